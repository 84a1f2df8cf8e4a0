"""Multi-GPU sharding of the frontend path: frames are sharded contiguously over the
ranks (rank r owns frames [r*B, (r+1)*B) of every global batch), extraction needs no communication,
and the ONE exchange step is the boundary descriptor block: the match job (t, t-1) with t = r*B needs
the descriptors of frame r*B - 1, which the previous rank produced.  One all_gather of the fixed-size
block {n, descriptors[cap x 32]} per step (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import os
import torch
import torch.distributed as dist


def shard_range(world: int, rank: int, frames_per_rank: int):
    """global frame indices owned by `rank` in one global batch"""
    return range(rank * frames_per_rank, (rank + 1) * frames_per_rank)


def pack_boundary(desc_last: torch.Tensor, n_last: torch.Tensor) -> torch.Tensor:
    """desc_last: [cap, 32] uint8, n_last: int32 scalar tensor -> one flat uint8 block (4 + cap*32 bytes)"""
    return torch.cat([n_last.reshape(1).to(torch.int32).view(torch.uint8), desc_last.reshape(-1)])


def unpack_boundary(block: torch.Tensor, cap: int):
    n = block[:4].view(torch.int32)[0]
    return block[4:4 + cap * 32].view(cap, 32), n


def exchange_boundary(desc_last: torch.Tensor, n_last: torch.Tensor, cap: int, group=None):
    """all_gather every rank's last-frame block; return (desc, n) of the PREVIOUS rank (the last rank's
    block of the previous global batch wraps around to rank 0, like a streaming sequence would)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = pack_boundary(desc_last, n_last)
    if world == 1 and not (dist.is_initialized() and os.environ.get("DVS_FORCE_COLLECTIVE") == "1"):
        return unpack_boundary(mine, cap)
    out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    prev = (rank - 1) % world
    return unpack_boundary(out[prev * mine.numel():(prev + 1) * mine.numel()], cap)
