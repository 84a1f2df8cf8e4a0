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


def _block_bytes(cap: int) -> int:
    """descriptors first (so every rank's descriptor rows start 64-byte aligned inside the gathered buffer), then n, padded"""
    return (cap * 32 + 4 + 63) // 64 * 64


def pack_boundary(desc_last: torch.Tensor, n_last: torch.Tensor) -> torch.Tensor:
    """desc_last: [cap, 32] uint8, n_last: int32 scalar / 1-element tensor -> one flat uint8 block"""
    cap = desc_last.shape[0]
    block = torch.zeros(_block_bytes(cap), dtype=torch.uint8, device=desc_last.device)
    block[:cap * 32].copy_(desc_last.reshape(-1))
    block[cap * 32:cap * 32 + 4].copy_(n_last.reshape(1).to(torch.int32).view(torch.uint8))
    return block


def unpack_boundary(block: torch.Tensor, cap: int):
    """views into the block: (desc [cap, 32] uint8, n int32 0-dim) — usable in place by the matcher (data_ptr)"""
    n = block[cap * 32:cap * 32 + 4].view(torch.int32)[0]
    return block[:cap * 32].view(cap, 32), n


def exchange_boundary(desc_last: torch.Tensor, n_last: torch.Tensor, cap: int, group=None):
    """all_gather every rank's last-frame block; return (desc, n) of the PREVIOUS rank (the last rank's
    block of the previous global batch wraps around to rank 0, like a streaming sequence would)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1 and not (dist.is_initialized() and os.environ.get("DVS_FORCE_COLLECTIVE") == "1"):
        return desc_last, n_last.reshape(-1)[0]   # one rank: the predecessor is the caller's own last frame, in place
    mine = pack_boundary(desc_last, n_last)
    out = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    prev = (rank - 1) % world
    return unpack_boundary(out[prev * mine.numel():(prev + 1) * mine.numel()], cap)
