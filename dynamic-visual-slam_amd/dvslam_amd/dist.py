"""Multi-GPU sharding of the frontend path: frames are sharded contiguously over the
ranks (rank r owns frames [r*B, (r+1)*B) of every global batch), extraction needs no communication,
and the ONE exchange step is the boundary descriptor block: the match job (t, t-1) with t = r*B needs
the descriptors of frame r*B - 1, which the previous rank produced as the last frame of the SAME batch (rank 0:
the last rank's last frame of the batch before).  One all-gather of the fixed-size block {descriptors[cap x 32], n}
per global batch; the result for rank 0 comes from the previous call's gather.

Two implementations of the same block layout:
* `Comm` — the product path: dvs_comm_* / dvs_exchange_boundary of the C-ABI (RCCL ncclAllGather over xGMI on the
  caller's HIP stream, gather buffers owned by the communicator, nothing allocated per step).  This is what a C++ host
  calls and what bench.py runs on GPUs.
* `BoundaryExchanger` / `exchange_boundary` — the torch.distributed test double (gloo in the CPU tests), with the block
  and the gather buffer preallocated once."""
import ctypes as C
import os
import numpy as np
import torch
import torch.distributed as dist

ID_BYTES = 128


def shard_range(world: int, rank: int, frames_per_rank: int):
    """global frame indices owned by `rank` in one global batch"""
    return range(rank * frames_per_rank, (rank + 1) * frames_per_rank)


def level_shards(level_pixels, world: int):
    """SURVEY.md §8e "Partitioning", small batches: pyramid levels -> ranks, balanced by pixel count (longest-processing-time
    greedy: levels in decreasing size, each to the least loaded rank; ties -> lowest rank).  Returns one bit mask of levels per
    rank; every level is owned by exactly one rank.  720p, 8 ranks: {L0} {L1} {L2} {L3} {L4} {L5} {L6} {L7}; 4 ranks:
    {L0} {L1} {L2,L7} {L3,L4,L5,L6} by the greedy rule."""
    load = [0] * world
    masks = [0] * world
    for l in sorted(range(len(level_pixels)), key=lambda i: (-level_pixels[i], i)):
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += level_pixels[l]
        masks[r] |= 1 << l
    return masks


def _block_bytes(cap: int) -> int:
    """descriptors first (so every rank's descriptor rows start 64-byte aligned inside the gathered buffer), then n, padded;
    = dvs_boundary_block_bytes(cap)"""
    return (cap * 32 + 4 + 63) // 64 * 64


def pack_boundary(desc_last: torch.Tensor, n_last: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """desc_last: [cap, 32] uint8, n_last: int32 scalar / 1-element tensor -> one flat uint8 block (into `out` if given)"""
    cap = desc_last.shape[0]
    block = out if out is not None else torch.zeros(_block_bytes(cap), dtype=torch.uint8, device=desc_last.device)
    block[:cap * 32].copy_(desc_last.reshape(-1))
    block[cap * 32:cap * 32 + 4].copy_(n_last.reshape(1).to(torch.int32).view(torch.uint8))
    return block


def unpack_boundary(block: torch.Tensor, cap: int):
    """views into the block: (desc [cap, 32] uint8, n int32 0-dim) — usable in place by the matcher (data_ptr)"""
    n = block[cap * 32:cap * 32 + 4].view(torch.int32)[0]
    return block[:cap * 32].view(cap, 32), n


class BoundaryExchanger:
    """torch.distributed form of the exchange step with everything preallocated: the gather buffer [world][block] three times (a
    call's result points into this call's and the previous call's buffer and stays valid while the next call gathers into the
    third); this rank packs straight into its slot.  Same contract as dvs_exchange_boundary: called once per global batch with
    this rank's last frame, returns the predecessor of this rank's FIRST frame — rank r >= 1: rank r - 1's block of this call;
    rank 0: the last rank's block of the previous call, (None, None) on the first call."""

    def __init__(self, cap: int, device, group=None):
        self.cap, self.group = cap, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.blk = _block_bytes(cap)
        self.bufs = [torch.zeros(self.world * self.blk, dtype=torch.uint8, device=device) for _ in range(3)]
        self.turn = 0
        self.calls = 0

    def __call__(self, desc_last: torch.Tensor, n_last: torch.Tensor):
        out = self.bufs[self.turn]
        prev_out = self.bufs[(self.turn + 2) % 3] if self.calls > 0 else None
        self.turn = (self.turn + 1) % 3
        self.calls += 1
        mine = out[self.rank * self.blk:(self.rank + 1) * self.blk]
        pack_boundary(desc_last, n_last, mine)
        if self.world > 1:
            dist.all_gather_into_tensor(out, mine, group=self.group)
        if self.rank > 0:
            return unpack_boundary(out[(self.rank - 1) * self.blk:self.rank * self.blk], self.cap)
        if prev_out is None:
            return None, None
        return unpack_boundary(prev_out[(self.world - 1) * self.blk:self.world * self.blk], self.cap)


_exchangers = {}


def exchange_boundary(desc_last: torch.Tensor, n_last: torch.Tensor, cap: int, group=None):
    """one call per global batch with this rank's last frame; returns (desc, n) of the frame BEFORE this rank's first frame of the
    batch in the global order (see BoundaryExchanger), (None, None) where the sequence starts"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    key = (cap, str(desc_last.device), id(group), world)
    if key not in _exchangers:
        _exchangers[key] = BoundaryExchanger(cap, desc_last.device, group)
    return _exchangers[key](desc_last, n_last)


class Comm:
    """dvs_comm of the C-ABI: RCCL communicator + the boundary exchange on a raw HIP stream.  `bcast_id(id_bytes_or_None)`
    is the caller's out-of-band broadcast of the 128-byte unique id from rank 0 (bench.py: torch.distributed's store)."""

    @classmethod
    def loopback(cls, device: int, world: int):
        """`world` logical ranks of this process on one device (dvs_comm_create_loopback): a list of Comm, one per rank, each to be
        driven by its own thread — a collective call blocks until every rank of the group has made it"""
        from ._lib import lib, check
        hs = (C.c_void_p * world)()
        check(lib().dvs_comm_create_loopback(device, world, hs))
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c._L, c._check, c.h, c.rank, c.world = lib(), check, C.c_void_p(hs[r]), r, world
            out.append(c)
        return out

    def __init__(self, device: int, rank: int, world: int, bcast_id):
        from ._lib import lib, check
        self._L, self._check = lib(), check
        ident = None
        if rank == 0:
            buf = (C.c_uint8 * ID_BYTES)()
            check(self._L.dvs_comm_get_unique_id(buf))
            ident = bytes(buf)
        ident = bcast_id(ident)
        assert isinstance(ident, (bytes, bytearray)) and len(ident) == ID_BYTES
        h = C.c_void_p()
        check(self._L.dvs_comm_create(device, rank, world, (C.c_uint8 * ID_BYTES).from_buffer_copy(ident), C.byref(h)))
        self.h, self.rank, self.world = h, rank, world

    @property
    def rccl_version(self):
        return int(self._L.dvs_comm_rccl_version())

    def exchange_boundary(self, stream: int, d_desc_last: int, d_n_last: int, cap: int):
        """one call per global batch with this rank's last frame -> (device pointer of the descriptors, of the count) of the frame
        before this rank's first frame in the global order, (0, 0) where the sequence starts; asynchronous on `stream`"""
        pd, pn = C.c_void_p(), C.c_void_p()
        self._check(self._L.dvs_exchange_boundary(self.h, stream, d_desc_last, d_n_last, cap, C.byref(pd), C.byref(pn)))
        return pd.value or 0, pn.value or 0

    def all_gather(self, stream: int, d_send: int, d_recv: int, nbytes: int):
        self._check(self._L.dvs_comm_all_gather(self.h, stream, d_send, d_recv, nbytes))

    def close(self):
        if self.h:
            self._L.dvs_comm_destroy(self.h)
            self.h = None


class LevelShardedExtractor:
    """SURVEY.md §8e, batches of fewer than 8 frames: every rank holds the same level-0 frames, extracts only its own pyramid
    levels (dvs_orb_extract_levels_device, the resize chain rebuilt up to its top level), one in-place ncclAllGather of the
    level-slotted blocks, then dvs_orb_merge_levels_device restores the reference's level-major order on every rank.  Results are
    bit-identical to extract_batch_device (tests/test_gpu_orb.py).  `comm` is a Comm (or None with world == 1)."""

    def __init__(self, orb, comm, rank: int, world: int, rows: int, cols: int, nimg: int):
        from ._lib import DeviceBuffer
        self.orb, self.comm, self.rank, self.world, self.nimg = orb, comm, rank, world, nimg
        px = [int(np.prod(orb.level_size(rows, cols, l))) for l in range(orb.nlevels)]
        self.masks = level_shards(px, world)
        self.owner = np.array([next(r for r in range(world) if self.masks[r] >> l & 1) for l in range(orb.nlevels)], np.int32)
        self.block_bytes = orb.level_block_bytes(nimg)
        self.gather = DeviceBuffer(world * self.block_bytes, device=orb.device)

    def extract(self, d_imgs, rows, cols, step, frame_stride, d_kps, d_desc, capacity, d_nout):
        """asynchronous on the extractor's stream (the gather is enqueued on the same stream: it depends on the extraction and the
        merge depends on it)"""
        mine = self.gather.ptr + self.rank * self.block_bytes
        self.orb.extract_levels_device(d_imgs, self.nimg, rows, cols, step, frame_stride, self.masks[self.rank], mine)
        if self.world > 1:
            self.comm.all_gather(self.orb.get_stream(), mine, self.gather.ptr, self.block_bytes)
        self.orb.merge_levels_device(self.gather.ptr, self.world, self.owner, self.nimg, d_kps, d_desc, capacity, d_nout)
