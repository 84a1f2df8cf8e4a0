"""Multi-GPU sharding of the frontend path: frames are sharded contiguously over the
ranks (rank r owns frames [r*B, (r+1)*B) of every global batch), extraction needs no communication,
and the ONE exchange step is the boundary descriptor block: the match job (t, t-1) with t = r*B needs
the descriptors of frame r*B - 1, which the previous rank produced as the last frame of the SAME batch (rank 0:
the last rank's last frame of the batch before).  One all-gather of the fixed-size block {descriptors[cap x 32], n}
per global batch; the result for rank 0 comes from the previous call's gather.

ONE implementation — dvs_exchange_boundary of the C-ABI (csrc/comm.hip) — behind two thin ctypes wrappers:
* `Comm` — RCCL ncclAllGather over xGMI on the caller's HIP stream (or the loopback group), gather buffers owned by the
  communicator, nothing allocated per step.  This is what a C++ host calls and what bench.py runs on GPUs.
* `HostComm` — the same C function over a host-transport communicator (dvs_comm_create_host): blocks in host memory, the
  all-gather is the caller's callback.  The CPU tests drive it from two OS processes with gloo as the transport."""
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


class HostComm:
    """dvs_comm_create_host of the C-ABI: the SAME exchange step (csrc/comm.hip: buffer rotation, this rank's slot, the predecessor with
    its wrap-around to the previous call) with the blocks in host memory and the caller's all-gather as the transport.  No device is
    touched, so two OS processes without a GPU run the product's rank logic (tests/test_adapters_and_dist.py: gloo as the transport).
    `all_gather(buf, rank, nbytes)`: buf is a uint8 numpy view of the whole receive buffer [world * nbytes]; this rank's block already
    sits at buf[rank * nbytes:(rank + 1) * nbytes]; fill in the others (in place)."""

    def __init__(self, rank: int, world: int, all_gather):
        from ._lib import lib, check
        self._L, self._check, self.rank, self.world = lib(), check, rank, world
        self.error = None

        def _cb(user, send, recv, nbytes):
            try:
                buf = np.ctypeslib.as_array((C.c_uint8 * (nbytes * world)).from_address(recv))
                assert send == recv + rank * nbytes, "in-place contract"
                all_gather(buf, rank, nbytes)
                return 0
            except Exception as e:   # noqa: BLE001 — reported through the C-ABI's status, kept for the caller
                self.error = e
                return 1
        self._cb = HOST_ALL_GATHER(_cb)   # (kept alive with the object)
        h = C.c_void_p()
        check(self._L.dvs_comm_create_host(rank, world, self._cb, None, C.byref(h)))
        self.h = h

    def exchange_boundary(self, desc_last: np.ndarray, n_last: int):
        """one call per global batch with this rank's LAST frame ([cap, 32] uint8, count) -> (descriptors [cap, 32], count) of the frame
        before this rank's FIRST frame in the global order — views into the communicator's buffers, valid until the call after next —
        or (None, None) where the sequence starts"""
        desc_last = np.ascontiguousarray(desc_last, np.uint8)
        cap = desc_last.shape[0]
        n = np.array([n_last], np.int32)
        pd, pn = C.c_void_p(), C.c_void_p()
        self._check(self._L.dvs_exchange_boundary(self.h, None, desc_last.ctypes.data, n.ctypes.data, cap, C.byref(pd), C.byref(pn)))
        if not pd.value:
            return None, None
        d = np.ctypeslib.as_array((C.c_uint8 * (cap * 32)).from_address(pd.value)).reshape(cap, 32)
        return d, int(C.c_int32.from_address(pn.value).value)

    def reset_sequence(self):
        self._check(self._L.dvs_comm_reset_sequence(self.h))

    def close(self):
        if self.h:
            self._L.dvs_comm_destroy(self.h)
            self.h = None


HOST_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


def gloo_all_gather(group=None):
    """transport of HostComm over a torch.distributed group (gloo in the CPU tests)"""
    def fn(buf, rank, nbytes):
        t = torch.from_numpy(buf)
        dist.all_gather_into_tensor(t, t[rank * nbytes:(rank + 1) * nbytes].clone(), group=group)
    return fn


class Comm:
    """dvs_comm of the C-ABI: RCCL communicator + the boundary exchange on a raw HIP stream.  `bcast_id(id_bytes_or_None)`
    is the caller's out-of-band broadcast of the 128-byte unique id from rank 0 (bench.py: torch.distributed's store)."""

    @classmethod
    def loopback(cls, device: int, world: int, hooks: bool = False):
        """`world` logical ranks of this process on one device (dvs_comm_create_loopback): a list of Comm, one per rank, each to be
        driven by its own thread — a collective call blocks until every rank of the group has made it"""
        from ._lib import lib, test_lib, check
        L = test_lib() if hooks else lib()   # (the library of the pipelines the ranks will be attached to)
        hs = (C.c_void_p * world)()
        check(L.dvs_comm_create_loopback(device, world, hs))
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c._L, c._check, c.h, c.rank, c.world = L, check, C.c_void_p(hs[r]), r, world
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
