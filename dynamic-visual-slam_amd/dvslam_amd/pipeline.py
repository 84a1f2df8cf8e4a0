"""The streaming extract + match step that bench.py times and tests/test_gpu_pipeline.py checks against the oracle — a thin ctypes
caller of the C-ABI's `dvs_pipeline_*` (csrc/pipeline.hip), which owns the schedule: ONE statement of it, in C++, shared by bench,
the Python tests and the C++ host program tests/cpp/pipeline_stream.cpp (include/dvslam/streaming_pipeline.hpp).

Step i extracts batch i (B frames resident in HBM: pyramid -> FAST -> quad-tree -> blur -> orientation + rBRIEF, reference
ORBextractor.cpp:1086-1167) and matches batch i - 1 (B jobs: frame t against frame t - 1, frontend.cpp:1123); `nsets` output sets
rotate (>= 3 when pipelined — refused by dvs_pipeline_create otherwise).  With a communicator (frames sharded contiguously over
ranks) the frame before this rank's first frame comes from `dvs_exchange_boundary`.  `pipelined=False` is the
plain schedule: every batch's match behind its own extraction on one stream.  `lanes`: 0 = by batch size, 1 = the two-stream software
pipeline, 2..4 = the small-batch lane schedule (whole steps in flight on independent extractor / matcher pairs).  `quadtree_async`:
1 / -1 / 0 = the four-stream form of the two-stream pipeline on / off / by batch size (7..24 frames per step).

Pure ctypes: no torch in here."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE
from .matcher import BFMatcher
from .orb import ORBextractor


class _Ptr:
    """a device pointer inside the pipeline's output arena, with DeviceBuffer's download()"""

    def __init__(self, ptr, nbytes, device, L):
        self.ptr, self.nbytes, self.device, self.L = int(ptr or 0), int(nbytes), device, L

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _lib.check(self.L.dvs_memcpy_d2h(self.device, _lib.ptr(out), self.ptr, out.nbytes))
        return out


class StreamingPipeline:
    def __init__(self, B, rows, cols, nfeatures=2000, device=0, nsets=0, pipelined=True, params=(1.2, 8, 20, 7), lanes=0, quadtree_async=0,
                 hooks=False):
        # hooks=True: through lib/libdvslam_hip_test.so, which also exports the handles inside (`orb`, `mat`, the streams `T`, `M`) for the
        # tests that look at single stages; the product library exposes the step, its results and the measurement calls only
        self.L = L = _lib.test_lib() if hooks else _lib.lib()
        self.hooks = bool(hooks)
        self.B, self.rows, self.cols, self.device, self.nsets, self.pipelined = B, rows, cols, device, nsets, pipelined
        prm = _lib.PipelineParams(_lib.OrbParams(nfeatures, params[0], params[1], params[2], params[3], (C.c_int32 * 7)(*([0] * 7)), B),
                                  B, rows, cols, nsets, int(bool(pipelined)), lanes, quadtree_async)
        h = C.c_void_p()
        code = L.dvs_pipeline_create(C.byref(prm), device, C.byref(h))
        if code == -6 and pipelined and nsets < 3:
            raise ValueError(L.dvs_last_error().decode(errors="replace"))
        _lib.check(code)
        self._h = h
        self.nsets = nsets = int(L.dvs_pipeline_nsets(h))   # nsets = 0 asks for the schedule's default
        self.quadtree_async = bool(L.dvs_pipeline_quadtree_async(h))   # the four-stream form (dvs_pipeline_params::quadtree_async)
        self.lanes = int(L.dvs_pipeline_lanes(h))   # 0: serial, 1: two-stream software pipeline, >= 2: lane schedule
        st0 = _lib.PipelineSet()
        _lib.check(L.dvs_pipeline_get_set(h, 0, C.byref(st0)))
        self.cap = cap = int(st0.capacity)
        if hooks:   # non-owning views of the handles inside
            self.orb = ORBextractor.from_handle(L.dvs_pipeline_extractor(h), nfeatures, params[1], params[0], device, B, L=L)
            self.mat = BFMatcher.from_handle(L.dvs_pipeline_matcher(h), L=L)
            self.T = self.orb.get_stream()
            self.M = int(L.dvs_pipeline_match_stream(h) or 0)
        self.kps, self.desc, self.n, self.idx, self.dist = [], [], [], [], []
        for s in range(nsets):
            st = _lib.PipelineSet()
            _lib.check(L.dvs_pipeline_get_set(h, s, C.byref(st)))
            self.kps.append(_Ptr(st.d_kps, B * cap * 28, device, L)); self.desc.append(_Ptr(st.d_desc, B * cap * 32, device, L))
            self.n.append(_Ptr(st.d_n, B * 4, device, L))
            self.idx.append(_Ptr(st.d_idx, B * cap * 4, device, L)); self.dist.append(_Ptr(st.d_dist, B * cap * 4, device, L))
        self.comm = None

    @property
    def i(self):
        return int(self.L.dvs_pipeline_steps(self._h))

    def reset(self):
        """synchronise and restart the sequence at step 0"""
        _lib.check(self.L.dvs_pipeline_reset(self._h))

    # ---- measurement (dvs_pipeline_set_serialized / _stage_timing / _get_stage_times: bench.py's per-stage report) ----
    def set_serialized(self, on):
        """True: every kernel of an extraction alone on the main stream (per-kernel durations); False: the shipped schedule"""
        _lib.check(self.L.dvs_pipeline_set_serialized(self._h, int(bool(on))))

    def stage_timing(self, on=True):
        _lib.check(self.L.dvs_pipeline_stage_timing(self._h, int(bool(on))))

    def stage_times(self, reset=True):
        """({stage: ms}, {stage: launch sequences}) accumulated since the last reset"""
        from .orb import STAGES
        ms = np.zeros(len(STAGES), np.float64); calls = np.zeros(len(STAGES), np.int64)
        _lib.check(self.L.dvs_pipeline_get_stage_times(self._h, _lib.ptr(ms), _lib.ptr(calls), int(reset)))
        return {k: float(ms[i]) for i, k in enumerate(STAGES)}, {k: int(calls[i]) for i, k in enumerate(STAGES)}

    def attach_comm(self, comm):
        """frame-sharded run: `comm` is a dist.Comm (RCCL, or one rank of `dist.Comm.loopback`) made by the SAME library as this pipeline"""
        assert comm is None or comm._L is self.L, "communicator and pipeline must come from one library (hooks=... on both)"
        self.comm = comm
        _lib.check(self.L.dvs_pipeline_attach_comm(self._h, comm.h if comm is not None else None))

    def _last(self, s):
        return self.desc[s].ptr + (self.B - 1) * self.cap * 32, self.n[s].ptr + (self.B - 1) * 4

    def step(self, d_img, d_next=0, match=True):
        """extraction of the batch at device pointer `d_img` (B frames, tight rows) + the match of the previous step's batch;
        `d_next`: the batch the NEXT step will extract (its pyramid is built ahead), 0 if unknown"""
        _lib.check(self.L.dvs_pipeline_step(self._h, d_img, d_next or None, 0 if match else 1))

    def flush(self):
        """the match of the last extracted batch (the pipelined schedule runs it one step late)"""
        _lib.check(self.L.dvs_pipeline_flush(self._h))

    def synchronize(self):
        _lib.check(self.L.dvs_pipeline_synchronize(self._h))

    # ---- results (after synchronize) ----
    def outputs(self, i):
        """(n[B], keypoints[B][cap], descriptors[B][cap][32]) of batch i — must still be resident (i > last step - nsets)"""
        assert self.i - self.nsets <= i < self.i
        s = i % self.nsets
        return (self.n[s].download(np.int32, self.B), self.kps[s].download(np.uint8, self.B * self.cap * 28).view(KP_DTYPE).reshape(self.B, self.cap),
                self.desc[s].download(np.uint8, self.B * self.cap * 32).reshape(self.B, self.cap, 32))

    def matches(self, j):
        """(trainIdx[B][cap], distance[B][cap]) of batch j's match jobs (row f = frame f against frame f - 1)"""
        s = j % self.nsets
        return (self.idx[s].download(np.int32, self.B * self.cap).reshape(self.B, self.cap),
                self.dist[s].download(np.int32, self.B * self.cap).reshape(self.B, self.cap))

    def close(self):
        if getattr(self, "_h", None):
            self.L.dvs_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
