"""The streaming extract + match step that bench.py times and tests/test_gpu_pipeline.py checks against the oracle: ONE statement
of the schedule, shared by both.

Step i of a pipeline extracts batch i (B frames resident in HBM: pyramid -> FAST -> quad-tree -> blur -> orientation + rBRIEF,
reference ORBextractor.cpp:1086-1167) and matches batch i - 1 (B jobs: frame t against frame t - 1, frontend.cpp:1123), software
pipelined over two streams:

  * the extractor's main stream runs FAST and the quad-tree; its prefetch stream builds the NEXT batch's pyramid beside FAST
    (`hint_next_batch_device`); its auxiliary stream runs the blur and — deferred — the descriptor stage beside the next FAST;
  * the match of batch i - 1 runs on the match stream, released by the extractor behind batch i's FAST (`set_after_fast_event`) so
    that the matrix-core match runs beside the quad-tree / blur phase;
  * `nsets` output sets rotate: step i writes set i % nsets; its last reader is the match of batch i + 1 (frame 0 of batch i + 1
    against the last frame of batch i), handed to the extractor as the reuse guard of step i + nsets.  That match is enqueued in
    step i + 2, so the pipelined schedule needs nsets >= 3: with two sets step i + 2 would overwrite the set the match enqueued
    BEHIND it still reads (no event of that match exists yet when the extraction is enqueued) — refused in the constructor.

With a communicator (frames sharded contiguously over ranks) the frame before this rank's first frame comes
from `comm.exchange_boundary` (csrc/comm.hip: one all-gather of every rank's last frame per global batch) instead of the previous
batch's last frame.  `pipelined=False` is the plain schedule: every batch's match behind its own extraction on one stream.

Pure ctypes: no torch in here."""
import numpy as np

from . import _lib
from ._lib import KP_DTYPE
from .matcher import BFMatcher
from .orb import ORBextractor


class StreamingPipeline:
    def __init__(self, B, rows, cols, nfeatures=2000, device=0, nsets=4, pipelined=True, params=(1.2, 8, 20, 7)):
        if pipelined and nsets < 3:
            raise ValueError("the pipelined schedule rotates at least 3 output sets (the match of batch i + 1 reads batch i's last frame "
                             "and is enqueued in step i + 2)")
        self.L = _lib.lib()
        self.B, self.rows, self.cols, self.device, self.nsets, self.pipelined = B, rows, cols, device, nsets, pipelined
        self.orb = ORBextractor(nfeatures, *params, device=device, max_batch=B)
        self.cap = cap = self.orb.capacity
        # streams are created only when used, back to back and before any communicator comes up: every HIP stream is a hardware queue
        # (an idle fourth stream in the extractor handle cost 0.2 ms per step; RCCL initialised first moved the same job between
        # 48 k and 72 k frames/s depending on GPU_MAX_HW_QUEUES)
        self.T = self.orb.get_stream()
        self.M = _lib.stream_create(device) if pipelined else self.T      # match (and boundary exchange: the match is its only consumer)
        self.mat = BFMatcher(device=device, stream=self.M)
        mk = lambda n: [_lib.DeviceBuffer(n, device) for _ in range(nsets)]
        self.kps, self.desc, self.n = mk(B * cap * 28), mk(B * cap * 32), mk(B * 4)
        self.idx, self.dist = mk(B * cap * 4), mk(B * cap * 4)
        for b in self.desc + self.n:
            _lib.check(self.L.dvs_memset(device, b.ptr, 0, b.nbytes))
        self.ev_ext = [_lib.event_create(device) for _ in range(nsets)]     # batch's outputs complete (recorded by the library)
        self.ev_match = [_lib.event_create(device) for _ in range(nsets)]   # batch's match complete
        self.ev_fast = _lib.event_create(device)
        self.comm = None
        self.i = 0
        if pipelined:
            self.orb.set_after_fast_event(self.ev_fast)
        self.orb.set_output_event(self.ev_ext[0], defer=pipelined)

    def attach_comm(self, comm):
        """frame-sharded run: `comm.exchange_boundary(stream, d_desc_last, d_n_last, cap)` (dist.Comm / dist.LoopbackComm)"""
        self.comm = comm

    def _last(self, s):
        return self.desc[s].ptr + (self.B - 1) * self.cap * 32, self.n[s].ptr + (self.B - 1) * 4

    def _match(self, j, behind_fast=False):
        """enqueue the B match jobs of batch j on the match stream (its extraction is ordered by events)"""
        L, B, cap, M = self.L, self.B, self.cap, self.M
        sj = j % self.nsets
        prev_desc = prev_n = 0
        if self.pipelined:
            L.dvs_stream_wait_event(M, self.ev_ext[sj])
        if self.comm is not None:
            # the one exchange step, once per global batch: every rank's LAST frame of batch j; this rank's first frame is matched against
            # the frame before it in the global order — the previous rank's last frame of the same batch, or (rank 0) the last rank's of
            # the batch before.  Depends only on batch j's extraction; shares the match stream.
            prev_desc, prev_n = self.comm.exchange_boundary(M, *self._last(sj), cap)
        elif j > 0:
            prev_desc, prev_n = self._last((j - 1) % self.nsets)        # one GPU: the previous batch's last frame, read in place
        if behind_fast:
            L.dvs_stream_wait_event(M, self.ev_fast)                      # released behind the FAST of the step just enqueued
        self.mat.match_sequence_device(self.desc[sj].ptr, self.n[sj].ptr, cap, B, prev_desc, prev_n, self.idx[sj].ptr, self.dist[sj].ptr)
        L.dvs_event_record(self.ev_match[sj], M)

    def step(self, d_img, d_next=0, match=True):
        """extraction of the batch at device pointer `d_img` (B frames, tight rows) + the match of the previous step's batch;
        `d_next`: the batch the NEXT step will extract (its pyramid is built ahead), 0 if unknown"""
        i = self.i
        self.i += 1
        s = i % self.nsets
        B, rows, cols, cap = self.B, self.rows, self.cols, self.cap
        if self.pipelined:
            if i >= self.nsets:
                self.orb.set_reuse_guard_event(self.ev_match[(i - self.nsets + 1) % self.nsets])   # the last reader of the set this step overwrites
            if d_next:
                self.orb.hint_next_batch_device(d_next)
            self.orb.set_output_event(self.ev_ext[s])
        self.orb.extract_batch_device(d_img, B, rows, cols, cols, rows * cols, self.kps[s].ptr, self.desc[s].ptr, cap, self.n[s].ptr)
        if not match:
            return
        if self.pipelined:
            if i >= 1:
                self._match(i - 1, behind_fast=True)
        else:
            self._match(i)

    def flush(self):
        """the match of the last extracted batch (the pipelined schedule runs it one step late)"""
        if self.pipelined and self.i >= 1:
            self._match(self.i - 1)

    def synchronize(self):
        self.orb.synchronize()
        _lib.stream_synchronize(self.M)

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
        self.synchronize()
        self.orb.set_output_event(0, defer=False)
        self.orb.set_after_fast_event(0)
        for e in self.ev_ext + self.ev_match + [self.ev_fast]:
            self.L.dvs_event_destroy(e)
        self.mat.close()
        if self.M != self.T:
            _lib.stream_destroy(self.M)
        self.orb.close()
