import ctypes as C
import numpy as np
from ._lib import lib, test_lib, check, ptr


class BFMatcher:
    """Python mirror of cv::BFMatcher(cv::NORM_HAMMING, crossCheck=false) as the reference uses it
    (frontend.cpp:220,614,1123; backend.cpp:222,1072).  match(query, train) returns one
    (queryIdx=i, trainIdx, distance) per query row as two int32 arrays (trainIdx, distance)."""

    def __init__(self, device=0, stream=None, hooks=False):
        """stream: raw hipStream_t (int) the matcher enqueues on from the start; None = a stream of its own.  hooks=True: all calls through
        lib/libdvslam_hip_test.so (an object shared with a hooks=True extractor or pipeline must live in the same library)"""
        self._L = test_lib() if hooks else lib()
        h = C.c_void_p()
        if stream is None:
            check(self._L.dvs_matcher_create(device, C.byref(h)))
        else:
            check(self._L.dvs_matcher_create_on_stream(device, stream, C.byref(h)))
        self._h = h

    @classmethod
    def from_handle(cls, handle, L=None):
        """non-owning view of a dvs_matcher* that lives inside another handle (dvs_pipeline_matcher: test library only)"""
        m = cls.__new__(cls)
        m._L, m._h, m._owned = (L or lib()), C.c_void_p(handle), False
        return m

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._L.dvs_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def match(self, query, train):
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        idx = np.zeros(len(q), np.int32); dist = np.zeros(len(q), np.int32)
        if len(t) == 0:  # cv: empty train -> empty result
            return np.zeros(0, np.int32), np.zeros(0, np.int32)
        check(self._L.dvs_match_hamming(self._h, ptr(q), len(q), ptr(t), len(t), ptr(idx), ptr(dist)))
        return idx, dist

    def match_thresh(self, query, train, max_dist, cap=None):
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        cap = cap if cap is not None else max(len(q) * len(t), 1)
        pairs = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        check(self._L.dvs_match_hamming_thresh(self._h, ptr(q), len(q), ptr(t), len(t), max_dist, ptr(pairs), cap, C.byref(n)))
        return n.value, pairs[:min(n.value, cap)].copy()

    def match_batch_device(self, d_q, d_nq, q_stride_rows, d_t, d_nt, t_stride_rows, npairs, d_idx, d_dist):
        check(self._L.dvs_match_hamming_batch_device(self._h, d_q, d_nq, q_stride_rows, d_t, d_nt, t_stride_rows, npairs, d_idx, d_dist))

    def match_sequence_device(self, d_desc, d_n, stride_rows, nframes, d_prev_desc, d_prev_n, d_idx, d_dist):
        """frame p vs frame p-1 of a device-resident run; frame 0 vs (d_prev_desc, d_prev_n) or nothing (0, 0)"""
        check(self._L.dvs_match_hamming_sequence_device(self._h, d_desc, d_n, stride_rows, nframes, d_prev_desc or None,
                                                        d_prev_n or None, d_idx, d_dist))

    def set_stream(self, stream_ptr):
        check(self._L.dvs_matcher_set_stream(self._h, stream_ptr))

    def synchronize(self):
        check(self._L.dvs_matcher_synchronize(self._h))
