import ctypes as C
import numpy as np
from ._lib import lib, check, ptr, KP_DTYPE


class FrontendGlue:
    """ctypes driver of the N1 / N2 entry points (include/dvslam_hip.h, 'glue either side of the path')."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        check(self._L.dvs_matcher_create(device, C.byref(h)))
        self._h = h
        L, vp, i32, sz, f32, dbl = self._L, C.c_void_p, C.c_int32, C.c_size_t, C.c_float, C.c_double
        L.dvs_bgr_to_gray.argtypes = [vp, vp, i32, i32, sz, vp, sz, i32]
        L.dvs_filter_depth.argtypes = [vp, vp, vp, i32, vp, i32, i32, sz, f32, f32, vp, vp, vp, C.POINTER(i32)]
        L.dvs_filter_matches.argtypes = [vp, vp, vp, i32, f32, vp, C.POINTER(i32)]
        L.dvs_backproject.argtypes = [vp, vp, i32, vp, i32, i32, sz, f32, f32, f32, f32, vp, vp, vp, vp, C.POINTER(i32)]
        L.dvs_associate.argtypes = [vp, vp, vp, i32, vp, vp, i32, vp, vp, dbl, dbl, dbl, dbl, dbl, dbl, vp]

    def close(self):
        if getattr(self, "_h", None):
            self._L.dvs_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bgr_to_gray(self, bgr, variant=0):
        bgr = np.asarray(bgr); rows, cols, _ = bgr.shape
        assert bgr.dtype == np.uint8 and bgr.strides[2] == 1 and bgr.strides[1] == 3
        gray = np.zeros((rows, cols), np.uint8)
        check(self._L.dvs_bgr_to_gray(self._h, ptr(bgr), rows, cols, bgr.strides[0], ptr(gray), cols, variant))
        return gray

    def filter_depth(self, kps, desc, depth, min_depth=0.3, max_depth=3.0):
        kps = np.ascontiguousarray(kps, KP_DTYPE); n = len(kps)
        desc = np.ascontiguousarray(desc, np.uint8) if desc is not None else None
        depth = np.asarray(depth); assert depth.dtype == np.uint16 and depth.strides[1] == 2
        ok = np.zeros(n, KP_DTYPE); od = np.zeros((n, 32), np.uint8); oi = np.zeros(n, np.int32); m = C.c_int32()
        check(self._L.dvs_filter_depth(self._h, ptr(kps), ptr(desc) if desc is not None else None, n, ptr(depth), depth.shape[0], depth.shape[1],
                                       depth.strides[0], min_depth, max_depth, ptr(ok), ptr(od), ptr(oi), C.byref(m)))
        return ok[:m.value], od[:m.value], oi[:m.value]

    def filter_matches(self, idx, dist, max_distance=50.0):
        idx = np.ascontiguousarray(idx, np.int32); dist = np.ascontiguousarray(dist, np.int32); n = len(idx)
        out = np.zeros((n, 3), np.int32); m = C.c_int32()
        check(self._L.dvs_filter_matches(self._h, ptr(idx), ptr(dist), n, max_distance, ptr(out), C.byref(m)))
        return out[:m.value]

    def backproject(self, kps, depth, fx, fy, cx, cy, R, t):
        kps = np.ascontiguousarray(kps, KP_DTYPE); n = len(kps)
        depth = np.asarray(depth); R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64).reshape(3)
        w = np.zeros((n, 3), np.float64); oi = np.zeros(n, np.int32); m = C.c_int32()
        check(self._L.dvs_backproject(self._h, ptr(kps), n, ptr(depth), depth.shape[0], depth.shape[1], depth.strides[0], fx, fy, cx, cy,
                                      ptr(R), ptr(t), ptr(w), ptr(oi), C.byref(m)))
        return w[:m.value], oi[:m.value]

    def associate(self, obs_desc, obs_px, lm_desc, lm_xyz, R, t, fx, fy, cx, cy, max_desc=50.0, max_reproj=5.0):
        obs_desc = np.ascontiguousarray(obs_desc, np.uint8).reshape(-1, 32); obs_px = np.ascontiguousarray(obs_px, np.float32).reshape(-1, 2)
        lm_desc = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32); lm_xyz = np.ascontiguousarray(lm_xyz, np.float32).reshape(-1, 3)
        R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64).reshape(3)
        best = np.full(len(obs_desc), -1, np.int32)
        check(self._L.dvs_associate(self._h, ptr(obs_desc), ptr(obs_px), len(obs_desc), ptr(lm_desc), ptr(lm_xyz), len(lm_desc), ptr(R), ptr(t),
                                    fx, fy, cx, cy, max_desc, max_reproj, ptr(best)))
        return best
