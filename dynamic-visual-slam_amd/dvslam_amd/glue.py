import ctypes as C
import numpy as np
from ._lib import lib, check, ptr, KP_DTYPE, KeyframeHeader


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
        L.dvs_harris_responses.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, i32, f32, vp]
        L.dvs_keyframe_cdr_capacity.argtypes = [C.c_char_p, i32]; L.dvs_keyframe_cdr_capacity.restype = sz
        L.dvs_publish_keyframe.argtypes = [vp, C.POINTER(KeyframeHeader), vp, vp, i32, vp, i32, i32, sz, f32, f32, f32, f32, vp, vp, vp, sz,
                                           C.POINTER(sz), C.POINTER(i32)]

    def find_fundamental_ransac(self, pts1, pts2, threshold=2.0, confidence=0.99, max_iters=1000, seed=1):
        """cv::findFundamentalMat(pts1, pts2, mask, FM_RANSAC, threshold, confidence) -> (F 3x3, mask uint8[n], inliers of the model)"""
        p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
        F = np.zeros(9); mask = np.zeros(max(len(p1), 1), np.uint8); n = C.c_int32()
        check(self._L.dvs_find_fundamental_ransac(self._h, ptr(p1), ptr(p2), len(p1), threshold, confidence, max_iters, seed, ptr(F), ptr(mask),
                                                  C.byref(n)))
        return F.reshape(3, 3), mask[:len(p1)], n.value

    def find_fundamental_cv(self, pts1, pts2, threshold=2.0, confidence=0.99, max_iters=1000):
        """cv::findFundamentalMat(FM_RANSAC) with OpenCV's own sample sequence and 7-point solver (>= 15 points)
        -> (F 3x3, mask uint8[n], inliers of the model, iterations run)"""
        p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
        F = np.zeros(9); mask = np.zeros(max(len(p1), 1), np.uint8); n = C.c_int32(); it = C.c_int32()
        check(self._L.dvs_find_fundamental_cv(self._h, ptr(p1), ptr(p2), len(p1), threshold, confidence, max_iters, ptr(F), ptr(mask), C.byref(n), C.byref(it)))
        return F.reshape(3, 3), mask[:len(p1)], n.value, it.value

    def find_fundamental_cv_batch(self, pts1_list, pts2_list, threshold=2.0, confidence=0.99, max_iters=1000):
        """-> list of (mask uint8[n_b], inliers of the model, iterations run)"""
        nprob = len(pts1_list)
        off = np.zeros(nprob + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in pts1_list])
        cat = lambda ps: np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in ps]), np.float32)
        p1, p2 = cat(pts1_list), cat(pts2_list)
        mask = np.zeros(max(int(off[-1]), 1), np.uint8); nin = np.zeros(nprob, np.int32); its = np.zeros(nprob, np.int32)
        check(self._L.dvs_find_fundamental_cv_batch(self._h, nprob, ptr(off), ptr(p1), ptr(p2), threshold, confidence, max_iters, None, ptr(mask), ptr(nin), ptr(its)))
        return [(mask[off[b]:off[b + 1]].copy(), int(nin[b]), int(its[b])) for b in range(nprob)]

    def solve_pnp_ransac(self, pts3d, pts2d, K4, iterations=100, reproj_err=4.0, confidence=0.99, seed=1):
        """cv::solvePnPRansac(obj, img, K, noArray, rvec, tvec, false, iterations, reproj_err, confidence, inliers)
        -> (success, rvec, tvec, inlier indices)"""
        o = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3); i2 = np.ascontiguousarray(pts2d, np.float32).reshape(-1, 2)
        K = np.ascontiguousarray(K4, np.float64)
        rvec = np.zeros(3); tvec = np.zeros(3); inl = np.zeros(max(len(o), 1), np.int32); nin = C.c_int32(); ok = C.c_int32()
        check(self._L.dvs_solve_pnp_ransac(self._h, ptr(o), ptr(i2), len(o), ptr(K), iterations, reproj_err, confidence, seed, ptr(rvec), ptr(tvec),
                                           ptr(inl), C.byref(nin), C.byref(ok)))
        return bool(ok.value), rvec, tvec, inl[:nin.value].copy()

    def solve_pnp_ransac_cv(self, pts3d, pts2d, K4, iterations=100, reproj_err=4.0, confidence=0.99):
        """cv::solvePnPRansac as OpenCV 4.x runs it with its default flags (dvs_solve_pnp_ransac_cv): EPnP on cv::RNG's 5-point samples,
        float scoring, adaptive stop, solvePnP(ITERATIVE) refit -> (success, rvec, tvec, inlier indices, iterations run)"""
        return self.solve_pnp_ransac_cv_batch([pts3d], [pts2d], K4, iterations, reproj_err, confidence)[0]

    def solve_pnp_ransac_cv_batch(self, pts3d_list, pts2d_list, K4, iterations=100, reproj_err=4.0, confidence=0.99):
        nprob = len(pts3d_list)
        off = np.zeros(nprob + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in pts3d_list])
        o = (np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 3) for p in pts3d_list]), np.float32) if off[-1]
             else np.zeros((1, 3), np.float32))
        i2 = (np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in pts2d_list]), np.float32) if off[-1]
              else np.zeros((1, 2), np.float32))
        K = np.ascontiguousarray(K4, np.float64)
        rv = np.zeros((max(nprob, 1), 3)); tv = np.zeros((max(nprob, 1), 3)); inl = np.zeros(max(int(off[-1]), 1), np.int32)
        nin = np.zeros(max(nprob, 1), np.int32); ok = np.zeros(max(nprob, 1), np.int32); its = np.zeros(max(nprob, 1), np.int32)
        self._L.dvs_solve_pnp_ransac_cv_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        check(self._L.dvs_solve_pnp_ransac_cv_batch(self._h, nprob, ptr(off), ptr(o), ptr(i2), ptr(K), iterations, reproj_err, confidence, ptr(rv), ptr(tv),
                                                    ptr(inl), ptr(nin), ptr(ok), ptr(its)))
        return [(bool(ok[b]), rv[b].copy(), tv[b].copy(), inl[off[b]:off[b] + nin[b]].copy(), int(its[b])) for b in range(nprob)]

    def find_fundamental_ransac_batch(self, pts1_list, pts2_list, seeds, threshold=2.0, confidence=0.99, max_iters=1000):
        """many independent problems in one launch sequence -> list of (mask uint8[n_b], inliers of the model)"""
        nprob = len(pts1_list)
        off = np.zeros(nprob + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in pts1_list])
        cat = lambda ps: (np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in ps]), np.float32) if off[-1]
                          else np.zeros((1, 2), np.float32))
        p1, p2 = cat(pts1_list), cat(pts2_list)
        sd = np.ascontiguousarray(seeds, np.uint64)
        mask = np.zeros(max(int(off[-1]), 1), np.uint8); nin = np.zeros(max(nprob, 1), np.int32)
        self._L.dvs_find_fundamental_ransac_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int32,
                                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        check(self._L.dvs_find_fundamental_ransac_batch(self._h, nprob, ptr(off), ptr(p1), ptr(p2), threshold, confidence, max_iters, ptr(sd), None,
                                                        ptr(mask), ptr(nin)))
        return [(mask[off[b]:off[b + 1]].copy(), int(nin[b])) for b in range(nprob)]

    def solve_pnp_ransac_batch(self, pts3d_list, pts2d_list, K4, seeds, iterations=100, reproj_err=4.0, confidence=0.99):
        """-> list of (success, rvec, tvec, inlier indices)"""
        nprob = len(pts3d_list)
        off = np.zeros(nprob + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in pts3d_list])
        o = (np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 3) for p in pts3d_list]), np.float32) if off[-1]
             else np.zeros((1, 3), np.float32))
        i2 = (np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in pts2d_list]), np.float32) if off[-1]
              else np.zeros((1, 2), np.float32))
        K = np.ascontiguousarray(K4, np.float64); sd = np.ascontiguousarray(seeds, np.uint64)
        rv = np.zeros((max(nprob, 1), 3)); tv = np.zeros((max(nprob, 1), 3)); inl = np.zeros(max(int(off[-1]), 1), np.int32)
        nin = np.zeros(max(nprob, 1), np.int32); ok = np.zeros(max(nprob, 1), np.int32)
        self._L.dvs_solve_pnp_ransac_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        check(self._L.dvs_solve_pnp_ransac_batch(self._h, nprob, ptr(off), ptr(o), ptr(i2), ptr(K), iterations, reproj_err, confidence, ptr(sd), ptr(rv), ptr(tv),
                                                 ptr(inl), ptr(nin), ptr(ok)))
        return [(bool(ok[b]), rv[b].copy(), tv[b].copy(), inl[off[b]:off[b] + nin[b]].copy()) for b in range(nprob)]

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

    def harris_responses(self, img, xs, ys, block_size=7, k=0.04):
        """cv::ORB's HARRIS_SCORE measure at integer pixel positions of one image (pyramid layer)"""
        img = np.asarray(img); xs = np.ascontiguousarray(xs, np.int32); ys = np.ascontiguousarray(ys, np.int32)
        out = np.zeros(len(xs), np.float32)
        check(self._L.dvs_harris_responses(self._h, ptr(img), img.shape[0], img.shape[1], img.strides[0], ptr(xs), ptr(ys), len(xs), block_size, k, ptr(out)))
        return out

    @staticmethod
    def _header(stamp, frame_id, keyframe_id, t, q_xyzw):
        h = KeyframeHeader()
        h.stamp_sec, h.stamp_nanosec = int(stamp[0]), int(stamp[1])
        h.frame_id = frame_id.encode() if isinstance(frame_id, str) else frame_id
        h.keyframe_id = int(keyframe_id)
        for k in range(3):
            h.translation[k] = float(t[k])
        for k in range(4):
            h.rotation_xyzw[k] = float(q_xyzw[k])
        return h

    def publish_keyframe(self, kps, desc, depth, fx, fy, cx, cy, R, t, stamp=(0, 0), frame_id="camera_link", keyframe_id=0,
                         q_xyzw=(0.0, 0.0, 0.0, 1.0)):
        """publishKeyframe (frontend.cpp:699-776) as the Keyframe.msg CDR payload: returns (bytes, n_landmarks)"""
        kps = np.ascontiguousarray(kps, KP_DTYPE); n = len(kps)
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        depth = np.asarray(depth); R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64).reshape(3)
        hdr = self._header(stamp, frame_id, keyframe_id, t, q_xyzw)
        cap = self._L.dvs_keyframe_cdr_capacity(hdr.frame_id, n)
        out = np.zeros(cap, np.uint8); size = C.c_size_t(); m = C.c_int32()
        check(self._L.dvs_publish_keyframe(self._h, C.byref(hdr), ptr(kps), ptr(desc), n, ptr(depth), depth.shape[0], depth.shape[1],
                                           depth.strides[0], fx, fy, cx, cy, ptr(R), ptr(t), ptr(out), cap, C.byref(size), C.byref(m)))
        return out[:size.value].tobytes(), m.value

    def associate(self, obs_desc, obs_px, lm_desc, lm_xyz, R, t, fx, fy, cx, cy, max_desc=50.0, max_reproj=5.0):
        obs_desc = np.ascontiguousarray(obs_desc, np.uint8).reshape(-1, 32); obs_px = np.ascontiguousarray(obs_px, np.float32).reshape(-1, 2)
        lm_desc = np.ascontiguousarray(lm_desc, np.uint8).reshape(-1, 32); lm_xyz = np.ascontiguousarray(lm_xyz, np.float32).reshape(-1, 3)
        R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64).reshape(3)
        best = np.full(len(obs_desc), -1, np.int32)
        check(self._L.dvs_associate(self._h, ptr(obs_desc), ptr(obs_px), len(obs_desc), ptr(lm_desc), ptr(lm_xyz), len(lm_desc), ptr(R), ptr(t),
                                    fx, fy, cx, cy, max_desc, max_reproj, ptr(best)))
        return best


def unpack_keyframe(payload, cap_n=4096):
    """dvs_keyframe_unpack_cdr (host code, no GPU): dict of the message's fields as flat arrays"""
    L = lib()
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
    L.dvs_keyframe_unpack_cdr.argtypes = [vp, sz, C.POINTER(KeyframeHeader), vp, sz, vp, vp, vp, vp, vp, i32, C.POINTER(i32), C.POINTER(i32)]
    buf = np.frombuffer(payload, np.uint8)
    hdr = KeyframeHeader(); fid = C.create_string_buffer(256)
    lid = np.zeros(cap_n, np.uint64); xyz = np.zeros((cap_n, 3)); oid = np.zeros(cap_n, np.uint64); px = np.zeros((cap_n, 2))
    desc = np.zeros((cap_n, 32), np.uint8); nl = C.c_int32(); no = C.c_int32()
    check(L.dvs_keyframe_unpack_cdr(buf.ctypes.data, len(buf), C.byref(hdr), C.cast(fid, vp), 256, ptr(lid), ptr(xyz), ptr(oid), ptr(px), ptr(desc),
                                    cap_n, C.byref(nl), C.byref(no)))
    return dict(stamp=(hdr.stamp_sec, hdr.stamp_nanosec), frame_id=fid.value.decode(), keyframe_id=hdr.keyframe_id,
                translation=np.array(hdr.translation[:]), rotation_xyzw=np.array(hdr.rotation_xyzw[:]),
                landmark_ids=lid[:nl.value], landmark_xyz=xyz[:nl.value], obs_landmark_ids=oid[:no.value], obs_pixels=px[:no.value],
                obs_desc=desc[:no.value])


def cv_ransac_subsets_nocheck(n, model_points, iterations):
    """host only: the samples of a RANSAC callback WITHOUT checkSubset (cv::solvePnPRansac's 5-point samples): dvs_cv_ransac_subsets(NULL, NULL, ...)"""
    idx = np.zeros((max(iterations, 1), model_points), np.int32); found = C.c_int32()
    check(lib().dvs_cv_ransac_subsets(None, None, n, model_points, iterations, ptr(idx), C.byref(found)))
    return idx[:found.value]


def cv_ransac_subsets(pts1, pts2, model_points, iterations):
    """host only: the sample sequence cv::findFundamentalMat's RANSAC draws (dvs_cv_ransac_subsets) -> (idx[found][model_points], found)"""
    p1 = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
    idx = np.zeros((max(iterations, 1), model_points), np.int32); found = C.c_int32()
    check(lib().dvs_cv_ransac_subsets(ptr(p1), ptr(p2), len(p1), model_points, iterations, ptr(idx), C.byref(found)))
    return idx[:found.value], found.value
