"""ctypes bindings to oracle/_build/liboracle.so — the CPU restatement of the reference path.
TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline."""
import ctypes as C
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


FLAGS = "-O2 -ffp-contract=off (portable x86-64)"


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def use_native():
    """bench.py's cpu_baseline: rebuild the oracle with -O2 -march=native ON THIS MACHINE and load that copy.
    Must be called before the first lib().  The native build is deleted first so that a copy made on another CPU never runs
    here.  Falls back to the portable build (and says so in FLAGS) if the compiler is missing."""
    global _SO, FLAGS
    assert "_lib" not in globals(), "use_native() must precede the first oracle call"
    nat = os.path.join(_ROOT, "oracle", "_build", "native")
    try:
        subprocess.check_call(["rm", "-rf", nat])
        subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle"), "native"])
        _SO = os.path.join(nat, "liboracle.so")
        FLAGS = open(os.path.join(nat, "flags.txt")).read().strip().split(" -std")[0] + " -ffp-contract=off"
    except Exception as e:  # noqa: BLE001
        FLAGS += f" [native build failed: {e}]"
    return FLAGS


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def lib():
    global _lib
    try:
        return _lib
    except NameError:
        pass
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orc_orb_create.restype = vp; L.orc_orb_create.argtypes = [i32, f32, i32, i32, i32]
    L.orc_orb_destroy.argtypes = [vp]
    L.orc_orb_set_gauss_kernel.argtypes = [vp, vp]
    L.orc_orb_extract.restype = i32; L.orc_orb_extract.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32]
    L.orc_orb_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orc_orb_level_size.argtypes = [vp, i32, i32, i32, vp, vp]
    L.orc_orb_get_level.restype = i32; L.orc_orb_get_level.argtypes = [vp, i32, i32, vp, i32]
    L.orc_orb_get_candidates.restype = i32; L.orc_orb_get_candidates.argtypes = [vp, i32, vp, i32]
    L.orc_orb_get_level_keypoints.restype = i32; L.orc_orb_get_level_keypoints.argtypes = [vp, i32, vp, i32]
    L.orc_resize_linear_u8.argtypes = [vp, i32, i32, sz, vp, i32, i32, sz]
    L.orc_fast.restype = i32; L.orc_fast.argtypes = [vp, i32, i32, sz, i32, vp, i32]
    L.orc_gauss7.argtypes = [vp, i32, i32, vp, vp]
    L.orc_fast_atan2.restype = f32; L.orc_fast_atan2.argtypes = [f32, f32]
    L.orc_ic_angle.restype = f32; L.orc_ic_angle.argtypes = [vp, sz, f32, f32]
    L.orc_descriptor.argtypes = [vp, sz, f32, f32, f32, vp]
    L.orc_distribute.restype = i32; L.orc_distribute.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32]
    L.orc_std_sort_nodes.argtypes = [vp, vp, i32, vp]
    L.orc_brief_pattern.restype = C.POINTER(C.c_int8)
    L.orc_sincosf.argtypes = [f32, vp, vp]
    L.orc_match_hamming.argtypes = [vp, i32, vp, i32, i32, vp, vp]
    L.orc_match_hamming256.argtypes = [vp, i32, vp, i32, vp, vp]
    L.orc_match_hamming_thresh.restype = i32; L.orc_match_hamming_thresh.argtypes = [vp, i32, vp, i32, i32, vp, i32]
    L.orc_cvorb_create.restype = vp; L.orc_cvorb_create.argtypes = [i32, f32, i32, i32, i32, i32]
    L.orc_cvorb_destroy.argtypes = [vp]
    L.orc_cvorb_detect_and_compute.restype = i32; L.orc_cvorb_detect_and_compute.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32]
    L.orc_cvorb_level.restype = i32; L.orc_cvorb_level.argtypes = [vp, i32, i32, vp, i32, vp, vp]
    L.orc_resize_linear_exact_u8.argtypes = [vp, i32, i32, sz, vp, i32, i32, sz]
    L.orc_retain_best.restype = i32; L.orc_retain_best.argtypes = [vp, i32, i32, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleORB:
    """Mirror of ORB_SLAM3::ORBextractor (ORBextractor.hpp:44-110) on the CPU oracle."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.h = self.L.orc_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def __del__(self):
        try:
            self.L.orc_orb_destroy(self.h)
        except Exception:
            pass

    def set_gauss_kernel(self, k7):
        k = np.asarray(k7, dtype=np.int32); assert k.size == 7
        self.L.orc_orb_set_gauss_kernel(self.h, _p(k))

    def tables(self):
        n = self.nlevels
        s = np.zeros(n, np.float32); inv = np.zeros(n, np.float32); f = np.zeros(n, np.int32); um = np.zeros(16, np.int32)
        self.L.orc_orb_tables(self.h, _p(s), _p(inv), _p(f), _p(um))
        return s, inv, f, um

    def level_size(self, cols, rows, level):
        w = C.c_int(); h = C.c_int()
        self.L.orc_orb_level_size(self.h, cols, rows, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def extract(self, img):
        """returns (keypoints structured array, descriptors N x 32 uint8) or raises on error code"""
        img = np.asarray(img)
        rows, cols = (img.shape if img.size else (0, 0))
        cap = self.nfeatures + 3 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        step = img.strides[0] if img.size else 0
        n = self.L.orc_orb_extract(self.h, _p(img) if img.size else None, rows, cols, step, _p(kps), _p(desc), cap)
        if n < 0:
            return n, None, None
        self._shape = (rows, cols)
        return n, kps[:n].copy(), desc[:n].copy()

    def level(self, l, blurred=False):
        rows, cols = self._shape
        w, h = self.level_size(cols, rows, l)
        buf = np.zeros((h, w), np.uint8)
        n = self.L.orc_orb_get_level(self.h, l, int(blurred), _p(buf), buf.size)
        return buf if n == buf.size else None

    def candidates(self, l):
        cap = 1 << 20
        buf = np.zeros((cap, 3), np.int32)
        n = self.L.orc_orb_get_candidates(self.h, l, _p(buf), cap)
        assert n >= 0
        return buf[:n].copy()

    def level_keypoints(self, l):
        cap = self.nfeatures + 64
        kps = np.zeros(cap, KP_DTYPE)
        n = self.L.orc_orb_get_level_keypoints(self.h, l, _p(kps), cap)
        assert n >= 0
        return kps[:n].copy()


class OracleCvORB:
    """oracle/cvorb_oracle.cpp: cv::ORB::create(...)->detectAndCompute as the reference's test uses it (test_dbow2_integration.cpp:19,38)"""

    def __init__(self, nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, scoreType=0, fastThreshold=20):
        self._L = lib()
        self._h = self._L.orc_cvorb_create(nfeatures, scaleFactor, nlevels, edgeThreshold, scoreType, fastThreshold)
        self.nlevels = nlevels

    def __del__(self):
        try:
            self._L.orc_cvorb_destroy(self._h)
        except Exception:
            pass

    def detectAndCompute(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        cap = 1 << 16
        kps = np.zeros(cap, KP_DTYPE); desc = np.zeros((cap, 32), np.uint8)
        n = self._L.orc_cvorb_detect_and_compute(self._h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), _p(desc), cap)
        assert n >= 0, n
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l, blurred=False):
        w, h = C.c_int(), C.c_int()
        buf = np.zeros(1 << 24, np.uint8)
        assert self._L.orc_cvorb_level(self._h, l, int(blurred), _p(buf), buf.size, C.byref(w), C.byref(h)) == 0
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()


def resize_linear_exact(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_exact_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def retain_best(responses, n_points):
    """KeyPointsFilter::retainBest with the REAL std::nth_element / std::partition -> surviving original indices in their order"""
    r = np.ascontiguousarray(responses, np.float32)
    perm = np.zeros(max(len(r), 1), np.int32)
    k = lib().orc_retain_best(_p(r), len(r), n_points, _p(perm))
    return perm[:k].copy()


def match(q, t):
    """cv::BFMatcher(NORM_HAMMING).match restatement -> (train_idx int32[nq], dist int32[nq])"""
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    idx = np.zeros(len(q), np.int32); d = np.zeros(len(q), np.int32)
    lib().orc_match_hamming(_p(q), len(q), _p(t), len(t), q.shape[1] if q.ndim == 2 else 32, _p(idx), _p(d))
    return idx, d


def match_thresh(q, t, max_dist, cap=None):
    q = np.ascontiguousarray(q, np.uint8); t = np.ascontiguousarray(t, np.uint8)
    cap = cap if cap is not None else max(len(q) * len(t), 1)
    pairs = np.zeros((cap, 3), np.int32)
    n = lib().orc_match_hamming_thresh(_p(q), len(q), _p(t), len(t), max_dist, _p(pairs), cap)
    return n, pairs[:min(n, cap)].copy()


# ------------------------------------------------ bundle adjustment oracle ----------------------------------------------
class BaSummary(C.Structure):
    _fields_ = [("termination", C.c_int32), ("num_successful_steps", C.c_int32), ("num_iterations", C.c_int32),
                ("reserved", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class OracleBA:
    """CPU restatement of WeightedSquaredReprojectionError + Ceres autodiff/Huber/manifold/LM (oracle/ba_oracle.cpp)"""

    def __init__(self, prob):
        L = lib()
        vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
        L.orc_ba_create.restype = vp
        L.orc_ba_create.argtypes = [i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, dbl, dbl, dbl, dbl, dbl, dbl]
        L.orc_ba_destroy.argtypes = [vp]
        L.orc_ba_evaluate_raw.argtypes = [vp, vp, vp, vp, vp]
        L.orc_ba_evaluate.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_ba_normal_equations.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_ba_solve.argtypes = [vp, i32, dbl, dbl, dbl, vp]
        L.orc_ba_evaluate_mt.argtypes = [vp, i32, i32, vp]
        L.orc_ba_get_trace.argtypes = [vp, vp, i32]; L.orc_ba_get_trace.restype = i32
        L.orc_ba_get_parameters.argtypes = [vp, vp, vp, vp]
        self.L = L
        self.K, self.Ln, self.R = prob["K"], prob["L"], len(prob["cam_idx"])
        self._keep = [np.ascontiguousarray(prob[k]) for k in ("q", "t", "X", "cam_idx", "lm_idx", "uv", "pose_fixed", "lm_fixed")]
        q, t, X, cam, lm, uv, pf, lf = self._keep
        self.h = L.orc_ba_create(self.K, _p(q), _p(t), self.Ln, _p(X), self.R, _p(cam.astype(np.int32)), _p(lm.astype(np.int32)), _p(uv),
                                 _p(pf.astype(np.uint8)), _p(lf.astype(np.uint8)), prob["fx"], prob["fy"], prob["cx"], prob["cy"],
                                 prob["sigma"], prob["huber"])

    def __del__(self):
        try:
            self.L.orc_ba_destroy(self.h)
        except Exception:
            pass

    def evaluate_raw(self):
        R = self.R
        r = np.zeros((R, 2)); jq = np.zeros((R, 2, 4)); jt = np.zeros((R, 2, 3)); jx = np.zeros((R, 2, 3))
        self.L.orc_ba_evaluate_raw(self.h, _p(r), _p(jq), _p(jt), _p(jx))
        return r, jq, jt, jx

    def evaluate(self):
        R = self.R
        cost = C.c_double(); r = np.zeros((R, 2)); jp = np.zeros((R, 2, 6)); jl = np.zeros((R, 2, 3)); g = np.zeros(6 * self.K + 3 * self.Ln)
        self.L.orc_ba_evaluate(self.h, C.byref(cost), _p(r), _p(jp), _p(jl), _p(g))
        return cost.value, r, jp, jl, g

    def normal_equations(self):
        hpp = np.zeros((self.K, 6, 6)); hll = np.zeros((self.Ln, 3, 3)); w = np.zeros((self.R, 6, 3)); g = np.zeros(6 * self.K + 3 * self.Ln)
        cost = C.c_double()
        self.L.orc_ba_normal_equations(self.h, _p(hpp), _p(hll), _p(w), _p(g), C.byref(cost))
        return hpp, hll, w, g, cost.value

    def solve(self, max_iterations=10, ftol=1e-6, gtol=1e-10, ptol=1e-8):
        s = BaSummary()
        self.L.orc_ba_solve(self.h, max_iterations, ftol, gtol, ptol, C.byref(s))
        return s

    def trace(self):
        n = self.L.orc_ba_get_trace(self.h, None, 0)
        rows = np.zeros((n, 6))
        self.L.orc_ba_get_trace(self.h, _p(rows), n)
        return rows

    def evaluate_mt(self, nthreads, reps=1):
        """`reps` full evaluations with the residual blocks split over `nthreads` threads (timed CPU baseline)"""
        cost = C.c_double()
        self.L.orc_ba_evaluate_mt(self.h, nthreads, reps, C.byref(cost))
        return cost.value

    def parameters(self):
        q = np.zeros((self.K, 4)); t = np.zeros((self.K, 3)); X = np.zeros((self.Ln, 3))
        self.L.orc_ba_get_parameters(self.h, _p(q), _p(t), _p(X))
        return q, t, X


# ------------------------------------------------ frontend / backend glue oracle (N1, N2) -------------------------------
def _glue_lib():
    L = lib()
    vp, i32, sz, f32, dbl = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double
    L.orc_bgr_to_gray.argtypes = [vp, i32, i32, sz, vp, sz, i32]
    L.orc_filter_depth.restype = i32; L.orc_filter_depth.argtypes = [vp, vp, i32, vp, i32, i32, sz, f32, f32, vp, vp, vp]
    L.orc_filter_matches.restype = i32; L.orc_filter_matches.argtypes = [vp, vp, i32, f32, vp]
    L.orc_backproject.restype = i32; L.orc_backproject.argtypes = [vp, i32, vp, sz, f32, f32, f32, f32, vp, vp, vp, vp]
    L.orc_associate.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp, dbl, dbl, dbl, dbl, dbl, dbl, vp]
    L.orc_publish_keyframe_cdr.restype = sz
    L.orc_publish_keyframe_cdr.argtypes = [i32, C.c_uint32, C.c_char_p, C.c_uint64, vp, vp, vp, vp, i32, vp, sz, f32, f32, f32, f32, vp, vp, vp, sz, vp]
    L.orc_unpack_keyframe_cdr.restype = i32
    L.orc_unpack_keyframe_cdr.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    return L


def harris_responses(img, xs, ys, block_size=7, k=0.04):
    img = np.ascontiguousarray(img, np.uint8); xs = np.ascontiguousarray(xs, np.int32); ys = np.ascontiguousarray(ys, np.int32)
    out = np.zeros(len(xs), np.float32)
    L = _glue_lib()
    L.orc_harris_responses.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    L.orc_harris_responses(_p(img), img.shape[0], img.shape[1], img.shape[1], _p(xs), _p(ys), len(xs), block_size, k, _p(out))
    return out


def publish_keyframe(kps, desc, depth, fx, fy, cx, cy, R, t, stamp=(0, 0), frame_id="camera_link", keyframe_id=0, q_xyzw=(0, 0, 0, 1)):
    """oracle: Keyframe.msg built as objects and serialised by a generic CDR stream -> (bytes, n_landmarks)"""
    kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
    R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64).reshape(3); q = np.ascontiguousarray(q_xyzw, np.float64)
    n = len(kps); cap = 256 + 96 * (n + 1); out = np.zeros(cap, np.uint8); m = C.c_int32()
    size = _glue_lib().orc_publish_keyframe_cdr(int(stamp[0]), int(stamp[1]), frame_id.encode(), int(keyframe_id), _p(t), _p(q), _p(kps), _p(desc), n,
                                                _p(depth), depth.shape[1] * 2, fx, fy, cx, cy, _p(R), _p(t), _p(out), cap, C.byref(m))
    assert size <= cap
    return out[:size].tobytes(), m.value


def unpack_keyframe(payload, cap_n=4096):
    buf = np.frombuffer(payload, np.uint8)
    sec = C.c_int32(); nsec = C.c_uint32(); fid = C.create_string_buffer(256); kid = C.c_uint64()
    tr = np.zeros(3); rot = np.zeros(4); lid = np.zeros(cap_n, np.uint64); xyz = np.zeros((cap_n, 3)); oid = np.zeros(cap_n, np.uint64)
    px = np.zeros((cap_n, 2)); desc = np.zeros((cap_n, 32), np.uint8); nl = C.c_int32(); no = C.c_int32()
    rc = _glue_lib().orc_unpack_keyframe_cdr(buf.ctypes.data, len(buf), C.byref(sec), C.byref(nsec), C.cast(fid, C.c_void_p), 256, C.byref(kid), _p(tr), _p(rot),
                                             _p(lid), _p(xyz), _p(oid), _p(px), _p(desc), cap_n, C.byref(nl), C.byref(no))
    assert rc == 0, rc
    return dict(stamp=(sec.value, nsec.value), frame_id=fid.value.decode(), keyframe_id=kid.value, translation=tr, rotation_xyzw=rot,
                landmark_ids=lid[:nl.value], landmark_xyz=xyz[:nl.value], obs_landmark_ids=oid[:no.value], obs_pixels=px[:no.value],
                obs_desc=desc[:no.value])


def bgr_to_gray(bgr, variant=0):
    bgr = np.ascontiguousarray(bgr); rows, cols, _ = bgr.shape
    g = np.zeros((rows, cols), np.uint8)
    _glue_lib().orc_bgr_to_gray(_p(bgr), rows, cols, cols * 3, _p(g), cols, variant)
    return g


def filter_depth(kps, desc, depth, dmin=0.3, dmax=3.0):
    kps = np.ascontiguousarray(kps, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8); depth = np.ascontiguousarray(depth, np.uint16)
    n = len(kps); ok = np.zeros(n, KP_DTYPE); od = np.zeros((n, 32), np.uint8); oi = np.zeros(n, np.int32)
    m = _glue_lib().orc_filter_depth(_p(kps), _p(desc), n, _p(depth), depth.shape[0], depth.shape[1], depth.shape[1] * 2, dmin, dmax, _p(ok), _p(od), _p(oi))
    return ok[:m], od[:m], oi[:m]


def filter_matches(idx, dist, maxd=50.0):
    idx = np.ascontiguousarray(idx, np.int32); dist = np.ascontiguousarray(dist, np.int32)
    out = np.zeros((len(idx), 3), np.int32)
    m = _glue_lib().orc_filter_matches(_p(idx), _p(dist), len(idx), maxd, _p(out))
    return out[:m]


def backproject(kps, depth, fx, fy, cx, cy, R, t):
    kps = np.ascontiguousarray(kps, KP_DTYPE); depth = np.ascontiguousarray(depth, np.uint16)
    R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64)
    n = len(kps); w = np.zeros((n, 3)); oi = np.zeros(n, np.int32)
    m = _glue_lib().orc_backproject(_p(kps), n, _p(depth), depth.shape[1] * 2, fx, fy, cx, cy, _p(R), _p(t), _p(w), _p(oi))
    return w[:m], oi[:m]


def associate(obs_desc, obs_px, lm_desc, lm_xyz, R, t, fx, fy, cx, cy, max_desc=50.0, max_reproj=5.0):
    obs_desc = np.ascontiguousarray(obs_desc, np.uint8); obs_px = np.ascontiguousarray(obs_px, np.float32)
    lm_desc = np.ascontiguousarray(lm_desc, np.uint8); lm_xyz = np.ascontiguousarray(lm_xyz, np.float32)
    R = np.ascontiguousarray(R, np.float64); t = np.ascontiguousarray(t, np.float64)
    best = np.zeros(len(obs_desc), np.int32)
    _glue_lib().orc_associate(_p(obs_desc), _p(obs_px), len(obs_desc), _p(lm_desc), _p(lm_xyz), len(lm_desc), _p(R), _p(t), fx, fy, cx, cy,
                              max_desc, max_reproj, _p(best))
    return best


# ------------------------------------------------ robust estimation stages (N4) -------------------------------------------------
def _ransac_lib():
    L = lib()
    vp, i32, dbl, u64 = C.c_void_p, C.c_int, C.c_double, C.c_uint64
    L.orc_splitmix64.restype = u64; L.orc_splitmix64.argtypes = [u64]
    L.orc_sample_distinct.argtypes = [u64, i32, i32, i32, vp]
    L.orc_quartic_real_roots.restype = i32; L.orc_quartic_real_roots.argtypes = [dbl, dbl, dbl, dbl, dbl, vp]
    L.orc_p3p.restype = i32; L.orc_p3p.argtypes = [vp, vp, vp]
    L.orc_eight_point.restype = i32; L.orc_eight_point.argtypes = [vp, vp, i32, vp]
    L.orc_find_fundamental_ransac.argtypes = [vp, vp, i32, dbl, dbl, i32, u64, vp, vp, vp]
    L.orc_cv_rng_next.restype = C.c_uint32; L.orc_cv_rng_next.argtypes = [C.POINTER(u64)]
    L.orc_cv_subsets.restype = i32; L.orc_cv_subsets.argtypes = [vp, vp, i32, i32, i32, vp]
    L.orc_seven_point.restype = i32; L.orc_seven_point.argtypes = [vp, vp, vp, vp]
    L.orc_find_fundamental_cv.argtypes = [vp, vp, i32, dbl, dbl, i32, vp, vp, vp]
    L.orc_solve_pnp_ransac.restype = i32
    L.orc_solve_pnp_ransac.argtypes = [vp, vp, i32, vp, i32, dbl, dbl, u64, vp, vp, vp, vp, vp]
    L.orc_solve_pnp_ransac_cv.restype = i32
    L.orc_solve_pnp_ransac_cv.argtypes = [vp, vp, i32, vp, i32, dbl, dbl, vp, vp, vp, vp, vp, vp]
    L.orc_cv_subsets_nocheck.restype = i32; L.orc_cv_subsets_nocheck.argtypes = [i32, i32, i32, vp]
    L.orc_epnp.argtypes = [vp, vp, i32, vp, vp, vp]; L.orc_epnp.restype = None
    L.orc_solve_pnp_iterative.restype = i32; L.orc_solve_pnp_iterative.argtypes = [vp, vp, i32, vp, vp, vp]
    return L


def sample_distinct(seed, h, n, k):
    idx = np.zeros(k, np.int32)
    _ransac_lib().orc_sample_distinct(seed, h, n, k, _p(idx))
    return idx


def quartic_real_roots(a):
    r = np.zeros(4)
    n = _ransac_lib().orc_quartic_real_roots(*[float(v) for v in a], _p(r))
    return np.sort(r[:n])


def p3p(P, j):
    out = np.zeros((4, 12))
    n = _ransac_lib().orc_p3p(_p(np.ascontiguousarray(P, np.float64)), _p(np.ascontiguousarray(j, np.float64)), _p(out))
    return [(out[i, :9].reshape(3, 3).copy(), out[i, 9:].copy()) for i in range(n)]


def eight_point(p1, p2):
    F = np.zeros(9)
    p1 = np.ascontiguousarray(p1, np.float32); p2 = np.ascontiguousarray(p2, np.float32)
    ok = _ransac_lib().orc_eight_point(_p(p1), _p(p2), len(p1), _p(F))
    return F.reshape(3, 3) if ok else None


def find_fundamental_ransac(p1, p2, threshold=2.0, confidence=0.99, max_iters=1000, seed=1):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    F = np.zeros(9); mask = np.zeros(max(len(p1), 1), np.uint8); sel = np.zeros(3, np.int32)
    _ransac_lib().orc_find_fundamental_ransac(_p(p1), _p(p2), len(p1), threshold, confidence, max_iters, seed, _p(F), _p(mask), _p(sel))
    return F.reshape(3, 3), mask[:len(p1)], sel


def cv_rng_sequence(state, count):
    """cv::RNG(state).next() x count"""
    st = C.c_uint64(state)
    return [int(_ransac_lib().orc_cv_rng_next(C.byref(st))) for _ in range(count)]


def cv_subsets(p1, p2, model_points, iterations):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    idx = np.zeros((max(iterations, 1), model_points), np.int32)
    found = _ransac_lib().orc_cv_subsets(_p(p1), _p(p2), len(p1), model_points, iterations, _p(idx))
    return idx[:found], found


def seven_point(p1, p2, idx7):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    F = np.zeros((3, 9)); idx = np.ascontiguousarray(idx7, np.int32)
    n = _ransac_lib().orc_seven_point(_p(p1), _p(p2), _p(idx), _p(F))
    return [F[k].reshape(3, 3).copy() for k in range(n)]


def find_fundamental_cv(p1, p2, threshold=2.0, confidence=0.99, max_iters=1000):
    p1 = np.ascontiguousarray(p1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(p2, np.float32).reshape(-1, 2)
    F = np.zeros(9); mask = np.zeros(max(len(p1), 1), np.uint8); sel = np.zeros(3, np.int32)
    _ransac_lib().orc_find_fundamental_cv(_p(p1), _p(p2), len(p1), threshold, confidence, max_iters, _p(F), _p(mask), _p(sel))
    return F.reshape(3, 3), mask[:len(p1)], sel


def solve_pnp_ransac_cv(obj, img, K4, iterations=100, reproj_err=4.0, confidence=0.99):
    """cv::solvePnPRansac by OpenCV's procedure (oracle/pnp_cv_oracle.cpp) -> (ok, rvec, tvec, inliers, sel = (best iteration, iterations run,
    best count), the RANSAC stage's model (rvec, tvec))"""
    obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3); img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
    K4 = np.ascontiguousarray(K4, np.float64)
    rvec = np.zeros(3); tvec = np.zeros(3); inl = np.zeros(max(len(obj), 1), np.int32); nin = C.c_int(); sel = np.zeros(3, np.int32); m6 = np.zeros(6)
    ok = _ransac_lib().orc_solve_pnp_ransac_cv(_p(obj), _p(img), len(obj), _p(K4), iterations, reproj_err, confidence, _p(rvec), _p(tvec), _p(inl),
                                               C.byref(nin), _p(sel), _p(m6))
    return bool(ok), rvec, tvec, inl[:nin.value].copy(), sel, m6


def cv_subsets_nocheck(n, k, iters):
    idx = np.zeros((iters, k), np.int32)
    _ransac_lib().orc_cv_subsets_nocheck(n, k, iters, _p(idx))
    return idx


def epnp(obj, img, K4):
    obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3); img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
    r = np.zeros(3); t = np.zeros(3)
    _ransac_lib().orc_epnp(_p(obj), _p(img), len(obj), _p(np.ascontiguousarray(K4, np.float64)), _p(r), _p(t))
    return r, t


def solve_pnp_iterative(obj, img, K4):
    obj = np.ascontiguousarray(obj, np.float64).reshape(-1, 3); img = np.ascontiguousarray(img, np.float64).reshape(-1, 2)
    r = np.zeros(3); t = np.zeros(3)
    ok = _ransac_lib().orc_solve_pnp_iterative(_p(obj), _p(img), len(obj), _p(np.ascontiguousarray(K4, np.float64)), _p(r), _p(t))
    return bool(ok), r, t


def solve_pnp_ransac(obj, img, K4, iterations=100, reproj_err=4.0, confidence=0.99, seed=1):
    obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3); img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
    K4 = np.ascontiguousarray(K4, np.float64)
    rvec = np.zeros(3); tvec = np.zeros(3); inl = np.zeros(max(len(obj), 1), np.int32); nin = C.c_int(); sel = np.zeros(3, np.int32)
    ok = _ransac_lib().orc_solve_pnp_ransac(_p(obj), _p(img), len(obj), _p(K4), iterations, reproj_err, confidence, seed, _p(rvec), _p(tvec), _p(inl),
                                            C.byref(nin), _p(sel))
    return bool(ok), rvec, tvec, inl[:nin.value].copy(), sel
