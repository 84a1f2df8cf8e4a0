import ctypes as C
import numpy as np
from ._lib import lib, check, ptr, KP_DTYPE, DvsError


class CvOrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("edge_threshold", C.c_int32),
                ("first_level", C.c_int32), ("wta_k", C.c_int32), ("score_type", C.c_int32), ("patch_size", C.c_int32),
                ("fast_threshold", C.c_int32)]


class CvORB:
    """Python mirror of cv::ORB as the reference uses it (test_dbow2_integration.cpp:19,38): CvORB.create(nfeatures, ...) and
    detectAndCompute(image) -> (keypoints, descriptors) on the HIP library (dvs_cvorb_*, csrc/cvorb.hip)."""
    HARRIS_SCORE, FAST_SCORE = 0, 1

    def __init__(self, nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2, scoreType=0, patchSize=31,
                 fastThreshold=20, device=0):
        self._L = lib()
        self._L.dvs_cvorb_create.argtypes = [C.POINTER(CvOrbParams), C.c_int32, C.POINTER(C.c_void_p)]
        self._L.dvs_cvorb_destroy.argtypes = [C.c_void_p]; self._L.dvs_cvorb_destroy.restype = None
        self._L.dvs_cvorb_detect_and_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int32,
                                                         C.POINTER(C.c_int32)]
        self._L.dvs_cvorb_get_level.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        p = CvOrbParams(nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold)
        h = C.c_void_p()
        check(self._L.dvs_cvorb_create(C.byref(p), device, C.byref(h)))
        self._h, self.nfeatures, self.nlevels = h, nfeatures, nlevels
        self._cap = nfeatures + 64

    create = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def close(self):
        if getattr(self, "_h", None):
            self._L.dvs_cvorb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detectAndCompute(self, image, mask=None):
        assert mask is None, "detection masks are not built"
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2 and image.strides[1] == 1
        rows, cols = image.shape
        for _ in range(2):
            kps = np.zeros(self._cap, KP_DTYPE); desc = np.zeros((self._cap, 32), np.uint8)
            n = C.c_int32()
            st = self._L.dvs_cvorb_detect_and_compute(self._h, ptr(image), rows, cols, image.strides[0], ptr(kps), ptr(desc), self._cap, C.byref(n))
            if st == -3 and n.value > self._cap:     # DVS_ERR_CAPACITY: retainBest kept ties beyond the quota
                self._cap = n.value + 64
                continue
            check(st)
            return kps[:n.value].copy(), desc[:n.value].copy()
        raise DvsError(-3, "capacity")

    def level(self, l, blurred=False):
        w, h = C.c_int32(), C.c_int32()
        buf = np.zeros(1 << 24, np.uint8)
        check(self._L.dvs_cvorb_get_level(self._h, l, int(blurred), ptr(buf), buf.size, C.byref(w), C.byref(h)))
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()


def retain_best_host(responses, n_points):
    """csrc/lsort.h's restatement of KeyPointsFilter::retainBest (std::nth_element + std::partition) -> surviving original indices"""
    from ._lib import test_lib
    L = test_lib()
    r = np.ascontiguousarray(responses, np.float32)
    perm = np.zeros(max(len(r), 1), np.int32); k = C.c_int32()
    L.dvs_test_retain_best_host(ptr(r), len(r), n_points, ptr(perm), C.byref(k))
    return perm[:k.value].copy()


def retain_best_device(responses, n_points):
    """the same through the wavefront routine of the cv::ORB kernels (needs the GPU)"""
    from ._lib import test_lib
    L = test_lib()
    r = np.ascontiguousarray(responses, np.float32)
    perm = np.zeros(max(len(r), 1), np.int32); k = C.c_int32()
    check(L.dvs_test_retain_best_device(ptr(r), len(r), n_points, ptr(perm), C.byref(k)))
    return perm[:k.value].copy()
