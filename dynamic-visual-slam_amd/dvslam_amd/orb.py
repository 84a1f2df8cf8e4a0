import ctypes as C
import numpy as np
from ._lib import lib, test_lib, check, ptr, OrbParams, KP_DTYPE, DvsError

STAGES = ("pyramid", "fast", "octree", "blur", "describe")


class ORBextractor:
    """Python mirror of ORB_SLAM3::ORBextractor (reference ORBextractor.hpp:44-110) on the HIP library.

    ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST); calling it on a gray uint8
    image returns (n, keypoints, descriptors) where n is operator()'s return value (-1 for an empty
    image, ORBextractor.cpp:1090-1091), keypoints is a cv::KeyPoint-layout structured array and
    descriptors an n x 32 uint8 array."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0, max_batch=1,
                 gauss_kernel=None, hooks=False):
        # hooks=True: every call of this object goes through lib/libdvslam_hip_test.so, which also exports the scheduling / introspection
        # hooks (hint_next_batch_device, set_*_event, set_overlap, candidates, level_keypoints, stage timing); the product library does not
        self._L = test_lib() if hooks else lib()
        self.hooks = bool(hooks)
        p = OrbParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, (C.c_int32 * 7)(*(gauss_kernel or [0] * 7)), max_batch)
        h = C.c_void_p()
        check(self._L.dvs_orb_create(C.byref(p), device, C.byref(h)))
        self._h = h
        self.nfeatures, self.nlevels, self.scaleFactor, self.device, self.max_batch = nfeatures, nlevels, scaleFactor, device, max_batch
        self.capacity = self._L.dvs_orb_max_keypoints(self._h)

    @classmethod
    def from_handle(cls, handle, nfeatures, nlevels, scaleFactor, device, max_batch, L=None):
        """non-owning view of a dvs_orb* that lives inside another handle (dvs_pipeline_extractor: test library only)"""
        o = cls.__new__(cls)
        o._L, o._h, o._owned = (L or lib()), C.c_void_p(handle), False
        o.hooks = L is not None and L is not lib()
        o.nfeatures, o.nlevels, o.scaleFactor, o.device, o.max_batch = nfeatures, nlevels, scaleFactor, device, max_batch
        o.capacity = o._L.dvs_orb_max_keypoints(o._h)
        return o

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                self._L.dvs_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- getters of the reference class (ORBextractor.hpp:62-82) --
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def _tables(self):
        n = self.nlevels
        t = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32), np.zeros(16, np.int32)]
        check(self._L.dvs_orb_get_tables(self._h, *[ptr(a) for a in t]))
        return t

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        return self._tables()[4]

    def umax(self):
        return self._tables()[5]

    def level_size(self, rows, cols, level):
        r, c = C.c_int32(), C.c_int32()
        check(self._L.dvs_orb_level_size(self._h, rows, cols, level, C.byref(r), C.byref(c)))
        return c.value, r.value  # (w, h)

    # -- operator() --
    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        image = np.asarray(image)
        if image.size == 0:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (ORBextractor.cpp:1094)"
        assert image.strides[1] == 1
        rows, cols = image.shape
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int32()
        st = self._L.dvs_orb_extract(self._h, ptr(image), rows, cols, image.strides[0], ptr(kps), ptr(desc), self.capacity, C.byref(n))
        if st == -1:
            return -1, np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        check(st)
        self._shape = (rows, cols)
        return n.value, kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images):
        images = [np.ascontiguousarray(im) for im in images]
        rows, cols = images[0].shape
        n = len(images)
        arr = (C.c_void_p * n)(*[im.ctypes.data for im in images])
        kps = np.zeros((n, self.capacity), KP_DTYPE)
        desc = np.zeros((n, self.capacity, 32), np.uint8)
        nout = np.zeros(n, np.int32)
        check(self._L.dvs_orb_extract_batch(self._h, arr, n, rows, cols, images[0].strides[0], ptr(kps), ptr(desc), self.capacity, ptr(nout)))
        self._shape = (rows, cols)
        return nout, kps, desc

    def hint_next_batch_device(self, d_next_imgs):
        """announce the NEXT batch (device pointer, same layout as the coming extract_batch_device call): its pyramid is built
        beside this batch's descriptor stage and the match that follows; one-shot, results unchanged"""
        check(self._L.dvs_orb_hint_next_batch_device(self._h, d_next_imgs))

    def set_output_event(self, hip_event, defer=None):
        """hipEvent_t (int, 0 to clear) recorded where a device-resident call's outputs are complete; defer=True/False also switches
        the deferred descriptor stage (dvs_orb_set_defer_outputs)"""
        check(self._L.dvs_orb_set_output_event(self._h, hip_event or None))
        if defer is not None:
            check(self._L.dvs_orb_set_defer_outputs(self._h, int(bool(defer))))

    def set_reuse_guard_event(self, hip_event):
        """one-shot: the next device-resident call writes its outputs only behind this hipEvent_t"""
        check(self._L.dvs_orb_set_reuse_guard_event(self._h, hip_event or None))

    def set_after_fast_event(self, hip_event):
        """hipEvent_t (int, 0 to clear) recorded behind FAST by every following extract_batch_device (scheduling hook)"""
        check(self._L.dvs_orb_set_after_fast_event(self._h, hip_event or None))

    def extract_batch_device(self, d_imgs, nimg, rows, cols, step, frame_stride, d_kps, d_desc, capacity, d_nout):
        """raw device pointers (ints); asynchronous on the handle's stream"""
        check(self._L.dvs_orb_extract_batch_device(self._h, d_imgs, nimg, rows, cols, step, frame_stride, d_kps, d_desc, capacity, d_nout))
        self._shape = (rows, cols)

    def level_block_bytes(self, nimg):
        return int(self._L.dvs_orb_level_block_bytes(self._h, nimg))

    def extract_levels_device(self, d_imgs, nimg, rows, cols, step, frame_stride, level_mask, d_block):
        """level-sharded extraction: this rank's levels into a level-slotted block; asynchronous"""
        check(self._L.dvs_orb_extract_levels_device(self._h, d_imgs, nimg, rows, cols, step, frame_stride, level_mask, d_block))
        self._shape = (rows, cols)

    def merge_levels_device(self, d_blocks, world, level_owner, nimg, d_kps, d_desc, capacity, d_nout):
        """gathered level-slotted blocks of all ranks -> the reference's level-major output; asynchronous"""
        own = np.ascontiguousarray(level_owner, np.int32)
        check(self._L.dvs_orb_merge_levels_device(self._h, d_blocks, world, ptr(own), nimg, d_kps, d_desc, capacity, d_nout))

    def synchronize(self):
        check(self._L.dvs_orb_synchronize(self._h))

    def get_stream(self):
        """the hipStream_t (int) the handle currently enqueues on (its own stream unless set_stream was called)"""
        return int(self._L.dvs_orb_get_stream(self._h) or 0)

    def set_stream(self, stream_ptr):
        check(self._L.dvs_orb_set_stream(self._h, stream_ptr))

    def set_overlap(self, on):
        check(self._L.dvs_orb_set_overlap(self._h, int(on)))

    # -- parity introspection (mvImagePyramid is public in the reference, ORBextractor.hpp:84) --
    def level(self, l, blurred=False, frame=0):
        rows, cols = self._shape
        w, h = self.level_size(rows, cols, l)
        buf = np.zeros((h, w), np.uint8)
        check(self._L.dvs_orb_get_level(self._h, frame, l, int(blurred), ptr(buf), buf.size))
        return buf

    def candidates(self, l, frame=0):
        cap = 1 << 20
        buf = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        check(self._L.dvs_orb_get_candidates(self._h, frame, l, ptr(buf), cap, C.byref(n)))
        return buf[:n.value].copy()

    def level_keypoints(self, l, frame=0):
        cap = self.nfeatures + 64
        buf = np.zeros((cap, 3), np.int32)
        n = C.c_int32()
        check(self._L.dvs_orb_get_level_keypoints(self._h, frame, l, ptr(buf), cap, C.byref(n)))
        return buf[:n.value].copy()

    def enable_stage_timing(self, on=True):
        check(self._L.dvs_orb_enable_stage_timing(self._h, int(on)))

    def stage_times(self, reset=True):
        ms = np.zeros(5, np.float64); calls = np.zeros(5, np.int64)
        check(self._L.dvs_orb_get_stage_times(self._h, ptr(ms), ptr(calls), int(reset)))
        return dict(zip(STAGES, ms.tolist())), dict(zip(STAGES, calls.tolist()))
