import ctypes as C
import os
import subprocess
import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO_PATH = os.environ.get("DVSLAM_HIP_SO") or os.path.join(_PKG, "lib", "libdvslam_hip.so")   # (the override: A/B runs of two builds)

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])

STATUS = {0: "DVS_OK", -1: "DVS_ERR_EMPTY", -2: "DVS_ERR_UNSUPPORTED", -3: "DVS_ERR_CAPACITY", -4: "DVS_ERR_HIP",
          -5: "DVS_ERR_NO_DEVICE", -6: "DVS_ERR_ARG"}


class DvsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32), ("ini_th_fast", C.c_int32),
                ("min_th_fast", C.c_int32), ("gauss_kernel", C.c_int32 * 7), ("max_batch", C.c_int32)]


class PipelineParams(C.Structure):
    _fields_ = [("orb", OrbParams), ("batch", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32), ("nsets", C.c_int32), ("pipelined", C.c_int32),
                ("lanes", C.c_int32), ("quadtree_async", C.c_int32)]


class PipelineSet(C.Structure):
    _fields_ = [("d_kps", C.c_void_p), ("d_desc", C.c_void_p), ("d_n", C.c_void_p), ("d_idx", C.c_void_p), ("d_dist", C.c_void_p),
                ("ev_extracted", C.c_void_p), ("ev_matched", C.c_void_p), ("capacity", C.c_int32)]


class KeyframeHeader(C.Structure):
    _fields_ = [("stamp_sec", C.c_int32), ("stamp_nanosec", C.c_uint32), ("frame_id", C.c_char_p), ("keyframe_id", C.c_uint64),
                ("translation", C.c_double * 3), ("rotation_xyzw", C.c_double * 4)]


class BaSummary(C.Structure):
    _fields_ = [("termination", C.c_int32), ("num_successful_steps", C.c_int32), ("num_iterations", C.c_int32),
                ("linear_solver", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double)]


def build_library():
    """(re)build lib/libdvslam_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j4", "-C", _PKG])


_lib = None


def _bind(L):
    """argument types of the product ABI (include/dvslam_hip.h) — both libraries export it"""
    vp, i32, sz, dbl = C.c_void_p, C.c_int32, C.c_size_t, C.c_double
    L.dvs_last_error.restype = C.c_char_p
    L.dvs_device_count.restype = i32
    L.dvs_device_arch.argtypes = [i32, C.c_char_p, i32]
    L.dvs_malloc.argtypes = [i32, sz, C.POINTER(vp)]
    L.dvs_free.argtypes = [i32, vp]
    L.dvs_memcpy_h2d.argtypes = [i32, vp, vp, sz]
    L.dvs_memcpy_d2h.argtypes = [i32, vp, vp, sz]
    L.dvs_memset.argtypes = [i32, vp, C.c_int, sz]
    L.dvs_orb_create.argtypes = [C.POINTER(OrbParams), i32, C.POINTER(vp)]
    L.dvs_orb_destroy.argtypes = [vp]; L.dvs_orb_destroy.restype = None
    L.dvs_orb_max_keypoints.argtypes = [vp]
    L.dvs_orb_set_stream.argtypes = [vp, vp]
    L.dvs_orb_get_stream.argtypes = [vp]; L.dvs_orb_get_stream.restype = vp
    L.dvs_orb_synchronize.argtypes = [vp]
    L.dvs_orb_get_tables.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.dvs_orb_level_size.argtypes = [vp, i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.dvs_orb_extract.argtypes = [vp, vp, i32, i32, sz, vp, vp, i32, C.POINTER(i32)]
    L.dvs_orb_extract_batch.argtypes = [vp, vp, i32, i32, i32, sz, vp, vp, i32, vp]
    L.dvs_stream_create.argtypes = [i32, i32, C.POINTER(vp)]
    L.dvs_stream_destroy.argtypes = [vp]
    L.dvs_stream_synchronize.argtypes = [vp]
    L.dvs_event_create.argtypes = [i32, C.POINTER(vp)]
    L.dvs_event_query.argtypes = [vp, C.POINTER(i32)]
    L.dvs_event_create_timing.argtypes = [i32, C.POINTER(vp)]
    L.dvs_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.dvs_event_destroy.argtypes = [vp]
    L.dvs_event_synchronize.argtypes = [vp]
    L.dvs_event_record.argtypes = [vp, vp]
    L.dvs_stream_wait_event.argtypes = [vp, vp]
    L.dvs_orb_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, sz, sz, vp, vp, i32, vp]
    L.dvs_orb_level_block_bytes.argtypes = [vp, i32]; L.dvs_orb_level_block_bytes.restype = sz
    L.dvs_orb_extract_levels_device.argtypes = [vp, vp, i32, i32, i32, sz, sz, C.c_uint32, vp]
    L.dvs_orb_merge_levels_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, vp]
    L.dvs_orb_get_level.argtypes = [vp, i32, i32, i32, vp, i32]
    L.dvs_matcher_create.argtypes = [i32, C.POINTER(vp)]
    L.dvs_matcher_create_on_stream.argtypes = [i32, vp, C.POINTER(vp)]
    L.dvs_match_hamming_sequence_device.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp, vp]
    L.dvs_matcher_destroy.argtypes = [vp]; L.dvs_matcher_destroy.restype = None
    L.dvs_matcher_set_stream.argtypes = [vp, vp]
    L.dvs_matcher_synchronize.argtypes = [vp]
    L.dvs_match_hamming.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.dvs_match_hamming_batch_device.argtypes = [vp, vp, vp, i32, vp, vp, i32, i32, vp, vp]
    L.dvs_match_hamming_thresh.argtypes = [vp, vp, i32, vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.dvs_comm_get_unique_id.argtypes = [vp]
    L.dvs_comm_create.argtypes = [i32, i32, i32, vp, C.POINTER(vp)]
    L.dvs_comm_create_loopback.argtypes = [i32, i32, C.POINTER(vp)]
    L.dvs_comm_destroy.argtypes = [vp]; L.dvs_comm_destroy.restype = None
    L.dvs_comm_create_host.argtypes = [i32, i32, vp, vp, C.POINTER(vp)]
    L.dvs_comm_is_host.argtypes = [vp]
    L.dvs_comm_reset_sequence.argtypes = [vp]
    L.dvs_comm_rank.argtypes = [vp]
    L.dvs_comm_world.argtypes = [vp]
    L.dvs_comm_rccl_version.restype = i32
    L.dvs_boundary_block_bytes.argtypes = [i32]; L.dvs_boundary_block_bytes.restype = sz
    L.dvs_exchange_boundary.argtypes = [vp, vp, vp, vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.dvs_comm_all_gather.argtypes = [vp, vp, vp, vp, sz]
    L.dvs_pipeline_create.argtypes = [C.POINTER(PipelineParams), i32, C.POINTER(vp)]
    L.dvs_pipeline_destroy.argtypes = [vp]; L.dvs_pipeline_destroy.restype = None
    L.dvs_pipeline_attach_comm.argtypes = [vp, vp]
    L.dvs_pipeline_step.argtypes = [vp, vp, vp, i32]
    L.dvs_pipeline_flush.argtypes = [vp]
    L.dvs_pipeline_synchronize.argtypes = [vp]
    L.dvs_pipeline_reset.argtypes = [vp]
    L.dvs_pipeline_steps.argtypes = [vp]; L.dvs_pipeline_steps.restype = C.c_int64
    L.dvs_pipeline_get_set.argtypes = [vp, C.c_int64, C.POINTER(PipelineSet)]
    L.dvs_pipeline_lanes.argtypes = [vp]
    L.dvs_pipeline_nsets.argtypes = [vp]
    L.dvs_pipeline_quadtree_async.argtypes = [vp]
    L.dvs_pipeline_set_serialized.argtypes = [vp, i32]
    L.dvs_pipeline_stage_timing.argtypes = [vp, i32]
    L.dvs_pipeline_get_stage_times.argtypes = [vp, vp, vp, i32]
    L.dvs_find_fundamental_ransac.argtypes = [vp, vp, vp, i32, dbl, dbl, i32, C.c_uint64, vp, vp, C.POINTER(i32)]
    L.dvs_solve_pnp_ransac.argtypes = [vp, vp, vp, i32, vp, i32, dbl, dbl, C.c_uint64, vp, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.dvs_find_fundamental_cv.argtypes = [vp, vp, vp, i32, dbl, dbl, i32, vp, vp, C.POINTER(i32), C.POINTER(i32)]
    L.dvs_find_fundamental_cv_batch.argtypes = [vp, i32, vp, vp, vp, dbl, dbl, i32, vp, vp, vp, vp]
    L.dvs_cv_ransac_subsets.argtypes = [vp, vp, i32, i32, i32, vp, C.POINTER(i32)]
    if hasattr(L, "dvs_ba_create"):
        L.dvs_ba_create.argtypes = [i32, C.POINTER(vp)]
        L.dvs_ba_destroy.argtypes = [vp]; L.dvs_ba_destroy.restype = None
        L.dvs_ba_set_stream.argtypes = [vp, vp]
        L.dvs_ba_synchronize.argtypes = [vp]
        L.dvs_ba_set_problem.argtypes = [vp, i32, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, dbl, dbl, dbl, dbl, dbl, dbl]
        L.dvs_ba_evaluate.argtypes = [vp, vp, vp, vp, vp, vp]
        L.dvs_ba_evaluate_raw.argtypes = [vp, vp, vp, vp, vp]
        L.dvs_ba_normal_equations.argtypes = [vp, vp, vp, vp, vp, vp]
        L.dvs_ba_evaluate_device.argtypes = [vp, i32]
        L.dvs_ba_solve.argtypes = [vp, i32, dbl, dbl, dbl, C.POINTER(BaSummary)]
        L.dvs_ba_solve_device.argtypes = [vp, i32, dbl, dbl, dbl, C.POINTER(BaSummary)]
        L.dvs_ba_get_parameters.argtypes = [vp, vp, vp, vp]
        L.dvs_ba_get_trace.argtypes = [vp, vp, i32, C.POINTER(i32)]
        L.dvs_ba_pose_from_rt.argtypes = [vp, vp, vp, vp]
        L.dvs_ba_pose_to_rt.argtypes = [vp, vp, vp, vp]


def _bind_hooks(L):
    """... and of what only lib/libdvslam_hip_test.so exports beside the dvs_test_* functions: the extractor's scheduling / introspection hooks
    (include/dvslam_hip_test.h; hidden in the product library since round 5)"""
    vp, i32, sz, dbl = C.c_void_p, C.c_int32, C.c_size_t, C.c_double
    L.dvs_orb_use_own_stream.argtypes = [vp]
    L.dvs_orb_set_overlap.argtypes = [vp, i32]
    L.dvs_matcher_use_own_stream.argtypes = [vp]
    L.dvs_orb_hint_next_batch_device.argtypes = [vp, vp]
    L.dvs_orb_set_after_fast_event.argtypes = [vp, vp]
    L.dvs_orb_set_output_event.argtypes = [vp, vp]
    L.dvs_orb_set_defer_outputs.argtypes = [vp, i32]
    L.dvs_orb_set_reuse_guard_event.argtypes = [vp, vp]
    L.dvs_orb_set_async_quadtree.argtypes = [vp, i32]
    L.dvs_orb_set_tail_stream.argtypes = [vp, vp]
    L.dvs_orb_chain_graph_launches.argtypes = [vp]; L.dvs_orb_chain_graph_launches.restype = C.c_int64
    L.dvs_orb_get_candidates.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.dvs_orb_get_level_keypoints.argtypes = [vp, i32, i32, vp, i32, C.POINTER(i32)]
    L.dvs_orb_enable_stage_timing.argtypes = [vp, i32]
    L.dvs_orb_get_stage_times.argtypes = [vp, vp, vp, i32]
    L.dvs_orb_create_single_stream.argtypes = [C.POINTER(OrbParams), i32, C.POINTER(vp)]
    L.dvs_orb_create_on_stream.argtypes = [C.POINTER(OrbParams), i32, vp, C.POINTER(vp)]
    L.dvs_pipeline_extractor.argtypes = [vp]; L.dvs_pipeline_extractor.restype = vp
    L.dvs_pipeline_matcher.argtypes = [vp]; L.dvs_pipeline_matcher.restype = vp
    L.dvs_pipeline_match_stream.argtypes = [vp]; L.dvs_pipeline_match_stream.restype = vp


def lib():
    """Load the HIP library; raise loudly if it is missing (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(f"{SO_PATH} not found: build it with `make -C {_PKG}` (hipcc --offload-arch=gfx950); "
                           "dvslam_amd has no CPU fallback")
    L = C.CDLL(SO_PATH)
    _bind(L)
    _lib = L
    return L


TEST_SO_PATH = os.path.join(os.path.dirname(SO_PATH), "libdvslam_hip_test.so")
_test_lib = None


def test_lib():
    """lib/libdvslam_hip_test.so: the same sources built with -DDVS_TEST_HOOKS (include/dvslam_hip_test.h): the whole product ABI plus the
    dvs_test_* functions and the extractor's scheduling / introspection hooks, which the product library does not export.  Wrapper objects
    built with hooks=True make ALL their calls through it."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    if not os.path.exists(TEST_SO_PATH):
        subprocess.check_call(["make", "-s", "-j4", "-C", _PKG, "test-lib"])
    L = C.CDLL(TEST_SO_PATH)
    _bind(L)
    _bind_hooks(L)
    vp, i32, dbl = C.c_void_p, C.c_int32, C.c_double
    L.dvs_test_stream_delay.argtypes = [vp, i32]
    L.dvs_test_sort_nodes.argtypes = [vp, vp, i32, vp]; L.dvs_test_sort_nodes.restype = None
    L.dvs_test_sort_nodes_ranked.argtypes = [vp, vp, i32, vp]; L.dvs_test_sort_nodes_ranked.restype = None
    L.dvs_test_sort_nodes_device.argtypes = [vp, vp, i32, vp]; L.dvs_test_sort_nodes_device.restype = C.c_int
    L.dvs_test_quartic_roots.argtypes = [dbl, dbl, dbl, dbl, dbl, vp]
    L.dvs_test_p3p.argtypes = [vp, vp, vp]
    L.dvs_test_sincosf.argtypes = [C.c_float, vp, vp]; L.dvs_test_sincosf.restype = None
    L.dvs_test_geometry.argtypes = [C.POINTER(OrbParams), i32, i32, vp, vp, vp, vp, vp, vp]
    L.dvs_test_retain_best_host.argtypes = [vp, i32, i32, vp, C.POINTER(i32)]; L.dvs_test_retain_best_host.restype = None
    L.dvs_test_retain_best_device.argtypes = [vp, i32, i32, vp, C.POINTER(i32)]
    _test_lib = L
    return L


def check(code):
    if code != 0:
        msgs = [L.dvs_last_error().decode(errors="replace") for L in (_lib, _test_lib) if L is not None]
        raise DvsError(code, " | ".join(m for m in msgs if m) or "(no message)")


def device_count():
    return lib().dvs_device_count()


def kernel_source_digest():
    """sha256 (first 16 hex digits) over EVERY source of the library (csrc/*.hip, *.h, *.inc, in name order) — stamps profile data
    (profiles/pmc_traffic.json) with the code it was collected on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_PKG, "csrc")
    for name in sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h", ".inc"))):
        h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def stream_create(device=0, high_priority=False):
    """raw hipStream_t (int) created by the library (wrap with torch.cuda.ExternalStream when torch should use it)"""
    out = C.c_void_p()
    check(lib().dvs_stream_create(device, int(high_priority), C.byref(out)))   # True / 1: highest, -1: lowest, False / 0: default
    return int(out.value)


def event_create(device=0):
    """raw hipEvent_t (int), timing disabled"""
    out = C.c_void_p()
    check(lib().dvs_event_create(device, C.byref(out)))
    return int(out.value)


def timing_event_create(device=0):
    out = C.c_void_p()
    check(lib().dvs_event_create_timing(device, C.byref(out)))
    return int(out.value)


def event_elapsed_ms(start, stop):
    ms = C.c_float()
    check(lib().dvs_event_elapsed_ms(start, stop, C.byref(ms)))
    return float(ms.value)


def stream_synchronize(stream):
    check(lib().dvs_stream_synchronize(stream))


def stream_destroy(stream):
    check(lib().dvs_stream_destroy(stream))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DeviceBuffer:
    """Raw HBM allocation through the C-ABI (keeps tests/bench free of any torch dependency)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(lib().dvs_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib().dvs_memcpy_h2d(self.device, self.ptr, ptr(arr), arr.nbytes))
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib().dvs_memcpy_d2h(self.device, ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().dvs_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
