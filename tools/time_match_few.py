"""Run ON THE GPU BOX (under rocprofv3 --kernel-trace --stats for kernel times): the few-jobs match of a frame sequence, 1 / 2 / 4 / 6 / 8 jobs
of 2000 x 2000 through dvs_match_hamming_sequence_device; DVS_MATCH_LDS=0 selects k_match<16, 1> instead of k_match_lds."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT + "/dynamic-visual-slam_amd")
import numpy as np
import dvslam_amd
from dvslam_amd import _lib
rng = np.random.default_rng(1)
cap = 2024
m = dvslam_amd.BFMatcher()
for B in (1, 2, 4, 6, 8):
    desc = rng.integers(0, 256, (B + 1, cap, 32), dtype=np.uint8)
    n = np.full(B + 1, 2000, np.int32)
    d_desc = _lib.DeviceBuffer(desc.nbytes).upload(desc); d_n = _lib.DeviceBuffer(n.nbytes).upload(n)
    d_idx = _lib.DeviceBuffer(B * cap * 4); d_dist = _lib.DeviceBuffer(B * cap * 4)
    L = _lib.lib()
    def run():
        _lib.check(L.dvs_match_hamming_sequence_device(m._h, d_desc.ptr + cap * 32, d_n.ptr + 4, cap, B, d_desc.ptr, d_n.ptr, d_idx.ptr, d_dist.ptr))
    for _ in range(5): run()
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): run()
    m.synchronize()
    print(f"{B} jobs: {1e6 * (time.perf_counter() - t0) / 200:.1f} us per call (back to back)")
