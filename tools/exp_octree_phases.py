#!/usr/bin/env python3
"""phase timeline of the quad-tree workgroup of (frame 0, level l): DVS_DEBUG=2 stamps (100 MHz wall clock)"""
import os, sys
os.environ["DVS_DEBUG"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import dvslam_amd
from dvslam_amd import synth, _lib
import bench
NAMES = {1: "start", 2: "gather done", 3: "roots done", 4: "sweep: counted", 5: "sweep: scanned", 6: "sweep: rebuilt", 7: "expand list",
         8: "ordered: counted+keys", 9: "ordered: sorted", 10: "ordered: cut", 11: "ordered: rebuilt", 12: "loop done", 13: "end"}
dev = torch.device("cuda", 0); rows, cols, B = 720, 1280, int(sys.argv[1]) if len(sys.argv) > 1 else 64
d_img, _ = bench.make_batches(synth, torch, dev, B, 1, 0, rows, cols, True)
orb = dvslam_amd.ORBextractor(2000, 1.2, 8, 20, 7, device=0, max_batch=B)
if len(sys.argv) > 2 and sys.argv[2] == "alone":
    orb.set_overlap(False)
cap = orb.capacity
k = _lib.DeviceBuffer(B * cap * 28); d = _lib.DeviceBuffer(B * cap * 32); n = _lib.DeviceBuffer(B * 4)
for rep in range(3):
    orb.extract_batch_device(d_img[0].data_ptr(), B, rows, cols, cols, rows * cols, k.ptr, d.ptr, cap, n.ptr)
orb.synchronize()
L = _lib.lib()
for level in (0, 1, 4, 7):
    out = np.zeros(64, np.uint64)
    assert L.dvs_test_octree_stamps(orb._h, level, out.ctypes.data) == 0
    cnt = int(out[0]); st = [(int(v >> np.uint64(56)), int(v & np.uint64(0xFFFFFFFFFFFFFF))) for v in out[1:1 + cnt]]
    t0 = st[0][1]
    print(f"level {level}: {cnt} stamps, total {(st[-1][1] - t0) / 100:.1f} us")
    prev = t0
    for i, t in st:
        print(f"   {NAMES.get(i, i):26s} +{(t - prev) / 100:6.1f} us   @{(t - t0) / 100:7.1f}")
        prev = t
