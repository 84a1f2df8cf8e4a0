#!/usr/bin/env python3
"""quad-tree stage time per pyramid level (level-masked extraction, 64 frames): which workgroups set the kernel's duration"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd")); sys.path.insert(0, ROOT)
import torch
import dvslam_amd
from dvslam_amd import synth, _lib
import bench
dev = torch.device("cuda", 0); rows, cols, B = 720, 1280, 64
d_img, _ = bench.make_batches(synth, torch, dev, B, 1, 0, rows, cols, True)
orb = dvslam_amd.ORBextractor(2000, 1.2, 8, 20, 7, device=0, max_batch=B)
orb.set_overlap(False)
blk = _lib.DeviceBuffer(orb.level_block_bytes(B))
out = {}
for name, mask in [("all", 0xff)] + [(f"L{l}", 1 << l) for l in range(8)]:
    for rep in range(3):
        orb.extract_levels_device(d_img[0].data_ptr(), B, rows, cols, cols, rows * cols, mask, blk.ptr)
    orb.synchronize()
    orb.enable_stage_timing(True)
    for rep in range(10):
        orb.extract_levels_device(d_img[0].data_ptr(), B, rows, cols, cols, rows * cols, mask, blk.ptr)
    orb.synchronize()
    ms, calls = orb.stage_times()
    orb.enable_stage_timing(False)
    out[name] = {k: round(ms[k] / max(calls[k], 1), 4) for k in ms}
    print(name, out[name], flush=True)
