#!/usr/bin/env python3
"""Experiment: throughput of N independent extractor handles (own streams) each with batch B."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import numpy as np
import dvslam_amd
from dvslam_amd import synth
from dvslam_amd._lib import DeviceBuffer
rows, cols = 720, 1280
frames = [synth.make_frame(t) for t in range(8)]
def run(nh, B, steps=10):
    hs = []
    for i in range(nh):
        orb = dvslam_amd.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=B)
        mat = dvslam_amd.BFMatcher()
        cap = orb.capacity
        d_img = DeviceBuffer(B * rows * cols).upload(np.stack([frames[j % 8] for j in range(B)]))
        d_k = DeviceBuffer((B + 1) * cap * 28); d_d = DeviceBuffer((B + 1) * cap * 32); d_n = DeviceBuffer((B + 1) * 4)
        d_i = DeviceBuffer(B * cap * 4); d_dd = DeviceBuffer(B * cap * 4)
        dvslam_amd._lib.check(dvslam_amd.lib().dvs_memset(0, d_n.ptr, 0, (B + 1) * 4))
        mat.set_stream(dvslam_amd.lib().dvs_orb_get_stream(orb._h))
        hs.append((orb, mat, cap, d_img, d_k, d_d, d_n, d_i, d_dd))
    def step():
        for orb, mat, cap, d_img, d_k, d_d, d_n, d_i, d_dd in hs:
            orb.extract_batch_device(d_img.ptr, B, rows, cols, cols, rows * cols, d_k.ptr + cap * 28, d_d.ptr + cap * 32, cap, d_n.ptr + 4)
            mat.match_batch_device(d_d.ptr + cap * 32, d_n.ptr + 4, cap, d_d.ptr, d_n.ptr, cap, B, d_i.ptr, d_dd.ptr)
    for _ in range(3): step()
    for h in hs: h[0].synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    for h in hs: h[0].synchronize()
    dt = time.perf_counter() - t0
    print(f"handles={nh} batch={B}: {nh * B * steps / dt:9.0f} frames/s  ({1e3 * dt / steps:.3f} ms/step)", flush=True)
for nh, B in [(1, 64), (2, 32), (2, 64), (4, 16), (4, 32), (1, 128), (1, 16)]:
    run(nh, B)
