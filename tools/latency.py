"""Single-frame latency of the host entry points (the reference's live pattern: one frame per callback): extract (+ match vs the
previous frame) through dvs_orb_extract / dvs_match_hamming with host buffers, PCIe included."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import numpy as np
import dvslam_amd
from dvslam_amd import synth
for rows, cols, nf in [(480, 640, 500), (720, 1280, 1000), (720, 1280, 2000)]:
    frames = [synth.make_frame(t, cols=cols, rows=rows) for t in range(8)]
    g = dvslam_amd.ORBextractor(nf, 1.2, 8, 20, 7); m = dvslam_amd.BFMatcher()
    n, k, prev = g(frames[0])
    for f in frames[1:4]:
        n, k, d = g(f); m.match(d, prev); prev = d
    te = tm = 0.0; it = 0
    for rep in range(10):
        for f in frames:
            t0 = time.perf_counter(); n, k, d = g(f); t1 = time.perf_counter(); m.match(d, prev); t2 = time.perf_counter()
            te += t1 - t0; tm += t2 - t1; it += 1; prev = d
    print(f"{cols}x{rows} nfeatures={nf}: extract {1e3 * te / it:.3f} ms, match {1e3 * tm / it:.3f} ms per frame ({n} keypoints)")
