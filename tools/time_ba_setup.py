"""per-window cost of the BA adapter path: dvs_ba_set_problem + dvs_ba_solve_device on one persistent handle, window after window
(SlidingWindowBA::optimize does exactly this).  usage: python tools/time_ba_setup.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import numpy as np
import dvslam_amd
from dvslam_amd import synth
g = None
for K, L in ((10, 2000), (5, 400)):
    P = synth.make_ba_problem(K=K, L=L, seed=42)
    g = dvslam_amd.BAProblem(P)
    g.solve_device(20)
    ts, tp = [], []
    for r in range(9):
        t0 = time.perf_counter(); g.set_problem(P); t1 = time.perf_counter(); s = g.solve_device(20); t2 = time.perf_counter()
        tp.append(t1 - t0); ts.append(t2 - t1)
    print(f"{K} x {L}: set_problem {1e3 * sorted(tp)[4]:.3f} ms, solve_device {1e3 * sorted(ts)[4]:.3f} ms (median of 9, same handle), cost {s.final_cost!r}")
