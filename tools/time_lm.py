"""LM solve time on the device (10 KF x 2000 LM), best of a few repeats.  usage: python tools/time_lm.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "dynamic-visual-slam_amd"))
import dvslam_amd
from dvslam_amd import synth
P = synth.make_ba_problem(K=10, L=2000, seed=42)
gd = dvslam_amd.BAProblem(P, device=0); gd.solve_device(20)   # warm-up: a whole solve
ts = []
for r in range(7):
    gd = dvslam_amd.BAProblem(P, device=0); gd.solve_device(0)
    t0 = time.perf_counter(); sd = gd.solve_device(20); ts.append(time.perf_counter() - t0)
print("iterations", sd.num_iterations, "final_cost", repr(sd.final_cost), "ms", [round(1e3 * t, 3) for t in ts])
