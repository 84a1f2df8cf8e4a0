"""device LM solve (10 KF x 2000 LM) — run under rocprofv3 --kernel-trace --stats to see which kernel dominates"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import dvslam_amd
from dvslam_amd import synth
P = synth.make_ba_problem(K=10, L=2000, seed=42)
g = dvslam_amd.BAProblem(P); g.solve_device(1)
for _ in range(5):
    g = dvslam_amd.BAProblem(P); g.solve_device(0)
    t0 = time.perf_counter(); s = g.solve_device(20); dt = time.perf_counter() - t0
print("ms per solve", 1e3 * dt, "iterations", s.num_iterations)
