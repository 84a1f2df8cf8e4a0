"""BA evaluation, 64 windows per launch pair (the bench's batched configuration) — run under rocprofv3 --kernel-trace --stats"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import numpy as np
import dvslam_amd
from dvslam_amd import synth
from bench import _replicate_ba
P = synth.make_ba_problem(K=10, L=2000, seed=42)
g = dvslam_amd.BAProblem(_replicate_ba(P, 64))
g.evaluate_device(10); g.synchronize()
t0 = time.perf_counter(); g.evaluate_device(100); g.synchronize(); dt = time.perf_counter() - t0
print("us per window", 1e6 * dt / 100 / 64)
