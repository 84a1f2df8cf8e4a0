"""diagnostics: per-phase host timestamps of the first solves (DVS_LM_POLL_DEBUG=1).  usage: DVS_LM_POLL_DEBUG=1 python tools/exp_lm_debug.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "dynamic-visual-slam_amd"))
import dvslam_amd
from dvslam_amd import synth
P = synth.make_ba_problem(K=10, L=2000, seed=42)
gd = dvslam_amd.BAProblem(P, device=0); gd.solve_device(20)
for r in range(2):
    gd = dvslam_amd.BAProblem(P, device=0); gd.solve_device(0)
    sys.stderr.write(f"=== timed solve {r}\n"); sys.stderr.flush()
    t0 = time.perf_counter(); sd = gd.solve_device(20); dt = time.perf_counter() - t0
    sys.stderr.write(f"=== {1e3 * dt:.3f} ms\n"); sys.stderr.flush()
