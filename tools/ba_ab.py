"""BA timings for A/B runs of two builds (DVSLAM_HIP_SO selects the library): single-window evaluation, 64 windows per launch, device LM.
usage: python tools/ba_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import torch  # noqa: F401
import dvslam_amd
from dvslam_amd import synth
from bench import _replicate_ba
P = synth.make_ba_problem(K=10, L=2000, seed=42)
g = dvslam_amd.BAProblem(P)
g.evaluate_device(50); g.synchronize()
t0 = time.perf_counter(); g.evaluate_device(1000); g.synchronize(); dt1 = (time.perf_counter() - t0) / 1000
gb = dvslam_amd.BAProblem(_replicate_ba(P, 64))
gb.evaluate_device(10); gb.synchronize()
t0 = time.perf_counter(); gb.evaluate_device(100); gb.synchronize(); dtW = (time.perf_counter() - t0) / 100
gd = dvslam_amd.BAProblem(P); gd.solve_device(20)
ts = []
for r in range(7):
    gd = dvslam_amd.BAProblem(P); gd.solve_device(0)
    t0 = time.perf_counter(); sd = gd.solve_device(20); ts.append(time.perf_counter() - t0)
print(os.environ.get("DVSLAM_HIP_SO", "lib"), "single us/eval %.2f" % (1e6 * dt1), "batched us/eval %.3f" % (1e6 * dtW / 64), "LM ms %.3f" % (1e3 * sorted(ts)[3]),
      "cost", repr(sd.final_cost))
