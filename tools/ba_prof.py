import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import dvslam_amd
from dvslam_amd import synth
P = synth.make_ba_problem(K=10, L=2000, seed=42)
g = dvslam_amd.BAProblem(P)
g.evaluate_device(50); g.synchronize()
t0 = time.perf_counter(); g.evaluate_device(500); g.synchronize(); dt = time.perf_counter() - t0
print("us per eval", 1e6 * dt / 500)
