import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import dvslam_amd
from dvslam_amd import synth
P = synth.make_ba_problem(K=10, L=2000, seed=42)
for M in (1, 2, 4, 8, 16, 32):
    gs = [dvslam_amd.BAProblem(P) for _ in range(M)]
    for g in gs: g.evaluate_device(20)
    for g in gs: g.synchronize()
    iters = 200
    t0 = time.perf_counter()
    for _ in range(iters // 10):
        for g in gs: g.evaluate_device(10)
    for g in gs: g.synchronize()
    dt = time.perf_counter() - t0
    print(f"windows in flight {M:3d}: {M * iters / dt:10.0f} evals/s   ({1e6 * dt / iters:.1f} us per round)", flush=True)
    for g in gs: g.close()
