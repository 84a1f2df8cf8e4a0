import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import numpy as np
import dvslam_amd
from dvslam_amd import synth
P = synth.make_ba_problem(K=10, L=2000, seed=42)
def replicate(P, W):
    Q = dict(P)
    K, L = P["K"], P["L"]
    Q["K"], Q["L"] = K * W, L * W
    Q["q"] = np.tile(P["q"], (W, 1)); Q["t"] = np.tile(P["t"], (W, 1)); Q["X"] = np.tile(P["X"], (W, 1))
    Q["cam_idx"] = np.concatenate([P["cam_idx"] + w * K for w in range(W)]).astype(np.int32)
    Q["lm_idx"] = np.concatenate([P["lm_idx"] + w * L for w in range(W)]).astype(np.int32)
    Q["uv"] = np.tile(P["uv"], (W, 1)); Q["pose_fixed"] = np.tile(P["pose_fixed"], W); Q["lm_fixed"] = np.tile(P["lm_fixed"], W)
    return Q
c1 = dvslam_amd.BAProblem(P).evaluate()[0]
for W in (1, 8, 32, 64, 128):
    g = dvslam_amd.BAProblem(replicate(P, W))
    c = g.evaluate()[0]
    assert abs(c - W * c1) <= 1e-9 * abs(W * c1), (c, W * c1)
    g.evaluate_device(10); g.synchronize()
    iters = 100
    t0 = time.perf_counter(); g.evaluate_device(iters); g.synchronize(); dt = time.perf_counter() - t0
    print(f"W={W:4d} windows per launch: {W * iters / dt:12.0f} window-evals/s  ({1e6 * dt / iters:.1f} us per launch pair)", flush=True)
    g.close()
