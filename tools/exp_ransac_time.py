"""per-call latency of the tracking stages' C-ABI entry points (host buffers in / out).  usage: python tools/exp_ransac_time.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "dynamic-visual-slam_amd"))
import numpy as np
import dvslam_amd
g = dvslam_amd.FrontendGlue()
rng = np.random.default_rng(5)
n = 600
X = np.concatenate([rng.uniform(-1, 1, (n, 2)), rng.uniform(1.2, 2.5, (n, 1))], 1)
f, cx, cy = 600.0, 320.0, 240.0
def proj(X, t):
    Y = X + t
    return np.stack([f * Y[:, 0] / Y[:, 2] + cx, f * Y[:, 1] / Y[:, 2] + cy], 1)
p1 = proj(X, np.zeros(3)).astype(np.float32); p2 = proj(X, np.array([0.03, 0.01, 0.02])).astype(np.float32)
p2[:60] += rng.uniform(-30, 30, (60, 2)).astype(np.float32)
K4 = np.array([f, f, cx, cy])
for _ in range(5):
    g.find_fundamental_ransac(p1, p2, 2.0, 0.99, 1000, 7); g.solve_pnp_ransac(X.astype(np.float32), p2, K4, 100, 4.0, 0.99, 9)
for name, fn in [("fundamental", lambda s: g.find_fundamental_ransac(p1, p2, 2.0, 0.99, 1000, s)),
                 ("pnp", lambda s: g.solve_pnp_ransac(X.astype(np.float32), p2, K4, 100, 4.0, 0.99, s))]:
    ts = []
    for s in range(200):
        t0 = time.perf_counter(); fn(s); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: median {1e6 * ts[100]:.1f} us, p10 {1e6 * ts[20]:.1f}, p90 {1e6 * ts[180]:.1f}")
