"""per-call latency of the tracking stages' C-ABI entry points (host buffers in / out).  usage: python tools/time_ransac.py"""
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
                 ("fundamental_cv", lambda s: g.find_fundamental_cv(p1, p2, 2.0, 0.99, 1000)),
                 ("pnp", lambda s: g.solve_pnp_ransac(X.astype(np.float32), p2, K4, 100, 4.0, 0.99, s))]:
    ts = []
    for s in range(200):
        t0 = time.perf_counter(); fn(s); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"{name}: median {1e6 * ts[100]:.1f} us, p10 {1e6 * ts[20]:.1f}, p90 {1e6 * ts[180]:.1f}")

# the per-frame glue entry points at the replay's sizes (640 x 480 depth, ~1000 keypoints)
from dvslam_amd import synth
gray = synth.make_traj_frame(3, 640, 480) if hasattr(synth, "make_traj_frame") else synth.make_frame(3, cols=640, rows=480)
orb = dvslam_amd.ORBextractor(1000, 1.2, 8, 20, 7)
nk, k, d = orb(gray)
depth = np.full((480, 640), 1500, np.uint16)
for _ in range(5): g.filter_depth(k, d, depth)
ts = []
for s in range(200):
    t0 = time.perf_counter(); g.filter_depth(k, d, depth); ts.append(time.perf_counter() - t0)
ts.sort(); print(f"filter_depth ({nk} keypoints): median {1e6 * ts[100]:.1f} us")
ts = []
for s in range(200):
    t0 = time.perf_counter(); orb(gray); ts.append(time.perf_counter() - t0)
ts.sort(); print(f"extract 640x480/1000: median {1e6 * ts[100]:.1f} us")
