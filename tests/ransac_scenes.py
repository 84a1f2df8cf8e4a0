"""Synthetic two-view / PnP scenes with known ground truth for the robust-estimation tests (tests/test_ransac.py)."""
import numpy as np


def rot(axis, ang):
    a = np.asarray(axis, np.float64); a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def rodrigues_to_R(w):
    th = np.linalg.norm(w)
    return np.eye(3) if th < 1e-15 else rot(w / th, th)


def two_view(n=600, outlier_frac=0.3, noise=0.5, seed=0, planar=False, fx=600.0, fy=600.0, cx=320.0, cy=240.0, rows=480, cols=640):
    """n correspondences between two views of a 3D scene (camera 2 = R, t applied to camera-1 coordinates); the last
    outlier_frac of them (shuffled) are uniform random point pairs.  Returns dict with pts1, pts2, X (camera-1 frame), truth mask,
    R, t, K4."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R = rot(rng.normal(size=3), np.deg2rad(rng.uniform(2, 8)))
    t = rng.normal(size=3) * 0.08
    X = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-0.9, 0.9, n), np.full(n, 1.5) if planar else rng.uniform(1.0, 3.0, n)], axis=1)
    X2 = X @ R.T + t
    p1 = np.stack([fx * X[:, 0] / X[:, 2] + cx, fy * X[:, 1] / X[:, 2] + cy], 1) + rng.normal(0, noise, (n, 2))
    p2 = np.stack([fx * X2[:, 0] / X2[:, 2] + cx, fy * X2[:, 1] / X2[:, 2] + cy], 1) + rng.normal(0, noise, (n, 2))
    truth = np.ones(n, bool)
    no = int(round(outlier_frac * n))
    if no:
        bad = rng.permutation(n)[:no]
        truth[bad] = False
        p2[bad] = np.stack([rng.uniform(0, cols, no), rng.uniform(0, rows, no)], 1)
    return dict(pts1=p1.astype(np.float32), pts2=p2.astype(np.float32), X=X.astype(np.float32), truth=truth, R=R, t=t,
                K4=np.array([fx, fy, cx, cy]))


def sampson_truth_error(F, sc):
    """mean symmetric epipolar distance of the TRUE inliers under F (pixels)"""
    p1 = np.concatenate([sc["pts1"].astype(np.float64), np.ones((len(sc["pts1"]), 1))], 1)
    p2 = np.concatenate([sc["pts2"].astype(np.float64), np.ones((len(sc["pts2"]), 1))], 1)
    l2 = p1 @ F.T; l1 = p2 @ F
    d2 = np.abs((p2 * l2).sum(1)) / np.hypot(l2[:, 0], l2[:, 1]); d1 = np.abs((p1 * l1).sum(1)) / np.hypot(l1[:, 0], l1[:, 1])
    return np.maximum(d1, d2)[sc["truth"]].mean()


def iou(a, b):
    a = np.asarray(a, bool); b = np.asarray(b, bool)
    u = (a | b).sum()
    return 1.0 if u == 0 else (a & b).sum() / u
