"""Deterministic synthetic inputs for tests and bench.py.

Frames: a procedurally textured planar "world" canvas (rectangles + discs over a smooth gradient)
viewed through a camera that translates (+3,+1) px and rotates 0.5 deg per frame (nearest-neighbour
warp, integer-exact), plus per-frame uniform sensor noise of +-3 % full scale.  RGB = gray x3,
depth = 1500 mm plane with a 50 mm checker.  BA problems: see make_ba_problem().
Only numpy's PCG64 integer/uniform streams are used, so results are identical on every machine.
"""
import numpy as np

_CANVAS_CACHE = {}


def _canvas(seed: int, cw: int, ch: int) -> np.ndarray:
    key = (seed, cw, ch)
    if key in _CANVAS_CACHE:
        return _CANVAS_CACHE[key]
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:ch, 0:cw]
    img = (96 + 48 * np.sin(xx / 173.0) * np.cos(yy / 131.0)).astype(np.int32)
    n_rect = int(400 * (cw * ch) / (1280 * 720))
    n_disc = int(200 * (cw * ch) / (1280 * 720))
    for _ in range(n_rect):
        w = int(rng.integers(6, 140)); h = int(rng.integers(6, 140))
        x = int(rng.integers(-20, cw)); y = int(rng.integers(-20, ch))
        img[max(y, 0):max(y + h, 0), max(x, 0):max(x + w, 0)] = int(rng.integers(0, 256))
    for _ in range(n_disc):
        r = int(rng.integers(4, 50)); x = int(rng.integers(0, cw)); y = int(rng.integers(0, ch))
        v = int(rng.integers(0, 256))
        y0, y1, x0, x1 = max(y - r, 0), min(y + r + 1, ch), max(x - r, 0), min(x + r + 1, cw)
        sub = img[y0:y1, x0:x1]
        m = (yy[y0:y1, x0:x1] - y) ** 2 + (xx[y0:y1, x0:x1] - x) ** 2 <= r * r
        sub[m] = v
    out = np.clip(img, 0, 255).astype(np.uint8)
    _CANVAS_CACHE[key] = out
    return out


def make_frame(t: int = 0, cols: int = 1280, rows: int = 720, seed: int = 1234, noise: int = 8) -> np.ndarray:
    """Gray uint8 frame number t of the synthetic sequence (C-contiguous rows x cols)."""
    margin = 260 + 4 * 64
    cw, ch = cols + 2 * margin, rows + 2 * margin
    canvas = _canvas(seed, cw, ch)
    th = np.deg2rad(0.5 * t)
    c, s = np.cos(th), np.sin(th)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    dx, dy = xx - cols / 2.0, yy - rows / 2.0
    sx = np.rint(c * dx - s * dy + cols / 2.0 + margin + 3 * t).astype(np.int64)
    sy = np.rint(s * dx + c * dy + rows / 2.0 + margin + 1 * t).astype(np.int64)
    np.clip(sx, 0, cw - 1, out=sx); np.clip(sy, 0, ch - 1, out=sy)
    frame = canvas[sy, sx].astype(np.int32)
    if noise > 0:
        rng = np.random.Generator(np.random.PCG64(seed * 7919 + t))
        frame += rng.integers(-noise, noise + 1, size=frame.shape, dtype=np.int32)
    return np.ascontiguousarray(np.clip(frame, 0, 255).astype(np.uint8))


def traj_state(t: float):
    """Bounded camera trajectory of the 1000-frame replay sequence (BASELINE configs[4]): roll angle [rad] and image-plane offset
    [px] of frame t.  <= 7 px and <= 0.26 deg of motion per frame, |offset| <= 240 px, |roll| <= 12 deg, so every frame stays on
    the canvas however long the sequence runs (make_frame's straight-line motion leaves it after ~170 frames)."""
    th = np.deg2rad(12.0) * np.sin(2 * np.pi * t / 300.0)
    ox = 180.0 * np.sin(2 * np.pi * t / 450.0) + 60.0 * np.sin(2 * np.pi * t / 97.0)
    oy = 110.0 * np.sin(2 * np.pi * t / 333.0 + 0.7)
    return th, ox, oy


def make_traj_frame(t: int, cols: int = 640, rows: int = 480, seed: int = 1234, noise: int = 8) -> np.ndarray:
    """frame t of the bounded-trajectory sequence: the canvas sampled at R(th) (p - c) + c + (ox, oy) (nearest neighbour, so the
    generator is integer-exact), plus per-frame uniform sensor noise"""
    margin = 260 + 4 * 64
    cw, ch = cols + 2 * margin, rows + 2 * margin
    canvas = _canvas(seed, cw, ch)
    th, ox, oy = traj_state(t)
    c, s = np.cos(th), np.sin(th)
    yy, xx = np.mgrid[0:rows, 0:cols].astype(np.float64)
    dx, dy = xx - cols / 2.0, yy - rows / 2.0
    sx = np.rint(c * dx - s * dy + cols / 2.0 + margin + ox).astype(np.int64)
    sy = np.rint(s * dx + c * dy + rows / 2.0 + margin + oy).astype(np.int64)
    np.clip(sx, 0, cw - 1, out=sx); np.clip(sy, 0, ch - 1, out=sy)
    frame = canvas[sy, sx].astype(np.int32)
    if noise > 0:
        rng = np.random.Generator(np.random.PCG64(seed * 7919 + 104729 + t))
        frame += rng.integers(-noise, noise + 1, size=frame.shape, dtype=np.int32)
    return np.ascontiguousarray(np.clip(frame, 0, 255).astype(np.uint8))


def traj_pose(t: float, f: float, z0: float):
    """closed-form camera-to-world pose (R, T) of frame t for a fronto-parallel plane at depth z0 and focal length f [px]: the
    pixel p sees the canvas point R(th) (p - c) + offset, i.e. world = Rz(th) x_cam + (ox, oy, 0) z0 / f"""
    th, ox, oy = traj_state(t)
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]]), np.array([ox * z0 / f, oy * z0 / f, 0.0])


def make_rgbd(t: int = 0, cols: int = 1280, rows: int = 720, seed: int = 1234):
    """(bgr uint8 HxWx3, depth uint16 HxW in mm) as the frontend receives them (FE:1076-1077)."""
    g = make_frame(t, cols, rows, seed)
    bgr = np.repeat(g[:, :, None], 3, axis=2)
    yy, xx = np.mgrid[0:rows, 0:cols]
    depth = (1500 + 50 * (((xx // 64) + (yy // 64)) & 1)).astype(np.uint16)
    return bgr, depth


def make_descriptors(n: int, seed: int) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def _quat_from_axis_angle(axis, ang):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])


def _quat_mul(a, b):
    w1, x1, y1, z1 = a; w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _quat_rot(q, p):
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return p @ R.T


def make_ba_problem(K: int = 10, L: int = 2000, seed: int = 42, pixel_noise: float = 1.0, outlier_frac: float = 0.02,
                    pose_noise=(0.01, np.deg2rad(0.5)), lm_noise: float = 0.02, visibility: float = 1.0):
    """Sliding-window BA problem in the optimiser's own parameterisation (world->camera quaternion
    (w,x,y,z) + translation, BA.hpp:92-165): K keyframes on a 1 m arc looking at the landmark
    centroid, L landmarks in [-2,2]x[-1.5,1.5]x[2,6] m, every landmark seen in every keyframe.
    Returns a dict of float64/int32 arrays (initial = perturbed values, gt_* = ground truth)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fx = fy = 900.0; cx, cy = 640.0, 360.0
    X = np.stack([rng.uniform(-2, 2, L), rng.uniform(-1.5, 1.5, L), rng.uniform(2, 6, L)], axis=1)
    centroid = np.array([0.0, 0.0, 4.0])
    q_gt = np.zeros((K, 4)); t_gt = np.zeros((K, 3))
    for k in range(K):
        a = (k / max(K - 1, 1) - 0.5) * 1.0 / 4.0          # 1 m of arc on a 4 m radius
        cam_c = centroid + 4.0 * np.array([np.sin(a), 0.0, -np.cos(a)])
        q = _quat_from_axis_angle([0, 1, 0], -a)              # world->camera rotation
        q_gt[k] = q
        t_gt[k] = -_quat_rot(q, cam_c[None, :])[0]
    cam_idx, lm_idx, uv = [], [], []
    for k in range(K):
        pc = _quat_rot(q_gt[k], X) + t_gt[k]
        u = fx * pc[:, 0] / pc[:, 2] + cx + rng.normal(0, pixel_noise, L)
        v = fy * pc[:, 1] / pc[:, 2] + cy + rng.normal(0, pixel_noise, L)
        vis = rng.uniform(0, 1, L) < visibility
        out = rng.uniform(0, 1, L) < outlier_frac
        u = np.where(out, rng.uniform(0, 1280, L), u); v = np.where(out, rng.uniform(0, 720, L), v)
        sel = np.nonzero(vis)[0]
        cam_idx.append(np.full(sel.size, k, np.int32)); lm_idx.append(sel.astype(np.int32))
        uv.append(np.stack([u[sel], v[sel]], axis=1))
    q0 = q_gt.copy(); t0 = t_gt.copy()
    for k in range(1, K):
        dq = _quat_from_axis_angle(rng.normal(0, 1, 3), rng.normal(0, pose_noise[1]))
        q0[k] = _quat_mul(dq, q_gt[k]); t0[k] = t_gt[k] + rng.normal(0, pose_noise[0], 3)
    X0 = X + rng.normal(0, lm_noise, X.shape)
    pose_fixed = np.zeros(K, np.uint8); pose_fixed[0] = 1
    return dict(K=K, L=L, q=np.ascontiguousarray(q0), t=np.ascontiguousarray(t0), X=np.ascontiguousarray(X0),
                cam_idx=np.concatenate(cam_idx), lm_idx=np.concatenate(lm_idx), uv=np.ascontiguousarray(np.concatenate(uv)),
                pose_fixed=pose_fixed, lm_fixed=np.zeros(L, np.uint8), fx=fx, fy=fy, cx=cx, cy=cy, sigma=1.0,
                huber=1.345, gt_q=q_gt, gt_t=t_gt, gt_X=X)
