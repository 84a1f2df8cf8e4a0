"""N1 / N2 rows: BGR->gray, depth filter, distance filter, back-projection, backend association.
CPU: known-answers of the oracle restatements; GPU: HIP entry points vs the oracle — bit-exact (integer / index work,
doubles produced by identically ordered operations)."""
import numpy as np
import pytest
from dvslam_amd import synth


def _scene(seed=0, n=700):
    rng = np.random.default_rng(seed)
    from oracle_bindings import KP_DTYPE
    kps = np.zeros(n, KP_DTYPE)
    kps["x"] = rng.uniform(0, 639.49, n).astype(np.float32); kps["y"] = rng.uniform(0, 479.49, n).astype(np.float32)
    kps["x"][:5] = [10.5, 11.5, 100.49999, 0.0, 639.4]          # .5 cases: std::round goes away from zero
    kps["response"] = rng.integers(7, 200, n); kps["octave"] = rng.integers(0, 8, n); kps["class_id"] = -1
    desc = synth.make_descriptors(n, seed + 1)
    depth = rng.integers(0, 4000, (480, 640)).astype(np.uint16)
    depth[::7, ::5] = 0; depth[3, 3] = 300; depth[4, 4] = 3000; depth[5, 5] = 299; depth[6, 6] = 3001
    return kps, desc, depth


def test_oracle_known_answers(oracle):
    bgr = np.zeros((2, 3, 3), np.uint8)
    bgr[0, 0] = [255, 255, 255]; bgr[0, 1] = [255, 0, 0]; bgr[0, 2] = [0, 255, 0]; bgr[1, 0] = [0, 0, 255]; bgr[1, 1] = [10, 20, 30]
    g = oracle.bgr_to_gray(bgr, 0)
    assert g[0, 0] == 255 and g[0, 1] == 29 and g[0, 2] == 150 and g[1, 0] == 76          # 0.114 / 0.587 / 0.299 weights
    assert g[1, 1] == (10 * 3735 + 20 * 19235 + 30 * 9798 + 16384) >> 15
    assert oracle.bgr_to_gray(bgr, 1)[1, 1] == (10 * 1868 + 20 * 9617 + 30 * 4899 + 8192) >> 14
    kps, desc, depth = _scene()
    ok, od, oi = oracle.filter_depth(kps, desc, depth)
    x = np.floor(np.abs(kps["x"]) + 0.5).astype(int); y = np.floor(np.abs(kps["y"]) + 0.5).astype(int)
    inside = (x < 640) & (y < 480)
    d = depth[np.minimum(y, 479), np.minimum(x, 639)].astype(np.float32) * np.float32(0.001)
    keep = inside & ~((d < np.float32(0.3)) | (d > np.float32(3.0)))
    assert (oi == np.nonzero(keep)[0]).all() and (od == desc[keep]).all() and ok.tobytes() == kps[keep].tobytes()
    idx = np.arange(6, dtype=np.int32)[::-1].copy(); dist = np.array([49, 50, 0, 256, 51, 12], np.int32)
    assert oracle.filter_matches(idx, dist).tolist() == [[0, 5, 49], [2, 3, 0], [5, 0, 12]]


@pytest.mark.gpu
def test_glue_parity(gpu, oracle):
    from dvslam_amd import FrontendGlue
    g = FrontendGlue()
    bgr, depth16 = synth.make_rgbd(0, cols=642, rows=481)
    rng = np.random.default_rng(3)
    bgr = np.ascontiguousarray(rng.integers(0, 256, bgr.shape, dtype=np.uint8))            # arbitrary colours, odd width
    for variant in (0, 1):
        assert (g.bgr_to_gray(bgr, variant) == oracle.bgr_to_gray(bgr, variant)).all()
    big = np.zeros((100, 50 * 3 + 7), np.uint8); view = big[:, 2:2 + 150].reshape(100, 50, 3)  # unaligned rows
    view[:] = rng.integers(0, 256, (100, 50, 3))
    assert (g.bgr_to_gray(view) == oracle.bgr_to_gray(np.ascontiguousarray(view))).all()
    kps, desc, depth = _scene(5, 2011)
    a = g.filter_depth(kps, desc, depth); b = oracle.filter_depth(kps, desc, depth)
    assert a[0].tobytes() == b[0].tobytes() and (a[1] == b[1]).all() and (a[2] == b[2]).all() and 0 < len(b[2]) < len(kps)
    e = g.filter_depth(kps[:0], desc[:0], depth)
    assert len(e[0]) == 0
    idx = rng.integers(0, 2000, 2024).astype(np.int32); dist = rng.integers(0, 120, 2024).astype(np.int32)
    assert (g.filter_matches(idx, dist) == oracle.filter_matches(idx, dist)).all()
    R = np.array([[0.9975, -0.0499, 0.05], [0.0524, 0.9974, -0.0498], [-0.0474, 0.0523, 0.9975]]); t = np.array([0.1, -0.2, 0.05])
    w1, i1 = g.backproject(kps, depth, 615.5, 616.25, 320.1, 241.3, R, t); w2, i2 = oracle.backproject(kps, depth, 615.5, 616.25, 320.1, 241.3, R, t)
    assert (i1 == i2).all() and w1.tobytes() == w2.tobytes() and len(i2) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nobs,nlm,seed", [(1, 1, 0), (300, 1500, 1), (1200, 4000, 2), (64, 0, 3)])
def test_association_parity(gpu, oracle, nobs, nlm, seed):
    """backend.cpp:1064-1120 on a database snapshot: Hamming gate 50, reprojection gate 5 px, first-in-order ties"""
    from dvslam_amd import FrontendGlue
    rng = np.random.default_rng(seed)
    lm_desc = synth.make_descriptors(max(nlm, 1), 50 + seed)[:nlm]
    lm_xyz = np.stack([rng.uniform(-2, 2, nlm), rng.uniform(-1.5, 1.5, nlm), rng.uniform(2, 6, nlm)], axis=1).astype(np.float32)
    obs_desc = synth.make_descriptors(nobs, 60 + seed)
    R = np.eye(3); t = np.array([0.05, 0.02, -0.1]); fx = fy = 600.0; cx, cy = 320.0, 240.0
    obs_px = rng.uniform(0, 640, (nobs, 2)).astype(np.float32)
    for i in range(min(nobs, nlm)):            # make most observations true re-observations with small descriptor / pixel noise
        j = int(rng.integers(0, nlm))
        d = lm_desc[j].copy(); flips = rng.integers(0, 256, int(rng.integers(0, 60)))
        for f in flips: d[f // 8] ^= 1 << (f % 8)
        obs_desc[i] = d
        pc = R.T @ (lm_xyz[j].astype(np.float64) - t)
        obs_px[i] = [fx * pc[0] / pc[2] + cx + rng.normal(0, 2.5), fy * pc[1] / pc[2] + cy + rng.normal(0, 2.5)]
    if nlm > 10 and nobs > 3:                  # exact duplicates in the database: the first one in order must win
        lm_desc[7] = lm_desc[3]; lm_xyz[7] = lm_xyz[3]
    got = FrontendGlue().associate(obs_desc, obs_px, lm_desc, lm_xyz, R, t, fx, fy, cx, cy)
    ref = oracle.associate(obs_desc, obs_px, lm_desc, lm_xyz, R, t, fx, fy, cx, cy) if nlm else np.full(nobs, -1, np.int32)
    assert (got == ref).all()
    if nlm > 100:
        assert (ref >= 0).sum() > 0.2 * min(nobs, nlm) and (ref < 0).sum() > 0


# ---------------------------------------------------------------- Keyframe.msg on the wire (row N3) ------------------------------
def _hand_cdr(stamp, frame_id, kf_id, trans, rot, landmarks, observations):
    """Third, independent statement of the layout, straight from the CDR rules: align every primitive to its size relative to
    the byte after the 4-byte encapsulation header."""
    import struct
    b = bytearray()

    def put(fmt, v):
        size = struct.calcsize(fmt)
        while len(b) % size:
            b.append(0)
        b.extend(struct.pack("<" + fmt, v))

    put("i", stamp[0]); put("I", stamp[1])
    put("I", len(frame_id) + 1); b.extend(frame_id.encode() + b"\0")
    put("Q", kf_id)
    for v in trans: put("d", v)
    for v in rot: put("d", v)
    put("I", len(landmarks))
    for lid, x, y, z in landmarks:
        put("Q", lid); put("d", x); put("d", y); put("d", z)
    put("I", len(observations))
    for lid, u, v, d in observations:
        put("Q", lid); put("d", u); put("d", v); put("I", len(d)); b.extend(bytes(d))
    return bytes([0, 1, 0, 0]) + bytes(b)


def _kf_case(n, seed, all_invalid=False):
    kps, desc, depth = _scene(seed, max(n, 8))
    kps, desc = kps[:n], desc[:n]
    if all_invalid:
        depth[:] = 0
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]) @ np.diag([1.0, 1.0, 1.0])
    t = np.array([0.25, -1.5, 3.0])
    return kps, desc, depth, R, t


def test_keyframe_cdr_oracle_known_answer_and_host_unpack(oracle, hiplib):
    from oracle_bindings import KP_DTYPE
    from dvslam_amd.glue import unpack_keyframe
    from dvslam_amd import DvsError
    kps = np.zeros(3, KP_DTYPE)
    kps["x"] = [10.0, 30.25, 50.0]; kps["y"] = [20.0, 40.5, 60.0]
    depth = np.zeros((100, 100), np.uint16)
    depth[20, 10] = 1500; depth[41, 30] = 100; depth[60, 50] = 2000       # keypoint 1: 0.1 m -> rejected; (40.5 rounds to 41)
    desc = (np.arange(96, dtype=np.uint8).reshape(3, 32) * 3 + 1).astype(np.uint8)
    R = np.eye(3); t = np.array([1.0, 2.0, 3.0]); q = (0.0, 0.0, 0.0, 1.0)
    fx = fy = np.float32(500.0); cx = np.float32(5.0); cy = np.float32(15.0)

    def world(i):
        z = np.float32(depth[int(np.floor(kps["y"][i] + 0.5)), int(np.floor(kps["x"][i] + 0.5))]) * np.float32(0.001)
        x = (kps["x"][i] - cx) * z / fx; y = (kps["y"][i] - cy) * z / fy
        return (float(x) + 1.0, float(y) + 2.0, float(z) + 3.0)

    for fid in ["camera_link", "", "abc", "abcd", "abcdefg"]:            # every alignment phase of the uint64 behind the string
        want = _hand_cdr((7, 99), fid, 42, t, q, [(0,) + world(0), (2,) + world(2)],
                         [(0, 10.0, 20.0, desc[0]), (2, 50.0, 60.0, desc[2])])
        got, m = oracle.publish_keyframe(kps, desc, depth, 500.0, 500.0, 5.0, 15.0, R, t, (7, 99), fid, 42, q)
        assert m == 2 and got == want, fid
        u = unpack_keyframe(got)                                            # the library's subscriber-side parser (host code)
        o = oracle.unpack_keyframe(got)
        assert u["frame_id"] == fid == o["frame_id"] and u["keyframe_id"] == 42 and u["stamp"] == (7, 99)
        for k in ("translation", "rotation_xyzw", "landmark_ids", "landmark_xyz", "obs_landmark_ids", "obs_pixels", "obs_desc"):
            assert np.array_equal(u[k], o[k]), k
        assert list(u["landmark_ids"]) == [0, 2] and np.array_equal(u["obs_desc"][1], desc[2])
    empty, m0 = oracle.publish_keyframe(kps[:0], desc[:0], depth, 500.0, 500.0, 5.0, 15.0, R, t, (1, 2), "camera_link", 3, q)
    assert m0 == 0 and empty == _hand_cdr((1, 2), "camera_link", 3, t, q, [], [])
    assert len(unpack_keyframe(empty)["landmark_ids"]) == 0
    with pytest.raises(DvsError):
        unpack_keyframe(got[:-5])                                           # truncated descriptor
    with pytest.raises(DvsError):
        unpack_keyframe(b"\x00\x00\x00\x00" + got[4:])                      # big-endian flag: refused
    assert hiplib.dvs_keyframe_cdr_capacity(b"camera_link", 2) == len(_hand_cdr((7, 99), "camera_link", 42, t, q, [(0, 0, 0, 0)] * 2, [(0, 0, 0, desc[0])] * 2))


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,fid,all_invalid", [(700, 0, "camera_link", False), (1, 3, "a", False), (0, 1, "camera_link", False),
                                                    (257, 5, "", False), (300, 2, "camera_link", True), (2024, 9, "optical_frame_12", False)])
def test_publish_keyframe_parity(gpu, oracle, n, seed, fid, all_invalid):
    from dvslam_amd import FrontendGlue
    from dvslam_amd.glue import unpack_keyframe
    kps, desc, depth, R, t = _kf_case(n, seed, all_invalid)
    q = (0.0, 0.0, np.sqrt(0.5), np.sqrt(0.5))
    g = FrontendGlue()
    got, m = g.publish_keyframe(kps, desc, depth, 600.0, 601.0, 320.5, 240.25, R, t, (12, 345678), fid, 77, q)
    want, m2 = oracle.publish_keyframe(kps, desc, depth, 600.0, 601.0, 320.5, 240.25, R, t, (12, 345678), fid, 77, q)
    assert m == m2 and got == want
    u = unpack_keyframe(got)
    w, oi = oracle.backproject(kps, depth, 600.0, 601.0, 320.5, 240.25, R, t) if n else (np.zeros((0, 3)), np.zeros(0, np.int32))
    assert np.array_equal(u["landmark_ids"], oi.astype(np.uint64)) and np.array_equal(u["landmark_xyz"], w)
    assert np.array_equal(u["obs_desc"], desc[oi]) and np.array_equal(u["obs_pixels"][:, 0], kps["x"][oi].astype(np.float64))


# ---------------------------------------------------------------- Harris score (row N4) ---------------------------------------
def _harris_numpy(img, xs, ys, bs=7, k=np.float32(0.04)):
    """independent statement: whole-image integer gradient maps, box sums, float32 expression in the source's order"""
    I = img.astype(np.int64)
    Ix = np.zeros_like(I); Iy = np.zeros_like(I)
    Ix[1:-1, 1:-1] = (I[1:-1, 2:] - I[1:-1, :-2]) * 2 + (I[:-2, 2:] - I[:-2, :-2]) + (I[2:, 2:] - I[2:, :-2])
    Iy[1:-1, 1:-1] = (I[2:, 1:-1] - I[:-2, 1:-1]) * 2 + (I[2:, :-2] - I[:-2, :-2]) + (I[2:, 2:] - I[:-2, 2:])
    r = bs // 2
    out = np.zeros(len(xs), np.float32)
    scale = np.float32(1.0) / (np.float32(4 * bs) * np.float32(255.0))
    s4 = scale * scale * scale * scale
    for i, (x, y) in enumerate(zip(xs, ys)):
        if not (x - r - 1 >= 0 and y - r - 1 >= 0 and x - r + bs <= img.shape[1] - 1 and y - r + bs <= img.shape[0] - 1):
            continue
        wx = Ix[y - r:y - r + bs, x - r:x - r + bs]; wy = Iy[y - r:y - r + bs, x - r:x - r + bs]
        a = np.float32(int((wx * wx).sum())); b = np.float32(int((wy * wy).sum())); c = np.float32(int((wx * wy).sum()))
        s = a + b
        out[i] = (a * b - c * c - (k * s) * s) * s4
    return out


def test_harris_oracle_known_answers(oracle):
    ramp = np.tile(np.arange(64, dtype=np.uint8), (48, 1))                 # I = x: Ix = 8, Iy = 0 everywhere
    r = oracle.harris_responses(ramp, [20, 30], [20, 10])
    a = np.float32(49 * 64)
    scale = np.float32(1.0) / (np.float32(28) * np.float32(255.0))
    want = (a * np.float32(0) - np.float32(0) - (np.float32(0.04) * a) * a) * (scale * scale * scale * scale)
    assert r[0] == want == r[1] and want < 0
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (90, 120)).astype(np.uint8)
    img[30:60, 40:80] = 200                                                  # a real corner
    xs = np.concatenate([rng.integers(0, 120, 300), [40, 79, 3, 4, 115, 116]]); ys = np.concatenate([rng.integers(0, 90, 300), [30, 59, 3, 4, 85, 86]])
    for bs in (7, 3, 8, 2):
        assert np.array_equal(oracle.harris_responses(img, xs, ys, bs), _harris_numpy(img, xs, ys, bs)), bs
    assert oracle.harris_responses(img, [40], [30])[0] > oracle.harris_responses(img, [60], [45])[0] == 0.0  # corner vs flat interior


@pytest.mark.gpu
def test_harris_parity(gpu, oracle):
    from dvslam_amd import FrontendGlue
    g = FrontendGlue()
    frame = synth.make_frame(3, cols=640, rows=480)
    rng = np.random.default_rng(8)
    xs = np.concatenate([rng.integers(0, 640, 3000), [0, 3, 4, 635, 636, 639]]); ys = np.concatenate([rng.integers(0, 480, 3000), [0, 3, 4, 475, 476, 479]])
    for bs in (7, 5, 8, 1):
        got = g.harris_responses(frame, xs, ys, bs); want = oracle.harris_responses(frame, xs, ys, bs)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), bs
    assert (g.harris_responses(frame, xs, ys) != 0).sum() > 2500
    view = frame[:, :333]                                                      # rows wider than the image (step != cols)
    assert np.array_equal(g.harris_responses(view, xs // 2, ys), oracle.harris_responses(np.ascontiguousarray(view), xs // 2, ys))


@pytest.mark.gpu
def test_out_of_image_keypoints_are_dropped_not_faulted(gpu, oracle):
    """keypoints whose rounded position lies outside the depth image are undefined behaviour in the reference (cv::Mat::at);
    the GPU entry points drop them instead of reading out of bounds"""
    from dvslam_amd import FrontendGlue
    from dvslam_amd.glue import unpack_keyframe
    kps, desc, depth = _scene(11, 64)
    depth[:] = 1500
    kps["x"][:4] = [-5.0, 100000.0, 10.0, 639.6]; kps["y"][:4] = [10.0, 10.0, -0.6, 479.6]
    g = FrontendGlue()
    R = np.eye(3); t = np.zeros(3)
    w, oi = g.backproject(kps, depth, 600.0, 600.0, 320.0, 240.0, R, t)   # ONE policy on every entry point: dropped
    assert len(oi) == 60 and list(oi[:2]) == [4, 5]
    w2, oi2 = g.backproject(kps[4:], depth, 600.0, 600.0, 320.0, 240.0, R, t)
    assert len(oi2) == 60 and (w == w2).all()
    payload, m = g.publish_keyframe(kps, desc, depth, 600.0, 600.0, 320.0, 240.0, R, t)
    assert m == 60 and list(unpack_keyframe(payload)["landmark_ids"][:2]) == [4, 5]
