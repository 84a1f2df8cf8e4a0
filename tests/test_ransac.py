"""The frontend's robust-estimation stages: cv::findFundamentalMat(FM_RANSAC, 2.0, 0.99) and
cv::solvePnPRansac(100, 4.0, 0.99) as frontend.cpp:635, 911-921, 1146-1147 call them.

Two forms.  The library's own estimators (8-point / P3P over a documented sampler): tolerances are stated per assertion as inlier-set
IoU / pose error (a) against synthetic ground truth, for the CPU oracle and the HIP path alike, and (b) between the HIP path and the
oracle, which share the sampler but not their numerical routines.  And cv::findFundamentalMat as OpenCV 4.x itself runs it
(dvs_find_fundamental_cv: cv::RNG sample sequence, 7-point solver, RANSAC from 15 points on and LMedS below; at the end of this file) — restated from the published algorithm,
unpinned like everything else here (no OpenCV in the image)."""
import numpy as np
import pytest
from dvslam_amd._lib import test_lib as _hooks   # lib/libdvslam_hip_test.so: the dvs_test_* hooks (not in the product library)
import ransac_scenes as rs


# ---------------------------------------------------------------- CPU: the oracle's pieces against independent statements
def test_sampler_is_deterministic_distinct_and_uniform(oracle):
    a = oracle.sample_distinct(7, 3, 100, 8); b = oracle.sample_distinct(7, 3, 100, 8)
    assert (a == b).all() and len(set(a.tolist())) == 8 and a.min() >= 0 and a.max() < 100
    assert (oracle.sample_distinct(7, 4, 100, 8) != a).any() and (oracle.sample_distinct(8, 3, 100, 8) != a).any()
    assert sorted(oracle.sample_distinct(1, 0, 8, 8).tolist()) == list(range(8))          # n == k: a permutation
    hist = np.zeros(20)
    for h in range(4000):
        hist[oracle.sample_distinct(99, h, 20, 3)] += 1
    assert hist.min() > 0.8 * 600 and hist.max() < 1.2 * 600                               # 12 000 draws over 20 bins


def test_quartic_roots_against_numpy(oracle):
    rng = np.random.Generator(np.random.PCG64(3))
    for _ in range(200):
        r = rng.uniform(-3, 3, 4)
        if rng.uniform() < 0.5:                                                            # a complex pair
            c = np.poly(np.array([r[0], r[1], complex(r[2], abs(r[3]) + 0.1), complex(r[2], -abs(r[3]) - 0.1)])).real
            want = np.sort(r[:2])
        else:
            c = np.poly(r); want = np.sort(r)
        if np.diff(want).min(initial=1.0) < 1e-3:
            continue
        c = c * rng.uniform(0.5, 2.0)
        got = oracle.quartic_real_roots(c)
        assert len(got) == len(want) and np.allclose(got, want, rtol=1e-7, atol=1e-7), (c, got, want)


def test_p3p_recovers_the_pose_from_exact_data(oracle):
    rng = np.random.Generator(np.random.PCG64(5))
    bad = 0
    for planar in (False, True):
        for _ in range(100):
            R = rs.rot(rng.normal(size=3), rng.uniform(0, 0.6)); t = rng.normal(size=3) * 0.3
            P = np.stack([rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), np.full(3, 1.5) if planar else rng.uniform(1, 3, 3)], 1)
            Q = P @ R.T + t
            j = Q / np.linalg.norm(Q, axis=1)[:, None]
            sols = oracle.p3p(P, j)
            err = min([np.abs(Rs - R).max() + np.abs(ts - t).max() for Rs, ts in sols] or [9.0])
            bad += err > 1e-6
            for Rs, ts in sols:                                                            # every returned pose is a rigid motion
                assert np.allclose(Rs @ Rs.T, np.eye(3), atol=1e-9) and np.linalg.det(Rs) > 0.999
    assert bad <= 3, f"{bad} of 200 exact triangles not recovered (near-degenerate ones may be)"


def test_product_quartic_and_p3p_on_the_host(hiplib):
    """the HIP library's own quartic (Ferrari + Newton) and P3P routines, compiled for the host (dvs_test_*): roots against
    numpy, exact poses recovered — including fronto-parallel planar triangles, where the quartic has (near-)double roots"""
    rng = np.random.Generator(np.random.PCG64(3))
    for _ in range(300):
        r = rng.uniform(-3, 3, 4)
        cplx = rng.uniform() < 0.5
        c = np.poly(np.array([r[0], r[1], complex(r[2], abs(r[3]) + 0.1), complex(r[2], -abs(r[3]) - 0.1)])).real if cplx else np.poly(r)
        want = np.sort(r[:2] if cplx else r)
        if np.diff(want).min(initial=1.0) < 1e-3:
            continue
        c = c * rng.uniform(0.5, 2.0)
        out = np.zeros(4)
        n = _hooks().dvs_test_quartic_roots(*[float(v) for v in c], out.ctypes.data)
        assert n == len(want) and np.allclose(np.sort(out[:n]), want, rtol=1e-7, atol=1e-7), (c, out[:n], want)
    bad = 0
    for planar in (False, True):
        for _ in range(200):
            R = rs.rot(rng.normal(size=3), rng.uniform(0, 0.6)) if not planar or rng.uniform() < 0.5 else rs.rot([0, 0, 1], rng.uniform(0, 0.3))
            t = rng.normal(size=3) * 0.3
            P = np.stack([rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3), np.full(3, 1.5) if planar else rng.uniform(1, 3, 3)], 1)
            Q = P @ R.T + t
            j = Q / np.linalg.norm(Q, axis=1)[:, None]
            out = np.zeros(48)
            n = _hooks().dvs_test_p3p(np.ascontiguousarray(P).ctypes.data, np.ascontiguousarray(j).ctypes.data, out.ctypes.data)
            sols = [(out[12 * i:12 * i + 9].reshape(3, 3), out[12 * i + 9:12 * i + 12]) for i in range(n)]
            bad += min([np.abs(Rs - R).max() + np.abs(ts - t).max() for Rs, ts in sols] or [9.0]) > 1e-6
    assert bad <= 6, f"{bad} of 400 exact triangles not recovered"


def test_eight_point_on_exact_correspondences(oracle):
    sc = rs.two_view(n=40, outlier_frac=0.0, noise=0.0, seed=2)
    F = oracle.eight_point(sc["pts1"], sc["pts2"])
    assert F is not None and abs(np.linalg.det(F)) < 1e-9
    assert rs.sampson_truth_error(F, sc) < 1e-3            # float32 pixel coordinates: sub-milli-pixel epipolar error


@pytest.mark.parametrize("seed,outliers", [(0, 0.3), (1, 0.5), (2, 0.1)])
def test_oracle_fundamental_ransac_against_ground_truth(oracle, seed, outliers):
    sc = rs.two_view(n=600, outlier_frac=outliers, noise=0.5, seed=seed)
    F, mask, sel = oracle.find_fundamental_ransac(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000, seed=11)
    assert sel[0] >= 0 and sel[1] <= 1000 and mask.sum() == sel[2]
    recall = (mask.astype(bool) & sc["truth"]).sum() / sc["truth"].sum()
    false_in = (mask.astype(bool) & ~sc["truth"]).sum() / max((~sc["truth"]).sum(), 1)
    # the returned model is the best MINIMAL-sample model (as RANSACPointSetRegistrator returns it: no refit), fitted to 8 noisy
    # points, so some true inliers fall outside the 2 px band; an outlier near the epipolar line passes by chance (4 px band)
    assert recall > (0.85 if outliers <= 0.3 else 0.7) and false_in < 0.08, (recall, false_in)
    assert rs.sampson_truth_error(F, sc) < (1.5 if outliers <= 0.3 else 2.5)
    assert sel[1] < 1000 or outliers >= 0.5, "the adaptive rule must stop early on easy data"


@pytest.mark.parametrize("seed,planar", [(0, False), (1, True), (2, False)])
def test_oracle_pnp_ransac_against_ground_truth(oracle, seed, planar):
    sc = rs.two_view(n=400, outlier_frac=0.3, noise=0.5, seed=seed, planar=planar)
    ok, rvec, tvec, inl, sel = oracle.solve_pnp_ransac(sc["X"], sc["pts2"], sc["K4"], 100, 4.0, 0.99, seed=5)
    assert ok and sel[0] >= 0
    mask = np.zeros(400, bool); mask[inl] = True
    assert rs.iou(mask, sc["truth"]) > 0.85     # the inlier set is the best MINIMAL-sample pose's (as OpenCV returns it), not the refined pose's
    R = rs.rodrigues_to_R(rvec)
    ang = np.arccos(np.clip((np.trace(R.T @ sc["R"]) - 1) / 2, -1, 1))
    assert ang < 3e-3 and np.abs(tvec - sc["t"]).max() < (8e-3 if planar else 4e-3), (ang, tvec - sc["t"])


def test_oracle_degenerate_inputs(oracle):
    z2 = np.zeros((0, 2), np.float32)
    F, mask, sel = oracle.find_fundamental_ransac(z2, z2)
    assert sel[0] == -1 and mask.size == 0
    p = np.full((20, 2), 7.0, np.float32)                                                  # all points identical: no model
    F, mask, sel = oracle.find_fundamental_ransac(p, p)
    assert sel[0] == -1 and mask.sum() == 0
    ok, rvec, tvec, inl, sel = oracle.solve_pnp_ransac(np.zeros((3, 3), np.float32), np.zeros((3, 2), np.float32), [600, 600, 320, 240])
    assert not ok and inl.size == 0
    for n in (20, 10):                                                                     # OpenCV's procedure: no sample passes checkSubset -> no model
        p = np.full((n, 2), 7.0, np.float32)
        F, mask, sel = oracle.find_fundamental_cv(p, p)
        assert sel[0] == -1 and sel[1] == 0 and mask.sum() == 0 and (F == 0).all()
    q = np.stack([np.arange(30, dtype=np.float32), 2 * np.arange(30, dtype=np.float32) + 1], 1)   # all points on one line
    F, mask, sel = oracle.find_fundamental_cv(q, q + 3)
    assert sel[0] == -1 and mask.sum() == 0


# ---------------------------------------------------------------- GPU: the HIP stages against ground truth and against the oracle
@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,outliers", [(600, 0, 0.3), (2000, 1, 0.5), (60, 2, 0.1), (9, 3, 0.0)])
def test_gpu_fundamental_ransac(gpu, oracle, n, seed, outliers):
    from dvslam_amd import FrontendGlue
    sc = rs.two_view(n=n, outlier_frac=outliers, noise=0.5, seed=seed)
    g = FrontendGlue()
    F, mask, nin = g.find_fundamental_ransac(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000, seed=11)
    F2, mask2, sel = oracle.find_fundamental_ransac(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000, seed=11)
    assert nin == mask.sum()
    # same sampler, same estimator, different numerical routines: the same hypothesis wins unless two are within rounding,
    # and the inlier sets agree up to correspondences sitting on the 2 px boundary
    assert rs.iou(mask, mask2) > 0.97, (mask.sum(), mask2.sum())
    if n >= 60:
        recall = (mask.astype(bool) & sc["truth"]).sum() / sc["truth"].sum()
        false_in = (mask.astype(bool) & ~sc["truth"]).sum() / max((~sc["truth"]).sum(), 1)
        assert recall > (0.85 if outliers <= 0.3 else 0.7) and false_in < 0.1, (recall, false_in)
        assert rs.sampson_truth_error(F, sc) < (1.5 if outliers <= 0.3 else 2.5) and abs(np.linalg.det(F)) < 1e-6
    F3, mask3, nin3 = g.find_fundamental_ransac(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000, seed=11)
    assert (mask3 == mask).all() and (F3 == F).all(), "deterministic for a given seed"


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,planar,outliers", [(400, 0, False, 0.3), (400, 1, True, 0.3), (1500, 2, False, 0.6), (12, 3, True, 0.0)])
def test_gpu_pnp_ransac(gpu, oracle, n, seed, planar, outliers):
    from dvslam_amd import FrontendGlue
    sc = rs.two_view(n=n, outlier_frac=outliers, noise=0.5, seed=seed, planar=planar)
    g = FrontendGlue()
    ok, rvec, tvec, inl = g.solve_pnp_ransac(sc["X"], sc["pts2"], sc["K4"], 100, 4.0, 0.99, seed=5)
    ok2, rvec2, tvec2, inl2, sel = oracle.solve_pnp_ransac(sc["X"], sc["pts2"], sc["K4"], 100, 4.0, 0.99, seed=5)
    assert ok and ok2
    m = np.zeros(n, bool); m[inl] = True
    m2 = np.zeros(n, bool); m2[inl2] = True
    assert (np.diff(inl) > 0).all() and rs.iou(m, m2) > 0.97
    assert rs.iou(m, sc["truth"]) > 0.85
    R = rs.rodrigues_to_R(rvec)
    ang = np.arccos(np.clip((np.trace(R.T @ sc["R"]) - 1) / 2, -1, 1))
    assert ang < 4e-3 and np.abs(tvec - sc["t"]).max() < 1e-2, (ang, tvec - sc["t"])
    if rs.iou(m, m2) == 1.0:    # same inlier set -> the two refinements minimise the same cost: same optimum
        assert np.abs(rvec - rvec2).max() < 1e-6 and np.abs(tvec - tvec2).max() < 1e-6, (rvec - rvec2, tvec - tvec2)


@pytest.mark.gpu
def test_gpu_ransac_degenerate_inputs(gpu):
    from dvslam_amd import FrontendGlue
    g = FrontendGlue()
    z2 = np.zeros((0, 2), np.float32)
    F, mask, nin = g.find_fundamental_ransac(z2, z2)
    assert nin == 0 and mask.size == 0
    p = np.full((20, 2), 7.0, np.float32)
    F, mask, nin = g.find_fundamental_ransac(p, p)
    assert nin == 0 and mask.sum() == 0 and (F == 0).all()
    ok, rvec, tvec, inl = g.solve_pnp_ransac(np.zeros((3, 3), np.float32), np.zeros((3, 2), np.float32), [600, 600, 320, 240])
    assert not ok and inl.size == 0
    for n in (20, 10):                                                                     # dvs_find_fundamental_cv: no sample passes checkSubset
        p = np.full((n, 2), 7.0, np.float32)
        F, mask, nin, its = g.find_fundamental_cv(p, p)
        assert nin == 0 and its == 0 and mask.sum() == 0 and (F == 0).all()
    q = np.stack([np.arange(30, dtype=np.float32), 2 * np.arange(30, dtype=np.float32) + 1], 1)
    F, mask, nin, its = g.find_fundamental_cv(q, q + 3)
    assert nin == 0 and mask.sum() == 0 and (F == 0).all()
    rng = np.random.Generator(np.random.PCG64(1))                                          # pure noise: no consistent pose
    ok, rvec, tvec, inl = g.solve_pnp_ransac(rng.uniform(-1, 1, (50, 3)) + [0, 0, 2], rng.uniform(0, 640, (50, 2)), [600, 600, 320, 240])
    assert inl.size < 15


@pytest.mark.gpu
def test_gpu_ransac_batches_equal_the_single_calls(gpu):
    """dvs_find_fundamental_ransac_batch / dvs_solve_pnp_ransac_batch: many independent problems (ragged sizes, one too small, one
    empty) in one launch sequence; every problem gets bit for bit what its single call gives (mask, inlier count; pose, inlier list)"""
    from dvslam_amd import FrontendGlue
    g = FrontendGlue()
    sizes = [600, 9, 0, 250, 5, 1200, 64, 8]
    scenes = [rs.two_view(n=max(n, 1), outlier_frac=0.3, noise=0.5, seed=10 + i, planar=(i % 2 == 1)) for i, n in enumerate(sizes)]
    p1 = [sc["pts1"][:n] for sc, n in zip(scenes, sizes)]; p2 = [sc["pts2"][:n] for sc, n in zip(scenes, sizes)]
    X = [sc["X"][:n] for sc, n in zip(scenes, sizes)]
    seeds = [101 + 7 * i for i in range(len(sizes))]
    fb = g.find_fundamental_ransac_batch(p1, p2, seeds, 2.0, 0.99, 1000)
    pb = g.solve_pnp_ransac_batch(X, p2, scenes[0]["K4"], seeds, 100, 4.0, 0.99)
    for i, n in enumerate(sizes):
        F, mask, nin = g.find_fundamental_ransac(p1[i], p2[i], 2.0, 0.99, 1000, seed=seeds[i])
        assert (fb[i][0] == mask).all() and fb[i][1] == nin, i
        ok, rvec, tvec, inl = g.solve_pnp_ransac(X[i], p2[i], scenes[0]["K4"], 100, 4.0, 0.99, seed=seeds[i])
        assert pb[i][0] == ok and (pb[i][3] == inl).all(), i
        assert (pb[i][1].view(np.uint64) == rvec.view(np.uint64)).all() and (pb[i][2].view(np.uint64) == tvec.view(np.uint64)).all(), i
    big = g.find_fundamental_ransac_batch([p1[0]] * 200, [p2[0]] * 200, list(range(200)), 2.0, 0.99, 1000)   # results leave by a copy command
    F, mask, nin = g.find_fundamental_ransac(p1[0], p2[0], 2.0, 0.99, 1000, seed=137)
    assert (big[137][0] == mask).all() and big[137][1] == nin


# ---------------------------------------------------------------- cv::findFundamentalMat as OpenCV itself runs it (dvs_find_fundamental_cv)
def _py_cv_rng(state):
    """cv::RNG::next() — a third statement (Python integers) of the multiply-with-carry generator"""
    state = state if state else 0xffffffff
    while True:
        state = ((state & 0xffffffff) * 4164903690 + (state >> 32)) & 0xffffffffffffffff
        yield state & 0xffffffff


def _py_cv_subsets(p1, p2, k, iters):
    """RANSACPointSetRegistrator::getSubset over `iters` iterations, third statement (Python, float64 arithmetic on the float32 points)"""
    n = len(p1)
    rng = _py_cv_rng(0xffffffffffffffff)
    eps = float(np.finfo(np.float32).eps)

    def collinear(p, idx):
        i = len(idx) - 1
        for j in range(i):
            dx1 = float(p[idx[j], 0]) - float(p[idx[i], 0]); dy1 = float(p[idx[j], 1]) - float(p[idx[i], 1])
            for m in range(j):
                dx2 = float(p[idx[m], 0]) - float(p[idx[i], 0]); dy2 = float(p[idx[m], 1]) - float(p[idx[i], 1])
                if abs(dx2 * dy1 - dy2 * dx1) <= eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                    return True
        return False
    out = []
    for _ in range(iters):
        for attempt in range(10000):
            idx = []
            for i in range(k):
                v = next(rng) % n
                while v in idx:
                    v = next(rng) % n
                idx.append(v)
            if not collinear(p1, idx) and not collinear(p2, idx):
                break
        else:
            return out
        out.append(idx)
    return out


def test_cv_rng_and_subsets_three_statements(hiplib, oracle):
    """cv::RNG and getSubset: the product's host routine (dvs_cv_ransac_subsets), the oracle's and a Python statement give the same
    sequence — on integer pixel coordinates (ORB keypoints of level 0), where collinear samples DO occur and are drawn again"""
    from dvslam_amd import glue
    g = _py_cv_rng(0xffffffffffffffff)
    assert oracle.cv_rng_sequence(0xffffffffffffffff, 50) == [next(g) for _ in range(50)]
    g = _py_cv_rng(0)
    assert oracle.cv_rng_sequence(0, 5) == [next(g) for _ in range(5)]                      # state 0 -> 0xffffffff (cv::RNG's constructor)
    rng = np.random.Generator(np.random.PCG64(9))
    for n, k, iters in ((40, 7, 60), (300, 7, 200), (16, 7, 50), (500, 5, 100)):
        p1 = rng.integers(0, 48, (n, 2)).astype(np.float32)                                # a small integer grid: many collinear triples
        p2 = p1 + rng.integers(-2, 3, (n, 2)).astype(np.float32)
        want = _py_cv_subsets(p1, p2, k, iters)
        a, fa = glue.cv_ransac_subsets(p1, p2, k, iters)
        b, fb = oracle.cv_subsets(p1, p2, k, iters)
        assert fa == fb == len(want) == iters
        assert a.tolist() == want and b.tolist() == want
        assert all(len(set(r)) == k for r in want)
    # without the collinearity redraw the sequence would differ: the test above is not vacuous
    p1 = rng.integers(0, 48, (40, 2)).astype(np.float32); p2 = p1.copy()
    plain = []
    g = _py_cv_rng(0xffffffffffffffff)
    for _ in range(60):
        idx = []
        for i in range(7):
            v = next(g) % 40
            while v in idx:
                v = next(g) % 40
            idx.append(v)
        plain.append(idx)
    assert plain != _py_cv_subsets(p1, p2, 7, 60)


def test_oracle_seven_point_on_exact_correspondences(oracle):
    """the 7-point models of exact correspondences: one of them is the true epipolar geometry, all are singular"""
    sc = rs.two_view(n=40, outlier_frac=0.0, noise=0.0, seed=4)
    rng = np.random.Generator(np.random.PCG64(2))
    for _ in range(20):
        idx = rng.choice(40, 7, replace=False)
        Fs = oracle.seven_point(sc["pts1"], sc["pts2"], idx)
        assert 1 <= len(Fs) <= 3
        assert min(rs.sampson_truth_error(F, sc) for F in Fs) < 2e-2
        for F in Fs:
            assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-9


@pytest.mark.parametrize("seed,outliers", [(0, 0.3), (1, 0.5), (2, 0.1)])
def test_oracle_fundamental_cv_against_ground_truth(oracle, seed, outliers):
    sc = rs.two_view(n=600, outlier_frac=outliers, noise=0.5, seed=seed)
    F, mask, sel = oracle.find_fundamental_cv(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000)
    assert sel[0] >= 0 and sel[1] <= 1000 and mask.sum() == sel[2]
    recall = (mask.astype(bool) & sc["truth"]).sum() / sc["truth"].sum()
    false_in = (mask.astype(bool) & ~sc["truth"]).sum() / max((~sc["truth"]).sum(), 1)
    assert recall > (0.8 if outliers <= 0.3 else 0.65) and false_in < 0.1, (recall, false_in)   # a 7-point minimal model, no refit
    assert sel[1] < 1000 or outliers >= 0.5


@pytest.mark.parametrize("n,seed", [(14, 0), (11, 1), (8, 2), (12, 3)])
def test_oracle_fundamental_lmeds_below_fifteen_points(oracle, n, seed):
    """below 15 points cv::findFundamentalMat(FM_RANSAC) runs LMedS: 300 iterations at confidence 0.99, the smallest median wins"""
    sc = rs.two_view(n=n, outlier_frac=0.0 if seed != 3 else 0.25, noise=0.3, seed=seed)
    F, mask, sel = oracle.find_fundamental_cv(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000)
    assert sel[1] == 300 and sel[0] >= 0 and mask.sum() == sel[2]
    # With n close to 7 the winning model's median lies among (or right behind) its own seven sample points, whose error is ~0: sigma
    # collapses and little more than the sample survives — LMedS as published, not a defect of the restatement
    assert mask.sum() >= 7 and abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-9
    F2, mask2, sel2 = oracle.find_fundamental_cv(sc["pts1"][:7], sc["pts2"][:7])
    assert sel2[0] == -1 and mask2.sum() == 0                                                # the reference never calls below 8 (frontend.cpp:627)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,outliers", [(600, 0, 0.3), (2000, 1, 0.5), (60, 2, 0.1), (15, 3, 0.0)])
def test_gpu_fundamental_cv(gpu, oracle, n, seed, outliers):
    """dvs_find_fundamental_cv against the oracle's statement of the same OpenCV algorithm: the same samples (integer arithmetic), the
    same loop — iterations run, inliers of the best model and the inlier mask agree; the models come from different numerical routines
    (elimination on raw coordinates / eigen-decomposition on normalised ones), so a correspondence within rounding of the threshold
    may flip (IoU bar 0.99) and F agrees to 1e-6 relative when the masks are identical"""
    from dvslam_amd import FrontendGlue
    sc = rs.two_view(n=n, outlier_frac=outliers, noise=0.5, seed=seed)
    if seed == 2:
        sc["pts1"] = np.round(sc["pts1"]); sc["pts2"] = np.round(sc["pts2"])                # integer pixel coordinates (level-0 keypoints)
    g = FrontendGlue()
    F, mask, nin, its = g.find_fundamental_cv(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000)
    F2, mask2, sel = oracle.find_fundamental_cv(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000)
    assert nin == mask.sum() and sel[0] >= 0
    assert rs.iou(mask, mask2) > 0.99, (mask.sum(), mask2.sum())
    if (mask == mask2).all():
        assert its == sel[1] and nin == sel[2]
        assert np.abs(F / np.linalg.norm(F) - F2 / np.linalg.norm(F2)).max() < 1e-6 or np.abs(F / np.linalg.norm(F) + F2 / np.linalg.norm(F2)).max() < 1e-6
    assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-9 and (F[2, 2] == 1.0 or F[2, 2] == 0.0)
    if n >= 60:
        recall = (mask.astype(bool) & sc["truth"]).sum() / sc["truth"].sum()
        assert recall > (0.8 if outliers <= 0.3 else 0.65)
    F3, mask3, nin3, its3 = g.find_fundamental_cv(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000)
    assert (mask3 == mask).all() and (F3 == F).all() and its3 == its
    # batch = single, and fewer than 15 points are refused (OpenCV would run LMedS)
    fb = g.find_fundamental_cv_batch([sc["pts1"], sc["pts1"][: max(15, n // 2)]], [sc["pts2"], sc["pts2"][: max(15, n // 2)]])
    assert (fb[0][0] == mask).all() and fb[0][1] == nin and fb[0][2] == its
    Fh, mh, nh, ih = g.find_fundamental_cv(sc["pts1"][: max(15, n // 2)], sc["pts2"][: max(15, n // 2)])
    assert (fb[1][0] == mh).all() and fb[1][1] == nh
    with pytest.raises(Exception):
        g.find_fundamental_cv(sc["pts1"][:7], sc["pts2"][:7])


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(14, 0), (11, 1), (8, 2), (12, 3), (9, 4)])
def test_gpu_fundamental_lmeds_below_fifteen_points(gpu, oracle, n, seed):
    """dvs_find_fundamental_cv below 15 points (LMedS) against the oracle's statement: the same samples, the same 300 iterations.  At 14
    points mask, count and F agree; from 8 to 13 the algorithm's own score is rounding noise (below) and only what it defines is compared"""
    from dvslam_amd import FrontendGlue
    sc = rs.two_view(n=n, outlier_frac=0.25 if seed >= 3 else 0.0, noise=0.3, seed=seed)
    g = FrontendGlue()
    F, mask, nin, its = g.find_fundamental_cv(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000)
    F2, mask2, sel = oracle.find_fundamental_cv(sc["pts1"], sc["pts2"], 2.0, 0.99, 1000)
    assert its == sel[1] == 300 and nin == mask.sum()
    if n == 14:
        # 14 points: the median (index 7) is the best error OUTSIDE the model's own seven sample points — a well-defined score
        assert (mask == mask2).all() and nin == sel[2]
        if nin >= 7:
            a, b = F / np.linalg.norm(F), F2 / np.linalg.norm(F2)
            assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-6
    else:
        # 8 .. 13 points: index n / 2 < 7, so EVERY model's median is the error of one of its own sample points — rounding noise around
        # zero — and which model "wins" is decided by that noise (in OpenCV: by its SVD's).  What is defined: sigma falls to its floor
        # (0.001), the inliers are the points within 1e-6 px^2 of the winner, i.e. its own seven sample points (plus any that fit to
        # that precision), and the winner is one of the 300 drawn samples.  Both implementations must satisfy exactly that.
        from dvslam_amd import glue
        subsets, found = glue.cv_ransac_subsets(sc["pts1"], sc["pts2"], 7, 300)
        assert found == 300
        for mk in (mask, mask2):
            inl = set(np.flatnonzero(mk).tolist())
            assert len(inl) >= 7 and any(set(row.tolist()) <= inl for row in subsets)
    if nin < 7:
        assert (F == 0).all()                                                                # OpenCV returns an empty matrix, the mask stays
    # mixed batch: RANSAC and LMedS problems side by side = the single calls
    big = rs.two_view(n=200, outlier_frac=0.2, noise=0.5, seed=seed + 10)
    fb = g.find_fundamental_cv_batch([big["pts1"], sc["pts1"], big["pts1"][:40]], [big["pts2"], sc["pts2"], big["pts2"][:40]])
    assert (fb[1][0] == mask).all() and fb[1][1] == nin and fb[1][2] == 300
    Fs, ms, ns, it_s = g.find_fundamental_cv(big["pts1"], big["pts2"])
    assert (fb[0][0] == ms).all() and fb[0][2] == it_s


# ---------------------------------------------------------------- cv::solvePnPRansac as OpenCV itself runs it (dvs_solve_pnp_ransac_cv)
def _pnp_scene(n, seed, outliers, noise=0.5, planar=False):
    sc = rs.two_view(n, outliers, noise, seed, planar=planar)
    return sc["X"], sc["pts2"], sc["K4"], sc                        # camera-1 points against their (noisy, partly wrong) image in view 2


def test_cv_pnp_subsets_three_statements(hiplib, oracle):
    """the RANSAC stage's 5-point samples: product host routine, oracle and a Python statement of cv::RNG + getSubset (no checkSubset)"""
    from dvslam_amd import glue
    for n, iters in ((6, 40), (37, 100), (600, 100)):
        g = _py_cv_rng(0xffffffffffffffff)
        want = []
        for _ in range(iters):
            idx = []
            for i in range(5):
                v = next(g) % n
                while v in idx:
                    v = next(g) % n
                idx.append(v)
            want.append(idx)
        assert glue.cv_ransac_subsets_nocheck(n, 5, iters).tolist() == want
        assert oracle.cv_subsets_nocheck(n, 5, iters).tolist() == want


def test_oracle_epnp_and_iterative_refit_on_exact_data(oracle):
    """the two solvers of the OpenCV procedure on noise-free data: EPnP on 5 general points and solvePnP(ITERATIVE) on many points,
    planar (homography initialisation) and not (DLT initialisation), recover the pose"""
    rng = np.random.Generator(np.random.PCG64(5))
    K4 = np.array([600.0, 610.0, 320.0, 240.0])
    for trial in range(20):
        R = rs.rot(rng.normal(size=3), np.deg2rad(rng.uniform(1, 25))); t = rng.normal(size=3) * 0.2
        w = np.array(rs_rodrigues(R))
        X = np.stack([rng.uniform(-1, 1, 60), rng.uniform(-0.8, 0.8, 60), rng.uniform(1.5, 3.0, 60)], 1)
        Xc = X @ R.T + t
        uv = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1)
        r5, t5 = oracle.epnp(X[:5], uv[:5], K4)
        assert np.abs(r5 - w).max() < 1e-4 and np.abs(t5 - t).max() < 1e-4, (trial, r5, w)       # float image points: ~1e-7 relative
        ok, ri, ti = oracle.solve_pnp_iterative(X, uv, K4)
        assert ok and np.abs(ri - w).max() < 1e-6 and np.abs(ti - t).max() < 1e-6   # CvLevMarq stops on a relative step below FLT_EPSILON
        Xp = X.copy(); Xp[:, 2] = 2.0
        Xc = Xp @ R.T + t
        uvp = np.stack([K4[0] * Xc[:, 0] / Xc[:, 2] + K4[2], K4[1] * Xc[:, 1] / Xc[:, 2] + K4[3]], 1)
        ok, ri, ti = oracle.solve_pnp_iterative(Xp, uvp, K4)
        assert ok and np.abs(ri - w).max() < 1e-6 and np.abs(ti - t).max() < 1e-6   # CvLevMarq stops on a relative step below FLT_EPSILON


def rs_rodrigues(R):
    """rotation matrix -> Rodrigues vector (angle < pi)"""
    th = np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return v * (th / (2 * np.sin(th))) if th > 1e-12 else v * 0.5


@pytest.mark.parametrize("seed,outliers", [(0, 0.0), (1, 0.2), (2, 0.4), (3, 0.45)])
def test_oracle_pnp_cv_against_ground_truth(oracle, seed, outliers):
    X, uv, K4, sc = _pnp_scene(500, seed, outliers)
    ok, rvec, tvec, inl, sel, m6 = oracle.solve_pnp_ransac_cv(X, uv, K4, 100, 4.0, 0.99)
    assert ok and sel[0] >= 0 and sel[1] <= 100
    got = np.zeros(len(X), bool); got[inl] = True
    assert rs.iou(got, sc["truth"]) > 0.93                                              # 0.5 px noise against a 4 px gate
    R = rs.rodrigues_to_R(rvec)
    assert np.abs(R - sc["R"]).max() < 2e-3 and np.abs(tvec - sc["t"]).max() < 4e-3
    if outliers == 0.0:
        assert sel[1] <= 3                                                             # RANSACUpdateNumIters stops a clean problem at once


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,outliers", [(600, 0, 0.0), (600, 1, 0.3), (300, 2, 0.5), (120, 3, 0.2), (40, 4, 0.25), (12, 5, 0.0), (6, 6, 0.0), (1500, 7, 0.35)])
def test_gpu_pnp_cv(gpu, oracle, n, seed, outliers):
    """dvs_solve_pnp_ransac_cv against the oracle's statement of the same OpenCV procedure: the SAME samples (host-made on both sides, checked
    above), the loop stops at the SAME iteration, the SAME inlier list — index for index — and the refitted pose to 1e-6 (Jacobi
    eigen-decompositions with different orderings / thresholds on the two sides; CvLevMarq ends on a relative step below FLT_EPSILON)"""
    from dvslam_amd import FrontendGlue
    X, uv, K4, sc = _pnp_scene(n, seed, outliers)
    ok2, r2, t2, inl2, sel2, m6 = oracle.solve_pnp_ransac_cv(X, uv, K4, 100, 4.0, 0.99)
    g = FrontendGlue()
    ok, r, t, inl, its = g.solve_pnp_ransac_cv(X, uv, K4, 100, 4.0, 0.99)
    assert ok == ok2 and its == sel2[1], (ok, ok2, its, sel2)
    assert inl.tolist() == inl2.tolist()
    assert np.abs(r - r2).max() < 1e-6 and np.abs(t - t2).max() < 1e-6, (r, r2, t, t2)
    if n >= 40:
        got = np.zeros(len(X), bool); got[inl] = True
        assert rs.iou(got, sc["truth"]) > 0.9 and np.abs(rs.rodrigues_to_R(r) - sc["R"]).max() < 5e-3
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,outliers", [(600, 0, 0.0), (600, 1, 0.3), (120, 3, 0.2), (12, 5, 0.0)])
def test_gpu_pnp_cv_against_real_opencv_when_present(gpu, n, seed, outliers):
    """THE PIN THIS REPO LACKS: where `cv2` is importable, dvs_solve_pnp_ransac_cv is compared with cv2.solvePnPRansac itself on the
    same correspondences — same call as frontend.cpp:911-921 (100 iterations, 4 px, 0.99, no distortion).  The image this repo is built and
    judged in has no OpenCV (parity unpinned), so this test SKIPS there; on a machine with OpenCV 4.x it states the bar:
    the same inlier set up to the documented deviations (Jacobi eigen-decompositions where OpenCV runs its SVD: IoU >= 0.98) and the refitted
    pose to 1e-4."""
    cv2 = pytest.importorskip("cv2")
    from dvslam_amd import FrontendGlue
    X, uv, K4, sc = _pnp_scene(n, seed, outliers)
    K = np.array([[K4[0], 0, K4[2]], [0, K4[1], K4[3]], [0, 0, 1]], np.float64)
    ok2, r2, t2, inl2 = cv2.solvePnPRansac(X.astype(np.float32), uv.astype(np.float32), K, None, iterationsCount=100, reprojectionError=4.0, confidence=0.99)
    g = FrontendGlue()
    ok, r, t, inl, its = g.solve_pnp_ransac_cv(X, uv, K4, 100, 4.0, 0.99)
    g.close()
    assert bool(ok) == bool(ok2)
    if ok2:
        a = np.zeros(len(X), bool); a[inl] = True
        b = np.zeros(len(X), bool); b[np.asarray(inl2).reshape(-1)] = True
        assert rs.iou(a, b) >= 0.98, (int(a.sum()), int(b.sum()))
        assert np.abs(r - np.asarray(r2).reshape(3)).max() < 1e-4 and np.abs(t - np.asarray(t2).reshape(3)).max() < 1e-4


@pytest.mark.gpu
def test_gpu_pnp_cv_batch_and_refusals(gpu, oracle):
    """many problems in one launch sequence = the single calls bit for bit; fewer than 6 points are refused per problem; an exactly planar
    point set (EPnP's degenerate case, in OpenCV as here) must not fault or hang"""
    from dvslam_amd import FrontendGlue
    g = FrontendGlue()
    scenes = [_pnp_scene(n, 10 + i, o) for i, (n, o) in enumerate([(400, 0.3), (5, 0.0), (90, 0.1), (0, 0.0), (700, 0.45)])]
    singles = [g.solve_pnp_ransac_cv(X, uv, K4) if len(X) else (False, np.zeros(3), np.zeros(3), np.zeros(0, np.int32), 0) for X, uv, K4, _ in scenes]
    batch = g.solve_pnp_ransac_cv_batch([s[0] for s in scenes], [s[1] for s in scenes], scenes[0][2])
    for a, b in zip(singles, batch):
        assert a[0] == b[0] and a[4] == b[4] and a[3].tolist() == b[3].tolist()
        assert (a[1].view(np.uint64) == b[1].view(np.uint64)).all() and (a[2].view(np.uint64) == b[2].view(np.uint64)).all()
    assert not batch[1][0] and len(batch[1][3]) == 0 and not batch[3][0]               # 5 and 0 points: refused
    Xp, uvp, K4, _ = _pnp_scene(300, 3, 0.2, planar=True)
    ok, r, t, inl, its = g.solve_pnp_ransac_cv(Xp, uvp, K4)
    assert 1 <= its <= 100 and np.isfinite(r).all() and np.isfinite(t).all()
    g.close()
