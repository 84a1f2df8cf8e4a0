"""CPU: structural known-answers for the BA oracle (oracle/ba_oracle.cpp).  The reference pins nothing
here (no tests touch bundle_adjustment.hpp), so these are the properties derivable from its text."""
import ctypes as C
import numpy as np
import pytest
from dvslam_amd import synth


def _rot(q, p):
    q = q / np.linalg.norm(q)
    return synth._quat_rot(q, p[None, :])[0]


def _residual(P, q, t, X, uv):
    pc = _rot(q, X) + t
    if pc[2] <= 0.1:
        return np.zeros(2)
    return np.array([P["fx"] * pc[0] / pc[2] + P["cx"] - uv[0], P["fy"] * pc[1] / pc[2] + P["cy"] - uv[1]]) / P["sigma"]


def test_autodiff_matches_finite_differences_and_functor(oracle):
    P = synth.make_ba_problem(K=3, L=25, seed=3)
    P["q"] = P["q"] * np.array([[1.0], [1.7], [0.6]])          # un-normalised quaternions: the functor normalises internally
    P["sigma"] = 1.5
    o = oracle.OracleBA(P)
    r, jq, jt, jx = o.evaluate_raw()
    eps = 1e-6
    for i in range(0, len(r), 7):
        c, l = P["cam_idx"][i], P["lm_idx"][i]
        q, t, X, uv = P["q"][c], P["t"][c], P["X"][l], P["uv"][i]
        assert np.allclose(r[i], _residual(P, q, t, X, uv), rtol=1e-12, atol=1e-10)
        for k in range(4):
            d = np.zeros(4); d[k] = eps
            fd = (_residual(P, q + d, t, X, uv) - _residual(P, q - d, t, X, uv)) / (2 * eps)
            assert np.allclose(jq[i][:, k], fd, rtol=1e-5, atol=1e-4)
        for k in range(3):
            d = np.zeros(3); d[k] = eps
            assert np.allclose(jt[i][:, k], (_residual(P, q, t + d, X, uv) - _residual(P, q, t - d, X, uv)) / (2 * eps), rtol=1e-5, atol=1e-4)
            assert np.allclose(jx[i][:, k], (_residual(P, q, t, X + d, uv) - _residual(P, q, t, X - d, uv)) / (2 * eps), rtol=1e-5, atol=1e-4)


def test_point_behind_camera_gives_zero_residual_and_jacobian(oracle):
    P = synth.make_ba_problem(K=2, L=10, seed=4)
    P["X"][3] = [0.0, 0.0, -50.0]                              # z_c <= 0.1 for every camera (bundle_adjustment.hpp:545-550)
    o = oracle.OracleBA(P)
    r, jq, jt, jx = o.evaluate_raw()
    m = P["lm_idx"] == 3
    assert (r[m] == 0).all() and (jq[m] == 0).all() and (jt[m] == 0).all() and (jx[m] == 0).all()
    assert (r[~m] != 0).any()


def test_huber_and_manifold_pieces(oracle):
    L = oracle.lib()
    L.orc_ba_huber.argtypes = [C.c_double, C.c_double, C.c_void_p]
    L.orc_ba_quat_plus.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rho = np.zeros(3)
    L.orc_ba_huber(1.345, 1.0, rho.ctypes.data)
    assert rho.tolist() == [1.0, 1.0, 0.0]
    L.orc_ba_huber(1.345, 9.0, rho.ctypes.data)
    assert np.isclose(rho[0], 2 * 1.345 * 3 - 1.345 ** 2) and np.isclose(rho[1], 1.345 / 3) and np.isclose(rho[2], -rho[1] / 18)
    # Plus keeps unit norm and Plus(x, 0) = x; d Plus / d delta at 0 equals the PlusJacobian used in evaluate()
    x = np.array([0.3, -0.5, 0.2, 0.78]); x /= np.linalg.norm(x)
    out = np.zeros(4)
    L.orc_ba_quat_plus(x.ctypes.data, np.zeros(3).ctypes.data, out.ctypes.data)
    assert (out == x).all()
    J = np.zeros((4, 3))
    for k in range(3):
        d = np.zeros(3); d[k] = 1e-7
        a = np.zeros(4); b = np.zeros(4)
        L.orc_ba_quat_plus(x.ctypes.data, d.ctypes.data, a.ctypes.data); L.orc_ba_quat_plus(x.ctypes.data, (-d).ctypes.data, b.ctypes.data)
        J[:, k] = (a - b) / 2e-7
        assert abs(np.linalg.norm(a) - 1) < 1e-12
    expect = np.array([[x[3], x[2], -x[1]], [-x[2], x[3], x[0]], [x[1], -x[0], x[3]], [-x[0], -x[1], -x[2]]])
    assert np.allclose(J, expect, atol=1e-7)


def test_local_jacobian_gradient_consistency(oracle):
    P = synth.make_ba_problem(K=4, L=60, seed=5)
    o = oracle.OracleBA(P)
    cost, r, jp, jl, g = o.evaluate()
    hpp, hll, w, g2, cost2 = o.normal_equations()
    assert cost == cost2 and (g == g2).all()
    K = P["K"]
    gp = np.zeros((K, 6)); gl = np.zeros((P["L"], 3))
    for i in range(len(r)):
        if not P["pose_fixed"][P["cam_idx"][i]]:
            gp[P["cam_idx"][i]] += jp[i].T @ r[i]
        gl[P["lm_idx"][i]] += jl[i].T @ r[i]
    assert np.allclose(g[:6 * K].reshape(K, 6), gp, rtol=1e-12, atol=1e-12) and np.allclose(g[6 * K:].reshape(-1, 3), gl, rtol=1e-12, atol=1e-12)
    assert (g[:6] == 0).all() and (hpp[0] == 0).all()         # first pose is the fixed gauge (bundle_adjustment.hpp:781-785)
    i = len(r) - 1                                             # an observation of a free camera
    assert np.allclose(w[i], jp[i].T @ jl[i], rtol=1e-12)


def test_noise_free_problem_is_a_fixed_point(oracle):
    P = synth.make_ba_problem(K=4, L=50, seed=6, pixel_noise=0.0, outlier_frac=0.0, pose_noise=(0.0, 0.0), lm_noise=0.0)
    o = oracle.OracleBA(P)
    cost, r, jp, jl, g = o.evaluate()
    assert cost < 1e-18 and np.abs(g).max() < 1e-6
    s = o.solve(10)
    assert s.termination == 0 and s.final_cost < 1e-18


def test_lm_converges_on_the_synthetic_window(oracle):
    P = synth.make_ba_problem(K=5, L=200, seed=7)
    o = oracle.OracleBA(P)
    s = o.solve(60)                                             # Huber tail with 2 % outliers converges slowly (IRLS-like)
    assert s.termination == 0 and s.num_successful_steps >= 2
    assert s.final_cost < 0.6 * s.initial_cost
    q, t, X = o.parameters()
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-9)
    assert (q[0] == P["q"][0]).all() and (t[0] == P["t"][0]).all()
    P2 = dict(P); P2["q"], P2["t"], P2["X"] = q, t, X            # inlier reprojection error is at the noise level (1 px)
    r = oracle.OracleBA(P2).evaluate_raw()[0]
    assert np.median(np.linalg.norm(r, axis=1)) < 1.6
    s2 = oracle.OracleBA(P).solve(2)                            # iteration limit -> NO_CONVERGENCE (success=false in the reference)
    assert s2.termination == 1 and s2.num_iterations == 2


# ---- the LM loop bracketed from outside its own transcription (tests/ba_bracket.py) --------------------------------------
@pytest.mark.parametrize("name", ["3x60", "5x200", "10x2000"])
def test_oracle_lm_reaches_the_scipy_optimum(oracle, name):
    """oracle/ba_oracle.cpp's restatement of ceres::Solve, run to convergence, lands on the optimum an independent
    scipy.optimize.least_squares(huber) solve found (tools/gen_ba_scipy_golden.py): relative cost <= 1e-6"""
    import ba_bracket as bb
    kw, G = bb.scipy_golden(name)
    o = oracle.OracleBA(synth.make_ba_problem(**kw))
    s = o.solve(100, 1e-14, 1e-14, 1e-14)
    assert abs(s.initial_cost - float(G["initial_cost"])) <= 1e-9 * s.initial_cost, "same problem, same Huber cost at the start"
    assert abs(s.final_cost - float(G["optimum_cost"])) <= 1e-6 * float(G["optimum_cost"])
    q, t, X = o.parameters()
    ang, dc, dX, scale = bb.gauge_aligned_errors(q, t, X, G["q"], G["t"], G["X"])
    assert ang < 2e-4 and dc < 2e-3, (ang, dc, dX, scale)      # poses agree up to the free scale


@pytest.mark.parametrize("kw", __import__("ba_bracket").HARD[:2], ids=["5x200 seed 4", "5x200 seed 3"])
def test_oracle_follows_the_documented_trust_region_schedule(oracle, kw):
    import ba_bracket as bb
    o = oracle.OracleBA(synth.make_ba_problem(**kw))
    s = o.solve(40)
    tr = o.trace()
    nacc, nfail = bb.check_schedule(tr)
    assert nacc == s.num_successful_steps and nfail >= 2, "these windows must exercise rejected / invalid steps"
    assert len(tr) == s.num_iterations
