"""Shared checks that bracket the Levenberg-Marquardt loop from OUTSIDE its own transcription (the product's
dvs_ba_solve* and oracle/ba_oracle.cpp's solveLM are twins, so agreeing with each other proves little):

* scipy_golden(name): optimum of the same window found by scipy.optimize.least_squares (tools/gen_ba_scipy_golden.py — other
  residual code, minimal rotation parameterisation, reflective trust region), committed under tests/golden/;
* check_schedule(trace): Ceres' documented trust-region schedule asserted on a solver's own iteration log
  (ceres-solver docs, "Levenberg-Marquardt" / trust_region_minimizer.cc + levenberg_marquardt_strategy.cc):
  initial radius 1e4; on a successful step radius <- min(1e16, radius / max(1/3, 1 - (2 rho - 1)^3)) and the decrease
  factor resets to 2; on an unsuccessful (rejected or invalid) step radius <- radius / decrease_factor and the factor doubles
  (so consecutive failures divide by 2, 4, 8 ...); a step is accepted iff rho > 1e-3;
* gauge_aligned_errors: pose / landmark agreement of two solutions up to the similarity the problem leaves free."""
import os
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HARD = [  # windows whose LM run rejects steps and goes on (found with the oracle): K, L, seed, pose noise (m, rad), landmark noise, outliers
    dict(K=5, L=200, seed=4, pose_noise=(0.3, np.deg2rad(12)), lm_noise=0.5, outlier_frac=0.2),
    dict(K=5, L=200, seed=3, pose_noise=(0.2, np.deg2rad(8)), lm_noise=0.3, outlier_frac=0.1),
    dict(K=10, L=2000, seed=42, pose_noise=(0.3, np.deg2rad(10)), lm_noise=0.3, outlier_frac=0.1),
]


def scipy_golden(name):
    """-> (make_ba_problem keyword arguments, dict with initial_cost, optimum_cost, q, t, X of scipy's optimum)"""
    import json
    z = np.load(os.path.join(GOLD, f"ba_scipy_{name}.npz"))
    return json.loads(str(z["make_ba_problem_kwargs"])), {k: z[k] for k in z.files}


def check_schedule(trace):
    """trace: [n, 6] rows {radius, kind, cost change, model change, rho, candidate cost}.  Returns (#accepted, #unsuccessful)."""
    assert len(trace) >= 1 and trace[0, 0] == 1e4, "initial trust-region radius 1e4"
    factor = 2.0
    nacc = nfail = 0
    for i in range(len(trace)):
        radius, kind, dc, dm, rho = trace[i, 0], int(trace[i, 1]), trace[i, 2], trace[i, 3], trace[i, 4]
        if kind in (3, 4):                       # tolerance reached on the candidate: the loop ends here
            assert i == len(trace) - 1
            break
        if kind in (1, 2):
            assert dm > 0 and abs(rho - dc / dm) <= 1e-12 * abs(rho), "relative decrease = cost change / model cost change"
            assert (kind == 1) == (rho > 1e-3), "accepted iff relative decrease > min_relative_decrease (1e-3)"
        if kind == 1:
            expect = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            factor = 2.0
            nacc += 1
        else:
            expect = radius / factor
            factor *= 2.0
            nfail += 1
        if i + 1 < len(trace):
            assert abs(trace[i + 1, 0] - expect) <= 1e-12 * expect, f"radius after iteration {i} (kind {kind}): {trace[i + 1, 0]} != {expect}"
    return nacc, nfail


def _R(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def gauge_aligned_errors(q, t, X, q2, t2, X2):
    """Both solutions fix pose 0, so they differ by at most a scaling about camera 0's centre.  Returns (max rotation angle
    [rad] between corresponding poses, max camera-centre distance and max landmark distance after rescaling solution 2 by the
    least-squares scale, the scale)."""
    C0 = -_R(q[0]).T @ t[0]
    cen = lambda qq, tt: np.stack([-_R(qq[k]).T @ tt[k] for k in range(len(qq))])  # noqa: E731
    c1, c2 = cen(q, t) - C0, cen(q2, t2) - C0
    a, b = X - C0, X2 - C0
    s = float((a * b).sum() / (b * b).sum())
    ang = max(np.arccos(np.clip((np.trace(_R(q[k]).T @ _R(q2[k])) - 1) / 2, -1, 1)) for k in range(len(q)))
    return ang, float(np.abs(c1 - s * c2).max()), float(np.abs(a - s * b).max()), s
