import ctypes as C
import numpy as np
from ._lib import lib, check, ptr, BaSummary, DvsError

TERMINATION = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}


class BAProblem:
    """Direct driver of the dvs_ba_* C-ABI: problem in the optimiser's parameterisation (bundle_adjustment.hpp:92-165)."""

    def __init__(self, prob, device=0):
        self._L = lib()
        h = C.c_void_p()
        check(self._L.dvs_ba_create(device, C.byref(h)))
        self._h = h
        self.set_problem(prob)

    def set_problem(self, prob):
        """a new window on the same handle (what SlidingWindowBA::optimize does per call): device memory is reused when it fits"""
        self.K, self.L, self.R = int(prob["K"]), int(prob["L"]), len(prob["cam_idx"])
        a = lambda k, dt: np.ascontiguousarray(prob[k], dt)
        self._keep = [a("q", np.float64), a("t", np.float64), a("X", np.float64), a("cam_idx", np.int32), a("lm_idx", np.int32),
                      a("uv", np.float64), a("pose_fixed", np.uint8), a("lm_fixed", np.uint8)]
        q, t, X, cam, lm, uv, pf, lf = self._keep
        check(self._L.dvs_ba_set_problem(self._h, self.K, ptr(q), ptr(t), self.L, ptr(X), self.R, ptr(cam), ptr(lm), ptr(uv), ptr(pf),
                                         ptr(lf), prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["sigma"], prob["huber"]))

    def close(self):
        if getattr(self, "_h", None):
            self._L.dvs_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def evaluate_raw(self):
        R = self.R
        r = np.zeros((R, 2)); jq = np.zeros((R, 2, 4)); jt = np.zeros((R, 2, 3)); jx = np.zeros((R, 2, 3))
        check(self._L.dvs_ba_evaluate_raw(self._h, ptr(r), ptr(jq), ptr(jt), ptr(jx)))
        return r, jq, jt, jx

    def evaluate(self):
        R = self.R
        cost = C.c_double(); r = np.zeros((R, 2)); jp = np.zeros((R, 2, 6)); jl = np.zeros((R, 2, 3)); g = np.zeros(6 * self.K + 3 * self.L)
        check(self._L.dvs_ba_evaluate(self._h, C.byref(cost), ptr(r), ptr(jp), ptr(jl), ptr(g)))
        return cost.value, r, jp, jl, g

    def normal_equations(self):
        hpp = np.zeros((self.K, 6, 6)); hll = np.zeros((self.L, 3, 3)); w = np.zeros((self.R, 6, 3)); g = np.zeros(6 * self.K + 3 * self.L)
        cost = C.c_double()
        check(self._L.dvs_ba_normal_equations(self._h, ptr(hpp), ptr(hll), ptr(w), ptr(g), C.byref(cost)))
        return hpp, hll, w, g, cost.value

    def evaluate_device(self, iters):
        check(self._L.dvs_ba_evaluate_device(self._h, iters))

    def synchronize(self):
        check(self._L.dvs_ba_synchronize(self._h))

    def set_stream(self, s):
        check(self._L.dvs_ba_set_stream(self._h, s))

    def solve(self, max_iterations=10, ftol=1e-6, gtol=1e-10, ptol=1e-8):
        s = BaSummary()
        check(self._L.dvs_ba_solve(self._h, max_iterations, ftol, gtol, ptol, C.byref(s)))
        return s

    def solve_device(self, max_iterations=10, ftol=1e-6, gtol=1e-10, ptol=1e-8):
        """dvs_ba_solve_device: the same trust-region loop with the Schur complement / Cholesky / back-substitution on the GPU"""
        s = BaSummary()
        check(self._L.dvs_ba_solve_device(self._h, max_iterations, ftol, gtol, ptol, C.byref(s)))
        return s

    def trace(self):
        """dvs_ba_get_trace: [iterations, 6] = radius, kind (0 invalid, 1 accepted, 2 rejected, 3 ptol, 4 ftol), cost change,
        model cost change, relative decrease, candidate cost of the last solve"""
        n = C.c_int32()
        check(self._L.dvs_ba_get_trace(self._h, None, 0, C.byref(n)))
        rows = np.zeros((n.value, 6))
        check(self._L.dvs_ba_get_trace(self._h, ptr(rows), n.value, C.byref(n)))
        return rows

    def parameters(self):
        q = np.zeros((self.K, 4)); t = np.zeros((self.K, 3)); X = np.zeros((self.L, 3))
        check(self._L.dvs_ba_get_parameters(self._h, ptr(q), ptr(t), ptr(X)))
        return q, t, X


class SlidingWindowBA:
    """Python mirror of the reference class (bundle_adjustment.hpp:652-904).

    optimize(keyframes, landmarks, observations, max_iterations=10) with
      keyframes   : list of (frame_id, R 3x3, t 3) in the caller's convention (fromRt inverts it, :138-165)
      landmarks   : list of (id, category, (x, y, z), fixed)
      observations: list of ((u, v), landmark_id, category, frame_id)
    returns a dict with the OptimizationResult fields (:419-432)."""

    def __init__(self, fx, fy, cx, cy, sigma_pixels=1.0, device=0):
        self.fx, self.fy, self.cx, self.cy, self.sigma, self.device = fx, fy, cx, cy, sigma_pixels, device

    def optimize(self, keyframes, landmarks, observations, max_iterations=10):
        L_ = lib()
        res = dict(success=False, final_cost=0.0, iterations_completed=0, frames_optimized=len(keyframes),
                   landmarks_optimized=len(landmarks), message="", optimized_poses={}, optimized_landmarks={})
        if not keyframes or not landmarks or not observations:
            res["message"] = "Insufficient input data for optimization"          # :750-755
            return res
        frame_slot = {}                                                           # std::map<int, ...> keyed by frame_id
        q = []; t = []
        for fid, R, tt in keyframes:
            qq = np.zeros(4); tr = np.zeros(3)
            check(L_.dvs_ba_pose_from_rt(ptr(np.ascontiguousarray(R, np.float64)), ptr(np.ascontiguousarray(tt, np.float64).reshape(3)), ptr(qq), ptr(tr)))
            frame_slot[fid] = len(q); q.append(qq); t.append(tr)
        lm_slot = {}; X = []; fixed = []; cat = {}
        for lid, category, pos, fx_ in landmarks:                                  # keyed by id only (:762, :790)
            lm_slot[lid] = len(X); X.append(np.asarray(pos, np.float64)); fixed.append(1 if fx_ else 0); cat[lid] = category
        cam = []; lm = []; uv = []
        for (u, v), lid, _c, fid in observations:                                  # unknown ids are skipped (:805-809)
            if fid in frame_slot and lid in lm_slot:
                cam.append(frame_slot[fid]); lm.append(lm_slot[lid]); uv.append((u, v))
        if not cam:
            res["message"] = "No valid observation constraints"                    # :829-834
            return res
        pose_fixed = np.zeros(len(q), np.uint8)
        pose_fixed[frame_slot[keyframes[0][0]]] = 1                                # first keyframe in the vector is the gauge (:781-785)
        prob = dict(K=len(q), L=len(X), q=np.array(q), t=np.array(t), X=np.array(X), cam_idx=np.array(cam, np.int32),
                    lm_idx=np.array(lm, np.int32), uv=np.array(uv, np.float64), pose_fixed=pose_fixed, lm_fixed=np.array(fixed, np.uint8),
                    fx=self.fx, fy=self.fy, cx=self.cx, cy=self.cy, sigma=self.sigma, huber=1.345)
        p = BAProblem(prob, self.device)
        try:                                                                        # options at :839-847
            s = p.solve_device(max_iterations, 1e-6, 1e-10, 1e-8)
        except DvsError as e:                                                       # shapes outside the device solver's window limits
            if e.code != -2:
                raise
            s = p.solve(max_iterations, 1e-6, 1e-10, 1e-8)
        res["success"] = s.termination == 0                                        # :860
        res["final_cost"] = s.final_cost
        res["iterations_completed"] = s.num_successful_steps                       # :862
        res["message"] = ("Bundle adjustment converged successfully" if res["success"]
                          else "Bundle adjustment failed to converge: " + TERMINATION[s.termination])
        qo, to, Xo = p.parameters()
        for fid, slot in frame_slot.items():
            R = np.zeros((3, 3)); tt = np.zeros(3)
            check(L_.dvs_ba_pose_to_rt(ptr(np.ascontiguousarray(qo[slot])), ptr(np.ascontiguousarray(to[slot])), ptr(R), ptr(tt)))
            res["optimized_poses"][fid] = (R, tt)
        for lid, slot in lm_slot.items():
            res["optimized_landmarks"][(lid, cat[lid])] = Xo[slot].copy()
        p.close()
        return res
