#!/usr/bin/env python3
"""File-driven stand-in for bag_playback.launch.xml WITH the tracking stages (BASELINE configs[4]):
a synthetic RGB-D sequence runs through the two nodes' per-frame logic, once on the MI355X (every stage through the C-ABI) and
once on the CPU oracle, and the two trajectories are compared with each other and with the generator's closed-form ground truth.

Per frame (Frontend::syncCallback, frontend.cpp:1068-1324):
  gray -> ORB extract -> filterDepth -> match vs previous frame -> distance < 50 -> findFundamentalMat(RANSAC, 2 px, 0.99) inliers
  -> feature culling (matched first, then <= 200 unmatched with response >= 50, best first; :1171-1219)
  -> estimateCameraPose: 3D points from the PREVIOUS depth image, solvePnPRansac(100, 4 px, 0.99), inverse motion, isMotionOutlier
     (0.5 m / 0.2 rad), pose accumulation R_ / t_ (:843-962)
  -> isKeyframe: match vs last keyframe + F-RANSAC, keyframe if < 150 consistent matches or > 30 frames (:601-662)
  -> publishKeyframe as Keyframe.msg CDR bytes (:699-790)
Backend (Backend::syncCallback / bundleAdjustmentCallback, backend.cpp:709-989): unpack, associate observations with the landmark
database (Hamming < 50, reprojection < 5 px), new landmarks otherwise, and every `ba_every` keyframes SlidingWindowBA over the
last 5 keyframes (the reference runs it on a 2 s wall timer; a replay has no wall clock).

Two phases.  What does not depend on the pose — gray -> ORB extract -> filterDepth -> match vs the previous frame
(frontend.cpp:1084-1132) — runs for ALL frames first, 64 frames per call through dvs_orb_extract_batch_device /
dvs_filter_depth_batch_device / dvs_match_hamming_sequence_device with everything resident in HBM (`BatchedFrontEnd`; frames shard
contiguously over `--shards` ranks, each re-extracting the one frame before its range, so the shards need no exchange and their
results are gathered where phase 2 runs).  Phase 2 is the sequential part (:1136-1324): fundamental-matrix gate, feature culling,
PnP + pose accumulation, the keyframe test against the LAST KEYFRAME (whose identity is only known sequentially) and the backend.

The scene is a fronto-parallel textured plane at Z0 (depth image constant), the camera translates in the image plane and rolls
(dvslam_amd.synth.traj_state), so ground truth is exact.  RANSAC seeds are the frame index: both pipelines draw the same samples."""
import argparse
import json
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


# Variant switches of the replay (set by run() / the command line; both pipelines read the same ones):
#   PNP_MODE "own": dvs_solve_pnp_ransac — P3P + LM with the library's documented sampler (the default: valid on the exactly planar scene);
#            "cv":  dvs_solve_pnp_ransac_cv — cv::solvePnPRansac by OpenCV's procedure (EPnP on cv::RNG's 5-point samples + iterative refit).
#   RELIEF_MM: the depth image becomes plane + a 64-pixel checker of this height (SURVEY.md section 8d's "50 mm checker"), so that the 3D
#            points of a frame are NOT coplanar — EPnP's 4-control-point formulation is rank-deficient on an exactly planar set, in
#            OpenCV as here.  The frames stay renderings of the plane: a 3 % depth error moves a reprojection by < 0.25 px per frame.
PNP_MODE = "own"
RELIEF_MM = 0


def make_depth(rows, cols, z0):
    d = np.full((rows, cols), int(round(z0 * 1000)), np.uint16)
    if RELIEF_MM:
        yy, xx = np.mgrid[0:rows, 0:cols]
        d[((yy // 64 + xx // 64) & 1) == 1] += np.uint16(RELIEF_MM)
    return d


def rodrigues_to_R(w):
    th = np.linalg.norm(w)
    if th < 1e-15:
        return np.eye(3)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def rot_angle(R):
    return float(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1)))


def quat_xyzw(R):
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


class HipStages:
    """every stage through the C-ABI of libdvslam_hip.so (ctypes drivers in dvslam_amd)"""
    name = "hip"

    def __init__(self, nfeatures):
        import dvslam_amd
        self.orb = dvslam_amd.ORBextractor(nfeatures, 1.2, 8, 20, 7)
        self.mat = dvslam_amd.BFMatcher()
        self.g = dvslam_amd.FrontendGlue()
        self._ba = dvslam_amd

    def extract(self, gray):
        n, k, d = self.orb(gray)
        return k, d

    def filter_depth(self, k, d, depth):
        return self.g.filter_depth(k, d, depth)[:2]

    def match(self, q, t):
        return self.mat.match(q, t)

    def fundamental_inliers(self, p1, p2, seed):
        # cv::findFundamentalMat(FM_RANSAC, 2.0, 0.99) as OpenCV runs it: its own sample sequence and 7-point solver, RANSAC from 15 points
        # on and LMedS below (the reference calls with >= 8, frontend.cpp:627); `seed` belongs to the library's own estimator, unused here
        return self.g.find_fundamental_cv(p1, p2, 2.0, 0.99, 1000)[1].astype(bool)

    def pnp(self, obj, img, K4, seed):
        if PNP_MODE == "cv":
            ok, rvec, tvec, inl, _ = self.g.solve_pnp_ransac_cv(obj, img, K4, 100, 4.0, 0.99)
            return ok, rvec, tvec, len(inl)
        ok, rvec, tvec, inl = self.g.solve_pnp_ransac(obj, img, K4, 100, 4.0, 0.99, seed)
        return ok, rvec, tvec, len(inl)

    def publish(self, *a, **kw):
        return self.g.publish_keyframe(*a, **kw)[0]

    # ---- the same stages over MANY frames at once (track_batched): one launch sequence per stage ----
    def fundamental_inliers_batch(self, p1_list, p2_list, seeds):
        return [m.astype(bool) for m, _, _ in self.g.find_fundamental_cv_batch(p1_list, p2_list, 2.0, 0.99, 1000)]

    def pnp_batch(self, obj_list, img_list, K4, seeds):
        if PNP_MODE == "cv":
            return [(ok, rvec, tvec, len(inl)) for ok, rvec, tvec, inl, _ in self.g.solve_pnp_ransac_cv_batch(obj_list, img_list, K4, 100, 4.0, 0.99)]
        return [(ok, rvec, tvec, len(inl)) for ok, rvec, tvec, inl in self.g.solve_pnp_ransac_batch(obj_list, img_list, K4, seeds, 100, 4.0, 0.99)]

    def match_many_vs_one(self, q_list, train):
        """keyframe-pair jobs: every query set against ONE train set (the last keyframe), one dvs_match_hamming_batch_device call"""
        from dvslam_amd import _lib
        n = len(q_list)
        cap = max(max(len(q) for q in q_list), 1)
        Q = np.zeros((n, cap, 32), np.uint8)
        for i, q in enumerate(q_list):
            Q[i, :len(q)] = q
        nq = np.array([len(q) for q in q_list], np.int32); nt = np.full(n, len(train), np.int32)
        D = lambda a: _lib.DeviceBuffer(max(a.nbytes, 4)).upload(a)
        dQ, dT, dnq, dnt = D(Q), D(np.ascontiguousarray(train, np.uint8)), D(nq), D(nt)
        di, dd = _lib.DeviceBuffer(n * cap * 4), _lib.DeviceBuffer(n * cap * 4)
        self.mat.match_batch_device(dQ.ptr, dnq.ptr, cap, dT.ptr, dnt.ptr, 0, n, di.ptr, dd.ptr)   # train stride 0: the same set for every job
        self.mat.synchronize()
        idx = di.download(np.int32, n * cap).reshape(n, cap); dist = dd.download(np.int32, n * cap).reshape(n, cap)
        return [(idx[i, :nq[i]].copy(), dist[i, :nq[i]].copy()) for i in range(n)]

    def unpack(self, payload):
        from dvslam_amd.glue import unpack_keyframe
        return unpack_keyframe(payload)

    def associate(self, *a):
        return self.g.associate(*a)

    def solve_ba(self, prob, iters):
        b = self._ba.BAProblem(prob)
        s = b.solve_device(iters)
        return s, b.parameters()


class CpuStages:
    """the same stages on the CPU oracle (tests/oracle_bindings.py): the checker, and the CPU reference of the comparison"""
    name = "cpu"

    def __init__(self, nfeatures):
        import oracle_bindings as ob
        self.ob = ob
        self.orb = ob.OracleORB(nfeatures, 1.2, 8, 20, 7)

    def extract(self, gray):
        n, k, d = self.orb.extract(gray)
        return k, d

    def filter_depth(self, k, d, depth):
        return self.ob.filter_depth(k, d, depth)[:2]

    def match(self, q, t):
        return self.ob.match(q, t)

    def fundamental_inliers(self, p1, p2, seed):
        return self.ob.find_fundamental_cv(p1, p2, 2.0, 0.99, 1000)[1].astype(bool)

    def pnp(self, obj, img, K4, seed):
        if PNP_MODE == "cv":
            ok, rvec, tvec, inl, sel, _ = self.ob.solve_pnp_ransac_cv(obj, img, K4, 100, 4.0, 0.99)
            return ok, rvec, tvec, len(inl)
        ok, rvec, tvec, inl, sel = self.ob.solve_pnp_ransac(obj, img, K4, 100, 4.0, 0.99, seed)
        return ok, rvec, tvec, len(inl)

    def publish(self, *a, **kw):
        return self.ob.publish_keyframe(*a, **kw)[0]

    def unpack(self, payload):
        return self.ob.unpack_keyframe(payload)

    def associate(self, *a):
        return self.ob.associate(*a)

    def solve_ba(self, prob, iters):
        o = self.ob.OracleBA(prob)
        s = o.solve(iters)
        return s, o.parameters()


class BatchedFrontEnd:
    """phase 1 on the MI355X: extraction + depth filter + match against the previous frame for a run of frames, B per call,
    device-resident between the three stages; one bulk download per batch.  Returns per frame (keypoints, descriptors, trainIdx, dist)
    after the depth filter; frame 0 of the run has no match (idx / dist None)."""

    def __init__(self, nfeatures, rows, cols, B=64, device=0):
        import ctypes as C
        import dvslam_amd
        from dvslam_amd import _lib
        self.C, self._lib, self.L = C, _lib, _lib.lib()
        self.B, self.rows, self.cols, self.device = B, rows, cols, device
        self.orb = dvslam_amd.ORBextractor(nfeatures, 1.2, 8, 20, 7, device=device, max_batch=B)
        self.mat = dvslam_amd.BFMatcher(device=device, stream=self.orb.get_stream())     # one stream: the three stages depend on each other
        self.cap = cap = self.orb.capacity
        vp, i32, sz, f32 = C.c_void_p, C.c_int32, C.c_size_t, C.c_float
        self.L.dvs_filter_depth_batch_device.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, i32, sz, sz, f32, f32, vp, vp, vp, vp]
        D = lambda n: _lib.DeviceBuffer(n, device)
        self.d_img = D(B * rows * cols); self.d_depth = D(rows * cols * 2)
        self.d_k, self.d_d, self.d_n = D(B * cap * 28), D(B * cap * 32), D(B * 4)
        # filtered outputs in two sets: the first frame of a batch is matched against the last frame of the batch before
        self.f_k = [D(B * cap * 28) for _ in range(2)]; self.f_d = [D(B * cap * 32) for _ in range(2)]; self.f_n = [D(B * 4) for _ in range(2)]
        self.d_idx, self.d_dist = D(B * cap * 4), D(B * cap * 4)

    def run_into_block(self, frames, depth, blk):
        """phase 1 of a contiguous run of frames with the results left ON THE DEVICE in `blk` (a RankBlock: frame slots in structure-of-arrays
        form, the send block of the multi-rank gather): every batch writes its filtered keypoints / descriptors / counts and its matches
        straight into its slots, the first frame of a batch reads its predecessor from the slot before.  Asynchronous but for the uploads."""
        B, cap, rows, cols = self.B, self.cap, self.rows, self.cols
        assert len(frames) <= blk.slots and cap == blk.cap
        self.d_depth.upload(np.ascontiguousarray(depth, np.uint16))
        for b0 in range(0, len(frames), B):
            nb = min(B, len(frames) - b0)
            self.orb.synchronize()                    # the image slots are reused: the previous batch's extraction has read them
            for i in range(nb):
                fr = np.ascontiguousarray(frames[b0 + i])
                self._lib.check(self.L.dvs_memcpy_h2d(self.device, self.d_img.ptr + i * rows * cols, self._lib.ptr(fr), fr.nbytes))
            self.orb.extract_batch_device(self.d_img.ptr, nb, rows, cols, cols, rows * cols, self.d_k.ptr, self.d_d.ptr, cap, self.d_n.ptr)
            self._lib.check(self.L.dvs_filter_depth_batch_device(self.mat._h, self.d_k.ptr, self.d_d.ptr, self.d_n.ptr, cap, nb, self.d_depth.ptr, rows, cols,
                                                                 cols * 2, 0, 0.3, 3.0, blk.kps(b0), blk.desc(b0), None, blk.n(b0)))
            prev = (blk.desc(b0 - 1), blk.n(b0 - 1)) if b0 else (0, 0)
            self.mat.match_sequence_device(blk.desc(b0), blk.n(b0), cap, nb, prev[0], prev[1], blk.idx(b0), blk.dist(b0))

    def run(self, frames, depth):
        C, B, cap, rows, cols = self.C, self.B, self.cap, self.rows, self.cols
        self.d_depth.upload(np.ascontiguousarray(depth, np.uint16))
        out = []
        for b0 in range(0, len(frames), B):
            nb = min(B, len(frames) - b0)
            s, sp = (b0 // B) & 1, ((b0 // B) & 1) ^ 1
            for i in range(nb):                       # frame by frame into its slot: no host-side copy of the batch
                fr = np.ascontiguousarray(frames[b0 + i])
                self._lib.check(self.L.dvs_memcpy_h2d(self.device, self.d_img.ptr + i * rows * cols, self._lib.ptr(fr), fr.nbytes))
            self.orb.extract_batch_device(self.d_img.ptr, nb, rows, cols, cols, rows * cols, self.d_k.ptr, self.d_d.ptr, cap, self.d_n.ptr)
            self._lib.check(self.L.dvs_filter_depth_batch_device(self.mat._h, self.d_k.ptr, self.d_d.ptr, self.d_n.ptr, cap, nb, self.d_depth.ptr, rows, cols,
                                                                 cols * 2, 0, 0.3, 3.0, self.f_k[s].ptr, self.f_d[s].ptr, None, self.f_n[s].ptr))
            prev = (self.f_d[sp].ptr + (B - 1) * cap * 32, self.f_n[sp].ptr + (B - 1) * 4) if b0 else (0, 0)   # full batches precede b0
            self.mat.match_sequence_device(self.f_d[s].ptr, self.f_n[s].ptr, cap, nb, prev[0], prev[1], self.d_idx.ptr, self.d_dist.ptr)
            self.orb.synchronize()
            n = self.f_n[s].download(np.int32, nb)
            k = self.f_k[s].download(np.uint8, nb * cap * 28).view(self._lib.KP_DTYPE).reshape(nb, cap)
            d = self.f_d[s].download(np.uint8, nb * cap * 32).reshape(nb, cap, 32)
            idx = self.d_idx.download(np.int32, nb * cap).reshape(nb, cap); dist = self.d_dist.download(np.int32, nb * cap).reshape(nb, cap)
            for f in range(nb):
                first = b0 + f == 0
                out.append((k[f, :n[f]].copy(), d[f, :n[f]].copy(), None if first else idx[f, :n[f]].copy(), None if first else dist[f, :n[f]].copy()))
        return out


class RankBlock:
    """One rank's phase-1 results as ONE device block — what dvs_comm_all_gather moves: `slots` frame slots in structure-of-arrays form
    {int32 n[slots]; keypoints[slots][cap] x 28 B; descriptors[slots][cap][32]; int32 trainIdx[slots][cap]; int32 distance[slots][cap]},
    every array 256-byte aligned.  Same layout on every rank (slots = the longest shard + the one re-extracted frame before it)."""

    def __init__(self, slots, cap):
        up = lambda v: (v + 255) // 256 * 256
        self.slots, self.cap = slots, cap
        self.o_n = 0
        self.o_k = up(slots * 4)
        self.o_d = self.o_k + up(slots * cap * 28)
        self.o_i = self.o_d + up(slots * cap * 32)
        self.o_s = self.o_i + up(slots * cap * 4)
        self.nbytes = self.o_s + up(slots * cap * 4)
        self.base = 0

    def at(self, base):
        self.base = base
        return self

    def n(self, f): return self.base + self.o_n + 4 * f
    def kps(self, f): return self.base + self.o_k + f * self.cap * 28
    def desc(self, f): return self.base + self.o_d + f * self.cap * 32
    def idx(self, f): return self.base + self.o_i + f * self.cap * 4
    def dist(self, f): return self.base + self.o_s + f * self.cap * 4

    def unpack(self, raw, nframes, first_has_match):
        """host bytes of one block -> per frame (keypoints, descriptors, trainIdx, distance)"""
        from dvslam_amd._lib import KP_DTYPE
        S, cap = self.slots, self.cap
        n = raw[self.o_n:self.o_n + 4 * S].view(np.int32)
        k = raw[self.o_k:self.o_k + S * cap * 28].view(KP_DTYPE).reshape(S, cap)
        d = raw[self.o_d:self.o_d + S * cap * 32].reshape(S, cap, 32)
        ii = raw[self.o_i:self.o_i + S * cap * 4].view(np.int32).reshape(S, cap)
        dd = raw[self.o_s:self.o_s + S * cap * 4].view(np.int32).reshape(S, cap)
        out = []
        for f in range(nframes):
            m = int(n[f])
            nomatch = f == 0 and not first_has_match
            out.append((k[f, :m].copy(), d[f, :m].copy(), None if nomatch else ii[f, :m].copy(), None if nomatch else dd[f, :m].copy()))
        return out


def shard_bounds(n, world, r):
    return r * n // world, (r + 1) * n // world


def sharded_front_end_rank(frames_of, n, depth, nfeatures, rank, world, comm, device=0, B=64, fe=None):
    """BASELINE configs[4] across `world` ranks (bag_playback.launch.xml:1-8 starts ONE frontend; here its pose-independent half,
    frontend.cpp:1084-1132, is sharded): this rank runs phase 1 on its contiguous frame range [a, b) — re-extracting frame a - 1 instead of
    receiving it — with the results left on the device, then ONE dvs_comm_all_gather of the per-rank blocks brings every shard to every
    rank (the tracking rank needs them all; RCCL has no gather-to-one cheaper than this at 8 ranks x ~9 MB).  `frames_of(t)` yields frame t.
    Returns the per-frame list of the whole sequence (on every rank)."""
    from dvslam_amd import _lib
    rows, cols = depth.shape
    fe = fe or BatchedFrontEnd(nfeatures, rows, cols, B, device)
    slots = max(shard_bounds(n, world, r)[1] - shard_bounds(n, world, r)[0] for r in range(world)) + 1
    blk = RankBlock(slots, fe.cap)
    gather = _lib.DeviceBuffer(world * blk.nbytes, device)
    _lib.check(_lib.lib().dvs_memset(device, gather.ptr, 0, gather.nbytes))
    a, b = shard_bounds(n, world, rank)
    lo = max(a - 1, 0)
    mine = gather.ptr + rank * blk.nbytes
    if b > a:
        fe.run_into_block([frames_of(t) for t in range(lo, b)], depth, blk.at(mine))
    if world > 1:
        comm.all_gather(fe.orb.get_stream(), mine, gather.ptr, blk.nbytes)      # in place: this rank's block already sits in its slot
    fe.orb.synchronize()
    raw = gather.download(np.uint8, world * blk.nbytes)
    out = []
    for r in range(world):
        ra, rb = shard_bounds(n, world, r)
        if rb <= ra:
            continue
        rlo = max(ra - 1, 0)
        part = blk.unpack(raw[r * blk.nbytes:(r + 1) * blk.nbytes], rb - rlo, first_has_match=False)
        out += part[ra - rlo:]
    return out


def sharded_front_end_loopback(frames, depth, nfeatures, world, B=64):
    """the same on ONE GPU: `world` logical ranks of this process (dvs_comm_create_loopback), one host thread, extractor and communicator
    each — the collective path of the multi-rank program without the second GPU.  Returns rank 0's view."""
    import threading
    from dvslam_amd import dist as dvdist
    rows, cols = depth.shape
    comms = dvdist.Comm.loopback(0, world)
    fes = [BatchedFrontEnd(nfeatures, rows, cols, B, 0) for _ in range(world)]
    res, err = [None] * world, []

    def work(r):
        try:
            res[r] = sharded_front_end_rank(lambda t: frames[t], len(frames), depth, nfeatures, r, world, comms[r], 0, B, fes[r])
        except Exception as e:   # noqa: BLE001
            err.append((r, repr(e)))
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in comms:
        c.close()
    assert not err, err
    return res


def batched_front_end(frames, depth, nfeatures, shards=1, B=64, fe=None):
    """phase 1 over `shards` contiguous frame ranges (one per rank on a multi-GPU node; here one after the other on the one GPU): shard
    r runs frames [a_r - 1, b_r) — it re-extracts the frame before its range instead of receiving it — and contributes [a_r, b_r)"""
    rows, cols = frames[0].shape
    n = len(frames)
    fe = fe or BatchedFrontEnd(nfeatures, rows, cols, B)
    out = []
    for r in range(shards):
        a, b = r * n // shards, (r + 1) * n // shards
        if b <= a:
            continue
        lo = max(a - 1, 0)
        part = fe.run(frames[lo:b], depth)
        out += part[a - lo:]
    return out


def pts_of(kps, idx):
    return np.stack([kps["x"][idx], kps["y"][idx]], 1).astype(np.float32)


def track(stages, n_frames, cols, rows, f, z0, nfeatures, ba_every=10, verbose=False, frames=None, pre=None):
    """runs the sequence through `stages`; returns the per-frame visual-odometry poses (camera-to-world, frame 0 = identity),
    keyframe list, backend results"""
    from dvslam_amd import synth
    cx, cy = cols / 2.0, rows / 2.0
    K4 = np.array([f, f, cx, cy])
    depth = make_depth(rows, cols, z0)
    R_, t_ = np.eye(3), np.zeros(3)                       # Frontend::R_, t_ (frontend.cpp:162-163)
    poses = []
    prev_k = prev_d = None
    last_kf_k = last_kf_d = None
    since_kf = 0
    kf_frames, kf_payloads, kf_poses = [], [], []
    stats = dict(matches=[], geometric=[], pnp_inliers=[], pose_updates=0, motion_outliers=0, pnp_failures=0)
    t_stage = 0.0
    for t in range(n_frames):
        gray = frames[t] if frames is not None else synth.make_traj_frame(t, cols, rows)
        t0 = time.perf_counter()
        if pre is not None:                               # phase 1 ran ahead (BatchedFrontEnd): extraction, depth filter, match vs previous
            fk, fd, pre_idx, pre_dist = pre[t]
        else:
            k, d = stages.extract(gray)
            fk, fd = stages.filter_depth(k, d, depth)     # filterDepth (:1100)
        publish = None
        if prev_k is None:                                # first frame (:1285-1300): becomes the first keyframe
            backend_k, backend_d = fk, fd
            publish = True
        else:
            idx, dist = (pre_idx, pre_dist) if pre is not None else stages.match(fd, prev_d)   # :1123
            q = np.nonzero(dist < 50)[0]                  # :1126-1132
            tr = idx[q]
            stats["matches"].append(len(q))
            if len(q) >= 8:                               # :1136-1153
                m = stages.fundamental_inliers(pts_of(prev_k, tr), pts_of(fk, q), seed=2 * t)
                q, tr = q[m], tr[m]
            stats["geometric"].append(len(q))
            # feature culling (:1171-1219)
            matched = set(q.tolist())
            order = list(q)
            un = [i for i in range(len(fk)) if i not in matched]
            un.sort(key=lambda i: -float(fk["response"][i]))
            order += [i for i in un[:200] if fk["response"][i] >= 50.0][:200]
            sel = np.array(order, np.int64)
            backend_k, backend_d = fk[sel], fd[sel]
            # estimateCameraPose (:843-962)
            if len(q) >= 5:
                pp = pts_of(prev_k, tr); cp = pts_of(fk, q)
                xi = np.floor(pp[:, 0] + 0.5).astype(np.int64); yi = np.floor(pp[:, 1] + 0.5).astype(np.int64)   # std::round, positive coordinates
                inb = (xi >= 0) & (yi >= 0) & (xi < cols) & (yi < rows)
                dp = np.zeros(len(pp), np.float32)
                dp[inb] = depth[yi[inb], xi[inb]].astype(np.float32) * np.float32(0.001)
                ok = inb & ~((dp <= np.float32(0.3)) | (dp > np.float32(3.0)))
                obj = np.stack([(pp[:, 0] - np.float32(cx)) * dp / np.float32(f), (pp[:, 1] - np.float32(cy)) * dp / np.float32(f), dp], 1)[ok]
                img = cp[ok]
                if len(obj) >= 6:
                    good, rvec, tvec, nin = stages.pnp(obj, img, K4, seed=2 * t + 1)
                    if good:
                        stats["pnp_inliers"].append(nin)
                        Rr = rodrigues_to_R(rvec)
                        Ri, ti = Rr.T, -Rr.T @ tvec
                        if np.linalg.norm(ti) > 0.5 or rot_angle(Ri) > 0.2:      # isMotionOutlier (:549-570)
                            stats["motion_outliers"] += 1
                        else:
                            t_ = t_ + R_ @ ti
                            R_ = R_ @ Ri
                            stats["pose_updates"] += 1
                    else:
                        stats["pnp_failures"] += 1
            # isKeyframe (:601-662)
            crit = False
            if last_kf_d is not None and len(last_kf_d) and len(backend_d):
                ki, kd = stages.match(backend_d, last_kf_d)
                kq = np.nonzero(kd < 50)[0]
                ktr = ki[kq]
                if len(kq) >= 8:
                    km = stages.fundamental_inliers(pts_of(last_kf_k, ktr), pts_of(backend_k, kq), seed=2 * t + 1000003)
                    kq = kq[km]
                crit = len(kq) < 150
            if crit or since_kf > 30:
                since_kf = 0; publish = True
            else:
                since_kf += 1; publish = False
        if publish:
            last_kf_k, last_kf_d = backend_k, backend_d
            payload = stages.publish(backend_k, backend_d, depth, f, f, cx, cy, R_, t_, stamp=(t, 0), frame_id="camera_link", keyframe_id=len(kf_frames),
                                     q_xyzw=quat_xyzw(R_))
            kf_frames.append(t); kf_payloads.append(payload); kf_poses.append((R_.copy(), t_.copy()))
        prev_k, prev_d = fk, fd
        t_stage += time.perf_counter() - t0
        poses.append((R_.copy(), t_.copy()))
        if verbose and t % 100 == 0:
            print(f"[{stages.name}] frame {t}: {len(fk)} keypoints, {len(kf_frames)} keyframes", flush=True)
    backend = run_backend(stages, kf_payloads, kf_poses, f, cx, cy, ba_every)
    return dict(poses=poses, keyframes=kf_frames, stats=stats, backend=backend, seconds_in_stages=t_stage)


def track_batched(stages, n_frames, cols, rows, f, z0, nfeatures, ba_every, frames, pre):
    """The same tracking as track(), organised by DEPENDENCE instead of by frame.  Of the frontend's per-frame work only
    two things are sequential: WHICH frame is the last keyframe, and the running pose product.  Everything they consume is pose-
    independent and runs for all frames at once, through the batch entry points:
      A. fundamental-matrix gate of (t, t - 1) for every t (dvs_find_fundamental_ransac_batch, seeds as in track()) -> feature culling
      B. PnP of (t, t - 1) for every t (dvs_solve_pnp_ransac_batch) -> relative motions
      C. the keyframe chain: per keyframe k0 the candidates k0 + 1 .. k0 + 32 matched against k0 in ONE batched match call
         (keyframe-pair match jobs, frontend.cpp:614), their fundamental-matrix gates in ONE batch, then the first that qualifies
      D. pose products, Keyframe messages, backend.
    Results are those of track() with the same stages, bit for bit (every problem gets what its single call gives)."""
    cx, cy = cols / 2.0, rows / 2.0
    K4 = np.array([f, f, cx, cy])
    depth = make_depth(rows, cols, z0)
    t0w = time.perf_counter()
    stats = dict(matches=[], geometric=[], pnp_inliers=[], pose_updates=0, motion_outliers=0, pnp_failures=0)
    # ---- A: distance filter + fundamental-matrix gate for every frame pair
    Q, TR = [None] * n_frames, [None] * n_frames
    jobs = []
    for t in range(1, n_frames):
        fk, fd, idx, dist = pre[t]
        q = np.nonzero(dist < 50)[0]; tr = idx[q]
        stats["matches"].append(len(q))
        Q[t], TR[t] = q, tr
        if len(q) >= 8:
            jobs.append(t)
    masks = stages.fundamental_inliers_batch([pts_of(pre[t - 1][0], TR[t]) for t in jobs], [pts_of(pre[t][0], Q[t]) for t in jobs], [2 * t for t in jobs])
    for t, m in zip(jobs, masks):
        Q[t], TR[t] = Q[t][m], TR[t][m]
    BK, BD = [None] * n_frames, [None] * n_frames          # culled features per frame (what a keyframe publishes / is tested with)
    BK[0], BD[0] = pre[0][0], pre[0][1]
    for t in range(1, n_frames):
        fk, fd = pre[t][0], pre[t][1]
        q = Q[t]
        stats["geometric"].append(len(q))
        matched = np.zeros(len(fk), bool); matched[q] = True
        un = np.nonzero(~matched)[0]
        un = un[np.argsort(-fk["response"][un].astype(np.float64), kind="stable")][:200]
        un = un[fk["response"][un] >= 50.0]
        sel = np.concatenate([q, un]).astype(np.int64)
        BK[t], BD[t] = fk[sel], fd[sel]
    # ---- B: PnP of every frame pair
    pj, obj_l, img_l = [], [], []
    for t in range(1, n_frames):
        q, tr = Q[t], TR[t]
        if len(q) < 5:
            continue
        pp = pts_of(pre[t - 1][0], tr); cp = pts_of(pre[t][0], q)
        xi = np.floor(pp[:, 0] + 0.5).astype(np.int64); yi = np.floor(pp[:, 1] + 0.5).astype(np.int64)
        inb = (xi >= 0) & (yi >= 0) & (xi < cols) & (yi < rows)
        dp = np.zeros(len(pp), np.float32)
        dp[inb] = depth[yi[inb], xi[inb]].astype(np.float32) * np.float32(0.001)
        ok = inb & ~((dp <= np.float32(0.3)) | (dp > np.float32(3.0)))
        obj = np.stack([(pp[:, 0] - np.float32(cx)) * dp / np.float32(f), (pp[:, 1] - np.float32(cy)) * dp / np.float32(f), dp], 1)[ok]
        if len(obj) >= 6:
            pj.append(t); obj_l.append(obj); img_l.append(cp[ok])
    motion = {}
    for t, r in zip(pj, stages.pnp_batch(obj_l, img_l, K4, [2 * t + 1 for t in pj])):
        motion[t] = r
    # ---- C: keyframe chain
    kf_frames = [0]
    k0 = 0
    while True:
        cand = list(range(k0 + 1, min(k0 + 33, n_frames)))
        if not cand:
            break
        nxt = None
        if len(BD[k0]):
            res = stages.match_many_vs_one([BD[t] for t in cand], BD[k0])
            KQ, fj = {}, []
            for t, (ki, kd) in zip(cand, res):
                kq = np.nonzero(kd < 50)[0] if len(BD[t]) else np.zeros(0, np.int64)
                KQ[t] = (kq, ki[kq] if len(kq) else kq)
                if len(kq) >= 8:
                    fj.append(t)
            km = stages.fundamental_inliers_batch([pts_of(BK[k0], KQ[t][1]) for t in fj], [pts_of(BK[t], KQ[t][0]) for t in fj],
                                                  [2 * t + 1000003 for t in fj]) if fj else []
            cnt = {t: len(KQ[t][0]) for t in cand}
            for t, m in zip(fj, km):
                cnt[t] = int(m.sum())
        since = 0
        for t in cand:
            crit = bool(len(BD[k0]) and len(BD[t]) and cnt[t] < 150)
            if crit or since > 30:
                nxt = t
                break
            since += 1
        if nxt is None:
            break
        kf_frames.append(nxt); k0 = nxt
    # ---- D: poses, messages, backend
    R_, t_ = np.eye(3), np.zeros(3)
    poses, kf_payloads, kf_poses = [], [], []
    kfset = set(kf_frames)
    for t in range(n_frames):
        if t in motion:
            good, rvec, tvec, nin = motion[t]
            if good:
                stats["pnp_inliers"].append(nin)
                Rr = rodrigues_to_R(rvec)
                Ri, ti = Rr.T, -Rr.T @ tvec
                if np.linalg.norm(ti) > 0.5 or rot_angle(Ri) > 0.2:
                    stats["motion_outliers"] += 1
                else:
                    t_ = t_ + R_ @ ti
                    R_ = R_ @ Ri
                    stats["pose_updates"] += 1
            else:
                stats["pnp_failures"] += 1
        if t in kfset:
            payload = stages.publish(BK[t], BD[t], depth, f, f, cx, cy, R_, t_, stamp=(t, 0), frame_id="camera_link", keyframe_id=len(kf_payloads),
                                     q_xyzw=quat_xyzw(R_))
            kf_payloads.append(payload); kf_poses.append((R_.copy(), t_.copy()))
        poses.append((R_.copy(), t_.copy()))
    t_stage = time.perf_counter() - t0w
    backend = run_backend(stages, kf_payloads, kf_poses, f, cx, cy, ba_every)
    return dict(poses=poses, keyframes=kf_frames, stats=stats, backend=backend, seconds_in_stages=t_stage)


def run_backend(stages, payloads, kf_poses, f, cx, cy, ba_every, window=5, max_iterations=20):
    """Backend::syncCallback's association + landmark database, and SlidingWindowBA::optimize over the last `window` keyframes
    every `ba_every` keyframes (backend.cpp:895-960; the optimised poses replace the stored keyframe poses, :1356-1392)"""
    db_xyz, db_desc = np.zeros((0, 3), np.float32), np.zeros((0, 32), np.uint8)
    kf_R, kf_t, obs_by_kf, assoc, ba_runs = [], [], [], [], []
    for k, payload in enumerate(payloads):
        msg = stages.unpack(payload)
        Rk, Tk = kf_poses[k]
        px = msg["obs_pixels"].astype(np.float32)
        best = (stages.associate(msg["obs_desc"], px, db_desc, db_xyz, Rk, Tk, f, f, cx, cy, 50.0, 5.0) if len(db_xyz)
                else np.full(len(px), -1, np.int32))
        taken, new_xyz, new_desc, obs, n_assoc = set(), [], [], [], 0
        for i in range(len(px)):
            j = int(best[i])
            if j >= 0 and j not in taken:
                taken.add(j); lid = j; n_assoc += 1
            else:
                lid = len(db_xyz) + len(new_xyz)
                new_xyz.append(msg["landmark_xyz"][i]); new_desc.append(msg["obs_desc"][i])
            obs.append((float(px[i, 0]), float(px[i, 1]), lid))
        if new_xyz:
            db_xyz = np.vstack([db_xyz, np.asarray(new_xyz, np.float32)]); db_desc = np.vstack([db_desc, np.asarray(new_desc, np.uint8)])
        kf_R.append(Rk.copy()); kf_t.append(Tk.copy()); obs_by_kf.append(obs); assoc.append(n_assoc)
        if ba_every and (k + 1) % ba_every == 0 and k + 1 >= 2:
            win = list(range(max(0, k + 1 - window), k + 1))
            seen = {}
            for w in win:
                for (_, _, lid) in obs_by_kf[w]:
                    seen[lid] = seen.get(lid, 0) + 1
            lids = sorted(l for l, c in seen.items() if c >= 2)
            if len(lids) < 10:
                continue
            slot = {l: i for i, l in enumerate(lids)}
            q = np.zeros((len(win), 4)); tt = np.zeros((len(win), 3))
            for wi, w in enumerate(win):                  # CameraPose::fromRt: world->camera quaternion (w, x, y, z) + translation
                Rcw = kf_R[w].T
                xyzw = quat_xyzw(Rcw)
                q[wi] = [xyzw[3], xyzw[0], xyzw[1], xyzw[2]]; tt[wi] = -Rcw @ kf_t[w]
            cam, lm, uv = [], [], []
            for wi, w in enumerate(win):
                for (u, v, lid) in obs_by_kf[w]:
                    if lid in slot:
                        cam.append(wi); lm.append(slot[lid]); uv.append((u, v))
            pf = np.zeros(len(win), np.uint8); pf[0] = 1
            prob = dict(K=len(win), L=len(lids), q=q, t=tt, X=db_xyz[lids].astype(np.float64), cam_idx=np.array(cam, np.int32),
                        lm_idx=np.array(lm, np.int32), uv=np.array(uv, np.float64), pose_fixed=pf, lm_fixed=np.zeros(len(lids), np.uint8),
                        fx=f, fy=f, cx=cx, cy=cy, sigma=1.0, huber=1.345)
            s, (qo, to, Xo) = stages.solve_ba(prob, max_iterations)
            ba_runs.append(dict(keyframe=k, window=len(win), landmarks=len(lids), observations=len(cam), termination=int(s.termination),
                                steps=int(s.num_successful_steps), initial_cost=float(s.initial_cost), final_cost=float(s.final_cost)))
            if s.termination == 0:                        # updateOptimizedResults only on success (backend.cpp:967)
                for wi, w in enumerate(win):
                    ww, x, y, z = qo[wi] / np.linalg.norm(qo[wi])
                    Rcw = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                                    [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                                    [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])
                    kf_R[w] = Rcw.T; kf_t[w] = -Rcw.T @ to[wi]
                db_xyz[lids] = Xo.astype(np.float32)
    return dict(landmarks=int(len(db_xyz)), associations=assoc, ba=ba_runs, kf_R=kf_R, kf_t=kf_t)


def ground_truth(n_frames, f, z0):
    from dvslam_amd import synth
    R0, T0 = synth.traj_pose(0, f, z0)
    out = []
    for t in range(n_frames):
        R, T = synth.traj_pose(t, f, z0)
        out.append((R0.T @ R, R0.T @ (T - T0)))            # relative to frame 0, where the frontend starts from identity
    return out


def rmse(a, b):
    e = [np.linalg.norm(pa[1] - pb[1]) for pa, pb in zip(a, b)]
    r = [np.degrees(rot_angle(pa[0].T @ pb[0])) for pa, pb in zip(a, b)]
    return dict(translation_m=float(np.sqrt(np.mean(np.square(e)))), rotation_deg=float(np.sqrt(np.mean(np.square(r)))),
                max_translation_m=float(np.max(e)), max_rotation_deg=float(np.max(r)))


def run(n_frames=1000, cols=640, rows=480, f=600.0, z0=1.5, nfeatures=1000, ba_every=5, with_cpu=True, verbose=False, batched=True, shards=1, pnp="own",
        relief_mm=0):
    global PNP_MODE, RELIEF_MM
    saved = (PNP_MODE, RELIEF_MM)
    PNP_MODE, RELIEF_MM = pnp, relief_mm
    try:
        return _run(n_frames, cols, rows, f, z0, nfeatures, ba_every, with_cpu, verbose, batched, shards, pnp, relief_mm)
    finally:
        PNP_MODE, RELIEF_MM = saved


def _run(n_frames, cols, rows, f, z0, nfeatures, ba_every, with_cpu, verbose, batched, shards, pnp, relief_mm):
    from dvslam_amd import synth
    frames = [synth.make_traj_frame(t, cols, rows) for t in range(n_frames)]
    gt = ground_truth(n_frames, f, z0)
    res = dict(config=dict(frames=n_frames, resolution=[cols, rows], nfeatures=nfeatures, focal_px=f, plane_depth_m=z0, ba_every_keyframes=ba_every,
                           ransac="findFundamentalMat(RANSAC, 2 px, 0.99) / solvePnPRansac(100, 4 px, 0.99), seeds = frame index",
                           pnp=("cv::solvePnPRansac by OpenCV's procedure (EPnP + iterative refit)" if pnp == "cv" else "P3P + LM, own sampler"),
                           depth_relief_mm=relief_mm))
    t0 = time.perf_counter()
    pre, t_phase1 = None, 0.0
    if batched:
        depth = make_depth(rows, cols, z0)
        fe = BatchedFrontEnd(nfeatures, rows, cols)          # handles and buffers: set up once, like the stages' (not per frame)
        fe.run(frames[:2], depth)                            # ... and the first call's lazy workspace allocation / kernel load
        t0 = time.perf_counter()
        pre = batched_front_end(frames, depth, nfeatures, shards, fe=fe)
        t_phase1 = time.perf_counter() - t0
    if batched:
        hip = track_batched(HipStages(nfeatures), n_frames, cols, rows, f, z0, nfeatures, ba_every, frames, pre)
    else:
        hip = track(HipStages(nfeatures), n_frames, cols, rows, f, z0, nfeatures, ba_every, verbose, frames, pre)
    hip["seconds_in_stages"] += t_phase1
    res["config"]["phases"] = (f"extract + depth filter + match batched on the device ({shards} shard(s), 64 frames per call); fundamental-matrix gates, "
                               "PnP and keyframe-pair match jobs as batches over frames; keyframe chain and pose products sequential" if batched
                               else "one frame per call")
    res["hip"] = dict(wall_s=time.perf_counter() - t0, ms_per_frame_in_stages=1e3 * hip["seconds_in_stages"] / n_frames,
                      ms_per_frame_phase1=1e3 * t_phase1 / n_frames,
                      keyframes=len(hip["keyframes"]), rmse_vs_ground_truth=rmse(hip["poses"], gt),
                      median_matches=float(np.median(hip["stats"]["matches"])), median_geometric=float(np.median(hip["stats"]["geometric"])),
                      median_pnp_inliers=float(np.median(hip["stats"]["pnp_inliers"])), pose_updates=hip["stats"]["pose_updates"],
                      motion_outliers=hip["stats"]["motion_outliers"], pnp_failures=hip["stats"]["pnp_failures"],
                      landmarks=hip["backend"]["landmarks"], ba_runs=len(hip["backend"]["ba"]),
                      ba_converged=sum(1 for b in hip["backend"]["ba"] if b["termination"] == 0))
    if with_cpu:
        t0 = time.perf_counter()
        cpu = track(CpuStages(nfeatures), n_frames, cols, rows, f, z0, nfeatures, ba_every, verbose, frames)
        res["cpu"] = dict(wall_s=time.perf_counter() - t0, ms_per_frame_in_stages=1e3 * cpu["seconds_in_stages"] / n_frames,
                          keyframes=len(cpu["keyframes"]), rmse_vs_ground_truth=rmse(cpu["poses"], gt), landmarks=cpu["backend"]["landmarks"],
                          ba_runs=len(cpu["backend"]["ba"]), ba_converged=sum(1 for b in cpu["backend"]["ba"] if b["termination"] == 0))
        res["hip_vs_cpu"] = dict(rmse=rmse(hip["poses"], cpu["poses"]), same_keyframes=hip["keyframes"] == cpu["keyframes"],
                                 keyframes_in_common=len(set(hip["keyframes"]) & set(cpu["keyframes"])))
        res["_raw"] = dict(hip=hip, cpu=cpu)
    else:
        res["_raw"] = dict(hip=hip)
    return res


def summarize_hip(hip, n_frames, gt, t_phase1, wall):
    return dict(wall_s=wall, ms_per_frame_in_stages=1e3 * hip["seconds_in_stages"] / n_frames, ms_per_frame_phase1=1e3 * t_phase1 / n_frames,
                keyframes=len(hip["keyframes"]), rmse_vs_ground_truth=rmse(hip["poses"], gt),
                median_matches=float(np.median(hip["stats"]["matches"])), median_geometric=float(np.median(hip["stats"]["geometric"])),
                median_pnp_inliers=float(np.median(hip["stats"]["pnp_inliers"])), pose_updates=hip["stats"]["pose_updates"],
                motion_outliers=hip["stats"]["motion_outliers"], pnp_failures=hip["stats"]["pnp_failures"],
                landmarks=hip["backend"]["landmarks"], ba_runs=len(hip["backend"]["ba"]),
                ba_converged=sum(1 for b in hip["backend"]["ba"] if b["termination"] == 0))


def compare_with_golden(hip, path):
    """the CPU oracle pipeline's recorded run of the same sequence (tools/gen_replay_golden.py)"""
    g = np.load(path)
    e = rmse(hip["poses"], list(zip(g["R"], g["t"])))
    return dict(file=os.path.relpath(path, ROOT), same_match_counts=hip["stats"]["matches"] == g["matches"].tolist(),
                same_keyframes=hip["keyframes"] == g["keyframes"].tolist(), pose_rmse_vs_cpu_pipeline=e,
                landmarks_cpu=int(g["landmarks"]), landmarks_hip=int(hip["backend"]["landmarks"]))


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(gpus, argv):
    """`replay_tracking.py --gpus N` outside a launcher: start the N ranks (one process per GPU) as a CHILD process — nothing in this
    process has touched the GPU or imported torch — and exit with its code (as bench.py does)"""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def run_ranks(a):
    """one process per GPU (BASELINE configs[4] "across 8 GPUs"): phase 1 sharded over the ranks, ONE dvs_comm_all_gather, the sequential
    tracking + backend on rank 0"""
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.dry_launch:
        print(json.dumps({"dry_launch": True, "rank": rank, "world": world, "local_rank": local, "pid": os.getpid()}), flush=True)
        return
    import torch  # noqa: F401  (one ROCm stack per process: tests/conftest.py)
    import torch.distributed as dist
    from dvslam_amd import synth
    from dvslam_amd import dist as dvdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")                                     # only the hand-over of the RCCL unique id and the final barrier
    n, cols, rows, nf, f, z0 = a.frames, a.cols, a.rows, a.nfeatures, 600.0, 1.5
    depth = make_depth(rows, cols, z0)
    fe = BatchedFrontEnd(nf, rows, cols, 64, local)                      # handles (and their streams) before the communicator comes up
    fe.run([synth.make_traj_frame(t, cols, rows) for t in range(2)], depth)

    def bcast_id(ident):
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    comm = dvdist.Comm(local, rank, world, bcast_id)
    dist.barrier()
    t0 = time.perf_counter()
    pre = sharded_front_end_rank(lambda t: synth.make_traj_frame(t, cols, rows), n, depth, nf, rank, world, comm, local, 64, fe)
    t_phase1 = time.perf_counter() - t0                                  # includes this rank's share of the frame synthesis
    if rank == 0:
        hip = track_batched(HipStages(nf), n, cols, rows, f, z0, nf, a.ba_every, None, pre)
        hip["seconds_in_stages"] += t_phase1
        r = dict(config=dict(frames=n, resolution=[cols, rows], nfeatures=nf, gpus=world, rccl_version=comm.rccl_version,
                             phases=f"phase 1 sharded over {world} ranks (one process per GPU), one dvs_comm_all_gather (ncclAllGather), tracking on rank 0"),
                 hip=summarize_hip(hip, n, ground_truth(n, f, z0), t_phase1, time.perf_counter() - t0))
        if a.golden:
            r["golden"] = compare_with_golden(hip, a.golden)
        print(json.dumps(r, indent=1), flush=True)
        if a.out:
            json.dump(r, open(a.out, "w"), indent=1)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="N >= 1: one process per GPU (self-launched through torch.distributed.run unless already "
                                                        "under a launcher): phase 1 sharded over the ranks, one all-gather, tracking on rank 0")
    ap.add_argument("--loopback", type=int, default=0, help="N logical ranks of ONE process on one GPU (dvs_comm_create_loopback) instead of N processes")
    ap.add_argument("--dry-launch", action="store_true", help="--gpus: only start the ranks and report them (launcher self-test, no GPU work)")
    ap.add_argument("--golden", default="", help="npz of the CPU oracle pipeline's recorded run to compare with (tests/golden/replay_1000_cpu.npz)")
    ap.add_argument("--ba-every", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--cols", type=int, default=640)
    ap.add_argument("--rows", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--per-frame", action="store_true", help="round 2's form: one frame per call through the host entry points")
    ap.add_argument("--shards", type=int, default=1, help="contiguous frame ranges of phase 1 (one per rank on a multi-GPU node)")
    ap.add_argument("--pnp", choices=("own", "cv"), default="own", help="cv: dvs_solve_pnp_ransac_cv (OpenCV's procedure); use with --relief-mm")
    ap.add_argument("--relief-mm", type=int, default=0, help="depth = plane + a 64-px checker of this height: non-coplanar 3D points (EPnP needs them)")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.gpus >= 1:
        if "WORLD_SIZE" not in os.environ:
            sys.exit(self_launch(a.gpus, sys.argv[1:]))
        run_ranks(a)
        sys.exit(0)
    import torch  # noqa: F401  (one ROCm stack per process: tests/conftest.py)
    if a.loopback >= 1:
        from dvslam_amd import synth
        frames = [synth.make_traj_frame(t, a.cols, a.rows) for t in range(a.frames)]
        depth = np.full((a.rows, a.cols), 1500, np.uint16)
        t0 = time.perf_counter()
        pre = sharded_front_end_loopback(frames, depth, a.nfeatures, a.loopback)[0]
        t1 = time.perf_counter() - t0
        hip = track_batched(HipStages(a.nfeatures), a.frames, a.cols, a.rows, 600.0, 1.5, a.nfeatures, a.ba_every, None, pre)
        hip["seconds_in_stages"] += t1
        r = dict(config=dict(frames=a.frames, loopback_ranks=a.loopback), hip=summarize_hip(hip, a.frames, ground_truth(a.frames, 600.0, 1.5), t1, time.perf_counter() - t0))
        if a.golden:
            r["golden"] = compare_with_golden(hip, a.golden)
        print(json.dumps(r, indent=1))
        sys.exit(0)
    r = run(a.frames, a.cols, a.rows, nfeatures=a.nfeatures, with_cpu=not a.no_cpu, verbose=True, batched=not a.per_frame, shards=a.shards, pnp=a.pnp,
            relief_mm=a.relief_mm)
    r.pop("_raw")
    print(json.dumps(r, indent=1))
    if a.out:
        json.dump(r, open(a.out, "w"), indent=1)
