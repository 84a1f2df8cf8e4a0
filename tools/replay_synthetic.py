#!/usr/bin/env python3
"""Synthetic stand-in for bag_playback.launch.xml, WITHOUT the tracking stages (solvePnPRansac / findFundamentalMat are row N4):
keyframe poses come from the generator's closed-form ground truth plus noise, everything in between runs through the C-ABI
exactly in the order the two nodes use it:

  frontend (frontend.cpp:1076-1132, 699-790)   BGR -> gray -> ORB extract -> match vs previous keyframe (+ distance filter)
                                               -> depth filter -> publishKeyframe as Keyframe.msg CDR bytes
  backend  (backend.cpp:709-832, 1064-1173)    unpack the payload -> associate observations with the landmark database
                                               (Hamming < 50, reprojection < 5 px) -> new landmarks for the rest
                                               -> SlidingWindowBA::optimize over the window -> refined poses / landmarks

The scene is the synthetic sequence's textured plane at Z0 in front of a camera that translates in x / y and rolls about its
axis (dvslam_amd/synth.py: frame t samples the canvas at R(0.5 deg * t) (p - c) + c + (3 t, t)), so ground truth is exact.
Reports the pose RMSE against ground truth before and after bundle adjustment."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))


def ground_truth(t, cols, rows, f, z0):
    """camera-to-world (R, T) of frame t: world = canvas plane in metres, origin under the image centre of frame 0"""
    th = np.deg2rad(0.5 * t)
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
    T = np.array([3.0 * t * z0 / f, 1.0 * t * z0 / f, 0.0])
    return R, T


def rot_z_small(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def quat_xyzw(R):
    w = np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2])) / 2.0
    return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])


def run(n_keyframes=6, stride=2, cols=640, rows=480, f=600.0, z0=1.5, nfeatures=800, pose_noise=(0.004, 0.15), seed=3, max_iterations=60,
        reproj_gate=5.0, verbose=False):
    import dvslam_amd
    from dvslam_amd import synth
    from dvslam_amd.glue import unpack_keyframe
    rng = np.random.default_rng(seed)
    cx, cy = cols / 2.0, rows / 2.0
    orb = dvslam_amd.ORBextractor(nfeatures, 1.2, 8, 20, 7)
    mat = dvslam_amd.BFMatcher()
    glue = dvslam_amd.FrontendGlue()
    depth = np.full((rows, cols), int(round(z0 * 1000)), np.uint16)

    # ------------------------------------------------------------------ frontend
    payloads, gt, noisy, prev_desc, match_counts = [], [], [], None, []
    for k in range(n_keyframes):
        t = k * stride
        gray = synth.make_frame(t, cols, rows)
        bgr = np.repeat(gray[:, :, None], 3, axis=2)
        g2 = glue.bgr_to_gray(bgr)                                  # cv::cvtColor (frontend.cpp:1084)
        assert np.array_equal(g2, gray)                             # equal channels -> the same gray image
        n, kps, desc = orb(g2)
        kps, desc, _ = glue.filter_depth(kps, desc, depth)          # filterDepth (frontend.cpp:1100)
        if prev_desc is not None:                                   # matcher_.match + distance < 50 (frontend.cpp:1123-1132)
            idx, dist = mat.match(desc, prev_desc)
            match_counts.append(len(glue.filter_matches(idx, dist, 50.0)))
        prev_desc = desc
        Rg, Tg = ground_truth(t, cols, rows, f, z0)
        if k == 0:
            Rn, Tn = Rg, Tg                                         # the first keyframe is the gauge of the window
        else:
            Rn = Rg @ rot_z_small(np.deg2rad(rng.normal(0, pose_noise[1])))
            Tn = Tg + rng.normal(0, pose_noise[0], 3) * np.array([1.0, 1.0, 0.3])
        gt.append((Rg, Tg)); noisy.append((Rn, Tn))
        payload, m = glue.publish_keyframe(kps, desc, depth, f, f, cx, cy, Rn, Tn, stamp=(t, 0), frame_id="camera_link", keyframe_id=k,
                                           q_xyzw=quat_xyzw(Rn))
        payloads.append(payload)

    # ------------------------------------------------------------------ backend
    db_xyz, db_desc, db_ids = np.zeros((0, 3), np.float32), np.zeros((0, 32), np.uint8), []
    keyframes, observations, next_id, assoc_counts = [], [], 0, []
    for k, payload in enumerate(payloads):
        msg = unpack_keyframe(payload)
        assert msg["keyframe_id"] == k and msg["frame_id"] == "camera_link"
        Rk, Tk = noisy[k]
        assert np.allclose(msg["translation"], Tk) and np.allclose(msg["rotation_xyzw"], quat_xyzw(Rk))
        obs_px = msg["obs_pixels"].astype(np.float32)
        best = (glue.associate(msg["obs_desc"], obs_px, db_desc, db_xyz, Rk, Tk, f, f, cx, cy, 50.0, reproj_gate)
                if len(db_ids) else np.full(len(obs_px), -1, np.int32))
        taken, new_xyz, new_desc, n_assoc = set(), [], [], 0
        for i in range(len(obs_px)):
            j = int(best[i])
            if j >= 0 and j not in taken:                           # one observation per landmark and keyframe
                taken.add(j); lid = db_ids[j]; n_assoc += 1
            else:
                lid = next_id; next_id += 1
                db_ids.append(lid); new_xyz.append(msg["landmark_xyz"][i]); new_desc.append(msg["obs_desc"][i])
            observations.append(((float(obs_px[i, 0]), float(obs_px[i, 1])), lid, "unlabeled", k))
        if new_xyz:
            db_xyz = np.vstack([db_xyz, np.asarray(new_xyz, np.float32)]); db_desc = np.vstack([db_desc, np.asarray(new_desc, np.uint8)])
        keyframes.append((k, Rk, Tk))
        assoc_counts.append(n_assoc)
    # landmarks seen in at least two keyframes constrain the poses; the rest only add free parameters
    seen = {}
    for _, lid, _, fid in observations:
        seen.setdefault(lid, set()).add(fid)
    keep = {lid for lid, fr in seen.items() if len(fr) >= 2}
    slot = {lid: i for i, lid in enumerate(db_ids)}
    landmarks = [(lid, "unlabeled", tuple(float(v) for v in db_xyz[slot[lid]]), False) for lid in sorted(keep)]
    obs = [o for o in observations if o[1] in keep]
    ba = dvslam_amd.SlidingWindowBA(f, f, cx, cy)
    out = ba.optimize(keyframes, landmarks, obs, max_iterations)

    def rmse(poses):
        # one pose is the gauge, the global scale of the window stays free (as in the reference: only keyframes[0] is held
        # constant, bundle_adjustment.hpp:781-785): compare after the least-squares scale about the gauge keyframe
        a = np.array([poses[k][1] - poses[0][1] for k in range(1, n_keyframes)]); b = np.array([gt[k][1] - gt[0][1] for k in range(1, n_keyframes)])
        sc = float((a * b).sum() / max((a * a).sum(), 1e-30))
        e = [np.linalg.norm(sc * (poses[k][1] - poses[0][1]) - (gt[k][1] - gt[0][1])) for k in range(1, n_keyframes)]
        a = [np.degrees(np.arccos(np.clip((np.trace(poses[k][0].T @ gt[k][0]) - 1) / 2, -1, 1))) for k in range(1, n_keyframes)]
        return float(np.sqrt(np.mean(np.square(e)))), float(np.sqrt(np.mean(np.square(a))))

    before = rmse(noisy)
    after = rmse([out["optimized_poses"][k] for k in range(n_keyframes)])
    res = dict(keyframes=n_keyframes, landmarks=len(landmarks), observations=len(obs), frontend_matches=match_counts, associations=assoc_counts,
               ba=dict(success=out["success"], final_cost=out["final_cost"], steps=out["iterations_completed"], message=out["message"]),
               rmse_translation_m=dict(before=before[0], after=after[0]), rmse_rotation_deg=dict(before=before[1], after=after[1]))
    if verbose:
        import json
        print(json.dumps(res, indent=1))
    return res


if __name__ == "__main__":
    run(verbose=True)
