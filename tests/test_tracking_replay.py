"""The per-frame logic of the two nodes with the tracking stages in place (tools/replay_tracking.py: extraction, depth filter,
match, fundamental-matrix RANSAC, feature culling, PnP RANSAC + pose accumulation, keyframe decision, Keyframe.msg, backend
association, sliding-window BA) on the bounded-trajectory synthetic sequence.  The 1000-frame run of BASELINE configs[4] is
recorded in profiles/r02_replay_1000.json; here a short prefix keeps the suite fast.  Tolerances (floating point, RANSAC):
HIP against the CPU oracle pipeline <= 3 mm / 0.1 deg RMS over 60 frames with identical keyframe decisions; either against the
closed-form ground truth <= 5 cm / 2 deg (frame-to-frame visual odometry on a fronto-parallel plane drifts in tilt)."""
import os
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_trajectory_generator_is_bounded_and_consistent():
    from dvslam_amd import synth
    ts = np.arange(0, 5000)
    th, ox, oy = synth.traj_state(ts)
    assert np.abs(np.degrees(th)).max() <= 12.0 + 1e-9 and np.abs(ox).max() <= 240.0 and np.abs(oy).max() <= 110.0
    assert np.hypot(np.diff(ox), np.diff(oy)).max() < 7.0 and np.abs(np.degrees(np.diff(th))).max() < 0.26
    a = synth.make_traj_frame(7, 320, 240); b = synth.make_traj_frame(7, 320, 240)
    assert a.shape == (240, 320) and a.dtype == np.uint8 and (a == b).all() and (synth.make_traj_frame(8, 320, 240) != a).any()
    R, T = synth.traj_pose(40, 600.0, 1.5)
    assert np.allclose(R @ R.T, np.eye(3)) and T[2] == 0.0


def test_cpu_pipeline_tracks_the_ground_truth(oracle):
    import replay_tracking as rt
    from dvslam_amd import synth
    n = 12
    frames = [synth.make_traj_frame(t, 640, 480) for t in range(n)]
    cpu = rt.track(rt.CpuStages(1000), n, 640, 480, 600.0, 1.5, 1000, 5, False, frames)
    e = rt.rmse(cpu["poses"], rt.ground_truth(n, 600.0, 1.5))
    assert cpu["stats"]["pose_updates"] == n - 1 and cpu["stats"]["motion_outliers"] == 0 and cpu["keyframes"][0] == 0
    assert min(cpu["stats"]["geometric"]) > 300 and min(cpu["stats"]["pnp_inliers"]) > 250
    assert e["translation_m"] < 0.02 and e["rotation_deg"] < 0.8, e


@pytest.mark.gpu
def test_hip_pipeline_against_cpu_pipeline_and_ground_truth(gpu, oracle):
    import replay_tracking as rt
    r = rt.run(n_frames=60, ba_every=2, with_cpu=True)
    raw = r.pop("_raw")
    assert r["hip_vs_cpu"]["same_keyframes"] and r["hip"]["keyframes"] >= 2
    assert r["hip_vs_cpu"]["rmse"]["translation_m"] < 3e-3 and r["hip_vs_cpu"]["rmse"]["rotation_deg"] < 0.1, r["hip_vs_cpu"]
    for side in ("hip", "cpu"):
        assert r[side]["rmse_vs_ground_truth"]["translation_m"] < 0.05 and r[side]["rmse_vs_ground_truth"]["rotation_deg"] < 2.0, r[side]
    assert r["hip"]["pose_updates"] == 59 and r["hip"]["pnp_failures"] == 0 and r["hip"]["landmarks"] == r["cpu"]["landmarks"]
    # the extraction / match / glue stages are bit-exact, so both pipelines see the same matches; only the RANSAC stages differ
    assert raw["hip"]["stats"]["matches"] == raw["cpu"]["stats"]["matches"]
    assert raw["hip"]["backend"]["associations"] == raw["cpu"]["backend"]["associations"]
