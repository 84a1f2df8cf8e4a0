"""End-to-end composition of the C-ABI pieces in the order the two nodes use them (tools/replay_synthetic.py): frontend stages ->
Keyframe.msg CDR payload -> backend unpack -> landmark association -> SlidingWindowBA on the device.  Tracking (PnP / RANSAC, row
N4) is replaced by ground-truth poses plus noise, so this checks the data flow and the numerics of the composition, not SLAM
accuracy: the scene is a plane and all landmarks are free, which leaves camera tilt weakly constrained (from a perfect start
the optimum sits ~0.2 deg / 5 mm away)."""
import os
import sys
import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
pytestmark = pytest.mark.gpu


def test_synthetic_replay_composes(gpu):
    import replay_synthetic as r
    o = r.run(n_keyframes=6, nfeatures=800, pose_noise=(0.004, 0.15), seed=3)
    assert o["keyframes"] == 6 and o["landmarks"] > 400 and o["observations"] > 2 * o["landmarks"] - 1
    assert all(m > 300 for m in o["frontend_matches"])                 # consecutive keyframes share most of their features
    assert o["associations"][0] == 0 and all(a > 100 for a in o["associations"][1:])
    assert np.isfinite(o["ba"]["final_cost"]) and o["ba"]["steps"] >= 5
    assert o["rmse_translation_m"]["after"] < 0.015 and o["rmse_rotation_deg"]["after"] < 0.6
    # deterministic: the same run again gives the same numbers (fixed-order reductions everywhere)
    o2 = r.run(n_keyframes=6, nfeatures=800, pose_noise=(0.004, 0.15), seed=3)
    assert o2["ba"]["final_cost"] == o["ba"]["final_cost"] and o2["associations"] == o["associations"]


def test_synthetic_replay_from_ground_truth_stays_put(gpu):
    import replay_synthetic as r
    o = r.run(n_keyframes=6, nfeatures=800, pose_noise=(0.0, 0.0), seed=3)
    assert o["ba"]["success"]
    assert o["rmse_translation_m"]["after"] < 0.01 and o["rmse_rotation_deg"]["after"] < 0.4
