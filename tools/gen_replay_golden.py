#!/usr/bin/env python3
"""tests/golden/replay_1000_cpu.npz: the 1000-frame tracking replay (BASELINE configs[4]) through the CPU ORACLE stages only
(tools/replay_tracking.py, CpuStages) — keyframe decisions, per-frame poses, match statistics.  tests/test_tracking_replay.py compares
the HIP pipeline's 1000-frame run with it on the GPU box without spending 74 s of oracle time there.  No GPU needed.
usage: python tools/gen_replay_golden.py"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import replay_tracking as rt  # noqa: E402
from dvslam_amd import synth  # noqa: E402

N, COLS, ROWS, F, Z0, NF, BA_EVERY = 1000, 640, 480, 600.0, 1.5, 1000, 5
frames = [synth.make_traj_frame(t, COLS, ROWS) for t in range(N)]
cpu = rt.track(rt.CpuStages(NF), N, COLS, ROWS, F, Z0, NF, BA_EVERY, True, frames)
R = np.stack([p[0] for p in cpu["poses"]]); T = np.stack([p[1] for p in cpu["poses"]])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "replay_1000_cpu.npz"), keyframes=np.array(cpu["keyframes"], np.int32), R=R, t=T,
                    matches=np.array(cpu["stats"]["matches"], np.int32), geometric=np.array(cpu["stats"]["geometric"], np.int32),
                    associations=np.array(cpu["backend"]["associations"], np.int32), landmarks=np.int32(cpu["backend"]["landmarks"]),
                    config=np.array([N, COLS, ROWS, NF, BA_EVERY], np.int32))
print("keyframes", len(cpu["keyframes"]), "landmarks", cpu["backend"]["landmarks"])
