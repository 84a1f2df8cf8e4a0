#!/usr/bin/env python3
"""Generate tests/golden/* from the CPU oracle (oracle/).  The reference ships no golden vectors and
cannot be built or imported here (needs OpenCV/Ceres/ROS 2), so these fixtures pin the ORACLE — they
detect drift of the oracle and give the GPU tests a second, file-based comparison target."""
import hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import oracle_bindings as ob
from dvslam_amd import synth
ob.build()
G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)

meta = dict(frame=0, seed=1234, nfeatures=300, nlevels=5)
img = synth.make_frame(meta["frame"], cols=320, rows=240, seed=meta["seed"])
meta["image_sha256"] = hashlib.sha256(img.tobytes()).hexdigest()
o = ob.OracleORB(meta["nfeatures"], 1.2, meta["nlevels"], 20, 7)
n, kps, desc = o.extract(img)
assert n > 100, n
np.savez_compressed(os.path.join(G, "orb_320x240.npz"), n=n, kps=kps, desc=desc,
                    **{f"cand{l}": o.candidates(l) for l in range(meta["nlevels"])})
json.dump(meta, open(os.path.join(G, "orb_320x240.json"), "w"), indent=1)
print("orb_320x240:", n, "keypoints")

q = synth.make_descriptors(300, 7); t = synth.make_descriptors(257, 8)
t[5] = q[3]; t[100] = q[3]; t[17] = q[9]          # exact duplicates -> tie-break on lowest index
idx, d = ob.match(q, t)
np.savez_compressed(os.path.join(G, "match_300x257.npz"), q=q, t=t, idx=idx, dist=d)
print("match:", d.min(), d.max())
