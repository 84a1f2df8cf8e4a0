import sys, numpy as np
sys.path.insert(0, "dynamic-visual-slam_amd")
import dvslam_amd
from dvslam_amd import synth
e = dvslam_amd.ORBextractor(2000, 1.2, 8, 20, 7, device=0, max_batch=64)
imgs = np.stack([synth.make_frame(i) for i in range(64)])
for _ in range(3):
    e.extract_batch(imgs)
