import sys, os, numpy as np
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0, ROOT+"/dynamic-visual-slam_amd")
import dvslam_amd
from dvslam_amd import synth, _lib
img = synth.make_frame(0, 1280, 720)
d = _lib.DeviceBuffer(img.nbytes).upload(img)
for nl in (2, 3, 4, 6, 8):
    g = dvslam_amd.ORBextractor(2000, 1.2, nl, 20, 7, max_batch=1, hooks=True)   # (scheduling hooks: the test library)
    cap = g.capacity
    k, de, n = _lib.DeviceBuffer(cap*28), _lib.DeviceBuffer(cap*32), _lib.DeviceBuffer(4)
    g.set_overlap(False)
    for it in range(3):
        g.extract_batch_device(d.ptr, 1, 720, 1280, 1280, 720*1280, k.ptr, de.ptr, cap, n.ptr)
    g.synchronize()
    g.enable_stage_timing(True)
    for it in range(50):
        g.extract_batch_device(d.ptr, 1, 720, 1280, 1280, 720*1280, k.ptr, de.ptr, cap, n.ptr)
    ms, calls = g.stage_times()
    print(nl, {s: round(1e3*ms[s]/max(calls[s],1),1) for s in ms})
