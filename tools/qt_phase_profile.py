"""Run ON THE GPU BOX with a library built with -DDVS_QT_PROF (see tools/qt_phase_profile.sh): 100 MHz time stamps of the level-0 quad-tree
of frame 0, phase by phase -> microseconds."""
import sys, os, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
os.environ["DVSLAM_HIP_SO"] = ROOT + "/dynamic-visual-slam_amd/lib/libdvslam_hip_prof.so"
sys.path.insert(0, ROOT + "/dynamic-visual-slam_amd")
import numpy as np
import dvslam_amd
from dvslam_amd import synth, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
imgs = np.stack([synth.make_frame(i, 1280, 720) for i in range(B)])
d = _lib.DeviceBuffer(imgs.nbytes).upload(imgs)
g = dvslam_amd.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=B, hooks=True)   # (scheduling hooks: the test library)
cap = g.capacity
k, de, n = _lib.DeviceBuffer(B * cap * 28), _lib.DeviceBuffer(B * cap * 32), _lib.DeviceBuffer(4 * B)
g.set_overlap(False)
for it in range(4):
    g.extract_batch_device(d.ptr, B, 720, 1280, 1280, 720 * 1280, k.ptr, de.ptr, cap, n.ptr)
g.synchronize()
out = np.zeros(96, np.uint64)
L = _lib.lib()
L.dvs_prof_qt.argtypes = [C.c_void_p]
assert L.dvs_prof_qt(out.ctypes.data) == 0
t = out.astype(np.int64)
us = lambda a, b: (t[b] - t[a]) / 100.0
print(f"candidates {t[92]}, nodes {t[93]}, last stamp {t[94]}")
print(f"gather {us(0, 1):.2f}  roots {us(1, 2):.2f}")
i = 3; prev = 2; s = 0
last = int(t[94])
while i + 2 < min(last, 40) + 1 and t[i] and i < 40:
    print(f"sweep {s}: count {us(prev, i):.2f}  scan {us(i, i + 1):.2f}  rebuild {us(i + 1, i + 2):.2f}")
    prev = i + 2; i += 3; s += 1
if last > 40:
    print(f"ordered phase: list {us(40, 41):.2f}")
    j = 41
    while j + 4 < last + 1:
        print(f"  iteration: count+fill {us(j, j + 1):.2f}  sort {us(j + 1, j + 2):.2f}  select {us(j + 2, j + 3):.2f}  rebuild {us(j + 3, j + 4):.2f}")
        j += 5 if False else 4
        if j < last: j += 0
    print(f"  (stamps 40..{last - 1}: {[round((t[x] - t[40]) / 100.0, 2) for x in range(40, last)]})")
print(f"final {us(90, 91):.2f}   total {us(0, 91):.2f}   (main loop {us(2, 90):.2f})")
