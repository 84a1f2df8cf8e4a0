import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-visual-slam_amd"))
import numpy as np
import dvslam_amd
from dvslam_amd import synth
L = dvslam_amd.lib()
for (rows, cols, nf) in ((480, 640, 500), (720, 1280, 2000)):
    img = synth.make_frame(1, cols=cols, rows=rows)
    g = dvslam_amd.ORBextractor(nf, 1.2, 8, 20, 7)
    for _ in range(5): g(img)
    print(rows, cols, "graph active:", L.dvs_test_graph_active(g._h))
    t0 = time.perf_counter()
    for _ in range(200): g(img)
    dt = (time.perf_counter() - t0) / 200
    # pieces: host staging memcpy alone
    buf = np.empty_like(img)
    t0 = time.perf_counter()
    for _ in range(200): np.copyto(buf, img)
    dc = (time.perf_counter() - t0) / 200
    print(f"  extract {dt*1e3:.3f} ms per frame; a {img.nbytes/1e6:.2f} MB host memcpy alone {dc*1e3:.3f} ms")
