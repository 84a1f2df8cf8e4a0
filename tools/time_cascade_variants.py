"""Run ON THE GPU BOX: stage times of the extractor at 1 / 4 / 8 frames for the pyramid cascade's tile sizes (DVS_CASC_TW / DVS_CASC_TH are read
when the handle builds its geometry) with a hash of the outputs, which must not move.  (The 512- and 1024-thread forms of the kernel that
this script also compared in round 4 were not kept: EXPERIMENTS.md.)"""
import sys, os, hashlib, subprocess, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT + "/dynamic-visual-slam_amd")
    import numpy as np
    import dvslam_amd
    from dvslam_amd import synth, _lib
    B = int(sys.argv[2])
    imgs = np.stack([synth.make_frame(i, 1280, 720) for i in range(B)])
    d = _lib.DeviceBuffer(imgs.nbytes).upload(imgs)
    g = dvslam_amd.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=B)
    cap = g.capacity
    k, de, n = _lib.DeviceBuffer(B * cap * 28), _lib.DeviceBuffer(B * cap * 32), _lib.DeviceBuffer(4 * B)
    g.set_overlap(False)
    for it in range(3):
        g.extract_batch_device(d.ptr, B, 720, 1280, 1280, 720 * 1280, k.ptr, de.ptr, cap, n.ptr)
    g.synchronize()
    nn = n.download(np.int32, B)
    kk = k.download(np.uint8, B * cap * 28).reshape(B, cap * 28); dd = de.download(np.uint8, B * cap * 32).reshape(B, cap * 32)
    hk = hashlib.sha1(b"".join(kk[i, :28 * nn[i]].tobytes() + dd[i, :32 * nn[i]].tobytes() for i in range(B))).hexdigest()[:12]
    g.enable_stage_timing(True)
    for it in range(50):
        g.extract_batch_device(d.ptr, B, 720, 1280, 1280, 720 * 1280, k.ptr, de.ptr, cap, n.ptr)
    ms, calls = g.stage_times()
    print(json.dumps({"B": B, "env": {k_: v for k_, v in os.environ.items() if k_.startswith("DVS_CASC")}, "hash": hk,
                      "us": {s: round(1e3 * ms[s] / max(calls[s], 1), 1) for s in ms}}))
else:
    for B in (1, 4, 8):
        for tw, thh in [(128, 64), (128, 32), (64, 32), (64, 16), (32, 32)]:
            env = dict(os.environ, DVS_CASC_TW=str(tw), DVS_CASC_TH=str(thh))
            r = subprocess.run([sys.executable, __file__, "one", str(B)], env=env, capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr[-400:], flush=True)
