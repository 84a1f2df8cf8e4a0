"""GPU parity: HIP extractor (through the C-ABI) vs the CPU oracle, stage by stage and end to end.
Bar: bit-exact — pyramid bytes, blurred bytes, candidate lists (order included), per-level quad-tree
output (order included), final keypoints (all 7 fields, float bits) and 256-bit descriptors."""
import hashlib
import json
import os
import numpy as np
import pytest
from dvslam_amd._lib import test_lib as _hooks   # lib/libdvslam_hip_test.so: the dvs_test_* hooks (not in the product library)
from dvslam_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _pair(oracle, nf, nl, **kw):
    """(HIP extractor, oracle).  hooks=True: the extractor lives in lib/libdvslam_hip_test.so, which also exports the stage introspection
    (candidate lists, per-level keypoints) and the scheduling hooks; the default is the product library"""
    from dvslam_amd import ORBextractor
    return ORBextractor(nf, 1.2, nl, 20, 7, **kw), oracle.OracleORB(nf, 1.2, nl, 20, 7)


def _assert_same_result(n, kps, desc, n2, kps2, desc2):
    assert n == n2
    for f in kps.dtype.names:
        assert (kps[f].view(np.uint32) == kps2[f].view(np.uint32)).all(), f"keypoint field {f} differs"
    assert (desc == desc2).all()


@pytest.mark.parametrize("rows,cols,nf,nl,frame", [(240, 320, 300, 5, 0), (480, 640, 500, 8, 1), (720, 1280, 2000, 8, 0),
                                                   (360, 1000, 700, 6, 3), (600, 400, 1000, 7, 2),
                                                   (481, 643, 600, 6, 1), (250, 330, 200, 4, 4)])  # widths % 4 != 0: generic blur path
def test_stage_and_end_to_end_parity(gpu, oracle, rows, cols, nf, nl, frame):
    img = synth.make_frame(frame, cols=cols, rows=rows)
    g, o = _pair(oracle, nf, nl, hooks=True)     # stage by stage: the test library (same sources + the introspection hooks) ...
    n, kps, desc = g(img)
    n2, kps2, desc2 = o.extract(img)
    gp, _ = _pair(oracle, nf, nl)                # ... and end to end through the PRODUCT library as well
    _assert_same_result(*gp(img), n2, kps2, desc2)
    for l in range(nl):
        assert (gp.level(l) == o.level(l)).all(), f"pyramid level {l} (product library)"
    gp.close()
    for l in range(nl):
        assert (g.level(l) == o.level(l)).all(), f"pyramid level {l}"
        assert (g.candidates(l) == o.candidates(l)).all() if len(o.candidates(l)) == len(g.candidates(l)) else False, f"candidates level {l}"
        lk = o.level_keypoints(l)
        ref = np.stack([lk["x"].astype(np.int32) - 16, lk["y"].astype(np.int32) - 16, lk["response"].astype(np.int32)], axis=1)
        got = g.level_keypoints(l)
        assert got.shape == ref.shape and (got == ref).all(), f"quad-tree level {l}"
        if len(lk):
            assert (g.level(l, blurred=True) == o.level(l, blurred=True)).all(), f"blurred level {l}"
    _assert_same_result(n, kps, desc, n2, kps2, desc2)
    g.close()


def test_reference_frontend_configuration(gpu, oracle):
    """extractor exactly as the frontend builds it: (1000, 1.2f, 8, 20, 7) on 1280x720 (frontend.cpp:205-211)"""
    img = synth.make_frame(5)
    g, o = _pair(oracle, 1000, 8)
    _assert_same_result(*g(img), *o.extract(img))


def test_golden_fixture(gpu):
    meta = json.load(open(os.path.join(GOLD, "orb_320x240.json")))
    img = synth.make_frame(meta["frame"], cols=320, rows=240, seed=meta["seed"])
    assert hashlib.sha256(img.tobytes()).hexdigest() == meta["image_sha256"]
    from dvslam_amd import ORBextractor
    g = ORBextractor(meta["nfeatures"], 1.2, meta["nlevels"], 20, 7, hooks=True)
    n, kps, desc = g(img)
    gold = np.load(os.path.join(GOLD, "orb_320x240.npz"))
    assert n == int(gold["n"]) and kps.tobytes() == gold["kps"].tobytes() and (desc == gold["desc"]).all()
    for l in range(meta["nlevels"]):
        assert (g.candidates(l) == gold[f"cand{l}"]).all()


def test_empty_flat_and_error_behaviour(gpu, oracle):
    from dvslam_amd import ORBextractor, DvsError
    g = ORBextractor(500, 1.2, 8, 20, 7)
    assert g(np.zeros((0, 0), np.uint8))[0] == -1                       # ORBextractor.cpp:1090-1091
    n, kps, desc = g(np.full((480, 640), 128, np.uint8))                 # flat image: zero keypoints, descriptors released
    assert n == 0 and len(kps) == 0 and desc.shape == (0, 32)
    with pytest.raises(DvsError) as e:                                   # sizes where the reference divides by zero
        g(np.zeros((120, 160), np.uint8))
    assert e.value.code == -2
    img = synth.make_frame(0, cols=640, rows=480)                        # handle still usable afterwards
    _assert_same_result(*g(img), *oracle.OracleORB(500, 1.2, 8, 20, 7).extract(img))


def test_threshold_fallback_cells(gpu, oracle):
    """low-contrast image: most cells have no corner at 20 and fall back to 7 (ORBextractor.cpp:843-846)"""
    img = synth.make_frame(0, cols=640, rows=480)
    low = (100 + (img.astype(np.int32) - 128) // 6).astype(np.uint8)
    g, o = _pair(oracle, 800, 8)
    r = g(low); r2 = o.extract(low)
    _assert_same_result(*r, *r2)
    assert (r[1]["response"] < 20).any() and r[0] > 100


@pytest.mark.parametrize("ini,mn", [(5, 12), (20, 20), (40, 3), (7, 20)])
@pytest.mark.parametrize("cols", [640, 643])   # aligned rows: k_fast_wave; odd width: k_fast_cell
def test_threshold_orders_follow_the_two_literal_calls(gpu, oracle, ini, mn, cols):
    """cv::FAST at iniThFAST, then — only for a cell left empty — at minThFAST, whatever the order of the two numbers
    (ORBextractor.cpp:826-846): with minTh > iniTh the second call can only find a subset of nothing"""
    from dvslam_amd import ORBextractor
    img = synth.make_frame(3, cols=cols, rows=480)
    low = (100 + (img.astype(np.int32) - 128) // 3).astype(np.uint8)
    for im in (img, low):
        g = ORBextractor(600, 1.2, 6, ini, mn, hooks=True); o = oracle.OracleORB(600, 1.2, 6, ini, mn)
        r = g(im); r2 = o.extract(im)
        gp = ORBextractor(600, 1.2, 6, ini, mn)     # the product library end to end
        _assert_same_result(*gp(im), *r2); gp.close()
        for l in range(6):
            assert len(g.candidates(l)) == len(o.candidates(l)) and (g.candidates(l) == o.candidates(l)).all(), f"candidates level {l}"
        _assert_same_result(*r, *r2)
        g.close()


def test_noncontiguous_step_and_repeat_determinism(gpu, oracle):
    big = np.zeros((480, 700), np.uint8)
    img = synth.make_frame(2, cols=640, rows=480)
    big[:, 13:653] = img
    view = big[:, 13:653]                     # step 700, unaligned origin
    g, o = _pair(oracle, 500, 8)
    a = g(view); b = g(np.ascontiguousarray(view)); c = o.extract(img)
    _assert_same_result(*a, *b)
    _assert_same_result(*a, *c)


def test_batch_and_device_resident_paths(gpu, oracle):
    from dvslam_amd import ORBextractor
    from dvslam_amd._lib import DeviceBuffer, KP_DTYPE
    rows, cols, nf = 480, 640, 500
    frames = [synth.make_frame(t, cols=cols, rows=rows) for t in range(5)]
    g = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=3)          # 5 frames through batches of 3 + 2
    nout, kps, desc = g.extract_batch(frames)
    o = oracle.OracleORB(nf, 1.2, 8, 20, 7)
    refs = [o.extract(f) for f in frames]
    for i, (n2, k2, d2) in enumerate(refs):
        _assert_same_result(int(nout[i]), kps[i, :nout[i]], desc[i, :nout[i]], n2, k2, d2)
    # device-resident: frames already in HBM, outputs stay in HBM
    cap = g.capacity
    d_img = DeviceBuffer(3 * rows * cols).upload(np.stack(frames[:3]))
    d_k = DeviceBuffer(3 * cap * 28); d_d = DeviceBuffer(3 * cap * 32); d_n = DeviceBuffer(3 * 4)
    g.extract_batch_device(d_img.ptr, 3, rows, cols, cols, rows * cols, d_k.ptr, d_d.ptr, cap, d_n.ptr)
    g.synchronize()
    n3 = d_n.download(np.int32, 3); k3 = d_k.download(KP_DTYPE, 3 * cap).reshape(3, cap); dd = d_d.download(np.uint8, 3 * cap * 32).reshape(3, cap, 32)
    for i in range(3):
        _assert_same_result(int(n3[i]), k3[i, :n3[i]], dd[i, :n3[i]], *refs[i])


def test_full_size_properties(gpu):
    """BASELINE config 2 (1280x720, 2000 kp) through size-independent properties."""
    from dvslam_amd import ORBextractor
    g = ORBextractor(2000, 1.2, 8, 20, 7)
    img = synth.make_frame(7)
    n, kps, desc = g(img)
    quota = g.features_per_level()
    assert 1900 <= n <= 2024
    for l in range(8):
        m = kps["octave"] == l
        assert m.sum() <= quota[l] + 2
        w, h = g.level_size(720, 1280, l)
        s = g.GetScaleFactors()[l]
        assert (kps["x"][m] >= 19 * s).all() and (kps["x"][m] < (w - 19) * s).all()
        assert (kps["y"][m] >= 19 * s).all() and (kps["y"][m] < (h - 19) * s).all()
    assert (np.diff(kps["octave"]) >= 0).all()
    n2, kps2, desc2 = g(img)                      # idempotence
    _assert_same_result(n, kps, desc, n2, kps2, desc2)
    # keypoints are distinct pixels per level
    key = np.stack([kps["octave"], kps["x"].view(np.int32), kps["y"].view(np.int32)], axis=1)
    assert len(np.unique(key, axis=0)) == n


def test_resolution_changes_and_large_batch(gpu, oracle):
    """one handle, changing resolutions (workspace rebuild) and a 20-frame batch through max_batch = 16"""
    from dvslam_amd import ORBextractor
    g = ORBextractor(400, 1.2, 6, 20, 7, max_batch=16)
    o = oracle.OracleORB(400, 1.2, 6, 20, 7)
    for (rows, cols) in [(480, 640), (240, 320), (480, 640), (300, 404)]:
        img = synth.make_frame(1, cols=cols, rows=rows)
        _assert_same_result(*g(img), *o.extract(img))
    frames = [synth.make_frame(t, cols=640, rows=480) for t in range(20)]
    nout, kps, desc = g.extract_batch(frames)
    for i, fr in enumerate(frames):
        n2, k2, d2 = o.extract(fr)
        _assert_same_result(int(nout[i]), kps[i, :nout[i]], desc[i, :nout[i]], n2, k2, d2)


def test_overlap_on_off_identical(gpu):
    from dvslam_amd import ORBextractor
    img = synth.make_frame(3)
    a = ORBextractor(2000, 1.2, 8, 20, 7); b = ORBextractor(2000, 1.2, 8, 20, 7, hooks=True)
    b.set_overlap(False)
    _assert_same_result(*a(img), *b(img))


def test_full_hd_3000_features(gpu, oracle):
    """larger than the BASELINE configs: 1920x1080, 3000 features (more cells, bigger quad-tree quota)"""
    img = synth.make_frame(2, cols=1920, rows=1080)
    g, o = _pair(oracle, 3000, 8)
    _assert_same_result(*g(img), *o.extract(img))


@pytest.mark.gpu
def test_workgroup_sort_matches_std_sort(gpu, hiplib, oracle):
    """The quad-tree's workgroup sort (one wavefront per introsort range, rank-placed leaves) must reproduce std::sort's
    permutation on tie-heavy keys, including the depth-limit heapsort fallback (ORBextractor.cpp:538-553, 700)."""
    rng = np.random.default_rng(9)
    cases = []
    for n in [1, 2, 16, 17, 18, 33, 64, 65, 100, 257, 434, 1000, 1500]:
        for kc, kx in [(2, 2), (3, 40), (50, 5), (1000, 1000), (1, 1)]:
            cases.append((rng.integers(2, 2 + kc, n), rng.integers(0, kx, n) * 7))
    n = 1024
    for _ in range(4):   # median-of-3 killers: heapsort fallback
        base = np.concatenate([np.arange(1, n // 2 + 1, 2), np.arange(n // 2 + 1, n + 1), np.arange(2, n // 2 + 1, 2)])[:n]
        base = np.resize(base, n)
        cases.append((base // rng.integers(1, 4), rng.integers(0, 3, n)))
    cases.append((np.arange(700), np.zeros(700)))
    cases.append((np.arange(700)[::-1], np.zeros(700)))
    for count, ulx in cases:
        count = np.ascontiguousarray(count, np.int32); ulx = np.ascontiguousarray(ulx, np.int32)
        m = len(count)
        a = np.zeros(m, np.int32); b = np.zeros(m, np.int32)
        assert _hooks().dvs_test_sort_nodes_device(count.ctypes.data, ulx.ctypes.data, m, a.ctypes.data) == 0
        oracle.lib().orc_std_sort_nodes(count.ctypes.data, ulx.ctypes.data, m, b.ctypes.data)
        assert (a == b).all(), f"n={m}"


@pytest.mark.gpu
def test_next_batch_hint_is_result_neutral(gpu, oracle):
    """dvs_orb_hint_next_batch_device: the announced batch's pyramid is built beside the current batch's FAST into a second
    buffer and consumed by the next call.  Results must equal the oracle whether the hint is right, wrong (another buffer is
    extracted next) or absent, across alternating batches."""
    from dvslam_amd import ORBextractor
    from dvslam_amd._lib import DeviceBuffer, KP_DTYPE
    rows, cols, nf, B = 240, 320, 300, 2
    frames = [synth.make_frame(t, cols=cols, rows=rows) for t in range(6)]
    o = oracle.OracleORB(nf, 1.2, 8, 20, 7)
    refs = [o.extract(f) for f in frames]
    g = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=B, hooks=True)
    cap = g.capacity
    bufs = [DeviceBuffer(B * rows * cols).upload(np.stack(frames[2 * i:2 * i + 2])) for i in range(3)]
    d_k = DeviceBuffer(B * cap * 28); d_d = DeviceBuffer(B * cap * 32); d_n = DeviceBuffer(B * 4)

    def run(i, hint):
        if hint is not None:
            g.hint_next_batch_device(bufs[hint].ptr)
        g.extract_batch_device(bufs[i].ptr, B, rows, cols, cols, rows * cols, d_k.ptr, d_d.ptr, cap, d_n.ptr)
        g.synchronize()
        n3 = d_n.download(np.int32, B); k3 = d_k.download(KP_DTYPE, B * cap).reshape(B, cap)
        dd = d_d.download(np.uint8, B * cap * 32).reshape(B, cap, 32)
        for j in range(B):
            _assert_same_result(int(n3[j]), k3[j, :n3[j]], dd[j, :n3[j]], *refs[2 * i + j])

    run(0, 1)       # builds batch 1's pyramid beside batch 0
    run(1, 2)       # consumes it, prefetches batch 2
    run(2, 0)       # consumes, prefetches batch 0
    run(1, None)    # wrong guess: batch 1 is extracted, the prefetched pyramid of batch 0 is dropped
    run(0, 0)       # same buffer announced as its own successor (the bench's steady state)
    run(0, None)    # consumes
    run(2, None)    # plain call


@pytest.mark.gpu
def test_randomized_configurations(gpu, oracle):
    """seeded sweep over resolutions (odd and even widths), feature budgets, level counts, scale factors and FAST thresholds,
    each through the host entry point AND the device-resident batch entry point (tight pitch = cols)"""
    from dvslam_amd import ORBextractor
    from dvslam_amd._lib import DeviceBuffer, KP_DTYPE
    rng = np.random.default_rng(2026)
    done = 0
    for it in range(40):
        rows = int(rng.integers(150, 420)); cols = int(rng.integers(200, 560))
        if it % 3 == 0:
            cols &= ~3
        nl = int(rng.integers(2, 9)); nf = int(rng.integers(60, 900))
        sf = float(rng.choice([1.2, 1.2, 1.1, 1.3, 1.44]))
        ini, mn = [(20, 7), (20, 7), (30, 10), (12, 5)][int(rng.integers(0, 4))]
        img = synth.make_frame(int(rng.integers(0, 40)), cols=cols, rows=rows)
        try:
            g = ORBextractor(nf, sf, nl, ini, mn, max_batch=2)
            n, kps, desc = g(img)
        except Exception as e:                                   # sizes the reference itself cannot handle (nCols / nIni == 0)
            assert getattr(e, "code", None) == -2, (rows, cols, nf, nl, sf, e)
            continue
        o = oracle.OracleORB(nf, sf, nl, ini, mn)
        n2, kps2, desc2 = o.extract(img)
        _assert_same_result(n, kps, desc, n2, kps2, desc2)
        cap = g.capacity
        two = np.stack([img, img[::-1].copy()])
        d_img = DeviceBuffer(two.nbytes).upload(two)
        d_k = DeviceBuffer(2 * cap * 28); d_d = DeviceBuffer(2 * cap * 32); d_n = DeviceBuffer(8)
        g.extract_batch_device(d_img.ptr, 2, rows, cols, cols, rows * cols, d_k.ptr, d_d.ptr, cap, d_n.ptr)
        g.synchronize()
        n3 = d_n.download(np.int32, 2); k3 = d_k.download(KP_DTYPE, 2 * cap).reshape(2, cap); dd = d_d.download(np.uint8, 2 * cap * 32).reshape(2, cap, 32)
        _assert_same_result(int(n3[0]), k3[0, :n3[0]], dd[0, :n3[0]], n2, kps2, desc2)
        _assert_same_result(int(n3[1]), k3[1, :n3[1]], dd[1, :n3[1]], *o.extract(two[1]))
        g.close()
        done += 1
    assert done >= 25


@pytest.mark.parametrize("world,rows,cols,nf,nimg", [(3, 480, 640, 1000, 2), (8, 720, 1280, 2000, 1), (2, 240, 320, 500, 3)])
def test_level_sharded_extraction_equals_whole_frame_extraction(gpu, world, rows, cols, nf, nimg):
    """SURVEY.md §8e, small batches: every rank extracts only its pyramid levels into a level-slotted block; the gathered blocks
    merged (dvs_orb_merge_levels_device) are bit-identical to the unsharded extraction.  The `world` ranks run one after the other
    on this one GPU, each with its own handle as on its own GPU; the all-gather itself is the plain byte gather covered by
    test_comm_exchange_single_rank_on_gpu and the gloo tests."""
    from dvslam_amd import ORBextractor, _lib
    from dvslam_amd import dist as dvdist
    frames = np.stack([synth.make_frame(3 + i, cols=cols, rows=rows) for i in range(nimg)])
    d_img = _lib.DeviceBuffer(frames.nbytes).upload(frames)
    ref = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=nimg)
    cap = ref.capacity
    d_k = _lib.DeviceBuffer(nimg * cap * 28); d_d = _lib.DeviceBuffer(nimg * cap * 32); d_n = _lib.DeviceBuffer(nimg * 4)
    ref.extract_batch_device(d_img.ptr, nimg, rows, cols, cols, rows * cols, d_k.ptr, d_d.ptr, cap, d_n.ptr); ref.synchronize()
    n0 = d_n.download(np.int32, nimg); k0 = d_k.download(np.uint8, nimg * cap * 28).reshape(nimg, cap, 28)
    de0 = d_d.download(np.uint8, nimg * cap * 32).reshape(nimg, cap, 32)
    px = [int(np.prod(ref.level_size(rows, cols, l))) for l in range(8)]
    masks = dvdist.level_shards(px, world)
    owner = [next(r for r in range(world) if masks[r] >> l & 1) for l in range(8)]
    bb = ref.level_block_bytes(nimg)
    gathered = _lib.DeviceBuffer(world * bb)                       # what ncclAllGather leaves on every rank: [world][block]
    for r in range(world):
        h = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=nimg)
        assert h.level_block_bytes(nimg) == bb
        h.extract_levels_device(d_img.ptr, nimg, rows, cols, cols, rows * cols, masks[r], gathered.ptr + r * bb); h.synchronize()
    d_k2 = _lib.DeviceBuffer(nimg * cap * 28); d_d2 = _lib.DeviceBuffer(nimg * cap * 32); d_n2 = _lib.DeviceBuffer(nimg * 4)
    ref.merge_levels_device(gathered.ptr, world, owner, nimg, d_k2.ptr, d_d2.ptr, cap, d_n2.ptr); ref.synchronize()
    n1 = d_n2.download(np.int32, nimg); k1 = d_k2.download(np.uint8, nimg * cap * 28).reshape(nimg, cap, 28)
    de1 = d_d2.download(np.uint8, nimg * cap * 32).reshape(nimg, cap, 32)
    assert (n1 == n0).all() and n0.min() > 100
    for f in range(nimg):
        assert (k1[f, :n0[f]] == k0[f, :n0[f]]).all() and (de1[f, :n0[f]] == de0[f, :n0[f]]).all(), f


def test_scheduling_hooks_leave_results_unchanged(gpu):
    """dvs_orb_set_output_event (deferred descriptor stage: the next call's FAST runs beside it) and dvs_orb_set_after_fast_event
    (a caller stream released behind FAST): a pipelined caller gets the same bytes as the plain calls, with the consumer — here the
    match of the previous batch, on its own stream — ordered only by the two events."""
    from dvslam_amd import ORBextractor, BFMatcher, _lib
    L = _lib.lib()
    rows, cols, nf, B, NBATCH = 480, 640, 800, 3, 5
    frames = [np.stack([synth.make_frame(10 * b + i, cols=cols, rows=rows) for i in range(B)]) for b in range(NBATCH)]
    d_img = [_lib.DeviceBuffer(f.nbytes).upload(f) for f in frames]
    plain = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=B)
    cap = plain.capacity
    mk = lambda: (_lib.DeviceBuffer(B * cap * 28), _lib.DeviceBuffer(B * cap * 32), _lib.DeviceBuffer(B * 4))
    ref = []
    for b in range(NBATCH):
        k, d, n = mk()
        plain.extract_batch_device(d_img[b].ptr, B, rows, cols, cols, rows * cols, k.ptr, d.ptr, cap, n.ptr); plain.synchronize()
        ref.append((k.download(np.uint8, B * cap * 28), d.download(np.uint8, B * cap * 32), n.download(np.int32, B)))
    m0 = BFMatcher()
    ridx = _lib.DeviceBuffer(B * cap * 4); rdist = _lib.DeviceBuffer(B * cap * 4)
    # pipelined: outputs by event, match of batch b - 1 on its own stream behind batch b's FAST (the hooks: test library)
    g = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=B, hooks=True)
    mstream = _lib.stream_create(0)
    mat = BFMatcher(stream=mstream)
    ev_out = [_lib.event_create(0) for _ in range(NBATCH)]
    ev_fast = _lib.event_create(0)
    ev_match = [_lib.event_create(0) for _ in range(NBATCH)]
    g.set_after_fast_event(ev_fast)
    outs = [mk() for _ in range(NBATCH)]
    idx = [_lib.DeviceBuffer(B * cap * 4) for _ in range(NBATCH)]; dist = [_lib.DeviceBuffer(B * cap * 4) for _ in range(NBATCH)]
    for b in range(NBATCH):
        k, d, n = outs[b]
        g.set_output_event(ev_out[b], defer=(b % 2 == 0))     # deferred and joined calls alternate
        if b >= 2:
            g.set_reuse_guard_event(ev_match[b - 2])              # (formally: the match that read the oldest outputs)
        if b + 1 < NBATCH:
            g.hint_next_batch_device(d_img[b + 1].ptr)
        g.extract_batch_device(d_img[b].ptr, B, rows, cols, cols, rows * cols, k.ptr, d.ptr, cap, n.ptr)
        if b >= 1:
            assert L.dvs_stream_wait_event(mstream, ev_fast) == 0          # behind batch b's FAST
            assert L.dvs_stream_wait_event(mstream, ev_out[b - 1]) == 0    # batch b - 1 complete (deferred descriptor stage)
            pk = outs[b - 2] if b >= 2 else None
            mat.match_sequence_device(outs[b - 1][1].ptr, outs[b - 1][2].ptr, cap, B, pk[1].ptr + (B - 1) * cap * 32 if pk else 0,
                                      pk[2].ptr + (B - 1) * 4 if pk else 0, idx[b - 1].ptr, dist[b - 1].ptr)
            assert L.dvs_event_record(ev_match[b - 1], mstream) == 0
    g.synchronize(); _lib.stream_synchronize(mstream)
    for b in range(NBATCH):
        k, d, n = outs[b]
        n1 = n.download(np.int32, B)
        assert (n1 == ref[b][2]).all() and n1.min() > 100
        k1 = k.download(np.uint8, B * cap * 28).reshape(B, cap, 28); d1 = d.download(np.uint8, B * cap * 32).reshape(B, cap, 32)
        k0 = ref[b][0].reshape(B, cap, 28); d0 = ref[b][1].reshape(B, cap, 32)
        for f in range(B):
            assert (k1[f, :n1[f]] == k0[f, :n1[f]]).all() and (d1[f, :n1[f]] == d0[f, :n1[f]]).all(), (b, f)
    for b in range(NBATCH - 1):   # matches of batches 0 .. NBATCH - 2 against the same jobs run serially
        pk = outs[b - 1] if b >= 1 else None
        m0.match_sequence_device(outs[b][1].ptr, outs[b][2].ptr, cap, B, pk[1].ptr + (B - 1) * cap * 32 if pk else 0,
                                 pk[2].ptr + (B - 1) * 4 if pk else 0, ridx.ptr, rdist.ptr)
        m0.synchronize()
        n1 = outs[b][2].download(np.int32, B)
        a = idx[b].download(np.int32, B * cap).reshape(B, cap); r = ridx.download(np.int32, B * cap).reshape(B, cap)
        ad = dist[b].download(np.int32, B * cap).reshape(B, cap); rd = rdist.download(np.int32, B * cap).reshape(B, cap)
        for f in range(B):
            if b == 0 and f == 0:
                continue                      # no predecessor: nothing is written for it
            assert (a[f, :n1[f]] == r[f, :n1[f]]).all() and (ad[f, :n1[f]] == rd[f, :n1[f]]).all(), (b, f)
    g.set_output_event(0, defer=False); g.set_after_fast_event(0)
    for e in ev_out + ev_match + [ev_fast]:
        L.dvs_event_destroy(e)
    _lib.stream_destroy(mstream)


@pytest.mark.parametrize("env", [{"DVS_BLUR_MFMA": "1"}, {"DVS_BLUR_MFMA": "2"}, {"DVS_FAST_BYTE_DMA": "0"}, {"DVS_BLUR_MFMA": "1", "DVS_CASCADE": "0"}, {"DVS_CASCADE": "1"},
                                 {"DVS_HOST_POLL": "0"}, {"DVS_OCT_T": "256"}, {"DVS_OCT_T": "512", "DVS_CASCADE": "0"}, {"DVS_NO_OVERLAP": "1"}, {"DVS_DESC_ORDER": "0"}])
@pytest.mark.parametrize("rows,cols,nf,nl", [(480, 640, 500, 8), (720, 1280, 2000, 8), (360, 1000, 700, 6), (250, 332, 200, 4), (200, 136, 150, 3)])
def test_opt_in_kernel_variants_are_bit_identical(gpu, oracle, env, rows, cols, nf, nl):
    """Every switch dvs_orb_create reads from the environment (csrc/orb.hip: the matrix-core blur, the dword-aligned FAST tile origin,
    the one-launch pyramid cascade on / off, the copy-command result path of the host entry points, the quad-tree workgroup size,
    no stream overlap, the descriptor stage visiting keypoints in list order instead of tile by tile) must reproduce the oracle bit for bit — blurred levels, candidates and the final result — including widths
    that are not a multiple of the 32-column strips / 128-column super-strips and rows not a multiple of 32"""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        g, o = _pair(oracle, nf, nl, max_batch=2, hooks=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    frames = [synth.make_frame(t, cols=cols, rows=rows) for t in (2, 7)]
    nout, kps, desc = g.extract_batch(frames)
    for i, img in enumerate(frames):
        n2, k2, d2 = o.extract(img)
        _assert_same_result(int(nout[i]), kps[i, :nout[i]], desc[i, :nout[i]], n2, k2, d2)
    n, k1, d1 = g(frames[1])
    n2, k2, d2 = o.extract(frames[1])
    for l in range(nl):
        if len(o.level_keypoints(l)):
            assert (g.level(l, blurred=True) == o.level(l, blurred=True)).all(), f"blurred level {l}"
        assert len(o.candidates(l)) == len(g.candidates(l)) and (g.candidates(l) == o.candidates(l)).all(), f"candidates level {l}"
    _assert_same_result(n, k1, d1, n2, k2, d2)
    g.close()


def test_deferred_stage_held_back_two_steps(gpu):
    """A deferred descriptor stage that is held back (here: its reuse guard completes 50 ms late) while the next calls' FAST and
    quad-tree run on: what it still reads — its pyramid, the blurred block, ITS level keypoint lists — must not be rewritten under
    it.  (Round 2 rotated two list sets: the quad-tree of call k + 2 overwrote the lists stage k was still reading.)  Every batch
    must equal the plain, unpipelined extraction."""
    import ctypes as C
    from dvslam_amd import ORBextractor, _lib
    L = _lib.lib()
    rows, cols, nf, B, NBATCH = 480, 640, 800, 3, 7
    frames = [np.stack([synth.make_frame(10 * b + i, cols=cols, rows=rows) for i in range(B)]) for b in range(NBATCH)]
    d_img = [_lib.DeviceBuffer(f.nbytes).upload(f) for f in frames]
    plain = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=B)
    cap = plain.capacity
    mk = lambda: (_lib.DeviceBuffer(B * cap * 28), _lib.DeviceBuffer(B * cap * 32), _lib.DeviceBuffer(B * 4))
    ref = []
    for b in range(NBATCH):
        k, d, n = mk()
        plain.extract_batch_device(d_img[b].ptr, B, rows, cols, cols, rows * cols, k.ptr, d.ptr, cap, n.ptr); plain.synchronize()
        ref.append((k.download(np.uint8, B * cap * 28), d.download(np.uint8, B * cap * 32), n.download(np.int32, B)))
    g = ORBextractor(nf, 1.2, 8, 20, 7, max_batch=B, hooks=True)
    side = _lib.stream_create(0)
    ev_out = [_lib.event_create(0) for _ in range(NBATCH)]
    ev_slow = _lib.event_create(0)
    outs = [mk() for _ in range(NBATCH)]
    held = 1                                                      # the call whose descriptor stage is held back
    for b in range(NBATCH):
        k, d, n = outs[b]
        g.set_output_event(ev_out[b], defer=True)
        if b == held:
            assert _hooks().dvs_test_stream_delay(side, 50000) == 0      # 50 ms: hundreds of calls of this size
            assert L.dvs_event_record(ev_slow, side) == 0
            g.set_reuse_guard_event(ev_slow)
        if b + 1 < NBATCH:
            g.hint_next_batch_device(d_img[b + 1].ptr)
        g.extract_batch_device(d_img[b].ptr, B, rows, cols, cols, rows * cols, k.ptr, d.ptr, cap, n.ptr)
        if b == held + 3:
            done = C.c_int32(-1)
            assert L.dvs_event_query(ev_out[held], C.byref(done)) == 0
            assert done.value == 0, "the held-back stage finished before three more calls were enqueued: the test did not delay it"
    g.synchronize(); _lib.stream_synchronize(side)
    for b in range(NBATCH):
        k, d, n = outs[b]
        n1 = n.download(np.int32, B)
        assert (n1 == ref[b][2]).all() and n1.min() > 100, b
        k1 = k.download(np.uint8, B * cap * 28).reshape(B, cap, 28); d1 = d.download(np.uint8, B * cap * 32).reshape(B, cap, 32)
        k0 = ref[b][0].reshape(B, cap, 28); d0 = ref[b][1].reshape(B, cap, 32)
        for f in range(B):
            assert (k1[f, :n1[f]] == k0[f, :n1[f]]).all() and (d1[f, :n1[f]] == d0[f, :n1[f]]).all(), (b, f)
    g.set_output_event(0, defer=False)
    for e in ev_out + [ev_slow]:
        L.dvs_event_destroy(e)
    _lib.stream_destroy(side)
