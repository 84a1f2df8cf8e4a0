"""CPU: pins the oracle (oracle/orb_oracle.cpp) to every known answer derivable from the reference
text and to the committed golden fixtures.  The reference holds no golden vectors for
this path, so parity stays "unpinned" beyond these."""
import hashlib
import json
import os
import numpy as np
import pytest
from dvslam_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_level_sizes_quotas_umax(oracle):
    o = oracle.OracleORB(2000, 1.2, 8, 20, 7)
    assert [o.level_size(1280, 720, l) for l in range(8)] == [(1280, 720), (1067, 600), (889, 500), (741, 417),
                                                               (617, 347), (514, 289), (429, 241), (357, 201)]
    assert [o.level_size(640, 480, l) for l in range(8)] == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231),
                                                             (257, 193), (214, 161), (179, 134)]
    _, _, f, um = o.tables()
    assert f.tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert um.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert oracle.OracleORB(1000, 1.2, 8, 20, 7).tables()[2].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.OracleORB(500, 1.2, 8, 20, 7).tables()[2].tolist() == [109, 90, 75, 63, 52, 44, 36, 31]


def test_pattern_table(oracle):
    p = np.ctypeslib.as_array(oracle.lib().orc_brief_pattern(), shape=(1024,)).copy()
    assert p.min() == -13 and p.max() == 12
    assert p[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12]          # ORBextractor.cpp:151-152
    assert p[-8:].tolist() == [7, 0, 12, -2, -1, -6, 0, -11]      # ORBextractor.cpp:405-406
    assert hashlib.sha256(p.astype(np.int8).tobytes()).hexdigest() == \
        "2164181aea6ff9ac426ca512d5130d15e1f6e3cd47b1cbdd568bbe1e55d49023"


def test_empty_and_unsupported(oracle):
    o = oracle.OracleORB(500, 1.2, 8, 20, 7)
    assert o.extract(np.zeros((0, 0), np.uint8))[0] == -1      # ORBextractor.cpp:1090-1091
    assert o.extract(np.zeros((120, 160), np.uint8))[0] == -2  # level 7 is 45x33: nCols == 0 in the reference (UB)


def test_fast_known_answers(oracle):
    L = oracle.lib()
    img = np.full((32, 32), 100, np.uint8)
    out = np.zeros((64, 3), np.int32)
    assert L.orc_fast(oracle._p(img), 32, 32, 32, 20, oracle._p(out), 64) == 0     # flat image
    img[16, 16] = 200                                                             # isolated bright pixel:
    n = L.orc_fast(oracle._p(img), 32, 32, 32, 20, oracle._p(out), 64)            # all 16 ring px darker by 100
    assert n == 1 and out[0].tolist() == [16, 16, 99]                             # score = min|d| - 1
    # a 9-arc exactly: ring positions 0..8 darker by 50, the rest equal -> corner; 8-arc -> not
    for arc, expect in ((9, 1), (8, 0)):
        img = np.full((32, 32), 100, np.uint8)
        ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
                (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
        for k in range(arc):
            img[16 + ring[k][1], 16 + ring[k][0]] = 50
        n = L.orc_fast(oracle._p(img), 32, 32, 32, 20, oracle._p(out), 64)
        got = [tuple(r) for r in out[:n].tolist() if r[0] == 16 and r[1] == 16]
        assert len(got) == expect
        if expect:
            assert got[0][2] == 49


def test_gauss_and_resize_properties(oracle):
    L = oracle.lib()
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int32)
    img = np.full((40, 50), 77, np.uint8)
    out = np.zeros_like(img)
    L.orc_gauss7(oracle._p(img), 50, 40, oracle._p(out), oracle._p(k))
    assert (out == 77).all()                     # kernel sums to 256 -> constants are preserved
    imp = np.zeros((21, 21), np.uint8); imp[10, 10] = 255
    out = np.zeros_like(imp)
    L.orc_gauss7(oracle._p(imp), 21, 21, oracle._p(out), oracle._p(k))
    exp = (255 * np.outer(k, k) + 32768) >> 16   # impulse response = rounded outer product
    assert (out[7:14, 7:14] == exp).all()
    src = np.full((60, 72), 131, np.uint8); dst = np.zeros((50, 60), np.uint8)
    L.orc_resize_linear_u8(oracle._p(src), 72, 60, 72, oracle._p(dst), 60, 50, 60)
    assert (dst == 131).all()


def test_fast_atan2(oracle):
    L = oracle.lib()
    assert L.orc_fast_atan2(0.0, 1.0) == 0.0
    assert abs(L.orc_fast_atan2(1.0, 1.0) - 45.0) < 0.01
    assert abs(L.orc_fast_atan2(1.0, 0.0) - 90.0) < 0.01
    assert abs(L.orc_fast_atan2(-1.0, -1.0) - 225.0) < 0.01
    rng = np.random.default_rng(0)
    y = rng.integers(-20000, 20000, 500).astype(np.float32); x = rng.integers(-20000, 20000, 500).astype(np.float32)
    got = np.array([L.orc_fast_atan2(float(a), float(b)) for a, b in zip(y, x)])
    ref = np.degrees(np.arctan2(y.astype(np.float64), x.astype(np.float64))) % 360
    assert np.abs(((got - ref + 180) % 360) - 180).max() < 0.02   # documented accuracy ~0.3 deg; typically 0.01


@pytest.mark.parametrize("shape,nf", [((480, 640), 500), ((720, 1280), 2000)])
def test_extract_structural_properties(oracle, shape, nf):
    img = synth.make_frame(0, cols=shape[1], rows=shape[0])
    o = oracle.OracleORB(nf, 1.2, 8, 20, 7)
    n, kps, desc = o.extract(img)
    assert n == len(kps) == len(desc) and desc.shape[1] == 32
    quota = o.tables()[2]
    assert nf * 0.6 <= n <= nf + 3 * 8
    prev_oct = -1
    for l in range(8):
        lk = o.level_keypoints(l)
        w, h = o.level_size(shape[1], shape[0], l)
        assert len(lk) <= quota[l] + 2
        assert ((lk["x"] >= 19) & (lk["x"] < w - 19) & (lk["y"] >= 19) & (lk["y"] < h - 19)).all()
        assert ((lk["angle"] >= 0) & (lk["angle"] <= 360)).all()
    assert (np.diff(kps["octave"]) >= 0).all()            # level-major output order
    assert (kps["class_id"] == -1).all()
    s = o.tables()[0]
    assert (kps["size"] == np.array([float(int(31 * s[o_])) for o_ in kps["octave"]], np.float32)).all()


def test_golden_fixture_oracle(oracle):
    """Fixtures generated by tools/gen_golden.py from THIS oracle (the reference ships none)."""
    meta = json.load(open(os.path.join(GOLD, "orb_320x240.json")))
    img = synth.make_frame(meta["frame"], cols=320, rows=240, seed=meta["seed"])
    assert hashlib.sha256(img.tobytes()).hexdigest() == meta["image_sha256"]
    o = oracle.OracleORB(meta["nfeatures"], 1.2, meta["nlevels"], 20, 7)
    n, kps, desc = o.extract(img)
    gold = np.load(os.path.join(GOLD, "orb_320x240.npz"))
    assert n == int(gold["n"])
    assert kps.tobytes() == gold["kps"].tobytes()
    assert (desc == gold["desc"]).all()


def test_fast_against_independent_numpy_definition(oracle):
    """Second, independent restatement of FAST-9/16 straight from its definition (Rosten & Drummond; score = largest
    threshold for which the pixel still has a 9-arc, as cv::FAST's cornerScore returns), incl. 3x3 NMS with a zero border:
    guards the oracle's transcription of OpenCV's early-exit loops."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (40, 47), dtype=np.uint8)
    img[10:25, 12:30] = 200; img[18:35, 5:20] = 30           # some real corners besides the noise
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    H, W = img.shape
    I = img.astype(np.int32)
    for t in (20, 7, 60):
        score = np.zeros((H, W), np.int32)
        for y in range(3, H - 3):
            for x in range(3, W - 3):
                d = np.array([I[y, x] - I[y + dy, x + dx] for dx, dy in ring])
                best = -1
                for s in range(16):
                    arc = d[[(s + i) % 16 for i in range(9)]]
                    best = max(best, arc.min(), (-arc).min())       # darker arc: all d > thr ; brighter: all -d > thr
                if best > t:                                         # corner at t  <=>  some arc strictly beyond t
                    score[y, x] = best - 1
        exp = []
        for y in range(3, H - 3):
            for x in range(3, W - 3):
                s = score[y, x]
                if s > 0:                                            # a corner at t (score >= t > 0)
                    nb = score[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                    if s > nb.max():                                 # strict 3x3 maximum, non-corners count as 0
                        exp.append((x, y, s))
        out = np.zeros((4096, 3), np.int32)
        n = oracle.lib().orc_fast(oracle._p(img), W, H, W, t, oracle._p(out), 4096)
        assert [tuple(r) for r in out[:n].tolist()] == exp, t
