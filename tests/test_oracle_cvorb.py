"""cv::ORB-compatible mode, CPU side: the oracle restatement (oracle/cvorb_oracle.cpp) against what the
reference's own test asserts (test_dbow2_integration.cpp:41-42), an independent bracket for INTER_LINEAR_EXACT, and the library's
restatement of std::nth_element + std::partition (csrc/lsort.h) against the real libstdc++ routines the oracle calls."""
import numpy as np
import pytest
from dvslam_amd import synth


def disc_image():
    """test_dbow2_integration.cpp:14-17: three filled circles on black (cv::circle's rasteriser may differ on a few rim pixels)"""
    img = np.zeros((480, 640), np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    for cx, cy, r in ((100, 100, 50), (300, 200, 30), (500, 300, 40)):
        img[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = 255
    return img


def test_reference_test_assertions_hold_for_the_oracle(oracle):
    """EXPECT_GT(descriptors.rows, 0); EXPECT_EQ(descriptors.cols, 32) for cv::ORB::create(100) on the disc image"""
    kps, desc = oracle.OracleCvORB(100).detectAndCompute(disc_image())
    assert len(desc) > 0 and desc.shape[1] == 32 and len(kps) == len(desc)
    assert (kps["class_id"] == -1).all() and (kps["octave"] >= 0).all() and (kps["octave"] < 8).all()
    assert ((kps["angle"] >= 0) & (kps["angle"] < 360)).all()


def test_structure_on_a_textured_frame(oracle):
    img = synth.make_frame(0, cols=640, rows=480)
    o = oracle.OracleCvORB(500)
    kps, desc = o.detectAndCompute(img)
    assert 400 <= len(kps) <= 520                      # quotas are met on a textured frame; ties may add a few
    # level-major order, octave-consistent size, points at least edgeThreshold from the level's border
    assert (np.diff(kps["octave"]) >= 0).all()
    for l in range(8):
        s = np.float32(np.float64(np.float32(1.2)) ** l)
        m = kps["octave"] == l
        assert (kps["size"][m] == np.float32(31) * s).all()
        w, h = o.level(l).shape[1], o.level(l).shape[0]
        x, y = kps["x"][m] / s, kps["y"][m] / s
        assert (np.round(x) >= 31).all() and (np.round(x) < w - 31).all() and (np.round(y) >= 31).all() and (np.round(y) < h - 31).all()
    # HARRIS responses are sorted into "the N best" per level: within a level every kept response >= the smallest kept one (trivial),
    # and the level sizes follow cvRound(dim / 1.2^l)
    assert o.level(1).shape == (400, 533) and o.level(7).shape == (134, 179)


def test_linear_exact_resize_is_bilinear(oracle):
    """independent bracket: torch's float bilinear (align_corners=False, the same sample positions) within one gray level"""
    import torch
    img = synth.make_frame(1, cols=640, rows=480)
    got = oracle.resize_linear_exact(img, 533, 400).astype(np.float32)
    ref = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float32))[None, None], size=(400, 533), mode="bilinear",
                                          align_corners=False, antialias=False)[0, 0].numpy()
    assert np.abs(got - ref).max() <= 1.0
    assert np.abs(got - ref).mean() < 0.3


@pytest.mark.parametrize("n,npts,kind", [(10, 3, "rand"), (100, 50, "ties"), (1000, 200, "ties"), (5000, 868, "rand"), (5000, 434, "fewvals"),
                                         (4, 2, "ties"), (3, 3, "rand"), (7, 0, "rand"), (20000, 868, "ints"), (777, 776, "ties"), (64, 1, "ties")])
def test_retain_best_replica_equals_libstdcxx(hiplib, oracle, n, npts, kind):
    """csrc/lsort.h (nth_element = introselect, partition = bidirectional) leaves the same survivors IN THE SAME ORDER as the real
    std::nth_element + std::partition called by the oracle, on inputs with many equal responses (FAST scores are small integers)"""
    from dvslam_amd import cvorb
    rng = np.random.default_rng(n * 31 + npts)
    if kind == "rand":
        r = rng.normal(size=n).astype(np.float32)
    elif kind == "ties":
        r = rng.integers(7, 40, size=n).astype(np.float32)
    elif kind == "fewvals":
        r = rng.choice(np.array([-1.5, 0.0, 0.25, 3.0], np.float32), size=n)
    else:
        r = rng.integers(20, 255, size=n).astype(np.float32)
    want = oracle.retain_best(r, npts)
    got = cvorb.retain_best_host(r, npts)
    assert len(got) == len(want) and (got == want).all()
    if npts > 0 and n > npts:                          # the set: everything not smaller than the n-th largest response
        thr = np.sort(r)[::-1][npts - 1]
        assert sorted(got.tolist()) == sorted(np.nonzero(r >= thr)[0].tolist())


def test_adversarial_depth_limit(hiplib, oracle):
    """median-of-3 killer sequences push introselect to its heap_select fallback"""
    from dvslam_amd import cvorb
    for n in (64, 500, 4096):
        k = n // 2
        r = np.zeros(n, np.float32)                    # Musser's anti-quicksort pattern for median-of-3
        for i in range(k):
            r[i] = i + 1 if i % 2 == 0 else k + i + (1 if i % 2 else 0)
            r[k + i] = 2 * (i + 1)
        r = -r
        for npts in (1, n // 3, n - 1):
            want = oracle.retain_best(r, npts); got = cvorb.retain_best_host(r, npts)
            assert len(got) == len(want) and (got == want).all(), (n, npts)
