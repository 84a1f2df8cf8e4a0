"""GPU parity of the cv::ORB-compatible mode: dvs_cvorb_* (csrc/cvorb.hip, through the C-ABI) against
oracle/cvorb_oracle.cpp.  Bar: bit-exact — every pyramid level (INTER_LINEAR_EXACT chain), every blurred level, the keypoints (all 7
cv::KeyPoint fields, float bit patterns, in retainBest's libstdc++ order) and the 256-bit descriptors."""
import numpy as np
import pytest
from dvslam_amd import synth
from test_oracle_cvorb import disc_image

pytestmark = pytest.mark.gpu


def _same(kg, dg, ko, do):
    assert len(kg) == len(ko), (len(kg), len(ko))
    for f in kg.dtype.names:
        assert (kg[f].view(np.uint32) == ko[f].view(np.uint32)).all(), f"keypoint field {f} differs"
    assert (dg == do).all()


def test_disc_image_of_the_reference_test(gpu, oracle):
    """test_dbow2_integration.cpp:14-19, 33-43: cv::ORB::create(100) on three filled discs: rows > 0, 32 columns, and the same rows
    through HIP and oracle"""
    from dvslam_amd import CvORB
    img = disc_image()
    g = CvORB.create(100)
    kg, dg = g.detectAndCompute(img)
    ko, do = oracle.OracleCvORB(100).detectAndCompute(img)
    assert len(dg) > 0 and dg.shape[1] == 32
    _same(kg, dg, ko, do)
    g.close()


@pytest.mark.parametrize("rows,cols,nf,nl,score,frame", [(480, 640, 500, 8, 0, 0), (720, 1280, 2000, 8, 0, 1), (480, 640, 100, 8, 1, 2),
                                                         (361, 487, 300, 5, 0, 3), (720, 1280, 1000, 8, 1, 4), (200, 260, 150, 3, 0, 5),
                                                         (130, 150, 50, 8, 0, 6)])   # the last: upper levels smaller than 2 x edgeThreshold
def test_parity_on_textured_frames(gpu, oracle, rows, cols, nf, nl, score, frame):
    from dvslam_amd import CvORB
    img = synth.make_frame(frame, cols=cols, rows=rows)
    g = CvORB(nf, 1.2, nl, 31, 0, 2, score, 31, 20)
    o = oracle.OracleCvORB(nf, 1.2, nl, 31, score, 20)
    kg, dg = g.detectAndCompute(img)
    ko, do = o.detectAndCompute(img)
    for l in range(nl):
        assert (g.level(l) == o.level(l)).all(), f"pyramid level {l}"
        assert (g.level(l, blurred=True) == o.level(l, blurred=True)).all(), f"blurred level {l}"
    assert len(ko) > 20
    _same(kg, dg, ko, do)
    kg2, dg2 = g.detectAndCompute(img)                  # handle reuse
    _same(kg2, dg2, ko, do)
    g.close()


def test_empty_image_and_unsupported_parameters(gpu):
    from dvslam_amd import CvORB, DvsError
    g = CvORB(100)
    k, d = g.detectAndCompute(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    k, d = g.detectAndCompute(np.full((240, 320), 128, np.uint8))   # nothing to detect: 0 rows (cv::ORB releases the descriptors)
    assert len(k) == 0
    with pytest.raises(DvsError):
        CvORB(100, WTA_K=3)
    with pytest.raises(DvsError):
        CvORB(100, edgeThreshold=10)


@pytest.mark.parametrize("n,npts,kind", [(10, 3, "rand"), (1000, 200, "ties"), (5000, 868, "rand"), (30000, 868, "ints"), (30000, 434, "ties"),
                                         (4, 2, "ties"), (3, 3, "rand"), (7, 0, "rand"), (777, 776, "ties"), (100000, 868, "ints")])
def test_wavefront_retain_best_equals_libstdcxx(gpu, oracle, n, npts, kind):
    """the wavefront-parallel nth_element / partition of the kernels (exact Hoare swaps as rank pairs) against the real libstdc++
    routines, order included"""
    from dvslam_amd import cvorb
    rng = np.random.default_rng(n * 17 + npts)
    r = (rng.normal(size=n) if kind == "rand" else rng.integers(7, 40, size=n) if kind == "ties" else rng.integers(20, 255, size=n)).astype(np.float32)
    want = oracle.retain_best(r, npts)
    got = cvorb.retain_best_device(r, npts)
    assert len(got) == len(want) and (got == want).all()
