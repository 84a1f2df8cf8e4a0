"""The oracle is unpinned (the reference holds no golden vectors and OpenCV is absent), so its OpenCV restatements are bracketed
from the outside with independent FLOAT implementations that ship in this image: they cannot fix the last bit (OpenCV's 8-bit
paths are fixed point) but they catch a wrong sampling grid, border rule, kernel or patch geometry."""
import numpy as np
import pytest
from dvslam_amd import synth


def test_resize_against_torch_bilinear(oracle):
    """cv::resize INTER_LINEAR samples at (x + 0.5) * scale - 0.5 with edge clamping = torch's bilinear, align_corners=False;
    OpenCV's Q11 coefficients + two rounding steps stay within 1 gray level of the float result."""
    torch = pytest.importorskip("torch")
    L = oracle.lib()
    img = synth.make_frame(2, cols=640, rows=480)
    for (dw, dh) in [(533, 400), (444, 333), (320, 240), (639, 479)]:
        dst = np.zeros((dh, dw), np.uint8)
        L.orc_resize_linear_u8(oracle._p(img), 640, 480, 640, oracle._p(dst), dw, dh, dw)
        ref = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float64))[None, None], size=(dh, dw), mode="bilinear",
                                              align_corners=False)[0, 0].numpy()
        err = np.abs(dst.astype(np.float64) - ref)
        assert err.max() <= 1.0 + 1e-9, (dw, dh, err.max())
        assert (err > 0.75).mean() < 0.02          # and it is a rounding of it, not a shifted grid


def test_gaussian_blur_against_scipy(oracle):
    """cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101): the normalised float kernel exp(-x^2 / 8) with 'mirror' borders."""
    ndi = pytest.importorskip("scipy.ndimage")
    L = oracle.lib()
    k = np.array([18, 34, 48, 56, 48, 34, 18], np.int32)
    g = np.exp(-np.arange(-3, 4) ** 2 / 8.0); g /= g.sum()
    # the Q8 kernel is that Gaussian up to the rounding OpenCV's bit-exact kernel generator distributes so that the sum stays 256
    # (straight rounding would give 18 34 49 55 49 34 18 = 257)
    assert np.abs(k / 256.0 - g).max() < 0.004 and k.sum() == 256
    img = synth.make_frame(5, cols=320, rows=240)
    out = np.zeros_like(img)
    L.orc_gauss7(oracle._p(img), 320, 240, oracle._p(out), oracle._p(k))
    ref = ndi.correlate1d(ndi.correlate1d(img.astype(np.float64), g, axis=1, mode="mirror"), g, axis=0, mode="mirror")
    err = np.abs(out.astype(np.float64) - ref)                   # borders included: REFLECT_101 = scipy's 'mirror'
    assert err.max() <= 4.0 and err.mean() < 0.6 and err[:4].mean() < 0.7 and err[:, :4].mean() < 0.7


def test_intensity_centroid_against_numpy(oracle):
    """IC_Angle (ORBextractor.cpp:76-103): first moments over the circular 31-px patch -> atan2(m01, m10), against a direct mask"""
    L = oracle.lib()
    img = synth.make_frame(7, cols=200, rows=160)
    umax = np.zeros(16, np.int32)
    vmax = int(np.floor(15 * np.sqrt(2.0) / 2 + 1)); vmin = int(np.ceil(15 * np.sqrt(2.0) / 2))
    for v in range(vmax + 1):
        umax[v] = int(round(np.sqrt(15.0 * 15.0 - v * v)))
    v0 = 0
    for v in range(15, vmin - 1, -1):                                      # the symmetric completion of the table (:452-468)
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0; v0 += 1
    for (x, y) in [(40, 40), (100, 80), (150, 120), (33, 121)]:
        m10 = m01 = 0
        for v in range(-15, 16):
            for u in range(-umax[abs(v)], umax[abs(v)] + 1):
                I = int(img[y + v, x + u]); m10 += u * I; m01 += v * I
        want = np.degrees(np.arctan2(float(m01), float(m10))) % 360
        got = L.orc_ic_angle(oracle._p(img), 200, float(x), float(y))
        assert abs(((got - want + 180) % 360) - 180) < 0.02, (x, y, got, want)
