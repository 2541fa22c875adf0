"""The HIP flow path against analytic truth and against an INDEPENDENT derivation of its constants -- no oracle involved.

tests/test_gpu_flow.py compares every kernel with oracle/farneback_oracle.c, and the host code that prepares the kernels'
constants (csrc/mavflow.cpp: gaussian_kernel, prepare_poly) is a close relative of the oracle's: a slip shared by both would
pass every GPU-vs-oracle test.  Here the same library entry points are held to things neither of them wrote:
  * the polynomial-expansion taps and the four inverse-moment constants against numpy float64 (closed-form normalised Gaussian,
    np.linalg.inv of the 6x6 moment matrix of the basis 1, x, y, x^2, y^2, xy), the pyramid blur taps against the closed form;
  * mav_stage_polyexp on an exact quadratic image (the expansion must return its coefficients);
  * mav_stage_blur_resize on constants, on the 3x3 binomial filter written out in numpy, and on linear ramps (a normalised
    symmetric blur and a half-pixel-centre bilinear resize both map a ramp to the same ramp at the mapped coordinates);
  * mav_farneback on pure translations and on the synthetic radial field, against the known flow.
Reference call being replaced: /root/reference/src/farneback.py:76-80 (cv2.calcOpticalFlowFarneback, arithmetic in OpenCV;
SURVEY.md Appendix A lists what only a live cv2 could still confirm, U1-U6).
"""
import numpy as np
import pytest

from mavflow import synth

pytestmark = pytest.mark.gpu


def gaussian(n, sigma):
    x = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * sigma * sigma))
    return x, g / g.sum()


@pytest.mark.parametrize("poly_n,poly_sigma", [(8, 1.2), (5, 1.1), (7, 1.5)])
def test_polyexp_constants_against_numpy(mav, poly_n, poly_sigma):
    from mavflow import _lib
    fb = _lib.fb_defaults()
    fb.poly_n, fb.poly_sigma = poly_n, poly_sigma
    with _lib.Context(96, 64, 1, fb) as c:
        k = c.stage_coefficients()
    x, g = gaussian(poly_n, poly_sigma)
    half = slice(poly_n, None)                                       # the library stores the centre tap first
    np.testing.assert_allclose(k["g"], g[half], rtol=3e-7, atol=1e-12)
    np.testing.assert_allclose(k["xg"], (x * g)[half], rtol=3e-7, atol=1e-12)
    np.testing.assert_allclose(k["xxg"], (x * x * g)[half], rtol=3e-7, atol=1e-12)
    # moment matrix of the weighted basis (1, x, y, x^2, y^2, xy), weight g(x) g(y): G_ij = sum w b_i b_j
    X, Y = np.meshgrid(x, x)
    wgt = np.outer(g, g)
    basis = [np.ones_like(X), X, Y, X * X, Y * Y, X * Y]
    G = np.array([[np.sum(wgt * bi * bj) for bj in basis] for bi in basis])
    inv = np.linalg.inv(G)
    expect = [inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]]            # ig11, ig03, ig33, ig55
    np.testing.assert_allclose(k["ig"], expect, rtol=2e-5)
    # structure the kernels rely on: x and y are interchangeable, the odd moments vanish
    assert abs(inv[2, 2] - inv[1, 1]) < 1e-12 and abs(inv[0, 4] - inv[0, 3]) < 1e-12 and abs(inv[0, 1]) < 1e-12


def test_pyramid_blur_taps_against_closed_form(mav):
    from mavflow import _lib
    with _lib.Context(3840, 2160, 1, _lib.fb_defaults(levels=5)) as c:
        assert c.num_layers() == 5
        for k in range(5):
            w, h, sigma, ksize = c.layer_dims(k)
            scale = 0.4 ** k
            assert (w, h) == (int(np.rint(3840 * scale)), int(np.rint(2160 * scale)))      # cvRound = round half even = np.rint
            assert sigma == pytest.approx((1 / scale - 1) / 2, rel=1e-12)
            assert ksize == max(int(np.rint(sigma * 5)) | 1, 3)
            taps = c.stage_coefficients(k)["blur"]
            if k == 0:
                assert taps.tolist() == [0.25, 0.5, 0.25]                                   # sigma 0, ksize 3: the fixed kernel
            else:
                xs = np.arange(ksize) - (ksize - 1) / 2
                g = np.exp(-xs * xs / (2 * sigma * sigma))
                np.testing.assert_allclose(taps, g / g.sum(), rtol=3e-7, atol=1e-12)
            assert abs(float(taps.astype(np.float64).sum()) - 1.0) < 2e-7 * ksize


def test_polyexp_recovers_a_quadratic(mav):
    """R = (y-linear, x-linear, yy, xx, xy) of an exact quadratic, away from the border (poly_n = 8 pixels)."""
    from mavflow import _lib
    h, w = 64, 80
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = 40.0, 30.0
    a, b, c2, d, e = 0.7, -0.3, 0.02, 0.015, -0.01                    # x, y, xx, yy, xy
    img = 100 + a * (x - cx) + b * (y - cy) + c2 * (x - cx) ** 2 + d * (y - cy) ** 2 + e * (x - cx) * (y - cy)
    with _lib.Context(w, h, 1) as c:
        assert c.num_layers() == 1
        R = c.stage_polyexp(img.astype(np.float32), 0)
    s = np.s_[10:-10, 10:-10]
    gx = a + 2 * c2 * (x - cx) + e * (y - cy)
    gy = b + 2 * d * (y - cy) + e * (x - cx)
    np.testing.assert_allclose(R[1][s], gx[s], atol=2e-3)
    np.testing.assert_allclose(R[0][s], gy[s], atol=2e-3)
    np.testing.assert_allclose(R[3][s], c2, atol=2e-4)
    np.testing.assert_allclose(R[2][s], d, atol=2e-4)
    np.testing.assert_allclose(R[4][s], e, atol=2e-4)


def test_blur_resize_constant_binomial_and_ramps(mav):
    from mavflow import _lib
    W, H = 200, 120
    with _lib.Context(W, H, 1) as c:
        assert c.num_layers() == 2
        w1, h1, sigma1, k1 = c.layer_dims(1)
        assert (w1, h1, k1) == (80, 48, 5) and sigma1 == pytest.approx(0.75)
        const = np.full((H, W), 77, np.uint8)
        np.testing.assert_allclose(c.stage_blur_resize(const, 0), 77.0, atol=1e-4)
        np.testing.assert_allclose(c.stage_blur_resize(const, 1), 77.0, atol=1e-4)
        # layer 0 = the 3x3 binomial filter with BORDER_REFLECT_101, written out in numpy (exact in float32: dyadic weights)
        rng = np.random.default_rng(0)
        img = rng.integers(0, 256, (H, W)).astype(np.uint8)
        p = np.pad(img.astype(np.float64), 1, mode="reflect")
        kk = np.array([0.25, 0.5, 0.25])
        ref = sum(kk[i] * kk[j] * p[i:i + H, j:j + W] for i in range(3) for j in range(3))
        assert np.array_equal(c.stage_blur_resize(img, 0), ref.astype(np.float32))
        # ramps: blur(ramp) = ramp away from the border, resize(INTER_LINEAR) samples it at (d + 0.5) * S/s - 0.5
        dx, dy = np.arange(w1), np.arange(h1)
        sx, sy = (dx + 0.5) * (W / w1) - 0.5, (dy + 0.5) * (H / h1) - 0.5
        ramp_x = np.tile(np.arange(W, dtype=np.uint8), (H, 1))
        got = c.stage_blur_resize(ramp_x, 1)
        np.testing.assert_allclose(got[4:-4, 4:-4], np.tile(sx, (h1, 1))[4:-4, 4:-4], atol=2e-4)
        ramp_y = np.tile(np.arange(H, dtype=np.uint8)[:, None], (1, W))
        got = c.stage_blur_resize(ramp_y, 1)
        np.testing.assert_allclose(got[4:-4, 4:-4], np.tile(sy[:, None], (1, w1))[4:-4, 4:-4], atol=2e-4)


@pytest.mark.parametrize("shift", [(1.5, -0.75), (-3.0, 2.0)])
def test_farneback_recovers_a_uniform_translation(mav, shift):
    from mavflow import _lib
    W, H = 320, 240
    rng = np.random.default_rng(5)
    fx, fy, amp, ph = synth._texture_params(rng)
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    t0 = synth._eval_separable(x, y, fx, fy, amp, ph)
    t1 = synth._eval_separable(x - shift[0], y - shift[1], fx, fy, amp, ph)
    A = 119.5 / np.abs(t0).max()
    f0 = np.rint(127.5 + A * t0).astype(np.uint8)
    f1 = np.clip(np.rint(127.5 + A * t1), 0, 255).astype(np.uint8)
    with _lib.Context(W, H, 1) as c:
        flow = c.farneback(f0, f1)[0]
    s = np.s_[30:-30, 30:-30]
    err = np.hypot(flow[..., 0][s] - shift[0], flow[..., 1][s] - shift[1])
    assert err.mean() < 0.05, err.mean()
    assert np.percentile(err, 99) < 0.3


def test_farneback_tracks_the_radial_field_640x480(mav):
    from mavflow import _lib
    W, H = 640, 480
    f0, f1, truth = synth.make_pair(W, H, 0)
    with _lib.Context(W, H, 1) as c:
        flow = c.farneback(f0, f1)[0]
    err = np.hypot(flow[..., 0] - truth[..., 0], flow[..., 1] - truth[..., 1])
    inner = np.ones((H, W), bool)
    inner[:20] = inner[-20:] = False
    inner[:, :20] = inner[:, -20:] = False
    inner[H // 4 - 24:H // 4 + 48, W // 4 - 24:W // 4 + 48] = False
    assert err[inner].mean() < 0.1, err[inner].mean()
    radial = synth.true_flow(W, H, patch=False)
    dev = np.hypot(flow[..., 0] - radial[..., 0], flow[..., 1] - radial[..., 1])
    assert dev[H // 4:H // 4 + 24, W // 4:W // 4 + 24].mean() > 5 * err[inner].mean()
