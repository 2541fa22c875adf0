"""The C restatement of Farneback (oracle/farneback_oracle.c) against analytic ground truth.  CPU only.

PARITY UNPINNED vs cv2 (OpenCV is not in the image, the reference holds no Farneback fixture); these tests
bound the restatement by known flow instead."""
import numpy as np
import pytest

from oracle import fb_oracle
from mavflow import synth


def test_layers_and_dims(fb_oracle):
    p = fb_oracle_params()
    assert fb_oracle.num_layers(1920, 1080, p) == 2
    assert fb_oracle.layer_dims(1920, 1080, p, 1) == (768, 432, 0.75, 5)
    assert fb_oracle.layer_dims(1920, 1080, p, 0) == (1920, 1080, 0.0, 3)
    p5 = fb_oracle_params(levels=5)
    assert fb_oracle.num_layers(3840, 2160, p5) == 5
    dims = [fb_oracle.layer_dims(3840, 2160, p5, k) for k in range(5)]
    assert [d[3] for d in dims] == [3, 5, 13, 37, 95]
    assert dims[4][:2] == (98, 55)
    assert fb_oracle.num_layers(64, 48, p) == 1            # 0.4 * 48 < 32 -> no coarse layer


def fb_oracle_params(levels=1):
    return fb_oracle_mod().default_params(levels)


def fb_oracle_mod():
    from oracle import fb_oracle as m
    return m


def test_polyexp_on_quadratic(fb_oracle):
    """A quadratic image is reproduced exactly by the polynomial expansion away from the border."""
    h, w = 64, 80
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = 40.0, 30.0
    a, b, c, d, e = 0.7, -0.3, 0.02, 0.015, -0.01      # x, y, xx, yy, xy
    I = 100 + a * (x - cx) + b * (y - cy) + c * (x - cx) ** 2 + d * (y - cy) ** 2 + e * (x - cx) * (y - cy)
    R = fb_oracle.polyexp(I.astype(np.float32))
    s = np.s_[10:-10, 10:-10]
    gx = a + 2 * c * (x - cx) + e * (y - cy)
    gy = b + 2 * d * (y - cy) + e * (x - cx)
    np.testing.assert_allclose(R[..., 1][s], gx[s], atol=2e-3)      # x-linear
    np.testing.assert_allclose(R[..., 0][s], gy[s], atol=2e-3)      # y-linear
    np.testing.assert_allclose(R[..., 3][s], c, atol=2e-4)          # xx
    np.testing.assert_allclose(R[..., 2][s], d, atol=2e-4)          # yy
    np.testing.assert_allclose(R[..., 4][s], e, atol=2e-4)          # xy


def test_blur_resize_identity_and_constant(fb_oracle):
    img = np.full((48, 64), 77, np.uint8)
    out = fb_oracle.blur_resize(img, 64, 48, 3, 0.0)
    np.testing.assert_allclose(out, 77.0, atol=1e-4)
    out = fb_oracle.blur_resize(img, 26, 19, 5, 0.75)
    np.testing.assert_allclose(out, 77.0, atol=1e-4)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    out = fb_oracle.blur_resize(img, 64, 48, 3, 0.0)
    f = img.astype(np.float64)
    p = np.pad(f, 1, mode="reflect")
    k = np.array([0.25, 0.5, 0.25])
    ref = sum(k[i] * k[j] * p[i:i + 48, j:j + 64] for i in range(3) for j in range(3))
    np.testing.assert_allclose(out, ref, atol=1e-3)


@pytest.mark.parametrize("shift", [(1.5, -0.75), (-3.0, 2.0)])
def test_uniform_translation(fb_oracle, shift):
    """Pure translation of a smooth texture: interior flow equals the shift."""
    W, H = 320, 240
    rng = np.random.default_rng(5)
    fx, fy, amp, ph = synth._texture_params(rng)
    x = np.arange(W, dtype=np.float64); y = np.arange(H, dtype=np.float64)
    t0 = synth._eval_separable(x, y, fx, fy, amp, ph)
    t1 = synth._eval_separable(x - shift[0], y - shift[1], fx, fy, amp, ph)
    A = 119.5 / np.abs(t0).max()
    f0 = np.rint(127.5 + A * t0).astype(np.uint8)
    f1 = np.clip(np.rint(127.5 + A * t1), 0, 255).astype(np.uint8)
    flow = fb_oracle.calc(f0, f1)
    s = np.s_[30:-30, 30:-30]
    err = np.hypot(flow[..., 0][s] - shift[0], flow[..., 1][s] - shift[1])
    assert err.mean() < 0.05, err.mean()
    assert np.percentile(err, 99) < 0.3


def test_radial_flow_640x480(fb_oracle):
    """BASELINE config 1 plumbing: 640x480 synthetic pair, flow close to the analytic field away from the patch."""
    W, H = 640, 480
    f0, f1, truth = synth.make_pair(W, H, 0)
    flow = fb_oracle.calc(f0, f1)
    err = np.hypot(flow[..., 0] - truth[..., 0], flow[..., 1] - truth[..., 1])
    inner = np.ones((H, W), bool)
    inner[:20] = inner[-20:] = False
    inner[:, :20] = inner[:, -20:] = False
    inner[H // 4 - 24:H // 4 + 48, W // 4 - 24:W // 4 + 48] = False
    assert err[inner].mean() < 0.1, err[inner].mean()
    # the 24x24 moving patch is smaller than the 13x13 window at the coarse layer, so Farneback only registers
    # it as a disturbance of the radial field -- which is all the detector needs
    radial = synth.true_flow(W, H, patch=False)
    dev = np.hypot(flow[..., 0] - radial[..., 0], flow[..., 1] - radial[..., 1])
    assert dev[H // 4:H // 4 + 24, W // 4:W // 4 + 24].mean() > 5 * err[inner].mean()


def test_bad_arguments(fb_oracle):
    a = np.zeros((48, 64), np.uint8)
    with pytest.raises(ValueError):
        fb_oracle.calc(a, np.zeros((48, 65), np.uint8))
    p = fb_oracle_params()
    p.pyr_scale = 1.0
    with pytest.raises(ValueError):
        fb_oracle.calc(a, a, p)


def test_oracle_vs_cv2_when_available(fb_oracle):
    """Pins the restatement wherever OpenCV exists (it does not in the build image: skipped, parity unpinned)."""
    cv2 = pytest.importorskip("cv2")
    f0, f1, _ = synth.make_pair(640, 480, 0)
    ref = cv2.calcOpticalFlowFarneback(f0, f1, None, 0.4, 1, 12, 10, 8, 1.2, 0)
    got = fb_oracle.calc(f0, f1)
    e = np.hypot(got[..., 0] - ref[..., 0], got[..., 1] - ref[..., 1])
    assert e.mean() <= 1e-3 and e.max() <= 1e-1, (e.mean(), e.max())


# ---- the checker's view of its own stability (oracle/tolerances.py, round 6) ---------------------------------------------------------
def test_sweep_record_and_python_driven_pyramid(fb_oracle):
    """calc(want_sys) changes nothing about the flow; the record holds the system the flow was solved from and the flow the sweep
    replaced; the pyramid driven layer by layer from Python (tools/worst_pixel.py) is calc() bit for bit."""
    f0, f1, _ = synth.make_pair(200, 150, 5, k=0.02, patch=False)
    p = fb_oracle_params(levels=2)
    flow = fb_oracle.calc(f0, f1, p)
    flow2, rec = fb_oracle.calc(f0, f1, p, want_sys=True)
    assert np.array_equal(flow, flow2) and rec.shape == (150, 200, 7) and np.isfinite(rec).all()
    g11, g12, g22, h1, h2 = (rec[..., i] for i in range(5))
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    assert np.array_equal(((g11 * h2 - g12 * h1) * idet).astype(np.float32), flow[..., 0])
    assert np.array_equal(((g22 * h1 - g12 * h2) * idet).astype(np.float32), flow[..., 1])
    seen = []
    staged = fb_oracle.pyramid(f0, f1, p, lambda k, it, fl, M, s, R0, R1: seen.append((k, it, fl.copy(), s)))
    assert np.array_equal(staged, flow)
    assert fb_oracle.num_layers(200, 150, p) == 2                          # 0.16 * 150 < 32: the third layer is not built
    assert [(k, it) for k, it, _, _ in seen] == [(k, it) for k in (1, 0) for it in range(10)]
    assert np.array_equal(seen[-1][3], rec)                               # the record IS the finest layer's last sweep
    assert np.array_equal(seen[-2][2], rec[..., 5:7].astype(np.float32))   # "before" = the sweep before it


def test_twins_are_close_where_the_iteration_settles_and_the_restatement_is_mirror_symmetric(fb_oracle):
    """The sensitivity twins (float32 window sums on the frames and on the mirrored frames; NOT OpenCV) stay within the flow gate of
    calc() on a friendly texture -- its unstable class is empty there -- and the restatement itself is mirror symmetric: calc() of
    the mirrored frames, mirrored back, is calc() up to float32 storage rounding."""
    from oracle import tolerances as tol
    f0, f1, _ = synth.make_pair(320, 240, 3)
    ref = fb_oracle.calc(f0, f1)
    twins = fb_oracle.twins(f0, f1)
    assert len(twins) == 2 and not np.array_equal(ref, twins[0]) and not np.array_equal(twins[0], twins[1])
    for t in twins:
        tol.check_flow(t, ref, "twin")
    assert not tol.unstable_mask(ref, twins).any() and tol.sensitivity(ref, twins).max() < 1e-3
    m = fb_oracle.calc(np.ascontiguousarray(f0[:, ::-1]), np.ascontiguousarray(f1[:, ::-1]))[:, ::-1].copy()
    m[..., 0] = -m[..., 0]
    assert tol.epe(m, ref).max() < 1e-4
    flow, rec, flips = fb_oracle.calc_tracked(f0, f1)
    assert np.array_equal(flow, ref) and flips.shape == (240, 320) and flips.dtype == np.uint8
    assert (flips > 0).mean() < 0.01                                       # a settled field: hardly a pixel still crosses the border test


def test_flow_gate_classes():
    """oracle/tolerances.py: the strict gate for everybody; a frame outside it passes only with its unstable pixels (the oracle's own
    twins apart by >= 0.01 px, or a flip of the border test in the window) set aside, few of them, nothing wild, fewer still (0.5 % of the
    frame) actually beyond 0.05 px, and a TIGHTER maximum on all the rest."""
    from oracle import tolerances as tol
    H, W = 200, 300
    exp = np.zeros((H, W, 2), np.float32)
    got = exp.copy()
    got[100, 150, 0] = 0.2                                             # one pixel off by 0.2 px
    with pytest.raises(AssertionError, match="max EPE"):
        tol.check_flow(got, exp)                                       # strict: no twins, no excuse
    twins = [exp.copy(), exp.copy()]
    with pytest.raises(AssertionError, match="max EPE"):
        tol.check_flow(got, exp, "", twins)                            # the oracle is stable there: still a failure
    twins[1][98, 152, 1] = 0.02                                        # ONE of the twins moves 0.02 px inside the pixel's window
    e = tol.check_flow(got, exp, "", twins)
    assert e.max() == pytest.approx(0.2) and tol.unstable_mask(exp, twins).sum() == 13 * 13
    flips = np.zeros((H, W), np.uint8)
    flips[103, 147] = 1                                                # ... or a pixel of the window flipped the border test
    assert tol.check_flow(got, exp, "", [exp.copy(), exp.copy()], 6, flips).max() == pytest.approx(0.2)
    got[100, 150, 0] = 4.5
    with pytest.raises(AssertionError, match="unstable pixels"):
        tol.check_flow(got, exp, "", twins)                            # nothing wild, even there
    got[100, 150, 0] = 0.2
    got[10, 10, 1] = 0.06                                              # a second pixel, stable: the excused frame's rest is held to 0.05 px
    with pytest.raises(AssertionError, match="stable pixels"):
        tol.check_flow(got, exp, "", twins)
    got[10, 10, 1] = 0.04
    tol.check_flow(got, exp, "", twins)
    wide = [exp.copy(), exp.copy()]
    wide[0][40:60:12, 20:200:12, 0] = 0.02                             # 7.8 % of the frame unstable (inside the 10 % the class may take) ...
    bad = got.copy()
    bad[38:58, 20:40, 0] = 0.07                                        # ... and 400 of those pixels (0.67 % of the frame) beyond 0.05 px
    assert tol.unstable_mask(exp, wide).mean() < tol.FLOW_UNSTABLE_FRAC
    with pytest.raises(AssertionError, match="too many pixels beyond"):
        tol.check_flow(bad, exp, "", [wide[0], twins[1]])             # the exception is for a few pixels, not for a region
    bad[42:58, 20:40, 0] = 0.0                                         # 80 of them (0.13 %): excused
    tol.check_flow(bad, exp, "", [wide[0], twins[1]])
    twins[0][::12, ::12, 0] = 0.3                                      # the oracle unstable (nearly) everywhere: not a frame to excuse
    with pytest.raises(AssertionError, match="too many unstable"):
        tol.check_flow(got, exp, "", twins)
    got[100, 150, 0] = 0.1                                             # ... but a frame inside the strict gate passes however unstable the oracle is
    tol.check_flow(got, exp, "", twins)
    got[5, 5, 0] = np.nan
    with pytest.raises(AssertionError, match="non-finite"):
        tol.check_flow(got, exp, "", [exp.copy()])
    assert not tol.flow_epe_ok(np.full((4, 4), np.nan)) and tol.flow_epe_ok(np.zeros((4, 4)))
    assert (tol.FLOW_EPE_MAX, tol.FLOW_EPE_MAX_STABLE, tol.FLOW_UNSTABLE_S, tol.FLOW_UNSTABLE_FRAC, tol.FLOW_EXCUSED_FRAC) == (0.15, 0.05, 0.01, 0.10, 5e-3)
