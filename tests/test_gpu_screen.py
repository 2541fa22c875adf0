"""The single-precision screen of the phi / mask kernel, bounded by test.

When phi itself is not requested (the configuration bench.py times) k_phi_mask decides most pixels in float32 -- in the
tangent form  phi > T  <=>  dot <= 0 or |cross| > tan(T) dot  for thresholds up to 17 degrees (margin 4e-6 |f||d|(1 + tan T),
i.e. ~2e-4 degrees), in the arccos-argument form  arg < cos T  (band 1e-4) for larger dynamic thresholds, 1e-5 relative around
the two magnitude gates -- and sends only the pixels inside those bands down the exact double path.
The claim is that the masks cannot change.  These tests plant pixels at +-{1e-10 ... 0.1} degrees and +-{1e-8 ... 1e-3}
(arccos-argument units) around
every decision -- the fixed 15 degree threshold, the dynamic threshold 0.75 + 8/mag over mag in [0.5, 200] (and, with other
parameters, all the way to 178 degrees), both magnitude gates -- for an FoE inside the image, on a pixel centre, and 1e4 px
outside it, and require the screened masks (float32 flow, the kernel instance of the fused path, through mav_stage_phi_mask;
float64 flow through mav_phi_mask) to equal the oracle's bit for bit.  Reference: /root/reference/src/processor.py:333-341,
focus_of_expansion.py:150-184.
"""
import numpy as np
import pytest

from oracle import foe_oracle as fo

pytestmark = pytest.mark.gpu
W, H = 640, 480
DELTAS = np.array([0.0, 1e-8, 3e-8, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1.9e-5, 2.1e-5, 3e-5, 6e-5, 9.5e-5, 1.05e-4, 1.5e-4, 3e-4, 1e-3])
# the same in the ANGLE domain (degrees): the tangent-form screen's band is ~2e-4 degrees wide at any threshold
EPS_DEG = np.array([0.0, 1e-10, 1e-9, 1e-8, 1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 2e-4, 3e-4, 1e-3, 1e-2, 0.1])


def planted_field(foe, rng, dyn_a=0.25, dyn_b=0.5, dyn_c=8.0, fixed_deg=15.0, mag_lo=0.5, mag_hi=200.0, gates=(0.5, 1.0)):
    """A flow field (H, W, 2) float64 whose every pixel sits a chosen distance from one of the kernel's decisions."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    rx, ry = xx - foe[0], yy - foe[1]
    theta_r = np.arctan2(ry, rx)
    kind = rng.integers(0, 4, (H, W))                 # 0 fixed threshold, 1 dynamic threshold, 2 / 3 magnitude gates
    mag = np.exp(rng.uniform(np.log(mag_lo), np.log(mag_hi), (H, W)))
    delta = rng.choice(DELTAS, (H, W)) * rng.choice([-1.0, 1.0], (H, W))
    side = rng.choice([-1.0, 1.0], (H, W))
    # magnitude-gate pixels: |flow| = gate * (1 + delta), angle well above / below the thresholds at random
    for g_idx, k in ((0, 2), (1, 3)):
        sel = kind == k
        mag[sel] = gates[g_idx] * (1.0 + delta[sel])
    with np.errstate(all="ignore"):
        T = np.where(kind == 0, fixed_deg, dyn_a + dyn_b + dyn_c / mag)
    T = np.clip(T, 0.0, 179.9)
    in_angle = rng.random((H, W)) < 0.5               # half the threshold pixels are offset in degrees, half in arccos-argument units
    arg = np.cos(np.deg2rad(T)) + np.where(kind < 2, np.where(in_angle, 0.0, delta), rng.uniform(-0.5, 0.5, (H, W)))
    eps = rng.choice(EPS_DEG, (H, W)) * rng.choice([-1.0, 1.0], (H, W))
    ang = np.arccos(np.clip(arg, -1.0, 1.0)) + np.deg2rad(np.where((kind < 2) & in_angle, eps, 0.0))
    a = theta_r + side * ang
    flow = np.stack([mag * np.cos(a), mag * np.sin(a)], axis=-1)
    return flow


def oracle_masks(flow64, foe, sky=None, **thr):
    with np.errstate(all="ignore"):
        return fo.threshold_masks(fo.get_phi(flow64, foe), fo.get_magnitude(flow64), sky, **thr)


@pytest.fixture(scope="module")
def ctx(mav):
    from mavflow import _lib
    with _lib.Context(W, H, 2) as c:
        yield c


@pytest.mark.parametrize("foe", [(0.55 * W + 0.3, 0.45 * H - 0.2), (320.0, 240.0), (1.0e4 + 0.37, -7.0e3 - 0.11), (-3.3e3, 250.5)])
def test_screen_default_thresholds(ctx, foe):
    rng = np.random.default_rng(int(abs(foe[0])) % 1000)
    flow = planted_field(foe, rng)
    f32 = flow.astype(np.float32)
    sky = np.zeros((H, W), bool)
    sky[::7, ::5] = True
    # float32 flow: the template instance of the fused path.  The oracle sees the same rounded values (derotation with zero
    # rates is the exact promotion to double)
    ef, ed = oracle_masks(f32.astype(np.float64), foe, sky)
    _, mf, md, box = ctx.stage_phi_mask(f32, foe, sky=sky)                       # screen ON
    phi, mf_x, md_x, box_x = ctx.stage_phi_mask(f32, foe, sky=sky, want_phi=True)   # every pixel on the exact path
    assert np.array_equal(mf_x[0], ef) and np.array_equal(md_x[0], ed)
    assert np.array_equal(mf[0], ef), int((mf[0] != ef).sum())
    assert np.array_equal(md[0], ed), int((md[0] != ed).sum())
    assert tuple(box[0]) == tuple(box_x[0]) == fo.simple_bounding_box(ef)
    # float64 flow through the host entry point.  Here the planted offsets survive exactly (delta = 0 puts a pixel ON its
    # threshold), so the exact path's own verdict at such a pixel hangs on the last bit of arccos (device library vs numpy's):
    # the screen is held to the device's exact path, bit for bit, and that path to the oracle outside a 4-ulp band of phi.
    with np.errstate(all="ignore"):
        phi_o, mag_o = fo.get_phi(flow, foe), fo.get_magnitude(flow)
        ef, ed = fo.threshold_masks(phi_o, mag_o, sky)
        band = 4 * np.spacing(180.0)
        near_f = np.abs(phi_o - 15.0) <= band
        near_d = np.abs(phi_o - (0.25 + (0.5 + 8 / mag_o))) <= band
    _, mf, md, _ = ctx.phi_mask(flow, foe, sky=sky, want_phi=False)
    _, mf_x, md_x, _ = ctx.phi_mask(flow, foe, sky=sky, want_phi=True)
    assert np.array_equal(mf[0], mf_x[0]) and np.array_equal(md[0], md_x[0])
    assert not ((mf_x[0] != ef) & ~near_f).any() and not ((md_x[0] != ed) & ~near_d).any()
    assert (near_f | near_d).sum() > 100                               # the knife-edge pixels are really there
    # the planted field really exercises both outcomes of both masks
    assert 0.05 < ef.mean() < 0.95 and 0.05 < ed.mean() < 0.95


def test_screen_other_thresholds_up_to_178_degrees(ctx):
    """dyn_min_mag lowered to 0.04: the dynamic threshold 0.75 + 8/mag now spans 0.79 ... 178.5 degrees (cos T down to -0.9997,
    where __cosf's absolute error matters most), fixed threshold 3 degrees."""
    from mavflow import _lib
    th = _lib.thr_defaults()
    th.fixed_deg, th.fixed_min_mag, th.dyn_min_mag = 3.0, 0.3, 0.04
    kw = dict(fixed_deg=3.0, fixed_min_mag=0.3, dyn_min_mag=0.04)
    foe = (211.25, 301.5)
    rng = np.random.default_rng(99)
    flow = planted_field(foe, rng, fixed_deg=3.0, mag_lo=0.045, mag_hi=200.0, gates=(0.04, 0.3))
    f32 = flow.astype(np.float32)
    ef, ed = oracle_masks(f32.astype(np.float64), foe, None, **kw)
    _, mf, md, _ = ctx.stage_phi_mask(f32, foe, params=th)
    assert np.array_equal(mf[0], ef), int((mf[0] != ef).sum())
    assert np.array_equal(md[0], ed), int((md[0] != ed).sum())
    assert 0.05 < ed.mean() < 0.95


def test_screen_with_derotation(ctx):
    """Rates switched on: the kernel derotates in double on the fly before it screens; the planted offsets no longer sit on the
    thresholds exactly, the masks must still match the oracle's derotate -> phi -> threshold chain."""
    foe = (300.5, 200.25)
    rng = np.random.default_rng(5)
    f32 = planted_field(foe, rng).astype(np.float32)
    omega, dt = np.array([0.0004, -0.0003, 0.0002]) * 30, 1 / 30.0
    der = fo.derotate(f32, omega, dt)
    ef, ed = oracle_masks(der, foe)
    _, mf, md, _ = ctx.stage_phi_mask(f32, foe, omega=omega, dt=dt)
    assert np.array_equal(mf[0], ef) and np.array_equal(md[0], ed)


def test_screen_on_farneback_flow_1080p(mav, fb_oracle):
    """The 1080p case of tests/test_gpu_fullsize.py with phi NOT requested (screen on), against the oracle."""
    from mavflow import _lib, synth
    Wf, Hf = 1920, 1080
    prev, nxt = synth.make_batch(Wf, Hf, 2, distinct=2)
    smp = np.stack([synth.foe_samples(Wf, Hf, b) for b in range(2)])
    with _lib.Context(Wf, Hf, 2) as c:
        out = c.process_batch(prev, nxt, smp, want_phi=False)
    for b in range(2):
        chain = fo.run_chain(out["flow"][b], smp[b])
        assert tuple(out["results"][b]["foe"]) == tuple(chain["foe"])
        assert np.array_equal(out["mask_fixed"][b], chain["fixed"]) and np.array_equal(out["mask_dyn"][b], chain["total"])
        assert tuple(out["results"][b]["box"]) == tuple(chain["box"])
