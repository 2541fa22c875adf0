"""The reference's float32 path (frame index 0): Detector.derotate returns the float32 flow itself
(/root/reference/src/detector.py:80-81), so get_FOE_dense's |flow2| gate (focus_of_expansion.py:78), get_phi (:163-177) and the
threshold block (processor.py:333-341) run in float32.  Fixtures: tests/golden/frame0_io.npz, written by the reference's own
Python (tools/gen_golden.py).

Bar: FoE bit-exact; masks bit-exact except where the reference's own float32 phi sits within ARCCOS_ULPS float32 ulps of the
threshold that decided the pixel; phi within ARCCOS_ULPS ulps.  The band exists because numpy's float32 arccos is a SIMD
routine (AVX-512: up to 2 ulp from the correctly rounded value, 35 % of the arguments differ; without AVX-512 the host libm)
-- the reference's own result at such a pixel depends on the CPU it ran on.  The fixtures here were made on an AVX-512 host.
"""
import numpy as np
import pytest

from oracle import foe_oracle as fo
from mavflow import synth

ARCCOS_ULPS = 4


@pytest.fixture(scope="module")
def f0():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "frame0_io.npz"), allow_pickle=False)


def ulps32(a, b):
    """distance in float32 units in the last place between two float32 arrays of non-negative values"""
    return np.abs(a.astype(np.float32).view(np.int32).astype(np.int64) - b.astype(np.float32).view(np.int32).astype(np.int64))


def thresholds32(mag32):
    """the float32 thresholds of processor.py:333-341 for a float32 magnitude image"""
    with np.errstate(all="ignore"):
        return np.float32(0.25) + (np.float32(0.5) + np.float32(8) / mag32)


def assert_masks(got_fixed, got_total, exp_fixed, exp_total, phi_ref, mag_ref):
    """bit-exact outside the arccos band; returns how many pixels the band excused (0 expected at fixture sizes)"""
    with np.errstate(all="ignore"):
        near_fixed = ulps32(phi_ref, np.full_like(phi_ref, 15.0)) <= ARCCOS_ULPS
        thr = thresholds32(mag_ref)
        near_dyn = np.isfinite(thr) & (ulps32(phi_ref, np.where(np.isfinite(thr), thr, 0).astype(np.float32)) <= ARCCOS_ULPS)
    bad_f = (got_fixed != exp_fixed) & ~near_fixed
    bad_d = (got_total != exp_total) & ~near_dyn
    assert not bad_f.any(), f"{int(bad_f.sum())} fixed-mask pixels differ away from 15 degrees"
    assert not bad_d.any(), f"{int(bad_d.sum())} dynamic-mask pixels differ away from their threshold"
    return int((got_fixed != exp_fixed).sum() + (got_total != exp_total).sum())


# ---- the oracle against the reference-generated fixtures (CPU) --------------------------------------------------------------
def test_oracle_float32_phi_and_masks(f0):
    flow = f0["f0_flow"]
    assert flow.dtype == np.float32
    phi = fo.get_phi(flow, tuple(f0["f0_foe"]))
    mag = fo.get_magnitude(flow)
    assert phi.dtype == np.float32 and mag.dtype == np.float32
    assert mag.tobytes() == f0["f0_mag"].tobytes()                       # sqrt is correctly rounded everywhere: bit-exact
    assert ulps32(phi, f0["f0_phi"]).max() <= ARCCOS_ULPS
    for tag, sky in (("nosky", None), ("sky", f0["f0_sky"])):
        fixed, total = fo.threshold_masks(phi, mag, sky)
        assert_masks(fixed, total, f0[f"f0_fixed_{tag}"], f0[f"f0_total_{tag}"], f0["f0_phi"], f0["f0_mag"])


def test_oracle_float32_gate(f0):
    gate = f0["f0_gate_flow"]
    smp = _samples(int(f0["f0_gate_seed"]), *gate.shape[:2])
    assert float(f0["f0_gate_disagree_fraction"]) > 0.2                  # the fixture really separates the two gates
    assert tuple(f0["f0_gate_foe"]) != tuple(f0["f0_gate_foe_if_double"])
    assert np.array(fo.get_foe_dense(gate, smp)).tobytes() == f0["f0_gate_foe"].tobytes()
    assert np.array(fo.get_foe_dense(gate.astype(np.float64), smp)).tobytes() == f0["f0_gate_foe_if_double"].tobytes()


def test_oracle_frame0_chain(f0):
    W, H = 640, 480
    fl = synth.synthetic_flow(W, H, seed=3)
    out = fo.run_chain(fl, synth.foe_samples(W, H, 0), (1.0, 2.0, 3.0), 1 / 30.0, current_frame_index=0)     # rates are ignored
    assert np.array(out["foe"]).tobytes() == f0["chain0_foe"].tobytes()
    assert out["phi"].dtype == np.float32 == np.dtype(str(f0["chain0_phi_dtype"]))
    assert ulps32(out["phi"], f0["chain0_phi"]).max() <= ARCCOS_ULPS
    exp_f = np.unpackbits(f0["chain0_fixed_bits"]).reshape(H, W).astype(bool)
    exp_t = np.unpackbits(f0["chain0_total_bits"]).reshape(H, W).astype(bool)
    assert_masks(out["fixed"], out["total"], exp_f, exp_t, f0["chain0_phi"], fo.get_magnitude(fl))


def _samples(seed, H, W, n=1000):
    state = np.random.get_state()
    try:
        np.random.seed(seed)
        s = np.zeros((2 * n, 2), np.uint32)
        s[:, 0] = np.random.randint(0, H, 2 * n)
        s[:, 1] = np.random.randint(0, W, 2 * n)
    finally:
        np.random.set_state(state)
    return s


# ---- libmavflow against the same fixtures (GPU) ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx_small(mav):
    from mavflow import _lib
    with _lib.Context(160, 120, 4) as c:
        yield c


@pytest.mark.gpu
def test_gpu_float32_phi_and_masks(ctx_small, f0):
    flow = f0["f0_flow"]
    phi, mf, md, mx = ctx_small.phi_mask(flow, f0["f0_foe"])             # float32 in -> float32 arithmetic
    assert phi.dtype == np.float32 and mx.dtype == np.float32
    assert ulps32(phi[0], f0["f0_phi"]).max() <= ARCCOS_ULPS
    assert ulps32(mx, np.array([f0["f0_max_flow"]])).max() <= ARCCOS_ULPS
    assert_masks(mf[0], md[0], f0["f0_fixed_nosky"], f0["f0_total_nosky"], f0["f0_phi"], f0["f0_mag"])
    _, mf, md, _ = ctx_small.phi_mask(flow, f0["f0_foe"], sky=f0["f0_sky"], want_phi=False)
    assert_masks(mf[0], md[0], f0["f0_fixed_sky"], f0["f0_total_sky"], f0["f0_phi"], f0["f0_mag"])
    # the same field promoted to double takes the double path and is NOT the float32 answer everywhere
    phi64, _, _, _ = ctx_small.phi_mask(flow.astype(np.float64), f0["f0_foe"])
    assert phi64.dtype == np.float64 and np.any(phi64[0].astype(np.float32) != phi[0])


@pytest.mark.gpu
def test_gpu_float32_gate(ctx_small, f0):
    gate = f0["f0_gate_flow"]
    smp = _samples(int(f0["f0_gate_seed"]), *gate.shape[:2])
    assert ctx_small.foe_dense(gate, smp)[0].tobytes() == f0["f0_gate_foe"].tobytes()
    assert ctx_small.foe_dense(gate.astype(np.float64), smp)[0].tobytes() == f0["f0_gate_foe_if_double"].tobytes()


@pytest.mark.gpu
def test_gpu_frame0_chain_through_detect(mav, f0):
    """mav_detect with the frame0 flag: pair 0 is the reference's frame 0 (float32, rates ignored), pair 1 the same field as a
    later frame (derotated, double) -- one call, two arithmetic types."""
    from mavflow import _lib
    W, H = 640, 480
    fl = synth.synthetic_flow(W, H, seed=3)
    smp = synth.foe_samples(W, H, 0)
    dt = 1 / 30.0
    omega = np.array([0.013, -0.021, 0.008]) / dt
    with _lib.Context(W, H, 2) as c:
        out = c.detect(np.stack([fl, fl]), np.stack([smp, smp]), omega=np.stack([omega, omega]), dt=[dt, dt], frame0=[1, 0],
                       want_phi=True)
        out2 = c.detect(np.stack([fl, fl]), np.stack([smp, smp]), omega=np.stack([omega, omega]), dt=[dt, dt], frame0=[1, 0])
    r = out["results"]
    assert r[0]["foe"].tobytes() == f0["chain0_foe"].tobytes()
    assert ulps32(out["phi"][0], f0["chain0_phi"]).max() <= ARCCOS_ULPS              # float32 values, widened
    assert np.array_equal(out["phi"][0], out["phi"][0].astype(np.float32))
    exp_f = np.unpackbits(f0["chain0_fixed_bits"]).reshape(H, W).astype(bool)
    exp_t = np.unpackbits(f0["chain0_total_bits"]).reshape(H, W).astype(bool)
    assert_masks(out["mask_fixed"][0], out["mask_dyn"][0], exp_f, exp_t, f0["chain0_phi"], fo.get_magnitude(fl))
    b = r[0]["box"]
    if np.array_equal(out["mask_fixed"][0], exp_f):
        assert [b[0], b[1], b[2] - b[0], b[3] - b[1]] == list(f0["chain0_box"])
    assert tuple(b) == fo.simple_bounding_box(out["mask_fixed"][0])
    # pair 1: the double path of every later frame, against the oracle
    ref = fo.run_chain(fl, smp, omega, dt)
    assert tuple(r[1]["foe"]) == tuple(ref["foe"])
    assert np.array_equal(out["mask_fixed"][1], ref["fixed"]) and np.array_equal(out["mask_dyn"][1], ref["total"])
    assert tuple(r[1]["box"]) == tuple(ref["box"])
    # without phi the double pair takes the screened kernel path; nothing may change
    assert out2["results"].tobytes() == r.tobytes()
    assert np.array_equal(out2["mask_fixed"], out["mask_fixed"]) and np.array_equal(out2["mask_dyn"], out["mask_dyn"])
