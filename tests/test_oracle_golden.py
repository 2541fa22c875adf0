"""Pin the numpy oracle (oracle/foe_oracle.py) bit-for-bit to fixtures produced by the reference's own Python
(tools/gen_golden.py -> tests/golden/foe_chain.npz).  CPU only."""
import numpy as np

from oracle import foe_oracle as fo
from mavflow import synth


def beq(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def test_line_intersection(golden):
    pts, fl, exp = golden["li_pts"], golden["li_flow"], golden["li_out"]
    L = pts.shape[0]
    # build a tiny flow image + sample list that reproduces the L independent pairs
    flow = np.zeros((512, 512, 2), np.float32)
    samples = np.zeros((2 * L, 2), np.uint32)
    for i in range(L):
        for j in range(2):
            x, y = pts[i, j]
            flow[y, x] = fl[i, j]
            samples[i + j * L] = (y, x)
    # duplicates would overwrite each other; the fixture's coordinates are distinct enough to check that
    for i in range(L):
        for j in range(2):
            x, y = pts[i, j]
            assert np.array_equal(flow[y, x], fl[i, j])
    got = fo.line_intersections(flow, samples, mag_threshold=0.0)
    assert beq(got, exp)
    assert exp[5, 0] == 0.0 and exp[5, 1] == 0.0          # parallel -> (False, False)


def test_foe_dense_and_phi(golden):
    for c in range(4):
        flow = golden["foe_flow"][c]
        if golden["foe_flow_is_f32"][c]:
            flow = flow.astype(np.float32)
        foe = fo.get_foe_dense(flow, golden[f"foe_samples_{c}"])
        assert beq(np.array(foe), golden["foe_out"][c]), c
        phi = fo.get_phi(golden["foe_flow"][c], tuple(golden["foe_out"][c]))
        assert beq(phi, golden["phi_out"][c]), c
    assert beq(np.array(fo.get_foe_dense(np.zeros((120, 160, 2), np.float32), golden["foe_samples_0"])),
               golden["foe_zero"])
    assert tuple(golden["foe_zero"]) == (0.0, 0.0)


def test_ransac_edges(golden):
    for tag in ("iso", "tie", "clus"):
        assert beq(np.array(fo.ransac(golden[f"ransac_{tag}_in"])), golden[f"ransac_{tag}_out"]), tag
    assert tuple(golden["ransac_iso_out"]) == (0.0, 0.0)
    assert tuple(golden["ransac_tie_out"]) == (10.0, 10.0)
    assert fo.ransac(np.zeros((0, 2))) == tuple(golden["ransac_empty_out"]) == (0.0, 0.0)


def test_phi_special(golden):
    fz = golden["phi_zero_flow_in"]
    out = fo.get_phi(fz, (70.5, 40.25))
    assert beq(out, golden["phi_zero_flow_out"])
    assert np.all(out[10:20, 10:20] == 90.0)
    assert fo.get_phi(fz, (np.nan, np.nan)).shape[0] == int(golden["phi_nan_identity_len"]) == 0
    assert beq(fo.get_phi(fz, (float("nan"), 3.0)), golden["phi_float_nan_out"])
    assert beq(fo.get_phi(golden["foe_flow"][1], (80.0, 60.0)), golden["phi_on_pixel_out"])
    p32 = fo.get_phi(golden["foe_flow"][0].astype(np.float32), tuple(golden["foe_out"][0]))
    assert str(p32.dtype) == str(golden["phi_f32_dtype"]) == "float32"
    # float32 arccos is libm/SIMD dependent across machines: tolerance only
    np.testing.assert_allclose(p32, golden["phi_f32_out"], rtol=0, atol=1e-3)


def test_thresholds_and_magnitude(golden):
    phi = golden["phi_out"][1]
    mag = fo.get_magnitude(golden["foe_flow"][1])
    assert beq(mag, golden["mag_out"])
    for tag, sky in (("nosky", None), ("sky", golden["thr_sky"])):
        fixed, total = fo.threshold_masks(phi, mag, sky)
        assert beq(fixed, golden[f"thr_{tag}_fixed"]), tag
        assert beq(total, golden[f"thr_{tag}_total"]), tag
    assert golden["thr_nosky_fixed"].sum() > 0 and golden["thr_nosky_total"].sum() > 0


def _rect(box):
    x0, y0, x1, y1 = box
    return np.array([x0, y0, x1 - x0, y1 - y0])


def test_bbox(golden):
    assert np.array_equal(_rect(fo.simple_bounding_box(golden["bbox_a_in"])), golden["bbox_a"])
    assert list(golden["bbox_a"]) == [7, 5, 12, 3]
    assert np.array_equal(_rect(fo.simple_bounding_box(np.zeros((120, 160), np.uint8))), golden["bbox_empty"])
    assert list(golden["bbox_empty"]) == [-1, -1, 0, 0]
    assert np.array_equal(_rect(fo.simple_bounding_box(golden["bbox_gray_in"])), golden["bbox_gray"])
    assert np.array_equal(_rect(fo.simple_bounding_box(golden["thr_nosky_fixed"])), golden["bbox_fixed"])


def test_tpr_fpr(golden):
    got = fo.calculate_tpr_fpr(golden["tpr_gt"], 255 * golden["thr_nosky_fixed"])
    assert beq(np.array(got, dtype=np.float64), golden["tpr_out"])


def test_derotate(golden):
    dt = float(golden["derot_dt"])
    omega = golden["derot_dangle"] / dt
    out = fo.derotate(golden["derot_in"], omega, dt, 1)
    assert out.dtype == np.float64
    assert beq(out, golden["derot_out"])
    f = golden["derot_in"]
    assert fo.derotate(f, omega, dt, 0) is f and bool(golden["derot_frame0_same"])


def test_full_chain_640x480(golden):
    W, H = 640, 480
    fl = synth.synthetic_flow(W, H, seed=3)
    smp = synth.foe_samples(W, H, 0)
    assert int(smp.astype(np.int64).sum()) == int(golden["chain_samples_sum"])
    dt = float(golden["derot_dt"])
    r = fo.run_chain(fl, smp, golden["derot_dangle"] / dt, dt)
    assert beq(np.array(r["foe"]), golden["chain_foe"])
    assert np.array_equal(np.packbits(r["fixed"]), golden["chain_fixed_bits"])
    assert np.array_equal(np.packbits(r["total"]), golden["chain_total_bits"])
    assert beq(r["phi"][::16, ::16].copy(), golden["chain_phi_sub"])
    assert np.array_equal(_rect(r["box"]), golden["chain_box"])
    assert float(np.max(r["phi"])) == float(golden["chain_max_flow"])
