#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on seeded inputs.

Run in the build container only (needs /root/reference):   python tools/gen_golden.py
The reference is imported in place; nothing of it is copied.  cv2 / imutils / flow_vis / airsim are not
installed here, and none of the functions exercised below touches them, so empty placeholder modules stand
in for the import statements (SURVEY.md section 8c recipe).  Bytecode writing is disabled so the read-only
reference tree is left untouched.
"""
import sys

sys.dont_write_bytecode = True
import os
import types

import numpy as np

REF = "/root/reference/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "mav-detection_amd"))


def _placeholders():
    import matplotlib
    matplotlib.use("Agg")
    cv2 = types.ModuleType("cv2")
    cv2.TERM_CRITERIA_EPS = 2
    cv2.TERM_CRITERIA_COUNT = 1
    cv2.COLORMAP_JET = 2
    cv2.VideoCapture = object
    cv2.VideoWriter = object
    for name, mod in (("cv2", cv2), ("imutils", types.ModuleType("imutils")),
                      ("flow_vis", types.ModuleType("flow_vis")), ("airsim", types.ModuleType("airsim"))):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)


class _FakeDataset:
    """Duck-typed stand-in for datasets.dataset.Dataset: only what Detector.__init__/derotate read."""

    def __init__(self, W, H, dangle, dt):
        self.capture_size = (W, H)
        self._dangle = np.asarray(dangle, dtype=np.float64)
        self._dt = dt

    def get_delta_time(self, i):
        return self._dt

    def get_angular_difference(self, a, b):
        return self._dangle


def main():
    _placeholders()
    import focus_of_expansion as ref_foe
    import im_helpers as ref_im
    import lucas_kanade as ref_lk
    import utils as ref_utils
    from mavflow import synth

    os.makedirs(OUT, exist_ok=True)
    out = {}

    # ---- line_intersection (utils.py:183-197) ---------------------------------------------------------
    rng = np.random.default_rng(1)
    L = 64
    pts = rng.integers(0, 500, (L, 2, 2)).astype(np.uint32)          # (x, y) of the two base points
    fl = rng.normal(0, 4, (L, 2, 2)).astype(np.float32)
    fl[5, 1] = fl[5, 0]                                               # parallel pair -> (False, False)
    fl[9] = 0                                                         # degenerate: both lines are points
    res = np.zeros((L, 2))
    for i in range(L):
        c1, c2 = pts[i, 0], pts[i, 1]
        res[i, :] = ref_utils.line_intersection((c1, fl[i, 0] + c1), (c2, fl[i, 1] + c2))
    out.update(li_pts=pts, li_flow=fl, li_out=res)

    # ---- FoE dense + ransac + phi on small stored fields ----------------------------------------------
    W, H = 160, 120
    lk = ref_lk.LucasKanade(np.zeros((H, W, 3), np.uint8))
    foe = ref_foe.FocusOfExpansion(lk)

    def ref_get_foe(flow, seed):
        np.random.seed(seed)
        return foe.get_FOE_dense(flow)

    def samples_for(seed, h, w, n=1000):
        np.random.seed(seed)
        s = np.zeros((2 * n, 2), np.uint32)
        s[:, 0] = np.random.randint(0, h, 2 * n)
        s[:, 1] = np.random.randint(0, w, 2 * n)
        return s

    flows, foes, seeds, phis = [], [], [], []
    for case in range(4):
        f = synth.true_flow(W, H, k=0.08, patch=True)
        f += np.random.default_rng(50 + case).normal(0, 0.3, f.shape)
        f = f.astype(np.float32 if case % 2 == 0 else np.float64)
        if case == 3:
            f[:, : W // 2] = 0                                        # half the field below the magnitude gate
        seed = 99 + case
        e = ref_get_foe(f, seed)
        flows.append(f.astype(np.float64))                            # exact (f32 -> f64)
        foes.append(e)
        seeds.append(seed)
        phis.append(foe.get_phi(f.astype(np.float64), e))
    out.update(foe_flow=np.stack(flows), foe_flow_is_f32=np.array([1, 0, 1, 0]), foe_seed=np.array(seeds),
               foe_out=np.array(foes), phi_out=np.stack(phis))
    for case in range(4):
        out[f"foe_samples_{case}"] = samples_for(seeds[case], H, W)

    # all-zero flow -> (0, 0)
    out["foe_zero"] = np.array(ref_get_foe(np.zeros((H, W, 2), np.float32), 5))

    # ransac edge cases (focus_of_expansion.py:32-54)
    iso = np.array([[0.0, 0.0], [100.0, 0.0], [0.0, 100.0], [300.0, 300.0]])
    tie = np.array([[10.0, 10.0], [12.0, 10.0], [500.0, 500.0], [501.0, 500.0]])
    clus = np.concatenate([rng.normal(50, 5, (40, 2)), rng.normal(300, 40, (60, 2))])
    out.update(ransac_iso_in=iso, ransac_iso_out=np.array(foe.ransac(iso)),
               ransac_tie_in=tie, ransac_tie_out=np.array(foe.ransac(tie)),
               ransac_clus_in=clus, ransac_clus_out=np.array(foe.ransac(clus)),
               ransac_empty_out=np.array(foe.ransac(np.zeros((0, 2)))))

    # phi special cases (focus_of_expansion.py:150-184)
    fz = flows[1].copy()
    fz[10:20, 10:20] = 0.0                                            # zero-flow pixels -> 90 deg
    out.update(phi_zero_flow_in=fz, phi_zero_flow_out=foe.get_phi(fz, (70.5, 40.25)),
               phi_nan_identity_len=np.array(foe.get_phi(fz, (np.nan, np.nan)).shape[0]),
               phi_float_nan_out=foe.get_phi(fz, (float("nan"), 3.0)))
    # FoE exactly on a pixel centre: distance 0 -> norm floor 1e-6
    out.update(phi_on_pixel_out=foe.get_phi(flows[1], (80.0, 60.0)))
    # float32 flow keeps float32 arithmetic in the reference (zeros_like): record dtype + values
    p32 = foe.get_phi(flows[0].astype(np.float32), foes[0])
    out.update(phi_f32_out=p32, phi_f32_dtype=np.array(str(p32.dtype)))

    # ---- threshold block (processor.py:333-341), literally ---------------------------------------------
    phi = phis[1]
    flow_mag = ref_im.get_magnitude(flows[1])
    sky_mask = np.zeros((H, W), dtype=bool)
    sky_mask[:15, :] = True
    for tag, sky in (("nosky", np.zeros((H, W), dtype=bool)), ("sky", sky_mask)):
        with np.errstate(all="ignore"):
            angle_threshold_max = phi > (0.25 + (0.5 + 8 / flow_mag))
            angle_threshold_min = phi < (0.25 - (0.5 + 8 / flow_mag))
            angle_threshold = np.logical_or(angle_threshold_min, angle_threshold_max)
            total_mask = (flow_mag > 0.5) * ~sky * angle_threshold
            fixed_angle_threshold = 15
            estimate_fixed = phi * (flow_mag > 1.0) * ~sky > fixed_angle_threshold
        out[f"thr_{tag}_total"] = total_mask
        out[f"thr_{tag}_fixed"] = estimate_fixed
    out["thr_sky"] = sky_mask
    out["mag_out"] = flow_mag

    # ---- get_simple_bounding_box (im_helpers.py:55-84) -------------------------------------------------
    m = np.zeros((H, W), np.uint8)
    m[5:9, 7:20] = 255
    r = ref_im.get_simple_bounding_box(m)
    out.update(bbox_a_in=m, bbox_a=np.array([r.topleft[0], r.topleft[1], r.size[0], r.size[1]]))
    r = ref_im.get_simple_bounding_box(np.zeros((H, W), np.uint8))
    out.update(bbox_empty=np.array([r.topleft[0], r.topleft[1], r.size[0], r.size[1]]))
    g = (rng.integers(0, 256, (H, W)) * (rng.random((H, W)) > 0.995)).astype(np.uint8)
    g[:, :3] = 0
    g[40, 100] = 250
    r = ref_im.get_simple_bounding_box(g)
    out.update(bbox_gray_in=g, bbox_gray=np.array([r.topleft[0], r.topleft[1], r.size[0], r.size[1]]))
    r = ref_im.get_simple_bounding_box(out["thr_nosky_fixed"])
    out.update(bbox_fixed=np.array([r.topleft[0], r.topleft[1], r.size[0], r.size[1]]))

    # ---- calculate_tpr_fpr (im_helpers.py:244-252) -----------------------------------------------------
    gt = np.zeros((H, W), np.uint8)
    gt[30:60, 40:70] = 255
    with np.errstate(all="ignore"):
        t = ref_im.calculate_tpr_fpr(gt, 255 * out["thr_nosky_fixed"])
    out.update(tpr_gt=gt, tpr_out=np.array(t, dtype=np.float64))

    # ---- Rectangle (utils.py:13-104) -------------------------------------------------------------------
    r1 = ref_utils.Rectangle.from_points((7, 5), (19, 8))
    r2 = ref_utils.Rectangle.from_center((15.0, 8.0), (10.0, 6.0))
    out.update(rect_vals=np.array([*r1.topleft, *r1.size, *r1.get_center(), r1.get_area(),
                                   ref_utils.Rectangle.calculate_iou(r1, r2),
                                   ref_utils.Rectangle((0, 0), (0, 0)).get_area()], dtype=np.float64),
               rect_yolo=np.array(r1.to_yolo(np.array([W, H]))))

    # ---- derotate (detector.py:70-117) -----------------------------------------------------------------
    import detector as ref_det
    dangle = np.array([0.013, -0.021, 0.008])
    dt = 1.0 / 30.0
    det = ref_det.Detector(_FakeDataset(W, H, dangle, dt))
    f32 = flows[0].astype(np.float32)
    out.update(derot_in=f32, derot_dangle=dangle, derot_dt=np.array(dt),
               derot_out=det.derotate(0, 1, f32), derot_frame0_same=np.array(det.derotate(-1, 0, f32) is f32))
    # enum surface (names + values)
    out["algo_names"] = np.array([a.name for a in ref_det.Detector.Algorithm])
    out["algo_values"] = np.array([a.value[0] for a in ref_det.Detector.Algorithm])

    # ---- full chain at 640x480 from the seeded recipe (inputs are regenerated, not stored) ------------
    W2, H2 = 640, 480
    lk2 = ref_lk.LucasKanade(np.zeros((H2, W2, 3), np.uint8))
    foe2 = ref_foe.FocusOfExpansion(lk2)
    fl2 = synth.synthetic_flow(W2, H2, seed=3)                         # float32
    det2 = ref_det.Detector(_FakeDataset(W2, H2, dangle, dt))
    der2 = det2.derotate(0, 1, fl2)
    smp = synth.foe_samples(W2, H2, 0)
    np.random.seed(1234)
    e2 = foe2.get_FOE_dense(der2)
    phi2 = foe2.get_phi(der2, e2)
    mag2 = ref_im.get_magnitude(der2)
    nosky = np.zeros((H2, W2), dtype=bool)
    with np.errstate(all="ignore"):
        hi = phi2 > (0.25 + (0.5 + 8 / mag2))
        lo = phi2 < (0.25 - (0.5 + 8 / mag2))
        total2 = (mag2 > 0.5) * ~nosky * np.logical_or(lo, hi)
        fixed2 = phi2 * (mag2 > 1.0) * ~nosky > 15
    r = ref_im.get_simple_bounding_box(fixed2)
    out.update(chain_foe=np.array(e2), chain_samples_sum=np.array(int(smp.astype(np.int64).sum())),
               chain_fixed_bits=np.packbits(fixed2), chain_total_bits=np.packbits(total2),
               chain_phi_sub=phi2[::16, ::16].copy(), chain_box=np.array([r.topleft[0], r.topleft[1], r.size[0], r.size[1]]),
               chain_max_flow=np.array(foe2.max_flow))

    np.savez_compressed(os.path.join(OUT, "foe_chain.npz"), **out)
    print("wrote", os.path.join(OUT, "foe_chain.npz"), {k: getattr(v, "shape", None) for k, v in out.items()})
    window_search(ref_det, ref_im, ref_utils)
    frame0_and_io(ref_foe, ref_im, ref_lk, ref_utils, ref_det, flows[0].astype(np.float32), foes[0], synth)


def _threshold_block(phi, flow_mag, sky):
    """processor.py:333-341, literally (dtype follows the inputs, as in the reference)."""
    with np.errstate(all="ignore"):
        angle_threshold_max = phi > (0.25 + (0.5 + 8 / flow_mag))
        angle_threshold_min = phi < (0.25 - (0.5 + 8 / flow_mag))
        angle_threshold = np.logical_or(angle_threshold_min, angle_threshold_max)
        total_mask = (flow_mag > 0.5) * ~sky * angle_threshold
        fixed_angle_threshold = 15
        estimate_fixed = phi * (flow_mag > 1.0) * ~sky > fixed_angle_threshold
    return estimate_fixed, total_mask


def frame0_and_io(ref_foe, ref_im, ref_lk, ref_utils, ref_det, flow32, foe_case0, synth):
    """tests/golden/frame0_io.npz:
    (1) the reference's FLOAT32 path -- for frame index 0 Detector.derotate returns the float32 flow itself (detector.py:80-81)
        and get_FOE_dense's |flow2| gate (:78), get_phi (:163-177) and the threshold block then run in float32;
    (2) the .flo bytes utils.write_flow produces and what utils.read_flow returns for them (utils.py:204-257);
    (3) the JSON text processor.py:83-84 writes for a filled FrameResult (utils.get_json, utils.py:350-361)."""
    import json
    import tempfile
    import frame_result as ref_fr
    out = {}
    H, W = flow32.shape[:2]
    assert flow32.dtype == np.float32
    lk = ref_lk.LucasKanade(np.zeros((H, W, 3), np.uint8))
    foe = ref_foe.FocusOfExpansion(lk)

    # ---- (1a) phi + masks in float32 --------------------------------------------------------------------------------
    e = (float(foe_case0[0]), float(foe_case0[1]))
    phi32 = foe.get_phi(flow32, e)
    mag32 = ref_im.get_magnitude(flow32)
    assert phi32.dtype == np.float32 and mag32.dtype == np.float32
    sky = np.zeros((H, W), dtype=bool)
    sky[:15, :] = True
    nosky = np.zeros((H, W), dtype=bool)
    f_ns, t_ns = _threshold_block(phi32, mag32, nosky)
    f_s, t_s = _threshold_block(phi32, mag32, sky)
    out.update(f0_flow=flow32, f0_foe=np.array(e), f0_phi=phi32, f0_mag=mag32, f0_sky=sky, f0_fixed_nosky=f_ns, f0_total_nosky=t_ns,
               f0_fixed_sky=f_s, f0_total_sky=t_s, f0_max_flow=np.array(foe.max_flow))

    # ---- (1b) the float32 |flow2| gate: a radial field of magnitude 2.5 +- rounding, so that the float32 norm and the double
    #      norm of the same float32 vector fall on different sides of 2.5 for a fair share of the pixels; every line passes through
    #      the centre, all candidates tie and RANSAC returns the FIRST one, which depends on which pairs the gate lets through ----
    yy, xx = np.mgrid[0:H, 0:W]
    dx, dy = xx - 71.3, yy - 52.9
    r = np.hypot(dx, dy)
    gate = (2.5 * np.stack([dx / r, dy / r], axis=-1)).astype(np.float32)
    m32 = ref_im.get_magnitude(gate)
    m64 = ref_im.get_magnitude(gate.astype(np.float64))
    out["f0_gate_disagree_fraction"] = np.array(float(np.mean((m32 < 2.5) != (m64 < 2.5))))
    seed = None
    for cand in range(1000, 1400):
        np.random.seed(cand)
        a = foe.get_FOE_dense(gate)
        np.random.seed(cand)
        b = foe.get_FOE_dense(gate.astype(np.float64))
        if a != b:
            seed, f32_foe, f64_foe = cand, a, b
            break
    assert seed is not None, "no seed separates the float32 gate from the double gate"
    out.update(f0_gate_flow=gate, f0_gate_seed=np.array(seed), f0_gate_foe=np.array(f32_foe), f0_gate_foe_if_double=np.array(f64_foe))

    # ---- (1c) whole chain for frame index 0 at 640x480: derotate(-1, 0, f32) is the identity ----------------------------
    W2, H2 = 640, 480
    lk2 = ref_lk.LucasKanade(np.zeros((H2, W2, 3), np.uint8))
    foe2 = ref_foe.FocusOfExpansion(lk2)
    fl2 = synth.synthetic_flow(W2, H2, seed=3)
    class _DS:
        capture_size = (W2, H2)
        def get_delta_time(self, i): return 1.0 / 30.0
        def get_angular_difference(self, a, b): return np.array([0.013, -0.021, 0.008])
    det2 = ref_det.Detector(_DS())
    der = det2.derotate(-1, 0, fl2)
    assert der is fl2 and der.dtype == np.float32
    np.random.seed(1234)
    e2 = foe2.get_FOE_dense(der)
    phi2 = foe2.get_phi(der, e2)
    mag2 = ref_im.get_magnitude(der)
    fixed2, total2 = _threshold_block(phi2, mag2, np.zeros((H2, W2), dtype=bool))
    rct = ref_im.get_simple_bounding_box(fixed2)
    out.update(chain0_foe=np.array(e2), chain0_fixed_bits=np.packbits(fixed2), chain0_total_bits=np.packbits(total2),
               chain0_phi=phi2, chain0_box=np.array([rct.topleft[0], rct.topleft[1], rct.size[0], rct.size[1]]),
               chain0_phi_dtype=np.array(str(phi2.dtype)))

    # ---- (2) .flo --------------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(21)
    flo = rng.normal(0, 3, (37, 53, 2)).astype(np.float32)
    flo[0, 0] = (np.float32(1e-30), np.float32(-1e30))
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, "a.flo")
        ref_utils.write_flow(fn, flo)
        raw = open(fn, "rb").read()
        back = ref_utils.read_flow(fn)
        ref_utils.write_flow(fn, flo[..., 0].astype(np.float64), flo[..., 1].astype(np.float64))     # separate u, v planes
        raw_uv = open(fn, "rb").read()
    out.update(flo_in=flo, flo_bytes=np.frombuffer(raw, np.uint8), flo_read=back, flo_bytes_uv=np.frombuffer(raw_uv, np.uint8),
               flo_read_dtype=np.array(str(back.dtype)))

    # ---- (3) FrameResult -> JSON text (processor.py:83-84) -----------------------------------------------------------------
    r = ref_fr.FrameResult()
    r.foe_dense = (np.float64(297.87096720308574), np.float64(222.21492686396977))
    r.foe_gt = (352.0, 216.0)
    r.center_phi = np.float64(-143.13010235415598)
    r.tpr_fixed, r.fpr_fixed = np.float64(0.75), np.float64(0.001953125)
    r.tpr, r.fpr = np.float64(1.0) / np.float64(3.0), np.float64("nan")
    r.sky_tpr, r.sky_fpr = (0.0, 0.0)
    r.drone_flow_pixels = (np.float32(6.0), np.float32(-3.0))
    r.drone_size_pixels = np.sum(np.ones((24, 24)) > 0)          # a numpy integer, as processor.py:360 produces
    r.time = 4 * (1 / 30.0)
    text = json.dumps(ref_utils.get_json(r), indent=4, sort_keys=True)
    out["json_text"] = np.array(text)
    np.savez_compressed(os.path.join(OUT, "frame0_io.npz"), **out)
    print("wrote", os.path.join(OUT, "frame0_io.npz"), "gate seed", seed, "gate disagreement", float(out["f0_gate_disagree_fraction"]))


def _blob_image(W, H, seed, blobs):
    rng = np.random.default_rng(seed)
    img = (rng.integers(0, 40, (H, W)) * (rng.random((H, W)) > 0.97)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for (cx, cy, rad, amp) in blobs:
        d2 = (xx - cx) ** 2 + (yy - cy) ** 2
        img = np.maximum(img, (amp * np.exp(-d2 / (2.0 * rad * rad))).astype(np.uint8))
    return img


def window_search(ref_det, ref_im, ref_utils):
    """Detector.optimize_window (detector.py:314-358) and the level-0 scan of analyze_pyramid (detector.py:296-310 on
    im_helpers.sliding_window :38-52).  The upper pyramid levels need imutils / cv2 and cannot be generated here."""
    out = {}
    W, H = 200, 150
    cases = [
        ("mid", _blob_image(W, H, 11, [(120, 80, 9, 220)]), (88, 48, 64, 64)),
        ("corner", _blob_image(W, H, 12, [(6, 5, 7, 200)]), (0, 0, 64, 64)),          # growth runs into the wrap-around of negative slices
        ("edge", _blob_image(W, H, 13, [(192, 140, 10, 250), (60, 60, 5, 90)]), (136, 86, 64, 64)),
        ("zero", np.zeros((H, W), np.uint8), (16, 16, 64, 64)),
        ("small", _blob_image(W, H, 14, [(100, 75, 30, 255)]), (90, 70, 8, 8)),
    ]
    for tag, img, win in cases:
        rgb = np.repeat(img[..., None], 3, axis=2)
        score, rect = ref_det.Detector.optimize_window(None, rgb, ref_utils.Rectangle((win[0], win[1]), (win[2], win[3])))
        out[f"opt_{tag}_img"] = img
        out[f"opt_{tag}_in"] = np.array(win, np.int64)
        out[f"opt_{tag}_out"] = np.array([score, rect.topleft[0], rect.topleft[1], rect.size[0], rect.size[1]], np.float64)
        # level-0 scan: the loop body of analyze_pyramid on the reference's own sliding_window generator
        best = (0, 0, 0, 0, 0)
        for (x, y, window) in ref_im.sliding_window(rgb, stepSize=16, windowSize=(64, 64)):
            if window.shape[0] != 64 or window.shape[1] != 64:
                continue
            s = np.sum(window)
            am = np.unravel_index(window.argmax(), window.shape)
            if best[0] < s:
                best = (int(s), x, y, int(am[0]), int(am[1]))
        out[f"scan_{tag}"] = np.array(best, np.int64)
    out["cases"] = np.array([c[0] for c in cases])
    np.savez_compressed(os.path.join(OUT, "window_search.npz"), **out)
    print("wrote", os.path.join(OUT, "window_search.npz"))


if __name__ == "__main__":
    main()
