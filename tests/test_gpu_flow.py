"""GPU parity for the dense-flow stage: every kernel of libmavflow's Farneback path against the C restatement
(oracle/farneback_oracle.c) on identical inputs, stage by stage and end to end, through the C-ABI.

Tolerances (floating point; the CPU path mixes f32 storage with f64 accumulators, the GPU path is f32 throughout):
  stages    absolute, stated per test
  flow      end-point error vs the oracle: mean <= 1e-4 px, p99.9 <= 1e-2 px, max <= 0.15 px (the strict gate; the shape fuzz sets pixels aside at which the oracle itself is not reproducible):
            oracle/tolerances.py   (tightened to the measured level; north_star "stated EPE tolerance")
PARITY UNPINNED vs cv2 itself: OpenCV is not installable here (see oracle/farneback_oracle.c)."""
import os

import numpy as np
import pytest

from mavflow import synth
from oracle.tolerances import check_flow

pytestmark = pytest.mark.gpu


def soa(a):            # (h, w, 5) -> (5, h, w)
    return np.ascontiguousarray(np.moveaxis(a, -1, 0))


def epe(a, b):
    return np.hypot(a[..., 0] - b[..., 0], a[..., 1] - b[..., 1])


@pytest.fixture(scope="module")
def ctx640(mav):
    from mavflow import _lib
    with _lib.Context(640, 480, 4) as c:
        yield c


@pytest.fixture(scope="module")
def pair640():
    return synth.make_pair(640, 480, 0)


def test_layers_match_oracle(ctx640, fb_oracle):
    from oracle import fb_oracle as fbo
    p = fbo.default_params()
    assert ctx640.num_layers() == fb_oracle.num_layers(640, 480, p) == 2
    for k in range(2):
        assert ctx640.layer_dims(k) == fb_oracle.layer_dims(640, 480, p, k)


@pytest.mark.parametrize("k", [0, 1])
def test_blur_resize(ctx640, fb_oracle, pair640, k):
    w, h, sigma, ks = ctx640.layer_dims(k)
    got = ctx640.stage_blur_resize(pair640[0], k)
    exp = fb_oracle.blur_resize(pair640[0], w, h, ks, sigma)
    np.testing.assert_allclose(got, exp, rtol=0, atol=2e-4)


@pytest.mark.parametrize("size,levels", [((640, 480), 1), ((1920, 1080), 1), ((1000, 562), 1), ((333, 227), 1), ((3840, 2160), 5)])
def test_fused_blur_resize_equals_the_two_pass_form(mav, fb_oracle, size, levels):
    """Layers with a short Gaussian (ksize <= 13) go through ONE kernel that keeps the horizontal pass in LDS; the separable
    two-pass form (H x w scratch in memory) stays for the long ones.  Same functions, same tap order: bit-identical layer images,
    interior and borders, ragged sizes, noise as well as texture -- and both inside the oracle's tolerance."""
    from mavflow import _lib
    W, H = size
    rng = np.random.default_rng(5)
    imgs = [synth.make_pair(W, H, 3)[0], rng.integers(0, 256, (H, W), dtype=np.uint8)]
    with _lib.Context(W, H, 1, _lib.fb_defaults(levels=levels)) as c:
        fused_layers = [l["layer"] for l in c.schedule_info(1)["layers"] if l["blur"] == "fused"]
        assert fused_layers == ([1] if levels == 1 else [1, 2])
        for k in range(1, c.num_layers()):
            w, h, sigma, ks = c.layer_dims(k)
            for img in imgs:
                a, b = c.stage_blur_resize(img, k), c.stage_blur_resize(img, k, two_pass=True)
                assert np.array_equal(a, b), (k, int((a != b).sum()))
                np.testing.assert_allclose(a, fb_oracle.blur_resize(img, w, h, ks, sigma), rtol=0, atol=2e-4)


@pytest.mark.parametrize("k", [0, 1])
def test_polyexp(ctx640, fb_oracle, pair640, k):
    w, h, sigma, ks = ctx640.layer_dims(k)
    I = fb_oracle.blur_resize(pair640[0], w, h, ks, sigma)
    got = ctx640.stage_polyexp(I, k)
    exp = soa(fb_oracle.polyexp(I))
    np.testing.assert_allclose(got, exp, rtol=0, atol=2e-4)      # |R| ~ 1e1, f32 sums of ~300 terms of size ~1e2


def _stage_inputs(ctx, fb_oracle, pair, k):
    w, h, sigma, ks = ctx.layer_dims(k)
    R = [fb_oracle.polyexp(fb_oracle.blur_resize(pair[i], w, h, ks, sigma)) for i in range(2)]
    rng = np.random.default_rng(3)
    flow = (synth.true_flow(w, h, k=0.01) + rng.normal(0, 0.2, (h, w, 2))).astype(np.float32)
    flow[0, 0] = (-5.0, -7.0)            # leaves the image -> border branch
    flow[h - 1, w - 1] = (9.0, 3.0)
    return R, flow


@pytest.mark.parametrize("k", [0, 1])
def test_update_matrices(ctx640, fb_oracle, pair640, k):
    R, flow = _stage_inputs(ctx640, fb_oracle, pair640, k)
    got = ctx640.stage_update_matrices(soa(R[0]), soa(R[1]), flow, k)
    exp = soa(fb_oracle.update_matrices(R[0], R[1], flow))
    scale = np.abs(exp).max()
    np.testing.assert_allclose(got, exp, rtol=1e-4, atol=1e-6 * scale)


@pytest.mark.parametrize("k,update", [(0, True), (1, True), (0, False)])
def test_blur_iter(ctx640, fb_oracle, pair640, k, update):
    R, flow = _stage_inputs(ctx640, fb_oracle, pair640, k)
    M = fb_oracle.update_matrices(R[0], R[1], flow)
    eflow, eM = fb_oracle.blur_iter(R[0], R[1], flow, M, 12, update)
    gflow, gM = ctx640.stage_blur_iter(soa(R[0]), soa(R[1]), soa(M), k, update)
    e = epe(gflow, eflow)
    assert e.max() < 2e-3, e.max()
    if update:
        scale = np.abs(eM).max()
        np.testing.assert_allclose(gM, soa(eM), rtol=2e-3, atol=2e-5 * scale)


def _check_flow(got, exp, tag=""):
    return check_flow(got, exp, tag)


def test_flow_640x480(ctx640, fb_oracle, pair640):
    """BASELINE config 1 shape on the GPU path."""
    got = ctx640.farneback(pair640[0], pair640[1])[0]
    exp = fb_oracle.calc(pair640[0], pair640[1])
    e = _check_flow(got, exp, "640x480")
    print(f"\nEPE vs oracle 640x480: mean {e.mean():.3e} p99.9 {np.percentile(e, 99.9):.3e} max {e.max():.3e}")


def test_flow_batch_and_groups(ctx640, fb_oracle):
    """batch > group exercises the group loop; every slot must equal its single-pair result."""
    prev, nxt = synth.make_batch(640, 480, 3, distinct=3)
    ctx640.set_option("group", 2)
    got = ctx640.farneback(prev, nxt)
    ctx640.set_option("group", 1)
    one = ctx640.farneback(prev, nxt)
    assert np.array_equal(got, one)
    for b in (0, 2):
        _check_flow(got[b], fb_oracle.calc(prev[b], nxt[b]), f"batch {b}")


def test_flow_ragged_size(mav, fb_oracle):
    """Sizes that are no multiple of any tile edge, one layer only (0.4 * 50 < 32)."""
    from mavflow import _lib
    W, H = 117, 50
    f0, f1, _ = synth.make_pair(W, H, 4, patch=False)
    with _lib.Context(W, H, 1) as c:
        assert c.num_layers() == 1
        got = c.farneback(f0, f1)[0]
    _check_flow(got, fb_oracle.calc(f0, f1), "117x50")


def test_flow_other_params(mav, fb_oracle):
    """Non-default parameters take the generic (runtime-size) kernel paths."""
    from mavflow import _lib
    from oracle import fb_oracle as fbo
    W, H = 320, 240
    f0, f1, _ = synth.make_pair(W, H, 2)
    fb = _lib.fb_defaults()
    fb.pyr_scale, fb.levels, fb.winsize, fb.iterations, fb.poly_n, fb.poly_sigma = 0.5, 3, 15, 3, 5, 1.1
    p = fbo.Params(0.5, 3, 15, 3, 5, 1.1, 0)
    with _lib.Context(W, H, 1, fb) as c:
        assert c.num_layers() == fb_oracle.num_layers(W, H, p) == 3
        got = c.farneback(f0, f1)[0]
    _check_flow(got, fb_oracle.calc(f0, f1, p), "pyr 0.5 / 3 levels / win 15 / n 5")


def test_flow_720p(mav, fb_oracle):
    """BASELINE config 2: 1280x720, batch 1."""
    from mavflow import _lib
    f0, f1, truth = synth.make_pair(1280, 720, 1)
    with _lib.Context(1280, 720, 1) as c:
        got = c.farneback(f0, f1)[0]
    e = _check_flow(got, fb_oracle.calc(f0, f1), "720p")
    print(f"\nEPE vs oracle 720p: mean {e.mean():.3e} p99.9 {np.percentile(e, 99.9):.3e}")


def test_zero_and_constant_frames(ctx640):
    z = np.zeros((480, 640), np.uint8)
    flow = ctx640.farneback(z, z)[0]
    assert np.all(flow == 0)
    c = np.full((480, 640), 200, np.uint8)
    flow = ctx640.farneback(c, c)[0]
    assert np.abs(flow).max() < 1e-3


def test_shape_errors(ctx640):
    with pytest.raises(ValueError):
        ctx640.farneback(np.zeros((480, 641), np.uint8), np.zeros((480, 641), np.uint8))
    with pytest.raises(ValueError):
        ctx640.farneback(np.zeros((5, 480, 640), np.uint8), np.zeros((5, 480, 640), np.uint8))   # batch > max_batch


def test_flow_4k_five_layers(mav, fb_oracle):
    """BASELINE config 5 shape: 3840x2160, levels=5 (blur kernels of 95/37/13/5/3 taps), one pair."""
    from mavflow import _lib
    from oracle import fb_oracle as fbo
    W, H = 3840, 2160
    f0, f1, _ = synth.make_pair(W, H, 7, k=0.004)
    fb = _lib.fb_defaults(levels=5)
    with _lib.Context(W, H, 1, fb) as c:
        assert c.num_layers() == 5 and [c.layer_dims(k)[3] for k in range(5)] == [3, 5, 13, 37, 95]
        got = c.farneback(f0, f1)[0]
    e = _check_flow(got, fb_oracle.calc(f0, f1, fbo.default_params(levels=5)), "4K / 5 layers")
    print(f"\nEPE vs oracle 4K: mean {e.mean():.3e} p99.9 {np.percentile(e, 99.9):.3e}")


def test_flow_vs_cv2_when_available(ctx640, pair640):
    """The real bar: EPE against cv2.calcOpticalFlowFarneback itself.  OpenCV is not installed in the build
    image nor on the GPU box, so this is skipped there; it runs unchanged wherever `import cv2` works."""
    cv2 = pytest.importorskip("cv2")
    ref = cv2.calcOpticalFlowFarneback(pair640[0], pair640[1], None, 0.4, 1, 12, 10, 8, 1.2, 0)
    got = ctx640.farneback(pair640[0], pair640[1])[0]
    _check_flow(got, ref, "vs cv2")


def test_shape_and_parameter_fuzz(mav):
    """tools/fuzz_shapes.py, 15 cases: random sizes (multiples of 4 and not), batch sizes, pyramid / window / poly parameters
    and noise frames through the fused entry point -- flow within the EPE gate, FoE / masks / boxes bit-exact."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_shapes.py"), "15", "7"], cwd=root, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "all 15 cases passed" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _fuzz_case(seed, case):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        from tools.fuzz_shapes import fuzz_cases
    finally:
        sys.path.remove(root)
    return list(fuzz_cases(case + 1, seed))[case]


@pytest.mark.parametrize("seed,case,shape,worst_pair,mechanism", [
    (123, 10, (1048, 925, 6, 5), 5, "expanding"),      # round 5's 0.269 px, the frame the maximum gate was raised for
    (4, 65, (1096, 605, 5, 3), 3, "expanding"),        # the worst frame six seeds of the fuzz hold: 1.49 px
    (4, 21, (552, 521, 3, 5), 1, "border cycle"),      # 0.146 px along the right image border: a three-sweep limit cycle, one sweep apart
    (5, 43, (1196, 905, 3, 5), 1, "expanding"),        # out of sample (seeds 5 - 10, run after the gate was written): 0.35 px, and 5.4 % of the
                                                       # frame unstable -- the one bound of the first form that did not hold (it said 5 %)
])
def test_frames_outside_the_strict_gate_are_unstable_in_the_oracle_itself(mav, fb_oracle, seed, case, shape, worst_pair, mechanism):
    """The frames of tools/fuzz_shapes.py on which GPU and restatement part visibly, pinned as what profiles/r06/worst_pixel*.txt and
    border_cycle.txt show them to be: NOT a kernel defect (every single sweep of the finest layer, applied to the restatement's own
    M / R0 / R1, agrees with it to < 1e-4 px everywhere) and NOT an ill-conditioned 2x2 system (the determinant at the worst pixel
    loses < 5x to cancellation), but pixels at which the restatement's own result is not reproducible -- its float32-sums twins
    part, or a pixel of the window sits on the image-border test.  With those set aside the rest of the frame is inside a TIGHTER
    gate than the strict one (oracle/tolerances.py)."""
    from mavflow import _lib
    from oracle import tolerances as tol
    cs = _fuzz_case(seed, case)
    assert (cs["W"], cs["H"], cs["B"], cs["fb"].levels) == shape
    fb, po, b = cs["fb"], cs["po"], worst_pair
    with _lib.Context(cs["W"], cs["H"], cs["B"], fb) as c:
        got = c.farneback(cs["prev"], cs["nxt"])
    prev, nxt = cs["prev"][b], cs["nxt"][b]
    ref, rec, flips = fb_oracle.calc_tracked(prev, nxt, po)
    twins = fb_oracle.twins(prev, nxt, po)
    e = tol.check_flow(got[b], ref, f"seed {seed} case {case} pair {b}", twins, fb.winsize // 2, flips)      # the two-class gate holds
    un = tol.unstable_mask(ref, twins, fb.winsize // 2, flips)
    y, x = np.unravel_index(int(e.argmax()), e.shape)
    assert un[y, x] and e[~un].max() <= tol.FLOW_EPE_MAX_STABLE, "the large errors sit on unstable pixels only"
    assert tol.conditioning(rec[y, x][None])[1][0] < 5.0, "the worst pixel's 2x2 system is WELL conditioned"
    S = tol.sensitivity(ref, twins, fb.winsize // 2)
    if mechanism == "expanding":
        assert not tol.flow_epe_ok(e), "outside the strict gate"
        assert S[y, x] >= tol.FLOW_UNSTABLE_S and tol.last_step(ref, rec)[y, x] > 0.2, "the oracle's twins part there; its iteration still moves"
    else:
        assert e.max() > 0.1 and S[y, x] < tol.FLOW_UNSTABLE_S, "the twins agree there ..."
        assert tol._window_max((flips > 0).astype(np.float64), fb.winsize // 2)[y, x] > 0, "... but a pixel of the window flips the border test"
    # per sweep the kernel is right: the finest layer's ten sweeps of the GPU on the ORACLE's own intermediates
    steps = []
    fb_oracle.pyramid(prev, nxt, po, lambda k, it, fl, M, s, R0, R1: steps.append((it, fl.copy(), M, R0, R1)) if k == 0 else None)
    with _lib.Context(cs["W"], cs["H"], 1, fb) as c1:
        worst_sweep = 0.0
        for it, of, oM, oR0, oR1 in steps:
            f1, _ = c1.stage_blur_iter(soa(oR0), soa(oR1), soa(oM), 0, it < fb.iterations - 1)
            worst_sweep = max(worst_sweep, float(tol.epe(f1, of).max()))
    assert worst_sweep < 1e-4, worst_sweep
    print(f"\nseed {seed} case {case} pair {b}: max EPE {e.max():.3f} px at ({x}, {y}), S there {S[y, x]:.3f}; {int(un.sum())} unstable pixels "
          f"({un.mean():.2e} of the frame); stable rest: mean {e[~un].mean():.2e} p99.9 {np.percentile(e[~un], 99.9):.2e} max {e[~un].max():.2e}; "
          f"one GPU sweep on the oracle's M: <= {worst_sweep:.1e} px")


def test_overlapped_upload_path(mav):
    """Pinned memory + copy stream + fence: frames uploaded asynchronously give the same flow as the synchronous entry point."""
    from mavflow import _lib
    W, H, B = 320, 240, 2
    prev, nxt = synth.make_batch(W, H, B, distinct=2)
    with _lib.Context(W, H, B) as c:
        ref = c.farneback(prev, nxt)
        hp, hn = c.pinned_like(prev), c.pinned_like(nxt)
        dp, dn, df = c.alloc(prev.nbytes), c.alloc(nxt.nbytes), c.alloc(ref.nbytes)
        for _ in range(3):                                   # re-use of the same buffers across iterations must stay ordered
            c.upload_async(dp, hp); c.upload_async(dn, hn); c.upload_fence()
            c.farneback_dev(dp.ptr, dn.ptr, B, df.ptr)
        c.sync()
        got = df.download(np.float32, ref.shape)
    assert np.array_equal(got, ref)


def test_double_buffered_uploads_with_different_content(mav):
    """The documented deployment pattern -- upload batch k+1 into the other buffer set while batch k computes -- with DIFFERENT
    frames in every batch: an upload must not overwrite a set that an earlier, still running batch reads (mav_upload_async
    orders the copy stream behind the compute stream), and a batch must not start before its own upload has landed
    (mav_upload_fence).  Every batch's flow must equal the synchronous entry point's."""
    from mavflow import _lib
    W, H, B, N = 640, 480, 4, 6
    batches = []
    for k in range(N):
        prev, nxt = synth.make_batch(W, H, B, distinct=2)
        batches.append((np.roll(prev, 17 * k + 3, axis=2), np.roll(nxt, 17 * k + 3, axis=2)))
    with _lib.Context(W, H, B) as c:
        ref = [c.farneback(p, n) for p, n in batches]
        pinned = [(c.pinned_like(p), c.pinned_like(n)) for p, n in batches]
        sets = [(c.alloc(batches[0][0].nbytes), c.alloc(batches[0][1].nbytes)) for _ in range(2)]
        outs = [c.alloc(ref[0].nbytes) for _ in range(N)]
        c.upload_async(sets[0][0], pinned[0][0]); c.upload_async(sets[0][1], pinned[0][1]); c.upload_fence()
        for k in range(N):
            cur, nx = sets[k & 1], sets[(k + 1) & 1]
            if k + 1 < N:                                   # batch k+1 crosses PCIe into the set batch k-1 may still be reading
                c.upload_async(nx[0], pinned[k + 1][0]); c.upload_async(nx[1], pinned[k + 1][1])
            c.farneback_dev(cur[0].ptr, cur[1].ptr, B, outs[k].ptr)
            c.upload_fence()
        c.sync()
        for k in range(N):
            assert np.array_equal(outs[k].download(np.float32, ref[k].shape), ref[k]), k


@pytest.mark.parametrize("size", [(640, 480), (1920, 1080)])
def test_band_major_sweeps_are_bit_identical(mav, size):
    """Option "bands": the finest layer's sweeps of a pair in band-major order over skewed horizontal bands (how frames beyond
    the Infinity Cache are swept).  Same tiles, same arithmetic: the flow must equal the sweep-major schedule bit for bit, for
    one pair per call and for per-pair sweeps inside a batch."""
    from mavflow import _lib
    W, H = size
    prev, nxt = synth.make_batch(W, H, 3, distinct=3)
    with _lib.Context(W, H, 3) as c:
        c.set_option("pairs_in_flight", 1)
        c.set_option("bands", 1)
        ref = c.farneback(prev, nxt)
        one = c.farneback(prev[:1], nxt[:1])
        for pif in (1, 2):                               # one stream, and two pairs in flight on two streams
            c.set_option("pairs_in_flight", pif)
            for bands in (1, 2, 3):
                c.set_option("bands", bands)
                assert np.array_equal(c.farneback(prev, nxt), ref), (pif, bands)
                assert np.array_equal(c.farneback(prev[:1], nxt[:1]), one), (pif, bands)
        c.set_option("group", 2)                       # a group of 2 + a group of 1
        assert np.array_equal(c.farneback(prev, nxt), ref)


@pytest.mark.parametrize("size,batch", [((1280, 720), 1), ((1280, 720), 2), ((640, 480), 3), ((1920, 1080), 1), ((1000, 562), 1), ((58, 174), 3)])
def test_small_groups_give_the_same_results_as_big_ones(mav, size, batch):
    """Small groups (BASELINE config 2: one 1280x720 pair per call) send prev and next through ONE blur and ONE expansion launch per
    layer (two runs of images in one grid) where big groups launch per frame set.  Same kernels on the same data: flow, masks and
    records of a call must not depend on how the batch is cut into groups or calls -- one pair at a time, the whole batch, frames in
    adjacent or far-apart device buffers -- bit for bit, call after call."""
    from mavflow import _lib
    W, H = size
    prev, nxt = synth.make_batch(W, H, batch, distinct=min(batch, 4))
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(batch)])
    with _lib.Context(W, H, max(batch, 4)) as c:
        # small groups also get their whole pyramid from two launches (option "small_batch"): the reference is the per-layer form
        c.set_option("small_batch", 0)
        assert not c.schedule_info(batch)["pyramid_in_two_launches"]
        ref = c.process_batch(prev, nxt, smp)
        c.set_option("small_batch", 1)
        assert c.schedule_info(1)["pyramid_in_two_launches"] == (c.num_layers() > 1)
        whole = c.process_batch(prev, nxt, smp)
        assert np.array_equal(whole["flow"], ref["flow"]) and whole["results"].tobytes() == ref["results"].tobytes()
        seq = synth.make_sequence(W, H, batch + 1)                       # a frame sequence through the same form
        c.set_option("small_batch", 0)
        seq_ref = c.farneback_sequence(seq)
        c.set_option("small_batch", 1)
        assert np.array_equal(c.farneback_sequence(seq), seq_ref)
        for rep in range(2):
            for b in range(batch):
                one = c.process_batch(prev[b:b + 1], nxt[b:b + 1], smp[b:b + 1])
                for key in ("flow", "mask_fixed", "mask_dyn"):
                    assert np.array_equal(one[key][0], ref[key][b]), (rep, b, key)
                assert one["results"].tobytes() == ref["results"][b:b + 1].tobytes()
        c.set_option("group", 1)
        out = c.process_batch(prev, nxt, smp)
        assert np.array_equal(out["flow"], ref["flow"]) and out["results"].tobytes() == ref["results"].tobytes()
        c.set_option("group", 4)
        # device pointers, frames in two far-apart allocations (not a frame sequence)
        dp = c.alloc(prev.nbytes).upload(prev)
        gap = c.alloc(3 << 20)
        dn = c.alloc(nxt.nbytes).upload(nxt)
        out = c.alloc(ref["flow"].nbytes)
        for rep in range(3):
            c.farneback_dev(dp.ptr, dn.ptr, batch, out.ptr)
        c.sync()
        assert np.array_equal(out.download(np.float32, ref["flow"].shape), ref["flow"])
        del gap


def test_options_are_reported_and_validated(mav):
    """Every scheduling switch is an option of the context (the library reads no environment variable): set / get round trip,
    range errors as ValueError, and mav_schedule_info reports the options and the per-layer plan a call will take."""
    from mavflow import _lib
    src = "".join(open(os.path.join(os.path.dirname(_lib.__file__), "..", "csrc", f)).read()
                  for f in ("mavflow.cpp", "kernels_flow.hip", "kernels_detect.hip", "kernels_window.hip"))
    assert "getenv" not in src
    with _lib.Context(1920, 1080, 64) as c:
        info = c.schedule_info(64)
        assert info["group"] == 16 and info["pairs_in_flight"] == 2 and info["pairs_per_group"] == 16
        assert [(l["w"], l["h"]) for l in info["layers"]] == [(1920, 1080), (768, 432)]
        assert info["layers"][0]["sweeps"].startswith("two pairs in flight") and info["layers"][0]["bands"] == 2
        assert info["layers"][1]["blur"] == "fused" and info["layers"][1]["pairs_per_launch"] == 4
        for name, v in (("band_mb", 40), ("coarse_half", 3), ("strip", 20), ("phi_yloop", 4), ("phi_screen", 0), ("share_m", 0),
                        ("coarse_cache_mb", 100), ("bands", 3), ("group_fine", 2), ("small_batch", 0), ("sweep_write_through", 1),
                        ("deep_batch", 0), ("coarse_bands", 1), ("band_phase", 2), ("deep_frac", 32), ("band_skew", 0)):
            c.set_option(name, v)
            assert c.get_option(name) == v
            assert c.schedule_info(64)[name] == v
        c.set_option("bands", 0)                                     # back to automatic
        c.set_option("group_fine", 1)
        c.set_option("band_mb", 96)
        assert c.schedule_info(64)["layers"][0]["bands"] == 2
        for name, v in (("pairs_in_flight", 3), ("bands", 9), ("group", 0), ("no_such_option", 1), ("recompute", 1), ("pipeline", 1)):
            with pytest.raises(ValueError):
                c.set_option(name, v)
    with _lib.Context(640, 480, 2) as c:                                 # the deep set is sized by deep_frac: fixed once flow has been computed
        assert c.schedule_info(2)["deep_frac"] == 6
        prev, nxt = synth.make_batch(640, 480, 2, distinct=2)
        c.farneback(prev, nxt)
        with pytest.raises(_lib.MavflowError):
            c.set_option("deep_frac", 32)


def test_winsize_beyond_the_fast_kernel(mav, fb_oracle):
    """winsize 20 takes the general sweep kernel with 53 KB... up to 160 KB of dynamic LDS (granted per device in mav_create)."""
    from mavflow import _lib
    from oracle import fb_oracle as fbo
    W, H = 320, 240
    f0, f1, _ = synth.make_pair(W, H, 6)
    fb = _lib.fb_defaults()
    fb.winsize, fb.iterations = 40, 2                   # 5 x (32 + 40)^2 floats = 104 KB > the 64 KB default limit
    with _lib.Context(W, H, 1, fb) as c:
        got = c.farneback(f0, f1)[0]
    _check_flow(got, fb_oracle.calc(f0, f1, fbo.Params(0.4, 1, 40, 2, 8, 1.2, 0)), "winsize 40")


def test_frame_sequence_shares_expansions_bit_identical(mav):
    """A video (n + 1 frames, pair i = frames i, i + 1: src/farneback.py:76-80 with prevgray carried over) handed over as two views of
    ONE array is recognised (next == prev + one frame): uploaded once, every frame blurred and expanded once per group.  The flow must
    equal the two-batch form on separate copies bit for bit -- host pointers, device pointers, the fused chain, groups of 2 + 2 + 1,
    a single pair -- and option share_frames = 0 must change nothing but the work done."""
    from mavflow import _lib
    W, H, B = 640, 480, 5
    frames = synth.make_sequence(W, H, B + 1)
    assert (frames[0] != frames[1]).mean() > 0.5
    prev, nxt = frames[:-1].copy(), frames[1:].copy()           # separate arrays: no aliasing to recognise
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
    with _lib.Context(W, H, B) as c:
        c.set_option("group", 2)
        ref = c.farneback(prev, nxt)
        assert np.array_equal(c.farneback_sequence(frames), ref)
        assert np.array_equal(c.farneback(frames[:-1], frames[1:]), ref)
        assert np.array_equal(c.farneback_sequence(frames[:2]), ref[:1])                 # one pair: two frames in a group of one
        assert np.array_equal(c.farneback_sequence(frames[1:5]), ref[1:4])
        # device pointers: one buffer holding the run
        d = c.alloc(frames.nbytes); d.upload(frames)
        out = c.alloc(ref.nbytes)
        c.farneback_dev(d.ptr, d.ptr + W * H, B, out.ptr)
        c.sync()
        assert np.array_equal(out.download(np.float32, ref.shape), ref)
        # the fused chain
        a = c.process_batch(prev, nxt, smp)
        b = c.process_batch(frames[:-1], frames[1:], smp)
        for key in ("flow", "mask_fixed", "mask_dyn"):
            assert np.array_equal(a[key], b[key]), key
        assert a["results"].tobytes() == b["results"].tobytes()
        c.set_option("share_frames", 0)
        assert np.array_equal(c.farneback_sequence(frames), ref)
        c.set_option("share_frames", 1)
        c.set_option("group", 5)                                                            # one group of five pairs = six frames
        assert np.array_equal(c.farneback_sequence(frames), ref)


@pytest.mark.parametrize("size,batch,group", [((640, 480), 5, 4), ((640, 480), 7, 3), ((1000, 562), 4, 4), ((1920, 1080), 5, 5)])
def test_two_pairs_in_flight_give_the_same_flow(mav, size, batch, group):
    """The default schedule of the finest layer: pair s of a group on stream s & 1, M through slot s & 1, every pair band-major with
    its initial M built band by band (option "pairs_in_flight" = 2).  Same tiles and arithmetic as the one-stream, sweep-major,
    whole-frame-initial-M schedule: bit-identical flow, call after call, for odd and even group sizes and a ragged last group, and
    through the fused chain."""
    from mavflow import _lib
    W, H = size
    prev, nxt = synth.make_batch(W, H, batch, distinct=min(batch, 4))
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(batch)])
    with _lib.Context(W, H, batch) as c:
        c.set_option("group", group)
        c.set_option("pairs_in_flight", 1)
        ref = c.farneback(prev, nxt)
        chain = c.process_batch(prev, nxt, smp)
        c.set_option("pairs_in_flight", 2)
        for rep in range(3):
            out = c.farneback(prev, nxt)
            assert np.array_equal(out, ref), (rep, int((out != ref).sum()))
        assert np.array_equal(c.farneback(prev[:3], nxt[:3]), ref[:3])
        assert np.array_equal(c.farneback(prev[:1], nxt[:1]), ref[:1])
        two = c.process_batch(prev, nxt, smp)
        for key in ("flow", "mask_fixed", "mask_dyn"):
            assert np.array_equal(two[key], chain[key]), key
        assert two["results"].tobytes() == chain["results"].tobytes()
        for wt in (0, 1, -1):                        # the sweeps' M' through plain or write-through (sc1) stores: the same values either way
            c.set_option("sweep_write_through", wt)
            assert np.array_equal(c.farneback(prev, nxt), ref), wt
        # the profile's interval union: launches of one class overlap, the busy time stays below their sum
        c.profile_enable(True)
        c.farneback(prev, nxt)
        prof = c.profile_get()
        busy = c.profile_busy("blur_iter")
        c.profile_enable(False)
        assert 0 < busy <= prof["blur_iter"][0] * 1.0001


@pytest.mark.parametrize("size,levels,batch,group,band_mb", [((1920, 1080), 3, 5, 2, 8), ((1000, 562), 4, 7, 3, 8), ((640, 480), 2, 6, 2, 8),
                                                             ((3840, 2160), 5, 3, 2, 96)])
def test_deep_layers_once_per_call_and_banded_coarse_layers_are_bit_identical(mav, size, levels, batch, group, band_mb):
    """Round 4's two schedule changes for many-layer pyramids (BASELINE config 5: 3840x2160, five layers).
    "deep_batch": the layers at the top of the pyramid (each at most 1/32 of the frame) run ONCE for all pairs of a call -- their images
    from one launch, their expansions from one launch, their sweeps over all pairs -- and the groups start below them.
    "coarse_bands": a coarse layer whose per-pair working set exceeds a band is swept like the finest layer (pairs alternating between
    the two streams, band-major, initial M band by band).  Same tile functions on the same data: the flow must equal the per-group,
    one-stream schedule bit for bit -- ragged last group, a frame sequence, one pair, call after call -- and the chain after it too."""
    from mavflow import _lib
    W, H = size
    prev, nxt = synth.make_batch(W, H, batch, distinct=min(batch, 3))
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(batch)])
    with _lib.Context(W, H, batch, _lib.fb_defaults(levels=levels)) as c:
        c.set_option("group", group)
        c.set_option("band_mb", band_mb)
        if size == (1000, 562):
            c.set_option("deep_frac", 32)                    # (only layers 2 .. are deep then: the groups start at layer 1)
        for name in ("deep_batch", "coarse_bands"):
            c.set_option(name, 0)
        c.set_option("pairs_in_flight", 1)
        info = c.schedule_info(batch)
        assert info["deep_layers_from"] == 0 and all(l["sweeps"] == "one stream" for l in info["layers"])
        ref = c.farneback(prev, nxt)
        chain = c.process_batch(prev, nxt, smp)
        seq = synth.make_sequence(W, H, batch + 1)
        seq_ref = c.farneback_sequence(seq)
        c.set_option("pairs_in_flight", 2)
        for deep, cb in ((1, 0), (0, 1), (1, 1)):
            c.set_option("deep_batch", deep)
            c.set_option("coarse_bands", cb)
            info = c.schedule_info(batch)
            n_layers = len(info["layers"])
            if deep and n_layers > 1:
                assert info["deep_layers_from"] == (2 if size == (1000, 562) else 1) and info["deep_pairs"] == batch
            w1, h1 = info["layers"][1]["w"], info["layers"][1]["h"]
            big = w1 % 4 == 0 and w1 * h1 * 80 > (band_mb << 20) and ((h1 + 15) // 16) // 12 >= 2
            assert big == (size in ((1920, 1080), (3840, 2160)))
            if cb and big:
                assert info["layers"][1]["sweeps"].startswith("two pairs in flight") and info["layers"][1]["bands"] >= 2, info["layers"][1]
            for rep in range(2):
                out = c.farneback(prev, nxt)
                assert np.array_equal(out, ref), (deep, cb, rep, int((out != ref).sum()))
            assert np.array_equal(c.farneback_sequence(seq), seq_ref), (deep, cb)
            assert np.array_equal(c.farneback(prev[:1], nxt[:1]), ref[:1])
            assert np.array_equal(c.farneback(prev[:group + 1], nxt[:group + 1]), ref[:group + 1])
        # "band_phase": the second stream's pairs on a partition shifted by half a band (J + 1 bands, the outer two of half size), so that
        # one stream's initial-M launches fall into the other's sweeps; with band_mb = 8 the finest layer has up to 5 bands of 12 tile rows
        for bp, mb in ((0, band_mb), (1, band_mb), (1, max(band_mb, 20)), (3, band_mb)):
            c.set_option("band_phase", bp)
            c.set_option("band_mb", mb)
            for rep in range(2):
                out = c.farneback(prev, nxt)
                assert np.array_equal(out, ref), ("band_phase", bp, mb, rep, int((out != ref).sum()))
        c.set_option("band_mb", band_mb)
        c.set_option("band_phase", 0)
        # "band_skew": boundaries moved down by (iterations - 1) / 2 tile rows (default) so that every band has the same average size over
        # its sweeps; 0 = equal bands, other shifts for the test -- any monotone partition gives the same flow
        for bs in (0, 2, 7, -1):
            c.set_option("band_skew", bs)
            out = c.farneback(prev, nxt)
            assert np.array_equal(out, ref), ("band_skew", bs, int((out != ref).sum()))
        c.set_option("band_phase", 1)
        two = c.process_batch(prev, nxt, smp)
        for key in ("flow", "mask_fixed", "mask_dyn"):
            assert np.array_equal(two[key], chain[key]), key
        assert two["results"].tobytes() == chain["results"].tobytes()
