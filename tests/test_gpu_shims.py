"""The reference-shaped Python classes on the GPU path: same calls, same seeds, same answers as the fixtures the
reference's own code produced (tests/golden) and as the oracle."""
import logging

import numpy as np
import pytest

from oracle import foe_oracle as fo
from oracle.tolerances import check_flow
from mavflow import synth

pytestmark = pytest.mark.gpu
PHI_ATOL = 4 * np.spacing(180.0)


class FakeDataset:
    def __init__(self, W, H, dangle, dt):
        self.capture_size = (W, H)
        self.dangle, self.dt = np.asarray(dangle, np.float64), dt

    def get_delta_time(self, i):
        return self.dt

    def get_angular_difference(self, a, b):
        return self.dangle


def test_focus_of_expansion_seeded_like_the_reference(mav, golden):
    from mavflow.detector import LucasKanade
    from mavflow.focus_of_expansion import FocusOfExpansion
    foe = FocusOfExpansion(LucasKanade(np.zeros((120, 160, 3), np.uint8)))
    for c in range(4):
        flow = golden["foe_flow"][c]
        if golden["foe_flow_is_f32"][c]:
            flow = flow.astype(np.float32)
        np.random.seed(int(golden["foe_seed"][c]))
        got = foe.get_FOE_dense(flow)
        assert isinstance(got, tuple) and got == tuple(golden["foe_out"][c]), c
        phi = foe.get_phi(golden["foe_flow"][c], got)
        np.testing.assert_allclose(phi, golden["phi_out"][c], rtol=0, atol=PHI_ATOL)
        assert abs(foe.max_flow - golden["phi_out"][c].max()) <= PHI_ATOL
    assert foe.get_phi(golden["foe_flow"][0], (np.nan, np.nan)).shape == (0,)
    fixed, total = foe.get_masks(golden["foe_flow"][1], tuple(golden["foe_out"][1]), golden["thr_sky"])
    assert np.array_equal(fixed, golden["thr_sky_fixed"]) and np.array_equal(total, golden["thr_sky_total"])


def test_constructors_consume_the_same_random_draws(mav):
    """Detector / LucasKanade / FocusOfExpansion constructors must leave the global RNG where the reference leaves it."""
    from mavflow.detector import Detector
    from mavflow.focus_of_expansion import FocusOfExpansion
    np.random.seed(7)
    d = Detector(FakeDataset(160, 120, (0, 0, 0), 1.0))
    FocusOfExpansion(d.lucas_kanade)
    after = np.random.randint(0, 1 << 30)
    np.random.seed(7)
    np.random.randint(20, 100, 1000); np.random.randint(20, 140, 1000)          # detector.py:33-36
    np.random.randint(0, 255, (2666, 3))                                          # lucas_kanade.py:32
    np.random.randint(0, 255, (2666, 3)); np.random.randint(0, 2666, 2666)        # focus_of_expansion.py:24,26
    assert after == np.random.randint(0, 1 << 30)


def test_detector_derotate_and_window(mav, golden):
    from mavflow.detector import Detector
    dt = float(golden["derot_dt"])
    d = Detector(FakeDataset(160, 120, golden["derot_dangle"], dt))
    f = golden["derot_in"]
    assert d.derotate(-1, 0, f) is f
    out = d.derotate(0, 1, f)
    assert out.dtype == np.float64 and out.tobytes() == golden["derot_out"].tobytes()
    assert d.algorithm is Detector.Algorithm.ESSENTIAL and not d.is_homography_based()
    img = np.zeros((120, 160), np.uint8)
    img[40:60, 70:100] = 200
    from mavflow import im_helpers
    rgb = im_helpers.to_rgb(img.astype(np.float64), 255.0)
    score, rect, window, amax = d.analyze_pyramid(rgb)
    es, ex, ey = fo.analyze_pyramid_level0(rgb[..., 0])
    assert (score, rect.topleft, rect.size) == (es, (ex, ey), (64, 64)) and window.shape == (64, 64, 3)
    assert d.analyze_pyramid(np.zeros((120, 160, 3), np.uint8))[0] == 0


def test_im_helpers_on_gpu(mav, golden):
    from mavflow import im_helpers
    r = im_helpers.get_simple_bounding_box(golden["bbox_a_in"])
    assert [r.topleft[0], r.topleft[1], r.size[0], r.size[1]] == list(golden["bbox_a"])
    r = im_helpers.get_simple_bounding_box(np.zeros((120, 160), np.uint8))
    assert [r.topleft[0], r.topleft[1], r.size[0], r.size[1]] == list(golden["bbox_empty"])
    got = im_helpers.calculate_tpr_fpr(golden["tpr_gt"], 255 * golden["thr_nosky_fixed"])
    assert np.array(got, np.float64).tobytes() == golden["tpr_out"].tobytes()
    mag = im_helpers.get_magnitude(golden["foe_flow"][1])
    assert mag.tobytes() == golden["mag_out"].tobytes()


class FakeCapture:
    def __init__(self, frames):
        self.frames, self.i = frames, 0

    def read(self):
        f = self.frames[min(self.i, len(self.frames) - 1)]
        self.i += 1
        return True, np.repeat(f[..., None], 3, axis=2)


def test_farneback_class(mav, fb_oracle):
    from mavflow.farneback import Farneback
    f0, f1, _ = synth.make_pair(320, 240, 5)
    black = np.zeros_like(f0)
    fb = Farneback(FakeCapture([f0, f1, black, black]), None)
    vis = fb.process()
    assert vis.shape == (240, 320, 3) and vis.dtype == np.uint8
    ref = fb_oracle.calc(f0, f1)
    check_flow(fb.flow, ref, "Farneback.process")
    fb.process()                              # f1 -> black: some flow, new visualisation
    vis_prev = fb.prev_result
    vis2 = fb.process()                       # black -> black: constant (zero) magnitude = "invalid frame": previous result is kept
    assert np.all(fb.flow == 0) and np.array_equal(vis2, vis_prev)


def test_processor_loops_agree_with_the_oracle(mav):
    from mavflow.processor import Processor, SyntheticDataset
    from mavflow.run_config import RunConfig
    W, H, N = 320, 240, 4
    def make():
        ds = SyntheticDataset(W, H, N, use_farneback=True, dangle=(0.004, -0.002, 0.001))
        cfg = RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING")
        np.random.seed(11)
        return Processor(cfg), ds
    p1, ds1 = make()
    res = p1.run_detection()                   # flow seam -> one mav_detect call per frame
    p2, ds2 = make()
    res_b = p2.run_detection_batched(batch=2)  # frames -> mav_process_batch, two pairs per call
    p3, ds3 = make()
    res_s = p3.run_detection_staged()          # the reference-named calls one by one
    assert sorted(res) == sorted(res_b) == sorted(res_s) == [0, 1, 2]
    for i in res:                              # frame 0 (float32 path) and the later frames agree across all three loops
        assert vars(res[i]) == vars(res_s[i]) == vars(res_b[i]), i
        assert np.array_equal(p1.estimate_fixed, p3.estimate_fixed) and np.array_equal(p1.total_mask, p3.total_mask)
    # the oracle, fed the same flow and the same random draws
    np.random.seed(11)
    make()                                     # constructors consume their draws again
    for i in range(N - 1):
        smp = np.zeros((2000, 2), np.uint32)
        smp[:, 0] = np.random.randint(0, H, 2000); smp[:, 1] = np.random.randint(0, W, 2000)
        flow = ds1.get_flow_uv(i)
        ref = fo.run_chain(flow, smp, ds1.dangle / ds1.dt, ds1.dt, None, current_frame_index=i)
        for r in (res[i], res_b[i]):
            assert r.foe_dense == ref["foe"], (i, r.foe_dense, ref["foe"])
        seg = ds1.get_segmentation(i)[..., 0]
        with np.errstate(all="ignore"):
            exp = fo.calculate_tpr_fpr(seg, 255 * ref["fixed"])
        assert (res[i].tpr_fixed, res[i].fpr_fixed) == tuple(exp) == (res_b[i].tpr_fixed, res_b[i].fpr_fixed)
        assert res[i].drone_size_pixels == 24 * 24 and res[i].time == i * ds1.dt
        for p_ in (p1, p2):
            bx = p_.detection_boxes[i]
            assert tuple(int(v) for v in (bx.topleft + bx.size)) == \
                (ref["box"][0], ref["box"][1], ref["box"][2] - ref["box"][0], ref["box"][3] - ref["box"][1])


def test_float64_flow_keeps_its_precision_and_loop_attributes_exist(mav):
    """A dataset whose get_flow_uv returns float64 is evaluated in float64 from the start by the reference (derotate, get_FOE_dense and
    get_phi all promote): Context.detect() refuses to narrow it, Processor.run_detection() routes such frames through the float64
    kernels, and both loops then agree with the numpy chain on the float64 field.  The loop also exposes flow_uv_derotated / flow_mag
    (processor.py:306-307), derived on first access; and a dataset that refills ONE segmentation array per frame is followed."""
    from mavflow import _lib
    from mavflow.processor import Processor, SyntheticDataset
    from mavflow.run_config import RunConfig
    W, H, N = 320, 240, 4

    class F64Dataset(SyntheticDataset):
        constant_segmentation = False

        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.constant_segmentation = False
            self._one_seg = np.zeros((H, W, 3), np.uint8)

        def get_flow_uv(self, i):
            f = self.get_gt_of(i).astype(np.float64)
            return f + 1e-9 * np.arange(W)[None, :, None]           # not representable in float32

        def get_segmentation(self, i):                                # the SAME array object, refilled per frame
            self._one_seg[...] = 0
            self._one_seg[10 * (i + 1):10 * (i + 1) + 24, 40:64] = 255
            return self._one_seg

    def make():
        ds = F64Dataset(W, H, N, use_farneback=False, dangle=(0.004, -0.002, 0.001))
        cfg = RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING")
        np.random.seed(5)
        return Processor(cfg), ds
    p1, ds1 = make()
    res = p1.run_detection()
    p2, _ = make()
    res_s = p2.run_detection_staged()
    assert sorted(res) == sorted(res_s) == [0, 1, 2]
    np.random.seed(5)
    make()
    for i in range(N - 1):
        assert vars(res[i]) == vars(res_s[i]), i
        smp = np.zeros((2000, 2), np.uint32)
        smp[:, 0] = np.random.randint(0, H, 2000); smp[:, 1] = np.random.randint(0, W, 2000)
        ref = fo.run_chain(ds1.get_flow_uv(i), smp, ds1.dangle / ds1.dt, ds1.dt, None, current_frame_index=i)
        assert res[i].foe_dense == ref["foe"], i
        assert res[i].center_phi != res[(i + 1) % (N - 1)].center_phi           # the segmentation moved with the frame
    assert p1.flow_uv_derotated.dtype == np.float64 and p1.flow_mag.shape == (H, W)
    with _lib.Context(W, H, 1) as c:
        with pytest.raises(TypeError):
            c.detect(ds1.get_flow_uv(1), np.zeros((2000, 2), np.uint32))
    # the float32 loop derives the two attributes lazily
    ds = SyntheticDataset(W, H, 3, use_farneback=False, dangle=(0.004, -0.002, 0.001))
    p = Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
    p.run_detection()
    exp = p.detector.derotate(0, 1, ds.get_flow_uv(1))
    from mavflow import im_helpers
    assert np.array_equal(p.flow_uv_derotated, exp) and np.array_equal(p.flow_mag, im_helpers.get_magnitude(exp))


def test_pinned_pool_is_capped(mav):
    """Idle page-locked result blocks are capped in total (least recently used sizes go first), not kept for ever per distinct size."""
    from mavflow import _lib
    pool = _lib._PinnedPool()
    pool.CAP_BYTES = 3 << 20
    with _lib.Context(64, 64, 1) as c:
        for mb in (1, 2, 1, 2, 3, 1):
            a = pool.empty(c, (mb << 20,), np.uint8)
            a[0] = 1
            del a
            import gc
            gc.collect()
            assert pool.idle_bytes <= pool.CAP_BYTES, (mb, pool.idle_bytes)
        assert sum(len(v) * k for k, v in pool.free.items()) == pool.idle_bytes
        for lst in pool.free.values():                  # leave nothing page-locked behind
            while lst:
                c.lib.mav_host_free(None, lst.pop())


def test_farneback_flow_provider_fills_the_reference_flow_seam(mav, tmp_path, fb_oracle):
    """Dataset.get_flow_uv (/root/reference/src/datasets/dataset.py:205-212) answered by Farneback on the GPU; the .flo files
    it leaves behind are readable through the reference's own layout."""
    from mavflow.flow_provider import FarnebackFlowProvider, FloFlowProvider
    W, H = 320, 240
    frames = [synth.make_pair(W, H, 9)[0], synth.make_pair(W, H, 9)[1], synth.make_pair(W, H, 10)[1]]
    img = str(tmp_path / "seq" / "images")
    prov = FarnebackFlowProvider(lambda i: frames[i], W, H, img_path=img, write_flo=True)
    f0 = prov.get_flow_uv(0)
    f1 = prov.get_flow_uv(1)
    prov.release()
    assert f0.dtype == np.float32 and f0.shape == (H, W, 2)
    ref = fb_oracle.calc(frames[0], frames[1])
    check_flow(f0, ref, "FarnebackFlowProvider")
    files = FloFlowProvider(img)
    assert np.array_equal(files.get_flow_uv(0), f0) and np.array_equal(files.get_flow_uv(1), f1)
    # BGR frames are converted on the device (cv2.cvtColor at farneback.py:74): a gray replica gives the same flow
    prov = FarnebackFlowProvider(lambda i: np.repeat(frames[i][..., None], 3, axis=2), W, H)
    assert np.array_equal(prov.get_flow_uv(0), f0)
    prov.release()
    # window > 1: the frame-by-frame loop runs batched as frame sequences (every frame expanded once), same fields bit for bit
    video = synth.make_sequence(W, H, 8)
    one = FarnebackFlowProvider(lambda i: video[i], W, H)
    calls = []
    win = FarnebackFlowProvider(lambda i: (calls.append(i), video[i])[1], W, H, window=3, n_frames=len(video))
    for i in range(len(video) - 1):
        assert np.array_equal(win.get_flow_uv(i), one.get_flow_uv(i)), i
    assert calls == [0, 1, 2, 3, 3, 4, 5, 6, 6, 7]               # windows of 3, 3 and 1 pairs: 4 + 4 + 2 frames fetched
    assert np.array_equal(win.get_flow_uv(5), one.get_flow_uv(5))   # a random access recomputes its window
    one.release(); win.release()
    with pytest.raises(ValueError):
        FarnebackFlowProvider(lambda i: video[i], W, H, window=4)


def test_pyramid_generator_yields_every_level(mav):
    """im_helpers.pyramid (/root/reference/src/im_helpers.py:12-35) + sliding_window, iterated as Detector.analyze_pyramid does
    (detector.py:296-310), must reproduce analyze_pyramid's own answer -- same levels, same images."""
    from mavflow import im_helpers
    from mavflow.detector import Detector
    from oracle import pyramid_oracle as po
    W, H = 400, 300
    rng = np.random.default_rng(3)
    gray = (rng.integers(0, 30, (H, W)) * (rng.random((H, W)) > 0.9)).astype(np.uint8)
    gray[200:260, 120:200] = 90                                    # a blob that wins at a coarser level
    rgb = np.repeat(gray[..., None], 3, axis=2)
    levels = list(im_helpers.pyramid(rgb, scale=1.5))
    dims = im_helpers._ctx(W, H).pyramid_dims(1.5)
    assert [l.shape[:2] for l in levels] == [(h, w) for (w, h) in dims] and len(levels) >= 4
    exp = list(po.pyramid(gray, 1.5))
    for l, lv in enumerate(levels[1:], 1):                         # levels >= 1 equal the oracle's INTER_AREA cascade
        assert np.array_equal(lv[..., 0], exp[l]), l
    best = (0, None, None, None)
    for lv in levels:
        for (x, y, window) in im_helpers.sliding_window(lv, stepSize=16, windowSize=(64, 64)):
            if window.shape[0] != 64 or window.shape[1] != 64:
                continue
            s = int(np.sum(window))
            if best[0] < s:
                best = (s, x, y, window)
    d = Detector(type("DS", (), {"capture_size": (W, H)})())
    score, rect, window, _ = d.analyze_pyramid(rgb)
    assert (score, rect.topleft) == (best[0], (best[1], best[2])) and np.array_equal(window, best[3])
