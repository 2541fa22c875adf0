"""The reference-shaped loops on the device-resident plumbing (mavflow/pipeline.py, mavflow/processor.py) and the entry points that
plumbing added to the C-ABI: mav_upload_gather, mav_download_async / mav_marker_*, mav_tpr_fpr_counts_dev, mav_bgr2gray_dev.

VERDICT r04 #1: the three loops (run_detection, run_detection_staged, run_detection_batched) must produce identical FrameResults;
nothing that is not read may cross PCIe, and what IS read later must still be right (DeviceArray handles survive buffer re-use)."""
import ctypes as C
import json
import logging
import os

import numpy as np
import pytest

from oracle import foe_oracle as fo
from oracle import gray_oracle
from oracle.tolerances import check_flow
from mavflow import synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _counts(gt, img):
    """positives, negatives, true / false positives of src/im_helpers.py:244-252, products in wide integers."""
    g = np.asarray(gt).astype(np.int64)
    m = np.asarray(img).astype(np.int64)
    return (int(np.sum(g > 127)), int(np.sum((255 - g) > 127)), int(np.sum(g * m > 127)), int(np.sum((255 - g) * m > 127)))


def _processor(ds):
    from mavflow.processor import Processor
    from mavflow.run_config import RunConfig
    return Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))


def test_three_loops_fill_identical_frame_results_and_json_files(mav, tmp_path):
    """N - 1 = 7 frames, batch 3 (a ragged last batch), rotation on, a per-frame segmentation and sky (nothing declared constant) in
    one dataset and the constant form in the other: FrameResults and the JSON files of all three loops are the same, frame by frame."""
    from mavflow.processor import SyntheticDataset
    W, H, N = 320, 240, 8

    class PerFrame(SyntheticDataset):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.constant_segmentation = False
            self.constant_sky_segmentation = False

        def get_segmentation(self, i):
            seg = np.zeros((H, W, 3), np.uint8)
            seg[H // 4 + i:H // 4 + 24 + i, W // 4:W // 4 + 24] = 255
            return seg

        def get_sky_segmentation(self, i):
            sky = np.zeros((H, W), bool)
            sky[:10 + 3 * i] = True
            return sky

    for cls in (SyntheticDataset, PerFrame):
        runs = {}
        for loop in ("run_detection", "run_detection_staged", "run_detection_batched"):
            out_dir = tmp_path / f"{cls.__name__}_{loop}"
            ds = cls(W, H, N, use_farneback=True, dangle=(0.004, -0.002, 0.001), results_path=str(out_dir))
            np.random.seed(23)
            p = _processor(ds)
            res = p.run_detection_batched(batch=3) if loop == "run_detection_batched" else getattr(p, loop)()
            masks = (np.array(p.estimate_fixed), np.array(p.total_mask))
            runs[loop] = (res, masks, out_dir, dict(p.detection_boxes))
            p.release()
        base = runs["run_detection_staged"]
        assert sorted(base[0]) == list(range(N - 1))
        for loop in ("run_detection", "run_detection_batched"):
            res, masks, out_dir, boxes = runs[loop]
            for i in range(N - 1):
                assert vars(res[i]) == vars(base[0][i]), (cls.__name__, loop, i)
                a, b = (out_dir / f"image_{i:05d}.json").read_text(), (base[2] / f"image_{i:05d}.json").read_text()
                assert a == b and json.loads(a)["foe_dense"] == list(res[i].foe_dense)
            assert np.array_equal(masks[0], base[1][0]) and np.array_equal(masks[1], base[1][1]), (cls.__name__, loop)
            assert sorted(boxes) == list(range(N - 1))
        # the boxes the fast loops record are get_simple_bounding_box of the fixed mask (checked on the last frame, whose mask we hold)
        bx = runs["run_detection"][3][N - 2]
        assert tuple(int(v) for v in bx.topleft + bx.size) == tuple(
            v for v in (lambda b: (b[0], b[1], b[2] - b[0], b[3] - b[1]))(fo.simple_bounding_box(base[1][0])))


def test_fast_loop_leaves_flow_and_masks_on_the_device_until_read(mav):
    from mavflow.pipeline import DeviceArray
    from mavflow.processor import SyntheticDataset
    W, H, N = 320, 240, 4
    ds = SyntheticDataset(W, H, N, use_farneback=True)
    np.random.seed(3)
    p = _processor(ds)
    p.run_detection()
    for h in (p.flow_uv, p.estimate_fixed, p.total_mask):
        assert isinstance(h, DeviceArray) and h.on_device
    assert p.flow_uv.shape == (H, W, 2) and p.flow_uv.dtype == np.float32 and p.estimate_fixed.dtype == np.bool_
    # reading works like reading an array: ufuncs, functions, indexing, methods
    fixed = np.asarray(p.estimate_fixed)
    assert not p.estimate_fixed.on_device and fixed.dtype == np.bool_ and fixed.shape == (H, W)
    assert int(np.sum(p.total_mask)) == int(np.count_nonzero(np.asarray(p.total_mask)))
    assert (255 * p.estimate_fixed).dtype == np.int64 and (255 * p.estimate_fixed).max() in (0, 255)
    assert p.flow_uv[3, 4].shape == (2,) and p.flow_uv.astype(np.float64).dtype == np.float64
    assert np.linalg.norm(p.flow_uv, axis=-1).shape == (H, W)
    # the derived attributes of the reference's loop still work from a device-resident flow
    assert p.flow_uv_derotated.shape == (H, W, 2) and p.flow_mag.shape == (H, W)
    p.release()


def test_device_array_handles_survive_the_reuse_of_their_buffers(mav):
    """A handle somebody keeps is brought to the host by its owner right before the owner overwrites the memory."""
    from mavflow import _lib
    from mavflow.pipeline import DetectPipeline, FlowStage
    W, H = 256, 192
    prev, nxt = synth.make_batch(W, H, 4, distinct=4)
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(4)])
    with _lib.Context(W, H, 1) as ctx:
        ref = [ctx.process_batch(prev[b:b + 1], nxt[b:b + 1], smp[b:b + 1]) for b in range(4)]
        stage = FlowStage(ctx)
        kept = [stage.flow_of(prev[b], nxt[b]) for b in range(4)]         # two buffers, four flows: 0 and 1 must have been retired
        assert [h.on_device for h in kept] == [False, False, True, True]
        for b in range(4):
            assert np.array_equal(np.asarray(kept[b]), ref[b]["flow"][0]), b
        pipe = DetectPipeline(ctx, 1, slots=2, keep_flow=True)
        outs = []
        for b in range(4):
            outs.append(pipe.collect(pipe.submit(smp[b], prev=[prev[b]], nxt=[nxt[b]])))
        for b in range(4):
            assert outs[b]["results"].tobytes() == ref[b]["results"].tobytes(), b
            assert np.array_equal(outs[b]["mask_fixed"][0], ref[b]["mask_fixed"][0]) and np.array_equal(outs[b]["mask_dyn"][0], ref[b]["mask_dyn"][0]), b
            assert np.array_equal(outs[b]["flow"][0], ref[b]["flow"][0]), b
        with pytest.raises(ValueError):
            pipe.collect(0)                                               # already collected
        # a refused submit (wrong sample count, wrong frame size, no input at all) takes no slot and leaves the pipeline usable
        turn = pipe._turn
        for bad in (dict(samples=smp[0][:10], prev=[prev[0]], nxt=[nxt[0]]), dict(samples=smp[0], prev=[prev[0][:-1]], nxt=[nxt[0][:-1]]),
                    dict(samples=smp[0]), dict(samples=smp[0], prev=[prev[0]], nxt=[nxt[0]], omega=np.zeros(5))):
            with pytest.raises(ValueError):
                pipe.submit(**bad)
        assert pipe._turn == turn
        again = pipe.collect(pipe.submit(smp[2], prev=[prev[2]], nxt=[nxt[2]]))
        assert again["results"].tobytes() == ref[2]["results"].tobytes()
        pipe.close()
        stage.close()


def test_pipeline_batches_in_flight_video_layout_and_counts(mav):
    """Two batches in flight through two slots; a video (pair k = frames k, k + 1 as the SAME array objects) takes the frame-sequence
    layout; counts against a shared and a per-pair ground truth equal calculate_tpr_fpr's."""
    from mavflow import _lib
    from mavflow.pipeline import DetectPipeline
    W, H, B = 320, 240, 4
    seq = synth.make_sequence(W, H, 2 * B + 1)
    frames = [np.ascontiguousarray(seq[k]) for k in range(2 * B + 1)]
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(2 * B)])
    rng = np.random.default_rng(2)
    gts = [(rng.integers(0, 2, (H, W)) * 255).astype(np.uint8) for _ in range(2 * B)]
    gts[1][::3] = 100                                                    # values that are neither 0 nor 255
    omega = np.tile([0.01, -0.02, 0.005], (2 * B, 1)) / 0.04
    with _lib.Context(W, H, B) as ctx:
        pipe = DetectPipeline(ctx, B)
        t0 = pipe.submit(smp[:B], prev=frames[:B], nxt=frames[1:B + 1], omega=omega[:B], dt=np.full(B, 0.04), frame0=[True] + [False] * (B - 1), gt=gts[:B])
        t1 = pipe.submit(smp[B:], prev=frames[B:2 * B], nxt=frames[B + 1:], omega=omega[B:], dt=np.full(B, 0.04), frame0=[False] * B, gt_shared=gts[0])
        o0, o1 = pipe.collect(t0), pipe.collect(t1)
        pipe.close()
    with _lib.Context(W, H, B) as ctx:
        r0 = ctx.process_batch(seq[:B].copy(), seq[1:B + 1].copy(), smp[:B], omega=omega[:B], dt=np.full(B, 0.04), frame0=[True] + [False] * (B - 1))
        r1 = ctx.process_batch(seq[B:2 * B].copy(), seq[B + 1:].copy(), smp[B:], omega=omega[B:], dt=np.full(B, 0.04), frame0=[False] * B)
        for o, r, gt_of in ((o0, r0, lambda k: gts[k]), (o1, r1, lambda k: gts[0])):
            assert o["results"].tobytes() == r["results"].tobytes()
            for k in range(B):
                assert np.array_equal(o["mask_fixed"][k], r["mask_fixed"][k]) and np.array_equal(o["mask_dyn"][k], r["mask_dyn"][k]), k
                assert tuple(o["counts_fixed"][k]) == tuple(ctx.tpr_fpr_counts(gt_of(k), r["mask_fixed"][k:k + 1].view(np.uint8), 255)[0]), k
                assert tuple(o["counts_dyn"][k]) == tuple(_counts(gt_of(k), 255 * r["mask_dyn"][k].astype(np.int64))), k


def test_upload_gather_pageable_pinned_replicated_and_large(mav):
    """mav_upload_gather: sources larger than a staging chunk, page-locked and pageable sources mixed, one source repeated."""
    from mavflow import _lib
    from mavflow.pipeline import _ptr_array
    rng = np.random.default_rng(8)
    with _lib.Context(64, 64, 1) as ctx:
        ctx.set_option("upload_threads", 3)
        for nbytes, count in (((20 << 20) + 12345, 3), (4096, 40), (1, 5), ((16 << 20), 2)):
            srcs = [rng.integers(0, 256, nbytes, dtype=np.uint8) for _ in range(count)]
            srcs[1] = ctx.pinned_like(srcs[1])                              # a page-locked source goes straight from where it is
            srcs[-1] = srcs[0]                                              # the same array twice
            buf = ctx.alloc(nbytes * count)
            for ordered in (0, 1):
                _lib.check(ctx.lib.mav_upload_gather(ctx.h, buf.ptr, _ptr_array(srcs), count, nbytes, ordered))
                ctx.upload_fence()
                got = buf.download(np.uint8, (count, nbytes))
                for k in range(count):
                    assert np.array_equal(got[k], srcs[k]), (nbytes, k, ordered)
            buf.free()
        with pytest.raises(ValueError):
            ctx.lib.mav_upload_gather.restype = C.c_int
            _lib.check(ctx.lib.mav_upload_gather(ctx.h, None, _ptr_array([np.zeros(4, np.uint8)]), 1, 4, 0))
        with pytest.raises(_lib.MavflowError):
            ctx.set_option("upload_threads", 2)                             # the staging threads exist already
        dev = ctx.alloc(64)
        with pytest.raises(ValueError):                                     # a device pointer among the sources is refused, not read as host memory
            _lib.check(ctx.lib.mav_upload_gather(ctx.h, ctx.alloc(64).ptr, (C.c_void_p * 1)(dev.ptr), 1, 64, 0))


@pytest.mark.parametrize("W,H", [(64, 48), (37, 29)])
def test_tpr_fpr_counts_dev_matches_the_host_counts(mav, W, H):
    """16-byte-addressable images take the vector path, 37 x 29 the byte path; one mask or two; shared or per-pair ground truth."""
    from mavflow import _lib
    rng = np.random.default_rng(4)
    B = 3
    gt = rng.integers(0, 256, (B, H, W), dtype=np.uint8)
    gt[0][gt[0] < 128] = 0
    mf = (rng.random((B, H, W)) < 0.3).astype(np.uint8)
    md = (rng.random((B, H, W)) < 0.6).astype(np.uint8) * 7                 # any nonzero byte counts as set
    with _lib.Context(W, H, B) as ctx:
        dg, dmf, dmd = ctx.alloc(gt.nbytes).upload(gt), ctx.alloc(mf.nbytes).upload(mf), ctx.alloc(md.nbytes).upload(md)
        dcf, dcd = ctx.alloc(B * 32), ctx.alloc(B * 32)
        for value in (255, 1):
            for images in (B, 1):
                _lib.check(ctx.lib.mav_tpr_fpr_counts_dev(ctx.h, dg.ptr, images, dmf.ptr, dmd.ptr, value, B, dcf.ptr, dcd.ptr))
                ctx.sync()
                cf, cd = dcf.download(np.int64, (B, 4)), dcd.download(np.int64, (B, 4))
                for b in range(B):
                    g = gt[b if images == B else 0]
                    assert tuple(cf[b]) == tuple(_counts(g, value * (mf[b] != 0).astype(np.int64))), (value, images, b)
                    assert tuple(cd[b]) == tuple(_counts(g, value * (md[b] != 0).astype(np.int64))), (value, images, b)
            _lib.check(ctx.lib.mav_tpr_fpr_counts_dev(ctx.h, dg.ptr, B, None, dmd.ptr, value, B, None, dcd.ptr))    # the dynamic mask alone
            ctx.sync()
            cd = dcd.download(np.int64, (B, 4))
            assert tuple(cd[1]) == tuple(_counts(gt[1], value * (md[1] != 0).astype(np.int64)))
        with pytest.raises(ValueError):
            _lib.check(ctx.lib.mav_tpr_fpr_counts_dev(ctx.h, dg.ptr, 2, dmf.ptr, None, 255, B, dcf.ptr, None))        # gt_images neither 1 nor batch


def test_png_sequence_through_farneback_and_the_flow_provider(mav, fb_oracle, tmp_path):
    """image_%05d.png files -> PngSequenceCapture -> Farneback (the reference's class) and FarnebackFlowProvider.from_png_sequence, host
    and device-resident: BGR -> gray on the GPU equals the fixed-point oracle, the flow equals the CPU Farneback on those gray frames."""
    from mavflow.farneback import Farneback
    from mavflow.flow_provider import FarnebackFlowProvider
    from mavflow.pipeline import DeviceArray, FlowStage
    z = np.load(os.path.join(GOLDEN, "png_frames.npz"), allow_pickle=False)
    for k in range(3):
        (tmp_path / f"image_{k:05d}.png").write_bytes(z[f"png_seq{k}"].tobytes())
    gray = [gray_oracle.bgr_to_gray(z[f"bgr_seq{k}"]) for k in range(3)]
    ref = [fb_oracle.calc(gray[k], gray[k + 1]) for k in range(2)]
    fb = Farneback.from_png_sequence(str(tmp_path))
    assert np.array_equal(fb.prevgray, gray[0])
    fb.process()
    check_flow(fb.flow, ref[0], "Farneback.from_png_sequence, pair 0")
    fb.process()
    check_flow(fb.flow, ref[1], "pair 1")
    host = FarnebackFlowProvider.from_png_sequence(str(tmp_path))
    dev = FarnebackFlowProvider.from_png_sequence(str(tmp_path), on_device=True)
    for k in range(2):
        a, b = host.get_flow_uv(k), dev.get_flow_uv(k)
        assert isinstance(b, DeviceArray) and b.on_device and not isinstance(a, DeviceArray)
        assert np.array_equal(a, np.asarray(b)) and np.array_equal(a, fb.flow if k == 1 else a)
        check_flow(a, ref[k], f"provider pair {k}")
    with pytest.raises(OSError):
        host.get_flow_uv(2)                                                 # frame 3 does not exist
    # video mode: one frame per step, the previous gray frame stays on the device (the class's prevgray)
    stage = FlowStage(dev.ctx)
    assert stage.flow_next(z["bgr_seq0"]) is None
    f01 = stage.flow_next(z["bgr_seq1"])
    f12 = stage.flow_next(gray[2])                                          # a gray frame is taken as it is
    assert np.array_equal(np.asarray(f01), host.get_flow_uv(0)) and np.array_equal(np.asarray(f12), host.get_flow_uv(1))
    stage.close()
    host.release(); dev.release()


@pytest.mark.parametrize("W,H,B", [(322, 243, 3), (250, 190, 2), (641, 359, 4)])
def test_pipeline_equals_the_host_pointer_call_at_odd_sizes(mav, W, H, B):
    """Frame sizes that are no multiple of 4 / 16 (unaligned per-pair offsets, the byte path of the count kernel, the general blur): the
    pipeline's records, masks, counts and flow equal mav_process_batch's, with per-pair sky masks and ground truths, rotation and a
    frame-0 pair, over more submits than there are slots."""
    from mavflow import _lib
    from mavflow.pipeline import DetectPipeline
    rng = np.random.default_rng(W)
    prev, nxt = synth.make_batch(W, H, B, distinct=B)
    smp = np.zeros((B, 2000, 2), np.uint32)
    smp[..., 0] = rng.integers(0, H, (B, 2000)); smp[..., 1] = rng.integers(0, W, (B, 2000))
    sky = rng.random((B, H, W)) < 0.05
    gt = (rng.integers(0, 2, (B, H, W)) * 255).astype(np.uint8)
    omega = rng.normal(0, 0.2, (B, 3))
    dt = np.full(B, 1 / 30.0)
    f0 = [True] + [False] * (B - 1)
    with _lib.Context(W, H, B) as ctx:
        ref = ctx.process_batch(prev, nxt, smp, omega=omega, dt=dt, sky=sky, frame0=f0)
        pipe = DetectPipeline(ctx, B, keep_flow=True)
        outs = [pipe.collect(pipe.submit(smp, prev=list(prev), nxt=list(nxt), omega=omega, dt=dt, frame0=f0, sky=list(sky), gt=list(gt)))
                for _ in range(4)]
        part = pipe.collect(pipe.submit(smp[:1], prev=[prev[0]], nxt=[nxt[0]], omega=omega[:1], dt=dt[:1], frame0=f0[:1], sky=[sky[0]],
                                        gt_shared=gt[0]))                  # fewer pairs than the pipeline's batch, shared ground truth
        for o in outs:
            assert o["results"].tobytes() == ref["results"].tobytes()
            for k in range(B):
                assert np.array_equal(o["mask_fixed"][k], ref["mask_fixed"][k]) and np.array_equal(o["mask_dyn"][k], ref["mask_dyn"][k]), k
                assert np.array_equal(o["flow"][k], ref["flow"][k]), k
                assert tuple(o["counts_fixed"][k]) == _counts(gt[k], 255 * ref["mask_fixed"][k].astype(np.int64)), k
                assert tuple(o["counts_dyn"][k]) == _counts(gt[k], 255 * ref["mask_dyn"][k].astype(np.int64)), k
        assert part["results"].tobytes() == ref["results"][:1].tobytes() and len(part["mask_fixed"]) == 1
        assert tuple(part["counts_fixed"][0]) == _counts(gt[0], 255 * ref["mask_fixed"][0].astype(np.int64))
        pipe.close()


def test_a_video_dataset_takes_the_frame_sequence_layout_and_the_loops_still_agree(mav, fb_oracle):
    """SyntheticDataset(video=True): pair i = (frame i, frame i + 1) as the same array objects.  The batched loop hands such batches to
    the library as ONE run of n + 1 frames (every inner frame uploaded and expanded once); the flow is the pair-by-pair flow bit for bit,
    so all three loops fill identical FrameResults, and the flow matches the CPU Farneback and the analytic zoom."""
    from mavflow import pipeline
    from mavflow.processor import SyntheticDataset
    W, H, N = 320, 240, 9
    runs = {}
    seen = []
    orig = pipeline.DetectPipeline.submit

    def spy(self, samples, prev=None, nxt=None, **kw):
        if prev is not None:
            seen.append(len(prev) > 1 and all(nxt[k] is prev[k + 1] for k in range(len(prev) - 1)))
        return orig(self, samples, prev=prev, nxt=nxt, **kw)
    pipeline.DetectPipeline.submit = spy
    try:
        for loop in ("run_detection", "run_detection_staged", "run_detection_batched"):
            ds = SyntheticDataset(W, H, N, use_farneback=True, video=True, dangle=(0.002, 0.001, -0.001))
            np.random.seed(5)
            p = _processor(ds)
            runs[loop] = p.run_detection_batched(batch=4) if loop == "run_detection_batched" else getattr(p, loop)()
            if loop == "run_detection":
                flow3 = np.array(ds.get_flow_uv(3))
                f3, f4, truth = ds._pair(3)
            p.release()
    finally:
        pipeline.DetectPipeline.submit = orig
    assert seen == [True, True]                                   # both batches of 4 went in as frame sequences
    for i in range(N - 1):
        assert vars(runs["run_detection"][i]) == vars(runs["run_detection_staged"][i]) == vars(runs["run_detection_batched"][i]), i
    check_flow(flow3, fb_oracle.calc(f3, f4), "video pair 3")
    inner = (slice(20, H - 20), slice(20, W - 20))
    assert np.abs(flow3[inner] - truth[inner]).mean() < 0.05      # the analytic zoom field (a sanity bound, not a parity bound)


def test_lanes_give_the_single_context_results(mav):
    """A stream of one-pair calls spread over three contexts taken in turn (pipeline.LanedFlowStage / LanedPipeline, what the one-frame
    loop uses up to 1080p): flow, records, masks and counts of every pair equal the single-context results; a DeviceArray flow is
    followed to the lane that holds it; the tickets collect in submission order with `depth` batches outstanding."""
    from collections import deque
    from mavflow import _lib
    from mavflow.pipeline import LanedFlowStage, LanedPipeline, auto_lanes
    assert (auto_lanes(1280, 720), auto_lanes(1920, 1080), auto_lanes(3840, 2160), auto_lanes(1920, 1080, 64), auto_lanes(640, 480, 2)) == (4, 3, 1, 1, 4)
    assert (auto_lanes(1280, 720, uploads=False), auto_lanes(1920, 1080, uploads=False), auto_lanes(3840, 2160, uploads=False)) == (3, 2, 1)
    W, H, n = 320, 240, 7
    prev, nxt = synth.make_batch(W, H, n, distinct=n)
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(n)])
    gt = np.zeros((H, W), np.uint8)
    gt[60:84, 80:104] = 255
    with _lib.Context(W, H, 1) as c0:
        ref = [c0.process_batch(prev[b:b + 1], nxt[b:b + 1], smp[b:b + 1]) for b in range(n)]
    ctxs = [_lib.Context(W, H, 1) for _ in range(3)]
    stage, pipe = LanedFlowStage(ctxs), LanedPipeline(ctxs, 1)
    assert stage.lanes == 3 and pipe.depth == 3
    pending, outs, flows = deque(), [], []
    for b in range(n):
        f = stage.flow_of(prev[b], nxt[b])
        assert f.ctx is ctxs[b % 3]
        flows.append(f)
        pending.append(pipe.submit(smp[b], flow=f, gt_shared=gt))
        assert pending[-1][0] == b % 3                                  # the ticket's lane is the one that holds the flow
        while len(pending) > pipe.depth:
            outs.append(pipe.collect(pending.popleft()))
    while pending:
        outs.append(pipe.collect(pending.popleft()))
    for b in range(n):
        assert outs[b]["results"].tobytes() == ref[b]["results"].tobytes(), b
        assert np.array_equal(outs[b]["mask_fixed"][0], ref[b]["mask_fixed"][0]) and np.array_equal(outs[b]["mask_dyn"][0], ref[b]["mask_dyn"][0]), b
        assert np.array_equal(np.asarray(flows[b]), ref[b]["flow"][0]), b
        assert tuple(outs[b]["counts_fixed"][0]) == _counts(gt, 255 * ref[b]["mask_fixed"][0].astype(np.int64)), b
    # frames (no flow handle) take the lanes in turn
    t = [pipe.submit(smp[b], prev=[prev[b]], nxt=[nxt[b]]) for b in range(3)]
    assert [x[0] for x in t] == [0, 1, 2]
    for b in range(3):
        assert pipe.collect(t[b])["results"].tobytes() == ref[b]["results"].tobytes()
    other = _lib.Context(W, H, 1)
    with pytest.raises(ValueError):
        from mavflow.pipeline import FlowStage
        st = FlowStage(other)
        pipe.submit(smp[0], flow=st.flow_of(prev[0], nxt[0]))               # a flow on a context the pipeline does not span
    st.close(); other.close()
    pipe.close(); stage.close()
    for c in ctxs:
        c.close()


def test_host_flow_seam_fast_loop_equals_the_staged_loop(mav):
    """Dataset.get_flow_uv returning a HOST float32 field (what a .flo file gives): the fast loop uploads it once per frame through the
    gather (one lane: the chain is PCIe-bound) and must fill the FrameResults the reference-named calls fill; the masks it leaves behind
    are the staged loop's."""
    from mavflow.processor import SyntheticDataset
    W, H, N = 320, 240, 6
    res = {}
    for loop in ("run_detection", "run_detection_staged"):
        ds = SyntheticDataset(W, H, N, use_farneback=False, dangle=(0.003, -0.001, 0.002))
        np.random.seed(17)
        p = _processor(ds)
        out = getattr(p, loop)()
        res[loop] = (out, np.array(p.estimate_fixed), np.array(p.total_mask))
        if loop == "run_detection":
            assert len(p._ctxs) == 1 and not isinstance(p.flow_uv, type(p.estimate_fixed))     # a host array in, one lane
        p.release()
    a, b = res["run_detection"], res["run_detection_staged"]
    assert sorted(a[0]) == list(range(N - 1))
    for i in range(N - 1):
        assert vars(a[0][i]) == vars(b[0][i]), i
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_loops_agree_over_random_shapes_lanes_and_batches(mav):
    """A small fuzz of the three loops: frame sizes that are no multiple of 4 / 16, 1 - 3 lanes, batch sizes that divide the run or not,
    with and without rotation, Farneback or host-flow seam.  FrameResults of the fast loops == the staged loop's, frame by frame."""
    from mavflow.processor import SyntheticDataset
    rng = np.random.default_rng(2025)
    for case in range(6):
        W, H = int(rng.integers(100, 200)) * (2 if case % 2 else 1), int(rng.integers(90, 160))
        N = int(rng.integers(4, 9))
        lanes = int(rng.integers(1, 4))
        batch = int(rng.integers(1, 5))
        use_fb = case % 3 != 2
        dangle = tuple(rng.normal(0, 0.003, 3)) if case % 2 else (0.0, 0.0, 0.0)
        runs = {}
        for loop in ("run_detection_staged", "run_detection") + (("run_detection_batched",) if use_fb else ()):
            ds = SyntheticDataset(W, H, N, use_farneback=use_fb, dangle=dangle, lanes=lanes, seed=case)
            np.random.seed(100 + case)
            p = _processor(ds)
            runs[loop] = p.run_detection_batched(batch=batch) if loop == "run_detection_batched" else getattr(p, loop)()
            p.release()
        base = runs.pop("run_detection_staged")
        assert sorted(base) == list(range(N - 1)), (case, W, H)
        for loop, res in runs.items():
            for i in range(N - 1):
                assert vars(res[i]) == vars(base[i]), (case, W, H, N, lanes, batch, loop, i)


def test_a_seam_that_hides_its_lanes_is_followed_anyway(mav):
    """get_flow_uv handing out DeviceArrays from two contexts in turn without showing the stage (any user-written seam): the loop
    collects the contexts it has seen, spans its pipeline over them, and fills the FrameResults of the staged loop."""
    from mavflow import _lib, pipeline
    from mavflow.processor import SyntheticDataset
    W, H, N = 256, 192, 7

    class Hidden(SyntheticDataset):
        def get_flow_uv(self, i):
            if not hasattr(self, "_mine"):
                self._mine_ctxs = [_lib.Context(W, H, 1) for _ in range(2)]
                self._mine = [pipeline.FlowStage(c) for c in self._mine_ctxs]
            f0, f1, _ = self._pair(i)
            return self._mine[i % 2].flow_of(f0, f1)

        def release(self):
            for st in getattr(self, "_mine", []):
                st.close()
            for c in getattr(self, "_mine_ctxs", []):
                c.close()

    res = {}
    for loop in ("run_detection", "run_detection_staged"):
        ds = Hidden(W, H, N, use_farneback=True, dangle=(0.001, 0.002, -0.001))
        np.random.seed(9)
        p = _processor(ds)
        res[loop] = getattr(p, loop)()
        if loop == "run_detection":
            assert len(p._lane_seen) == 2
            # the pipeline over the first lane alone went when the one over both lanes was built
            assert len(p._pipes) == 1 and len(next(iter(p._pipes.values())).ctxs) == 2
        p.release()
    for i in range(N - 1):
        assert vars(res["run_detection"][i]) == vars(res["run_detection_staged"][i]), i


# ---- round 6: one loop iteration as ONE call (mav_frame_step), the contexts' worker threads, deferred flow ------------------------------
def _same_outputs(a, b, tag):
    assert a["results"].tobytes() == b["results"].tobytes(), tag
    n = len(a["mask_fixed"])
    for k in range(n):
        assert np.array_equal(np.asarray(a["mask_fixed"][k]), np.asarray(b["mask_fixed"][k])), (tag, k)
        assert np.array_equal(np.asarray(a["mask_dyn"][k]), np.asarray(b["mask_dyn"][k])), (tag, k)
    for key in ("counts_fixed", "counts_dyn"):
        assert (a[key] is None) == (b[key] is None) and (a[key] is None or np.array_equal(a[key], b[key])), (tag, key)


def test_frame_step_posted_inline_and_plain_calls_agree(mav):
    """Every way a batch can reach the detection -- frames, a video run, a host flow field, a deferred Farneback flow (gray and BGR
    frames, pair mode and video mode) -- through the fused step posted to the worker thread, the same step enqueued by the calling
    thread, and the plain synchronous call: identical records, masks and counts."""
    from mavflow import _lib
    from mavflow.pipeline import DetectPipeline, FlowStage
    W, H, B = 256, 192, 3
    seq = synth.make_sequence(W, H, B + 2)
    frames = [np.ascontiguousarray(seq[k]) for k in range(B + 2)]
    prev, nxt = synth.make_batch(W, H, B, distinct=B)
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
    rng = np.random.default_rng(5)
    gts = [(rng.integers(0, 2, (H, W)) * 255).astype(np.uint8) for _ in range(B)]
    skies = [np.zeros((H, W), bool) for _ in range(B)]
    for k, sk in enumerate(skies):
        sk[:20 + 7 * k] = True
    omega, dts = np.tile([0.01, -0.02, 0.005], (B, 1)) / 0.04, np.full(B, 0.04)
    f0 = [True] + [False] * (B - 1)
    with _lib.Context(W, H, B) as ctx:
        ref = ctx.process_batch(prev, nxt, smp, omega=omega, dt=dts, frame0=f0, sky=np.stack(skies))
        ref_counts = [np.stack([ctx.tpr_fpr_counts(gts[k], ref[m][k:k + 1].view(np.uint8), 255)[0] for k in range(B)]) for m in ("mask_fixed", "mask_dyn")]
        ref_vid = ctx.process_batch(seq[:B].copy(), seq[1:B + 1].copy(), smp)
        flows = ctx.farneback(prev, nxt)
    outs = {}
    for worker in (True, False):
        with _lib.Context(W, H, B) as ctx:
            pipe = DetectPipeline(ctx, B, worker=worker)
            o = pipe.collect(pipe.submit(smp, prev=list(prev), nxt=list(nxt), omega=omega, dt=dts, frame0=f0, sky=skies, gt=gts))
            assert o["results"].tobytes() == ref["results"].tobytes() and np.array_equal(o["counts_fixed"], ref_counts[0]) and np.array_equal(o["counts_dyn"], ref_counts[1])
            for k in range(B):
                assert np.array_equal(o["mask_fixed"][k], ref["mask_fixed"][k]) and np.array_equal(o["mask_dyn"][k], ref["mask_dyn"][k])
            v = pipe.collect(pipe.submit(smp, prev=frames[:B], nxt=frames[1:B + 1]))                       # a video: one run of B + 1 frames
            assert v["results"].tobytes() == ref_vid["results"].tobytes()
            hf = pipe.collect(pipe.submit(smp, flow=[flows[b] for b in range(B)], omega=omega, dt=dts, frame0=f0, sky=skies, gt_shared=gts[0]))
            assert hf["results"].tobytes() == ref["results"].tobytes()
            outs[worker] = (o, v, hf)
            pipe.close()
    for a, b, tag in zip(outs[True], outs[False], ("frames", "video", "host flow")):
        _same_outputs(a, b, tag)
    # deferred flow: FlowStage hands out a handle, the pipeline of the same context takes upload + Farneback + detection as one step
    bgr = [np.repeat(f[..., None], 3, axis=2) for f in frames]                                                # gray values replicated: same luma
    for worker in (True, False):
        with _lib.Context(W, H, 1) as ctx:
            ref1 = [ctx.process_batch(seq[k:k + 1].copy(), seq[k + 1:k + 2].copy(), smp[k:k + 1]) for k in range(B)]
            stage, pipe = FlowStage(ctx), DetectPipeline(ctx, 1, worker=worker)
            for mode in ("pairs", "video", "bgr pairs", "bgr video"):
                src = bgr if mode.startswith("bgr") else frames
                if mode.endswith("video"):
                    stage._have_prev = False
                    assert stage.flow_next(src[0]) is None
                got = []
                for k in range(B):
                    h = stage.flow_of(src[k], src[k + 1]) if mode.endswith("pairs") else stage.flow_next(src[k + 1])
                    assert h._deferred is not None and h.on_device
                    t = pipe.submit(smp[k], flow=h)
                    assert h._deferred is None and stage._open is None
                    got.append((pipe.collect(t), h))
                for k in range(B):
                    assert got[k][0]["results"].tobytes() == ref1[k]["results"].tobytes(), (worker, mode, k)
                    assert np.array_equal(got[k][0]["mask_fixed"][0], ref1[k]["mask_fixed"][0]), (worker, mode, k)
                    assert np.array_equal(np.asarray(got[k][1]), ref1[k]["flow"][0]), (worker, mode, k)   # the handle reads the flow the step computed
            pipe.close(); stage.close()


def test_deferred_flow_is_computed_when_somebody_looks_first(mav):
    """A deferred handle that is read, kept across later flows, or dropped unread -- before any pipeline sees it."""
    from mavflow import _lib
    from mavflow.pipeline import DetectPipeline, FlowStage
    W, H = 256, 192
    seq = synth.make_sequence(W, H, 6)
    fr = [np.ascontiguousarray(f) for f in seq]
    smp = synth.foe_samples(W, H, 0)
    with _lib.Context(W, H, 1) as ctx:
        ref = [ctx.farneback(seq[k:k + 1].copy(), seq[k + 1:k + 2].copy())[0] for k in range(5)]
        stage = FlowStage(ctx)
        h0 = stage.flow_of(fr[0], fr[1])
        assert h0._deferred is not None
        assert np.array_equal(np.asarray(h0), ref[0]) and h0._deferred is None          # read: computed on the spot
        h1 = stage.flow_of(fr[1], fr[2])
        h2 = stage.flow_of(fr[2], fr[3])                                                # the next plan settles the open one (h1 is held: computed)
        assert h1._deferred is None and h2._deferred is not None
        h3 = stage.flow_of(fr[3], fr[4])                                                # h1's buffer is re-used: h1 goes to the host first
        assert not h1.on_device and np.array_equal(np.asarray(h1), ref[1]) and np.array_equal(np.asarray(h2), ref[2])
        del h3                                                                          # dropped unread: never computed
        stage.flow_next(fr[0])                                                          # video mode from here
        a = stage.flow_next(fr[1])
        del a                                                                           # dropped: its FRAME must still arrive (next pair's prev)
        b = stage.flow_next(fr[2])
        pipe = DetectPipeline(ctx, 1)
        out = pipe.collect(pipe.submit(smp, flow=b))
        assert np.array_equal(np.asarray(b), ref[1])
        with _lib.Context(W, H, 1) as c2:
            assert out["results"].tobytes() == c2.detect(ref[1][None], smp[None])["results"].tobytes()
        pipe.close(); stage.close()
        nodefer = FlowStage(ctx, defer=False)
        assert nodefer.flow_of(fr[0], fr[1])._deferred is None
        nodefer.close()


def test_posted_step_errors_surface_at_wait_and_drain(mav):
    from mavflow import _lib
    W, H = 64, 48
    with _lib.Context(W, H, 1) as ctx:
        bad = _lib.FrameStep()
        bad.n = 5                                              # beyond max_batch
        t = ctx.post_step(bad)
        with pytest.raises(ValueError, match="outside"):
            ctx.wait_step(t)
        ok = _lib.FrameStep()
        ok.n = 1                                               # nothing to do: a legal empty step
        t2 = ctx.post_step(ok)
        ctx.wait_step(t2)
        t3 = ctx.post_step(bad)
        with pytest.raises(ValueError):
            ctx.sync()                                         # any other call of the binding drains the worker first and reports
        ctx.sync()
        with pytest.raises(ValueError):
            ctx.wait_step(t3)                                  # ... and the ticket still tells
        with pytest.raises(ValueError):
            ctx.wait_step(999)
        with pytest.raises(ValueError):
            _lib.check(ctx.lib.mav_frame_step_dev(ctx.h, C.byref(bad)))
    # a context destroyed with steps still queued: the worker finishes the one in hand, drops the rest, and the process goes on
    ctx = _lib.Context(W, H, 1)
    for _ in range(200):
        ctx.post_step(ok)
    ctx.close()
    with _lib.Context(W, H, 1) as again:
        again.wait_step(again.post_step(ok))


def test_upload_gather_reads_page_locked_sources_before_it_returns(mav):
    """the header promises that on return every source has been read.  A page-locked source is overwritten right after
    the call while the compute stream is still busy -- with the copy stream and with inline uploads (where the copy would sit behind
    the compute stream's kernels) -- and the device must hold the ORIGINAL bytes.  With MAV_GATHER_SOURCES_HELD the caller keeps them."""
    from mavflow import _lib
    from mavflow.pipeline import _ptr_array
    W, H, B = 640, 480, 8
    prev, nxt = synth.make_batch(W, H, B, distinct=2)
    rng = np.random.default_rng(3)
    n = 4 << 20
    for inline in (0, 1):
        with _lib.Context(W, H, B) as ctx:
            ctx.set_option("inline_uploads", inline)
            dp, dn, df = ctx.alloc(prev.nbytes).upload(prev), ctx.alloc(nxt.nbytes).upload(nxt), ctx.alloc(8 * W * H * B)
            src = ctx.pinned_like(rng.integers(0, 256, n, dtype=np.uint8))
            page = rng.integers(0, 256, n, dtype=np.uint8)
            want = np.stack([src.copy(), page.copy()])
            dst = ctx.alloc(2 * n)
            for flags in (0, _lib.GATHER_ORDERED):
                src[:] = want[0]; page[:] = want[1]
                ctx.sync()
                for _ in range(3):
                    ctx.farneback_dev(dp.ptr, dn.ptr, B, df.ptr)                         # ~10 ms of kernels ahead on the compute stream
                _lib.check(ctx.lib.mav_upload_gather(ctx.h, dst.ptr, _ptr_array([src, page]), 2, n, flags))
                src[:] = 0xAA; page[:] = 0x55                                            # the caller re-uses its buffers at once
                ctx.upload_fence()
                got = dst.download(np.uint8, (2, n))
                assert np.array_equal(got, want), (inline, flags)
            src[:] = want[0]
            _lib.check(ctx.lib.mav_upload_gather(ctx.h, dst.ptr, _ptr_array([src]), 1, n, _lib.GATHER_SOURCES_HELD))
            ctx.upload_fence(); ctx.sync()                                               # held until the work behind it has completed
            assert np.array_equal(dst.download(np.uint8, (n,)), want[0])


def test_retired_handles_dropped_at_once_do_not_leak_their_blocks_to_another_lane(mav):
    """a retired handle's page-locked block is the target of a copy that is only ENQUEUED.  Drop the handle at once and
    let another lane materialise same-size handles: nobody may be handed that block before the copy has landed.  Two lanes, many
    rounds; every materialised mask is compared with a synchronous download of the same device memory."""
    from mavflow import _lib
    from mavflow.pipeline import DetectPipeline, _retire_all
    W, H = 640, 480
    prev, nxt = synth.make_batch(W, H, 2, distinct=2)
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(2)])
    lanes = [_lib.Context(W, H, 1) for _ in range(2)]
    for c in lanes:
        c.set_option("inline_uploads", 1)
    pipes = [DetectPipeline(c, 1, slots=3) for c in lanes]
    pool = _lib._pinned
    seen_pending = 0
    for rnd in range(12):
        a, b = rnd & 1, (rnd & 1) ^ 1
        oa = pipes[a].collect(pipes[a].submit(smp[a], prev=[prev[a]], nxt=[nxt[a]]))
        # lane a: two more batches behind which the retiring copies queue up (~0.4 ms of kernels), then retire lane a's handles and
        # drop them immediately
        ts = [pipes[a].submit(smp[a], prev=[prev[a]], nxt=[nxt[a]]) for _ in range(2)]
        held = [oa["mask_fixed"][0], oa["mask_dyn"][0]]
        refs = [r for sl in pipes[a].slots for r in sl.handles if r() is not None]
        _retire_all(refs)
        assert all(h._pending is not None for h in held)
        del held, oa
        seen_pending += len(pool.pending)
        # lane b: same-size handles materialised right now -- they take blocks from the pool
        ob = pipes[b].collect(pipes[b].submit(smp[b], prev=[prev[b]], nxt=[nxt[b]]))
        sb = pipes[b].slots[(pipes[b]._turn - 1) % 3]
        for key, buf in (("mask_fixed", sb.mf), ("mask_dyn", sb.md)):
            got = np.asarray(ob[key][0]).copy()
            lanes[a].sync()                                                              # lane a's late copies land now (into blocks nobody else may hold)
            sync = buf.download(np.uint8, (H, W)).view(np.bool_)
            assert np.array_equal(got, sync) and np.array_equal(np.asarray(ob[key][0]), sync), (rnd, key)
        for t in ts:
            pipes[a].collect(t)
    assert seen_pending > 0, "the scenario never had a block waiting for its copy: the test did not exercise the guard"
    for p in pipes:
        p.close()
    for c in lanes:
        c.close()
    assert not pool.guard or all(m.done() for m in pool.guard.values())
