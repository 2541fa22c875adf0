"""Processor -- the detection loop of /root/reference/src/processor.py:277-396 (FoE branch) on libmavflow.

run_detection() keeps the reference's shape: one frame index at a time, the same order of operations
(:305-341), the same FrameResult fields (:353-362); run_detection_staged() is that loop through the reference-named
calls one at a time.  run_detection_batched() is the MI355X form of the same loop:
frame pairs are independent once the flow no longer comes from files, so they go through the fused
mav_process_batch_dev entry point `batch` pairs at a time, two batches in flight.

What the fast loops move (mavflow/pipeline.py): frames (or a host flow field) in; one 32-byte record and eight counts per frame
out.  Flow, derotated flow and both masks stay on the device as DeviceArray handles (Processor.flow_uv, .estimate_fixed,
.total_mask) that turn into host arrays the moment somebody reads them; the calculate_tpr_fpr counts of both masks (:350-351) are
taken on the device against a ground truth that is uploaded once when the dataset declares it constant.

Each loop writes `{results_path}/image_{i:05d}.json` per frame as the reference's write() does (:83-84) when the dataset (or the
constructor) names a results_path.  Video / PNG output and the homography branch are outside the hot path and are not reproduced."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib, im_helpers, pipeline, synth, utils
from .detector import Detector
from .focus_of_expansion import FocusOfExpansion
from .frame_result import FrameResult
from .run_config import RunConfig


class SyntheticDataset:
    """A dataset object with the getters run_detection uses (datasets/dataset.py:152-344), fed by mavflow.synth: a textured
    scene under radial (FoE) motion plus one 24x24 patch moving against it.  Flow comes from libmavflow's Farneback
    (use_farneback=True, the seam this build adds) or from the analytic field (the reference's .flo seam)."""

    def __init__(self, W: int = 640, H: int = 480, N: int = 6, use_farneback: bool = True, dt: float = 1 / 30.0,
                 dangle=(0.0, 0.0, 0.0), seed: int = 0, distinct: Optional[int] = None, results_path: Optional[str] = None,
                 video: bool = False, lanes: Optional[int] = None):
        self.capture_size = (W, H)
        self.resolution = np.array([W, H])
        self.constant_segmentation = True                # one segmentation image for all frames (Processor derives it once)
        self.constant_sky_segmentation = True            # ... and one sky mask
        self.N = N
        self.distinct = distinct                         # pair i shows the content of pair i % distinct (long runs from few pictures)
        # video = True: ONE sequence of frames (synth.make_sequence: a texture under a growing zoom), pair i = (frame i, frame i + 1) as
        # the SAME array objects -- how a video runs through the reference's loop; the batched loop then uploads and expands every frame once
        self.video = video
        self._frames = None
        self.lanes = lanes                               # contexts the Farneback seam takes in turn (None: pipeline.auto_lanes for one pair)
        self.results_path = results_path                 # where Processor writes image_%05d.json (dataset.py: results_path)
        self.sequence = f"synthetic-{seed}"
        self.use_farneback = use_farneback
        self.dt = dt
        self.dangle = np.asarray(dangle, np.float64)
        self.seed = seed
        self._pairs = {}
        self._gt32 = {}
        self._bgr = {}
        self._seg = self._sky = self._depth = None       # the constant images are built once, like files read once
        self._ctxs = []
        self._stage: Optional[pipeline.LanedFlowStage] = None
        self._frame_cursor = 0

    VIDEO_K = 0.004                                      # zoom per frame of the video form (synth.make_sequence)

    def _pair(self, i: int):
        if self.distinct:
            i %= self.distinct
        if i not in self._pairs:
            W, H = self.capture_size
            if self.video:
                if self._frames is None:
                    n = (self.distinct or (self.N - 1)) + 1
                    self._frames = list(synth.make_sequence(W, H, n, seed=self.seed, k=self.VIDEO_K))
                # frame j is the texture zoomed by j k about c = (0.55 W, 0.45 H): a point seen at x in frame i sits at
                # x + k (x - c) / (1 - (i + 1) k) in frame i + 1
                g = self.VIDEO_K / (1.0 - (i + 1) * self.VIDEO_K)
                truth = np.empty((H, W, 2))
                truth[..., 0] = (g * (np.arange(W) - 0.55 * W))[None, :]
                truth[..., 1] = (g * (np.arange(H) - 0.45 * H))[:, None]
                self._pairs[i] = (self._frames[i], self._frames[i + 1], truth)
            else:
                self._pairs[i] = synth.make_pair(W, H, self.seed * 1000 + i)
        return self._pairs[i]

    def frame_pair(self, i: int) -> Tuple[np.ndarray, np.ndarray]:
        f0, f1, _ = self._pair(i)
        return f0, f1

    def get_frame(self) -> np.ndarray:
        i = self._frame_cursor % self.N
        self._frame_cursor += 1
        if i not in self._bgr:
            self._bgr[i] = np.repeat(self._pair(i)[1][..., None], 3, axis=2)
        return self._bgr[i]

    def get_flow_uv(self, i: int):
        """float32 (H, W, 2): the analytic field (a host array, what a .flo file would hold) or libmavflow's Farneback -- then as a
        pipeline.DeviceArray: array-like, on the device until read, so the loop's next stage takes it from where it is."""
        f0, f1, truth = self._pair(i)
        if not self.use_farneback:
            return self.get_gt_of(i)
        if self._stage is None:
            W, H = self.capture_size
            n = self.lanes or pipeline.auto_lanes(W, H, 1)
            self._ctxs = [_lib.Context(W, H, 1) for _ in range(n)]
            self._stage = pipeline.LanedFlowStage(self._ctxs)
        return self._stage.flow_of(f0, f1)

    def get_gt_of(self, i: int) -> np.ndarray:
        if self.distinct:
            i %= self.distinct
        if i not in self._gt32:
            self._gt32[i] = self._pair(i)[2].astype(np.float32)
        return self._gt32[i]

    def get_segmentation(self, i: int) -> np.ndarray:
        if self._seg is None:
            W, H = self.capture_size
            self._seg = np.zeros((H, W, 3), np.uint8)
            self._seg[H // 4:H // 4 + 24, W // 4:W // 4 + 24] = 255
        return self._seg

    def get_sky_segmentation(self, i: int) -> np.ndarray:
        if self._sky is None:
            self._sky = np.zeros((self.capture_size[1], self.capture_size[0]), dtype=bool)
        return self._sky

    def get_depth(self, i: int) -> np.ndarray:
        if self._depth is None:
            self._depth = np.ones((self.capture_size[1], self.capture_size[0]), np.float32)
        return self._depth

    def validate_sky_segment(self, sky_mask, depth_buffer) -> Tuple[float, float]:
        return (0.0, 0.0)

    def get_gt_foe(self, i: int) -> Tuple[float, float]:
        return (0.55 * self.capture_size[0], 0.45 * self.capture_size[1])

    def get_time(self, i: int) -> float:
        return i * self.dt

    def get_delta_time(self, i: int) -> float:
        return self.dt

    def get_angular_difference(self, a: int, b: int) -> np.ndarray:
        return self.dangle

    def release(self) -> None:
        if self._stage is not None:
            self._stage.close()
            for c in self._ctxs:
                c.close()
            self._ctxs, self._stage = [], None


class Processor:
    def __init__(self, config: RunConfig, results_path: Optional[str] = None) -> None:
        self.config = config
        self.logger = config.logger
        self.sequence = config.sequence
        self.debug_mode = config.debug
        self.headless = config.headless
        self.dataset = config.get_dataset()
        self.detector = Detector(self.dataset)
        self.detection_results: Dict[int, FrameResult] = dict()
        # extra: get_simple_bounding_box of every frame's fixed mask, which the device records with the FoE (the reference derives no
        # box in this loop; FrameResult and its JSON stay exactly the reference's)
        self.detection_boxes: Dict[int, utils.Rectangle] = dict()
        self.frame_step_size = 1
        self.frame_index, self.start_frame = 0, 100
        self.is_exiting = False
        self.focus_of_expansion = FocusOfExpansion(self.detector.lucas_kanade)
        self.flow_uv = None
        self._derot, self._derot_frame = None, 0
        # {results_path}/image_{i:05d}.json per frame (processor.py:83-84); the reference takes the directory from its dataset
        self.results_path = results_path if results_path is not None else getattr(self.dataset, "results_path", None)
        self._ctxs = []                                 # this loop's own contexts (never the helpers' shared, evictable ones)
        self._pipes = {}
        self._stale_pipes = []                           # pipelines over a subset of the lanes now in use, until their last frames are collected
        self._seg_val = None
        self._center = None

    def is_active(self) -> bool:
        return self.frame_index < self.dataset.N - 1 and not self.is_exiting

    # -- device plumbing --------------------------------------------------------------------------------------------------------
    def _pipeline(self, ctxs, batch: int) -> "pipeline.LanedPipeline":
        """The (cached) pipeline over these contexts for `batch` pairs per submit."""
        key = (tuple(id(c) for c in ctxs), batch)
        pipe = self._pipes.get(key)
        if pipe is None or any(a is not b or not b.alive for a, b in zip(pipe.ctxs, ctxs)):
            if pipe is not None:
                pipe.close()
            # a seam that shows its lanes one by one (_flow_ctxs' fallback) makes this grow [c0] -> [c0, c1] -> ...: the pipelines over
            # the smaller sets would keep their three slots of device and page-locked buffers on the same contexts until release()
            # (closed by _drop_stale_pipes once the loop has collected what it still has in flight on them)
            mine = set(key[0])
            for k in [k for k in self._pipes if k[1] == batch and k != key and set(k[0]) < mine]:
                self._stale_pipes.append(self._pipes.pop(k))
            pipe = pipeline.LanedPipeline(ctxs, batch)
            pipe.set_params(foe_params=self.focus_of_expansion._foe_params(1000))
            self._pipes[key] = pipe
        return pipe

    def _own_ctxs(self, batch: int = 1, lanes: int = 1):
        """This loop's own contexts (never the helpers' shared, evictable ones): `lanes` of them, each for `batch` pairs."""
        W, H = self.dataset.capture_size
        ok = len(self._ctxs) >= lanes and all(c.alive and c.max_batch >= batch and (c.W, c.H) == (W, H) for c in self._ctxs[:lanes])
        if not ok:
            self._close_pipes()
            for c in self._ctxs:
                c.close()
            self._ctxs = [_lib.Context(W, H, batch) for _ in range(lanes)]
        return self._ctxs[:lanes]

    def _flow_ctxs(self, flow) -> list:
        """The contexts a device-resident flow seam works on: every lane of the stage that produced this handle when the dataset shows
        it (SyntheticDataset, FarnebackFlowProvider: attribute _stage), else the handle's own context."""
        for holder in (self.dataset, getattr(self.dataset, "_flow", None)):
            st = getattr(holder, "_stage", None)
            if isinstance(st, pipeline.LanedFlowStage) and any(s.ctx is flow.ctx for s in st.stages):
                return [s.ctx for s in st.stages]
        # a seam that does not show its stage: the contexts its handles have come from so far (a seam that takes two or three contexts
        # in turn is recognised within its first frames; the pipeline is then rebuilt once over all of them)
        self._lane_seen = [c for c in getattr(self, "_lane_seen", []) if c.alive]
        if not any(c is flow.ctx for c in self._lane_seen):
            self._lane_seen.append(flow.ctx)
        return list(self._lane_seen)

    def _drop_stale_pipes(self) -> None:
        for pipe in self._stale_pipes:
            pipe.close()
        self._stale_pipes = []

    def _close_pipes(self) -> None:
        self._drop_stale_pipes()
        for pipe in self._pipes.values():
            pipe.close()
        self._pipes = {}

    # -- validation tail shared by the loops (processor.py:343-362) ---------------------------------------------------
    def _fill_result(self, i: int, foe_dense, sky_scores, counts_fixed=None, counts_dyn=None, estimate_fixed=None, total_mask=None) -> FrameResult:
        """The FrameResult of frame i.  The TPR / FPR pairs come from the counts the device took of both masks (the fast loops), or,
        when masks are given instead, from calculate_tpr_fpr on them as the reference's lines read (the staged loop)."""
        r = FrameResult()
        r.foe_dense = foe_dense
        r.foe_gt = utils.assert_type(self.dataset.get_gt_foe(i))
        segmentation, rows, cols = self._segmentation(i)
        # ground-truth flow of the drone: derotated at the drone's pixels only (pointwise, same values as derotating the frame)
        gt = utils.assert_type(self.dataset.get_gt_of(i))
        with np.errstate(all="ignore"):
            drone_flow_avg_gt = np.average(self.detector.derotate_at(i - self.frame_step_size, i, gt[rows, cols], rows, cols), axis=0)
        center = self._gt_center(segmentation)
        r.center_phi = np.rad2deg(np.arctan2(center[1] - r.foe_gt[1], center[0] - r.foe_gt[0]))
        if counts_fixed is not None:
            r.tpr_fixed, r.fpr_fixed = im_helpers._rates(counts_fixed)
            r.tpr, r.fpr = im_helpers._rates(counts_dyn)
        else:
            r.tpr_fixed, r.fpr_fixed = im_helpers.calculate_tpr_fpr(segmentation, 255 * estimate_fixed)     # as processor.py:350-351
            r.tpr, r.fpr = im_helpers.calculate_tpr_fpr(segmentation, 255 * total_mask)
        r.sky_tpr, r.sky_fpr = sky_scores
        r.drone_flow_pixels = (drone_flow_avg_gt[0], drone_flow_avg_gt[1])
        r.drone_size_pixels = np.int64(rows.size)              # == np.sum(segmentation > 127), a numpy integer as in the reference
        r.time = self.dataset.get_time(i)
        return r

    def _store(self, i: int, r: FrameResult) -> None:
        self.detection_results[i] = r
        self.config.results[i] = r
        if self.results_path is not None:                      # what write() leaves per frame (processor.py:83-84), same text
            os.makedirs(self.results_path, exist_ok=True)
            with open(f"{self.results_path}/image_{i:05d}.json", "w") as f:
                f.write(json.dumps(utils.get_json(r), indent=4, sort_keys=True))

    def _segmentation(self, i: int):
        """Channel 0 of the dataset's segmentation image, contiguous, with the coordinates of its drone pixels (> 127) and the
        centre of its bounding box (processor.py:332,344-347).  Derived per frame, as the reference's loop does, unless the dataset
        DECLARES its segmentation constant (attribute `constant_segmentation`, SyntheticDataset): then once.  (Caching on the
        array's identity alone would hand a dataset that refills one preallocated array the first frame's pixels for ever.)"""
        seg3 = self.dataset.get_segmentation(i)
        if getattr(self.dataset, "constant_segmentation", False) and self._seg_val is not None:
            return self._seg_val
        seg = np.ascontiguousarray(seg3[..., 0])
        rows, cols = np.nonzero(seg > 127)
        self._seg_val = (seg, rows, cols)
        self._center = None
        return self._seg_val

    def _gt_center(self, segmentation: np.ndarray):
        """get_simple_bounding_box(segmentation).get_center() (processor.py:346-347) of the image _segmentation() last derived."""
        if self._center is None:
            self._center = im_helpers.get_simple_bounding_box(segmentation).get_center()
        return self._center

    # The reference's loop also leaves flow_uv_derotated and flow_mag behind (processor.py:306-307).  The fused call never forms
    # them on the host; they are derived on first access from the frame's flow, through the same shims the staged loop uses.
    @property
    def flow_uv_derotated(self):
        if self._derot is None and self.flow_uv is not None:
            self._derot = self.detector.derotate(self._derot_frame - self.frame_step_size, self._derot_frame, np.asarray(self.flow_uv))
        return self._derot

    @flow_uv_derotated.setter
    def flow_uv_derotated(self, v):
        self._derot = v

    @property
    def flow_mag(self):
        d = self.flow_uv_derotated
        return None if d is None else im_helpers.get_magnitude(d)

    def _rates(self, i: int):
        dt = self.dataset.get_delta_time(i)
        return np.asarray(self.dataset.get_angular_difference(i - self.frame_step_size, i), np.float64) / dt, dt

    def _sky_and_gt(self, ids):
        """(submit() keywords for the sky masks and the ground truth of these frames, their segmentation tuples).  A dataset that declares
        an image constant has it uploaded once for the run; otherwise one image per frame travels with the batch."""
        kw = {}
        skies = [self.dataset.get_sky_segmentation(i) for i in ids]
        if getattr(self.dataset, "constant_sky_segmentation", False):
            kw["sky_shared"] = skies[0]
        else:
            kw["sky"] = skies
        segs = [self._segmentation(i) for i in ids]
        if getattr(self.dataset, "constant_segmentation", False):
            kw["gt_shared"] = segs[0][0]
        else:
            kw["gt"] = [sg[0] for sg in segs]
        return kw, skies

    def run_detection(self) -> Dict[int, FrameResult]:
        """The reference's loop (processor.py:283-341): one frame index at a time, flow from the dataset's seam (get_flow_uv: a
        .flo file -> a host array, or Farneback on the GPU -> a DeviceArray), then derotation -> FoE -> phi -> masks -> box -> the
        calculate_tpr_fpr counts of both masks in ONE enqueue on the context that holds the flow (pipeline.DetectPipeline): a host
        field crosses PCIe once, a device field not at all; a 32-byte record and eight counts come back.  estimate_fixed /
        total_mask are DeviceArray handles (read them and they are host arrays).  Frame 0 takes the reference's float32 path
        (detector.py:80-81).  The sample coordinates are drawn from np.random exactly where get_FOE_dense draws them.
        Software-pipelined: frame i's FrameResult is filled in (and its JSON written) after the following frames -- as many as the
        pipeline has lanes (pipeline.auto_lanes: 3 - 4 contexts taken in turn for frames up to 1080p, whose one-pair chains of launches
        then interleave on the GPU) -- have been enqueued; results, files and their order are those of the plain loop."""
        from collections import deque
        pending = deque()

        def finish(n_keep: int) -> None:
            while len(pending) > n_keep:
                self._finish_frame(*pending.popleft())

        pipe = None
        while self.is_active():
            i = self.frame_index
            self.dataset.get_frame()
            self.flow_uv = self.dataset.get_flow_uv(i)
            if self.flow_uv is None:
                raise ValueError("Could not load flow field.")
            self._derot, self._derot_frame = None, i
            if self.flow_uv.dtype != np.float32:
                # a float64 field is evaluated in float64 from the start by the reference: the fused float32 call would narrow
                # it, so this frame goes through the float64 kernels (the staged calls)
                finish(0)
                self._staged_frame(i)
                continue
            on_dev = isinstance(self.flow_uv, pipeline.DeviceArray) and self.flow_uv.on_device
            # (a host flow field -- the .flo seam -- is a 3-launch chain behind a 16.6 MB upload at 1080p: PCIe-bound, one lane; measured
            # 0.43 ms per frame with one context, 0.47 - 0.53 with two)
            ctxs = self._flow_ctxs(self.flow_uv) if on_dev else self._own_ctxs(1, 1)
            now = self._pipeline(ctxs, 1)
            if now is not pipe:                                         # the flow moved to other contexts: drain the old pipeline first
                finish(0)
                self._drop_stale_pipes()
                pipe = now
            kw, skies = self._sky_and_gt([i])
            self.sky_mask = skies[0]
            sky = self.dataset.validate_sky_segment(self.sky_mask, utils.assert_type(self.dataset.get_depth(i)))
            rand1 = np.empty((2000, 2), dtype=np.uint32)                 # focus_of_expansion.py:69-71
            rand1[..., 0] = np.random.randint(0, self.flow_uv.shape[0], 2000)
            rand1[..., 1] = np.random.randint(0, self.flow_uv.shape[1], 2000)
            omega, dt = self._rates(i) if i >= 1 else (np.zeros(3), 1.0)
            ticket = pipe.submit(rand1, flow=self.flow_uv, omega=omega, dt=dt, frame0=[i < 1], **kw)
            pending.append((pipe, i, ticket, sky))
            finish(pipe.depth)
            self.frame_index += 1
        finish(0)
        return self.detection_results

    def _finish_frame(self, pipe, i: int, ticket, sky) -> None:
        out = pipe.collect(ticket)
        rec = out["results"][0]
        self.estimate_fixed, self.total_mask = out["mask_fixed"][0], out["mask_dyn"][0]
        r = self._fill_result(i, (float(rec["foe"][0]), float(rec["foe"][1])), sky, out["counts_fixed"][0], out["counts_dyn"][0])
        self.detection_boxes[i] = utils.Rectangle.from_box(rec["box"])
        self._store(i, r)

    def run_detection_staged(self) -> Dict[int, FrameResult]:
        """The same loop through the reference-named calls one by one (Detector.derotate, get_FOE_dense, the masks): every call
        ships its arrays across PCIe, as a maintainer who only swaps the imports would get.  Kept as the parity check of
        those shims; run_detection() is the fast form."""
        while self.is_active():
            i = self.frame_index
            self.dataset.get_frame()
            self.flow_uv = self.dataset.get_flow_uv(i)
            if self.flow_uv is None:
                raise ValueError("Could not load flow field.")
            self._staged_frame(i)
        return self.detection_results

    def _staged_frame(self, i: int) -> None:
        """One frame through the reference-named calls (processor.py:306-341), flow already in self.flow_uv."""
        self._derot_frame = i
        self.flow_uv_derotated = self.detector.derotate(i - self.frame_step_size, i, np.asarray(self.flow_uv))
        self.sky_mask = self.dataset.get_sky_segmentation(i)
        sky = self.dataset.validate_sky_segment(self.sky_mask, utils.assert_type(self.dataset.get_depth(i)))
        foe = self.focus_of_expansion.get_FOE_dense(self.flow_uv_derotated)
        fixed, total = self.focus_of_expansion.get_masks(self.flow_uv_derotated, foe, self.sky_mask)
        self.estimate_fixed, self.total_mask = fixed, total
        r = self._fill_result(i, foe, sky, estimate_fixed=fixed, total_mask=total)
        self._store(i, r)
        self.frame_index += 1

    def run_detection_batched(self, batch: int = 8) -> Dict[int, FrameResult]:
        """The same loop with frame pairs in flight `batch` at a time through the fused entry point (frames -> flow -> derotation ->
        FoE -> masks -> box -> counts), two batches in flight: while batch k computes, batch k + 1's frames are gathered from the
        dataset's arrays and cross PCIe, and batch k - 1's FrameResults are filled in.  Needs a dataset that hands out frame pairs
        (frame_pair(i)).  The sample coordinates are drawn per frame in frame order, as get_FOE_dense draws them."""
        from collections import deque
        W, H = self.dataset.capture_size
        idx = list(range(self.frame_index, self.dataset.N - 1))
        # small batches (one pair at 720p / 1080p ...) are spread over 2 - 3 contexts taken in turn; a big batch keeps two pairs in
        # flight inside its one context
        pipe = self._pipeline(self._own_ctxs(batch, pipeline.auto_lanes(W, H, batch)), batch)
        pending = deque()
        for b0 in range(0, len(idx), batch):
            ids = idx[b0:b0 + batch]
            pairs = [self.dataset.frame_pair(i) for i in ids]
            samples = np.empty((len(ids), 2000, 2), np.uint32)
            for k in range(len(ids)):                      # same draws, same order as get_FOE_dense
                samples[k, :, 0] = np.random.randint(0, H, 2000)
                samples[k, :, 1] = np.random.randint(0, W, 2000)
            dts = np.array([self.dataset.get_delta_time(i) for i in ids], np.float64)
            # frame 0 is never derotated and runs in float32 (detector.py:80-81): flagged per pair
            omega = np.stack([np.asarray(self.dataset.get_angular_difference(i - self.frame_step_size, i), np.float64) / dt
                              for i, dt in zip(ids, dts)])
            kw, skies = self._sky_and_gt(ids)
            sky_scores = [self.dataset.validate_sky_segment(sk, utils.assert_type(self.dataset.get_depth(i))) for i, sk in zip(ids, skies)]
            ticket = pipe.submit(samples, prev=[p[0] for p in pairs], nxt=[p[1] for p in pairs], omega=omega, dt=dts,
                                 frame0=[i < 1 for i in ids], **kw)
            pending.append((ids, ticket, sky_scores))
            while len(pending) > pipe.depth:
                self._finish_batch(pipe, *pending.popleft())
        while pending:
            self._finish_batch(pipe, *pending.popleft())
        self.frame_index = self.dataset.N - 1
        return self.detection_results

    def _finish_batch(self, pipe, ids, ticket, sky_scores) -> None:
        out = pipe.collect(ticket)
        for k, i in enumerate(ids):
            rec = out["results"][k]
            r = self._fill_result(i, (float(rec["foe"][0]), float(rec["foe"][1])), sky_scores[k], out["counts_fixed"][k], out["counts_dyn"][k])
            self.detection_boxes[i] = utils.Rectangle.from_box(rec["box"])
            self._store(i, r)
        self.estimate_fixed, self.total_mask = out["mask_fixed"][-1], out["mask_dyn"][-1]     # of the last frame, as the loop leaves them

    def release(self) -> None:
        self._close_pipes()
        for c in self._ctxs:
            c.close()
        self._ctxs = []
        self.dataset.release()
