"""Processor -- the detection loop of /root/reference/src/processor.py:277-396 (FoE branch) on libmavflow.

run_detection() keeps the reference's shape: one frame index at a time, the same order of operations
(:305-341), the same FrameResult fields (:353-362); run_detection_staged() is that loop through the reference-named
calls one at a time.  run_detection_batched() is the MI355X form of the same loop:
frame pairs are independent once the flow no longer comes from files, so they go through the fused
mav_process_batch entry point `batch` pairs at a time.  File / video / PNG output and the homography branch are
outside the hot path and are not reproduced."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib, im_helpers, synth, utils
from .detector import Detector
from .focus_of_expansion import FocusOfExpansion
from .frame_result import FrameResult
from .run_config import RunConfig


class SyntheticDataset:
    """A dataset object with the getters run_detection uses (datasets/dataset.py:152-344), fed by mavflow.synth: a textured
    scene under radial (FoE) motion plus one 24x24 patch moving against it.  Flow comes from libmavflow's Farneback
    (use_farneback=True, the seam this build adds) or from the analytic field (the reference's .flo seam)."""

    def __init__(self, W: int = 640, H: int = 480, N: int = 6, use_farneback: bool = True, dt: float = 1 / 30.0,
                 dangle=(0.0, 0.0, 0.0), seed: int = 0):
        self.capture_size = (W, H)
        self.resolution = np.array([W, H])
        self.constant_segmentation = True                # one segmentation image for all frames (Processor derives it once)
        self.N = N
        self.sequence = f"synthetic-{seed}"
        self.use_farneback = use_farneback
        self.dt = dt
        self.dangle = np.asarray(dangle, np.float64)
        self.seed = seed
        self._pairs = {}
        self._gt32 = {}
        self._bgr = {}
        self._seg = self._sky = self._depth = None       # the constant images are built once, like files read once
        self._ctx: Optional[_lib.Context] = None
        self._frame_cursor = 0

    def _pair(self, i: int):
        if i not in self._pairs:
            self._pairs[i] = synth.make_pair(self.capture_size[0], self.capture_size[1], self.seed * 1000 + i)
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

    def get_flow_uv(self, i: int) -> np.ndarray:
        f0, f1, truth = self._pair(i)
        if not self.use_farneback:
            return self.get_gt_of(i)                     # the analytic field, float32 (what a .flo file would hold)
        if self._ctx is None:
            self._ctx = _lib.Context(self.capture_size[0], self.capture_size[1], 1)
        return self._ctx.farneback(f0, f1)[0]

    def get_gt_of(self, i: int) -> np.ndarray:
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
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


class Processor:
    def __init__(self, config: RunConfig) -> None:
        self.config = config
        self.logger = config.logger
        self.sequence = config.sequence
        self.debug_mode = config.debug
        self.headless = config.headless
        self.dataset = config.get_dataset()
        self.detector = Detector(self.dataset)
        self.detection_results: Dict[int, FrameResult] = dict()
        self.frame_step_size = 1
        self.frame_index, self.start_frame = 0, 100
        self.is_exiting = False
        self.focus_of_expansion = FocusOfExpansion(self.detector.lucas_kanade)
        self.flow_uv = None
        self._derot, self._derot_frame = None, 0

    def is_active(self) -> bool:
        return self.frame_index < self.dataset.N - 1 and not self.is_exiting

    # -- validation tail shared by the loops (processor.py:343-362) ---------------------------------------------------
    def _fill_result(self, i: int, foe_dense, estimate_fixed, total_mask, sky_scores, masks_on_device: bool = False) -> FrameResult:
        r = FrameResult()
        r.foe_dense = foe_dense
        r.foe_gt = utils.assert_type(self.dataset.get_gt_foe(i))
        segmentation, rows, cols = self._segmentation(i)
        if masks_on_device:      # first thing: both masks are still where the detection call left them on the device, and stay
            # there only until the next call on that context -- count there (mav_last_masks_tpr_fpr) instead of re-uploading
            (r.tpr_fixed, r.fpr_fixed), (r.tpr, r.fpr) = im_helpers.tpr_fpr_of_last_masks(segmentation, 255)
        # ground-truth flow of the drone: derotated at the drone's pixels only (pointwise, same values as derotating the frame)
        gt = utils.assert_type(self.dataset.get_gt_of(i))
        with np.errstate(all="ignore"):
            drone_flow_avg_gt = np.average(self.detector.derotate_at(i - self.frame_step_size, i, gt[rows, cols], rows, cols), axis=0)
        center = self._gt_center(segmentation)
        r.center_phi = np.rad2deg(np.arctan2(center[1] - r.foe_gt[1], center[0] - r.foe_gt[0]))
        if not masks_on_device:
            r.tpr_fixed, r.fpr_fixed = im_helpers.calculate_tpr_fpr(segmentation, 255 * estimate_fixed)     # as processor.py:350-351
            r.tpr, r.fpr = im_helpers.calculate_tpr_fpr(segmentation, 255 * total_mask)
        r.sky_tpr, r.sky_fpr = sky_scores
        r.drone_flow_pixels = (drone_flow_avg_gt[0], drone_flow_avg_gt[1])
        r.drone_size_pixels = np.int64(rows.size)              # == np.sum(segmentation > 127), a numpy integer as in the reference
        r.time = self.dataset.get_time(i)
        return r

    def _segmentation(self, i: int):
        """Channel 0 of the dataset's segmentation image, contiguous, with the coordinates of its drone pixels (> 127) and the
        centre of its bounding box (processor.py:332,344-347).  Derived per frame, as the reference's loop does, unless the dataset
        DECLARES its segmentation constant (attribute `constant_segmentation`, SyntheticDataset): then once.  (Caching on the
        array's identity alone would hand a dataset that refills one preallocated array the first frame's pixels for ever.)"""
        seg3 = self.dataset.get_segmentation(i)
        if getattr(self.dataset, "constant_segmentation", False) and getattr(self, "_seg_val", None) is not None:
            return self._seg_val
        seg = np.ascontiguousarray(seg3[..., 0])
        rows, cols = np.nonzero(seg > 127)
        self._seg_val = (seg, rows, cols)
        self._center = None
        return self._seg_val

    def _gt_center(self, segmentation: np.ndarray):
        """get_simple_bounding_box(segmentation).get_center() (processor.py:346-347) of the image _segmentation() last derived.
        Evaluated on first use, AFTER the detection call's masks have been counted on the device: the box is a device call on the
        same context and would displace them."""
        if self._center is None:
            self._center = im_helpers.get_simple_bounding_box(segmentation).get_center()
        return self._center

    # The reference's loop also leaves flow_uv_derotated and flow_mag behind (processor.py:306-307).  The fused call never forms
    # them on the host; they are derived on first access from the frame's flow, through the same shims the staged loop uses.
    @property
    def flow_uv_derotated(self):
        if self._derot is None and self.flow_uv is not None:
            self._derot = self.detector.derotate(self._derot_frame - self.frame_step_size, self._derot_frame, self.flow_uv)
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

    def run_detection(self) -> Dict[int, FrameResult]:
        """The reference's loop (processor.py:283-341): one frame index at a time, flow from the dataset's seam
        (get_flow_uv: a .flo file or Farneback on the GPU), then derotation -> FoE -> phi -> masks in ONE device call
        (mav_detect): the float32 field crosses PCIe once, the masks come back, nothing else moves.  Frame 0 takes the
        reference's float32 path (detector.py:80-81).  The sample coordinates are drawn from np.random exactly where
        get_FOE_dense draws them."""
        W, H = self.dataset.capture_size
        ctx = im_helpers._ctx(W, H)
        while self.is_active():
            i = self.frame_index
            self.dataset.get_frame()
            self.flow_uv = self.dataset.get_flow_uv(i)
            if self.flow_uv is None:
                raise ValueError("Could not load flow field.")
            self._derot, self._derot_frame = None, i
            if np.asarray(self.flow_uv).dtype != np.float32:
                # a float64 field is evaluated in float64 from the start by the reference: the fused float32 call would narrow
                # it, so this frame goes through the float64 kernels (the staged calls)
                self._staged_frame(i)
                continue
            self.sky_mask = self.dataset.get_sky_segmentation(i)
            sky = self.dataset.validate_sky_segment(self.sky_mask, utils.assert_type(self.dataset.get_depth(i)))
            rand1 = np.zeros((2000, 2), dtype=np.uint32)                 # focus_of_expansion.py:69-71
            rand1[..., 0] = np.random.randint(0, self.flow_uv.shape[0], 2000)
            rand1[..., 1] = np.random.randint(0, self.flow_uv.shape[1], 2000)
            omega, dt = self._rates(i) if i >= 1 else (np.zeros(3), 1.0)
            out = ctx.detect(self.flow_uv, rand1, omega=omega, dt=dt, sky=self.sky_mask, frame0=[i < 1],
                             foe_params=self.focus_of_expansion._foe_params(1000))
            rec = out["results"][0]
            self.estimate_fixed, self.total_mask = out["mask_fixed"][0], out["mask_dyn"][0]
            r = self._fill_result(i, (float(rec["foe"][0]), float(rec["foe"][1])), self.estimate_fixed, self.total_mask, sky,
                                  masks_on_device=True)
            self.detection_results[i] = r
            self.config.results[i] = r
            self.frame_index += 1
        return self.detection_results

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
        self.flow_uv_derotated = self.detector.derotate(i - self.frame_step_size, i, self.flow_uv)
        self.sky_mask = self.dataset.get_sky_segmentation(i)
        sky = self.dataset.validate_sky_segment(self.sky_mask, utils.assert_type(self.dataset.get_depth(i)))
        foe = self.focus_of_expansion.get_FOE_dense(self.flow_uv_derotated)
        fixed, total = self.focus_of_expansion.get_masks(self.flow_uv_derotated, foe, self.sky_mask)
        self.estimate_fixed, self.total_mask = fixed, total
        r = self._fill_result(i, foe, fixed, total, sky)
        self.detection_results[i] = r
        self.config.results[i] = r
        self.frame_index += 1

    def run_detection_batched(self, batch: int = 8) -> Dict[int, FrameResult]:
        """The same loop with frame pairs in flight `batch` at a time through the fused entry point (frames -> flow ->
        derotation -> FoE -> masks -> box).  Needs a dataset that hands out frame pairs (frame_pair(i))."""
        W, H = self.dataset.capture_size
        idx = list(range(self.frame_index, self.dataset.N - 1))
        with _lib.Context(W, H, batch) as ctx:
            for b0 in range(0, len(idx), batch):
                ids = idx[b0:b0 + batch]
                prev = np.stack([self.dataset.frame_pair(i)[0] for i in ids])
                nxt = np.stack([self.dataset.frame_pair(i)[1] for i in ids])
                samples = np.zeros((len(ids), 2000, 2), np.uint32)
                for k in range(len(ids)):                      # same draws, same order as get_FOE_dense
                    samples[k, :, 0] = np.random.randint(0, H, 2000)
                    samples[k, :, 1] = np.random.randint(0, W, 2000)
                dts = np.array([self.dataset.get_delta_time(i) for i in ids], np.float64)
                rot = [i >= 1 for i in ids]
                omega = np.stack([np.asarray(self.dataset.get_angular_difference(i - 1, i), np.float64) / dt
                                  for i, dt in zip(ids, dts)])
                # frame 0 is never derotated and runs in float32 (detector.py:80-81): flagged per pair below
                sky = np.stack([self.dataset.get_sky_segmentation(i) for i in ids])
                out = ctx.process_batch(prev, nxt, samples, omega=omega, dt=dts, sky=sky, frame0=[not r_ for r_ in rot])
                for k, i in enumerate(ids):
                    rec = out["results"][k]
                    r = self._fill_result(i, (float(rec["foe"][0]), float(rec["foe"][1])), out["mask_fixed"][k], out["mask_dyn"][k],
                                          (0.0, 0.0))
                    r.box = utils.Rectangle.from_box(rec["box"])   # extra: the detection box (the reference never stores one)
                    self.detection_results[i] = r
                    self.config.results[i] = r
        self.frame_index = self.dataset.N - 1
        return self.detection_results

    def release(self) -> None:
        self.dataset.release()
