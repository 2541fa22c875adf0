"""Detector -- the Algorithm switch, IMU derotation and the window search (analyze_pyramid, optimize_window) of the
reference's Detector (/root/reference/src/detector.py:14-117,280-358,430-433) on libmavflow.  The homography / affine / essential-matrix
branches (cv2 RANSAC estimators, dead in run_detection) are not part of the hot path."""
from __future__ import annotations

from enum import Enum
from typing import Any, Tuple

import numpy as np

from . import im_helpers, utils
from .frame_result import FrameResult


class LucasKanade:
    """Only what the hot path reads from the reference's sparse tracker (lucas_kanade.py:9-32): the frame shape, the
    corner budget, and the constructor's draw from the global RNG."""

    def __init__(self, old_frame: np.ndarray) -> None:
        self.old_frame = old_frame
        self.num_corners = 2000
        self.minimum_num_corners = self.num_corners // 3
        self.total_num_corners = self.num_corners + self.minimum_num_corners
        self.color = np.random.randint(0, 255, (self.total_num_corners, 3))


class Detector:
    class Algorithm(Enum):
        NONE = 0,
        FOE = 1,
        AFFINE = 2,
        HOMOGRAPHY = 3,
        FUNDAMENTAL = 4,
        ESSENTIAL = 5,

    def __init__(self, dataset, algorithm: "Detector.Algorithm" = None, use_sparse_of: bool = False) -> None:
        self.dataset = dataset
        self.algorithm = Detector.Algorithm.ESSENTIAL if algorithm is None else algorithm
        self.use_sparse_of = use_sparse_of
        W, H = self.dataset.capture_size[0], self.dataset.capture_size[1]
        self.sample_size = 1000
        self.border_offset = 20
        # same draws, same order as the reference's constructor (keeps a seeded global RNG stream aligned)
        self.sample_y = np.random.randint(self.border_offset, H - self.border_offset, self.sample_size)
        self.sample_x = np.random.randint(self.border_offset, W - self.border_offset, self.sample_size)
        self.coords = np.column_stack((self.sample_x, self.sample_y))
        self.confidence = 0
        self.prev_frame = np.zeros((H, W, 3), dtype=np.uint8)
        self.lucas_kanade = LucasKanade(self.prev_frame)
        self.fov = 90
        self.focal_length = 1 / np.tan(np.deg2rad(self.fov) / 2)
        self.frame_result = FrameResult()

    def derotate(self, previous_frame_index: int, current_frame_index: int, flow_uv: np.ndarray) -> np.ndarray:
        """Subtract the rotational flow predicted from the IMU rates; float64 out.  Frame 0 is returned untouched."""
        if current_frame_index < 1:
            return flow_uv
        dt = self.dataset.get_delta_time(current_frame_index)
        omega = np.asarray(self.dataset.get_angular_difference(previous_frame_index, current_frame_index), np.float64) / dt
        W, H = self.dataset.capture_size[0], self.dataset.capture_size[1]
        flow_uv = np.asarray(flow_uv)
        if flow_uv.dtype != np.float32:
            # a float64 field (no dataset of the reference produces one: .flo files and Farneback are float32) keeps its precision:
            # the reference's own numpy arithmetic on the host, not the float32-input kernel
            rows, cols = np.mgrid[0:flow_uv.shape[0], 0:flow_uv.shape[1]]
            return self.derotate_at(previous_frame_index, current_frame_index, flow_uv.astype(np.float64, copy=False), rows, cols)
        return im_helpers._ctx(W, H).derotate(flow_uv, omega, dt)[0]

    def derotate_at(self, previous_frame_index: int, current_frame_index: int, flow_values: np.ndarray, rows: np.ndarray,
                    cols: np.ndarray) -> np.ndarray:
        """derotate() restricted to the pixels (rows[k], cols[k]) whose flow vectors are flow_values[k]: the correction is
        pointwise, so selecting first and derotating after gives the values the reference gets by derotating the whole field
        and selecting (processor.py:310,343-345 need the ground-truth flow at the few hundred drone pixels only).  Host numpy in
        the reference's operation order (detector.py:83-117); float64 out, frame 0 untouched."""
        if current_frame_index < 1:
            return flow_values
        dt = self.dataset.get_delta_time(current_frame_index)
        w, h = self.dataset.capture_size[0], self.dataset.capture_size[1]
        omega = np.asarray(self.dataset.get_angular_difference(previous_frame_index, current_frame_index), np.float64) / dt
        x = -(np.asarray(cols) / w - 0.5) * 2.0
        y = -(np.asarray(rows) / h - 0.5) * 2.0
        du = +omega[0] * x * y - omega[1] * x ** 2 - omega[1] + omega[2] * y
        dv = -omega[2] * x + omega[0] + omega[0] * y ** 2 - omega[1] * x * y
        du = du * (w * dt / 2)
        dv = dv * (h * dt / 2)
        return flow_values - np.stack([du, dv], axis=-1)

    @staticmethod
    def _gray_of(img: np.ndarray, who: str):
        a = np.asarray(img)
        if a.dtype != np.uint8:
            raise TypeError(f"{who} expects the u8 image im_helpers.to_rgb produces")
        if a.ndim == 3:
            if not (np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 0], a[..., 2])):
                raise ValueError(f"{who}: 3-channel input must be a gray replica (im_helpers.to_rgb)")
            return a, np.ascontiguousarray(a[..., 0]), 1
        return a, a, 3

    def analyze_pyramid(self, img: np.ndarray) -> Tuple[float, utils.Rectangle, np.ndarray, Any]:
        """Highest-sum 64x64 window (stride 16, first maximum wins) over every pyramid level (scale 1.5, INTER_AREA).
        Returns (score, Rectangle, window, argmax inside the window) like the reference (detector.py:280-312): the
        rectangle is in the winning level's own coordinates and `window` is that level's sub-image."""
        a, gray, mult = self._gray_of(img, "analyze_pyramid")
        H, W = gray.shape
        ctx = im_helpers._ctx(W, H)
        score, x, y, level, ay, ax = (int(v) for v in ctx.analyze_pyramid(gray)[0])
        if score == 0:
            return (0, utils.Rectangle((0, 0), (0, 0)), np.zeros(0), 0)
        lv = gray if level == 0 else ctx.pyramid_level(gray, level)
        window = lv[y:y + 64, x:x + 64]
        if a.ndim == 3:
            window = np.repeat(window[..., None], 3, axis=2)
        return (score // mult, utils.Rectangle((x, y), (64, 64)), window, (ay, ax, 0) if a.ndim == 3 else (ay, ax))

    def optimize_window(self, mag_img: np.ndarray, window: utils.Rectangle) -> Tuple[float, utils.Rectangle]:
        """Greedy corner walk of detector.py:314-358 (the window grows / shrinks while the enclosed sum rises)."""
        a, gray, mult = self._gray_of(mag_img, "optimize_window")
        H, W = gray.shape
        win = [int(window.get_left()), int(window.get_top()), int(window.get_right()) - int(window.get_left()),
               int(window.get_bottom()) - int(window.get_top())]
        score, out = im_helpers._ctx(W, H).optimize_window(gray, [win])
        if int(score[0]) == 0:
            return (0.0, window)
        x, y, w, h = (int(v) for v in out[0])
        return (float(int(score[0]) // mult), utils.Rectangle.from_points((x, y), (x + w, y + h)))

    def is_homography_based(self) -> bool:
        return self.algorithm in [Detector.Algorithm.HOMOGRAPHY]
