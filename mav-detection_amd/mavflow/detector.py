"""Detector -- the Algorithm switch, IMU derotation and the level-0 window search of the reference's Detector
(/root/reference/src/detector.py:14-117,280-312,430-433) on libmavflow.  The homography / affine / essential-matrix
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
        return im_helpers._ctx(W, H).derotate(np.asarray(flow_uv, np.float32), omega, dt)[0]

    def analyze_pyramid(self, img: np.ndarray) -> Tuple[float, utils.Rectangle, np.ndarray, Any]:
        """Highest-sum 64x64 window (stride 16, first maximum wins) of pyramid level 0.
        Returns (score, Rectangle, window, argmax inside the window) like the reference."""
        a = np.asarray(img)
        if a.dtype != np.uint8:
            raise TypeError("analyze_pyramid expects the u8 image im_helpers.to_rgb produces")
        if a.ndim == 3:
            if not (np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 0], a[..., 2])):
                raise ValueError("analyze_pyramid: 3-channel input must be a gray replica (im_helpers.to_rgb)")
            gray, mult = np.ascontiguousarray(a[..., 0]), 1
        else:
            gray, mult = a, 3
        H, W = gray.shape
        score, x, y = (int(v) for v in im_helpers._ctx(W, H).window_max(gray)[0])
        if score == 0:
            return (0, utils.Rectangle((0, 0), (0, 0)), np.zeros(0), 0)
        window = a[y:y + 64, x:x + 64]
        return (score // mult, utils.Rectangle((x, y), (64, 64)), window, np.unravel_index(window.argmax(), window.shape))

    def is_homography_based(self) -> bool:
        return self.algorithm in [Detector.Algorithm.HOMOGRAPHY]
