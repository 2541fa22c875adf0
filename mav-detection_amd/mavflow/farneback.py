"""Farneback -- the reference's stateful wrapper (/root/reference/src/farneback.py:12-107) with
cv2.calcOpticalFlowFarneback replaced by libmavflow's HIP path; the parameters are the reference's literals (:78-80).

`capture` is anything with .read() -> (ok, BGR u8 frame) (cv2.VideoCapture has that shape).  process() keeps the
reference's behaviour -- it returns a BGR visualisation and falls back to the previous one when the frame produces no
flow -- and additionally exposes the dense field as .flow, which the reference computes and throws away."""
from __future__ import annotations

import numpy as np

from . import _lib


def bgr_to_gray(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(COLOR_BGR2GRAY) on u8: fixed-point (B*1868 + G*9617 + R*4899 + 8192) >> 14.
    Host form of the formula (used by tests as the checker of mav_bgr2gray); the class below converts on the GPU."""
    a = np.asarray(img)
    if a.ndim == 2:
        return np.ascontiguousarray(a, np.uint8)
    b, g, r = (a[..., i].astype(np.uint32) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def flow_to_bgr(flow: np.ndarray) -> np.ndarray:
    """Direction -> hue, magnitude -> value (visualisation only; not a parity surface)."""
    fx, fy = flow[..., 0].astype(np.float64), flow[..., 1].astype(np.float64)
    mag = np.hypot(fx, fy)
    ang = np.arctan2(fy, fx) % (2 * np.pi)
    top = mag.max()
    v = np.zeros_like(mag) if top <= 0 else np.clip(2.0 * 255 * mag / top, 0, 255)
    h6 = ang / (np.pi / 3)
    k = lambda n: (n + h6) % 6
    chan = lambda n: v - v * np.clip(np.minimum(k(n), 4 - k(n)), 0, 1)
    rgb = np.stack([chan(5), chan(3), chan(1)], axis=-1)
    return np.rint(rgb[..., ::-1]).astype(np.uint8)


class Farneback:
    PARAMS = dict(pyr_scale=0.4, levels=1, winsize=12, iterations=10, poly_n=8, poly_sigma=1.2, flags=0)

    def __init__(self, capture, output=None) -> None:
        self.capture = capture
        self.output = output
        _, prev = self.capture.read()
        H, W = prev.shape[:2]
        fb = _lib.fb_defaults()
        for k, v in self.PARAMS.items():
            setattr(fb, k, v)
        self.ctx = _lib.Context(W, H, 1, fb)
        self.prevgray = self._gray(prev)
        self.flow = np.zeros((H, W, 2), np.float32)
        self.history_length = 1
        self.prev_result = np.zeros((H, W, 3), np.uint8)

    def _gray(self, img: np.ndarray) -> np.ndarray:
        return self.ctx.bgr2gray(img)[0] if img.ndim == 3 else np.ascontiguousarray(img, np.uint8)

    def process(self) -> np.ndarray:
        _, img = self.capture.read()
        gray = self._gray(img)
        self.flow = self.ctx.farneback(self.prevgray, gray)[0]
        self.prevgray = gray
        result = flow_to_bgr(self.flow)
        mag = np.hypot(self.flow[..., 0], self.flow[..., 1])
        if mag.max() == mag.min():                     # "invalid frame" (min-max normalised magnitude sums to 0, :89): keep the previous result
            result = self.prev_result
        self.prev_result = result
        return result
