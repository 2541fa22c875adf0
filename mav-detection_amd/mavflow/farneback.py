"""Farneback -- the reference's stateful wrapper (/root/reference/src/farneback.py:12-107) with
cv2.calcOpticalFlowFarneback replaced by libmavflow's HIP path; the parameters are the reference's literals (:78-80).

`capture` is anything with .read() -> (ok, BGR u8 frame) (cv2.VideoCapture has that shape).  process() keeps the
reference's behaviour -- it returns a BGR visualisation and falls back to the previous one when the frame produces no
flow -- and additionally exposes the dense field as .flow, which the reference computes and throws away."""
from __future__ import annotations

import numpy as np

from . import _lib


def flow_to_hsv(flow: np.ndarray):
    """The HSV image the reference composes from a flow field (farneback.py:83-94), including its quirks: hue = angle / 2
    in degrees truncated to u8, saturation 255, value = 2 x min-max-normalised magnitude stored into a u8 array (numpy
    wraps it modulo 256, so the upper half of the magnitude range folds over), and pixels with value < 1 repainted as
    (127, 255, 255).  cv2.cartToPolar's angle is a ~0.3 degree approximation, so hues can differ by one step from cv2's:
    visualisation only, not a parity surface.  Returns (hsv u8, invalid_frame)."""
    fx, fy = flow[..., 0].astype(np.float32), flow[..., 1].astype(np.float32)
    mag = np.sqrt(fx * fx + fy * fy)
    ang = np.arctan2(fy, fx).astype(np.float32)
    ang = np.where(ang < 0, ang + np.float32(2 * np.pi), ang)
    lo, hi = float(mag.min()), float(mag.max())
    norm = np.zeros_like(mag) if hi - lo <= np.finfo(np.float64).eps else (mag - lo) * np.float32(255.0 / (hi - lo))
    hsv = np.zeros(flow.shape[:2] + (3,), np.uint8)
    hsv[..., 0] = (ang * 180 / np.pi / 2).astype(np.int64) & 255
    hsv[..., 1] = 255
    hsv[..., 2] = (norm * 2.0).astype(np.int64) & 255
    mask = hsv[..., 2] < 1
    hsv[mask, 0] = 127
    hsv[mask, 2] = 255
    return hsv, bool(np.sum((norm * 2.0).astype(np.int64) & 255) < 1)


def hsv_to_bgr(hsv: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(COLOR_HSV2BGR) on u8 (H in [0, 180)): the float sector formula, rounded to nearest."""
    h = hsv[..., 0].astype(np.float32) * np.float32(6.0 / 180.0)
    s = hsv[..., 1].astype(np.float32) / np.float32(255)
    v = hsv[..., 2].astype(np.float32) / np.float32(255)
    sector = np.floor(h)
    f = h - sector
    sector = sector.astype(np.int64) % 6
    tab = np.stack([v, v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))], axis=-1)
    idx = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])   # (b, g, r) picks per sector
    pick = idx[sector]
    bgr = np.take_along_axis(tab, pick, axis=-1)
    return np.clip(np.rint(bgr * 255.0), 0, 255).astype(np.uint8)


def flow_to_bgr(flow: np.ndarray) -> np.ndarray:
    return hsv_to_bgr(flow_to_hsv(flow)[0])


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

    @classmethod
    def from_png_sequence(cls, img_path: str, img_format: str = "image_%05d.png", output=None) -> "Farneback":
        """Farneback(cv2.VideoCapture(f'{img_path}/image_%05d.png'), output) with the capture this build provides for PNG sequences
        (frame_source.PngSequenceCapture; src/datasets/dataset.py:38,57)."""
        from .frame_source import PngSequenceCapture
        cap = PngSequenceCapture(f"{img_path}/{img_format}")
        if not cap.isOpened():
            raise OSError(f"no PNG sequence at {img_path}/{img_format}")
        return cls(cap, output)

    def _gray(self, img: np.ndarray) -> np.ndarray:
        return self.ctx.bgr2gray(img)[0] if img.ndim == 3 else np.ascontiguousarray(img, np.uint8)

    def process(self) -> np.ndarray:
        _, img = self.capture.read()
        gray = self._gray(img)
        self.flow = self.ctx.farneback(self.prevgray, gray)[0]
        self.prevgray = gray
        self.hsv, invalid_frame = flow_to_hsv(self.flow)
        result = hsv_to_bgr(self.hsv)
        if invalid_frame:                              # the normalised magnitude sums to 0 (:89): keep the previous result (:97-98)
            result = self.prev_result
        self.prev_result = result
        return result
