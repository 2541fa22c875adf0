"""The image helpers the hot path names (/root/reference/src/im_helpers.py), on libmavflow where they touch whole frames:
get_magnitude (:150-159), get_simple_bounding_box (:55-84), calculate_tpr_fpr (:244-252), to_int / to_rgb (:162-200),
pyramid / sliding_window (:12-52, level 0 only: the upper levels need cv2.resize(INTER_AREA), absent here)."""
from __future__ import annotations

from typing import Iterator, Tuple

import numpy as np

from . import _lib
from .utils import Rectangle

_ctx_cache = {}


def _ctx(W: int, H: int, batch: int = 1) -> "_lib.Context":
    """One cached context per frame size (the reference's helpers are free functions with no state)."""
    key = (W, H)
    c = _ctx_cache.get(key)
    if c is None or c.max_batch < batch:
        c = _lib.Context(W, H, max(batch, 1))
        _ctx_cache[key] = c
    return c


def get_magnitude(img: np.ndarray) -> np.ndarray:
    """|v| along the last axis in the input's float type -- a pure elementwise numpy expression (host glue; the fused
    GPU path computes the magnitude inside its phi kernel and never calls this)."""
    return np.linalg.norm(img, axis=-1)


def get_simple_bounding_box(img: np.ndarray) -> Rectangle:
    """Box around all pixels above 0.1 * max(img); empty -> topleft (-1, -1), size (0, 0).  u8 / bool images run on the GPU."""
    a = np.asarray(img)
    if a.ndim == 3:
        a = a.max(axis=2)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    if a.dtype != np.uint8:
        raise TypeError("get_simple_bounding_box: u8 or bool image expected (the reference applies it to segmentation masks)")
    H, W = a.shape
    return Rectangle.from_box(_ctx(W, H).bbox(a)[0])


def calculate_tpr_fpr(gt_img: np.ndarray, img: np.ndarray) -> Tuple[float, float]:
    """TPR / FPR of a detection image against a u8 ground truth.  `img` is 255 * mask in the reference's call sites
    (processor.py:350-351); any array whose nonzero pixels are 255 (or a bool / 0-1 mask) is accepted."""
    gt = np.ascontiguousarray(gt_img, np.uint8)
    m = np.asarray(img)
    mask = (m != 0).astype(np.uint8)
    if m.dtype != np.bool_ and mask.any() and not np.isin(m[m != 0], (1, 255)).all():
        raise ValueError("calculate_tpr_fpr: detection image must be a 0/1 mask or 255 * mask")
    H, W = gt.shape
    pos, neg, tp, fp = (int(v) for v in _ctx(W, H).tpr_fpr_counts(gt, mask)[0])
    with np.errstate(all="ignore"):
        return (np.float64(tp) / np.float64(pos), np.float64(fp) / np.float64(neg))


def to_int(img: np.ndarray, type: type = np.uint8, normalize: bool = False, max_value: float = None) -> np.ndarray:
    out = img
    if normalize:
        if max_value is None:
            max_value = np.max(img)
        elif max_value <= 0.0:
            max_value = 1.0
        out = np.abs(out) * 255 / max_value
    return np.around(out).astype(type)


def to_rgb(img: np.ndarray, max_value: float = None) -> np.ndarray:
    """Grayscale -> 3 equal u8 channels (what cv2.cvtColor(GRAY2RGB) of the normalised image yields)."""
    g = to_int(img, np.uint8, True, max_value=max_value)
    return np.repeat(g[..., None], 3, axis=2)


def pyramid(image: np.ndarray, scale: float = 1.5, minSize: Tuple[int, int] = (30, 30)) -> Iterator[np.ndarray]:
    """Level 0 only.  The reference's further levels go through imutils.resize -> cv2.resize(INTER_AREA); neither library
    exists here and nothing pins their output, so they are not reproduced."""
    yield image


def sliding_window(image: np.ndarray, stepSize: int, windowSize: Tuple[int, int]):
    for y in range(0, image.shape[0], stepSize):
        for x in range(0, image.shape[1], stepSize):
            yield (x, y, image[y:y + windowSize[1], x:x + windowSize[0]])
