"""The image helpers the hot path names (/root/reference/src/im_helpers.py), on libmavflow where they touch whole frames:
get_magnitude (:150-159), get_simple_bounding_box (:55-84), calculate_tpr_fpr (:244-252), to_int / to_rgb (:162-200),
pyramid / sliding_window (:12-52; the levels above 0 are imutils.resize -> cv2.resize(INTER_AREA) restated on the device,
parity unpinned at the cv2 boundary -- DESIGN.md section 2)."""
from __future__ import annotations

from typing import Iterator, Tuple

import numpy as np

from . import _lib
from .utils import Rectangle

_CTX_CACHE_SIZES = 12                # frame sizes kept at one time (a 1080p image pyramid at scale 1.5 has 9 levels)
_ctx_cache = {}                      # (W, H) -> Context, least recently used first (dicts keep insertion order)


def _ctx(W: int, H: int, batch: int = 1) -> "_lib.Context":
    """A cached context for this frame size.  The reference's helpers are free functions with no state (im_helpers.py:55-84,
    244-252): the cache holds the contexts of the _CTX_CACHE_SIZES most recently used frame sizes.  Such contexts never compute
    flow, so each holds only the staging blocks of its calls (no Farneback workspace: mav_create allocates none) -- a dozen idle
    ones cost a few MB.  A context the cache lets go of (least recently used size, or one replaced by a larger batch) is only
    DROPPED, never closed: a caller may still hold it (pyramid() across its yields, a loop across its frames) and must find it
    usable; the library object is destroyed when the last reference goes (Context.__del__)."""
    key = (W, H)
    c = _ctx_cache.pop(key, None)
    if c is not None and (c.max_batch < batch or not c.h):
        c = None
    if c is None:
        c = _lib.Context(W, H, max(batch, 1))
    _ctx_cache[key] = c              # most recently used last
    while len(_ctx_cache) > _CTX_CACHE_SIZES:
        _ctx_cache.pop(next(iter(_ctx_cache)))
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
    """TPR / FPR of a detection image against a u8 ground truth, the reference's literal counts (:244-252):
    tp = sum(gt * img > 127), fp = sum((255 - gt) * img > 127) over positives = sum(gt > 127), negatives = sum(255 - gt > 127).
    `img` is 255 * mask at the reference's call sites (processor.py:350-351: an int64 image, so gt * img never wraps and
    gt * 255 > 127 <=> gt >= 1); a bool mask multiplies by 1 (gt * img > 127 <=> gt > 127).  Both forms are counted on the
    device as numpy would; anything else (other values, or a uint8 image whose product with gt would wrap) is rejected
    rather than guessed at."""
    gt = np.ascontiguousarray(gt_img, np.uint8)
    mask, value = _mask_and_value(img)
    H, W = gt.shape
    return _rates(_ctx(W, H).tpr_fpr_counts(gt, mask, value)[0])


def _mask_and_value(img: np.ndarray):
    """(u8 0/1 mask, the value its set pixels carry in the reference's product gt * img): bool -> 1, 0/1 -> 1, 255 * mask -> 255.
    Three reductions instead of a sort: all nonzero entries equal the maximum iff sum == count * max (entries are >= 0)."""
    m = np.asarray(img)
    if m.dtype == np.bool_:
        return m.view(np.uint8), 1
    mask = (m != 0).astype(np.uint8)
    n = int(np.count_nonzero(mask))
    if n == 0:
        return mask, 1
    top, low = int(m.max()), int(m.min())
    if low < 0 or top not in (1, 255) or int(m.sum(dtype=np.int64)) != n * top:
        raise ValueError("calculate_tpr_fpr: detection image must be a bool mask, a 0/1 mask or 255 * mask")
    if top == 255 and m.dtype.itemsize == 1:
        raise ValueError("calculate_tpr_fpr: a uint8 255-image would wrap in gt * img; pass 255 * mask (int) as the reference does")
    return mask, top


def _rates(counts):
    pos, neg, tp, fp = (int(v) for v in counts)
    with np.errstate(all="ignore"):
        return (np.float64(tp) / np.float64(pos), np.float64(fp) / np.float64(neg))


def tpr_fpr_of_last_masks(gt_img: np.ndarray, mask_value: int = 255):
    """calculate_tpr_fpr(gt, 255 * estimate_fixed) and calculate_tpr_fpr(gt, 255 * total_mask) (processor.py:350-351) for the
    masks the last detection call on this frame size's context left on the device: ((tpr_fixed, fpr_fixed), (tpr, fpr))."""
    gt = np.ascontiguousarray(gt_img, np.uint8)
    H, W = gt.shape
    cf, cd = _ctx(W, H).last_masks_tpr_fpr(gt, mask_value)
    return _rates(cf[0]), _rates(cd[0])


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
    """Every level of the reference's generator (:12-35): the image itself, then imutils.resize(previous, width=int(w / scale))
    = cv2.resize(INTER_AREA) until a side drops below minSize.  Levels >= 1 come from the device (mav_stage_pyramid_level, the
    same cascade Detector.analyze_pyramid scans), so iterating pyramid() + sliding_window() as the reference does sees exactly
    the images analyze_pyramid scored.  u8 images with 1 or 3 equal channels (what im_helpers.to_rgb produces); the INTER_AREA
    arithmetic is a restatement of OpenCV's, unpinned (no cv2 here)."""
    a = np.asarray(image)
    yield image
    if a.dtype != np.uint8:
        raise TypeError("pyramid: u8 image expected (im_helpers.to_rgb output)")
    if tuple(minSize) != (30, 30):
        raise ValueError("pyramid: only the reference's minSize (30, 30) is implemented")
    gray = a
    if a.ndim == 3:
        if not (np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 0], a[..., 2])):
            raise ValueError("pyramid: 3-channel input must be a gray replica (im_helpers.to_rgb)")
        gray = np.ascontiguousarray(a[..., 0])
    H, W = gray.shape
    ctx = _ctx(W, H)
    for level in range(1, len(ctx.pyramid_dims(scale))):
        lv = ctx.pyramid_level(gray, level, scale)
        yield np.repeat(lv[..., None], a.shape[2], axis=2) if a.ndim == 3 else lv


def sliding_window(image: np.ndarray, stepSize: int, windowSize: Tuple[int, int]):
    for y in range(0, image.shape[0], stepSize):
        for x in range(0, image.shape[1], stepSize):
            yield (x, y, image[y:y + windowSize[1], x:x + windowSize[0]])
