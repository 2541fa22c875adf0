"""TEST INFRASTRUCTURE -- CPU restatement of the window search of Detector.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package; the product never does.

  analyze_pyramid   src/detector.py:280-312  with pyramid() / sliding_window()  src/im_helpers.py:12-52
  optimize_window   src/detector.py:314-358

Pinning status
  * optimize_window and the level-0 window scan are pure numpy in the reference: PINNED bit for bit by
    tests/golden/window_search.npz (tools/gen_golden.py runs the reference's own Detector.optimize_window and
    im_helpers.sliding_window).
  * pyramid levels >= 1 go through imutils.resize -> cv2.resize(INTER_AREA).  Neither imutils nor opencv-python is in the
    reference tree or in this image (both unpinned in requirements.txt:1,4) and the reference holds no fixture for the
    call: PARITY UNPINNED.  The restatement follows the published algorithms:
      - imutils.resize (imutils/convenience.py, 0.5.x): r = width / float(w); dim = (width, int(h * r));
        cv2.resize(image, dim, interpolation=cv2.INTER_AREA)
      - OpenCV 4.x modules/imgproc/src/resize.cpp: scale = 1 / (dsize / ssize) in double; the non-integer-ratio area path
        (computeResizeAreaTab + resizeArea_<uchar, float>): float weights, float accumulation in tab order (x taps
        first, then rows), saturate_cast<uchar> = round-half-even.  Builds with FMA contraction (aarch64) may differ in
        the last bit before rounding; x86-64 wheels do not contract.
"""
from __future__ import annotations

import math

import numpy as np

WIN = 64
STEP = 16


# ---- cv2.resize(INTER_AREA), general (non-integer ratio) path --------------------------------------------------------
def area_tab(ssize: int, dsize: int, scale: float):
    """computeResizeAreaTab (resize.cpp): list of (di, si, alpha as float32) in table order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = math.ceil(fsx1)
        sx2 = math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _ranked(tab):
    """Split a tab into passes: pass r holds the r-th entry of every destination index (order of accumulation)."""
    passes, seen = [], {}
    for di, si, a in tab:
        r = seen.get(di, 0)
        seen[di] = r + 1
        while len(passes) <= r:
            passes.append(([], [], []))
        passes[r][0].append(di); passes[r][1].append(si); passes[r][2].append(a)
    return [(np.array(d), np.array(s), np.array(a, np.float32)) for d, s, a in passes]


def is_area_fast(ssize: int, dsize: int) -> bool:
    scale = 1.0 / (dsize / ssize)
    iscale = int(scale)
    return abs(scale - iscale) < np.finfo(np.float64).eps


def resize_area_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_AREA) for a shrinking, non-integer ratio, 1-channel u8."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    sh, sw = img.shape
    scale_x = 1.0 / (dw / sw)
    scale_y = 1.0 / (dh / sh)
    assert scale_x >= 1 and scale_y >= 1, "INTER_AREA enlarging falls back to a linear path (not restated)"
    assert not (is_area_fast(sw, dw) and is_area_fast(sh, dh)), "integer-ratio fast path not restated"
    S = img.astype(np.float32)
    buf = np.zeros((sh, dw), np.float32)
    for di, si, a in _ranked(area_tab(sw, dw, scale_x)):
        buf[:, di] = buf[:, di] + S[:, si] * a[None, :]
    acc = np.zeros((dh, dw), np.float32)
    for di, si, b in _ranked(area_tab(sh, dh, scale_y)):
        acc[di, :] = acc[di, :] + b[:, None] * buf[si, :]
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def next_level_dims(w: int, h: int, scale: float = 1.5):
    """im_helpers.pyramid :28 + imutils.resize: w' = int(w / scale); r = w' / float(w); h' = int(h * r)."""
    wn = int(w / scale)
    r = wn / float(w)
    return wn, int(h * r)


def pyramid_dims(w: int, h: int, scale: float = 1.5, min_size=(30, 30)):
    dims = [(w, h)]
    while True:
        w, h = next_level_dims(w, h, scale)
        if h < min_size[1] or w < min_size[0]:
            break
        dims.append((w, h))
    return dims


def pyramid(img: np.ndarray, scale: float = 1.5, min_size=(30, 30)):
    """im_helpers.pyramid (:12-35): level l is the INTER_AREA shrink of level l-1 (cascaded)."""
    img = np.asarray(img)
    levels = [img]
    for (w, h) in pyramid_dims(img.shape[1], img.shape[0], scale, min_size)[1:]:
        img = resize_area_u8(img, w, h)
        levels.append(img)
    return levels


# ---- analyze_pyramid -------------------------------------------------------------------------------------------------
def _scan(img: np.ndarray):
    """sliding_window (:38-52) + the loop body of analyze_pyramid on one level: list of (score, x, y) of complete windows
    in scan order; score = sum over the 3 equal channels."""
    H, W = img.shape
    ii = np.zeros((H + 1, W + 1), dtype=np.int64)
    ii[1:, 1:] = np.cumsum(np.cumsum(img.astype(np.int64), axis=0), axis=1)
    for y in range(0, H, STEP):
        if y + WIN > H:
            continue
        for x in range(0, W, STEP):
            if x + WIN > W:
                continue
            yield 3 * int(ii[y + WIN, x + WIN] - ii[y, x + WIN] - ii[y + WIN, x] + ii[y, x]), x, y


def analyze_pyramid(img_u8: np.ndarray, scale: float = 1.5):
    """(score, x, y, level, argmax_row, argmax_col): first window, levels scanned from 0 upwards, with the strictly
    largest sum; x, y are in that level's own coordinates (the reference does not rescale them).  All zero when no
    window has a positive sum.  argmax = np.unravel_index(window.argmax(), window.shape)[:2] (channel index is 0)."""
    best = (0, 0, 0, 0, 0, 0)
    for lv, im in enumerate(pyramid(img_u8, scale)):
        for s, x, y in _scan(im):
            if best[0] < s:
                w = im[y:y + WIN, x:x + WIN]
                ay, ax = np.unravel_index(w.argmax(), w.shape)
                best = (s, x, y, lv, int(ay), int(ax))
    return best


# ---- optimize_window -------------------------------------------------------------------------------------------------
def _slice_sum(ii: np.ndarray, H: int, W: int, top: int, bottom: int, left: int, right: int) -> int:
    """3 * sum(img[top:bottom, left:right]) with Python's slice rules (negative indices wrap, ends clip)."""
    def norm(a, n):
        if a < 0:
            a += n
        return min(max(a, 0), n)
    t, b, l, r = norm(top, H), norm(bottom, H), norm(left, W), norm(right, W)
    if t >= b or l >= r:
        return 0
    return 3 * int(ii[b, r] - ii[t, r] - ii[b, l] + ii[t, l])


def optimize_window(img_u8: np.ndarray, window):
    """Detector.optimize_window on the 3-channel replica of a u8 image.  window = (x, y, w, h) integers.
    Returns (score, (x, y, w, h)); (0, window) when no neighbour has a positive sum."""
    img = np.asarray(img_u8)
    H, W = img.shape
    ii = np.zeros((H + 1, W + 1), dtype=np.int64)
    ii[1:, 1:] = np.cumsum(np.cumsum(img.astype(np.int64), axis=0), axis=1)
    x, y, w, h = (int(v) for v in window)
    res_score, res = 0, (x, y, x + w, y + h)                 # left, top, right, bottom
    while True:
        l, t, r, b = res
        best_s, best = 0, res
        for hh in (0, 1):
            for i in (-1, 1):
                for j in (-1, 1):
                    cand = (l + i, t + j, r, b) if hh == 0 else (l, t, r + i, b + j)
                    s = _slice_sum(ii, H, W, cand[1], cand[3], cand[0], cand[2])
                    if s > best_s:
                        best_s, best = s, cand
        if best_s <= res_score:
            break
        res_score, res = best_s, best
    l, t, r, b = res
    return res_score, (l, t, r - l, b - t)
