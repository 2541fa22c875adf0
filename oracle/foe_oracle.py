"""CPU restatement (numpy, IEEE double, reference operation order) of the post-flow chain. TEST INFRASTRUCTURE ONLY.

This is the *oracle* for derotation, the FoE fit, phi, the threshold masks and the boxes.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path (mavflow + libmavflow.so)
never does.

Pinned: every function here is checked bit-for-bit against fixtures under tests/golden/ that were produced by
running the reference's own Python (tools/gen_golden.py imports /root/reference/src in this container) -- see
tests/test_oracle_golden.py.  Citations are relative to /root/reference/.
"""
from __future__ import annotations

import numpy as np


def get_magnitude(v: np.ndarray) -> np.ndarray:
    """src/im_helpers.py:150-159 -- np.linalg.norm(axis=-1) == sqrt(x*x + y*y) in the input's float type."""
    v = np.asarray(v)
    if not np.issubdtype(v.dtype, np.inexact):
        v = v.astype(np.float64)
    return np.sqrt(v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1])


def derotate(flow: np.ndarray, omega, dt: float, current_frame_index: int = 1) -> np.ndarray:
    """src/detector.py:70-117.  omega = angular difference / dt (3 doubles); returns float64 (H, W, 2).

    For current_frame_index < 1 the reference returns its input unchanged (:80-81)."""
    if current_frame_index < 1:
        return flow
    h, w = flow.shape[:2]
    xs = np.tile(np.arange(w), (h, 1))
    ys = np.tile(np.arange(h), (w, 1)).T
    x = -(xs / w - 0.5) * 2.0
    y = -(ys / h - 0.5) * 2.0
    o0, o1, o2 = (np.float64(o) for o in omega)
    du = +o0 * x * y - o1 * x ** 2 - o1 + o2 * y
    dv = -o2 * x + o0 + o0 * y ** 2 - o1 * x * y
    du = du * (w * dt / 2)
    dv = dv * (h * dt / 2)
    return flow - np.stack([du, dv], axis=-1)


def line_intersections(flow: np.ndarray, samples: np.ndarray, mag_threshold: float = 2.5) -> np.ndarray:
    """Vectorised body of get_FOE_dense's loop (src/focus_of_expansion.py:74-83) with utils.line_intersection
    (src/utils.py:183-197).  samples: (2N, 2) uint32 (row, col).  Returns the (N, 2) float64 `intersections`
    array *before* the x != 0.0 filter; skipped / parallel pairs stay (0, 0)."""
    samples = np.asarray(samples)
    N = samples.shape[0] // 2
    r1, c1 = samples[:N, 0], samples[:N, 1]
    r2, c2 = samples[N:2 * N, 0], samples[N:2 * N, 1]
    f1 = flow[r1, c1, :]
    f2 = flow[r2, c2, :]
    keep = ~(get_magnitude(f2) < mag_threshold)            # `if mag < thr: continue`
    p1x, p1y = c1.astype(np.float64), r1.astype(np.float64)
    p2x, p2y = c2.astype(np.float64), r2.astype(np.float64)
    q1x = f1[:, 0].astype(np.float64) + p1x               # float32 + uint32 -> float64 in numpy
    q1y = f1[:, 1].astype(np.float64) + p1y
    q2x = f2[:, 0].astype(np.float64) + p2x
    q2y = f2[:, 1].astype(np.float64) + p2y
    xd0, xd1 = p1x - q1x, p2x - q2x
    yd0, yd1 = p1y - q1y, p2y - q2y
    div = xd0 * yd1 - xd1 * yd0
    d0 = p1x * q1y - p1y * q1x
    d1 = p2x * q2y - p2y * q2x
    with np.errstate(all="ignore"):
        x = (d0 * xd1 - d1 * xd0) / div
        y = (d0 * yd1 - d1 * yd0) / div
    ok = keep & ~(div == 0)
    out = np.zeros((N, 2))
    out[ok, 0] = x[ok]
    out[ok, 1] = y[ok]
    return out


def ransac(estimates: np.ndarray, ransac_threshold: float = 30.0):
    """src/focus_of_expansion.py:32-54: first candidate with the strictly largest inlier count (count > 0)."""
    est = np.asarray(estimates, dtype=np.float64).reshape(-1, 2)
    if est.shape[0] == 0:
        return (0.0, 0.0)
    dx = est[:, None, 0] - est[None, :, 0]
    dy = est[:, None, 1] - est[None, :, 1]
    with np.errstate(all="ignore"):
        dist = np.sqrt(dx * dx + dy * dy)
    score = (dist < ransac_threshold).sum(axis=1) - 1
    best = int(np.argmax(score))                            # argmax returns the first maximum
    if score[best] > 0:
        return (float(est[best, 0]), float(est[best, 1]))
    return (0.0, 0.0)


def get_foe_dense(flow: np.ndarray, samples: np.ndarray, mag_threshold: float = 2.5, ransac_threshold: float = 30.0):
    """src/focus_of_expansion.py:56-86 with the random draws passed in as `samples`."""
    inter = line_intersections(flow, samples, mag_threshold)
    inter = inter[inter[:, 0] != 0.0, :]
    return ransac(inter, ransac_threshold)


def get_phi(flow: np.ndarray, foe) -> np.ndarray:
    """src/focus_of_expansion.py:150-184 (degrees; dtype follows the flow dtype exactly as zeros_like does)."""
    if foe[0] is np.nan:
        return np.zeros(0)
    h, w = flow.shape[:2]
    xs = np.tile(np.arange(w), (h, 1))
    ys = np.tile(np.arange(h), (w, 1)).T
    d2 = np.zeros_like(flow)
    d2[..., 0] = xs - foe[0]
    d2[..., 1] = ys - foe[1]
    fm = get_magnitude(flow)
    dist = get_magnitude(d2)
    with np.errstate(all="ignore"):
        norm = np.maximum(np.ones_like(fm) * 1e-6, fm * dist)
        arg = (flow[..., 0] * d2[..., 0] + flow[..., 1] * d2[..., 1]) / norm
        arg = np.clip(arg, -1, 1)
        ang = np.arccos(arg)
    ang[np.isnan(ang)] = 0
    return np.rad2deg(ang)


def threshold_masks(phi: np.ndarray, mag: np.ndarray, sky: np.ndarray | None = None,
                    fixed_deg=15, fixed_min_mag=1.0, dyn_min_mag=0.5, dyn_a=0.25, dyn_b=0.5, dyn_c=8):
    """src/processor.py:333-341.  Returns (estimate_fixed, total_mask) as bool arrays."""
    if sky is None:
        sky = np.zeros(phi.shape, dtype=bool)
    with np.errstate(all="ignore"):
        hi = phi > (dyn_a + (dyn_b + dyn_c / mag))
        lo = phi < (dyn_a - (dyn_b + dyn_c / mag))
    ang = np.logical_or(lo, hi)
    total = (mag > dyn_min_mag) * ~sky * ang
    fixed = phi * (mag > fixed_min_mag) * ~sky > fixed_deg
    return fixed, total


def simple_bounding_box(img: np.ndarray):
    """src/im_helpers.py:55-84.  Returns (x0, y0, x1, y1) inclusive; (-1, -1, -1, -1) when nothing is set.
    (The reference's Rectangle is topleft=(x0, y0), size=(x1 - x0, y1 - y0).)"""
    img = np.asarray(img)
    thr = 0.1 * np.max(img)
    mask = img > thr
    if mask.ndim == 3:
        mask = mask.any(axis=2)
    rows = np.flatnonzero(mask.any(axis=1))
    cols = np.flatnonzero(mask.any(axis=0))
    if rows.size == 0:
        return (-1, -1, -1, -1)
    return (int(cols[0]), int(rows[0]), int(cols[-1]), int(rows[-1]))


def calculate_tpr_fpr(gt: np.ndarray, img: np.ndarray):
    """src/im_helpers.py:244-252 (dtype promotion left to numpy exactly as the reference does)."""
    positives = np.sum(gt > 127)
    negatives = np.sum((255 - gt) > 127)
    tp = np.sum((gt * img) > 127)
    fp = np.sum(((255 - gt) * img) > 127)
    with np.errstate(all="ignore"):
        return (tp / positives, fp / negatives)


def analyze_pyramid_level0(img_u8: np.ndarray, win: int = 64, step: int = 16):
    """Level 0 of src/detector.py:280-312 on a 1-channel u8 image replicated to 3 channels by to_rgb
    (src/im_helpers.py:162-173): first window (row-major scan) with the strictly largest sum.
    Returns (score, x, y); (0, 0, 0) when every window sums to 0."""
    img = np.asarray(img_u8)
    H, W = img.shape[:2]
    ii = np.zeros((H + 1, W + 1), dtype=np.int64)
    ii[1:, 1:] = np.cumsum(np.cumsum(img.astype(np.int64), axis=0), axis=1)
    best = (0, 0, 0)
    for y in range(0, H, step):
        if y + win > H:
            continue
        for x in range(0, W, step):
            if x + win > W:
                continue
            s = 3 * int(ii[y + win, x + win] - ii[y, x + win] - ii[y + win, x] + ii[y, x])
            if best[0] < s:
                best = (s, x, y)
    return best


def run_chain(flow_f32: np.ndarray, samples: np.ndarray, omega=(0.0, 0.0, 0.0), dt: float = 1.0,
              sky: np.ndarray | None = None, current_frame_index: int = 1):
    """The order of operations of src/processor.py:305-341 on one flow field."""
    der = derotate(flow_f32, omega, dt, current_frame_index)
    mag = get_magnitude(der)
    foe = get_foe_dense(der, samples)
    phi = get_phi(der, foe)
    fixed, total = threshold_masks(phi, mag, sky)
    box = simple_bounding_box(fixed)
    return dict(foe=foe, phi=phi, mag=mag, fixed=fixed, total=total, box=box)
