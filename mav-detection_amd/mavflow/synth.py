"""Synthetic frame pairs for tests and bench (numpy only; SURVEY.md section 8(d) recipe).

Closed-form texture  tex(x, y) = 127.5 + A * sum_j a_j sin(2 pi (fx_j x + fy_j y) + phi_j)  so that the second
frame is an *exact* warp of the first (no interpolation): frame1(x, y) = round(tex(x - u, y - v)) with the radial
field (u, v) = k * (x - 0.55 W, y - 0.45 H) plus one 24x24 "MAV" patch moving (6, -3) px.  Both the texture and the
radial warp are separable in x and y, so whole frames are two small matrix products.
"""
from __future__ import annotations

import contextlib

import numpy as np

N_WAVES = 32


@contextlib.contextmanager
def _one_blas_thread():
    """The two matrix products below run on ONE BLAS thread.  Why a frame synthesiser cares: OpenBLAS starts one thread per core it can
    SEE (64 on the 256-core GPU hosts) and its idle threads spin for tens of milliseconds after a product before they sleep; inside a
    container whose cgroup grants a CPU-time quota (16 cores per 100 ms period there) the spinners burn the quota and the kernel then
    freezes EVERY thread of the cgroup until the period ends -- a one-pair GPU call issued 30 - 80 ms after a synthesis took 25 - 80 ms
    instead of 0.3 (profiles/r06/stall_bisect.txt: throttle counters of the cgroup, gone with one BLAS thread).  No-op without threadpoolctl."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        yield
        return
    with threadpool_limits(limits=1, user_api="blas"):
        yield


def _texture_params(rng: np.random.Generator):
    fx = np.empty(N_WAVES)
    fy = np.empty(N_WAVES)
    j = 0
    while j < N_WAVES:
        cx, cy = rng.uniform(-1 / 6, 1 / 6, 2)
        if np.hypot(cx, cy) >= 1 / 64:
            fx[j], fy[j] = cx, cy
            j += 1
    amp = rng.uniform(0.5, 1.0, N_WAVES)
    phase = rng.uniform(0, 2 * np.pi, N_WAVES)
    return fx, fy, amp, phase


def _eval_separable(xs: np.ndarray, ys: np.ndarray, fx, fy, amp, phase) -> np.ndarray:
    """sum_j amp_j sin(2pi fx_j xs + 2pi fy_j ys + phase_j) on the grid ys x xs."""
    ax = 2 * np.pi * np.outer(fx, xs) + phase[:, None]          # (J, W)
    by = 2 * np.pi * np.outer(fy, ys)                            # (J, H)
    left = np.concatenate([np.cos(by) * amp[:, None], np.sin(by) * amp[:, None]], axis=0).T   # (H, 2J)
    right = np.concatenate([np.sin(ax), np.cos(ax)], axis=0)                                  # (2J, W)
    with _one_blas_thread():
        return left @ right


def _eval_points(px: np.ndarray, py: np.ndarray, fx, fy, amp, phase) -> np.ndarray:
    arg = 2 * np.pi * (fx[:, None, None] * px[None] + fy[:, None, None] * py[None]) + phase[:, None, None]
    with _one_blas_thread():
        return np.tensordot(amp, np.sin(arg), axes=1)


def true_flow(W: int, H: int, k: float = 0.01, patch=True) -> np.ndarray:
    """The analytic flow (H, W, 2) float64 that make_pair warps by."""
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    flow = np.empty((H, W, 2))
    flow[..., 0] = (k * (x - 0.55 * W))[None, :]
    flow[..., 1] = (k * (y - 0.45 * H))[:, None]
    if patch:
        x0, y0 = W // 4, H // 4
        flow[y0:y0 + 24, x0:x0 + 24, 0] = 6.0
        flow[y0:y0 + 24, x0:x0 + 24, 1] = -3.0
    return flow


def make_pair(W: int, H: int, pair_index: int = 0, k: float = 0.01, patch: bool = True):
    """Return (frame0 u8 HxW, frame1 u8 HxW, true flow HxWx2 f64) for one synthetic pair."""
    rng = np.random.default_rng(20240 + pair_index)
    fx, fy, amp, phase = _texture_params(rng)
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    t0 = _eval_separable(x, y, fx, fy, amp, phase)
    span = np.abs(t0).max()
    A = 119.5 / span
    f0 = np.rint(127.5 + A * t0).astype(np.uint8)
    xs = x - k * (x - 0.55 * W)
    ys = y - k * (y - 0.45 * H)
    t1 = _eval_separable(xs, ys, fx, fy, amp, phase)
    if patch:
        x0, y0 = W // 4, H // 4
        px, py = np.meshgrid(x[x0:x0 + 24] - 6.0, y[y0:y0 + 24] + 3.0)
        t1[y0:y0 + 24, x0:x0 + 24] = _eval_points(px, py, fx, fy, amp, phase)
    f1 = np.clip(np.rint(127.5 + A * t1), 0, 255).astype(np.uint8)
    return f0, f1, true_flow(W, H, k, patch)


def make_batch(W: int, H: int, batch: int, distinct: int = 4):
    """A batch of pairs: `distinct` freshly generated pairs, the rest cyclic shifts of them (cheap, different content)."""
    prev = np.empty((batch, H, W), np.uint8)
    nxt = np.empty((batch, H, W), np.uint8)
    base = [make_pair(W, H, i)[:2] for i in range(min(distinct, batch))]
    for b in range(batch):
        f0, f1 = base[b % len(base)]
        s = b // len(base)
        if s:
            f0 = np.roll(f0, (7 * s, 13 * s), axis=(0, 1))
            f1 = np.roll(f1, (7 * s, 13 * s), axis=(0, 1))
        prev[b], nxt[b] = f0, f1
    return prev, nxt


def make_sequence(W: int, H: int, n_frames: int, seed: int = 0, k: float = 0.004) -> np.ndarray:
    """A synthetic VIDEO: n_frames u8 (n, H, W) of one texture under a growing radial zoom about (0.55 W, 0.45 H), frame j an
    exact warp of the texture by j * k * (x - cx, y - cy).  Pair i of the sequence is (frame i, frame i + 1): every inner frame
    belongs to two pairs, which is how a video runs through the reference's loop (src/farneback.py:76-80)."""
    rng = np.random.default_rng(20240 + 1000 + seed)
    fx, fy, amp, phase = _texture_params(rng)
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    A = 119.5 / np.abs(_eval_separable(x, y, fx, fy, amp, phase)).max()
    out = np.empty((n_frames, H, W), np.uint8)
    for j in range(n_frames):
        t = _eval_separable(x - j * k * (x - 0.55 * W), y - j * k * (y - 0.45 * H), fx, fy, amp, phase)
        out[j] = np.clip(np.rint(127.5 + A * t), 0, 255).astype(np.uint8)
    return out


def foe_samples(W: int, H: int, pair_index: int = 0, n_pairs: int = 1000) -> np.ndarray:
    """(2N, 2) uint32 (row, col) sample coordinates drawn exactly as focus_of_expansion.py:70-71 does."""
    state = np.random.get_state()
    try:
        np.random.seed(1234 + pair_index)
        out = np.zeros((2 * n_pairs, 2), dtype=np.uint32)
        out[:, 0] = np.random.randint(0, H, 2 * n_pairs)
        out[:, 1] = np.random.randint(0, W, 2 * n_pairs)
    finally:
        np.random.set_state(state)
    return out


def synthetic_flow(W: int, H: int, seed: int = 0, noise: float = 0.05, dtype=np.float32) -> np.ndarray:
    """A radial flow field + patch + noise (for the FoE / phi / mask stages on their own)."""
    rng = np.random.default_rng(777 + seed)
    flow = true_flow(W, H, k=0.02)
    flow += rng.normal(0, noise, flow.shape)
    return flow.astype(dtype)
