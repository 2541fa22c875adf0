"""Window search of Detector on the GPU (analyze_pyramid over all levels, optimize_window) against
oracle/pyramid_oracle.py and the reference-generated fixtures in tests/golden/window_search.npz.
Integer / byte work: everything is compared bit for bit.  The INTER_AREA levels are parity-unpinned at the cv2 boundary
(see the oracle's header); optimize_window and the level-0 scan are pinned by the fixtures."""
import os

import numpy as np
import pytest

from oracle import pyramid_oracle as po

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "window_search.npz"))
CASES = [str(c) for c in G["cases"]]


def _ctx(W, H, B=1):
    from mavflow import _lib
    return _lib.Context(W, H, B)


def _blobs(W, H, seed, n=3):
    rng = np.random.default_rng(seed)
    img = (rng.integers(0, 30, (H, W)) * (rng.random((H, W)) > 0.98)).astype(np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(n):
        cx, cy, rad, amp = rng.integers(0, W), rng.integers(0, H), rng.uniform(3, 40), rng.integers(60, 256)
        img = np.maximum(img, (amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * rad * rad))).astype(np.uint8))
    return img


@pytest.mark.parametrize("size", [(640, 480), (1920, 1080), (333, 217), (100, 70)])
def test_pyramid_dims_and_levels_bit_exact(size):
    W, H = size
    img = np.random.default_rng(W).integers(0, 256, (H, W)).astype(np.uint8)
    with _ctx(W, H) as ctx:
        assert ctx.pyramid_dims() == po.pyramid_dims(W, H)
        levels = po.pyramid(img)
        for l, exp in enumerate(levels):
            got = ctx.pyramid_level(img, l)
            assert got.shape == exp.shape and np.array_equal(got, exp), (size, l, int(np.abs(got.astype(int) - exp).max()))


def test_pyramid_other_scale_and_errors():
    W, H = 640, 480
    img = _blobs(W, H, 5)
    with _ctx(W, H) as ctx:
        assert ctx.pyramid_dims(1.3) == po.pyramid_dims(W, H, 1.3)
        for l, exp in enumerate(po.pyramid(img, 1.3)):
            assert np.array_equal(ctx.pyramid_level(img, l, 1.3), exp), l
        with pytest.raises(ValueError):
            ctx.analyze_pyramid(img, 2.0)            # 640x480 -> 320x240: OpenCV's integer-ratio path, not restated
        with pytest.raises(ValueError):
            ctx.analyze_pyramid(img, 1.0)
        with pytest.raises(ValueError):
            ctx.pyramid_level(img, 40)


def test_analyze_pyramid_matches_oracle_batch():
    W, H, B = 640, 480, 6
    imgs = np.stack([_blobs(W, H, 100 + b) for b in range(B)])
    imgs[2] = 0                                       # no positive window -> all zeros
    # four dots 70 px apart: a level-0 window holds one of them, the level-1 window at (96, 96) holds all four -> level 1 wins
    imgs[3] = 0
    for (cx, cy) in ((150, 150), (220, 150), (150, 220), (220, 220)):
        imgs[3, cy:cy + 6, cx:cx + 6] = 255
    with _ctx(W, H, B) as ctx:
        got = ctx.analyze_pyramid(imgs)
    for b in range(B):
        assert tuple(int(v) for v in got[b]) == po.analyze_pyramid(imgs[b]), b
    assert tuple(got[2]) == (0, 0, 0, 0, 0, 0)
    assert tuple(got[3][1:4]) == (96, 96, 1)


def test_analyze_pyramid_fullsize():
    W, H = 1920, 1080
    imgs = np.stack([_blobs(W, H, 7, n=6), _blobs(W, H, 8, n=1)])
    with _ctx(W, H, 2) as ctx:
        got = ctx.analyze_pyramid(imgs)
    for b in range(2):
        assert tuple(int(v) for v in got[b]) == po.analyze_pyramid(imgs[b]), b


@pytest.mark.parametrize("tag", CASES)
def test_optimize_window_matches_reference_fixture(tag):
    img, win, exp = G[f"opt_{tag}_img"], G[f"opt_{tag}_in"], G[f"opt_{tag}_out"]
    with _ctx(img.shape[1], img.shape[0]) as ctx:
        score, out = ctx.optimize_window(img, [win])
    assert float(score[0]) == exp[0] and tuple(int(v) for v in out[0]) == tuple(int(v) for v in exp[1:]), (tag, score, out, exp)


@pytest.mark.parametrize("tag", CASES)
def test_level0_scan_matches_reference_fixture(tag):
    img = G[f"opt_{tag}_img"]
    with _ctx(img.shape[1], img.shape[0]) as ctx:
        wm = ctx.window_max(img)[0]
        full = ctx.analyze_pyramid(img)[0]
    exp = G[f"scan_{tag}"]
    assert tuple(int(v) for v in wm) == tuple(int(v) for v in exp[:3])
    if full[3] == 0:
        assert tuple(int(v) for v in full[[0, 1, 2, 4, 5]]) == tuple(int(v) for v in exp)


def test_optimize_window_batch_against_oracle():
    W, H, B = 640, 480, 5
    rng = np.random.default_rng(3)
    imgs = np.stack([_blobs(W, H, 40 + b, n=2) for b in range(B)])
    wins = np.stack([[rng.integers(-5, W - 70), rng.integers(-5, H - 70), 64, 64] for _ in range(B)]).astype(np.int32)
    wins[1] = [W - 30, H - 30, 64, 64]                # sticks out of the image: slices clip
    wins[2] = [0, 0, 8, 8]
    with _ctx(W, H, B) as ctx:
        score, out = ctx.optimize_window(imgs, wins)
    for b in range(B):
        es, ew = po.optimize_window(imgs[b], wins[b])
        assert (int(score[b]), tuple(int(v) for v in out[b])) == (es, ew), b


def test_detector_window_search_shim():
    from mavflow import detector, utils

    class DS:
        capture_size = (640, 480)
    np.random.seed(0)
    d = detector.Detector(DS())
    gray = _blobs(640, 480, 77)
    rgb = np.repeat(gray[..., None], 3, axis=2)
    score, rect, window, amax = d.analyze_pyramid(rgb)
    es, ex, ey, lv, ay, ax = po.analyze_pyramid(gray)
    assert (score, rect.topleft, rect.size) == (es, (ex, ey), (64, 64)) and window.shape == (64, 64, 3)
    assert amax == (ay, ax, 0) and window[ay, ax, 0] == window.max()
    s2, r2 = d.optimize_window(rgb, rect) if lv == 0 else d.optimize_window(rgb, utils.Rectangle((ex, ey), (64, 64)))
    os_, ow = po.optimize_window(gray, (ex, ey, 64, 64))
    assert isinstance(s2, float) and s2 == float(os_) and (r2.topleft, r2.size) == ((ow[0], ow[1]), (ow[2], ow[3]))
    z, rz = d.optimize_window(np.zeros((480, 640, 3), np.uint8), rect)
    assert z == 0.0 and rz is rect
