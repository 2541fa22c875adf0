"""The ONE statement of the end-to-end flow tolerance (test infrastructure, like everything under oracle/).

north_star: "flow fields match cv2.calcOpticalFlowFarneback on identical inputs to a stated EPE tolerance"; SURVEY 8(d) proposed
mean <= 1e-2 px / p99.9 <= 1e-1 px "tighten after first measurement".  Measured over rounds 1 - 6 against the C restatement
(f32 device sums vs the f32 / f64 mix of OpenCV's CPU path, ten feedback iterations per layer): 1080p mean 2.5e-6 / p99.9 1.4e-4 /
max 3.6e-3; worst of six unfriendly pictures mean 1.4e-5 / p99.9 1.3e-3 / max 2.2e-2; 4K with five layers mean 1.6e-5 / p99.9 1.9e-3 /
max 6.2e-2.

THE STRICT GATE (every pixel of a frame, GPU flow against fb_oracle.calc):

    mean EPE <= 1e-4 px,  p99.9 <= 1e-2 px,  max <= 0.15 px.

Every end-to-end test, bench.py and smoke() run it and nothing else.  The mean and p99.9 gates sit a factor 2 - 7 above the worst
measurement of a well-behaved frame, so a dropped sweep, a wrong border weight or a half-precision intermediate (each moves the mean
by >= 1e-3 px) fails every end-to-end test; a border or tile-seam defect of a single pixel fails the maximum.

WHAT THE STRICT GATE CANNOT HOLD, and why (round 6; profiles/r06/worst_pixel*.txt, border_cycle.txt, flow_gate_survey.txt).  Farneback's
update is an iteration  flow -> M(flow) -> box sums -> 2x2 solve -> flow,  ten sweeps per layer, coarse to fine.  The shape fuzz
(tools/fuzz_shapes.py: ~1 050 frames over six seeds) holds five frames outside the strict gate -- worst 1.49 px (seed 4, case 65,
pair 3) -- and round 5 answered the first of them by raising the maximum to 0.5 px for every pixel of every frame, on the belief
that "an isolated pixel's 2x2 system is nearly singular".  The probes say otherwise, the same way in all five:

  * per SWEEP the kernels agree with the restatement to 1e-6 .. 5e-5 px everywhere (one GPU sweep applied to the oracle's own M, R0,
    R1, every sweep of every layer), and the systems at the worst pixels are well conditioned (det 14 .. 61, the determinant loses
    < 2x to cancellation);
  * the offending pixels form clusters in regions where the iteration DOES NOT SETTLE -- false matches several pixels off the true
    motion in which the oracle's own flow still moves 0.3 .. 3 px per sweep at the end -- so the sweep-to-sweep map expands and a
    rounding difference of 1e-6 px grows 1.5 .. 5x per sweep through the twenty to forty sweeps of the finer layers (seed 4 / case
    65: 1.4e-3 px after layer 2, 3.7e-2 after layer 1, 1.49 after layer 0);
  * a second, rarer mechanism (seed 4, case 21: 0.146 px, just inside the gate): the update has ONE discontinuity -- whether a
    pixel's displaced position still lies inside the second image (bilinear sample of R1) or not -- and along the image border a
    pixel can sit on it in a limit cycle; restatement and GPU then walk the same three-sweep cycle one sweep apart.

The restatement can tell where this happens, from itself alone:

    S      = the largest distance from fb_oracle.calc of its two float32-sums TWINS (fb_oracle.twins: the same C code with its
             window sums and the solve rounded to float32 -- OpenCV sums in double -- once on the frames as they are, once on the
             mirrored frames; NOT OpenCV and never an expected value), as the maximum over the pixel's winsize window;
    flips  = pixels whose inside / outside decision changed in one of the finest layer's last four updates
             (fb_oracle.calc_tracked), dilated by the same window.

    A pixel is UNSTABLE when S >= 0.01 px or a flipped pixel lies in its window: there the restatement cannot say what a float32
    implementation (OpenCV's own SIMD / IPP builds included) returns.

Over the survey's ~1 050 frames every pixel that is NOT unstable is within 0.034 px of the oracle, and every larger error sits on an
unstable pixel (up to 1.49 px).  Hence THE TWO-CLASS GATE, for a caller that hands over twins (and flips) -- the shape fuzz and its
pinned regression cases, nobody else:

    a frame inside the strict gate passes as it is;
    a frame outside it passes if
      (a) over the pixels that are not unstable  mean <= 1e-4 px, p99.9 <= 1e-2 px and max <= 0.05 px  -- so every pixel further off
          than 0.05 px is an unstable one  (measured over the nine frames below: <= 3.2e-5 / 5.2e-3 / 1.8e-2),
      (b) the pixels that USE the exception -- further off than 0.05 px -- are at most 0.5 % of the frame (measured 0.02 - 0.17 %:
          13 - 388 pixels beyond 0.15 px) and none is further off than 4 px (a sanity bound: no accuracy claim there),
      (c) the unstable class is at most 10 % of the frame, so that (a) speaks for at least nine tenths of it (measured 2.3 - 6.0 %).

OUT-OF-SAMPLE CHECK (profiles/r06/flow_gate_survey_oos.txt, strict_gate_failures.txt).  The criteria above were written from seeds
0 - 4 and 123; six further seeds (5 - 10, 1 071 frames) were run afterwards.  Four more frames fail the strict gate (0.18 - 0.35 px), all
four with every off pixel on an unstable one and stable pixels within 9.2e-3 px -- (a) and the sanity bound held unchanged.  What did NOT
hold was the first form of (c), "unstable pixels <= 5 % of the frame" (3.6 % at most in the first six seeds): seed 5 / case 43 has 5.4 %
and 6.0 % in two of its pairs.  The share of unstable pixels is a property of the restatement alone (frames well inside the strict gate
have up to 59 %), not of the GPU's distance from it, so the bound that matters was moved to where the GPU is measured -- (b), new --
and (c) widened to 10 %.  tests/test_gpu_flow.py pins that frame too.

Candidate criteria the survey rejected: the last sweep's determinant or cancellation (the worst pixels have ordinary ones); how far
the oracle's flow moved in its last sweep (5 - 15 % of all pixels move more than 0.02 px there and are perfectly reproducible);
a single twin with a 0.15 px threshold (round 6's first attempt: the twin's perturbation differs from the GPU's, and in an expanding
region their final distances differ by 10x -- seeds 2 and 3 showed 0.42 and 0.58 px where one twin had moved 0.04 and 0.10).

Used by tests/, __graft_entry__.smoke() and bench.py's verification legs; nothing else states a flow gate.
"""
import numpy as np

FLOW_EPE_MEAN = 1e-4           # px, mean end-point error over a frame
FLOW_EPE_P999 = 1e-2           # px, 99.9th percentile
FLOW_EPE_MAX = 0.15            # px, any pixel, strict gate (worst measured on a well-behaved frame: 0.062, 4K / 5 layers)
FLOW_EPE_MAX_STABLE = 0.05     # px, any stable pixel of a frame that needed its unstable pixels excused (worst measured: 0.018; 0.034 over all frames)
FLOW_UNSTABLE_S = 0.01         # px: one of the oracle's own float32-sums twins moves at least this far (window maximum) -> unstable pixel
FLOW_EPE_SANITY_UNSTABLE = 4.0 # px, any unstable pixel: no accuracy claim there, only that nothing is wild
FLOW_UNSTABLE_FRAC = 0.10      # unstable pixels per frame at most, in a frame that fails the strict gate (measured: 2.3 - 6.0 %; 5 % until the out-of-sample seeds)
FLOW_EXCUSED_FRAC = 5e-3       # pixels beyond FLOW_EPE_MAX_STABLE (all of them unstable) per frame at most (measured: 0.02 - 0.17 %)
FLOW_GATE_TEXT = f"mean <= {FLOW_EPE_MEAN:g} px, p99.9 <= {FLOW_EPE_P999:g} px, max <= {FLOW_EPE_MAX:g} px"


def epe(a, b) -> np.ndarray:
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return np.hypot(d[..., 0], d[..., 1])


def _window_max(a, radius):
    """maximum over the (2 radius + 1)^2 neighbourhood, edges replicated"""
    for axis in (0, 1):
        n = a.shape[axis]
        p = np.take(a, np.clip(np.arange(-radius, n + radius), 0, n - 1), axis=axis)
        out = a.copy()
        for d in range(2 * radius + 1):
            np.maximum(out, np.take(p, np.arange(d, d + n), axis=axis), out=out)
        a = out
    return a


def sensitivity(exp, twins, radius=6) -> np.ndarray:
    """S of the module docstring: the largest |exp - twin| over the oracle's twins (fb_oracle.twins: one array or a list), as the
    maximum over the (2 radius + 1)^2 window a pixel's 2x2 system is summed over (radius = winsize // 2)."""
    tw = twins if isinstance(twins, (list, tuple)) else [twins]
    d = epe(tw[0], exp)
    for t in tw[1:]:
        np.maximum(d, epe(t, exp), out=d)
    return _window_max(d, radius)


def unstable_mask(exp, twins, radius=6, flips=None) -> np.ndarray:
    """The pixels at which the restatement itself is not reproducible: S >= FLOW_UNSTABLE_S, or -- with `flips` from
    fb_oracle.calc_tracked -- a pixel of the window changed sides of the image-border test in one of the last four updates."""
    m = sensitivity(exp, twins, radius) >= FLOW_UNSTABLE_S
    if flips is not None:
        m |= _window_max((np.asarray(flips) > 0).astype(np.float64), radius) > 0
    return m


def last_step(exp, record, radius=6):
    """How far the ORACLE's own flow still moved in the finest layer's last sweep (px), per pixel, as the maximum over the
    (2 radius + 1)^2 window.  `record` = fb_oracle.calc(want_sys=True)[1].  Diagnostic (tools/worst_pixel.py), not a gate."""
    r = np.asarray(record, np.float64)
    d = np.hypot(np.asarray(exp, np.float64)[..., 0] - r[..., 5], np.asarray(exp, np.float64)[..., 1] - r[..., 6])
    return _window_max(np.where(np.isnan(d), 0.0, d), radius)


def conditioning(sys):
    """(det, cancel) of the oracle's 2x2 systems (..., >= 5) = (g11, g12, g22, h1, h2, ...) (fb_oracle.calc(want_sys=True)):
    det = g11 g22 - g12^2 and cancel = (g11 g22 + g12^2) / (|det| + 1e-3), the factor by which a relative rounding error of the
    window sums grows in the solve's denominator det + 1e-3.  Diagnostic (tools/worst_pixel.py), not a gate: the worst pixels
    measured are well conditioned."""
    s = np.asarray(sys, np.float64)
    a, b = s[..., 0] * s[..., 2], s[..., 1] * s[..., 1]
    det = a - b
    cancel = (a + b) / (np.abs(det) + 1e-3)
    return np.where(np.isnan(det), np.inf, det), np.where(np.isnan(cancel), 1.0, cancel)


def _strict(e, tag="", max_px=None):
    if e.mean() > FLOW_EPE_MEAN:
        return "mean EPE" + tag
    if np.percentile(e, 99.9) > FLOW_EPE_P999:
        return "p99.9 EPE" + tag
    if e.max() > (FLOW_EPE_MAX if max_px is None else max_px):
        return "max EPE" + tag
    return None


def flow_gate(e, unstable=None):
    """None when the end-point-error field `e` (H, W) is inside the gate, else the name of the first gate it fails.
    `unstable`: boolean (H, W) from unstable_mask(), or None = no pixel is excused.  A frame inside the strict gate passes as it
    is; only a frame that fails it is looked at again with its unstable pixels set aside -- those may be at most FLOW_UNSTABLE_FRAC
    of the frame and nothing wild, the pixels beyond FLOW_EPE_MAX_STABLE at most FLOW_EXCUSED_FRAC of it, and every statistic of the
    strict gate (with that tighter maximum) must hold over the rest."""
    e = np.asarray(e)
    if not np.isfinite(e).all():
        return "non-finite EPE"
    failed = _strict(e)
    if failed is None or unstable is None or not unstable.any():
        return failed
    if unstable.mean() > FLOW_UNSTABLE_FRAC:
        return f"{failed}; too many unstable pixels to excuse"
    if e[unstable].max() > FLOW_EPE_SANITY_UNSTABLE:
        return "max EPE (unstable pixels)"
    if (e > FLOW_EPE_MAX_STABLE).mean() > FLOW_EXCUSED_FRAC:
        return f"{failed}; too many pixels beyond {FLOW_EPE_MAX_STABLE:g} px"
    return _strict(e[~unstable], " (stable pixels)", FLOW_EPE_MAX_STABLE) if (~unstable).any() else None


def flow_epe_ok(e, unstable=None) -> bool:
    """True when an end-point-error field is inside the gate (NaN anywhere fails)."""
    return flow_gate(e, unstable) is None


def check_flow(got, exp, tag="", twins=None, radius=6, flips=None) -> np.ndarray:
    """assert `got` (H, W, 2) matches `exp` inside the gate; returns the EPE field.  `twins` = fb_oracle.twins of the same frames (and
    `flips` from fb_oracle.calc_tracked) switch the unstable-pixel class on (radius = winsize // 2); without them every pixel is held
    to the strict gate."""
    assert np.isfinite(np.asarray(got)).all(), (tag, "non-finite flow")
    e = epe(got, exp)
    unstable = None if twins is None else unstable_mask(exp, twins, radius, flips)
    failed = flow_gate(e, unstable)
    stats = (tag, float(e.mean()), float(np.percentile(e, 99.9)), float(e.max()))
    if unstable is not None:
        stats += (f"{int(unstable.sum())} unstable pixels", float(e[~unstable].max()) if (~unstable).any() else 0.0)
    assert failed is None, (failed,) + stats
    return e
