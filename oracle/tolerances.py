"""The ONE statement of the end-to-end flow tolerance (test infrastructure, like everything under oracle/).

north_star: "flow fields match cv2.calcOpticalFlowFarneback on identical inputs to a stated EPE tolerance"; SURVEY 8(d) proposed
mean <= 1e-2 px / p99.9 <= 1e-1 px "tighten after first measurement".  Measured over rounds 1 - 4 against the C restatement
(f32 device sums vs the f32 / f64 mix of OpenCV's CPU path, ten feedback iterations per layer): 1080p mean 2.5e-6 / p99.9 1.4e-4 /
max 3.6e-3; worst of six unfriendly pictures mean 1.4e-5 / p99.9 1.3e-3 / max 2.2e-2; 4K with five layers mean 1.6e-5 / p99.9 1.9e-3 /
max 6.5e-2.  Round 5's extended shape fuzz (tools/fuzz_shapes.py, 140 cases over two seeds) found the worst single pixel: 0.269 px on
a 1048x925 frame with a four-layer pyramid (seed 123, case 10: mean 4.0e-5, p99.9 3.8e-3) -- an isolated pixel whose 2x2 system is
nearly singular, where float32 and float64 sums part; the 0.15 px first set for the maximum was too tight for that and is 0.5 now.
The mean and p99.9 gates sit a factor 2 - 7 above the worst measurement, so a dropped sweep, a wrong border weight or a
half-precision intermediate (each moves the mean by >= 1e-3 px) fails every end-to-end test.

Used by tests/, __graft_entry__.smoke() and bench.py's verification legs; nothing else states a flow gate.
"""
import numpy as np

FLOW_EPE_MEAN = 1e-4      # px, mean end-point error over a frame
FLOW_EPE_P999 = 1e-2      # px, 99.9th percentile
FLOW_EPE_MAX = 0.5        # px, any single pixel (worst measured: 0.269)
FLOW_GATE_TEXT = f"mean <= {FLOW_EPE_MEAN:g} px, p99.9 <= {FLOW_EPE_P999:g} px, max <= {FLOW_EPE_MAX:g} px"


def epe(a, b) -> np.ndarray:
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return np.hypot(d[..., 0], d[..., 1])


def flow_epe_ok(e) -> bool:
    """True when an end-point-error field is inside the gate (NaN anywhere fails)."""
    e = np.asarray(e)
    return bool(np.isfinite(e).all() and e.mean() <= FLOW_EPE_MEAN and np.percentile(e, 99.9) <= FLOW_EPE_P999 and e.max() <= FLOW_EPE_MAX)


def check_flow(got, exp, tag="") -> np.ndarray:
    """assert `got` (..., 2) matches `exp` inside the gate; returns the EPE field."""
    assert np.isfinite(np.asarray(got)).all(), (tag, "non-finite flow")
    e = epe(got, exp)
    stats = (tag, float(e.mean()), float(np.percentile(e, 99.9)), float(e.max()))
    assert e.mean() <= FLOW_EPE_MEAN, ("mean EPE",) + stats
    assert np.percentile(e, 99.9) <= FLOW_EPE_P999, ("p99.9 EPE",) + stats
    assert e.max() <= FLOW_EPE_MAX, ("max EPE",) + stats
    return e
