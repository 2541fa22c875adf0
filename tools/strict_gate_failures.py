"""The fuzz frames outside the strict flow gate, one line each (run on the GPU box): what oracle/tolerances.py's two-class gate looks at.
For every pair of the named cases of tools/fuzz_shapes.py that fails the strict gate: the whole frame's mean / p99.9 / max EPE, the share
of unstable pixels (oracle twins >= FLOW_UNSTABLE_S apart, or a border-test flip in the window), how many pixels lie beyond the stable
maximum (0.05 px) and beyond the strict maximum (0.15 px) and how many of THOSE are stable (must be 0), the statistics over the stable
pixels, the worst unstable pixel, and the gate's verdict.
usage: python tools/strict_gate_failures.py seed:case[,seed:case...]      default: the frames of profiles/r06/flow_gate_survey*.txt"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib
from oracle import fb_oracle
from oracle import tolerances as tol
from tools.fuzz_shapes import fuzz_cases

DEFAULT = "123:10,2:65,3:65,4:65,5:43,7:32,5:10"          # seeds 0 - 4 and 123: the survey the gate was written from; 5 - 10: the check after it
names = [tuple(int(v) for v in t.split(":")) for t in (sys.argv[1] if len(sys.argv) > 1 else DEFAULT).split(",")]
orc = fb_oracle.load()
print("seed case pair  frame        | whole frame: mean  p99.9    max     | unstable share | px > 0.05 (stable) | px > 0.15 (stable) |"
      " stable: mean  p99.9    max     | unstable max | gate")
for seed, want in names:
    cs = [c for c in fuzz_cases(want + 1, seed)][want]
    W, H, B, fb, po, prev, nxt = (cs[k] for k in ("W", "H", "B", "fb", "po", "prev", "nxt"))
    with _lib.Context(W, H, B, fb) as c:
        got = c.farneback(prev, nxt)
    for b in range(B):
        e = tol.epe(got[b], orc.calc(prev[b], nxt[b], po))
        if tol.flow_gate(e) is None:
            continue
        ref, _, flips = orc.calc_tracked(prev[b], nxt[b], po)
        un = tol.unstable_mask(ref, orc.twins(prev[b], nxt[b], po), fb.winsize // 2, flips)
        st = e[~un]
        over = [(int((e > t).sum()), int(((e > t) & ~un).sum())) for t in (tol.FLOW_EPE_MAX_STABLE, tol.FLOW_EPE_MAX)]
        print(f"{seed:4d} {want:4d} {b:4d}  {W:4d}x{H:<4d} L{fb.levels} | {e.mean():.2e} {np.percentile(e, 99.9):.2e} {e.max():.2e} |"
              f"   {un.mean():8.4f}     | {over[0][0]:7d} ({over[0][1]})       | {over[1][0]:7d} ({over[1][1]})       |"
              f" {st.mean():.2e} {np.percentile(st, 99.9):.2e} {st.max():.2e} |  {e[un].max():.3e}  | {tol.flow_gate(e, un) or 'passes'}"
              f"   [{100.0 * over[0][0] / e.size:.3f} % of the frame beyond {tol.FLOW_EPE_MAX_STABLE} px]", flush=True)
