"""End-point error of the GPU flow against the oracle's own reproducibility (run on the GPU box): the measurement behind
oracle/tolerances.py's two-class gate.  Workloads: the timed configurations (720p, 1080p, 4K / 5 layers), the six unfriendly 1080p pictures
of tests/test_gpu_content.py, and tools/fuzz_shapes.py's cases for several seeds.  Per frame: S = the largest distance of the oracle's two
float32-sums twins (fb_oracle.twins) from the oracle, as a maximum over the winsize window; EPE of the GPU flow in bins of S; and for
candidate thresholds of S the share of "unstable" pixels, the statistics over the STABLE pixels and the verdicts.
usage: python tools/flow_gate_survey.py [n_fuzz_cases] [seed,seed,...]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd"); sys.path.insert(0, "tests")
import numpy as np
from mavflow import _lib, synth
from oracle import fb_oracle
from oracle import tolerances as tol
from tools.fuzz_shapes import fuzz_cases

n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 80
seeds = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 123, 1, 2, 3, 4]
orc = fb_oracle.load()
SEDGES = [0, 1e-5, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 1, np.inf]
THRESHOLDS = [3e-3, 1e-2, 3e-2]
stot = {i: [0, 0.0, 0.0] for i in range(len(SEDGES) - 1)}
frames = []


def account(tag, got, prev, nxt, po, winsize):
    ref, _, flips = orc.calc_tracked(prev, nxt, po)
    e = tol.epe(got, ref)
    S = tol.sensitivity(ref, orc.twins(prev, nxt, po), winsize // 2)
    fl = tol._window_max((flips > 0).astype(np.float64), winsize // 2) > 0
    for i, (lo, hi) in enumerate(zip(SEDGES[:-1], SEDGES[1:])):
        m = (S >= lo) & (S < hi)
        if m.any():
            t = stot[i]
            t[0] += int(m.sum()); t[1] = max(t[1], float(e[m].max())); t[2] += float(e[m].sum())
    row = {"tag": tag, "mean": float(e.mean()), "p999": float(np.percentile(e, 99.9)), "max": float(e.max()), "strict": tol.flow_gate(e)}
    for th in THRESHOLDS:
        un = (S >= th) | fl
        st = e[~un] if (~un).any() else np.zeros(1)
        row[th] = (float(un.mean()), float(st.mean()), float(np.percentile(st, 99.9)), float(st.max()), float(e[un].max()) if un.any() else 0.0)
    row["gate"] = tol.flow_gate(e, (S >= tol.FLOW_UNSTABLE_S) | fl)
    row["flip_share"], row["flip_max"] = float(fl.mean()), (float(e[fl].max()) if fl.any() else 0.0)
    frames.append(row)


for (W, H, levels, tag) in ((1280, 720, 1, "720p"), (1920, 1080, 1, "1080p"), (3840, 2160, 5, "4K / 5 layers")):
    fb = _lib.fb_defaults(levels)
    po = fb_oracle.default_params(levels)
    for idx in (0, 7):
        f0, f1, _ = synth.make_pair(W, H, idx)
        with _lib.Context(W, H, 1, fb) as c:
            got = c.farneback(f0[None], f1[None])[0]
        account(f"{tag} pair {idx}", got, f0, f1, po, fb.winsize)
    print("done", tag, flush=True)
import test_gpu_content as tc
for name, make in tc.CASES.items():
    f0, f1 = make()
    with _lib.Context(tc.W, tc.H, 1) as c:
        got = c.farneback(f0[None], f1[None])[0]
    account(name, got, f0, f1, fb_oracle.default_params(), 12)
print("done content", flush=True)
for seed in seeds:
    for cs in fuzz_cases(n_fuzz, seed):
        with _lib.Context(cs["W"], cs["H"], cs["B"], cs["fb"]) as c:
            got = c.farneback(cs["prev"], cs["nxt"])
        for b in range(cs["B"]):
            account(f"fuzz seed {seed} case {cs['case']} pair {b} ({cs['W']}x{cs['H']}, {cs['fb'].levels} levels, it {cs['fb'].iterations})",
                    got[b], cs["prev"][b], cs["nxt"][b], cs["po"], cs["fb"].winsize)
    print("done fuzz seed", seed, flush=True)

print(f"\n{len(frames)} frames.  EPE (GPU vs oracle) by S = max over the two twins of |calc - twin|, maximum over the winsize window")
print("   S [px)                     pixels      max EPE    mean EPE")
for i, (lo, hi) in enumerate(zip(SEDGES[:-1], SEDGES[1:])):
    n, mx, sm = stot[i]
    if n:
        print(f"   [{lo:6g}, {hi:6g})     {n:12d}   {mx:10.3e}  {sm / n:10.3e}")
print(f"\nframes failing the STRICT gate (no pixel excused): {sum(f['strict'] is not None for f in frames)};  failing the gate of oracle/tolerances.py "
      f"(S >= {tol.FLOW_UNSTABLE_S} px unstable): {sum(f['gate'] is not None for f in frames)}")
for th in THRESHOLDS:
    worst_frac = max(f[th][0] for f in frames)
    print(f"\nthreshold S >= {th} px: largest unstable share of a frame {worst_frac:.3e}; over the STABLE pixels of every frame: worst mean "
          f"{max(f[th][1] for f in frames):.3e}, worst p99.9 {max(f[th][2] for f in frames):.3e}, worst max {max(f[th][3] for f in frames):.3e}; "
          f"largest EPE on an unstable pixel {max(f[th][4] for f in frames):.3e}")
print(f"\nbranch flips alone (a pixel of the window changed sides of the border test in the last four updates): largest share of a frame "
      f"{max(f['flip_share'] for f in frames):.3e}, largest EPE on such a pixel {max(f['flip_max'] for f in frames):.3e}, largest EPE on a pixel WITHOUT "
      f"(and with S < {tol.FLOW_UNSTABLE_S}) {max(f[tol.FLOW_UNSTABLE_S][3] for f in frames):.3e}")
print("\nthe frames that fail the strict gate, and the twenty with the largest EPE: whole frame mean | p99.9 | max || at S >= 0.01 or flipped: unstable share | "
      "stable mean | stable p99.9 | stable max | unstable max || strict gate | two-class gate | workload")
shown = sorted(frames, key=lambda f: -f["max"])
shown = [f for f in shown if f["strict"] is not None] + [f for f in shown if f["strict"] is None][:20]
for f in shown:
    t = f[1e-2]
    print(f"   {f['mean']:9.3e} {f['p999']:9.3e} {f['max']:9.3e} || {t[0]:9.3e} {t[1]:9.3e} {t[2]:9.3e} {t[3]:9.3e} {t[4]:9.3e} || {str(f['strict']):10s} {str(f['gate']):24s} {f['tag']}")
