"""End-point error of the GPU flow against how far the oracle's own iteration still moves (run on the GPU box): the measurement
behind oracle/tolerances.py's two-class gate.  Workloads: the timed configurations (720p, 1080p, 4K / 5 layers), the six unfriendly
1080p pictures of tests/test_gpu_content.py, and tools/fuzz_shapes.py's cases for two seeds.
usage: python tools/flow_gate_survey.py [n_fuzz_cases]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd"); sys.path.insert(0, "tests")
import numpy as np
from mavflow import _lib, synth
from oracle import fb_oracle
from oracle.tolerances import epe, last_step, conditioning, _window_max, FLOW_UNSTABLE_S, flow_gate
from tools.fuzz_shapes import fuzz_cases

n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 80
orc = fb_oracle.load()
EDGES = [0, 1e-3, 1e-2, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, np.inf]
tot = {i: [0, 0.0, 0.0, 0.0] for i in range(len(EDGES) - 1)}     # pixels, max EPE, max EPE / step, sum EPE
SEDGES = [0, 1e-5, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 1, np.inf]
stot = {r: {i: [0, 0.0, 0.0] for i in range(len(SEDGES) - 1)} for r in (0, 6)}
worst_frames = []
frames = []


def account(tag, got, prev, nxt, po, winsize):
    ref, rec = orc.calc(prev, nxt, po, want_sys=True)
    e = epe(got, ref)
    st = last_step(ref, rec, winsize // 2)
    twin = epe(orc.calc_f32sums(prev, nxt, po), ref)
    for r in (0, 6):
        sw = _window_max(twin, r) if r else twin
        for i, (lo, hi) in enumerate(zip(SEDGES[:-1], SEDGES[1:])):
            m = (sw >= lo) & (sw < hi)
            if m.any():
                t = stot[r][i]
                t[0] += int(m.sum()); t[1] = max(t[1], float(e[m].max())); t[2] += float(e[m].sum())
    un = _window_max(twin, winsize // 2) >= FLOW_UNSTABLE_S
    frames.append((float(un.mean()), int(un.sum()), float(e[un].max()) if un.any() else 0.0, float(e[~un].max()), flow_gate(e, un), flow_gate(e), tag))
    worst_frames.append((float(e.max()), tag, float(e.mean()), float(np.percentile(e, 99.9)), float(st.flat[int(e.argmax())]),
                         float(conditioning(rec)[1].flat[int(e.argmax())]), float(twin.flat[int(e.argmax())]), float(sw.flat[int(e.argmax())]),
                         float(twin.max()), float(twin.mean())))
    for i, (lo, hi) in enumerate(zip(EDGES[:-1], EDGES[1:])):
        m = (st >= lo) & (st < hi)
        if m.any():
            t = tot[i]
            t[0] += int(m.sum()); t[1] = max(t[1], float(e[m].max())); t[3] += float(e[m].sum())
            if lo > 0:
                t[2] = max(t[2], float((e[m] / st[m]).max()))


for (W, H, levels, tag) in ((1280, 720, 1, "720p"), (1920, 1080, 1, "1080p"), (3840, 2160, 5, "4K / 5 layers")):
    fb = _lib.fb_defaults(levels)
    po = fb_oracle.default_params(levels)
    for idx in (0, 7):
        f0, f1, _ = synth.make_pair(W, H, idx)
        with _lib.Context(W, H, 1, fb) as c:
            got = c.farneback(f0[None], f1[None])[0]
        account(f"{tag} pair {idx}", got, f0, f1, po, fb.winsize)
        print("done", tag, idx, flush=True)
import test_gpu_content as tc
for name, make in tc.CASES.items():
    f0, f1 = make()
    with _lib.Context(tc.W, tc.H, 1) as c:
        got = c.farneback(f0[None], f1[None])[0]
    account(name, got, f0, f1, fb_oracle.default_params(), 12)
    print("done", name, flush=True)
for seed in (0, 123):
    for cs in fuzz_cases(n_fuzz, seed):
        with _lib.Context(cs["W"], cs["H"], cs["B"], cs["fb"]) as c:
            got = c.farneback(cs["prev"], cs["nxt"])
        for b in range(cs["B"]):
            account(f"fuzz seed {seed} case {cs['case']} pair {b} ({cs['W']}x{cs['H']}, {cs['fb'].levels} levels, it {cs['fb'].iterations})",
                    got[b], cs["prev"][b], cs["nxt"][b], cs["po"], cs["fb"].winsize)
    print("done fuzz seed", seed, flush=True)

print("\nEPE by the oracle's last-sweep step (window maximum), all workloads")
print("   step [px)                  pixels      max EPE    mean EPE   max EPE / step")
for i, (lo, hi) in enumerate(zip(EDGES[:-1], EDGES[1:])):
    n, mx, ratio, sm = tot[i]
    if n:
        print(f"   [{lo:6g}, {hi:6g})     {n:12d}   {mx:10.3e}  {sm / n:10.3e}   {ratio:10.3e}")
for r in (0, 6):
    print(f"\nEPE (GPU vs oracle) by the oracle's own sensitivity S = |calc - calc_f32sums|" + (f", maximum over the {2 * r + 1} x {2 * r + 1} window" if r else ", per pixel"))
    print("   S [px)                     pixels      max EPE    mean EPE")
    for i, (lo, hi) in enumerate(zip(SEDGES[:-1], SEDGES[1:])):
        n, mx, sm = stot[r][i]
        if n:
            print(f"   [{lo:6g}, {hi:6g})     {n:12d}   {mx:10.3e}  {sm / n:10.3e}")
print("\nthe fifteen worst frames: max EPE | mean | p99.9 | step at the worst pixel | cancellation there | S there | S there (13 x 13 max) | frame max S | frame mean S | workload")
for mx, tag, mean, p999, st, can, s0, s6, smax, smean in sorted(worst_frames, reverse=True)[:15]:
    print(f"   {mx:9.3e}  {mean:9.3e}  {p999:9.3e}  {st:9.3e}  {can:7.1f}  {s0:9.3e}  {s6:9.3e}  {smax:9.3e}  {smean:9.3e}   {tag}")
print("\nper frame, the gate of oracle/tolerances.py: unstable fraction | unstable pixels | max EPE unstable | max EPE stable | gate with the twin | strict gate | workload")
for fr in sorted(frames, key=lambda f: (-f[0], -f[3]))[:25]:
    print(f"   {fr[0]:9.3e}  {fr[1]:7d}  {fr[2]:9.3e}  {fr[3]:9.3e}  {str(fr[4]):28s} {str(fr[5]):10s} {fr[6]}")
print(f"frames: {len(frames)}; failing the gate with the twin: {sum(f[4] is not None for f in frames)}; failing the strict gate: {sum(f[5] is not None for f in frames)}")
