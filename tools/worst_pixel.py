"""Where GPU and oracle part on the fuzz's worst frame (run on the GPU box).

Replays case `case` of `tools/fuzz_shapes.py` with seed `seed` (default: seed 123, case 10 = 1048 x 925, pyr_scale 0.5, 4 extra layers,
the frame behind gpurun_out/fuzz80.log's 0.269 px), finds the worst pixel of the worst pair and prints

  1. how the end-point error is distributed over that frame (counts above 0.01 / 0.05 / 0.15 px, the worst pixel's neighbourhood),
  2. for the worst pixel, layer by layer and sweep by sweep: the oracle's float64 system (g11, g12, g22, h1, h2), det + 1e-3,
     the cancellation in the determinant, the oracle's flow, the GPU's float32 flow after the same sweep (stage hooks, the GPU's own
     intermediates fed forward) and the EPE there, next to the layer's maximum EPE,
  3. the ONE-sweep error: the GPU's sweep applied to the ORACLE's M (what float32 sums cost per sweep, before feedback),
  4. max EPE against candidate explanations: conditioning of the 2x2 system, how far the oracle's iteration still moves, the
     oracle's own sensitivity to float32 rounding of its sums (two twins) and flips of the image-border test -- the last two are
     what oracle/tolerances.py calls unstable.

usage: python tools/worst_pixel.py [seed] [case]
"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib
from oracle import fb_oracle
from oracle.tolerances import epe, conditioning, last_step, sensitivity, FLOW_UNSTABLE_S
from tools.fuzz_shapes import fuzz_cases

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 123
want = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cs = [c for c in fuzz_cases(want + 1, seed)][want]
W, H, B, fb, po, prev, nxt = (cs[k] for k in ("W", "H", "B", "fb", "po", "prev", "nxt"))
orc = fb_oracle.load()
soa = lambda a: np.ascontiguousarray(np.moveaxis(a, -1, 0))
aos = lambda a: np.ascontiguousarray(np.moveaxis(a, 0, -1))
print(f"seed {seed} case {want}: {W}x{H} B={B} pyr_scale={fb.pyr_scale} levels={fb.levels} winsize={fb.winsize} iterations={fb.iterations} poly_n={fb.poly_n}")

with _lib.Context(W, H, B, fb) as c:
    gflow = c.farneback(prev, nxt)
refs = [orc.calc(prev[b], nxt[b], po, want_sys=True) for b in range(B)]
E = np.stack([epe(gflow[b], refs[b][0]) for b in range(B)])
for b in range(B):
    print(f"  pair {b}: mean {E[b].mean():.3e}  p99.9 {np.percentile(E[b], 99.9):.3e}  max {E[b].max():.3e}")
b, y, x = (int(v) for v in np.unravel_index(int(E.argmax()), E.shape))
e, (ref, sys_) = E[b], refs[b]
print(f"\n1. worst pixel: pair {b}, (x, y) = ({x}, {y}), EPE {e[y, x]:.4f} px;  oracle flow {ref[y, x]}  GPU flow {gflow[b, y, x]}")
for t in (0.01, 0.05, 0.15):
    print(f"   pixels above {t:4.2f} px: {int((e > t).sum()):6d} of {e.size}")
y0, x0 = max(0, y - 3), max(0, x - 3)
print("   EPE in the 7 x 7 neighbourhood (rows y-3 .. y+3):")
for r in e[y0:y + 4, x0:x + 4]:
    print("     " + " ".join(f"{v:7.4f}" for v in r))

# 2. the staged chains: oracle (float64 sums) and GPU (stage hooks, its own intermediates fed forward)
otrace = {}
def on_sweep(k, it, flow, M_before, s, R0, R1):
    otrace[(k, it)] = (flow.copy(), M_before, s, R0, R1)
oflow = orc.pyramid(prev[b], nxt[b], po, on_sweep)
assert np.array_equal(oflow, ref), "the Python-driven oracle pyramid must equal fbo_calc"
gtrace = {}
one_sweep = {}
with _lib.Context(W, H, 1, fb) as c:
    nl = c.num_layers()
    gf = None
    for k in range(nl - 1, -1, -1):
        w, h, _, _ = c.layer_dims(k)
        gf = np.zeros((h, w, 2), np.float32) if gf is None else orc.resize_flow(gf, w, h, 1.0 / fb.pyr_scale)
        R0, R1 = (c.stage_polyexp(c.stage_blur_resize(img, k), k) for img in (prev[b], nxt[b]))
        M = c.stage_update_matrices(R0, R1, gf, k)
        for it in range(fb.iterations):
            upd = it < fb.iterations - 1
            # the GPU's sweep on the ORACLE's M and expansions: the error of ONE sweep
            of, oM, _, oR0, oR1 = otrace[(k, it)]
            f1, _ = c.stage_blur_iter(soa(oR0), soa(oR1), soa(oM), k, upd)
            one_sweep[(k, it)] = epe(f1, of)
            gf, Mn = c.stage_blur_iter(R0, R1, M, k, upd)
            M = Mn if upd else M
            gtrace[(k, it)] = gf.copy()
    d = float(np.abs(gf - gflow[b]).max())
    print(f"\n2. staged GPU chain vs the library's own call: max |difference| {d:.3e} px"
          + ("  (bit-identical)" if d == 0 else "  (the staged chain up-samples the coarse flow on the host)"))
print("   'step' = how far the oracle's own flow moved in this sweep at the pixel (px)")
print("   layer sweep |        g11         g12         g22          h1          h2 |   det+1e-3    cancel |  oracle (u, v)        GPU (u, v)         |     step | EPE here   layer max (at)")
for k in range(nl - 1, -1, -1):
    w, h, _, _ = orc.layer_dims(W, H, po, k)
    xk, yk = min(w - 1, int(x * w / W)), min(h - 1, int(y * h / H))
    for it in range(fb.iterations):
        of, _, s, _, _ = otrace[(k, it)]
        g = gtrace[(k, it)]
        ek = epe(g, of)
        my, mx = np.unravel_index(int(ek.argmax()), ek.shape)
        g11, g12, g22, h1, h2, ub, vb = s[yk, xk]
        det, cancel = conditioning(s[yk, xk][None])
        step = float(np.hypot(of[yk, xk, 0] - ub, of[yk, xk, 1] - vb))
        print(f"   {k:5d} {it:5d} | {g11:11.5f} {g12:11.5f} {g22:11.5f} {h1:11.5f} {h2:11.5f} | {det[0] + 1e-3:10.3e} {cancel[0]:9.1f} |"
              f" ({of[yk, xk, 0]:8.4f},{of[yk, xk, 1]:8.4f}) ({g[yk, xk, 0]:8.4f},{g[yk, xk, 1]:8.4f}) | {step:8.2e} | {ek[yk, xk]:8.2e}   {ek.max():8.2e} ({mx}, {my})")

print("\n3. ONE GPU sweep on the oracle's own M / R0 / R1 (float32 sums vs float64 sums, no feedback): EPE at the worst pixel, frame max")
for k in range(nl - 1, -1, -1):
    w, h, _, _ = orc.layer_dims(W, H, po, k)
    xk, yk = min(w - 1, int(x * w / W)), min(h - 1, int(y * h / H))
    print(f"   layer {k}: " + "  ".join(f"{one_sweep[(k, it)][yk, xk]:.1e}/{one_sweep[(k, it)].max():.1e}" for it in range(fb.iterations)))

# 4. error against conditioning, all pairs of the case
print("\n4. max / mean EPE by the last sweep's conditioning (all pairs of the case)")
det_all = np.stack([conditioning(refs[i][1])[0] for i in range(B)])
can_all = np.stack([conditioning(refs[i][1])[1] for i in range(B)])
edges = [0, 2, 5, 10, 20, 50, 100, 200, 500, 1000, 1e4, 1e5, np.inf]
print("   cancellation (g11 g22 + g12^2) / (det + 1e-3)      pixels      max EPE     mean EPE")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (can_all >= lo) & (can_all < hi)
    if m.any():
        print(f"   [{lo:8g}, {hi:8g})                          {int(m.sum()):10d}   {E[m].max():10.3e}   {E[m].mean():10.3e}")
edges = [-np.inf, 1e-3, 1e-2, 1e-1, 1, 10, 100, 1e3, np.inf]
print("   determinant g11 g22 - g12^2                         pixels      max EPE     mean EPE")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (det_all >= lo) & (det_all < hi)
    if m.any():
        print(f"   [{lo:8g}, {hi:8g})                          {int(m.sum()):10d}   {E[m].max():10.3e}   {E[m].mean():10.3e}")
print(f"   worst pixel: det {det_all[b, y, x]:.4e}  cancellation {can_all[b, y, x]:.1f}")
st_all = np.stack([last_step(refs[i][0], refs[i][1], fb.winsize // 2) for i in range(B)])
edges = [0, 1e-3, 1e-2, 0.02, 0.05, 0.1, 0.2, 0.5, 1, 2, 5, np.inf]
print("   the oracle's last-sweep step, window maximum [px)   pixels      max EPE     mean EPE")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (st_all >= lo) & (st_all < hi)
    if m.any():
        print(f"   [{lo:8g}, {hi:8g})                          {int(m.sum()):10d}   {E[m].max():10.3e}   {E[m].mean():10.3e}")
print(f"   worst pixel: the oracle's flow moved {st_all[b, y, x]:.3f} px in the last sweep (window maximum)")
from oracle.tolerances import _window_max, unstable_mask
tw_all = [orc.twins(prev[i], nxt[i], po) for i in range(B)]
S_all = np.stack([sensitivity(refs[i][0], tw_all[i], fb.winsize // 2) for i in range(B)])
flips_b = orc.calc_tracked(prev[b], nxt[b], po)[2]
fl_b = _window_max((flips_b > 0).astype(np.float64), fb.winsize // 2) > 0
edges = [0, 1e-5, 1e-4, 1e-3, FLOW_UNSTABLE_S, 0.05, 0.15, 0.5, 1, np.inf]
print("   S = max over the oracle's two float32-sums twins of |calc - twin|, window maximum [px)     pixels      max EPE     mean EPE")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (S_all >= lo) & (S_all < hi)
    if m.any():
        print(f"   [{lo:8g}, {hi:8g})                          {int(m.sum()):10d}   {E[m].max():10.3e}   {E[m].mean():10.3e}")
un_b = unstable_mask(refs[b][0], tw_all[b], fb.winsize // 2, flips_b)
print(f"   worst pixel: S = {S_all[b, y, x]:.3f} px; a pixel of its window changed sides of the image-border test in the last four updates: {bool(fl_b[y, x])} "
      f"({int((flips_b > 0).sum())} such pixels in the frame)")
print(f"   unstable pixels of the pair (S >= {FLOW_UNSTABLE_S} or flipped): {int(un_b.sum())} = {un_b.mean():.2e} of the frame; over the stable rest: "
      f"mean {E[b][~un_b].mean():.3e}  p99.9 {np.percentile(E[b][~un_b], 99.9):.3e}  max {E[b][~un_b].max():.3e} px")
