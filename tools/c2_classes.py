"""Per-class GPU time of ONE small-batch call from HIP events around every launch (no tracer): python tools/c2_classes.py [W H B]"""
import sys; sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth
W, H, B = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1280, 720, 1)
ctx = _lib.Context(W, H, B)
prev, nxt = synth.make_batch(W, H, B, distinct=1); smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt); ds = ctx.alloc(smp.nbytes).upload(smp)
dr = ctx.alloc(32 * B); dmf = ctx.alloc(B * W * H); dmd = ctx.alloc(B * W * H)
def call(): ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, B, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)
for _ in range(300): call()
ctx.sync()
N = 50
ctx.profile_enable(True)
for _ in range(N): call()
prof = ctx.profile_get(); ctx.profile_enable(False)
tot = 0
for k, (ms, n) in prof.items():
    if n: print(f"  {k:18s} {n / N:5.1f} launches per call  {1e3 * ms / N:7.2f} us per call  {1e3 * ms / n:6.2f} us per launch")
    tot += ms
print(f"  sum {1e3 * tot / N:.1f} us per call (events around every launch add their own ~1-2 us each)")
