#!/usr/bin/env python3
"""Where does a step's time go, WITHOUT a tracer?  (Under rocprofv3 the host side of the ~1 700 launches of a step becomes the
bottleneck: a traced 4K step takes 31 - 41 ms instead of 26.)  One mav_process_batch_dev call with HIP events around every launch
(profile mode 1) or around every run of launches of one class on a stream (mode 2, a tenth of the events), its wall time next to an
unprofiled call's, and the time per set of kernel classes running concurrently.

    python tools/untraced_anatomy.py [W H batch levels]          (default 3840 2160 16 5)"""
import collections
import sys
import time

sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

W, H, B, L = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (3840, 2160, 16, 5)
ctx = _lib.Context(W, H, B, _lib.fb_defaults(levels=L))
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
prev, nxt = synth.make_batch(W, H, B, distinct=min(B, 4))
smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
dp, dn, ds = ctx.alloc(prev.nbytes).upload(prev), ctx.alloc(nxt.nbytes).upload(nxt), ctx.alloc(smp.nbytes).upload(smp)
dr, dmf, dmd = ctx.alloc(B * 32), ctx.alloc(B * W * H), ctx.alloc(B * W * H)


def call():
    ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, B, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)


def wall(n=5):
    ctx.sync()
    t = time.perf_counter()
    for _ in range(n):
        call()
    ctx.sync()
    return 1e3 * (time.perf_counter() - t) / n


for _ in range(3):
    call()
print(f"{W}x{H} batch {B} levels {L}: unprofiled {wall():.3f} ms per call")
names = None
for mode in (2, 1):
    ctx.profile_enable(mode)
    ms = wall(1)
    names = list(ctx.profile_get().keys())
    k, st, t0, t1 = ctx.profile_intervals()
    ctx.profile_enable(False)
    ev = []
    for i in range(len(k)):
        c = names[k[i]].replace("blur_iter_coarse", "sweep").replace("blur_iter", "sweep")
        ev.append((float(t0[i]), 1, c)); ev.append((float(t1[i]), -1, c))
    ev.sort()
    act, acc, last = collections.Counter(), collections.Counter(), ev[0][0]
    for t, d, c in ev:
        acc[tuple(sorted((a, b) for a, b in act.items() if b > 0))] += t - last
        last = t
        act[c] += d
    print(f"profile mode {mode}: {ms:.3f} ms for the call, {len(k)} intervals, span {ev[-1][0] - ev[0][0]:.3f} ms")
    for key, v in acc.most_common(14):
        print(f"   {v:8.3f} ms   {dict(key) if key else 'nothing running'}")
    per = collections.Counter()
    for i in range(len(k)):
        per[(names[k[i]], int(st[i]))] += float(t1[i] - t0[i])
    print("   sum per class and stream:", {f"{a}@{b}": round(v, 3) for (a, b), v in sorted(per.items())})
ctx.close()
