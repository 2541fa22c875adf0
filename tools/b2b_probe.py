"""Why is a back-to-back stream of one-pair calls slower than synchronised calls at some frame sizes?  Per-kernel-class times of
N back-to-back mav_process_batch_dev calls (events around every launch) next to the wall time per call with and without a sync."""
import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1280, 720)
N = 100
ctx = _lib.Context(W, H, 1)
prev, nxt = synth.make_batch(W, H, 1, distinct=1)
smp = np.stack([synth.foe_samples(W, H, 0)])
dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt); ds = ctx.alloc(smp.nbytes).upload(smp)
dr = ctx.alloc(32); dmf = ctx.alloc(W * H); dmd = ctx.alloc(W * H)
dev = lambda: ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, 1, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)
for _ in range(10): dev()
ctx.sync()
t0 = time.perf_counter()
for _ in range(N): dev(); ctx.sync()
print(f"{W}x{H} synced: {(time.perf_counter() - t0) / N * 1e3:.3f} ms per call")
for rep in range(2):
    ctx.timer_start()
    t0 = time.perf_counter()
    for _ in range(N): dev()
    enq = time.perf_counter() - t0
    print(f"back to back: {ctx.timer_stop() / N:.3f} ms per call (events), host enqueue {enq / N * 1e3:.3f} ms per call")
ctx.profile_enable(True)
for _ in range(N): dev()
prof = ctx.profile_get()
ctx.profile_enable(False)
print({k: (round(v[0] / N, 4), v[1] // N) for k, v in prof.items()})
