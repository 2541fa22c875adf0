"""Single-pair latency of the drop-in calls (what the reference's frame-by-frame loop would see).

For each frame size: wall time per call of (a) mav_process_batch_dev + sync with everything resident, (b) the enqueue part
alone, (c) the host-pointer mav_farneback (upload, compute, flow download), (d) the host-pointer mav_process_batch without
the flow download.  Prints min / median / p99 / max and the indices of calls slower than 3x the median.
Run on the GPU box:  python tools/latency_probe.py [calls]
"""
import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def stats(name, t):
    t = np.asarray(t) * 1e3
    med = np.median(t)
    slow = np.nonzero(t > 3 * med)[0]
    print(f"  {name:34s} min {t.min():7.3f}  median {med:7.3f}  p99 {np.percentile(t, 99):7.3f}  max {t.max():7.3f} ms"
          f"   slow calls: {list(slow[:8])}", flush=True)


for W, H in ((640, 480), (1280, 720), (1920, 1080)):
    print(f"{W}x{H}, one pair per call, {CALLS} calls", flush=True)
    ctx = _lib.Context(W, H, 1)
    prev, nxt = synth.make_batch(W, H, 1, distinct=1)
    smp = np.stack([synth.foe_samples(W, H, 0)])
    dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt); ds = ctx.alloc(smp.nbytes).upload(smp)
    dr = ctx.alloc(32); dmf = ctx.alloc(W * H); dmd = ctx.alloc(W * H)

    def dev():
        ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, 1, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)

    for _ in range(5):
        dev()
    ctx.sync()
    full, enq = [], []
    for _ in range(CALLS):
        t0 = time.perf_counter(); dev(); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        full.append(t2 - t0); enq.append(t1 - t0)
    stats("process_batch_dev + sync", full)
    stats("  of which enqueue", enq)
    ctx.timer_start()
    for _ in range(CALLS):
        dev()
    print(f"  GPU time per call, back to back      {ctx.timer_stop() / CALLS:7.3f} ms", flush=True)

    for _ in range(3):
        ctx.farneback(prev, nxt)
    t = []
    for _ in range(CALLS):
        t0 = time.perf_counter(); ctx.farneback(prev, nxt); t.append(time.perf_counter() - t0)
    stats("mav_farneback (host pointers)", t)
    for _ in range(3):
        ctx.process_batch(prev, nxt, smp, want_flow=False)
    t = []
    for _ in range(CALLS):
        t0 = time.perf_counter(); ctx.process_batch(prev, nxt, smp, want_flow=False); t.append(time.perf_counter() - t0)
    stats("mav_process_batch (host, no flow out)", t)
    ctx.close()

# ---- the reference-shaped loop at 1080p: one Processor.run_detection iteration against process_batch(batch = 1) ----

W, H = 1920, 1080
print(f"{W}x{H}: Processor loops, per iteration", flush=True)
ctx = _lib.Context(W, H, 1)
prev, nxt = synth.make_batch(W, H, 1, distinct=1)
smp = np.stack([synth.foe_samples(W, H, 0)])
flow = ctx.farneback(prev, nxt)
for fn, name in ((lambda: ctx.process_batch(prev, nxt, smp, want_flow=False), "process_batch(batch=1), masks out"),
                 (lambda: ctx.process_batch(prev, nxt, smp, want_flow=True), "process_batch(batch=1), flow + masks out"),
                 (lambda: ctx.detect(flow, smp), "detect(batch=1): f32 flow in, masks out")):
    for _ in range(3):
        fn()
    t = []
    for _ in range(min(CALLS, 50)):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    stats(name, t)
ctx.close()
# ---- the reference-shaped loops themselves: bench.py's `api_loop` leg (pre-generated 1080p SyntheticDataset, second run timed) ----
sys.path.insert(0, ".")
import bench
for batch in (64, 1):
    leg = bench.api_loop_leg(W, H, batch=batch, n_batches=6 if batch > 1 else 64, n_unbatched=96, staged=True, only_batched=(batch == 1))
    for k, v in leg.items():
        if k != "workload":
            print(f"  {k:42s} {v}", flush=True)
