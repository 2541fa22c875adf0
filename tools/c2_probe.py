"""Back-to-back small-batch calls of the fused path (BASELINE config 2: 1280x720, one pair): HIP-event time per call.
usage: python tools/c2_probe.py [W H batch calls] [name=value ...]      (name=value: mav_set_option before the loop)
Run under `rocprofv3 --kernel-trace` and feed the trace to tools/trace_timeline.py to see one call's launches and gaps."""
import sys
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

pos = [a for a in sys.argv[1:] if "=" not in a]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
W, H, B, CALLS = (int(pos[i]) if len(pos) > i else d for i, d in enumerate((1280, 720, 1, 200)))
ctx = _lib.Context(W, H, B)
for k, v in opts:
    ctx.set_option(k, int(v))
prev, nxt = synth.make_batch(W, H, B, distinct=min(B, 4))
smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt); ds = ctx.alloc(smp.nbytes).upload(smp)
dr = ctx.alloc(32 * B); dmf = ctx.alloc(B * W * H); dmd = ctx.alloc(B * W * H)


def call():
    ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, B, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)


for _ in range(10):
    call()
ctx.sync()
best = []
for rep in range(3):
    ctx.timer_start()
    for _ in range(CALLS):
        call()
    best.append(ctx.timer_stop() / CALLS)
print(f"{W}x{H} batch {B} {dict(opts)}: GPU ms per call, back to back: " + " ".join(f"{t:.4f}" for t in best), flush=True)
import time
ctx.sync()
enq = []
for _ in range(50):
    t0 = time.perf_counter(); call(); enq.append(time.perf_counter() - t0); ctx.sync()
enq = np.array(enq) * 1e3
print(f"    host enqueue per call (idle stream): median {np.median(enq):.4f} ms, min {enq.min():.4f}", flush=True)
ctx.close()
