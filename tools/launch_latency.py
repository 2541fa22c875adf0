import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth
W, H, B = 1280, 720, 1
ctx = _lib.Context(W, H, B)
prev, nxt = synth.make_batch(W, H, B, distinct=1)
smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt); ds = ctx.alloc(smp.nbytes).upload(smp)
dr = ctx.alloc(32 * B); dmf = ctx.alloc(B * W * H); dmd = ctx.alloc(B * W * H)
def run(): ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, B, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)
for _ in range(3): run()
ctx.sync()
for trial in range(3):
    t0 = time.perf_counter(); ctx.timer_start(); t1 = time.perf_counter()
    for _ in range(20): run()
    t2 = time.perf_counter(); ms = ctx.timer_stop(); t3 = time.perf_counter(); ctx.sync(); t4 = time.perf_counter()
    print(f"timer_start {1e3*(t1-t0):.3f} ms, enqueue 20 steps {1e3*(t2-t1):.3f} ms, timer_stop {1e3*(t3-t2):.3f} ms (events {ms:.3f} ms), sync {1e3*(t4-t3):.3f}")
