"""Are the rare long calls periodic in wall time (something outside the process) or in call count (something we do)?
Runs N single-pair calls at 1280x720 with a sync after each and prints when the slow ones happened."""
import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
W, H = 1280, 720
ctx = _lib.Context(W, H, 1)
prev, nxt = synth.make_batch(W, H, 1, distinct=1)
smp = np.stack([synth.foe_samples(W, H, 0)])
dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt); ds = ctx.alloc(smp.nbytes).upload(smp)
dr = ctx.alloc(32)
for _ in range(5):
    ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, 1, dr.ptr)
ctx.sync()
T0 = time.perf_counter(); ts = np.empty(N); te = np.empty(N)
for i in range(N):
    ts[i] = time.perf_counter(); ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, 1, dr.ptr); ctx.sync(); te[i] = time.perf_counter()
d = (te - ts) * 1e3
med = np.median(d)
slow = np.nonzero(d > 5 * med)[0]
print(f"{N} calls, median {med:.3f} ms, total {te[-1] - T0:.2f} s, {len(slow)} slow calls")
for i in slow:
    print(f"  call {i:5d} at t = {ts[i] - T0:7.3f} s took {d[i]:7.2f} ms")
