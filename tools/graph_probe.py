"""One call of the fused path captured into a hipGraph and replayed, against the same call enqueued launch by launch (run on the GPU box).
BASELINE config 2 (1280x720, one pair) is a chain of ~27 dependent launches, each paying a kernel boundary: is a graph launch cheaper?
The capture wraps the library from OUTSIDE -- hipStreamBeginCapture on mav_stream(ctx), one mav_process_batch_dev, hipStreamEndCapture --
so nothing of the library changes for the experiment.  Prints GPU ms per call back to back (HIP events on the context's stream), wall
time of one call + sync, the host time of the enqueue, the graph's node count, and whether the replay's records and masks equal the
eager call's byte for byte.
usage: python tools/graph_probe.py [W H batch calls] [levels=N] [name=value ...]"""
import ctypes as C
import sys
import time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

pos = [a for a in sys.argv[1:] if "=" not in a]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
W, H, B, CALLS = (int(pos[i]) if len(pos) > i else d for i, d in enumerate((1280, 720, 1, 300)))
hip = C.CDLL("libamdhip64.so.7")                       # the runtime libmavflow.so is linked against: the same handle


def chk(rc, what):
    if rc != 0:
        hip.hipGetErrorString.restype = C.c_char_p
        raise RuntimeError(f"{what}: HIP error {rc} ({hip.hipGetErrorString(rc).decode()})")


levels = next((int(v) for k, v in opts if k == "levels"), 1)           # levels=5: BASELINE config 5's pyramid (a context parameter, not an option)
ctx = _lib.Context(W, H, B, _lib.fb_defaults(levels=levels))
for k, v in opts:
    if k != "levels":
        ctx.set_option(k, int(v))
prev, nxt = synth.make_batch(W, H, B, distinct=min(B, 4))
smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt); ds = ctx.alloc(smp.nbytes).upload(smp)
dr = ctx.alloc(32 * B); dmf = ctx.alloc(B * W * H); dmd = ctx.alloc(B * W * H)
ctx.lib.mav_stream.restype = C.c_void_p
stream = C.c_void_p(ctx.lib.mav_stream(ctx.h))


def eager():
    ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, B, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)


def outputs():
    ctx.sync()
    return [b.download(np.uint8, (n,)).copy() for b, n in ((dr, 32 * B), (dmf, B * W * H), (dmd, B * W * H))]


for _ in range(5):
    eager()                                            # warm: workspace, tables, second stream -- a capture must not allocate
want = outputs()
for b in (dr, dmf, dmd):
    b.upload(np.zeros(b.nbytes, np.uint8))
graph, gexec = C.c_void_p(), C.c_void_p()
chk(hip.hipStreamBeginCapture(stream, 1), "hipStreamBeginCapture (thread-local mode)")
try:
    eager()
finally:
    rc = hip.hipStreamEndCapture(stream, C.byref(graph))
chk(rc, "hipStreamEndCapture")
n_nodes = C.c_size_t(0)
chk(hip.hipGraphGetNodes(graph, None, C.byref(n_nodes)), "hipGraphGetNodes")
chk(hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")


def replay():
    chk(hip.hipGraphLaunch(gexec, stream), "hipGraphLaunch")


replay()
got = outputs()
same = all(np.array_equal(a, b) for a, b in zip(want, got))
print(f"{W}x{H} batch {B} {dict(opts)}: graph of {n_nodes.value} nodes; replay's records and masks equal the eager call's: {same}", flush=True)
for name, fn in (("eager", eager), ("graph", replay), ("eager", eager), ("graph", replay)):
    for _ in range(10):
        fn()
    ctx.sync()
    ts = []
    for rep in range(3):
        ctx.timer_start()
        for _ in range(CALLS):
            fn()
        ts.append(ctx.timer_stop() / CALLS)
    lat, enq = [], []
    for _ in range(200):
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        enq.append(t1 - t0); lat.append(t2 - t0)
    lat, enq = np.sort(np.array(lat)) * 1e3, np.array(enq) * 1e3
    print(f"  {name}: GPU ms per call back to back " + " ".join(f"{t:.4f}" for t in ts) +
          f";  one call + sync: median {np.median(lat):.4f} p99 {lat[int(0.99 * len(lat))]:.4f} ms;  host enqueue median {np.median(enq):.4f} ms", flush=True)
chk(hip.hipGraphExecDestroy(gexec), "hipGraphExecDestroy")
chk(hip.hipGraphDestroy(graph), "hipGraphDestroy")
ctx.close()
