"""Probe: a stream of ONE-PAIR calls (BASELINE config 2's shape of work; the one-frame loop of the API) spread over several contexts.
Each context owns its streams and workspace, so consecutive calls on different contexts are independent chains that the GPU may
interleave: one chain's kernel boundaries, tails and latency-bound coarse-layer launches fall into the other's launches.
    python tools/lanes_probe.py [W H calls]
Prints ms per pair for 1, 2, 3 contexts taking the calls in turn (same total number of calls)."""
import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

a = sys.argv[1:]
W, H, CALLS = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1280, 720, 300)
prev, nxt = synth.make_batch(W, H, 1, distinct=1)
smp = np.stack([synth.foe_samples(W, H, 0)])


def setup():
    c = _lib.Context(W, H, 1)
    b = [c.alloc(prev.nbytes).upload(prev), c.alloc(nxt.nbytes).upload(nxt), c.alloc(smp.nbytes).upload(smp), c.alloc(32), c.alloc(W * H), c.alloc(W * H)]
    return c, b


def run(cb):
    c, b = cb
    c.process_batch_dev(b[0].ptr, b[1].ptr, b[2].ptr, 1, b[3].ptr, mf_ptr=b[4].ptr, md_ptr=b[5].ptr)


for n in (1, 2, 3):
    ctxs = [setup() for _ in range(n)]
    for k in range(30):
        run(ctxs[k % n])
    for cb in ctxs:
        cb[0].sync()
    best = []
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(CALLS):
            run(ctxs[k % n])
        for cb in ctxs:
            cb[0].sync()
        best.append((time.perf_counter() - t0) / CALLS)
    print(f"{W}x{H}, one pair per call, {n} context(s) in turn: ms per pair " + " ".join(f"{1e3 * t:.4f}" for t in best), flush=True)
    for cb in ctxs:
        cb[0].close()
