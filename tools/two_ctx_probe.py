"""Probe: do several contexts (streams) working on different pairs concurrently beat one context?

    python tools/two_ctx_probe.py [W H B  group_fine,...  nctx,...]

Two effects pull in opposite directions: launches of different streams are out of phase, so the memory phases of one overlap
the compute phases of the other (a single launch of ~1.6 rounds of workgroups runs its phases in lockstep); but the working
sets of the pairs in flight must share the 256 MB Infinity Cache.  At 1920x1080 one pair's finest-layer set is 166 MB, so two
streams spill; at 960x540 (41.5 MB per pair) four pairs fit, which isolates the first effect:
    python tools/two_ctx_probe.py 960 540 64 4 1      vs      ... 960 540 64 2 2      vs      ... 960 540 64 1 4
"""
import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

a = sys.argv[1:]
W, H, B = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1920, 1080, 64)
GFS = [int(v) for v in a[3].split(",")] if len(a) >= 4 else [1]
NCTX = [int(v) for v in a[4].split(",")] if len(a) >= 5 else [1, 2, 4]
prev, nxt = synth.make_batch(W, H, B, distinct=4)
smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])


def setup(n, lo):
    c = _lib.Context(W, H, n)
    bufs = [c.alloc(prev[lo:lo+n].nbytes).upload(prev[lo:lo+n]), c.alloc(nxt[lo:lo+n].nbytes).upload(nxt[lo:lo+n]),
            c.alloc(smp[lo:lo+n].nbytes).upload(smp[lo:lo+n]), c.alloc(32 * n), c.alloc(n * W * H), c.alloc(n * W * H)]
    return c, bufs, n


def run(cb):
    c, b, n = cb
    c.process_batch_dev(b[0].ptr, b[1].ptr, b[2].ptr, n, b[3].ptr, mf_ptr=b[4].ptr, md_ptr=b[5].ptr)


for nctx in NCTX:
    ctxs = [setup(B // nctx, i * (B // nctx)) for i in range(nctx)]
    for gf in GFS:
        for cb in ctxs: cb[0].set_option("group_fine", gf)
        for _ in range(2):
            for cb in ctxs: run(cb)
        for cb in ctxs: cb[0].sync()
        t0 = time.perf_counter()
        for _ in range(4):
            for cb in ctxs: run(cb)
        for cb in ctxs: cb[0].sync()
        dt = (time.perf_counter() - t0) / 4
        print(f"{W}x{H}: {nctx} context(s), group_fine={gf}: {B / dt:.0f} pairs/s ({dt * 1e3:.2f} ms per {B} pairs)", flush=True)
    for cb in ctxs: cb[0].close()
