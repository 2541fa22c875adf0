"""Probe: does running two contexts (two streams, different pairs) concurrently beat one context? (tail/gap filling vs Infinity-Cache sharing)"""
import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth
W, H, B = 1920, 1080, 64
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
for nctx in (1, 2, 4):
    ctxs = [setup(B // nctx, i * (B // nctx)) for i in range(nctx)]
    for gf in (1,):
        for cb in ctxs: cb[0].set_option("group_fine", gf)
        for _ in range(2):
            for cb in ctxs: run(cb)
        for cb in ctxs: cb[0].sync()
        t0 = time.perf_counter()
        for _ in range(4):
            for cb in ctxs: run(cb)
        for cb in ctxs: cb[0].sync()
        dt = (time.perf_counter() - t0) / 4
        print(f"{nctx} context(s), group_fine={gf}: {B / dt:.0f} pairs/s ({dt * 1e3:.2f} ms per 64 pairs)")
    for cb in ctxs: cb[0].close()
