"""A/B of two BUILDS of libmavflow on one box: the headline loop (and optionally the 4K share and the 720p single pair) for the library at
the given path.   python tools/ab_lib.py <libmavflow.so> [steps]
Build a variant into csrc/build_diag/ (never into the package) and alternate:  for i in 1 2 3; do ab_lib.py A.so; ab_lib.py B.so; done"""
import sys, time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib
_lib.load(sys.argv[1])
from mavflow import synth
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
out = []
for (W, H, B, levels, n) in ((1920, 1080, 64, 1, steps), (3840, 2160, 16, 5, max(5, steps // 3)), (1280, 720, 1, 1, 300)):
    ctx = _lib.Context(W, H, B, _lib.fb_defaults(levels=levels))
    prev, nxt = synth.make_batch(W, H, B, distinct=min(B, 4))
    smp = np.stack([synth.foe_samples(W, H, b) for b in range(B)])
    dp, dn, ds = ctx.alloc(prev.nbytes).upload(prev), ctx.alloc(nxt.nbytes).upload(nxt), ctx.alloc(smp.nbytes).upload(smp)
    dr, dmf, dmd = ctx.alloc(32 * B), ctx.alloc(B * W * H), ctx.alloc(B * W * H)
    call = lambda: ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, B, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)
    for _ in range(max(3, n // 10)):
        call()
    ctx.sync()
    reps = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        ctx.sync()
        reps.append((time.perf_counter() - t0) / n)
    out.append(f"{W}x{H}x{B}: {1e3 * sorted(reps)[1]:.4f} ms ({B / sorted(reps)[1]:.0f} pairs/s)")
    import zlib
    out.append("crc %08x" % zlib.crc32(dr.download(np.uint8, (32 * B,))))
    ctx.close()
print(sys.argv[1].split("/")[-1], " | ".join(out), flush=True)
