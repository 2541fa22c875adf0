"""Which HIP call blocks in the rare 40 - 80 ms one-pair call?  (run on the GPU box)

    LD_PRELOAD=tools/hipstamps/libhipstamps.so python tools/stall_stamps.py [calls]

tools/hipstamps interposes every HIP runtime entry point libmavflow.so uses and stamps its start and end on the host.  Per frame size
a fresh context runs 5 warm-up calls and then `calls` single-pair mav_process_batch_dev calls (everything resident), each followed by
a stream synchronisation.  For every call slower than 2 ms the probe prints every HIP call inside it -- offset from the call's start,
duration, and the host-side gap before it -- so the blocking call (or a gap BETWEEN calls: Python, the OS) is named.  The series is
then repeated (a) in the same context, (b) in a second context of the same size, (c) after a one-second pause, (d) with a device
synchronisation + 50 ms pause between warm-up and series, to tell "once per process / context / idle period" apart.
"""
import ctypes as C
import gc
import os
import sys
import time
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

CALLS = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipstamps", "libhipstamps.so")
if "libhipstamps" not in os.environ.get("LD_PRELOAD", ""):
    sys.exit("run with LD_PRELOAD=tools/hipstamps/libhipstamps.so")
hs = C.CDLL(SO)
hs.hipstamps_count.restype = C.c_size_t
hs.hipstamps_read.restype = C.c_size_t
hs.hipstamps_read.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p]
hs.hipstamps_name.restype = C.c_char_p
hs.hipstamps_now.restype = C.c_uint64
hs.hipstamps_enable(1)


def records(first, n):
    out = np.empty((n, 3), np.uint64)
    got = hs.hipstamps_read(first, n, out.ctypes.data)
    return out[:got]


def series(ctx, bufs, calls, tag):
    dp, dn, ds, dr, dmf, dmd = bufs
    hs.hipstamps_reset()
    first, t0s, t1s, t2s = [], [], [], []
    for _ in range(calls):
        first.append(hs.hipstamps_count())
        a = hs.hipstamps_now()
        ctx.process_batch_dev(dp.ptr, dn.ptr, ds.ptr, 1, dr.ptr, mf_ptr=dmf.ptr, md_ptr=dmd.ptr)
        b = hs.hipstamps_now()
        ctx.sync()
        c = hs.hipstamps_now()
        t0s.append(a); t1s.append(b); t2s.append(c)
    first.append(hs.hipstamps_count())
    t0s, t1s, t2s = (np.asarray(v, np.uint64) for v in (t0s, t1s, t2s))
    full = (t2s - t0s) / 1e6
    enq = (t1s - t0s) / 1e6
    slow = np.nonzero(full > 2.0)[0]
    print(f"  {tag}: {calls} calls  median {np.median(full):.3f}  p99 {np.percentile(full, 99):.3f}  max {full.max():.3f} ms "
          f"(enqueue median {np.median(enq):.3f} max {enq.max():.3f});  calls above 2 ms: {[(int(i), round(float(full[i]), 2)) for i in slow[:6]]}", flush=True)
    # the longest single HIP call of the series, by name
    rec = records(0, first[-1])
    if len(rec):
        dur = (rec[:, 2] - rec[:, 1]) / 1e3
        names = {}
        for i in np.argsort(-dur)[:4]:
            names[int(i)] = (hs.hipstamps_name(int(rec[i, 0])).decode(), float(dur[i]))
        print("    longest HIP calls of the series: " + ", ".join(f"{n} {d:.1f} us" for n, d in names.values()), flush=True)
    for i in slow[:2]:
        r = records(first[i], first[i + 1] - first[i])
        print(f"    call {int(i)} ({full[i]:.2f} ms, enqueue {enq[i]:.2f} ms): {len(r)} HIP calls; those above 50 us, and host gaps above 50 us:")
        prev_end = int(t0s[i])
        for k, (nid, a, b) in enumerate(r):
            gap = (int(a) - prev_end) / 1e3
            d = (int(b) - int(a)) / 1e3
            if gap > 50:
                print(f"      [{k:3d}] +{(int(a) - int(t0s[i])) / 1e3:9.1f} us  host gap of {gap:9.1f} us BEFORE {hs.hipstamps_name(int(nid)).decode()}")
            if d > 50:
                print(f"      [{k:3d}] +{(int(a) - int(t0s[i])) / 1e3:9.1f} us  {hs.hipstamps_name(int(nid)).decode():24s} took {d:9.1f} us")
            prev_end = int(b)
        tail = (int(t2s[i]) - prev_end) / 1e3
        print(f"      after the last HIP call: {tail:.1f} us to the end of the call")
    return full


def make(W, H):
    ctx = _lib.Context(W, H, 1)
    prev, nxt = synth.make_batch(W, H, 1, distinct=1)
    smp = np.stack([synth.foe_samples(W, H, 0)])
    bufs = (ctx.alloc(prev.nbytes).upload(prev), ctx.alloc(nxt.nbytes).upload(nxt), ctx.alloc(smp.nbytes).upload(smp),
            ctx.alloc(32), ctx.alloc(W * H), ctx.alloc(W * H))
    for _ in range(5):
        ctx.process_batch_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, 1, bufs[3].ptr, mf_ptr=bufs[4].ptr, md_ptr=bufs[5].ptr)
    ctx.sync()
    return ctx, bufs


gc.disable()          # rule Python's collector out: no cycle collection inside a series
for W, H in ((1280, 720), (1920, 1080)):
    print(f"{W}x{H}", flush=True)
    ctx, bufs = make(W, H)
    series(ctx, bufs, CALLS, "fresh context, 5 warm-up calls")
    series(ctx, bufs, CALLS, "(a) same context again")
    ctx2, bufs2 = make(W, H)
    series(ctx2, bufs2, CALLS, "(b) second context, same size")
    time.sleep(1.0)
    series(ctx2, bufs2, CALLS, "(c) the same after a 1 s pause")
    ctx2.close(); ctx.close()
    ctx3, bufs3 = make(W, H)
    time.sleep(0.05)
    series(ctx3, bufs3, CALLS, "(d) fresh context, 50 ms pause after the warm-up")
    ctx3.close()
