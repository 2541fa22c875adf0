#!/usr/bin/env python3
"""Layer-image stage alone: HIP-event time of the blur_resize kernel class (3x3 finest layer + coarse layers) per farneback call,
1080p batch 64 / 1 level and 3840x2160 batch 16 / 5 levels.   python tools/blur_probe.py"""
import sys
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth

for (W, H, B, L) in ((1920, 1080, 64, 1), (3840, 2160, 16, 5)):
    ctx = _lib.Context(W, H, B, _lib.fb_defaults(levels=L))
    prev, nxt = synth.make_batch(W, H, B, distinct=2)
    dp, dn = ctx.alloc(prev.nbytes).upload(prev), ctx.alloc(nxt.nbytes).upload(nxt)
    flow = ctx.alloc(B * W * H * 8)
    for _ in range(2):
        ctx.farneback_dev(dp.ptr, dn.ptr, B, flow.ptr)
    ctx.sync()
    ctx.profile_enable(1)
    n = 5
    for _ in range(n):
        ctx.farneback_dev(dp.ptr, dn.ptr, B, flow.ptr)
    ctx.sync()
    prof = ctx.profile_get()
    ctx.profile_enable(False)
    print(f"{W}x{H} b{B} L{L}: blur_resize {prof['blur_resize'][0] / n:.3f} ms per call in {prof['blur_resize'][1] // n} launches; "
          f"polyexp {prof['polyexp'][0] / n:.3f} ms")
    ctx.close()
