"""Diagnostic: where do the waves of the sweep kernel spend their cycles?  Needs the -DMAV_STAMPS build
(`make -C mav-detection_amd/csrc diag` -> mav-detection_amd/csrc/build_diag/libmavflow_diag.so); never used by the product.
    python tools/phase_stamps.py [batch] [group_fine] [path of the diagnostic library]"""
import ctypes as C, os, sys
sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, synth
_lib.load(sys.argv[3] if len(sys.argv) > 3 else "mav-detection_amd/csrc/build_diag/libmavflow_diag.so")
W, H, B = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
gf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.Context(W, H, B)
ctx.set_option("group_fine", gf)
prev, nxt = synth.make_batch(W, H, B, distinct=2)
flow = ctx.alloc(B * W * H * 8); dp = ctx.alloc(prev.nbytes).upload(prev); dn = ctx.alloc(nxt.nbytes).upload(nxt)
ctx.farneback_dev(dp.ptr, dn.ptr, B, flow.ptr); ctx.sync()
buf = (C.c_ulonglong * 8)()
ctx.lib.mav_debug_read_stamps(buf, 1)
ctx.farneback_dev(dp.ptr, dn.ptr, B, flow.ptr); ctx.sync()
ctx.lib.mav_debug_read_stamps(buf, 1)
v = np.array(list(buf), dtype=np.float64)
waves = v[7]
names = ["A: loads+vertical sums", "barrier wait", "B: LDS reads+sums+solve+park", "C first half (2 px)", "C total", "-", "whole wave", "waves"]
print(f"group_fine={gf}, waves {waves:.0f}; shader cycles per wave (s_memtime), updating sweeps of all layers")
for i in (0, 1, 2, 3, 4, 6):
    print(f"  {names[i]:32s} {v[i] / waves:10.1f} ticks/wave   {100 * v[i] / v[6]:5.1f} %")
