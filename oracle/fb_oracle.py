"""ctypes loader for oracle/farneback_oracle.c (the Farneback checker). TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED at the cv2 boundary (see the C file's header): OpenCV is absent from the image and from the
reference tree, and the reference holds no fixture for cv2.calcOpticalFlowFarneback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libfboracle.so")


class Params(C.Structure):
    _fields_ = [("pyr_scale", C.c_double), ("levels", C.c_int), ("winsize", C.c_int), ("iterations", C.c_int),
                ("poly_n", C.c_int), ("poly_sigma", C.c_double), ("flags", C.c_int)]


def default_params(levels: int = 1) -> Params:
    """The reference's literal call, src/farneback.py:78-80."""
    return Params(0.4, levels, 12, 10, 8, 1.2, 0)


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "farneback_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return SO


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.fbo_calc.restype = C.c_int
        lib.fbo_num_layers.restype = C.c_int

    def num_layers(self, W, H, p):
        return self.lib.fbo_num_layers(W, H, C.byref(p))

    def layer_dims(self, W, H, p, k):
        w, h, ks, sg = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        self.lib.fbo_layer_dims(W, H, C.byref(p), k, C.byref(w), C.byref(h), C.byref(sg), C.byref(ks))
        return w.value, h.value, sg.value, ks.value

    def blur_resize(self, img, w, h, ksize, sigma):
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        out = np.empty((h, w), np.float32)
        self.lib.fbo_blur_resize(_p(img, C.c_uint8), W, H, w, h, ksize, C.c_double(sigma), _p(out, C.c_float))
        return out

    def polyexp(self, I, n=8, sigma=1.2):
        I = np.ascontiguousarray(I, np.float32)
        h, w = I.shape
        R = np.empty((h, w, 5), np.float32)
        self.lib.fbo_polyexp(_p(I, C.c_float), w, h, n, C.c_double(sigma), _p(R, C.c_float))
        return R

    def update_matrices(self, R0, R1, flow):
        h, w = flow.shape[:2]
        R0 = np.ascontiguousarray(R0, np.float32); R1 = np.ascontiguousarray(R1, np.float32)
        flow = np.ascontiguousarray(flow, np.float32)
        M = np.empty((h, w, 5), np.float32)
        self.lib.fbo_update_matrices(_p(R0, C.c_float), _p(R1, C.c_float), _p(flow, C.c_float), w, h, _p(M, C.c_float))
        return M

    def blur_iter(self, R0, R1, flow, M, winsize=12, update=True, want_sys=False):
        """One FarnebackUpdateFlow_Blur sweep; returns (new flow, new M) without touching the inputs -- and, with want_sys, the
        (h, w, 7) float64 record (g11, g12, g22, h1, h2, u_before, v_before): the system every pixel was solved from and the flow
        the sweep replaced."""
        h, w = flow.shape[:2]
        R0 = np.ascontiguousarray(R0, np.float32); R1 = np.ascontiguousarray(R1, np.float32)
        flow = np.array(flow, np.float32, order="C", copy=True)
        M = np.array(M, np.float32, order="C", copy=True)
        sys = np.empty((h, w, 7), np.float64) if want_sys else None
        self.lib.fbo_blur_iter_sys(_p(R0, C.c_float), _p(R1, C.c_float), _p(flow, C.c_float), _p(M, C.c_float), w, h,
                                   winsize, int(bool(update)), _p(sys, C.c_double) if want_sys else None)
        return (flow, M, sys) if want_sys else (flow, M)

    def resize_flow(self, prev, w, h, mul):
        prev = np.ascontiguousarray(prev, np.float32)
        ph, pw = prev.shape[:2]
        out = np.empty((h, w, 2), np.float32)
        self.lib.fbo_resize_flow(_p(prev, C.c_float), pw, ph, w, h, C.c_double(mul), _p(out, C.c_float))
        return out

    def calc(self, prev, nxt, p=None, want_sys=False):
        """cv2.calcOpticalFlowFarneback(prev, next, None, *p) restated; returns float32 (H, W, 2) -- and, with want_sys, the
        (H, W, 7) float64 record of the finest layer's last sweep (see blur_iter; NaN when iterations == 0), from which
        oracle/tolerances.py reads where the iteration has settled."""
        p = p or default_params()
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        if prev.shape != nxt.shape or prev.ndim != 2:
            raise ValueError("prev/next must be equal-size single-channel u8 images")
        H, W = prev.shape
        flow = np.empty((H, W, 2), np.float32)
        sys = np.full((H, W, 7), np.nan) if want_sys else None
        rc = self.lib.fbo_calc_sys(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), W, H, C.byref(p), _p(flow, C.c_float),
                                   _p(sys, C.c_double) if want_sys else None)
        if rc != 0:
            raise ValueError(f"fbo_calc failed: {rc}")
        return (flow, sys) if want_sys else flow

    def calc_f32sums(self, prev, nxt, p=None):
        """NOT OpenCV and never an expected value: calc() with every window sum and the 2x2 solve rounded to float32 (OpenCV sums
        in double).  |calc - calc_f32sums| is how far the restatement's own result moves under float32 rounding of its sums."""
        p = p or default_params()
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        H, W = prev.shape
        flow = np.empty((H, W, 2), np.float32)
        rc = self.lib.fbo_calc_f32sums(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), W, H, C.byref(p), _p(flow, C.c_float))
        if rc != 0:
            raise ValueError(f"fbo_calc_f32sums failed: {rc}")
        return flow

    def calc_tracked(self, prev, nxt, p=None):
        """calc() + the (H, W, 7) record of the finest layer's last sweep + flips (H, W) uint8: in how many of the finest layer's last
        four updates a pixel's displaced position changed sides of the image border test (the iteration's one discontinuity)."""
        p = p or default_params()
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        H, W = prev.shape
        flow = np.empty((H, W, 2), np.float32)
        sys = np.full((H, W, 7), np.nan)
        flips = np.zeros((H, W), np.uint8)
        rc = self.lib.fbo_calc_track(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), W, H, C.byref(p), _p(flow, C.c_float), _p(sys, C.c_double),
                                     _p(flips, C.c_uint8))
        if rc != 0:
            raise ValueError(f"fbo_calc_track failed: {rc}")
        return flow, sys, flips

    def twins(self, prev, nxt, p=None):
        """Two float32-sums re-runs of the restatement whose distance from calc() marks where its own result is not reproducible
        (oracle/tolerances.py): calc_f32sums on the frames as they are, and on the horizontally MIRRORED frames (result mirrored
        back, u negated) -- the same algorithm, mathematically the same flow, but every sliding sum runs the other way, so the
        rounding differences are a second, independent sample.  NOT OpenCV, never an expected value."""
        a = self.calc_f32sums(prev, nxt, p)
        prev_m, nxt_m = np.ascontiguousarray(np.asarray(prev)[:, ::-1]), np.ascontiguousarray(np.asarray(nxt)[:, ::-1])
        b = self.calc_f32sums(prev_m, nxt_m, p)[:, ::-1].copy()
        b[..., 0] = -b[..., 0]
        return [a, b]

    def pyramid(self, prev, nxt, p=None, on_sweep=None):
        """calc() driven layer by layer from Python with the stage functions above (bit-identical to calc(): tested).
        on_sweep(k, it, flow, M_before, sys, R0, R1) is called after every sweep (tools/worst_pixel.py)."""
        p = p or default_params()
        prev = np.ascontiguousarray(prev, np.uint8); nxt = np.ascontiguousarray(nxt, np.uint8)
        H, W = prev.shape
        flow = None
        for k in range(self.num_layers(W, H, p) - 1, -1, -1):
            w, h, sigma, ksize = self.layer_dims(W, H, p, k)
            flow = np.zeros((h, w, 2), np.float32) if flow is None else self.resize_flow(flow, w, h, 1.0 / p.pyr_scale)
            R0, R1 = (self.polyexp(self.blur_resize(img, w, h, ksize, sigma), p.poly_n, p.poly_sigma) for img in (prev, nxt))
            M = self.update_matrices(R0, R1, flow)
            for it in range(p.iterations):
                M_before = M
                flow, M, sys = self.blur_iter(R0, R1, flow, M, p.winsize, it < p.iterations - 1, want_sys=True)
                if on_sweep is not None:
                    on_sweep(k, it, flow, M_before, sys, R0, R1)
        return flow


def load() -> Oracle:
    return Oracle(C.CDLL(build()))
