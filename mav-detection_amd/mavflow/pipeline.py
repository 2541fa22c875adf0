"""Device-resident plumbing of the reference-shaped loops (Processor.run_detection / run_detection_batched,
/root/reference/src/processor.py:283-362) -- what keeps them from moving bytes nobody reads.

The reference's loop body hands every intermediate over as a host numpy array: the flow field (16.6 MB at 1080p), both masks,
the ground truth for the TPR / FPR counts.  On a GPU none of them has to cross PCIe: the flow feeds the FoE fit and the phi stage
where it was computed, the masks are counted against the ground truth where they were written, and what the loop stores per frame
is a 32-byte record plus eight integers.  This module provides

    DeviceArray     an array-like handle of a result that stays on the device until somebody LOOKS at it (np.asarray, indexing,
                    arithmetic); Processor.flow_uv / estimate_fixed / total_mask are such handles in the fast loops, so code that
                    reads them still works and code that does not pays nothing.
    FlowStage       Dataset.get_flow_uv's Farneback seam (src/datasets/dataset.py:205-212) without the round trip: two frames in
                    (mav_upload_gather), flow left on the device in one of two alternating buffers, DeviceArray out.
    DetectPipeline  mav_process_batch_dev / mav_detect_dev + mav_tpr_fpr_counts_dev with double-buffered slots: submit(batch k + 1)
                    while batch k computes, collect(batch k) waits for batch k's marker only.  Frames are handed over as the reference
                    has them -- one numpy array per frame -- and gathered into the slot's device buffers by the library's staging
                    threads (mav_upload_gather); records and counts come back through page-locked blocks.

Nothing here computes: every number is produced by the kernels behind include/mavflow.h.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import check


class DeviceArray(np.lib.mixins.NDArrayOperatorsMixin):
    """A (read-only) result that lives in device memory owned by a FlowStage / DetectPipeline slot.  It is copied to the host the
    first time it is looked at, and -- so that a handle somebody still holds never goes stale -- by its owner right before the
    owner re-uses the memory (`_retire`).  After that it is an ordinary host array behind the same object."""

    __slots__ = ("ctx", "ptr", "shape", "dtype", "_store_dtype", "_host", "_pending", "__weakref__")

    def __init__(self, ctx: "_lib.Context", ptr: int, shape, dtype, store_dtype=None):
        self.ctx, self.ptr, self.shape, self.dtype = ctx, int(ptr), tuple(shape), np.dtype(dtype)
        self._store_dtype = np.dtype(store_dtype) if store_dtype is not None else self.dtype
        self._host = None
        self._pending = None                          # (page-locked block, _Marker): a copy to the host that has been enqueued

    @property
    def on_device(self) -> bool:
        return self._host is None and self._pending is None

    @property
    def ndim(self) -> int:
        return len(self.shape)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    @property
    def nbytes(self) -> int:
        return self.size * self.dtype.itemsize

    def __len__(self) -> int:
        return self.shape[0]

    def _materialize(self) -> np.ndarray:
        if self._host is None:
            if self._pending is not None:             # retired: the copy was enqueued, wait for it (and only for it)
                buf, marker = self._pending
                marker.wait()
                self._pending = None
            else:
                if not self.ctx.h:
                    raise _lib.MavflowError("the context that holds this array has been closed")
                buf = _lib._pinned.empty(self.ctx, self.shape, self._store_dtype)
                check(self.ctx.lib.mav_memcpy_d2h(self.ctx.h, _lib._ptr(buf), self.ptr, buf.nbytes))
            self._host = buf.view(self.dtype) if self.dtype != self._store_dtype else buf
        return self._host

    def _retire(self) -> bool:
        """The owner is about to enqueue work that overwrites the device memory: enqueue the copy to the host AHEAD of it, on the same
        stream (no host synchronisation: the loop that holds a handle of an older batch must not drain the batch in flight).  The
        owner records a marker behind the copies of all handles it retires (set through _retired_behind).  True if a copy was enqueued."""
        if self._host is not None or self._pending is not None:
            return False
        buf = _lib._pinned.empty(self.ctx, self.shape, self._store_dtype)
        check(self.ctx.lib.mav_download_async(self.ctx.h, _lib._ptr(buf), self.ptr, buf.nbytes))
        self._pending = (buf, None)
        return True

    def __array__(self, dtype=None, copy=None):
        a = self._materialize()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        inputs = tuple(np.asarray(x) if isinstance(x, DeviceArray) else x for x in inputs)
        if "out" in kw:
            kw["out"] = tuple(np.asarray(x) if isinstance(x, DeviceArray) else x for x in kw["out"])
        return getattr(ufunc, method)(*inputs, **kw)

    def __array_function__(self, func, types, args, kwargs):
        def host(x):
            if isinstance(x, DeviceArray):
                return x._materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(host(v) for v in x)
            return x
        return func(*host(args), **{k: host(v) for k, v in kwargs.items()})

    def __getitem__(self, k):
        return self._materialize()[k]

    def __iter__(self):
        return iter(self._materialize())

    def __getattr__(self, name):                      # .sum(), .astype(), .view(), .T ... : whatever ndarray offers
        if name.startswith("__") or name in DeviceArray.__slots__:   # (a slot that is not set yet must not send us into _materialize)
            raise AttributeError(name)
        return getattr(self._materialize(), name)

    def __repr__(self):
        where = "device" if self._host is None else "host"
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, {where})"


class _Marker:
    """mav_marker_* as an object several handles can share; destroyed with its last holder."""

    def __init__(self, ctx: "_lib.Context"):
        self.ctx = ctx
        m = C.c_void_p()
        check(ctx.lib.mav_marker_create(ctx.h, C.byref(m)))
        self.m = m
        check(ctx.lib.mav_marker_record(ctx.h, m))

    def wait(self) -> None:
        check(self.ctx.lib.mav_marker_wait(self.ctx.h, self.m))

    def __del__(self):
        try:
            self.ctx.lib.mav_marker_destroy(None, self.m)
        except Exception:
            pass


def _retire_all(handles) -> None:
    """Every handle of `handles` (weak references) that somebody still holds gets its copy to the host enqueued, one marker behind them."""
    moved = []
    for wr in handles:
        h = wr()
        if h is not None and h._retire():
            moved.append(h)
    handles.clear()
    if moved:
        marker = _Marker(moved[0].ctx)
        for h in moved:
            h._pending = (h._pending[0], marker)


def _as_frames(frames, H: int, W: int, name: str):
    """A batch of frames as the reference hands them over: a sequence of (H, W) u8 arrays (or one (B, H, W) array) -> list of
    C-contiguous arrays, no copy unless an element is not contiguous."""
    out = []
    for k, f in enumerate(frames):
        a = np.asarray(f)
        if a.shape != (H, W) or a.dtype != np.uint8:
            raise ValueError(f"{name}[{k}]: expected ({H}, {W}) uint8, got {a.shape} {a.dtype}")
        out.append(a if a.flags.c_contiguous else np.ascontiguousarray(a))
    return out


def _ptr_array(arrays):
    return (C.c_void_p * len(arrays))(*[a.ctypes.data for a in arrays])


class _Fence:
    """One re-recordable marker: "the last enqueued reader of this buffer".  wait() before the host overwrites the buffer through an
    UNORDERED copy (the copy stream does not wait for the compute stream); a no-op when nothing was recorded or it has long finished."""

    def __init__(self, ctx: "_lib.Context"):
        self.ctx, self.armed = ctx, False
        m = C.c_void_p()
        check(ctx.lib.mav_marker_create(ctx.h, C.byref(m)))
        self.m = m

    def record(self) -> None:
        check(self.ctx.lib.mav_marker_record(self.ctx.h, self.m))
        self.armed = True

    def wait(self) -> None:
        if self.armed:
            check(self.ctx.lib.mav_marker_wait(self.ctx.h, self.m))
            self.armed = False

    def destroy(self) -> None:
        self.ctx.lib.mav_marker_destroy(None, self.m)


class FlowStage:
    """cv2.calcOpticalFlowFarneback(prev, next, ...) (src/farneback.py:76-80) whose result stays where the next stage reads it.
    Frames are (H, W) u8 gray or (H, W, 3) u8 BGR as a capture hands them out; BGR frames are converted on the device
    (cv2.cvtColor(COLOR_BGR2GRAY), src/farneback.py:21,74 -> mav_bgr2gray_dev).

    Nothing here waits for the work in flight: the gray frames live in a ring of four slots (pairs alternate between slots 0|1 and
    2|3, a video advances one slot per frame), a new frame is copied into a slot as soon as the LAST flow that read that slot has
    finished -- two calls ago -- while the previous call's flow and whatever the caller enqueued behind it are still running; the flow
    fields alternate between two buffers the same way (DeviceArray handles of older calls are brought over before their buffer is
    re-used)."""

    RING = 4

    def __init__(self, ctx: "_lib.Context"):
        self.ctx = ctx
        n0 = ctx.W * ctx.H
        self._gray = ctx.alloc(self.RING * n0)
        self._gray_fence = [_Fence(ctx) for _ in range(self.RING)]
        self._bgr = None                                  # staging of BGR frames (first BGR frame allocates it): two frames
        self._bgr_fence = _Fence(ctx)
        self._flow = [ctx.alloc(8 * n0), ctx.alloc(8 * n0)]
        self._handles = [[], []]
        self._turn = 0
        self._pair_turn = 0
        self._have_prev = False                           # video mode (flow_next): slot of the previous frame
        self._prev_slot = 0

    def _upload(self, frames, slots) -> None:
        """frames[k] -> gray slot slots[k] of the ring."""
        ctx, n0 = self.ctx, self.ctx.W * self.ctx.H
        arrs = []
        for k, f in enumerate(frames):
            a = np.asarray(f)
            if a.dtype != np.uint8 or a.shape[:2] != (ctx.H, ctx.W) or not (a.ndim == 2 or (a.ndim == 3 and a.shape[2] == 3)):
                raise ValueError(f"frame {k}: expected ({ctx.H}, {ctx.W}) or ({ctx.H}, {ctx.W}, 3) uint8, got {a.shape} {a.dtype}")
            arrs.append(a if a.flags.c_contiguous else np.ascontiguousarray(a))
        for sl in slots:
            self._gray_fence[sl].wait()
        gray = [(a, sl) for a, sl in zip(arrs, slots) if a.ndim == 2]
        bgr = [(a, sl) for a, sl in zip(arrs, slots) if a.ndim == 3]
        if len(gray) == 2 and slots[1] == slots[0] + 1:
            check(ctx.lib.mav_upload_gather(ctx.h, self._gray.ptr + slots[0] * n0, _ptr_array([g[0] for g in gray]), 2, n0, 0))
        else:
            for a, sl in gray:
                check(ctx.lib.mav_upload_gather(ctx.h, self._gray.ptr + sl * n0, _ptr_array([a]), 1, n0, 0))
        if bgr:
            if self._bgr is None:
                self._bgr = ctx.alloc(6 * n0)
            self._bgr_fence.wait()
            for j, (a, sl) in enumerate(bgr):
                check(ctx.lib.mav_upload_gather(ctx.h, self._bgr.ptr + j * 3 * n0, _ptr_array([a]), 1, 3 * n0, 0))
        check(ctx.lib.mav_upload_fence(ctx.h))
        for j, (a, sl) in enumerate(bgr):
            check(ctx.lib.mav_bgr2gray_dev(ctx.h, self._bgr.ptr + j * 3 * n0, 1, self._gray.ptr + sl * n0))
        if bgr:
            self._bgr_fence.record()

    def _flow_into_next_buffer(self, prev_slot: int, next_slot: int) -> DeviceArray:
        ctx, n0 = self.ctx, self.ctx.W * self.ctx.H
        k = self._turn
        self._turn ^= 1
        _retire_all(self._handles[k])
        ctx.farneback_dev(self._gray.ptr + prev_slot * n0, self._gray.ptr + next_slot * n0, 1, self._flow[k].ptr)
        for sl in (prev_slot, next_slot):
            self._gray_fence[sl].record()
        h = DeviceArray(ctx, self._flow[k].ptr, (ctx.H, ctx.W, 2), np.float32)
        self._handles[k].append(weakref.ref(h))
        return h

    def flow_of(self, prev: np.ndarray, nxt: np.ndarray) -> DeviceArray:
        """Flow prev -> next as a DeviceArray (H, W, 2) float32.  The handle stays valid: the buffer it points to is re-used by the
        call after the next one, which first brings a still-referenced handle over to the host."""
        self._have_prev = False
        s0 = 2 * self._pair_turn
        self._pair_turn ^= 1
        self._upload([prev, nxt], [s0, s0 + 1])
        return self._flow_into_next_buffer(s0, s0 + 1)

    def flow_next(self, frame: np.ndarray) -> Optional[DeviceArray]:
        """Video mode, the reference's Farneback.process() (src/farneback.py:73-81): the flow from the previous frame handed in to this
        one; the previous frame's gray image is still on the device (the class's `prevgray`), so one frame crosses PCIe per step.
        None for the first frame."""
        slot = (self._prev_slot + 1) % self.RING if self._have_prev else 0
        self._upload([frame], [slot])
        out = self._flow_into_next_buffer(self._prev_slot, slot) if self._have_prev else None
        self._have_prev, self._prev_slot = True, slot
        return out

    def close(self):
        for hs in self._handles:
            _retire_all(hs)
        if self.ctx.h:
            self.ctx.sync()                               # the copies of retired handles have landed before the buffers go
        for b in [self._gray, self._bgr] + self._flow:
            if b is not None:
                b.free()
        for f in self._gray_fence + [self._bgr_fence]:
            f.destroy()


class _Slot:
    pass


class DetectPipeline:
    """The fused loop body (src/processor.py:305-351) for up to `batch` pairs per submit, `slots` submits in flight."""

    N_PAIRS = 1000                                        # focus_of_expansion.py:67

    def __init__(self, ctx: "_lib.Context", batch: int, slots: int = 3, keep_flow: bool = False):
        """slots = 3: one batch computing, one being enqueued, and the one before still referenced by whoever holds its handles (the
        loops keep the last finished frame's masks as attributes) -- its buffers are not needed yet, so nothing has to be brought over."""
        if batch > ctx.max_batch:
            raise ValueError(f"batch {batch} exceeds the context's max_batch {ctx.max_batch}")
        self.ctx, self.B, self.keep_flow = ctx, int(batch), bool(keep_flow)
        self.n0 = ctx.W * ctx.H
        B, n0 = self.B, self.n0
        self._par_off = {}
        off = 0
        for name, nbytes in (("samples", B * 4 * self.N_PAIRS * 4), ("omega", B * 24), ("dt", B * 8), ("frame0", (B + 7) & ~7)):
            self._par_off[name] = (off, nbytes)
            off += nbytes
        self._par_bytes = off
        self._out_bytes = B * 32 + 2 * B * 32                 # records, counts of the fixed mask, counts of the dynamic mask
        self.slots = []
        for _ in range(slots):
            s = _Slot()
            s.frames = None                                   # (2 B + 1) frames, allocated by the first submit that brings frames
            s.flow_in = None                                  # B flow fields, allocated by the first submit that brings host flow
            s.flow_out = ctx.alloc(8 * n0 * B) if keep_flow else None
            s.mf, s.md = ctx.alloc(n0 * B), ctx.alloc(n0 * B)
            s.par = ctx.alloc(self._par_bytes)
            s.out = ctx.alloc(self._out_bytes)
            s.sky = s.gt = None
            s.h_par = _lib._pinned.empty(ctx, (self._par_bytes,), np.uint8)
            s.h_out = _lib._pinned.empty(ctx, (self._out_bytes,), np.uint8)
            m = C.c_void_p()
            check(ctx.lib.mav_marker_create(ctx.h, C.byref(m)))
            s.marker = m
            s.handles = []
            s.n = 0
            s.flow_handle = None
            s.busy = False
            self.slots.append(s)
        self._turn = 0
        self._shared = {}                                     # "gt" / "sky": (source array, device buffer, images)
        self._sky_any = None                                  # (shared sky array, does it mask anything)
        self.foe_params = _lib.foe_defaults()
        self.thr_params = _lib.thr_defaults()

    # -- shared images (a segmentation / sky mask that is the same for every frame of the run) ---------------------------------
    def _shared_image(self, kind: str, img: np.ndarray, replicate: int) -> "_lib.DeviceBuffer":
        cur = self._shared.get(kind)
        if cur is not None and cur[0] is img and cur[2] == replicate:
            return cur[1]
        a = np.ascontiguousarray(np.asarray(img).reshape(self.ctx.H, self.ctx.W))
        a = a.view(np.uint8) if a.dtype == np.bool_ else a
        if a.dtype != np.uint8:
            raise ValueError(f"{kind}: u8 or bool image expected, got {a.dtype}")
        ctx = self.ctx
        if cur is not None:
            ctx.sync()
            cur[1].free()
        buf = ctx.alloc(self.n0 * replicate)
        check(ctx.lib.mav_upload_gather(ctx.h, buf.ptr, _ptr_array([a] * replicate), replicate, self.n0, 1))
        check(ctx.lib.mav_upload_fence(ctx.h))
        self._shared[kind] = (img, buf, replicate, a)         # `a` kept: the id of `img` must not be recycled while cached
        return buf

    def _per_pair_images(self, slot, attr: str, imgs, n: int) -> int:
        arrs = []
        for k, m in enumerate(imgs):
            a = np.asarray(m)
            if a.shape != (self.ctx.H, self.ctx.W):
                raise ValueError(f"{attr}[{k}]: expected ({self.ctx.H}, {self.ctx.W}), got {a.shape}")
            a = a.view(np.uint8) if a.dtype == np.bool_ else a
            if a.dtype != np.uint8:
                raise ValueError(f"{attr}[{k}]: u8 or bool image expected, got {a.dtype}")
            arrs.append(a if a.flags.c_contiguous else np.ascontiguousarray(a))
        if len(arrs) != n:
            raise ValueError(f"{attr}: {len(arrs)} images for {n} pairs")
        buf = getattr(slot, attr)
        if buf is None:
            buf = self.ctx.alloc(self.n0 * self.B)
            setattr(slot, attr, buf)
        check(self.ctx.lib.mav_upload_gather(self.ctx.h, buf.ptr, _ptr_array(arrs), n, self.n0, 0))
        return buf.ptr

    # -- submit / collect -------------------------------------------------------------------------------------------------------
    def submit(self, samples, prev: Optional[Sequence[np.ndarray]] = None, nxt: Optional[Sequence[np.ndarray]] = None, flow=None,
               omega=None, dt=None, frame0=None, sky=None, sky_shared=None, gt=None, gt_shared=None) -> int:
        """Enqueue one batch: frames (prev / nxt: one array per pair) or flow (a DeviceArray of this context, or float32 host arrays:
        one (H, W, 2) array per pair) -> FoE, masks, box records and, when a ground truth is given, the calculate_tpr_fpr counts of both
        masks.  Returns the ticket for collect().  Nothing is waited for except the slot's own previous batch."""
        ctx, lib = self.ctx, self.ctx.lib
        H, W, n0 = ctx.H, ctx.W, self.n0
        dev_flow = None
        if flow is not None and isinstance(flow, DeviceArray):
            if flow.ctx is not ctx:
                raise ValueError("a DeviceArray flow must live on this pipeline's context")
            if flow.on_device:
                if flow.shape != (H, W, 2) or flow.dtype != np.float32:
                    raise ValueError(f"flow: expected ({H}, {W}, 2) float32, got {flow.shape} {flow.dtype}")
                dev_flow, n = flow.ptr, 1
            else:
                flow = [flow._materialize()]
        if dev_flow is None and flow is not None:
            fl = [np.asarray(f) for f in (flow if isinstance(flow, (list, tuple)) else [flow])]
            for k, f in enumerate(fl):
                if f.shape != (H, W, 2) or f.dtype != np.float32:
                    raise ValueError(f"flow[{k}]: expected ({H}, {W}, 2) float32, got {f.shape} {f.dtype}")
            fl = [f if f.flags.c_contiguous else np.ascontiguousarray(f) for f in fl]
            n = len(fl)
        elif dev_flow is None:
            if prev is None or nxt is None:
                raise ValueError("submit() needs either flow or both prev and nxt")
            p, q = _as_frames(prev, H, W, "prev"), _as_frames(nxt, H, W, "next")
            if len(p) != len(q):
                raise ValueError("prev and next differ in length")
            n = len(p)
        if not 1 <= n <= self.B:
            raise ValueError(f"{n} pairs outside [1, {self.B}]")
        smp = np.asarray(samples)
        if smp.size != n * 4 * self.N_PAIRS:
            raise ValueError(f"samples: expected {n} x {2 * self.N_PAIRS} x 2 values, got shape {smp.shape}")
        for name, v, per in (("omega", omega, 3), ("dt", dt, 1), ("frame0", frame0, 1)):
            if v is not None and np.size(v) != n * per:
                raise ValueError(f"{name}: expected {n * per} values, got {np.size(v)}")

        # the arguments are in order: take the next slot (only now -- a refused call leaves the pipeline as it was)
        si = self._turn
        self._turn = (self._turn + 1) % len(self.slots)
        s = self.slots[si]
        if s.busy:                                            # its previous batch was never collected: finish it before the buffers go
            check(lib.mav_marker_wait(ctx.h, s.marker))
            s.busy = False
        _retire_all(s.handles)
        s.n = n

        # small per-pair parameters: packed into the slot's page-locked block, one asynchronous copy
        hp = s.h_par
        o, _ = self._par_off["samples"]
        hp[o:o + n * 16 * self.N_PAIRS].view(np.uint32)[:] = smp.reshape(-1)
        o_om, _ = self._par_off["omega"]
        o_dt, _ = self._par_off["dt"]
        o_f0, _ = self._par_off["frame0"]
        if omega is not None:
            hp[o_om:o_om + n * 24].view(np.float64)[:] = np.asarray(omega, np.float64).reshape(-1)
            hp[o_dt:o_dt + n * 8].view(np.float64)[:] = 1.0 if dt is None else np.asarray(dt, np.float64).reshape(-1)
        if frame0 is not None:
            hp[o_f0:o_f0 + n] = np.asarray(frame0).reshape(-1).astype(np.uint8)
        check(lib.mav_upload_async_unordered(ctx.h, s.par.ptr, _lib._ptr(hp), self._par_bytes))

        # frames / flow: gathered from the caller's arrays by the library's staging threads; this slot's buffers are idle (its
        # previous batch has been waited for), so the copies need no ordering against the batch that is computing now
        flow_ptr = None
        if dev_flow is not None:
            flow_ptr = dev_flow
        elif flow is not None:
            if s.flow_in is None:
                s.flow_in = ctx.alloc(8 * n0 * self.B)
            check(lib.mav_upload_gather(ctx.h, s.flow_in.ptr, _ptr_array(fl), n, 8 * n0, 0))
            flow_ptr = s.flow_in.ptr
        else:
            if s.frames is None:
                s.frames = ctx.alloc((2 * self.B + 1) * n0)
            if n > 1 and all(q[k] is p[k + 1] for k in range(n - 1)):
                # a video: pair k = (frame k, frame k + 1).  One run of n + 1 frames, next = prev + one frame: the library
                # recognises the layout and blurs / expands every frame once (mav_farneback, "frame sequences")
                check(lib.mav_upload_gather(ctx.h, s.frames.ptr, _ptr_array(p + [q[-1]]), n + 1, n0, 0))
                prev_ptr, next_ptr = s.frames.ptr, s.frames.ptr + n0
            else:
                check(lib.mav_upload_gather(ctx.h, s.frames.ptr, _ptr_array(p + q), 2 * n, n0, 0))
                prev_ptr, next_ptr = s.frames.ptr, s.frames.ptr + n * n0
        sky_ptr = None
        if sky_shared is not None:
            if self._sky_any is None or self._sky_any[0] is not sky_shared:
                self._sky_any = (sky_shared, bool(np.asarray(sky_shared).any()))
            if self._sky_any[1]:                              # an all-False sky changes no mask: same result as no sky at all
                sky_ptr = self._shared_image("sky", sky_shared, self.B).ptr
        elif sky is not None:
            sky_ptr = self._per_pair_images(s, "sky", sky, n)
        gt_ptr, gt_images = None, 0
        if gt_shared is not None:
            gt_ptr, gt_images = self._shared_image("gt", gt_shared, 1).ptr, 1
        elif gt is not None:
            gt_ptr, gt_images = self._per_pair_images(s, "gt", gt, n), n
        check(lib.mav_upload_fence(ctx.h))

        par = s.par.ptr
        smp_ptr = par + self._par_off["samples"][0]
        om_ptr = par + o_om if omega is not None else None
        dt_ptr = par + o_dt if omega is not None else None
        f0_ptr = par + o_f0 if frame0 is not None else None
        res_ptr = s.out.ptr
        if flow_ptr is None:
            out_flow = s.flow_out.ptr if s.flow_out is not None else None
            check(lib.mav_process_batch_dev(ctx.h, prev_ptr, next_ptr, smp_ptr, om_ptr, dt_ptr, f0_ptr, sky_ptr, n, C.byref(self.foe_params),
                                            C.byref(self.thr_params), out_flow, None, s.mf.ptr, s.md.ptr, res_ptr))
            s.flow_handle = None if out_flow is None else (out_flow, )
        else:
            check(lib.mav_detect_dev(ctx.h, flow_ptr, smp_ptr, om_ptr, dt_ptr, f0_ptr, sky_ptr, n, C.byref(self.foe_params),
                                     C.byref(self.thr_params), None, s.mf.ptr, s.md.ptr, res_ptr))
            s.flow_handle = None
        s.has_counts = gt_ptr is not None
        nout = n * 32
        if s.has_counts:
            check(lib.mav_tpr_fpr_counts_dev(ctx.h, gt_ptr, gt_images, s.mf.ptr, s.md.ptr, 255, n, res_ptr + self.B * 32, res_ptr + 2 * self.B * 32))
            nout = self._out_bytes
        check(lib.mav_download_async(ctx.h, _lib._ptr(s.h_out), res_ptr, nout))
        check(lib.mav_marker_record(ctx.h, s.marker))
        s.busy = True
        return si

    def collect(self, ticket: int) -> dict:
        """Wait for that batch (and only that batch) and return its records (n,) RESULT_DTYPE, the (n, 4) int64 counts of both masks
        (None without a ground truth) and, per pair, lazy handles of the masks (H, W) bool -- and of the flow, if the pipeline keeps
        it.  A handle costs nothing until it is looked at; it stays valid after the slot is re-used (the slot brings it over first)."""
        s = self.slots[ticket]
        if not s.busy:
            raise ValueError("this ticket has been collected already")
        ctx = self.ctx
        check(ctx.lib.mav_marker_wait(ctx.h, s.marker))
        s.busy = False
        n, B = s.n, self.B
        res = s.h_out[:n * 32].view(_lib.RESULT_DTYPE).copy()
        cf = cd = None
        if s.has_counts:
            cf = s.h_out[B * 32:B * 32 + n * 32].view(np.int64).reshape(n, 4).copy()
            cd = s.h_out[2 * B * 32:2 * B * 32 + n * 32].view(np.int64).reshape(n, 4).copy()
        n0 = self.n0
        mf = [DeviceArray(ctx, s.mf.ptr + k * n0, (ctx.H, ctx.W), np.bool_, np.uint8) for k in range(n)]
        md = [DeviceArray(ctx, s.md.ptr + k * n0, (ctx.H, ctx.W), np.bool_, np.uint8) for k in range(n)]
        flow = None
        if s.flow_handle is not None:
            flow = [DeviceArray(ctx, s.flow_handle[0] + k * 8 * n0, (ctx.H, ctx.W, 2), np.float32) for k in range(n)]
        for h in mf + md + (flow or []):
            s.handles.append(weakref.ref(h))
        return dict(results=res, counts_fixed=cf, counts_dyn=cd, mask_fixed=mf, mask_dyn=md, flow=flow)

    def close(self):
        ctx = self.ctx
        if not ctx.h:
            return
        for s in self.slots:
            if s.busy:
                ctx.lib.mav_marker_wait(ctx.h, s.marker)
                s.busy = False
            _retire_all(s.handles)
        ctx.sync()
        for s in self.slots:
            for b in (s.frames, s.flow_in, s.flow_out, s.mf, s.md, s.par, s.out, s.sky, s.gt):
                if b is not None:
                    b.free()
            ctx.lib.mav_marker_destroy(ctx.h, s.marker)
        for rec in self._shared.values():
            rec[1].free()
        self._shared = {}
        self.slots = []


# ---- lanes: a stream of SMALL calls spread over several contexts ------------------------------------------------------------------------
# One pair of a 720p or 1080p video is a chain of ~27 dependent launches of which many fill a fraction of the chip (a coarse-layer sweep
# at 720p is 144 workgroups on 256 CUs) and all pay a kernel boundary.  A context is single-threaded and owns its streams and workspace;
# distinct contexts are independent (include/mavflow.h) -- so consecutive one-pair calls given to two or three contexts IN TURN are
# independent chains the GPU interleaves: one chain's boundaries, tails and latency-bound launches fall into the other's launches.
# Measured (tools/lanes_probe.py, one pair per call, ms per pair with 1 / 2 / 3 contexts): 1280x720 0.299 / 0.207 / 0.181,
# 1920x1080 0.523 / 0.429 / 0.458, 640x480 0.207 / 0.124 / 0.101 (one stream per lane: _one_stream_per_lane).  Beyond ~200 MB of sweep working set per call the chains fight over
# the 256 MB Infinity Cache and a second lane loses (a 64-pair batch already keeps two pairs in flight inside its one context).
def auto_lanes(W: int, H: int, batch: int = 1) -> int:
    """Contexts a stream of `batch`-pair calls at this frame size is spread over: 3 up to 100 MB of finest-layer sweep working set per
    call (80 B per pixel and pair), 2 up to 200 MB (one 1080p pair: 166 MB), 1 beyond."""
    ws = 80 * W * H * batch
    return 3 if ws <= (100 << 20) else (2 if ws <= (200 << 20) else 1)


def _one_stream_per_lane(ctxs) -> None:
    """Several lanes: every context keeps ALL its work -- uploads included -- on its one compute stream (option "inline_uploads"; the copy
    and pair streams of a context are only created when first used).  The HIP runtime maps streams onto a small pool of hardware queues
    (4 by default): streams beyond it share queues, two lanes whose streams share a queue do not overlap at all, and a lane whose copy
    stream sits on another lane's queue waits for that lane's kernels.  Measured with two lanes at 1080p, the one-frame loop: 0.48 ms
    per frame when the runtime happened to spread the six streams well, 0.61 (= one lane) after an earlier context had shifted the
    assignment, 0.79 with an eight-queue pool; one stream per lane makes it 0.47 - 0.48 in every order (profiles/r05/lanes_probe.txt).
    A single lane keeps its copy stream: there the upload of frame i + 1 overlaps the chain of frame i."""
    for c in ctxs:
        c.set_option("inline_uploads", 1 if len(ctxs) > 1 else 0)


class LanedFlowStage:
    """FlowStage over several contexts taken in turn (flow_of); the video form (flow_next) keeps its previous frame on ONE device
    buffer and therefore stays on the first lane."""

    def __init__(self, ctxs):
        _one_stream_per_lane(ctxs)
        self.stages = [FlowStage(c) for c in ctxs]
        self._turn = 0

    @property
    def lanes(self) -> int:
        return len(self.stages)

    def flow_of(self, prev: np.ndarray, nxt: np.ndarray) -> DeviceArray:
        st = self.stages[self._turn]
        self._turn = (self._turn + 1) % len(self.stages)
        return st.flow_of(prev, nxt)

    def flow_next(self, frame: np.ndarray) -> Optional[DeviceArray]:
        return self.stages[0].flow_next(frame)

    def close(self):
        for st in self.stages:
            st.close()


class LanedPipeline:
    """DetectPipeline over several contexts.  A batch whose flow is a DeviceArray goes to the lane (context) that holds it; anything
    else takes the lanes in turn.  Tickets are (lane, slot); collect them in submission order.  `depth` = how many submitted batches a
    loop should keep uncollected so that every lane has work (the number of lanes)."""

    def __init__(self, ctxs, batch: int, **kw):
        _one_stream_per_lane(ctxs)
        self.pipes = [DetectPipeline(c, batch, **kw) for c in ctxs]
        self._turn = 0

    @property
    def depth(self) -> int:
        return len(self.pipes)

    @property
    def ctxs(self):
        return [p.ctx for p in self.pipes]

    def set_params(self, foe_params=None, thr_params=None) -> None:
        for p in self.pipes:
            if foe_params is not None:
                p.foe_params = foe_params
            if thr_params is not None:
                p.thr_params = thr_params

    def submit(self, samples, flow=None, **kw):
        if isinstance(flow, DeviceArray) and flow.on_device:
            lane = next((k for k, p in enumerate(self.pipes) if p.ctx is flow.ctx), None)
            if lane is None:
                raise ValueError("a DeviceArray flow must live on one of this pipeline's contexts")
        else:
            lane = self._turn
            self._turn = (self._turn + 1) % len(self.pipes)
        return lane, self.pipes[lane].submit(samples, flow=flow, **kw)

    def collect(self, ticket) -> dict:
        lane, t = ticket
        return self.pipes[lane].collect(t)

    def close(self):
        for p in self.pipes:
            p.close()
