"""Device-resident plumbing of the reference-shaped loops (Processor.run_detection / run_detection_batched,
/root/reference/src/processor.py:283-362) -- what keeps them from moving bytes nobody reads.

The reference's loop body hands every intermediate over as a host numpy array: the flow field (16.6 MB at 1080p), both masks,
the ground truth for the TPR / FPR counts.  On a GPU none of them has to cross PCIe: the flow feeds the FoE fit and the phi stage
where it was computed, the masks are counted against the ground truth where they were written, and what the loop stores per frame
is a 32-byte record plus eight integers.  This module provides

    DeviceArray     an array-like handle of a result that stays on the device until somebody LOOKS at it (np.asarray, indexing,
                    arithmetic); Processor.flow_uv / estimate_fixed / total_mask are such handles in the fast loops, so code that
                    reads them still works and code that does not pays nothing.
    FlowStage       Dataset.get_flow_uv's Farneback seam (src/datasets/dataset.py:205-212) without the round trip: frames in, flow left
                    on the device in one of two alternating buffers, DeviceArray out -- DEFERRED: nothing is enqueued until the handle
                    reaches a DetectPipeline (then upload + Farneback + detection are one step) or somebody reads it.
    DetectPipeline  one mav_frame_step per submit (uploads of the caller's per-frame arrays, Farneback, detection, TPR / FPR counts, result
                    download, marker), posted to the context's worker thread; double-buffered slots: submit(batch k + 1) while batch k
                    computes, collect(batch k) waits for batch k only; records and counts come back through page-locked blocks.
    Laned...        the same over several contexts taken in turn, one stream each, in a priority class of their own.

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

    __slots__ = ("ctx", "ptr", "shape", "dtype", "_store_dtype", "_host", "_pending", "_deferred", "__weakref__")

    def __init__(self, ctx: "_lib.Context", ptr: int, shape, dtype, store_dtype=None):
        self.ctx, self.ptr, self.shape, self.dtype = ctx, int(ptr), tuple(shape), np.dtype(dtype)
        self._store_dtype = np.dtype(store_dtype) if store_dtype is not None else self.dtype
        self._host = None
        self._pending = None                          # (page-locked block, _Marker): a copy to the host that has been enqueued
        self._deferred = None                         # a flow FlowStage has not enqueued yet (_DeferredFlow): see FlowStage.flow_of

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
        if self._deferred is not None:                # nobody took the flow into a fused step: compute it now
            self._deferred.flush()
        if self._host is None:
            if self._pending is not None:             # retired: the copy was enqueued, wait for it (and only for it)
                buf, marker = self._pending
                marker.wait()
                self._pending = None
            else:
                if not self.ctx.alive:
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
        if self._deferred is not None:
            self._deferred.flush()
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
        check(self.ctx.lib.mav_marker_wait(None, self.m))

    def done(self) -> bool:
        d = C.c_int(0)
        check(self.ctx.lib.mav_marker_query(None, self.m, C.byref(d)))
        return bool(d.value)

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
            # the page-locked block is the target of a copy that is only ENQUEUED: should the handle be dropped before the copy has
            # landed, the pool must not lend the block to anybody else until the marker has fired
            _lib._pinned.guard_until(h._pending[0], marker)


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


class _DeferredFlow:
    """A flow FlowStage has decided everything about -- frame slots, flow buffer -- but not enqueued: DetectPipeline.submit takes it
    into ONE fused step (upload -> Farneback -> detection -> counts -> download: mav_frame_step) when the handle reaches it, and
    anybody else who looks at the handle first gets it computed on the spot (flush)."""

    __slots__ = ("stage", "frames", "slots", "prev_slot", "next_slot", "k", "handle", "bgr")

    def flush(self) -> None:
        self.stage._flush(self)


class FlowStage:
    """cv2.calcOpticalFlowFarneback(prev, next, ...) (src/farneback.py:76-80) whose result stays where the next stage reads it.
    Frames are (H, W) u8 gray or (H, W, 3) u8 BGR as a capture hands them out; BGR frames are converted on the device
    (cv2.cvtColor(COLOR_BGR2GRAY), src/farneback.py:21,74 -> mav_bgr2gray_dev).

    Nothing here waits for the work in flight: the gray frames live in a ring of four slots (pairs alternate between slots 0|1 and
    2|3, a video advances one slot per frame), a new frame is copied into a slot as soon as the LAST flow that read that slot has
    finished -- two calls ago -- while the previous call's flow and whatever the caller enqueued behind it are still running; the flow
    fields alternate between two buffers the same way (DeviceArray handles of older calls are brought over before their buffer is
    re-used).

    The flow is DEFERRED (round 6): flow_of / flow_next decide where everything goes and hand out the handle, but enqueue nothing.
    When the handle reaches a DetectPipeline of the same context -- the reference's loop: get_flow_uv(i), then the detection on it --
    the frames' upload, Farneback and the detection travel as one mav_frame_step; when somebody reads the handle first, or the next
    flow is asked for, the flow is computed on the spot.  Same launches, same results either way."""

    RING = 4

    def __init__(self, ctx: "_lib.Context", defer: bool = True):
        self.ctx, self.defer = ctx, bool(defer)
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
        self._open: Optional[_DeferredFlow] = None        # the one flow that has been handed out but not enqueued

    def _check_frames(self, frames):
        ctx, arrs = self.ctx, []
        for k, f in enumerate(frames):
            a = np.asarray(f)
            if a.dtype != np.uint8 or a.shape[:2] != (ctx.H, ctx.W) or not (a.ndim == 2 or (a.ndim == 3 and a.shape[2] == 3)):
                raise ValueError(f"frame {k}: expected ({ctx.H}, {ctx.W}) or ({ctx.H}, {ctx.W}, 3) uint8, got {a.shape} {a.dtype}")
            arrs.append(a if a.flags.c_contiguous else np.ascontiguousarray(a))
        return arrs

    def _upload(self, arrs, slots) -> None:
        """arrs[k] (checked frames) -> gray slot slots[k] of the ring."""
        ctx, n0 = self.ctx, self.ctx.W * self.ctx.H
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

    def _plan(self, arrs, slots, prev_slot: int, next_slot: int) -> DeviceArray:
        """The next flow buffer for the flow prev_slot -> next_slot (arrs go to `slots` first); handles of the buffer's previous user
        are brought over.  Returns the handle -- deferred, or enqueued at once when the stage does not defer."""
        self._settle_open()
        ctx = self.ctx
        k = self._turn
        self._turn ^= 1
        _retire_all(self._handles[k])
        h = DeviceArray(ctx, self._flow[k].ptr, (ctx.H, ctx.W, 2), np.float32)
        self._handles[k].append(weakref.ref(h))
        d = _DeferredFlow()
        d.stage, d.frames, d.slots, d.prev_slot, d.next_slot, d.k = self, arrs, list(slots), prev_slot, next_slot, k
        d.bgr = bool(arrs) and arrs[0].ndim == 3
        d.handle = weakref.ref(h)
        # one fused step takes gray frames into CONSECUTIVE slots, or BGR frames (at most the two the staging holds) likewise
        fusable = all(a.ndim == arrs[0].ndim for a in arrs) and all(slots[i + 1] == slots[i] + 1 for i in range(len(slots) - 1))
        if self.defer and fusable:
            h._deferred = d
            self._open = d
        else:
            self._flush(d)
        return h

    def _settle_open(self) -> None:
        """Before the next flow is planned: the one handed out earlier that nobody has enqueued.  Its handle still held -> compute it;
        handle gone -> nobody will read it, but in video mode its frame is the next pair's `prev`: the upload still happens."""
        d = self._open
        if d is None:
            return
        if d.handle() is not None:
            self._flush(d)
            return
        self._open = None
        if self._have_prev:
            self._upload(d.frames, d.slots)
        d.frames = None

    def _flush(self, d: _DeferredFlow) -> None:
        """Enqueue a planned flow through the plain calls (uploads, fence, Farneback, slot markers)."""
        ctx, n0 = self.ctx, self.ctx.W * self.ctx.H
        if self._open is d:
            self._open = None
        h = d.handle()
        if h is not None:
            h._deferred = None
        self._upload(d.frames, d.slots)
        ctx.farneback_dev(self._gray.ptr + d.prev_slot * n0, self._gray.ptr + d.next_slot * n0, 1, self._flow[d.k].ptr)
        for sl in (d.prev_slot, d.next_slot):
            self._gray_fence[sl].record()
        d.frames = None

    def _take(self, d: _DeferredFlow, step: "_lib.FrameStep", keep: list):
        """Write the flow part of a fused step for the deferred flow `d` (called by DetectPipeline.submit, which posts the step): the
        gather of its frames, BGR -> gray, Farneback, the markers.  Returns a callable to run once the step HAS been posted."""
        ctx, n0 = self.ctx, self.ctx.W * self.ctx.H
        nf = len(d.frames)
        waits = [self._gray_fence[sl] for sl in d.slots if self._gray_fence[sl].armed]
        records = [self._gray_fence[sl] for sl in dict.fromkeys((d.prev_slot, d.next_slot))]
        g = step.gather[step.n_gather]
        srcs = _ptr_array(d.frames)
        keep.extend((srcs, d.frames))
        g.src_host, g.count = C.cast(srcs, C.POINTER(C.c_void_p)), nf
        if d.bgr:
            if self._bgr is None:
                self._bgr = ctx.alloc(6 * n0)
            if self._bgr_fence.armed:
                waits.append(self._bgr_fence)
            records.append(self._bgr_fence)
            g.bytes_each, g.dst_dev = 3 * n0, self._bgr.ptr
            step.bgr_dev, step.n_bgr, step.gray_dev = self._bgr.ptr, nf, self._gray.ptr + d.slots[0] * n0
        else:
            g.bytes_each, g.dst_dev = n0, self._gray.ptr + d.slots[0] * n0
        step.n_gather += 1
        step.compute_flow = 1
        step.prev_dev, step.next_dev = self._gray.ptr + d.prev_slot * n0, self._gray.ptr + d.next_slot * n0
        step.flow_dev = self._flow[d.k].ptr
        if waits:
            wa = (C.c_void_p * len(waits))(*[f.m.value for f in waits])
            keep.append(wa)
            step.wait_before, step.n_wait_before = C.cast(wa, C.POINTER(C.c_void_p)), len(waits)
        ra = (C.c_void_p * len(records))(*[f.m.value for f in records])
        keep.append(ra)
        step.record_after_flow, step.n_record_after_flow = C.cast(ra, C.POINTER(C.c_void_p)), len(records)

        def posted():
            for f in waits:
                f.armed = False                      # the step waits for them before it overwrites the slots
            for f in records:
                f.armed = True
            if self._open is d:
                self._open = None
            h = d.handle()
            if h is not None:
                h._deferred = None
            d.frames = None
        return posted

    def flow_of(self, prev: np.ndarray, nxt: np.ndarray) -> DeviceArray:
        """Flow prev -> next as a DeviceArray (H, W, 2) float32.  The handle stays valid: the buffer it points to is re-used by the
        call after the next one, which first brings a still-referenced handle over to the host."""
        arrs = self._check_frames([prev, nxt])
        self._have_prev = False
        s0 = 2 * self._pair_turn
        self._pair_turn ^= 1
        return self._plan(arrs, [s0, s0 + 1], s0, s0 + 1)

    def flow_next(self, frame: np.ndarray) -> Optional[DeviceArray]:
        """Video mode, the reference's Farneback.process() (src/farneback.py:73-81): the flow from the previous frame handed in to this
        one; the previous frame's gray image is still on the device (the class's `prevgray`), so one frame crosses PCIe per step.
        None for the first frame."""
        arrs = self._check_frames([frame])
        slot = (self._prev_slot + 1) % self.RING if self._have_prev else 0
        if self._have_prev:
            out = self._plan(arrs, [slot], self._prev_slot, slot)
        else:
            self._settle_open()
            self._upload(arrs, [slot])
            out = None
        self._have_prev, self._prev_slot = True, slot
        return out

    def close(self):
        self._open = None                                 # a flow nobody asked for is not computed
        for hs in self._handles:
            for wr in hs:
                h = wr()
                if h is not None and h._deferred is not None:
                    h._deferred.flush()                   # ... unless its handle is still held: it stays readable
        for hs in self._handles:
            _retire_all(hs)
        if self.ctx.alive:
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

    def __init__(self, ctx: "_lib.Context", batch: int, slots: int = 3, keep_flow: bool = False, worker: bool = True):
        """slots = 3: one batch computing, one being enqueued, and the one before still referenced by whoever holds its handles (the
        loops keep the last finished frame's masks as attributes) -- its buffers are not needed yet, so nothing has to be brought over.
        worker: submit() posts its step to the context's worker thread (mav_frame_step_post) instead of enqueueing it itself."""
        if batch > ctx.max_batch:
            raise ValueError(f"batch {batch} exceeds the context's max_batch {ctx.max_batch}")
        self.ctx, self.B, self.keep_flow, self.worker = ctx, int(batch), bool(keep_flow), bool(worker)
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
            s.ticket, s.keep = 0, None
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

    # -- submit / collect -------------------------------------------------------------------------------------------------------
    def _images(self, name: str, imgs, n: int):
        """per-pair sky masks / ground-truth images as the reference hands them over -> checked u8 arrays"""
        arrs = []
        for k, m in enumerate(imgs):
            a = np.asarray(m)
            if a.shape != (self.ctx.H, self.ctx.W):
                raise ValueError(f"{name}[{k}]: expected ({self.ctx.H}, {self.ctx.W}), got {a.shape}")
            a = a.view(np.uint8) if a.dtype == np.bool_ else a
            if a.dtype != np.uint8:
                raise ValueError(f"{name}[{k}]: u8 or bool image expected, got {a.dtype}")
            arrs.append(a if a.flags.c_contiguous else np.ascontiguousarray(a))
        if len(arrs) != n:
            raise ValueError(f"{name}: {len(arrs)} images for {n} pairs")
        return arrs

    @staticmethod
    def _add_gather(step, keep, arrays, bytes_each, dst_ptr) -> None:
        srcs = _ptr_array(arrays)
        keep.extend((srcs, arrays))
        g = step.gather[step.n_gather]
        g.src_host, g.count, g.bytes_each, g.dst_dev = C.cast(srcs, C.POINTER(C.c_void_p)), len(arrays), bytes_each, dst_ptr
        step.n_gather += 1

    def submit(self, samples, prev: Optional[Sequence[np.ndarray]] = None, nxt: Optional[Sequence[np.ndarray]] = None, flow=None,
               omega=None, dt=None, frame0=None, sky=None, sky_shared=None, gt=None, gt_shared=None) -> int:
        """Enqueue one batch: frames (prev / nxt: one array per pair) or flow (a DeviceArray of this context, or float32 host arrays:
        one (H, W, 2) array per pair) -> FoE, masks, box records and, when a ground truth is given, the calculate_tpr_fpr counts of both
        masks.  Returns the ticket for collect().  Nothing is waited for except the slot's own previous batch.

        The whole batch -- uploads, fence, Farneback (frames, or a flow FlowStage has deferred), detection, counts, result download,
        marker -- is ONE mav_frame_step, posted to the context's worker thread (worker=True, the default: this thread goes on at once
        and several lanes enqueue side by side) or enqueued by this thread in one call (worker=False)."""
        ctx, lib = self.ctx, self.ctx.lib
        H, W, n0 = ctx.H, ctx.W, self.n0
        dev_flow = deferred = None
        if flow is not None and isinstance(flow, DeviceArray):
            if flow.ctx is not ctx:
                raise ValueError("a DeviceArray flow must live on this pipeline's context")
            if flow.on_device:
                if flow.shape != (H, W, 2) or flow.dtype != np.float32:
                    raise ValueError(f"flow: expected ({H}, {W}, 2) float32, got {flow.shape} {flow.dtype}")
                dev_flow, n, deferred = flow.ptr, 1, flow._deferred
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
        sky_arrs = self._images("sky", sky, n) if (sky is not None and sky_shared is None) else None
        gt_arrs = self._images("gt", gt, n) if (gt is not None and gt_shared is None) else None
        # images shared by the whole run: uploaded once, ahead of everything (plain calls: they drain the worker the first time only)
        sky_ptr = None
        if sky_shared is not None:
            if self._sky_any is None or self._sky_any[0] is not sky_shared:
                self._sky_any = (sky_shared, bool(np.asarray(sky_shared).any()))
            if self._sky_any[1]:                              # an all-False sky changes no mask: same result as no sky at all
                sky_ptr = self._shared_image("sky", sky_shared, self.B).ptr
        gt_ptr, gt_images = None, 0
        if gt_shared is not None:
            gt_ptr, gt_images = self._shared_image("gt", gt_shared, 1).ptr, 1

        # the arguments are in order: take the next slot (only now -- a refused call leaves the pipeline as it was)
        si = self._turn
        self._turn = (self._turn + 1) % len(self.slots)
        s = self.slots[si]
        if s.busy:                                            # its previous batch was never collected: finish it before the buffers go
            self._wait_slot(s)
        _retire_all(s.handles)
        s.n = n
        step = _lib.FrameStep()
        keep = []                                             # what the step reads on the host: alive until the slot's marker has fired
        step.n = n

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
        step.par_host, step.par_dev, step.par_bytes = hp.ctypes.data, s.par.ptr, self._par_bytes
        step.off_samples, step.off_omega, step.off_dt, step.off_frame0 = o, o_om, o_dt, o_f0
        step.has_omega, step.has_frame0 = int(omega is not None), int(frame0 is not None)

        # frames / flow: gathered from the caller's arrays by the library's staging threads; this slot's buffers are idle (its
        # previous batch has been waited for), so the copies need no ordering against the batch that is computing now
        posted = None
        s.flow_handle = None
        if deferred is not None:                              # a flow FlowStage planned: its frames, Farneback and the detection in one step
            posted = deferred.stage._take(deferred, step, keep)
        elif dev_flow is not None:
            step.flow_dev = dev_flow
        elif flow is not None:
            if s.flow_in is None:
                s.flow_in = ctx.alloc(8 * n0 * self.B)
            self._add_gather(step, keep, fl, 8 * n0, s.flow_in.ptr)
            step.flow_dev = s.flow_in.ptr
        else:
            if s.frames is None:
                s.frames = ctx.alloc((2 * self.B + 1) * n0)
            if n > 1 and all(q[k] is p[k + 1] for k in range(n - 1)):
                # a video: pair k = (frame k, frame k + 1).  One run of n + 1 frames, next = prev + one frame: the library
                # recognises the layout and blurs / expands every frame once (mav_farneback, "frame sequences")
                self._add_gather(step, keep, p + [q[-1]], n0, s.frames.ptr)
                step.prev_dev, step.next_dev = s.frames.ptr, s.frames.ptr + n0
            else:
                self._add_gather(step, keep, p + q, n0, s.frames.ptr)
                step.prev_dev, step.next_dev = s.frames.ptr, s.frames.ptr + n * n0
            step.compute_flow = 1
            if s.flow_out is not None:                        # (else: the context's own flow buffer, which no handle points into)
                step.flow_dev = s.flow_out.ptr
                s.flow_handle = (s.flow_out.ptr, )
        for attr, arrs in (("sky", sky_arrs), ("gt", gt_arrs)):
            if arrs is not None:
                buf = getattr(s, attr)
                if buf is None:
                    buf = ctx.alloc(n0 * self.B)
                    setattr(s, attr, buf)
                self._add_gather(step, keep, arrs, n0, buf.ptr)
                if attr == "sky":
                    sky_ptr = buf.ptr
                else:
                    gt_ptr, gt_images = buf.ptr, n
        step.detect = 1
        step.sky_dev, step.gt_dev, step.gt_images = sky_ptr, gt_ptr, gt_images
        step.foe, step.thr = self.foe_params, self.thr_params
        step.mask_fixed_dev, step.mask_dyn_dev, step.out_dev = s.mf.ptr, s.md.ptr, s.out.ptr
        step.off_counts_fixed, step.off_counts_dyn = self.B * 32, 2 * self.B * 32
        s.has_counts = gt_ptr is not None
        step.out_host, step.out_bytes = s.h_out.ctypes.data, (self._out_bytes if s.has_counts else n * 32)
        step.record_done = s.marker
        if self.worker:
            s.ticket = ctx.post_step(step)
        else:
            s.ticket = 0
            check(lib.mav_frame_step_dev(ctx.h, C.byref(step)))
        if posted is not None:
            posted()
        s.keep = keep
        s.busy = True
        return si

    def _wait_slot(self, s) -> None:
        try:
            if s.ticket:
                self.ctx.wait_step(s.ticket, s.marker)
            else:
                check(self.ctx.lib.mav_marker_wait(None, s.marker))
        finally:                                          # a step that failed has nothing more to wait for: the slot is free either way
            s.busy = False
            s.keep = None

    def collect(self, ticket: int) -> dict:
        """Wait for that batch (and only that batch) and return its records (n,) RESULT_DTYPE, the (n, 4) int64 counts of both masks
        (None without a ground truth) and, per pair, lazy handles of the masks (H, W) bool -- and of the flow, if the pipeline keeps
        it.  A handle costs nothing until it is looked at; it stays valid after the slot is re-used (the slot brings it over first)."""
        s = self.slots[ticket]
        if not s.busy:
            raise ValueError("this ticket has been collected already")
        ctx = self.ctx
        self._wait_slot(s)
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
        if not ctx.alive:
            return
        for s in self.slots:
            if s.busy:
                try:
                    self._wait_slot(s)
                except Exception:                             # noqa: BLE001 -- a failed step must not keep the buffers from going
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
def auto_lanes(W: int, H: int, batch: int = 1, uploads: bool = True) -> int:
    """Contexts a stream of `batch`-pair calls at this frame size is spread over.  Device-resident inputs (uploads=False): 3 up to
    100 MB of finest-layer sweep working set per call (80 B per pixel and pair), 2 up to 200 MB (one 1080p pair: 166 MB), 1 beyond.
    With the frames crossing PCIe inside every call (the reference-shaped loops: a lane's upload sits on its one stream, in front of
    its chain) one more lane fills those gaps: 4 up to 100 MB, 3 up to 200 MB.  One-frame loop, ms per frame with 2 / 3 / 4 / 5 lanes:
    1280x720 0.27 / 0.213 / 0.195 / 0.239, 1920x1080 0.477 / 0.443 / 0.47 (profiles/r06/lanes_api_probe.txt,
    lane_queue_priority.txt).  Four lanes need all four hardware queues of their class -- which they have since the lanes' streams
    live in a priority class of their own (_one_stream_per_lane); before that the fourth lane lost 20 % inside bench.py."""
    ws = 80 * W * H * batch
    if uploads:
        return 4 if ws <= (100 << 20) else (3 if ws <= (200 << 20) else 1)
    return 3 if ws <= (100 << 20) else (2 if ws <= (200 << 20) else 1)


LANE_STREAM_PRIORITY = -1     # high


def _one_stream_per_lane(ctxs) -> None:
    """Several lanes: every context keeps ALL its work -- uploads included -- on its one compute stream (option "inline_uploads"; the copy
    and pair streams of a context are only created when first used).  The HIP runtime maps streams onto a small pool of hardware queues
    (4 by default): streams beyond it share queues, two lanes whose streams share a queue do not overlap at all, and a lane whose copy
    stream sits on another lane's queue waits for that lane's kernels.  Measured with two lanes at 1080p, the one-frame loop: 0.48 ms
    per frame when the runtime happened to spread the six streams well, 0.61 (= one lane) after an earlier context had shifted the
    assignment, 0.79 with an eight-queue pool; one stream per lane makes it 0.47 - 0.48 in every order (profiles/r05/lanes_probe.txt).
    A single lane keeps its copy stream: there the upload of frame i + 1 overlaps the chain of frame i."""
    want = 1 if len(ctxs) > 1 else 0
    for c in ctxs:
        if c.get_option("inline_uploads") != want:        # (setting it drains the context's streams: not on a context that has it already)
            c.set_option("inline_uploads", want)
        # ... and the lanes' streams live in a priority class of their own (LANE_STREAM_PRIORITY): the runtime keeps a pool of hardware
        # queues per class, so which queues the lanes get no longer depends on what other streams the process has made -- with ONE idle
        # context created before the lanes the same three-lane loop ran at 0.31 instead of 0.215 ms per 720p frame
        # (profiles/r06/lane_queue_priority.txt)
        pr = LANE_STREAM_PRIORITY if len(ctxs) > 1 else 0
        if c.get_option("stream_priority") != pr:
            c.set_option("stream_priority", pr)


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
