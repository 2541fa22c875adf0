"""ctypes binding of libmavflow.so (include/mavflow.h).  No CPU fallback: if the library or a GPU is missing the
product path raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libmavflow.so")          # the one library the product loads; no environment override

MAV_OK, MAV_ERR_ARG, MAV_ERR_HIP, MAV_ERR_OOM, MAV_ERR_STATE = 0, -1, -2, -3, -4


class MavflowError(RuntimeError):
    """HIP / device failure inside libmavflow (the counterpart of cv2.error at the flow-operator seam)."""


class FbParams(C.Structure):
    _fields_ = [("pyr_scale", C.c_double), ("levels", C.c_int), ("winsize", C.c_int), ("iterations", C.c_int),
                ("poly_n", C.c_int), ("poly_sigma", C.c_double), ("flags", C.c_int)]


class FoeParams(C.Structure):
    _fields_ = [("n_pairs", C.c_int), ("mag_threshold", C.c_double), ("ransac_threshold", C.c_double)]


class ThrParams(C.Structure):
    _fields_ = [("fixed_deg", C.c_double), ("fixed_min_mag", C.c_double), ("dyn_min_mag", C.c_double),
                ("dyn_a", C.c_double), ("dyn_b", C.c_double), ("dyn_c", C.c_double)]


class Result(C.Structure):
    _fields_ = [("box", C.c_int32 * 4), ("foe", C.c_double * 2)]


RESULT_DTYPE = np.dtype([("box", np.int32, (4,)), ("foe", np.float64, (2,))])
assert RESULT_DTYPE.itemsize == C.sizeof(Result) == 32

# every symbol include/mavflow.h declares (tests check the library exports each of them)
EXPORTS = [
    "mav_fb_defaults", "mav_foe_defaults", "mav_thr_defaults", "mav_create", "mav_destroy", "mav_last_error",
    "mav_device_count", "mav_set_option", "mav_num_layers", "mav_layer_dims", "mav_farneback", "mav_derotate",
    "mav_foe_dense", "mav_ransac", "mav_bgr2gray", "mav_phi_mask", "mav_bbox", "mav_window_max", "mav_tpr_fpr_counts", "mav_process_batch",
    "mav_farneback_dev", "mav_process_batch_dev", "mav_sync", "mav_stream", "mav_dev_alloc", "mav_dev_free",
    "mav_memcpy_h2d", "mav_memcpy_d2h", "mav_host_alloc", "mav_host_free", "mav_upload_async", "mav_upload_fence", "mav_timer_start", "mav_timer_stop", "mav_profile_enable", "mav_profile_get", "mav_profile_busy",
    "mav_comm_unique_id", "mav_comm_init", "mav_comm_destroy", "mav_allgather_results", "mav_stage_blur_resize",
    "mav_stage_polyexp", "mav_stage_update_matrices", "mav_stage_blur_iter",
    "mav_analyze_pyramid", "mav_pyramid_levels", "mav_pyramid_dims", "mav_optimize_window", "mav_stage_pyramid_level",
    "mav_detect", "mav_detect_dev", "mav_last_flow_dev", "mav_foe_dense_f32", "mav_phi_mask_f32", "mav_stage_coefficients",
    "mav_stage_phi_mask", "mav_last_masks_tpr_fpr", "mav_get_option", "mav_schedule_info", "mav_stage_blur_resize_two_pass",
    "mav_membw_probe", "mav_runtime_info", "mav_upload_async_unordered", "mav_mem_info", "mav_profile_intervals",
    "mav_upload_gather", "mav_download_async", "mav_marker_create", "mav_marker_record", "mav_marker_wait", "mav_marker_destroy",
    "mav_tpr_fpr_counts_dev", "mav_bgr2gray_dev", "mav_png_unfilter", "mav_comm_count",
    "mav_marker_query", "mav_frame_step_dev", "mav_frame_step_post", "mav_frame_step_wait", "mav_worker_drain",
]

GATHER_ORDERED, GATHER_SOURCES_HELD = 1, 2      # mav_upload_gather flags
STEP_MAX_GATHER = 4


class Gather(C.Structure):
    _fields_ = [("src_host", C.POINTER(C.c_void_p)), ("count", C.c_int), ("bytes_each", C.c_size_t), ("dst_dev", C.c_void_p)]


class FrameStep(C.Structure):
    """mav_frame_step (include/mavflow.h): one iteration of the reference's loop as one call."""
    _fields_ = [("n", C.c_int),
                ("wait_before", C.POINTER(C.c_void_p)), ("n_wait_before", C.c_int),
                ("par_host", C.c_void_p), ("par_dev", C.c_void_p), ("par_bytes", C.c_size_t),
                ("gather", Gather * STEP_MAX_GATHER), ("n_gather", C.c_int),
                ("bgr_dev", C.c_void_p), ("n_bgr", C.c_int), ("gray_dev", C.c_void_p),
                ("compute_flow", C.c_int), ("prev_dev", C.c_void_p), ("next_dev", C.c_void_p), ("flow_dev", C.c_void_p),
                ("record_after_flow", C.POINTER(C.c_void_p)), ("n_record_after_flow", C.c_int),
                ("detect", C.c_int),
                ("off_samples", C.c_size_t), ("off_omega", C.c_size_t), ("off_dt", C.c_size_t), ("off_frame0", C.c_size_t),
                ("has_omega", C.c_int), ("has_frame0", C.c_int),
                ("sky_dev", C.c_void_p), ("gt_dev", C.c_void_p), ("gt_images", C.c_int),
                ("foe", FoeParams), ("thr", ThrParams),
                ("mask_fixed_dev", C.c_void_p), ("mask_dyn_dev", C.c_void_p), ("out_dev", C.c_void_p),
                ("off_counts_fixed", C.c_size_t), ("off_counts_dyn", C.c_size_t),
                ("out_host", C.c_void_p), ("out_bytes", C.c_size_t), ("record_done", C.c_void_p)]

_lib = None


def load(path: str | None = None) -> C.CDLL:
    """Load libmavflow.so (built in-tree by `make -C mav-detection_amd/csrc` / __graft_entry__.build()).
    path: an explicit other build of the same ABI, given by a diagnostic tool BEFORE anything else loads the library
    (tools/phase_stamps.py and its -DMAV_STAMPS build); the product never passes one."""
    global _lib
    if _lib is not None:
        if path is not None and os.path.abspath(path) != _lib._name:
            raise RuntimeError(f"libmavflow is already loaded from {_lib._name}")
        return _lib
    so = os.path.abspath(path) if path is not None else SO_PATH
    if not os.path.exists(so):
        raise ImportError(f"{so} is missing: build it with __graft_entry__.build(); there is no CPU fallback")
    lib = C.CDLL(so)
    lib.mav_last_error.restype = C.c_char_p
    lib.mav_stream.restype = C.c_void_p
    lib.mav_last_flow_dev.restype = C.c_void_p
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ("mav_last_error", "mav_stream", "mav_fb_defaults", "mav_foe_defaults", "mav_thr_defaults", "mav_last_flow_dev"):
            fn.restype = C.c_int
    lib.mav_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(FbParams)]
    lib.mav_destroy.argtypes = [C.c_void_p]
    lib.mav_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
    lib.mav_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_long)]
    lib.mav_schedule_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    lib.mav_runtime_info.argtypes = [C.c_char_p, C.c_size_t]
    lib.mav_membw_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    lib.mav_num_layers.argtypes = [C.c_void_p]
    lib.mav_mem_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 4
    lib.mav_layer_dims.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
    vp = C.c_void_p
    lib.mav_farneback.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.mav_farneback_dev.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.mav_derotate.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    lib.mav_foe_dense.argtypes = [vp, vp, vp, C.c_int, C.POINTER(FoeParams), vp]
    lib.mav_ransac.argtypes = [vp, vp, C.c_int, C.c_double, vp]
    lib.mav_bgr2gray.argtypes = [vp, vp, C.c_int, vp]
    lib.mav_phi_mask.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(ThrParams), vp, vp, vp, vp]
    lib.mav_bbox.argtypes = [vp, vp, C.c_int, vp]
    lib.mav_window_max.argtypes = [vp, vp, C.c_int, vp]
    lib.mav_tpr_fpr_counts.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
    lib.mav_last_masks_tpr_fpr.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    lib.mav_analyze_pyramid.argtypes = [vp, vp, C.c_int, C.c_double, vp]
    lib.mav_pyramid_levels.argtypes = [vp, C.c_double]
    lib.mav_pyramid_dims.argtypes = [vp, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mav_optimize_window.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    lib.mav_stage_pyramid_level.argtypes = [vp, vp, C.c_double, C.c_int, vp]
    # ctx, prev, next, samples, omega, dt, frame0, sky, batch, foe params, thr params, flow, phi, mask_fixed, mask_dyn, results
    pb = [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(FoeParams), C.POINTER(ThrParams), vp, vp, vp, vp, vp]
    lib.mav_process_batch.argtypes = pb
    lib.mav_process_batch_dev.argtypes = pb
    # ctx, flow, samples, omega, dt, frame0, sky, batch, foe params, thr params, phi, mask_fixed, mask_dyn, results
    dt_ = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(FoeParams), C.POINTER(ThrParams), vp, vp, vp, vp]
    lib.mav_detect.argtypes = dt_
    lib.mav_detect_dev.argtypes = dt_
    lib.mav_last_flow_dev.argtypes = [vp]
    lib.mav_foe_dense_f32.argtypes = [vp, vp, vp, C.c_int, C.POINTER(FoeParams), vp]
    lib.mav_phi_mask_f32.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(ThrParams), vp, vp, vp, vp]
    lib.mav_stage_coefficients.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    lib.mav_stage_phi_mask.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.POINTER(ThrParams), vp, vp, vp, vp]
    lib.mav_sync.argtypes = [vp]
    lib.mav_stream.argtypes = [vp]
    lib.mav_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.mav_dev_free.argtypes = [vp, vp]
    lib.mav_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mav_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mav_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.mav_host_free.argtypes = [vp, vp]
    lib.mav_upload_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mav_upload_async_unordered.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mav_upload_fence.argtypes = [vp]
    lib.mav_upload_gather.argtypes = [vp, vp, C.POINTER(vp), C.c_int, C.c_size_t, C.c_int]
    lib.mav_download_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mav_marker_create.argtypes = [vp, C.POINTER(vp)]
    lib.mav_marker_record.argtypes = [vp, vp]
    lib.mav_marker_wait.argtypes = [vp, vp]
    lib.mav_marker_destroy.argtypes = [vp, vp]
    lib.mav_marker_query.argtypes = [vp, vp, C.POINTER(C.c_int)]
    lib.mav_frame_step_dev.argtypes = [vp, C.POINTER(FrameStep)]
    lib.mav_frame_step_post.argtypes = [vp, C.POINTER(FrameStep), C.POINTER(C.c_uint64)]
    lib.mav_frame_step_wait.argtypes = [vp, C.c_uint64, vp]
    lib.mav_worker_drain.argtypes = [vp]
    lib.mav_tpr_fpr_counts_dev.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
    lib.mav_bgr2gray_dev.argtypes = [vp, vp, C.c_int, vp]
    lib.mav_png_unfilter.argtypes = [vp, C.c_int, C.c_size_t, C.c_int, vp]
    lib.mav_timer_start.argtypes = [vp]
    lib.mav_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.mav_profile_enable.argtypes = [vp, C.c_int]
    lib.mav_profile_get.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    lib.mav_profile_busy.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double)]
    lib.mav_profile_intervals.argtypes = [vp, C.POINTER(C.c_int), vp, vp, vp, vp]
    lib.mav_comm_unique_id.argtypes = [vp]
    lib.mav_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.mav_comm_destroy.argtypes = [vp]
    lib.mav_comm_count.argtypes = [vp, C.POINTER(C.c_int)]
    lib.mav_allgather_results.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.mav_stage_blur_resize.argtypes = [vp, vp, C.c_int, vp]
    lib.mav_stage_blur_resize_two_pass.argtypes = [vp, vp, C.c_int, vp]
    lib.mav_stage_polyexp.argtypes = [vp, vp, C.c_int, vp]
    lib.mav_stage_update_matrices.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    lib.mav_stage_blur_iter.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map an error code to the exception the reference raises at that seam."""
    if rc == MAV_OK:
        return
    msg = load().mav_last_error().decode("utf-8", "replace")
    if rc == MAV_ERR_ARG:
        raise ValueError(msg)
    if rc == MAV_ERR_OOM:
        raise MemoryError(msg)
    raise MavflowError(f"[{rc}] {msg}")


def fb_defaults(levels: int | None = None) -> FbParams:
    p = FbParams()
    load().mav_fb_defaults(C.byref(p))
    if levels is not None:
        p.levels = levels
    return p


def foe_defaults() -> FoeParams:
    p = FoeParams()
    load().mav_foe_defaults(C.byref(p))
    return p


def thr_defaults() -> ThrParams:
    p = ThrParams()
    load().mav_thr_defaults(C.byref(p))
    return p


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype, shape=None, name="array"):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


class _PinnedPool:
    """Page-locked host memory for the arrays the host-pointer entry points hand back.  A fresh numpy array is pageable and
    untouched: a 16.6 MB flow field costs ~4000 first-touch page faults plus the runtime's staging copy, more than the GPU
    needs to compute it.  Results are therefore written into page-locked blocks (PCIe rate, no faults) that return to this
    pool when the last numpy array or view over them is garbage-collected.
    Module-level: a block lent out survives the context that allocated it."""

    CAP_BYTES = 2 << 30                                  # idle blocks kept for re-use: at most this much page-locked memory in all

    def __init__(self):
        self.free = {}                                   # nbytes -> [ptr, ...]
        self.idle_bytes = 0
        self.stamp = {}                                  # nbytes -> tick of the last use of that size (least recently used goes first)
        self.tick = 0
        # A block that is the TARGET of a device -> host copy still enqueued (a retired DeviceArray, pipeline.py) must not be lent out
        # again before that copy has landed, even if every array over it has been dropped: guard[ptr] = the marker recorded behind
        # the copy; a returning block whose marker has not fired waits in `pending` (checked with mav_marker_query, never blocking).
        self.guard = {}                                  # ptr -> marker object (has .done())
        self.pending = []                                # [(marker, nbytes, ptr), ...]

    def guard_until(self, arr: np.ndarray, marker) -> None:
        """`arr` (from empty()) is being written by a copy that completes when `marker.done()`: keep its block out of circulation until then."""
        base = arr
        while isinstance(getattr(base, "base", None), np.ndarray):
            base = base.base
        self.guard[base.ctypes.data] = marker

    def _reap(self) -> None:
        if self.pending:
            still = []
            for marker, nbytes, ptr in self.pending:
                if marker.done():
                    self._shelve(nbytes, ptr)
                else:
                    still.append((marker, nbytes, ptr))
            self.pending = still

    def empty(self, ctx: "Context", shape, dtype) -> np.ndarray:
        import weakref
        dtype = np.dtype(dtype)
        nbytes = max(1, int(np.prod(shape)) * dtype.itemsize)
        self.tick += 1
        self.stamp[nbytes] = self.tick
        self._reap()
        lst = self.free.get(nbytes)
        if lst:
            ptr = lst.pop()
            self.idle_bytes -= nbytes
        else:
            p = C.c_void_p()
            check(ctx.lib.mav_host_alloc(ctx._h, nbytes, C.byref(p)))
            ptr = p.value
        # the finalizer hangs on the ctypes object that OWNS the memory in numpy's eyes: every array or view derived from it
        # (numpy collapses view chains onto the owner) keeps it alive, so the block returns only when the last of them is gone
        owner = (C.c_uint8 * nbytes).from_address(ptr)
        weakref.finalize(owner, self._give, nbytes, ptr).atexit = False      # at interpreter exit the OS reclaims; no HIP calls then
        return np.ctypeslib.as_array(owner).view(dtype)[:int(np.prod(shape))].reshape(shape)

    def _give(self, nbytes, ptr):
        """A block comes back (the last array over it is gone) -- to the shelf, or, while a copy into it is still enqueued, to `pending`."""
        marker = self.guard.pop(ptr, None)
        if marker is not None and not marker.done():
            self.pending.append((marker, nbytes, ptr))
            return
        self._shelve(nbytes, ptr)

    def _shelve(self, nbytes, ptr):
        """At most 4 idle blocks per size and CAP_BYTES idle in all: a long-running caller with changing batch
        sizes (a 64-pair 1080p flow block is 1 GB) must not pile up page-locked memory -- blocks of the least recently used sizes
        are released first, then the returning block itself if it alone exceeds the cap."""
        lst = self.free.setdefault(nbytes, [])
        if len(lst) >= 4 or nbytes > self.CAP_BYTES:
            load().mav_host_free(None, ptr)
            return
        lst.append(ptr)
        self.idle_bytes += nbytes
        for size in sorted(self.free, key=lambda k: self.stamp.get(k, 0)):
            while self.idle_bytes > self.CAP_BYTES and self.free[size] and not (size == nbytes and len(self.free[size]) == 1):
                load().mav_host_free(None, self.free[size].pop())
                self.idle_bytes -= size
        if self.idle_bytes > self.CAP_BYTES:             # only this block's own size is left
            load().mav_host_free(None, lst.pop())
            self.idle_bytes -= nbytes


_pinned = _PinnedPool()


class DeviceBuffer:
    """A hipMalloc'd buffer owned through the C-ABI (bench / multi-GPU path)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        check(ctx.lib.mav_dev_alloc(ctx._h, self.nbytes, C.byref(p)))      # (an allocation touches no stream: no need to drain the worker)
        self.ptr = p.value

    def upload(self, a: np.ndarray):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(self.ctx.lib.mav_memcpy_h2d(self.ctx.h, self.ptr, _ptr(a), a.nbytes))
        return self

    def download(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(self.ctx.lib.mav_memcpy_d2h(self.ctx.h, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.lib.mav_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """One mav_ctx: (device, W, H, max_batch, Farneback parameters)."""

    def __init__(self, W: int, H: int, max_batch: int = 1, fb: FbParams | None = None, device: int = 0):
        self.lib = load()
        self.W, self.H, self.max_batch = int(W), int(H), int(max_batch)
        self.fb = fb if fb is not None else fb_defaults()
        h = C.c_void_p()
        check(self.lib.mav_create(C.byref(h), device, self.W, self.H, self.max_batch, C.byref(self.fb)))
        self._h = h
        self._posted = False                  # steps have been posted to the context's worker thread since the last drain

    # A context is single-threaded.  Once a step has been posted (post_step) the library's worker thread is that thread until it has
    # enqueued everything posted; `h` -- what every other call of this binding passes as the context -- therefore drains the worker
    # first.  post_step / wait_step use the raw handle.
    @property
    def h(self):
        if self._posted:
            self.drain()
        return self._h

    @h.setter
    def h(self, v):
        self._h = v

    @property
    def alive(self) -> bool:
        """Not closed (asks nothing of the worker thread, unlike `h`)."""
        return bool(self._h)

    def post_step(self, step: "FrameStep") -> int:
        """mav_frame_step_post: hand one loop iteration to the context's worker thread; returns its ticket at once.  The caller keeps
        the step's host buffers alive until wait_step(ticket, marker) has returned."""
        t = C.c_uint64()
        check(self.lib.mav_frame_step_post(self._h, C.byref(step), C.byref(t)))
        self._posted = True
        return t.value

    def wait_step(self, ticket: int, marker=None) -> None:
        """The step has been enqueued and, with its record_done marker given, has finished on the device; raises what the step raised."""
        check(self.lib.mav_frame_step_wait(self._h, ticket, marker))

    def drain(self) -> None:
        """Every posted step has been enqueued (mav_worker_drain); raises the first failure among them."""
        self._posted = False
        check(self.lib.mav_worker_drain(self._h))

    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):
                self.lib.mav_host_free(self._h, p)
            self._pinned = []
            self.lib.mav_destroy(self._h)            # (the worker finishes the step in hand, drops what is still queued, joins)
            self._h = None
            self._posted = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- introspection ---------------------------------------------------------------------------------------
    def set_option(self, name: str, value: int):
        check(self.lib.mav_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_long()
        check(self.lib.mav_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def membw_probe(self, bytes_per_buffer: int, reps: int = 20) -> float:
        """GB/s of a plain 3-reads-1-write streaming kernel over four buffers of that size (calibration for bench.py's roofline)."""
        g = C.c_double()
        check(self.lib.mav_membw_probe(self.h, int(bytes_per_buffer), int(reps), C.byref(g)))
        return g.value

    def schedule_info(self, batch: int) -> dict:
        """The schedule a call of `batch` pairs takes with the options in effect (every option, group split, per-layer plan)."""
        import json
        buf = C.create_string_buffer(8192)
        check(self.lib.mav_schedule_info(self.h, int(batch), buf, len(buf)))
        return json.loads(buf.value.decode())

    def mem_info(self) -> dict:
        """Device memory in bytes: free / total of the GPU, what this context holds in all, and its Farneback workspace alone
        (0 until a call computes flow)."""
        v = [C.c_size_t() for _ in range(4)]
        check(self.lib.mav_mem_info(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("dev_free", "dev_total", "ctx_bytes", "workspace_bytes"), (x.value for x in v)))

    def num_layers(self) -> int:
        return self.lib.mav_num_layers(self.h)

    def layer_dims(self, k: int):
        w, h, ks, sg = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        check(self.lib.mav_layer_dims(self.h, k, C.addressof(w), C.addressof(h), C.addressof(ks), C.addressof(sg)))
        return w.value, h.value, sg.value, ks.value

    def sync(self):
        check(self.lib.mav_sync(self.h))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def pinned_like(self, a: np.ndarray) -> np.ndarray:
        """A page-locked host copy of `a` (numpy view over hipHostMalloc memory; freed with the context)."""
        p = C.c_void_p()
        check(self.lib.mav_host_alloc(self.h, a.nbytes, C.byref(p)))
        self._pinned = getattr(self, "_pinned", []) + [p.value]
        out = np.ctypeslib.as_array((C.c_uint8 * a.nbytes).from_address(p.value)).view(a.dtype).reshape(a.shape)
        out[...] = a
        return out

    def upload_async(self, dst: DeviceBuffer, src: np.ndarray, ordered: bool = True):
        """Copy on the copy stream.  ordered (default): behind everything enqueued on the compute stream so far (safe for a buffer
        set an earlier batch may still read); ordered=False: no wait -- for a destination no enqueued work touches."""
        fn = self.lib.mav_upload_async if ordered else self.lib.mav_upload_async_unordered
        check(fn(self.h, dst.ptr, _ptr(src), src.nbytes))

    def upload_fence(self):
        check(self.lib.mav_upload_fence(self.h))

    # -- host-array entry points -----------------------------------------------------------------------------
    def _imgs(self, a, name):
        a = np.asarray(a)
        if a.ndim == 2:
            a = a[None]
        if a.ndim != 3 or a.shape[1:] != (self.H, self.W):
            raise ValueError(f"{name}: expected (batch, {self.H}, {self.W}) u8, got {a.shape}")
        if a.dtype != np.uint8:
            raise ValueError(f"{name}: expected uint8, got {a.dtype}")
        return np.ascontiguousarray(a)

    def farneback(self, prev, nxt) -> np.ndarray:
        prev, nxt = self._imgs(prev, "prev"), self._imgs(nxt, "next")
        if prev.shape != nxt.shape:
            raise ValueError("prev and next differ in shape")
        B = prev.shape[0]
        flow = _pinned.empty(self, (B, self.H, self.W, 2), np.float32)
        check(self.lib.mav_farneback(self.h, _ptr(prev), _ptr(nxt), B, _ptr(flow)))
        return flow

    def farneback_sequence(self, frames) -> np.ndarray:
        """Flow of every consecutive pair of a run of frames (n + 1, H, W) u8 -> (n, H, W, 2) float32.  The two batches handed to
        the library are views of the one array (next = prev + one frame), which it recognises: the run is uploaded once and every
        frame is blurred and expanded once instead of twice.  Same flow, bit for bit, as farneback(frames[:-1], frames[1:]) on
        separate copies."""
        frames = self._imgs(frames, "frames")
        if frames.shape[0] < 2:
            raise ValueError("a sequence needs at least two frames")
        return self.farneback(frames[:-1], frames[1:])

    def derotate(self, flow, omega, dt) -> np.ndarray:
        flow = np.asarray(flow, np.float32)
        flow = flow[None] if flow.ndim == 3 else flow
        B = flow.shape[0]
        flow = _arr(flow, np.float32, (B, self.H, self.W, 2), "flow")
        omega = _arr(np.asarray(omega, np.float64).reshape(B, 3), np.float64)
        dt = _arr(np.asarray(dt, np.float64).reshape(B), np.float64)
        out = np.empty((B, self.H, self.W, 2), np.float64)
        check(self.lib.mav_derotate(self.h, _ptr(flow), _ptr(omega), _ptr(dt), B, _ptr(out)))
        return out

    def foe_dense(self, flow, samples, params: FoeParams | None = None) -> np.ndarray:
        """get_FOE_dense + ransac.  The arithmetic follows the array's dtype as numpy's does in the reference: a float32
        field (frame index 0, never derotated) has its |flow2| gate evaluated in float32, anything else runs in double."""
        flow = np.asarray(flow)
        f32 = flow.dtype == np.float32
        if not f32:
            flow = np.asarray(flow, np.float64)
        flow = flow[None] if flow.ndim == 3 else flow
        B = flow.shape[0]
        p = params or foe_defaults()
        flow = _arr(flow, flow.dtype, (B, self.H, self.W, 2), "flow")
        samples = _arr(np.asarray(samples).reshape(B, 2 * p.n_pairs, 2), np.uint32)
        foe = np.empty((B, 2), np.float64)
        fn = self.lib.mav_foe_dense_f32 if f32 else self.lib.mav_foe_dense
        check(fn(self.h, _ptr(flow), _ptr(samples), B, C.byref(p), _ptr(foe)))
        return foe

    def ransac(self, estimates, ransac_threshold: float = 30.0):
        est = _arr(np.asarray(estimates, np.float64).reshape(-1, 2), np.float64)
        foe = np.empty(2, np.float64)
        check(self.lib.mav_ransac(self.h, _ptr(est) if est.shape[0] else None, est.shape[0], float(ransac_threshold), _ptr(foe)))
        return (float(foe[0]), float(foe[1]))

    def bgr2gray(self, bgr) -> np.ndarray:
        a = np.asarray(bgr)
        a = a[None] if a.ndim == 3 else a
        a = _arr(a, np.uint8, (a.shape[0], self.H, self.W, 3), "bgr")
        gray = np.empty((a.shape[0], self.H, self.W), np.uint8)
        check(self.lib.mav_bgr2gray(self.h, _ptr(a), a.shape[0], _ptr(gray)))
        return gray

    def phi_mask(self, flow, foe, sky=None, params: ThrParams | None = None, want_phi=True):
        """get_phi + the threshold block.  dtype in = dtype of the arithmetic and of phi, as in the reference: float32 flow
        (frame index 0) -> float32 phi, float64 flow -> float64 phi."""
        flow = np.asarray(flow)
        f32 = flow.dtype == np.float32
        ft = np.float32 if f32 else np.float64
        flow = np.asarray(flow, ft)
        flow = flow[None] if flow.ndim == 3 else flow
        B = flow.shape[0]
        flow = _arr(flow, ft, (B, self.H, self.W, 2), "flow")
        foe = _arr(np.asarray(foe, np.float64).reshape(B, 2), np.float64)
        sky = None if sky is None else _arr(np.asarray(sky).reshape(B, self.H, self.W).astype(np.uint8), np.uint8)
        p = params or thr_defaults()
        phi = np.empty((B, self.H, self.W), ft) if want_phi else None
        mf = np.empty((B, self.H, self.W), np.uint8)
        md = np.empty((B, self.H, self.W), np.uint8)
        mx = np.empty(B, ft) if want_phi else None      # max(phi) needs the exact path, like phi itself
        fn = self.lib.mav_phi_mask_f32 if f32 else self.lib.mav_phi_mask
        check(fn(self.h, _ptr(flow), _ptr(foe), _ptr(sky), B, C.byref(p), _ptr(phi), _ptr(mf), _ptr(md), _ptr(mx)))
        return phi, mf.view(np.bool_), md.view(np.bool_), mx

    def bbox(self, img) -> np.ndarray:
        img = self._imgs(img, "img")
        box = np.empty((img.shape[0], 4), np.int32)
        check(self.lib.mav_bbox(self.h, _ptr(img), img.shape[0], _ptr(box)))
        return box

    def window_max(self, img) -> np.ndarray:
        img = self._imgs(img, "img")
        out = np.empty((img.shape[0], 3), np.int64)
        check(self.lib.mav_window_max(self.h, _ptr(img), img.shape[0], _ptr(out)))
        return out

    def pyramid_dims(self, scale: float = 1.5):
        """[(w, h)] of every level of im_helpers.pyramid for this frame size."""
        n = self.lib.mav_pyramid_levels(self.h, scale)
        if n < 0:
            check(n)
        dims = []
        for l in range(n):
            w, h = C.c_int(), C.c_int()
            check(self.lib.mav_pyramid_dims(self.h, scale, l, C.byref(w), C.byref(h)))
            dims.append((w.value, h.value))
        return dims

    def analyze_pyramid(self, img, scale: float = 1.5) -> np.ndarray:
        """(batch, 6) int64: score, x, y, level, argmax_row, argmax_col (detector.py:280-312, all levels)."""
        img = self._imgs(img, "img")
        out = np.empty((img.shape[0], 6), np.int64)
        check(self.lib.mav_analyze_pyramid(self.h, _ptr(img), img.shape[0], scale, _ptr(out)))
        return out

    def pyramid_level(self, img, level: int, scale: float = 1.5) -> np.ndarray:
        img = self._imgs(img, "img")
        w, h = C.c_int(), C.c_int()
        check(self.lib.mav_pyramid_dims(self.h, scale, level, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.uint8)
        check(self.lib.mav_stage_pyramid_level(self.h, _ptr(img[:1]), scale, level, _ptr(out)))
        return out

    def optimize_window(self, img, windows):
        """Detector.optimize_window (detector.py:314-358): windows (batch, 4) = x, y, w, h -> (scores int64, windows int32)."""
        img = self._imgs(img, "img")
        B = img.shape[0]
        win = _arr(np.asarray(windows).reshape(B, 4), np.int32)
        score = np.empty(B, np.int64)
        out = np.empty((B, 4), np.int32)
        check(self.lib.mav_optimize_window(self.h, _ptr(img), B, _ptr(win), _ptr(score), _ptr(out)))
        return score, out

    def tpr_fpr_counts(self, gt, mask, mask_value: int = 255) -> np.ndarray:
        """(batch, 4) = positives, negatives, true positives, false positives of calculate_tpr_fpr(gt, mask_value * mask)."""
        gt = self._imgs(gt, "gt")
        mask = self._imgs(np.asarray(mask).astype(np.uint8), "mask")
        out = np.empty((gt.shape[0], 4), np.int64)
        check(self.lib.mav_tpr_fpr_counts(self.h, _ptr(gt), _ptr(mask), int(mask_value), gt.shape[0], _ptr(out)))
        return out

    def _detect_args(self, B, samples, omega, dt, frame0, sky, fp):
        samples = _arr(np.asarray(samples).reshape(B, 2 * fp.n_pairs, 2), np.uint32)
        omega = None if omega is None else _arr(np.asarray(omega, np.float64).reshape(B, 3), np.float64)
        dt = None if dt is None else _arr(np.asarray(dt, np.float64).reshape(B), np.float64)
        frame0 = None if frame0 is None else _arr(np.asarray(frame0).reshape(B).astype(np.uint8), np.uint8)
        sky = None if sky is None else _arr(np.asarray(sky).reshape(B, self.H, self.W).astype(np.uint8), np.uint8)
        return samples, omega, dt, frame0, sky

    def last_masks_tpr_fpr(self, gt, mask_value: int = 255):
        """calculate_tpr_fpr counts of BOTH masks of the most recent detect / process_batch / phi_mask call, taken where they
        still are (on the device): ((batch, 4) fixed, (batch, 4) dynamic) = positives, negatives, true / false positives."""
        gt = self._imgs(gt, "gt")
        B = gt.shape[0]
        cf, cd = np.empty((B, 4), np.int64), np.empty((B, 4), np.int64)
        check(self.lib.mav_last_masks_tpr_fpr(self.h, _ptr(gt), int(mask_value), B, _ptr(cf), _ptr(cd)))
        return cf, cd

    def process_batch(self, prev, nxt, samples, omega=None, dt=None, sky=None, foe_params=None, thr_params=None,
                      want_flow=True, want_phi=False, want_masks=True, frame0=None):
        """Fused loop body of Processor.run_detection (processor.py:305-341) for a batch of pairs.  frame0: per-pair flags of
        the reference's frame index 0 (no derotation, float32 arithmetic; detector.py:80-81)."""
        prev, nxt = self._imgs(prev, "prev"), self._imgs(nxt, "next")
        B = prev.shape[0]
        fp = foe_params or foe_defaults()
        tp = thr_params or thr_defaults()
        samples, omega, dt, frame0, sky = self._detect_args(B, samples, omega, dt, frame0, sky, fp)
        flow = _pinned.empty(self, (B, self.H, self.W, 2), np.float32) if want_flow else None
        phi = _pinned.empty(self, (B, self.H, self.W), np.float64) if want_phi else None
        mf = _pinned.empty(self, (B, self.H, self.W), np.uint8) if want_masks else None
        md = _pinned.empty(self, (B, self.H, self.W), np.uint8) if want_masks else None
        res = np.empty(B, RESULT_DTYPE)
        check(self.lib.mav_process_batch(self.h, _ptr(prev), _ptr(nxt), _ptr(samples), _ptr(omega), _ptr(dt), _ptr(frame0), _ptr(sky),
                                         B, C.byref(fp), C.byref(tp), _ptr(flow), _ptr(phi), _ptr(mf), _ptr(md), _ptr(res)))
        return dict(flow=flow, phi=phi, mask_fixed=None if mf is None else mf.view(np.bool_),
                    mask_dyn=None if md is None else md.view(np.bool_), results=res)

    def detect(self, flow, samples, omega=None, dt=None, sky=None, foe_params=None, thr_params=None, want_phi=False,
               want_masks=True, frame0=None):
        """processor.py:305-341 from the reference's own flow seam: a float32 (B, H, W, 2) field (Dataset.get_flow_uv) in,
        derotation -> FoE -> phi -> masks -> box on the device, masks and records out.
        The field must BE float32 (what .flo files and Farneback give): the reference evaluates a float64 field in float64 from the
        start, which this fused call does not do -- such input goes through derotate / foe_dense / phi_mask (the float64 kernels),
        as Processor.run_detection does by itself."""
        flow = np.asarray(flow)
        if flow.dtype != np.float32:
            raise TypeError(f"detect() takes a float32 flow field, got {flow.dtype}: narrowing would change FoE and masks against the "
                            "reference; use derotate / foe_dense / phi_mask for float64 fields")
        flow = flow[None] if flow.ndim == 3 else flow
        B = flow.shape[0]
        flow = _arr(flow, np.float32, (B, self.H, self.W, 2), "flow")
        fp = foe_params or foe_defaults()
        tp = thr_params or thr_defaults()
        samples, omega, dt, frame0, sky = self._detect_args(B, samples, omega, dt, frame0, sky, fp)
        phi = _pinned.empty(self, (B, self.H, self.W), np.float64) if want_phi else None
        mf = _pinned.empty(self, (B, self.H, self.W), np.uint8) if want_masks else None
        md = _pinned.empty(self, (B, self.H, self.W), np.uint8) if want_masks else None
        res = np.empty(B, RESULT_DTYPE)
        check(self.lib.mav_detect(self.h, _ptr(flow), _ptr(samples), _ptr(omega), _ptr(dt), _ptr(frame0), _ptr(sky), B,
                                  C.byref(fp), C.byref(tp), _ptr(phi), _ptr(mf), _ptr(md), _ptr(res)))
        return dict(phi=phi, mask_fixed=None if mf is None else mf.view(np.bool_),
                    mask_dyn=None if md is None else md.view(np.bool_), results=res)

    # -- device-pointer path (bench, multi-GPU) --------------------------------------------------------------
    def process_batch_dev(self, prev_ptr, next_ptr, samples_ptr, batch, results_ptr, flow_ptr=None, omega_ptr=None,
                          dt_ptr=None, sky_ptr=None, phi_ptr=None, mf_ptr=None, md_ptr=None, foe_params=None,
                          thr_params=None, frame0_ptr=None):
        fp = foe_params or foe_defaults()
        tp = thr_params or thr_defaults()
        check(self.lib.mav_process_batch_dev(self.h, prev_ptr, next_ptr, samples_ptr, omega_ptr, dt_ptr, frame0_ptr, sky_ptr, batch,
                                             C.byref(fp), C.byref(tp), flow_ptr, phi_ptr, mf_ptr, md_ptr, results_ptr))

    def last_flow(self, pair: int) -> np.ndarray:
        """Flow field (H, W, 2) float32 of pair `pair` of the most recent process_batch_dev / farneback_dev call, downloaded
        from wherever that call wrote it (the caller's buffer or the context's workspace)."""
        p = self.lib.mav_last_flow_dev(self.h)
        if not p:
            raise ValueError("no flow has been computed on this context yet")
        if not 0 <= pair < self.max_batch:
            raise ValueError(f"pair {pair} outside [0, {self.max_batch})")
        out = np.empty((self.H, self.W, 2), np.float32)
        check(self.lib.mav_memcpy_d2h(self.h, _ptr(out), p + pair * out.nbytes, out.nbytes))
        return out

    def farneback_dev(self, prev_ptr, next_ptr, batch, flow_ptr):
        check(self.lib.mav_farneback_dev(self.h, prev_ptr, next_ptr, batch, flow_ptr))

    # -- multi-GPU record exchange (RCCL through the library, on the context's stream) ------------------------
    def comm_unique_id(self) -> np.ndarray:
        """A fresh ncclUniqueId (128 bytes, uint8) -- rank 0 creates it, every rank passes the same bytes to comm_init."""
        uid = np.zeros(128, np.uint8)
        check(self.lib.mav_comm_unique_id(_ptr(uid)))
        return uid

    def comm_init(self, uid, rank: int, nranks: int):
        uid = np.ascontiguousarray(np.asarray(uid, np.uint8).reshape(128))
        comm = C.c_void_p()
        check(self.lib.mav_comm_init(self.h, _ptr(uid), int(rank), int(nranks), C.byref(comm)))
        return comm

    def allgather(self, comm, local_ptr, bytes_per_rank: int, all_ptr):
        """ncclAllGather of bytes_per_rank bytes from every rank, enqueued on the context's stream (no host sync)."""
        check(self.lib.mav_allgather_results(self.h, comm, local_ptr, int(bytes_per_rank), all_ptr))

    def comm_count(self, comm) -> int:
        n = C.c_int()
        check(self.lib.mav_comm_count(comm, C.byref(n)))
        return n.value

    def comm_destroy(self, comm):
        check(self.lib.mav_comm_destroy(comm))

    def timer_start(self):
        check(self.lib.mav_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self.lib.mav_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        check(self.lib.mav_profile_enable(self.h, int(on)))

    def profile_get(self) -> dict:
        n = C.c_int(16)
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        cnt = (C.c_long * 16)()
        check(self.lib.mav_profile_get(self.h, C.byref(n), names, ms, cnt))
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n.value)}

    def profile_intervals(self):
        """(class index, stream, t0 ms, t1 ms) arrays of every profiled launch / run since profile_enable; class names in
        profile_get()'s order."""
        n = C.c_int(0)
        check(self.lib.mav_profile_intervals(self.h, C.byref(n), None, None, None, None))
        k, st = np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        t0, t1 = np.empty(n.value, np.float32), np.empty(n.value, np.float32)
        check(self.lib.mav_profile_intervals(self.h, C.byref(n), _ptr(k), _ptr(st), _ptr(t0), _ptr(t1)))
        return k[:n.value], st[:n.value], t0[:n.value], t1[:n.value]

    def profile_busy(self, *names: str) -> float:
        """ms during which at least one launch of the named kernel classes was running (union of the launches' intervals)."""
        out = C.c_double()
        check(self.lib.mav_profile_busy(self.h, ",".join(names).encode(), C.byref(out)))
        return out.value

    # -- stage hooks (parity tests) --------------------------------------------------------------------------
    def stage_phi_mask(self, flow32, foe, omega=None, dt=None, sky=None, params: ThrParams | None = None, want_phi=False):
        """The phi / mask / box stage of the fused path (float32 flow, double arithmetic, screen on unless phi is wanted)
        with a caller-supplied FoE."""
        flow = np.asarray(flow32, np.float32)
        flow = flow[None] if flow.ndim == 3 else flow
        B = flow.shape[0]
        flow = _arr(flow, np.float32, (B, self.H, self.W, 2), "flow")
        foe = _arr(np.asarray(foe, np.float64).reshape(B, 2), np.float64)
        omega = None if omega is None else _arr(np.asarray(omega, np.float64).reshape(B, 3), np.float64)
        dt = None if dt is None else _arr(np.asarray(dt, np.float64).reshape(B), np.float64)
        sky = None if sky is None else _arr(np.asarray(sky).reshape(B, self.H, self.W).astype(np.uint8), np.uint8)
        p = params or thr_defaults()
        phi = np.empty((B, self.H, self.W), np.float64) if want_phi else None
        mf = np.empty((B, self.H, self.W), np.uint8)
        md = np.empty((B, self.H, self.W), np.uint8)
        box = np.empty((B, 4), np.int32)
        check(self.lib.mav_stage_phi_mask(self.h, _ptr(flow), _ptr(foe), _ptr(omega), _ptr(dt), _ptr(sky), B, C.byref(p), _ptr(phi),
                                          _ptr(mf), _ptr(md), _ptr(box)))
        return phi, mf.view(np.bool_), md.view(np.bool_), box

    def stage_coefficients(self, k: int = -1):
        """The constants the flow kernels use: dict(g, xg, xxg (poly_n + 1 each, centre first), ig = [ig11, ig03, ig33, ig55],
        blur = layer k's Gaussian taps when k >= 0)."""
        n = self.fb.poly_n
        g, xg, xxg = (np.empty(n + 1, np.float32) for _ in range(3))
        ig = np.empty(4, np.float32)
        blur = np.empty(self.layer_dims(k)[3], np.float32) if k >= 0 else None
        check(self.lib.mav_stage_coefficients(self.h, k, _ptr(g), _ptr(xg), _ptr(xxg), _ptr(ig), _ptr(blur)))
        return dict(g=g, xg=xg, xxg=xxg, ig=ig, blur=blur)

    def stage_blur_resize(self, img, k, two_pass=False):
        img = _arr(img, np.uint8, (self.H, self.W), "img")
        w, h, _, _ = self.layer_dims(k)
        out = np.empty((h, w), np.float32)
        fn = self.lib.mav_stage_blur_resize_two_pass if two_pass else self.lib.mav_stage_blur_resize
        check(fn(self.h, _ptr(img), k, _ptr(out)))
        return out

    def stage_polyexp(self, I, k):
        w, h, _, _ = self.layer_dims(k)
        I = _arr(I, np.float32, (h, w), "I")
        R = np.empty((5, h, w), np.float32)
        check(self.lib.mav_stage_polyexp(self.h, _ptr(I), k, _ptr(R)))
        return R

    def stage_update_matrices(self, R0, R1, flow, k):
        w, h, _, _ = self.layer_dims(k)
        R0 = _arr(R0, np.float32, (5, h, w), "R0"); R1 = _arr(R1, np.float32, (5, h, w), "R1")
        flow = _arr(flow, np.float32, (h, w, 2), "flow")
        M = np.empty((5, h, w), np.float32)
        check(self.lib.mav_stage_update_matrices(self.h, _ptr(R0), _ptr(R1), _ptr(flow), k, _ptr(M)))
        return M

    def stage_blur_iter(self, R0, R1, M, k, update=True):
        w, h, _, _ = self.layer_dims(k)
        R0 = _arr(R0, np.float32, (5, h, w), "R0"); R1 = _arr(R1, np.float32, (5, h, w), "R1")
        M = _arr(M, np.float32, (5, h, w), "M")
        flow = np.empty((h, w, 2), np.float32)
        Mo = np.empty((5, h, w), np.float32)
        check(self.lib.mav_stage_blur_iter(self.h, _ptr(R0), _ptr(R1), _ptr(M), k, int(bool(update)), _ptr(flow), _ptr(Mo)))
        return flow, (Mo if update else None)


def runtime_info() -> dict:
    """HIP version of the build, HIP runtime / driver versions of this process, RCCL version once loaded (mav_runtime_info)."""
    import json
    buf = C.create_string_buffer(512)
    check(load().mav_runtime_info(buf, len(buf)))
    return json.loads(buf.value.decode())


def device_count() -> int:
    return load().mav_device_count()
