/*
 * mavflow.h -- C-ABI of libmavflow.so: the MI355X (gfx950) replacement for the per-frame-pair vision math of
 * evroon/mav-detection (Farneback flow -> derotation -> FoE fit -> phi -> threshold masks -> box).
 *
 * The reference is pure Python with no FFI of its own; each entry point below replaces the Python call cited
 * next to it (paths relative to the reference root) and is bound through ctypes (INTEGRATION.md shows the stub
 * a maintainer adds on the reference side).
 *
 * Conventions
 *   - plain C, no torch types; every buffer is caller-allocated, C-contiguous, never retained past the call.
 *   - entry points without a suffix take HOST pointers and are synchronous; `_dev` variants take DEVICE
 *     pointers (hipMalloc'd by anyone in this process, e.g. mav_dev_alloc or a torch tensor's data_ptr()),
 *     enqueue on the context's stream and return without synchronising (call mav_sync).
 *   - return value: 0 = OK, <0 = error (MAV_ERR_*); mav_last_error() gives the message. No abort(), no C++
 *     exceptions cross the boundary.
 *   - a mav_ctx is single-threaded (as the reference's loop is); distinct contexts are independent.
 *   - images are (batch, H, W) u8; flow is (batch, H, W, 2) interleaved (u, v); masks are (batch, H, W) u8 0/1.
 */
#ifndef MAVFLOW_H
#define MAVFLOW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAV_OK 0
#define MAV_ERR_ARG (-1)   /* bad argument (NULL, size mismatch, pyr_scale >= 1 ...): cv2.error / ValueError in the shim */
#define MAV_ERR_HIP (-2)   /* a HIP runtime call failed */
#define MAV_ERR_OOM (-3)   /* device allocation failed */
#define MAV_ERR_STATE (-4) /* context unusable / no GPU */

typedef struct mav_ctx mav_ctx;

/* src/farneback.py:76-80 -- the literal argument list of cv2.calcOpticalFlowFarneback.
 * Defaults (mav_fb_defaults): 0.4, 1, 12, 10, 8, 1.2, 0. Only flags == 0 (box window) is implemented. */
typedef struct {
    double pyr_scale;
    int levels, winsize, iterations, poly_n;
    double poly_sigma;
    int flags;
} mav_fb_params;

/* src/focus_of_expansion.py:21-23,67 -- N = 1000 line pairs, |flow2| gate 2.5, RANSAC radius 30 px. */
typedef struct {
    int n_pairs;
    double mag_threshold, ransac_threshold;
} mav_foe_params;

/* src/processor.py:333-341 -- fixed: phi*(mag > fixed_min_mag)*~sky > fixed_deg;
 * dynamic: (mag > dyn_min_mag) * ~sky * (phi > dyn_a + (dyn_b + dyn_c/mag) | phi < dyn_a - (dyn_b + dyn_c/mag)).
 * Defaults: 15, 1.0, 0.5, 0.25, 0.5, 8. */
typedef struct {
    double fixed_deg, fixed_min_mag, dyn_min_mag, dyn_a, dyn_b, dyn_c;
} mav_thr_params;

/* One record per frame pair: what the multi-GPU all-gather moves (32 bytes). box = x0, y0, x1, y1 inclusive
 * of the fixed-threshold mask (src/im_helpers.py:55-84 semantics), all -1 when the mask is empty. */
typedef struct {
    int32_t box[4];
    double foe[2];
} mav_result;

void mav_fb_defaults(mav_fb_params*);
void mav_foe_defaults(mav_foe_params*);
void mav_thr_defaults(mav_thr_params*);

/* ---- context ------------------------------------------------------------------------------------------- */
/* One context per (device, W, H, max_batch); owns its streams, pyramid tables and -- from the first call that computes flow on --
 * the Farneback workspace (174 MB per 1080p slot, 16 slots by default): a context used only for mav_bbox / mav_tpr_fpr_counts /
 * mav_phi_mask / mav_detect / the window search holds a few KB plus the staging blocks of its calls.
 * Size bound: max_batch <= 65535 and max_batch * W * H <= 2^30 pixels (1920x1080 x 512 and 3840x2160 x 128, the global batches of
 * BASELINE configs 4 and 5, are 1 % under it): MAV_ERR_ARG above, so that no per-batch element count leaves 32 bits. */
#define MAV_MAX_BATCH_PIXELS ((size_t)1 << 30)
int mav_create(mav_ctx** out, int device, int W, int H, int max_batch, const mav_fb_params* fb /* NULL = defaults */);
int mav_destroy(mav_ctx*);
const char* mav_last_error(void); /* thread-local, never NULL */
int mav_device_count(void);       /* <= 0 when no GPU is visible */
/* Scheduling / tuning options (every switch of the library is here: it reads no environment variable).  MAV_ERR_ARG for unknown
 * names or values out of range.
 *   "group"            pairs per launch for everything but the finest layer's sweeps (default 16 up to 4 Mpx frames, 8 above)
 *   "group_fine"       pairs per launch for the finest layer's sweeps (default 1: one pair's working set stays in the Infinity Cache;
 *                      0 = same as group)
 *   "pairs_in_flight"  1 | 2 (default 2): the finest layer's per-pair work of a group alternates between two streams, every pair swept
 *                      band by band (bands of <= "band_mb" MB of working set, default 96, or "bands" when set) so that both stay in
 *                      the Infinity Cache; the coarse layers alternate sub-groups of "coarse_half" pairs (0 = half the count that fits
 *                      "coarse_cache_mb", default 220) between the two streams
 *   "bands"            J in [1, 8]: a pair's finest-layer sweeps run band by band over J skewed horizontal bands; 0 = automatic (the
 *                      default: 1 up to ~2.6 Mpx, above that as many as keep a band's working set inside the Infinity Cache; the
 *                      two-stream schedule sizes its bands by "band_mb")
 *   "share_m"          one-stream schedule: all pairs of a group ping-pong M through the first slot's buffers (default 1)
 *   "share_frames"     0: treat a frame sequence (see mav_farneback) as independent pairs (default 1)
 *   "small_batch"      default 1: a group whose finest-layer working set is at most 200 MB (one 1080p pair, two 720p pairs: a chain of
 *                      launches that each fill a fraction of the chip) gets the layer images of its whole pyramid from ONE launch and
 *                      all polynomial expansions from ONE launch instead of two launches per layer
 *   "deep_batch"       default 1: when a call has more than one group, the coarse layers of the pyramid (every layer of at most
 *                      1/"deep_frac" of the frame, default 6: all layers above the finest at pyr_scale 0.4) run ONCE for up to 64 pairs of
 *                      the call before the groups start, instead of once per group; "deep_frac" can be set until the first call that
 *                      computes flow (MAV_ERR_STATE afterwards)
 *   "coarse_bands"     default 0: 1 = a coarse layer whose per-pair working set exceeds "band_mb" (layer 1 of the 4K preset, 106 MB) is
 *                      swept like the finest layer, pairs alternating between the two streams band by band (measured slower: off)
 *   "band_skew"        default -1: the band boundaries are moved down by (iterations - 1) / 2 tile rows -- sweep `it` shifts every
 *                      boundary up by `it` rows, so this gives every band the same average size over its sweeps (even launches, even
 *                      cache footprints); n >= 0: by n rows (0 = equal bands)
 *   "band_phase"       default 0; n > 0: in the two-stream schedule the pairs of the second stream use a band partition shifted by half a
 *                      band whenever a pair has at least n bands, so that the two streams do not build their bands' initial M (HBM-bound)
 *                      at the same moments (measured slower: the lockstep of the two streams protects the Infinity Cache)
 *   "sweep_write_through"  -1 (default): the sweeps' M' stores are write-through (sc1) in the two-stream schedules, plain otherwise;
 *                      0 / 1: never / always
 *   "strip"            width in tiles of the column strips of the XCD-aware tile order (0 = automatic)
 *   "phi_screen"       0: every pixel of the phi / threshold stage takes the exact path (default 1: float32 screen in front of it)
 *   "phi_yloop"        16-row blocks per workgroup of the phi kernel (0 = automatic)
 *   "upload_threads"   host threads that stage pageable sources for mav_upload_gather (default 4, the caller included; until its first call)
 *   "inline_uploads"   default 0; 1 = mav_upload_async / _unordered / mav_upload_gather enqueue their copies on the context's COMPUTE stream
 *                      and mav_upload_fence is a no-op: the context then owns one stream (its copy stream and its second compute stream are
 *                      created by the first call that needs them), i.e. one of the runtime's few hardware queues.  For contexts that take a
 *                      stream of small calls in turn with other contexts ("lanes", mavflow/pipeline.py): streams beyond the runtime's queue
 *                      pool share queues, and lanes that share one do not overlap
 *   "stream_priority"  default 0; -1 = the context's compute stream is re-created in the HIGH priority class (the call drains the context).
 *                      The runtime keeps a pool of hardware queues per priority class and hands a new stream the least-used queue of its
 *                      class: which queue a lane gets otherwise depends on every stream the process has ever made (one idle context created
 *                      before three lanes: 0.31 instead of 0.215 ms per 1280x720 frame).  Lanes take a class of their own
 * None of them changes a result bit (tests/test_gpu_flow.py, tests/test_gpu_screen.py). */
int mav_set_option(mav_ctx*, const char* name, long value);
int mav_get_option(mav_ctx*, const char* name, long* value);
/* The schedule a call of `batch` pairs takes with the options in effect, as one line of JSON: every option above, the group split,
 * whether the small-batch schedule applies and, per layer, the blur form and how the sweeps run (pairs per launch, bands).
 * bench.py prints it into its record and hashes it together with the kernel sources. */
int mav_schedule_info(mav_ctx*, int batch, char* buf, size_t cap);
/* Device memory: free / total bytes of the context's GPU (hipMemGetInfo), the bytes this context holds in all (workspace, flow
 * workspace, detection scratch, staging blocks of the host-pointer calls, window-search buffers) and the Farneback workspace alone
 * (0 until a call computes flow).  Any pointer may be NULL. */
int mav_mem_info(mav_ctx*, size_t* dev_free, size_t* dev_total, size_t* ctx_bytes, size_t* workspace_bytes);
int mav_num_layers(const mav_ctx*);
int mav_layer_dims(const mav_ctx*, int k, int* w, int* h, int* ksize, double* sigma);

/* ---- host-pointer entry points (synchronous) ----------------------------------------------------------- */
/* cv2.calcOpticalFlowFarneback(prev, next, None, *fb)   [src/farneback.py:76-80] for `batch` pairs.
 * FRAME SEQUENCES: the reference calls this with next = the frame after prev and keeps `prevgray` for the following call, i.e. a
 * video of n + 1 frames is n pairs whose inner frames each appear twice.  When the two batches are views of ONE run of batch + 1
 * frames -- next == prev + W*H, in every entry point that takes prev / next, host or device pointers -- the library uploads the run
 * once and blurs / expands every frame once per group instead of twice.  The flow is bit-identical to the two-batch form. */
int mav_farneback(mav_ctx*, const uint8_t* prev, const uint8_t* next, int batch, float* flow);
/* Detector.derotate [src/detector.py:70-117]: omega = angular difference / dt, (batch,3); dt (batch). */
int mav_derotate(mav_ctx*, const float* flow, const double* omega, const double* dt, int batch, double* flow_out);
/* FocusOfExpansion.get_FOE_dense + ransac [src/focus_of_expansion.py:32-86]; samples (batch, 2N, 2) = (row, col)
 * drawn by the caller exactly as :70-71 does (the GPU never generates them). foe (batch, 2). */
int mav_foe_dense(mav_ctx*, const double* flow, const uint32_t* samples, int batch, const mav_foe_params*, double* foe);
/* FocusOfExpansion.ransac [src/focus_of_expansion.py:32-54] on caller-supplied estimates (count, 2), count <= 4096:
 * first estimate with the strictly largest number of others within ransac_threshold; (0, 0) when none has a neighbour. */
int mav_ransac(mav_ctx*, const double* estimates, int count, double ransac_threshold, double* foe /* 2 */);
/* cv2.cvtColor(img, COLOR_BGR2GRAY) [src/farneback.py:21,74] for (batch, H, W, 3) u8 frames -> (batch, H, W) u8. */
int mav_bgr2gray(mav_ctx*, const uint8_t* bgr, int batch, uint8_t* gray);
/* FocusOfExpansion.get_phi [src/focus_of_expansion.py:150-184] + threshold block [src/processor.py:333-341].
 * sky: (batch,H,W) u8 or NULL; phi (degrees), mask_fixed, mask_dyn, max_phi (batch) are each optional (NULL). */
int mav_phi_mask(mav_ctx*, const double* flow, const double* foe, const uint8_t* sky, int batch, const mav_thr_params*,
                 double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, double* max_phi);
/* The same two calls handed a FLOAT32 flow array, which is what the reference does for frame index 0 (derotate returns its
 * input, src/detector.py:80-81): numpy then evaluates the |flow2| gate (:78), get_phi (:163-177, zeros_like keeps float32) and
 * the threshold block in float32.  Same arithmetic here, in numpy's operation order; the line intersections stay in double
 * (float32 + uint32 promotes).  phi / max_phi are float32.  arccos: correctly rounded float32 (numpy's own float32 arccos is a
 * SIMD routine up to 2 ulp away from that, host dependent -- see tests/test_frame0.py). */
int mav_foe_dense_f32(mav_ctx*, const float* flow, const uint32_t* samples, int batch, const mav_foe_params*, double* foe);
int mav_phi_mask_f32(mav_ctx*, const float* flow, const double* foe, const uint8_t* sky, int batch, const mav_thr_params*,
                     float* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, float* max_phi);
/* im_helpers.get_simple_bounding_box [src/im_helpers.py:55-84] on u8 images: box (batch,4) = x0,y0,x1,y1, -1 if empty. */
int mav_bbox(mav_ctx*, const uint8_t* img, int batch, int32_t* box);
/* Level 0 of Detector.analyze_pyramid [src/detector.py:280-312] on the 3-channel replica of a u8 image:
 * out (batch,3) = score, x, y of the first 64x64 / stride-16 window with the strictly largest sum. */
int mav_window_max(mav_ctx*, const uint8_t* img, int batch, int64_t* out);
/* Detector.analyze_pyramid [src/detector.py:280-312] over ALL levels of pyramid() [src/im_helpers.py:12-35; each level =
 * imutils.resize(previous, width=int(w/scale)) = cv2.resize(INTER_AREA), until a side drops below 30 px] with
 * sliding_window() [src/im_helpers.py:38-52], on the 3-channel replica of a u8 image (the reference passes scale 1.5).
 * out (batch,6) = score, x, y, level, argmax_row, argmax_col: the first window in scan order (level 0 first) with the
 * strictly largest sum; x, y in that level's own coordinates (the reference does not rescale them); argmax = position of
 * the window's first maximum (np.unravel_index(window.argmax(), ...)); all 0 when no window has a positive sum.
 * MAV_ERR_ARG when scale <= 1 or a level's size ratio is a whole number in both axes (OpenCV's fast-area path). */
int mav_analyze_pyramid(mav_ctx*, const uint8_t* img, int batch, double scale, int64_t* out);
int mav_pyramid_levels(const mav_ctx*, double scale);                          /* number of levels (>= 1), or MAV_ERR_* */
int mav_pyramid_dims(const mav_ctx*, double scale, int level, int* w, int* h); /* size of level `level` */
/* Detector.optimize_window [src/detector.py:314-358] on the 3-channel replica of a u8 image: greedy growth / shrink of a
 * window by moving one corner diagonally by one pixel per step while the enclosed sum rises (Python slice semantics for
 * windows that leave the image).  window_in / window_out (batch,4) = x, y, w, h; score (batch) = 3 * enclosed sum, 0 and the
 * unchanged window when no neighbour has a positive sum. */
int mav_optimize_window(mav_ctx*, const uint8_t* img, int batch, const int32_t* window_in, int64_t* score, int32_t* window_out);
/* im_helpers.calculate_tpr_fpr [src/im_helpers.py:244-252]: positives = #(gt > 127), negatives = #(255 - gt > 127),
 * tp = #(gt * img > 127), fp = #((255 - gt) * img > 127), where img = mask_value at the set pixels of `mask` (any nonzero byte)
 * and 0 elsewhere, products in wide integers as numpy forms them for the reference's own argument 255 * mask (mask_value 255,
 * src/processor.py:350-351) or for a bool mask (mask_value 1).  gt: any u8 image.
 * counts (batch,4) = positives, negatives, true positives, false positives. */
int mav_tpr_fpr_counts(mav_ctx*, const uint8_t* gt, const uint8_t* mask, int mask_value, int batch, int64_t* counts);
/* The same counts for the two masks the most recent mav_detect / mav_process_batch / mav_phi_mask(_f32) call on this context
 * produced, which are still resident on the device: the validation tail of the loop [src/processor.py:350-351] without moving
 * the masks again.  gt (batch, H, W) u8; counts_fixed / counts_dyn (batch, 4) each, either may be NULL.  MAV_ERR_STATE when
 * no such call precedes or its batch differs. */
int mav_last_masks_tpr_fpr(mav_ctx*, const uint8_t* gt, int mask_value, int batch, int64_t* counts_fixed, int64_t* counts_dyn);

/* The fused loop body of Processor.run_detection [src/processor.py:305-341] for `batch` pairs:
 * frames -> flow -> (derotate) -> FoE -> phi -> masks -> box. omega/dt NULL = no rotation (dt = 1);
 * sky NULL = no sky; flow / mask_fixed / mask_dyn / phi outputs are optional (NULL). results (batch).
 * frame0 (batch) u8 flags or NULL: a nonzero flag marks a pair as the reference's frame index 0, for which
 * Detector.derotate returns the float32 flow untouched [src/detector.py:80-81] and every numpy expression after it
 * (|flow2| gate of get_FOE_dense :78, get_phi :163-177, the threshold block) therefore runs in FLOAT32: that pair is not
 * derotated and is evaluated in float32 arithmetic in numpy's operation order (phi, if requested, is the float32 value
 * widened).  Pairs without the flag follow the float64 path of every later frame. */
int mav_process_batch(mav_ctx*, const uint8_t* prev, const uint8_t* next, const uint32_t* samples, const double* omega,
                      const double* dt, const uint8_t* frame0, const uint8_t* sky, int batch, const mav_foe_params*,
                      const mav_thr_params*, float* flow, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn,
                      mav_result* results);
/* The same loop body from the reference's own flow seam [src/datasets/dataset.py:205-212 -> src/processor.py:305-341]:
 * `flow` is the float32 (batch, H, W, 2) field Dataset.get_flow_uv returns (a .flo file, or mav_farneback's output);
 * derotation, FoE, phi, masks and box as in mav_process_batch.  One upload of the flow, no other transfer of it. */
int mav_detect(mav_ctx*, const float* flow, const uint32_t* samples, const double* omega, const double* dt,
               const uint8_t* frame0, const uint8_t* sky, int batch, const mav_foe_params*, const mav_thr_params*, double* phi,
               uint8_t* mask_fixed, uint8_t* mask_dyn, mav_result* results);

/* ---- device-pointer entry points (asynchronous on the context's stream) -------------------------------- */
int mav_farneback_dev(mav_ctx*, const uint8_t* prev, const uint8_t* next, int batch, float* flow);
int mav_process_batch_dev(mav_ctx*, const uint8_t* prev, const uint8_t* next, const uint32_t* samples, const double* omega,
                          const double* dt, const uint8_t* frame0, const uint8_t* sky, int batch, const mav_foe_params*,
                          const mav_thr_params*, float* flow, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn,
                          mav_result* results);
int mav_detect_dev(mav_ctx*, const float* flow, const uint32_t* samples, const double* omega, const double* dt,
                   const uint8_t* frame0, const uint8_t* sky, int batch, const mav_foe_params*, const mav_thr_params*,
                   double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, mav_result* results);
/* Device pointer of the flow field the most recent mav_process_batch_dev / mav_farneback_dev call on this context wrote
 * (the caller's buffer, or the context's own workspace when the caller passed flow == NULL); NULL before the first call.
 * Lets a caller that keeps the flow in the workspace (bench.py) still inspect it. */
/* cv2.cvtColor(COLOR_BGR2GRAY) [src/farneback.py:21,74] on device pointers: (batch, H, W, 3) u8 -> (batch, H, W) u8. */
int mav_bgr2gray_dev(mav_ctx*, const uint8_t* bgr, int batch, uint8_t* gray);
const float* mav_last_flow_dev(const mav_ctx*);
/* im_helpers.calculate_tpr_fpr [src/im_helpers.py:244-252, called at src/processor.py:350-351] for device-resident masks against a
 * device-resident ground truth, counts left on the device (4 x int64 per pair: positives, negatives, true / false positives): the
 * validation tail of a batch as one more launch behind mav_process_batch_dev / mav_detect_dev.  gt_images = batch: one ground-truth
 * image per pair; gt_images = 1: ONE image shared by every pair (a sequence whose segmentation does not change).  Either mask (with
 * its counts buffer) may be NULL; both masks share one pass over the ground truth. */
int mav_tpr_fpr_counts_dev(mav_ctx*, const uint8_t* gt, int gt_images, const uint8_t* mask_fixed, const uint8_t* mask_dyn, int mask_value,
                           int batch, int64_t* counts_fixed, int64_t* counts_dyn);
int mav_sync(mav_ctx*);
void* mav_stream(mav_ctx*); /* the context's hipStream_t */

/* device memory + copies for callers without their own HIP binding */
int mav_dev_alloc(mav_ctx*, size_t bytes, void** out);
int mav_dev_free(mav_ctx*, void* p);
int mav_memcpy_h2d(mav_ctx*, void* dst, const void* src, size_t bytes);
int mav_memcpy_d2h(mav_ctx*, void* dst, const void* src, size_t bytes);

/* Overlapped uploads: pinned host memory and copies on a second stream, ordered against the compute stream in BOTH
 * directions: mav_upload_async first makes the copy stream wait for everything enqueued on the context's stream so far
 * (so a buffer set is never overwritten while an earlier batch still reads it), and mav_upload_fence makes everything
 * enqueued on the context's stream AFTER the fence wait for the copies issued BEFORE it.  Double-buffered batches:
 *   upload_async(set B) ; process_batch_dev(set A) ; upload_fence() ; upload_async(set A) ; process_batch_dev(set B) ; ... */
int mav_host_alloc(mav_ctx*, size_t bytes, void** out);
int mav_host_free(mav_ctx* /* may be NULL: the memory may outlive its context */, void* p);
int mav_upload_async(mav_ctx*, void* dst_dev, const void* src_host, size_t bytes);
/* The same copy WITHOUT the wait for the compute stream: for a destination no enqueued work touches (the other buffer set).  With
 * mav_upload_async the order  process_batch_dev(set A) ; upload_async(set B)  makes the copy wait for batch A -- correct, but the
 * overlap is gone; this form overlaps in either order.  Overwriting a buffer that enqueued work still reads is the caller's bug. */
int mav_upload_async_unordered(mav_ctx*, void* dst_dev, const void* src_host, size_t bytes);
int mav_upload_fence(mav_ctx*);
/* GATHER upload: `count` separate host arrays of bytes_each bytes -> one contiguous device buffer (array i at dst_dev + i * bytes_each),
 * on the copy stream.  This is the shape the reference's loop hands its data over in: one numpy array per frame from
 * Dataset.get_frame / get_flow_uv [src/datasets/dataset.py:205-230], i.e. 128 separate 2 MB arrays for a batch of 64 pairs at 1080p.
 * Pageable sources are copied into a ring of page-locked chunks by a few worker threads (option "upload_threads", default 4: the
 * calling thread plus three; settable until the first call) and every chunk crosses PCIe while the next one is being filled;
 * sources that are page-locked already (mav_host_alloc) are sent from where they are.  On return every source has been READ (the
 * caller may overwrite it) -- for a page-locked source that means the call waits for its transfer (copy stream), or, with option
 * "inline_uploads" (where the transfer would queue behind the compute stream's kernels), stages it like a pageable one; the
 * transfers of staged sources complete on the copy stream -- mav_upload_fence orders the compute stream behind them.
 * flags: MAV_GATHER_ORDERED: as mav_upload_async, the copies wait for everything enqueued on the compute stream so far (without it:
 * as mav_upload_async_unordered).  MAV_GATHER_SOURCES_HELD: the caller keeps every source alive and unchanged until work enqueued
 * behind the copies has completed (a marker recorded after mav_upload_fence has fired): page-locked sources are then read by the DMA
 * engine whenever the stream gets there and the call waits for nothing.  The same pointer may appear more than once (replication). */
#define MAV_GATHER_ORDERED 1
#define MAV_GATHER_SOURCES_HELD 2
int mav_upload_gather(mav_ctx*, void* dst_dev, const void* const* src_host, int count, size_t bytes_each, int flags);
/* Device -> host copy enqueued on the context's stream (dst_host should be page-locked: mav_host_alloc); complete after mav_sync or
 * after a marker recorded behind it. */
int mav_download_async(mav_ctx*, void* dst_host, const void* src_dev, size_t bytes);
/* Markers: "everything enqueued on the context's stream so far" as an object the host can wait for WITHOUT draining the stream
 * (mav_sync also waits for whatever was enqueued after the marker).  A loop that keeps two batches in flight records one per batch
 * behind the batch's result download and waits for it when it needs those results. */
int mav_marker_create(mav_ctx*, void** marker_out);
int mav_marker_record(mav_ctx*, void* marker);
int mav_marker_wait(mav_ctx* /* may be NULL */, void* marker);
/* *done = 1 when everything the marker was recorded behind has completed (or it was never recorded), else 0; never blocks. */
int mav_marker_query(mav_ctx* /* may be NULL */, void* marker, int* done);
int mav_marker_destroy(mav_ctx* /* may be NULL */, void* marker);

/* ---- one iteration of the reference's loop as ONE call ---------------------------------------------------------------------------
 * Processor.run_detection's body [src/processor.py:283-362] is, per frame: read a frame, get the flow (Farneback here), derotate,
 * FoE, phi, masks, TPR / FPR counts against the segmentation, store a record.  Through the entry points above that is a dozen calls
 * per frame (gather upload, fence, Farneback, markers, parameter upload, detect, counts, download, marker) -- at 1280x720, where the
 * GPU needs 0.18 ms per frame, the calling thread needs longer than that to issue them.  mav_frame_step describes the whole
 * iteration; mav_frame_step_dev enqueues it in one call, and mav_frame_step_post hands it to the context's WORKER thread (created by
 * the first post) and returns at once, so that a single-threaded host loop feeding two or three contexts in turn ("lanes") pays a
 * few microseconds per frame and the contexts enqueue side by side.  Every part is optional; what is present runs in this order:
 *   1. the host waits for `wait_before` markers (the last readers of buffers the uploads overwrite),
 *   2. uploads: the packed parameter block (par_host -> par_dev; par_host should be page-locked) and the `gather` lists (host arrays ->
 *      device: frames, a host flow field, per-pair sky masks / ground truth), as mav_upload_gather does, then the upload fence,
 *   3. BGR -> gray of `n_bgr` frames (mav_bgr2gray_dev),
 *   4. compute_flow: flow_dev = Farneback(prev_dev, next_dev) for n pairs (mav_farneback_dev; the frame-sequence layout is
 *      recognised), then the `record_after_flow` markers ("the frame buffers have been read"),
 *   5. detection on flow_dev (mav_detect_dev): samples / omega / dt / frame0 at their offsets inside par_dev, sky_dev, masks,
 *      n records to out_dev; with gt_dev the TPR / FPR counts of both masks (mav_tpr_fpr_counts_dev) to out_dev + off_counts_*,
 *   6. out_dev[0 : out_bytes] -> out_host (page-locked), then `record_done`.
 * The library copies the struct and the pointer arrays it refers to (gather[i].src_host, wait_before, record_after_flow) when the
 * step is posted; the HOST BUFFERS themselves (frames, par_host, out_host) must stay valid and unchanged until the step's
 * `record_done` marker has fired (mav_frame_step_wait). */
typedef struct {
    const void* const* src_host; /* count host arrays of bytes_each bytes ... */
    int count;
    size_t bytes_each;
    void* dst_dev;               /* ... to dst_dev + i * bytes_each */
} mav_gather;
#define MAV_STEP_MAX_GATHER 4
typedef struct mav_frame_step {
    int n; /* pairs (1 <= n <= max_batch) */
    /* 1. */
    void* const* wait_before;
    int n_wait_before;
    /* 2. */
    const void* par_host;
    void* par_dev;
    size_t par_bytes;
    mav_gather gather[MAV_STEP_MAX_GATHER];
    int n_gather;
    /* 3. */
    const uint8_t* bgr_dev; /* n_bgr frames (H, W, 3) u8, gathered there by a `gather` entry */
    int n_bgr;
    uint8_t* gray_dev;      /* n_bgr frames (H, W) u8 out */
    /* 4. */
    int compute_flow;
    const uint8_t* prev_dev; /* n gray frames each */
    const uint8_t* next_dev;
    float* flow_dev;         /* (n, H, W, 2) float32: written by step 4 (NULL: into the context's own flow buffer), or gathered in step 2, or resident */
    void* const* record_after_flow;
    int n_record_after_flow;
    /* 5. */
    int detect;              /* 0: stop after step 4 (a flow-only step) */
    size_t off_samples, off_omega, off_dt, off_frame0; /* byte offsets inside par_dev */
    int has_omega, has_frame0;
    const uint8_t* sky_dev;  /* n sky masks or NULL */
    const uint8_t* gt_dev;   /* ground truth for the counts, or NULL: no counts */
    int gt_images;           /* n, or 1 = one image shared by every pair */
    mav_foe_params foe;
    mav_thr_params thr;
    uint8_t* mask_fixed_dev; /* (n, H, W) u8 each */
    uint8_t* mask_dyn_dev;
    void* out_dev;           /* n mav_result records at offset 0 */
    size_t off_counts_fixed, off_counts_dyn; /* n x 4 int64 each, inside out_dev */
    /* 6. */
    void* out_host;
    size_t out_bytes;
    void* record_done;
} mav_frame_step;
int mav_frame_step_dev(mav_ctx*, const mav_frame_step*);
/* Post the step to the context's worker thread.  From the first post on until mav_worker_drain (or mav_destroy) the worker is the
 * thread that owns the context: the caller may post further steps and wait for tickets, and must call mav_worker_drain before any
 * other entry point of this context (the Python binding does so by itself).  *ticket identifies the step.  mav_destroy lets the
 * worker finish the step it is enqueueing and DROPS the ones still queued. */
int mav_frame_step_post(mav_ctx*, const mav_frame_step*, uint64_t* ticket);
/* Block until step `ticket` has been enqueued by the worker and, when `marker` is not NULL (the step's record_done), until that
 * marker has fired.  Returns the step's own return code (and sets mav_last_error to its message). */
int mav_frame_step_wait(mav_ctx*, uint64_t ticket, void* marker);
/* Block until the worker has enqueued every posted step (device work may still be running); returns the first failed step's code
 * since the last drain, MAV_OK if none.  No-op for a context that never posted. */
int mav_worker_drain(mav_ctx*);

/* Frame decode in front of the path [src/datasets/dataset.py:57,223-230: cv2.VideoCapture over image_%05d.png; src/farneback.py:17-21]:
 * the un-filtering pass of a PNG image, host memory, no context.  raw = the inflated IDAT stream of a non-interlaced image (per row a
 * filter-type byte + stride bytes), bpp = bytes per complete pixel (1 for bit depths below 8), out = rows x stride bytes.  Filters 0 - 4 of
 * the PNG specification.  MAV_ERR_ARG for any other filter type.  (zlib inflate and chunk parsing: mavflow/frame_source.py, Python stdlib.) */
int mav_png_unfilter(const uint8_t* raw, int rows, size_t stride, int bpp, uint8_t* out);

/* HIP-event timing on the context's stream (bench.py): start/stop bracket enqueued work; stop synchronises. */
int mav_timer_start(mav_ctx*);
int mav_timer_stop(mav_ctx*, float* ms);
/* Per-kernel-class profiling with HIP events (separate pass, never inside a timed region).  on = 1: events around every launch
 * (durations and launch counts per class); on = 2: events around every RUN of consecutive launches of one class on a stream -- a tenth
 * of the events, for mav_profile_busy, which then sees the two streams overlap almost undisturbed (total_ms / launches then count runs).
 * mav_profile_get: name[i] / total_ms[i] / launches[i] for i < *n (caller passes capacity in *n). */
int mav_profile_enable(mav_ctx*, int on);
int mav_profile_get(mav_ctx*, int* n, const char** names, double* total_ms, long* launches);
/* Milliseconds during which at least one launch of the named kernel classes (comma-separated, names as mav_profile_get reports
 * them) was running in the profiled calls: the union of the launches' intervals.  With two pairs in flight (option
 * "pairs_in_flight") launches of one class overlap and their summed durations exceed the wall time. */
int mav_profile_busy(mav_ctx*, const char* names, double* busy_ms);
/* The profiled launches (mode 1) or runs of launches (mode 2) themselves, as intervals: class index in mav_profile_get's order, stream
 * (0 = the context's stream, 1 = its second compute stream), start / end in ms since mav_profile_enable.  *n = capacity in, count out;
 * with NULL arrays *n returns the number of intervals.  tools/untraced_anatomy.py builds the step's timeline from it without a tracer. */
int mav_profile_intervals(mav_ctx*, int* n, int* kernel_class, int* stream, float* t0_ms, float* t1_ms);

/* Calibration for the roofline record: GB/s this GPU delivers, now, to a plain streaming kernel with the sweep kernel's mix of
 * 3 reads : 1 write (four temporary buffers of bytes_per_buffer each, float4 per thread, `reps` timed launches on the context's
 * stream).  4 x 32 MB stays inside the 256 MB Infinity Cache, 4 x 1 GB does not. */
int mav_membw_probe(mav_ctx*, size_t bytes_per_buffer, int reps, double* gbs);

/* multi-GPU: gather `bytes_per_rank` bytes from every rank (RCCL ncclAllGather over xGMI) on the context's stream.
 * comm is an ncclComm_t created by the caller (mav_comm_* helpers below wrap RCCL's own bootstrap). */
/* One line of JSON: the HIP version libmavflow was built with, the HIP runtime / driver versions the process actually runs (in the
 * multi-GPU bench torch loads its own runtime first and libmavflow binds to it) and RCCL's version once it is loaded (0 before).
 * mav_comm_init returns MAV_ERR_STATE when the running runtime's major version differs from the build's. */
int mav_runtime_info(char* buf, size_t cap);
int mav_comm_unique_id(void* id128 /* 128 bytes out */);
int mav_comm_init(mav_ctx*, const void* id128, int rank, int nranks, void** comm_out);
int mav_comm_count(void* comm, int* nranks); /* ncclCommCount: the ranks the communicator really spans */
int mav_comm_destroy(void* comm);
int mav_allgather_results(mav_ctx*, void* comm, const void* local_dev, size_t bytes_per_rank, void* all_dev);

/* ---- stage hooks (diagnostics / parity tests; host pointers, one image or pair, SoA planes) ------------- */
/* layer image I_k of one frame: convertTo(f32) -> GaussianBlur -> resize   (h_k, w_k) */
int mav_stage_blur_resize(mav_ctx*, const uint8_t* img, int k, float* out);
/* the same through the separable two-pass kernels (H x w scratch in memory) that layers with a long Gaussian use: layers with a
 * short one go through ONE fused kernel in the product path, and the two forms must agree bit for bit */
int mav_stage_blur_resize_two_pass(mav_ctx*, const uint8_t* img, int k, float* out);
/* FarnebackPolyExp of a (h, w) f32 image at layer k -> R as 5 planes (5, h, w) */
int mav_stage_polyexp(mav_ctx*, const float* I, int k, float* R);
/* FarnebackUpdateMatrices at layer k: R0, R1 (5,h,w), flow (h,w,2) -> M (5,h,w) */
int mav_stage_update_matrices(mav_ctx*, const float* R0, const float* R1, const float* flow, int k, float* M);
/* one FarnebackUpdateFlow_Blur sweep at layer k: M (5,h,w) -> flow (h,w,2) and, if update != 0, M_out (5,h,w) */
int mav_stage_blur_iter(mav_ctx*, const float* R0, const float* R1, const float* M, int k, int update, float* flow,
                        float* M_out);

/* The phi / mask / box stage exactly as the fused path runs it (float32 flow from the flow stage, derotation on the fly,
 * double arithmetic, single-precision screen when phi == NULL) but with a caller-supplied FoE (batch, 2) instead of the RANSAC
 * fit: lets a test plant pixels around every threshold for a known FoE.  omega / dt / sky / phi / masks / box may be NULL;
 * box (batch, 4) = x0, y0, x1, y1 of the fixed mask. */
int mav_stage_phi_mask(mav_ctx*, const float* flow, const double* foe, const double* omega, const double* dt, const uint8_t* sky,
                       int batch, const mav_thr_params*, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, int32_t* box);

/* The constants the flow kernels run on, as the library derived them (parity tests compare them with an independent
 * derivation): FarnebackPrepareGaussian(poly_n, poly_sigma) -> g, xg, xxg (poly_n + 1 floats each, centre tap first) and
 * ig = {ig11, ig03, ig33, ig55}; and, when k >= 0, layer k's GaussianBlur taps (its ksize floats).  Any pointer may be NULL. */
int mav_stage_coefficients(mav_ctx*, int k, float* g, float* xg, float* xxg, float* ig, float* blur_taps);

/* pyramid level `level` (>= 0) of one u8 image: (h_l, w_l) u8, sizes from mav_pyramid_dims */
int mav_stage_pyramid_level(mav_ctx*, const uint8_t* img, double scale, int level, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* MAVFLOW_H */
