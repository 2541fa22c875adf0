// Host side of libmavflow.so: the C-ABI of include/mavflow.h over the gfx950 kernels.
// Owns the context (stream, pyramid tables, workspace), schedules the per-layer launches in groups of pairs that
// keep the iteration working set cache-sized, and maps every failure to an error code + message.
#include <dlfcn.h>
#include <math.h>
#include <float.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "mavflow_internal.h"

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? MAV_ERR_OOM : MAV_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                         \
    } while (0)
#define CHK(call)              \
    do {                       \
        int rc_ = (call);      \
        if (rc_ != MAV_OK) return rc_; \
    } while (0)

extern "C" const char* mav_last_error(void) { return g_err.c_str(); }

extern "C" void mav_fb_defaults(mav_fb_params* p) { *p = mav_fb_params{0.4, 1, 12, 10, 8, 1.2, 0}; }
extern "C" void mav_foe_defaults(mav_foe_params* p) { *p = mav_foe_params{1000, 2.5, 30.0}; }
extern "C" void mav_thr_defaults(mav_thr_params* p) { *p = mav_thr_params{15.0, 1.0, 0.5, 0.25, 0.5, 8.0}; }

extern "C" int mav_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- small host math ------------------------------------------------------------------------------------
static int cv_round(double v) { return (int)nearbyint(v); }
static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}
// smallest double T with fl(sqrt(T)) >= t, so that  sqrt(v) < t  <=>  v < T  for every double v (sqrt is monotone
// and correctly rounded on both sides).  Lets the kernels compare squared magnitudes without changing a single result.
static double sq_threshold(double t)
{
    if (!(t > 0)) return 0.0;
    double T = t * t;
    while (sqrt(nextafter(T, -INFINITY)) >= t) T = nextafter(T, -INFINITY);
    while (sqrt(T) < t) T = nextafter(T, INFINITY);
    return T;
}

// float32 twin for frame-0 pairs: numpy compares sqrt_f32(m2) with the threshold rounded to float32 (a Python float next to a
// float32 scalar is "weak"), so  sqrtf(v) < (float)t  <=>  v < T32  with T32 the smallest float whose rounded root is >= (float)t.
static float sq_threshold_f32(double t)
{
    const float tf = (float)t;
    if (!(tf > 0)) return 0.f;
    float T = tf * tf;
    while (sqrtf(nextafterf(T, -INFINITY)) >= tf) T = nextafterf(T, -INFINITY);
    while (sqrtf(T) < tf) T = nextafterf(T, INFINITY);
    return T;
}

static void gaussian_kernel(int n, double sigma, std::vector<float>& k)  // getGaussianKernel(n, sigma, CV_32F)
{
    static const float small_tab[4][7] = {{1.f},
                                          {0.25f, 0.5f, 0.25f},
                                          {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                          {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float* fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : nullptr;
    const double sx = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double s2 = -0.5 / (sx * sx);
    k.resize(n);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        const double t = fixed ? (double)fixed[i] : exp(s2 * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

static bool inv6_cholesky(const double G[36], double inv[36])
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = G[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            if (i == j) {
                if (!(s > 0)) return false;
                L[i * 6 + j] = sqrt(s);
            } else
                L[i * 6 + j] = s / L[j * 6 + j];
        }
    for (int c = 0; c < 6; c++) {
        double y[6], x[6];
        for (int i = 0; i < 6; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= L[i * 6 + k] * y[k];
            y[i] = s / L[i * 6 + i];
        }
        for (int i = 5; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 6; k++) s -= L[k * 6 + i] * x[k];
            x[i] = s / L[i * 6 + i];
        }
        for (int i = 0; i < 6; i++) inv[i * 6 + c] = x[i];
    }
    return true;
}

static bool prepare_poly(int n, double sigma, PolyCoef* pc)  // FarnebackPrepareGaussian
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    std::vector<float> gb(2 * n + 1), xgb(2 * n + 1), xxgb(2 * n + 1);
    float *g = gb.data() + n, *xg = xgb.data() + n, *xxg = xxgb.data() + n;
    double s = 0.;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G[36];
    memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0] += g[y] * g[x];
            G[7] += g[y] * g[x] * x * x;
            G[21] += g[y] * g[x] * x * x * x * x;
            G[35] += g[y] * g[x] * x * x * y * y;
        }
    G[14] = G[3] = G[4] = G[18] = G[24] = G[7];
    G[28] = G[21];
    G[22] = G[27] = G[35];
    double inv[36];
    if (!inv6_cholesky(G, inv)) return false;
    pc->n = n;
    for (int k = 0; k <= n; k++) { pc->g[k] = g[k]; pc->xg[k] = xg[k]; pc->xxg[k] = xxg[k]; }
    pc->ig11 = (float)inv[7]; pc->ig03 = (float)inv[3]; pc->ig33 = (float)inv[21]; pc->ig55 = (float)inv[35];
    return true;
}

// resize(INTER_LINEAR): source index and weight for destination index o of d along an axis of S source pixels -- the host twin of the
// kernels' resize_coord (same expression, same roundings: this file is compiled without FMA contraction): (o + 0.5) * scale - 0.5 in
// double, rounded to float, floor, clamp.
static void resize_coord_host(int o, int S, int d, double scale, int* s0, float* f)
{
    if (d == S) { *s0 = o; *f = 0.f; return; }
    const double p = (o + 0.5) * scale;
    float t = (float)(p - 0.5);
    int s = (int)floorf(t);
    t -= (float)s;
    if (s < 0) { t = 0.f; s = 0; }
    if (s >= S - 1) { t = 0.f; s = S - 1; }
    *s0 = s; *f = t;
}

// ---- context -------------------------------------------------------------------------------------------
struct Layer {
    int w, h, ksize;
    double sigma;
    float* g = nullptr;      // device copy of the Gaussian taps (getGaussianKernel(ksize, sigma, CV_32F))
    int* coord = nullptr;    // device: xs[w] | xf[w] | ys[h] | yf[h] -- the resize coordinates of the layer's columns and rows (coarse layers)
};

// ---- gather uploads: many host arrays -> one contiguous device buffer ----------------------------------------------------------
// The reference's loop is handed one numpy array per frame (Dataset.get_frame / get_flow_uv, src/datasets/dataset.py:205-230); a batch
// of 64 pairs is 128 separate pageable 2 MB arrays.  hipMemcpy from pageable memory stages through the runtime's own bounce buffer on
// the calling thread (~6 GB/s here); np.stack + one copy touches every byte twice.  The stager copies the sources into a ring of
// page-locked chunks on a few worker threads (memcpy at DRAM rate) and issues one H2D per chunk on the copy stream while the next
// chunk is being filled, so the PCIe transfer hides behind the staging.  A source that is already page-locked (mav_host_alloc, the
// Python layer's pinned pool) is copied straight from where it is.
struct Stager {
    enum { NCHUNK = 4 };
    static constexpr size_t CHUNK = (size_t)16 << 20, PIECE = (size_t)512 << 10;
    struct Seg { char* dst; const char* src; size_t n; };
    void* chunk[NCHUNK] = {nullptr};
    hipEvent_t sent[NCHUNK] = {nullptr};
    bool in_flight[NCHUNK] = {false};
    unsigned next_chunk = 0;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::vector<Seg> segs;
    std::atomic<size_t> next_seg{0};
    unsigned long long generation = 0;
    int busy = 0;
    bool stop = false;

    void run_segments()
    {
        for (;;) {
            const size_t i = next_seg.fetch_add(1);
            if (i >= segs.size()) return;
            memcpy(segs[i].dst, segs[i].src, segs[i].n);
        }
    }
    void worker()
    {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv_job.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            lk.unlock();
            run_segments();
            lk.lock();
            if (--busy == 0) cv_done.notify_one();
        }
    }
    // copy every segment of `segs` (the calling thread takes part), return when all are done
    void copy_all()
    {
        next_seg.store(0);
        if (workers.empty() || segs.size() < 2) { run_segments(); return; }
        {
            std::lock_guard<std::mutex> lk(m);
            busy = (int)workers.size();
            generation++;
        }
        cv_job.notify_all();
        run_segments();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return busy == 0; });
    }
    void shutdown()
    {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv_job.notify_all();
        for (auto& t : workers) t.join();
        workers.clear();
        for (int i = 0; i < NCHUNK; i++) {
            if (sent[i]) { hipEventSynchronize(sent[i]); hipEventDestroy(sent[i]); }
            if (chunk[i]) hipHostFree(chunk[i]);
        }
    }
};
enum KernelId { K_BLUR_RESIZE, K_POLYEXP, K_UPDATE, K_ITER, K_ITER_COARSE, K_FOE, K_PHI, K_MISC, K_COUNT };
static const char* const kKernelNames[K_COUNT] = {"blur_resize", "polyexp", "update_matrices", "blur_iter", "blur_iter_coarse",
                                                  "foe_ransac", "phi_mask_box", "misc"};
struct ProfRec { int kid; hipEvent_t a, b; int stream; };
struct ProfInterval { int kid; float t0, t1; int stream; };      // ms since the profile was switched on

struct mav_ctx {
    int device = 0, W = 0, H = 0, max_batch = 0, group = 0, group_fine = 1;
    mav_fb_params fb;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // uploads that overlap the compute stream (mav_upload_async / mav_upload_fence)
    hipEvent_t copy_done = nullptr, compute_mark = nullptr;
    hipEvent_t gather_done = nullptr;    // mav_upload_gather: "the page-locked sources sent from where they are have been read"
    struct Stager* stager = nullptr;     // mav_upload_gather: page-locked ring + worker threads (created by its first call)
    struct Worker* worker = nullptr;     // mav_frame_step_post: the thread that enqueues posted steps (created by the first post)
    int upload_threads = 4;              // option "upload_threads"
    bool inline_uploads = false;         // option "inline_uploads": uploads go on the compute stream (no copy stream, no cross-stream events)
    int stream_priority = 0;             // option "stream_priority": the compute stream's priority class (the runtime keeps a queue pool per class)
    const float* last_flow = nullptr;    // where the latest farneback / process_batch call wrote its flow (mav_last_flow_dev)
    const uint8_t *last_mf = nullptr, *last_md = nullptr;   // masks of the latest host-pointer detection call, still in their
    int last_mask_batch = 0;                                // staging blocks (mav_last_masks_tpr_fpr)
    std::vector<Layer> layers;
    PolyCoef pc;
    // workspace (group slots)
    size_t n0 = 0, n1 = 0;
    // Workspace of `group` slots.
    struct WorkSet {
        float *Htmp = nullptr;       // scratch of the two-pass blur+resize (layers whose Gaussian is too long for the fused kernel)
        // R = R0 | R1 in ONE allocation (R1 = R0 + 5 n0 group): the expansions of the two frames of every pair.  For a frame SEQUENCE
        // (next = prev + one frame) the group's g + 1 frames are expanded once into slots 0 .. g and pair s reads slots s and s + 1.
        float *I = nullptr, *R = nullptr, *Ma = nullptr, *Mb = nullptr, *fc[2] = {nullptr, nullptr};
        // small groups (flow_group): the coarse layers' images / expansions, every layer in a region of its own (layer k: c_off[k]
        // floats per frame in, c_stride[k] floats per frame), so that the whole pyramid can be blurred and expanded in two launches
        float *Ic = nullptr, *Rc = nullptr;
    } ws;
    std::vector<size_t> c_off, c_stride;      // per layer (index 0 unused); c_total = sum of the strides
    size_t c_total = 0;
    // DEEP LAYERS (deep_layers): layers kd .. top, each at most 1/deep_frac of the frame -- every coarse layer of a 0.4-scale pyramid.
    // When a call has more than one group they run ONCE for up to deep_cap pairs of the call (their images from one launch per tile code,
    // all expansions from one launch, sweeps over all pairs in cache-sized sub-groups) before the groups start; the groups then begin at
    // layer kd - 1 with the deep flow as their coarser layer.  Buffers of their own, every layer in a compact region.
    struct DeepSet { float *I = nullptr, *R = nullptr, *Ma = nullptr, *Mb = nullptr, *f[2] = {nullptr, nullptr}; } deep;
    int kd = 0, deep_cap = 0;                 // kd = 0: no deep layer
    // option "deep_frac" (before the first flow call): a layer is deep when its pixels x deep_frac <= the frame's.  6: every coarse layer
    // of a 0.4-scale pyramid (layer 1 is 0.16 of the frame).  Measured with 32 (layers 2 - 4 of the 4K preset only) vs 6: 1080p, 64 pairs
    // 2 579 - 2 637 vs 2 653 - 2 686 pairs/s (+2.4 %: layer 1's sixteen sub-groups of 4 pairs run back to back for the whole call instead
    // of four per group behind a fork / join each); 4K 607 - 609 vs 609 - 610 (profiles/r04/ab_deep_frac.log)
    int deep_frac = 6;
    bool deep_batch = true;                   // option "deep_batch"
    int band_skew = -1;                       // option "band_skew": tile rows the band boundaries move down to compensate the sweeps' skew
                                              // (sweeps_band_major); -1 = (iterations - 1) / 2, 0 = equal bands
    int band_phase = 0;                       // option "band_phase": n > 0 = the second stream's pairs use a partition shifted by half a band
                                              // when a pair has at least n bands (sweeps_band_major); 0 = never (default: measured slower)
    bool coarse_bands = false;                // option "coarse_bands": a coarse layer whose per-pair working set exceeds band_mb is swept like the finest one
                                              // (measured at 3840x2160 / 5 layers: 584 vs 594 pairs/s -- half-size launches cost more than the cache returns; off)
    int small_g = 0;                          // pairs the Ic / Rc buffers were sized for (0: none)
    bool small_batch = true;                  // option "small_batch"
    int sweep_wt = -1;                        // option "sweep_write_through": -1 = in the two-stream schedules only (default), 0 / 1 = never / always
    int small_batch_mb = 200;                 // option "small_batch_mb": ... for groups of at most this much finest-layer sweep working set
    int bands = 1;                   // option "bands": the finest layer's sweeps in band-major order over this many skewed bands
    // option "pairs_in_flight" (1 or 2): the finest layer's per-pair work (initial M + sweeps) of a group alternates between the
    // compute stream and pair_stream, every pair band-major over bands of at most pif_band_mb of working set (layer_sweeps)
    // band_mb 96: 2 bands at 1080p (83 MB each), 7 at 3840x2160 (95 MB, ~1 160 tiles per launch on 1 280 resident slots): 608 vs 605
    // pairs/s with 8 bands of 83 MB, 597 with 6 of 111 MB, 567 with 5 (profiles/r04/ab_band_mb_4k.log)
    int pairs_in_flight = 2, pif_band_mb = 96;
    bool bands_set = false;          // "bands" given explicitly: that many bands in either schedule
    int bands_auto = 1;              // what "bands" = 0 restores
    bool group_fine_set = false;     // "group_fine" given explicitly
    hipStream_t pair_stream = nullptr;
    hipEvent_t pif_fork = nullptr, pif_join = nullptr;
    bool share_frames = true;        // option "share_frames": expand a frame once when next == prev + one frame (a frame sequence)
    int coarse_cache_mb = 220;       // coarse layers: pairs per launch capped so that the sweeps' working set stays below this (0 = no cap)
    // tuning options that used to be environment variables (mav_set_option / mav_get_option; all reported by mav_schedule_info)
    bool share_m = true;             // "share_m": one-stream schedule, every pair of a group ping-pongs M through the first slot's buffers
    int coarse_half = 0;             // "coarse_half": pairs per launch of the coarse layers' two-stream schedule (0 = half the cache-sized count)
    int strip = 0;                   // "strip": width in tiles of the column strips of the XCD-aware tile order (0 = automatic)
    bool phi_screen = true;          // "phi_screen": the float32 screen in front of the exact phi / threshold arithmetic
    int phi_yloop = 0;               // "phi_yloop": 16-row blocks per workgroup of the phi kernel (0 = automatic)
    size_t htmp_stride = 0;
    bool ws_ready = false;         // the Farneback workspace exists (ensure_workspace: allocated by the first call that computes flow)
    size_t ws_bytes = 0;           // its size
    size_t group_bytes = 0, deep_bytes = 0;   // ws_bytes = the group slots + the deep set (0 until a call of more than one group)
    float* flow_ws = nullptr;      // lazily allocated (max_batch) when the caller does not want the flow
    // detection scratch (max_batch)
    FoeScratch foe_sc{nullptr, nullptr, nullptr};
    int foe_sc_n = 0;
    double* foe_dev = nullptr;
    int32_t* box_acc = nullptr;
    unsigned long long* u64_scratch = nullptr;  // [max_batch*8]
    int* i32_scratch = nullptr;                 // [max_batch]
    DerotParams* derot_dev = nullptr;
    // staging buffers of the host-pointer entry points: slot i of a call re-uses the block slot i of the previous call
    // left behind (grow-only), so the staged path performs no hipMalloc / hipFree once warm
    struct Block { void* p = nullptr; size_t cap = 0; };
    std::vector<Block> scratch;
    size_t scratch_next = 0;
    uint8_t* pyr_ws = nullptr;                  // analyze_pyramid level images (lazily, max_batch)
    size_t pyr_ws_bytes = 0;
    unsigned long long* sat = nullptr;          // optimize_window summed-area tables (lazily, max_batch)
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int profiling = 0;               // 0 off; 1 = HIP events around every launch; 2 = around every RUN of launches of one class on a stream
    struct OpenRun { hipStream_t st; int kid; hipEvent_t a; };
    std::vector<OpenRun> open_runs;  // mode 2: the run in progress on each stream
    std::vector<ProfRec> prof;
    std::vector<ProfInterval> prof_iv;          // every profiled launch as an interval (mav_profile_busy: union over concurrent streams)
    hipEvent_t prof_base = nullptr;
    double prof_ms[K_COUNT] = {0};
    long prof_n[K_COUNT] = {0};
};

// mode 2: the end event of a run is recorded when the next launch on that stream belongs to another class (or at collection): it
// completes, in stream order, when the run's last kernel has -- two events per run instead of two per launch, so that the
// overlap of the two streams is measured almost undisturbed (mav_profile_busy)
static void close_run(mav_ctx* c, size_t i)
{
    mav_ctx::OpenRun r = c->open_runs[i];
    hipEvent_t b = nullptr;
    hipEventCreate(&b);
    hipEventRecord(b, r.st);
    c->prof.push_back({r.kid, r.a, b, r.st == c->stream ? 0 : 1});
    c->open_runs.erase(c->open_runs.begin() + i);
}
static void prof_close_stream(mav_ctx* c, hipStream_t st)       // before a stream waits for another one: the wait is not part of the run
{
    for (size_t i = 0; i < c->open_runs.size(); i++) if (c->open_runs[i].st == st) { close_run(c, i); return; }
}
struct ProfScope {
    mav_ctx* c; int kid; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(mav_ctx* c_, int k, hipStream_t st_ = nullptr) : c(c_), kid(k), st(st_ ? st_ : c_->stream)
    {
        if (c->profiling == 1) { hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a, st); }
        else if (c->profiling == 2) {
            for (size_t i = 0; i < c->open_runs.size(); i++)
                if (c->open_runs[i].st == st) {
                    if (c->open_runs[i].kid == kid) return;          // the run goes on
                    close_run(c, i);
                    break;
                }
            hipEventCreate(&a);
            hipEventRecord(a, st);
            c->open_runs.push_back({st, kid, a});
        }
    }
    ~ProfScope()
    {
        if (c->profiling == 1) { hipEventRecord(b, st); c->prof.push_back({kid, a, b, st == c->stream ? 0 : 1}); }
    }
};

static void free_layer(Layer& l)
{
    if (l.g) hipFree(l.g);
    if (l.coord) hipFree(l.coord);
    l.g = nullptr; l.coord = nullptr;
}

// Pairs of the largest group the small-group schedule can take (is_small_group) when the context runs groups of `group` pairs.
static int small_group_cap(const mav_ctx* c, int group)
{
    size_t sg = c->c_total ? ((size_t)c->small_batch_mb << 20) / (c->n0 * 80) : 0;
    return (int)(sg > (size_t)group ? (size_t)group : sg);
}
// Workspace for `group` slots.  The new buffers are allocated in full before the old ones are released: when an allocation fails the
// context keeps its previous group and stays usable (the caller sees MAV_ERR_OOM).
static int alloc_group(mav_ctx* c, int group)
{
    const size_t g = (size_t)group, nc = 2 * (c->n1 ? c->n1 : 1);
    // I / I2 hold the 2 g frames of a group (prev and next in one launch; a frame sequence of g pairs has g + 1 <= 2 g frames);
    // Htmp holds g + 1 (the two-pass blur never runs over more frames per launch)
    enum { NB = 9 };
    // Ic / Rc: for the largest group the small-group schedule can take (is_small_group), 2 g frames of every coarse layer
    const size_t sg = (size_t)small_group_cap(c, group);
    const size_t elems[NB] = {c->n0 * 2 * g, 10 * c->n0 * g, 5 * c->n0 * g, 5 * c->n0 * g, nc * g, nc * g, c->htmp_stride * (g + 1),
                              sg ? 2 * sg * c->c_total : 1, sg ? 10 * sg * c->c_total : 1};
    float* fresh[NB] = {nullptr};
    size_t total = 0;
    for (int i = 0; i < NB; i++) {
        const hipError_t e = hipMalloc(&fresh[i], sizeof(float) * elems[i]);
        if (e != hipSuccess) {
            for (int j = 0; j < NB; j++) if (fresh[j]) hipFree(fresh[j]);
            (void)hipGetLastError();
            return fail(e == hipErrorOutOfMemory ? MAV_ERR_OOM : MAV_ERR_HIP, "workspace for group %d (%zu bytes for buffer %d): %s", group,
                        sizeof(float) * elems[i], i, hipGetErrorString(e));
        }
        total += sizeof(float) * elems[i];
    }
    mav_ctx::WorkSet& w = c->ws;
    float** bufs[NB] = {&w.I, &w.R, &w.Ma, &w.Mb, &w.fc[0], &w.fc[1], &w.Htmp, &w.Ic, &w.Rc};
    for (int i = 0; i < NB; i++) { if (*bufs[i]) hipFree(*bufs[i]); *bufs[i] = fresh[i]; }
    c->group = group;
    c->small_g = (int)sg;
    c->ws_ready = true;
    c->group_bytes = total;
    c->ws_bytes = c->group_bytes + c->deep_bytes;
    return MAV_OK;
}
// The deep layers' work set (deep_layers) does not depend on the group and is reachable only by calls of more than one group
// (use_deep_batch): allocated by the first such call -- a context whose calls never exceed one group (max_batch <= group, or one-pair
// calls) never holds it (150 MB per pair at 3840x2160 / 5 levels) --, kept across "group" changes until mav_destroy.
static int ensure_deep(mav_ctx* c)
{
    if (c->kd <= 0 || c->deep.I) return MAV_OK;
    const size_t D = (size_t)c->deep_cap, dt = c->c_total - c->c_off[c->kd], top = c->c_stride[c->kd];
    const size_t de[6] = {2 * D * dt, 10 * D * dt, 5 * D * top, 5 * D * top, 2 * D * top, 2 * D * top};
    float* d[6] = {nullptr};
    size_t bytes = 0;
    for (int i = 0; i < 6; i++) {
        const hipError_t e = hipMalloc(&d[i], sizeof(float) * de[i]);
        if (e != hipSuccess) {
            for (int j = 0; j < 6; j++) if (d[j]) hipFree(d[j]);
            (void)hipGetLastError();
            return fail(e == hipErrorOutOfMemory ? MAV_ERR_OOM : MAV_ERR_HIP, "deep-layer workspace (%zu bytes): %s", sizeof(float) * de[i], hipGetErrorString(e));
        }
        bytes += sizeof(float) * de[i];
    }
    c->deep.I = d[0]; c->deep.R = d[1]; c->deep.Ma = d[2]; c->deep.Mb = d[3]; c->deep.f[0] = d[4]; c->deep.f[1] = d[5];
    c->deep_bytes = bytes;                     // only once the whole set exists
    c->ws_bytes = c->group_bytes + c->deep_bytes;
    return MAV_OK;
}
// The Farneback workspace (174 MB per 1080p slot, 16 slots by default) belongs to the calls that compute flow: a context created for
// mav_bbox / mav_tpr_fpr_counts / mav_phi_mask / mav_detect never pays for it (the reference's helpers are stateless free functions,
// src/im_helpers.py:55-84,244-252).  Allocated by the first call that needs it, kept until mav_destroy.
static int ensure_workspace(mav_ctx* c)
{
    return c->ws_ready ? MAV_OK : alloc_group(c, c->group);
}

static int ensure_copy_stream(mav_ctx* c)
{
    if (c->copy_stream) return MAV_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&c->copy_done, &c->compute_mark}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return MAV_OK;
}
static int ensure_pair_stream(mav_ctx* c)
{
    if (c->pair_stream) return MAV_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamCreateWithFlags(&c->pair_stream, hipStreamNonBlocking));
    for (hipEvent_t* e : {&c->pif_fork, &c->pif_join}) HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    return MAV_OK;
}

static int sync_all_streams(mav_ctx* c)
{
    for (hipStream_t st : {c->stream, c->pair_stream})
        if (st) HIPCHK(hipStreamSynchronize(st));
    return MAV_OK;
}

static void stop_worker(mav_ctx* c);
extern "C" int mav_destroy(mav_ctx* c)
{
    if (!c) return MAV_OK;
    hipSetDevice(c->device);
    stop_worker(c);                  // (finishes the step it is enqueueing, drops the rest, leaves)
    (void)sync_all_streams(c);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
    if (c->stager) { c->stager->shutdown(); delete c->stager; c->stager = nullptr; }
    for (auto& l : c->layers) free_layer(l);
    {
        mav_ctx::WorkSet& w = c->ws;
        void* wb[] = {w.I, w.R, w.Ma, w.Mb, w.fc[0], w.fc[1], w.Htmp, w.Ic, w.Rc, c->deep.I, c->deep.R, c->deep.Ma, c->deep.Mb, c->deep.f[0], c->deep.f[1]};
        for (void* b : wb) if (b) hipFree(b);
    }
    void* bufs[] = {c->flow_ws, c->foe_sc.cand, c->foe_sc.count, c->foe_sc.best_key, c->foe_sc.done, c->foe_dev, c->box_acc, c->u64_scratch,
                    c->i32_scratch, c->derot_dev, c->pyr_ws, c->sat};
    for (void* b : bufs) if (b) hipFree(b);
    for (auto& blk : c->scratch) if (blk.p) hipFree(blk.p);
    for (auto& r : c->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    if (c->prof_base) hipEventDestroy(c->prof_base);
    for (hipEvent_t e : {c->t0, c->t1, c->copy_done, c->compute_mark, c->gather_done, c->pif_fork, c->pif_join}) if (e) hipEventDestroy(e);
    for (hipStream_t st : {c->pair_stream, c->copy_stream, c->stream}) if (st) hipStreamDestroy(st);
    delete c;
    return MAV_OK;
}

extern "C" int mav_create(mav_ctx** out, int device, int W, int H, int max_batch, const mav_fb_params* fbp)
{
    if (!out) return fail(MAV_ERR_ARG, "mav_create: out is NULL");
    *out = nullptr;
    mav_fb_params fb;
    if (fbp) fb = *fbp; else mav_fb_defaults(&fb);
    if (W < 1 || H < 1 || max_batch < 1) return fail(MAV_ERR_ARG, "mav_create: bad size W=%d H=%d max_batch=%d", W, H, max_batch);
    if (max_batch > 65535 || (size_t)max_batch * (size_t)W * (size_t)H > MAV_MAX_BATCH_PIXELS)
        return fail(MAV_ERR_ARG, "mav_create: max_batch %d x %dx%d exceeds the bound (max_batch <= 65535, max_batch * W * H <= %zu pixels)", max_batch,
                    W, H, (size_t)MAV_MAX_BATCH_PIXELS);
    if (!(fb.pyr_scale > 0 && fb.pyr_scale < 1)) return fail(MAV_ERR_ARG, "pyr_scale must be in (0, 1), got %g", fb.pyr_scale);
    if (fb.levels < 0 || fb.winsize < 2 || fb.winsize > 64 || fb.iterations < 1 || fb.poly_n < 1 || fb.poly_n > MAV_MAX_POLY_N)
        return fail(MAV_ERR_ARG, "unsupported Farneback parameters (levels=%d winsize=%d iterations=%d poly_n=%d)", fb.levels,
                    fb.winsize, fb.iterations, fb.poly_n);
    if (fb.flags != 0) return fail(MAV_ERR_ARG, "only flags == 0 (box window, no initial flow) is implemented, got %d", fb.flags);
    if (fb.winsize / 2 != 6 && blur_iter_lds_bytes(fb.winsize) > (size_t)160 * 1024)
        return fail(MAV_ERR_ARG, "winsize %d needs %zu bytes of LDS per workgroup in the general sweep kernel, the CU has 163840", fb.winsize,
                    blur_iter_lds_bytes(fb.winsize));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(MAV_ERR_STATE, "no HIP device visible: libmavflow has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(MAV_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
    HIPCHK(hipSetDevice(device));
    if (fb.winsize / 2 != 6) {      // the general sweep kernel asks for more dynamic LDS than the default limit: per device, checked
        const char* err = blur_iter_prepare(fb.winsize);
        if (err) return fail(MAV_ERR_HIP, "winsize %d: %s", fb.winsize, err);
    }

    mav_ctx* c = new mav_ctx();
    c->device = device; c->W = W; c->H = H; c->max_batch = max_batch; c->fb = fb;
    auto bail = [&](int code) { mav_destroy(c); return code; };
#define HIPB(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail(e_ == hipErrorOutOfMemory ? MAV_ERR_OOM : MAV_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); return bail(e_ == hipErrorOutOfMemory ? MAV_ERR_OOM : MAV_ERR_HIP); } } while (0)
    // The compute stream now; the copy stream (overlapped uploads) and the pair stream (two pairs in flight) with their events when a
    // call first needs them (ensure_copy_stream / ensure_pair_stream): a context that serves one-pair calls as one of several LANES
    // (mavflow/pipeline.py) then owns exactly one stream, i.e. one hardware queue of the runtime's small pool -- streams beyond the
    // pool's size share queues, and two lanes whose streams share one do not overlap at all.
    HIPB(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPB(hipEventCreate(&c->t0));
    HIPB(hipEventCreate(&c->t1));
    if (!prepare_poly(fb.poly_n, fb.poly_sigma, &c->pc)) { fail(MAV_ERR_ARG, "poly_sigma %g gives a singular moment matrix", fb.poly_sigma); return bail(MAV_ERR_ARG); }

    // layer selection exactly as optflowgf.cpp (A.1): levels is the number of EXTRA layers actually reachable
    int levels = 0;
    {
        double scale = 1;
        for (levels = 0; levels < fb.levels; levels++) {
            scale *= fb.pyr_scale;
            if (W * scale < 32 || H * scale < 32) break;
        }
    }
    c->layers.resize(levels + 1);
    for (int k = 0; k <= levels; k++) {
        Layer& l = c->layers[k];
        double scale = 1;
        for (int i = 0; i < k; i++) scale *= fb.pyr_scale;
        l.sigma = (1. / scale - 1) * 0.5;
        l.ksize = cv_round(l.sigma * 5) | 1;
        if (l.ksize < 3) l.ksize = 3;
        l.w = cv_round(W * scale);
        l.h = cv_round(H * scale);
        if (l.w < 1 || l.h < 1) { fail(MAV_ERR_ARG, "layer %d collapses to %dx%d", k, l.w, l.h); return bail(MAV_ERR_ARG); }
        std::vector<float> g;
        gaussian_kernel(l.ksize, l.sigma, g);
        HIPB(hipMalloc(&l.g, g.size() * sizeof(float)));
        HIPB(hipMemcpy(l.g, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice));
        if (k > 0) {                                    // the layer's resize coordinates, once (the kernels used to evaluate them per thread, in double)
            std::vector<int> tab(2 * (size_t)(l.w + l.h));
            for (int x = 0; x < l.w; x++) resize_coord_host(x, W, l.w, (double)W / l.w, &tab[x], (float*)&tab[l.w + x]);
            for (int y = 0; y < l.h; y++) resize_coord_host(y, H, l.h, (double)H / l.h, &tab[2 * l.w + y], (float*)&tab[2 * l.w + l.h + y]);
            HIPB(hipMalloc(&l.coord, tab.size() * sizeof(int)));
            HIPB(hipMemcpy(l.coord, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    c->n0 = (size_t)W * H;
    c->n1 = levels >= 1 ? (size_t)c->layers[1].w * c->layers[1].h : 0;
    c->c_off.assign(levels + 1, 0); c->c_stride.assign(levels + 1, 0);
    for (int k = 1; k <= levels; k++) {                 // slot strides rounded up to 64 floats: every slot stays 256-byte aligned
        c->c_stride[k] = ((size_t)c->layers[k].w * c->layers[k].h + 63) & ~(size_t)63;
        c->c_off[k] = c->c_total;
        c->c_total += c->c_stride[k];
    }
    for (int k = levels; k >= 1; k--)                   // deep layers: the top of the pyramid, every layer at most 1/deep_frac of the frame
        if ((size_t)c->layers[k].w * c->layers[k].h * (size_t)c->deep_frac <= c->n0) c->kd = k; else break;
    c->deep_cap = max_batch < 64 ? max_batch : 64;
    c->htmp_stride = (size_t)H * W;                    // any layer (even layer 0 when its fast form does not apply) fits
    // group: pairs per launch for everything but the finest layer's sweeps (see flow_group).
    // (1080p, 64 pairs: 27.0 - 27.5 ms with groups of 16 or 32, 27.8 - 28.1 with 8, 28.4 with 4 -- the batched blur / expansion
    // launches of 16 pairs run 10 - 15 % faster than two of 8; at 3840x2160 8 and 16 are equal, 4 is slower: profiles/r02/ab_group*.log)
    int group = (size_t)W * H <= ((size_t)4 << 20) ? 16 : 8;
    if (group > max_batch) group = max_batch;
    // One pair's finest-layer working set (80 B/px) fits the 256 MB Infinity Cache up to ~2.6 Mpx.  Beyond that a pair is swept
    // band by band (sweeps_band_major): the one-stream schedule ("pairs_in_flight" = 1) over bands of at most ~230 MB (measured at
    // 3840x2160, 16 pairs: 3 bands 31.6 ms, 4 bands 32.4, 5 bands 34.3, 6 bands 35.0, batched sweep-major 35.4); the default
    // two-stream schedule sizes its own bands ("band_mb").
    if ((size_t)W * H * 80 > (size_t)200 << 20) {
        c->bands = (int)(((size_t)W * H * 80 + ((size_t)230 << 20) - 1) / ((size_t)230 << 20));
        if (c->bands > 8) c->bands = 8;
    }
    c->bands_auto = c->bands;
    c->group = group;                                   // the workspace itself comes with the first call that computes flow (ensure_workspace)
    c->small_g = small_group_cap(c, group);
    const size_t B = (size_t)max_batch;
    HIPB(hipMalloc(&c->foe_dev, sizeof(double) * 2 * B));
    HIPB(hipMalloc(&c->box_acc, sizeof(int32_t) * 4 * B));
    HIPB(hipMalloc(&c->u64_scratch, sizeof(unsigned long long) * 8 * B));   // two count blocks of 4 per pair
    HIPB(hipMalloc(&c->i32_scratch, sizeof(int) * B));
    HIPB(hipMalloc(&c->derot_dev, sizeof(DerotParams) * B));
    HIPB(hipMalloc(&c->foe_sc.count, sizeof(int) * B));
    HIPB(hipMalloc(&c->foe_sc.best_key, sizeof(unsigned long long) * B));
    HIPB(hipMalloc(&c->foe_sc.done, sizeof(unsigned) * 2 * B));
    HIPB(hipMemset(c->foe_sc.done, 0, sizeof(unsigned) * 2 * B));
#undef HIPB
    *out = c;
    return MAV_OK;
}

// ---- options: every scheduling / tuning switch of the library lives here (no environment variables) -----------------------------
struct OptionDesc { const char* name; long lo, hi; };
static const OptionDesc kOptions[] = {
    {"group", 1, 1 << 20}, {"group_fine", 0, 1 << 20}, {"bands", 0, 8}, {"pairs_in_flight", 1, 2}, {"band_mb", 8, 1 << 20},
    {"coarse_cache_mb", 0, 1 << 20}, {"coarse_half", 0, 1 << 20}, {"share_m", 0, 1}, {"share_frames", 0, 1}, {"strip", 0, 1 << 20},
    {"phi_screen", 0, 1}, {"phi_yloop", 0, 1 << 20}, {"small_batch", 0, 1}, {"sweep_write_through", -1, 1}, {"deep_batch", 0, 1},
    {"coarse_bands", 0, 1}, {"band_phase", 0, 64}, {"band_skew", -1, 64}, {"deep_frac", 1, 1 << 20},
};
static long* option_slot(mav_ctx* c, const char* name, long* tmp)
{
    // int / bool members behind one long-typed view: *tmp carries the value, write_back stores it
    struct { const char* n; long v; } cur[] = {
        {"group", c->group}, {"group_fine", c->group_fine}, {"bands", c->bands}, {"pairs_in_flight", c->pairs_in_flight},
        {"band_mb", c->pif_band_mb}, {"coarse_cache_mb", c->coarse_cache_mb}, {"coarse_half", c->coarse_half}, {"share_m", c->share_m},
        {"share_frames", c->share_frames}, {"strip", c->strip}, {"phi_screen", c->phi_screen}, {"phi_yloop", c->phi_yloop},
        {"small_batch", c->small_batch}, {"sweep_write_through", c->sweep_wt}, {"deep_batch", c->deep_batch}, {"coarse_bands", c->coarse_bands}, {"band_phase", c->band_phase}, {"band_skew", c->band_skew}, {"deep_frac", c->deep_frac},
    };
    for (auto& e : cur) if (!strcmp(e.n, name)) { *tmp = e.v; return tmp; }
    return nullptr;
}
extern "C" int mav_get_option(mav_ctx* c, const char* name, long* value)
{
    if (!c || !name || !value) return fail(MAV_ERR_ARG, "mav_get_option: NULL argument");
    if (!strcmp(name, "upload_threads")) { *value = c->upload_threads; return MAV_OK; }
    if (!strcmp(name, "inline_uploads")) { *value = c->inline_uploads; return MAV_OK; }
    if (!strcmp(name, "stream_priority")) { *value = c->stream_priority; return MAV_OK; }
    long tmp;
    if (!option_slot(c, name, &tmp)) return fail(MAV_ERR_ARG, "unknown option '%s'", name);
    *value = tmp;
    return MAV_OK;
}
extern "C" int mav_set_option(mav_ctx* c, const char* name, long value)
{
    if (!c || !name) return fail(MAV_ERR_ARG, "mav_set_option: NULL argument");
    if (!strcmp(name, "inline_uploads")) {      // not part of the launch schedule either
        if (value < 0 || value > 1) return fail(MAV_ERR_ARG, "option 'inline_uploads' must be 0 or 1, got %ld", value);
        HIPCHK(hipSetDevice(c->device));
        if (c->copy_stream) HIPCHK(hipStreamSynchronize(c->copy_stream));      // nothing of the old mode is left in flight
        HIPCHK(hipStreamSynchronize(c->stream));
        c->inline_uploads = value != 0;
        return MAV_OK;
    }
    if (!strcmp(name, "stream_priority")) {     // not part of the launch schedule
        // The HIP runtime maps streams onto a pool of (by default four) hardware queues PER PRIORITY CLASS, least-used first; which
        // queue a new stream gets depends on every stream the process has ever made.  Lanes -- contexts that take a stream of small calls
        // in turn and must not share a queue (pipeline.py) -- ask for a class of their own: -1 = high.  The context's compute stream is
        // re-created in that class; nothing may be in flight (the call drains it).
        int lo = 0, hi = 0;
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));            // lo = least (numerically largest), hi = greatest
        if (value < hi || value > lo) return fail(MAV_ERR_ARG, "option 'stream_priority' must be in [%d, %d], got %ld", hi, lo, value);
        if ((int)value == c->stream_priority) return MAV_OK;
        CHK(mav_worker_drain(c));
        CHK(sync_all_streams(c));
        if (c->copy_stream) HIPCHK(hipStreamSynchronize(c->copy_stream));
        hipStream_t fresh = nullptr;
        HIPCHK(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, (int)value));
        HIPCHK(hipStreamDestroy(c->stream));
        c->stream = fresh;
        c->stream_priority = (int)value;
        return MAV_OK;
    }
    if (!strcmp(name, "upload_threads")) {      // host side of mav_upload_gather; not part of the launch schedule (mav_schedule_info)
        if (value < 1 || value > 64) return fail(MAV_ERR_ARG, "option 'upload_threads' must be in [1, 64], got %ld", value);
        if (c->stager) return fail(MAV_ERR_STATE, "upload_threads must be set before the first mav_upload_gather call");
        c->upload_threads = (int)value;
        return MAV_OK;
    }
    const OptionDesc* d = nullptr;
    for (const auto& o : kOptions) if (!strcmp(o.name, name)) d = &o;
    if (!d) return fail(MAV_ERR_ARG, "unknown option '%s'", name);
    if (value < d->lo || value > d->hi) return fail(MAV_ERR_ARG, "option '%s' must be in [%ld, %ld], got %ld", name, d->lo, d->hi, value);
    const int v = (int)value;
    if (!strcmp(name, "group")) {
        const int g = v > c->max_batch ? c->max_batch : v;
        HIPCHK(hipSetDevice(c->device));
        CHK(sync_all_streams(c));
        if (g == c->group) return MAV_OK;
        if (!c->ws_ready) { c->group = g; c->small_g = small_group_cap(c, g); return MAV_OK; }    // nothing allocated yet: the plan changes
        return alloc_group(c, g);
    }
    if (!strcmp(name, "group_fine")) { c->group_fine = v; c->group_fine_set = true; }
    else if (!strcmp(name, "bands")) {          // 0 = back to automatic
        c->bands_set = v > 0;
        c->bands = v > 0 ? v : c->bands_auto;
    }
    else if (!strcmp(name, "pairs_in_flight")) c->pairs_in_flight = v;
    else if (!strcmp(name, "band_mb")) c->pif_band_mb = v;
    else if (!strcmp(name, "coarse_cache_mb")) c->coarse_cache_mb = v;
    else if (!strcmp(name, "coarse_half")) c->coarse_half = v;
    else if (!strcmp(name, "share_m")) c->share_m = v != 0;
    else if (!strcmp(name, "share_frames")) c->share_frames = v != 0;
    else if (!strcmp(name, "strip")) c->strip = v;
    else if (!strcmp(name, "phi_screen")) c->phi_screen = v != 0;
    else if (!strcmp(name, "phi_yloop")) c->phi_yloop = v;
    else if (!strcmp(name, "small_batch")) c->small_batch = v != 0;
    else if (!strcmp(name, "sweep_write_through")) c->sweep_wt = v;
    else if (!strcmp(name, "deep_batch")) c->deep_batch = v != 0;
    else if (!strcmp(name, "coarse_bands")) c->coarse_bands = v != 0;
    else if (!strcmp(name, "band_phase")) c->band_phase = v;
    else if (!strcmp(name, "band_skew")) c->band_skew = v;
    else if (!strcmp(name, "deep_frac")) {
        if (c->ws_ready) return fail(MAV_ERR_STATE, "deep_frac must be set before the first call that computes flow");
        c->deep_frac = v; c->kd = 0;
        for (int k = (int)c->layers.size() - 1; k >= 1; k--)
            if ((size_t)c->layers[k].w * c->layers[k].h * (size_t)v <= c->n0) c->kd = k; else break;
    }
    return MAV_OK;
}

extern "C" int mav_num_layers(const mav_ctx* c) { return c ? (int)c->layers.size() : 0; }
extern "C" int mav_layer_dims(const mav_ctx* c, int k, int* w, int* h, int* ksize, double* sigma)
{
    if (!c || k < 0 || k >= (int)c->layers.size()) return fail(MAV_ERR_ARG, "mav_layer_dims: bad layer %d", k);
    if (w) *w = c->layers[k].w; if (h) *h = c->layers[k].h; if (ksize) *ksize = c->layers[k].ksize; if (sigma) *sigma = c->layers[k].sigma;
    return MAV_OK;
}

// Device memory: what the GPU has free / in total (hipMemGetInfo) and what THIS context holds -- the Farneback workspace (0 until a
// call computes flow), the flow workspace, the detection scratch, the staging blocks of the host-pointer calls, the window-search buffers.
extern "C" int mav_mem_info(mav_ctx* c, size_t* dev_free, size_t* dev_total, size_t* ctx_bytes, size_t* workspace_bytes)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_mem_info: NULL context");
    HIPCHK(hipSetDevice(c->device));
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    if (dev_free) *dev_free = fr;
    if (dev_total) *dev_total = tot;
    if (workspace_bytes) *workspace_bytes = c->ws_bytes;
    if (ctx_bytes) {
        const size_t B = (size_t)c->max_batch;
        size_t n = c->ws_bytes + c->pyr_ws_bytes;
        if (c->flow_ws) n += sizeof(float) * 2 * c->n0 * B;
        if (c->sat) n += sizeof(unsigned long long) * (size_t)(c->W + 1) * (c->H + 1) * B;
        if (c->foe_sc.cand) n += sizeof(double) * 2 * (size_t)c->foe_sc_n * B;
        n += B * (sizeof(double) * 2 + sizeof(int32_t) * 4 + sizeof(unsigned long long) * 8 + sizeof(int) + sizeof(DerotParams) + sizeof(int) +
                  sizeof(unsigned long long) + sizeof(unsigned) * 2);
        for (const auto& l : c->layers) n += sizeof(float) * l.ksize + (l.coord ? sizeof(int) * 2 * (size_t)(l.w + l.h) : 0);
        for (const auto& b : c->scratch) n += b.cap;
        *ctx_bytes = n;
    }
    return MAV_OK;
}

extern "C" int mav_sync(mav_ctx* c)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_sync: NULL context");
    HIPCHK(hipStreamSynchronize(c->stream));
    return MAV_OK;
}
extern "C" void* mav_stream(mav_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int mav_dev_alloc(mav_ctx* c, size_t bytes, void** out)
{
    if (!c || !out) return fail(MAV_ERR_ARG, "mav_dev_alloc: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMalloc(out, bytes ? bytes : 1));
    return MAV_OK;
}
extern "C" int mav_dev_free(mav_ctx* c, void* p)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_dev_free: NULL context");
    if (p) HIPCHK(hipFree(p));
    return MAV_OK;
}
extern "C" int mav_memcpy_h2d(mav_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c || !dst || !src) return fail(MAV_ERR_ARG, "mav_memcpy_h2d: NULL argument");
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MAV_OK;
}
extern "C" int mav_memcpy_d2h(mav_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!c || !dst || !src) return fail(MAV_ERR_ARG, "mav_memcpy_d2h: NULL argument");
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MAV_OK;
}

// Pinned host memory + uploads on a second stream: the next batch's frames cross PCIe while the current batch computes.
extern "C" int mav_host_alloc(mav_ctx* c, size_t bytes, void** out)
{
    if (!c || !out) return fail(MAV_ERR_ARG, "mav_host_alloc: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return MAV_OK;
}
extern "C" int mav_host_free(mav_ctx* c, void* p)
{
    (void)c;                     // page-locked memory outlives the context that allocated it (NULL context allowed)
    if (p) HIPCHK(hipHostFree(p));
    return MAV_OK;
}
extern "C" int mav_upload_async(mav_ctx* c, void* dst_dev, const void* src_host, size_t bytes)
{
    if (!c || !dst_dev || !src_host) return fail(MAV_ERR_ARG, "mav_upload_async: NULL argument");
    if (c->inline_uploads) { HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream)); return MAV_OK; }
    CHK(ensure_copy_stream(c));
    // the copy may overwrite a buffer that work already enqueued on the compute stream still reads (the previous user of a
    // double-buffered set): order the copy stream behind everything enqueued there so far
    HIPCHK(hipEventRecord(c->compute_mark, c->stream));
    HIPCHK(hipStreamWaitEvent(c->copy_stream, c->compute_mark, 0));
    HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->copy_stream));
    return MAV_OK;
}
extern "C" int mav_upload_async_unordered(mav_ctx* c, void* dst_dev, const void* src_host, size_t bytes)
{
    if (!c || !dst_dev || !src_host) return fail(MAV_ERR_ARG, "mav_upload_async_unordered: NULL argument");
    if (c->inline_uploads) { HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream)); return MAV_OK; }
    CHK(ensure_copy_stream(c));
    // no wait for the compute stream: the caller vouches that nothing enqueued so far touches dst_dev (a buffer set that work
    // already enqueued does not use), so the copy overlaps that work whatever the order of the two calls
    HIPCHK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->copy_stream));
    return MAV_OK;
}
extern "C" int mav_upload_fence(mav_ctx* c)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_upload_fence: NULL context");
    if (c->inline_uploads || !c->copy_stream) return MAV_OK;          // the copies ARE on the compute stream / there have been none
    HIPCHK(hipEventRecord(c->copy_done, c->copy_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->copy_done, 0));   // work enqueued after this call sees the uploaded bytes
    return MAV_OK;
}

static int ensure_gather_event(mav_ctx* c)
{
    if (!c->gather_done) HIPCHK(hipEventCreateWithFlags(&c->gather_done, hipEventDisableTiming));
    return MAV_OK;
}
static int ensure_stager(mav_ctx* c)
{
    if (c->stager) return MAV_OK;
    Stager* s = new Stager();
    for (int i = 0; i < Stager::NCHUNK; i++) {
        if (hipHostMalloc(&s->chunk[i], Stager::CHUNK, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s->sent[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            s->shutdown();
            delete s;
            return fail(MAV_ERR_OOM, "page-locked staging ring (%d x %zu bytes)", (int)Stager::NCHUNK, (size_t)Stager::CHUNK);
        }
    }
    for (int t = 1; t < c->upload_threads; t++) s->workers.emplace_back([s] { s->worker(); });   // the calling thread is the first copier
    c->stager = s;
    return MAV_OK;
}
// 1: page-locked host memory (send it from where it is); 0: plain host memory (stage it); -1: device memory (a caller's mistake)
static int host_ptr_kind(const void* p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof(a));
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return 0; }   // plain malloc'd memory: an error or "unregistered"
    if (a.type == hipMemoryTypeHost) return 1;
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeArray) return -1;
    return a.type == hipMemoryTypeManaged ? 1 : 0;
}

extern "C" int mav_upload_gather(mav_ctx* c, void* dst_dev, const void* const* src_host, int count, size_t bytes_each, int flags)
{
    if (!c || !dst_dev || !src_host || count < 1) return fail(MAV_ERR_ARG, "mav_upload_gather: NULL argument or count < 1");
    for (int i = 0; i < count; i++) if (!src_host[i]) return fail(MAV_ERR_ARG, "mav_upload_gather: source %d is NULL", i);
    if (!bytes_each) return MAV_OK;
    const bool ordered = (flags & MAV_GATHER_ORDERED) != 0, held = (flags & MAV_GATHER_SOURCES_HELD) != 0;
    std::vector<int> kind(count);
    for (int i = 0; i < count; i++) {
        kind[i] = (i > 0 && src_host[i] == src_host[i - 1]) ? kind[i - 1] : host_ptr_kind(src_host[i]);
        if (kind[i] < 0) return fail(MAV_ERR_ARG, "mav_upload_gather: source %d is a device pointer (host arrays expected)", i);
        // "on return every source has been read": a page-locked source is read by the DMA engine when the stream gets there.  With
        // inline uploads that is behind whatever the compute stream still has to run -- waiting for it would drain the lane -- so such
        // a source is staged like a pageable one unless the caller vouches that it holds its sources (MAV_GATHER_SOURCES_HELD)
        if (kind[i] == 1 && !held && c->inline_uploads) kind[i] = 0;
    }
    HIPCHK(hipSetDevice(c->device));
    if (!c->inline_uploads) CHK(ensure_copy_stream(c));
    const hipStream_t cs = c->inline_uploads ? c->stream : c->copy_stream;
    if (ordered && !c->inline_uploads) {   // as mav_upload_async: behind everything enqueued on the compute stream so far
        HIPCHK(hipEventRecord(c->compute_mark, c->stream));
        HIPCHK(hipStreamWaitEvent(c->copy_stream, c->compute_mark, 0));
    }
    char* dst = (char*)dst_dev;
    Stager* s = nullptr;
    size_t fill = 0;                           // bytes staged in the chunk being filled
    size_t chunk_dst = 0;                      // device offset the chunk being filled starts at
    bool direct = false;                       // a page-locked source was sent from where it is
    auto flush = [&]() -> int {                // copy the collected segments, send the chunk
        if (!fill) return MAV_OK;
        const unsigned k = s->next_chunk % Stager::NCHUNK;
        s->copy_all();
        s->segs.clear();
        HIPCHK(hipMemcpyAsync(dst + chunk_dst, s->chunk[k], fill, hipMemcpyHostToDevice, cs));
        HIPCHK(hipEventRecord(s->sent[k], cs));
        s->in_flight[k] = true;
        s->next_chunk++;
        fill = 0;
        return MAV_OK;
    };
    auto open_chunk = [&](size_t dev_off) -> int {   // the next ring slot, once its previous transfer has left it
        const unsigned k = s->next_chunk % Stager::NCHUNK;
        if (s->in_flight[k]) { HIPCHK(hipEventSynchronize(s->sent[k])); s->in_flight[k] = false; }
        chunk_dst = dev_off;
        return MAV_OK;
    };
    for (int i = 0; i < count; i++) {
        const char* src = (const char*)src_host[i];
        const size_t dev_off = (size_t)i * bytes_each;
        if (kind[i] == 1) {                    // straight from where it is; whatever was staged before it goes first (keeps nothing waiting)
            if (s) CHK(flush());
            HIPCHK(hipMemcpyAsync(dst + dev_off, src, bytes_each, hipMemcpyHostToDevice, cs));
            direct = true;
            continue;
        }
        if (!s) { CHK(ensure_stager(c)); s = c->stager; }
        size_t done = 0;
        while (done < bytes_each) {
            if (!fill) CHK(open_chunk(dev_off + done));
            const size_t n = std::min(bytes_each - done, Stager::CHUNK - fill);
            char* into = (char*)s->chunk[s->next_chunk % Stager::NCHUNK] + fill;
            for (size_t o = 0; o < n; o += Stager::PIECE)
                s->segs.push_back({into + o, src + done + o, std::min(Stager::PIECE, n - o)});
            fill += n; done += n;
            if (fill == Stager::CHUNK) CHK(flush());
        }
    }
    if (s) {
        CHK(flush());
        // the staged sources have been read; the ring's last transfers may still be in flight (the next call waits for a slot before it refills it)
    }
    if (direct && !held) {
        // page-locked sources sent from where they are (copy stream: nothing but copies ahead of them): the caller may overwrite them
        // when this call returns, so wait until the engine has read them
        CHK(ensure_gather_event(c));
        HIPCHK(hipEventRecord(c->gather_done, cs));
        HIPCHK(hipEventSynchronize(c->gather_done));
    }
    return MAV_OK;
}

extern "C" int mav_download_async(mav_ctx* c, void* dst_host, const void* src_dev, size_t bytes)
{
    if (!c || !dst_host || !src_dev) return fail(MAV_ERR_ARG, "mav_download_async: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    return MAV_OK;
}

// Markers: "everything enqueued on the context's stream so far" as an object the host can wait for without draining the stream
// (mav_sync waits for the work enqueued AFTER the marker too).  The pipelined loop records one per batch.
extern "C" int mav_marker_create(mav_ctx* c, void** out)
{
    if (!c || !out) return fail(MAV_ERR_ARG, "mav_marker_create: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    hipEvent_t e = nullptr;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = (void*)e;
    return MAV_OK;
}
extern "C" int mav_marker_record(mav_ctx* c, void* marker)
{
    if (!c || !marker) return fail(MAV_ERR_ARG, "mav_marker_record: NULL argument");
    HIPCHK(hipEventRecord((hipEvent_t)marker, c->stream));
    return MAV_OK;
}
extern "C" int mav_marker_wait(mav_ctx* c, void* marker)
{
    (void)c;                     // waiting needs no context: a marker may outlive the one it was recorded on (mav_destroy drains the streams)
    if (!marker) return fail(MAV_ERR_ARG, "mav_marker_wait: NULL marker");
    HIPCHK(hipEventSynchronize((hipEvent_t)marker));
    return MAV_OK;
}
extern "C" int mav_marker_query(mav_ctx* c, void* marker, int* done)
{
    (void)c;
    if (!marker || !done) return fail(MAV_ERR_ARG, "mav_marker_query: NULL argument");
    const hipError_t e = hipEventQuery((hipEvent_t)marker);
    if (e == hipSuccess) { *done = 1; return MAV_OK; }
    if (e == hipErrorNotReady) { (void)hipGetLastError(); *done = 0; return MAV_OK; }
    HIPCHK(e);
    return MAV_OK;
}
extern "C" int mav_marker_destroy(mav_ctx* c, void* marker)
{
    (void)c;
    if (marker) HIPCHK(hipEventDestroy((hipEvent_t)marker));
    return MAV_OK;
}

extern "C" int mav_timer_start(mav_ctx* c)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_timer_start: NULL context");
    HIPCHK(hipEventRecord(c->t0, c->stream));
    return MAV_OK;
}
extern "C" int mav_timer_stop(mav_ctx* c, float* ms)
{
    if (!c || !ms) return fail(MAV_ERR_ARG, "mav_timer_stop: NULL argument");
    HIPCHK(hipEventRecord(c->t1, c->stream));
    HIPCHK(hipEventSynchronize(c->t1));
    HIPCHK(hipEventElapsedTime(ms, c->t0, c->t1));
    return MAV_OK;
}

static int prof_collect(mav_ctx* c)
{
    while (!c->open_runs.empty()) close_run(c, c->open_runs.size() - 1);
    if (c->prof.empty()) return MAV_OK;
    CHK(sync_all_streams(c));
    for (auto& r : c->prof) {
        float ms = 0, t0 = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->prof_ms[r.kid] += ms; c->prof_n[r.kid]++;
            if (c->prof_base && hipEventElapsedTime(&t0, c->prof_base, r.a) == hipSuccess) c->prof_iv.push_back({r.kid, t0, t0 + ms, r.stream});
        }
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    c->prof.clear();
    return MAV_OK;
}
extern "C" int mav_profile_enable(mav_ctx* c, int on)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_profile_enable: NULL context");
    CHK(prof_collect(c));
    if (on) {
        memset(c->prof_ms, 0, sizeof(c->prof_ms)); memset(c->prof_n, 0, sizeof(c->prof_n));
        c->prof_iv.clear();
        if (!c->prof_base) HIPCHK(hipEventCreate(&c->prof_base));
        HIPCHK(hipEventRecord(c->prof_base, c->stream));
    }
    c->profiling = on == 2 ? 2 : (on != 0);
    return MAV_OK;
}
// Time during which at least one launch of the named kernel classes (comma-separated names of mav_profile_get) was running:
// the union of the profiled launches' intervals.  With two pairs in flight launches of one class overlap, their summed
// durations exceed the wall time and this is the figure a rate has to be quoted on.
extern "C" int mav_profile_busy(mav_ctx* c, const char* names, double* busy_ms)
{
    if (!c || !names || !busy_ms) return fail(MAV_ERR_ARG, "mav_profile_busy: NULL argument");
    CHK(prof_collect(c));
    bool want[K_COUNT] = {false};
    for (int i = 0; i < K_COUNT; i++) {
        const char* p = strstr(names, kKernelNames[i]);
        const size_t len = strlen(kKernelNames[i]);
        while (p) {                                          // whole-name match between commas
            if ((p == names || p[-1] == ',') && (p[len] == 0 || p[len] == ',')) { want[i] = true; break; }
            p = strstr(p + 1, kKernelNames[i]);
        }
    }
    std::vector<std::pair<float, float>> iv;
    for (const auto& r : c->prof_iv) if (want[r.kid]) iv.push_back({r.t0, r.t1});
    std::sort(iv.begin(), iv.end());
    double busy = 0;
    float lo = 0, hi = -1;
    for (const auto& x : iv) {
        if (hi < lo || x.first > hi) { if (hi >= lo) busy += hi - lo; lo = x.first; hi = x.second; }
        else if (x.second > hi) hi = x.second;
    }
    if (hi >= lo) busy += hi - lo;
    *busy_ms = busy;
    return MAV_OK;
}
// Every profiled launch (mode 1) or run of launches (mode 2) since mav_profile_enable as an interval: class index (order of
// mav_profile_get), stream (0 = the context's stream, 1 = its second compute stream), start / end in ms since the profile was switched
// on.  *n: capacity in, count out (the total when the arrays are NULL).
extern "C" int mav_profile_intervals(mav_ctx* c, int* n, int* kernel_class, int* stream, float* t0_ms, float* t1_ms)
{
    if (!c || !n) return fail(MAV_ERR_ARG, "mav_profile_intervals: NULL argument");
    CHK(prof_collect(c));
    const int total = (int)c->prof_iv.size();
    if (!kernel_class || !stream || !t0_ms || !t1_ms) { *n = total; return MAV_OK; }
    const int k = total < *n ? total : *n;
    for (int i = 0; i < k; i++) {
        kernel_class[i] = c->prof_iv[i].kid; stream[i] = c->prof_iv[i].stream; t0_ms[i] = c->prof_iv[i].t0; t1_ms[i] = c->prof_iv[i].t1;
    }
    *n = k;
    return MAV_OK;
}
extern "C" int mav_profile_get(mav_ctx* c, int* n, const char** names, double* total_ms, long* launches)
{
    if (!c || !n) return fail(MAV_ERR_ARG, "mav_profile_get: NULL argument");
    CHK(prof_collect(c));
    int cap = *n, k = 0;
    for (int i = 0; i < K_COUNT && k < cap; i++, k++) {
        if (names) names[k] = kKernelNames[i];
        if (total_ms) total_ms[k] = c->prof_ms[i];
        if (launches) launches[k] = c->prof_n[i];
    }
    *n = k;
    return MAV_OK;
}

// Calibration: GB/s of a plain 3-reads-1-write float4 streaming kernel over four buffers of bytes_per_buffer each (the sweeps' mix).
extern "C" int mav_membw_probe(mav_ctx* c, size_t bytes_per_buffer, int reps, double* gbs)
{
    if (!c || !gbs || reps < 1 || bytes_per_buffer < 4096) return fail(MAV_ERR_ARG, "mav_membw_probe: bad argument");
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = bytes_per_buffer & ~(size_t)4095;
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = MAV_OK;
    for (int i = 0; i < 4 && rc == MAV_OK; i++)
        if (hipMalloc(&buf[i], bytes) != hipSuccess) rc = fail(MAV_ERR_OOM, "mav_membw_probe: %zu bytes", bytes);
    float ms = 0.f;
    if (rc == MAV_OK) {
        for (int i = 0; i < 3; i++) hipMemsetAsync(buf[i], 0, bytes, c->stream);
        for (int w = 0; w < 2; w++) launch_probe_r3w1(c->stream, buf[0], buf[1], buf[2], buf[3], bytes / 16);
        hipEventRecord(c->t0, c->stream);
        for (int r = 0; r < reps; r++) launch_probe_r3w1(c->stream, buf[0], buf[1], buf[2], buf[3], bytes / 16);
        hipEventRecord(c->t1, c->stream);
        if (hipEventSynchronize(c->t1) != hipSuccess || hipEventElapsedTime(&ms, c->t0, c->t1) != hipSuccess || hipGetLastError() != hipSuccess)
            rc = fail(MAV_ERR_HIP, "mav_membw_probe: launch or timing failed");
    }
    for (float* b : buf) if (b) hipFree(b);
    if (rc == MAV_OK) *gbs = 4.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
    return rc;
}

static int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MAV_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return MAV_OK;
}

// ---- Farneback: the schedule of one group of pairs --------------------------------------------------------------------
static BlurParams blur_of(const mav_ctx* c, const Layer& l)
{
    BlurParams bp{l.ksize, (l.ksize == 3 && l.sigma <= 0) ? 1 : 0, l.g, (double)c->W / l.w, (double)c->H / l.h, nullptr, nullptr, nullptr, nullptr};
    if (l.coord) {
        bp.xs = l.coord; bp.xf = (const float*)(l.coord + l.w);
        bp.ys = l.coord + 2 * l.w; bp.yf = (const float*)(l.coord + 2 * l.w + l.h);
    }
    return bp;
}

// How the sweeps of layer k run for a group of g pairs (plan_sweeps decides, layer_sweeps executes, mav_schedule_info reports).
enum { SW_SEQ = 0, SW_TWO_PAIRS = 1, SW_COARSE_TWO = 2 };
struct SweepPlan {
    int mode;        // SW_SEQ: sub-groups one after the other on one stream; SW_TWO_PAIRS: finest layer, pairs alternate between two
                     // streams, each band-major; SW_COARSE_TWO: coarse layer, sub-groups of `half` pairs alternate between two streams
    int sub;         // pairs per launch (SW_SEQ)
    int J;           // horizontal bands per pair (finest layer, one pair per launch); 1 = sweep-major
    int half;        // SW_COARSE_TWO: pairs per launch
    bool m_per_sub;  // the initial M of a sub-group is built right before its sweeps
};
static SweepPlan plan_sweeps(const mav_ctx* c, int k, int g, bool bands_ok)
{
    const Layer& l = c->layers[k];
    const int T = blur_iter_tile_rows(l.h), I = c->fb.iterations;
    SweepPlan p{SW_SEQ, g, 1, 0, false};
    // The finest layer's ten sweeps re-read R0/R1 and M: run them `group_fine` pairs at a time so that one sub-group's working
    // set stays resident in the 256 MB Infinity Cache between sweeps.  Coarse layers are small: as many pairs per launch as keep
    // the sweeps' working set cache-sized, to fill the 256 CUs.
    if (k == 0 && c->group_fine > 0 && c->group_fine < g) p.sub = c->group_fine;
    if (k > 0 && c->coarse_cache_mb > 0) {
        const size_t ws = (size_t)l.w * l.h * 80, cap = (size_t)c->coarse_cache_mb << 20;
        const int fit = (int)(cap / (ws ? ws : 1));
        if (fit < p.sub) p.sub = fit > 1 ? fit : 1;
    }
    // with per-sub-group sweeps the initial M of a sub-group is built right before its sweeps: M, R0 and R1 are then still in the
    // Infinity Cache when the first sweep reads them (measured -0.5 ms per 64 pairs; doing the same with the blur and the expansion
    // costs more in small launches than it returns)
    p.m_per_sub = p.sub < g;
    const size_t ws_pair = (size_t)l.w * l.h * 80, band_bytes = (size_t)c->pif_band_mb << 20;
    // A COARSE layer whose per-pair working set exceeds a band (layer 1 of the 4K preset: 1536 x 864, 106 MB) is swept exactly like the
    // finest one: pairs alternate between the two streams, each band-major, so that the two pairs in flight occupy 2 x <= band_mb of the
    // Infinity Cache instead of 2 x 106 MB (option "coarse_bands").
    const bool big_coarse = k > 0 && c->coarse_bands && bands_ok && ws_pair > band_bytes && T / (I + 2) >= 2;
    if (big_coarse) { p.sub = 1; p.m_per_sub = true; }
    if (k > 0 && !big_coarse && c->pairs_in_flight == 2 && g >= 2) {
        // COARSE LAYERS with two sub-groups in flight: sub-groups of half the cache-sized count alternate between the compute stream
        // and pair_stream, each with its own M slots, for the same reason as the pairs of the finest layer below -- 25.5 - 25.6 vs
        // 26.0 - 26.6 ms per 64 pairs at 1080p with 4 + 4 instead of 8 pairs per launch (3 + 3: 25.7 - 25.8; 8 + 8: 26.4;
        // profiles/r02/ab_coarse_two*.log).
        p.mode = SW_COARSE_TWO;
        p.half = p.sub / 2 > 0 ? p.sub / 2 : 1;
        if (c->coarse_half > 0) p.half = c->coarse_half;
        if (2 * p.half > g) p.half = (g + 1) / 2;
        return p;
    }
    if ((k == 0 || big_coarse) && c->pairs_in_flight == 2 && p.m_per_sub && p.sub == 1 && g >= 2) {
        // TWO PAIRS IN FLIGHT (finest layer, one pair per launch).  Pair s of the group runs on stream s & 1 -- the compute stream
        // and pair_stream -- and ping-pongs M through slot s & 1.  The two streams never wait for each other inside the group
        // (different pairs: no dependency; one fork and one join event per group), so one stream's launches fill the kernel
        // boundaries and the fill / drain of the other's.  What keeps this inside the 256 MB Infinity Cache is the band-major order:
        // each pair is swept (and its initial M built) band by band, bands of at most band_mb (96 MB) of working set -- 2 bands at 1080p, 7 at
        // 3840x2160 -- so the hot set is 2 x 83 MB, what ONE whole 1080p pair occupies in the one-stream schedule.  Measured
        // (profiles/r02/ab_two_pairs*.log): 1080p 25.9 - 26.3 vs 27.1 - 27.4 ms per 64 pairs, 4K 28.7 vs 30.1 ms per 16 pairs; two full
        // pairs without bands 28.6 ms, three or four streams 27.9 - 28.1 ms.  Same tiles, same arithmetic as every other schedule:
        // bit-identical flow (tests/test_gpu_flow.py).
        int J = (c->bands_set && k == 0) ? c->bands : (int)((ws_pair + band_bytes - 1) / band_bytes);
        const int Jmax = T / (I + 2);          // a band needs iterations + 2 tile rows (the skew must not reach the image top)
        if (J > Jmax) J = Jmax;
        if (J < 1) J = 1;
        if (J == 1 || bands_ok) { p.mode = SW_TWO_PAIRS; p.J = J; return p; }
    }
    // one stream; bands (finest layer, one pair per launch): at least iterations + 2 tile rows each
    if (k == 0 && (p.m_per_sub || g == 1) && p.sub == 1 && T >= (I + 2) * c->bands && bands_ok) p.J = c->bands;
    return p;
}

// The finest layer's sweeps of one pair in BAND-MAJOR order, for frames whose per-pair working set (80 B per pixel: M in, M out,
// R0, R1) does not fit the 256 MB Infinity Cache -- 664 MB at 3840x2160.  The image is cut into J horizontal bands of tile rows
// and ALL sweeps of a band run before the next band starts, so that a band's M, R0 and R1 stay cache-resident across its sweeps
// exactly as a whole 1080p pair does.  What makes this legal without recomputing halos is a skew: sweep `it` processes band j as
// tile rows [A_j - it, A_(j+1) - it) (16 pixel rows per tile row; the first band starts at row 0, the last ends at the bottom).
//   * reads: sweep it on band j needs sweep it - 1 within 6 pixels of its rows = output of (it - 1, j), just computed, and of
//     (it - 1, j - 1), computed with the previous band;
//   * the M ping-pong: (it, j) writes the buffer that (it - 1, j + 1) will read later, but only up to tile row A_(j+1) - it - 1,
//     while that reader starts 6 pixels above tile row A_(j+1) - it + 1; and the rows of band j - 1 that (it, j) itself reads have
//     been overwritten by (it + 1, j - 1) only up to one tile row above them.  A skew of one tile row (16 >= 6 pixels) covers both.
// Every tile is computed exactly once on the same tile grid: results are bit-identical to the sweep-major schedule
// (tests/test_gpu_flow.py).  One stream, no events.
// upd != nullptr: the initial M (UpdateMatrices from the coarser layer's flow) is built band by band too, right before a band's
// first sweep: pixel rows [16 a0 - 8, 16 a1 + 8) -- what that sweep reads (6-pixel halo) -- which lie below everything the bands
// above have written into Ma (their odd sweeps end one whole tile row higher); the rows two neighbouring bands both need are
// simply built twice, to the same values.
struct BandUpdate { const float* flow_prev; size_t fc_stride; int pw, ph; float mul; };
// phase = 1 (the pairs of the second stream, option "band_phase", off by default): the partition is shifted by half a band -- J + 1 bands,
// the first and the last of half size.  Two streams that start a group together with the same partition stay in lockstep: both build a
// band's initial M (HBM-bound) at the same moments and both sweep (cache-bound) at the same moments -- untraced at 3840x2160: 1.3 ms per
// step with two initial-M launches running and no sweep, 1.0 ms with one (tools/untraced_anatomy.py).  Shifted by half a band, one
// stream's initial M does fall into the other's sweeps (3.6 ms per step) -- and the step gets SLOWER: 581 vs 602 pairs/s at 4K, 2 320 vs
// 2 630 at 1080p (profiles/r04/ab_band_phase.log): the 66 MB a band's initial M streams in from HBM push the sweeping pair's band out
// of the Infinity Cache (sweep launches 37.1 vs 34.6 ms summed).  The lockstep is worth keeping.  The band arguments above hold for any
// monotone sequence of boundaries (a band whose rows have all moved above the image top at a late sweep is empty and skipped; its
// successor then starts at row 0): bit-identical (tests/test_gpu_flow.py).
static void sweeps_band_major(mav_ctx* c, hipStream_t st, int kid, float* Ma, float* Mb, size_t ms, const float* r0, const float* r1, size_t rs, int gs,
                              int lw, int lh, int T, int J, float* fo, size_t fstride, const BandUpdate* upd = nullptr, bool two_streams = false,
                              int phase = 0)
{
    const bool wt = c->sweep_wt < 0 ? two_streams : c->sweep_wt != 0;
    const int I = c->fb.iterations;
    const int NBands = phase ? J + 1 : J;
    auto bound = [&](int j) -> int {                      // first tile row of band j; bound(NBands) = T
        if (j <= 0) return 0;
        if (j >= NBands) return T;
        if (phase) return (int)((long long)T * (2 * j - 1) / (2 * J));
        // Sweep `it` shifts every boundary up by `it` tile rows: over a band's sweeps the first band loses (I - 1) / 2 rows on average
        // and the last one gains as many.  Boundaries moved down by that amount give every band the same AVERAGE size -- launches and
        // cache footprints stay even (option "band_skew"; 1080p: first band 38 of 68 tile rows instead of 34; equal split: 0).
        int b = (int)((long long)T * j / J) + (c->band_skew < 0 ? (I - 1) / 2 : c->band_skew);
        const int lo = j, hi = T - (NBands - j);             // at least one tile row per band
        return b < lo ? lo : (b > hi ? hi : b);
    };
    J = NBands;
    for (int j = 0; j < J; j++) {
        const int a0 = bound(j), a1 = bound(j + 1);
        if (a1 <= a0) continue;
        if (upd) {
            ProfScope ps(c, K_UPDATE, st);
            launch_update_matrices(st, r0, r1, rs, upd->flow_prev, upd->fc_stride, upd->pw, upd->ph, upd->mul, gs, lw, lh, Ma, ms,
                                   a0 == 0 ? 0 : a0 * 16 - 8, j == J - 1 ? lh : a1 * 16 + 8);
        }
        for (int it = 0; it < I; it++) {
            const int update = it < I - 1;
            int ty0 = a0 - it, ty1 = j == J - 1 ? T : a1 - it;
            if (ty0 < 0) ty0 = 0;
            if (ty1 <= ty0) continue;
            ProfScope ps(c, kid, st);
            launch_blur_iter(st, (it & 1) ? Mb : Ma, (it & 1) ? Ma : Mb, ms, r0, r1, rs, gs, lw, lh, c->fb.winsize, update, !update, fo,
                             fstride, ty0, ty1, c->strip, wt);
        }
    }
}

// initial M + the `iterations` sweeps of gs pairs, sweep-major, on stream ss
static void sweeps_plain(mav_ctx* c, hipStream_t ss, int kid, float* Min, float* Mout, size_t ms, const float* r0, const float* r1, size_t rs, int gs,
                         const Layer& l, float* fo, size_t fstride, bool two_streams = false)
{
    const bool wt = c->sweep_wt < 0 ? two_streams : c->sweep_wt != 0;
    for (int it = 0; it < c->fb.iterations; it++) {
        const int update = it < c->fb.iterations - 1;
        { ProfScope ps(c, kid, ss);
          launch_blur_iter(ss, Min, Mout, ms, r0, r1, rs, gs, l.w, l.h, c->fb.winsize, update, !update, fo, fstride, 0, -1, c->strip, wt); }
        if (update) { float* t = Min; Min = Mout; Mout = t; }
    }
}

// Initial M and the `iterations` sweeps of layer k for g pairs whose expansions lie at r0 / r1 (slot stride rs), starting on
// stream st.  flow_prev = the coarser layer's flow (pw x ph, slot stride fc_stride; nullptr at the top layer); the layer's flow goes to
// fdst (slot stride fstride).  M ping-pongs through Ma / Mb (slot stride ms).  On return everything has been joined back into st.
static int layer_sweeps(mav_ctx* c, hipStream_t st, int k, int g, const float* r0g, const float* r1g, size_t rs, const float* flow_prev,
                        size_t fc_stride, int pw, int ph, float* fdst, size_t fstride, float* Ma, float* Mb, size_t ms)
{
    const Layer& l = c->layers[k];
    const float mul = (float)(1. / c->fb.pyr_scale);
    const bool bands_ok = blur_iter_bands_ok(l.w, c->fb.winsize, ms, rs, fstride, Ma, Mb, r0g, r1g, fdst);
    const SweepPlan p = plan_sweeps(c, k, g, bands_ok);
    const int kid = k == 0 ? K_ITER : K_ITER_COARSE;
    const int T = blur_iter_tile_rows(l.h);
    if (p.mode == SW_TWO_PAIRS) {
        CHK(ensure_pair_stream(c));
        HIPCHK(hipEventRecord(c->pif_fork, st));
        HIPCHK(hipStreamWaitEvent(c->pair_stream, c->pif_fork, 0));
        for (int s0 = 0; s0 < g; s0++) {
            const hipStream_t ss = (s0 & 1) ? c->pair_stream : st;
            float *Min = Ma + (size_t)(s0 & 1) * ms, *Mout = Mb + (size_t)(s0 & 1) * ms;
            const float *r0 = r0g + (size_t)s0 * rs, *r1 = r1g + (size_t)s0 * rs;
            float* fo = fdst + (size_t)s0 * fstride;
            const BandUpdate bu{flow_prev ? flow_prev + (size_t)s0 * fc_stride : nullptr, fc_stride, pw, ph, mul};
            if (p.J > 1) {
                sweeps_band_major(c, ss, kid, Min, Mout, ms, r0, r1, rs, 1, l.w, l.h, T, p.J, fo, fstride, &bu, true,
                                  (c->band_phase && (s0 & 1) && p.J >= c->band_phase) ? 1 : 0);
                continue;
            }
            { ProfScope ps(c, K_UPDATE, ss);
              launch_update_matrices(ss, r0, r1, rs, bu.flow_prev, fc_stride, pw, ph, mul, 1, l.w, l.h, Min, ms); }
            sweeps_plain(c, ss, kid, Min, Mout, ms, r0, r1, rs, 1, l, fo, fstride, true);
        }
        prof_close_stream(c, c->pair_stream); prof_close_stream(c, st);
        HIPCHK(hipEventRecord(c->pif_join, c->pair_stream));
        HIPCHK(hipStreamWaitEvent(st, c->pif_join, 0));
        return MAV_OK;
    }
    if (p.mode == SW_COARSE_TWO) {
        CHK(ensure_pair_stream(c));
        HIPCHK(hipEventRecord(c->pif_fork, st));
        HIPCHK(hipStreamWaitEvent(c->pair_stream, c->pif_fork, 0));
        int idx = 0;
        for (int s0 = 0; s0 < g; s0 += p.half, idx++) {
            const int gs = g - s0 < p.half ? g - s0 : p.half;
            const hipStream_t ss = (idx & 1) ? c->pair_stream : st;
            const size_t m_off = (size_t)(idx & 1) * p.half * ms;
            const float *r0 = r0g + (size_t)s0 * rs, *r1 = r1g + (size_t)s0 * rs;
            { ProfScope ps(c, K_UPDATE, ss);
              launch_update_matrices(ss, r0, r1, rs, flow_prev ? flow_prev + (size_t)s0 * fc_stride : nullptr, fc_stride, pw, ph, mul, gs, l.w, l.h,
                                     Ma + m_off, ms); }
            sweeps_plain(c, ss, K_ITER_COARSE, Ma + m_off, Mb + m_off, ms, r0, r1, rs, gs, l, fdst + (size_t)s0 * fstride, fstride, true);
        }
        prof_close_stream(c, c->pair_stream); prof_close_stream(c, st);
        HIPCHK(hipEventRecord(c->pif_join, c->pair_stream));
        HIPCHK(hipStreamWaitEvent(st, c->pif_join, 0));
        return MAV_OK;
    }
    if (!p.m_per_sub) {
        ProfScope ps(c, K_UPDATE, st);
        launch_update_matrices(st, r0g, r1g, rs, flow_prev, fc_stride, pw, ph, mul, g, l.w, l.h, Ma, ms);
    }
    // Sub-groups are swept one after the other on one stream, so they all ping-pong M through the SAME two buffers (the first
    // sub-group's slots; option "share_m"): the M lines then stay hot in the Infinity Cache from pair to pair instead of leaving a
    // dead 83 MB copy behind per pair.  Measured at 1080p, 64 pairs: 27.6 / 28.0 ms shared vs 28.2 / 28.6 ms with per-slot buffers.
    for (int s0 = 0; s0 < g; s0 += p.sub) {
        const int gs = g - s0 < p.sub ? g - s0 : p.sub;
        const size_t m_off = (p.m_per_sub && c->share_m) ? 0 : (size_t)s0 * ms;
        const float *r0 = r0g + (size_t)s0 * rs, *r1 = r1g + (size_t)s0 * rs;
        if (p.m_per_sub) {
            ProfScope ps(c, K_UPDATE, st);
            launch_update_matrices(st, r0, r1, rs, flow_prev ? flow_prev + (size_t)s0 * fc_stride : nullptr, fc_stride, pw, ph, mul, gs, l.w,
                                   l.h, Ma + m_off, ms);
        }
        float* fo = fdst + (size_t)s0 * fstride;
        if (p.J > 1 && gs == 1) sweeps_band_major(c, st, kid, Ma + m_off, Mb + m_off, ms, r0, r1, rs, gs, l.w, l.h, T, p.J, fo, fstride);
        else sweeps_plain(c, st, kid, Ma + m_off, Mb + m_off, ms, r0, r1, rs, gs, l, fo, fstride);
    }
    return MAV_OK;
}

// Layer images and polynomial expansions of both frames of g pairs at layer k, on stream st, through image buffer I into the
// expansion set R (R0 | R1 in one piece); *r0 / *r1 = where pair 0's two expansions lie (pair s: + s * 5 n0).
// seq: `prev` is a run of g + 1 consecutive frames and pair s = (frame s, frame s + 1): every frame is blurred and expanded ONCE.
// Otherwise, when the 2 g layer images are small (merge_frames), prev and next go through ONE blur and ONE expansion launch of 2 g
// images instead of two of g: a group of one or two pairs is bound by launch latency, not by bytes.
static void layer_expansions(mav_ctx* c, hipStream_t st, int k, const uint8_t* prev, const uint8_t* next, int g, bool seq, float* I, float* R,
                             const float** r0, const float** r1)
{
    mav_ctx::WorkSet& w = c->ws;
    const size_t n0 = c->n0;
    const Layer& l = c->layers[k];
    if (seq) {
        { ProfScope ps(c, K_BLUR_RESIZE, st);
          launch_blur_resize(st, prev, nullptr, 0, n0, g + 1, c->W, c->H, l.w, l.h, blur_of(c, l), w.Htmp, c->htmp_stride, I, n0); }
        { ProfScope ps(c, K_POLYEXP, st);
          launch_polyexp(st, I, n0, g + 1, l.w, l.h, c->pc, R, 5 * n0); }
        *r0 = R; *r1 = R + 5 * n0;
        return;
    }
    float* R1 = R + 5 * n0 * (size_t)g;
    *r0 = R; *r1 = R1;
    // (the two-pass blur's scratch holds g + 1 frames)
    const bool no_tmp = !blur_resize_needs_tmp(prev, next, n0, c->W, c->H, l.w, l.h, blur_of(c, l), I, n0);
    const bool merge_frames = (no_tmp || 2 * g <= g + 1) && (size_t)2 * g * l.w * l.h * sizeof(float) <= ((size_t)48 << 20);
    if (merge_frames) {
        { ProfScope ps(c, K_BLUR_RESIZE, st);
          launch_blur_resize(st, prev, next, g, n0, 2 * g, c->W, c->H, l.w, l.h, blur_of(c, l), w.Htmp, c->htmp_stride, I, n0); }
        { ProfScope ps(c, K_POLYEXP, st);
          launch_polyexp(st, I, n0, 2 * g, l.w, l.h, c->pc, R, 5 * n0); }
        return;
    }
    const uint8_t* img[2] = {prev, next};
    float* Rs[2] = {R, R1};
    for (int i = 0; i < 2; i++) {
        { ProfScope ps(c, K_BLUR_RESIZE, st);
          launch_blur_resize(st, img[i], nullptr, 0, n0, g, c->W, c->H, l.w, l.h, blur_of(c, l), w.Htmp, c->htmp_stride, I, n0); }
        { ProfScope ps(c, K_POLYEXP, st);
          launch_polyexp(st, I, n0, g, l.w, l.h, c->pc, Rs[i], 5 * n0); }
    }
}

// does a group of g pairs take the small-group schedule?
static bool is_small_group(const mav_ctx* c, int g)
{
    return c->small_batch && c->layers.size() > 1 && (int)c->layers.size() <= MAV_MAX_JOBS && g <= c->small_g &&
           (size_t)g * c->n0 * 80 <= ((size_t)c->small_batch_mb << 20);
}

// One group of g pairs: every coarse layer completely (top layer first: images, expansions, initial M, sweeps), then the finest layer.
// SMALL GROUPS (is_small_group: one 1080p pair, two 720p pairs ...; BASELINE config 2) are a chain of ~30 dependent launches that
// each fill a fraction of the chip, ~4.5 us of boundary apiece.  For them the whole pyramid's layer images come from ONE launch and
// all expansions from ONE launch (k_blur_multi / k_polyexp_multi: the workgroups of several layers in one grid, every layer into a
// region of its own in Ic / Rc), instead of two launches per layer: the small layers ride along with the finest one.  Same tile
// functions on the same data: bit-identical flow (tests/test_gpu_flow.py).
// (Measured for such groups and not kept: the finest layer's images and expansions on a side stream underneath the coarse chain --
// every event record / wait costs ~6 us on the compute stream and the overlapped kernels slow the chain's own: 0.320 vs 0.306 ms per
// 1280x720 pair, profiles/r03/c2_side_stream.txt; all sweeps of a layer in one launch of resident workgroups that hand M' over through
// flags -- a cross-CU hand-off costs what the kernel boundary costs: 0.419 vs 0.305 ms, profiles/r03/c2_resident_sweeps.txt.  For big
// groups overlapping the finest layer's preparation with the coarse sweeps loses too: profiles/r02/ab_overlap_fine_prep_with_coarse_sweeps.log.)
// The layer images of layers k_lo .. k_hi for F frames (the first `split` from run prev, the rest from run img2; img2 == nullptr: one
// run) into regions Ik(k) with slot stride sk(k): every layer blur_multi_ok accepts through ONE launch, the others (long Gaussians)
// through the two-pass kernels in chunks the H x w scratch holds; then ALL their expansions through one launch (per MAV_MAX_JOBS layers).
template <typename IkFn, typename RkFn, typename SkFn>
static void pyramid_multi(mav_ctx* c, hipStream_t st, const uint8_t* prev, const uint8_t* img2, int split, int F, int k_lo, int k_hi, IkFn Ik,
                          RkFn Rk, SkFn sk)
{
    mav_ctx::WorkSet& w = c->ws;
    const size_t n0 = c->n0;
    BlurJobs bj{0, 0, {}};
    PolyJobs pj{0, 0, {}};
    auto flush_blur = [&]() { if (bj.n) { ProfScope ps(c, K_BLUR_RESIZE, st); launch_blur_multi(st, prev, img2, split, n0, F, c->W, c->H, bj, F > 8); bj.n = 0; } };
    auto flush_poly = [&]() { if (pj.n) { ProfScope ps(c, K_POLYEXP, st); launch_polyexp_multi(st, pj, F, c->pc); pj.n = 0; } };
    for (int k = k_lo; k <= k_hi; k++) {
        const Layer& l = c->layers[k];
        if (blur_multi_ok(prev, img2, n0, c->W, c->H, l.w, l.h, blur_of(c, l), Ik(k), sk(k))) {
            if (bj.n == MAV_MAX_JOBS) flush_blur();
            BlurJob& J = bj.j[bj.n++];
            J.out = Ik(k); J.out_stride = sk(k); J.bp = blur_of(c, l); J.w = l.w; J.h = l.h;
        } else {
            // a long Gaussian (or an unaligned finest layer): launches of its own.  The two-pass scratch holds (group + 1) H x W floats;
            // a frame needs H x w of it
            ProfScope ps(c, K_BLUR_RESIZE, st);
            const size_t per = (size_t)c->H * l.w, cap_frames = (c->htmp_stride * (size_t)(c->group + 1)) / per;
            const int chunk = (int)(cap_frames < (size_t)F ? cap_frames : (size_t)F);
            for (int f0 = 0; f0 < F; f0 += chunk) {
                const int n = F - f0 < chunk ? F - f0 : chunk;
                const bool from2 = img2 && f0 >= split;
                const uint8_t* a = from2 ? img2 + (size_t)(f0 - split) * n0 : prev + (size_t)f0 * n0;
                launch_blur_resize(st, a, from2 ? nullptr : img2, from2 ? 0 : split - f0, n0, n, c->W, c->H, l.w, l.h, blur_of(c, l), w.Htmp, per,
                                   Ik(k) + (size_t)f0 * sk(k), sk(k));
            }
        }
        if (pj.n == MAV_MAX_JOBS) { flush_blur(); flush_poly(); }
        PolyJob& P = pj.j[pj.n++];
        P.I = Ik(k); P.R = Rk(k); P.I_stride = sk(k); P.R_stride = 5 * sk(k); P.w = l.w; P.h = l.h;
    }
    flush_blur();
    flush_poly();
}

// DEEP LAYERS (kd .. top) of D pairs at once, on the compute stream; the flow of layer kd lands in deep.f[kd & 1], slot stride
// 2 * c_stride[kd].  See mav_ctx::DeepSet.  Same tile functions on the same data as the per-group path: bit-identical flow.
static int deep_layers(mav_ctx* c, hipStream_t st, const uint8_t* prev, const uint8_t* next, int D, bool seq)
{
    const int L = (int)c->layers.size(), kd = c->kd;
    const int F = seq ? D + 1 : 2 * D;
    const uint8_t* img2 = seq ? nullptr : next;
    const size_t base = c->c_off[kd];
    auto Ik = [&](int k) { return c->deep.I + (size_t)F * (c->c_off[k] - base); };
    auto Rk = [&](int k) { return c->deep.R + 5 * (size_t)F * (c->c_off[k] - base); };
    auto sk = [&](int k) { return c->c_stride[k]; };
    pyramid_multi(c, st, prev, img2, D, F, kd, L - 1, Ik, Rk, sk);
    const float* flow_prev = nullptr;
    int pw = 0, ph = 0;
    size_t fp_stride = 0;
    for (int k = L - 1; k >= kd; k--) {
        const size_t rs = 5 * sk(k);
        const float *r0 = Rk(k), *r1 = Rk(k) + (seq ? rs : rs * (size_t)D);
        CHK(layer_sweeps(c, st, k, D, r0, r1, rs, flow_prev, fp_stride, pw, ph, c->deep.f[k & 1], 2 * sk(k), c->deep.Ma, c->deep.Mb, 5 * sk(k)));
        flow_prev = c->deep.f[k & 1]; fp_stride = 2 * sk(k); pw = c->layers[k].w; ph = c->layers[k].h;
    }
    return MAV_OK;
}

// One group of g pairs: every coarse layer completely (top layer first: images, expansions, initial M, sweeps), then the finest layer.
// deep_flow != nullptr: the layers from kd up have been computed already (deep_layers); the group starts at layer kd - 1 with
// deep_flow (this group's first pair; slot stride deep_stride) as its coarser layer.
// SMALL GROUPS (is_small_group: one 1080p pair, two 720p pairs ...; BASELINE config 2) are a chain of ~30 dependent launches that
// each fill a fraction of the chip, ~4.5 us of boundary apiece.  For them the whole pyramid's layer images come from ONE launch and
// all expansions from ONE launch (k_blur_multi / k_polyexp_multi: the workgroups of several layers in one grid, every layer into a
// region of its own in Ic / Rc), instead of two launches per layer: the small layers ride along with the finest one.  Same tile
// functions on the same data: bit-identical flow (tests/test_gpu_flow.py).
// (Measured for such groups and not kept: the finest layer's images and expansions on a side stream underneath the coarse chain;
// all sweeps of a layer in one launch of resident workgroups that hand M' over through flags -- HISTORY.md.)
static int flow_group(mav_ctx* c, const uint8_t* prev, const uint8_t* next, int g, bool seq, float* flow_out, const float* deep_flow = nullptr,
                      size_t deep_stride = 0)
{
    mav_ctx::WorkSet& w = c->ws;
    const hipStream_t st = c->stream;
    const int L = (int)c->layers.size();
    const size_t n0 = c->n0, fc_stride = 2 * (c->n1 ? c->n1 : 1);
    const float* flow_prev = deep_flow;
    size_t fp_stride = deep_flow ? deep_stride : fc_stride;
    const int k_top = deep_flow ? c->kd - 1 : L - 1;
    int pw = deep_flow ? c->layers[c->kd].w : 0, ph = deep_flow ? c->layers[c->kd].h : 0;
    if (!deep_flow && is_small_group(c, g)) {
        const int F = seq ? g + 1 : 2 * g;                                // frames: one run of g + 1, or the prev run and the next run
        const uint8_t* img2 = seq ? nullptr : next;
        auto Ik = [&](int k) { return k ? w.Ic + (size_t)F * c->c_off[k] : w.I; };
        auto Rk = [&](int k) { return k ? w.Rc + 5 * (size_t)F * c->c_off[k] : w.R; };
        auto sk = [&](int k) { return k ? c->c_stride[k] : n0; };
        pyramid_multi(c, st, prev, img2, g, F, 0, L - 1, Ik, Rk, sk);
        for (int k = L - 1; k >= 0; k--) {
            const size_t rs = 5 * sk(k);
            const float *r0 = Rk(k), *r1 = Rk(k) + (seq ? rs : rs * (size_t)g);
            CHK(layer_sweeps(c, st, k, g, r0, r1, rs, flow_prev, fc_stride, pw, ph, k ? w.fc[k & 1] : flow_out, k ? fc_stride : 2 * n0, w.Ma, w.Mb, 5 * n0));
            flow_prev = w.fc[k & 1]; pw = c->layers[k].w; ph = c->layers[k].h;
        }
        return MAV_OK;
    }
    const float *r0 = nullptr, *r1 = nullptr;
    for (int k = k_top; k >= 0; k--) {
        layer_expansions(c, st, k, prev, next, g, seq, w.I, w.R, &r0, &r1);
        CHK(layer_sweeps(c, st, k, g, r0, r1, 5 * n0, flow_prev, fp_stride, pw, ph, k ? w.fc[k & 1] : flow_out, k ? fc_stride : 2 * n0, w.Ma, w.Mb, 5 * n0));
        flow_prev = w.fc[k & 1]; fp_stride = fc_stride; pw = c->layers[k].w; ph = c->layers[k].h;
    }
    return MAV_OK;
}

// does a call of `batch` pairs run its deep layers once for the whole call (deep_layers) instead of once per group?
static bool use_deep_batch(const mav_ctx* c, int batch) { return c->deep_batch && c->kd > 0 && batch > c->group; }

extern "C" int mav_farneback_dev(mav_ctx* c, const uint8_t* prev, const uint8_t* next, int batch, float* flow)
{
    if (!c || !prev || !next || !flow) return fail(MAV_ERR_ARG, "mav_farneback: NULL argument");
    if (batch < 1 || batch > c->max_batch) return fail(MAV_ERR_ARG, "batch %d outside [1, %d]", batch, c->max_batch);
    HIPCHK(hipSetDevice(c->device));
    CHK(ensure_workspace(c));
    // A frame SEQUENCE -- the caller's two batches are views of one run of batch + 1 consecutive frames, next = prev + one frame,
    // which is how a video goes through the reference's loop (src/farneback.py:76-80 with prevgray = the last call's frame) -- has
    // every inner frame in two pairs.  Each group then blurs and expands its g + 1 frames once instead of 2 g (same arithmetic per
    // frame: the flow is bit-identical to the two-batch form; tests/test_gpu_flow.py).  Option "share_frames" = 0 switches it off.
    const bool seq = c->share_frames && next == prev + c->n0;
    const bool deep = use_deep_batch(c, batch);
    if (deep) CHK(ensure_deep(c));
    const int chunk = deep ? c->deep_cap : batch;
    for (int d0 = 0; d0 < batch; d0 += chunk) {
        const int D = batch - d0 < chunk ? batch - d0 : chunk;
        const size_t dstride = deep ? 2 * c->c_stride[c->kd] : 0;
        // (Measured and not kept: the deep chain on a stream of its own underneath the first group's top-layer images / expansions --
        // 592 vs 597 pairs/s at 3840x2160 / 5 layers / 16 pairs: the fork / join events cost more than the overlap hides.)
        if (deep) CHK(deep_layers(c, c->stream, prev + (size_t)d0 * c->n0, next + (size_t)d0 * c->n0, D, seq));
        for (int g0 = d0; g0 < d0 + D; g0 += c->group) {
            const int g = d0 + D - g0 < c->group ? d0 + D - g0 : c->group;
            CHK(flow_group(c, prev + (size_t)g0 * c->n0, next + (size_t)g0 * c->n0, g, seq, flow + (size_t)g0 * 2 * c->n0,
                           deep ? c->deep.f[c->kd & 1] + (size_t)(g0 - d0) * dstride : nullptr, dstride));
            CHK(check_launch("farneback kernels"));
        }
    }
    c->last_flow = flow;
    return MAV_OK;
}
extern "C" const float* mav_last_flow_dev(const mav_ctx* c) { return c ? c->last_flow : nullptr; }

// The schedule a call of `batch` pairs takes with the options in effect, as one line of JSON (bench.py prints it and hashes it):
// every option of mav_set_option, the group split and, per layer, how its sweeps run.
extern "C" int mav_schedule_info(mav_ctx* c, int batch, char* buf, size_t cap)
{
    if (!c || !buf || cap < 2) return fail(MAV_ERR_ARG, "mav_schedule_info: NULL argument");
    if (batch < 1 || batch > c->max_batch) return fail(MAV_ERR_ARG, "batch %d outside [1, %d]", batch, c->max_batch);
    std::string o = "{";
    char t[256];
    for (const auto& d : kOptions) {
        long v = 0;
        option_slot(c, d.name, &v);
        snprintf(t, sizeof(t), "\"%s\": %ld, ", d.name, v);
        o += t;
    }
    const int g = batch < c->group ? batch : c->group;
    const bool deep = use_deep_batch(c, batch);
    const int D = deep ? (batch < c->deep_cap ? batch : c->deep_cap) : 0;
    snprintf(t, sizeof(t), "\"pairs_per_group\": %d, \"pyramid_in_two_launches\": %s, \"deep_layers_from\": %d, \"deep_pairs\": %d, \"layers\": [", g,
             (!deep && is_small_group(c, g)) ? "true" : "false", deep ? c->kd : 0, D);
    o += t;
    static const char* const mode_names[] = {"one stream", "two pairs in flight, band-major", "two sub-groups in flight"};
    for (int k = 0; k < (int)c->layers.size(); k++) {
        const Layer& l = c->layers[k];
        const SweepPlan p = plan_sweeps(c, k, (deep && k >= c->kd) ? D : g, l.w % 4 == 0 && c->fb.winsize / 2 == 6);
        snprintf(t, sizeof(t), "%s{\"layer\": %d, \"w\": %d, \"h\": %d, \"blur\": \"%s\", \"sweeps\": \"%s\", \"pairs_per_launch\": %d, \"bands\": %d}",
                 k ? ", " : "", k, l.w, l.h,
                 (l.w == c->W && l.h == c->H) ? "3x3" : (blur_resize_is_fused(c->W, c->H, l.w, l.h, l.ksize) ? "fused" : "two-pass"),
                 mode_names[p.mode], p.mode == SW_COARSE_TWO ? p.half : (p.mode == SW_TWO_PAIRS ? 1 : p.sub), p.J);
        o += t;
    }
    o += "]}";
    if (o.size() + 1 > cap) return fail(MAV_ERR_ARG, "mav_schedule_info: buffer of %zu bytes too small (%zu needed)", cap, o.size() + 1);
    memcpy(buf, o.c_str(), o.size() + 1);
    return MAV_OK;
}

// ---- detection ---------------------------------------------------------------------------------------------
static int ensure_foe_scratch(mav_ctx* c, int N)
{
    if (N <= c->foe_sc_n) return MAV_OK;
    if (c->foe_sc.cand) hipFree(c->foe_sc.cand);
    c->foe_sc.cand = nullptr; c->foe_sc_n = 0;
    HIPCHK(hipMalloc(&c->foe_sc.cand, sizeof(double) * 2 * (size_t)N * c->max_batch));
    c->foe_sc_n = N;
    return MAV_OK;
}

// Per-pair derotation constants.  Device pointers stay on the device (a tiny kernel packs them: no host round trip, the
// stream never drains); host pointers are packed here and copied.
static int upload_derot(mav_ctx* c, const double* omega, const double* dt, const uint8_t* frame0, int batch, bool host_ptrs,
                        const DerotParams** out)
{
    *out = nullptr;
    if (!omega && !frame0) return MAV_OK;
    if (!host_ptrs) {
        launch_make_derot(c->stream, omega, dt, frame0, batch, c->W, c->H, c->derot_dev);
        *out = c->derot_dev;
        return MAV_OK;
    }
    std::vector<DerotParams> dp(batch);
    for (int b = 0; b < batch; b++) {
        const double d = dt ? dt[b] : 1.0;
        dp[b].o0 = omega ? omega[3 * b] : 0.0; dp[b].o1 = omega ? omega[3 * b + 1] : 0.0; dp[b].o2 = omega ? omega[3 * b + 2] : 0.0;
        dp[b].sx = c->W * d / 2;   // w * dt / 2   (detector.py:101)
        dp[b].sy = c->H * d / 2;
        dp[b].mode = (frame0 && frame0[b]) ? MAV_PAIR_FRAME0 : (omega ? MAV_PAIR_DEROTATE : MAV_PAIR_PROMOTE);
    }
    HIPCHK(hipMemcpyAsync(c->derot_dev, dp.data(), sizeof(DerotParams) * batch, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));  // dp is a local vector
    *out = c->derot_dev;
    return MAV_OK;
}

static int detect_dev(mav_ctx* c, const float* flow32, const double* flow64, const DerotParams* derot, const uint32_t* samples,
                      const uint8_t* sky, int batch, const mav_foe_params* fp, const mav_thr_params* tp, const double* foe_in,
                      double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, unsigned long long* max_phi_bits, double* foe_out,
                      mav_result* results, int32_t* box_out)
{
    const int W = c->W, H = c->H;
    const double* foe = foe_in;
    const bool want_phi = tp || phi || mask_fixed || mask_dyn || results || box_out || max_phi_bits;
    if (samples) {
        mav_foe_params f;
        if (fp) f = *fp; else mav_foe_defaults(&f);
        if (f.n_pairs < 1 || f.n_pairs > 4096) return fail(MAV_ERR_ARG, "n_pairs %d outside [1, 4096]", f.n_pairs);
        CHK(ensure_foe_scratch(c, f.n_pairs));
        double* fo = foe_out ? foe_out : c->foe_dev;
        ProfScope ps(c, K_FOE);
        // the candidates kernel also initialises the pair's box accumulators and tickets, the vote's last workgroup writes the FoE:
        // two launches where round 2 had four (candidates, vote, FoE finalize, box init)
        int32_t* init_box = want_phi ? c->box_acc : nullptr;
        if (flow32)
            launch_foe_f32(c->stream, flow32, derot, samples, batch, W, H, f.n_pairs, sq_threshold(f.mag_threshold),
                           sq_threshold_f32(f.mag_threshold), sq_threshold(f.ransac_threshold), c->foe_sc, fo, init_box, max_phi_bits);
        else
            launch_foe_f64(c->stream, flow64, samples, batch, W, H, f.n_pairs, sq_threshold(f.mag_threshold),
                           sq_threshold(f.ransac_threshold), c->foe_sc, fo, init_box, max_phi_bits);
        foe = fo;
    }
    if (want_phi) {
        if (!foe) return fail(MAV_ERR_ARG, "phi/mask stage needs a FoE (samples or foe)");
        mav_thr_params t;
        if (tp) t = *tp; else mav_thr_defaults(&t);
        if (!samples) { ProfScope ps(c, K_MISC); launch_box_init(c->stream, c->box_acc, max_phi_bits, c->foe_sc.done, batch); }
        // the pair's record (box, FoE) is written by the phi kernel's last workgroup of the pair: no finalize launch
        PhiLaunch pl;
        pl.done = c->foe_sc.done; pl.results = results; pl.box_out = box_out; pl.screen = c->phi_screen; pl.yloop = c->phi_yloop;
        ProfScope ps(c, K_PHI);
        if (flow32)
            launch_phi_mask_f32(c->stream, flow32, derot, foe, sky, batch, W, H, t, phi, mask_fixed, mask_dyn, c->box_acc, max_phi_bits, pl);
        else
            launch_phi_mask_f64(c->stream, flow64, foe, sky, batch, W, H, t, phi, mask_fixed, mask_dyn, c->box_acc, max_phi_bits, pl);
    }
    return check_launch("detection kernels");
}

extern "C" int mav_detect_dev(mav_ctx* c, const float* flow, const uint32_t* samples, const double* omega, const double* dt,
                              const uint8_t* frame0, const uint8_t* sky, int batch, const mav_foe_params* fp,
                              const mav_thr_params* tp, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, mav_result* results)
{
    if (!c || !flow || !samples || !results) return fail(MAV_ERR_ARG, "mav_detect: NULL argument");
    if (batch < 1 || batch > c->max_batch) return fail(MAV_ERR_ARG, "batch %d outside [1, %d]", batch, c->max_batch);
    HIPCHK(hipSetDevice(c->device));
    const DerotParams* derot = nullptr;
    CHK(upload_derot(c, omega, dt, frame0, batch, false, &derot));
    mav_thr_params t;
    if (tp) t = *tp; else mav_thr_defaults(&t);
    return detect_dev(c, flow, nullptr, derot, samples, sky, batch, fp, &t, nullptr, phi, mask_fixed, mask_dyn, nullptr, nullptr,
                      results, nullptr);
}

extern "C" int mav_process_batch_dev(mav_ctx* c, const uint8_t* prev, const uint8_t* next, const uint32_t* samples,
                                     const double* omega, const double* dt, const uint8_t* frame0, const uint8_t* sky, int batch,
                                     const mav_foe_params* fp, const mav_thr_params* tp, float* flow, double* phi,
                                     uint8_t* mask_fixed, uint8_t* mask_dyn, mav_result* results)
{
    if (!c || !prev || !next || !samples || !results) return fail(MAV_ERR_ARG, "mav_process_batch: NULL argument");
    if (batch < 1 || batch > c->max_batch) return fail(MAV_ERR_ARG, "batch %d outside [1, %d]", batch, c->max_batch);
    HIPCHK(hipSetDevice(c->device));
    if (!flow) {
        if (!c->flow_ws) HIPCHK(hipMalloc(&c->flow_ws, sizeof(float) * 2 * c->n0 * c->max_batch));
        flow = c->flow_ws;
    }
    CHK(mav_farneback_dev(c, prev, next, batch, flow));
    return mav_detect_dev(c, flow, samples, omega, dt, frame0, sky, batch, fp, tp, phi, mask_fixed, mask_dyn, results);
}

// ---- one iteration of the reference's loop as one call (include/mavflow.h: mav_frame_step) ---------------------------------------
static int frame_step_enqueue(mav_ctx* c, const mav_frame_step* s)
{
    if (s->n < 1 || s->n > c->max_batch) return fail(MAV_ERR_ARG, "mav_frame_step: n %d outside [1, %d]", s->n, c->max_batch);
    if (s->n_gather < 0 || s->n_gather > MAV_STEP_MAX_GATHER) return fail(MAV_ERR_ARG, "mav_frame_step: n_gather %d outside [0, %d]", s->n_gather, (int)MAV_STEP_MAX_GATHER);
    if (s->n_wait_before < 0 || s->n_record_after_flow < 0 || (s->n_wait_before && !s->wait_before) || (s->n_record_after_flow && !s->record_after_flow))
        return fail(MAV_ERR_ARG, "mav_frame_step: marker list without markers");
    if (s->compute_flow && (!s->prev_dev || !s->next_dev)) return fail(MAV_ERR_ARG, "mav_frame_step: compute_flow needs prev_dev and next_dev");
    if (s->detect && ((!s->flow_dev && !s->compute_flow) || !s->par_dev || !s->out_dev)) return fail(MAV_ERR_ARG, "mav_frame_step: detect needs a flow, par_dev and out_dev");
    if (s->detect && s->gt_dev && (!s->mask_fixed_dev || !s->mask_dyn_dev)) return fail(MAV_ERR_ARG, "mav_frame_step: the counts need both masks");
    if (s->n_bgr && (!s->bgr_dev || !s->gray_dev)) return fail(MAV_ERR_ARG, "mav_frame_step: n_bgr without bgr_dev / gray_dev");
    if (s->out_bytes && (!s->out_host || !s->out_dev)) return fail(MAV_ERR_ARG, "mav_frame_step: out_bytes without out_host / out_dev");
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i < s->n_wait_before; i++)
        if (s->wait_before[i]) HIPCHK(hipEventSynchronize((hipEvent_t)s->wait_before[i]));
    if (s->par_bytes) {
        if (!s->par_host || !s->par_dev) return fail(MAV_ERR_ARG, "mav_frame_step: par_bytes without par_host / par_dev");
        CHK(mav_upload_async_unordered(c, s->par_dev, s->par_host, s->par_bytes));
    }
    for (int i = 0; i < s->n_gather; i++)
        CHK(mav_upload_gather(c, s->gather[i].dst_dev, s->gather[i].src_host, s->gather[i].count, s->gather[i].bytes_each, MAV_GATHER_SOURCES_HELD));
    CHK(mav_upload_fence(c));
    if (s->n_bgr) CHK(mav_bgr2gray_dev(c, s->bgr_dev, s->n_bgr, s->gray_dev));
    float* flow = s->flow_dev;
    if (s->compute_flow) {
        if (!flow) {                             // nobody wants to see the flow: the context's own buffer, as mav_process_batch_dev
            if (!c->flow_ws) HIPCHK(hipMalloc(&c->flow_ws, sizeof(float) * 2 * c->n0 * c->max_batch));
            flow = c->flow_ws;
        }
        CHK(mav_farneback_dev(c, s->prev_dev, s->next_dev, s->n, flow));
    }
    for (int i = 0; i < s->n_record_after_flow; i++)
        if (s->record_after_flow[i]) HIPCHK(hipEventRecord((hipEvent_t)s->record_after_flow[i], c->stream));
    if (s->detect) {
        const char* par = (const char*)s->par_dev;
        CHK(mav_detect_dev(c, flow, (const uint32_t*)(par + s->off_samples), s->has_omega ? (const double*)(par + s->off_omega) : nullptr,
                           s->has_omega ? (const double*)(par + s->off_dt) : nullptr, s->has_frame0 ? (const uint8_t*)(par + s->off_frame0) : nullptr,
                           s->sky_dev, s->n, &s->foe, &s->thr, nullptr, s->mask_fixed_dev, s->mask_dyn_dev, (mav_result*)s->out_dev));
        if (s->gt_dev)
            CHK(mav_tpr_fpr_counts_dev(c, s->gt_dev, s->gt_images, s->mask_fixed_dev, s->mask_dyn_dev, 255, s->n,
                                       (int64_t*)((char*)s->out_dev + s->off_counts_fixed), (int64_t*)((char*)s->out_dev + s->off_counts_dyn)));
    }
    if (s->out_bytes) CHK(mav_download_async(c, s->out_host, s->out_dev, s->out_bytes));
    if (s->record_done) HIPCHK(hipEventRecord((hipEvent_t)s->record_done, c->stream));
    return MAV_OK;
}

extern "C" int mav_frame_step_dev(mav_ctx* c, const mav_frame_step* s)
{
    if (!c || !s) return fail(MAV_ERR_ARG, "mav_frame_step_dev: NULL argument");
    CHK(mav_worker_drain(c));
    return frame_step_enqueue(c, s);
}

// The worker: one thread per context that has been posted to.  A context is single-threaded; while steps are posted the worker IS
// that thread (the poster touches nothing of the context but this queue).  Three lanes = three workers enqueueing side by side
// while the loop's own thread draws samples and fills in FrameResults.
struct StepJob {
    mav_frame_step s;
    std::vector<const void*> src[MAV_STEP_MAX_GATHER];
    std::vector<void*> wait_before, record_after;
    uint64_t ticket = 0;
};
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::deque<StepJob> q;
    uint64_t posted = 0, done = 0;
    bool stop = false;
    int first_rc = MAV_OK;                                     // first failure since the last drain
    std::string first_err;
    std::map<uint64_t, std::pair<int, std::string>> failed;   // by ticket, until somebody waits for it
};
static void worker_main(mav_ctx* c)
{
    Worker* w = c->worker;
    (void)hipSetDevice(c->device);
    std::unique_lock<std::mutex> lk(w->m);
    for (;;) {
        w->cv_job.wait(lk, [&] { return w->stop || !w->q.empty(); });
        if (w->stop) {
            // mav_destroy: steps nobody waited for are DROPPED, not enqueued -- the host buffers they point to may be gone with
            // whoever posted them; their tickets report MAV_ERR_STATE to a waiter that is still there
            for (StepJob& j : w->q) w->failed[j.ticket] = {MAV_ERR_STATE, "the context was destroyed before this step was enqueued"};
            if (!w->q.empty()) w->done = w->q.back().ticket;
            w->q.clear();
            w->cv_done.notify_all();
            return;
        }
        StepJob job = std::move(w->q.front());
        w->q.pop_front();
        lk.unlock();
        const int rc = frame_step_enqueue(c, &job.s);
        lk.lock();
        if (rc != MAV_OK) {
            w->failed[job.ticket] = {rc, g_err};
            if (w->first_rc == MAV_OK) { w->first_rc = rc; w->first_err = g_err; }
        }
        w->done = job.ticket;
        w->cv_done.notify_all();
    }
}
static void stop_worker(mav_ctx* c)
{
    if (!c->worker) return;
    { std::lock_guard<std::mutex> lk(c->worker->m); c->worker->stop = true; }
    c->worker->cv_job.notify_all();
    c->worker->th.join();
    delete c->worker;
    c->worker = nullptr;
}
extern "C" int mav_frame_step_post(mav_ctx* c, const mav_frame_step* s, uint64_t* ticket)
{
    if (!c || !s || !ticket) return fail(MAV_ERR_ARG, "mav_frame_step_post: NULL argument");
    if (s->n_gather < 0 || s->n_gather > MAV_STEP_MAX_GATHER || s->n_wait_before < 0 || s->n_record_after_flow < 0)
        return fail(MAV_ERR_ARG, "mav_frame_step_post: list length out of range");
    if (!c->worker) {
        c->worker = new Worker();
        c->worker->th = std::thread(worker_main, c);
    }
    Worker* w = c->worker;
    StepJob job;
    job.s = *s;
    for (int i = 0; i < s->n_gather; i++) {
        if (s->gather[i].count < 1 || !s->gather[i].src_host) return fail(MAV_ERR_ARG, "mav_frame_step_post: gather list %d is empty", i);
        job.src[i].assign(s->gather[i].src_host, s->gather[i].src_host + s->gather[i].count);
        job.s.gather[i].src_host = job.src[i].data();          // (heap blocks: they keep their address when the job moves)
    }
    if (s->n_wait_before) { job.wait_before.assign(s->wait_before, s->wait_before + s->n_wait_before); job.s.wait_before = job.wait_before.data(); }
    if (s->n_record_after_flow) {
        job.record_after.assign(s->record_after_flow, s->record_after_flow + s->n_record_after_flow);
        job.s.record_after_flow = job.record_after.data();
    }
    {
        std::lock_guard<std::mutex> lk(w->m);
        job.ticket = *ticket = ++w->posted;
        w->q.push_back(std::move(job));
    }
    w->cv_job.notify_one();
    return MAV_OK;
}
extern "C" int mav_frame_step_wait(mav_ctx* c, uint64_t ticket, void* marker)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_frame_step_wait: NULL context");
    Worker* w = c->worker;
    if (!w || ticket == 0) return fail(MAV_ERR_ARG, "mav_frame_step_wait: no such ticket");
    {
        std::unique_lock<std::mutex> lk(w->m);
        if (ticket > w->posted) return fail(MAV_ERR_ARG, "mav_frame_step_wait: ticket %llu was never posted", (unsigned long long)ticket);
        w->cv_done.wait(lk, [&] { return w->done >= ticket; });
        auto it = w->failed.find(ticket);
        if (it != w->failed.end()) {
            const int rc = it->second.first;
            g_err = it->second.second;
            w->failed.erase(it);
            return rc;
        }
    }
    if (marker) HIPCHK(hipEventSynchronize((hipEvent_t)marker));
    return MAV_OK;
}
extern "C" int mav_worker_drain(mav_ctx* c)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_worker_drain: NULL context");
    Worker* w = c->worker;
    if (!w) return MAV_OK;
    std::unique_lock<std::mutex> lk(w->m);
    w->cv_done.wait(lk, [&] { return w->done == w->posted; });
    const int rc = w->first_rc;
    if (rc != MAV_OK) g_err = w->first_err;
    w->first_rc = MAV_OK;
    return rc;
}

// ---- host-pointer wrappers -----------------------------------------------------------------------------------
// A staging buffer of one host-pointer call.  It borrows the context's next scratch block (grown when too small, never
// shrunk or freed before mav_destroy): once a call shape has been seen the staged path allocates nothing.  The host entry
// points are synchronous (they end in mav_sync), so a block is idle again when the next call takes it.
struct DevBuf {
    void* p = nullptr;
    int alloc(mav_ctx* c, size_t bytes)
    {
        if (c->scratch_next == c->scratch.size()) c->scratch.emplace_back();
        mav_ctx::Block& b = c->scratch[c->scratch_next++];
        if (b.cap < bytes || !b.p) {
            if (b.p) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(b.p); b.p = nullptr; b.cap = 0; }
            const size_t want = bytes ? bytes : 1;
            HIPCHK(hipMalloc(&b.p, want));
            b.cap = want;
        }
        p = b.p;
        return MAV_OK;
    }
    int upload(mav_ctx* c, const void* src, size_t bytes)
    {
        CHK(alloc(c, bytes));
        HIPCHK(hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, c->stream));
        return MAV_OK;
    }
    template <typename T> T* as() { return (T*)p; }
};
// The two frame batches of a host-pointer call -> device.  When the caller's batches are views of one run of batch + 1 frames
// (next == prev + one frame) the run crosses PCIe once and keeps that layout on the device, which mav_farneback_dev recognises.
static int upload_frames(mav_ctx* c, const uint8_t* prev, const uint8_t* next, int batch, DevBuf& dp, DevBuf& dn, const uint8_t** dprev,
                         const uint8_t** dnext)
{
    const size_t n = c->n0 * batch;
    if (next == prev + c->n0) {
        CHK(dp.upload(c, prev, n + c->n0));
        CHK(dn.alloc(c, 1));                              // keeps the staging slots of the two call forms aligned
        *dprev = dp.as<uint8_t>(); *dnext = dp.as<uint8_t>() + c->n0;
        return MAV_OK;
    }
    CHK(dp.upload(c, prev, n)); CHK(dn.upload(c, next, n));
    *dprev = dp.as<uint8_t>(); *dnext = dn.as<uint8_t>();
    return MAV_OK;
}
static int download(mav_ctx* c, void* dst, const void* src, size_t bytes)
{
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return MAV_OK;
}
static int check_batch(mav_ctx* c, int batch, const char* fn)
{
    if (!c) return fail(MAV_ERR_ARG, "%s: NULL context", fn);
    if (batch < 1 || batch > c->max_batch) return fail(MAV_ERR_ARG, "%s: batch %d outside [1, %d]", fn, batch, c->max_batch);
    HIPCHK(hipSetDevice(c->device));
    c->scratch_next = 0;              // a new host-pointer call: its staging buffers start again at block 0
    c->last_mf = c->last_md = nullptr; c->last_mask_batch = 0;     // ... and may overwrite the previous call's masks
    return MAV_OK;
}

extern "C" int mav_farneback(mav_ctx* c, const uint8_t* prev, const uint8_t* next, int batch, float* flow)
{
    CHK(check_batch(c, batch, "mav_farneback"));
    if (!prev || !next || !flow) return fail(MAV_ERR_ARG, "mav_farneback: NULL argument");
    const size_t n = c->n0 * batch;
    DevBuf dp, dn, df;
    const uint8_t *dprev, *dnext;
    CHK(upload_frames(c, prev, next, batch, dp, dn, &dprev, &dnext)); CHK(df.alloc(c, n * 2 * sizeof(float)));
    CHK(mav_farneback_dev(c, dprev, dnext, batch, df.as<float>()));
    CHK(download(c, flow, df.p, n * 2 * sizeof(float)));
    return mav_sync(c);
}

extern "C" int mav_derotate(mav_ctx* c, const float* flow, const double* omega, const double* dt, int batch, double* flow_out)
{
    CHK(check_batch(c, batch, "mav_derotate"));
    if (!flow || !omega || !flow_out) return fail(MAV_ERR_ARG, "mav_derotate: NULL argument");
    const size_t n = c->n0 * batch * 2;
    DevBuf df, dout;
    CHK(df.upload(c, flow, n * sizeof(float))); CHK(dout.alloc(c, n * sizeof(double)));
    const DerotParams* derot = nullptr;
    CHK(upload_derot(c, omega, dt, nullptr, batch, true, &derot));
    launch_derotate(c->stream, df.as<float>(), derot, batch, c->W, c->H, dout.as<double>());
    CHK(check_launch("derotate"));
    CHK(download(c, flow_out, dout.p, n * sizeof(double)));
    return mav_sync(c);
}

extern "C" int mav_foe_dense(mav_ctx* c, const double* flow, const uint32_t* samples, int batch, const mav_foe_params* fp, double* foe)
{
    CHK(check_batch(c, batch, "mav_foe_dense"));
    if (!flow || !samples || !foe) return fail(MAV_ERR_ARG, "mav_foe_dense: NULL argument");
    mav_foe_params f;
    if (fp) f = *fp; else mav_foe_defaults(&f);
    if (f.n_pairs < 1 || f.n_pairs > 4096) return fail(MAV_ERR_ARG, "n_pairs %d outside [1, 4096]", f.n_pairs);
    DevBuf df, ds, dfoe;
    CHK(df.upload(c, flow, c->n0 * batch * 2 * sizeof(double)));
    CHK(ds.upload(c, samples, sizeof(uint32_t) * 4 * (size_t)f.n_pairs * batch));
    CHK(dfoe.alloc(c, sizeof(double) * 2 * batch));
    CHK(detect_dev(c, nullptr, df.as<double>(), nullptr, ds.as<uint32_t>(), nullptr, batch, &f, nullptr, nullptr, nullptr, nullptr,
                   nullptr, nullptr, dfoe.as<double>(), nullptr, nullptr));
    CHK(download(c, foe, dfoe.p, sizeof(double) * 2 * batch));
    return mav_sync(c);
}

extern "C" int mav_phi_mask(mav_ctx* c, const double* flow, const double* foe, const uint8_t* sky, int batch, const mav_thr_params* tp,
                            double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, double* max_phi)
{
    CHK(check_batch(c, batch, "mav_phi_mask"));
    if (!flow || !foe) return fail(MAV_ERR_ARG, "mav_phi_mask: NULL argument");
    const size_t n = c->n0 * batch;
    mav_thr_params t;
    if (tp) t = *tp; else mav_thr_defaults(&t);
    DevBuf df, dfoe, dsky, dphi, dmf, dmd;
    CHK(df.upload(c, flow, n * 2 * sizeof(double)));
    CHK(dfoe.upload(c, foe, sizeof(double) * 2 * batch));
    if (sky) CHK(dsky.upload(c, sky, n));
    if (phi) CHK(dphi.alloc(c, n * sizeof(double)));
    if (mask_fixed) CHK(dmf.alloc(c, n));
    if (mask_dyn) CHK(dmd.alloc(c, n));
    CHK(detect_dev(c, nullptr, df.as<double>(), nullptr, nullptr, dsky.as<uint8_t>(), batch, nullptr, &t, dfoe.as<double>(),
                   dphi.as<double>(), dmf.as<uint8_t>(), dmd.as<uint8_t>(), max_phi ? c->u64_scratch : nullptr, nullptr, nullptr,
                   nullptr));
    c->last_mf = dmf.as<uint8_t>(); c->last_md = dmd.as<uint8_t>(); c->last_mask_batch = batch;
    if (phi) CHK(download(c, phi, dphi.p, n * sizeof(double)));
    if (mask_fixed) CHK(download(c, mask_fixed, dmf.p, n));
    if (mask_dyn) CHK(download(c, mask_dyn, dmd.p, n));
    if (max_phi) CHK(download(c, max_phi, c->u64_scratch, sizeof(double) * batch));  // same bits
    return mav_sync(c);
}

// ---- the same two calls on a float32 flow array (the reference's frame index 0): float32 arithmetic ----------------
static int frame0_params(mav_ctx* c, int batch, const DerotParams** out)
{
    std::vector<uint8_t> all(batch, 1);
    return upload_derot(c, nullptr, nullptr, all.data(), batch, true, out);
}

extern "C" int mav_foe_dense_f32(mav_ctx* c, const float* flow, const uint32_t* samples, int batch, const mav_foe_params* fp, double* foe)
{
    CHK(check_batch(c, batch, "mav_foe_dense_f32"));
    if (!flow || !samples || !foe) return fail(MAV_ERR_ARG, "mav_foe_dense_f32: NULL argument");
    mav_foe_params f;
    if (fp) f = *fp; else mav_foe_defaults(&f);
    if (f.n_pairs < 1 || f.n_pairs > 4096) return fail(MAV_ERR_ARG, "n_pairs %d outside [1, 4096]", f.n_pairs);
    DevBuf df, ds, dfoe;
    CHK(df.upload(c, flow, c->n0 * batch * 2 * sizeof(float)));
    CHK(ds.upload(c, samples, sizeof(uint32_t) * 4 * (size_t)f.n_pairs * batch));
    CHK(dfoe.alloc(c, sizeof(double) * 2 * batch));
    const DerotParams* mode = nullptr;
    CHK(frame0_params(c, batch, &mode));
    CHK(detect_dev(c, df.as<float>(), nullptr, mode, ds.as<uint32_t>(), nullptr, batch, &f, nullptr, nullptr, nullptr, nullptr,
                   nullptr, nullptr, dfoe.as<double>(), nullptr, nullptr));
    CHK(download(c, foe, dfoe.p, sizeof(double) * 2 * batch));
    return mav_sync(c);
}

extern "C" int mav_phi_mask_f32(mav_ctx* c, const float* flow, const double* foe, const uint8_t* sky, int batch,
                                const mav_thr_params* tp, float* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, float* max_phi)
{
    CHK(check_batch(c, batch, "mav_phi_mask_f32"));
    if (!flow || !foe) return fail(MAV_ERR_ARG, "mav_phi_mask_f32: NULL argument");
    const size_t n = c->n0 * batch;
    mav_thr_params t;
    if (tp) t = *tp; else mav_thr_defaults(&t);
    DevBuf df, dfoe, dsky, dphi, dmf, dmd;
    CHK(df.upload(c, flow, n * 2 * sizeof(float)));
    CHK(dfoe.upload(c, foe, sizeof(double) * 2 * batch));
    if (sky) CHK(dsky.upload(c, sky, n));
    if (phi) CHK(dphi.alloc(c, n * sizeof(double)));
    if (mask_fixed) CHK(dmf.alloc(c, n));
    if (mask_dyn) CHK(dmd.alloc(c, n));
    const DerotParams* mode = nullptr;
    CHK(frame0_params(c, batch, &mode));
    CHK(detect_dev(c, df.as<float>(), nullptr, mode, nullptr, dsky.as<uint8_t>(), batch, nullptr, &t, dfoe.as<double>(),
                   dphi.as<double>(), dmf.as<uint8_t>(), dmd.as<uint8_t>(), max_phi ? c->u64_scratch : nullptr, nullptr, nullptr,
                   nullptr));
    c->last_mf = dmf.as<uint8_t>(); c->last_md = dmd.as<uint8_t>(); c->last_mask_batch = batch;
    // the kernel stores the float32 angles widened to double (one phi layout for both arithmetic types): narrow them back, exactly
    std::vector<double> wide;
    if (phi) { wide.resize(n); CHK(download(c, wide.data(), dphi.p, n * sizeof(double))); }
    if (mask_fixed) CHK(download(c, mask_fixed, dmf.p, n));
    if (mask_dyn) CHK(download(c, mask_dyn, dmd.p, n));
    std::vector<double> mx(max_phi ? batch : 0);
    if (max_phi) CHK(download(c, mx.data(), c->u64_scratch, sizeof(double) * batch));
    CHK(mav_sync(c));
    if (phi) for (size_t i = 0; i < n; i++) phi[i] = (float)wide[i];
    if (max_phi) for (int b = 0; b < batch; b++) max_phi[b] = (float)mx[b];
    return MAV_OK;
}

extern "C" int mav_bbox(mav_ctx* c, const uint8_t* img, int batch, int32_t* box)
{
    CHK(check_batch(c, batch, "mav_bbox"));
    if (!img || !box) return fail(MAV_ERR_ARG, "mav_bbox: NULL argument");
    DevBuf di, db;
    CHK(di.upload(c, img, c->n0 * batch)); CHK(db.alloc(c, sizeof(int32_t) * 4 * batch));
    launch_box_init(c->stream, c->box_acc, nullptr, nullptr, batch);
    launch_bbox_u8(c->stream, di.as<uint8_t>(), batch, c->W, c->H, c->i32_scratch, c->box_acc);
    launch_box_finalize(c->stream, c->box_acc, batch, db.as<int32_t>());
    CHK(check_launch("bbox"));
    CHK(download(c, box, db.p, sizeof(int32_t) * 4 * batch));
    return mav_sync(c);
}

extern "C" int mav_bgr2gray(mav_ctx* c, const uint8_t* bgr, int batch, uint8_t* gray)
{
    CHK(check_batch(c, batch, "mav_bgr2gray"));
    if (!bgr || !gray) return fail(MAV_ERR_ARG, "mav_bgr2gray: NULL argument");
    const size_t n = c->n0 * batch;
    DevBuf di, dg;
    CHK(di.upload(c, bgr, 3 * n)); CHK(dg.alloc(c, n));
    launch_bgr2gray(c->stream, di.as<uint8_t>(), n, dg.as<uint8_t>());
    CHK(check_launch("bgr2gray"));
    CHK(download(c, gray, dg.p, n));
    return mav_sync(c);
}

extern "C" int mav_bgr2gray_dev(mav_ctx* c, const uint8_t* bgr, int batch, uint8_t* gray)
{
    if (!c || !bgr || !gray) return fail(MAV_ERR_ARG, "mav_bgr2gray_dev: NULL argument");
    if (batch < 1) return fail(MAV_ERR_ARG, "mav_bgr2gray_dev: batch %d < 1", batch);
    HIPCHK(hipSetDevice(c->device));
    ProfScope ps(c, K_MISC);
    launch_bgr2gray(c->stream, bgr, c->n0 * (size_t)batch, gray);
    return check_launch("bgr2gray");
}

// ---- frame decode, host side: the un-filtering pass of a PNG image ------------------------------------------------------------------
// `raw` is the inflated IDAT stream of a non-interlaced image: per row one filter-type byte followed by `stride` bytes; bpp = bytes per
// complete pixel (1 for depths below 8).  Filters 0 - 4 of the PNG specification (None, Sub, Up, Average, Paeth), arithmetic modulo 256.
// Sub / Average / Paeth are recurrences along the row, Up / Average / Paeth along the column: inherently serial per image, a few
// milliseconds of plain C for a 1080p frame (a numpy formulation needs a Python-level loop per pixel for two of the five filters).
extern "C" int mav_png_unfilter(const uint8_t* raw, int rows, size_t stride, int bpp, uint8_t* out)
{
    if (!raw || !out || rows < 0 || bpp < 1 || bpp > 8) return fail(MAV_ERR_ARG, "mav_png_unfilter: bad argument");
    const uint8_t* prev = nullptr;
    for (int y = 0; y < rows; y++) {
        const uint8_t ft = raw[(size_t)y * (stride + 1)];
        const uint8_t* in = raw + (size_t)y * (stride + 1) + 1;
        uint8_t* o = out + (size_t)y * stride;
        const size_t b = (size_t)bpp;
        switch (ft) {
        case 0: memcpy(o, in, stride); break;
        case 1:
            for (size_t i = 0; i < stride; i++) o[i] = (uint8_t)(in[i] + (i >= b ? o[i - b] : 0));
            break;
        case 2:
            for (size_t i = 0; i < stride; i++) o[i] = (uint8_t)(in[i] + (prev ? prev[i] : 0));
            break;
        case 3:
            for (size_t i = 0; i < stride; i++) {
                const unsigned a = i >= b ? o[i - b] : 0, up = prev ? prev[i] : 0;
                o[i] = (uint8_t)(in[i] + ((a + up) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= b ? o[i - b] : 0, up = prev ? prev[i] : 0, ul = (prev && i >= b) ? prev[i - b] : 0;
                const int p = a + up - ul, pa = abs(p - a), pb = abs(p - up), pc = abs(p - ul);
                const int pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? up : ul);
                o[i] = (uint8_t)(in[i] + pred);
            }
            break;
        default: return fail(MAV_ERR_ARG, "mav_png_unfilter: row %d has filter type %d (valid: 0 - 4)", y, (int)ft);
        }
        prev = o;
    }
    return MAV_OK;
}

extern "C" int mav_ransac(mav_ctx* c, const double* estimates, int count, double ransac_threshold, double* foe)
{
    if (!c || !foe || (count > 0 && !estimates)) return fail(MAV_ERR_ARG, "mav_ransac: NULL argument");
    if (count < 0 || count > 4096) return fail(MAV_ERR_ARG, "mav_ransac: count %d outside [0, 4096]", count);
    HIPCHK(hipSetDevice(c->device));
    const int N = count > 0 ? count : 1;
    CHK(ensure_foe_scratch(c, N));
    if (count > 0) HIPCHK(hipMemcpyAsync(c->foe_sc.cand, estimates, sizeof(double) * 2 * count, hipMemcpyHostToDevice, c->stream));
    launch_ransac_only(c->stream, c->foe_sc, count, c->foe_sc_n, sq_threshold(ransac_threshold), c->foe_dev);
    CHK(check_launch("ransac"));
    CHK(download(c, foe, c->foe_dev, sizeof(double) * 2));
    return mav_sync(c);
}

extern "C" int mav_window_max(mav_ctx* c, const uint8_t* img, int batch, int64_t* out)
{
    CHK(check_batch(c, batch, "mav_window_max"));
    if (!img || !out) return fail(MAV_ERR_ARG, "mav_window_max: NULL argument");
    DevBuf di, dout;
    CHK(di.upload(c, img, c->n0 * batch)); CHK(dout.alloc(c, sizeof(int64_t) * 3 * batch));
    launch_window_max(c->stream, di.as<uint8_t>(), batch, c->W, c->H, c->u64_scratch, dout.as<int64_t>());
    CHK(check_launch("window_max"));
    CHK(download(c, out, dout.p, sizeof(int64_t) * 3 * batch));
    return mav_sync(c);
}

// ---- window search: analyze_pyramid / optimize_window -----------------------------------------------------------
// Level sizes exactly as the reference derives them (im_helpers.py:28-33 + imutils.resize): w' = int(w / scale),
// r = w' / float(w), h' = int(h * r); stop when a side drops below 30.
static int pyr_plan(const mav_ctx* c, double scale, int batch_cap, PyrPlan* p)
{
    if (!(scale > 1.0)) return fail(MAV_ERR_ARG, "pyramid scale %g must be > 1", scale);
    int w = c->W, h = c->H;
    size_t off = 0;
    unsigned base = 0;
    p->n = 0;
    for (;;) {
        if (p->n == MAV_PYR_MAX) return fail(MAV_ERR_ARG, "pyramid scale %g gives more than %d levels", scale, MAV_PYR_MAX);
        const int n = p->n++;
        p->w[n] = w; p->h[n] = h; p->base[n] = base; p->off[n] = off;
        if (w >= 64 && h >= 64) base += (unsigned)(((w - 64) / 16 + 1) * ((h - 64) / 16 + 1));
        if (n > 0) off += ((size_t)w * h * batch_cap + 255) & ~(size_t)255;
        const int wn = (int)((double)w / scale);
        if (wn < 1) break;
        const double r = (double)wn / (double)w;
        const int hn = (int)((double)h * r);
        if (hn < 30 || wn < 30) break;
        // cv2.resize takes its integer-ratio "fast area" path when both ratios are whole numbers; only the general path is restated
        const double sx = 1.0 / ((double)wn / w), sy = 1.0 / ((double)hn / h);
        if (fabs(sx - (int)sx) < DBL_EPSILON && fabs(sy - (int)sy) < DBL_EPSILON)
            return fail(MAV_ERR_ARG, "pyramid level %d -> %d has an integer ratio (%dx%d -> %dx%d): OpenCV's fast-area path is not implemented",
                        n, n + 1, w, h, wn, hn);
        w = wn; h = hn;
    }
    p->base[p->n] = base;
    return MAV_OK;
}
static size_t pyr_bytes(const PyrPlan& p, int batch_cap)
{
    size_t total = 0;
    for (int l = 1; l < p.n; l++) total += ((size_t)p.w[l] * p.h[l] * batch_cap + 255) & ~(size_t)255;
    return total ? total : 256;
}
static int ensure_pyr_ws(mav_ctx* c, const PyrPlan& p)
{
    const size_t need = pyr_bytes(p, c->max_batch);
    if (need <= c->pyr_ws_bytes) return MAV_OK;
    if (c->pyr_ws) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->pyr_ws); }
    c->pyr_ws = nullptr; c->pyr_ws_bytes = 0;
    if (hipMalloc(&c->pyr_ws, need) != hipSuccess) return fail(MAV_ERR_OOM, "pyramid workspace (%zu bytes)", need);
    c->pyr_ws_bytes = need;
    return MAV_OK;
}
// levels 1 .. upto of `batch` images (device pointer img0) into the workspace
static void build_pyramid(mav_ctx* c, const PyrPlan& p, const uint8_t* img0, int batch, int upto)
{
    for (int l = 1; l <= upto && l < p.n; l++) {
        const uint8_t* src = l == 1 ? img0 : c->pyr_ws + p.off[l - 1];
        launch_area_resize(c->stream, src, (size_t)p.w[l - 1] * p.h[l - 1], p.w[l - 1], p.h[l - 1], c->pyr_ws + p.off[l],
                           (size_t)p.w[l] * p.h[l], p.w[l], p.h[l], batch);
    }
}

extern "C" int mav_pyramid_levels(const mav_ctx* c, double scale)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_pyramid_levels: NULL context");
    PyrPlan p;
    CHK(pyr_plan(c, scale, 1, &p));
    return p.n;
}
extern "C" int mav_pyramid_dims(const mav_ctx* c, double scale, int level, int* w, int* h)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_pyramid_dims: NULL context");
    PyrPlan p;
    CHK(pyr_plan(c, scale, 1, &p));
    if (level < 0 || level >= p.n) return fail(MAV_ERR_ARG, "pyramid level %d outside [0, %d)", level, p.n);
    if (w) *w = p.w[level];
    if (h) *h = p.h[level];
    return MAV_OK;
}

extern "C" int mav_analyze_pyramid(mav_ctx* c, const uint8_t* img, int batch, double scale, int64_t* out)
{
    CHK(check_batch(c, batch, "mav_analyze_pyramid"));
    if (!img || !out) return fail(MAV_ERR_ARG, "mav_analyze_pyramid: NULL argument");
    PyrPlan p;
    CHK(pyr_plan(c, scale, c->max_batch, &p));
    CHK(ensure_pyr_ws(c, p));
    DevBuf di, dout;
    CHK(di.upload(c, img, c->n0 * batch)); CHK(dout.alloc(c, sizeof(int64_t) * 6 * batch));
    HIPCHK(hipMemsetAsync(c->u64_scratch, 0, sizeof(unsigned long long) * batch, c->stream));
    build_pyramid(c, p, di.as<uint8_t>(), batch, p.n - 1);
    for (int l = 0; l < p.n; l++)
        launch_level_scan(c->stream, l == 0 ? di.as<uint8_t>() : c->pyr_ws + p.off[l], (size_t)p.w[l] * p.h[l], batch, p.w[l], p.h[l],
                          p.base[l], c->u64_scratch);
    launch_pyramid_finalize(c->stream, c->u64_scratch, p, di.as<uint8_t>(), c->pyr_ws, batch, dout.as<int64_t>());
    CHK(check_launch("analyze_pyramid"));
    CHK(download(c, out, dout.p, sizeof(int64_t) * 6 * batch));
    return mav_sync(c);
}

extern "C" int mav_stage_pyramid_level(mav_ctx* c, const uint8_t* img, double scale, int level, uint8_t* out)
{
    CHK(check_batch(c, 1, "mav_stage_pyramid_level"));
    if (!img || !out) return fail(MAV_ERR_ARG, "mav_stage_pyramid_level: NULL argument");
    PyrPlan p;
    CHK(pyr_plan(c, scale, c->max_batch, &p));
    if (level < 0 || level >= p.n) return fail(MAV_ERR_ARG, "pyramid level %d outside [0, %d)", level, p.n);
    CHK(ensure_pyr_ws(c, p));
    DevBuf di;
    CHK(di.upload(c, img, c->n0));
    build_pyramid(c, p, di.as<uint8_t>(), 1, level);
    CHK(check_launch("area_resize"));
    CHK(download(c, out, level == 0 ? di.p : (void*)(c->pyr_ws + p.off[level]), (size_t)p.w[level] * p.h[level]));
    return mav_sync(c);
}

extern "C" int mav_optimize_window(mav_ctx* c, const uint8_t* img, int batch, const int32_t* window_in, int64_t* score,
                                   int32_t* window_out)
{
    CHK(check_batch(c, batch, "mav_optimize_window"));
    if (!img || !window_in || !score || !window_out) return fail(MAV_ERR_ARG, "mav_optimize_window: NULL argument");
    if (!c->sat && hipMalloc(&c->sat, sizeof(unsigned long long) * (size_t)(c->W + 1) * (c->H + 1) * c->max_batch) != hipSuccess)
        return fail(MAV_ERR_OOM, "summed-area tables");
    DevBuf di, dw, ds, dwo;
    CHK(di.upload(c, img, c->n0 * batch)); CHK(dw.upload(c, window_in, sizeof(int32_t) * 4 * batch));
    CHK(ds.alloc(c, sizeof(int64_t) * batch)); CHK(dwo.alloc(c, sizeof(int32_t) * 4 * batch));
    launch_optimize_window(c->stream, di.as<uint8_t>(), batch, c->W, c->H, c->sat, dw.as<int32_t>(), ds.as<int64_t>(), dwo.as<int32_t>());
    CHK(check_launch("optimize_window"));
    CHK(download(c, score, ds.p, sizeof(int64_t) * batch));
    CHK(download(c, window_out, dwo.p, sizeof(int32_t) * 4 * batch));
    return mav_sync(c);
}

extern "C" int mav_tpr_fpr_counts(mav_ctx* c, const uint8_t* gt, const uint8_t* mask, int mask_value, int batch, int64_t* counts)
{
    CHK(check_batch(c, batch, "mav_tpr_fpr_counts"));
    if (!gt || !mask || !counts) return fail(MAV_ERR_ARG, "mav_tpr_fpr_counts: NULL argument");
    if (mask_value < 1 || mask_value > 65535) return fail(MAV_ERR_ARG, "mav_tpr_fpr_counts: mask_value %d outside [1, 65535]", mask_value);
    DevBuf dg, dm;
    CHK(dg.upload(c, gt, c->n0 * batch)); CHK(dm.upload(c, mask, c->n0 * batch));
    launch_tpr_fpr(c->stream, dg.as<uint8_t>(), dm.as<uint8_t>(), (unsigned)mask_value, batch, c->W, c->H, c->u64_scratch);
    CHK(check_launch("tpr_fpr"));
    CHK(download(c, counts, c->u64_scratch, sizeof(int64_t) * 4 * batch));
    return mav_sync(c);
}

// frames (mav_process_batch) or a float32 flow field (mav_detect) in, masks and records out: one body for both
static int process_host(mav_ctx* c, const char* fn, const uint8_t* prev, const uint8_t* next, const float* flow_in,
                        const uint32_t* samples, const double* omega, const double* dt, const uint8_t* frame0, const uint8_t* sky,
                        int batch, const mav_foe_params* fp, const mav_thr_params* tp, float* flow_out, double* phi,
                        uint8_t* mask_fixed, uint8_t* mask_dyn, mav_result* results)
{
    CHK(check_batch(c, batch, fn));
    if ((!flow_in && (!prev || !next)) || !samples || !results) return fail(MAV_ERR_ARG, "%s: NULL argument", fn);
    mav_foe_params f;
    if (fp) f = *fp; else mav_foe_defaults(&f);
    if (f.n_pairs < 1 || f.n_pairs > 4096) return fail(MAV_ERR_ARG, "n_pairs %d outside [1, 4096]", f.n_pairs);
    const size_t n = c->n0 * batch;
    DevBuf dp, dn, ds, dflow, dres, dmf, dmd, dsky, dom, ddt, df0, dphi;      // the always-present buffers take the first blocks
    const uint8_t *dprev = nullptr, *dnext = nullptr;
    if (flow_in) CHK(dflow.upload(c, flow_in, n * 2 * sizeof(float)));
    else { CHK(upload_frames(c, prev, next, batch, dp, dn, &dprev, &dnext)); CHK(dflow.alloc(c, n * 2 * sizeof(float))); }
    CHK(ds.upload(c, samples, sizeof(uint32_t) * 4 * (size_t)f.n_pairs * batch));
    CHK(dres.alloc(c, sizeof(mav_result) * batch));
    if (mask_fixed) CHK(dmf.alloc(c, n));
    if (mask_dyn) CHK(dmd.alloc(c, n));
    if (sky) CHK(dsky.upload(c, sky, n));
    if (omega) CHK(dom.upload(c, omega, sizeof(double) * 3 * batch));
    if (omega && dt) CHK(ddt.upload(c, dt, sizeof(double) * batch));
    if (frame0) CHK(df0.upload(c, frame0, batch));
    if (phi) CHK(dphi.alloc(c, n * sizeof(double)));
    if (!flow_in) CHK(mav_farneback_dev(c, dprev, dnext, batch, dflow.as<float>()));
    CHK(mav_detect_dev(c, dflow.as<float>(), ds.as<uint32_t>(), dom.as<double>(), ddt.as<double>(), df0.as<uint8_t>(),
                       dsky.as<uint8_t>(), batch, &f, tp, dphi.as<double>(), dmf.as<uint8_t>(), dmd.as<uint8_t>(), dres.as<mav_result>()));
    c->last_mf = dmf.as<uint8_t>(); c->last_md = dmd.as<uint8_t>(); c->last_mask_batch = batch;
    if (flow_out) CHK(download(c, flow_out, dflow.p, n * 2 * sizeof(float)));
    if (phi) CHK(download(c, phi, dphi.p, n * sizeof(double)));
    if (mask_fixed) CHK(download(c, mask_fixed, dmf.p, n));
    if (mask_dyn) CHK(download(c, mask_dyn, dmd.p, n));
    CHK(download(c, results, dres.p, sizeof(mav_result) * batch));
    return mav_sync(c);
}

extern "C" int mav_last_masks_tpr_fpr(mav_ctx* c, const uint8_t* gt, int mask_value, int batch, int64_t* counts_fixed, int64_t* counts_dyn)
{
    if (!c || !gt) return fail(MAV_ERR_ARG, "mav_last_masks_tpr_fpr: NULL argument");
    if (mask_value < 1 || mask_value > 65535) return fail(MAV_ERR_ARG, "mav_last_masks_tpr_fpr: mask_value %d outside [1, 65535]", mask_value);
    if (!c->last_mask_batch || batch != c->last_mask_batch || (counts_fixed && !c->last_mf) || (counts_dyn && !c->last_md))
        return fail(MAV_ERR_STATE, "mav_last_masks_tpr_fpr: no masks of a %d-pair detection call are resident", batch);
    HIPCHK(hipSetDevice(c->device));
    // the ground truth goes into the NEXT free staging block: the previous call's blocks (its masks among them) stay untouched
    const uint8_t *mf = c->last_mf, *md = c->last_md;
    const size_t mark = c->scratch_next;
    DevBuf dg;
    CHK(dg.upload(c, gt, c->n0 * batch));
    c->scratch_next = mark;          // the call is synchronous: its block is free again on return, a repeated call re-uses it
    // one pass over the ground truth for both masks
    unsigned long long *c0 = c->u64_scratch, *c1 = c->u64_scratch + 4 * (size_t)batch;
    const uint8_t* m0 = counts_fixed ? mf : md;
    const uint8_t* m1 = (counts_fixed && counts_dyn) ? md : nullptr;
    if (!counts_fixed && !counts_dyn) return MAV_OK;
    launch_tpr_fpr2(c->stream, dg.as<uint8_t>(), c->n0, m0, m1, (unsigned)mask_value, batch, c->W, c->H, c0, m1 ? c1 : nullptr);
    CHK(check_launch("tpr_fpr"));
    CHK(download(c, counts_fixed ? counts_fixed : counts_dyn, c0, sizeof(int64_t) * 4 * batch));
    if (m1) CHK(download(c, counts_dyn, c1, sizeof(int64_t) * 4 * batch));
    return mav_sync(c);
}

// calculate_tpr_fpr of one or two device-resident masks against a device-resident ground truth, counts left on the device: the
// validation tail of a batch [src/processor.py:350-351] as two more launches behind mav_process_batch_dev, no transfer, no sync.
extern "C" int mav_tpr_fpr_counts_dev(mav_ctx* c, const uint8_t* gt, int gt_images, const uint8_t* mask_fixed, const uint8_t* mask_dyn,
                                      int mask_value, int batch, int64_t* counts_fixed, int64_t* counts_dyn)
{
    if (!c || !gt) return fail(MAV_ERR_ARG, "mav_tpr_fpr_counts_dev: NULL argument");
    if (batch < 1 || batch > c->max_batch) return fail(MAV_ERR_ARG, "batch %d outside [1, %d]", batch, c->max_batch);
    if (gt_images != 1 && gt_images != batch) return fail(MAV_ERR_ARG, "mav_tpr_fpr_counts_dev: gt_images must be 1 (shared) or batch (%d), got %d", batch, gt_images);
    if (mask_value < 1 || mask_value > 65535) return fail(MAV_ERR_ARG, "mav_tpr_fpr_counts_dev: mask_value %d outside [1, 65535]", mask_value);
    if ((mask_fixed && !counts_fixed) || (mask_dyn && !counts_dyn) || (!mask_fixed && !mask_dyn))
        return fail(MAV_ERR_ARG, "mav_tpr_fpr_counts_dev: every mask needs its counts buffer, and at least one mask");
    HIPCHK(hipSetDevice(c->device));
    const size_t stride = gt_images == 1 && batch > 1 ? 0 : c->n0;
    const uint8_t* m0 = mask_fixed ? mask_fixed : mask_dyn;
    const uint8_t* m1 = (mask_fixed && mask_dyn) ? mask_dyn : nullptr;
    ProfScope ps(c, K_MISC);
    launch_tpr_fpr2(c->stream, gt, stride, m0, m1, (unsigned)mask_value, batch, c->W, c->H,
                    (unsigned long long*)(mask_fixed ? counts_fixed : counts_dyn), (unsigned long long*)(m1 ? counts_dyn : nullptr));
    return check_launch("tpr_fpr");
}

extern "C" int mav_process_batch(mav_ctx* c, const uint8_t* prev, const uint8_t* next, const uint32_t* samples, const double* omega,
                                 const double* dt, const uint8_t* frame0, const uint8_t* sky, int batch, const mav_foe_params* fp,
                                 const mav_thr_params* tp, float* flow, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn,
                                 mav_result* results)
{
    return process_host(c, "mav_process_batch", prev, next, nullptr, samples, omega, dt, frame0, sky, batch, fp, tp, flow, phi,
                        mask_fixed, mask_dyn, results);
}

extern "C" int mav_detect(mav_ctx* c, const float* flow, const uint32_t* samples, const double* omega, const double* dt,
                          const uint8_t* frame0, const uint8_t* sky, int batch, const mav_foe_params* fp, const mav_thr_params* tp,
                          double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, mav_result* results)
{
    if (!flow) return fail(MAV_ERR_ARG, "mav_detect: NULL flow");
    return process_host(c, "mav_detect", nullptr, nullptr, flow, samples, omega, dt, frame0, sky, batch, fp, tp, nullptr, phi,
                        mask_fixed, mask_dyn, results);
}

// ---- stage hooks -------------------------------------------------------------------------------------------------
static int layer_of(mav_ctx* c, int k, const Layer** l)
{
    if (!c) return fail(MAV_ERR_ARG, "NULL context");
    if (k < 0 || k >= (int)c->layers.size()) return fail(MAV_ERR_ARG, "layer %d out of range", k);
    HIPCHK(hipSetDevice(c->device));
    c->scratch_next = 0;
    c->last_mf = c->last_md = nullptr; c->last_mask_batch = 0;
    *l = &c->layers[k];
    return MAV_OK;
}
extern "C" int mav_stage_phi_mask(mav_ctx* c, const float* flow, const double* foe, const double* omega, const double* dt,
                                  const uint8_t* sky, int batch, const mav_thr_params* tp, double* phi, uint8_t* mask_fixed,
                                  uint8_t* mask_dyn, int32_t* box)
{
    CHK(check_batch(c, batch, "mav_stage_phi_mask"));
    if (!flow || !foe) return fail(MAV_ERR_ARG, "mav_stage_phi_mask: NULL argument");
    const size_t n = c->n0 * batch;
    mav_thr_params t;
    if (tp) t = *tp; else mav_thr_defaults(&t);
    DevBuf df, dfoe, dsky, dphi, dmf, dmd, dbox;
    CHK(df.upload(c, flow, n * 2 * sizeof(float)));
    CHK(dfoe.upload(c, foe, sizeof(double) * 2 * batch));
    if (sky) CHK(dsky.upload(c, sky, n));
    if (phi) CHK(dphi.alloc(c, n * sizeof(double)));
    if (mask_fixed) CHK(dmf.alloc(c, n));
    if (mask_dyn) CHK(dmd.alloc(c, n));
    if (box) CHK(dbox.alloc(c, sizeof(int32_t) * 4 * batch));
    const DerotParams* derot = nullptr;
    CHK(upload_derot(c, omega, dt, nullptr, batch, true, &derot));
    CHK(detect_dev(c, df.as<float>(), nullptr, derot, nullptr, dsky.as<uint8_t>(), batch, nullptr, &t, dfoe.as<double>(),
                   dphi.as<double>(), dmf.as<uint8_t>(), dmd.as<uint8_t>(), nullptr, nullptr, nullptr, dbox.as<int32_t>()));
    if (phi) CHK(download(c, phi, dphi.p, n * sizeof(double)));
    if (mask_fixed) CHK(download(c, mask_fixed, dmf.p, n));
    if (mask_dyn) CHK(download(c, mask_dyn, dmd.p, n));
    if (box) CHK(download(c, box, dbox.p, sizeof(int32_t) * 4 * batch));
    return mav_sync(c);
}

extern "C" int mav_stage_coefficients(mav_ctx* c, int k, float* g, float* xg, float* xxg, float* ig, float* blur_taps)
{
    if (!c) return fail(MAV_ERR_ARG, "mav_stage_coefficients: NULL context");
    const int n = c->pc.n;
    for (int i = 0; i <= n; i++) {
        if (g) g[i] = c->pc.g[i];
        if (xg) xg[i] = c->pc.xg[i];
        if (xxg) xxg[i] = c->pc.xxg[i];
    }
    if (ig) { ig[0] = c->pc.ig11; ig[1] = c->pc.ig03; ig[2] = c->pc.ig33; ig[3] = c->pc.ig55; }
    if (blur_taps) {
        const Layer* l;
        CHK(layer_of(c, k, &l));
        HIPCHK(hipMemcpy(blur_taps, l->g, sizeof(float) * l->ksize, hipMemcpyDeviceToHost));   // what the kernels actually read
    }
    return MAV_OK;
}

static int stage_blur_resize(mav_ctx* c, const uint8_t* img, int k, bool two_pass, float* out)
{
    const Layer* l;
    CHK(layer_of(c, k, &l));
    if (!img || !out) return fail(MAV_ERR_ARG, "mav_stage_blur_resize: NULL argument");
    const size_t n = (size_t)l->w * l->h;
    // the two-pass form's H x w scratch (one frame: 4 bytes per pixel) is a staging block of this call: a diagnostic hook never
    // allocates the Farneback workspace (GBs at 1080p / 4K) nor freezes "deep_frac"
    DevBuf di, dout, dtmp;
    CHK(di.upload(c, img, c->n0)); CHK(dout.alloc(c, n * sizeof(float))); CHK(dtmp.alloc(c, c->htmp_stride * sizeof(float)));
    launch_blur_resize(c->stream, di.as<uint8_t>(), nullptr, 0, c->n0, 1, c->W, c->H, l->w, l->h, blur_of(c, *l), dtmp.as<float>(), c->htmp_stride,
                       dout.as<float>(), n, two_pass);
    CHK(check_launch("blur_resize"));
    CHK(download(c, out, dout.p, n * sizeof(float)));
    return mav_sync(c);
}
extern "C" int mav_stage_blur_resize(mav_ctx* c, const uint8_t* img, int k, float* out) { return stage_blur_resize(c, img, k, false, out); }
extern "C" int mav_stage_blur_resize_two_pass(mav_ctx* c, const uint8_t* img, int k, float* out) { return stage_blur_resize(c, img, k, true, out); }
extern "C" int mav_stage_polyexp(mav_ctx* c, const float* I, int k, float* R)
{
    const Layer* l;
    CHK(layer_of(c, k, &l));
    if (!I || !R) return fail(MAV_ERR_ARG, "mav_stage_polyexp: NULL argument");
    const size_t n = (size_t)l->w * l->h;
    DevBuf di, dr;
    CHK(di.upload(c, I, n * sizeof(float))); CHK(dr.alloc(c, 5 * n * sizeof(float)));
    launch_polyexp(c->stream, di.as<float>(), n, 1, l->w, l->h, c->pc, dr.as<float>(), 5 * n);
    CHK(check_launch("polyexp"));
    CHK(download(c, R, dr.p, 5 * n * sizeof(float)));
    return mav_sync(c);
}
extern "C" int mav_stage_update_matrices(mav_ctx* c, const float* R0, const float* R1, const float* flow, int k, float* M)
{
    const Layer* l;
    CHK(layer_of(c, k, &l));
    if (!R0 || !R1 || !flow || !M) return fail(MAV_ERR_ARG, "mav_stage_update_matrices: NULL argument");
    const size_t n = (size_t)l->w * l->h;
    DevBuf d0, d1, df, dm;
    CHK(d0.upload(c, R0, 5 * n * sizeof(float))); CHK(d1.upload(c, R1, 5 * n * sizeof(float)));
    CHK(df.upload(c, flow, 2 * n * sizeof(float))); CHK(dm.alloc(c, 5 * n * sizeof(float)));
    launch_update_matrices_flow(c->stream, d0.as<float>(), d1.as<float>(), 5 * n, df.as<float>(), 2 * n, 1, l->w, l->h, dm.as<float>(), 5 * n);
    CHK(check_launch("update_matrices"));
    CHK(download(c, M, dm.p, 5 * n * sizeof(float)));
    return mav_sync(c);
}
extern "C" int mav_stage_blur_iter(mav_ctx* c, const float* R0, const float* R1, const float* M, int k, int update, float* flow, float* M_out)
{
    const Layer* l;
    CHK(layer_of(c, k, &l));
    if (!R0 || !R1 || !M || !flow || (update && !M_out)) return fail(MAV_ERR_ARG, "mav_stage_blur_iter: NULL argument");
    const size_t n = (size_t)l->w * l->h;
    DevBuf d0, d1, dm, dmo, df;
    CHK(d0.upload(c, R0, 5 * n * sizeof(float))); CHK(d1.upload(c, R1, 5 * n * sizeof(float)));
    CHK(dm.upload(c, M, 5 * n * sizeof(float))); CHK(dmo.alloc(c, 5 * n * sizeof(float))); CHK(df.alloc(c, 2 * n * sizeof(float)));
    launch_blur_iter(c->stream, dm.as<float>(), dmo.as<float>(), 5 * n, d0.as<float>(), d1.as<float>(), 5 * n, 1, l->w, l->h,
                     c->fb.winsize, update, 1, df.as<float>(), 2 * n, 0, -1, c->strip);
    CHK(check_launch("blur_iter"));
    CHK(download(c, flow, df.p, 2 * n * sizeof(float)));
    if (update) CHK(download(c, M_out, dmo.p, 5 * n * sizeof(float)));
    return mav_sync(c);
}

// ---- RCCL (loaded lazily so the single-GPU path carries no collective library) ----------------------------------------
typedef int (*nccl_get_uid_t)(void*);
struct uid128 { char b[128]; };  // ncclUniqueId is passed by value: 128 bytes
typedef int (*nccl_init_rank2_t)(void**, int, uid128, int);
typedef int (*nccl_destroy_t)(void*);
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_t)(int);
static void* g_rccl = nullptr;
static int rccl_sym(const char* name, void** fn)
{
    if (!g_rccl) {
        g_rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl) g_rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!g_rccl) return fail(MAV_ERR_STATE, "cannot load RCCL: %s", dlerror());
    }
    *fn = dlsym(g_rccl, name);
    if (!*fn) return fail(MAV_ERR_STATE, "RCCL symbol %s missing", name);
    return MAV_OK;
}
// Versions at the seam where this library meets a runtime somebody else loaded: in the multi-GPU bench torch has already mapped its
// own HIP runtime and RCCL (same SONAMEs), so libmavflow -- built by this tree's hipcc -- binds to those.
typedef int (*nccl_get_version_t)(int*);
extern "C" int mav_runtime_info(char* buf, size_t cap)
{
    if (!buf || cap < 16) return fail(MAV_ERR_ARG, "mav_runtime_info: NULL argument");
    int rt = 0, drv = 0, nccl = 0;
    if (hipRuntimeGetVersion(&rt) != hipSuccess) rt = -1;
    if (hipDriverGetVersion(&drv) != hipSuccess) drv = -1;
    void* fn = nullptr;
    if (g_rccl && (fn = dlsym(g_rccl, "ncclGetVersion"))) ((nccl_get_version_t)fn)(&nccl);
    snprintf(buf, cap, "{\"built_with_hip\": \"%d.%d.%d\", \"hip_runtime\": %d, \"hip_runtime_major\": %d, \"hip_driver\": %d, \"rccl\": %d}",
             HIP_VERSION_MAJOR, HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt, rt > 0 ? rt / 10000000 : -1, drv, nccl);
    return MAV_OK;
}
extern "C" int mav_comm_unique_id(void* id128)
{
    if (!id128) return fail(MAV_ERR_ARG, "mav_comm_unique_id: NULL");
    void* fn;
    CHK(rccl_sym("ncclGetUniqueId", &fn));
    int rc = ((nccl_get_uid_t)fn)(id128);
    return rc ? fail(MAV_ERR_HIP, "ncclGetUniqueId failed: %d", rc) : MAV_OK;
}
extern "C" int mav_comm_init(mav_ctx* c, const void* id128, int rank, int nranks, void** comm_out)
{
    if (!c || !id128 || !comm_out) return fail(MAV_ERR_ARG, "mav_comm_init: NULL argument");
    HIPCHK(hipSetDevice(c->device));
    // the collective runs on this context's stream inside whatever HIP runtime the process loaded first: refuse a runtime of another
    // major version than the one the kernels' host code was compiled against (launch ABI, stream handles) instead of finding out later
    int rt = 0;
    HIPCHK(hipRuntimeGetVersion(&rt));
    if (rt / 10000000 != HIP_VERSION_MAJOR)
        return fail(MAV_ERR_STATE, "mav_comm_init: libmavflow was built with HIP %d.%d but the process runs HIP runtime %d (major %d): rebuild against "
                    "the runtime the launcher loads", HIP_VERSION_MAJOR, HIP_VERSION_MINOR, rt, rt / 10000000);
    void* fn;
    CHK(rccl_sym("ncclCommInitRank", &fn));
    uid128 id;
    memcpy(&id, id128, sizeof(id));
    int rc = ((nccl_init_rank2_t)fn)(comm_out, nranks, id, rank);
    return rc ? fail(MAV_ERR_HIP, "ncclCommInitRank failed: %d", rc) : MAV_OK;
}
typedef int (*nccl_comm_count_t)(void*, int*);
extern "C" int mav_comm_count(void* comm, int* nranks)
{
    if (!comm || !nranks) return fail(MAV_ERR_ARG, "mav_comm_count: NULL argument");
    void* fn;
    CHK(rccl_sym("ncclCommCount", &fn));
    int rc = ((nccl_comm_count_t)fn)(comm, nranks);
    return rc ? fail(MAV_ERR_HIP, "ncclCommCount failed: %d", rc) : MAV_OK;
}
extern "C" int mav_comm_destroy(void* comm)
{
    if (!comm) return MAV_OK;
    void* fn;
    CHK(rccl_sym("ncclCommDestroy", &fn));
    ((nccl_destroy_t)fn)(comm);
    return MAV_OK;
}
extern "C" int mav_allgather_results(mav_ctx* c, void* comm, const void* local_dev, size_t bytes_per_rank, void* all_dev)
{
    if (!c || !comm || !local_dev || !all_dev) return fail(MAV_ERR_ARG, "mav_allgather_results: NULL argument");
    void* fn;
    CHK(rccl_sym("ncclAllGather", &fn));
    int rc = ((nccl_allgather_t)fn)(local_dev, all_dev, bytes_per_rank, /* ncclInt8 */ 0, comm, c->stream);
    return rc ? fail(MAV_ERR_HIP, "ncclAllGather failed: %d", rc) : MAV_OK;
}
