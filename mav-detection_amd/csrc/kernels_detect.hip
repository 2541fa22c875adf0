// Detection kernels for gfx950: derotation, FoE candidates + RANSAC vote, phi + threshold masks + box, and the
// small reductions around them.  Compiled with -ffp-contract=off: every double operation below is written in the
// reference's numpy evaluation order so results are bit-identical to
//   /root/reference/src/detector.py:70-117            (derotate)
//   /root/reference/src/focus_of_expansion.py:32-86   (get_FOE_dense, ransac; utils.py:183-197 line_intersection)
//   /root/reference/src/focus_of_expansion.py:150-184 (get_phi)
//   /root/reference/src/processor.py:333-341          (threshold block)
//   /root/reference/src/im_helpers.py:55-84,244-252   (get_simple_bounding_box, calculate_tpr_fpr)
//   /root/reference/src/detector.py:280-312           (analyze_pyramid, level 0)
// with one documented exception: arccos comes from the device math library, not the host's libm.
#include "mavflow_internal.h"

#include <limits.h>
#include <algorithm>
#include <math.h>

#include <type_traits>

// Derotated flow vector at pixel (col, row) in double, reference operation order (detector.py:92-114).
static __device__ __forceinline__ void derot_apply(float2 f, const DerotParams* __restrict__ dp, int W, int H, int row, int col,
                                                   double* fu, double* fv)
{
    double u = (double)f.x, v = (double)f.y;
    if (dp && dp->mode == MAV_PAIR_DEROTATE) {
        const double x = -((double)col / (double)W - 0.5) * 2.0;
        const double y = -((double)row / (double)H - 0.5) * 2.0;
        const double o0 = dp->o0, o1 = dp->o1, o2 = dp->o2;
        double du = (((o0 * x) * y - o1 * (x * x)) - o1) + o2 * y;
        double dv = ((((-o2) * x + o0) + o0 * (y * y))) - (o1 * x) * y;
        du = du * dp->sx;
        dv = dv * dp->sy;
        u = u - du;
        v = v - dv;
    }
    *fu = u; *fv = v;
}
static __device__ __forceinline__ void derot_apply(double2 f, const DerotParams*, int, int, int, int, double* fu, double* fv)
{
    *fu = f.x; *fv = f.y;
}
static __device__ __forceinline__ float2 flow_raw(const float* __restrict__ flow, int W, int row, int col)
{
    return *(const float2*)(flow + ((size_t)row * W + col) * 2);
}
static __device__ __forceinline__ double2 flow_raw(const double* __restrict__ flow, int W, int row, int col)
{
    return *(const double2*)(flow + ((size_t)row * W + col) * 2);
}
template <typename FlowT>
static __device__ __forceinline__ void flow_at(const FlowT* __restrict__ flow, const DerotParams* __restrict__ dp, int W, int H,
                                               int row, int col, double* fu, double* fv)
{
    derot_apply(flow_raw(flow, W, row, col), dp, W, H, row, col, fu, fv);
}

__global__ __launch_bounds__(256) void k_derotate(const float* __restrict__ flow, const DerotParams* __restrict__ derot, int W,
                                                  int H, double* __restrict__ out)
{
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t img = (size_t)W * H * 2;
    double u, v;
    flow_at(flow + b * img, derot + b, W, H, y, x, &u, &v);
    *(double2*)(out + b * img + ((size_t)y * W + x) * 2) = make_double2(u, v);
}

void launch_derotate(hipStream_t st, const float* flow, const DerotParams* derot, int B, int W, int H, double* out)
{
    dim3 grid((W + 63) / 64, (H + 3) / 4, B);
    hipLaunchKernelGGL(k_derotate, grid, dim3(256), 0, st, flow, derot, W, H, out);
}

// ------------------------------------------------------------------------------------------------------------
// FoE candidates: one 1024-thread workgroup per pair.  Thread i intersects the flow lines through samples i and
// i+N; surviving rows (x != 0.0) are compacted IN INDEX ORDER (ballot + wave prefix) because RANSAC's tie-break
// is "first candidate wins".
// ------------------------------------------------------------------------------------------------------------
template <typename FlowT>
__global__ __launch_bounds__(1024) void k_foe_candidates(const FlowT* __restrict__ flow, const DerotParams* __restrict__ derot,
                                                         const uint32_t* __restrict__ samples, int W, int H, int N,
                                                         double mag2_thr, float mag2_thr_f32, FoeScratch sc,
                                                         int32_t* __restrict__ box_acc, unsigned long long* __restrict__ max_phi_bits)
{
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t img = (size_t)W * H * 2;
    const FlowT* fl = flow + b * img;
    const DerotParams* dp = derot ? derot + b : nullptr;
    // frame-0 pair (float32 flow handed to numpy as is): only the |flow2| gate runs in float32 there -- `flow + coord` is
    // float32 + uint32 = float64, so the intersections below are the same double arithmetic either way
    const bool f32_gate = std::is_same<FlowT, float>::value && dp && dp->mode == MAV_PAIR_FRAME0;
    const uint32_t* smp = samples + (size_t)b * 4 * N;
    double* cand = sc.cand + (size_t)b * 2 * N;
    if (tid == 0) {
        base_s = 0; sc.best_key[b] = 0ull;
        sc.done[2 * b] = 0u; sc.done[2 * b + 1] = 0u;           // tickets of the vote and of the phi kernel (both run after this one)
        if (box_acc) { box_acc[4 * b] = INT_MAX; box_acc[4 * b + 1] = INT_MAX; box_acc[4 * b + 2] = -1; box_acc[4 * b + 3] = -1; }
        if (max_phi_bits) max_phi_bits[b] = 0ull;
    }
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        double ix = 0.0, iy = 0.0;
        if (i < N) {
            const uint32_t r1 = smp[2 * i], c1 = smp[2 * i + 1];
            const uint32_t r2 = smp[2 * (i + N)], c2 = smp[2 * (i + N) + 1];
            if (r1 < (uint32_t)H && r2 < (uint32_t)H && c1 < (uint32_t)W && c2 < (uint32_t)W) {
                double f1x, f1y, f2x, f2y;
                flow_at(fl, dp, W, H, (int)r1, (int)c1, &f1x, &f1y);
                flow_at(fl, dp, W, H, (int)r2, (int)c2, &f2x, &f2y);
                const double mag2 = f2x * f2x + f2y * f2y;
                bool below = mag2 < mag2_thr;      // == sqrt(mag2) < mag_threshold, see sq_threshold() on the host
                if (f32_gate) {                    // np.linalg.norm of a float32 pair: float32 products, sum and root
                    const float gx = (float)f2x, gy = (float)f2y;
                    const float m2 = gx * gx + gy * gy;
                    below = m2 < mag2_thr_f32;
                }
                if (!below) {
                    const double p1x = (double)c1, p1y = (double)r1, p2x = (double)c2, p2y = (double)r2;
                    const double q1x = f1x + p1x, q1y = f1y + p1y, q2x = f2x + p2x, q2y = f2y + p2y;
                    const double xd0 = p1x - q1x, xd1 = p2x - q2x, yd0 = p1y - q1y, yd1 = p2y - q2y;
                    const double div = xd0 * yd1 - xd1 * yd0;
                    if (!(div == 0.0)) {
                        const double d0 = p1x * q1y - p1y * q1x;
                        const double d1 = p2x * q2y - p2y * q2x;
                        ix = (d0 * xd1 - d1 * xd0) / div;
                        iy = (d0 * yd1 - d1 * yd0) / div;
                        keep = (ix != 0.0);
                    }
                }
            }
        }
        const unsigned long long bal = __ballot(keep);
        const int pre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int k = 0; k < wv; k++) off += wave_tot[k];
        if (keep) {
            cand[2 * (off + pre)] = ix;
            cand[2 * (off + pre) + 1] = iy;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int k = 0; k < 16; k++) t += wave_tot[k];
            base_s += t;
        }
        __syncthreads();
    }
    if (tid == 0) sc.count[b] = base_s;
}

// RANSAC vote: score_i = #{j : |e_j - e_i| < r} - 1; winner = largest score, lowest index (strict > in the loop).
// One WAVE scores four candidates against all M: its lanes stride over j (one coalesced double2 load feeds the four tests), the
// counts are reduced over the wave.  Grid (ceil(N/16), B).  (Round 2's form -- one THREAD per candidate, 4 workgroups per pair --
// took 23 us for a single pair: a thousand dependent LDS reads per thread.)  The workgroup whose ticket is last reads the winner
// and writes the pair's FoE: no separate finalize launch.  Every workgroup's atomicMax has returned before its ticket is drawn.
__global__ __launch_bounds__(256) void k_ransac(FoeScratch sc, int N, double dist2_thr, double* __restrict__ foe)
{
    __shared__ unsigned long long wkey[4];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int M = sc.count[b];
    const double2* cand = (const double2*)(sc.cand + (size_t)b * 2 * N);
    const int i0 = ((int)blockIdx.x * 4 + wv) * 4;
    unsigned long long key = 0ull;
    if (i0 < M) {
        double2 e[4];
        int cnt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = cand[min(i0 + k, M - 1)];
        for (int j = lane; j < M; j += 64) {
            const double2 c = cand[j];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double dx = c.x - e[k].x, dy = c.y - e[k].y;
                const double d2 = dx * dx + dy * dy;
                cnt[k] += (d2 < dist2_thr) ? 1 : 0;  // == (sqrt(d2) < ransac_threshold)
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            for (int o = 32; o > 0; o >>= 1) cnt[k] += __shfl_xor(cnt[k], o);
            const int score = cnt[k] - 1;
            if (i0 + k < M && score > 0) {
                const unsigned long long kk = ((unsigned long long)(unsigned)score << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)(i0 + k));
                key = kk > key ? kk : key;
            }
        }
    }
    if (lane == 0) wkey[wv] = key;
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int k = 1; k < 4; k++) key = wkey[k] > key ? wkey[k] : key;
        unsigned long long seen = 0ull;
        if (key) seen = atomicMax(&sc.best_key[b], key);              // returning form: performed before the ticket below
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");
        const unsigned ticket = atomicAdd(&sc.done[2 * b], 1u);
        if (ticket == gridDim.x - 1) {
            const unsigned long long best = __hip_atomic_load(&sc.best_key[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double x = 0.0, y = 0.0;
            if (best) {
                const unsigned idx = 0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull);
                x = sc.cand[(size_t)b * 2 * N + 2 * idx];
                y = sc.cand[(size_t)b * 2 * N + 2 * idx + 1];
            }
            foe[2 * b] = x;
            foe[2 * b + 1] = y;
            __hip_atomic_store(&sc.done[2 * b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename FlowT>
static void launch_foe_t(hipStream_t st, const FlowT* flow, const DerotParams* derot, const uint32_t* samples, int B, int W,
                         int H, int N, double mag2_thr, float mag2_thr_f32, double dist2_thr, FoeScratch s, double* foe,
                         int32_t* box_acc, unsigned long long* max_phi_bits)
{
    hipLaunchKernelGGL(k_foe_candidates<FlowT>, dim3(B), dim3(1024), 0, st, flow, derot, samples, W, H, N, mag2_thr, mag2_thr_f32, s,
                       box_acc, max_phi_bits);
    hipLaunchKernelGGL(k_ransac, dim3((N + 15) / 16, B), dim3(256), 0, st, s, N, dist2_thr, foe);
}
void launch_foe_f32(hipStream_t st, const float* flow, const DerotParams* derot, const uint32_t* samples, int B, int W, int H,
                    int N, double mag2_thr, float mag2_thr_f32, double dist2_thr, FoeScratch s, double* foe, int32_t* box_acc,
                    unsigned long long* max_phi_bits)
{
    launch_foe_t<float>(st, flow, derot, samples, B, W, H, N, mag2_thr, mag2_thr_f32, dist2_thr, s, foe, box_acc, max_phi_bits);
}
void launch_foe_f64(hipStream_t st, const double* flow, const uint32_t* samples, int B, int W, int H, int N, double mag2_thr,
                    double dist2_thr, FoeScratch s, double* foe, int32_t* box_acc, unsigned long long* max_phi_bits)
{
    launch_foe_t<double>(st, flow, nullptr, samples, B, W, H, N, mag2_thr, 0.f, dist2_thr, s, foe, box_acc, max_phi_bits);
}

// ------------------------------------------------------------------------------------------------------------
// phi + both threshold masks + box extents + max(phi), one pass over the flow.  Workgroup = 4 rows x 256 columns,
// lane-contiguous float2/double2 loads; box extents and max(phi) are reduced per wave, then one atomic per wave.
// ------------------------------------------------------------------------------------------------------------
__global__ void k_box_init(int32_t* box_acc, unsigned long long* max_phi_bits, unsigned* done, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    box_acc[4 * b] = INT_MAX; box_acc[4 * b + 1] = INT_MAX; box_acc[4 * b + 2] = -1; box_acc[4 * b + 3] = -1;
    if (max_phi_bits) max_phi_bits[b] = 0ull;
    if (done) done[2 * b + 1] = 0u;
}
// (the fused chain needs no such launch: k_foe_candidates initialises its pair's accumulators)
void launch_box_init(hipStream_t st, int32_t* box_acc, unsigned long long* max_phi_bits, unsigned* done, int B)
{
    hipLaunchKernelGGL(k_box_init, dim3((B + 63) / 64), dim3(64), 0, st, box_acc, max_phi_bits, done, B);
}

// Box extents of one wave -> the pair's accumulators.  The accumulators only ever grow, so a (possibly stale) read that
// already contains this wave's extents makes the four same-address atomics unnecessary: after the first few waves of a
// pair almost every wave skips them (they were the phi kernel's bottleneck: thousands of atomics on four words).
static __device__ __forceinline__ void wave_box_commit(int x0, int y0, int x1, int y1, int32_t* acc)
{
    for (int o = 32; o > 0; o >>= 1) {
        x0 = min(x0, __shfl_xor(x0, o)); y0 = min(y0, __shfl_xor(y0, o));
        x1 = max(x1, __shfl_xor(x1, o)); y1 = max(y1, __shfl_xor(y1, o));
    }
    if ((threadIdx.x & 63) == 0 && x1 >= 0) {
        const int c0 = __hip_atomic_load(&acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int c1 = __hip_atomic_load(&acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int c2 = __hip_atomic_load(&acc[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int c3 = __hip_atomic_load(&acc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // returning atomics whose results are waited for: when the workgroup draws its ticket (k_phi_mask's tail) they have been performed
        int seen = 0;
        if (x0 < c0) seen |= atomicMin(&acc[0], x0);
        if (y0 < c1) seen |= atomicMin(&acc[1], y0);
        if (x1 > c2) seen |= atomicMax(&acc[2], x1);
        if (y1 > c3) seen |= atomicMax(&acc[3], y1);
        asm volatile("s_waitcnt vmcnt(0)" ::"v"(seen) : "memory");
    }
}

// Screening constants for the single-precision fast path (host-computed, see phi_screen()).
struct PhiScreen {
    int enabled;         // 0: every pixel takes the exact double path
    float tan_fixed;     // tan(fixed_deg), fixed_deg in (0, 80)
    float margin_fixed;  // MAV_SCREEN_TAN_MARGIN * (1 + tan_fixed)
    float fmm2, dmm2;    // fixed_min_mag^2, dyn_min_mag^2
    float dyn_ab, dyn_c; // dyn_a + dyn_b, dyn_c
};
#define MAV_SCREEN_TAN_MAX_DEG 17.0f     // the degree-9 series of tan below is good to 5e-8 relative up to 0.2967 rad
#define MAV_SCREEN_TAN_MARGIN 4e-6f      // decision margin, in units of |flow| * |p - FoE| * (1 + tan T); error bound 1.2e-6

// Exact phi of one pixel (degrees), numpy order of operations (focus_of_expansion.py:163-177).
static __device__ __forceinline__ double phi_exact(double u, double v, double d2x, double d2y, double* fm_out)
{
    const double fm = sqrt(u * u + v * v);
    const double dist = sqrt(d2x * d2x + d2y * d2y);
    const double prod = fm * dist;
    const double norm = (prod > 1e-6 || prod != prod) ? prod : 1e-6;  // np.maximum propagates NaN
    double arg = (u * d2x + v * d2y) / norm;
    *fm_out = fm;
    if (arg != arg) return 0.0;  // angle_diff[isnan] = 0
    arg = arg < -1.0 ? -1.0 : (arg > 1.0 ? 1.0 : arg);
    return acos(arg) * (180.0 / 3.141592653589793238462643383279502884);
}

// Single-precision screen of one pixel of the phi / mask stage.  Returns true when both verdicts are certain (then *fix_out /
// *dyn_out hold them); false sends the pixel to phi_pixel_exact().  For a threshold T below 90 degrees
//     phi > T   <=>   dot <= 0  or  |cross| > tan(T) * dot        (dot = f . d, cross = f x d, d = p - FoE),
// and this form is well conditioned exactly where the thresholds live (a few degrees): in float32 dot and |cross| are each
// within 4.2e-7 S of their exact values (S = |f| |d|), tan(T) within 6.5e-7 relative, so  r = |cross| - tan(T) dot  is within
// 1.2e-6 S (1 + tan T) of its exact value and its sign is the exact path's verdict whenever |r| exceeds the 4e-6 margin: an
// angular band of ~2e-4 degrees, ~1e-5 of the pixels.  (The arccos-argument form of round 1, arg < cos T with a 1e-4 band, is
// ill conditioned at small T -- d arg = sin T dT -- and sent every pixel within ~0.15 degrees of a 2-degree threshold down the
// double path.)  tan(T) of the dynamic threshold T = dyn_a + dyn_b + dyn_c / |f| comes from the degree-9 series, good to 5e-8
// relative up to 17 degrees (every |f| >= 0.5 with the reference's constants); larger dynamic thresholds go to the exact path.
// Magnitude gates: 1e-5 relative around both.  Branch-free, explicit FMAs (this file is built with -ffp-contract=off): the
// kernel is instruction-bound (PMC: 10 % of its cycles waiting on memory), so every instruction here is paid 133 M times a step.
static __device__ __forceinline__ bool phi_pixel_screen(float uf, float vf, float dxf, float dyf, bool notsky, const PhiScreen& scr,
                                                        bool* fix_out, bool* dyn_out)
{
    const float m2 = fmaf(uf, uf, vf * vf), dd = fmaf(dxf, dxf, dyf * dyf);
    const float prod2 = m2 * dd;
    const float dot = fmaf(uf, dxf, vf * dyf);
    const float crs = fabsf(fmaf(uf, dyf, -(vf * dxf)));
    const float S = __builtin_amdgcn_sqrtf(prod2);            // raw v_sqrt_f32 (1 ulp): S only scales the margins
    // norm floor (1e-6), inf / NaN and the two magnitude gates' bands stay on the exact path
    const bool ok = (int)(prod2 > 1e-8f) & (int)(prod2 < 1e30f) & (int)(fabsf(m2 - scr.fmm2) > 1e-5f * scr.fmm2) & (int)(fabsf(m2 - scr.dmm2) > 1e-5f * scr.dmm2);
    const bool act_f = (m2 > scr.fmm2) & notsky, act_d = (m2 > scr.dmm2) & notsky;
    const float rf = fmaf(-scr.tan_fixed, dot, crs);
    const bool ok_f = fabsf(rf) > scr.margin_fixed * S;
    const float T = fmaf(scr.dyn_c, __builtin_amdgcn_rsqf(m2), scr.dyn_ab);      // degrees, >= 0; raw v_rsq_f32 (1 ulp, in the budget)
    const float x = T * 0.017453292519943295f, x2 = x * x;
    const float tT = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 0.02186948854f, 0.05396825397f), 0.13333333333f), 0.33333333333f), 1.f);
    const float rd = fmaf(-tT, dot, crs);
    const bool ok_d = (T < MAV_SCREEN_TAN_MAX_DEG) & (fabsf(rd) > MAV_SCREEN_TAN_MARGIN * fmaf(S, tT, S));
    *fix_out = act_f & (rf > 0.f);
    *dyn_out = act_d & (rd > 0.f);
    return ok & (!act_f | ok_f) & (!act_d | ok_d);
}

// The exact path of one pixel: double arithmetic in numpy's order (phi_exact) and the literal threshold block.  Out of line on
// purpose: only ~1e-5 of the pixels come here when the screen is on, and sixteen inlined copies of the double sqrt / divide /
// acos sequences made the kernel 69 KB of code -- more than the instruction cache two CUs share.
// (everything by value, result in registers: no stack frame, the kernel needs no scratch memory)
struct PhiVerdict { double ph; unsigned bits; };     // bits: 1 = fixed mask, 2 = dynamic mask
static __device__ __noinline__ PhiVerdict phi_pixel_exact(double u, double v, double d2x, double d2y, int notsky, double fixed_deg,
                                                          double fixed_min_mag, double dyn_min_mag, double dyn_a, double dyn_b,
                                                          double dyn_c)
{
    double fm;
    const double ph = phi_exact(u, v, d2x, d2y, &fm);
    const double t = dyn_b + dyn_c / fm;
    const bool hi = ph > (dyn_a + t);
    const bool lo = ph < (dyn_a - t);
    const bool dyn = (fm > dyn_min_mag) && notsky && (lo || hi);
    const double gated = ((fm > fixed_min_mag) && notsky) ? ph : 0.0;
    return PhiVerdict{ph, (gated > fixed_deg ? 1u : 0u) | (dyn ? 2u : 0u)};
}

// One pixel of a frame-0 pair: get_phi (focus_of_expansion.py:163-177) and the threshold block (processor.py:333-341) as numpy
// evaluates them on a FLOAT32 flow array -- zeros_like keeps diff2 in float32 (the int64 - float64 coordinate difference is
// rounded into it), np.linalg.norm, the products, the quotient and the thresholds (8 / mag, 0.5 + ..., 0.25 + ...) are all
// float32 operations, rad2deg multiplies by 180f / pif.  No contraction (this file is built with -ffp-contract=off).
// arccos: the correctly rounded float32 value (double acos, rounded once).
struct ThrF32 { float fixed_deg, fixed_min_mag, dyn_min_mag, dyn_a, dyn_b, dyn_c; };
static __device__ __noinline__ PhiVerdict phi_pixel_f32(float u, float v, float d2x, float d2y, int notsky, float fixed_deg,
                                                        float fixed_min_mag, float dyn_min_mag, float dyn_a, float dyn_b, float dyn_c)
{
    const ThrF32 t{fixed_deg, fixed_min_mag, dyn_min_mag, dyn_a, dyn_b, dyn_c};
    const float fm = sqrtf(u * u + v * v);
    const float dist = sqrtf(d2x * d2x + d2y * d2y);
    const float prod = fm * dist;
    const float norm = (prod > 1e-6f || prod != prod) ? prod : 1e-6f;      // np.maximum propagates NaN
    float arg = (u * d2x + v * d2y) / norm;
    float ph = 0.f;                                                         // angle_diff[isnan] = 0
    if (arg == arg) {
        arg = arg < -1.f ? -1.f : (arg > 1.f ? 1.f : arg);
        ph = (float)acos((double)arg) * __int_as_float(0x42652EE0);         // 180.0f / 3.14159274f
    }
    const float tt = t.dyn_b + t.dyn_c / fm;
    const bool hi = ph > (t.dyn_a + tt);
    const bool lo = ph < (t.dyn_a - tt);
    const bool dyn = (fm > t.dyn_min_mag) && notsky && (lo || hi);
    const float gated = ((fm > t.fixed_min_mag) && notsky) ? ph : 0.f;
    return PhiVerdict{(double)ph, (gated > t.fixed_deg ? 1u : 0u) | (dyn ? 2u : 0u)};
}

// VEC = 4: one thread = 4 consecutive pixels x 4 rows (W % 4 == 0): float4 / double2 row loads, uchar4 mask stores;
// workgroup = 256 columns x 16 rows.  VEC = 1: any width, one pixel per lane and row, 4 rows per thread.
// ROT = false: the launch carries no per-pair parameters (no rates, no frame-0 pair): pass 1 has no derotation code at all.
template <typename FlowT, int VEC, bool ROT>
__global__ __launch_bounds__(256) void k_phi_mask(const FlowT* __restrict__ flow, const DerotParams* __restrict__ derot,
                                                  const double* __restrict__ foe, const uint8_t* __restrict__ sky, int W, int H,
                                                  mav_thr_params thr, PhiScreen scr, double* __restrict__ phi_out,
                                                  uint8_t* __restrict__ mfix, uint8_t* __restrict__ mdyn,
                                                  int32_t* __restrict__ box_acc, unsigned long long* __restrict__ max_phi_bits,
                                                  unsigned* __restrict__ done, mav_result* __restrict__ results,
                                                  int32_t* __restrict__ box_out)
{
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t npx = (size_t)W * H;
    const FlowT* fl = flow + b * npx * 2;
    const DerotParams* dp = (ROT && derot) ? derot + b : nullptr;
    const bool f32_pair = ROT && std::is_same<FlowT, float>::value && dp && dp->mode == MAV_PAIR_FRAME0;
    const double foex = foe[2 * b], foey = foe[2 * b + 1];
    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = -1, by1 = -1;
    double pmax = 0.0;
    const int xb = (blockIdx.x * 64 + lane) * VEC;
    const bool colok = xb < W;                            // lanes past the right edge stay for the wave reductions below
    const int xld = colok ? xb : 0;
    // a workgroup walks every gridDim.y-th block of 16 rows: fewer, longer-lived waves (the launch is wave-launch bound otherwise)
    for (int yblk = blockIdx.y; yblk * 16 < H; yblk += gridDim.y) {
    const int yb = yblk * 16 + wv * 4;
    // Pass 1 (every pixel, compact straight-line code): all flow vectors and sky words of the thread are requested up front, then
    // each pixel is screened in float32; bit 4r + j of fixb / dynb holds its verdicts, of `todo` that it still needs the exact path
    // (inside a guard band, derotated, screen off, or a frame-0 pair).
    decltype(flow_raw(fl, W, 0, 0)) raw[4][VEC];
    uint32_t skw[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int yc = min(yb + r, H - 1);                // rows past the bottom: re-read the last row, never stored
#pragma unroll
        for (int j = 0; j < VEC; j++) raw[r][j] = flow_raw(fl, W, yc, xld + j);
        const size_t oc = b * npx + (size_t)yc * W + xld;
        skw[r] = !sky ? 0u : (VEC == 4 ? *(const uint32_t*)(sky + oc) : (uint32_t)sky[oc]);
    }
    unsigned fixb = 0u, dynb = 0u, todo = 0u;
    const bool screen = scr.enabled && !f32_pair;
    float dxf[VEC], dyf[4];                               // p - FoE: formed in double as the exact path does, then rounded once
#pragma unroll
    for (int j = 0; j < VEC; j++) dxf[j] = (float)((double)(xb + j) - foex);
#pragma unroll
    for (int r = 0; r < 4; r++) dyf[r] = (float)((double)(yb + r) - foey);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (yb + r >= H || !colok) continue;
#pragma unroll
        for (int j = 0; j < VEC; j++) {
            const unsigned bit = 1u << (4 * r + j);
            const bool notsky = ((skw[r] >> (8 * j)) & 255u) == 0u;
            bool fix = false, dyn = false, sure = false;
            if (screen) {
                float uf = (float)raw[r][j].x, vf = (float)raw[r][j].y;
                if constexpr (ROT) {                      // derotation (a wave-uniform branch) in double, then the float32 screen
                    double u, v;
                    derot_apply(raw[r][j], dp, W, H, yb + r, xb + j, &u, &v);
                    uf = (float)u; vf = (float)v;
                }
                sure = phi_pixel_screen(uf, vf, dxf[j], dyf[r], notsky, scr, &fix, &dyn);
            }
            const unsigned m = sure ? bit : 0u;           // selects, not branches: this loop is the kernel's whole cost
            fixb |= fix ? m : 0u;
            dynb |= dyn ? m : 0u;
            todo |= bit ^ m;
        }
    }
    // Pass 2 (rare when the screen is on): the exact path, out of line, one pixel at a time; the pixel's inputs are re-read
    // (L1 / L2 hits) instead of being indexed out of registers.
    for (unsigned m = todo; m; m &= m - 1u) {
        const int k = __ffs(m) - 1, r = k >> 2, j = k & 3;
        const int x = xb + j, y = yb + r;
        const size_t o = b * npx + (size_t)y * W + x;
        const int notsky = !sky || sky[o] == 0;
        const auto rv = flow_raw(fl, W, y, x);
        const double d2x = (double)x - foex, d2y = (double)y - foey;
        PhiVerdict pv;
        if (f32_pair)       // diff2 is a float32 array in the reference: the double coordinate difference is rounded into it
            pv = phi_pixel_f32((float)rv.x, (float)rv.y, (float)d2x, (float)d2y, notsky, (float)thr.fixed_deg, (float)thr.fixed_min_mag,
                               (float)thr.dyn_min_mag, (float)thr.dyn_a, (float)thr.dyn_b, (float)thr.dyn_c);
        else {
            double u, v;
            derot_apply(rv, dp, W, H, y, x, &u, &v);
            pv = phi_pixel_exact(u, v, d2x, d2y, notsky, thr.fixed_deg, thr.fixed_min_mag, thr.dyn_min_mag, thr.dyn_a, thr.dyn_b, thr.dyn_c);
        }
        if (phi_out) phi_out[o] = pv.ph;
        pmax = pv.ph > pmax ? pv.ph : pmax;
        if (pv.bits & 1u) fixb |= 1u << k;
        if (pv.bits & 2u) dynb |= 1u << k;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = yb + r;
        if (y >= H || !colok) continue;
        const size_t o = b * npx + (size_t)y * W + xb;
        const unsigned f4 = (fixb >> (4 * r)) & 15u, d4 = (dynb >> (4 * r)) & 15u;
        if (VEC == 4) {
            // bit j -> byte j (0 / 1): spread the nibble over a dword
            if (mfix) *(uint32_t*)(mfix + o) = (f4 & 1u) | ((f4 & 2u) << 7) | ((f4 & 4u) << 14) | ((f4 & 8u) << 21);
            if (mdyn) *(uint32_t*)(mdyn + o) = (d4 & 1u) | ((d4 & 2u) << 7) | ((d4 & 4u) << 14) | ((d4 & 8u) << 21);
        } else {
            if (mfix) mfix[o] = (uint8_t)(f4 & 1u);
            if (mdyn) mdyn[o] = (uint8_t)(d4 & 1u);
        }
        if (f4) {
            bx0 = min(bx0, xb + __ffs(f4) - 1); bx1 = max(bx1, xb + 31 - __clz(f4));
            by0 = min(by0, y); by1 = max(by1, y);
        }
    }
    }   // yblk
    wave_box_commit(bx0, by0, bx1, by1, box_acc + 4 * b);
    if (max_phi_bits) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(pmax);  // phi >= 0: bit order == value order
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(bits, o);
            bits = other > bits ? other : bits;
        }
        if (lane == 0 && bits > __hip_atomic_load(&max_phi_bits[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&max_phi_bits[b], bits);
    }
    // The pair's record (box + FoE; im_helpers.py:55-84 gives -1 for an empty mask) is written by the workgroup that draws the
    // pair's last ticket: every workgroup's box atomics have returned (wave_box_commit waits for them) before its ticket is drawn,
    // and the accumulators are read with agent-scope loads.  No separate finalize launch.
    if (done) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned ticket = atomicAdd(&done[2 * b + 1], 1u);
            if (ticket == gridDim.x * gridDim.y - 1) {
                int x0 = __hip_atomic_load(&box_acc[4 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int y0 = __hip_atomic_load(&box_acc[4 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int x1 = __hip_atomic_load(&box_acc[4 * b + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int y1 = __hip_atomic_load(&box_acc[4 * b + 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (x1 < 0) { x0 = y0 = x1 = y1 = -1; }
                if (results) {
                    results[b].box[0] = x0; results[b].box[1] = y0; results[b].box[2] = x1; results[b].box[3] = y1;
                    results[b].foe[0] = foex; results[b].foe[1] = foey;
                }
                if (box_out) { box_out[4 * b] = x0; box_out[4 * b + 1] = y0; box_out[4 * b + 2] = x1; box_out[4 * b + 3] = y1; }
                __hip_atomic_store(&done[2 * b + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// The screen is usable when neither phi nor max(phi) is requested and the thresholds are in the regime where
// "lo" (phi < dyn_a - (dyn_b + dyn_c/mag)) can never fire, the fixed threshold lies in (0, 80) degrees and the dynamic one is >= 0.
static PhiScreen phi_screen(const mav_thr_params& t, const double* phi, const unsigned long long* max_phi_bits, bool screen_on)
{
    PhiScreen s{};
    const bool ok = !phi && !max_phi_bits && t.fixed_deg > 0.0 && t.fixed_deg < 80.0 && t.dyn_a - t.dyn_b <= 0.0 &&
                    t.dyn_c >= 0.0 && t.dyn_a + t.dyn_b >= 0.0 && t.fixed_min_mag > 1e-3 && t.dyn_min_mag > 1e-3 &&
                    t.fixed_min_mag < 1e6 && t.dyn_min_mag < 1e6 && t.dyn_c < 1e6 && t.dyn_a + t.dyn_b < 1e3;
    s.enabled = ok && screen_on;
    s.tan_fixed = ok ? (float)tan(t.fixed_deg * 3.141592653589793238462643383279502884 / 180.0) : 0.f;
    s.margin_fixed = MAV_SCREEN_TAN_MARGIN * (1.f + s.tan_fixed);
    s.fmm2 = (float)(t.fixed_min_mag * t.fixed_min_mag);
    s.dmm2 = (float)(t.dyn_min_mag * t.dyn_min_mag);
    s.dyn_ab = (float)(t.dyn_a + t.dyn_b);
    s.dyn_c = (float)t.dyn_c;
    return s;
}

template <typename FlowT>
static void launch_phi_mask_t(hipStream_t st, const FlowT* flow, const DerotParams* derot, const double* foe, const uint8_t* sky,
                              int B, int W, int H, mav_thr_params thr, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn,
                              int32_t* box_acc, unsigned long long* max_phi_bits, const PhiLaunch& pl)
{
    const PhiScreen scr = phi_screen(thr, phi, max_phi_bits, pl.screen);
    const bool vec = W % 4 == 0 && (((uintptr_t)sky | (uintptr_t)mask_fixed | (uintptr_t)mask_dyn) & 3) == 0;
    // Blocks of 16 rows per workgroup: enough that the whole launch stays near 4096 workgroups.  With one block each (34 816
    // workgroups at 1080p x 64 pairs) the kernel is bound by the rate at which waves can be launched, not by its bytes:
    // measured 0.55 ms, against 0.47 / 0.36 / 0.32 / 0.31 ms with 2 / 4 / 8 / 17 blocks per workgroup (option "phi_yloop" overrides).
    const int nby = (H + 15) / 16, gx = vec ? (W / 4 + 63) / 64 : (W + 63) / 64;
    int yloop = pl.yloop;
    if (yloop < 1) {
        const int want = 4096 / (gx * B > 0 ? gx * B : 1);
        yloop = want > 0 ? (nby + want - 1) / want : nby;
    }
    const dim3 grid(gx, (nby + yloop - 1) / yloop, B);
    auto k = vec ? (derot ? k_phi_mask<FlowT, 4, true> : k_phi_mask<FlowT, 4, false>)
                 : (derot ? k_phi_mask<FlowT, 1, true> : k_phi_mask<FlowT, 1, false>);
    hipLaunchKernelGGL(k, grid, dim3(256), 0, st, flow, derot, foe, sky, W, H, thr, scr, phi, mask_fixed, mask_dyn, box_acc, max_phi_bits,
                       (pl.results || pl.box_out) ? pl.done : (unsigned*)nullptr, pl.results, pl.box_out);
}
void launch_phi_mask_f32(hipStream_t st, const float* flow, const DerotParams* derot, const double* foe, const uint8_t* sky,
                         int B, int W, int H, mav_thr_params thr, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn,
                         int32_t* box_acc, unsigned long long* max_phi_bits, const PhiLaunch& pl)
{
    launch_phi_mask_t<float>(st, flow, derot, foe, sky, B, W, H, thr, phi, mask_fixed, mask_dyn, box_acc, max_phi_bits, pl);
}
void launch_phi_mask_f64(hipStream_t st, const double* flow, const double* foe, const uint8_t* sky, int B, int W, int H,
                         mav_thr_params thr, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, int32_t* box_acc,
                         unsigned long long* max_phi_bits, const PhiLaunch& pl)
{
    launch_phi_mask_t<double>(st, flow, nullptr, foe, sky, B, W, H, thr, phi, mask_fixed, mask_dyn, box_acc, max_phi_bits, pl);
}

__global__ void k_finalize(const int32_t* __restrict__ box_acc, const double* __restrict__ foe, int B, mav_result* __restrict__ res,
                           int32_t* __restrict__ box_only)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int x0 = box_acc[4 * b], y0 = box_acc[4 * b + 1], x1 = box_acc[4 * b + 2], y1 = box_acc[4 * b + 3];
    if (x1 < 0) { x0 = y0 = x1 = y1 = -1; }
    if (res) {
        res[b].box[0] = x0; res[b].box[1] = y0; res[b].box[2] = x1; res[b].box[3] = y1;
        res[b].foe[0] = foe[2 * b]; res[b].foe[1] = foe[2 * b + 1];
    }
    if (box_only) { box_only[4 * b] = x0; box_only[4 * b + 1] = y0; box_only[4 * b + 2] = x1; box_only[4 * b + 3] = y1; }
}
void launch_box_finalize(hipStream_t st, const int32_t* box_acc, int B, int32_t* box)
{
    hipLaunchKernelGGL(k_finalize, dim3((B + 63) / 64), dim3(64), 0, st, box_acc, (const double*)nullptr, B,
                       (mav_result*)nullptr, box);
}

// ------------------------------------------------------------------------------------------------------------
// get_simple_bounding_box on arbitrary u8 images: pass 1 max, pass 2 extents of (double)v > 0.1*max.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_u8_max(const uint8_t* __restrict__ img, int W, int H, int* __restrict__ maxv)
{
    const int b = blockIdx.z;
    const size_t npx = (size_t)W * H;
    const uint8_t* p = img + b * npx;
    int m = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256) m = max(m, (int)p[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&maxv[b], m);
}
__global__ __launch_bounds__(256) void k_u8_extents(const uint8_t* __restrict__ img, int W, int H, const int* __restrict__ maxv,
                                                    int32_t* __restrict__ box_acc)
{
    const int b = blockIdx.z;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const double thr = 0.1 * (double)maxv[b];
    int bx0 = INT_MAX, by0 = INT_MAX, bx1 = -1, by1 = -1;
    if (y < H)
        for (int x = blockIdx.x * 256 + (threadIdx.x & 63); x < min(W, (int)(blockIdx.x + 1) * 256); x += 64)
            if ((double)img[(size_t)b * W * H + (size_t)y * W + x] > thr) {
                bx0 = min(bx0, x); bx1 = max(bx1, x); by0 = min(by0, y); by1 = max(by1, y);
            }
    wave_box_commit(bx0, by0, bx1, by1, box_acc + 4 * b);
}
void launch_bbox_u8(hipStream_t st, const uint8_t* img, int B, int W, int H, int* maxv, int32_t* box_acc)
{
    hipMemsetAsync(maxv, 0, sizeof(int) * B, st);
    hipLaunchKernelGGL(k_u8_max, dim3(64, 1, B), dim3(256), 0, st, img, W, H, maxv);
    hipLaunchKernelGGL(k_u8_extents, dim3((W + 255) / 256, (H + 3) / 4, B), dim3(256), 0, st, img, W, H, maxv, box_acc);
}

// ------------------------------------------------------------------------------------------------------------
// analyze_pyramid level 0: 64x64 windows, stride 16, score = 3 * sum(u8) (the to_rgb replica has 3 equal channels);
// first window in row-major scan order with the strictly largest score.  One workgroup per window row.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window_max(const uint8_t* __restrict__ img, int W, int H, int nwx,
                                                    unsigned long long* __restrict__ key)
{
    extern __shared__ int colsum[];
    const int b = blockIdx.y, wy = blockIdx.x, tid = threadIdx.x;
    const uint8_t* p = img + (size_t)b * W * H + (size_t)wy * 16 * W;
    for (int x = tid; x < W; x += 256) {
        int s = 0;
        for (int r = 0; r < 64; r++) s += p[(size_t)r * W + x];
        colsum[x] = s;
    }
    __syncthreads();
    unsigned long long k = 0ull;
    for (int wx = tid; wx < nwx; wx += 256) {
        int s = 0;
        for (int c = 0; c < 64; c++) s += colsum[wx * 16 + c];
        const unsigned score = 3u * (unsigned)s;
        const unsigned idx = (unsigned)(wy * nwx + wx);
        if (score) {
            const unsigned long long kk = ((unsigned long long)score << 32) | (0xFFFFFFFFu - idx);
            k = kk > k ? kk : k;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(k, o);
        k = other > k ? other : k;
    }
    if ((tid & 63) == 0 && k) atomicMax(&key[b], k);
}
__global__ void k_window_finalize(const unsigned long long* __restrict__ key, int nwx, int B, int64_t* __restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned long long k = key[b];
    int64_t score = 0, x = 0, y = 0;
    if (k) {
        const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
        score = (int64_t)(k >> 32);
        x = (int64_t)(idx % nwx) * 16;
        y = (int64_t)(idx / nwx) * 16;
    }
    out[3 * b] = score; out[3 * b + 1] = x; out[3 * b + 2] = y;
}
void launch_window_max(hipStream_t st, const uint8_t* img, int B, int W, int H, unsigned long long* key, int64_t* out)
{
    hipMemsetAsync(key, 0, sizeof(unsigned long long) * B, st);
    const int nwx = W >= 64 ? (W - 64) / 16 + 1 : 0, nwy = H >= 64 ? (H - 64) / 16 + 1 : 0;
    if (nwx > 0 && nwy > 0)
        hipLaunchKernelGGL(k_window_max, dim3(nwy, B), dim3(256), sizeof(int) * (size_t)W, st, img, W, H, nwx, key);
    hipLaunchKernelGGL(k_window_finalize, dim3((B + 63) / 64), dim3(64), 0, st, key, nwx > 0 ? nwx : 1, B, out);
}

// calculate_tpr_fpr counts for gt (u8) against mask_value * mask: the reference's products in wide integers (no u8 wrap).
// One pass over the ground truth serves one mask or two (the fixed and the dynamic mask of a detection call); gt_stride = 0: one
// ground-truth image for every pair of the batch (a constant segmentation).  16 bytes per thread and load where the images are
// 16-byte addressable, byte by byte otherwise; the counts are integers, so the order of the additions is immaterial.
__device__ __forceinline__ void tpr_fpr_byte(unsigned gv, unsigned m0, unsigned m1, unsigned mask_value, unsigned& pos, unsigned& neg, unsigned& tp0,
                                             unsigned& fp0, unsigned& tp1, unsigned& fp1)
{
    const unsigned v0 = m0 ? mask_value : 0u, v1 = m1 ? mask_value : 0u;
    pos += gv > 127u;
    neg += (255u - gv) > 127u;
    tp0 += (gv * v0) > 127u;
    fp0 += ((255u - gv) * v0) > 127u;
    tp1 += (gv * v1) > 127u;
    fp1 += ((255u - gv) * v1) > 127u;
}
template <bool TWO, bool VEC>
__global__ __launch_bounds__(256) void k_tpr_fpr(const uint8_t* __restrict__ gt, size_t gt_stride, const uint8_t* __restrict__ mask0,
                                                 const uint8_t* __restrict__ mask1, unsigned mask_value, size_t npx,
                                                 unsigned long long* __restrict__ counts0, unsigned long long* __restrict__ counts1)
{
    const int b = blockIdx.y;
    const uint8_t* g = gt + b * gt_stride;
    const uint8_t* m0 = mask0 + b * npx;
    const uint8_t* m1 = TWO ? mask1 + b * npx : m0;
    unsigned pos = 0, neg = 0, tp0 = 0, fp0 = 0, tp1 = 0, fp1 = 0;
    if (VEC) {
        const size_t nv = npx / 16;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
            const uint4 gv = ((const uint4*)g)[i], a = ((const uint4*)m0)[i], c = TWO ? ((const uint4*)m1)[i] : a;
            const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w}, aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int k = 0; k < 4; k++)
                    tpr_fpr_byte((gw[w] >> (8 * k)) & 255u, (aw[w] >> (8 * k)) & 255u, (cw[w] >> (8 * k)) & 255u, mask_value, pos, neg, tp0, fp0, tp1, fp1);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256)
            tpr_fpr_byte(g[i], m0[i], TWO ? m1[i] : 0u, mask_value, pos, neg, tp0, fp0, tp1, fp1);
    }
    for (int o = 32; o > 0; o >>= 1) {
        pos += __shfl_xor(pos, o); neg += __shfl_xor(neg, o); tp0 += __shfl_xor(tp0, o); fp0 += __shfl_xor(fp0, o);
        if (TWO) { tp1 += __shfl_xor(tp1, o); fp1 += __shfl_xor(fp1, o); }
    }
    // one set of atomics per WORKGROUP (round 6): with one per wave a single 1280x720 pair issued 4 096 64-bit atomics onto eight
    // addresses and the kernel took 35 us for 2.8 MB -- all of it the atomics' serialisation
    __shared__ unsigned part[4][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[wave][0] = pos; part[wave][1] = neg; part[wave][2] = tp0; part[wave][3] = fp0; part[wave][4] = tp1; part[wave][5] = fp1;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const unsigned long long v = (unsigned long long)part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        const int k = threadIdx.x;                 // 0 pos, 1 neg, 2 tp0, 3 fp0, 4 tp1, 5 fp1
        if (k < 4) atomicAdd(&counts0[4 * b + k], v);
        if (TWO) {
            if (k < 2) atomicAdd(&counts1[4 * b + k], v);
            if (k >= 4) atomicAdd(&counts1[4 * b + k - 2], v);
        }
    }
}
void launch_tpr_fpr2(hipStream_t st, const uint8_t* gt, size_t gt_stride, const uint8_t* mask0, const uint8_t* mask1, unsigned mask_value, int B,
                     int W, int H, unsigned long long* counts0, unsigned long long* counts1)
{
    const size_t npx = (size_t)W * H;
    if (mask1 && counts1 == counts0 + 4 * (size_t)B) {           // the two count blocks follow each other (the fused step, full batch): one fill
        hipMemsetAsync(counts0, 0, sizeof(unsigned long long) * 8 * B, st);
    } else {
        hipMemsetAsync(counts0, 0, sizeof(unsigned long long) * 4 * B, st);
        if (mask1) hipMemsetAsync(counts1, 0, sizeof(unsigned long long) * 4 * B, st);
    }
    const bool vec = npx % 16 == 0 && gt_stride % 16 == 0 && (((uintptr_t)gt | (uintptr_t)mask0 | (uintptr_t)mask1) & 15) == 0;
    const dim3 grid(128, B), blk(256);
    if (mask1) {
        if (vec) hipLaunchKernelGGL((k_tpr_fpr<true, true>), grid, blk, 0, st, gt, gt_stride, mask0, mask1, mask_value, npx, counts0, counts1);
        else hipLaunchKernelGGL((k_tpr_fpr<true, false>), grid, blk, 0, st, gt, gt_stride, mask0, mask1, mask_value, npx, counts0, counts1);
    } else {
        if (vec) hipLaunchKernelGGL((k_tpr_fpr<false, true>), grid, blk, 0, st, gt, gt_stride, mask0, mask1, mask_value, npx, counts0, counts1);
        else hipLaunchKernelGGL((k_tpr_fpr<false, false>), grid, blk, 0, st, gt, gt_stride, mask0, mask1, mask_value, npx, counts0, counts1);
    }
}
void launch_tpr_fpr(hipStream_t st, const uint8_t* gt, const uint8_t* mask, unsigned mask_value, int B, int W, int H,
                    unsigned long long* counts)
{
    launch_tpr_fpr2(st, gt, (size_t)W * H, mask, nullptr, mask_value, B, W, H, counts, nullptr);
}

// cv2.cvtColor(COLOR_BGR2GRAY) on u8 (farneback.py:21,74): fixed point, (B*1868 + G*9617 + R*4899 + 8192) >> 14  (SURVEY A.7).
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t* __restrict__ bgr, size_t n, uint8_t* __restrict__ gray)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint8_t* p = bgr + 3 * i;
        gray[i] = (uint8_t)((p[0] * 1868u + p[1] * 9617u + p[2] * 4899u + 8192u) >> 14);
    }
}
void launch_bgr2gray(hipStream_t st, const uint8_t* bgr, size_t n, uint8_t* gray)
{
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, st, bgr, n, gray);
}

// FocusOfExpansion.ransac on caller-supplied estimates (already in s.cand[0..M), s.count[0] = M): vote + pick.
__global__ void k_set_count(FoeScratch sc, int M)
{
    sc.count[0] = M;
    sc.best_key[0] = 0ull;
    sc.done[0] = 0u;
}
void launch_ransac_only(hipStream_t st, FoeScratch s, int M, int N, double dist2_thr, double* foe)
{
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, s, M);
    hipLaunchKernelGGL(k_ransac, dim3(((M > 0 ? M : 1) + 15) / 16, 1), dim3(256), 0, st, s, N, dist2_thr, foe);
}

// DerotParams from device-resident omega (B,3) and dt (B, nullable = 1): sx = w*dt/2, sy = h*dt/2 (detector.py:101-102).
__global__ void k_make_derot(const double* __restrict__ omega, const double* __restrict__ dt, const uint8_t* __restrict__ frame0,
                             int B, int W, int H, DerotParams* __restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double d = dt ? dt[b] : 1.0;
    DerotParams p;
    p.o0 = omega ? omega[3 * b] : 0.0; p.o1 = omega ? omega[3 * b + 1] : 0.0; p.o2 = omega ? omega[3 * b + 2] : 0.0;
    p.sx = (double)W * d / 2;
    p.sy = (double)H * d / 2;
    p.mode = (frame0 && frame0[b]) ? MAV_PAIR_FRAME0 : (omega ? MAV_PAIR_DEROTATE : MAV_PAIR_PROMOTE);
    out[b] = p;
}
void launch_make_derot(hipStream_t st, const double* omega, const double* dt, const uint8_t* frame0, int B, int W, int H,
                       DerotParams* out)
{
    hipLaunchKernelGGL(k_make_derot, dim3((B + 63) / 64), dim3(64), 0, st, omega, dt, frame0, B, W, H, out);
}
