// Window search of Detector: the image pyramid of analyze_pyramid (imutils.resize ->
// cv2.resize INTER_AREA, scale 1.5), the 64x64 / stride-16 window scan over every level, and optimize_window.
// gfx950, wave64.  Compiled with -ffp-contract=off: the area resize accumulates in float in OpenCV's tap order.
// u8 byte work on images that shrink by 2.25x per level: HBM/latency bound, nothing here is hot.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mavflow_internal.h"

// ---- cv2.resize(INTER_AREA), non-integer ratio: computeResizeAreaTab + resizeArea_<uchar, float> (resize.cpp) ---------
// One destination index d of one axis covers source cells [d*scale, d*scale + scale): an optional partial first cell
// (s1 - 1, weight wf), whole cells s1 .. s2-1 (weight wm each) and an optional partial last cell (s2, weight wl).
struct AreaSpan { int s1, s2; float wf, wm, wl; bool first, last; };
__device__ __forceinline__ AreaSpan area_span(int d, double scale, int ssize)
{
    AreaSpan a;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    int s1 = (int)ceil(f1), s2 = (int)floor(f2);
    s2 = s2 < ssize - 1 ? s2 : ssize - 1;
    s1 = s1 < s2 ? s1 : s2;
    a.s1 = s1; a.s2 = s2;
    a.first = (double)s1 - f1 > 1e-3;
    a.last = f2 - (double)s2 > 1e-3;
    a.wf = (float)(((double)s1 - f1) / cell);
    a.wm = (float)(1.0 / cell);
    a.wl = (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell);
    return a;
}
__device__ __forceinline__ float area_row(const uint8_t* __restrict__ S, const AreaSpan& x)
{
    float buf = 0.f;                                   // buf[dx] += S[si] * alpha, in tab order
    if (x.first) buf = buf + (float)S[x.s1 - 1] * x.wf;
    for (int sx = x.s1; sx < x.s2; sx++) buf = buf + (float)S[sx] * x.wm;
    if (x.last) buf = buf + (float)S[x.s2] * x.wl;
    return buf;
}
__global__ __launch_bounds__(256) void k_area_resize(const uint8_t* __restrict__ src, size_t src_stride, int sw, int sh,
                                                     uint8_t* __restrict__ dst, size_t dst_stride, int dw, int dh,
                                                     double scale_x, double scale_y)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dw || dy >= dh) return;
    const uint8_t* S = src + blockIdx.z * src_stride;
    const AreaSpan x = area_span(dx, scale_x, sw), y = area_span(dy, scale_y, sh);
    float sum = 0.f;                                   // sum[dx] += beta * buf[dx], rows in tab order
    if (y.first) sum = sum + y.wf * area_row(S + (size_t)(y.s1 - 1) * sw, x);
    for (int sy = y.s1; sy < y.s2; sy++) sum = sum + y.wm * area_row(S + (size_t)sy * sw, x);
    if (y.last) sum = sum + y.wl * area_row(S + (size_t)y.s2 * sw, x);
    const float r = rintf(sum);                        // saturate_cast<uchar>(float): round half to even, clamp
    dst[blockIdx.z * dst_stride + (size_t)dy * dw + dx] = (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}
void launch_area_resize(hipStream_t st, const uint8_t* src, size_t src_stride, int sw, int sh, uint8_t* dst, size_t dst_stride,
                        int dw, int dh, int B)
{
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);   // hal::resize: 1. / inv_scale
    hipLaunchKernelGGL(k_area_resize, dim3((dw + 63) / 64, (dh + 3) / 4, B), dim3(256), 0, st, src, src_stride, sw, sh, dst,
                       dst_stride, dw, dh, scale_x, scale_y);
}

// ---- analyze_pyramid: the window scan of one level (detector.py:296-310) -----------------------------------------------
// One workgroup per window row: 64-row column sums in LDS, then 64-column sums per window.  key[b] keeps
// (score << 32) | ~index with index = position in the reference's scan order (levels, then rows, then columns), so the
// 64-bit maximum is "strictly larger score, first one wins".  Scores of 0 never enter (result[0] < score with 0 start).
__global__ __launch_bounds__(256) void k_level_scan(const uint8_t* __restrict__ img, size_t stride, int W, int H, int nwx,
                                                    unsigned idx_base, unsigned long long* __restrict__ key)
{
    extern __shared__ int colsum[];
    const int b = blockIdx.y, wy = blockIdx.x, tid = threadIdx.x;
    const uint8_t* p = img + (size_t)b * stride + (size_t)wy * 16 * W;
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
        const unsigned idx = idx_base + (unsigned)(wy * nwx + wx);
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
// Decode the winner and find np.unravel_index(window.argmax(), window.shape)[:2] inside it.  One workgroup per image.
__global__ __launch_bounds__(256) void k_pyramid_finalize(const unsigned long long* __restrict__ key, PyrPlan plan,
                                                          const uint8_t* __restrict__ img0, const uint8_t* __restrict__ ws,
                                                          int64_t* __restrict__ out)
{
    __shared__ unsigned red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned long long k = key[b];
    if (!k) {
        if (tid < 6) out[6 * b + tid] = 0;
        return;
    }
    const unsigned idx = 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
    int lv = 0;
    while (lv + 1 < plan.n && idx >= plan.base[lv + 1]) lv++;
    const int W = plan.w[lv], H = plan.h[lv];
    const int nwx = (W - 64) / 16 + 1;
    const unsigned local = idx - plan.base[lv];
    const int x = (int)(local % nwx) * 16, y = (int)(local / nwx) * 16;
    const uint8_t* im = (lv == 0 ? img0 : ws + plan.off[lv]) + (size_t)b * W * H;
    unsigned best = 0;                                  // (value << 12) | (4095 - position): first maximum in row-major order
    for (int p = tid; p < 64 * 64; p += 256) {
        const unsigned v = im[(size_t)(y + (p >> 6)) * W + x + (p & 63)];
        const unsigned kk = (v << 12) | (4095u - (unsigned)p);
        best = kk > best ? kk : best;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    if ((tid & 63) == 0) red[tid >> 6] = best;
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 4; i++) best = red[i] > best ? red[i] : best;
        const int pos = 4095 - (int)(best & 4095u);
        out[6 * b] = (int64_t)(k >> 32); out[6 * b + 1] = x; out[6 * b + 2] = y; out[6 * b + 3] = lv;
        out[6 * b + 4] = pos >> 6; out[6 * b + 5] = pos & 63;
    }
}
void launch_level_scan(hipStream_t st, const uint8_t* img, size_t stride, int B, int W, int H, unsigned idx_base,
                       unsigned long long* key)
{
    const int nwx = W >= 64 ? (W - 64) / 16 + 1 : 0, nwy = H >= 64 ? (H - 64) / 16 + 1 : 0;
    if (nwx > 0 && nwy > 0)
        hipLaunchKernelGGL(k_level_scan, dim3(nwy, B), dim3(256), sizeof(int) * (size_t)W, st, img, stride, W, H, nwx, idx_base, key);
}
void launch_pyramid_finalize(hipStream_t st, const unsigned long long* key, const PyrPlan& plan, const uint8_t* img0,
                             const uint8_t* ws, int B, int64_t* out)
{
    hipLaunchKernelGGL(k_pyramid_finalize, dim3(B), dim3(256), 0, st, key, plan, img0, ws, out);
}

// ---- optimize_window (detector.py:314-358) ---------------------------------------------------------------------------
// Summed-area table sat[(H+1) x (W+1)] of u64 (row 0 / column 0 are zero), then a greedy walk that needs 8 box sums per
// step.  The walk is inherently serial (each step depends on the previous winner): one wave per image, lane = candidate.
__global__ __launch_bounds__(256) void k_sat_rows(const uint8_t* __restrict__ img, int W, int H, unsigned long long* __restrict__ sat)
{
    __shared__ unsigned part[256];
    const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const uint8_t* row = img + ((size_t)b * H + y) * W;
    unsigned long long* out = sat + (size_t)b * (H + 1) * (W + 1) + (size_t)(y + 1) * (W + 1);
    const int chunk = (W + 255) / 256, x0 = tid * chunk, x1 = x0 + chunk < W ? x0 + chunk : W;
    unsigned s = 0;
    for (int x = x0; x < x1; x++) s += row[x];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                 // inclusive scan of the 256 chunk sums
        const unsigned v = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned run = tid ? part[tid - 1] : 0u;
    for (int x = x0; x < x1; x++) { run += row[x]; out[x + 1] = run; }
    if (tid == 0) out[0] = 0ull;
    if (y == 0)
        for (int x = tid; x <= W; x += 256) sat[(size_t)b * (H + 1) * (W + 1) + x] = 0ull;
}
__global__ __launch_bounds__(256) void k_sat_cols(int W, int H, unsigned long long* __restrict__ sat)
{
    const int x = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (x > W) return;
    unsigned long long* p = sat + (size_t)b * (H + 1) * (W + 1) + x;
    unsigned long long run = 0ull;
    for (int y = 1; y <= H; y++) { run += p[(size_t)y * (W + 1)]; p[(size_t)y * (W + 1)] = run; }
}
// Python slice index: negative wraps once, then clips to [0, n].
__device__ __forceinline__ int slice_index(int a, int n)
{
    if (a < 0) a += n;
    return a < 0 ? 0 : (a > n ? n : a);
}
__global__ __launch_bounds__(64) void k_optimize_window(const unsigned long long* __restrict__ sat, int W, int H,
                                                        const int32_t* __restrict__ win_in, int64_t* __restrict__ score_out,
                                                        int32_t* __restrict__ win_out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const unsigned long long* S = sat + (size_t)b * (H + 1) * (W + 1);
    int l = win_in[4 * b], t = win_in[4 * b + 1], r = l + win_in[4 * b + 2], bt = t + win_in[4 * b + 3];
    unsigned long long res = 0ull;
    // candidate order of the reference: h = 0 moves the top-left corner by (i, j) in (-1,-1), (-1,1), (1,-1), (1,1); h = 1 the
    // bottom-right corner.  Lanes 8..63 mirror lanes 0..7 (same values), so the xor reduction over 1, 2, 4 is enough.
    const int c = lane & 7, di = (c & 2) ? 1 : -1, dj = (c & 1) ? 1 : -1;
    const int max_steps = 4 * (W + H) + 64;             // every step strictly raises a sum bounded by the image: always reached early
    for (int step = 0; step < max_steps; step++) {
        const int cl = c < 4 ? l + di : l, ct = c < 4 ? t + dj : t, cr = c < 4 ? r : r + di, cb = c < 4 ? bt : bt + dj;
        const int y0 = slice_index(ct, H), y1 = slice_index(cb, H), x0 = slice_index(cl, W), x1 = slice_index(cr, W);
        unsigned long long s = 0ull;
        if (y0 < y1 && x0 < x1)
            s = 3ull * (S[(size_t)y1 * (W + 1) + x1] - S[(size_t)y0 * (W + 1) + x1] - S[(size_t)y1 * (W + 1) + x0] + S[(size_t)y0 * (W + 1) + x0]);
        unsigned long long k = (s << 3) | (unsigned long long)(7 - c);          // strictly larger score, first candidate wins
        for (int o = 1; o < 8; o <<= 1) {
            const unsigned long long other = __shfl_xor(k, o);
            k = other > k ? other : k;
        }
        const unsigned long long best = k >> 3;
        if (best <= res) break;                          // uniform across the wave
        res = best;
        const int w = 7 - (int)(k & 7ull), wi = (w & 2) ? 1 : -1, wj = (w & 1) ? 1 : -1;
        if (w < 4) { l += wi; t += wj; } else { r += wi; bt += wj; }
    }
    if (lane == 0) {
        score_out[b] = (int64_t)res;
        win_out[4 * b] = l; win_out[4 * b + 1] = t; win_out[4 * b + 2] = r - l; win_out[4 * b + 3] = bt - t;
    }
}
void launch_optimize_window(hipStream_t st, const uint8_t* img, int B, int W, int H, unsigned long long* sat, const int32_t* win_in,
                            int64_t* score, int32_t* win_out)
{
    hipLaunchKernelGGL(k_sat_rows, dim3(H, B), dim3(256), 0, st, img, W, H, sat);
    hipLaunchKernelGGL(k_sat_cols, dim3((W + 1 + 255) / 256, B), dim3(256), 0, st, W, H, sat);
    hipLaunchKernelGGL(k_optimize_window, dim3(B), dim3(64), 0, st, sat, W, H, win_in, score, win_out);
}
