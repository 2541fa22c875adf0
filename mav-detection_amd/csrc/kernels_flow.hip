// Dense-flow kernels for gfx950 (MI355X): Gaussian pyramid layer, polynomial expansion, UpdateMatrices and the
// fused {box blur -> 2x2 solve -> UpdateMatrices} iteration of Farneback's algorithm.
//
// What they compute is cv2.calcOpticalFlowFarneback as called at /root/reference/src/farneback.py:76-80
// (OpenCV modules/video/src/optflowgf.cpp).  How they compute it is new: R and M live
// in HBM as 5 separate f32 planes per pair (coalesced 128-B row segments per wave), every stencil stage stages
// its tile + halo through LDS once, box sums are separable sliding sums done in place in LDS, and the solve and
// the next UpdateMatrices are fused behind the blur so M makes exactly one HBM round trip per iteration.
// The path is HBM/LDS-bound stencil work: no MFMA.
#include "mavflow_internal.h"

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------------------------------------------------
// Layer image: convertTo(f32) -> GaussianBlur(ksize, sigma, REFLECT_101) -> resize(INTER_LINEAR), fused through
// the host-built 1-D tap tables (A.2).  One thread per output pixel; u8 source rows are re-read through L1/L2.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blur_resize(const uint8_t* __restrict__ img, size_t img_stride, int W, int H,
                                                     int w, int h, ResizeTables t, float* __restrict__ out,
                                                     size_t out_stride)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= w || dy >= h) return;
    const uint8_t* src = img + (size_t)blockIdx.z * img_stride;
    const int* xi = t.xi + (size_t)dx * t.taps;
    const float* xw = t.xw + (size_t)dx * t.taps;
    const int* yi = t.yi + (size_t)dy * t.taps;
    const float* yw = t.yw + (size_t)dy * t.taps;
    float acc = 0.f;
    for (int ty = 0; ty < t.taps; ty++) {
        const float wy = yw[ty];
        if (wy == 0.f) continue;
        const uint8_t* row = src + (size_t)yi[ty] * W;
        float r = 0.f;
        for (int tx = 0; tx < t.taps; tx++) r += xw[tx] * (float)row[xi[tx]];
        acc += wy * r;
    }
    out[(size_t)blockIdx.z * out_stride + (size_t)dy * w + dx] = acc;
}

void launch_blur_resize(hipStream_t st, const uint8_t* img, size_t img_stride, int G, int W, int H, int w, int h,
                        ResizeTables t, float* out, size_t out_stride)
{
    dim3 grid((w + 63) / 64, (h + 3) / 4, G);
    hipLaunchKernelGGL(k_blur_resize, grid, dim3(256), 0, st, img, img_stride, W, H, w, h, t, out, out_stride);
}

// ------------------------------------------------------------------------------------------------------------
// FarnebackPolyExp (A.4).  64x16 output tile per workgroup; the (64+2n) x (16+2n) source tile (edge-clamped,
// which is exactly OpenCV's row clamp + edge-triple replication) is staged in LDS, the vertical pass leaves
// three (64+2n) x 16 planes in LDS and the horizontal pass reads those.  Coefficients sit in kernel arguments
// (scalar loads).  Output: 5 planes.
// ------------------------------------------------------------------------------------------------------------
#define PX 64
#define PY 16
template <int N_T>
__global__ __launch_bounds__(256) void k_polyexp(const float* __restrict__ I, size_t I_stride, int w, int h, PolyCoef pc,
                                                 float* __restrict__ R, size_t R_stride)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = N_T > 0 ? N_T : pc.n;
    const int EX = PX + 2 * n, EY = PY + 2 * n;
    float* tile = smem;
    float* v0 = tile + EX * EY;
    float* v1 = v0 + PY * EX;
    float* v2 = v1 + PY * EX;
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * PX, y0 = blockIdx.y * PY;
    const float* src = I + (size_t)blockIdx.z * I_stride;

    for (int i = tid; i < EX * EY; i += 256) {
        const int ly = i / EX, lx = i - ly * EX;
        const int gx = clampi(x0 - n + lx, 0, w - 1), gy = clampi(y0 - n + ly, 0, h - 1);
        tile[i] = src[(size_t)gy * w + gx];
    }
    __syncthreads();
    for (int i = tid; i < PY * EX; i += 256) {
        const int ly = i / EX, lx = i - ly * EX;
        const float* c = tile + (ly + n) * EX + lx;
        float t0 = c[0] * pc.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 1; k <= n; k++) {
            const float a = c[-k * EX], b = c[k * EX];
            const float p = a + b;
            t0 += pc.g[k] * p;
            t1 += pc.xg[k] * (b - a);
            t2 += pc.xxg[k] * p;
        }
        v0[i] = t0; v1[i] = t1; v2[i] = t2;
    }
    __syncthreads();
    const size_t npx = (size_t)w * h;
    float* dst = R + (size_t)blockIdx.z * R_stride;
    for (int i = tid; i < PY * PX; i += 256) {
        const int ly = i >> 6, lx = i & 63;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= w || gy >= h) continue;
        const float* p0 = v0 + ly * EX + lx + n;
        const float* p1 = v1 + ly * EX + lx + n;
        const float* p2 = v2 + ly * EX + lx + n;
        float b1 = p0[0] * pc.g[0], b2 = 0.f, b3 = p1[0] * pc.g[0], b4 = 0.f, b5 = p2[0] * pc.g[0], b6 = 0.f;
#pragma unroll
        for (int k = 1; k <= n; k++) {
            const float a0 = p0[-k], c0 = p0[k], a1 = p1[-k], c1 = p1[k], a2 = p2[-k], c2 = p2[k];
            const float tg = c0 + a0;
            b1 += tg * pc.g[k];
            b4 += tg * pc.xxg[k];
            b2 += (c0 - a0) * pc.xg[k];
            b3 += (c1 + a1) * pc.g[k];
            b6 += (c1 - a1) * pc.xg[k];
            b5 += (c2 + a2) * pc.g[k];
        }
        const size_t o = (size_t)gy * w + gx;
        dst[o] = b3 * pc.ig11;
        dst[npx + o] = b2 * pc.ig11;
        dst[2 * npx + o] = b1 * pc.ig03 + b5 * pc.ig33;
        dst[3 * npx + o] = b1 * pc.ig03 + b4 * pc.ig33;
        dst[4 * npx + o] = b6 * pc.ig55;
    }
}

void launch_polyexp(hipStream_t st, const float* I, size_t I_stride, int G, int w, int h, const PolyCoef& pc, float* R,
                    size_t R_stride)
{
    const int n = pc.n;
    const size_t lds = sizeof(float) * ((size_t)(PX + 2 * n) * (PY + 2 * n) + 3 * (size_t)PY * (PX + 2 * n));
    dim3 grid((w + PX - 1) / PX, (h + PY - 1) / PY, G);
    if (n == 8)
        hipLaunchKernelGGL(k_polyexp<8>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, R, R_stride);
    else if (n == 7)
        hipLaunchKernelGGL(k_polyexp<7>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, R, R_stride);
    else if (n == 5)
        hipLaunchKernelGGL(k_polyexp<5>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, R, R_stride);
    else
        hipLaunchKernelGGL(k_polyexp<0>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, R, R_stride);
}

// ------------------------------------------------------------------------------------------------------------
// FarnebackUpdateMatrices for one pixel (A.5): bilinear gather of the 5 R1 planes at (x+u, y+v), border damping,
// the five products.  R0p/R1p point at plane 0 of the pair; planes are npx apart.
// ------------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void update_px(const float* __restrict__ R0p, const float* __restrict__ R1p, size_t npx,
                                                 int w, int h, int x, int y, float dx, float dy, float* __restrict__ Mp)
{
    const size_t idx = (size_t)y * w + x;
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= x1; fy -= y1;
    float r2, r3, r4, r5, r6;
    const float q0 = R0p[idx], q1 = R0p[npx + idx], q2 = R0p[2 * npx + idx], q3 = R0p[3 * npx + idx],
                q4 = R0p[4 * npx + idx];
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const float* p = R1p + (size_t)y1 * w + x1;
        r2 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r3 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r4 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r5 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1]; p += npx;
        r6 = a00 * p[0] + a01 * p[1] + a10 * p[w] + a11 * p[w + 1];
        r4 = (q2 + r4) * 0.5f;
        r5 = (q3 + r5) * 0.5f;
        r6 = (q4 + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = q2; r5 = q3; r6 = q4 * 0.5f;
    }
    r2 = (q0 - r2) * 0.5f;
    r3 = (q1 - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    const int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472}
        auto bw = [](int d) { return d < 2 ? 0.14f : 0.4472f; };
        const float scale = (x < BORDER ? bw(x) : 1.f) * (x >= w - BORDER ? bw(w - x - 1) : 1.f) *
                            (y < BORDER ? bw(y) : 1.f) * (y >= h - BORDER ? bw(h - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    Mp[idx] = r4 * r4 + r6 * r6;
    Mp[npx + idx] = (r4 + r5) * r6;
    Mp[2 * npx + idx] = r5 * r5 + r6 * r6;
    Mp[3 * npx + idx] = r4 * r2 + r6 * r3;
    Mp[4 * npx + idx] = r6 * r2 + r5 * r3;
}

// Initial M of a layer.  flow = 0 (top layer), resize(prevFlow)*mul evaluated inline (lower layers), or an
// explicit flow field (stage hook).  The upsampled flow is never written: the first blur sweep overwrites it.
template <int MODE>  // 0 zero, 1 upsample, 2 explicit
__global__ __launch_bounds__(256) void k_update_matrices(const float* __restrict__ R0, const float* __restrict__ R1,
                                                         size_t R_stride, const float* __restrict__ fsrc, size_t f_stride,
                                                         int pw, int ph, float mul, double scale_x, double scale_y, int w,
                                                         int h, float* __restrict__ M, size_t M_stride)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int s = blockIdx.z;
    float dx = 0.f, dy = 0.f;
    if (MODE == 1) {
        const float* pf = fsrc + (size_t)s * f_stride;
        float fy = (float)((y + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        if (sy < 0) { fy = 0.f; sy = 0; }
        if (sy >= ph - 1) { fy = 0.f; sy = ph - 1; }
        const int sy1 = sy + 1 < ph ? sy + 1 : sy;
        float fx = (float)((x + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= pw - 1) { fx = 0.f; sx = pw - 1; }
        const int sx1 = sx + 1 < pw ? sx + 1 : sx;
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const float2 p00 = *(const float2*)(pf + ((size_t)sy * pw + sx) * 2);
        const float2 p01 = *(const float2*)(pf + ((size_t)sy * pw + sx1) * 2);
        const float2 p10 = *(const float2*)(pf + ((size_t)sy1 * pw + sx) * 2);
        const float2 p11 = *(const float2*)(pf + ((size_t)sy1 * pw + sx1) * 2);
        dx = ((p00.x * a0 + p01.x * a1) * b0 + (p10.x * a0 + p11.x * a1) * b1) * mul;
        dy = ((p00.y * a0 + p01.y * a1) * b0 + (p10.y * a0 + p11.y * a1) * b1) * mul;
    } else if (MODE == 2) {
        const float2 f = *(const float2*)(fsrc + (size_t)s * f_stride + ((size_t)y * w + x) * 2);
        dx = f.x; dy = f.y;
    }
    update_px(R0 + (size_t)s * R_stride, R1 + (size_t)s * R_stride, (size_t)w * h, w, h, x, y, dx, dy,
              M + (size_t)s * M_stride);
}

void launch_update_matrices(hipStream_t st, const float* R0, const float* R1, size_t R_stride, const float* flow_prev,
                            size_t fp_stride, int pw, int ph, float mul, int G, int w, int h, float* M, size_t M_stride)
{
    dim3 grid((w + 63) / 64, (h + 3) / 4, G);
    if (flow_prev)
        hipLaunchKernelGGL(k_update_matrices<1>, grid, dim3(256), 0, st, R0, R1, R_stride, flow_prev, fp_stride, pw, ph, mul,
                           (double)pw / w, (double)ph / h, w, h, M, M_stride);
    else
        hipLaunchKernelGGL(k_update_matrices<0>, grid, dim3(256), 0, st, R0, R1, R_stride, (const float*)nullptr, (size_t)0,
                           0, 0, 0.f, 0.0, 0.0, w, h, M, M_stride);
}

void launch_update_matrices_flow(hipStream_t st, const float* R0, const float* R1, size_t R_stride, const float* flow,
                                 size_t f_stride, int G, int w, int h, float* M, size_t M_stride)
{
    dim3 grid((w + 63) / 64, (h + 3) / 4, G);
    hipLaunchKernelGGL(k_update_matrices<2>, grid, dim3(256), 0, st, R0, R1, R_stride, flow, f_stride, 0, 0, 0.f, 0.0, 0.0,
                       w, h, M, M_stride);
}

// ------------------------------------------------------------------------------------------------------------
// One FarnebackUpdateFlow_Blur sweep (A.6), fused: (2m+1)^2 box sum of the 5 M planes -> 2x2 solve -> flow, and
// (all sweeps but the last) UpdateMatrices with the new flow -> M'.  32x32 pixel tile per 256-thread workgroup.
//   phase 1  M tile + m-pixel halo (edge-clamped = replicate border) -> LDS, 5 planes, odd row pitch
//   phase 2  vertical sliding sums, one thread per (plane, column), written back in place
//   phase 3  horizontal sliding sums, one thread per (plane, row), in place
//   phase 4  per pixel: scale, solve, store flow; gather R1, store M'
// LDS for winsize 12: 5 x 1996 floats = 39.9 KB -> 4 workgroups (16 waves) per CU.
// ------------------------------------------------------------------------------------------------------------
static inline void iter_geometry(int m, int* ext, int* pitch, int* plane)
{
    *ext = MAV_TILE + 2 * m;
    *pitch = (*ext & 1) ? *ext : *ext + 1;          // odd pitch: phase 3 lanes (rows) hit distinct banks
    int p = *ext * *pitch;
    while ((p - *ext) % 32 != 0) p++;               // plane = ext (mod 32): phase 2 lanes stay on distinct banks across planes
    *plane = p;
}

size_t blur_iter_lds_bytes(int winsize)
{
    int ext, pitch, plane;
    iter_geometry(winsize / 2, &ext, &pitch, &plane);
    return sizeof(float) * 5 * (size_t)plane;
}

template <int M_T>
__global__ __launch_bounds__(256) void k_blur_iter(const float* __restrict__ M_in, float* __restrict__ M_out, size_t M_stride,
                                                   const float* __restrict__ R0, const float* __restrict__ R1, size_t R_stride,
                                                   int w, int h, int m_rt, int pitch, int plane, float scale, int do_update,
                                                   float* __restrict__ flow, size_t f_stride)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int m = M_T > 0 ? M_T : m_rt;
    const int ext = MAV_TILE + 2 * m;
    const int win = 2 * m + 1;
    const int tid = threadIdx.x;
    const int s = blockIdx.z;
    const int x0 = blockIdx.x * MAV_TILE, y0 = blockIdx.y * MAV_TILE;
    const size_t npx = (size_t)w * h;
    const float* Min = M_in + (size_t)s * M_stride;

    for (int c = 0; c < 5; c++) {
        const float* P = Min + c * npx;
        float* L = lds + c * plane;
        for (int i = tid; i < ext * ext; i += 256) {
            const int ly = i / ext, lx = i - ly * ext;
            const int gx = clampi(x0 - m + lx, 0, w - 1), gy = clampi(y0 - m + ly, 0, h - 1);
            L[ly * pitch + lx] = P[(size_t)gy * w + gx];
        }
    }
    __syncthreads();
    for (int t = tid; t < 5 * ext; t += 256) {
        const int c = t / ext, lx = t - c * ext;
        float* col = lds + c * plane + lx;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < win; k++) sum += col[k * pitch];
#pragma unroll 8
        for (int y = 0; y < MAV_TILE; y++) {
            const float out = sum;
            if (y < MAV_TILE - 1) sum += col[(y + win) * pitch] - col[y * pitch];
            col[y * pitch] = out;
        }
    }
    __syncthreads();
    for (int t = tid; t < 5 * MAV_TILE; t += 256) {
        const int c = t >> 5, y = t & 31;
        float* row = lds + c * plane + y * pitch;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < win; k++) sum += row[k];
#pragma unroll 8
        for (int x = 0; x < MAV_TILE; x++) {
            const float out = sum;
            if (x < MAV_TILE - 1) sum += row[x + win] - row[x];
            row[x] = out;
        }
    }
    __syncthreads();
    const float* R0p = R0 + (size_t)s * R_stride;
    const float* R1p = R1 + (size_t)s * R_stride;
    float* Mo = M_out + (size_t)s * M_stride;
    float* fo = flow + (size_t)s * f_stride;
    const int lx = tid & 31;
    const int gx = x0 + lx;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ly = (tid >> 5) + 8 * j;
        const int gy = y0 + ly;
        if (gx >= w || gy >= h) continue;
        const float* L = lds + ly * pitch + lx;
        const float g11 = L[0] * scale, g12 = L[plane] * scale, g22 = L[2 * plane] * scale, h1 = L[3 * plane] * scale,
                    h2 = L[4 * plane] * scale;
        // differences of products with one fused rounding each (Kahan): the CPU path does this step in double
        const float p = g12 * g12, pe = fmaf(g12, g12, -p);
        const float det = (fmaf(g11, g22, -p) - pe) + 1e-3f;
        const float idet = 1.f / det;
        const float qa = g12 * h1, qae = fmaf(g12, h1, -qa);
        const float qb = g12 * h2, qbe = fmaf(g12, h2, -qb);
        const float u = (fmaf(g11, h2, -qa) - qae) * idet;
        const float v = (fmaf(g22, h1, -qb) - qbe) * idet;
        *(float2*)(fo + ((size_t)gy * w + gx) * 2) = make_float2(u, v);
        if (do_update) update_px(R0p, R1p, npx, w, h, gx, gy, u, v, Mo);
    }
}

void launch_blur_iter(hipStream_t st, const float* M_in, float* M_out, size_t M_stride, const float* R0, const float* R1,
                      size_t R_stride, int G, int w, int h, int winsize, int do_update, float* flow, size_t f_stride)
{
    int ext, pitch, plane;
    const int m = winsize / 2;
    iter_geometry(m, &ext, &pitch, &plane);
    const size_t lds = sizeof(float) * 5 * (size_t)plane;
    const float scale = (float)(1.0 / ((double)winsize * winsize));
    dim3 grid((w + MAV_TILE - 1) / MAV_TILE, (h + MAV_TILE - 1) / MAV_TILE, G);
    if (m == 6)
        hipLaunchKernelGGL(k_blur_iter<6>, grid, dim3(256), lds, st, M_in, M_out, M_stride, R0, R1, R_stride, w, h, m, pitch,
                           plane, scale, do_update, flow, f_stride);
    else
        hipLaunchKernelGGL(k_blur_iter<0>, grid, dim3(256), lds, st, M_in, M_out, M_stride, R0, R1, R_stride, w, h, m, pitch,
                           plane, scale, do_update, flow, f_stride);
}
