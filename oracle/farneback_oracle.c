/*
 * farneback_oracle.c -- CPU restatement of cv2.calcOpticalFlowFarneback (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the *oracle* for the dense-flow stage of the hot path. It is a checker: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The product path (libmavflow.so)
 * never links or calls it.
 *
 * What it restates: the reference's only Farneback call site is
 *     /root/reference/src/farneback.py:76-80
 *         cv2.calcOpticalFlowFarneback(prev, next, None, pyr_scale=0.4, levels=1, winsize=12,
 *                                      iterations=10, poly_n=8, poly_sigma=1.2, flags=0)
 * whose arithmetic lives in a third-party dependency that is ABSENT from /root/reference and from this
 * image: `opencv-python`, version UNPINNED (requirements.txt:4).  The algorithm below follows OpenCV 4.x
 * `modules/video/src/optflowgf.cpp` (FarnebackOpticalFlowImpl::calc, FarnebackPrepareGaussian,
 * FarnebackPolyExp, FarnebackUpdateMatrices, FarnebackUpdateFlow_Blur) plus `GaussianBlur`
 * (getGaussianKernel, separable symmetric row/column filter, BORDER_REFLECT_101) and
 * `resize(INTER_LINEAR)` as published; SURVEY.md Appendix A is the written spec.
 *
 * PARITY UNPINNED at the cv2 boundary: the reference holds no golden vector, test or fixture for this
 * call and cv2 cannot be imported here, so nothing in this file could be checked against OpenCV itself.
 * What pins it instead: analytic-flow tests (tests/test_oracle_farneback.py) and GPU<->oracle agreement.
 *
 * Layout conventions (OpenCV's own): images row-major; R and M are 5-channel interleaved (CV_32FC5),
 * flow is 2-channel interleaved (u, v).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef struct {
    double pyr_scale;
    int levels, winsize, iterations, poly_n;
    double poly_sigma;
    int flags;
} fbo_params;

/* cvRound: round-half-to-even (SSE cvtsd2si semantics) */
static int cv_round(double v) { return (int)nearbyint(v); }
static int cv_floor(float v) { int i = (int)v; return i - (i > v); }

static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    }
    return p;
}

/* optflowgf.cpp calc(): number of extra layers actually used (A.1) */
int fbo_num_layers(int W, int H, const fbo_params* p)
{
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < p->levels; k++) {
        scale *= p->pyr_scale;
        if (W * scale < min_size || H * scale < min_size) break;
    }
    return k + 1; /* layers k..0 */
}

void fbo_layer_dims(int W, int H, const fbo_params* p, int k, int* w, int* h, double* sigma, int* ksize)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= p->pyr_scale;
    double s = (1. / scale - 1) * 0.5;
    int sz = cv_round(s * 5) | 1;
    if (sz < 3) sz = 3;
    *w = cv_round(W * scale);
    *h = cv_round(H * scale);
    *sigma = s;
    *ksize = sz;
}

/* getGaussianKernel(n, sigma, CV_32F) */
void fbo_gaussian_kernel(int n, double sigma, float* k)
{
    static const float small_tab[4][7] = {
        {1.f},
        {0.25f, 0.5f, 0.25f},
        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float* fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : 0;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

/* convertTo(CV_32F) -> GaussianBlur(ksize, sigma) -> resize(w,h,INTER_LINEAR)    (A.2) */
void fbo_blur_resize(const uint8_t* img, int W, int H, int w, int h, int ksize, double sigma, float* out)
{
    int r = ksize / 2;
    float* kern = (float*)malloc(sizeof(float) * ksize);
    fbo_gaussian_kernel(ksize, sigma, kern);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)W * H);
    float* blur = (float*)malloc(sizeof(float) * (size_t)W * H);
    /* row filter (symmetric form, float accumulation) */
    for (int y = 0; y < H; y++) {
        const uint8_t* s = img + (size_t)y * W;
        float* d = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float acc = kern[r] * (float)s[x];
            for (int j = 1; j <= r; j++)
                acc += kern[r + j] * ((float)s[reflect101(x - j, W)] + (float)s[reflect101(x + j, W)]);
            d[x] = acc;
        }
    }
    /* column filter */
    for (int y = 0; y < H; y++) {
        float* d = blur + (size_t)y * W;
        const float* c = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) d[x] = kern[r] * c[x];
        for (int j = 1; j <= r; j++) {
            const float* a = tmp + (size_t)reflect101(y - j, H) * W;
            const float* b = tmp + (size_t)reflect101(y + j, H) * W;
            float kj = kern[r + j];
            for (int x = 0; x < W; x++) d[x] += kj * (a[x] + b[x]);
        }
    }
    free(tmp);
    if (w == W && h == H) { /* resize to the same size is a copy */
        memcpy(out, blur, sizeof(float) * (size_t)W * H);
        free(blur); free(kern);
        return;
    }
    /* resize INTER_LINEAR, half-pixel centres, clamp */
    double scale_x = (double)W / w, scale_y = (double)H / h;
    int* xofs = (int*)malloc(sizeof(int) * w);
    float* xa = (float*)malloc(sizeof(float) * w);
    for (int dx = 0; dx < w; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= W - 1) { fx = 0; sx = W - 1; }
        xofs[dx] = sx; xa[dx] = fx;
    }
    for (int dy = 0; dy < h; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= H - 1) { fy = 0; sy = H - 1; }
        const float* s0 = blur + (size_t)sy * W;
        const float* s1 = blur + (size_t)(sy + 1 < H ? sy + 1 : sy) * W;
        float b0 = 1.f - fy, b1 = fy;
        for (int dx = 0; dx < w; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < W ? sx + 1 : sx;
            float a0 = 1.f - xa[dx], a1 = xa[dx];
            float h0 = s0[sx] * a0 + s0[sx1] * a1;
            float h1 = s1[sx] * a0 + s1[sx1] * a1;
            out[(size_t)dy * w + dx] = h0 * b0 + h1 * b1;
        }
    }
    free(xofs); free(xa); free(blur); free(kern);
}

/* solve the 6x6 SPD system by Cholesky and return the inverse (row-major) */
static void inv6_cholesky(const double G[36], double inv[36])
{
    double L[36];
    memset(L, 0, sizeof(L));
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = G[i * 6 + j];
            for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = (i == j) ? sqrt(s) : s / L[j * 6 + j];
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
}

/* FarnebackPrepareGaussian (A.3). g/xg/xxg point at the centre tap (index 0), valid for [-n, n]. */
void fbo_prepare_gaussian(int n, double sigma, float* g, float* xg, float* xxg, double* ig /*[4]*/)
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
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
            G[0 * 6 + 0] += g[y] * g[x];
            G[1 * 6 + 1] += g[y] * g[x] * x * x;
            G[3 * 6 + 3] += g[y] * g[x] * x * x * x * x;
            G[5 * 6 + 5] += g[y] * g[x] * x * x * y * y;
        }
    G[2 * 6 + 2] = G[0 * 6 + 3] = G[0 * 6 + 4] = G[3 * 6 + 0] = G[4 * 6 + 0] = G[1 * 6 + 1];
    G[4 * 6 + 4] = G[3 * 6 + 3];
    G[3 * 6 + 4] = G[4 * 6 + 3] = G[5 * 6 + 5];
    double inv[36];
    inv6_cholesky(G, inv);
    ig[0] = inv[1 * 6 + 1]; /* ig11 */
    ig[1] = inv[0 * 6 + 3]; /* ig03 */
    ig[2] = inv[3 * 6 + 3]; /* ig33 */
    ig[3] = inv[5 * 6 + 5]; /* ig55 */
}

/* FarnebackPolyExp (A.4): I (h x w f32) -> R (h x w x 5 f32) */
void fbo_polyexp(const float* src, int width, int height, int n, double sigma, float* dst)
{
    float* kbuf = (float*)malloc(sizeof(float) * (n * 6 + 3));
    float* g = kbuf + n;
    float* xg = g + n * 2 + 1;
    float* xxg = xg + n * 2 + 1;
    double ig[4];
    fbo_prepare_gaussian(n, sigma, g, xg, xxg, ig);
    double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
    float* _row = (float*)malloc(sizeof(float) * (size_t)(width + n * 2) * 3);
    float* row = _row + n * 3;

    for (int y = 0; y < height; y++) {
        float g0 = g[0], g1, g2;
        const float* srow0 = src + (size_t)y * width;
        const float* srow1 = 0;
        float* drow = dst + (size_t)y * width * 5;
        /* vertical part of the convolution (rows clamped) */
        for (int x = 0; x < width; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * width;
            srow1 = src + (size_t)(y + k < height - 1 ? y + k : height - 1) * width;
            for (int x = 0; x < width; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0; row[x * 3 + 1] = t1; row[x * 3 + 2] = t2;
            }
        }
        /* horizontal part: replicate the edge triples */
        for (int x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[width * 3 + x] = row[width * 3 + x - 3];
        }
        for (int x = 0; x < width; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(_row); free(kbuf);
}

/* Checker-side bookkeeping of the ONE discontinuity of the iteration: whether a pixel's displaced position (x + u, y + v) lies
 * inside the second image (bilinear sample of R1) or not (the else-branch below).  A pixel whose decision changes from one update
 * to the next sits on that discontinuity: an implementation that differs by one ulp may flip it one sweep earlier or later and is
 * then a visible distance away for as long as the flip repeats (limit cycles along the image border, profiles/r06/border_cycle.txt).
 * state / flips: one byte per pixel, NULL = not tracked; count_from: flips are counted in updates with index >= count_from. */
typedef struct { uint8_t* state; uint8_t* flips; int update_index, count_from; } fbo_branch_track;
static _Thread_local fbo_branch_track* g_track = 0;    /* set by calc_ex for the finest layer only (per calling thread) */

/* FarnebackUpdateMatrices (A.5), rows [y0, y1) */
void fbo_update_matrices_rows(const float* R0_, const float* R1, const float* flow_, int width, int height,
                              float* matM, int y0, int y1)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    size_t step1 = (size_t)width * 5;
    for (int y = y0; y < y1; y++) {
        const float* flow = flow_ + (size_t)y * width * 2;
        const float* R0 = R0_ + (size_t)y * width * 5;
        float* M = matM + (size_t)y * width * 5;
        for (int x = 0; x < width; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor(fx), y1_ = cv_floor(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= y1_;
            if (g_track) {
                const uint8_t in = (unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1_ < (unsigned)(height - 1);
                uint8_t* st = g_track->state + (size_t)y * width + x;
                if (g_track->update_index > 0 && g_track->update_index >= g_track->count_from && *st != in) {
                    uint8_t* fl = g_track->flips + (size_t)y * width + x;
                    if (*fl < 255) (*fl)++;
                }
                *st = in;
            }
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1_ < (unsigned)(height - 1)) {
                const float* ptr = R1 + (size_t)y1_ * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) * (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) * (y >= height - BORDER ? border[height - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

void fbo_update_matrices(const float* R0, const float* R1, const float* flow, int w, int h, float* M)
{
    fbo_update_matrices_rows(R0, R1, flow, w, h, M, 0, h);
}

/* FarnebackUpdateFlow_Blur (A.6): one sweep; M is rewritten in row stripes behind the sweep when update != 0.
 * sys (nullable, h x w x 7 doubles) receives, per pixel, the scaled 2x2 system it was solved from and the flow the sweep
 * replaced: (g11, g12, g22, h1, h2, u_before, v_before) -- the checker's view of how well conditioned a pixel is and of
 * how far the iteration still moves there (oracle/tolerances.py); OpenCV has no such output.
 * acc_float != 0 is NOT OpenCV: the same sweep with every window sum and the solve rounded to float32 after each operation
 * (OpenCV accumulates them in double).  It exists to measure how far the restatement's OWN result moves under float32 rounding
 * of its sums -- the numerically unstable pixels of oracle/tolerances.py -- and is never the expected value of a test. */
#define RND(v) (acc_float ? (double)(float)(v) : (double)(v))
void fbo_blur_iter_ex(const float* R0, const float* R1, float* flow_, float* matM, int width, int height,
                      int block_size, int update_matrices, double* sys, int acc_float)
{
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = (1 << 10) / width > block_size ? (1 << 10) / width : block_size;
    double scale = RND(1. / (block_size * block_size));
    double* _vsum = (double*)malloc(sizeof(double) * (size_t)(width + m * 2 + 2) * 5);
    double* vsum = _vsum + (m + 1) * 5;

    const float* srow0 = matM;
    for (int x = 0; x < width * 5; x++) vsum[x] = RND(srow0[x] * (m + 2));
    for (int y = 1; y < m; y++) {
        srow0 = matM + (size_t)(y < height - 1 ? y : height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] = RND(vsum[x] + srow0[x]);
    }
    for (int y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float* flow = flow_ + (size_t)y * width * 2;
        srow0 = matM + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * width * 5;
        const float* srow1 = matM + (size_t)(y + m < height - 1 ? y + m : height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] = RND(vsum[x] + RND(srow1[x] - srow0[x]));
        for (int x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        g11 = RND(vsum[0] * (m + 2)); g12 = RND(vsum[1] * (m + 2)); g22 = RND(vsum[2] * (m + 2));
        h1 = RND(vsum[3] * (m + 2)); h2 = RND(vsum[4] * (m + 2));
        for (int x = 1; x < m; x++) {
            g11 = RND(g11 + vsum[x * 5]); g12 = RND(g12 + vsum[x * 5 + 1]); g22 = RND(g22 + vsum[x * 5 + 2]);
            h1 = RND(h1 + vsum[x * 5 + 3]); h2 = RND(h2 + vsum[x * 5 + 4]);
        }
        for (int x = 0; x < width; x++) {
            g11 = RND(g11 + RND(vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5]));
            g12 = RND(g12 + RND(vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4]));
            g22 = RND(g22 + RND(vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3]));
            h1 = RND(h1 + RND(vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2]));
            h2 = RND(h2 + RND(vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1]));
            double g11_ = RND(g11 * scale), g12_ = RND(g12 * scale), g22_ = RND(g22 * scale), h1_ = RND(h1 * scale), h2_ = RND(h2 * scale);
            double idet = RND(1. / RND(RND(RND(g11_ * g22_) - RND(g12_ * g12_)) + 1e-3));
            if (sys) {
                double* o = sys + ((size_t)y * width + x) * 7;
                o[0] = g11_; o[1] = g12_; o[2] = g22_; o[3] = h1_; o[4] = h2_;
                o[5] = flow[x * 2]; o[6] = flow[x * 2 + 1];
            }
            flow[x * 2] = (float)(RND(RND(g11_ * h2_) - RND(g12_ * h1_)) * idet);
            flow[x * 2 + 1] = (float)(RND(RND(g22_ * h1_) - RND(g12_ * h2_)) * idet);
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update_matrices && (y1 == height || y1 >= y0 + min_update_stripe)) {
            fbo_update_matrices_rows(R0, R1, flow_, width, height, matM, y0, y1);
            y0 = y1;
        }
    }
    free(_vsum);
}
#undef RND

void fbo_blur_iter_sys(const float* R0, const float* R1, float* flow_, float* matM, int width, int height,
                       int block_size, int update_matrices, double* sys)
{
    fbo_blur_iter_ex(R0, R1, flow_, matM, width, height, block_size, update_matrices, sys, 0);
}

void fbo_blur_iter(const float* R0, const float* R1, float* flow_, float* matM, int width, int height,
                   int block_size, int update_matrices)
{
    fbo_blur_iter_sys(R0, R1, flow_, matM, width, height, block_size, update_matrices, 0);
}

/* resize(prevFlow -> (w,h), INTER_LINEAR) * (1/pyr_scale) */
void fbo_resize_flow(const float* prev, int pw, int ph, int w, int h, double mul, float* flow)
{
    double scale_x = (double)pw / w, scale_y = (double)ph / h;
    float fmul = (float)mul;
    for (int dy = 0; dy < h; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= ph - 1) { fy = 0; sy = ph - 1; }
        int sy1 = sy + 1 < ph ? sy + 1 : sy;
        float b0 = 1.f - fy, b1 = fy;
        for (int dx = 0; dx < w; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = cv_floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= pw - 1) { fx = 0; sx = pw - 1; }
            int sx1 = sx + 1 < pw ? sx + 1 : sx;
            float a0 = 1.f - fx, a1 = fx;
            for (int c = 0; c < 2; c++) {
                float h0 = prev[((size_t)sy * pw + sx) * 2 + c] * a0 + prev[((size_t)sy * pw + sx1) * 2 + c] * a1;
                float h1 = prev[((size_t)sy1 * pw + sx) * 2 + c] * a0 + prev[((size_t)sy1 * pw + sx1) * 2 + c] * a1;
                flow[((size_t)dy * w + dx) * 2 + c] = (h0 * b0 + h1 * b1) * fmul;
            }
        }
    }
}

/* FarnebackOpticalFlowImpl::calc, flags == 0 path. Returns 0 on success, <0 on bad arguments.
 * sys_last (nullable, H x W x 7 doubles): the finest layer's last sweep as fbo_blur_iter_sys reports it. */
static int calc_ex(const uint8_t* prev, const uint8_t* next, int W, int H, const fbo_params* p, float* flow0, double* sys_last,
                   int acc_float, uint8_t* flips_out)
{
    if (!prev || !next || !flow0 || W <= 0 || H <= 0) return -1;
    if (!(p->pyr_scale > 0 && p->pyr_scale < 1) || p->levels < 0 || p->winsize < 2 || p->iterations < 0 ||
        p->poly_n < 1 || p->flags != 0)
        return -2;
    const uint8_t* img[2] = {prev, next};
    int levels = fbo_num_layers(W, H, p) - 1;
    float* prevFlow = 0;
    int pw = 0, ph = 0;
    for (int k = levels; k >= 0; k--) {
        int w, h, ksize;
        double sigma;
        fbo_layer_dims(W, H, p, k, &w, &h, &sigma, &ksize);
        float* flow = (k > 0) ? (float*)malloc(sizeof(float) * (size_t)w * h * 2) : flow0;
        if (!prevFlow) memset(flow, 0, sizeof(float) * (size_t)w * h * 2);
        else fbo_resize_flow(prevFlow, pw, ph, w, h, 1. / p->pyr_scale, flow);
        float* R[2];
        float* I = (float*)malloc(sizeof(float) * (size_t)w * h);
        for (int i = 0; i < 2; i++) {
            R[i] = (float*)malloc(sizeof(float) * (size_t)w * h * 5);
            fbo_blur_resize(img[i], W, H, w, h, ksize, sigma, I);
            fbo_polyexp(I, w, h, p->poly_n, p->poly_sigma, R[i]);
        }
        free(I);
        float* M = (float*)malloc(sizeof(float) * (size_t)w * h * 5);
        fbo_branch_track tr = {0, 0, 0, 0};
        if (k == 0 && flips_out) {              /* finest layer: track the inside / outside decision over its updates; count the last four */
            tr.state = (uint8_t*)calloc((size_t)w * h, 1);
            tr.flips = flips_out;
            memset(flips_out, 0, (size_t)w * h);
            tr.count_from = p->iterations - 1 - 4 > 1 ? p->iterations - 1 - 4 : 1;
            g_track = &tr;
        }
        fbo_update_matrices(R[0], R[1], flow, w, h, M);           /* update 0: from the up-sampled flow */
        for (int i = 0; i < p->iterations; i++) {
            tr.update_index = i + 1;                              /* update i + 1: from sweep i's flow (none after the last sweep) */
            fbo_blur_iter_ex(R[0], R[1], flow, M, w, h, p->winsize, i < p->iterations - 1,
                             (k == 0 && i == p->iterations - 1) ? sys_last : 0, acc_float);
        }
        g_track = 0;
        free(tr.state);
        free(M); free(R[0]); free(R[1]);
        if (prevFlow) free(prevFlow);
        prevFlow = (k > 0) ? flow : 0;
        pw = w; ph = h;
    }
    return 0;
}

int fbo_calc_sys(const uint8_t* prev, const uint8_t* next, int W, int H, const fbo_params* p, float* flow0, double* sys_last)
{
    return calc_ex(prev, next, W, H, p, flow0, sys_last, 0, 0);
}

/* ... and flips (nullable, H x W bytes): per pixel, in how many of the finest layer's last four updates the inside / outside
 * decision of its displaced position differed from the update before (fbo_branch_track). */
int fbo_calc_track(const uint8_t* prev, const uint8_t* next, int W, int H, const fbo_params* p, float* flow0, double* sys_last, uint8_t* flips)
{
    return calc_ex(prev, next, W, H, p, flow0, sys_last, 0, flips);
}

int fbo_calc(const uint8_t* prev, const uint8_t* next, int W, int H, const fbo_params* p, float* flow0)
{
    return calc_ex(prev, next, W, H, p, flow0, 0, 0, 0);
}

/* NOT OpenCV: calc() with float32 window sums (see fbo_blur_iter_ex); the sensitivity twin of oracle/tolerances.py. */
int fbo_calc_f32sums(const uint8_t* prev, const uint8_t* next, int W, int H, const fbo_params* p, float* flow0)
{
    return calc_ex(prev, next, W, H, p, flow0, 0, 1, 0);
}
