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

#include <algorithm>

static __device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------------------------------------------------
// Workgroup -> tile mapping shared by the tiled kernels (64 x 16 pixel tiles over G images).
// Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own 4 MB L2), so workgroup b
// takes tile (b & 7) * per + (b >> 3): every XCD walks ONE contiguous run of tiles and the halo rows / gather rows that
// vertically adjacent tiles share are re-read from that XCD's L2 instead of HBM (a plain 2-D grid puts neighbours on
// different XCDs: measured 3.9x read over-fetch in the expansion kernel).  Inside an image the run goes through column
// STRIPS of strip_w tiles, row by row inside a strip: the tiles resident on an XCD at one time (~160) then span several
// tile rows of the strip, so a tile's upper neighbour is still in L2 when it is needed -- on a 3840-wide layer one tile row
// is 60 tiles and without strips the reuse distance exceeds the L2.  Speed only: every tile is computed exactly once and
// nothing depends on the order.  Grid = n_tiles rounded up to a multiple of 8.
// ------------------------------------------------------------------------------------------------------------
struct TileMap { int tiles_x, tiles_y, per_img, n_tiles, strip_w, xcd, ty0; };   // xcd = 0: plain row-major order; ty0: first tile row (a band launch)
static __device__ __forceinline__ bool tile_of_block(const TileMap& tm, int* s, int* tx, int* ty)
{
    const int per = ((int)gridDim.x + 7) >> 3;
    const int tile = tm.xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile >= tm.n_tiles) return false;
    const int img = tile / tm.per_img;
    int tr = tile - img * tm.per_img;
    const int strip_tiles = tm.strip_w * tm.tiles_y;          // every strip but the last is strip_w tiles wide
    const int st = tr / strip_tiles;
    tr -= st * strip_tiles;
    const int x_base = st * tm.strip_w;
    const int sw = min(tm.strip_w, tm.tiles_x - x_base);
    const int row = tr / sw;
    *s = img; *ty = row + tm.ty0; *tx = x_base + tr - row * sw;
    return true;
}
// rows [ty0, ty1) of the tile grid only (a band of the image); ty1 < 0 = all rows.  strip > 0: strips of that many tiles (option "strip").
static TileMap make_tile_map(int w, int h, int G, int tile_w, int tile_h, int ty0 = 0, int ty1 = -1, int strip = 0)
{
    TileMap tm;
    tm.xcd = 1;
    tm.tiles_x = (w + tile_w - 1) / tile_w;
    tm.tiles_y = (h + tile_h - 1) / tile_h;
    tm.ty0 = ty0 > 0 ? (ty0 < tm.tiles_y ? ty0 : tm.tiles_y) : 0;
    if (ty1 >= 0 && ty1 < tm.tiles_y) tm.tiles_y = ty1;
    tm.tiles_y = tm.tiles_y > tm.ty0 ? tm.tiles_y - tm.ty0 : 0;
    tm.per_img = tm.tiles_x * tm.tiles_y;
    tm.n_tiles = tm.per_img * G;
    int sw = tm.tiles_x;
    if (strip > 0) sw = strip < tm.tiles_x ? strip : tm.tiles_x;
    else if (tm.tiles_x > 40) { const int ns = (tm.tiles_x + 29) / 30; sw = (tm.tiles_x + ns - 1) / ns; }
    tm.strip_w = sw;
    return tm;
}
static inline unsigned tile_grid(const TileMap& tm) { return (unsigned)(((tm.n_tiles + 7) / 8) * 8); }

#ifdef MAV_STAMPS   // diagnostic build only (tools/phase_stamps.py): per-phase wave cycles of the sweep kernel, never in the product .so
#define MAV_STAMP_WAVES (1 << 17)
__device__ unsigned long long g_phase_cycles[MAV_STAMP_WAVES * 8];   // one row per wave slot of a launch: plain += (launches are serial)
#define STAMP(var) unsigned long long var; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0)
#define STAMP_ADD(i, a, b) do { if ((threadIdx.x & 63) == 0) { const unsigned wid = (blockIdx.x * 4u + (threadIdx.x >> 6)) & (MAV_STAMP_WAVES - 1); g_phase_cycles[wid * 8 + (i)] += (unsigned long long)((b) - (a)); } } while (0)
extern "C" int mav_debug_read_stamps(unsigned long long* out, int reset)
{
    static unsigned long long host[MAV_STAMP_WAVES * 8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_phase_cycles), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 8; i++) out[i] = 0;
    for (size_t w = 0; w < MAV_STAMP_WAVES; w++) for (int i = 0; i < 8; i++) out[i] += host[w * 8 + i];
    if (reset) { static unsigned long long z[MAV_STAMP_WAVES * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define STAMP(var)
#define STAMP_ADD(i, a, b)
#endif

// ------------------------------------------------------------------------------------------------------------
// Layer image: convertTo(f32) -> GaussianBlur(ksize, sigma, REFLECT_101) -> resize(INTER_LINEAR)   (A.2).
// ------------------------------------------------------------------------------------------------------------
// Two separable passes for the coarse layers (any ksize / scale), table-free: along each axis the output is
//   a0 * B[s0] + a1 * B[s0 + 1],   B[c] = sum_t g[t] * src[reflect101(c - r + t)]
// with (s0, a1) from the half-pixel-centre rule of resize(INTER_LINEAR).  Away from the border the two Gaussian sums share
// their pixels (B[s0+1] is B[s0] shifted by one), so a tap costs one load and two FMAs; g[t] is wave-uniform (scalar loads).
//   pass H   tmp[y][dx]   for every source row y            (H x w f32; 4 rows per thread share the column arithmetic)
//   pass V   out[dy][dx]  from tmp                          (h x w f32)
// MACs per output drop from (ksize+1)^2 to ~(ksize+1)(H/h + 1): the 95-tap layer of the 4K / 5-layer preset costs the same as
// the 5-tap one.
static __device__ __forceinline__ void resize_coord(int o, int S, int d, double scale, int* s0, float* f)
{
#pragma clang fp contract(off)                       // (o + 0.5) * scale - 0.5 with separate roundings, as OpenCV's host code evaluates it
    if (d == S) { *s0 = o; *f = 0.f; return; }
    float t = (float)((o + 0.5) * scale - 0.5);
    int s = (int)floorf(t);
    t -= s;
    if (s < 0) { t = 0.f; s = 0; }
    if (s >= S - 1) { t = 0.f; s = S - 1; }
    *s0 = s; *f = t;
}
static __device__ __forceinline__ int reflect101d(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// Image z of a launch over two runs of images (the `prev` frames and the `next` frames of a group): z < split comes from run a.
static __device__ __forceinline__ const uint8_t* image_of(const uint8_t* a, const uint8_t* b, int split, size_t stride, int z)
{
    return z < split ? a + (size_t)z * stride : b + (size_t)(z - split) * stride;
}

// Horizontal pass at destination column (s0, f) for the four source rows y0 .. y0 + 3 (clamped to the image): the ONE spelling of
// this arithmetic, shared by the two-pass and the fused kernel so that both give the same bits.
static __device__ __forceinline__ void blur_h4(const uint8_t* __restrict__ base, int W, int H, int y0, int s0, float f,
                                               const BlurParams& bp, float out[4])
{
    const int r = bp.ksize >> 1;
    const int s1 = s0 + 1 < W ? s0 + 1 : s0;
    const uint8_t* rows[4];
#pragma unroll
    for (int k = 0; k < 4; k++) rows[k] = base + (size_t)min(y0 + k, H - 1) * W;
    float b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f};
    if (s0 - r >= 0 && s0 + 1 + r < W) {               // interior: B[s0+1] re-uses B[s0]'s pixels shifted by one
        float prev[4];
        if (bp.ksize == 5) {                               // all 24 bytes of the thread requested before any is used
            uint8_t px[4][6];
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int t = 0; t < 6; t++) px[k][t] = rows[k][s0 - 2 + t];
#pragma unroll
            for (int t = 0; t < 5; t++) {
                const float g = bp.g[t];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    b0[k] = fmaf(g, (float)px[k][t], b0[k]);
                    b1[k] = fmaf(g, (float)px[k][t + 1], b1[k]);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) prev[k] = (float)rows[k][s0 - r];
            // long kernels (the 4K / 5-layer preset has 13-, 37- and 95-tap layers): the ksize + 1 consecutive bytes of a row
            // come as ALIGNED dwords and are re-aligned with v_alignbyte -- a quarter of the load instructions.  Only words
            // that hold a needed byte are read (an aligned dword never crosses a page, so nothing unmapped is touched).
            // Same products in the same order as the byte loop: identical results.
            const uint32_t* wp[4];
            unsigned sh[4];
            uint32_t lo[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uintptr_t a = (uintptr_t)(rows[k] + (s0 - r));
                wp[k] = (const uint32_t*)(a & ~(uintptr_t)3);
                sh[k] = (unsigned)(a & 3);
                lo[k] = wp[k][0];
            }
            const int nbytes = bp.ksize + 1;                 // byte 0 = prev, byte j = tap j - 1's "next"
            for (int c = 0; 4 * c < nbytes; c++) {
                uint32_t cur[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // the needed bytes sit at positions sh .. sh + nbytes - 1 of the word stream
                    const uint32_t hi = c + 1 <= ((int)sh[k] + nbytes - 1) / 4 ? wp[k][c + 1] : 0u;
                    cur[k] = __builtin_amdgcn_alignbyte(hi, lo[k], sh[k]);
                    lo[k] = hi;
                }
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int j = 4 * c + b;
                    if (j >= nbytes) break;
                    if (j > 0) {
                        const float g = bp.g[j - 1];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const float nxt = (float)((cur[k] >> (8 * b)) & 0xffu);
                            b0[k] = fmaf(g, prev[k], b0[k]);
                            b1[k] = fmaf(g, nxt, b1[k]);
                            prev[k] = nxt;
                        }
                    }
                }
            }
        }
    } else {
        for (int t = 0; t < bp.ksize; t++) {
            const float g = bp.g[t];
            const int c0 = reflect101d(s0 - r + t, W), c1 = reflect101d(s1 - r + t, W);
#pragma unroll
            for (int k = 0; k < 4; k++) { b0[k] = fmaf(g, (float)rows[k][c0], b0[k]); b1[k] = fmaf(g, (float)rows[k][c1], b1[k]); }
        }
    }
    const float a0 = 1.f - f;
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = fmaf(b1[k], f, b0[k] * a0);
}

// The same horizontal pass for four rows of a source region STAGED IN LDS (stage_rows: columns outside the image already hold their
// BORDER_REFLECT_101 pixels, so one code path serves interior and border tiles).  rd(k, wi) = aligned dword wi of the thread's k-th
// row; the thread's first tap is byte `off` of that stream; tap t multiplies byte off + t for B[s0] and byte off + t + 1 for
// B[s0 + 1] (when s0 is the last column resize_coord gives f = 0 and B[s0 + 1] drops out exactly).  The bytes come as dwords
// re-aligned with v_alignbyte.  Same products in the same order with the same fused roundings as blur_h4: identical bits.
template <int KS, typename WordFn>      // KS > 0: ksize known at compile time (the loops unroll: all LDS reads of a thread issue together)
static __device__ __forceinline__ void blur_h4_stream_t(WordFn rd, int off, float f, const BlurParams& bp, float out[4])
{
    const int w0 = off >> 2;
    const unsigned sh = (unsigned)(off & 3);
    const int nbytes = (KS > 0 ? KS : bp.ksize) + 1;          // byte 0 = the first "prev", byte j = tap j - 1's "next"
    float b0[4] = {0.f, 0.f, 0.f, 0.f}, b1[4] = {0.f, 0.f, 0.f, 0.f}, prev[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t lo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) lo[k] = rd(k, w0);
    auto step = [&](int c) {
        uint32_t cur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t hi = rd(k, w0 + c + 1);              // (one word of slack behind every staged row)
            cur[k] = __builtin_amdgcn_alignbyte(hi, lo[k], sh);
            lo[k] = hi;
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int j = 4 * c + b;
            if (j >= nbytes) break;
            if (j == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) prev[k] = (float)(cur[k] & 0xffu);
            } else {
                const float g = bp.g[j - 1];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float nxt = (float)((cur[k] >> (8 * b)) & 0xffu);
                    b0[k] = fmaf(g, prev[k], b0[k]);
                    b1[k] = fmaf(g, nxt, b1[k]);
                    prev[k] = nxt;
                }
            }
        }
        };
    if constexpr (KS > 0) {
#pragma unroll
        for (int c = 0; 4 * c < KS + 1; c++) step(c);
    } else {
        for (int c = 0; 4 * c < nbytes; c++) step(c);
    }
    const float a0 = 1.f - f;
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = fmaf(b1[k], f, b0[k] * a0);
}
template <typename WordFn>
static __device__ __forceinline__ void blur_h4_stream(WordFn rd, int off, float f, const BlurParams& bp, float out[4])
{
    if (bp.ksize == 5) blur_h4_stream_t<5>(rd, off, f, bp, out);          // the reference's preset: layer 1 (scale 0.4, sigma 0.75)
    else if (bp.ksize == 13) blur_h4_stream_t<13>(rd, off, f, bp, out);
    else blur_h4_stream_t<0>(rd, off, f, bp, out);
}

// The horizontal pass once more, for LONG Gaussians (the two-pass kernels: 37 and 95 taps at 3840 x 2160 / 5 layers), on tap PAIRS
// kept in LDS.  blur_h4_stream's generic loop fetched every tap with a vector load from global memory and waited for it -- one
// memory round trip per tap (the taps sit behind a pointer the compiler cannot prove constant next to the kernel's own stores) --
// and shuffled registers to pack two rows into one v_pk_fma_f32.  Here the (b0, b1) accumulators of ONE row are the packed pair:
// source byte t of the row contributes  g[t] * p  to B[s0]  and  g[t - 1] * p  to B[s0 + 1], i.e. one packed FMA of the pair
// G2[t] = (g[t], g[t - 1]) (LDS, one broadcast ds_read_b64 per tap for all four rows) with the splat (p, p):
//     B[s0]     accumulates g[0] p[0], g[1] p[1], ...            -- the order of blur_h4 / blur_h4_stream
//     B[s0 + 1] accumulates g[0] p[1], g[1] p[2], ...            -- likewise
// G2[0] = (g[0], 0), G2[ksize] = (0, g[ksize - 1]) and the padding up to a multiple of four taps is (0, 0): a fused multiply-add
// with a zero tap returns its accumulator unchanged (pixels are finite, the accumulators start at +0), so the bits are those of the
// other forms (tests/test_gpu_flow.py: fused == two-pass == oracle).
typedef float mav_f2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ int tap_pairs_padded(int ksize) { return (ksize + 1 + 3) & ~3; }
static __device__ __forceinline__ void stage_tap_pairs(const BlurParams& bp, mav_f2* __restrict__ G2)
{
    const int n = tap_pairs_padded(bp.ksize);
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        mav_f2 v;
        v.x = t < bp.ksize ? bp.g[t] : 0.f;
        v.y = (t >= 1 && t <= bp.ksize) ? bp.g[t - 1] : 0.f;
        G2[t] = v;
    }
}
template <typename WordFn>
static __device__ __forceinline__ void blur_h4_pairs(WordFn rd, int off, float f, int ksize, const mav_f2* __restrict__ G2, float out[4])
{
    const int w0 = off >> 2;
    const unsigned sh = (unsigned)(off & 3);
    const int n_words = tap_pairs_padded(ksize) >> 2;
    mav_f2 acc[4];
    uint32_t lo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { acc[k] = (mav_f2)(0.f, 0.f); lo[k] = rd(k, w0); }
    for (int c = 0; c < n_words; c++) {
        uint32_t cur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t hi = rd(k, w0 + c + 1);              // (one word of slack behind every staged row)
            cur[k] = __builtin_amdgcn_alignbyte(hi, lo[k], sh);
            lo[k] = hi;
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const mav_f2 G = G2[4 * c + b];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float p = (float)((cur[k] >> (8 * b)) & 0xffu);
                acc[k] = __builtin_elementwise_fma(G, (mav_f2)(p, p), acc[k]);
            }
        }
    }
    const float a0 = 1.f - f;
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = fmaf(acc[k].y, f, acc[k].x * a0);
}

// Source rows [y0, y0 + n_rows) (row index clamped to the image), byte columns [xb, xb + 4 words) of one image -> LDS, row pitch
// pitch_w dwords; xb is a multiple of 4 (possibly negative).  Columns outside the image hold their BORDER_REFLECT_101 pixels.
// Dword-addressable rows: coalesced dword loads; frames whose width is no multiple of 4: byte by byte.
template <int NB = 8>      // NB loads of a thread in flight before the first LDS store
static __device__ __forceinline__ void stage_rows(const uint8_t* __restrict__ base, int W, int H, int y0, int n_rows, int xb, int words,
                                                  bool dword_ok, uint32_t* __restrict__ sw, int pitch_w)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t* sb = (uint8_t*)sw;
    if (dword_ok) {
        // words inside the image as dwords (xb and W are multiples of 4); the few columns outside it byte by byte, reflected.
        // A wave's (row, 64-word chunk) items go eight at a time: eight independent loads in flight, then eight LDS stores (a plain
        // load -> store loop serialises one memory round trip per row: 11 of them per wave for the preset's layer 1).
        const int wlo = xb < 0 ? (-xb) >> 2 : 0, whi = min(words, (W - xb) >> 2);
        const int n_left = 4 * wlo, n_out = n_left + 4 * max(words - whi, 0);
        const int n_chunks = (whi - wlo + 63) >> 6, rows_w = n_rows > wv ? (n_rows - wv + 3) >> 2 : 0;
        const int n_items = rows_w * n_chunks;
        for (int it0 = 0; it0 < n_items; it0 += NB) {
            uint32_t v[NB];
            int dst[NB];
#pragma unroll
            for (int u = 0; u < NB; u++) {
                const int it = it0 + u;
                const int ri = n_chunks == 1 ? it : it / n_chunks, ch = it - ri * n_chunks;
                const int row = wv + 4 * ri, wd = wlo + 64 * ch + lane;
                const bool ok = it < n_items && wd < whi;
                dst[u] = ok ? row * pitch_w + wd : -1;
                const uint32_t* g = (const uint32_t*)(base + (size_t)min(y0 + (ok ? row : 0), H - 1) * W + xb);
                v[u] = g[ok ? wd : wlo];
            }
#pragma unroll
            for (int u = 0; u < NB; u++)
                if (dst[u] >= 0) sw[dst[u]] = v[u];
        }
        if (n_out > 0)
            for (int row = wv; row < n_rows; row += 4) {
                const uint8_t* p = base + (size_t)min(y0 + row, H - 1) * W;
                for (int j = lane; j < n_out; j += 64) {
                    const int col = j < n_left ? j : 4 * whi + (j - n_left);
                    sb[row * pitch_w * 4 + col] = p[reflect101d(xb + col, W)];
                }
            }
    } else {
        for (int row = wv; row < n_rows; row += 4) {
            const uint8_t* p = base + (size_t)min(y0 + row, H - 1) * W;
            for (int j = lane; j < 4 * words; j += 64) sb[row * pitch_w * 4 + j] = p[reflect101d(xb + j, W)];
        }
    }
}
// the source columns the 64 destination columns from dx0 on need: first staged byte column (multiple of 4) and dwords per row.
// s0 = this lane's source column (lane = destination column dx0 + lane, clamped to w - 1): the tile's first and last source
// columns are lane 0's and lane 63's, read across the wave instead of evaluating resize_coord (double arithmetic) twice more.
static __device__ __forceinline__ void tile_columns(int s0, const BlurParams& bp, int* xb, int* words)
{
    const int c0 = __builtin_amdgcn_readlane(s0, 0), c1 = __builtin_amdgcn_readlane(s0, 63);
    const int r = bp.ksize >> 1;
    *xb = (c0 - r) & ~3;
    *words = ((c1 + 1 + r - *xb) >> 2) + 2;                  // + 1: blur_h4_stream reads one word ahead
}

// Vertical pass at destination row (s0, f): row(y) = the horizontal pass's value at source row y of this thread's column.
template <typename RowFn>
static __device__ __forceinline__ float blur_v1(RowFn row, int H, int s0, float f, const BlurParams& bp)
{
    const int r = bp.ksize >> 1;
    const int s1 = s0 + 1 < H ? s0 + 1 : s0;
    float b0 = 0.f, b1 = 0.f;
    if (s0 - r >= 0 && s0 + 1 + r < H) {
        if (bp.ksize == 5) {                               // the six rows of the thread requested together
            float px[6];
#pragma unroll
            for (int t = 0; t < 6; t++) px[t] = row(s0 - 2 + t);
#pragma unroll
            for (int t = 0; t < 5; t++) { b0 = fmaf(bp.g[t], px[t], b0); b1 = fmaf(bp.g[t], px[t + 1], b1); }
        } else {
            float prev = row(s0 - r);
#pragma unroll 8
            for (int t = 0; t < bp.ksize; t++) {
                const float g = bp.g[t];
                const float nxt = row(s0 - r + t + 1);
                b0 = fmaf(g, prev, b0);
                b1 = fmaf(g, nxt, b1);
                prev = nxt;
            }
        }
    } else {
        for (int t = 0; t < bp.ksize; t++) {
            const float g = bp.g[t];
            b0 = fmaf(g, row(reflect101d(s0 - r + t, H)), b0);
            b1 = fmaf(g, row(reflect101d(s1 - r + t, H)), b1);
        }
    }
    return fmaf(b1, f, b0 * (1.f - f));
}

// direct form (no LDS): every thread walks its own bytes in global memory.  Fallback for scales whose staged rows would not fit LDS.
__global__ __launch_bounds__(256) void k_blur_resize_h_direct(const uint8_t* __restrict__ img, const uint8_t* __restrict__ img2, int split,
                                                              size_t img_stride, int W, int H, int w, BlurParams bp,
                                                              float* __restrict__ tmp, size_t tmp_stride)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 4;
    if (dx >= w || y0 >= H) return;
    const uint8_t* base = image_of(img, img2, split, img_stride, blockIdx.z);
    int s0; float f;
    resize_coord(dx, W, w, bp.scale_x, &s0, &f);
    float o[4];
    blur_h4(base, W, H, y0, s0, f, bp, o);
    float* dst = tmp + (size_t)blockIdx.z * tmp_stride + dx;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (y0 + k < H) dst[(size_t)(y0 + k) * w] = o[k];
}
// Staged form: a workgroup owns 64 destination columns x rows_blk source rows; the u8 bytes those need go through LDS once
// (coalesced dword loads; a thread of the direct form issues (ksize + 1) / 4 dependent dword loads per row from an address of its
// own -- 24 round trips for the 95-tap layer of the 4K preset -- and neighbouring lanes re-read each other's bytes).
__global__ __launch_bounds__(256) void k_blur_resize_h(const uint8_t* __restrict__ img, const uint8_t* __restrict__ img2, int split,
                                                       size_t img_stride, int W, int H, int w, BlurParams bp,
                                                       float* __restrict__ tmp, size_t tmp_stride, int rows_blk, int pitch_w, int dword_ok)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t srows[];         // [rows_blk][pitch_w], then the tap pairs (stage_tap_pairs)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int dx0 = blockIdx.x * 64;
    const uint8_t* base = image_of(img, img2, split, img_stride, blockIdx.z);
    const int dx = dx0 + lane;
    mav_f2* G2 = (mav_f2*)(srows + ((rows_blk * pitch_w + 1) & ~1));
    stage_tap_pairs(bp, G2);                                                 // (visible after the first barrier below)
    int s0; float f;
    if (bp.xs) { s0 = bp.xs[min(dx, w - 1)]; f = bp.xf[min(dx, w - 1)]; }
    else resize_coord(min(dx, w - 1), W, w, bp.scale_x, &s0, &f);
    int xb, words;
    tile_columns(s0, bp, &xb, &words);
    const int off = s0 - (bp.ksize >> 1) - xb;
    float* dst = tmp + (size_t)blockIdx.z * tmp_stride + dx;
    // a workgroup walks every gridDim.y-th block of rows (the launcher gives every block its own workgroup)
    for (int yb = blockIdx.y; yb * rows_blk < H; yb += gridDim.y) {
        const int y0 = yb * rows_blk;
        const int n_rows = min(rows_blk, H - y0);
        if (yb != (int)blockIdx.y) __syncthreads();             // everybody has finished reading the previous block's rows
        // (24 instead of 8 staging loads in flight per thread: 161 / 175 vs 129 / 169 us per launch -- the registers cost occupancy)
        stage_rows(base, W, H, y0, n_rows, xb, min(words, pitch_w), dword_ok != 0, srows, pitch_w);
        __syncthreads();
        for (int i = wv * 4; i < n_rows; i += 16) {
            float o[4];
            blur_h4_pairs([&](int k, int wi) { return srows[min(i + k, n_rows - 1) * pitch_w + wi]; }, off, f, bp.ksize, G2, o);
            if (dx < w) {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (i + k < n_rows) dst[(size_t)(y0 + i + k) * w] = o[k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_blur_resize_v(const float* __restrict__ tmp, size_t tmp_stride, int H, int w, int h,
                                                       BlurParams bp, float* __restrict__ out, size_t out_stride)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= w || dy >= h) return;
    const float* src = tmp + (size_t)blockIdx.z * tmp_stride + dx;
    int s0; float f;
    resize_coord(dy, H, h, bp.scale_y, &s0, &f);
    out[(size_t)blockIdx.z * out_stride + (size_t)dy * w + dx] = blur_v1([&](int y) { return src[(size_t)y * w]; }, H, s0, f, bp);
}

// Both passes in ONE kernel for the layers whose Gaussian is short (ksize <= 13: layer 1 of the reference's preset, layers 1 - 2
// of the 4K / 5-layer one): a workgroup owns a 64 x FB_TH tile of the layer, runs the horizontal pass over the source rows the
// tile's vertical pass will read -- [s0(first row) - r, s0(last row) + 1 + r] clamped to the image; a reflected row index always
// falls inside that range -- into LDS (rows x 64 f32), and filters / resamples vertically from there.  The H x w f32 scratch of
// the two-pass form (written once, read ~(ksize + 1) h / H times: 2.95x the stage's algorithmic bytes) never exists.  Same
// functions, same tap order: bit-identical to the two-pass form (tests/test_gpu_flow.py).  The rows two vertically adjacent
// tiles share (2r + 1 of ~40 at scale 0.4) are filtered twice.
#define FB_TH 16
// pitch_w > 0: the tile's u8 source region (rows [ylo, yhi], the columns its 64 destination columns need) goes through LDS first
// (stage_rows: ~7 coalesced dword loads per thread instead of ~70 single-byte loads) and the horizontal pass reads it from there;
// pitch_w == 0 (regions too large for LDS): every thread reads its bytes from global memory.
// one 64 x FB_TH tile of one image: base = the u8 frame, out = the layer image of that frame
static __device__ __forceinline__ void blur_fused_tile(const uint8_t* __restrict__ base, float* __restrict__ out, int W, int H, int w, int h,
                                                       const BlurParams& bp, int rows_cap, int pitch_w, int dword_ok, int tile_x, int tile_y,
                                                       float* __restrict__ hrows)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // (Measured and not kept: a 1-D grid in tile_of_block's XCD-aware order, so that the 128-byte lines horizontally adjacent tiles
    // share are re-read from one XCD's L2 -- 72.0 vs 69.6 us per launch; workgroups that walk 4 - 8 tiles -- 104 vs 72 us.  PMC says
    // why: the kernel issues ~1000 VALU instructions per wave, index arithmetic and double-precision resize coordinates as much as
    // filter taps, so it is instruction-bound.  Hence: the tile's first / last source column come from lanes 0 / 63 of the wave's own
    // coordinates, and the 16 destination rows' source coordinates are evaluated once per wave, by lanes 0 - 15.)
    const int dx0 = tile_x * 64;
    const int dx = dx0 + lane, dxc = min(dx, w - 1);
    const int r = bp.ksize >> 1;
    int s0; float f;
    resize_coord(dxc, W, w, bp.scale_x, &s0, &f);
    int xb, words;
    tile_columns(s0, bp, &xb, &words);
    const int off = s0 - r - xb;
    uint32_t* sw = (uint32_t*)(hrows + rows_cap * 64);
    const int dy0 = tile_y * FB_TH, dy1 = min(dy0 + FB_TH, h) - 1;
    int row_s; float row_f;                                              // lane L < 16: source row / weight of destination row dy0 + L
    resize_coord(min(dy0 + (lane & (FB_TH - 1)), h - 1), H, h, bp.scale_y, &row_s, &row_f);
    const int sa = __builtin_amdgcn_readlane(row_s, 0), sb = __builtin_amdgcn_readlane(row_s, FB_TH - 1);    // (rows past h - 1 clamp to it)
    const int ylo = max(sa - r, 0), yhi = min(sb + 1 + r, H - 1);
    const int n_rows = min(yhi - ylo + 1, rows_cap);                      // (the host sized rows_cap for the worst tile)
    if (pitch_w > 0) {
        stage_rows(base, W, H, ylo, n_rows, xb, min(words, pitch_w), dword_ok != 0, sw, pitch_w);
        __syncthreads();
        for (int i = wv * 4; i < n_rows; i += 16) {
            float o[4];
            blur_h4_stream([&](int k, int wi) { return sw[min(i + k, n_rows - 1) * pitch_w + wi]; }, off, f, bp, o);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k < n_rows) hrows[(i + k) * 64 + lane] = o[k];
        }
    } else {
        for (int i = wv * 4; i < n_rows; i += 16) {
            float o[4];
            blur_h4(base, W, H, ylo + i, s0, f, bp, o);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (i + k < n_rows) hrows[(i + k) * 64 + lane] = o[k];
        }
    }
    __syncthreads();
    const float* col = hrows + lane - ylo * 64;
#pragma unroll
    for (int j = 0; j < FB_TH / 4; j++) {
        const int dy = dy0 + j * 4 + wv;                                  // wave-uniform
        if (dy > dy1) break;
        const int t0 = __builtin_amdgcn_readlane(row_s, j * 4 + wv);
        const float tf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row_f), j * 4 + wv));
        const float v = blur_v1([&](int y) { return col[y * 64]; }, H, t0, tf, bp);
        if (dx < w) out[(size_t)dy * w + dx] = v;
    }
}

// The fused tile again, for the cases that carry the load (ksize 5 and 13 -- layer 1 of the reference's preset, layers 1 - 2 of the
// 4K / 5-layer one; dword-addressable frames; coordinate tables present): same H-pass / V-pass functions on the same values as
// blur_fused_tile, hence the same bits, with the bookkeeping around them cut down -- round 3's form issued ~1000 VALU + ~520 SALU
// instructions per wave for ~190 filter FMAs per thread:
//   * the resize coordinates come from the layer's tables (bp.xs / xf / ys / yf) instead of double-precision arithmetic per thread;
//   * KS is a template argument of the WHOLE tile (no run-time dispatch inside the loops; the V pass unrolls for 13 taps too);
//   * staging: a wave owns rows wv, wv + 4, ...; row addresses are wave-uniform (scalar), a lane adds its word offset; six rows'
//     loads in flight; every staged row lies inside the image (ylo + row <= yhi <= H - 1), so no clamping;
//   * the H pass does not clamp its row index: rows past n_rows (at most 3, LDS the launch allocates) are computed and not stored;
//   * interior tiles (every row's taps inside the image) take a V pass without the reflect-101 tests.
// The (b0, b1)-pair form of blur_h4_pairs for a Gaussian whose length is a template argument (the fused tile: 5 and 13 taps): the
// KS + 1 tap pairs (g[t], g[t - 1]) are wave-uniform values the caller builds once per thread (TapPairs), byte t of a row costs one
// conversion and one packed FMA.  Same accumulation order as blur_h4_stream: same bits.
template <int KS> struct TapPairs {
    mav_f2 G[KS + 1];
    __device__ __forceinline__ void load(const BlurParams& bp)
    {
#pragma unroll
        for (int t = 0; t <= KS; t++) { G[t].x = t < KS ? bp.g[t] : 0.f; G[t].y = t >= 1 ? bp.g[t - 1] : 0.f; }
    }
};
template <int KS, typename WordFn>
static __device__ __forceinline__ void blur_h4_pairs_t(WordFn rd, int off, float f, const TapPairs<KS>& tp, float out[4])
{
    const int w0 = off >> 2;
    const unsigned sh = (unsigned)(off & 3);
    constexpr int NW = (KS + 1 + 3) / 4;
    mav_f2 acc[4];
    uint32_t lo[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { acc[k] = (mav_f2)(0.f, 0.f); lo[k] = rd(k, w0); }
#pragma unroll
    for (int c = 0; c < NW; c++) {
        uint32_t cur[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t hi = rd(k, w0 + c + 1);
            cur[k] = __builtin_amdgcn_alignbyte(hi, lo[k], sh);
            lo[k] = hi;
        }
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if (4 * c + b > KS) break;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float p = (float)((cur[k] >> (8 * b)) & 0xffu);
                acc[k] = __builtin_elementwise_fma(tp.G[4 * c + b], (mav_f2)(p, p), acc[k]);
            }
        }
    }
    const float a0 = 1.f - f;
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = fmaf(acc[k].y, f, acc[k].x * a0);
}
template <int KS>
static __device__ __forceinline__ float blur_v1_interior(const float* __restrict__ col, int s0, float f, const TapPairs<KS>& tp, const BlurParams& bp)
{
    constexpr int r = KS >> 1;
    float px[KS + 1];
#pragma unroll
    for (int t = 0; t <= KS; t++) px[t] = col[(s0 - r + t) * 64];
    if constexpr (KS >= 13) {
        mav_f2 acc = (mav_f2)(0.f, 0.f);                                     // (b0, b1): row t feeds b0 with g[t] and b1 with g[t - 1]
#pragma unroll
        for (int t = 0; t <= KS; t++) acc = __builtin_elementwise_fma(tp.G[t], (mav_f2)(px[t], px[t]), acc);
        return fmaf(acc.y, f, acc.x * (1.f - f));
    } else {
        float b0 = 0.f, b1 = 0.f;
#pragma unroll
        for (int t = 0; t < KS; t++) { const float g = bp.g[t]; b0 = fmaf(g, px[t], b0); b1 = fmaf(g, px[t + 1], b1); }
        return fmaf(b1, f, b0 * (1.f - f));
    }
}
template <int KS, int TH>
static __device__ __forceinline__ void blur_fused_tile_fast(const uint8_t* __restrict__ base, float* __restrict__ out, int W, int H, int w, int h,
                                                            const BlurParams& bp, int rows_cap, int pitch_w, int tile_x, int tile_y,
                                                            float* __restrict__ hrows)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int r = KS >> 1;
    TapPairs<KS> tp;
    if constexpr (KS >= 13) tp.load(bp);                                   // (the pair form: 13 taps only, see below)
    const int dx = tile_x * 64 + lane, dxc = min(dx, w - 1);
    const int dy0 = tile_y * TH, dy1 = min(dy0 + TH, h) - 1;
    const int dyc = min(dy0 + (lane & (TH - 1)), h - 1);
    const int s0 = bp.xs[dxc];
    const float f = bp.xf[dxc];
    const int row_s = bp.ys[dyc];                                           // lane L < TH: source row / weight of destination row dy0 + L
    const float row_f = bp.yf[dyc];
    int xb, words;
    tile_columns(s0, bp, &xb, &words);
    words = min(words, pitch_w);
    const int off = s0 - r - xb;
    uint32_t* sw = (uint32_t*)(hrows + rows_cap * 64);
    const int sa = __builtin_amdgcn_readlane(row_s, 0), sb = __builtin_amdgcn_readlane(row_s, TH - 1);
    const int ylo = max(sa - r, 0), yhi = min(sb + 1 + r, H - 1);
    const int n_rows = min(yhi - ylo + 1, rows_cap);
    // ---- stage the u8 source rows [ylo, ylo + n_rows), byte columns [xb, xb + 4 words)
    {
        const int wlo = xb < 0 ? (-xb) >> 2 : 0, whi = min(words, (W - xb) >> 2);
        const int nw = whi - wlo;                                            // dwords of a row that lie inside the image
        const uint8_t* g0 = base + (size_t)ylo * W + (xb + 4 * wlo);         // wave-uniform
        uint32_t* s00 = sw + wlo;
        for (int c0 = 0; c0 < nw; c0 += 64) {
            const int wd = c0 + lane;
            const bool ok = wd < nw;
            const unsigned lo = 4u * (unsigned)(ok ? wd : 0);
            for (int row0 = wv; row0 < n_rows; row0 += 24) {
                uint32_t v[6];
#pragma unroll
                for (int u = 0; u < 6; u++) {
                    const int row = min(row0 + 4 * u, n_rows - 1);           // (wave-uniform)
                    v[u] = *(const uint32_t*)(g0 + ((unsigned)(row * W) + lo));
                }
#pragma unroll
                for (int u = 0; u < 6; u++) {
                    const int row = row0 + 4 * u;
                    if (ok && row < n_rows) s00[row * pitch_w + wd] = v[u];
                }
            }
        }
        const int n_left = 4 * wlo, n_out = n_left + 4 * max(words - whi, 0);
        if (n_out > 0) {                                                     // the few columns outside the image, reflected, byte by byte
            uint8_t* sb8 = (uint8_t*)sw;
            for (int row = wv; row < n_rows; row += 4) {
                const uint8_t* p = base + (size_t)(ylo + row) * W;
                for (int j = lane; j < n_out; j += 64) {
                    const int col = j < n_left ? j : 4 * whi + (j - n_left);
                    sb8[row * pitch_w * 4 + col] = p[reflect101d(xb + col, W)];
                }
            }
        }
    }
    __syncthreads();
    // ---- horizontal pass: staged bytes -> hrows (n_rows x 64 f32)
    for (int i = wv * 4; i < n_rows; i += 16) {
        float o[4];
        const uint32_t* rw = sw + i * pitch_w;
        // (the pair form pays for 13 taps -- 184 vs 194 us per launch for layer 2 of the 4K preset -- and costs for 5: 64 vs 53 us per
        // launch for layer 1 at 1080p, where the compiler's own mix of scalar-tap FMAs is shorter)
        if constexpr (KS >= 13) blur_h4_pairs_t<KS>([&](int k, int wi) { return rw[k * pitch_w + wi]; }, off, f, tp, o);
        else blur_h4_stream_t<KS>([&](int k, int wi) { return rw[k * pitch_w + wi]; }, off, f, bp, o);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i + k < n_rows) hrows[(i + k) * 64 + lane] = o[k];
    }
    __syncthreads();
    // ---- vertical pass
    const float* col = hrows + lane - ylo * 64;
    const bool interior = sa - r >= 0 && sb + 1 + r < H;                    // wave-uniform: no row of the tile touches the border
#pragma unroll
    for (int j = 0; j < TH / 4; j++) {
        const int dy = dy0 + j * 4 + wv;                                     // wave-uniform
        if (dy > dy1) break;
        const int t0 = __builtin_amdgcn_readlane(row_s, j * 4 + wv);
        const float tf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row_f), j * 4 + wv));
        const float v = interior ? blur_v1_interior<KS>(col, t0, tf, tp, bp) : blur_v1([&](int y) { return col[y * 64]; }, H, t0, tf, bp);
        if (dx < w) out[(size_t)dy * w + dx] = v;
    }
}
// does a fused layer take the fast tile?  (dword-addressable frames, staged form, tables, ksize 5 or 13)
static __host__ __device__ __forceinline__ bool fused_fast_ok(const BlurParams& bp, int pitch_w, int dword_ok)
{
    return dword_ok && pitch_w > 0 && bp.xs != nullptr && (bp.ksize == 5 || bp.ksize == 13);
}
// Which tile code a fused launch carries.  One kernel per PATH: the paths differ a lot in registers (the generic tile and the 13-tap
// pair form want ~80 VGPRs, the 5-tap fast tile 44), and a kernel that contained them all ran the 5-tap layers at the occupancy of
// the hungriest (66 vs 53 us per launch for layer 1 at 1080p).
enum { FP_GENERIC = 0, FP_FAST5 = 1, FP_FAST13 = 2, FP_ANY = 3 };
static int fused_path_of(const BlurParams& bp, int pitch_w, int dword_ok)
{
    return !fused_fast_ok(bp, pitch_w, dword_ok) ? FP_GENERIC : (bp.ksize == 5 ? FP_FAST5 : FP_FAST13);
}
template <int PATH>
static __device__ __forceinline__ void blur_fused_any(const uint8_t* __restrict__ base, float* __restrict__ out, int W, int H, int w, int h,
                                                      const BlurParams& bp, int rows_cap, int pitch_w, int dword_ok, int th, int tile_x, int tile_y,
                                                      float* __restrict__ hrows)
{
    // th = tile height chosen by the host (fused_plan): 16, or 8 where 16 rows' source region does not fit LDS (fast tile only)
    const bool fast = PATH == FP_FAST5 || PATH == FP_FAST13 || (PATH == FP_ANY && fused_fast_ok(bp, pitch_w, dword_ok));
    if ((PATH == FP_FAST5 || PATH == FP_ANY) && fast && bp.ksize == 5) {
        if (th == 16) blur_fused_tile_fast<5, 16>(base, out, W, H, w, h, bp, rows_cap, pitch_w, tile_x, tile_y, hrows);
        else blur_fused_tile_fast<5, 8>(base, out, W, H, w, h, bp, rows_cap, pitch_w, tile_x, tile_y, hrows);
        return;
    }
    if ((PATH == FP_FAST13 || PATH == FP_ANY) && fast && bp.ksize == 13) {
        if (th == 16) blur_fused_tile_fast<13, 16>(base, out, W, H, w, h, bp, rows_cap, pitch_w, tile_x, tile_y, hrows);
        else blur_fused_tile_fast<13, 8>(base, out, W, H, w, h, bp, rows_cap, pitch_w, tile_x, tile_y, hrows);
        return;
    }
    if (PATH == FP_GENERIC || PATH == FP_ANY)
        blur_fused_tile(base, out, W, H, w, h, bp, rows_cap, pitch_w, dword_ok, tile_x, tile_y, hrows);
}
template <int PATH>
__global__ __launch_bounds__(256) void k_blur_resize_fused(const uint8_t* __restrict__ img, const uint8_t* __restrict__ img2, int split,
                                                           size_t img_stride, int W, int H, int w, int h, BlurParams bp,
                                                           float* __restrict__ out, size_t out_stride, int rows_cap, int pitch_w, int dword_ok, int th)
{
    extern __shared__ __attribute__((aligned(16))) float hrows[];       // [rows_cap][64] f32, then [rows_cap + 3][pitch_w] dwords of u8
    blur_fused_any<PATH>(image_of(img, img2, split, img_stride, blockIdx.z), out + (size_t)blockIdx.z * out_stride, W, H, w, h, bp, rows_cap,
                         pitch_w, dword_ok, th, blockIdx.x, blockIdx.y, hrows);
}

static int fused_blur_rows(int H, int h, int ksize, int th = FB_TH) { return (int)((th - 1) * ((double)H / h)) + (ksize | 1) + 4; }
// LDS of a fused tile: rows x 64 f32 of horizontal-pass results, then the staged u8 rows (three rows of slack: the fast tile's
// horizontal pass works on whole groups of four rows)
static size_t fused_lds_bytes(int rows, int pitch_w) { return (size_t)rows * 256 + (size_t)(rows + 3) * 4 * pitch_w; }
// dwords per staged source row: the columns 64 destination pixels need (63 scale + 1 + ksize), the 4-alignment slack and the word
// blur_h4_stream reads ahead
static int staged_pitch_words(int W, int w, int ksize) { return ((int)(63 * ((double)W / w)) + (ksize | 1) + 2 + 3) / 4 + 3; }
// How a fused layer's tiles are cut: 64 x 16 with the source region staged in LDS when that fits 64 KB; else, for the fast tile
// (fused_fast_ok), 64 x 8 staged (layer 2 of the 4K / 5-layer preset: 13 taps at scale 6.25 -- 110 source rows of 420 bytes for 16
// destination rows); else 64 x 16 unstaged (every thread reads its bytes from global memory).
struct FusedPlan { int th, rows, pitch_w; size_t lds; };
static FusedPlan fused_plan(int W, int H, int w, int h, const BlurParams& bp, int dword_ok)
{
    const int pitch = staged_pitch_words(W, w, bp.ksize);
    FusedPlan p{FB_TH, fused_blur_rows(H, h, bp.ksize), pitch, 0};
    if (fused_lds_bytes(p.rows, pitch) > 64 * 1024) {
        const int rows8 = fused_blur_rows(H, h, bp.ksize, 8);
        if (fused_fast_ok(bp, pitch, dword_ok) && fused_lds_bytes(rows8, pitch) <= 64 * 1024) { p.th = 8; p.rows = rows8; }
        else p.pitch_w = 0;
    }
    p.lds = fused_lds_bytes(p.rows, p.pitch_w);
    return p;
}
bool blur_resize_is_fused(int W, int H, int w, int h, int ksize)
{
    return !(w == W && h == H) && ksize <= 13 && H > 2 * ksize && W > 2 * ksize && (size_t)fused_blur_rows(H, h, ksize) * 256 <= 48 * 1024;
}

// Layer 0 (scale 1, sigma 0 -> fixed kernel [1/4, 1/2, 1/4], BORDER_REFLECT_101): every product is exact in f32,
// so this is bit-identical to OpenCV's row-then-column filter.  One thread = 4 consecutive pixels x 4 rows:
// six aligned u32 row loads + the two neighbour bytes per row, float4 stores.
static __device__ __forceinline__ void blur3_block(const uint8_t* __restrict__ src, float* __restrict__ dst, int W, int H, int bx, int by)
{
    const int x = (bx * 64 + (threadIdx.x & 63)) * 4;
    const int y0 = (by * 4 + (threadIdx.x >> 6)) * 4;
    if (x >= W || y0 >= H) return;
    const int xl = x == 0 ? 1 : x - 1;                    // reflect101
    const int xr = x + 4 >= W ? W - 2 : x + 4;
    // all 18 loads of the thread first (six rows x {left byte, aligned dword, right byte}); the outputs below are computed
    // unconditionally and only the stores are predicated, so no load is deferred behind a branch
    uint32_t qs[6];
    uint8_t ls[6], rs[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        int yy = y0 - 1 + r;
        yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);         // rows past the image (tail of the last group): any valid row
        const uint8_t* p = src + (size_t)yy * W;
        qs[r] = *(const uint32_t*)(p + x);
        ls[r] = p[xl];
        rs[r] = p[xr];
    }
    float hrow[6][4];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const uint32_t q = qs[r];
        const float a = (float)ls[r], b0 = (float)(q & 255u), b1 = (float)((q >> 8) & 255u), b2 = (float)((q >> 16) & 255u),
                    b3 = (float)(q >> 24), c = (float)rs[r];
        hrow[r][0] = 0.5f * b0 + 0.25f * (a + b1);
        hrow[r][1] = 0.5f * b1 + 0.25f * (b0 + b2);
        hrow[r][2] = 0.5f * b2 + 0.25f * (b1 + b3);
        hrow[r][3] = 0.5f * b3 + 0.25f * (b2 + c);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float4 o;
        o.x = 0.5f * hrow[r + 1][0] + 0.25f * (hrow[r][0] + hrow[r + 2][0]);
        o.y = 0.5f * hrow[r + 1][1] + 0.25f * (hrow[r][1] + hrow[r + 2][1]);
        o.z = 0.5f * hrow[r + 1][2] + 0.25f * (hrow[r][2] + hrow[r + 2][2]);
        o.w = 0.5f * hrow[r + 1][3] + 0.25f * (hrow[r][3] + hrow[r + 2][3]);
        if (y0 + r < H) *(float4*)(dst + (size_t)(y0 + r) * W + x) = o;
    }
}

// The layer images of SEVERAL layers in one launch (a small group's pyramid: launch_blur_multi): job 0.. = layers whose blur is the
// 3x3 form (layer 0) or the fused form; workgroup b belongs to the last job whose first_block is <= b.
template <int PATH>      // the tile code its fused jobs need (FP_ANY when they differ)
__global__ __launch_bounds__(256) void k_blur_multi(const uint8_t* __restrict__ img, const uint8_t* __restrict__ img2, int split, size_t img_stride,
                                                    int W, int H, int dword_ok, BlurJobs jobs)
{
    extern __shared__ __attribute__((aligned(16))) float hrows[];
    int k = 0;
    for (int i = 1; i < jobs.n; i++) k = (int)blockIdx.x >= jobs.j[i].first_block ? i : k;
    const BlurJob J = jobs.j[k];
    const int b = (int)blockIdx.x - J.first_block;
    const int z = b / (J.gx * J.gy), rem = b - z * J.gx * J.gy;
    const int by = rem / J.gx, bx = rem - by * J.gx;
    const uint8_t* base = image_of(img, img2, split, img_stride, z);
    float* out = J.out + (size_t)z * J.out_stride;
    if (J.fused) blur_fused_any<PATH>(base, out, W, H, J.w, J.h, J.bp, J.rows_cap, J.pitch_w, dword_ok, J.th, bx, by, hrows);
    else blur3_block(base, out, W, H, bx, by);
}
// Layer images of several layers of G frames in ONE launch.  jobs[i]: layer size, BlurParams, out / out_stride filled by the caller; a
// layer qualifies when blur_multi_ok says so (3x3 form or fused form).  Grid bookkeeping and the LDS size are filled here.
static bool blur3_fast_ok(const uint8_t* img, const uint8_t* img2, size_t img_stride, int W, int H, int w, int h, const BlurParams& bp,
                          const float* out, size_t out_stride);
bool blur_multi_ok(const uint8_t* img, const uint8_t* img2, size_t img_stride, int W, int H, int w, int h, BlurParams bp, const float* out,
                   size_t out_stride)
{
    return blur3_fast_ok(img, img2 ? img2 : img, img_stride, W, H, w, h, bp, out, out_stride) || blur_resize_is_fused(W, H, w, h, bp.ksize);
}
void launch_blur_multi(hipStream_t st, const uint8_t* img, const uint8_t* img2, int split, size_t img_stride, int G, int W, int H, BlurJobs jobs,
                       bool split_by_path)
{
    if (!img2) { img2 = img; split = G; }
    const int dword_ok = (W % 4 == 0 && img_stride % 4 == 0 && ((uintptr_t)img & 3) == 0 && ((uintptr_t)img2 & 3) == 0) ? 1 : 0;
    if (split_by_path) {      // many frames per layer (the deep layers of a whole call): one launch per tile code -- a launch that mixes
                              // them runs every layer at the occupancy of the hungriest path -- instead of one launch in all
        bool done[MAV_MAX_JOBS] = {false};
        for (int i = 0; i < jobs.n; i++) {
            if (done[i]) continue;
            auto path_of = [&](const BlurJob& J) {
                if (J.w == W && J.h == H) return -1;
                const FusedPlan fp = fused_plan(W, H, J.w, J.h, J.bp, dword_ok);
                return fused_path_of(J.bp, fp.pitch_w, dword_ok);
            };
            const int p = path_of(jobs.j[i]);
            BlurJobs sub{0, 0, {}};
            for (int k = i; k < jobs.n; k++)
                if (!done[k] && path_of(jobs.j[k]) == p) { sub.j[sub.n++] = jobs.j[k]; done[k] = true; }
            launch_blur_multi(st, img, img2, split, img_stride, G, W, H, sub, false);
        }
        return;
    }
    int blocks = 0;
    size_t lds = 0;
    for (int i = 0; i < jobs.n; i++) {
        BlurJob& J = jobs.j[i];
        J.fused = !(J.w == W && J.h == H);
        if (J.fused) {
            const FusedPlan fp = fused_plan(W, H, J.w, J.h, J.bp, dword_ok);
            J.rows_cap = fp.rows; J.pitch_w = fp.pitch_w; J.th = fp.th;
            J.gx = (J.w + 63) / 64; J.gy = (J.h + fp.th - 1) / fp.th;
            lds = std::max(lds, fp.lds);
        } else {
            J.rows_cap = J.pitch_w = 0;
            J.gx = (W / 4 + 63) / 64; J.gy = ((H + 3) / 4 + 3) / 4;
        }
        J.first_block = blocks;
        blocks += J.gx * J.gy * G;
    }
    int path = -1;                                                          // one tile code for all fused jobs, or FP_ANY
    for (int i = 0; i < jobs.n; i++)
        if (jobs.j[i].fused) {
            const int p = fused_path_of(jobs.j[i].bp, jobs.j[i].pitch_w, dword_ok);
            path = path < 0 ? p : (path == p ? p : FP_ANY);
        }
    switch (path) {
    case FP_FAST13: hipLaunchKernelGGL(k_blur_multi<FP_FAST13>, dim3(blocks), dim3(256), lds, st, img, img2, split, img_stride, W, H, dword_ok, jobs); break;
    case FP_GENERIC: hipLaunchKernelGGL(k_blur_multi<FP_GENERIC>, dim3(blocks), dim3(256), lds, st, img, img2, split, img_stride, W, H, dword_ok, jobs); break;
    case FP_ANY: hipLaunchKernelGGL(k_blur_multi<FP_ANY>, dim3(blocks), dim3(256), lds, st, img, img2, split, img_stride, W, H, dword_ok, jobs); break;
    default: hipLaunchKernelGGL(k_blur_multi<FP_FAST5>, dim3(blocks), dim3(256), lds, st, img, img2, split, img_stride, W, H, dword_ok, jobs); break;   // (also: no fused job)
    }
}

// G images: the first `split` from run img, the rest from run img2 (both with stride img_stride); split >= G: one run.
// two_pass: force the separable two-pass form through the tmp scratch (the stage hook compares the two forms).
__global__ __launch_bounds__(256) void k_blur3_u8(const uint8_t* __restrict__ img, const uint8_t* __restrict__ img2, int split,
                                                  size_t img_stride, int W, int H, float* __restrict__ out, size_t out_stride)
{
    blur3_block(image_of(img, img2, split, img_stride, blockIdx.z), out + (size_t)blockIdx.z * out_stride, W, H, blockIdx.x, blockIdx.y);
}

static bool blur3_fast_ok(const uint8_t* img, const uint8_t* img2, size_t img_stride, int W, int H, int w, int h, const BlurParams& bp,
                          const float* out, size_t out_stride)
{
    return w == W && h == H && bp.fixed3 && W % 4 == 0 && W >= 8 && H >= 2 && img_stride % 4 == 0 && out_stride % 4 == 0 &&
           ((uintptr_t)img & 3) == 0 && ((uintptr_t)img2 & 3) == 0 && ((uintptr_t)out & 15) == 0;
}
// true when launch_blur_resize (two_pass = false) will go through the tmp scratch for these operands
bool blur_resize_needs_tmp(const uint8_t* img, const uint8_t* img2, size_t img_stride, int W, int H, int w, int h, BlurParams bp,
                           const float* out, size_t out_stride)
{
    return !blur3_fast_ok(img, img2 ? img2 : img, img_stride, W, H, w, h, bp, out, out_stride) && !blur_resize_is_fused(W, H, w, h, bp.ksize);
}
void launch_blur_resize(hipStream_t st, const uint8_t* img, const uint8_t* img2, int split, size_t img_stride, int G, int W, int H, int w,
                        int h, BlurParams bp, float* tmp, size_t tmp_stride, float* out, size_t out_stride, bool two_pass)
{
    if (!img2) { img2 = img; split = G; }
    if (blur3_fast_ok(img, img2, img_stride, W, H, w, h, bp, out, out_stride)) {
        dim3 grid((W / 4 + 63) / 64, ((H + 3) / 4 + 3) / 4, G);
        hipLaunchKernelGGL(k_blur3_u8, grid, dim3(256), 0, st, img, img2, split, img_stride, W, H, out, out_stride);
        return;
    }
    // rows are dword-addressable when W, the image stride and both bases are multiples of 4 (else the staging goes byte by byte)
    const int dword_ok = (W % 4 == 0 && img_stride % 4 == 0 && ((uintptr_t)img & 3) == 0 && ((uintptr_t)img2 & 3) == 0) ? 1 : 0;
    const int pitch_w = staged_pitch_words(W, w, bp.ksize);
    if (!two_pass && blur_resize_is_fused(W, H, w, h, bp.ksize)) {
        const FusedPlan fp = fused_plan(W, H, w, h, bp, dword_ok);
        const dim3 grid((w + 63) / 64, (h + fp.th - 1) / fp.th, G);
        switch (fused_path_of(bp, fp.pitch_w, dword_ok)) {
        case FP_FAST5: hipLaunchKernelGGL(k_blur_resize_fused<FP_FAST5>, grid, dim3(256), fp.lds, st, img, img2, split, img_stride, W, H, w, h, bp, out, out_stride, fp.rows, fp.pitch_w, dword_ok, fp.th); break;
        case FP_FAST13: hipLaunchKernelGGL(k_blur_resize_fused<FP_FAST13>, grid, dim3(256), fp.lds, st, img, img2, split, img_stride, W, H, w, h, bp, out, out_stride, fp.rows, fp.pitch_w, dword_ok, fp.th); break;
        default: hipLaunchKernelGGL(k_blur_resize_fused<FP_GENERIC>, grid, dim3(256), fp.lds, st, img, img2, split, img_stride, W, H, w, h, bp, out, out_stride, fp.rows, fp.pitch_w, dword_ok, fp.th); break;
        }
        return;
    }
    int rows_blk = 16;
    while (rows_blk > 4 && (size_t)rows_blk * pitch_w * 4 > 48 * 1024) rows_blk >>= 1;
    if ((size_t)rows_blk * pitch_w * 4 <= 48 * 1024)
    {
        const int gx = (w + 63) / 64, nby = (H + rows_blk - 1) / rows_blk;
        const int gy = nby;                                               // one row block per workgroup (walking several measured slower: 69 vs 47 us)
        const size_t lds_h = (size_t)((rows_blk * pitch_w + 1) & ~1) * 4 + (size_t)((bp.ksize + 4) & ~3) * 8;    // staged rows + tap pairs
        hipLaunchKernelGGL(k_blur_resize_h, dim3(gx, gy, G), dim3(256), lds_h, st, img,
                           img2, split, img_stride, W, H, w, bp, tmp, tmp_stride, rows_blk, pitch_w, dword_ok);
    }
    else
        hipLaunchKernelGGL(k_blur_resize_h_direct, dim3((w + 63) / 64, ((H + 3) / 4 + 3) / 4, G), dim3(256), 0, st, img, img2, split, img_stride,
                           W, H, w, bp, tmp, tmp_stride);
    hipLaunchKernelGGL(k_blur_resize_v, dim3((w + 63) / 64, (h + 3) / 4, G), dim3(256), 0, st, (const float*)tmp, tmp_stride, H, w, h,
                       bp, out, out_stride);
}

// ------------------------------------------------------------------------------------------------------------
// FarnebackPolyExp (A.4).  64x16 output tile per workgroup; the (64+2n) x (16+2n) source tile (edge-clamped,
// which is exactly OpenCV's row clamp + edge-triple replication) is staged in LDS, the vertical pass leaves
// three (64+2n) x 16 planes in LDS and the horizontal pass reads those.  Coefficients sit in kernel arguments
// (scalar loads).  Output: 5 planes.
// ------------------------------------------------------------------------------------------------------------
#define PX 64
#define PY 16
// one 64 x 16 tile of one image: src / dst = the image's layer plane and its five expansion planes
template <int N_T>
static __device__ __forceinline__ void polyexp_tile(const float* __restrict__ src, float* __restrict__ dst, int w, int h, const PolyCoef& pc,
                                                    int tile_x, int tile_y, float* __restrict__ smem)
{
    const int n = N_T > 0 ? N_T : pc.n;
    const int EX = PX + 2 * n, EY = PY + 2 * n;
    float* tile = smem;
    float* v0 = tile + EX * EY;
    float* v1 = v0 + PY * EX;
    float* v2 = v1 + PY * EX;
    const int tid = threadIdx.x;
    const int x0 = tile_x * PX, y0 = tile_y * PY;

    if constexpr (N_T > 0) {
        // all of a thread's loads are issued before the first one is consumed (a load -> LDS-store loop would serialise
        // ten memory round trips per workgroup)
        constexpr int TOT = (PX + 2 * N_T) * (PY + 2 * N_T), CNT = (TOT + 255) / 256;
        float v[CNT];
#pragma unroll
        for (int j = 0; j < CNT; j++) {
            const int i = tid + 256 * j;
            if (i < TOT) {
                const int ly = i / EX, lx = i - ly * EX;
                const int gx = clampi(x0 - n + lx, 0, w - 1), gy = clampi(y0 - n + ly, 0, h - 1);
                v[j] = src[(size_t)gy * w + gx];
            }
        }
#pragma unroll
        for (int j = 0; j < CNT; j++)
            if (tid + 256 * j < TOT) tile[tid + 256 * j] = v[j];
    } else {
        for (int i = tid; i < EX * EY; i += 256) {
            const int ly = i / EX, lx = i - ly * EX;
            const int gx = clampi(x0 - n + lx, 0, w - 1), gy = clampi(y0 - n + ly, 0, h - 1);
            tile[i] = src[(size_t)gy * w + gx];
        }
    }
    __syncthreads();
    for (int i = tid; i < PY * EX; i += 256) {
        const int ly = i / EX, lx = i - ly * EX;
        const float* c = tile + (ly + n) * EX + lx;
        float t0 = c[0] * pc.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 1; k <= n; k++) {               // (explicit fmaf: the bits must not depend on which kernel this is inlined into)
            const float a = c[-k * EX], b = c[k * EX];
            const float p = a + b;
            t0 = fmaf(pc.g[k], p, t0);
            t1 = fmaf(pc.xg[k], b - a, t1);
            t2 = fmaf(pc.xxg[k], p, t2);
        }
        v0[i] = t0; v1[i] = t1; v2[i] = t2;
    }
    __syncthreads();
    const size_t npx = (size_t)w * h;
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
            b1 = fmaf(tg, pc.g[k], b1);
            b4 = fmaf(tg, pc.xxg[k], b4);
            b2 = fmaf(c0 - a0, pc.xg[k], b2);
            b3 = fmaf(c1 + a1, pc.g[k], b3);
            b6 = fmaf(c1 - a1, pc.xg[k], b6);
            b5 = fmaf(c2 + a2, pc.g[k], b5);
        }
        const size_t o = (size_t)gy * w + gx;
        dst[o] = b3 * pc.ig11;
        dst[npx + o] = b2 * pc.ig11;
        dst[2 * npx + o] = fmaf(b5, pc.ig33, b1 * pc.ig03);
        dst[3 * npx + o] = fmaf(b4, pc.ig33, b1 * pc.ig03);
        dst[4 * npx + o] = b6 * pc.ig55;
    }
}

template <int N_T>
__global__ __launch_bounds__(256) void k_polyexp(const float* __restrict__ I, size_t I_stride, int w, int h, PolyCoef pc, TileMap tm,
                                                 float* __restrict__ R, size_t R_stride)
{
    int img_s, tile_x, tile_y;
    if (!tile_of_block(tm, &img_s, &tile_x, &tile_y)) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    polyexp_tile<N_T>(I + (size_t)img_s * I_stride, R + (size_t)img_s * R_stride, w, h, pc, tile_x, tile_y, smem);
}

// The expansions of SEVERAL layers in one launch (a small group's whole pyramid: launch_polyexp_multi): workgroup b belongs to the
// last job whose first_block is <= b and takes that job's tile b - first_block (plain row-major order, image by image).
template <int N_T>
__global__ __launch_bounds__(256) void k_polyexp_multi(PolyJobs jobs, PolyCoef pc)
{
    int k = 0;
    for (int i = 1; i < jobs.n; i++) k = (int)blockIdx.x >= jobs.j[i].first_block ? i : k;
    const PolyJob J = jobs.j[k];
    const int tile = (int)blockIdx.x - J.first_block;
    const int img_s = tile / J.per_img, rem = tile - img_s * J.per_img;
    const int tile_y = rem / J.tiles_x, tile_x = rem - tile_y * J.tiles_x;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    polyexp_tile<N_T>(J.I + (size_t)img_s * J.I_stride, J.R + (size_t)img_s * J.R_stride, J.w, J.h, pc, tile_x, tile_y, smem);
}
// jobs[i]: G images of a (w x h) layer at I (slot stride I_stride) -> R (slot stride R_stride); first_block / tiles_x / per_img are filled here
void launch_polyexp_multi(hipStream_t st, PolyJobs jobs, int G, const PolyCoef& pc)
{
    const int n = pc.n;
    const size_t lds = sizeof(float) * ((size_t)(PX + 2 * n) * (PY + 2 * n) + 3 * (size_t)PY * (PX + 2 * n));
    int blocks = 0;
    for (int i = 0; i < jobs.n; i++) {
        PolyJob& J = jobs.j[i];
        J.tiles_x = (J.w + PX - 1) / PX;
        J.per_img = J.tiles_x * ((J.h + PY - 1) / PY);
        J.first_block = blocks;
        blocks += J.per_img * G;
    }
    if (n == 8) hipLaunchKernelGGL(k_polyexp_multi<8>, dim3(blocks), dim3(256), lds, st, jobs, pc);
    else if (n == 7) hipLaunchKernelGGL(k_polyexp_multi<7>, dim3(blocks), dim3(256), lds, st, jobs, pc);
    else if (n == 5) hipLaunchKernelGGL(k_polyexp_multi<5>, dim3(blocks), dim3(256), lds, st, jobs, pc);
    else hipLaunchKernelGGL(k_polyexp_multi<0>, dim3(blocks), dim3(256), lds, st, jobs, pc);
}

void launch_polyexp(hipStream_t st, const float* I, size_t I_stride, int G, int w, int h, const PolyCoef& pc, float* R,
                    size_t R_stride)
{
    const int n = pc.n;
    const size_t lds = sizeof(float) * ((size_t)(PX + 2 * n) * (PY + 2 * n) + 3 * (size_t)PY * (PX + 2 * n));
    TileMap tm = make_tile_map(w, h, G, PX, PY);
    tm.xcd = 0;        // plain row-major order: measured 3 % faster here than the XCD-banded order (the halo re-reads that miss
                       // L2 hit the Infinity Cache, which all XCDs share; the kernel is VALU / write bound, not read bound)
    const dim3 grid(tile_grid(tm));
    if (n == 8)
        hipLaunchKernelGGL(k_polyexp<8>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, tm, R, R_stride);
    else if (n == 7)
        hipLaunchKernelGGL(k_polyexp<7>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, tm, R, R_stride);
    else if (n == 5)
        hipLaunchKernelGGL(k_polyexp<5>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, tm, R, R_stride);
    else
        hipLaunchKernelGGL(k_polyexp<0>, grid, dim3(256), lds, st, I, I_stride, w, h, pc, tm, R, R_stride);
}

// ------------------------------------------------------------------------------------------------------------
// FarnebackUpdateMatrices for one pixel (A.5), register form: q[] = the 5 R0 values at (x, y), out[] = the 5 M values.
static __device__ __forceinline__ void update_core(const float q[5], const float* __restrict__ R1p, size_t npx, int w, int h,
                                                   int x, int y, float dx, float dy, float out[5])
{
    // Every multiply-add below is spelled out (fmaf) and contraction is off for the rest: the bits must not depend on which kernel
    // (or which template instantiation of one) the function is inlined into -- under -ffp-contract=fast (this file's setting until round 3) two
    // instantiations of the sweep kernel that differed in a store instruction fused `a*b + c*d` differently and their flows drifted
    // apart by up to 8e-4 px over twenty sweeps.  update_finish() below is the same arithmetic on values already gathered.
#pragma clang fp contract(off)
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    fx -= x1; fy -= y1;
    float r2, r3, r4, r5, r6;
    if ((unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1)) {
        const float gx = 1.f - fx, gy = 1.f - fy;
        const float a00 = gx * gy, a01 = fx * gy, a10 = gx * fy, a11 = fx * fy;
        const float* p = R1p + (size_t)y1 * w + x1;
        r2 = fmaf(a11, p[w + 1], fmaf(a10, p[w], fmaf(a01, p[1], a00 * p[0]))); p += npx;
        r3 = fmaf(a11, p[w + 1], fmaf(a10, p[w], fmaf(a01, p[1], a00 * p[0]))); p += npx;
        r4 = fmaf(a11, p[w + 1], fmaf(a10, p[w], fmaf(a01, p[1], a00 * p[0]))); p += npx;
        r5 = fmaf(a11, p[w + 1], fmaf(a10, p[w], fmaf(a01, p[1], a00 * p[0]))); p += npx;
        r6 = fmaf(a11, p[w + 1], fmaf(a10, p[w], fmaf(a01, p[1], a00 * p[0])));
        r4 = (q[2] + r4) * 0.5f;
        r5 = (q[3] + r5) * 0.5f;
        r6 = (q[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = q[2]; r5 = q[3]; r6 = q[4] * 0.5f;
    }
    r2 = (q[0] - r2) * 0.5f;
    r3 = (q[1] - r3) * 0.5f;
    r2 += fmaf(r4, dy, r6 * dx);
    r3 += fmaf(r6, dy, r5 * dx);
    const int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472}
        auto bw = [](int d) { return d < 2 ? 0.14f : 0.4472f; };
        const float scale = (x < BORDER ? bw(x) : 1.f) * (x >= w - BORDER ? bw(w - x - 1) : 1.f) *
                            (y < BORDER ? bw(y) : 1.f) * (y >= h - BORDER ? bw(h - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    out[0] = fmaf(r4, r4, r6 * r6);
    out[1] = (r4 + r5) * r6;
    out[2] = fmaf(r5, r5, r6 * r6);
    out[3] = fmaf(r4, r2, r6 * r3);
    out[4] = fmaf(r6, r2, r5 * r3);
}

// memory form: R0p/R1p/Mp point at plane 0 of the pair; planes are npx apart.
static __device__ __forceinline__ void update_px(const float* __restrict__ R0p, const float* __restrict__ R1p, size_t npx,
                                                 int w, int h, int x, int y, float dx, float dy, float* __restrict__ Mp)
{
    const size_t idx = (size_t)y * w + x;
    float q[5], o[5];
#pragma unroll
    for (int c = 0; c < 5; c++) q[c] = R0p[c * npx + idx];
    update_core(q, R1p, npx, w, h, x, y, dx, dy, o);
#pragma unroll
    for (int c = 0; c < 5; c++) Mp[c * npx + idx] = o[c];
}

// 2x2 solve of the blurred system (A.6).  Differences of products with one fused rounding each (Kahan): the CPU path
// does this step in double.
static __device__ __forceinline__ void solve_px(float g11, float g12, float g22, float h1, float h2, float* u, float* v)
{
    const float p = g12 * g12, pe = fmaf(g12, g12, -p);
    const float det = (fmaf(g11, g22, -p) - pe) + 1e-3f;
    const float idet = 1.f / det;
    const float qa = g12 * h1, qae = fmaf(g12, h1, -qa);
    const float qb = g12 * h2, qbe = fmaf(g12, h2, -qb);
    *u = (fmaf(g11, h2, -qa) - qae) * idet;
    *v = (fmaf(g22, h1, -qb) - qbe) * idet;
}

// resize(prevFlow -> layer size, INTER_LINEAR) * mul at one pixel (A.2): half-pixel centres, clamped, f32 weights.
static __device__ __forceinline__ float2 upsample_flow(const float* __restrict__ pf, int pw, int ph, float mul, double scale_x,
                                                       double scale_y, int x, int y)
{
#pragma clang fp contract(off)                       // products and sums rounded one by one, as resize()'s host code does
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
    return make_float2(((p00.x * a0 + p01.x * a1) * b0 + (p10.x * a0 + p11.x * a1) * b1) * mul,
                       ((p00.y * a0 + p01.y * a1) * b0 + (p10.y * a0 + p11.y * a1) * b1) * mul);
}

// Initial M of a layer.  flow = 0 (top layer), resize(prevFlow)*mul evaluated inline (lower layers), or an
// explicit flow field (stage hook).  The upsampled flow is never written: the first blur sweep overwrites it.
// One thread per pixel, 64 x 4 pixels per workgroup on a plain 2-D grid.  (Measured alternative, not kept: 64 x 16 tiles in the
// XCD-aware order with two gathers in flight per thread -- L2 absorbed the gather rows, the kernel ran 5 % slower: its reads
// that miss L2 are served by the Infinity Cache anyway, and the taller tile halves the number of workgroups in flight.)
template <int MODE>  // 0 zero, 1 upsample, 2 explicit
__global__ __launch_bounds__(256) void k_update_matrices(const float* __restrict__ R0, const float* __restrict__ R1,
                                                         size_t R_stride, const float* __restrict__ fsrc, size_t f_stride,
                                                         int pw, int ph, float mul, double scale_x, double scale_y, int w,
                                                         int h, float* __restrict__ M, size_t M_stride, int y_begin, int y_end)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = y_begin + blockIdx.y * 4 + (threadIdx.x >> 6);          // rows [y_begin, y_end) of the layer only
    if (x >= w || y >= y_end) return;
    const int s = blockIdx.z;
    float dx = 0.f, dy = 0.f;
    if (MODE == 1) {
        const float2 f = upsample_flow(fsrc + (size_t)s * f_stride, pw, ph, mul, scale_x, scale_y, x, y);
        dx = f.x; dy = f.y;
    } else if (MODE == 2) {
        const float2 f = *(const float2*)(fsrc + (size_t)s * f_stride + ((size_t)y * w + x) * 2);
        dx = f.x; dy = f.y;
    }
    update_px(R0 + (size_t)s * R_stride, R1 + (size_t)s * R_stride, (size_t)w * h, w, h, x, y, dx, dy,
              M + (size_t)s * M_stride);
}

void launch_update_matrices(hipStream_t st, const float* R0, const float* R1, size_t R_stride, const float* flow_prev,
                            size_t fp_stride, int pw, int ph, float mul, int G, int w, int h, float* M, size_t M_stride, int y_begin, int y_end)
{
    if (y_begin < 0) y_begin = 0;
    if (y_end < 0 || y_end > h) y_end = h;
    if (y_end <= y_begin) return;
    dim3 grid((w + 63) / 64, (y_end - y_begin + 3) / 4, G);
    if (flow_prev)
        hipLaunchKernelGGL(k_update_matrices<1>, grid, dim3(256), 0, st, R0, R1, R_stride, flow_prev, fp_stride, pw, ph, mul,
                           (double)pw / w, (double)ph / h, w, h, M, M_stride, y_begin, y_end);
    else
        hipLaunchKernelGGL(k_update_matrices<0>, grid, dim3(256), 0, st, R0, R1, R_stride, (const float*)nullptr, (size_t)0,
                           0, 0, 0.f, 0.0, 0.0, w, h, M, M_stride, y_begin, y_end);
}

void launch_update_matrices_flow(hipStream_t st, const float* R0, const float* R1, size_t R_stride, const float* flow,
                                 size_t f_stride, int G, int w, int h, float* M, size_t M_stride)
{
    dim3 grid((w + 63) / 64, (h + 3) / 4, G);
    hipLaunchKernelGGL(k_update_matrices<2>, grid, dim3(256), 0, st, R0, R1, R_stride, flow, f_stride, 0, 0, 0.f, 0.0, 0.0,
                       w, h, M, M_stride, 0, h);
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
__global__ __launch_bounds__(256) void k_blur_iter_generic(const float* __restrict__ M_in, float* __restrict__ M_out, size_t M_stride,
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
        float u, v;
        solve_px(g11, g12, g22, h1, h2, &u, &v);
        *(float2*)(fo + ((size_t)gy * w + gx) * 2) = make_float2(u, v);
        if (do_update) update_px(R0p, R1p, npx, w, h, gx, gy, u, v, Mo);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Fast form of the sweep for a compile-time half-window M_T (winsize 12 -> 6), image width a multiple of 4.
// Tile = 64 x 16 pixels per 256-thread workgroup; tiles are numbered so that the workgroups one XCD receives
// (blockIdx % 8, round-robin dispatch) walk one contiguous band of the image: halo rows are re-read from that XCD's L2.
// The kernel is built around memory latency (the v1-v3 profiles showed ~7 serialized round trips per tile):
//   entry    the R0 values phase C will need are requested first (they depend on nothing)
//   phase A  one thread per (plane, PAIR of tile columns): the 16+2m rows of its two columns arrive as float2 loads, all
//            independent and in flight together; vertical sliding sums in registers; only the 16 sums go to LDS
//   barrier  (the only one)
//   phase B  one thread per 4 consecutive pixels of a row.  Lanes are assigned by the hardware's ds_read_b128 lane
//            groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): each group reads 16 consecutive float4 of ONE row =
//            all 64 banks once, conflict-free at any pitch.  Horizontal sliding sums, 2x2 solve.  Wave v owns tile rows
//            4v..4v+3 in phases B and C, and nobody else reads those LDS rows, so it parks its flow in the rows of planes
//            0/1 it has just finished reading: no second barrier, no extra LDS.
//   phase C  UpdateMatrices with lane = pixel column (64 consecutive pixels per wave): the R1 gathers of two pixels
//            are in flight together; R0 loads, gathers and M' stores are all row-contiguous across the wave.
// flow is stored to HBM only by a layer's last sweep (earlier sweeps' flow is consumed inside the kernel).
// LDS: 5 x (16 x 76 + 12) floats = 24.6 KB -> 6 workgroups (24 waves) per CU.  Raw M never touches LDS.
// ------------------------------------------------------------------------------------------------------------
#define FT_X 64
#define FT_Y 16

struct GatherPx {
    float p00[5], p01[5], p10[5], p11[5];
    float fx, fy;
    bool inside;
};
// request the 2x2 neighbourhood of the 5 R1 planes around (x + dx, y + dy); out-of-image taps read a clamped address
static __device__ __forceinline__ void gather_issue(const float* __restrict__ R1p, size_t npx, int w, int h, int x, int y, float dx,
                                                    float dy, GatherPx& g)
{
    float fx = x + dx, fy = y + dy;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    g.fx = fx - x1; g.fy = fy - y1;
    g.inside = (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
    const int xc = g.inside ? x1 : 0, yc = g.inside ? y1 : 0;
    const float* p = R1p + (size_t)yc * w + xc;
#pragma unroll
    for (int c = 0; c < 5; c++) {
        g.p00[c] = p[0]; g.p01[c] = p[1]; g.p10[c] = p[w]; g.p11[c] = p[w + 1];
        p += npx;
    }
}
static __device__ __forceinline__ void update_finish(const float q[5], const GatherPx& g, int w, int h, int x, int y, float dx,
                                                     float dy, float out[5])
{
#pragma clang fp contract(off)                       // explicit fmaf only: see update_core()
    float r2, r3, r4, r5, r6;
    if (g.inside) {
        const float gx = 1.f - g.fx, gy = 1.f - g.fy;
        const float a00 = gx * gy, a01 = g.fx * gy, a10 = gx * g.fy, a11 = g.fx * g.fy;
        r2 = fmaf(a11, g.p11[0], fmaf(a10, g.p10[0], fmaf(a01, g.p01[0], a00 * g.p00[0])));
        r3 = fmaf(a11, g.p11[1], fmaf(a10, g.p10[1], fmaf(a01, g.p01[1], a00 * g.p00[1])));
        r4 = fmaf(a11, g.p11[2], fmaf(a10, g.p10[2], fmaf(a01, g.p01[2], a00 * g.p00[2])));
        r5 = fmaf(a11, g.p11[3], fmaf(a10, g.p10[3], fmaf(a01, g.p01[3], a00 * g.p00[3])));
        r6 = fmaf(a11, g.p11[4], fmaf(a10, g.p10[4], fmaf(a01, g.p01[4], a00 * g.p00[4])));
        r4 = (q[2] + r4) * 0.5f;
        r5 = (q[3] + r5) * 0.5f;
        r6 = (q[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = q[2]; r5 = q[3]; r6 = q[4] * 0.5f;
    }
    r2 = (q[0] - r2) * 0.5f;
    r3 = (q[1] - r3) * 0.5f;
    r2 += fmaf(r4, dy, r6 * dx);
    r3 += fmaf(r6, dy, r5 * dx);
    const int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        auto bw = [](int d) { return d < 2 ? 0.14f : 0.4472f; };
        const float scale = (x < BORDER ? bw(x) : 1.f) * (x >= w - BORDER ? bw(w - x - 1) : 1.f) *
                            (y < BORDER ? bw(y) : 1.f) * (y >= h - BORDER ? bw(h - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    out[0] = fmaf(r4, r4, r6 * r6);
    out[1] = (r4 + r5) * r6;
    out[2] = fmaf(r5, r5, r6 * r6);
    out[3] = fmaf(r4, r2, r6 * r3);
    out[4] = fmaf(r6, r2, r5 * r3);
}

struct __attribute__((packed, aligned(4))) F2U { float x, y; };     // two adjacent floats at 4-byte alignment
// AL = true: width a multiple of 4 and 16-byte aligned planes (vector loads / stores as written); AL = false: any width -- the
// column pairs are read as two floats at 4-byte alignment and the flow is stored pixel by pixel.
// WT = true: M' leaves the kernel through write-through (agent-scope, sc1) stores instead of staying dirty in the XCD's L2 until the
// end-of-kernel write-back.  With two launches in flight on two streams (the batch schedules) that is worth 2 % of a step -- 1080p,
// 64 pairs: 2 532 -> 2 577 pairs/s over three alternating runs, 4K / 5 layers, 16 pairs: 571 -> 583 (profiles/r03/ab_write_through.log) --
// the lines M' would occupy in L2 stay free for the M / R0 / R1 reads and no launch ends in a burst of write-backs that its
// neighbour on the other stream has to share; alone on one stream (one pair per call) the same stores cost 3 % (0.305 vs 0.297 ms
// per 1280x720 pair), so launch_blur_iter's callers ask for it in the two-stream schedules only.  Write-through for the initial M
// (k_update_matrices) and for the expansions (k_polyexp) was measured too: -0.7 % and +-0.
template <int M_T, bool AL = true, bool WT = false>
__global__ __launch_bounds__(256) void k_blur_iter_fast(const float* __restrict__ M_in, float* __restrict__ M_out,
                                                        size_t M_stride, const float* __restrict__ R0,
                                                        const float* __restrict__ R1, size_t R_stride, int w, int h,
                                                        TileMap tm, float scale,
                                                        int do_update, int store_flow, float* __restrict__ flow, size_t f_stride)
{
    constexpr int EXT_X = FT_X + 2 * M_T;              // 76
    constexpr int EXT_Y = FT_Y + 2 * M_T;              // 28
    constexpr int WIN = 2 * M_T + 1;                   // 13
    constexpr int PITCH = (EXT_X + 3) & ~3;            // 76
    constexpr int PLANE = FT_Y * PITCH + (EXT_X - (FT_Y * PITCH) % 32 + 64) % 32;   // = EXT_X (mod 32): phase-A lanes stay on distinct banks across planes
    static_assert(PLANE % 4 == 0 && EXT_X % 2 == 0, "plane must keep 16-byte alignment");
    __shared__ __attribute__((aligned(16))) float vs[5 * PLANE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    int s, tx, ty;                                       // XCD-aware tile order, see tile_of_block()
    if (!tile_of_block(tm, &s, &tx, &ty)) return;
    const int x0 = tx * FT_X, y0 = ty * FT_Y;
    const size_t npx = (size_t)w * h;
    const float* Min = M_in + (size_t)s * M_stride;
    const float* R0p = R0 + (size_t)s * R_stride;
    const float* R1p = R1 + (size_t)s * R_stride;

    STAMP(ts0);
    // entry: R0 of this thread's four phase-C pixels (rows 4*wv + j, column lane)
    float q[4][5];
    int gys[4];
    const int gxc = min(x0 + lane, w - 1);
    if (do_update) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            gys[j] = min(y0 + wv * 4 + j, h - 1);
            const size_t idx = (size_t)gys[j] * w + gxc;
#pragma unroll
            for (int c = 0; c < 5; c++) q[j][c] = R0p[c * npx + idx];
        }
    }

    if (x0 >= M_T && x0 + FT_X + M_T <= w) {           // all 76 columns inside the image: float2 columns, one pass
        constexpr int NP = EXT_X / 2;                  // 38 column pairs per plane, 190 threads busy
        for (int t = tid; t < 5 * NP; t += 256) {
            const int c = t / NP, pr = t - c * NP;
            const float* col = Min + c * npx + (x0 - M_T + 2 * pr);
            float2 v[EXT_Y];
#pragma unroll
            for (int i = 0; i < EXT_Y; i++) {
                const float* pp = col + (size_t)clampi(y0 - M_T + i, 0, h - 1) * w;
                if (AL) v[i] = *(const float2*)pp;
                else { const F2U t = *(const F2U*)pp; v[i] = make_float2(t.x, t.y); }
            }
            float* out = vs + c * PLANE + 2 * pr;
            float sx = 0.f, sy = 0.f;
#pragma unroll
            for (int i = 0; i < WIN; i++) { sx += v[i].x; sy += v[i].y; }
            *(float2*)out = make_float2(sx, sy);
#pragma unroll
            for (int y = 1; y < FT_Y; y++) {
                sx += v[y + WIN - 1].x - v[y - 1].x;
                sy += v[y + WIN - 1].y - v[y - 1].y;
                *(float2*)(out + y * PITCH) = make_float2(sx, sy);
            }
        }
    } else {                                           // tiles touching the left/right image edge: clamped scalar columns
        for (int t = tid; t < 5 * EXT_X; t += 256) {
            const int c = t / EXT_X, lx = t - c * EXT_X;
            const int gx = clampi(x0 - M_T + lx, 0, w - 1);
            const float* col = Min + c * npx + gx;
            float v[EXT_Y];
#pragma unroll
            for (int i = 0; i < EXT_Y; i++) v[i] = col[(size_t)clampi(y0 - M_T + i, 0, h - 1) * w];
            float* out = vs + c * PLANE + lx;
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < WIN; i++) sum += v[i];
            out[0] = sum;
#pragma unroll
            for (int y = 1; y < FT_Y; y++) {
                sum += v[y + WIN - 1] - v[y - 1];
                out[y * PITCH] = sum;
            }
        }
    }
    STAMP(ts1);
    __syncthreads();
    STAMP(ts2);

    {
        // hardware b128 lane group (quads of 4 lanes: g0 = quads {0,3,5,6}, g1 = quads {1,2,4,7}, +2 for lanes 32..63)
        const int quad = (lane & 31) >> 2;
        const int grp = (lane >> 5) * 2 + ((quad == 1 || quad == 2 || quad == 4 || quad == 7) ? 1 : 0);
        const int qpos = (quad == 0 || quad == 1) ? 0 : ((quad == 3 || quad == 2) ? 1 : ((quad == 5 || quad == 4) ? 2 : 3));
        const int pos = qpos * 4 + (lane & 3);          // 0..15 inside the group
        const int ly = wv * 4 + grp;                     // tile row 0..15
        const int lx0 = pos * 4;
        const int gx = x0 + lx0, gy = y0 + ly;
        float S[5][4];
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const float4* p = (const float4*)(vs + c * PLANE + ly * PITCH + lx0);
            constexpr int NV = (4 + 2 * M_T + 3) / 4;
            float f[4 * NV];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const float4 t4 = p[k];
                f[4 * k] = t4.x; f[4 * k + 1] = t4.y; f[4 * k + 2] = t4.z; f[4 * k + 3] = t4.w;
            }
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < WIN; k++) a += f[k];
            S[c][0] = a;
#pragma unroll
            for (int j = 1; j < 4; j++) {
                a += f[j + WIN - 1] - f[j - 1];
                S[c][j] = a;
            }
        }
        float u[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            solve_px(S[0][j] * scale, S[1][j] * scale, S[2][j] * scale, S[3][j] * scale, S[4][j] * scale, &u[j], &v[j]);
        if (store_flow && gx < w && gy < h) {
            float* fo = flow + (size_t)s * f_stride + ((size_t)gy * w + gx) * 2;
            if (AL) {                                        // w % 4 == 0: the 4 pixels are all inside or all outside
                *(float4*)fo = make_float4(u[0], v[0], u[1], v[1]);
                *(float4*)(fo + 4) = make_float4(u[2], v[2], u[3], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (gx + j < w) *(float2*)(fo + 2 * j) = make_float2(u[j], v[j]);
            }
        }
        if (!do_update) return;
        // park the flow in this wave's own rows of planes 0 (u) and 1 (v); every read of those rows by this wave is
        // older in program order, and no other wave touches them
        *(float4*)(vs + ly * PITCH + lx0) = make_float4(u[0], u[1], u[2], u[3]);
        *(float4*)(vs + PLANE + ly * PITCH + lx0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    STAMP(ts3);

    float* Mo = M_out + (size_t)s * M_stride;
    const bool colok = x0 + lane < w;
#pragma unroll
    for (int jb = 0; jb < 4; jb += 2) {
        GatherPx g[2];
        float fu[2], fv[2];
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            const int ly = wv * 4 + jb + jj;
            fu[jj] = vs[ly * PITCH + lane];
            fv[jj] = vs[PLANE + ly * PITCH + lane];
            gather_issue(R1p, npx, w, h, gxc, gys[jb + jj], fu[jj], fv[jj], g[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 2; jj++) {
            float o[5];
            update_finish(q[jb + jj], g[jj], w, h, gxc, gys[jb + jj], fu[jj], fv[jj], o);
            if (colok && y0 + wv * 4 + jb + jj < h) {
                const size_t idx = (size_t)gys[jb + jj] * w + gxc;
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    if constexpr (WT) __hip_atomic_store((unsigned*)(Mo + c * npx + idx), __float_as_uint(o[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else Mo[c * npx + idx] = o[c];
                }
            }
        }
#ifdef MAV_STAMPS
        if (jb == 0) { STAMP(tsm); STAMP_ADD(3, ts3, tsm); STAMP_ADD(5, tsm, tsm); }
#endif
    }
    STAMP(ts4);
    STAMP_ADD(0, ts0, ts1); STAMP_ADD(1, ts1, ts2); STAMP_ADD(2, ts2, ts3); STAMP_ADD(4, ts3, ts4); STAMP_ADD(6, ts0, ts4); STAMP_ADD(7, 0ull, 1ull);
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
int blur_iter_tile_rows(int h) { return (h + FT_Y - 1) / FT_Y; }
static bool blur_iter_vec_ok(int w, size_t M_stride, size_t R_stride, size_t f_stride, const void* M_in, const void* M_out, const void* R0,
                             const void* R1, const void* flow)
{
    return (w % 4 == 0) && (M_stride % 4 == 0) && (R_stride % 4 == 0) && (f_stride % 4 == 0) && aligned16(M_in) && aligned16(M_out) &&
           aligned16(R0) && aligned16(R1) && aligned16(flow);
}
bool blur_iter_bands_ok(int w, int winsize, size_t M_stride, size_t R_stride, size_t f_stride, const void* M_in, const void* M_out,
                        const void* R0, const void* R1, const void* flow)
{
    return winsize / 2 == 6 && blur_iter_vec_ok(w, M_stride, R_stride, f_stride, M_in, M_out, R0, R1, flow);
}

// tile rows [ty0, ty1) only (ty1 < 0: the whole layer).  Band launches exist for the fast form only (blur_iter_bands_ok).
void launch_blur_iter(hipStream_t st, const float* M_in, float* M_out, size_t M_stride, const float* R0, const float* R1,
                      size_t R_stride, int G, int w, int h, int winsize, int do_update, int store_flow, float* flow, size_t f_stride,
                      int ty0, int ty1, int strip, bool write_through)
{
    int ext, pitch, plane;
    const int m = winsize / 2;
    const float scale = (float)(1.0 / ((double)winsize * winsize));
    dim3 grid((w + MAV_TILE - 1) / MAV_TILE, (h + MAV_TILE - 1) / MAV_TILE, G);
    const bool vec_ok = blur_iter_vec_ok(w, M_stride, R_stride, f_stride, M_in, M_out, R0, R1, flow);
    if (m == 6 && vec_ok) {
        const TileMap tm = make_tile_map(w, h, G, FT_X, FT_Y, ty0, ty1, strip);
        if (tm.n_tiles == 0) return;
        if (write_through && do_update)
            hipLaunchKernelGGL((k_blur_iter_fast<6, true, true>), dim3(tile_grid(tm)), dim3(256), 0, st, M_in, M_out, M_stride, R0, R1, R_stride, w,
                               h, tm, scale, do_update, store_flow, flow, f_stride);
        else
            hipLaunchKernelGGL(k_blur_iter_fast<6>, dim3(tile_grid(tm)), dim3(256), 0, st, M_in, M_out, M_stride, R0, R1, R_stride, w, h,
                               tm, scale, do_update, store_flow, flow, f_stride);
        return;
    }
    if (m == 6 && f_stride % 2 == 0 && ((uintptr_t)flow & 7) == 0) {    // any width / alignment: relaxed form of the same kernel
        const TileMap tm = make_tile_map(w, h, G, FT_X, FT_Y, 0, -1, strip);
        if (write_through && do_update)
            hipLaunchKernelGGL((k_blur_iter_fast<6, false, true>), dim3(tile_grid(tm)), dim3(256), 0, st, M_in, M_out, M_stride, R0, R1, R_stride,
                               w, h, tm, scale, do_update, store_flow, flow, f_stride);
        else
            hipLaunchKernelGGL((k_blur_iter_fast<6, false>), dim3(tile_grid(tm)), dim3(256), 0, st, M_in, M_out, M_stride, R0, R1, R_stride,
                               w, h, tm, scale, do_update, store_flow, flow, f_stride);
        return;
    }
    iter_geometry(m, &ext, &pitch, &plane);
    const size_t lds = sizeof(float) * 5 * (size_t)plane;
    if (m == 6)
        hipLaunchKernelGGL(k_blur_iter_generic<6>, grid, dim3(256), lds, st, M_in, M_out, M_stride, R0, R1, R_stride, w, h, m,
                           pitch, plane, scale, do_update, flow, f_stride);
    else       // dynamic LDS above the default limit was granted by blur_iter_prepare() when the context was created
        hipLaunchKernelGGL(k_blur_iter_generic<0>, grid, dim3(256), lds, st, M_in, M_out, M_stride, R0, R1, R_stride, w, h, m,
                           pitch, plane, scale, do_update, flow, f_stride);
}

// The general-winsize sweep needs 5 x (32 + 2m)^2 floats of dynamic LDS; above the 64 KB default a kernel must be granted
// the size on every device it runs on.  Called by mav_create on the context's device; returns an error text or nullptr.
const char* blur_iter_prepare(int winsize)
{
    const size_t lds = blur_iter_lds_bytes(winsize);
    if (lds <= (size_t)64 * 1024) return nullptr;
    const hipError_t e = hipFuncSetAttribute((const void*)k_blur_iter_generic<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return e == hipSuccess ? nullptr : hipGetErrorString(e);
}

// ------------------------------------------------------------------------------------------------------------
// Calibration (mav_membw_probe, bench.py's `measured_ceiling`): what the memory system delivers to a plain streaming kernel with
// the sweeps' mix of 3 reads : 1 write (M, R0, R1 in; M' out), one float4 per thread and step, grid-stride.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_probe_r3w1(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                                    float4* __restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i], z = c[i];
        d[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
    }
}
void launch_probe_r3w1(hipStream_t st, const float* a, const float* b, const float* c, float* d, size_t n_float4)
{
    hipLaunchKernelGGL(k_probe_r3w1, dim3(256 * 8), dim3(256), 0, st, (const float4*)a, (const float4*)b, (const float4*)c, (float4*)d, n_float4);
}
