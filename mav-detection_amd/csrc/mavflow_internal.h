// Internal declarations shared by the host side (mavflow.cpp) and the kernel translation units.
// gfx950 only: wave = 64 lanes, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mavflow.h"

#define MAV_MAX_POLY_N 16
#define MAV_TILE 32  // flow-iteration tile edge (pixels)

// FarnebackPrepareGaussian output, centre tap first (index k = |offset|).
struct PolyCoef {
    int n;
    float g[MAV_MAX_POLY_N + 1], xg[MAV_MAX_POLY_N + 1], xxg[MAV_MAX_POLY_N + 1];
    float ig11, ig03, ig33, ig55;
};

// GaussianBlur(ksize, sigma) + resize(INTER_LINEAR) of one layer: the f32 Gaussian taps (device memory, getGaussianKernel) and
// the two resize ratios; everything else is arithmetic inside the kernels.
struct BlurParams {
    int ksize;
    int fixed3;          // ksize == 3 with the fixed kernel [1/4, 1/2, 1/4] (sigma == 0)
    const float* g;      // [ksize]
    double scale_x;      // W / w
    double scale_y;      // H / h
    // resize(INTER_LINEAR) coordinates of the layer's w columns and h rows as tables (device memory; built once per context by the host
    // in resize_coord's own arithmetic): source index and weight of the right / lower neighbour.  nullptr: evaluated in the kernel.
    const int* xs; const float* xf;      // [w]
    const int* ys; const float* yf;      // [h]
};

// How the detection kernels treat one pair's float32 flow (what numpy does with it in the reference):
//   PROMOTE   frame index >= 1 with zero rates: flow - 0.0 is the float32 field promoted to double, double arithmetic after it
//   DEROTATE  frame index >= 1: Detector.derotate (detector.py:83-117) in double, evaluated on the fly
//   FRAME0    frame index 0: derotate returns the float32 array itself (detector.py:80-81), float32 arithmetic after it
enum { MAV_PAIR_PROMOTE = 0, MAV_PAIR_DEROTATE = 1, MAV_PAIR_FRAME0 = 2 };
struct DerotParams {  // one pair; Detector.derotate
    double o0, o1, o2, sx, sy;  // sx = w*dt/2, sy = h*dt/2
    int mode;                   // MAV_PAIR_*
};

// Several layers in one launch (a small group's whole pyramid): job tables passed by value in the kernel arguments.
#define MAV_MAX_JOBS 6
struct PolyJob { const float* I; float* R; size_t I_stride, R_stride; int w, h, tiles_x, per_img, first_block, pad; };
struct PolyJobs { int n, pad; PolyJob j[MAV_MAX_JOBS]; };
struct BlurJob { float* out; size_t out_stride; BlurParams bp; int w, h, fused, gx, gy, first_block, rows_cap, pitch_w, th, pad; };
struct BlurJobs { int n, pad; BlurJob j[MAV_MAX_JOBS]; };

// ---- flow kernels (kernels_flow.hip) ----------------------------------------------------------------------
// All take G slots; slot s reads/writes base + s*stride (strides in elements).
// G images from two runs: the first `split` from img, the rest from img2 (same stride); img2 == nullptr: one run.  Layers with a
// short Gaussian (blur_resize_is_fused) go through one fused kernel and never touch tmp unless two_pass is set.
void launch_blur_resize(hipStream_t st, const uint8_t* img, const uint8_t* img2, int split, size_t img_stride, int G, int W, int H, int w,
                        int h, BlurParams bp, float* tmp /* G x H x w scratch for the separable passes */, size_t tmp_stride, float* out,
                        size_t out_stride, bool two_pass = false);
bool blur_resize_is_fused(int W, int H, int w, int h, int ksize);
bool blur_resize_needs_tmp(const uint8_t* img, const uint8_t* img2, size_t img_stride, int W, int H, int w, int h, BlurParams bp,
                           const float* out, size_t out_stride);
void launch_polyexp(hipStream_t st, const float* I, size_t I_stride, int G, int w, int h, const PolyCoef& pc, float* R,
                    size_t R_stride);
// jobs.j[i]: I, R, I_stride, R_stride, w, h filled by the caller; G images per job
void launch_polyexp_multi(hipStream_t st, PolyJobs jobs, int G, const PolyCoef& pc);
// jobs.j[i]: out, out_stride, bp, w, h filled by the caller, for layers blur_multi_ok accepts
bool blur_multi_ok(const uint8_t* img, const uint8_t* img2, size_t img_stride, int W, int H, int w, int h, BlurParams bp, const float* out,
                   size_t out_stride);
void launch_blur_multi(hipStream_t st, const uint8_t* img, const uint8_t* img2, int split, size_t img_stride, int G, int W, int H, BlurJobs jobs,
                       bool split_by_path = false /* one launch per tile code (fused_path_of) instead of one for all jobs */);
// flow_prev == nullptr: zero initial flow. Otherwise flow = resize(prev (ph x pw x 2))*mul, evaluated inline.
void launch_update_matrices(hipStream_t st, const float* R0, const float* R1, size_t R_stride, const float* flow_prev,
                            size_t fp_stride, int pw, int ph, float mul, int G, int w, int h, float* M, size_t M_stride,
                            int y_begin = 0, int y_end = -1 /* pixel rows [y_begin, y_end) only; -1 = to the bottom */);
// explicit per-pixel flow (h x w x 2), used by the stage hook
void launch_update_matrices_flow(hipStream_t st, const float* R0, const float* R1, size_t R_stride, const float* flow,
                                 size_t f_stride, int G, int w, int h, float* M, size_t M_stride);
void launch_blur_iter(hipStream_t st, const float* M_in, float* M_out, size_t M_stride, const float* R0, const float* R1,
                      size_t R_stride, int G, int w, int h, int winsize, int do_update, int store_flow, float* flow, size_t f_stride,
                      int ty0 = 0, int ty1 = -1 /* tile rows [ty0, ty1) of 16 pixel rows; ty1 < 0 = the whole layer */,
                      int strip = 0 /* width in tiles of the tile order's column strips; 0 = automatic */,
                      bool write_through = false /* M' through sc1 stores: see k_blur_iter_fast */);
int blur_iter_tile_rows(int h);                 // 16-pixel tile rows of a layer of height h
// band launches (ty0 / ty1) are honoured only by the fast sweep kernel: true when launch_blur_iter will take it for these operands
bool blur_iter_bands_ok(int w, int winsize, size_t M_stride, size_t R_stride, size_t f_stride, const void* M_in, const void* M_out,
                        const void* R0, const void* R1, const void* flow);
// store_flow == 0: the sweep's flow is consumed inside the kernel only (valid when do_update != 0)
size_t blur_iter_lds_bytes(int winsize);
const char* blur_iter_prepare(int winsize);   // grants the general sweep kernel its dynamic LDS on the current device

void launch_probe_r3w1(hipStream_t st, const float* a, const float* b, const float* c, float* d, size_t n_float4);   // calibration

// ---- detection kernels (kernels_detect.hip, compiled with -ffp-contract=off) --------------------------------
struct FoeScratch {
    double* cand;                  // [B][N][2] compacted candidates
    int* count;                    // [B]
    unsigned long long* best_key;  // [B]
    unsigned* done;                // [B][2] tickets: [0] workgroups of the vote that have finished, [1] of the phi kernel (both self-resetting)
};
// FlowT = float (optionally derotated on the fly) or double (already derotated).
void launch_foe_f32(hipStream_t st, const float* flow, const DerotParams* derot /*dev, [B] or null*/, const uint32_t* samples,
                    int B, int W, int H, int N, double mag2_thr, float mag2_thr_f32 /* frame-0 pairs */, double dist2_thr,
                    FoeScratch s, double* foe, int32_t* box_acc /* nullable: the pair's accumulators are initialised here too */,
                    unsigned long long* max_phi_bits /* nullable */);
void launch_foe_f64(hipStream_t st, const double* flow, const uint32_t* samples, int B, int W, int H, int N, double mag2_thr,
                    double dist2_thr, FoeScratch s, double* foe, int32_t* box_acc, unsigned long long* max_phi_bits);
// box_acc: [B][4] int32 accumulators (x0 min, y0 min, x1 max, y1 max), initialised by launch_box_init or by launch_foe_*.
void launch_box_init(hipStream_t st, int32_t* box_acc, unsigned long long* max_phi_bits, unsigned* done /* FoeScratch::done or null */, int B);
// How a phi launch ends and is tuned: results / box_out (either may be null) are written by the pair's last workgroup (needs done).
struct PhiLaunch {
    unsigned* done = nullptr;
    mav_result* results = nullptr;
    int32_t* box_out = nullptr;
    bool screen = true;      // option "phi_screen"
    int yloop = 0;           // option "phi_yloop" (0 = automatic)
};
void launch_phi_mask_f32(hipStream_t st, const float* flow, const DerotParams* derot, const double* foe, const uint8_t* sky,
                         int B, int W, int H, mav_thr_params thr, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn,
                         int32_t* box_acc, unsigned long long* max_phi_bits, const PhiLaunch& pl);
void launch_phi_mask_f64(hipStream_t st, const double* flow, const double* foe, const uint8_t* sky, int B, int W, int H,
                         mav_thr_params thr, double* phi, uint8_t* mask_fixed, uint8_t* mask_dyn, int32_t* box_acc,
                         unsigned long long* max_phi_bits, const PhiLaunch& pl);
void launch_box_finalize(hipStream_t st, const int32_t* box_acc, int B, int32_t* box);
void launch_derotate(hipStream_t st, const float* flow, const DerotParams* derot, int B, int W, int H, double* out);
void launch_bbox_u8(hipStream_t st, const uint8_t* img, int B, int W, int H, int* maxv /*[B] scratch*/, int32_t* box_acc);
void launch_window_max(hipStream_t st, const uint8_t* img, int B, int W, int H, unsigned long long* key /*[B]*/,
                       int64_t* out);
void launch_tpr_fpr(hipStream_t st, const uint8_t* gt, const uint8_t* mask, unsigned mask_value, int B, int W, int H,
                    unsigned long long* counts /*[B][4]*/);
// one pass over gt for one mask (mask1 = counts1 = nullptr) or two; gt_stride = 0: one ground-truth image for every pair
void launch_tpr_fpr2(hipStream_t st, const uint8_t* gt, size_t gt_stride, const uint8_t* mask0, const uint8_t* mask1, unsigned mask_value, int B,
                     int W, int H, unsigned long long* counts0 /*[B][4]*/, unsigned long long* counts1 /*[B][4]*/);
void launch_bgr2gray(hipStream_t st, const uint8_t* bgr, size_t n, uint8_t* gray);
void launch_ransac_only(hipStream_t st, FoeScratch s, int M, int N, double dist2_thr, double* foe);
void launch_make_derot(hipStream_t st, const double* omega /*null: no rotation*/, const double* dt, const uint8_t* frame0, int B,
                       int W, int H, DerotParams* out);

// ---- window search (kernels_window.hip, compiled with -ffp-contract=off) -----------------------------------------
#define MAV_PYR_MAX 32
// Levels of analyze_pyramid for one frame size: dims, first window index of each level in the reference's scan order
// (base[n] = total), and the byte offset of a level's image block (levels >= 1, batch images back to back) in the workspace.
struct PyrPlan {
    int n;
    int w[MAV_PYR_MAX], h[MAV_PYR_MAX];
    unsigned base[MAV_PYR_MAX + 1];
    size_t off[MAV_PYR_MAX];
};
void launch_area_resize(hipStream_t st, const uint8_t* src, size_t src_stride, int sw, int sh, uint8_t* dst, size_t dst_stride,
                        int dw, int dh, int B);
void launch_level_scan(hipStream_t st, const uint8_t* img, size_t stride, int B, int W, int H, unsigned idx_base,
                       unsigned long long* key /*[B], zeroed by the caller*/);
void launch_pyramid_finalize(hipStream_t st, const unsigned long long* key, const PyrPlan& plan, const uint8_t* img0,
                             const uint8_t* ws, int B, int64_t* out /*[B][6]*/);
void launch_optimize_window(hipStream_t st, const uint8_t* img, int B, int W, int H, unsigned long long* sat /*[B][(H+1)(W+1)]*/,
                            const int32_t* win_in, int64_t* score, int32_t* win_out);
