// Does the sweep kernel's ACCESS PATTERN (not its arithmetic) reach the streaming bandwidth?  Each 256-thread workgroup
// touches what one 64x16 sweep tile touches: 76x28 float2-column reads of 5 M planes, 64x16 dword reads of 5 R0 planes and
// 5 R1 planes, 64x16 dword writes of 5 M' planes -- no LDS, no barrier, no arithmetic beyond keeping the loads alive.
// build: hipcc --offload-arch=gfx950 -O3 tilebw.hip -o tilebw.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k_tile(const float* __restrict__ M, const float* __restrict__ R0, const float* __restrict__ R1,
                                              float* __restrict__ Mo, int w, int h, int tiles_x, int n_tiles)
{
    const int nb = gridDim.x, per = (nb + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * 64, y0 = ty * 16;
    const size_t npx = (size_t)w * h;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float acc = 0.f;
    if (tid < 190) {
        const int c = tid / 38, pr = tid - c * 38;
        int colx = x0 - 6 + 2 * pr; colx = colx < 0 ? 0 : (colx > w - 2 ? w - 2 : colx);
        const float* col = M + c * npx + colx;
        float2 v[28];
#pragma unroll
        for (int i = 0; i < 28; i++) { int yy = y0 - 6 + i; yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy); v[i] = *(const float2*)(col + (size_t)yy * w); }
#pragma unroll
        for (int i = 0; i < 28; i++) acc += v[i].x + v[i].y;
    }
    const int gx = x0 + lane;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int gy = min(y0 + wv * 4 + j, h - 1);
        const size_t idx = (size_t)gy * w + gx;
        float o[5];
#pragma unroll
        for (int c = 0; c < 5; c++) o[c] = R0[c * npx + idx] + R1[c * npx + idx];
#pragma unroll
        for (int c = 0; c < 5; c++) Mo[c * npx + idx] = o[c] + acc;
    }
}

int main(int argc, char** argv)
{
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 0;   // dummy dynamic LDS per workgroup: caps residency at 160/lds_kb per CU

    const int w = 1920, h = 1080, tiles_x = 30, tiles_y = 68, n_tiles = tiles_x * tiles_y;
    const size_t npx = (size_t)w * h, plane5 = 5 * npx * sizeof(float);
    for (int nsets : {1, 8}) {
        float *M[8], *R0[8], *R1[8], *Mo[8];
        for (int s = 0; s < nsets; s++) {
            hipMalloc(&M[s], plane5); hipMalloc(&R0[s], plane5); hipMalloc(&R1[s], plane5); hipMalloc(&Mo[s], plane5);
            hipMemset(M[s], 0, plane5); hipMemset(R0[s], 0, plane5); hipMemset(R1[s], 0, plane5);
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 40, nb = ((n_tiles + 7) / 8) * 8;
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_tile, dim3(nb), dim3(256), lds_kb * 1024, 0, M[r % nsets], R0[r % nsets], R1[r % nsets], Mo[r % nsets], w, h, tiles_x, n_tiles);
        hipEventRecord(e0);
        for (int r = 0; r < reps; r++) { const int s = r % nsets; hipLaunchKernelGGL(k_tile, dim3(nb), dim3(256), lds_kb * 1024, 0, M[s], R0[s], R1[s], Mo[s], w, h, tiles_x, n_tiles); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 4.0 * plane5;    // unique bytes per launch: M, R0, R1 read, M' written
        printf("lds %d KB: %d buffer set(s) (%s): %.1f us per launch, %.2f TB/s of unique bytes (80 B/px)\n", lds_kb, nsets,
               nsets == 1 ? "166 MB, Infinity-Cache resident" : "1.3 GB cycled, HBM", 1e3 * ms / reps, bytes * reps / (ms * 1e-3) / 1e12);
        for (int s = 0; s < nsets; s++) { hipFree(M[s]); hipFree(R0[s]); hipFree(R1[s]); hipFree(Mo[s]); }
    }
    return 0;
}
