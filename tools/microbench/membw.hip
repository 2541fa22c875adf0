// Memory-system calibration for MI355X: streaming copy / read at several footprints (HBM vs Infinity-Cache resident).
// build: hipcc --offload-arch=gfx950 -O3 membw.hip -o membw      run on the GPU box: ./membw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n)
{
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[0] = s;
}
// 3 reads : 1 write, the mix of the sweep kernel (R0, R1, M in; M' out)
__global__ __launch_bounds__(256) void k_r3w1(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                                              float4* __restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 x = a[i], y = b[i], z = c[i];
        d[i] = make_float4(x.x + y.x + z.x, x.y + y.y + z.y, x.z + y.z + z.z, x.w + y.w + z.w);
    }
}
// 1 read : 5 writes (the polynomial expansion: image in, five coefficient planes out) and write only
__global__ __launch_bounds__(256) void k_r1w5(const float4* __restrict__ a, float4* __restrict__ o, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = a[i];
        o[i] = x; o[n + i] = x; o[2 * n + i] = x; o[3 * n + i] = x; o[4 * n + i] = x;
    }
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ o, size_t n)
{
    const float4 x = make_float4(1.f, 2.f, 3.f, 4.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = x;
}
static void write_mixes()
{
    const size_t sizes_mb[] = {8, 64, 512};                       // size of the input plane; the output is 5x that
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float4 *a, *o;
        hipMalloc(&a, bytes); hipMalloc(&o, 5 * bytes);
        hipMemset(a, 1, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = mb >= 512 ? 10 : 50, grid = 256 * 8;
        float ms;
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_r1w5, dim3(grid), dim3(256), 0, 0, a, o, n);
        hipEventRecord(e0); for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_r1w5, dim3(grid), dim3(256), 0, 0, a, o, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double mix = 6.0 * bytes * reps / (ms * 1e-3) / 1e12;
        hipEventRecord(e0); for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, o, 5 * n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double wr = 5.0 * bytes * reps / (ms * 1e-3) / 1e12;
        printf("plane %4zu MB: 1r+5w %.2f TB/s (footprint %zu MB)   write only %.2f TB/s (footprint %zu MB)\n", mb, mix, 6 * mb, wr, 5 * mb);
        hipFree(a); hipFree(o);
    }
}
int main()
{
    write_mixes();
    const size_t sizes_mb[] = {16, 32, 64, 128, 512, 2048};
    float* out; hipMalloc(&out, 4);
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20, n = bytes / 16;
        float4 *a, *b, *c, *d;
        hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&d, bytes);
        hipMemset(a, 1, bytes); hipMemset(b, 1, bytes); hipMemset(c, 1, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = mb >= 512 ? 10 : 50, grid = 256 * 8;
        float ms;
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n);
        hipEventRecord(e0); for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double copy = 2.0 * bytes * reps / (ms * 1e-3) / 1e12;
        hipEventRecord(e0); for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, out, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double rd = 1.0 * bytes * reps / (ms * 1e-3) / 1e12;
        hipEventRecord(e0); for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_r3w1, dim3(grid), dim3(256), 0, 0, a, b, c, d, n); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); const double mix = 4.0 * bytes * reps / (ms * 1e-3) / 1e12;
        printf("buffer %5zu MB: copy %.2f TB/s (footprint %zu MB)   read %.2f TB/s   3r+1w %.2f TB/s (footprint %zu MB)\n", mb, copy, 2 * mb, rd, mix, 4 * mb);
        hipFree(a); hipFree(b); hipFree(c); hipFree(d);
    }
    return 0;
}
