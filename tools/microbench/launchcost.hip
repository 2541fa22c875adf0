// What does one dependent launch cost in a single-stream chain, and does it depend on the kernel-argument size or on the grid?
// build: hipcc --offload-arch=gfx950 -O3 launchcost.hip -o launchcost.bin      run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Big { int v[60]; };          // 240 bytes by value
__global__ void k_none(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 12345) p[1] = 1; }
__global__ void k_big(int* p, Big a, Big b) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == a.v[3] + b.v[59]) p[1] = 1; }
__global__ void k_store(float* q, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) q[i] = 1.f; }
template <typename F> static float chain(hipStream_t st, int n, F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < n; i++) f();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}
int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int* p; hipMalloc(&p, 64); hipMemset(p, 0, 64);
    float* q; hipMalloc(&q, 64 << 20);
    Big A{}, B{};
    const int N = 2000;
    for (int grid : {1, 144, 900, 4096}) {
        printf("grid %5d x 256 threads: no args %.2f us", grid, chain(st, N, [&] { hipLaunchKernelGGL(k_none, dim3(grid), dim3(256), 0, st, p); }));
        printf("   480 B of args %.2f us\n", chain(st, N, [&] { hipLaunchKernelGGL(k_big, dim3(grid), dim3(256), 0, st, p, A, B); }));
    }
    for (size_t mb : {1, 4, 18, 64}) {
        const size_t n = (mb << 20) / 4;
        printf("store %2zu MB per launch (plain stores): %.2f us per launch\n", mb, chain(st, 500, [&] { hipLaunchKernelGGL(k_store, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q, n); }));
    }
    return 0;
}
