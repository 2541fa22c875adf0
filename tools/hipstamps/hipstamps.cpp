// hipstamps: host-side time stamps around every HIP runtime call of a process (diagnostic; never part of the product).
//
// An LD_PRELOAD interposer: each HIP entry point libmavflow.so uses on its call paths is defined here, forwards to the real one
// (dlsym RTLD_NEXT) and records (call, start, end) of CLOCK_MONOTONIC into a ring.  Kernel launches written `k<<<...>>>` /
// hipLaunchKernelGGL compile to hipLaunchKernel, so every launch is seen.  tools/stall_stamps.py drives it:
//     LD_PRELOAD=tools/hipstamps/libhipstamps.so python tools/stall_stamps.py
// build: make -C tools/hipstamps
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <atomic>

namespace {
struct Rec { uint32_t id; uint32_t tid; uint64_t t0, t1; };
constexpr size_t CAP = 1u << 22;                 // 4 M records = 96 MB of address space, touched only as far as used
Rec* g_rec = nullptr;
std::atomic<size_t> g_n{0};
std::atomic<int> g_on{0};
const char* g_names[64];
std::atomic<int> g_nnames{0};

inline uint64_t now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec; }
int name_id(const char* s)
{
    int n = g_nnames.load();
    for (int i = 0; i < n; i++) if (g_names[i] == s) return i;
    int i = g_nnames.fetch_add(1);
    g_names[i] = s;
    return i;
}
struct Scope {
    uint32_t id; uint64_t t0; bool on;
    explicit Scope(int id_) : id(id_), t0(0), on(g_on.load(std::memory_order_relaxed) != 0) { if (on) t0 = now(); }
    ~Scope()
    {
        if (!on) return;
        uint64_t t1 = now();
        size_t i = g_n.fetch_add(1, std::memory_order_relaxed);
        if (i < CAP) g_rec[i] = Rec{id, 0, t0, t1};
    }
};
template <typename F> F real(const char* name)
{
    void* p = dlsym(RTLD_NEXT, name);
    if (!p) {       // the HIP runtime came in as a dependency of a dlopen()ed library (ctypes: RTLD_LOCAL): not in the global scope
        static void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!hip) hip = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_GLOBAL);
        if (hip) p = dlsym(hip, name);
    }
    if (!p) { fprintf(stderr, "hipstamps: %s not found\n", name); _exit(97); }
    return (F)p;
}
}  // namespace

#define WRAP(ret, name, params, args)                                   \
    extern "C" ret name params                                          \
    {                                                                   \
        static auto fn = real<ret(*) params>(#name);                   \
        static int id = name_id(#name);                                 \
        Scope s(id);                                                    \
        return fn args;                                                 \
    }

WRAP(hipError_t, hipLaunchKernel, (const void* f, dim3 g, dim3 b, void** a, size_t sh, hipStream_t st), (f, g, b, a, sh, st))
WRAP(hipError_t, hipEventRecord, (hipEvent_t e, hipStream_t st), (e, st))
WRAP(hipError_t, hipStreamWaitEvent, (hipStream_t st, hipEvent_t e, unsigned int fl), (st, e, fl))
WRAP(hipError_t, hipMemcpyAsync, (void* d, const void* s_, size_t n, hipMemcpyKind k, hipStream_t st), (d, s_, n, k, st))
WRAP(hipError_t, hipMemsetAsync, (void* d, int v, size_t n, hipStream_t st), (d, v, n, st))
WRAP(hipError_t, hipMemcpy, (void* d, const void* s_, size_t n, hipMemcpyKind k), (d, s_, n, k))
WRAP(hipError_t, hipMalloc, (void** p, size_t n), (p, n))
WRAP(hipError_t, hipFree, (void* p), (p))
WRAP(hipError_t, hipHostMalloc, (void** p, size_t n, unsigned int fl), (p, n, fl))
WRAP(hipError_t, hipHostFree, (void* p), (p))
WRAP(hipError_t, hipStreamSynchronize, (hipStream_t st), (st))
WRAP(hipError_t, hipEventSynchronize, (hipEvent_t e), (e))
WRAP(hipError_t, hipEventQuery, (hipEvent_t e), (e))
WRAP(hipError_t, hipDeviceSynchronize, (void), ())
WRAP(hipError_t, hipSetDevice, (int d), (d))
WRAP(hipError_t, hipEventCreate, (hipEvent_t* e), (e))
WRAP(hipError_t, hipEventCreateWithFlags, (hipEvent_t* e, unsigned fl), (e, fl))
WRAP(hipError_t, hipEventDestroy, (hipEvent_t e), (e))
WRAP(hipError_t, hipStreamCreateWithFlags, (hipStream_t* st, unsigned int fl), (st, fl))
WRAP(hipError_t, hipGetLastError, (void), ())
WRAP(hipError_t, hipPointerGetAttributes, (hipPointerAttribute_t* a, const void* p), (a, p))

extern "C" {
// control surface for the probe (ctypes on the same .so)
void hipstamps_enable(int on)
{
    if (on && !g_rec) g_rec = new Rec[CAP];
    g_on.store(on);
}
void hipstamps_reset(void) { g_n.store(0); }
size_t hipstamps_count(void) { size_t n = g_n.load(); return n < CAP ? n : CAP; }
uint64_t hipstamps_now(void) { return now(); }
// copies records [first, first + n) as (id, t0, t1) triples of uint64
size_t hipstamps_read(size_t first, size_t n, uint64_t* out)
{
    size_t have = hipstamps_count();
    if (first >= have) return 0;
    if (first + n > have) n = have - first;
    for (size_t i = 0; i < n; i++) { const Rec& r = g_rec[first + i]; out[3 * i] = r.id; out[3 * i + 1] = r.t0; out[3 * i + 2] = r.t1; }
    return n;
}
const char* hipstamps_name(int id) { return (id >= 0 && id < g_nnames.load()) ? g_names[id] : "?"; }
}
