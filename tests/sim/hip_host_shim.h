// TEST INFRASTRUCTURE: a minimal "HIP on the host" shim so the kernel sources under
// storm_amd/csrc compile with a host C++ compiler and run lane-accurately on the CPU
// (tests/sim/build_sim.py -> tests/sim/libstorm_sim.so).  Purpose: exercise the kernels'
// index math, LDS layouts, barriers and host launch code in the GPU-less build container.
// Every GPU thread is a fiber; a workgroup's fibers run on one OS thread; __syncthreads and
// wave-level collectives (shuffles, MFMA) are cooperative yield points.  The MFMA emulation
// encodes the documented gfx950 fragment layouts (cdna_hip_programming.md section 3); the real
// hardware check is the -m gpu test-suite.  Never used by the product.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ thread_local

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) uint2 { uint32_t x, y; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

namespace simrt {
struct Idx { unsigned x, y, z; };
Idx thread_idx();
Idx block_idx();
dim3 block_dim();
dim3 grid_dim();
void block_barrier();
typedef void (*CollFn)(const void* const* in, char (*out)[64], int nlanes, long long ctx);
void wave_collective(const void* my_in, void* my_out, size_t out_bytes, CollFn fn, long long ctx);
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
inline void noop_fn(const void* const*, char (*)[64], int, long long) {}
inline void wave_rendezvous() { int d = 0, r; wave_collective(&d, &r, 0, &noop_fn, 0); }
}  // namespace simrt

#define threadIdx (simrt::thread_idx())
#define blockIdx (simrt::block_idx())
#define blockDim (simrt::block_dim())
#define gridDim (simrt::grid_dim())
#define __syncthreads() simrt::block_barrier()

// ---- math / bit helpers ---------------------------------------------------------------------
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }

inline double atomicAdd(double* p, double v) {
    uint64_t* q = reinterpret_cast<uint64_t*>(p);
    uint64_t old = __atomic_load_n(q, __ATOMIC_RELAXED), neu;
    double o;
    do { memcpy(&o, &old, 8); const double n = o + v; memcpy(&neu, &n, 8); }
    while (!__atomic_compare_exchange_n(q, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return o;
}

namespace simrt {
template <typename V> inline void shfl_xor_fn(const void* const* in, char (*out)[64], int nlanes, long long mask) {
    for (int l = 0; l < nlanes; ++l) {
        const int src = (l ^ (int)mask) < nlanes ? (l ^ (int)mask) : l;
        memcpy(out[l], in[src], sizeof(V));
    }
}
}  // namespace simrt
namespace simrt {
inline void any_fn(const void* const* in, char (*out)[64], int nlanes, long long) {      // wave vote: OR over the lanes
    int any = 0;
    for (int l = 0; l < nlanes; ++l) any |= *static_cast<const int*>(in[l]);
    for (int l = 0; l < nlanes; ++l) memcpy(out[l], &any, sizeof(any));
}
}  // namespace simrt
namespace simrt {
template <typename V> inline void shfl_idx_fn(const void* const* in, char (*out)[64], int nlanes, long long src) {
    for (int l = 0; l < nlanes; ++l) memcpy(out[l], in[(int)src < nlanes ? (int)src : l], sizeof(V));
}
}  // namespace simrt
template <typename V> inline V __shfl(V v, int src, int width = 64) {       // (uniform source lane)
    (void)width;
    V r;
    simrt::wave_collective(&v, &r, sizeof(V), &simrt::shfl_idx_fn<V>, src);
    return r;
}
template <typename V> inline V __shfl_xor(V v, int mask, int width = 64) {
    (void)width;
    V r;
    simrt::wave_collective(&v, &r, sizeof(V), &simrt::shfl_xor_fn<V>, mask);
    return r;
}

// ---- MFMA emulation ----------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 sim_bf16x8;
typedef __attribute__((ext_vector_type(16))) float sim_f32x16;
namespace simrt {
struct MfmaIn { float a[8], b[8]; int nk; sim_f32x16 c; };
// D[i][j] += sum_{h in 0..1} sum_{e<nk} A(lane i+32h).a[e] * B(lane j+32h).b[e];
// lane holds column j = lane&31 and rows (r&3) + 8*(r>>2) + 4*(lane>>5), r in [0,16).
inline void mfma32_fn(const void* const* in, char (*out)[64], int nlanes, long long) {
    if (nlanes != 64) { fprintf(stderr, "simrt: MFMA with a partial wave\n"); abort(); }
    for (int l = 0; l < 64; ++l) {
        const MfmaIn* me = static_cast<const MfmaIn*>(in[l]);
        sim_f32x16 d = me->c;
        const int j = l & 31;
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            float acc = 0.f;
            for (int h = 0; h < 2; ++h) {
                const MfmaIn* A = static_cast<const MfmaIn*>(in[i + 32 * h]);
                const MfmaIn* B = static_cast<const MfmaIn*>(in[j + 32 * h]);
                for (int e = 0; e < me->nk; ++e) acc += A->a[e] * B->b[e];
            }
            d[r] += acc;
        }
        memcpy(out[l], &d, sizeof(d));
    }
}
inline float bf2f(__bf16 h) { uint16_t u; memcpy(&u, &h, 2); uint32_t w = (uint32_t)u << 16; float f; memcpy(&f, &w, 4); return f; }
inline sim_f32x16 mfma_bf16(sim_bf16x8 a, sim_bf16x8 b, sim_f32x16 c, int, int, int) {
    MfmaIn in; in.nk = 8; in.c = c;
    for (int e = 0; e < 8; ++e) { in.a[e] = bf2f(a[e]); in.b[e] = bf2f(b[e]); }
    sim_f32x16 r;
    wave_collective(&in, &r, sizeof(r), &mfma32_fn, 0);
    return r;
}
typedef __attribute__((ext_vector_type(8))) _Float16 sim_f16x8;
inline sim_f32x16 mfma_f16(sim_f16x8 a, sim_f16x8 b, sim_f32x16 c, int, int, int) {
    MfmaIn in; in.nk = 8; in.c = c;
    for (int e = 0; e < 8; ++e) { in.a[e] = (float)a[e]; in.b[e] = (float)b[e]; }
    sim_f32x16 r;
    wave_collective(&in, &r, sizeof(r), &mfma32_fn, 0);
    return r;
}
inline sim_f32x16 mfma_f32(float a, float b, sim_f32x16 c, int, int, int) {
    MfmaIn in; in.nk = 1; in.c = c; in.a[0] = a; in.b[0] = b;
    sim_f32x16 r;
    wave_collective(&in, &r, sizeof(r), &mfma32_fn, 0);
    return r;
}
}  // namespace simrt
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 simrt::mfma_bf16
#define __builtin_amdgcn_mfma_f32_32x32x2f32 simrt::mfma_f32
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 simrt::mfma_f16

// ---- runtime API stubs ---------------------------------------------------------------------------
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };
inline const char* hipGetErrorString(hipError_t) { return "sim"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    strcpy(p->gcnArchName, "host-sim"); p->multiProcessorCount = 0; p->totalGlobalMem = 0; return hipSuccess;
}
typedef int hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
// HIP graphs do not exist on the simulator: a capture attempt fails and storm_ncsnpp_forward stays on its eager path
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum { hipStreamNonBlocking = 1, hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 1; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, int) { return 1; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return 1; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t) { return 1; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return 1; }
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) \
    simrt::launch((grid), (block), (size_t)(lds), [=]() { (kern)(__VA_ARGS__); })

// ---- LDS-DMA queue model (used by storm_amd/csrc/hw.h's simulator versions of dma16 / vm_wait) ------------------------
// Host simulation of the asynchronous LDS-DMA queue (test infrastructure).  Two extremes bracket the hardware:
//   STORM_SIM_DMA unset : a copy lands the moment it is issued  -> exposes write-after-read hazards (a DMA issued
//                         while another wave may still read the destination);
//   STORM_SIM_DMA=late  : a copy lands only when the issuing lane's counted vm_wait<N> retires it (in issue order,
//                         leaving the N newest in flight)        -> exposes read-after-write hazards (a fragment read
//                         before the wait + barrier that publishes the data).
namespace simdma {
struct Entry { char* dst; char data[16]; };
inline bool late() { static const bool v = [] { const char* e = getenv("STORM_SIM_DMA"); return e && e[0] == 'l'; }(); return v; }
inline std::vector<Entry>& queue() {                     // per simulated thread (fibers of a workgroup share an OS thread)
    static thread_local std::vector<std::vector<Entry>> q(1024);
    return q[threadIdx.x];
}
inline void retire(int keep) {
    std::vector<Entry>& q = queue();
    const size_t n = q.size() > (size_t)keep ? q.size() - (size_t)keep : 0;
    for (size_t i = 0; i < n; ++i) memcpy(q[i].dst, q[i].data, 16);
    q.erase(q.begin(), q.begin() + (long)n);
}
}  // namespace simdma
