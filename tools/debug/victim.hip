// Debug: victim workgroups that detect interference from kernels on other streams.
//   mode 0: fill LDS with a pattern, idle, count the words that changed;  mode 1: re-read the pattern continuously (transient wrong READS);
//   mode 2: fp64 FMA chains in registers only (no LDS, no memory) against the value lane 0 .. 63 must all agree on.
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void lds_victim(unsigned* __restrict__ bad, unsigned* __restrict__ first, int spins, int words, int mode) {
    extern __shared__ unsigned lds[];
    const unsigned tag = 0xA5000000u | (blockIdx.x << 12);
    if (mode == 2) {
        double acc = 0.0;
        const float x0 = 0.3713f, w0 = 0.9371f;
        for (int k = 0; k < spins; ++k) {
            const float x = x0 + 1e-3f * (float)(k & 255), w = w0 - 1e-3f * (float)(k & 127);
            acc += (double)(x * w);
        }
        const double ref = __shfl(acc, 0, 64);          // every lane computes the same chain
        unsigned long long a, r;
        memcpy(&a, &acc, 8); memcpy(&r, &ref, 8);
        if (a != r) { if (atomicAdd(bad, 1u) == 0) { first[0] = threadIdx.x; first[1] = (unsigned)a; first[2] = blockIdx.x; first[3] = (unsigned)r; } }
        return;
    }
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = tag | (i & 0xfff);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spins) {}
    } else {
        int idx = threadIdx.x;
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spins) {
            for (int rep = 0; rep < 64; ++rep) {
                idx += 61 * (threadIdx.x & 15) + 17; if (idx >= words) idx -= words; if (idx >= words) idx %= words;
                const unsigned v = lds[idx];
                if (v != (tag | (idx & 0xfff))) { if (atomicAdd(bad, 1u) == 0) { first[0] = (unsigned)idx; first[1] = v; first[2] = blockIdx.x; first[3] = threadIdx.x; } }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const unsigned v = lds[i];
        if (v != (tag | (i & 0xfff))) {
            if (atomicAdd(bad, 1u) == 0) { first[0] = (unsigned)i; first[1] = v; first[2] = blockIdx.x; first[3] = 0xffffffffu; }
        }
    }
}
extern "C" int victim_launch(unsigned* bad, unsigned* first, int blocks, int lds_bytes, int spins, int mode, void* stream) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_victim), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(lds_victim, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, bad, first, spins, lds_bytes / 4, mode);
    return (int)hipGetLastError();
}
