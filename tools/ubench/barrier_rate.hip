// Micro-benchmark: cost of the workgroup barrier structure of conv_pipe.hip.  8 waves per CU (2 / SIMD):
//   mode 0: every wave: {16 MFMA, barrier}                     (lock-step)
//   mode 1: ping-pong: waves 0-3 and 4-7 alternate {16 MFMA | nothing} between barriers (2 barriers per 16 MFMA)
//   mode 2: ping-pong with 6 conflict-free ds_read_b128 in the idle interval and 6 in the MFMA interval
//   mode 3: 4 waves per CU (1 / SIMD): {32 MFMA (4x4 tiles... here 2x16 on 8 accumulators), barrier}
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int row = lane & 31, h = lane >> 5;
    const int off0 = row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
    const char* base = smem + (wave & 3) * 8192;
    bf16x8 a[2], b[4];
    for (int j = 0; j < 2; ++j) a[j] = *reinterpret_cast<const bf16x8*>(base + off0 + j * 4096);
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + off0 + (j & 1) * 4096);
    auto mma16 = [&]() {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi * 4 + ni], 0, 0, 0);
    };
    auto reads = [&](int it) {
        for (int j = 0; j < 2; ++j) a[j] = *reinterpret_cast<const bf16x8*>(base + (off0 ^ ((it & 3) << 5)) + j * 4096);
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + (off0 ^ ((it & 3) << 5)) + (j & 1) * 4096);
    };
    const int grp = wave >> 2;
    if (MODE == 0 || MODE == 3) {
        for (int it = 0; it < iters; ++it) { mma16(); if (MODE == 3) mma16(); bar(); }
    } else {
        if (grp == 1) bar();
        for (int it = 0; it < iters; ++it) {
            if (MODE == 2) reads(it);
            bar();                     // staging interval (the other group computes)
            if (MODE == 2) { __builtin_amdgcn_sched_barrier(0); }
            mma16();
            bar();
        }
        if (grp == 0) bar();
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

int main() {
    float* out;
    const int blocks = 256, iters = 4000;
    hipMalloc(&out, blocks * 512 * 4);
    for (int mode = 0; mode < 4; ++mode) {
        const int threads = mode == 3 ? 256 : 512;
        auto kern = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : k<3>;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 65536, 0, out, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 65536, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma_per_simd = (mode == 3 ? 32.0 : 32.0) * iters;          // MFMAs issued on one SIMD
        const double tf = 2.0 * 32 * 32 * 16 * mfma_per_simd * 4 * blocks / (ms * 1e-3) / 1e12;
        printf("mode %d: %.3f ms, %.1f ns per MFMA slot on a SIMD, %.0f TF/s\n", mode, ms, ms * 1e6 / mfma_per_simd, tf);
    }
    return 0;
}
