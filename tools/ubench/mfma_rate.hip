// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 from one wave / two waves per SIMD, with and without
// interleaved LDS fragment reads (the conv kernels' inner pattern).  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters, int waves_active, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) {
        if (rnd) {                                           // two pseudo-random bf16 in (-2, 2) per dword: realistic switching activity
            unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const unsigned lo = 0x3f00u | (h & 0x80ffu), hi = 0x3f00u | ((h >> 16) & 0x80ffu);
            reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
        } else reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.001f;
    }
    __syncthreads();
    if (wave >= waves_active) return;
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a[2], b[4];
    // conv_index.h layout: 128-B rows, 16-B slots XOR-swizzled by (row >> 1) & 7 -> conflict-free ds_read_b128
    const int row = lane & 31, h = lane >> 5;
    const int off0 = row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
    const char* base = smem + (wave & 3) * 8192;
    for (int j = 0; j < 2; ++j) a[j] = *reinterpret_cast<const bf16x8*>(base + off0 + j * 4096);
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + off0 + j * 4096 * 0 + j * 32 * 128 % 8192);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {   // re-read the fragments every 8 MFMAs (k-group it & 3 of the chunk), reads right before use
            for (int j = 0; j < 2; ++j) a[j] = *reinterpret_cast<const bf16x8*>(base + (off0 ^ ((it & 3) << 5)) + j * 4096);
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + (off0 ^ ((it & 3) << 5)) + (j & 1) * 4096);
        }
        if (MODE == 2) {   // same reads, software pipelined: fragments of the NEXT group are fetched before this group's MFMAs
            bf16x8 an[2], bn[4];
            for (int j = 0; j < 2; ++j) an[j] = *reinterpret_cast<const bf16x8*>(base + (off0 ^ (((it + 1) & 3) << 5)) + j * 4096);
            for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + (off0 ^ (((it + 1) & 3) << 5)) + (j & 1) * 4096);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi * 4 + ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            for (int j = 0; j < 2; ++j) a[j] = an[j];
            for (int j = 0; j < 4; ++j) b[j] = bn[j];
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi * 4 + ni], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// The 128 x 128 wave tile (16 accumulator tiles = 256 accumulator registers, 512-register budget, ONE wave per SIMD): 8 fragment
// reads per 16 MFMAs instead of 6 per 8.  MODE 0 registers only, 1 reads right before use, 2 software pipelined.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k16(float* out, unsigned long long* cyc, int iters, int waves_active, int rnd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) {
        if (rnd) {
            unsigned h = (unsigned)i * 2654435761u + (unsigned)blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const unsigned lo = 0x3f00u | (h & 0x80ffu), hi = 0x3f00u | ((h >> 16) & 0x80ffu);
            reinterpret_cast<unsigned*>(smem)[i] = lo | (hi << 16);
        } else reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.001f;
    }
    __syncthreads();
    if (wave >= waves_active) return;
    f32x16 acc[16];
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a[4], b[4];
    const int row = lane & 31, h = lane >> 5;
    const int off0 = row * 128 + ((h ^ ((row >> 1) & 7)) << 4);
    const char* base = smem + (wave & 3) * 1024;
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const bf16x8*>(base + off0 + j * 4096);
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + off0 + j * 4096);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const bf16x8*>(base + (off0 ^ ((it & 3) << 5)) + j * 4096);
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + (off0 ^ ((it & 3) << 5)) + j * 4096);
        }
        if (MODE == 2) {
            bf16x8 an[4], bn[4];
            for (int j = 0; j < 4; ++j) an[j] = *reinterpret_cast<const bf16x8*>(base + (off0 ^ (((it + 1) & 3) << 5)) + j * 4096);
            for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const bf16x8*>(base + 32768 + (off0 ^ (((it + 1) & 3) << 5)) + j * 4096);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi * 4 + ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            for (int j = 0; j < 4; ++j) { a[j] = an[j]; b[j] = bn[j]; }
            continue;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi * 4 + ni], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 16; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    std::vector<unsigned long long> h(blocks * 8);
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int mode = 0; mode < 3; ++mode)
        for (int wa : {4, 8}) {
            hipMemset(cyc, 0, blocks * 8 * 8);
            auto kern = mode == 0 ? k<0> : (mode == 1 ? k<1> : k<2>);
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 65536, 0, out, cyc, iters, wa, rnd);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 65536, 0, out, cyc, iters, wa, rnd);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
            double sum = 0; int n = 0;
            for (auto v : h) if (v) { sum += (double)v; ++n; }
            const double cyc_per_mfma = sum / n / (iters * 8.0);
            const double tf = 2.0 * 32 * 32 * 16 * 8.0 * iters * wa * blocks / (ms * 1e-3) / 1e12;
            printf("%s operands, mode %d (%s) waves/CU %d: %.1f ticks per MFMA per wave, %.3f ms, %.0f TF/s, clock ~%.2f GHz\n", rnd ? "random" : "regular small", mode,
                   mode == 0 ? "registers only" : (mode == 1 ? "LDS re-read each group" : "LDS re-read, pipelined"), wa, cyc_per_mfma, ms, tf, sum / n / (ms * 1e6));
        }
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(cyc, 0, blocks * 8 * 8);
        auto kern = mode == 0 ? k16<0> : (mode == 1 ? k16<1> : k16<2>);
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int it16 = iters / 2;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, out, cyc, it16, 4, rnd);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, out, cyc, it16, 4, rnd);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
        double sum = 0; int n = 0;
        for (auto v : h) if (v) { sum += (double)v; ++n; }
        const double tf = 2.0 * 32 * 32 * 16 * 16.0 * it16 * 4 * blocks / (ms * 1e-3) / 1e12;
        printf("%s operands, 128x128 wave tile, mode %d (%s) waves/CU 4: %.1f ticks per MFMA per wave, %.3f ms, %.0f TF/s\n", rnd ? "random" : "regular small", mode,
               mode == 0 ? "registers only" : (mode == 1 ? "8 LDS reads per 16 MFMAs" : "8 LDS reads per 16 MFMAs, pipelined"), sum / n / (it16 * 16.0), ms, tf);
    }
    return 0;
}
