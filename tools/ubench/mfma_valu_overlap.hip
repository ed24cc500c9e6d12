// Micro-benchmark: do the matrix pipe and the VALU overlap ACROSS the two waves of a SIMD?  Waves 0-3 of a 512-thread workgroup
// (one per SIMD) run back-to-back independent v_mfma_f32_32x32x16_bf16; waves 4-7 (the second wave of every SIMD) run the fused
// GroupNorm + SiLU transform's instruction mix (fma, exp2, add, rcp, mul, pack).  Modes: MFMA waves alone, VALU waves alone,
// both, both with s_setprio 1 on the MFMA waves; and the same VALU work inside the MFMA wave's own stream (interleaved).
// hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float silu_like(float x, float s, float t) {
    const float y = fmaf(x, s, t);
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
}

// mode bit 0: MFMA waves work, bit 1: VALU waves work, bit 2: setprio 1 on the MFMA waves, bit 3: VALU work interleaved
// into the MFMA waves' own stream (VALU waves idle)
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters, int mode, int valu_per_iter) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mfma_wave = wave < 4;
    float sink = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma_wave && (mode & 1)) {
        f32x16 acc[8];
        for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        bf16x8 a[2], b[4];
        for (int j = 0; j < 2; ++j) for (int e = 0; e < 8; ++e) a[j][e] = (__bf16)(0.001f * (lane + j + e));
        for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(0.002f * (lane + j - e));
        float v[8];
        for (int e = 0; e < 8; ++e) v[e] = 0.01f * (lane + e);
        if (mode & 4) __builtin_amdgcn_s_setprio(1);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i >> 2], b[i & 3], acc[i], 0, 0, 0);
                if (mode & 8) {                              // one element of transform work behind every MFMA
                    v[i] = silu_like(v[i], 1.0001f, 0.01f);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        for (int j = 0; j < 8; ++j) for (int r = 0; r < 16; ++r) sink += acc[j][r];
        for (int e = 0; e < 8; ++e) sink += v[e];
    } else if (!mfma_wave && (mode & 2)) {
        float v[8];
        for (int e = 0; e < 8; ++e) v[e] = 0.01f * (lane + e);
        for (int it = 0; it < iters; ++it)
            for (int r = 0; r < valu_per_iter; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = silu_like(v[e], 1.0001f, 0.01f);
        for (int e = 0; e < 8; ++e) sink += v[e];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = sink;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    const int blocks = 256, iters = 4000;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    std::vector<unsigned long long> h(blocks * 8);
    const char* names[] = {"", "MFMA waves alone", "VALU waves alone", "both", "", "MFMA alone (prio 1)", "", "both, MFMA waves at prio 1",
                           "", "MFMA + interleaved VALU, one wave / SIMD"};
    for (int vpi : {1, 2})
        for (int mode : {1, 2, 3, 7, 9}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, mode, vpi);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, mode, vpi);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
            double m = 0, v = 0; int nm = 0, nv = 0;
            for (int i = 0; i < blocks * 8; ++i) { if ((i & 7) < 4) { m += (double)h[i]; ++nm; } else { v += (double)h[i]; ++nv; } }
            // s_memtime ticks at 100 MHz: report per-iteration wall time instead, and the rates the two streams achieved
            const double us_iter = ms * 1e3 / iters;
            printf("valu/iter %d  %-44s %8.3f ms  %.4f us/iter  | MFMA waves %7.0f ticks, VALU waves %7.0f ticks\n", vpi, names[mode], ms,
                   us_iter, m / nm, v / nv);
        }
    return 0;
}
