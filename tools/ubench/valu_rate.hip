// Micro-benchmark: issue cost (cycles per wave64 instruction) of the VALU operations the fused GroupNorm + SiLU transform and the
// epilogues are made of, 8 independent chains per lane, one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    float v[8]; f32x2 p[8]; _Float16 h[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.5f + 0.01f * (lane + i); p[i] = f32x2{v[i], v[i] + 0.1f}; h[i] = (_Float16)v[i]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) v[i] = __builtin_amdgcn_exp2f(v[i]);
            if (OP == 1) v[i] = __builtin_amdgcn_rcpf(v[i]);
            if (OP == 2) h[i] = __builtin_amdgcn_rcph(h[i]);
            if (OP == 3) asm volatile("v_exp_f16_e32 %0, %0" : "+v"(h[i]));
            if (OP == 4) p[i] = __builtin_elementwise_fma(p[i], f32x2{1.0001f, 0.9999f}, f32x2{0.001f, 0.002f});
            if (OP == 5) v[i] = fmaf(v[i], 1.0001f, 0.001f);
            if (OP == 6) p[i] = p[i] * f32x2{1.0001f, 0.9999f};
            if (OP == 7) asm volatile("v_exp_f32_e32 %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[i]), "+v"(p[i].x));   // trans + independent fma
            if (OP == 8) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(v[i]));
            if (OP == 9) asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(v[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + (float)h[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    std::vector<unsigned long long> h(blocks * 4);
    const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_rcp_f16", "v_exp_f16", "v_pk_fma_f32", "v_fma_f32", "v_pk_mul_f32", "v_exp_f32 + independent v_fma_f32", "v_sqrt_f32", "v_rsq_f32"};
    void (*ks[])(float*, unsigned long long*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>};
    for (int op = 0; op < 10; ++op) {
        hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        printf("%-40s %.2f ticks per wave instruction (1 wave / SIMD, 8 independent chains)\n", names[op], s / h.size() / (iters * 8.0));
    }
    return 0;
}
