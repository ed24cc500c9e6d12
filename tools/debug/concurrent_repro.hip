// Standalone reproducer: do kernels on two HIP streams of ONE process corrupt each other on this platform?
// No Python, no torch, no allocator, no library of this repository: one file, two hipStream_t.
//
//   victim    - a direct 510-point real DFT out of LDS with fp64 accumulation (the structure of storm_amd/csrc/spectral.hip::stft_kernel and of
//               any LDS-staged DFT): every launch computes the SAME frames, so its output must be bit-identical from launch to launch;
//   aggressor - (a) a loop of independent v_mfma_f32_32x32x16_bf16 with a register footprint chosen on the command line (<= 128 VGPRs so its
//               waves share SIMDs with the victim's, or 256 so they cannot), (b) a VALU-only fp32 FMA loop of the same duration (control),
//               (c) nothing (solo control).
// The victim loops on stream 1 for `seconds` while the aggressor is kept queued on stream 2; after every victim launch its output is compared
// on the host with the solo result bit for bit.  Prints launches / corrupted launches per (victim, aggressor) pair and the first difference.
//
//   hipcc --offload-arch=gfx950 -O3 tools/debug/concurrent_repro.hip -o gpurun_out/concurrent_repro && gpurun_out/concurrent_repro [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int NFFT = 510, NBIN = 256;

__global__ void dft_victim(const float* __restrict__ frames, const float2* __restrict__ tw, float2* __restrict__ out) {
    __shared__ float xs[NFFT];
    __shared__ float2 tws[NFFT];
    const float* x = frames + (size_t)blockIdx.x * NFFT;
    for (int k = threadIdx.x; k < NFFT; k += blockDim.x) { xs[k] = x[k]; tws[k] = tw[k]; }
    __syncthreads();
    for (int f = threadIdx.x; f < NBIN; f += blockDim.x) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int k = 0; k < NFFT; ++k) {
            const float2 w = tws[idx];
            re += (double)(xs[k] * w.x);
            im -= (double)(xs[k] * w.y);
            idx += f; if (idx >= NFFT) idx -= NFFT;
        }
        out[(size_t)blockIdx.x * NBIN + f] = make_float2((float)re, (float)im);
    }
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// NACC accumulators of 16 registers each: 4 -> ~80 VGPRs (several waves per SIMD: shares SIMDs with the victim), 14 -> 256 (one wave per SIMD pair)
template <int NACC>
__global__ void __launch_bounds__(256) mfma_aggressor(float* __restrict__ sink, int iters, uint32_t seed) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    union { bf16x8 v; uint32_t u[4]; } a, b;
    uint32_t s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
    for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; a.u[j] = (s & 0x007f007fu) | 0x3f003f00u; s = s * 1664525u + 1013904223u; b.u[j] = (s & 0x007f007fu) | 0x3f003f00u; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[i], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) sink[0] = t;
}

// the same with the operand traffic of a convolution: every MFMA's B fragment is re-read from a 64-KiB LDS image (ds_read_b128), the image
// rewritten between rounds - MFMA + LDS reads + LDS writes + barriers on the aggressor's side, like conv_igemm / a GEMM main loop
__global__ void __launch_bounds__(256) mfma_lds_aggressor(float* __restrict__ sink, int iters, uint32_t seed) {
    __shared__ uint4 img[4096];                              // 64 KiB
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint32_t s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    union { bf16x8 v; uint32_t u[4]; uint4 q; } a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; a.u[j] = (s & 0x007f007fu) | 0x3f003f00u; }
    for (int it = 0; it < iters; ++it) {
        if ((it & 63) == 0) {
            __syncthreads();
            for (int k = threadIdx.x; k < 4096; k += 256) { s = s * 1664525u + 1013904223u; img[k] = make_uint4((s & 0x007f007fu) | 0x3f003f00u, (s >> 3 & 0x007f007fu) | 0x3f003f00u, (s >> 5 & 0x007f007fu) | 0x3f003f00u, (s >> 7 & 0x007f007fu) | 0x3f003f00u); }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b.q = img[(threadIdx.x + 67 * (4 * it + i)) & 4095];
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[i], 0, 0, 0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) sink[0] = t;
}

__global__ void __launch_bounds__(256) valu_aggressor(float* __restrict__ sink, int iters) {
    float x0 = 0.37f + threadIdx.x * 1e-3f, x1 = 0.11f, x2 = 0.93f, x3 = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { x0 = fmaf(x0, 0.999f, x1); x1 = fmaf(x1, 0.998f, x2); x2 = fmaf(x2, 0.997f, x3); x3 = fmaf(x3, 0.996f, x0 * 1e-3f); }
    }
    if (x0 + x1 + x2 + x3 == 12345.678f) sink[0] = x0;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), %d CUs; %.1f s per pair\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, seconds);
    const int nframes = 2048;                                // victim: 2048 workgroups x 256 threads, ~1-2 per CU at a time next to the aggressor
    std::vector<float> hf((size_t)nframes * NFFT);
    std::vector<float> htw(2 * NFFT);
    uint32_t s = 12345u;
    for (auto& v : hf) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) - (1 << 22)) * (1.0f / (1 << 22)); }
    for (int k = 0; k < NFFT; ++k) { htw[2 * k] = (float)cos(2.0 * M_PI * k / NFFT); htw[2 * k + 1] = (float)sin(2.0 * M_PI * k / NFFT); }
    float *dframes, *dsink; float2 *dtw, *dout;
    CK(hipMalloc(&dframes, hf.size() * 4)); CK(hipMalloc(&dtw, NFFT * 8)); CK(hipMalloc(&dout, (size_t)nframes * NBIN * 8)); CK(hipMalloc(&dsink, 64));
    CK(hipMemcpy(dframes, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtw, htw.data(), NFFT * 8, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreate(&ev));
    const size_t nout = (size_t)nframes * NBIN * 2;
    std::vector<float> gold(nout), got(nout);
    // solo golden + solo reproducibility
    hipLaunchKernelGGL(dft_victim, dim3(nframes), dim3(256), 0, sv, dframes, dtw, dout);
    CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(gold.data(), dout, nout * 4, hipMemcpyDeviceToHost));

    struct Pair { const char* name; int kind; };             // kind: 0 none, 1 valu, 2 mfma 4 acc, 3 mfma 14 acc, 4 mfma + LDS
    const Pair pairs[] = {{"solo (no aggressor)", 0}, {"VALU fp32 FMA loop", 1}, {"MFMA bf16 32x32x16, 4 accumulators (shares SIMDs)", 2},
                          {"MFMA bf16 32x32x16, 14 accumulators (256 VGPRs)", 3}, {"MFMA + 64 KiB LDS image (reads, rewrites, barriers)", 4}};
    int total_bad = 0;
    for (const Pair& p : pairs) {
        long launches = 0, bad = 0, agg = 0;
        size_t first_idx = 0; float first_got = 0, first_want = 0;
        const double t0 = now();
        int inflight = 0;
        while (now() - t0 < seconds) {
            // keep the aggressor's queue fed: a few launches ahead, each ~1-2 ms
            if (p.kind != 0) {
                if (hipEventQuery(ev) == hipSuccess || inflight == 0) {
                    for (int q = 0; q < 4; ++q) {
                        if (p.kind == 1) hipLaunchKernelGGL(valu_aggressor, dim3(prop.multiProcessorCount * 4), dim3(256), 0, sa, dsink, 20000);
                        if (p.kind == 2) hipLaunchKernelGGL((mfma_aggressor<4>), dim3(prop.multiProcessorCount * 4), dim3(256), 0, sa, dsink, 20000, (uint32_t)agg);
                        if (p.kind == 4) hipLaunchKernelGGL(mfma_lds_aggressor, dim3(prop.multiProcessorCount * 2), dim3(256), 0, sa, dsink, 20000, (uint32_t)agg);
                        if (p.kind == 3) hipLaunchKernelGGL((mfma_aggressor<14>), dim3(prop.multiProcessorCount * 2), dim3(256), 0, sa, dsink, 6000, (uint32_t)agg);
                        ++agg;
                    }
                    CK(hipEventRecord(ev, sa));
                    inflight = 1;
                }
            }
            hipLaunchKernelGGL(dft_victim, dim3(nframes), dim3(256), 0, sv, dframes, dtw, dout);
            CK(hipMemcpyAsync(got.data(), dout, nout * 4, hipMemcpyDeviceToHost, sv));
            CK(hipStreamSynchronize(sv));
            ++launches;
            if (memcmp(got.data(), gold.data(), nout * 4) != 0) {
                if (bad == 0) for (size_t i = 0; i < nout; ++i) if (memcmp(&got[i], &gold[i], 4) != 0) { first_idx = i; first_got = got[i]; first_want = gold[i]; break; }
                ++bad;
            }
        }
        CK(hipStreamSynchronize(sa));
        printf("victim DFT beside %-52s : %6ld launches, %6ld corrupted (aggressor launches %ld)", p.name, launches, bad, agg);
        if (bad) printf("  first: element %zu (frame %zu, bin %zu, %s) got %.9g want %.9g", first_idx, first_idx / (2 * NBIN), (first_idx / 2) % NBIN, first_idx & 1 ? "im" : "re", first_got, first_want);
        printf("\n");
        total_bad += bad != 0;
    }
    printf("%s\n", total_bad ? "RESULT: concurrent queues corrupt the DFT victim on this box" : "RESULT: no corruption observed");
    return 0;
}
