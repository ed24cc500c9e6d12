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

// ACC = double: the product's structure (LDS operands, fp64 accumulation); ACC = float: the same loop without any fp64 instruction
template <typename ACC>
__global__ void dft_victim(const float* __restrict__ frames, const float2* __restrict__ tw, float2* __restrict__ out) {
    __shared__ float xs[NFFT];
    __shared__ float2 tws[NFFT];
    const float* x = frames + (size_t)blockIdx.x * NFFT;
    for (int k = threadIdx.x; k < NFFT; k += blockDim.x) { xs[k] = x[k]; tws[k] = tw[k]; }
    __syncthreads();
    for (int f = threadIdx.x; f < NBIN; f += blockDim.x) {
        ACC re = 0, im = 0;
        int idx = 0;
        for (int k = 0; k < NFFT; ++k) {
            const float2 w = tws[idx];
            re += (ACC)(xs[k] * w.x);
            im -= (ACC)(xs[k] * w.y);
            idx += f; if (idx >= NFFT) idx -= NFFT;
        }
        out[(size_t)blockIdx.x * NBIN + f] = make_float2((float)re, (float)im);
    }
}
// the same arithmetic with the operands in REGISTERS / global memory only (no LDS instruction in the kernel): every thread streams its
// frame and the twiddles through the caches
__global__ void dft_victim_nolds(const float* __restrict__ frames, const float2* __restrict__ tw, float2* __restrict__ out) {
    const float* x = frames + (size_t)blockIdx.x * NFFT;
    for (int f = threadIdx.x; f < NBIN; f += blockDim.x) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int k = 0; k < NFFT; ++k) {
            const float2 w = tw[idx];
            re += (double)(x[k] * w.x);
            im -= (double)(x[k] * w.y);
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
template <bool MFMA, bool REWRITE>
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
        if ((it & 63) == 0 && (REWRITE || it == 0)) {
            __syncthreads();
            for (int k = threadIdx.x; k < 4096; k += 256) { s = s * 1664525u + 1013904223u; img[k] = make_uint4((s & 0x007f007fu) | 0x3f003f00u, (s >> 3 & 0x007f007fu) | 0x3f003f00u, (s >> 5 & 0x007f007fu) | 0x3f003f00u, (s >> 7 & 0x007f007fu) | 0x3f003f00u); }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b.q = img[(threadIdx.x + 67 * (4 * it + i)) & 4095];
            if (MFMA) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[i], 0, 0, 0);
            else { acc[i][0] += __uint_as_float(b.u[0]); acc[i][1] += __uint_as_float(b.u[1]); acc[i][2] += __uint_as_float(b.u[2]); acc[i][3] += __uint_as_float(b.u[3]); }
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

// ---- part 2: is it the LDS itself?  A victim that only fills its LDS with a pattern, waits and re-checks it (no arithmetic), beside
// aggressors that only WRITE their own LDS allocation (no MFMA), for several allocation sizes.  Two workgroups of two different
// kernels must never see each other's LDS whatever their sizes: any mismatch here is an LDS allocation overlap on the CU.
__global__ void __launch_bounds__(256) lds_pattern_victim(unsigned* __restrict__ bad, unsigned* __restrict__ first, int words, int spins) {
    extern __shared__ unsigned vlds[];
    const unsigned tag = 0xA5000000u | ((blockIdx.x & 0xfff) << 12);
    for (int i = threadIdx.x; i < words; i += blockDim.x) vlds[i] = tag | (i & 0xfff);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spins) {}
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += blockDim.x) {
        const unsigned v = vlds[i];
        if (v != (tag | (i & 0xfff))) { if (atomicAdd(bad, 1u) == 0) { first[0] = (unsigned)i; first[1] = v; first[2] = blockIdx.x; first[3] = tag | (i & 0xfff); } }
    }
}
__global__ void __launch_bounds__(256) lds_writer_aggressor(float* __restrict__ sink, int words, int rounds) {
    extern __shared__ unsigned alds[];
    unsigned acc = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) alds[i] = 0x3f803f80u + (unsigned)r;     // (1.0 bf16 pairs: what an operand tile looks like)
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += blockDim.x) acc += alds[(i * 33 + r) % words];
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = (float)acc;
}

static int part2(hipStream_t sv, hipStream_t sa, float* dsink, int cus, double seconds) {
    unsigned *dbad, *dfirst;
    CK(hipMalloc(&dbad, 4)); CK(hipMalloc(&dfirst, 16));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_writer_aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_pattern_victim), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t ev;
    CK(hipEventCreate(&ev));
    int any = 0;
    const int vict_kb[] = {6, 24};
    const int agg_kb[] = {0, 16, 32, 48, 64, 80, 96, 128, 152};
    for (int vk : vict_kb) for (int ak : agg_kb) {
        if (vk + ak > 160) continue;
        CK(hipMemset(dbad, 0, 4));
        long launches = 0, agg = 0;
        int inflight = 0;
        const double t0 = now();
        while (now() - t0 < seconds) {
            if (ak > 0 && (inflight == 0 || hipEventQuery(ev) == hipSuccess)) {
                for (int q = 0; q < 4; ++q) { hipLaunchKernelGGL(lds_writer_aggressor, dim3(cus * 2), dim3(256), ak * 1024, sa, dsink, ak * 256, 300); ++agg; }
                CK(hipEventRecord(ev, sa));
                inflight = 1;
            }
            hipLaunchKernelGGL(lds_pattern_victim, dim3(cus * 4), dim3(256), vk * 1024, sv, dbad, dfirst, vk * 256, 20000);
            CK(hipStreamSynchronize(sv));
            ++launches;
        }
        CK(hipStreamSynchronize(sa));
        unsigned bad = 0, first[4] = {0, 0, 0, 0};
        CK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(first, dfirst, 16, hipMemcpyDeviceToHost));
        printf("LDS pattern victim (%3d KiB) beside LDS writer (%3d KiB dynamic LDS): %6ld launches, %9u corrupted words", vk, ak, launches, bad);
        if (bad) printf("  first: word %u of workgroup %u holds %08x, expected %08x", first[0], first[2], first[1], first[3]);
        printf("\n");
        any += bad != 0;
    }
    return any;
}


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
    struct Victim { const char* name; int kind; };
    const Victim victims[] = {{"DFT out of LDS, fp64 accumulation", 0}, {"DFT out of LDS, fp32 accumulation", 1}, {"DFT from global memory (no LDS), fp64 accumulation", 2}};
    struct Pair { const char* name; int kind; };
    const Pair pairs[] = {{"solo (no aggressor)", 0}, {"VALU fp32 FMA loop", 1}, {"MFMA bf16 32x32x16, 4 accumulators (128 VGPRs: shares SIMDs)", 2},
                          {"MFMA bf16 32x32x16, 14 accumulators (448 VGPRs)", 3}, {"MFMA fed from a 64 KiB LDS image (reads, rewrites, barriers)", 4},
                          {"MFMA fed from a 64 KiB LDS image (reads only)", 5}, {"LDS image reads + rewrites feeding VALU adds (no MFMA)", 6}};
    auto launch_victim = [&](int kind) {
        if (kind == 0) hipLaunchKernelGGL(dft_victim<double>, dim3(nframes), dim3(256), 0, sv, dframes, dtw, dout);
        if (kind == 1) hipLaunchKernelGGL(dft_victim<float>, dim3(nframes), dim3(256), 0, sv, dframes, dtw, dout);
        if (kind == 2) hipLaunchKernelGGL(dft_victim_nolds, dim3(nframes), dim3(256), 0, sv, dframes, dtw, dout);
    };
    int total_bad = 0;
    for (const Victim& v : victims) {
        launch_victim(v.kind);                               // solo golden of this victim
        CK(hipStreamSynchronize(sv));
        CK(hipMemcpy(gold.data(), dout, nout * 4, hipMemcpyDeviceToHost));
        printf("victim: %s\n", v.name);
        for (const Pair& p : pairs) {
            long launches = 0, bad = 0, agg = 0, bad_elems = 0;
            size_t first_idx = 0; float first_got = 0, first_want = 0;
            const double t0 = now();
            int inflight = 0;
            while (now() - t0 < seconds) {
                if (p.kind != 0 && (inflight == 0 || hipEventQuery(ev) == hipSuccess)) {     // keep the aggressor's queue fed, a few launches ahead
                    for (int q = 0; q < 4; ++q) {
                        const int cus = prop.multiProcessorCount;
                        if (p.kind == 1) hipLaunchKernelGGL(valu_aggressor, dim3(cus * 4), dim3(256), 0, sa, dsink, 20000);
                        if (p.kind == 2) hipLaunchKernelGGL((mfma_aggressor<4>), dim3(cus * 4), dim3(256), 0, sa, dsink, 20000, (uint32_t)agg);
                        if (p.kind == 3) hipLaunchKernelGGL((mfma_aggressor<14>), dim3(cus * 2), dim3(256), 0, sa, dsink, 6000, (uint32_t)agg);
                        if (p.kind == 4) hipLaunchKernelGGL((mfma_lds_aggressor<true, true>), dim3(cus * 2), dim3(256), 0, sa, dsink, 20000, (uint32_t)agg);
                        if (p.kind == 5) hipLaunchKernelGGL((mfma_lds_aggressor<true, false>), dim3(cus * 2), dim3(256), 0, sa, dsink, 20000, (uint32_t)agg);
                        if (p.kind == 6) hipLaunchKernelGGL((mfma_lds_aggressor<false, true>), dim3(cus * 2), dim3(256), 0, sa, dsink, 20000, (uint32_t)agg);
                        ++agg;
                    }
                    CK(hipEventRecord(ev, sa));
                    inflight = 1;
                }
                launch_victim(v.kind);
                CK(hipMemcpyAsync(got.data(), dout, nout * 4, hipMemcpyDeviceToHost, sv));
                CK(hipStreamSynchronize(sv));
                ++launches;
                if (memcmp(got.data(), gold.data(), nout * 4) != 0) {
                    long n = 0;
                    for (size_t i = 0; i < nout; ++i) if (memcmp(&got[i], &gold[i], 4) != 0) { if (n == 0 && bad == 0) { first_idx = i; first_got = got[i]; first_want = gold[i]; } ++n; }
                    bad_elems += n;
                    ++bad;
                }
            }
            CK(hipStreamSynchronize(sa));
            printf("  beside %-62s : %6ld launches, %6ld corrupted (aggressor launches %ld)", p.name, launches, bad, agg);
            if (bad) printf("  %.1f wrong values per corrupted launch of %zu; first: frame %zu bin %zu %s got %.9g want %.9g", (double)bad_elems / bad, nout,
                            first_idx / (2 * NBIN), (first_idx / 2) % NBIN, first_idx & 1 ? "im" : "re", first_got, first_want);
            printf("\n");
            total_bad += bad != 0;
        }
    }
    printf("%s\n", total_bad ? "RESULT part 1: concurrent queues corrupt the DFT victim on this box" : "RESULT part 1: no corruption observed");
    const int p2 = part2(sv, sa, dsink, prop.multiProcessorCount, seconds / 3);
    printf("%s\n", p2 ? "RESULT part 2: workgroups of two kernels on different queues see each other's LDS (allocation overlap)" : "RESULT part 2: LDS allocations stay disjoint");
    return 0;
}
