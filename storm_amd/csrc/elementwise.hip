// Small memory-bound kernels around the network: input packing, time embedding, per-block
// temb projections, attention row-softmax and the output head.  Reference map in
// include/storm_hip.h.
#include "common.h"

namespace storm {

// ---- input packing: complex [B,F,T] x n_in -> NHWC [B][F][T][8], x -> 2x-1 ----------------
struct PackPtrs { const float* p[3]; };

template <typename T>
__global__ void pack_input_kernel(PackPtrs in, int n_in, T* __restrict__ out, long long npix) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < n_in) {
            const float2 z = reinterpret_cast<const float2*>(in.p[k])[i];
            v[2 * k] = 2.0f * z.x - 1.0f;
            v[2 * k + 1] = 2.0f * z.y - 1.0f;
        } else {
            v[2 * k] = 0.f; v[2 * k + 1] = 0.f;
        }
    }
    v[6] = 0.f; v[7] = 0.f;
    store8(out + i * 8, v);
}

// ---- time embedding ----------------------------------------------------------------------
// temb = SiLU(L2(SiLU(L1(cat[sin, cos](2 pi log(t) W)))))  (ncsnpp.py:298-317; the blocks consume SiLU(temb), layerspp.py:262).
// One wave per output row of a Linear: the row is read as 16-byte pieces (coalesced), the input vector sits in LDS, the dot
// product is one wave reduction.  Grid (B, TEMB_SPLIT): every workgroup computes the first layer (cheap) and its share of the
// second one.  (The first version gave every THREAD a row and walked it sequentially, uncoalesced: 67 us for 6 MFLOP.)
constexpr int TEMB_SPLIT = 4, TEMB_ROWS = 8, TEMB_ROWS1 = 8, TEMB_THREADS = 1024;       // (32 rows of the first layer in ONE pass measured no faster: 43.7 vs 41.2 us - the passes are not where this launch's time goes)
// (nor the two double-precision calls per projection: sine and cosine on separate threads 23.6 -> 23.3 us.  Alone and repeated the launch takes 23 us, inside an evaluation 41: there its 1.5 MB of fp32
//  weights come from HBM every time - the evaluation in between moved hundreds of MB through the L2 - and each dependent round trip is that much longer)
// TEMB_ROWS rows at once: all their loads are issued before the first reduction (one memory round trip per 8 rows, not per row)
// The bias of a row is fetched WITH its weights (lane r holds row r's): round 5 found the first form - lane 0 loading bias[n] inside `emit`, behind
// each row's reduction - a chain of eight exposed round trips per pass (47 us per launch at every batch size, rocprofv3; ~20 us of it bias latency).
template <int ROWS, typename F>
__device__ __forceinline__ void rows_dot(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ x, int K, int n0, int n_end,
                                         int nstep, int lane, F&& emit) {
    constexpr int TEMB_ROWS = ROWS;                  // rows per pass: all their loads are in flight together (one memory round trip per pass)
    for (int n = n0; n < n_end; n += nstep * TEMB_ROWS) {
        float acc[TEMB_ROWS];
        const int brow = n + (lane & (TEMB_ROWS - 1)) * nstep;
        const float bv = bias[brow < n_end ? brow : n];
#pragma unroll
        for (int r = 0; r < TEMB_ROWS; ++r) {
            acc[r] = 0.f;
            const int row = n + r * nstep < n_end ? n + r * nstep : n;            // (clamped: the value is dropped below)
            for (int k = 4 * lane; k < K; k += 256) {
                const float4 a = *reinterpret_cast<const float4*>(W + (long long)row * K + k), v = *reinterpret_cast<const float4*>(x + k);
                acc[r] = fmaf(a.x, v.x, acc[r]); acc[r] = fmaf(a.y, v.y, acc[r]); acc[r] = fmaf(a.z, v.z, acc[r]); acc[r] = fmaf(a.w, v.w, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < TEMB_ROWS; ++r) {
            const float v = wave_sum(acc[r]);
            const float b = __shfl(bv, r, 64);
            if (lane == 0 && n + r * nstep < n_end) emit(n + r * nstep, v + b);
        }
    }
}
__global__ __launch_bounds__(TEMB_THREADS)
void time_embedding_kernel(const float* __restrict__ t, const float* __restrict__ gW,
                           const float* __restrict__ W1, const float* __restrict__ b1,
                           const float* __restrict__ W2, const float* __restrict__ b2,
                           float* __restrict__ out, int nf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* emb = reinterpret_cast<float*>(smem);        // [2 nf]
    float* h1 = emb + 2 * nf;                            // [4 nf]
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
    const float lt = (float)log((double)t[b]);           // correctly rounded logf
    for (int k = tid; k < nf; k += nt) {
        const float xp = ((lt * gW[k]) * 2.0f) * 3.14159274101257324f;   // layerspp.py:40 op order
        emb[k] = (float)sin((double)xp);
        emb[nf + k] = (float)cos((double)xp);
    }
    __syncthreads();
    const int E = 2 * nf, Hd = 4 * nf;
    rows_dot<TEMB_ROWS1>(W1, b1, emb, E, wave, Hd, nw, lane, [&](int n, float v) { h1[n] = silu_f(v); });
    __syncthreads();
    const int per = (Hd + gridDim.y - 1) / gridDim.y, n0 = blockIdx.y * per, n1 = min(Hd, n0 + per);
    rows_dot<TEMB_ROWS>(W2, b2, h1, Hd, n0 + wave, n1, nw, lane, [&](int n, float v) { out[(long long)b * Hd + n] = silu_f(v); });   // blocks consume SiLU(temb)
}

// ---- dense: out[b][n] = W[n][:] . x[b][:] + bias[n]; one wave per output row n ------------
// The weight row is read ONCE per 8 batch rows (K <= 64 * DENSE_KMAX: once per wave) and the batch rows are accumulated side by side,
// so a wave has one memory round trip instead of B dependent ones (the first version: 40 us for the 22 Dense_0 layers of a forward,
// latency bound).  Wider rows (nf > 128: K = 4 nf > 512) walk the row in pieces of 64 * DENSE_KMAX columns; a lane's summation order
// (k = lane, lane + 64, ...) is the same for every K.
constexpr int DENSE_KMAX = 8, DENSE_BT = 8;
// Round 4: the eight batch rows of an iteration are staged in LDS once per workgroup (DENSE_BT x K floats) instead of being read
// from global memory by every wave - 64 four-byte loads per lane and iteration, 160 MB of L1 / L2 traffic for 32 KiB of data
// (the 22 Dense_0 layers of a forward: 40 -> ~15 us).  STAGE = false: rows wider than the LDS stage, read in place as before.
template <bool STAGE>
__global__ void dense_kernel(const float* __restrict__ x, const float* __restrict__ W,
                             const float* __restrict__ bias, float* __restrict__ out, int B, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const xs = reinterpret_cast<float*>(smem);        // [DENSE_BT][K] (STAGE)
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool live = n < N;                                 // (no early return: every wave takes part in the staging barriers)
    const float* w = W + (long long)(live ? n : 0) * K;
    const float bn = live ? bias[n] : 0.f;
    for (int b0 = 0; b0 < B; b0 += DENSE_BT) {
        if (STAGE) {
            if (b0 > 0) __syncthreads();
            for (int i = threadIdx.x; i < DENSE_BT * K; i += blockDim.x) {
                const int j = i / K, k = i - j * K;
                const int b = b0 + j < B ? b0 + j : B - 1;
                xs[i] = x[(long long)b * K + k];
            }
            __syncthreads();
        }
        float acc[DENSE_BT];
#pragma unroll
        for (int j = 0; j < DENSE_BT; ++j) acc[j] = 0.f;
        for (int k0 = 0; k0 < K; k0 += 64 * DENSE_KMAX) {
            float wr[DENSE_KMAX];
#pragma unroll
            for (int i = 0; i < DENSE_KMAX; ++i) wr[i] = k0 + lane + 64 * i < K ? w[k0 + lane + 64 * i] : 0.f;
#pragma unroll
            for (int j = 0; j < DENSE_BT; ++j) {
                const int b = b0 + j < B ? b0 + j : B - 1;
#pragma unroll
                for (int i = 0; i < DENSE_KMAX; ++i)
                    if (k0 + 64 * i < K) {
                        const int k = k0 + lane + 64 * i;
                        const float xv = k < K ? (STAGE ? xs[j * K + k] : x[(long long)b * K + k]) : 0.f;
                        acc[j] = fmaf(wr[i], xv, acc[j]);
                    }
            }
        }
#pragma unroll
        for (int j = 0; j < DENSE_BT; ++j) {
            const float v = wave_sum(acc[j]);
            if (live && lane == 0 && b0 + j < B) out[(long long)(b0 + j) * N + n] = v + bn;
        }
    }
}

// ---- row softmax: one wave per row, wave64 reductions -------------------------------------
template <typename T>
__global__ void softmax_rows_kernel(const float* __restrict__ s, T* __restrict__ p, long long rows, int L, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = s + row * ld;
    float m = -INFINITY;
    for (int k = lane; k < L; k += 64) m = fmaxf(m, x[k]);
    m = wave_max(m);
    float sum = 0.f;
    for (int k = lane; k < L; k += 64) sum += expf(x[k] - m);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    T* o = p + row * ld;
    for (int k = lane; k < L; k += 64) from_f32(o[k], expf(x[k] - m) * inv);
    for (int k = L + lane; k < ld; k += 64) from_f32(o[k], 0.0f);       // zero the row padding
}

// ---- output head --------------------------------------------------------------------------
template <typename T>
__global__ void output_head_kernel(const T* __restrict__ pyr, const float* __restrict__ t,
                                   const float* __restrict__ W, const float* __restrict__ bias, int cin,
                                   float* __restrict__ out, long long npix_per_b, long long npix, float sign) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int b = (int)(i / npix_per_b);
    float v[8];
    load8(pyr + i * 8, v);
    const float tb = t ? t[b] : 1.0f;
    float o0 = 0.f, o1 = 0.f;
    for (int c = 0; c < cin; ++c) {
        const float h = t ? v[c] / tb : v[c];
        o0 = fmaf(W[c], h, o0);
        o1 = fmaf(W[cin + c], h, o1);
    }
    reinterpret_cast<float2*>(out)[i] = make_float2(sign * (o0 + bias[0]), sign * (o1 + bias[1]));
}

}  // namespace storm

using namespace storm;

extern "C" int storm_pack_input(const float* const* cplx_in, int n_in, void* out, int B, int F, int T, int dtype,
                                storm_stream_t s) {
    STORM_CHECK(cplx_in && out && n_in >= 1 && n_in <= 3, "storm_pack_input: n_in=%d", n_in);
    PackPtrs pp = {{nullptr, nullptr, nullptr}};
    for (int i = 0; i < n_in; ++i) { STORM_CHECK(cplx_in[i], "storm_pack_input: null input %d", i); pp.p[i] = cplx_in[i]; }
    const long long npix = (long long)B * F * T;
    const int nb = cdiv(npix, 256);
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) hipLaunchKernelGGL((pack_input_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, pp, n_in, (bf16_t*)out, npix);
    else if (dtype == STORM_F16) hipLaunchKernelGGL((pack_input_kernel<half_t>), dim3(nb), dim3(256), 0, st, pp, n_in, (half_t*)out, npix);
    else if (dtype == STORM_F32) hipLaunchKernelGGL((pack_input_kernel<float>), dim3(nb), dim3(256), 0, st, pp, n_in, (float*)out, npix);
    else STORM_CHECK(false, "storm_pack_input: dtype %d", dtype);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_time_embedding(const float* t, const float* gfp_W, const float* W1, const float* b1,
                                    const float* W2, const float* b2, float* act_temb, int B, int nf,
                                    storm_stream_t s) {
    STORM_CHECK(t && gfp_W && W1 && b1 && W2 && b2 && act_temb && B > 0 && nf > 0, "storm_time_embedding: bad arguments");
    const size_t lds = (size_t)6 * nf * sizeof(float);
    STORM_CHECK(lds <= 64 * 1024, "storm_time_embedding: nf=%d too large", nf);
    STORM_CHECK(nf % 2 == 0, "storm_time_embedding: nf=%d must be even (16-byte row pieces)", nf);
    hipLaunchKernelGGL(time_embedding_kernel, dim3(B, TEMB_SPLIT), dim3(TEMB_THREADS), lds, (hipStream_t)s, t, gfp_W, W1, b1, W2, b2, act_temb, nf);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_dense(const float* x, const float* W, const float* bias, float* out, int B, int N, int K,
                           storm_stream_t s) {
    STORM_CHECK(x && W && bias && out && B > 0 && N > 0 && K > 0, "storm_dense: bad arguments");
    const size_t lds = (size_t)DENSE_BT * K * sizeof(float);
    if (lds <= 48 * 1024) hipLaunchKernelGGL(dense_kernel<true>, dim3(cdiv(N, 4)), dim3(256), lds, (hipStream_t)s, x, W, bias, out, B, N, K);
    else hipLaunchKernelGGL(dense_kernel<false>, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)s, x, W, bias, out, B, N, K);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_softmax_rows(const float* scores, void* probs, long long rows, int L, int ld, int dtype,
                                  storm_stream_t s) {
    STORM_CHECK(scores && probs && rows > 0 && L > 0 && ld >= L, "storm_softmax_rows: bad arguments");
    const int nb = cdiv(rows, 4);
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, scores, (bf16_t*)probs, rows, L, ld);
    else if (dtype == STORM_F16) hipLaunchKernelGGL((softmax_rows_kernel<half_t>), dim3(nb), dim3(256), 0, st, scores, (half_t*)probs, rows, L, ld);
    else if (dtype == STORM_F32) hipLaunchKernelGGL((softmax_rows_kernel<float>), dim3(nb), dim3(256), 0, st, scores, (float*)probs, rows, L, ld);
    else STORM_CHECK(false, "storm_softmax_rows: dtype %d", dtype);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_output_head(const void* pyr, const float* t, const float* W, const float* bias, int cin,
                                 float* out_cplx, int B, int F, int T, int negate, int dtype, storm_stream_t s) {
    STORM_CHECK(pyr && W && bias && out_cplx && cin >= 1 && cin <= 8, "storm_output_head: bad arguments (cin=%d)", cin);
    const long long per_b = (long long)F * T, npix = per_b * B;
    const int nb = cdiv(npix, 256);
    const float sign = negate ? -1.0f : 1.0f;
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) hipLaunchKernelGGL((output_head_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, (const bf16_t*)pyr, t, W, bias, cin, out_cplx, per_b, npix, sign);
    else if (dtype == STORM_F16) hipLaunchKernelGGL((output_head_kernel<half_t>), dim3(nb), dim3(256), 0, st, (const half_t*)pyr, t, W, bias, cin, out_cplx, per_b, npix, sign);
    else if (dtype == STORM_F32) hipLaunchKernelGGL((output_head_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)pyr, t, W, bias, cin, out_cplx, per_b, npix, sign);
    else STORM_CHECK(false, "storm_output_head: dtype %d", dtype);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

// ---- weight repacking ---------------------------------------------------------------------
namespace storm {
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int Cout, int Cin, int ntaps,
                                   int CoutP, int CinP, int transpose) {
    const long long total = (long long)ntaps * CoutP * CinP;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ci = (int)(i % CinP);
    const int co = (int)((i / CinP) % CoutP);
    const int tap = (int)(i / ((long long)CinP * CoutP));
    float v = 0.f;
    if (co < Cout && ci < Cin) {
        if (transpose) v = src[(long long)ci * Cout + co];                       // [Cin][Cout] (NIN.W)
        else v = src[((long long)co * Cin + ci) * ntaps + tap];                  // [Cout][Cin][taps]
    }
    from_f32(dst[i], v);
}
}  // namespace storm

static int pack_weight(const float* src, void* dst, int Cout, int Cin, int ntaps, int CoutP, int CinP,
                       int transpose, int dtype, storm_stream_t s) {
    STORM_CHECK(src && dst && Cout > 0 && Cin > 0 && ntaps > 0 && CoutP >= Cout && CinP >= Cin, "storm_pack_*: bad arguments");
    const long long total = (long long)ntaps * CoutP * CinP;
    const int nb = cdiv(total, 256);
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), dim3(nb), dim3(256), 0, st, src, (bf16_t*)dst, Cout, Cin, ntaps, CoutP, CinP, transpose);
    else if (dtype == STORM_F16) hipLaunchKernelGGL((pack_weight_kernel<half_t>), dim3(nb), dim3(256), 0, st, src, (half_t*)dst, Cout, Cin, ntaps, CoutP, CinP, transpose);
    else if (dtype == STORM_F32) hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(nb), dim3(256), 0, st, src, (float*)dst, Cout, Cin, ntaps, CoutP, CinP, transpose);
    else STORM_CHECK(false, "storm_pack_*: dtype %d", dtype);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_pack_conv_weight(const float* src, void* dst, int Cout, int Cin, int ntaps, int CoutP, int CinP,
                                      int dtype, storm_stream_t s) {
    return pack_weight(src, dst, Cout, Cin, ntaps, CoutP, CinP, 0, dtype, s);
}
extern "C" int storm_pack_matrix(const float* src, void* dst, int Cout, int Cin, int transpose, int CoutP, int CinP,
                                 int dtype, storm_stream_t s) {
    return pack_weight(src, dst, Cout, Cin, 1, CoutP, CinP, transpose, dtype, s);
}
