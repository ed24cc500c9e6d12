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
__global__ void time_embedding_kernel(const float* __restrict__ t, const float* __restrict__ gW,
                                      const float* __restrict__ W1, const float* __restrict__ b1,
                                      const float* __restrict__ W2, const float* __restrict__ b2,
                                      float* __restrict__ out, int nf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* emb = reinterpret_cast<float*>(smem);        // [2 nf]
    float* h1 = emb + 2 * nf;                            // [4 nf]
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const float lt = (float)log((double)t[b]);           // correctly rounded logf
    for (int k = tid; k < nf; k += nt) {
        const float xp = ((lt * gW[k]) * 2.0f) * 3.14159274101257324f;   // layerspp.py:40 op order
        emb[k] = (float)sin((double)xp);
        emb[nf + k] = (float)cos((double)xp);
    }
    __syncthreads();
    const int E = 2 * nf, Hd = 4 * nf;
    for (int n = tid; n < Hd; n += nt) {
        float acc = 0.f;
        const float* w = W1 + (long long)n * E;
        for (int k = 0; k < E; ++k) acc = fmaf(w[k], emb[k], acc);
        h1[n] = silu_f(acc + b1[n]);
    }
    __syncthreads();
    for (int n = tid; n < Hd; n += nt) {
        float acc = 0.f;
        const float* w = W2 + (long long)n * Hd;
        for (int k = 0; k < Hd; ++k) acc = fmaf(w[k], h1[k], acc);
        out[(long long)b * Hd + n] = silu_f(acc + b2[n]);   // blocks consume SiLU(temb)
    }
}

// ---- dense: out[b][n] = W[n][:] . x[b][:] + bias[n]; one wave per output row n ------------
__global__ void dense_kernel(const float* __restrict__ x, const float* __restrict__ W,
                             const float* __restrict__ bias, float* __restrict__ out, int B, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (long long)n * K;
    for (int b = 0; b < B; ++b) {
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) acc = fmaf(w[k], x[(long long)b * K + k], acc);
        acc = wave_sum(acc);
        if (lane == 0) out[(long long)b * N + n] = acc + bias[n];
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
    hipLaunchKernelGGL(time_embedding_kernel, dim3(B), dim3(256), lds, (hipStream_t)s, t, gfp_W, W1, b1, W2, b2, act_temb, nf);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_dense(const float* x, const float* W, const float* bias, float* out, int B, int N, int K,
                           storm_stream_t s) {
    STORM_CHECK(x && W && bias && out && B > 0 && N > 0 && K > 0, "storm_dense: bad arguments");
    hipLaunchKernelGGL(dense_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)s, x, W, bias, out, B, N, K);
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
