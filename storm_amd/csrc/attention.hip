// Fused single-head attention of AttnBlockpp (layerspp.py:75-91):
//   w = softmax_j(<q_i, k_j> * C^-1/2),  h_i = sum_j w_ij v_j      (the two einsums + F.softmax of :82-86)
// as ONE kernel with an online softmax: the [L][L] score matrix (268 MB fp32 at batch 16, L = 2048; 1.7 GB at
// 10-s utterances) is never written - round 1 materialised it (S GEMM -> row softmax -> P.V GEMM).
//
// CDNA4 layout ("swapped" products, so that everything per query is lane-local):
//   * a workgroup = 4 waves x 32 queries; key tiles of 32 keys, K [32][C] and V^T [C][32] staged in LDS (XOR-swizzled,
//     double buffered, the next tile's global loads in flight under the MFMAs: one barrier per tile).
//   * S^T = K Q^T per wave: A = K fragment (LDS), B = Q fragment (registers for the whole kernel), v_mfma_f32_32x32x16:
//     lane (query j = lane & 31, half h = lane >> 5) ends up with 16 of its query's 32 scores - a row max / sum is
//     15 VALU ops + one lane-pair exchange.
//   * O^T += V^T P^T: A = V^T fragment (LDS), B = P^T fragment = the lane's own probabilities packed in register order.
//     The contraction index of an MFMA k-group is summed over, so its 16 slots may carry the keys in ANY order as long
//     as both operands agree: slot (h, e) <-> key 16 m + 4 h + (e & 3) + 8 (e >> 2) is exactly the order the S^T
//     accumulator leaves them in, so P needs no shuffle; the V^T fragment is two 8-byte reads.
//   * the rescale factor of the online softmax is a per-lane scalar (one query per lane column of O^T).
// bf16 or fp16 operands (T), fp32 accumulation / softmax.  C = 32 * CT channels (CT in {1, 2, 4, 8}); any L (ragged tiles masked).
// fp32 (the parity path): attention_f32_kernel below, the same structure on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
#include <vector>
#include "conv_params.h"

namespace storm {

namespace attn { constexpr int BQ = 128, BK = 32, THREADS = 256; }
using attn::BQ; using attn::BK;
// (the kernel lives in namespace storm itself: the dynamic LDS array of every kernel is storm::smem)
constexpr int ATTN_THREADS = attn::THREADS;

__device__ __forceinline__ uint4 ldg16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

// SPLIT (round 5): blockIdx.z walks `nsplit` contiguous ranges of the key tiles; a workgroup leaves its UNNORMALISED accumulator (fp32) and
// its running (maximum, sum) per query in the caller's scratch, attention_combine_kernel below merges the ranges.  For calls whose query
// blocks do not fill the chip - one utterance is 16 workgroups, each a serial chain of 64 key tiles (154 us of a 2.8-ms evaluation) -
// the chain is what a launch lasts: splitting it 8 ways is 8 x the workgroups and an eighth of the chain.
template <typename T, int CT, bool SPLIT>
__device__ __forceinline__
void attention_body(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vT,
                    const float* __restrict__ bias, T* __restrict__ out, int L, int ldv, long long q_bs,
                    long long k_bs, long long v_bs, long long o_bs, float scale_log2e,
                    float* __restrict__ part_o, float* __restrict__ part_ml, int nsplit,
                    const int b, const int q0, const int zsplit, const int nbatch) {
    constexpr int C = 32 * CT, KG = C / 16;            // channels; 16-channel k-groups of the score product
    constexpr int KROW = C * 2, KSLOTS = KROW / 16;    // K tile row: bytes, 16-B slots
    constexpr int KTILE = BK * KROW, VTILE = C * 64;   // bytes per buffer
    constexpr int KPT = BK * KSLOTS / ATTN_THREADS > 0 ? BK * KSLOTS / ATTN_THREADS : 1;      // 16-B pieces per thread (K)
    constexpr int VPT = C * 4 / ATTN_THREADS > 0 ? C * 4 / ATTN_THREADS : 1;                  // 16-B pieces per thread (V^T)
    typedef typename Mma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const kbuf = smem;                            // [2][KTILE]
    char* const vbuf = smem + 2 * KTILE;                // [2][VTILE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const T* qb = q + (long long)b * q_bs;
    const T* kb = k + (long long)b * k_bs;
    const T* vb = vT + (long long)b * v_bs;

    // Q fragments of this lane's query (clamped: rows past L are computed and dropped)
    const int qi = min(q0 + wave * 32 + j, L - 1);
    Frag qf[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const uint4 v = ldg16(qb + (long long)qi * C + 16 * g + 8 * h);
        qf[g] = *reinterpret_cast<const Frag*>(&v);
    }

    f32x16 o[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- staging: global -> registers (issued a tile ahead) -> LDS ------------------------------------------------
    uint4 kreg[KPT], vreg[VPT];
    auto load_tile = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int u = tid + i * ATTN_THREADS;
            if (u < BK * KSLOTS) {
                const int r = u / KSLOTS, s = u % KSLOTS;
                kreg[i] = ldg16(kb + (long long)min(j0 + r, L - 1) * C + 8 * s);
            }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int u = tid + i * ATTN_THREADS;
            if (u < C * 4) {
                const int c = u >> 2, p = u & 3;
                vreg[i] = (j0 + 8 * p + 8 <= ldv) ? ldg16(vb + (long long)c * ldv + j0 + 8 * p) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int u = tid + i * ATTN_THREADS;
            if (u < BK * KSLOTS) {
                const int r = u / KSLOTS, s = u % KSLOTS;
                *reinterpret_cast<uint4*>(kbuf + buf * KTILE + r * KROW + ((s ^ (r & (KSLOTS - 1))) << 4)) = kreg[i];
            }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int u = tid + i * ATTN_THREADS;
            if (u < C * 4) {
                const int c = u >> 2, p = u & 3, x = (c >> 2) & 7;
                char* row = vbuf + buf * VTILE + c * 64;
                *reinterpret_cast<uint2*>(row + (((2 * p) ^ x) << 3)) = make_uint2(vreg[i].x, vreg[i].y);
                *reinterpret_cast<uint2*>(row + (((2 * p + 1) ^ x) << 3)) = make_uint2(vreg[i].z, vreg[i].w);
            }
        }
    };

    const int ntiles_all = (L + BK - 1) / BK;
    const int n_lo = SPLIT ? zsplit * ntiles_all / nsplit : 0;
    const int ntiles = SPLIT ? (zsplit + 1) * ntiles_all / nsplit : ntiles_all;      // (one past this range's last tile)
    load_tile(n_lo * BK);
    store_tile(0);
    __syncthreads();
    for (int n = n_lo; n < ntiles; ++n) {
        const int buf = (n - n_lo) & 1, j0 = n * BK;
        if (n + 1 < ntiles) load_tile(j0 + BK);                      // in flight under this tile's MFMAs
        // ---- S^T = K Q^T (32 keys x 32 queries per wave) ------------------------------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const char* kt = kbuf + buf * KTILE + j * KROW;              // A fragment: key j of the tile (lane & 31)
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const Frag kf = *reinterpret_cast<const Frag*>(kt + (((2 * g + h) ^ (j & (KSLOTS - 1))) << 4));
            Mma<T>::run(kf, qf[g], s);
        }
        // ---- online softmax over this lane's 16 keys + its partner's 16 ------------------------------------------------
        float mx = -INFINITY;
        if (j0 + BK <= L) {                               // (uniform) a full key tile: no masking
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] *= scale_log2e; mx = fmaxf(mx, s[r]); }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = j0 + cidx::acc_row(lane, r);
                s[r] = key < L ? s[r] * scale_log2e : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = hw_exp2(m_run - m_new);                    // (first tile: exp2(-inf) = 0 on a zero accumulator)
        float ps = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = hw_exp2(s[r] - m_new); ps += p[r]; }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (wave_any(alpha != 1.0f)) {               // after the first tiles the running maxima rarely move: 16 CT multiplications saved
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
        // ---- O^T += V^T P^T: two k-groups of 16 keys, slot (h, e) <-> key 16 m + 4 h + (e & 3) + 8 (e >> 2) ---------------
#pragma unroll
        for (int mk = 0; mk < 2; ++mk) {
            uint32_t pw[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pw[e] = pack2(p[8 * mk + 2 * e], p[8 * mk + 2 * e + 1], (T*)nullptr);
            const uint4 pv = make_uint4(pw[0], pw[1], pw[2], pw[3]);
            const Frag pf = *reinterpret_cast<const Frag*>(&pv);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int c = 32 * t + j, x = (c >> 2) & 7;
                const char* row = vbuf + buf * VTILE + c * 64;
                const uint2 lo = *reinterpret_cast<const uint2*>(row + (((4 * mk + h) ^ x) << 3));
                const uint2 hi = *reinterpret_cast<const uint2*>(row + (((4 * mk + h + 2) ^ x) << 3));
                const uint4 vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
                Mma<T>::run(*reinterpret_cast<const Frag*>(&vv), pf, o[t]);
            }
        }
        if (n + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    const int qrow = q0 + wave * 32 + j;
    if (SPLIT) {                                     // this key range's partial result: O (unnormalised), running maximum (log2 domain), sum
        if (qrow < L) {
            const long long prow = ((long long)zsplit * nbatch + b) * L + qrow;
            float* po = part_o + prow * C;
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(po + 32 * t + 8 * g + 4 * h) = make_float4(o[t][4 * g], o[t][4 * g + 1], o[t][4 * g + 2], o[t][4 * g + 3]);
            if (h == 0) *reinterpret_cast<float2*>(part_ml + prow * 2) = make_float2(m_run, l_run);     // (both lane halves hold the same pair)
        }
        return;
    }
    // ---- epilogue: h = O / l + b_v (rows of P sum to one, so the NIN_2 bias passes through), 16-bit, 8-byte stores ----------
    if (qrow < L) {
        const float inv = 1.0f / l_run;
        T* orow = out + (long long)b * o_bs + (long long)qrow * C;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * t + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[t][4 * g + e] * inv + (bias ? bias[c + e] : 0.f);
                *reinterpret_cast<uint2*>(orow + c) = make_uint2(pack2(v[0], v[1], (T*)nullptr), pack2(v[2], v[3], (T*)nullptr));
            }
    }
}

template <typename T, int CT, bool SPLIT>
__global__ __launch_bounds__(ATTN_THREADS)
void attention_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ vT,
                      const float* __restrict__ bias, T* __restrict__ out, int L, int ldv, long long q_bs,
                      long long k_bs, long long v_bs, long long o_bs, float scale_log2e,
                      float* __restrict__ part_o, float* __restrict__ part_ml, int nsplit) {
    attention_body<T, CT, SPLIT>(q, k, vT, bias, out, L, ldv, q_bs, k_bs, v_bs, o_bs, scale_log2e, part_o, part_ml, nsplit,
                                 (int)blockIdx.y, (int)blockIdx.x * BQ, (int)blockIdx.z, (int)gridDim.y);
}
// The query blocks of SEVERAL problems (ragged micro-batches of one stream: different L) in one launch, never split (the group fills the chip):
// item = (problem, batch item, query block) from a host-built list, the problem's tensors from a device table.  A query block computes what
// the unsplit kernel computes for it.
template <typename T, int CT>
__global__ __launch_bounds__(ATTN_THREADS)
void attention_group_kernel(const AttnProblem* __restrict__ tab, const AttnItem* __restrict__ items, const float* __restrict__ bias, float scale_log2e) {
    const AttnItem it = items[blockIdx.x];
    const AttnProblem& p = tab[it.problem];
    attention_body<T, CT, false>(static_cast<const T*>(p.q), static_cast<const T*>(p.k), static_cast<const T*>(p.vT), bias, static_cast<T*>(p.out), p.L, p.ldv,
                                 p.q_bs, p.k_bs, p.v_bs, p.o_bs, scale_log2e, nullptr, nullptr, 1, it.b, it.qblock * BQ, 0, 1);
}

// Merge of the key ranges: out = sum_s O_s 2^(m_s - m) / sum_s l_s 2^(m_s - m) + bias, m = max_s m_s; ranges in index order (fixed: bit-reproducible).
// One thread per (query, four channels).
template <typename T>
__global__ __launch_bounds__(256)
void attention_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, const float* __restrict__ bias,
                              T* __restrict__ out, int S, int B, int L, int C, long long o_bs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = C / 4;
    const long long row = i / c4;
    if (row >= (long long)B * L) return;
    const int c = (int)(i - row * c4) * 4;
    const long long rows = (long long)B * L;
    float m = -INFINITY;
    for (int sp = 0; sp < S; ++sp) m = fmaxf(m, part_ml[(sp * rows + row) * 2]);
    float l = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int sp = 0; sp < S; ++sp) {
        const float2 ml = *reinterpret_cast<const float2*>(part_ml + (sp * rows + row) * 2);
        const float w = hw_exp2(ml.x - m);
        const float4 v = *reinterpret_cast<const float4*>(part_o + (sp * rows + row) * C + c);
        l = fmaf(ml.y, w, l);
        a0 = fmaf(v.x, w, a0); a1 = fmaf(v.y, w, a1); a2 = fmaf(v.z, w, a2); a3 = fmaf(v.w, w, a3);
    }
    const float inv = 1.0f / l;
    const int b = (int)(row / L);
    const long long qrow = row - (long long)b * L;
    float v[4] = {a0 * inv, a1 * inv, a2 * inv, a3 * inv};
    if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bias[c + e];
    }
    *reinterpret_cast<uint2*>(out + (long long)b * o_bs + qrow * C + c) = make_uint2(pack2(v[0], v[1], (T*)nullptr), pack2(v[2], v[3], (T*)nullptr));
}

// Key ranges for a call (1 = none): only where the query blocks leave most CUs idle (fewer than 128 workgroups), doubling while the split
// launch still fits the chip and a range keeps at least four key tiles; never for the fp32 parity path.  Like conv_splitk_slices' small-call
// rule a decision on the CALL: across its threshold a row agrees to the rounding of its 16-bit output, not bit for bit.
int attn_splits(int B, int L, int dtype) {
    if (dtype == STORM_F32) return 1;
    const int ntiles = cdiv(L, BK);
    const int forced = switches().attn_split;                // (tests / A-B: 1 = never, 2 / 4 / 8 = that many where the tiles allow)
    if (forced == 1) return 1;
    if (forced >= 2) return forced <= ntiles ? forced : 1;
    if (switches().batch_invariant != 0) return 1;             // (the rule below is a decision on the call: not in the per-image mode)
    const long long wgs = (long long)cdiv(L, BQ) * B;
    const int cus = device_cus();                              // (256 on MI355X; honours STORM_CONV_CUS like the convolutions' ladder)
    if (wgs >= cus / 2) return 1;
    int S = 1;
    while (S < 8 && wgs * S * 2 <= cus && ntiles / (S * 2) >= 4) S *= 2;
    return S;
}
long long attn_scratch_bytes(int B, int L, int C, int S) { return S < 2 ? 0 : (long long)S * B * L * (C + 2) * 4; }

// ---- fp32 (parity path): the same online-softmax structure on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 sums).
// Key tiles of 32 keys, K [32][C] and V^T [C][32] fp32 in LDS (2 x 32 KB at C = 256, double buffered = 128 KB), Q fragments of the
// lane's query in registers (C / 2 floats: one 512-register wave per SIMD).  An MFMA contracts over two k-slots (the lane
// halves): for S^T slot h <-> channel 2 g + h; for O^T slot h of MFMA r <-> key acc_row(h, r), i.e. the B operand of MFMA r is
// the lane's own p[r] and the A operand one V^T element.
template <int CT>
__global__ __launch_bounds__(ATTN_THREADS, 1)
void attention_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ vT,
                          const float* __restrict__ bias, float* __restrict__ out, int L, int ldv, long long q_bs,
                          long long k_bs, long long v_bs, long long o_bs, float scale_log2e) {
    constexpr int C = 32 * CT;
    constexpr int KROW = C * 4 + 16;                    // padded rows: the 32 lanes of a fragment read hit 32 banks
    constexpr int KTILE = BK * KROW, VROW = BK * 4 + 16, VTILE = C * VROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const kbuf = smem;
    char* const vbuf = smem + 2 * KTILE;
    const int b = blockIdx.y, q0 = blockIdx.x * BQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float* qb = q + (long long)b * q_bs;
    const float* kb = k + (long long)b * k_bs;
    const float* vb = vT + (long long)b * v_bs;
    const int qi = min(q0 + wave * 32 + j, L - 1);
    float qf[C / 2];                                    // channel 2 g + h of this lane's query
#pragma unroll
    for (int g = 0; g < C / 2; ++g) qf[g] = qb[(long long)qi * C + 2 * g + h];
    f32x16 o[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    auto stage = [&](int j0, int buf) {                 // synchronous staging (this path is for parity, not speed)
        for (int u = tid; u < BK * (C / 4); u += ATTN_THREADS) {
            const int r = u / (C / 4), s4 = u % (C / 4);
            *reinterpret_cast<float4*>(kbuf + buf * KTILE + r * KROW + s4 * 16) =
                *reinterpret_cast<const float4*>(kb + (long long)min(j0 + r, L - 1) * C + 4 * s4);
        }
        for (int u = tid; u < C * (BK / 4); u += ATTN_THREADS) {
            const int c = u / (BK / 4), p4 = u % (BK / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j0 + 4 * p4 + 4 <= ldv) v = *reinterpret_cast<const float4*>(vb + (long long)c * ldv + j0 + 4 * p4);
            *reinterpret_cast<float4*>(vbuf + buf * VTILE + c * VROW + p4 * 16) = v;
        }
    };
    const int ntiles = (L + BK - 1) / BK;
    stage(0, 0);
    __syncthreads();
    for (int n = 0; n < ntiles; ++n) {
        const int buf = n & 1, j0 = n * BK;
        if (n + 1 < ntiles) stage(j0 + BK, buf ^ 1);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kt = reinterpret_cast<const float*>(kbuf + buf * KTILE + j * KROW);
#pragma unroll
        for (int g = 0; g < C / 2; ++g) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[2 * g + h], qf[g], s, 0, 0, 0);
        float mx = -INFINITY;
        if (j0 + BK <= L) {                               // (uniform) a full key tile: no masking
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] *= scale_log2e; mx = fmaxf(mx, s[r]); }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = j0 + cidx::acc_row(lane, r);
                s[r] = key < L ? s[r] * scale_log2e : -INFINITY;
                mx = fmaxf(mx, s[r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = hw_exp2(m_run - m_new);
        float ps = 0.f, p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = hw_exp2(s[r] - m_new); ps += p[r]; }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = m_new;
        if (wave_any(alpha != 1.0f)) {               // after the first tiles the running maxima rarely move: 16 CT multiplications saved
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {                  // MFMA r contracts over the two keys acc_row(h = 0, r), acc_row(h = 1, r)
            const int key = cidx::acc_row(lane, r);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const float v = *reinterpret_cast<const float*>(vbuf + buf * VTILE + (32 * t + j) * VROW + key * 4);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v, p[r], o[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    const int qrow = q0 + wave * 32 + j;
    if (qrow < L) {
        const float inv = 1.0f / l_run;
        float* orow = out + (long long)b * o_bs + (long long)qrow * C;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * t + 8 * g + 4 * h;
                float4 v;
                v.x = o[t][4 * g] * inv + (bias ? bias[c] : 0.f); v.y = o[t][4 * g + 1] * inv + (bias ? bias[c + 1] : 0.f);
                v.z = o[t][4 * g + 2] * inv + (bias ? bias[c + 2] : 0.f); v.w = o[t][4 * g + 3] * inv + (bias ? bias[c + 3] : 0.f);
                *reinterpret_cast<float4*>(orow + c) = v;
            }
    }
}

template <typename T, int CT>
static int attn_launch(const void* q, const void* k, const void* vT, const float* bias, void* out, int B, int L, int ldv,
                  long long q_bs, long long k_bs, long long v_bs, long long o_bs, float scale, void* scratch, long long scratch_bytes, hipStream_t st) {
    constexpr int C = 32 * CT;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int lds = F32 ? 2 * (BK * (C * 4 + 16)) + 2 * (C * (BK * 4 + 16)) : 2 * (BK * C * 2) + 2 * (C * 64);
    static bool attr_set = false;
    if constexpr (F32) {
        auto kern = attention_f32_kernel<CT>;
        if (!attr_set) {
            STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(cdiv(L, BQ), B), dim3(ATTN_THREADS), lds, st, (const float*)q, (const float*)k, (const float*)vT,
                           bias, (float*)out, L, ldv, q_bs, k_bs, v_bs, o_bs, scale * 1.44269504088896341f);
    } else {
        auto kern = attention_kernel<T, CT, false>;
        auto kern_s = attention_kernel<T, CT, true>;
        if (!attr_set) {
            STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern_s), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr_set = true;
        }
        const int S = scratch != nullptr ? attn_splits(B, L, sizeof(T) == 4 ? STORM_F32 : STORM_BF16) : 1;
        if (S >= 2 && scratch_bytes >= attn_scratch_bytes(B, L, C, S)) {
            float* const part_o = static_cast<float*>(scratch);
            float* const part_ml = part_o + (long long)S * B * L * C;
            hipLaunchKernelGGL(kern_s, dim3(cdiv(L, BQ), B, S), dim3(ATTN_THREADS), lds, st, (const T*)q, (const T*)k, (const T*)vT,
                               bias, (T*)out, L, ldv, q_bs, k_bs, v_bs, o_bs, scale * 1.44269504088896341f, part_o, part_ml, S);
            STORM_LAUNCH_CHECK();
            const long long nthreads = (long long)B * L * (C / 4);
            hipLaunchKernelGGL(attention_combine_kernel<T>, dim3((unsigned)cdiv(nthreads, 256LL)), dim3(256), 0, st, part_o, part_ml, bias, (T*)out,
                               S, B, L, C, o_bs);
        } else {
            hipLaunchKernelGGL(kern, dim3(cdiv(L, BQ), B), dim3(ATTN_THREADS), lds, st, (const T*)q, (const T*)k, (const T*)vT,
                               bias, (T*)out, L, ldv, q_bs, k_bs, v_bs, o_bs, scale * 1.44269504088896341f, (float*)nullptr, (float*)nullptr, 1);
        }
    }
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

template <typename T>
static int attn_dispatch(int C, const void* q, const void* k, const void* vT, const float* bias, void* out, int B, int L, int ldv,
                         long long q_bs, long long k_bs, long long v_bs, long long o_bs, float scale, void* scratch, long long scratch_bytes,
                         hipStream_t st) {
    switch (C) {
        case 32: return attn_launch<T, 1>(q, k, vT, bias, out, B, L, ldv, q_bs, k_bs, v_bs, o_bs, scale, scratch, scratch_bytes, st);
        case 64: return attn_launch<T, 2>(q, k, vT, bias, out, B, L, ldv, q_bs, k_bs, v_bs, o_bs, scale, scratch, scratch_bytes, st);
        case 128: return attn_launch<T, 4>(q, k, vT, bias, out, B, L, ldv, q_bs, k_bs, v_bs, o_bs, scale, scratch, scratch_bytes, st);
        default: return attn_launch<T, 8>(q, k, vT, bias, out, B, L, ldv, q_bs, k_bs, v_bs, o_bs, scale, scratch, scratch_bytes, st);
    }
}

// grouped launch (common.h): 16-bit operands only (the fp32 parity kernel runs problem by problem)
template <typename T, int CT>
static int attn_group_launch(const AttnProblem* dev_tab, const AttnItem* dev_items, int n_items, const float* bias, float scale, hipStream_t st) {
    constexpr int C = 32 * CT;
    constexpr int lds = 2 * (BK * C * 2) + 2 * (C * 64);
    auto kern = attention_group_kernel<T, CT>;
    static bool attr_set = false;
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_items), dim3(ATTN_THREADS), lds, st, dev_tab, dev_items, bias, scale * 1.44269504088896341f);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}
int attn_query_blocks(int L) { return cdiv(L, BQ); }
int launch_attention_group(const AttnProblem* dev_tab, const AttnItem* dev_items, int n_items, const float* bias, int C, float scale, int dtype, hipStream_t st) {
    STORM_CHECK(dev_tab && dev_items && n_items > 0 && (dtype == STORM_BF16 || dtype == STORM_F16) && (C == 32 || C == 64 || C == 128 || C == 256),
                "storm_attention (group): bad arguments");
#define STORM_AG(T_) (C == 32 ? attn_group_launch<T_, 1>(dev_tab, dev_items, n_items, bias, scale, st) : C == 64 ? attn_group_launch<T_, 2>(dev_tab, dev_items, n_items, bias, scale, st) : \
                      C == 128 ? attn_group_launch<T_, 4>(dev_tab, dev_items, n_items, bias, scale, st) : attn_group_launch<T_, 8>(dev_tab, dev_items, n_items, bias, scale, st))
    return dtype == STORM_F16 ? STORM_AG(half_t) : STORM_AG(bf16_t);
#undef STORM_AG
}

}  // namespace storm

extern "C" int storm_attention_supported(int C, int dtype) {
    return (dtype == STORM_BF16 || dtype == STORM_F16 || dtype == STORM_F32) && (C == 32 || C == 64 || C == 128 || C == 256);
}

// scratch for the key-range split of a call (0 = this call runs unsplit): fp32 [S][B][L][C] partial outputs + [S][B][L][2] (maximum, sum)
extern "C" long long storm_attention_scratch_bytes(int B, int L, int C, int dtype) {
    if (B <= 0 || L <= 0 || !storm_attention_supported(C, dtype)) return 0;
    return storm::attn_scratch_bytes(B, L, C, storm::attn_splits(B, L, dtype));
}

extern "C" int storm_attention_ws(const void* q, const void* k, const void* vT, const float* bias, void* out, int B, int L, int C,
                                  int ldv, long long q_bstride, long long k_bstride, long long vT_bstride, long long out_bstride,
                                  float scale, int dtype, void* scratch, long long scratch_bytes, storm_stream_t s) {
    using namespace storm;
    STORM_CHECK(q && k && vT && out && B > 0 && L > 0, "storm_attention: bad arguments");
    STORM_CHECK(ldv >= L && ldv % 8 == 0, "storm_attention: ldv=%d (L=%d)", ldv, L);
    if (!storm_attention_supported(C, dtype)) {
        set_error("storm_attention: C=%d dtype=%d is outside the fused kernel (use the GEMM + softmax path)", C, dtype);
        return STORM_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_F32) return attn_dispatch<float>(C, q, k, vT, bias, out, B, L, ldv, q_bstride, k_bstride, vT_bstride, out_bstride, scale, nullptr, 0, st);
    if (dtype == STORM_F16) return attn_dispatch<half_t>(C, q, k, vT, bias, out, B, L, ldv, q_bstride, k_bstride, vT_bstride, out_bstride, scale, scratch, scratch_bytes, st);
    return attn_dispatch<bf16_t>(C, q, k, vT, bias, out, B, L, ldv, q_bstride, k_bstride, vT_bstride, out_bstride, scale, scratch, scratch_bytes, st);
}

extern "C" int storm_attention(const void* q, const void* k, const void* vT, const float* bias, void* out, int B, int L, int C,
                               int ldv, long long q_bstride, long long k_bstride, long long vT_bstride, long long out_bstride,
                               float scale, int dtype, storm_stream_t s) {
    return storm_attention_ws(q, k, vT, bias, out, B, L, C, ldv, q_bstride, k_bstride, vT_bstride, out_bstride, scale, dtype, nullptr, 0, s);
}

// One launch for the fused attention of P problems (ragged micro-batches of a stream: different B and L, one layer): 16-bit operands, never
// split.  blob: device scratch >= storm_attention_group_blob_bytes; this convenience entry fills it with a synchronous copy (the whole-network
// object keeps a pinned image).  q / k: [B_p][L_p][C], vT: [B_p][C][ldv_p], out: [B_p][L_p][C] (contiguous per problem).
extern "C" long long storm_attention_group_blob_bytes(const int* B, const int* L, int P) {
    if (B == nullptr || L == nullptr || P < 1) return -1;
    long long items = 0;
    for (int g = 0; g < P; ++g) items += (long long)B[g] * storm::attn_query_blocks(L[g]);
    return ((long long)P * (long long)sizeof(storm::AttnProblem) + 255) / 256 * 256 + items * (long long)sizeof(storm::AttnItem);
}
extern "C" int storm_attention_group(const void* const* q, const void* const* k, const void* const* vT, void* const* out, const int* B, const int* L,
                                     const int* ldv, int P, const float* bias, int C, float scale, int dtype, void* blob, long long blob_bytes,
                                     storm_stream_t s) {
    using namespace storm;
    STORM_CHECK(q && k && vT && out && B && L && ldv && P >= 1 && blob, "storm_attention_group: bad arguments");
    if ((dtype != STORM_BF16 && dtype != STORM_F16) || !storm_attention_supported(C, dtype)) {
        set_error("storm_attention_group: C=%d dtype=%d is outside the grouped kernel (16-bit operands, C in {32, 64, 128, 256})", C, dtype);
        return STORM_ERR_UNSUPPORTED;
    }
    const long long need = storm_attention_group_blob_bytes(B, L, P);
    STORM_CHECK(blob_bytes >= need, "storm_attention_group: blob %lld < %lld bytes", blob_bytes, need);
    const long long tab = ((long long)P * (long long)sizeof(AttnProblem) + 255) / 256 * 256;
    std::vector<char> host((size_t)need);
    AttnProblem* t = reinterpret_cast<AttnProblem*>(host.data());
    AttnItem* it = reinterpret_cast<AttnItem*>(host.data() + tab);
    long long ni = 0;
    for (int g = 0; g < P; ++g) {
        STORM_CHECK(q[g] && k[g] && vT[g] && out[g] && B[g] > 0 && L[g] > 0 && ldv[g] >= L[g] && ldv[g] % 8 == 0, "storm_attention_group: problem %d", g);
        memset(&t[g], 0, sizeof(AttnProblem));
        t[g].q = q[g]; t[g].k = k[g]; t[g].vT = vT[g]; t[g].out = out[g]; t[g].L = L[g]; t[g].ldv = ldv[g];
        t[g].q_bs = (long long)L[g] * C; t[g].k_bs = (long long)L[g] * C; t[g].v_bs = (long long)C * ldv[g]; t[g].o_bs = (long long)L[g] * C;
        for (int b = 0; b < B[g]; ++b)
            for (int qb = 0; qb < attn_query_blocks(L[g]); ++qb) { AttnItem& a = it[ni++]; a.problem = g; a.b = b; a.qblock = qb; a.pad_ = 0; }
    }
    STORM_HIP(hipMemcpyAsync(blob, host.data(), (size_t)need, hipMemcpyHostToDevice, (hipStream_t)s));
#ifndef STORM_HOST_SIM
    STORM_HIP(hipStreamSynchronize((hipStream_t)s));
#endif
    return launch_attention_group(reinterpret_cast<const AttnProblem*>(blob), reinterpret_cast<const AttnItem*>(static_cast<char*>(blob) + tab), (int)ni, bias, C, scale,
                                  dtype, (hipStream_t)s);
}
