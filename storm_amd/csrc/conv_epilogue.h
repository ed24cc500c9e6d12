// The epilogue shared by the chunk-descriptor convolution kernels (conv_pipe.hip, conv_pipe128.hip, conv_duo.hip): one wave's
// WM x WN accumulator tiles (64 couts x WN pixel rows of 32) ->
//   LDS transpose (private staging, 32 rows x WM x 128 B per pass) -> (acc + bias + temb bias + skip) * scale -> 16-byte stores,
// plus the fused GroupNorm statistics of the stored values (layerspp.py:242-274: what follows every convolution of a
// ResnetBlockBigGANpp is a GroupNorm over its output).  What the three kernels' private copies had grown into, kept in ONE place:
//   * a pass is ONE pixel row of the tile: its validity and element offset are scalar, the lane adds its own (pixel, cout octet)
//     offset; the staged row's swizzle has two variants (iteration even / odd) - `lane` is opaque to the compiler in these
//     kernels, so generic row / column arithmetic cost 30 VALU instructions per store;
//   * skip operands are fetched ONE PASS ahead, each into the register its predecessor (same iteration, previous pass) has just
//     left - fetched where they are used, every one of the 16 loads per tile and wave exposed its full memory latency;
//   * three instantiations of the store loop picked by uniform branches: a tile inside the image with all of its couts valid and a
//     16-bit output (no per-lane masks, no row / column tests, base + 32-bit-offset stores) with / without a skip operand, and the
//     general one.
#pragma once
#include "conv_pipe_common.h"

namespace storm { namespace epi {
using namespace cidx;
using pipe::IC;

struct TileAt { int tile, b, ty0, tx0, cout0; };          // the tile being stored (the kernel is already issuing the next one's loads)

// ABL: the profiling instantiations of conv_pipe.hip (2048 no global stores / skip loads, 4096 non-temporal stores, 8192 arithmetic
// only); 0 everywhere else.
template <typename T, int WM, int WN, int ABL = 0, typename AP>
__device__ __forceinline__ void store_tile(const f32x16 (&acc)[WM][WN], char* const stage, AP ap, const TileAt& t, const int wm, const int wn,
                                           const int lane, const int imgH, const int imgW, const int BN, const int TH,
                                           float (&gsum)[8], float (&gsq)[8]) {
    constexpr int SROWS = 32;
    constexpr int LPR = WM * 4;                 // lanes per staged row (8 couts each)
    constexpr int RPI = 64 / LPR;               // rows per read iteration
    static_assert(LPR == 8 && RPI == 8 && WM == 2, "store loop index math");
    // epilogue parameters: read ONCE per tile and pinned in SGPRs (through the kernarg pointer the compiler re-loaded the
    // output pointer / strides inside the store loop: a scalar load + s_waitcnt lgkmcnt(0) and a 64-bit multiply per store)
    const int outC = pin(ap->outC), out_f32 = pin(ap->out_f32);
    const bool has_skip = ap->skip != nullptr;
    char* const out_b = as_global(reinterpret_cast<unsigned long long>(ap->out) +
                                  (unsigned long long)((long long)t.b * ap->out_bstride * (out_f32 ? 4 : (int)sizeof(T))));
    const int c8 = lane & (LPR - 1);
    const int co = t.cout0 + wm * WM * 32 + c8 * 8;
    float badd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) badd[e] = 0.f;
    if (co + 8 <= ap->Cout) {
        if (ap->bias) { float bb[8]; load8(ap->bias + co, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) badd[e] += bb[e]; }
        if (ap->tbias) { float bb[8]; load8(ap->tbias + (long long)t.b * ap->tbias_stride + co, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) badd[e] += bb[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (co + e < ap->Cout) {
                if (ap->bias) badd[e] += ap->bias[co + e];
                if (ap->tbias) badd[e] += ap->tbias[(long long)t.b * ap->tbias_stride + co + e];
            }
    }
    const bool co_ok = co < outC;
    const T* const skip_b = reinterpret_cast<const T*>(ap->skip) + (long long)t.b * ap->skip_bstride;
    // out = (acc + bias + temb bias + skip) * scale as packed fma: (acc [+ skip]) * scale + (bias * scale); channel pairs stay in
    // adjacent registers from the staging read to the 16-bit pack (v_pk_fma_f32 / v_pk_add_f32)
    f32x2 badd2[4], gsum2[4], gsq2[4];
    const f32x2 scale2 = {ap->scale, ap->scale};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        badd2[i] = f32x2{badd[2 * i] * ap->scale, badd[2 * i + 1] * ap->scale};
        gsum2[i] = f32x2{0.f, 0.f}; gsq2[i] = f32x2{0.f, 0.f};
    }
    const int l8 = (lane >> 3) & 7;                         // the pixel (of the 8 per iteration) this lane stores
    const int gx0 = t.tx0 + l8;
    const uint32_t o_lane = (uint32_t)(gx0 * outC + co);
    int srow[2][2];                                         // staged row it * 8 + l8, slots 2 c8 / 2 c8 + 1: stage_off<WM> without its `it * 8` rows
#pragma unroll
    for (int odd = 0; odd < 2; ++odd)
#pragma unroll
        for (int h = 0; h < 2; ++h) srow[odd][h] = stage_off<WM>(odd * 8 + l8, 2 * c8 + h) - odd * 8 * (WM * 128);
    auto run_passes = [&](auto fast_, auto skipk_) {
        constexpr bool FAST = decltype(fast_)::value;
        constexpr int SKIPK = decltype(skipk_)::value;          // 1: skip operand, 0: none, -1: run-time
        const bool skip_on = (SKIPK < 0 ? has_skip : SKIPK == 1) && !(ABL & 2048);
        uint4 skq[SROWS / RPI];
        auto skip_fetch = [&](int pass, int it) {
            const int gy = t.ty0 + wn * WN + pass;
            const uint32_t o_row = (uint32_t)(gy * imgW) * (uint32_t)outC;
            const uint32_t o = o_row + (uint32_t)(it * RPI) * (uint32_t)outC + o_lane;
            if (FAST) skq[it] = ld16(skip_b, o * (uint32_t)sizeof(T));
            else skq[it] = (gy < imgH && gx0 + it * RPI < imgW && co_ok) ? *reinterpret_cast<const uint4*>(skip_b + o) : make_uint4(0u, 0u, 0u, 0u);
        };
        if (skip_on) {
#pragma unroll
            for (int it = 0; it < SROWS / RPI; ++it) skip_fetch(0, it);
        }
#pragma unroll
        for (int pass = 0; pass < WN; ++pass) {
            if (pass > 0) wave_sync();
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = lane & 31;
                    const f32x16& c = acc[mi][pass];
                    *reinterpret_cast<float4*>(stage + stage_off<WM>(row, stage_wslot(lane, mi, g))) =
                        make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                }
            wave_sync();
            const int gy = t.ty0 + wn * WN + pass;
            if (FAST || gy < imgH) {
                const uint32_t o_row = (uint32_t)(gy * imgW) * (uint32_t)outC;                    // (uniform)
#pragma unroll
                for (int it = 0; it < SROWS / RPI; ++it) {
                    const char* const sp = stage + it * RPI * (WM * 128);
                    const float4 v0 = *reinterpret_cast<const float4*>(sp + srow[it & 1][0]);
                    const float4 v1 = *reinterpret_cast<const float4*>(sp + srow[it & 1][1]);
                    f32x2 v2[4] = {f32x2{v0.x, v0.y}, f32x2{v0.z, v0.w}, f32x2{v1.x, v1.y}, f32x2{v1.z, v1.w}};
                    const uint4 skv = skq[it];
                    if (skip_on && pass + 1 < WN) skip_fetch(pass + 1, it);
                    if (FAST || (gx0 + it * RPI < imgW && co_ok && !(ABL & 2048))) {
                        const uint32_t o = o_row + (uint32_t)(it * RPI) * (uint32_t)outC + o_lane;
                        if (skip_on) {
                            alignas(16) T sk[8];
                            *reinterpret_cast<uint4*>(sk) = skv;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v2[i] += f32x2{to_f32(sk[2 * i]), to_f32(sk[2 * i + 1])};
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            v2[i] = __builtin_elementwise_fma(v2[i], scale2, badd2[i]);
                            gsum2[i] += v2[i];
                            gsq2[i] = __builtin_elementwise_fma(v2[i], v2[i], gsq2[i]);
                        }
                        const float v[8] = {v2[0].x, v2[0].y, v2[1].x, v2[1].y, v2[2].x, v2[2].y, v2[3].x, v2[3].y};
                        if (FAST) {
                            st16(out_b, o * (uint32_t)sizeof(T), make_uint4(pack2(v[0], v[1], (T*)nullptr), pack2(v[2], v[3], (T*)nullptr),
                                                                            pack2(v[4], v[5], (T*)nullptr), pack2(v[6], v[7], (T*)nullptr)));
                        }
                        else if (out_f32) store8(reinterpret_cast<float*>(out_b) + o, v);
                        else if (ABL & 4096) {                       // (profiling A/B: non-temporal output stores)
                            uint32_t w4[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) w4[i] = pack2(v[2 * i], v[2 * i + 1], (T*)nullptr);
                            typedef uint32_t u32x4_nt __attribute__((ext_vector_type(4)));
                            u32x4_nt val = {w4[0], w4[1], w4[2], w4[3]};
                            __builtin_nontemporal_store(val, reinterpret_cast<u32x4_nt*>(reinterpret_cast<T*>(out_b) + o));
                        }
                        else if (ABL & 8192) {                       // (profiling: all of the epilogue's arithmetic, no global store)
#pragma unroll
                            for (int e = 0; e < 8; ++e) keep(v[e]);
                            keep(o);
                        }
                        else store8(reinterpret_cast<T*>(out_b) + o, v);
                    }
                }
            }
        }
    };
    const bool interior = t.ty0 + TH <= imgH && t.tx0 + TILE_W <= imgW && t.cout0 + BN <= outC && !out_f32 &&
                          !(ABL & (2048 | 4096 | 8192));
    if (interior) {
        if (has_skip) run_passes(std::true_type{}, IC<1>{});
        else run_passes(std::true_type{}, IC<0>{});
    } else run_passes(std::false_type{}, IC<-1>{});
#pragma unroll
    for (int i = 0; i < 4; ++i) { gsum[2 * i] = gsum2[i].x; gsum[2 * i + 1] = gsum2[i].y; gsq[2 * i] = gsq2[i].x; gsq[2 * i + 1] = gsq2[i].y; }
}

// Fused GroupNorm statistics: the wave's per-channel (sum, sum of squares) of what it stored -> the per-tile partials
// [B][8-row tiles][outC][2] every conv kernel writes (storm_conv_tiles).  The workgroup's WAVES_N pixel-row groups are summed in
// a fixed order through `red` ([WAVES_N][BN][2] floats of LDS that the caller guarantees free); a workgroup tile of TH rows is
// TH / 8 partial tiles (wave rows wn * WN / 8).  Two workgroup barriers.
template <int WM, int WN, int WAVES_N, int BN, int TH, typename AP>
__device__ __forceinline__ void write_stats(float (&gsum)[8], float (&gsq)[8], float* const red, AP ap, const TileAt& t, const int wm, const int wn,
                                            const int lane, const int tid, const int imgH, const int tiles_x, const int tiles_per_img) {
    constexpr int LPR = WM * 4;
    constexpr int T8 = TH / TILE_H;                         // partial tiles per workgroup tile
    constexpr int WPT = WAVES_N / T8;                       // wave rows per partial tile
    static_assert(TH % TILE_H == 0 && WAVES_N % T8 == 0 && WPT * WN == TILE_H, "statistics tile layout");
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) { gsum[e] += __shfl_xor(gsum[e], off, 64); gsq[e] += __shfl_xor(gsq[e], off, 64); }
    __syncthreads();                                        // every wave's staging reads are done (red may overlap a staging area)
    if (lane < LPR) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int chl = wm * WM * 32 + lane * 8 + e;
            red[(wn * BN + chl) * 2] = gsum[e];
            red[(wn * BN + chl) * 2 + 1] = gsq[e];
        }
    }
    __syncthreads();
    const int part = tid / BN, ch = tid % BN;
    if (tid < T8 * BN && t.cout0 + ch < ap->outC) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int w = 0; w < WPT; ++w) { s0 += red[((part * WPT + w) * BN + ch) * 2]; s1 += red[((part * WPT + w) * BN + ch) * 2 + 1]; }
        long long t8;
        if (T8 == 1) t8 = t.tile;
        else {
            const int tiles_y8 = (imgH + TILE_H - 1) / TILE_H;
            const int trem = t.tile - t.b * tiles_per_img;
            const int ty8 = T8 * (trem / tiles_x) + part;
            if (ty8 >= tiles_y8) return;
            t8 = ((long long)t.b * tiles_y8 + ty8) * tiles_x + trem % tiles_x;
        }
        float* dst = ap->gn_part + (t8 * ap->outC + t.cout0 + ch) * 2;
        dst[0] = s0; dst[1] = s1;
    }
}

}}  // namespace storm::epi
