// Implicit-GEMM convolution / NT-GEMM on the CDNA4 matrix cores.
//
//   out[b][pix][co] = ( sum_seg sum_tap sum_ci X_seg[b][pix + tap][ci] * W_seg[tap][co][ci]
//                       + bias[co] + tbias[b][co] + skip[b][pix][co] ) * scale
//
// Replaces nn.Conv2d 3x3 / 1x1, layers.NIN and the attention einsums of the reference
// (see include/storm_hip.h).  Design (MI355X-first, not a cuDNN-style translation):
//   * NHWC activations; a workgroup (4 wave64) owns an 8x32-pixel output tile x BN output
//     channels; the GEMM is D[co][pix] = W[co][k] * X[k][pix] so both MFMA operands are
//     16-byte contiguous along the channel (K) axis -> ds_read_b128 fragments, no transposes.
//   * per K-chunk (128 B of channels per pixel) the haloed 10x34 input patch is staged ONCE in
//     LDS and reused by all nine taps (9x fewer L2->LDS bytes than im2col);  weight tiles
//     [BN][chunk] stream per tap through a 2-deep LDS ring (one barrier per tap).
//   * staging: the main 3x3 instantiation copies both operands global -> LDS with `buffer_load ... lds`
//     (LDS-DMA: bank swizzle on the per-lane source offset, out-of-range offsets = hardware zero fill, the next
//     weight tile lands under the MFMAs of the current tap, one memory round trip per patch, fused GroupNorm
//     = in-place rewrite of a lane's own units); the other instantiations (1x1 / GEMM, prefetching 256-cout
//     tile, and the register-staged build kept behind STORM_CONV_DMA=0) stage through registers with branch-free
//     raw-buffer loads.
//   * LDS rows are XOR-swizzled (conv_index.h) so every ds_read_b128 group is conflict free; the patch image is
//     swizzled by pixel COLUMN, so a k-group's fragment addresses are ONE VGPR plus instruction immediates.
//   * ~75 KB LDS and <=256 VGPR per workgroup -> 2 workgroups / CU so one group's staging
//     overlaps the other's MFMA phase; block ids are mapped so each XCD's L2 sees a contiguous
//     run of pixel tiles and both cout halves of a tile.
//   * epilogue: accumulators go through an LDS transpose so that global stores (and the
//     skip / bias reads) are 16-32 B per lane, full 128-B lines per 4-8 lanes.
//   * bf16 operands: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  fp32 operands:
//     v_mfma_f32_32x32x2_f32 (exact fp32, used by the parity path).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "conv_index.h"
#include <vector>
#include "conv_params.h"
#include "conv_dispatch_table.h"

namespace storm {
using namespace cidx;

template <int TAPS, int WM, int WAVES_M, int WAVES_N, bool PF, bool FP = false, int ABL_ = 0>
struct ConvCfg {
    static constexpr int ABL = ABL_;           // profiling-only ablation mask (separate instantiations)
    static constexpr bool PREFETCH = PF;       // double-buffered patch, next chunk fetched under the MFMAs
    static constexpr bool DMA = FP;            // operands staged by LDS-DMA (buffer_load ... lds) instead of registers
    static constexpr int NWAVES = WAVES_M * WAVES_N;
    static constexpr int THREADS = 64 * NWAVES;
    static constexpr int WN = TILE_H / WAVES_N;          // pixel tile rows (of 32 px) per wave
    static constexpr int BN = WAVES_M * WM * 32;          // output channels per workgroup
    static constexpr int NPIX = Geo<TAPS>::NPIX;
    static constexpr int PATCH_BYTES = (FP ? (NPIX + 7) / 8 * 8 : NPIX) * PIX_BYTES;     // DMA: whole 8-row (1 KiB) pieces
    static constexpr int WBUF_BYTES = BN * PIX_BYTES;
    static constexpr int NPBUF = PF ? 2 : 1;
    static constexpr int SS_BYTES = FP ? 1024 : 0;                                      // DMA: (scale, shift) table of a chunk
    static constexpr int OFF_SS = NPBUF * PATCH_BYTES + 2 * WBUF_BYTES;
    static constexpr int MAIN_BYTES = OFF_SS + SS_BYTES;
    static constexpr int PPIECES = PATCH_BYTES / 1024;                                   // DMA pieces of a patch
    static constexpr int PUD = (PPIECES + NWAVES - 1) / NWAVES;                          // ... per wave
    static constexpr int WUD = BN / 8 / NWAVES;                                          // weight DMA pieces per wave and tap
    static constexpr int PR = (WN >= 2 && NWAVES * 64 * WM * 128 <= MAIN_BYTES) ? 2 : 1;   // pixel rows staged per epilogue pass
    static constexpr int STAGE_BYTES = NWAVES * 32 * PR * WM * 128;
    static constexpr int LDS_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
    static constexpr int BLOCKS_PER_CU = (160 * 1024) / LDS_BYTES >= 2 ? 2 : 1;
    static constexpr int MIN_WAVES_PER_SIMD = BLOCKS_PER_CU * NWAVES / 4;
    static constexpr int PU = (NPIX * 8 + THREADS - 1) / THREADS;   // patch 16-B units per thread
    static constexpr int WU = BN * 8 / THREADS;                      // weight units per thread
    static_assert(WN % PR == 0, "epilogue passes");
    static_assert(BN * 8 % THREADS == 0, "");
    static_assert(WAVES_N * WN == TILE_H, "");
};


template <typename T, int TAPS, int WM, int WAVES_M, int WAVES_N, bool PF, bool FP, int ABL>
__device__ __forceinline__ void conv_tile(const ConvParams& a, const int vblock, const int n_ct, const int tiles_per_xcd,
                                          const int ntiles, const int tiles_x, const int tiles_per_img) {
    typedef ConvCfg<TAPS, WM, WAVES_M, WAVES_N, PF, FP, ABL> Cfg;
    typedef typename Mma<T>::Frag Frag;
    constexpr int THREADS = Cfg::THREADS;
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int KC = 8 * PER16;              // channels per K-chunk (128 B)
    constexpr int KG = 2 * PER16;              // channels per k-group (one fragment slot pair)
    constexpr int WN = Cfg::WN, BN = Cfg::BN, PW = Geo<TAPS>::PW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const pbuf = smem;                                    // NPBUF patch buffers, then the 2-deep weight ring
    char* const wbuf = smem + Cfg::NPBUF * Cfg::PATCH_BYTES;

    const BlockMap bm = block_map(vblock, n_ct, tiles_per_xcd);
    if (bm.tile >= ntiles) return;
    const int b = bm.tile / tiles_per_img;
    const int trem = bm.tile - b * tiles_per_img;
    const int ty0 = (TAPS == 9) ? (trem / tiles_x) * TILE_H : 0;
    const int tx0 = (TAPS == 9) ? (trem % tiles_x) * TILE_W : 0;
    const long long npix = (long long)a.H * a.W;
    const long long lin0 = (long long)trem * (TILE_H * TILE_W);     // TAPS==1: linear pixel base
    const int cout0 = bm.ct * BN;

    int tid_ = threadIdx.x;
    launder(tid_);                       // per-tile opaque: keeps lane-derived address math out of the persistent loop's preheader
    const int tid = tid_, lane = tid & 63, wave = uniform(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    unsigned long long* const trace_rec = (ABL & 64) && a.trace ? a.trace + ((long long)vblock * (WAVES_M * WAVES_N) + wave) * TRACE_SLOTS : nullptr;
    auto stamp = [&](int idx) {                         // profiling build only (tools/conv_trace.py)
        if ((ABL & 64) && trace_rec && idx < TRACE_SLOTS) {
            const unsigned long long t = hw_memtime();
            if (lane == 0) trace_rec[idx] = t;
        }
    };
    if ((ABL & 64) && trace_rec && lane == 0) trace_rec[0] = hw_ids();
    stamp(1);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // ---- per-chunk scalars (one K-chunk = 128 B of channels of one run) ----------------------------
    struct Chunk {
        const T* src; const T* w; const float* gn_ss;
        BufRsrc sb, wb;                                // buffer views of src / w: out-of-range reads are zeros
        u32x4 ssrd, wsrd, tsrd;                        // (DMA staging) raw descriptors of src, w and the GN (scale, shift) pairs
        int C, cbeg, cvalid, CinP, w_rows, ntaps, kbeg, klim, gn_silu;
        long long w_tapstride;
    };
    auto get_chunk = [&](int r, int ch) {
        const ConvRun& R = a.run[r];               // r is wave-uniform: scalar loads from the kernarg segment
        Chunk c;
        c.src = reinterpret_cast<const T*>(R.src) + (long long)b * R.src_bstride;
        c.w = reinterpret_cast<const T*>(R.w) + (long long)b * R.w_bstride + R.wc0;
        c.sb = make_buf(c.src, (uint32_t)((long long)a.H * a.W * R.C * sizeof(T)));
        c.wb = make_buf(c.w, (uint32_t)(((long long)R.ntaps * R.w_tapstride - R.wc0) * sizeof(T)));
        if (Cfg::DMA) {
            c.ssrd = make_srd(c.src, (uint32_t)((long long)a.H * a.W * R.C * sizeof(T)));
            c.wsrd = make_srd(c.w, (uint32_t)(((long long)R.ntaps * R.w_tapstride - R.wc0) * sizeof(T)));
        }
        c.C = R.C; c.cbeg = R.c0 + ch * KC; c.cvalid = min(KC, R.cn - ch * KC);
        c.CinP = R.CinP; c.w_rows = R.w_rows; c.ntaps = R.ntaps; c.w_tapstride = R.w_tapstride;
        c.kbeg = ch * KC; c.klim = R.CinP - R.wc0;
        c.gn_ss = R.gn_ss ? R.gn_ss + 2 * ((long long)b * R.gn_C + R.wc0 + ch * KC) : nullptr;
        c.gn_silu = R.gn_silu;
        if (Cfg::DMA) c.tsrd = make_srd(c.gn_ss ? static_cast<const void*>(c.gn_ss) : static_cast<const void*>(c.src), c.gn_ss ? (uint32_t)c.cvalid * 8u : 0u);
        return c;
    };
    auto chunks_of = [&](int r) { return (a.run[r].cn + KC - 1) / KC; };

    // ---- loaders --------------------------------------------------------------------------------
    uint4 wreg[Cfg::WU];
    auto load_w = [&](const Chunk& c, int tp) {
        const uint32_t wsoff = (uint32_t)((long long)tp * c.w_tapstride * sizeof(T));
#pragma unroll
        for (int i = 0; i < Cfg::WU; ++i) {
            const int u = tid + i * THREADS;
            const int row = u >> 3, slot = u & 7;
            const int co = cout0 + row, k = c.kbeg + slot * PER16;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            v = buf_load16(c.wb, (co < c.w_rows && k < c.klim) ? (uint32_t)(co * c.CinP + k) * (uint32_t)sizeof(T) : BUF_OOB, wsoff);
            wreg[i] = v;
        }
    };
    auto store_w = [&](int buf) {
        char* dst = wbuf + buf * Cfg::WBUF_BYTES;
#pragma unroll
        for (int i = 0; i < Cfg::WU; ++i) {
            const int u = tid + i * THREADS;
            *reinterpret_cast<uint4*>(dst + lds_off(u >> 3, u & 7)) = wreg[i];
        }
    };
    // patch staging is split: issue (global loads -> registers) ... commit (GroupNorm affine + SiLU when fused,
    // then LDS); with PREFETCH the commit of chunk c+1 happens after the MFMAs of chunk c's first tap.
    uint4 preg[Cfg::PU];
    uint32_t pmask = 0;                          // bit i: unit i was inside the image (zero padding stays zero)
    auto patch_issue = [&](const Chunk& c, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < Cfg::PU; ++i) {
            if (i < i0 || i >= i1) continue;
            const int u = tid + i * THREADS;
            const int p = u >> 3, slot = u & 7;
            int pix;
            bool ok = (u < Cfg::NPIX * 8) && (slot * PER16 < c.cvalid);
            if (TAPS == 9) {
                const int py = p / PW, px = p - py * PW;
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                pix = gy * a.W + gx;
            } else {
                pix = (int)lin0 + p;
                ok = ok && pix < (int)npix;
            }
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            v = buf_load16(c.sb, ok ? (uint32_t)(pix * c.C + c.cbeg + slot * PER16) * (uint32_t)sizeof(T) : BUF_OOB, 0u);
            preg[i] = v;
            pmask = ok ? (pmask | (1u << i)) : (pmask & ~(1u << i));
        }
    };
    auto patch_commit = [&](const Chunk& c, char* dst, int i0, int i1) {
        float ss[16];
        if (c.gn_ss != nullptr) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { ss[i] = 1.f; ss[8 + i] = 0.f; }
            if ((tid & 7) * PER16 < c.cvalid) load_ss<PER16>(c.gn_ss, tid & 7, ss);
        }
#pragma unroll
        for (int i = 0; i < Cfg::PU; ++i) {
            if (i < i0 || i >= i1) continue;
            const int u = tid + i * THREADS;
            if (u < Cfg::NPIX * 8) {
                uint4 v = preg[i];
                if (c.gn_ss != nullptr && (pmask >> i) & 1u) v = gn_act_slot(v, ss, c.gn_silu, (T*)nullptr);
                *reinterpret_cast<uint4*>(dst + patch_off(u >> 3, (u >> 3) % PW, u & 7)) = v;
            }
        }
    };

    // ---- LDS-DMA staging (Cfg::DMA): the same LDS images, filled by `buffer_load_dwordx4 ... lds` -------------
    // An instruction's LDS image is lane-linear (1 KiB = 8 rows), so the bank swizzle goes on the per-lane SOURCE
    // offset; padding / ragged channels / rows past the matrix are out-of-range offsets = hardware zeros.  No
    // staging registers, no ds_write pass; the fused GroupNorm transform rewrites a lane's own units in place.
    const int wave_u = tid >> 6;
    auto w_dma = [&](const Chunk& c, int tp, int buf) {
        char* dst = wbuf + buf * Cfg::WBUF_BYTES + wave_u * (Cfg::WUD * 1024);
        const uint32_t wsoff = (uint32_t)((long long)tp * c.w_tapstride * sizeof(T));
#pragma unroll
        for (int j = 0; j < Cfg::WUD; ++j) {
            const int row = (wave_u * Cfg::WUD + j) * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            const int co = cout0 + row, k = c.kbeg + slot * PER16;
            dma16(c.wsrd, (co < c.w_rows && k < c.klim) ? (uint32_t)(co * c.CinP + k) * (uint32_t)sizeof(T) : BUF_OOB, wsoff, dst + j * 1024, lane);
        }
    };
    uint32_t dmask = 0;                          // bit i: DMA unit i of this lane is real input (gets the GN transform)
    auto patch_dma = [&](const Chunk& c, char* dst) {
#pragma unroll
        for (int i = 0; i < Cfg::PUD; ++i) {
            int k = wave_u + i * Cfg::NWAVES;                     // piece: patch rows 8k .. 8k+7
            if (k >= Cfg::PPIECES) k -= Cfg::NWAVES;               // surplus slot: the same piece again
            const int p = k * 8 + (lane >> 3);
            const int py = p / PW, px = p - py * PW;
            const int slot = (lane & 7) ^ ((px >> 1) & 7);
            int pix;
            bool ok = p < Cfg::NPIX && slot * PER16 < c.cvalid;
            if (TAPS == 9) {
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                pix = gy * a.W + gx;
            } else {
                pix = (int)lin0 + p;
                ok = ok && pix < (int)npix;
            }
            dma16(c.ssrd, ok ? (uint32_t)(pix * c.C + c.cbeg + slot * PER16) * (uint32_t)sizeof(T) : BUF_OOB, 0u, dst + k * 1024, lane);
            dmask = ok ? (dmask | (1u << i)) : (dmask & ~(1u << i));
        }
        dma16(c.tsrd, (uint32_t)lane * 16u, 0u, smem + Cfg::OFF_SS, lane);
    };
    auto patch_fixup_t = [&](char* dst, auto silu_) {           // fused GroupNorm-apply (+ SiLU), in place, own units
#pragma unroll
        for (int i = 0; i < Cfg::PUD; ++i) {
            const int k = wave_u + i * Cfg::NWAVES;
            if (k < Cfg::PPIECES && ((dmask >> i) & 1u)) {
                const int p = k * 8 + (lane >> 3);
                const int slot = (lane & 7) ^ (((p % PW) >> 1) & 7);
                uint4* const q = reinterpret_cast<uint4*>(dst + k * 1024 + lane * 16);
                float ss[16];
                load_ss<PER16>(reinterpret_cast<const float*>(smem + Cfg::OFF_SS), slot, ss);
                *q = gn_act_slot_t<decltype(silu_)::value>(*q, ss, (T*)nullptr);
            }
        }
    };
    auto patch_fixup = [&](const Chunk& c, char* dst) {          // (one uniform branch per patch, not a select per channel pair)
        if (c.gn_silu) patch_fixup_t(dst, std::true_type{}); else patch_fixup_t(dst, std::false_type{});
    };

    // per-lane fragment offsets: weights rows of mi are +32 rows (same swizzle) -> one VGPR + immediates; k-group j
    // flips bits 5-6 of the slot field (slot = 2j + lane / 32)
    const int abase = lds_off(wm * WM * 32 + (lane & 31), lane >> 5);

    auto compute = [&](const char* patch, const char* wb, int dy, int dx, int nk) {
        const int px = (lane & 31) + dx;
        const int pb = patch_off((wn * WN + dy) * PW + px, px, lane >> 5);       // pixel row ni: + ni * PW rows
        auto load_frags = [&](int j, Frag (&fa)[WM], Frag (&fb)[WN]) {
            const char* wa = wb + (abase ^ (j << 5));
            const char* pp = patch + (pb ^ (j << 5));
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const Frag*>(wa + mi * 32 * PIX_BYTES);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const Frag*>(pp + ni * PW * PIX_BYTES);
        };
        auto mma = [&](const Frag (&fa)[WM], const Frag (&fb)[WN]) {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) Mma<T>::run(fa[mi], fb[ni], acc[mi][ni]);
        };
        // ONE rolled loop (a second, unrolled copy of the MFMA code makes the register allocator keep two
        // homes for the accumulators).
        if (ABL & 3) {                          // profiling ablations (own instantiations, never dispatched in production)
            Frag fa[WM], fb[WN];
            load_frags(0, fa, fb);
#pragma unroll 1
            for (int j = 0; j < nk; ++j) {
                if (!(ABL & 2)) load_frags(j, fa, fb);
                if (!(ABL & 1)) mma(fa, fb);
                else {
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi) keep(fa[mi]);
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) keep(fb[ni]);
                }
            }
        } else {
#pragma unroll 1
            for (int j = 0; j < nk; ++j) {
                Frag fa[WM], fb[WN];
                load_frags(j, fa, fb);
                __builtin_amdgcn_sched_barrier(0);      // every fragment read of the k-group in flight before its first MFMA
                mma(fa, fb);
            }
        }
    };

    // ---- main loop: K-chunks (runs flattened) x taps ------------------------------------------------
    const int nruns = a.nruns;
    int r = 0, ch = 0, nch_r = chunks_of(0), step = 0, ci = 0;
    Chunk cur = get_chunk(0, 0);
    constexpr int HALF = (Cfg::PU + 1) / 2;      // non-prefetch mode stages the patch in two register halves
    if (PF) {
        patch_issue(cur, 0, Cfg::PU);
        patch_commit(cur, pbuf, 0, Cfg::PU);
    }
    if (Cfg::DMA) w_dma(cur, 0, 0);
    else { load_w(cur, 0); store_w(0); }
    stamp(2);
    while (true) {
        int nr = r, nc = ch + 1;
        if (nc == nch_r) { nc = 0; ++nr; }
        const bool has_nc = nr < nruns;
        const Chunk nxt = get_chunk(has_nc ? nr : r, has_nc ? nc : ch);
        if (!PF && !((ABL & (4 | 32)) && ci > 0)) {
            stamp(400 + 4 * ci);
            __syncthreads();                       // every wave finished reading the previous patch
            stamp(401 + 4 * ci);
            if (Cfg::DMA) {                                 // one memory round trip, no registers, no ds_write pass
                patch_dma(cur, pbuf);
                vm_wait<0>();                               // (also the weight tile issued before this stage)
                if (cur.gn_ss != nullptr) patch_fixup(cur, pbuf);
            } else {
                patch_issue(cur, 0, HALF); patch_commit(cur, pbuf, 0, HALF);
                stamp(402 + 4 * ci);
                patch_issue(cur, HALF, Cfg::PU); patch_commit(cur, pbuf, HALF, Cfg::PU);
            }
            stamp(403 + 4 * ci);
        }
        const char* const patch = pbuf + (PF ? (ci & 1) * Cfg::PATCH_BYTES : 0);
        const int nk = (cur.cvalid + KG - 1) / KG;
        const int ntaps = cur.ntaps;
        for (int tp = 0; tp < ntaps; ++tp) {
            // next step's weight tile -> registers; (PF, first tap) next chunk's patch -> registers.
            // Both fly during the MFMAs below and are written to LDS after them.
            const bool more_taps = tp + 1 < ntaps;
            const bool has_next = more_taps || has_nc;
            if (!Cfg::DMA && !(ABL & (4 | 16))) {
                if (more_taps) load_w(cur, tp + 1);
                else if (has_nc) load_w(nxt, 0);
            }
            const bool pf_now = PF && tp == 0 && has_nc && !(ABL & (4 | 32));
            if (pf_now) patch_issue(nxt, 0, Cfg::PU);
            stamp(4 + 4 * step);
            __syncthreads();                       // patch + wbuf[step&1] visible; ring slots of step-1 free
            if (Cfg::DMA && has_next && !(ABL & 16)) {            // next tile straight into the other ring slot: lands under the MFMAs
                if (more_taps) w_dma(cur, tp + 1, (step + 1) & 1);
                else w_dma(nxt, 0, (step + 1) & 1);
            }
            stamp(5 + 4 * step);
            int dy = 0, dx = 0;
            if (TAPS == 9) {
                if (ntaps == 9) { dy = tp / 3; dx = tp - dy * 3; } else { dy = 1; dx = 1; }
            }
            compute(patch, wbuf + (step & 1) * Cfg::WBUF_BYTES, dy, dx, nk);
            stamp(6 + 4 * step);
            if (Cfg::DMA) vm_wait<0>();            // own share of the next weight tile landed before the next barrier
            else if (has_next) store_w((step + 1) & 1);
            if (pf_now) patch_commit(nxt, pbuf + ((ci + 1) & 1) * Cfg::PATCH_BYTES, 0, Cfg::PU);
            stamp(7 + 4 * step);
            ++step;
        }
        if (!has_nc) break;
        if (nr != r) nch_r = chunks_of(nr);
        cur = nxt; r = nr; ch = nc; ++ci;
    }

    // ---- epilogue: LDS transpose -> (bias, temb bias, skip, scale) -> wide stores ---------
    stamp(500);
    __syncthreads();
    stamp(501);
    if (ABL & 8) {                                      // profiling ablation: keep the accumulators alive, store nothing
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) keep(acc[mi][ni]);
        return;
    }
    constexpr int PR = Cfg::PR;                 // pixel rows (of 32 px) staged per pass and wave
    constexpr int SROWS = 32 * PR;
    char* const stage = smem + wave * (SROWS * WM * 128);      // private to this wave: only wave-level ordering needed
    constexpr int LPR = WM * 4;                 // lanes per staged row (8 couts each)
    constexpr int RPI = 64 / LPR;               // rows per read iteration
    const int skipC = a.outC;
    const int c8 = lane & (LPR - 1);
    const int co = cout0 + wm * WM * 32 + c8 * 8;              // this lane's 8 output channels (same in every iteration)
    float badd[8];                              // bias + per-batch time-embedding bias, loaded once
#pragma unroll
    for (int e = 0; e < 8; ++e) badd[e] = 0.f;
    if (co + 8 <= a.Cout) {
        if (a.bias) { float bb[8]; load8(a.bias + co, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) badd[e] += bb[e]; }
        if (a.tbias) { float bb[8]; load8(a.tbias + (long long)b * a.tbias_stride + co, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) badd[e] += bb[e]; }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (co + e < a.Cout) {
                if (a.bias) badd[e] += a.bias[co + e];
                if (a.tbias) badd[e] += a.tbias[(long long)b * a.tbias_stride + co + e];
            }
    }
    const bool co_ok = co < a.outC;
    const T* const skip_b = reinterpret_cast<const T*>(a.skip) + (long long)b * a.skip_bstride;
    // out = (acc + bias + temb bias + skip) * scale, evaluated as packed fma: (acc [+ skip]) * scale + (bias * scale);
    // channel pairs stay in adjacent registers from the staging read to the bf16 pack (v_pk_fma_f32 / v_pk_add_f32)
    f32x2 badd2[4], gsum2[4], gsq2[4];
    const f32x2 scale2 = {a.scale, a.scale};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        badd2[i] = f32x2{badd[2 * i] * a.scale, badd[2 * i + 1] * a.scale};
        gsum2[i] = f32x2{0.f, 0.f}; gsq2[i] = f32x2{0.f, 0.f};
    }
    float gsum[8], gsq[8];                      // GroupNorm partials of this lane's 8 channels (fused statistics)
#pragma unroll
    for (int e = 0; e < 8; ++e) { gsum[e] = 0.f; gsq[e] = 0.f; }
    // store-loop constants: parameters read once (scalar), the lane's pixel / channel offset, the two staged-row swizzles
    const int outC = a.outC, out_f32 = a.out_f32, imgH = a.H, imgW = a.W;
    const bool has_skip = a.skip != nullptr;
    char* const out_b = reinterpret_cast<char*>(a.out) + (long long)b * a.out_bstride * (out_f32 ? 4 : (int)sizeof(T));
    const int l8 = (lane >> 3) & 7;             // LPR == 8: the pixel (of the 8 per iteration) this lane stores
    const int gx0 = tx0 + l8;
    const uint32_t o_lane = (uint32_t)(l8 * outC + co);
    int srow[2][2];
#pragma unroll
    for (int odd = 0; odd < 2; ++odd)
#pragma unroll
        for (int h = 0; h < 2; ++h) srow[odd][h] = stage_off<WM>(odd * 8 + l8, 2 * c8 + h) - odd * 8 * (WM * 128);
    // where iteration `it` of pass `pass` stores: validity and element offset inside this batch image (fits 32 bits)
    auto locate = [&](int pass, int it, bool& ok, uint32_t& o) {
        if constexpr (LPR == 8) {
            const int trow = wn * WN + pass * PR + it / 4, nb = (it % 4) * RPI;          // (it is a compile-time constant)
            if (TAPS == 9) {
                const int gy = ty0 + trow;
                ok = gy < imgH && gx0 + nb < imgW;
                o = (uint32_t)(gy * imgW + tx0 + nb) * (uint32_t)outC + o_lane;
            } else {
                const int pix_u = (int)lin0 + trow * TILE_W + nb;
                ok = pix_u + l8 < (int)npix;
                o = (uint32_t)pix_u * (uint32_t)outC + o_lane;
            }
        } else {
            const int row = it * RPI + lane / LPR;
            const int trow = wn * WN + pass * PR + (row >> 5), n = row & 31;
            int pix;
            if (TAPS == 9) {
                const int gy = ty0 + trow, gx = tx0 + n;
                ok = gy < imgH && gx < imgW;
                pix = gy * imgW + gx;
            } else {
                pix = (int)lin0 + trow * TILE_W + n;
                ok = pix < (int)npix;
            }
            o = (uint32_t)(pix * outC + co);
        }
    };
    // skip operands are fetched ONE PASS ahead, each into the registers its predecessor (same iteration, previous pass) has just
    // left; the first pass's before the first staging write (fetched where it is used, every load exposed its memory latency)
    // The store loop comes in three instantiations picked by uniform branches: a tile inside the image with all of its couts valid
    // and a 16-bit output (no per-lane masks or range tests, stores through base + 32-bit offset) with / without a skip operand,
    // and the general one.  With one wave per SIMD a workgroup issues a VALU instruction only every other slot
    // (tools/ubench/valu_rate), so the epilogue of this kernel is bound by its instruction count.
    constexpr int NIT = SROWS / RPI;
    auto run_passes = [&](auto fast_, auto skipk_) {
    constexpr bool FAST = decltype(fast_)::value;
    constexpr int SKIPK = decltype(skipk_)::value;          // 1: skip operand, 0: none, -1: run-time
    const bool skip_on = SKIPK < 0 ? has_skip : SKIPK == 1;
    Raw8<T> skq[NIT];
    auto skip_fetch = [&](int pass, int it) {
        bool ok; uint32_t o;
        locate(pass, it, ok, o);
        if (FAST || (ok && co_ok)) fetch8(skip_b + o, skq[it]);
    };
    if (skip_on) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) skip_fetch(0, it);
    }
#pragma unroll
    for (int pass = 0; pass < WN / PR; ++pass) {
        if (pass > 0) wave_sync();              // this wave's reads of the previous pass are done
#pragma unroll
        for (int nn = 0; nn < PR; ++nn)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = nn * 32 + (lane & 31);
                    const f32x16& c = acc[mi][pass * PR + nn];
                    *reinterpret_cast<float4*>(stage + stage_off<WM>(row, stage_wslot(lane, mi, g))) =
                        make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                }
        wave_sync();                            // LDS executes a wave's requests in order: no workgroup barrier
        // store loop: an iteration stores RPI pixels x (WM x 32) couts per wave.  With 8 lanes per pixel (WM = 2) everything that
        // does not depend on the lane is scalar - the pixel row's validity, the element offset of the iteration's first pixel -
        // and the lane adds its own (pixel l8, cout octet c8) offset; the staged row's swizzle has two variants (it even / odd).
        // (`lane` is opaque to the compiler here, so the generic row / column arithmetic cost 30 VALU instructions per store:
        // measured in conv_pipe.hip, 5.9 k of the epilogue's 12.1 k cycles per tile.)
#pragma unroll
        for (int it = 0; it < SROWS / RPI; ++it) {
            float4 v0, v1;
            bool ok;
            uint32_t o;
            locate(pass, it, ok, o);
            if constexpr (LPR == 8) {
                const char* const sp = stage + it * RPI * (WM * 128);
                v0 = *reinterpret_cast<const float4*>(sp + srow[it & 1][0]);
                v1 = *reinterpret_cast<const float4*>(sp + srow[it & 1][1]);
            } else {
                const int row = it * RPI + lane / LPR;
                v0 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, 2 * c8));
                v1 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, 2 * c8 + 1));
            }
            f32x2 v2[4] = {f32x2{v0.x, v0.y}, f32x2{v0.z, v0.w}, f32x2{v1.x, v1.y}, f32x2{v1.z, v1.w}};
            const Raw8<T> skv = skq[it];
            if (skip_on && pass + 1 < WN / PR) skip_fetch(pass + 1, it);
            if (FAST || (ok && co_ok)) {
                if (skip_on) {
                    float sk[8];
                    unpack8(skv, sk);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v2[i] += f32x2{sk[2 * i], sk[2 * i + 1]};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v2[i] = __builtin_elementwise_fma(v2[i], scale2, badd2[i]);
                    gsum2[i] += v2[i];
                    gsq2[i] = __builtin_elementwise_fma(v2[i], v2[i], gsq2[i]);
                }
                const float v[8] = {v2[0].x, v2[0].y, v2[1].x, v2[1].y, v2[2].x, v2[2].y, v2[3].x, v2[3].y};
                if constexpr (FAST && sizeof(T) == 2) {
                    st16(out_b, o * 2u, pack8<T>(v));
                }
                else if (out_f32) store8(reinterpret_cast<float*>(out_b) + o, v);
                else store8(reinterpret_cast<T*>(out_b) + o, v);
            }
        }
    }
    };
    bool interior = sizeof(T) == 2 && !out_f32 && cout0 + BN <= outC;
    if (TAPS == 9) interior = interior && ty0 + TILE_H <= imgH && tx0 + TILE_W <= imgW;
    else interior = interior && lin0 + TILE_H * TILE_W <= npix;
    if (interior) {
        if (has_skip) run_passes(std::true_type{}, std::integral_constant<int, 1>{});
        else run_passes(std::true_type{}, std::integral_constant<int, 0>{});
    } else run_passes(std::false_type{}, std::integral_constant<int, -1>{});
    stamp(502);
    if (ABL & 64) { vm_wait<0>(); stamp(503); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { gsum[2 * i] = gsum2[i].x; gsum[2 * i + 1] = gsum2[i].y; gsq[2 * i] = gsq2[i].x; gsq[2 * i + 1] = gsq2[i].y; }
    if (a.gn_part != nullptr) {
        // lanes with equal (lane % LPR) hold the same 8 channels: butterfly over the row lanes, then
        // across the WAVES_N waves of a cout range through LDS; one coalesced [BN][2] store per block.
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) { gsum[e] += __shfl_xor(gsum[e], off, 64); gsq[e] += __shfl_xor(gsq[e], off, 64); }
        __syncthreads();                         // staging reads of the last pass are done
        float* red = reinterpret_cast<float*>(smem);      // [WAVES_N][BN][2]
        if (lane < LPR) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int chl = wm * WM * 32 + lane * 8 + e;
                red[(wn * BN + chl) * 2] = gsum[e];
                red[(wn * BN + chl) * 2 + 1] = gsq[e];
            }
        }
        __syncthreads();
        if (tid < BN && cout0 + tid < a.outC) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_N; ++w) { s0 += red[(w * BN + tid) * 2]; s1 += red[(w * BN + tid) * 2 + 1]; }
            float* dst = a.gn_part + ((long long)bm.tile * a.outC + cout0 + tid) * 2;
            dst[0] = s0; dst[1] = s1;
        }
    }
}

// Persistent launch: at most (CUs x workgroups per CU) workgroups, each walking virtual block ids
// blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8, so a workgroup stays on its XCD's
// tile range).  A workgroup's output stores drain while it already stages the next tile, there is no
// per-tile dispatch gap, and the chip-wide load / store bursts of equal-length tiles de-phase.
template <typename T, int TAPS, int WM, int WAVES_M, int WAVES_N, bool PF, bool FP, int ABL>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (ConvCfg<TAPS, WM, WAVES_M, WAVES_N, PF, FP, ABL>::MIN_WAVES_PER_SIMD))
void conv_igemm_kernel(const ConvParams a, const int n_ct, const int tiles_per_xcd,
                       const int ntiles, const int tiles_x, const int tiles_per_img, const int total_vblocks) {
    // The parameter block is read through the kernarg segment pointer (hw.h: kernarg_of), re-laundered every tile: otherwise
    // LICM hoists every s_load of the block out of the tile loop and the ~150 live SGPRs spill into VGPRs and on into scratch.
    auto kp = kernarg_of(a);
    for (int vb = blockIdx.x; vb < total_vblocks; vb += gridDim.x) {
        if (vb != (int)blockIdx.x) __syncthreads();      // LDS of the previous tile (staging / statistics) is free
        relaunder(kp);
        conv_tile<T, TAPS, WM, WAVES_M, WAVES_N, PF, FP, ABL>(*(const ConvParams*)kp, vb, n_ct, tiles_per_xcd, ntiles, tiles_x, tiles_per_img);
    }
}


template <typename T, int TAPS, int WM, int WAVES_M, int WAVES_N, bool PF, bool FP = false, int ABL = 0>
static int launch_conv(const storm_conv_args& a, hipStream_t st) {
    typedef ConvCfg<TAPS, WM, WAVES_M, WAVES_N, PF, FP, ABL> Cfg;
    auto kern = conv_igemm_kernel<T, TAPS, WM, WAVES_M, WAVES_N, PF, FP, ABL>;
    static bool attr_set = false;          // per instantiation; benign race (idempotent)
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    int tiles_x = 1, tiles_per_img;
    if (TAPS == 9) {
        tiles_x = cdiv(a.W, TILE_W);
        tiles_per_img = tiles_x * cdiv(a.H, TILE_H);
    } else {
        tiles_per_img = cdiv((long long)a.H * a.W, TILE_H * TILE_W);
    }
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int n_ct = cdiv(a.outC, Cfg::BN);
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long vblocks = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(vblocks > 0 && vblocks < (1LL << 31), "storm_conv: grid %lld out of range", vblocks);
    long long grid = vblocks;
#if defined(STORM_PROFILING)
    if (const int persist = switches().conv_persist) {               // (A/B: persistent tile walk in this kernel family)
        const long long resident = (long long)((device_cus() + 7) / 8 * 8) * Cfg::BLOCKS_PER_CU * persist;      // multiple of 8
        if (grid > resident) grid = resident;
    }
#endif
    ConvParams prm = make_params(a);
#if defined(STORM_PROFILING)
    if (ABL & 64) {                                              // device buffer address handed over by tools/conv_trace.py
        prm.trace = reinterpret_cast<unsigned long long*>(switches().conv_trace_ptr);
    }
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, prm, n_ct,
                       tiles_per_xcd, (int)ntiles, tiles_x, tiles_per_img, (int)vblocks);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

// Which kernel a convolution runs on (exported as storm_conv_kernel_name, so that bench.py's roofline names the
// kernel the launcher really picked instead of re-deriving the rule).
//   0: conv_igemm 128 cout x 256 px, 4 waves (64x128 each), 2 workgroups / CU overlap each other's staging; LDS-DMA
//   1: conv_igemm 128 cout x 256 px, 8 waves (64x64 each), 2 workgroups / CU      (profiling build only)
//   2: conv_igemm 256 cout x 256 px, 8 waves, 1 workgroup / CU, patch double-buffered through registers
//   3: conv_pipe.hip, 256 cout x 256 px, 8 waves in two ping-pong groups, chunk-unrolled LDS-DMA pipeline (16-bit 3x3)
//   4: conv_pipe128.hip, 128 cout x 512 px, the same pipeline for layers with <= 128 output channels (16-bit 3x3)
//   5: conv_duo.hip, 128 cout x 256 px, 4 waves, two workgroups / CU, single-stream phases (16-bit 3x3, <= 128 output channels)
//   6: conv_thin.hip, 8 input channels (stem, input-skip 1x1s): operands straight from global memory (16-bit)
//   8: conv_narrow.hip, 3x3 to <= 4 output channels (the output pyramid): 36-row 1x1 GEMM over the haloed region + nine-point gather
//   9: conv_pipe.hip with 128 cout x 256 px per 8-wave workgroup (64 x 64 per wave): the pipelined kernel for 128 ... 511 pixel tiles
//  10: conv_pipe.hip split-K: the 128-cout tile with K cut into 4 (or 2) slices on as many workgroups (fp32 slabs in the caller's
//      scratch) + one combine launch - 3x3 layers with at most 8 128-cout tiles per image (needs storm_conv_args.splitk_ws)
//   7: conv_igemm 64 cout x 256 px, 4 waves (32x128 each), 2 workgroups / CU; LDS-DMA - for 3x3 layers with so few pixel tiles
//      that 128-cout tiles leave CUs without work (the 32 x 64 level: 128 pixel tiles x 2 cout tiles on 256 CUs x 2 slots)
// K slices storm_conv would use for this call (0 = none), whatever scratch the caller brought
static int splitk_slices_of(const storm_conv_args& a) {
    const int forced = switches().conv_variant;
    if (forced >= 0 && forced != 10) return 0;
    bool any9 = false;
    for (int s = 0; s < a.nseg; ++s) any9 = any9 || a.seg[s].ntaps == 9;
    if (!any9 || a.dtype == STORM_F32 || !conv_pipe_supports(a)) return 0;
    const int S = conv_splitk_slices(a);
    if (S == 0 && forced == 10 && !a.out_f32) {                    // (tests: force a split wherever there are two nine-tap chunks)
        const int n9 = cdiv(a.seg[0].Ca, 64) + (a.seg[0].Cb ? cdiv(a.seg[0].Cb, 64) : 0);
        return n9 >= 4 ? 4 : n9 >= 2 ? 2 : 0;
    }
    return S;
}

// The measured exceptions to the ladder below (conv_dispatch_table.h, regenerated by tools/tune_dispatch.py): -1 = no entry
static int table_variant(const storm_conv_args& a) {
    if (a.dtype == STORM_F32 || a.seg[0].ntaps != 9 || a.nseg > 2 || switches().conv_table == 0) return -1;
    const int k9 = a.seg[0].Ca + a.seg[0].Cb, k1 = a.nseg == 2 ? a.seg[1].Ca + a.seg[1].Cb : 0;
    if (a.nseg == 2 && a.seg[1].ntaps != 1) return -1;
    const int tiles_img = cdiv(a.W, TILE_W) * cdiv(a.H, TILE_H);
    for (const DispatchEntry& e : kDispatchTable)
        if (e.variant >= 0 && e.k9 == k9 && e.k1 == k1 && e.outC == a.outC && e.tiles_img == tiles_img && a.B >= e.images_lo && a.B <= e.images_hi) {
            const bool ok = e.variant == 0 || e.variant == 2 || e.variant == 7 || ((e.variant == 3 || e.variant == 9) && conv_pipe_supports(a)) ||
                            (e.variant == 4 && conv_pipe128_supports(a));
            return ok ? e.variant : -1;
        }
    return -1;
}

static int choose_variant(const storm_conv_args& a, bool any9) {
    const int forced = switches().conv_variant;                      // (test / A-B hook, storm_set_switch)
    if (forced >= 0 && forced != 10) return forced;
    if (any9) {
        const int S = splitk_slices_of(a);
        if (S >= 2 && a.splitk_ws != nullptr && a.splitk_ws_bytes >= conv_splitk_bytes(a, S)) return 10;
    }
    if (forced == 10) return conv_pipe_supports(a) ? 9 : 0;        // (forced split without scratch / with one chunk: the unsplit tile)
    if (conv_thin_supports(a)) return 6;
    if (conv_narrow_supports(a)) return 8;
    // STORM_BATCH_INVARIANT: the ladder below is a throughput decision on the LAUNCH (rounds of workgroups), and the structures it chooses between sum K - and the
    // fused GroupNorm partials (conv_epilogue.h:write_stats: 2 wave rows x 4 pixel rows against 4 x 2) - in different orders: with the switch the decision is
    // taken for ONE image whatever the batch, and the batch-ranged table stays out, so a row's bits do not depend on what it is batched with.
    const bool per_image = switches().batch_invariant != 0;
    const int Bq = per_image ? 1 : a.B;
    if (any9 && a.outC > 32 && !per_image) { const int tv = table_variant(a); if (tv >= 0) return tv; }
    const long long px_tiles = (long long)Bq * cdiv((long long)a.H * a.W, TILE_H * TILE_W);
    const int cin9 = a.seg[0].Ca + a.seg[0].Cb;
    const int cus = device_cus();
    if (a.outC > 128 && !(any9 && conv_pipe_supports(a)) && px_tiles >= 512) return 2;       // (1x1 / NIN / GEMMs, the fp32 parity path)
    if (a.outC > 128 && any9 && conv_pipe_supports(a)) {
        // conv_pipe's 256-cout tile (3) or its 128-cout tile (9: twice the workgroups, each with the same chain of phases but 8 MFMAs per phase
        // and wave instead of 16).  A launch lasts rounds x (phases x time per phase + ~12 us fixed): 0.6 us per phase for the 256-cout tile,
        // 0.36 us for the 128-cout one (profiles/r04_probe_small_half.txt), rounds = workgroups / CUs rounded up (one workgroup per CU).  Up to
        // 128 pixel tiles the half tile wins (one round either way: 256 -> 256 @ 16 x 32 x 64 48.6 vs 66.7 us); from 129 on its second round
        // costs more than the 256-cout tile's idle CUs.  Rounds 1 - 4 sent everything below 512 pixel tiles to the half tile: measured wrong
        // by the tuner in round 5 wherever a call is not the bench batch (profiles/r05_tune_dispatch.log: two utterances @ 128 x 256
        // 0.075 -> 0.069 ms plain, 0.098 -> 0.080 with a shortcut; ncsnpplarge 8 x 32 x 128 ties; three 10-s rows @ 64 x 320 0.093 -> 0.075).
        const int n_ph = 2 * (9 * (cdiv(a.seg[0].Ca, 64) + (a.seg[0].Cb ? cdiv(a.seg[0].Cb, 64) : 0)) +
                              (a.nseg == 2 ? cdiv(a.seg[1].Ca, 64) + (a.seg[1].Cb ? cdiv(a.seg[1].Cb, 64) : 0) : 0));
        // (a persistent workgroup's second, third ... tile hides its prologue under the previous epilogue: ~4 us per round + ~8 us once)
        const long long r3 = cdiv(px_tiles * cdiv(a.outC, 256), (long long)cus), r9 = cdiv(px_tiles * cdiv(a.outC, 128), (long long)cus);
        const double t3 = (double)r3 * (0.60 * n_ph + 4.0) + 8.0;
        const double t9 = (double)r9 * (0.36 * n_ph + 4.0) + 8.0;
        return t9 < t3 ? 9 : 3;
    }
    // <= 128 output channels.  Three structures - the generic tile (0: this file, two workgroups per CU hide each other's epilogue), conv_pipe128
    // (4: 16 x 32 pixel tiles, triple-buffered 32-channel chunks) and conv_pipe's 128-cout tile (9) - measured against each other on MI355X, each kernel
    // sustained, alternating (profiles/r04_duo_fair_ab.txt, r04_half_fair_ab.txt, r05_tune_dispatch*.log): within 5 % of each other wherever the chip
    // is full (the part runs these layers at its power cap, LAB_NOTES 2.3; conv_pipe128 3 - 8 % ahead from 256 input channels on, the 128-cout tile of
    // conv_pipe 8 % behind) - so between them the ROUNDS decide.  (The 64-cout generic tile, 7, remains for few-tile layers conv_pipe does not cover.)
    //  Round 5, second tuner pass over the ragged stream's shapes (profiles/r05_tune_dispatch_stream_shapes_before.log: 60 exceptions of 4 - 20 %,
    //  none at the bench batch): what decides between the three is how a layer's tiles QUANTISE into rounds of workgroups -
    //    generic tile (0): 8 x 32 pixels, two workgroups per CU, dispatched by the hardware: full rounds of 2 x CUs tiles cost one unit each;
    //      a remainder of at most CUs tiles runs one workgroup per CU, ~0.68 of a unit (a lone workgroup has the CU to itself);
    //    conv_pipe128 (4): 16 x 32 pixels, one persistent workgroup per CU: ceil(tiles / CUs) units (0.95 of a unit once K reaches 9 x 128 + 256 - 256 input
    //      channels, or 128 with a 256-channel shortcut: its triple-buffered 32-channel chunks amortise over a longer K);
    //    conv_pipe's 128-cout tile (9): 8 x 32 pixels, one persistent workgroup per CU: ceil(tiles / CUs) x 0.545 units.
    //  E.g. one 9-s utterance at 256 x 1152 (1152 / 576 tiles): 2.68 | 3.0 | 2.73 units - measured 0.69 | 0.83 | 0.72 ms.
    if (a.outC > 32 && any9 && cin9 >= 32) {
        const long long n0 = px_tiles, n4 = (long long)Bq * cdiv(a.H, 16) * cdiv(a.W, TILE_W);
        const long long rem = n0 % (2LL * cus);
        double best = (double)(n0 / (2LL * cus)) + (rem == 0 ? 0.0 : rem <= cus ? 0.68 : 1.0);
        int pick = 0;
        if (switches().conv_pipe128 != 0 && conv_pipe128_supports(a)) {
            const int k1 = a.nseg == 2 ? a.seg[1].Ca + a.seg[1].Cb : 0;                  // (a fused 1x1 shortcut lengthens K like input channels do)
            const double t4 = (double)cdiv(n4, (long long)cus) * (9 * cin9 + k1 >= 9 * 128 + 256 ? 0.95 : 1.0);
            // (a tie goes to conv_pipe128 up to 4096 pixel tiles - 0 ... 4 % ahead there in rounds 2 - 4 and in the tuner's passes - and to the generic tile above)
            if (t4 < best * 0.995 || (t4 <= best * 1.005 && n0 <= 4096)) { best = t4; pick = 4; }
        }
        if (conv_pipe_supports(a)) {
            const double t9 = (double)cdiv(n0, (long long)cus) * 0.545;
            if (t9 < best * 0.995) { best = t9; pick = 9; }
        }
        if (pick != 0) return pick;
    }
    if (any9 && a.outC >= 128 && px_tiles * cdiv(a.outC, 128) <= 256) return 7;
    return 0;
}

template <typename T>
static int dispatch_conv(const storm_conv_args& a, hipStream_t st) {
    bool any9 = false;
    for (int s = 0; s < a.nseg; ++s) any9 = any9 || a.seg[s].ntaps == 9;
    const bool small = a.outC <= 32;
    const int variant = choose_variant(a, any9);
    if (variant == 6 && conv_thin_supports(a)) return launch_conv_thin(a, st);
    if (variant == 8 && conv_narrow_supports(a)) return launch_conv_narrow(a, st);
#if defined(STORM_PROFILING)
    // work-skipping instantiations for tools/ (no MFMA, no fragment reads, ...): profiling build only
    const int abl = switches().conv_ablate;
    if (any9 && !small && abl && variant < 3) {
        const bool v2 = variant == 2;
        // (the generic 128-cout tile as production runs it: weights by LDS-DMA.  1 no MFMA, 2 no fragment reads, 4 patch staged for the first
        //  chunk only (no patch path / fused GroupNorm transform), 8 no epilogue, 16 no weight DMA after the first tile, 64 wave stamps)
        switch (abl) {
            case 1: return v2 ? launch_conv<T, 9, 2, 4, 2, true, false, 1>(a, st) : launch_conv<T, 9, 2, 2, 2, false, true, 1>(a, st);
            case 2: return v2 ? launch_conv<T, 9, 2, 4, 2, true, false, 2>(a, st) : launch_conv<T, 9, 2, 2, 2, false, true, 2>(a, st);
            case 4: return v2 ? launch_conv<T, 9, 2, 4, 2, true, false, 4>(a, st) : launch_conv<T, 9, 2, 2, 2, false, true, 4>(a, st);
            case 8: return v2 ? launch_conv<T, 9, 2, 4, 2, true, false, 8>(a, st) : launch_conv<T, 9, 2, 2, 2, false, true, 8>(a, st);
            case 16: return v2 ? launch_conv<T, 9, 2, 4, 2, true, false, 16>(a, st) : launch_conv<T, 9, 2, 2, 2, false, true, 16>(a, st);
            case 20: return launch_conv<T, 9, 2, 2, 2, false, true, 20>(a, st);
            case 32: return v2 ? launch_conv<T, 9, 2, 4, 2, true, false, 32>(a, st) : launch_conv<T, 9, 2, 2, 2, false, false, 32>(a, st);
            case 64: return v2 ? launch_conv<T, 9, 2, 4, 2, true, false, 64>(a, st)
                             : (switches().conv_dma ? launch_conv<T, 9, 2, 2, 2, false, true, 64>(a, st) : launch_conv<T, 9, 2, 2, 2, false, false, 64>(a, st));
            default: break;
        }
    }
#endif
    if (any9) {
        if (small) return launch_conv<T, 9, 1, 1, 4, false>(a, st);
        if (variant == 3 && conv_pipe_supports(a)) return launch_conv_pipe(a, st);
        if (variant == 9 && conv_pipe_supports(a)) return launch_conv_pipe_half(a, st);
        if (variant == 10) return launch_conv_pipe_splitk(a, splitk_slices_of(a), st);
        if (variant == 4 && conv_pipe128_supports(a)) return launch_conv_pipe128(a, st);
#if defined(STORM_WITH_DUO)                                          // conv_duo.hip: profiling library and the test simulator only (LAB_NOTES 2.3: a tie)
        if (variant == 5 && conv_duo_supports(a)) return launch_conv_duo(a, st);
#endif
        if (variant == 2) return launch_conv<T, 9, 2, 4, 2, true>(a, st);
        if (variant == 7) return launch_conv<T, 9, 1, 2, 2, false, true>(a, st);
#if defined(STORM_PROFILING)                                          // A/B instantiations: 8-wave geometry, register staging
        if (variant == 1) return launch_conv<T, 9, 2, 2, 4, false>(a, st);
        if (switches().conv_dma == 0) return launch_conv<T, 9, 2, 2, 2, false, false>(a, st);
#endif
        return launch_conv<T, 9, 2, 2, 2, false, true>(a, st);
    }
    if (small) return launch_conv<T, 1, 1, 1, 4, false>(a, st);
    if (variant == 2) return launch_conv<T, 1, 2, 4, 2, true>(a, st);
#if defined(STORM_PROFILING)
    if (variant == 1) return launch_conv<T, 1, 2, 2, 4, false>(a, st);
#endif
    return launch_conv<T, 1, 2, 2, 2, false>(a, st);
}

// Name of the kernel storm_conv launches for these arguments (as rocprofv3 prints it).
static const char* kernel_name_of(const storm_conv_args& a) {
    bool any9 = false;
    for (int s = 0; s < a.nseg; ++s) any9 = any9 || a.seg[s].ntaps == 9;
    const char* tn = a.dtype == STORM_BF16 ? "storm::bf16_t" : a.dtype == STORM_F16 ? "storm::half_t" : "float";
    const int taps = any9 ? 9 : 1;
    const char* shape;
    int variant = choose_variant(a, any9);
    if (variant == 8 && conv_narrow_supports(a)) return conv_narrow_kernel_name(a);
    if (a.outC <= 32) variant = -1;
    if (variant == 6 && conv_thin_supports(a)) return conv_thin_kernel_name(a.dtype, taps);
    if (a.outC <= 32) shape = "1, 1, 4, false, false";
    else if (any9 && variant == 10) return a.dtype == STORM_F16 ? "storm::conv_pipe_splitk_kernel<storm::half_t, 128, 8>" : "storm::conv_pipe_splitk_kernel<storm::bf16_t, 128, 8>";
    else if (any9 && (variant == 3 || variant == 9) && conv_pipe_supports(a)) return conv_pipe_kernel_name(a.dtype, variant == 9);
    else if (any9 && variant == 4 && conv_pipe128_supports(a)) return conv_pipe128_kernel_name(a.dtype);
#if defined(STORM_WITH_DUO)
    else if (any9 && variant == 5 && conv_duo_supports(a)) return conv_duo_kernel_name(a.dtype);
#endif
    else if (variant == 2) shape = "2, 4, 2, true, false";
    else if (any9 && variant == 7) shape = "1, 2, 2, false, true";
#if defined(STORM_PROFILING)
    else if (variant == 1) shape = "2, 2, 4, false, false";
    else if (any9 && switches().conv_dma == 0) shape = "2, 2, 2, false, false";
#endif
    else shape = any9 ? "2, 2, 2, false, true" : "2, 2, 2, false, false";
    static thread_local char buf[160];
    snprintf(buf, sizeof(buf), "storm::conv_igemm_kernel<%s, %d, %s, 0>", tn, taps, shape);
    return buf;
}

}  // namespace storm

extern "C" int storm_conv_tiles(const storm_conv_args* ap) {
    if (ap == nullptr) return 0;
    bool any9 = false;
    for (int s = 0; s < ap->nseg; ++s) any9 = any9 || ap->seg[s].ntaps == 9;
    using namespace storm::cidx;
    if (any9) return storm::cdiv(ap->W, TILE_W) * storm::cdiv(ap->H, TILE_H);
    return storm::cdiv((long long)ap->H * ap->W, TILE_H * TILE_W);
}

extern "C" long long storm_conv_splitk_bytes(const storm_conv_args* ap) {
    if (ap == nullptr || ap->nseg < 1 || ap->nseg > 2 || ap->B <= 0 || ap->H <= 0 || ap->W <= 0) return 0;
    return storm::conv_splitk_bytes(*ap, storm::splitk_slices_of(*ap));
}

// One launch for P problems that run the SAME 16-bit 3x3 layer on tensors of their own (storm_hip.h).  blob: device scratch of at least
// storm_conv_group_blob_bytes(args, P) bytes; the tables are built on the host and copied there (synchronously: a convenience entry for
// single layers and tests - the whole-network object keeps a pinned image and copies asynchronously).  bn: 256 / 128 couts per workgroup, 0 = by tile count.
extern "C" long long storm_conv_group_blob_bytes(const storm_conv_args* a, int P) {
    if (a == nullptr || P < 1) return -1;
    long long t = 0;
    for (int g = 0; g < P; ++g) t += (long long)a[g].B * storm::cdiv(a[g].H, 8) * storm::cdiv(a[g].W, 32);
    return (long long)P * (long long)sizeof(storm::pipe::PipeParams) + 256 + t * (long long)sizeof(storm::pipe::GroupTile);
}
extern "C" int storm_conv_group(const storm_conv_args* a, int P, void* blob, long long blob_bytes, int bn, storm_stream_t s) {
    STORM_CHECK(a != nullptr && P >= 1 && blob != nullptr, "storm_conv_group: bad arguments");
    const long long need = storm_conv_group_blob_bytes(a, P);
    STORM_CHECK(blob_bytes >= need, "storm_conv_group: blob %lld < %lld bytes", blob_bytes, need);
    std::vector<char> host((size_t)need);
    if (storm::conv_thin_supports(a[0])) {                   // 8-channel inputs (stem, input-skip 1x1s): conv_thin.hip's grouped form
        const long long img = storm::conv_thin_group_bytes(P);
        STORM_CHECK(blob_bytes >= img, "storm_conv_group: blob %lld < %lld bytes", blob_bytes, img);
        std::vector<char> image((size_t)img);
        const long long nt = storm::conv_thin_group_prepare(a, P, image.data());
        if (nt <= 0) { storm::set_error("storm_conv_group: the problems are not one layer of the 8-channel-input kernel"); return STORM_ERR_UNSUPPORTED; }
        STORM_HIP(hipMemcpyAsync(blob, image.data(), (size_t)img, hipMemcpyHostToDevice, (hipStream_t)s));
#ifndef STORM_HOST_SIM
        STORM_HIP(hipStreamSynchronize((hipStream_t)s));
#endif
        return storm::launch_conv_thin_group(blob, P, nt, a[0].seg[0].ntaps, a[0].dtype, (hipStream_t)s);
    }
    if (storm::conv_narrow_supports(a[0])) {                 // the output pyramid's convolutions (<= 4 planes): conv_narrow.hip's grouped form
        const long long tabn = storm::conv_narrow_group_bytes(P);
        storm::pipe::GroupTile* tl = reinterpret_cast<storm::pipe::GroupTile*>(host.data() + tabn);
        const long long ntn = storm::conv_narrow_group_prepare(a, P, host.data(), tl, (need - tabn) / (long long)sizeof(storm::pipe::GroupTile));
        if (ntn <= 0) { storm::set_error("storm_conv_group: the problems are not one layer of the narrow-output 3x3 kernel"); return STORM_ERR_UNSUPPORTED; }
        STORM_HIP(hipMemcpyAsync(blob, host.data(), (size_t)(tabn + ntn * (long long)sizeof(storm::pipe::GroupTile)), hipMemcpyHostToDevice, (hipStream_t)s));
#ifndef STORM_HOST_SIM
        STORM_HIP(hipStreamSynchronize((hipStream_t)s));
#endif
        return storm::launch_conv_narrow_group(a[0], blob, reinterpret_cast<const storm::pipe::GroupTile*>(static_cast<char*>(blob) + tabn), ntn, (hipStream_t)s);
    }
    const long long tab = ((long long)P * (long long)sizeof(storm::pipe::PipeParams) + 255) / 256 * 256;
    storm::pipe::PipeParams* table = reinterpret_cast<storm::pipe::PipeParams*>(host.data());
    storm::pipe::GroupTile* tiles = reinterpret_cast<storm::pipe::GroupTile*>(host.data() + tab);
    const long long nt = storm::conv_pipe_group_prepare(a, P, table, tiles, (need - tab) / (long long)sizeof(storm::pipe::GroupTile));
    if (nt <= 0) { storm::set_error("storm_conv_group: the problems are not one layer of the pipelined 3x3 kernel (16-bit operands, > 0 nine-tap chunks, same weights)"); return STORM_ERR_UNSUPPORTED; }
    STORM_HIP(hipMemcpyAsync(blob, host.data(), (size_t)(tab + nt * (long long)sizeof(storm::pipe::GroupTile)), hipMemcpyHostToDevice, (hipStream_t)s));
#ifndef STORM_HOST_SIM
    STORM_HIP(hipStreamSynchronize((hipStream_t)s));         // (the host image dies with this call)
#endif
    if (bn == 0) bn = nt * storm::cdiv(a[0].outC, 256) >= 512 ? 256 : 128;
    return storm::launch_conv_pipe_group(reinterpret_cast<const storm::pipe::PipeParams*>(blob), reinterpret_cast<const storm::pipe::GroupTile*>(static_cast<char*>(blob) + tab), nt, a[0].outC, bn,
                                  a[0].dtype, (hipStream_t)s);
}

extern "C" int storm_conv(const storm_conv_args* ap, storm_stream_t s) {
    using namespace storm;
    STORM_CHECK(ap != nullptr, "storm_conv: null args");
    const storm_conv_args& a = *ap;
    STORM_CHECK(a.nseg >= 1 && a.nseg <= 2, "storm_conv: nseg=%d", a.nseg);
    STORM_CHECK(a.B > 0 && a.H > 0 && a.W > 0, "storm_conv: bad shape B=%d H=%d W=%d", a.B, a.H, a.W);
    STORM_CHECK(a.outC > 0 && a.outC % 8 == 0 && a.Cout <= a.outC, "storm_conv: outC=%d Cout=%d", a.outC, a.Cout);
    STORM_CHECK(a.out != nullptr, "storm_conv: null out");
    const int per16 = a.dtype == STORM_F32 ? 4 : 8;
    for (int i = 0; i < a.nseg; ++i) {
        const storm_conv_seg& g = a.seg[i];
        STORM_CHECK(g.src_a && g.w, "storm_conv: seg %d null pointer", i);
        STORM_CHECK(g.ntaps == 9 || g.ntaps == 1, "storm_conv: seg %d ntaps=%d", i, g.ntaps);
        STORM_CHECK(g.Ca > 0 && g.Ca % 8 == 0 && g.Cb >= 0 && g.Cb % 8 == 0, "storm_conv: seg %d Ca=%d Cb=%d", i, g.Ca, g.Cb);
        STORM_CHECK((g.Cb == 0) == (g.src_b == nullptr), "storm_conv: seg %d src_b / Cb mismatch", i);
        STORM_CHECK(g.CinP % per16 == 0 && g.CinP >= g.Ca + g.Cb, "storm_conv: seg %d CinP=%d", i, g.CinP);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(s);
    if (a.dtype == STORM_BF16) return dispatch_conv<bf16_t>(a, st);
    if (a.dtype == STORM_F16) return dispatch_conv<half_t>(a, st);
    if (a.dtype == STORM_F32) return dispatch_conv<float>(a, st);
    STORM_CHECK(false, "storm_conv: dtype %d", a.dtype);
}

extern "C" const char* storm_conv_kernel_name(const storm_conv_args* ap) {
    return ap ? storm::kernel_name_of(*ap) : "";
}
