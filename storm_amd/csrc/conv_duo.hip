// Pipelined 3x3 implicit-GEMM convolution for layers with <= 128 output channels, TWO 4-wave workgroups per CU (bf16 / fp16
// operands): the full-resolution 3x3 layers of NCSN++ (layers.py:119-126 ddpm_conv3x3 with the fused pieces of
// layerspp.py:242-274 listed in include/storm_hip.h) - the layers conv_igemm.hip's generic loop ran at 0.32-0.36 of the MFMA peak.
//
// Geometry: conv_igemm.hip's - 128 output channels x (8 x 32) pixels per workgroup, every wave 64 couts x (4 x 32) pixels = 8
// accumulator tiles, two workgroups resident per CU (78 KiB of LDS, <= 256 registers each), so one's epilogue, tile start and
// memory stalls run under the other's MFMAs.  K loop: conv_pipe128.hip's - host-built 32-channel chunk descriptors
// (conv_params.h), tap bodies unrolled at compile time, every operand byte copied global -> LDS by `buffer_load ... lds`, counted
// vmcnt waits.  What is new here (round 4; tools/ubench/phase_stream.hip measured the arrangement before the kernel was written:
// 1238 TF/s for this stream against 1152 for the staging-interval form at K = 1152, fused operand, two workgroups per CU):
//   * ONE instruction stream and ONE barrier per phase.  A phase is one tap of a 32-channel chunk = two k-groups = 16 MFMAs per
//     wave; everything else a phase has to do lives in the gaps BETWEEN its MFMAs: the 12 fragment reads (second k-group, then the
//     next phase's first), the issue of 2 weight pieces + 1 patch piece, and the fused GroupNorm-apply + SiLU of one 1-KiB patch
//     piece cut into steps of a few VALU instructions per gap.  (The staging-interval form of conv_pipe.hip / conv_pipe128.hip / the
//     round-2 conv_duo runs that work between two barriers while the wave's own matrix pipe idles.)
//   * weights stream THREE phases ahead through the 4-slot ring (8 KiB per phase); a phase's one counted wait - "everything up to
//     the previous phase's weight pieces has landed" - therefore gives a weight piece ~1.5 phases and a patch piece (issued after
//     its phase's weights) ~2.5 phases before anything waits for it.
//   * patch slot i (pieces wave + 4 i of the next chunk) is issued in phase i, has landed by the end of phase i + 2 and is
//     transformed in phase i + 3 (3 .. 8) by the lane that fetched it.  Slots 0-4 (pieces <= 19) are ready by the barrier that
//     opens phase 8, whose second half pre-reads the next chunk's first k-group (patch rows 0-7: pieces <= 16); slot 5 (pieces
//     20, 21: the last halo row) is first read under tap (2, 0), six phases into the next chunk.
//   * a one-tap chunk (fused 1x1 shortcut) is one phase over the same haloed image, read under its centre tap; its successor's
//     image (6 pieces per wave) is issued at the head of the phase and waited for at its end (HBM-bound phases by construction; the
//     co-resident workgroup fills).
//
// Phase P (tap TP of chunk c):
//   gap 0-2   read k-group 1 of P                        gap 3     (fused) read the patch piece of slot TP-3 + its (scale, shift)s
//   gap 4-8   pre-read k-group 0 of P+1                  gap 4-6   DMA: weights of P+3 -> ring slot (P+3)&3, patch slot TP of c+1
//   gap 7-15  (fused) transform steps, write back        end       vmcnt: all but {patch of P-1, everything of P} landed | barrier
// LDS lifetimes: ring slot (P+3)&3 = (P-1)&3 was last read in phase P-1 (k-group 1) -> free behind the barrier that opens P; the
// weights of P+1 (issued in P-2, waited for at the end of P-1) are visible behind the same barrier and pre-read in P.  The
// other patch buffer was last read in the previous chunk's phase 8 (its k-group 1; the pre-reads of that phase read THIS
// chunk's buffer) -> pieces land in it from phase 0 on.  A transform's write (phase T) is visible to the other waves behind the
// barrier that opens T+1.
//
// K order: (32-channel chunk, tap, k-group), as conv_pipe128.hip: results agree with conv_igemm.hip / conv_pipe.hip to rounding
// (fp32 summation order), not bit for bit.
#include <cstdlib>
#include <cstring>
#include "conv_epilogue.h"

namespace storm {
using namespace cidx;

namespace duo {
using namespace pipe;

constexpr int BN = 128, TH = 8;                               // output channels x pixel rows (of 32 px) per workgroup
constexpr int KC = 32, PIXB = 64;                             // channels / bytes per pixel and K-chunk
constexpr int PW = TILE_W + 2;
constexpr int NWAVES = 4, THREADS = 256;
constexpr int WAVES_M = 2, WAVES_N = 2, WM = 2, WN = 4;       // wave grid; 32-cout tiles / pixel rows per wave
constexpr int NPIX = (TH + 2) * PW;
constexpr int PXP = 1024 / PIXB;                              // pixels per 1-KiB DMA piece
constexpr int PPIECES = (NPIX + PXP - 1) / PXP;               // 22
constexpr int PATCH_BYTES = PPIECES * 1024;
constexpr int NSLOT = (PPIECES + NWAVES - 1) / NWAVES;        // haloed pieces per wave: 6 (issued in phases 0-5)
constexpr int TLAG = 3;                                       // a slot is transformed this many phases after its issue
constexpr int WPHASE = BN * WROW, RINGB = 4 * WPHASE;
constexpr int NWD = BN / 16 / NWAVES;                         // weight DMA instructions per wave and phase: 2
constexpr int WAHEAD = 3;                                     // the weight stream runs this many phases ahead
constexpr int OFF_RING = 2 * PATCH_BYTES;
constexpr int OFF_SS = OFF_RING + RINGB;
constexpr int LDS_BYTES = OFF_SS + 2 * 1024;
constexpr int WSTAGE = 32 * WM * 128;                         // epilogue staging per wave
static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(PATCH_BYTES % 256 == 0, "k-group XOR must stay inside the slot field");
static_assert(NSLOT == 6 && NSLOT - 1 + TLAG <= 8, "slot i: issued in phase i, transformed in phase i + TLAG <= 8");
static_assert(((TH - 1) * PW + TILE_W - 1) / PXP <= NWAVES * (NSLOT - 1) - 1, "the pre-read of a chunk's first k-group (patch rows 0-7) touches pieces of slots 0-4 only");
static_assert(2 * WSTAGE + WAVES_N * BN * 8 <= PATCH_BYTES && 2 * WSTAGE <= 2 * WPHASE, "epilogue staging beside the next tile's loads");

STORM_HD int p_swz(int px, int slot) { return (slot ^ ((px >> 2) & 3)) << 4; }      // (conv_pipe128.hip)

// patch / table DMA instructions a wave issues in phase tp of a nine-tap chunk (behind that phase's weight pieces)
constexpr int n_patch(int tp) { return tp == 0 ? 2 : (tp < NSLOT ? 1 : 0); }

// two packed 16-bit values -> fp32 pair
__device__ __forceinline__ f32x2 unpack2(uint32_t w, bf16_t*) { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 unpack2(uint32_t w, half_t*) { return f32x2{f16_bits_to_f32((uint16_t)(w & 0xffffu)), f16_bits_to_f32((uint16_t)(w >> 16))}; }

}  // namespace duo
using namespace duo;

// ABL: profiling-only instantiations (libstorm_hip_prof.so, STORM_CONV_ABLATE): 64 wave stamps for tools/duo_trace.py (six per tile from
// slot 8; three per phase of a workgroup's first two tiles from slot 208), + 1 no patch DMA, + 2 no weight DMA, + 4 no transform
template <typename T, int ABL = 0>
__global__ __launch_bounds__(duo::THREADS, 2)
void conv_duo_kernel(const PipeParams a, const int n_ct, const int tiles_per_xcd,
                     const int ntiles, const int tiles_x, const int tiles_per_img, const int total_vblocks) {
    typedef typename Mma<T>::Frag Frag;
    constexpr bool TRACE = (ABL & 64) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PipeArgPtr ap = pipe_args(a);                            // parameter block through the kernarg segment pointer (conv_pipe.hip)
#define STORM_RELAUNDER() relaunder(ap)

    // persistent workgroups (two per CU) walking the XCD-aware virtual block ids
    int vb = blockIdx.x;
    while (vb < total_vblocks && block_map(vb, n_ct, tiles_per_xcd).tile >= ntiles) vb += gridDim.x;
    if (vb >= total_vblocks) return;
    int tile, b, ty0, tx0, cout0;                           // the tile whose loads are being ISSUED
    auto decode = [&](int v) {
        const BlockMap bm = block_map(v, n_ct, tiles_per_xcd);
        tile = bm.tile;
        b = bm.tile / tiles_per_img;
        const int trem = bm.tile - b * tiles_per_img;
        ty0 = (trem / tiles_x) * TH;
        tx0 = (trem % tiles_x) * TILE_W;
        cout0 = bm.ct * BN;
    };
    decode(vb);
    const int imgH = pin(ap->H), imgW = pin(ap->W);

    const int tid = threadIdx.x;
    int lane = tid & 63;                                    // re-laundered at every chunk
    const int wave = uniform(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    unsigned long long* const trace_rec = TRACE && ap->trace ? ap->trace + ((long long)blockIdx.x * NWAVES + wave) * TRACE_SLOTS : nullptr;
    int tstamp = 8;                                         // (profiling) next stamp slot: six per tile from slot 8
    auto stamp = [&]() {
        if (TRACE && trace_rec && tstamp < TRACE_SLOTS) {
            const unsigned long long t = hw_memtime();
            if ((threadIdx.x & 63) == 0) trace_rec[tstamp] = t;
            ++tstamp;
        }
    };
    if (TRACE && trace_rec && (threadIdx.x & 63) == 0) trace_rec[0] = hw_ids();
    int pstamp_i = 208, ptile = 0;                          // (profiling) per-phase stamps of the first two tiles
    auto pstamp = [&]() {
        if (TRACE && trace_rec && ptile < 2 && pstamp_i < TRACE_SLOTS) {
            const unsigned long long t = hw_memtime();
            if ((threadIdx.x & 63) == 0) trace_rec[pstamp_i] = t;
            ++pstamp_i;
        }
    };

    f32x16 acc[WM][WN];

    // ---- lane constants ---------------------------------------------------------------------------------------
    const int aoff = OFF_RING + w_off(wm * WM * 32 + (lane & 31), lane >> 5);
    const int aoff1 = aoff ^ 32;                            // second k-group of a phase
    int pbase[3];                                           // this lane's pixel of ni = 0 under tap (0, dx), k-group 0, current buffer
#pragma unroll
    for (int d = 0; d < 3; ++d) pbase[d] = ((wn * WN) * PW + (lane & 31)) * PIXB + p_swz((lane & 31) + d, lane >> 5);

    // patch entry of haloed piece wave + 4 i for this lane: (pixel index << 3) | logical 16-B slot that lands in this lane's
    // physical slot, or -1 (padding / past the patch: hardware zero fill); recomputed where used (conv_pipe128.hip)
    const int prow0 = wave * PXP + (lane >> 2);
    auto patch_entry = [&](int i) -> uint32_t {
        const int row = prow0 + NWAVES * PXP * i;
        const int py = row / PW, px = row - py * PW;
        const int slot = (lane & 3) ^ ((px >> 2) & 3);
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        const bool ok = row < NPIX && gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
        return ok ? (uint32_t)(((gy * imgW + gx) << 3) | slot) : 0xffffffffu;
    };
    // the six entries of the tile being computed, filled once per tile (part B): the main loop's DMA issue / transform steps cost a
    // compare and a multiply-add per slot instead of the divide / range-test chain above (measured with the wave stamps: with the
    // chain in the stream a patch DMA cost 0.3 us of a 0.7 us phase, three times a weight piece)
    uint32_t P[NSLOT];
    uint32_t R[NWD];                                        // per-lane source offsets of this wave's weight pieces of the current run

    // ---- descriptor state.  While chunk c is computed: pd_* = chunk c+1 (patch being issued / transformed); the weight stream has
    // its own cursor WAHEAD phases ahead: wc_left taps left in its chunk wci, wn_* = the stream fields of chunk wci + 1 --------------
    u32x4 pd_srd, pd_ss_srd, w_srd;
    const int nchunks_k = pin(ap->nchunks);                 // index of the terminator descriptor
    int pd_C2 = 0, pd_cbeg2 = 0, pd_cvalid = 0, pd_ntaps = 9, pd_gn = 0;
    float pd_k = 0.f, pd_a = 0.f;                           // SiLU as y * rcp(fma(pd_a, exp2(pd_k * y), 1)): (-log2 e, 1), or (0, 0) = identity
    int wci = 0, wc_left = 9, wn_ntaps = 9, wn_wrun = 0, wn_wsoff = 0, wn_neww = 0;
    int w_soff = 0, w_tapbytes = 0;
    auto load_patch_desc = [&](int i) {                     // descriptor i -> pd_*
        const ChunkDesc& d = ap->chunk[i < nchunks_k ? i : nchunks_k];
        pd_srd = make_srd(reinterpret_cast<const char*>(d.src + (unsigned long long)b * d.bstride), d.src_bytes);
        pd_gn = d.ss != 0ull;
        pd_ss_srd = make_srd(reinterpret_cast<const char*>(pd_gn ? d.ss + (unsigned long long)b * d.ss_bstride : d.src),
                             pd_gn ? (uint32_t)d.cvalid * 8u : 0u);
        pd_C2 = d.C2; pd_cbeg2 = d.cbeg2; pd_cvalid = d.cvalid; pd_ntaps = d.ntaps;
        pd_k = d.silu ? -1.44269504088896341f : 0.f; pd_a = d.silu ? 1.f : 0.f;
    };
    auto load_wn = [&](int i) {                             // descriptor i -> wn_*
        const ChunkDesc& d = ap->chunk[i < nchunks_k ? i : nchunks_k];
        wn_ntaps = d.ntaps; wn_wrun = d.wrun; wn_wsoff = d.w_soff; wn_neww = d.new_wrun;
    };
    auto enter_wrun = [&](int r) {
        const WRunDesc& W = ap->wrun[r];
        w_srd = make_srd(reinterpret_cast<const char*>(W.w), W.bytes);
        w_tapbytes = W.tapbytes;
#pragma unroll
        for (int j = 0; j < NWD; ++j) {
            const int row = (wave * NWD + j) * 16 + (lane >> 2);
            const int co = cout0 + row;                          // rows past the matrix: zeros (never stored)
            R[j] = co < W.rows ? (uint32_t)(co * W.CinP2 + ((lane & 3) ^ ((row >> 2) & 3)) * 16) : OOB;
        }
    };
    auto w_advance = [&]() {                                // the weight stream moves on by one phase
        if (--wc_left > 0) { w_soff += w_tapbytes; return; }
        ++wci;
        if (wn_neww) enter_wrun(wn_wrun);
        w_soff = wn_wsoff; wc_left = wn_ntaps;
        load_wn(wci + 1);
    };
    int ring_rd = 0;                                        // byte offset of the ring slot of the phase being read
    auto w_issue_piece = [&](int j, int ahead) {            // piece j of the stream's phase -> the slot `ahead` phases ahead
        if (ABL & 2) return;
        char* dst = smem + OFF_RING + ((ring_rd + ahead * WPHASE) & (RINGB - 1)) + (wave * NWD + j) * 1024;
        dma16(w_srd, R[j], (uint32_t)w_soff, dst, lane);
    };
    auto ring_next = [&]() { ring_rd = (ring_rd + WPHASE) & (RINGB - 1); };

    int par = 0;                                            // patch buffer of the chunk being read (folded into pbase)
    auto issue_table = [&](int into) { if (ABL & 1) return; dma16(pd_ss_srd, (uint32_t)lane * 16u, 0u, smem + OFF_SS + into * 1024, lane); };
    auto issue_slot = [&](int i, int into, uint32_t v) {    // haloed piece wave + 4 i; v = its patch entry
        const int k = wave + NWAVES * i;
        if (ABL & 1) return;
        if (k >= PPIECES) { issue_table(into); return; }     // surplus slot: identical table bytes again (keeps the VMEM count uniform)
        const bool ok = (int)v >= 0 && (int)(v & 7u) * 8 < pd_cvalid;
        dma16(pd_srd, ok ? mad24(v >> 3, (uint32_t)pd_C2, (v & 7u) * 16u) : OOB, (uint32_t)pd_cbeg2,
              smem + into * PATCH_BYTES + k * 1024, lane);
    };
    // fused GroupNorm-apply (+ SiLU) of one slot of the chunk being fetched, in place, by the lane that fetched the unit.  Two
    // forms of the same arithmetic (gn_act_slot's: packed affine, exp2, rcp): `commit_slot` in one go (a tile's first chunk), and
    // cut into the steps t_begin / t_a / t_b / t_c / t_end that the main loop places in its MFMA gaps.
    // (registers: the slot's data is rewritten in place pair by pair, and a pair's two scales + two shifts are read one step before use)
    uint32_t t_d[4]; f32x2 t_sc, t_sh, t_y, t_e; bool t_ok = false; int t_qoff = 0, t_ssoff = 0;
    auto t_load = [&](int i) {                              // (scale, shift) of channel pair i (table layout: an octet's 8 scales, then its 8 shifts)
        t_sc = *reinterpret_cast<const f32x2*>(smem + t_ssoff + 8 * i);
        t_sh = *reinterpret_cast<const f32x2*>(smem + t_ssoff + 32 + 8 * i);
    };
    auto t_begin = [&](int i, int into, uint32_t v) {
        const int k = wave + NWAVES * i;
        t_ok = pd_gn && k < PPIECES && (int)v >= 0 && (int)(v & 7u) * 8 < pd_cvalid;
        t_qoff = into * PATCH_BYTES + (k < PPIECES ? k : 0) * 1024 + lane * 16;
        t_ssoff = OFF_SS + into * 1024 + (int)(v & 3u) * 64;
        const uint4 q = *reinterpret_cast<const uint4*>(smem + t_qoff);
        t_d[0] = q.x; t_d[1] = q.y; t_d[2] = q.z; t_d[3] = q.w;
        t_load(0);
    };
    auto t_a = [&](int i) {                                 // channel pair i: unpack, affine, exponent scaling
        t_y = __builtin_elementwise_fma(unpack2(t_d[i], (T*)nullptr), t_sc, t_sh);
        t_e = t_y * f32x2{pd_k, pd_k};
        keep_rw(t_e);               // (every step's result is pinned where it is computed: only the final store is conditional, and the
    };                              //  optimiser otherwise sinks the whole transform into that branch, behind the phase's last MFMA)
    auto t_b = [&]() { t_e = __builtin_elementwise_fma(f32x2{pd_a, pd_a}, f32x2{hw_exp2(t_e.x), hw_exp2(t_e.y)}, f32x2{1.0f, 1.0f}); keep_rw(t_e); };
    auto t_c = [&](int i) {
        const f32x2 r = t_y * f32x2{hw_rcp(t_e.x), hw_rcp(t_e.y)};
        t_d[i] = pack2(r.x, r.y, (T*)nullptr);
        keep_rw(t_d[i]);
    };
    auto t_end = [&]() { if (t_ok) *reinterpret_cast<uint4*>(smem + t_qoff) = make_uint4(t_d[0], t_d[1], t_d[2], t_d[3]); };
    auto commit_slot = [&](int i, int into) {
        t_begin(i, into, P[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { t_a(j); if (j < 3) t_load(j + 1); t_b(); t_c(j); }
        t_end();
    };

    // ---- fragment reads / MFMAs (as conv_pipe128.hip) ---------------------------------------------------------------
    auto read_frags = [&](Frag (&fa)[WM], Frag (&fb)[WN], int ring, int pb, auto kg_, auto poff_, auto prow_) {
        constexpr int kg = decltype(kg_)::value, POFF = decltype(poff_)::value, PROW = decltype(prow_)::value;
        const char* wb = smem + ring + (kg ? aoff1 : aoff);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const Frag*>(wb + mi * 32 * WROW);
        const char* pp = smem + (pb ^ (kg << 5)) + POFF;
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const Frag*>(pp + ni * PROW);
    };
    auto read_a = [&](Frag& f, int ring, auto kg_, auto mi_) {
        constexpr int kg = decltype(kg_)::value, mi = decltype(mi_)::value;
        f = *reinterpret_cast<const Frag*>(smem + ring + (kg ? aoff1 : aoff) + mi * 32 * WROW);
    };
    auto read_b = [&](Frag& f, int pb, auto kg_, auto poff_, auto prow_, auto ni_) {
        constexpr int kg = decltype(kg_)::value, POFF = decltype(poff_)::value, PROW = decltype(prow_)::value, ni = decltype(ni_)::value;
        f = *reinterpret_cast<const Frag*>(smem + (pb ^ (kg << 5)) + POFF + ni * PROW);
    };
    auto mma1 = [&](const Frag (&fa)[WM], const Frag (&fb)[WN], int i) { Mma<T>::run(fa[i / WN], fb[i % WN], acc[i / WN][i % WN]); };
    Frag fa0[WM], fb0[WN], fa1[WM], fb1[WN];
    typedef IC<PW * PIXB> Prow9;
#define STORM_SB() __builtin_amdgcn_sched_barrier(0)

    // ---- one phase: tap TP of a nine-tap chunk ---------------------------------------------------------------------------
    auto phase9 = [&](auto t_) {
        constexpr int TP = decltype(t_)::value;
        constexpr int DX = TP % 3, POFF = ((TP / 3) * PW + DX) * PIXB;
        constexpr int DXN = (TP + 1) % 3, POFFN = (((TP + 1) / 3) * PW + DXN) * PIXB;       // the next tap of the chunk
        typedef IC<POFF> Poff; typedef IC<POFFN> PoffN; typedef IC<1> K1; typedef IC<0> K0;
        constexpr bool TR = TP >= TLAG && !(ABL & 4);                      // a slot is transformed in this phase (the arithmetic always runs - ONE loop
                                                             // body: a second copy gave the accumulators two homes and 600 spilled registers -
                                                             // and only a chunk with a fused operand writes its result back)
        constexpr int TS = TP - TLAG;
        const int pb = pbase[DX];
        const int into = par ^ 1;
        const int rn = (ring_rd + WPHASE) & (RINGB - 1);
        auto next_a = [&](auto mi_) {
            constexpr int mi = decltype(mi_)::value;
            if constexpr (TP < 8) read_a(fa0[mi], rn, K0{}, mi_);
            else { if (pd_ntaps == 9) read_a(fa0[mi], rn, K0{}, mi_); }
        };
        auto next_b = [&](auto ni_) {
            if constexpr (TP < 8) read_b(fb0[decltype(ni_)::value], pbase[DXN], K0{}, PoffN{}, Prow9{}, ni_);
            else { if (pd_ntaps == 9) read_b(fb0[decltype(ni_)::value], pbase[0] + (par ? -PATCH_BYTES : PATCH_BYTES), K0{}, IC<0>{}, Prow9{}, ni_); }
        };
        STORM_SB();
        prio(1);
        mma1(fa0, fb0, 0); read_a(fa1[0], ring_rd, K1{}, IC<0>{}); read_b(fb1[0], pb, K1{}, Poff{}, Prow9{}, IC<0>{}); STORM_SB();
        mma1(fa0, fb0, 1); read_b(fb1[1], pb, K1{}, Poff{}, Prow9{}, IC<1>{}); read_b(fb1[2], pb, K1{}, Poff{}, Prow9{}, IC<2>{}); STORM_SB();
        mma1(fa0, fb0, 2); read_b(fb1[3], pb, K1{}, Poff{}, Prow9{}, IC<3>{}); read_a(fa1[1], ring_rd, K1{}, IC<1>{}); STORM_SB();
        mma1(fa0, fb0, 3); if constexpr (TR) t_begin(TS, into, P[TS]); STORM_SB();
        mma1(fa0, fb0, 4); next_a(IC<0>{}); w_issue_piece(0, WAHEAD); STORM_SB();        // fa0[0]: last used by MFMA 3
        mma1(fa0, fb0, 5); next_b(IC<0>{}); w_issue_piece(1, WAHEAD); STORM_SB();        // fb0[0]: last used by MFMA 4
        mma1(fa0, fb0, 6); next_b(IC<1>{}); if constexpr (TP == 0) issue_table(into); if constexpr (TP < NSLOT) issue_slot(TP, into, P[TP]); STORM_SB();
        mma1(fa0, fb0, 7); next_b(IC<2>{}); if constexpr (TR) { t_a(0); t_load(1); } STORM_SB();
        mma1(fa1, fb1, 0); next_b(IC<3>{}); next_a(IC<1>{}); if constexpr (TR) t_b(); STORM_SB();   // fb0[3], fa0[1]: last used by MFMA 7
        mma1(fa1, fb1, 1); if constexpr (TR) { t_c(0); t_a(1); t_load(2); } STORM_SB();
        mma1(fa1, fb1, 2); if constexpr (TR) t_b(); STORM_SB();
        mma1(fa1, fb1, 3); if constexpr (TR) { t_c(1); t_a(2); t_load(3); } STORM_SB();
        mma1(fa1, fb1, 4); if constexpr (TR) t_b(); STORM_SB();
        mma1(fa1, fb1, 5); if constexpr (TR) { t_c(2); t_a(3); } STORM_SB();
        mma1(fa1, fb1, 6); if constexpr (TR) t_b(); STORM_SB();
        mma1(fa1, fb1, 7); if constexpr (TR) { t_c(3); t_end(); } STORM_SB();
        // everything up to the previous phase's weight pieces has landed: behind them this wave issued the previous phase's patch
        // DMAs and all of this phase's
        pstamp();
        vm_wait<(ABL & 3) ? 0 : n_patch(TP == 0 ? 8 : TP - 1) + NWD + n_patch(TP)>();
        pstamp();
        prio(0);
        raw_barrier();
        pstamp();
        STORM_SB();
        ring_next();
        w_advance();
    };
    // ---- the phase of a one-tap chunk (fused 1x1 shortcut): the SAME haloed image, read under its centre tap only (a third more
    // pixels fetched than a compact image would need, but one layout, one issue path and no layout branches in the nine-tap
    // stream); the whole patch of the next chunk (six slots per wave) is issued at the head of the phase ----------------------
    auto phase1 = [&]() {
        typedef IC<1> K1; typedef IC<0> K0; typedef IC<(PW + 1) * PIXB> Poff;
        const int pb = pbase[1];
        const int into = par ^ 1;
        read_frags(fa0, fb0, ring_rd, pb, K0{}, Poff{}, Prow9{});        // (not pre-read: the image has only just landed)
        issue_table(into);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) issue_slot(i, into, P[i]);
        STORM_SB();
        prio(1);
        mma1(fa0, fb0, 0); read_a(fa1[0], ring_rd, K1{}, IC<0>{}); read_b(fb1[0], pb, K1{}, Poff{}, Prow9{}, IC<0>{}); STORM_SB();
        mma1(fa0, fb0, 1); read_b(fb1[1], pb, K1{}, Poff{}, Prow9{}, IC<1>{}); read_b(fb1[2], pb, K1{}, Poff{}, Prow9{}, IC<2>{}); STORM_SB();
        mma1(fa0, fb0, 2); read_b(fb1[3], pb, K1{}, Poff{}, Prow9{}, IC<3>{}); read_a(fa1[1], ring_rd, K1{}, IC<1>{}); STORM_SB();
        mma1(fa0, fb0, 3); w_issue_piece(0, WAHEAD); STORM_SB();
        mma1(fa0, fb0, 4); w_issue_piece(1, WAHEAD); STORM_SB();
        mma1(fa0, fb0, 5); mma1(fa0, fb0, 6); mma1(fa0, fb0, 7); STORM_SB();
#pragma unroll
        for (int i = 0; i < WM * WN; ++i) mma1(fa1, fb1, i);
        STORM_SB();
        vm_wait<(ABL & 3) ? 0 : NWD>();                      // all but this phase's weight pieces: the next image has landed
        prio(0);
        raw_barrier();
        STORM_SB();
        ring_next();
        w_advance();
    };

    // ---- tile start, part A: chunk 0's patch -> buffer 0 / table 0, the first two taps' weights -> ring slots 0, 1: regions
    // the previous tile's epilogue staging does not touch
    auto tile_issue = [&]() {
        load_patch_desc(0);
        wci = 0; wc_left = 9;                                    // (the first chunk has nine taps)
        load_wn(0);
        enter_wrun(wn_wrun);
        w_soff = wn_wsoff;
        load_wn(1);
        ring_rd = 0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int j = 0; j < NWD; ++j) w_issue_piece(j, p);
            w_advance();
        }
        issue_table(0);
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) issue_slot(i, 0, patch_entry(i));
    };
    const int nchunks = pin(ap->nchunks), n9 = pin(ap->nchunks9);
    auto chunk_change = [&](int ci) {                       // chunk ci is done: the fetched buffer becomes current; descriptor of ci + 2
        par ^= 1;
        const int dlt = par ? PATCH_BYTES : -PATCH_BYTES;
#pragma unroll
        for (int d = 0; d < 3; ++d) pbase[d] += dlt;
        load_patch_desc(ci + 2);
        launder(lane);
    };
    tile_issue();
    while (true) {
        // ---- tile start, part B: everything issued has landed; fused GroupNorm transform of the first patch (all four waves); the
        // third tap's weights -> ring slot 2 (free now: the previous tile's epilogue staged there) ----------------------------------
        STORM_RELAUNDER();
        stamp();                                            // 0: tile start (part B)
        vm_wait<0>();
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) P[i] = patch_entry(i);
        if (pd_gn) {
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) commit_slot(i, 0);
        }
        load_patch_desc(1);
#pragma unroll
        for (int j = 0; j < NWD; ++j) w_issue_piece(j, 2);
        w_advance();
        raw_barrier();
        stamp();                                            // 1: chunk 0 landed and transformed, main loop begins
        read_frags(fa0, fb0, 0, pbase[0], IC<0>{}, IC<0>{}, Prow9{});   // first k-group of phase 0
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

        // ---- main loops: the nine-tap chunks, then the one-tap chunks of a fused 1x1 shortcut ---------------------------
        int ci = 0;
        for (; ci < n9; ++ci) {
            static_for<9>([&](auto t) { phase9(t); });
            chunk_change(ci);
        }
        for (; ci < nchunks; ++ci) {
            phase1();
            chunk_change(ci);
        }
        stamp();                                            // 2: main loop done
        if (TRACE) ++ptile;

        // ---- hand-over: this tile's coordinates go to the epilogue; the next tile's first loads are issued -----------------
        vm_wait<0>();                                       // trailing (zero-fill) patch / ring loads landed ...
        raw_barrier();                                      // ... and every wave is done reading: all of LDS is free
        const epi::TileAt et = {tile, b, ty0, tx0, cout0};
        STORM_RELAUNDER();
        int nvb = vb + gridDim.x;
        while (nvb < total_vblocks && block_map(nvb, n_ct, tiles_per_xcd).tile >= ntiles) nvb += gridDim.x;
        const bool has_next = nvb < total_vblocks;
        if (par) {                                          // the next tile starts in patch buffer 0 / ring slot 0
            par = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) pbase[d] -= PATCH_BYTES;
        }
        if (has_next) {
            vb = nvb;
            decode(vb);
            launder(lane);
            tile_issue();
        }
        STORM_RELAUNDER();
        stamp();                                            // 3: hand-over done (drain, barrier, next tile's first loads issued)

        // ---- epilogue (conv_epilogue.h).  Staging lives in patch buffer 1 (waves 0, 1; the statistics scratch behind them) and ring
        // slots 2, 3 (waves 2, 3): the next tile's first loads are landing in buffer 0 / table 0 / ring slots 0, 1 meanwhile.
        char* const stage = smem + (wave < 2 ? PATCH_BYTES + wave * WSTAGE : OFF_RING + 2 * WPHASE + (wave - 2) * WSTAGE);
        float gsum[8], gsq[8];
        epi::store_tile<T, WM, WN>(acc, stage, ap, et, wm, wn, lane, imgH, imgW, BN, TH, gsum, gsq);
        stamp();                                            // 4: epilogue stores issued
        if (ap->gn_part != nullptr)
            epi::write_stats<WM, WN, WAVES_N, BN, TH>(gsum, gsq, reinterpret_cast<float*>(smem + PATCH_BYTES + 2 * WSTAGE), ap, et, wm, wn, lane, tid,
                                                      imgH, tiles_x, tiles_per_img);
        stamp();                                            // 5: statistics written
        if (!has_next) break;
        __syncthreads();                                    // the statistics scratch / staging of this tile is free again
    }
}

#undef STORM_RELAUNDER
#undef STORM_SB

// ---- host side ---------------------------------------------------------------------------------------------------
bool conv_duo_supports(const storm_conv_args& a) {
    PipeParams p;
    return build_pipe_params(a, p, duo::KC);
}

template <typename T, int ABL = 0>
static int launch_duo(const storm_conv_args& a, hipStream_t st) {
    auto kern = conv_duo_kernel<T, ABL>;
    static bool attr_set = false;                       // per instantiation; benign race (idempotent)
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, duo::LDS_BYTES));
        attr_set = true;
    }
    PipeParams prm;
    STORM_CHECK(build_pipe_params(a, prm, duo::KC), "storm_conv: convolution outside the two-workgroup pipelined kernel's coverage");
    if (ABL & 64) prm.trace = reinterpret_cast<unsigned long long*>(switches().conv_trace_ptr);
    const int tiles_x = cdiv(a.W, TILE_W);
    const int tiles_per_img = tiles_x * cdiv(a.H, duo::TH);
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int n_ct = cdiv(a.outC, duo::BN);                            // (both cout tiles of a pixel tile on one XCD: block_map)
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long vblocks = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(vblocks > 0 && vblocks < (1LL << 31), "storm_conv: grid %lld out of range", vblocks);
    const long long resident = 2LL * ((device_cus() + 7) / 8 * 8);     // two workgroups per CU; a multiple of 8
    const long long grid = vblocks < resident ? vblocks : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(duo::THREADS), duo::LDS_BYTES, st, prm, n_ct, tiles_per_xcd, (int)ntiles,
                       tiles_x, tiles_per_img, (int)vblocks);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_conv_duo(const storm_conv_args& a, hipStream_t st) {
#if defined(STORM_PROFILING)
    if (a.dtype == STORM_BF16) switch (switches().conv_ablate) {
        case 64: return launch_duo<bf16_t, 64>(a, st);
        case 65: return launch_duo<bf16_t, 65>(a, st);       // + no patch DMA
        case 66: return launch_duo<bf16_t, 66>(a, st);       // + no weight DMA
        case 68: return launch_duo<bf16_t, 68>(a, st);       // + no transform
        case 67: return launch_duo<bf16_t, 67>(a, st);       // + no DMA of either kind
        default: break;
    }
#endif
    return a.dtype == STORM_F16 ? launch_duo<half_t>(a, st) : launch_duo<bf16_t>(a, st);
}

const char* conv_duo_kernel_name(int dtype) {
    return dtype == STORM_F16 ? "storm::conv_duo_kernel<storm::half_t, 0>" : "storm::conv_duo_kernel<storm::bf16_t, 0>";
}

}  // namespace storm
