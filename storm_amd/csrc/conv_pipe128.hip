// Software-pipelined 3x3 implicit-GEMM convolution for layers with <= 128 output channels (bf16 / fp16 operands): the
// full-resolution 3x3 layers of NCSN++ (layers.py:119-126 ddpm_conv3x3 with the fused pieces of layerspp.py:242-274 listed
// in include/storm_hip.h), which the 256-cout tile of conv_pipe.hip would half fill.
//
// Same math, arguments, epilogue and instruction-stream discipline as conv_pipe.hip (read its header first: chunk
// descriptors, compile-time-unrolled taps, LDS-DMA only, role-specialised ping-pong groups, one fragment read per MFMA
// gap).  What differs follows from the tile, 128 output channels x (16 x 32) pixels:
//   * every wave still owns 64 couts x (4 x 32) pixels = 8 accumulator tiles (wave grid 2 x 4: the leading group
//     computes pixel rows 0-7, the lagging group rows 8-15), so the MFMA : LDS-read ratio is unchanged (8 : 6);
//   * the haloed patch is 18 x 34 pixels, twice the 256-cout kernel's: a K-chunk is therefore 32 channels (64 B per
//     pixel, 39 KiB per patch), one PHASE = one tap (two k-groups, 16 MFMAs per wave), a nine-tap chunk = 9 phases;
//   * patches are TRIPLE buffered: while chunk c is read, chunk c+2 is fetched and chunk c+1 is waited for and
//     transformed (fused GroupNorm-apply + SiLU), one 1-KiB piece per lagging wave and phase.  A piece is waited for
//     exactly one chunk period (nine phases, ~10^4 cycles) after its issue - the VMEM sequence of a lagging wave is
//     periodic (table, slot 0, slot 1 | slot 2 | ... | slot 9: NVM = 11 per period), so the wait is vmcnt(NVM) in
//     every phase;
//   * weights (128 rows x 64 B per phase) stream through the same 4-slot ring, issued two phases (= two taps) ahead
//     by the leading waves, 2 pieces per wave and phase;
//   * a one-tap chunk (fused 1x1 shortcut) is one phase over a compact 16 x 32 image; its successor's image (8 pieces per
//     lagging wave) is issued two chunks ahead and waited for one phase later.  Those phases are HBM-bound by
//     construction (128 FLOP per input byte).
//
// Interval numbering as conv_pipe.hip: leading S(P) = 2P, C(P) = 2P+1; lagging one later.  LDS lifetimes, relative to the
// first interval 2P0 of chunk c:  buffer (c+2)%3 was last read by the lagging C of chunk c-1's last phase (2P0); the
// lagging group issues into it from its S(P0) (2P0+1).  Chunk c+1's slot j (pieces lw + 4j) is waited for and transformed in
// the lagging S(P0 + max(j-1, 0)) (<= 2P0+17); its first readers are the pre-reads of chunk c+1's first k-group in C(P0+8):
// leading (2P0+17) pixel rows 0-7 = pieces <= 16 = slots <= 4 (transformed by 2P0+7), lagging (2P0+18) rows 8-15 = pieces
// <= 34 = slots <= 8 (2P0+15).  Slot 9 (pieces 36-38: pixel row 16, columns 32-33, and row 17) is first read under tap (1, 1),
// four phases into chunk c+1.
//
// K order: (32-channel chunk, tap, k-group) - the fp32 accumulation order differs from conv_igemm.hip / conv_pipe.hip
// ((64-channel chunk, tap, k-group)), so outputs agree with those kernels to rounding, not bit for bit.
#include <cstdlib>
#include <cstring>
#include "conv_epilogue.h"

namespace storm {
using namespace cidx;

namespace pipe128 {
using namespace pipe;

constexpr int BN = 128, TH = 16;                              // output channels x pixel rows (of 32 px) per workgroup
constexpr int KC = 32, PIXB = 64;                             // channels / bytes per pixel and K-chunk
constexpr int PW = TILE_W + 2;
constexpr int NWAVES = 8, THREADS = 512;
constexpr int WAVES_M = 2, WAVES_N = 4, WM = 2, WN = 4;       // wave grid; 32-cout tiles / pixel rows per wave
constexpr int NPIX = (TH + 2) * PW;
constexpr int PXP = 1024 / PIXB;                              // pixels per 1-KiB DMA piece
constexpr int PPIECES = (NPIX + PXP - 1) / PXP;               // 39
constexpr int PATCH_BYTES = PPIECES * 1024;
constexpr int NBUF = 3;
constexpr int CPIECES = TH * TILE_W / PXP;                    // pieces of a compact (one-tap) image: 32
constexpr int NLAG = 4;                                       // patch-fetching (lagging) waves
constexpr int NSLOT = (PPIECES + NLAG - 1) / NLAG;            // haloed pieces per lagging wave: 10
constexpr int NSLOT1 = CPIECES / NLAG;                        // compact pieces per lagging wave: 8
constexpr int NVM = NSLOT + 1;                                // VMEM instructions of a lagging wave per nine-tap chunk period
constexpr int WPHASE = BN * WROW, RINGB = 4 * WPHASE;
constexpr int NWD = BN / 16 / 4;                              // weight DMA instructions per leading wave and phase
constexpr int OFF_RING = NBUF * PATCH_BYTES;
constexpr int OFF_SS = OFF_RING + RINGB;
constexpr int LDS_BYTES = OFF_SS + NBUF * 1024;
constexpr int PR = 1;
constexpr int WSTAGE = 32 * PR * WM * 128;                    // epilogue staging per wave
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(PATCH_BYTES % 256 == 0, "k-group XOR must stay inside the slot field");
static_assert(NSLOT * NLAG - PPIECES < NLAG && NSLOT == 10, "slot schedule: two slots in phase 0, one in each of phases 1-8");
static_assert(CPIECES % NLAG == 0 && NSLOT1 <= NSLOT, "compact pieces fit the same slots");
static_assert(NWAVES * WSTAGE <= (NBUF - 1) * PATCH_BYTES && WAVES_N * BN * 8 <= 2 * WPHASE, "epilogue staging beside the next tile's loads");

// Patch image: 64 B per pixel; the four 16-B slots are XOR-swizzled by the pixel COLUMN ((px >> 2) & 3): 16 consecutive
// px of a fragment read hit 16 distinct bank groups, and taps / pixel rows / buffers are plain additions.
STORM_HD int p_swz(int px, int slot) { return (slot ^ ((px >> 2) & 3)) << 4; }

}  // namespace pipe128
using namespace pipe128;

// ABL: profiling-only instantiations (libstorm_hip_prof.so, STORM_CONV_ABLATE; the product library holds ABL = 0 only).  64: per-tile wave
// stamps for tools/pipe128_trace.py; the work-skipping ones use conv_pipe's codes (profiles/r06_power_ablations_128cout.txt): 8 no weight
// DMA, 16 no fragment reads (operands stay what the registers hold), 128 no patch DMA / table / fused GroupNorm transform, 136 no DMA of
// either kind, 256 patch DMA from one hot 1-KiB region (same instructions and LDS writes, no memory-system traffic), 1024 no epilogue (accumulators kept alive, nothing staged, stored or reduced).  Results of those are garbage by design.
template <typename T, int ABL = 0>
__global__ __launch_bounds__(pipe128::THREADS, 2)
void conv_pipe128_kernel(const PipeParams a, const int n_ct, const int tiles_per_xcd,
                         const int ntiles, const int tiles_x, const int tiles_per_img, const int total_vblocks) {
    typedef typename Mma<T>::Frag Frag;
    constexpr bool TRACE = (ABL & 64) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // parameter block through the kernarg segment pointer, re-laundered per tile (see conv_pipe.hip)
    PipeArgPtr ap = pipe_args(a);
#define STORM_RELAUNDER() relaunder(ap)

    // persistent workgroups (at most one per CU) walking the XCD-aware virtual block ids
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
    const int grp = wave >> 2;                              // 0: leading (pixel rows 0-7), 1: lagging (rows 8-15), one interval behind
    const int lw = wave & 3;
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

    f32x16 acc[WM][WN];

    // ---- lane constants ---------------------------------------------------------------------------------------
    const int aoff = OFF_RING + w_off(wm * WM * 32 + (lane & 31), lane >> 5);
    const int aoff1 = aoff ^ 32;                            // second k-group of a phase
    int pbase[3];                                           // this lane's pixel of ni = 0 under tap (0, dx), k-group 0, current buffer
#pragma unroll
    for (int d = 0; d < 3; ++d) pbase[d] = ((wn * WN) * PW + (lane & 31)) * PIXB + p_swz((lane & 31) + d, lane >> 5);
    const int cdelta = wn * WN * (PW - TILE_W) * PIXB;      // haloed row index - compact row index of this wave's pixels

    // ---- patch entry of haloed piece lw + 4 i for this lane: (pixel index << 3) | logical 16-B slot that lands in this lane's
    // physical slot, or -1 (padding / past the patch: hardware zero fill).  Recomputed where it is used (a dozen VALU
    // operations against the ~500 issue cycles of a transform): a per-slot table would cost ten registers this kernel
    // does not have (it went to scratch, whose loads count in vmcnt and stalled every counted wait).
    const int prow0 = lw * PXP + (lane >> 2);               // patch pixel of slot 0
    auto patch_entry = [&](int i) -> uint32_t {
        const int row = prow0 + NLAG * PXP * i;
        const int py = row / PW, px = row - py * PW;
        const int slot = (lane & 3) ^ ((px >> 2) & 3);
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        const bool ok = row < NPIX && gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
        return ok ? (uint32_t)(((gy * imgW + gx) << 3) | slot) : 0xffffffffu;
    };
    uint32_t R[NWD];                                        // (leading waves) per-lane source offsets of the weight pieces of the current run

    // ---- descriptor state.  While chunk c is computed:  pd_* = chunk c+2 (patch being ISSUED), cm_* = chunk c+1 (patch being
    // waited for / transformed), w1_* / w2_* = weight stream positions of chunks c+1 / c+2 --------------------------------
    u32x4 pd_srd, pd_ss_srd, w_srd;
    const int nchunks_k = pin(ap->nchunks);                 // index of the terminator descriptor
    int pd_C2 = 0, pd_cbeg2 = 0, pd_cvalid = 0, pd_ntaps = 9, pd_gn = 0, pd_silu = 0;
    int cm_cvalid = 0, cm_gn = 0, cm_silu = 0;
    int w1_neww = 0, w1_wrun = 0, w1_wsoff = 0, w1_ntaps = 9, w2_neww = 0, w2_wrun = 0, w2_wsoff = 0;
    int w_soff = 0, w_tapbytes = 0;
    auto load_next = [&](int i) {                           // shift, then descriptor i -> pd_*, w2_*
        cm_cvalid = pd_cvalid; cm_gn = pd_gn; cm_silu = pd_silu;
        w1_neww = w2_neww; w1_wrun = w2_wrun; w1_wsoff = w2_wsoff; w1_ntaps = pd_ntaps;
        const ChunkDesc& d = ap->chunk[i < nchunks_k ? i : nchunks_k];
        pd_srd = make_srd(reinterpret_cast<const char*>(d.src + (unsigned long long)b * d.bstride), d.src_bytes);
        pd_gn = d.ss != 0ull;
        pd_ss_srd = make_srd(reinterpret_cast<const char*>(pd_gn ? d.ss + (unsigned long long)b * d.ss_bstride : d.src),
                             pd_gn ? (uint32_t)d.cvalid * 8u : 0u);
        pd_C2 = d.C2; pd_cbeg2 = d.cbeg2; pd_cvalid = d.cvalid; pd_ntaps = d.ntaps; pd_silu = d.silu;
        w2_wrun = d.wrun; w2_wsoff = d.w_soff; w2_neww = d.new_wrun;
    };
    auto enter_wrun = [&](int r) {                          // (leading waves: R = weight piece offsets of run r)
        const WRunDesc& W = ap->wrun[r];
        w_srd = make_srd(reinterpret_cast<const char*>(W.w), W.bytes);
        w_tapbytes = W.tapbytes;
        if (grp == 0) {
#pragma unroll
            for (int j = 0; j < NWD; ++j) {
                const int row = (lw * NWD + j) * 16 + (lane >> 2);
                const int co = cout0 + row;                      // rows past the matrix: zeros (never stored)
                R[j] = co < W.rows ? (uint32_t)(co * W.CinP2 + ((lane & 3) ^ ((row >> 2) & 3)) * 16) : OOB;
            }
        }
    };
    int ring_rd = 0;                                        // byte offset of the ring slot of the phase being read
    auto w_issue = [&]() {                                  // (leading) the stream's tap -> the slot two phases ahead
        if (ABL & 8) return;
        char* dst = smem + OFF_RING + (ring_rd ^ (2 * WPHASE)) + lw * (NWD * 1024);
#pragma unroll
        for (int j = 0; j < NWD; ++j) dma16(w_srd, R[j], (uint32_t)w_soff, dst + j * 1024, lane);
    };
    auto ring_next = [&]() { ring_rd = (ring_rd + WPHASE) & (RINGB - 1); };

    // patch buffers: rd = chunk being read (its offset is folded into pbase), cm = being waited for, is = being issued
    int rd_buf = 0, cm_buf = 1, is_buf = 2;
    // (lagging) the (scale, shift) table of the chunk being issued; every lagging wave fetches its own copy (identical
    // bytes), so that its own vmcnt orders it before its transforms
    auto issue_table = [&](int into) { if (ABL & 128) return; dma16(pd_ss_srd, (uint32_t)lane * 16u, 0u, smem + OFF_SS + into * 1024, lane); };
    auto issue_slot = [&](int i, int into) {                // haloed piece lw + 4 i
        if (ABL & 128) return;
        const int k = lw + NLAG * i;
        if (k >= PPIECES) { issue_table(into); return; }     // surplus slot: identical table bytes again (keeps the VMEM count uniform)
        const uint32_t v = patch_entry(i);
        const bool ok = (int)v >= 0 && (int)(v & 7u) * 8 < pd_cvalid;
        if (ABL & 256) {                                    // (profiling) the same DMA instructions and LDS writes from ONE hot 1-KiB region: no HBM / L2-miss traffic
            dma16(pd_srd, ok ? (uint32_t)lane * 16u : OOB, 0u, smem + into * PATCH_BYTES + k * 1024, lane);
            return;
        }
        dma16(pd_srd, ok ? mad24(v >> 3, (uint32_t)pd_C2, (v & 7u) * 16u) : OOB, (uint32_t)pd_cbeg2,
              smem + into * PATCH_BYTES + k * 1024, lane);
    };
    // one compact piece (one-tap chunk: TH x 32 pixels, no halo): piece k = pixel row k >> 1, columns 16 (k & 1) ..
    auto issue_compact = [&](int k, int into) {
        if (ABL & 128) return;
        const int trow = k >> 1, n = (k & 1) * PXP + (lane >> 2);
        const int slot = (lane & 3) ^ ((n >> 2) & 3);
        const int gy = ty0 + trow, gx = tx0 + n;
        const bool ok = gy < imgH && gx < imgW && slot * 8 < pd_cvalid;
        dma16(pd_srd, ok ? mad24((uint32_t)(gy * imgW + gx), (uint32_t)pd_C2, (uint32_t)slot * 16u) : OOB, (uint32_t)pd_cbeg2,
              smem + into * PATCH_BYTES + k * 1024, lane);
    };
    auto issue_any = [&](int i, int into) {                 // slot i in the layout of the chunk being issued
        if (pd_ntaps == 9) issue_slot(i, into);
        else if (i < NSLOT1) issue_compact(lw + NLAG * i, into);
        else issue_table(into);
    };
    // fused GroupNorm-apply (+ SiLU) of the chunk being waited for: in place, by the lane that fetched the unit, in the staging
    // interval that waits for it.  (Measured on MI355X, 256 -> 128 @ 256 x 512 x 16: the transform costs its stand-alone VALU
    // time - 2.68 M against 1.88 M cycles per launch - whoever runs it: sharing it slot-wise or half-slot-wise with the leading
    // waves, or dropping s_setprio, moved the launch time by < 2 %.  So it stays where it is simplest.)
    auto commit_slot = [&](int i, int into) {
        if (ABL & 128) return;
        const int k = lw + NLAG * i;
        if (k < PPIECES) {
            const uint32_t v = patch_entry(i);
            if ((int)v >= 0 && (int)(v & 7u) * 8 < cm_cvalid) {
                uint4* const q = reinterpret_cast<uint4*>(smem + into * PATCH_BYTES + k * 1024 + lane * 16);
                float ss[16];
                load_ss<8>(reinterpret_cast<const float*>(smem + OFF_SS + into * 1024), (int)(v & 7u), ss);
                *q = gn_act_slot(*q, ss, cm_silu, (T*)nullptr);
            }
        }
    };

    // ---- fragment reads / MFMAs --------------------------------------------------------------------------------
    auto read_frags = [&](Frag (&fa)[WM], Frag (&fb)[WN], int ring, int pb, auto kg_, auto poff_, auto prow_) {
        constexpr int kg = decltype(kg_)::value, POFF = decltype(poff_)::value, PROW = decltype(prow_)::value;
        if (ABL & 16) {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) keep_rw(fa[mi]);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) keep_rw(fb[ni]);
            return;
        }
        const char* wb = smem + ring + (kg ? aoff1 : aoff);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const Frag*>(wb + mi * 32 * WROW);
        const char* pp = smem + (pb ^ (kg << 5)) + POFF;
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const Frag*>(pp + ni * PROW);
    };
    auto read_a = [&](Frag& f, int ring, auto kg_, auto mi_) {
        constexpr int kg = decltype(kg_)::value, mi = decltype(mi_)::value;
        if (ABL & 16) { keep_rw(f); return; }
        f = *reinterpret_cast<const Frag*>(smem + ring + (kg ? aoff1 : aoff) + mi * 32 * WROW);
    };
    auto read_b = [&](Frag& f, int pb, auto kg_, auto poff_, auto prow_, auto ni_) {
        constexpr int kg = decltype(kg_)::value, POFF = decltype(poff_)::value, PROW = decltype(prow_)::value, ni = decltype(ni_)::value;
        if (ABL & 16) { keep_rw(f); return; }
        f = *reinterpret_cast<const Frag*>(smem + (pb ^ (kg << 5)) + POFF + ni * PROW);
    };
    auto mma_part = [&](const Frag (&fa)[WM], const Frag (&fb)[WN], int lo, int hi) {
#pragma unroll
        for (int i = 0; i < WM * WN; ++i)
            if (i >= lo && i < hi) Mma<T>::run(fa[i / WN], fb[i % WN], acc[i / WN][i % WN]);
    };
    Frag fa0[WM], fb0[WN], fa1[WM], fb1[WN];
    typedef IC<PW * PIXB> Prow9; typedef IC<TILE_W * PIXB> Prow1;

    // ---- one phase: tap TP of a chunk with NT taps ---------------------------------------------------------------
    auto phase = [&](auto t_, auto nt_) {
        constexpr int TP = decltype(t_)::value, NT = decltype(nt_)::value;
        constexpr int DX = NT == 9 ? TP % 3 : 0;
        constexpr int POFF = NT == 9 ? ((TP / 3) * PW + DX) * PIXB : 0;
        constexpr int DXN = NT == 9 ? (TP + 1) % 3 : 0;                          // the next tap of the chunk
        constexpr int POFFN = NT == 9 ? (((TP + 1) / 3) * PW + DXN) * PIXB : 0;
        typedef IC<POFF> Poff; typedef IC<POFFN> PoffN;
        typedef std::conditional_t<NT == 9, Prow9, Prow1> Prow;
        const int pb = NT == 9 ? pbase[DX] : pbase[0] - cdelta;
        // the weight stream moves to the phase after the next one
        if constexpr (NT == 9 && TP <= 6) w_soff += w_tapbytes;
        else if constexpr (NT == 9 && TP == 7) { if (w1_neww) enter_wrun(w1_wrun); w_soff = w1_wsoff; }
        else if (NT == 9 && w1_ntaps == 9) w_soff += w_tapbytes;
        else { if (w2_neww) enter_wrun(w2_wrun); w_soff = w2_wsoff; }
        // ================= S =================
        if constexpr (NT == 1) read_frags(fa0, fb0, ring_rd, pb, IC<0>{}, Poff{}, Prow{});   // (a compact image is never pre-read)
        if (grp == 0) {
            w_issue();
            vm_wait<NWD>();                                  // the weights of the next phase have landed
        } else if constexpr (NT == 9) {
            if constexpr (TP == 0) { issue_table(is_buf); issue_any(0, is_buf); issue_any(1, is_buf); }
            else issue_any(TP + 1, is_buf);
            vm_wait<NVM>();                                  // the same slot(s) of the previous period (chunk c+1) have landed
            if (cm_gn) {
                if constexpr (TP == 0) { commit_slot(0, cm_buf); commit_slot(1, cm_buf); }
                else commit_slot(TP + 1, cm_buf);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NSLOT1; ++i) issue_compact(lw + NLAG * i, is_buf);
            vm_wait<NSLOT1>();                               // everything older: the image of chunk c+1
        }
        raw_barrier();
        // ================= C =================
        __builtin_amdgcn_sched_barrier(0);
        prio(1);
        {
            // ONE fragment read per MFMA gap: the second k-group's six fragments arrive under MFMAs 0-2, then the first
            // k-group of the NEXT phase refills fa0 / fb0 as the MFMAs release them (conv_pipe.hip)
            const int rn = (ring_rd + WPHASE) & (RINGB - 1);
            auto next_a = [&](auto mi_) {
                constexpr int mi = decltype(mi_)::value;
                if constexpr (NT == 9 && TP < 8) read_a(fa0[mi], rn, IC<0>{}, mi_);
                else if constexpr (NT == 9) { if (w1_ntaps == 9) read_a(fa0[mi], rn, IC<0>{}, mi_); }
            };
            auto next_b = [&](auto ni_) {
                if constexpr (NT == 9 && TP < 8) read_b(fb0[decltype(ni_)::value], pbase[DXN], IC<0>{}, PoffN{}, Prow{}, ni_);
                else if constexpr (NT == 9) {
                    if (w1_ntaps == 9) read_b(fb0[decltype(ni_)::value], pbase[0] + (cm_buf - rd_buf) * PATCH_BYTES, IC<0>{}, IC<0>{}, Prow9{}, ni_);
                }
            };
            typedef IC<1> K1;
#define STORM_SB() __builtin_amdgcn_sched_barrier(0)
            mma_part(fa0, fb0, 0, 1); read_a(fa1[0], ring_rd, K1{}, IC<0>{}); read_b(fb1[0], pb, K1{}, Poff{}, Prow{}, IC<0>{}); STORM_SB();
            mma_part(fa0, fb0, 1, 2); read_b(fb1[1], pb, K1{}, Poff{}, Prow{}, IC<1>{}); read_b(fb1[2], pb, K1{}, Poff{}, Prow{}, IC<2>{}); STORM_SB();
            mma_part(fa0, fb0, 2, 3); read_b(fb1[3], pb, K1{}, Poff{}, Prow{}, IC<3>{}); read_a(fa1[1], ring_rd, K1{}, IC<1>{}); STORM_SB();
            mma_part(fa0, fb0, 3, 4); STORM_SB();
            mma_part(fa0, fb0, 4, 5); next_a(IC<0>{}); STORM_SB();          // fa0[0]: last used by MFMA 3
            mma_part(fa0, fb0, 5, 6); next_b(IC<0>{}); STORM_SB();          // fb0[0]: last used by MFMA 4
            mma_part(fa0, fb0, 6, 7); next_b(IC<1>{}); STORM_SB();
            mma_part(fa0, fb0, 7, 8); next_b(IC<2>{}); STORM_SB();
            mma_part(fa1, fb1, 0, 1); next_b(IC<3>{}); next_a(IC<1>{}); STORM_SB();   // fb0[3], fa0[1]: last used by MFMA 7
            mma_part(fa1, fb1, 1, WM * WN); STORM_SB();
#undef STORM_SB
            raw_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        prio(0);
        ring_next();
    };

    // ---- tile start, part A: chunk 0's patch -> buffer 0 / table 0 (lagging), the first two taps' weights -> ring slots 0, 1
    // (leading): regions the previous tile's epilogue staging does not touch
    auto tile_issue = [&]() {
        load_next(0);
        enter_wrun(w2_wrun);
        w_soff = w2_wsoff;
        if (grp == 0) {
            char* dst = smem + OFF_RING + lw * (NWD * 1024);
            if (!(ABL & 8)) {
#pragma unroll
                for (int j = 0; j < NWD; ++j) dma16(w_srd, R[j], (uint32_t)w_soff, dst + j * 1024, lane);
            }
            w_soff += w_tapbytes;                                // (the first chunk has nine taps)
            if (!(ABL & 8)) {
#pragma unroll
                for (int j = 0; j < NWD; ++j) dma16(w_srd, R[j], (uint32_t)w_soff, dst + WPHASE + j * 1024, lane);
            }
        } else {
            issue_table(0);
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) issue_slot(i, 0);
        }
    };
    const int nchunks = pin(ap->nchunks), n9 = pin(ap->nchunks9);
    auto chunk_change = [&](int ci) {                       // chunk ci is done: rotate the buffers, descriptor of chunk ci + 3
        const int old = rd_buf;
        rd_buf = cm_buf; cm_buf = is_buf; is_buf = old;
        const int dlt = (rd_buf - old) * PATCH_BYTES;
#pragma unroll
        for (int d = 0; d < 3; ++d) pbase[d] += dlt;
        load_next(ci + 3);
        launder(lane);
    };
    tile_issue();
    while (true) {
        // ---- tile start, part B: chunk 1 -> buffer 1 (after the previous tile's epilogue: it staged there); chunk 0 has landed;
        // fused GroupNorm transform of chunk 0 -------------------------------------------------------------------------
        STORM_RELAUNDER();
        stamp();                                            // 0: tile start (part B)
        load_next(1);
        if (grp == 1) {
            issue_table(1);
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) issue_any(i, 1);
            vm_wait<NVM>();                                  // chunk 0 (chunk 1 stays in flight: waited for slot by slot in chunk 0's phases)
            if (cm_gn) {
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) commit_slot(i, 0);
            }
        } else {
            vm_wait<0>();
        }
        load_next(2);
        raw_barrier();
        stamp();                                            // 1: chunk 0 landed and transformed, main loop begins
        read_frags(fa0, fb0, 0, pbase[0], IC<0>{}, IC<0>{}, Prow9{});   // first k-group of phase 0
        if (grp == 1) raw_barrier();                        // the lagging group starts one interval later
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

        // ---- main loops: the nine-tap chunks, then the one-tap chunks of a fused 1x1 shortcut ---------------------------
        int ci = 0;
        for (; ci < n9; ++ci) {
            static_for<9>([&](auto t) { phase(t, IC<9>{}); });
            chunk_change(ci);
        }
        for (; ci < nchunks; ++ci) {
            phase(IC<0>{}, IC<1>{});
            chunk_change(ci);
        }
        if (grp == 0) raw_barrier();                        // balance the stagger: every wave has executed the same barriers
        stamp();                                            // 2: main loop done

        // ---- hand-over: this tile's coordinates go to the epilogue; the next tile's first loads are issued -----------------
        vm_wait<0>();                                       // trailing (zero-fill) patch / ring loads landed ...
        raw_barrier();                                      // ... and every wave is done reading: all of LDS is free
        const int e_tile = tile, e_b = b, e_ty0 = ty0, e_tx0 = tx0, e_cout0 = cout0;
        STORM_RELAUNDER();
        int nvb = vb + gridDim.x;
        while (nvb < total_vblocks && block_map(nvb, n_ct, tiles_per_xcd).tile >= ntiles) nvb += gridDim.x;
        const bool has_next = nvb < total_vblocks;
        {                                                   // the next tile starts in patch buffer 0 / ring slot 0
            const int dlt = rd_buf * PATCH_BYTES;
#pragma unroll
            for (int d = 0; d < 3; ++d) pbase[d] -= dlt;
            rd_buf = 0; cm_buf = 1; is_buf = 2;
            ring_rd = 0;
        }
        if (has_next) {
            vb = nvb;
            decode(vb);
            launder(lane);
            tile_issue();
        }
        STORM_RELAUNDER();
        stamp();                                            // 3: hand-over done (drain, barrier, next tile's first loads issued)

        // ---- epilogue (conv_epilogue.h).  Staging lives in patch buffers 1 and 2, the statistics scratch in ring slots 2, 3: the next
        // tile's first loads are landing in buffer 0 / table 0 / ring slots 0, 1 meanwhile.  GroupNorm partials go out in the 8 x 32
        // pixel tile layout every conv kernel writes (storm_conv_tiles): this tile is two of them - pixel rows 0-7 (wave rows wn = 0, 1)
        // and rows 8-15 (wn = 2, 3).
        const epi::TileAt et = {e_tile, e_b, e_ty0, e_tx0, e_cout0};
        float gsum[8], gsq[8];
        if (ABL & 1024) {                                    // (profiling) no epilogue: the accumulators stay alive, nothing is staged or stored
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) keep(acc[mi][ni]);
        } else {
        epi::store_tile<T, WM, WN>(acc, smem + PATCH_BYTES + wave * WSTAGE, ap, et, wm, wn, lane, imgH, imgW, BN, TH, gsum, gsq);
        stamp();                                            // 4: epilogue stores issued
        if (ap->gn_part != nullptr)
            epi::write_stats<WM, WN, WAVES_N, BN, TH>(gsum, gsq, reinterpret_cast<float*>(smem + OFF_RING + 2 * WPHASE), ap, et, wm, wn, lane, tid,
                                                      imgH, tiles_x, tiles_per_img);
        }
        stamp();                                            // 5: statistics written
        if (!has_next) break;
        __syncthreads();                                    // the statistics scratch / staging of this tile is free again
    }
}

#undef STORM_RELAUNDER

// ---- host side ---------------------------------------------------------------------------------------------------
bool conv_pipe128_supports(const storm_conv_args& a) {
    if (a.outC > pipe128::BN) return false;
    PipeParams p;
    return build_pipe_params(a, p, pipe128::KC);
}

template <typename T, int ABL = 0>
static int launch_pipe128(const storm_conv_args& a, hipStream_t st) {
    constexpr bool TRACE = (ABL & 64) != 0;
    auto kern = conv_pipe128_kernel<T, ABL>;
    static bool attr_set = false;                       // per instantiation; benign race (idempotent)
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pipe128::LDS_BYTES));
        attr_set = true;
    }
    PipeParams prm;
    STORM_CHECK(a.outC <= pipe128::BN && build_pipe_params(a, prm, pipe128::KC), "storm_conv: convolution outside the 128-cout pipelined kernel's coverage");
    if (TRACE) prm.trace = reinterpret_cast<unsigned long long*>(switches().conv_trace_ptr);
    const int tiles_x = cdiv(a.W, TILE_W);
    const int tiles_per_img = tiles_x * cdiv(a.H, pipe128::TH);
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int n_ct = 1;
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long vblocks = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(vblocks > 0 && vblocks < (1LL << 31), "storm_conv: grid %lld out of range", vblocks);
    const long long resident = (device_cus() + 7) / 8 * 8;             // one workgroup per CU; a multiple of 8
    const long long grid = vblocks < resident ? vblocks : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pipe128::THREADS), pipe128::LDS_BYTES, st, prm, n_ct, tiles_per_xcd, (int)ntiles,
                       tiles_x, tiles_per_img, (int)vblocks);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_conv_pipe128(const storm_conv_args& a, hipStream_t st) {
#if defined(STORM_PROFILING)
    if (a.dtype == STORM_BF16) switch (switches().conv_ablate) {
        case 64: return launch_pipe128<bf16_t, 64>(a, st);
        case 8: return launch_pipe128<bf16_t, 8>(a, st);
        case 16: return launch_pipe128<bf16_t, 16>(a, st);
        case 128: return launch_pipe128<bf16_t, 128>(a, st);
        case 136: return launch_pipe128<bf16_t, 136>(a, st);
        case 1024: return launch_pipe128<bf16_t, 1024>(a, st);
        case 256: return launch_pipe128<bf16_t, 256>(a, st);
        default: break;
    }
#endif
    return a.dtype == STORM_F16 ? launch_pipe128<half_t>(a, st) : launch_pipe128<bf16_t>(a, st);
}

const char* conv_pipe128_kernel_name(int dtype) {
    return dtype == STORM_F16 ? "storm::conv_pipe128_kernel<storm::half_t, 0>" : "storm::conv_pipe128_kernel<storm::bf16_t, 0>";
}

}  // namespace storm
