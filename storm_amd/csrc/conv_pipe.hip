// Software-pipelined 3x3 implicit-GEMM convolution (bf16): the kernel behind every wide 3x3 layer of NCSN++
// (layers.py:119-126 ddpm_conv3x3, with the fused pieces of layerspp.py:242-274 listed in include/storm_hip.h).
//
// Same math, arguments and epilogue as conv_igemm.hip; what differs is the K loop, built for ONE 8-wave workgroup
// per CU (2 waves / SIMD, 256 registers each) and for an instruction stream in which nothing but MFMAs, fragment
// reads and DMA issues is left:
//   * tile: BN output channels x (TH x 32) pixels; every wave owns 64 couts x (4 x 32) pixels = 8 accumulator tiles
//     of v_mfma_f32_32x32x16_bf16.   <256, 8>: wave grid 4 (cout) x 2 (pixel rows).
//     <128, 8> (round 4): 128 couts x 256 pixels, wave grid 2 x 4, a wave owns 64 couts x (2 x 32) pixels = 4 accumulator tiles and
//     issues 8 MFMAs per phase instead of 16 - for layers with 128 ... 511 pixel tiles (the 32 x 64 level at batch 16), where the
//     256-cout tile leaves half of the CUs without a workgroup and the launch time is the serial chain of phases per workgroup.
//     Same chunk order, same phases, same epilogue: results are bit-identical to <256, 8>.
//   * the K dimension is a HOST-BUILT list of chunk descriptors (conv_params.h: 64 channels of one source tensor
//     under all of its taps), and the body of a chunk is compile-time unrolled over its taps (two bodies: nine taps,
//     one tap).  Tap offsets, pixel rows and k-groups are instruction immediates; the per-phase scalar work is one
//     ring-slot rotation and one weight-offset add.  (Round 1 walked taps / chunks / runs with a run-time cursor:
//     its bookkeeping alone - no MFMA, no DMA, no fragment reads - took 58 % of the kernel's time.)
//   * nothing asynchronous ever targets a VGPR.  Weights: every "phase" (half a tap of one 128-byte K-chunk:
//     BN rows x 64 B) is copied global -> LDS by `buffer_load_dwordx4 ... lds` into a 4-slot ring, issued two phases
//     before use.  The LDS image of such an instruction is lane-linear, so the bank swizzle is applied to the per-lane
//     SOURCE offset (constant for a whole run; tap / chunk / half ride in the scalar offset) and again on the read.
//   * the haloed (TH+2) x 34 pixel patch of the NEXT chunk is fetched the same way into the other patch buffer, ONE
//     1-KiB piece per wave and phase, and - with a fused GroupNorm-apply + SiLU - rewritten in place by the lane that
//     fetched it, one piece per phase, six phases later.  Padding pixels / ragged channels are out-of-range buffer
//     reads = hardware zeros.  The patch image is swizzled by pixel COLUMN, so a tap / pixel row change is an
//     instruction immediate.  A one-tap chunk (the fused 1x1 shortcut) uses a compact TH x 32 image (no halo).
//   * ping-pong with ROLES: waves 4-7 (the second wave of every SIMD, "lagging") run one barrier interval behind waves
//     0-3 ("leading"), so one group's MFMA interval C coincides with the other's staging interval S.  vmcnt retires in
//     order, so a wave that waits for L2-resident weights every phase would also wait for every HBM patch piece it
//     issued since: the LEADING group therefore streams all weights (4 pieces per wave and phase, waited for one
//     phase later) and the LAGGING group fetches and transforms all patches with lazy waits (a piece is waited for six
//     phases after its issue; the whole patch two phases before the chunk ends).
//   * fragment reads live in the MFMA intervals only: C(P) reads its second k-group under its first MFMAs and, once the
//     first k-group's MFMAs are issued, the FIRST k-group of phase P+1 into the registers they free - a staging
//     interval is DMA issue + wait + barrier, and the matrix pipe starts the moment a barrier opens.
//   * all VMEM of the loop is inline asm, so the counted `s_waitcnt vmcnt(N)` are the only waits and loads stay in
//     flight across the raw s_barriers (N = VMEM instructions this wave issued AFTER the one that must have landed).
//
// Phase = (tap, half), two k-groups, 16 MFMAs per wave:
//   S(P):  leading: DMA weights of phase P+2 -> ring slot (P+2)&3 | vmcnt: weights of P+1 landed
//          lagging: DMA one patch piece of the next chunk | vmcnt: the piece issued six phases ago | transform it
//          barrier
//   C(P):  2 MFMAs | read k-group 1 | 6 MFMAs | read k-group 0 of P+1 | 8 MFMAs | barrier
// Interval numbering (barriers): leading group S(P) = 2P, C(P) = 2P+1; lagging group one later.
// LDS lifetimes: ring slot (P+2)&3 = (P-2)&3 was last read in C(P-2) (lagging: interval 2P-2) -> free in the leading
// S(P) (2P); the weights of P+1 are waited for in the leading S(P) and visible from 2P+1 = the leading C(P), which
// pre-reads them.  The other patch buffer was last read by the lagging C(P0-1) (interval 2P0, P0 = first phase of the
// chunk): the lagging group issues from its S(P0) (2P0+1) on, transforms until its S(P0+16) (2P0+33), and the first
// reader is the leading C(P0+17) (2P0+35).  A ONE-tap chunk is fetched whole in the lagging S(P0) and waited for in its
// S(P0+1) (2P0+3): its successor's first k-group is therefore read at the START of the successor's S(0) (2P0+4), not
// pre-read.
#include <cstdlib>
#include <cstring>
#include "conv_epilogue.h"

namespace storm {
using namespace cidx;

namespace pipe {

constexpr int PW = TILE_W + 2;
constexpr int PIXB = 128, KC = 64, SLOTS = 8;                 // bytes / channels / 16-B slots per pixel and K-chunk
constexpr int PR = 1;                                         // pixel rows (of 32 px) staged per epilogue pass and wave
constexpr int NWAVES = 8, THREADS = 512;

template <int BN_, int TH_> struct PCfg {
    static constexpr int BN = BN_, TH = TH_;
    static constexpr int WAVES_M = BN / 64, WAVES_N = NWAVES / WAVES_M;
    static constexpr int WM = 2, WN = TH / WAVES_N;           // 32-cout tiles / pixel rows (32 px) per wave
    static constexpr int NPIX = (TH + 2) * PW;
    static constexpr int PPIECES = (NPIX + 7) / 8;            // 1-KiB DMA pieces (8 rows) of a haloed patch
    static constexpr int PATCH_BYTES = PPIECES * 1024;
    static constexpr int CPIECES = TH * 4;                    // pieces of a compact (one-tap) patch
    static constexpr int NLAG = 4;                            // patch-fetching (lagging) waves
    static constexpr int NSLOT = (PPIECES + NLAG - 1) / NLAG; // haloed pieces per lagging wave (one per phase)
    static constexpr int NSLOT1 = CPIECES / NLAG;             // compact pieces per lagging wave
    static constexpr int LAZY = 6;                            // phases between a piece's issue and its wait / transform
    static constexpr int WPHASE = BN * WROW, RINGB = 4 * WPHASE;
    static constexpr int NWD = BN / 16 / 4;                   // weight DMA instructions per LEADING wave and phase (16 rows each)
    static constexpr int OFF_RING = 2 * PATCH_BYTES;
    static constexpr int OFF_SS = OFF_RING + RINGB;
    static constexpr int MAIN_BYTES = OFF_SS + 2 * 1024;
    static constexpr int WSTAGE = 32 * PR * WM * 128;         // one wave's epilogue staging (one pixel row x 64 couts, fp32)
    // Epilogue staging beside the next tile's landing loads: waves 0-4 in patch buffer 1; waves 5-7 + the statistics scratch in ring
    // slots 2, 3 when they fit there (<256, 8>: 2 x 16 KiB), else in a region of their own behind the main image (<128, 8>)
    static constexpr int TAIL_BYTES = 3 * WSTAGE + WAVES_N * BN * 8;
    static constexpr bool TAIL_IN_RING = TAIL_BYTES <= 2 * WPHASE;
    static constexpr int OFF_TAIL = TAIL_IN_RING ? OFF_RING + 2 * WPHASE : MAIN_BYTES;
    static constexpr int LDS_BYTES = TAIL_IN_RING ? MAIN_BYTES : MAIN_BYTES + TAIL_BYTES;
    static_assert(WN == 4 || WN == 2, "wave tile is 64 couts x 4 (or 2) pixel rows");
    static_assert(5 * WSTAGE <= PATCH_BYTES, "epilogue staging of waves 0-4 inside patch buffer 1");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((WPHASE & (WPHASE - 1)) == 0, "ring rotation by mask");
    static_assert(PATCH_BYTES % 256 == 0, "k-group XOR must stay inside the slot field");
    static_assert(NSLOT * NLAG - PPIECES < NLAG, "at most one surplus slot per wave");
    static_assert(CPIECES % NLAG == 0 && NSLOT1 <= NSLOT, "compact pieces fit the same slots");
    static_assert(NSLOT - 1 + LAZY <= 16, "the whole patch is transformed two phases before a nine-tap chunk ends");
};

// Patch image: pixel row r, 128 B; its eight 16-B slots are XOR-swizzled by the pixel COLUMN ((px >> 1) & 7).
// 16 lanes of a fragment read are 16 consecutive px -> 16 distinct bank groups; and because the swizzle does not
// depend on the pixel row, taps / wave rows / buffers are plain additions.
STORM_HD int p_swz(int px, int slot) { return (slot ^ ((px >> 1) & 7)) << 4; }

}  // namespace pipe
using namespace pipe;

// ABL: profiling-only instantiations (built with -DSTORM_PROFILING into libstorm_hip_prof.so, never in the product
// library): 8 no weight DMA after the prologue, 16 no fragment reads, 32 no MFMAs, 128 no patch DMA / transform,
// 64 wave-timeline stamps (tools/conv_trace.py).
// SPLIT: the workgroup id carries a K slice as well - (pixel tile, cout tile, slice); a workgroup walks only its slice's chunk
// descriptors (the nine-tap chunks in `kslices` contiguous ranges, the one-tap chunks with the last range) and stores its raw fp32
// accumulators as slab `slice` of the output (the host points a.out at the slabs: fp32, no bias / skip / statistics).
// GROUP: ONE launch over the pixel tiles of several problems (conv_params.h: GroupTile) - the parameter block of a tile's problem is read
// from a device table instead of the kernarg segment, its (problem, image, row, column) from a host-built list; everything else - chunk
// descriptors, phases, epilogue - is the code of the plain instantiation, so a tile's output is the bits its own launch would write.
template <typename T, int BN, int TH, int ABL, bool SPLIT, bool GROUP = false>
__device__ __forceinline__ void conv_pipe_body(const PipeParams& a, const int n_ct, const int tiles_per_xcd,
                                               const int ntiles, const int tiles_x, const int tiles_per_img, const int total_vblocks,
                                               const PipeParams* gtab = nullptr, const GroupTile* glist = nullptr) {
    typedef PCfg<BN, TH> Cfg;
    constexpr bool TRACE = (ABL & 64) != 0;
    constexpr int WM = Cfg::WM, WN = Cfg::WN, WAVES_M = Cfg::WAVES_M, WAVES_N = Cfg::WAVES_N, NWD = Cfg::NWD;
    constexpr int PATCH_BYTES = Cfg::PATCH_BYTES, PPIECES = Cfg::PPIECES, CPIECES = Cfg::CPIECES;
    constexpr int NSLOT = Cfg::NSLOT, NSLOT1 = Cfg::NSLOT1, WPHASE = Cfg::WPHASE, RINGB = Cfg::RINGB;
    constexpr int OFF_RING = Cfg::OFF_RING, OFF_SS = Cfg::OFF_SS;
    typedef typename Mma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // The parameter block is read through the kernarg segment pointer, re-laundered at every tile and before every epilogue:
    // otherwise every scalar load of the block is hoisted out of the tile loop and the live SGPRs spill.
    PipeArgPtr ap = GROUP ? const_table(gtab) : pipe_args(a);
    const PipeArgPtr gtab4 = GROUP ? const_table(gtab) : PipeArgPtr();
    const typename KArg<GroupTile>::Ptr glist4 = GROUP ? const_table(glist) : typename KArg<GroupTile>::Ptr();
#define STORM_RELAUNDER() relaunder(ap)

    // Persistent workgroups: at most one per CU, each walking the virtual block ids blockIdx.x, + gridDim.x, ... (gridDim.x
    // is a multiple of 8, so a workgroup stays on its XCD's tile range).  The first patch and weights of the NEXT tile are
    // issued before this tile's epilogue, whose stores then drain under the next tile's main loop.
    int vb = blockIdx.x;
    while (vb < total_vblocks && block_map(vb, n_ct, tiles_per_xcd).tile >= ntiles) vb += gridDim.x;   // (padding ids of the XCD map)
    if (vb >= total_vblocks) return;
    int tile, b, ty0, tx0, cout0;                           // the tile whose loads are being ISSUED (= computed, until the hand-over)
    int nchunks = pin(ap->nchunks), n9 = pin(ap->nchunks9); // chunks of this workgroup's K loop (SPLIT: of its slice, from descriptor c0 on)
    int c0 = 0, eb_off = 0;
    int imgH = 0, imgW = 0;                                 // (GROUP: of the problem whose tile is being issued; else set once below)
    auto decode = [&](int v) {
        const BlockMap bm = block_map(v, n_ct, tiles_per_xcd);
        if constexpr (GROUP) {                              // the tile's problem and place from the host-built list (four scalar loads)
            const unsigned int gp = glist4[bm.tile].problem, gb = glist4[bm.tile].b, gyx = glist4[bm.tile].yx, gt = glist4[bm.tile].tile;
            ap = gtab4 + gp;
            tile = (int)gt; b = (int)gb; ty0 = (int)(gyx & 0xffffu); tx0 = (int)(gyx >> 16);
            imgH = ap->H; imgW = ap->W;
        } else {
        tile = bm.tile;
        b = bm.tile / tiles_per_img;
        const int trem = bm.tile - b * tiles_per_img;
        ty0 = (trem / tiles_x) * TH;
        tx0 = (trem % tiles_x) * TILE_W;
        }
        int ct = bm.ct;
        if constexpr (SPLIT) {
            const int nct = ap->split_nct, S = ap->kslices, n9all = ap->nchunks9;
            const int slice = ct / nct;
            ct -= slice * nct;
            c0 = slice * n9all / S;
            n9 = (slice + 1) * n9all / S - c0;
            nchunks = slice == S - 1 ? ap->nchunks - c0 : n9;
            eb_off = slice * ap->B;                         // slab `slice` = "batch items" slice * B ... of the fp32 output
        }
        cout0 = ct * BN;
    };
    decode(vb);
    if constexpr (!GROUP) { imgH = pin(ap->H); imgW = pin(ap->W); }   // (two SGPRs for the whole kernel: read in the pipelined loop)

    const int tid = threadIdx.x;
    int lane = tid & 63;                                    // re-laundered at every chunk (see relaunder())
    const int wave = uniform(tid >> 6);
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int grp = wave >> 2;                              // 0: leading, 1: lagging (one barrier interval behind)

    unsigned long long* const trace_rec = TRACE && ap->trace ? ap->trace + ((long long)blockIdx.x * NWAVES + wave) * TRACE_SLOTS : nullptr;
    auto stamp = [&](int idx) {                             // profiling build only (tools/conv_trace.py)
        if (TRACE && trace_rec && idx < TRACE_SLOTS) {
            const unsigned long long t = hw_memtime();
            if (lane == 0) trace_rec[idx] = t;
        }
    };
    if (TRACE && trace_rec && lane == 0) trace_rec[0] = hw_ids();
    stamp(1);

    f32x16 acc[WM][WN];

    // ---- lane constants -----------------------------------------------------------------------------------
    // weights: row of mi is +32 rows = +2048 B (same swizzle); the second k-group of a phase flips slot bit 1
    const int aoff = OFF_RING + w_off(wm * WM * 32 + (lane & 31), lane >> 5);
    const int aoff1 = aoff ^ 32;
    // patch: this lane's pixel of ni = 0 under tap (0, 0) + the k-group-0 swizzle term of tap column dx
    // (all other parts of a fragment address are instruction immediates; the buffer parity is added at a chunk change)
    int pbase[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) pbase[d] = ((wn * WN) * PW + (lane & 31)) * PIXB + p_swz((lane & 31) + d, lane >> 5);
    const int cdelta = wn * WN * (PW - TILE_W) * PIXB;      // haloed row index - compact row index of this wave's pixels (uniform)

    // ---- role registers ---------------------------------------------------------------------------------------
    // leading waves: R[j] = per-lane source offset of weight piece j of the current run;
    // lagging waves: R[i] = haloed patch piece lw + 4 i of this wave (issued in phase i of a nine-tap chunk):
    //   (pixel index << 3) | logical 16-B slot that lands in this lane's physical slot, or -1 (padding / past the
    //   patch: hardware zero fill).
    const int lw = wave & 3;
    uint32_t R[NSLOT];
    static_assert(NWD <= NSLOT, "role registers");
    auto patch_table = [&]() {                              // (lagging waves) for the tile (ty0, tx0)
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                const int k = lw + Cfg::NLAG * i;                  // (k >= PPIECES: surplus slot, never used as a piece)
                const int row = k * 8 + (lane >> 3);
                const int py = row / PW, px = row - py * PW;
                const int slot = (lane & 7) ^ ((px >> 1) & 7);
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                const bool ok = row < Cfg::NPIX && gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
                R[i] = ok ? (uint32_t)(((gy * imgW + gx) << 3) | slot) : 0xffffffffu;
            }
        }
    };

    // ---- issue-side state: the NEXT chunk's patch source and the weight stream ----------------------------------
    u32x4 nx_srd, nx_ss_srd, w_srd;
    const int nchunks_k = pin(ap->nchunks);                   // index of the terminator descriptor
    int nx_C2 = 0, nx_cbeg2 = 0, nx_cvalid = 0, nx_ntaps = 9, nx_gn = 0, nx_silu = 0, nx_wrun = 0, nx_wsoff = 0, nx_neww = 0;
    int w_soff = 0, w_tapbytes = 0;
    auto load_next = [&](int i) {                           // descriptor i -> nx_* (scalar loads from the kernarg segment)
        const ChunkDesc& d = ap->chunk[SPLIT ? (i < nchunks ? c0 + i : nchunks_k) : (i < nchunks_k ? i : nchunks_k)];
        nx_srd = make_srd(reinterpret_cast<const char*>(d.src + (unsigned long long)b * d.bstride), d.src_bytes);
        nx_gn = d.ss != 0ull;
        nx_ss_srd = make_srd(reinterpret_cast<const char*>(nx_gn ? d.ss + (unsigned long long)b * d.ss_bstride : d.src),
                             nx_gn ? (uint32_t)d.cvalid * 8u : 0u);
        nx_C2 = d.C2; nx_cbeg2 = d.cbeg2; nx_cvalid = d.cvalid; nx_ntaps = d.ntaps; nx_silu = d.silu;
        nx_wrun = d.wrun; nx_wsoff = d.w_soff; nx_neww = d.new_wrun;
    };
    auto enter_wrun = [&](int r) {                          // (leading waves: R = weight piece offsets)
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
    auto w_issue = [&](int half) {                          // (leading) half of the stream's tap -> the slot two phases ahead
        if (ABL & 8) return;
        char* dst = smem + OFF_RING + (ring_rd ^ (2 * WPHASE)) + lw * (NWD * 1024);
        const uint32_t so = (uint32_t)(w_soff + half * WROW);
#pragma unroll
        for (int j = 0; j < NWD; ++j) dma16(w_srd, R[j], so, dst + j * 1024, lane);
    };
    auto ring_next = [&]() { ring_rd = (ring_rd + WPHASE) & (RINGB - 1); };

    int par = 0;                                            // patch buffer of the chunk being read
    // (lagging) the (scale, shift) table of the next chunk: every lagging wave fetches its own copy (identical bytes), so
    // that its own vmcnt orders it before the transforms
    auto issue_table = [&](int into) { dma16(nx_ss_srd, (uint32_t)lane * 16u, 0u, smem + OFF_SS + into * 1024, lane); };
    // (lagging) haloed piece i of this wave -> the other buffer
    auto issue_slot = [&](int i, int into) {
        const int k = lw + Cfg::NLAG * i;
        if (k >= PPIECES) { issue_table(into); return; }     // surplus slot: identical table bytes again - harmless whenever
                                                             // they land (a repeated PIECE could land on its transformed image)
        uint32_t v = R[i];
        launder(v);                       // opaque per use: values DERIVED from the table entry must not be hoisted out of the tile loop
                                          // as eleven more live registers (one of them spilled: scratch reload + vmcnt(0) in the loop)
        const bool ok = (int)v >= 0 && (int)(v & 7u) * 8 < nx_cvalid;
        dma16(nx_srd, ok ? mad24(v >> 3, (uint32_t)nx_C2, (v & 7u) * 16u) : OOB, (uint32_t)nx_cbeg2,
              smem + into * PATCH_BYTES + k * 1024, lane);
    };
    // (lagging) one compact piece (one-tap chunk: TH x 32 pixels, no halo): piece k = pixel row k >> 2, columns 8 (k & 3) ..
    auto issue_compact = [&](int k, int into) {
        const int trow = k >> 2, n = (k & 3) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((n >> 1) & 7);
        const int gy = ty0 + trow, gx = tx0 + n;
        const bool ok = gy < imgH && gx < imgW && slot * 8 < nx_cvalid;
        dma16(nx_srd, ok ? mad24((uint32_t)(gy * imgW + gx), (uint32_t)nx_C2, (uint32_t)slot * 16u) : OOB, (uint32_t)nx_cbeg2,
              smem + into * PATCH_BYTES + k * 1024, lane);
    };
    auto issue_any = [&](int i, int into) {                 // (lagging) slot i, in the next chunk's layout
        if (nx_ntaps == 9) issue_slot(i, into);
        else if (i < NSLOT1) issue_compact(lw + Cfg::NLAG * i, into);
        else issue_table(into);                              // surplus slot (keeps the VMEM count uniform)
    };
    // (lagging) fused GroupNorm-apply (+ SiLU): in place, by the lane that fetched the unit (haloed layout only)
    auto commit_slot = [&](int i, int into) {
        const int k = lw + Cfg::NLAG * i;
        if (k < PPIECES) {
            uint32_t v = R[i];
            launder(v);
            if ((int)v >= 0 && (int)(v & 7u) * 8 < nx_cvalid) {
                uint4* const q = reinterpret_cast<uint4*>(smem + into * PATCH_BYTES + k * 1024 + lane * 16);
                float ss[16];
                load_ss<8>(reinterpret_cast<const float*>(smem + OFF_SS + into * 1024), (int)(v & 7u), ss);
                *q = gn_act_slot(*q, ss, nx_silu, (T*)nullptr);
            }
        }
    };

    // ---- fragment reads / MFMAs --------------------------------------------------------------------------------
    // k-group kg (0..3) of the chunk = k-group (kg & 1) of the ring phase at byte offset `ring`; POFF = byte offset of
    // the tap inside the patch image; PROW = bytes between consecutive pixel rows of the image (haloed: PW, compact: 32 px)
    auto read_frags = [&](Frag (&fa)[WM], Frag (&fb)[WN], int ring, int pb, auto kg_, auto poff_, auto prow_) {
        constexpr int kg = decltype(kg_)::value, POFF = decltype(poff_)::value, PROW = decltype(prow_)::value;
        if (ABL & 16) {                                      // operands stay whatever the prologue left in the registers
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) keep_rw(fa[mi]);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) keep_rw(fb[ni]);
            return;
        }
        const char* wb = smem + ring + ((kg & 1) ? aoff1 : aoff);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const Frag*>(wb + mi * 32 * WROW);
        const char* pp = smem + (pb ^ (kg << 5)) + POFF;
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const Frag*>(pp + ni * PROW);
    };
    auto read_a = [&](Frag& f, int ring, auto kg_, auto mi_) {           // one weight fragment
        constexpr int kg = decltype(kg_)::value, mi = decltype(mi_)::value;
        if (ABL & 16) { keep_rw(f); return; }
        f = *reinterpret_cast<const Frag*>(smem + ring + ((kg & 1) ? aoff1 : aoff) + mi * 32 * WROW);
    };
    auto read_b = [&](Frag& f, int pb, auto kg_, auto poff_, auto prow_, auto ni_) {   // one pixel fragment
        constexpr int kg = decltype(kg_)::value, POFF = decltype(poff_)::value, PROW = decltype(prow_)::value, ni = decltype(ni_)::value;
        if (ABL & 16) { keep_rw(f); return; }
        f = *reinterpret_cast<const Frag*>(smem + (pb ^ (kg << 5)) + POFF + ni * PROW);
    };
    auto mma_part = [&](const Frag (&fa)[WM], const Frag (&fb)[WN], int lo, int hi) {   // MFMAs lo..hi-1 of the k-group's WM x WN
        if (ABL & 32) {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) keep(fa[mi]);
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) keep(fb[ni]);
            return;
        }
#pragma unroll
        for (int i = 0; i < WM * WN; ++i)
            if (i >= lo && i < hi) Mma<T>::run(fa[i / WN], fb[i % WN], acc[i / WN][i % WN]);
    };
    Frag fa0[WM], fb0[WN], fa1[WM], fb1[WN];
    int tstep = 0;                                          // (profiling) tap-steps stamped so far
    typedef IC<PW * PIXB> Prow9; typedef IC<TILE_W * PIXB> Prow1;

    // ---- one tap-step (two phases) of a chunk with NT taps; TP = tap index ------------------------------------------
    auto tap_step = [&](auto t_, auto nt_) {
        constexpr int TP = decltype(t_)::value, NT = decltype(nt_)::value;
        constexpr int DX = NT == 9 ? TP % 3 : 0;
        constexpr int POFF = NT == 9 ? ((TP / 3) * PW + DX) * PIXB : 0;
        constexpr int DXN = NT == 9 ? (TP + 1) % 3 : 0;                          // the next tap of the chunk
        constexpr int POFFN = NT == 9 ? (((TP + 1) / 3) * PW + DXN) * PIXB : 0;
        typedef IC<POFF> Poff; typedef IC<POFFN> PoffN;
        typedef std::conditional_t<NT == 9, Prow9, Prow1> Prow;
        const int pb = NT == 9 ? pbase[DX] : pbase[0] - cdelta;
        const int into = par ^ 1;
        const int sb = tstep < 30 ? 4 + 16 * tstep : 4096;   // (profiling: the first 30 tap-steps are stamped)
        // the weight stream moves to the next tap (last tap: to the first tap of the next chunk)
        if constexpr (TP == NT - 1) {
            if (nx_neww) enter_wrun(nx_wrun);
            w_soff = nx_wsoff;
        } else {
            w_soff += w_tapbytes;
        }
        static_for<2>([&](auto h_) {
            constexpr int h = decltype(h_)::value;
            constexpr int p = 2 * TP + h;                    // phase of the chunk
            constexpr bool pdma = !(ABL & 128);
            // ================= S =================
            stamp(sb + (h ? 7 : 0));
            if constexpr (NT == 1 && h == 0) read_frags(fa0, fb0, ring_rd, pb, IC<0>{}, Poff{}, Prow{});   // (not pre-read: see top)
            if (grp == 0) {
                w_issue(h);
                stamp(sb + (h ? 12 : 1));
                vm_wait<(ABL & 8) ? 0 : NWD>();                  // the weights of the next phase have landed
            } else if constexpr (pdma && NT == 9) {
                if constexpr (p == 0) issue_table(into);
                if constexpr (p < NSLOT) issue_any(p, into);
                stamp(sb + (h ? 12 : 1));
                if constexpr (p >= Cfg::LAZY && p - Cfg::LAZY < NSLOT) {
                    constexpr int i = p - Cfg::LAZY;             // this piece has had LAZY phases to arrive
                    vm_wait<(p < NSLOT ? p : NSLOT - 1) - i>();
                    if (nx_gn) commit_slot(i, into);
                }
            } else if constexpr (pdma) {                         // one-tap chunk: the whole compact patch of the next chunk
                if constexpr (h == 0) {
#pragma unroll
                    for (int i = 0; i < NSLOT1; ++i) issue_compact(lw + Cfg::NLAG * i, into);
                } else vm_wait<0>();
                stamp(sb + (h ? 12 : 1));
            }
            stamp(sb + (h ? 8 : 2));
            raw_barrier();
            // ================= C =================
            stamp(sb + (h ? 9 : 4));
            __builtin_amdgcn_sched_barrier(0);
            prio(1);
            if constexpr (!(ABL & 256)) {
                // ONE fragment read per MFMA gap (an in-order wave cannot issue its next MFMA behind a burst of reads that
                // queue for the LDS): MFMA i of k-group 0 uses fa0[i / 4], fb0[i % 4]; the second k-group's six fragments
                // arrive under MFMAs 1-6, then the first k-group of the NEXT phase refills fa0 / fb0 as the MFMAs release them
                const int rn = (ring_rd + WPHASE) & (RINGB - 1);
                auto next_a = [&](auto mi_) {                    // fa0[mi] <- first k-group of the next phase
                    constexpr int mi = decltype(mi_)::value;
                    if constexpr (h == 0) read_a(fa0[mi], rn, IC<2>{}, mi_);
                    else if constexpr (TP < NT - 1) read_a(fa0[mi], rn, IC<0>{}, mi_);
                    else if constexpr (NT == 9) { if (nx_ntaps == 9) read_a(fa0[mi], rn, IC<0>{}, mi_); }
                };
                auto next_b = [&](auto ni_) {
                    constexpr int ni = decltype(ni_)::value;
                    if constexpr (h == 0) read_b(fb0[ni], pb, IC<2>{}, Poff{}, Prow{}, ni_);
                    else if constexpr (TP < NT - 1) read_b(fb0[ni], pbase[DXN], IC<0>{}, PoffN{}, Prow{}, ni_);
                    else if constexpr (NT == 9) { if (nx_ntaps == 9) read_b(fb0[ni], pbase[0] + (par ? -PATCH_BYTES : PATCH_BYTES), IC<0>{}, IC<0>{}, Prow9{}, ni_); }
                };
                typedef IC<2 * h + 1> K1;
#define STORM_SB() __builtin_amdgcn_sched_barrier(0)
                if constexpr (WM * WN == 4) {
                // <128, 8>: four MFMAs per k-group (fa[i / 2], fb[i % 2]); the same rule - one or two reads per MFMA gap, a fragment
                // register is refilled right after the last MFMA that reads it
                mma_part(fa0, fb0, 0, 1); read_a(fa1[0], ring_rd, K1{}, IC<0>{}); read_b(fb1[0], pb, K1{}, Poff{}, Prow{}, IC<0>{}); STORM_SB();
                mma_part(fa0, fb0, 1, 2); read_b(fb1[1], pb, K1{}, Poff{}, Prow{}, IC<1>{}); read_a(fa1[1], ring_rd, K1{}, IC<1>{}); STORM_SB();
                mma_part(fa0, fb0, 2, 3); next_a(IC<0>{}); STORM_SB();          // fa0[0]: last used by MFMA 1
                mma_part(fa0, fb0, 3, 4); next_b(IC<0>{}); STORM_SB();          // fb0[0]: last used by MFMA 2
                mma_part(fa1, fb1, 0, 1); next_b(IC<1>{}); next_a(IC<1>{}); STORM_SB();   // fb0[1], fa0[1]: last used by MFMA 3
                mma_part(fa1, fb1, 1, WM * WN); STORM_SB();
                stamp(sb + (h ? 10 : 6));
                raw_barrier();
                } else {
                // fragment registers are refilled as early as the MFMAs release them, so that every LDS read of the interval
                // has returned by MFMA 12: the wave then ARRIVES at the interval barrier and issues its last four MFMAs behind
                // it - the barrier's latency (~95 cycles) and the partner group's start-up hide under them
                mma_part(fa0, fb0, 0, 1); read_a(fa1[0], ring_rd, K1{}, IC<0>{}); read_b(fb1[0], pb, K1{}, Poff{}, Prow{}, IC<0>{}); STORM_SB();
                mma_part(fa0, fb0, 1, 2); read_b(fb1[1], pb, K1{}, Poff{}, Prow{}, IC<1>{}); read_b(fb1[2], pb, K1{}, Poff{}, Prow{}, IC<2>{}); STORM_SB();
                mma_part(fa0, fb0, 2, 3); read_b(fb1[3], pb, K1{}, Poff{}, Prow{}, IC<3>{}); read_a(fa1[1], ring_rd, K1{}, IC<1>{}); STORM_SB();
                mma_part(fa0, fb0, 3, 4); STORM_SB();
                mma_part(fa0, fb0, 4, 5); next_a(IC<0>{}); STORM_SB();          // fa0[0]: last used by MFMA 3
                mma_part(fa0, fb0, 5, 6); next_b(IC<0>{}); STORM_SB();          // fb0[0]: last used by MFMA 4
                mma_part(fa0, fb0, 6, 7); next_b(IC<1>{}); STORM_SB();
                mma_part(fa0, fb0, 7, 8); next_b(IC<2>{}); STORM_SB();
                mma_part(fa1, fb1, 0, 1); next_b(IC<3>{}); next_a(IC<1>{}); STORM_SB();   // fb0[3], fa0[1]: last used by MFMA 7
                if constexpr ((ABL & 512) != 0) {                // (profiling A/B: arrive at the barrier before the last four MFMAs)
                    mma_part(fa1, fb1, 1, 4); STORM_SB();
                    stamp(sb + (h ? 10 : 6));
                    raw_barrier();                               // (lgkmcnt(0): this interval's reads are complete)
                    STORM_SB();
                    mma_part(fa1, fb1, 4, WM * WN);
                } else {
                    mma_part(fa1, fb1, 1, WM * WN); STORM_SB();
                    stamp(sb + (h ? 10 : 6));
                    raw_barrier();
                }
                }
#undef STORM_SB
            } else {
            mma_part(fa0, fb0, 0, 2);                            // the matrix pipe starts at once (operands were pre-read)
            __builtin_amdgcn_sched_barrier(0);
            read_frags(fa1, fb1, ring_rd, pb, IC<2 * h + 1>{}, Poff{}, Prow{});
            __builtin_amdgcn_sched_barrier(0);
            mma_part(fa0, fb0, 2, WM * WN);
            __builtin_amdgcn_sched_barrier(0);
            {   // first k-group of the NEXT phase into the registers the MFMAs above have consumed
                const int rn = (ring_rd + WPHASE) & (RINGB - 1);
                if constexpr (h == 0) read_frags(fa0, fb0, rn, pb, IC<2>{}, Poff{}, Prow{});
                else if constexpr (TP < NT - 1) read_frags(fa0, fb0, rn, pbase[DXN], IC<0>{}, PoffN{}, Prow{});
                else if constexpr (NT == 9) {                    // next chunk: the other buffer (a one-tap chunk reads in its own S)
                    if (nx_ntaps == 9) read_frags(fa0, fb0, rn, pbase[0] + (par ? -PATCH_BYTES : PATCH_BYTES), IC<0>{}, IC<0>{}, Prow9{});
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_part(fa1, fb1, 0, WM * WN);
            stamp(sb + (h ? 10 : 6));
            raw_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
            prio(0);
            ring_next();
        });
        stamp(sb + 11);
        if (TRACE) ++tstep;
    };

    // ---- tile start, part A: issue the first chunk's patch (lagging waves) and the first tap's weights (leading waves)
    // into patch buffer 0 / ring slots 0, 1 / table 0 (regions the previous tile's epilogue staging does not touch)
    auto tile_issue = [&]() {
        patch_table();
        load_next(0);
        enter_wrun(nx_wrun);
        w_soff = nx_wsoff;
        if (grp == 0) {                                     // phases 0 and 1 -> ring slots 0 and 1
            char* dst = smem + OFF_RING + lw * (NWD * 1024);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < NWD; ++j) dma16(w_srd, R[j], (uint32_t)(w_soff + h * WROW), dst + h * WPHASE + j * 1024, lane);
        } else {
            issue_table(0);
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) issue_slot(i, 0);
        }
    };
    // chunk change: the fetched buffer becomes current; descriptor of the chunk after the next.  Lane-derived address
    // math is re-laundered so that none of it is hoisted out of the loops and kept in registers.
    auto chunk_change = [&](int ci) {
        par ^= 1;
        const int dlt = par ? PATCH_BYTES : -PATCH_BYTES;
#pragma unroll
        for (int d = 0; d < 3; ++d) pbase[d] += dlt;
        load_next(ci + 1);
        launder(lane);
    };
    tile_issue();
    bool first = true;
    while (true) {
        // ---- tile start, part B: everything issued has landed; fused GroupNorm transform of the first patch ----------
        STORM_RELAUNDER();
        vm_wait<0>();
        if (grp == 1 && nx_gn) {
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) commit_slot(i, 0);
        }
        load_next(1);
        raw_barrier();
        read_frags(fa0, fb0, 0, pbase[0], IC<0>{}, IC<0>{}, Prow9{});   // first k-group of phase 0
        if (grp == 1) raw_barrier();                        // the lagging group starts one interval later
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
        if (first) stamp(2);

        // ---- main loops: the nine-tap chunks, then the one-tap chunks of a fused 1x1 shortcut ---------------------------
        // (two loops in sequence, not one loop with two bodies: the accumulators then have one home per loop)
        int ci = 0;
        for (; ci < n9; ++ci) {
            static_for<9>([&](auto t) { tap_step(t, IC<9>{}); });
            chunk_change(ci + 1);
        }
        for (; ci < nchunks; ++ci) {
            tap_step(IC<0>{}, IC<1>{});
            chunk_change(ci + 1);
        }
        if (grp == 0) raw_barrier();                        // balance the stagger: every wave has executed the same barriers

        // ---- hand-over: this tile's coordinates go to the epilogue; the next tile's first loads are issued -----------------
        if (first) stamp(500);
        vm_wait<0>();                                       // trailing ring / patch re-loads landed ...
        raw_barrier();                                      // ... and every wave is done reading: all of LDS is free
        if (first) stamp(501);
        const int e_tile = tile, e_b = b + eb_off, e_ty0 = ty0, e_tx0 = tx0, e_cout0 = cout0;
        const int e_H = imgH, e_W = imgW;                   // (GROUP: the next tile may belong to another problem)
        PipeArgPtr ap_e = ap;
        STORM_RELAUNDER();
        int nvb = vb + gridDim.x;
        while (nvb < total_vblocks && block_map(nvb, n_ct, tiles_per_xcd).tile >= ntiles) nvb += gridDim.x;
        const bool has_next = nvb < total_vblocks;
        if (has_next) {
            vb = nvb;
            decode(vb);
            if (par) {                                      // the next tile starts in patch buffer 0 / ring slot 0
                par = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) pbase[d] -= PATCH_BYTES;
            }
            ring_rd = 0;
            launder(lane);
            tile_issue();
        }

            STORM_RELAUNDER();
        if (ABL & 1024) {                                   // (profiling: no epilogue; the accumulators stay live)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) keep(acc[mi][ni]);
            if (!has_next) break;
            first = false;
            __syncthreads();
            continue;
        }
        // ---- epilogue (conv_epilogue.h).  Staging lives in patch buffer 1 (waves 0-4) and ring slots 2, 3 (waves 5-7; the statistics
        // scratch behind them; <128, 8>: a region behind the main image, PCfg::OFF_TAIL): the next tile's first loads are landing in
        // buffer 0 / slots 0, 1 meanwhile.
        constexpr int WSTAGE = Cfg::WSTAGE;
        char* const stage = smem + (wave < 5 ? PATCH_BYTES + wave * WSTAGE : Cfg::OFF_TAIL + (wave - 5) * WSTAGE);
        const epi::TileAt et = {e_tile, e_b, e_ty0, e_tx0, e_cout0};
        float gsum[8], gsq[8];
        if constexpr (GROUP) relaunder(ap_e); else ap_e = ap;
        epi::store_tile<T, WM, WN, (ABL & (2048 | 4096 | 8192))>(acc, stage, ap_e, et, wm, wn, lane, e_H, e_W, BN, TH, gsum, gsq);
        if (first) stamp(502);
        if (ap_e->gn_part != nullptr)
            epi::write_stats<WM, WN, WAVES_N, BN, TH>(gsum, gsq, reinterpret_cast<float*>(smem + Cfg::OFF_TAIL + 3 * WSTAGE), ap_e, et, wm, wn, lane, tid,
                                                      e_H, tiles_x, tiles_per_img);
        if (!has_next) break;
        first = false;
        __syncthreads();                                    // the statistics scratch / staging of this tile is free again
    }
    if (TRACE) { vm_wait<0>(); stamp(503); }
}

#undef STORM_RELAUNDER

// (the parameter block must stay the kernels' FIRST argument: the body reads it in place through the kernarg segment pointer)
template <typename T, int BN, int TH, int ABL>
__global__ __launch_bounds__(pipe::THREADS, 2)
void conv_pipe_kernel(const PipeParams a, const int n_ct, const int tiles_per_xcd,
                      const int ntiles, const int tiles_x, const int tiles_per_img, const int total_vblocks) {
    conv_pipe_body<T, BN, TH, ABL, false>(a, n_ct, tiles_per_xcd, ntiles, tiles_x, tiles_per_img, total_vblocks);
}
template <typename T, int BN, int TH>
__global__ __launch_bounds__(pipe::THREADS, 2)
void conv_pipe_splitk_kernel(const PipeParams a, const int n_ct, const int tiles_per_xcd,
                             const int ntiles, const int tiles_x, const int tiles_per_img, const int total_vblocks) {
    conv_pipe_body<T, BN, TH, 0, true>(a, n_ct, tiles_per_xcd, ntiles, tiles_x, tiles_per_img, total_vblocks);
}

template <typename T, int BN, int TH>
__global__ __launch_bounds__(pipe::THREADS, 2)
void conv_pipe_group_kernel(const PipeParams* gtab, const GroupTile* glist, const int n_ct, const int tiles_per_xcd, const int ntiles, const int total_vblocks) {
    conv_pipe_body<T, BN, TH, 0, false, true>(*gtab, n_ct, tiles_per_xcd, ntiles, 1, 1, total_vblocks, gtab, glist);
}

// ---- host side ---------------------------------------------------------------------------------------------------
// The K loop as chunk descriptors (declared in conv_pipe_common.h).
bool pipe::build_pipe_params(const storm_conv_args& a, PipeParams& p, const int kc) {
    const int pixb = 2 * kc;                                           // bytes per pixel and chunk
    memset(&p, 0, sizeof(p));
    if ((a.dtype != STORM_BF16 && a.dtype != STORM_F16) || a.nseg < 1 || a.seg[0].ntaps != 9) return false;
    int n = 0, nw = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const storm_conv_seg& g = a.seg[s];
        if ((g.ntaps != 9 && g.ntaps != 1) || g.w_bstride != 0) return false;
        if (g.ntaps == 1 && g.gn_ss != nullptr) return false;
        if (s > 0 && g.ntaps == 9) return false;                       // nine-tap chunks first (a one-tap chunk never precedes one)
        for (int part = 0; part < 2; ++part) {
            if (part == 1 && g.Cb == 0) break;
            if (nw >= 4) return false;
            const int C = part == 0 ? g.Ca : g.Cb, wc0 = part == 0 ? 0 : g.Ca;
            const long long bstride = part == 0 ? g.bstride_a : g.bstride_b;
            const long long img_bytes = (long long)a.H * a.W * C * 2;
            const long long w_bytes = ((long long)g.ntaps * g.w_tapstride - wc0) * 2;
            if (img_bytes >= (1LL << 31) || w_bytes >= (1LL << 31) || g.w_tapstride * 2 >= (1LL << 31)) return false;
            WRunDesc& R = p.wrun[nw];
            R.w = (unsigned long long)(reinterpret_cast<const char*>(g.w) + 2LL * wc0);
            R.bytes = (unsigned int)w_bytes; R.CinP2 = g.CinP * 2; R.rows = g.w_rows; R.tapbytes = (int)(g.w_tapstride * 2);
            const int nch = (C + kc - 1) / kc;
            for (int ch = 0; ch < nch; ++ch) {
                if (n >= MAX_CHUNKS) return false;
                ChunkDesc& d = p.chunk[n++];
                d.src = (unsigned long long)(part == 0 ? g.src_a : g.src_b);
                d.bstride = (unsigned long long)(bstride * 2);
                d.src_bytes = (unsigned int)img_bytes;
                d.C2 = C * 2; d.cbeg2 = ch * pixb; d.cvalid = C - ch * kc < kc ? C - ch * kc : kc;
                d.ntaps = g.ntaps; d.silu = g.gn_silu;
                if (g.gn_ss) {
                    d.ss = (unsigned long long)(g.gn_ss + 2 * (wc0 + ch * kc));
                    d.ss_bstride = (unsigned int)((g.Ca + g.Cb) * 8);
                }
                d.wrun = nw; d.w_soff = ch * pixb; d.new_wrun = ch == 0;
            }
            ++nw;
        }
    }
    ChunkDesc& t = p.chunk[n];                                         // terminator: the last chunk's prefetch target
    t = p.chunk[n - 1];
    t.src_bytes = 0; t.ss = 0; t.ntaps = 9; t.new_wrun = 0; t.cvalid = 0;
    p.nchunks = n; p.nchunks9 = 0;
    for (int i = 0; i < n; ++i) p.nchunks9 += p.chunk[i].ntaps == 9;
    p.B = a.B; p.H = a.H; p.W = a.W;
    p.out = a.out; p.outC = a.outC; p.Cout = a.Cout; p.out_bstride = a.out_bstride;
    p.bias = a.bias; p.tbias = a.tbias; p.tbias_stride = a.tbias_stride; p.out_f32 = a.out_f32;
    p.skip = a.skip; p.skip_bstride = a.skip_bstride; p.scale = a.scale;
    p.gn_part = a.gn_part;
    p.trace = nullptr;
    return true;
}

bool conv_pipe_supports(const storm_conv_args& a) {
    PipeParams p;
    return build_pipe_params(a, p, KC);
}

template <typename T, int BN, int TH, int ABL>
static int launch_pipe(const storm_conv_args& a, hipStream_t st) {
    typedef PCfg<BN, TH> Cfg;
    auto kern = conv_pipe_kernel<T, BN, TH, ABL>;
    static bool attr_set = false;                       // per instantiation; benign race (idempotent)
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    PipeParams prm;
    STORM_CHECK(build_pipe_params(a, prm, KC), "storm_conv: convolution outside the pipelined kernel's coverage");
    const int tiles_x = cdiv(a.W, TILE_W);
    const int tiles_per_img = tiles_x * cdiv(a.H, TH);
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int n_ct = cdiv(a.outC, BN);
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long vblocks = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(vblocks > 0 && vblocks < (1LL << 31), "storm_conv: grid %lld out of range", vblocks);
    const long long resident = (device_cus() + 7) / 8 * 8;             // one workgroup per CU; a multiple of 8
    const long long grid = vblocks < resident ? vblocks : resident;
#if defined(STORM_PROFILING)
    if (ABL & 64) {   // device buffer address handed over by tools/conv_trace.py
        prm.trace = reinterpret_cast<unsigned long long*>(switches().conv_trace_ptr);
    }
#endif
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), Cfg::LDS_BYTES, st, prm, n_ct, tiles_per_xcd, (int)ntiles,
                       tiles_x, tiles_per_img, (int)vblocks);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_conv_pipe(const storm_conv_args& a, hipStream_t st) {
    if (a.dtype == STORM_F16) return launch_pipe<half_t, 256, 8, 0>(a, st);
#if defined(STORM_PROFILING)
    switch (switches().conv_ablate) {
        case 8: return launch_pipe<bf16_t, 256, 8, 8>(a, st);
        case 16: return launch_pipe<bf16_t, 256, 8, 16>(a, st);
        case 32: return launch_pipe<bf16_t, 256, 8, 32>(a, st);
        case 128: return launch_pipe<bf16_t, 256, 8, 128>(a, st);
        case 184: return launch_pipe<bf16_t, 256, 8, 184>(a, st);       // barriers + scalar skeleton only
        case 136: return launch_pipe<bf16_t, 256, 8, 136>(a, st);       // no DMA of either kind
        case 152: return launch_pipe<bf16_t, 256, 8, 152>(a, st);       // MFMAs + barriers only
        case 64: return launch_pipe<bf16_t, 256, 8, 64>(a, st);
        case 256: return launch_pipe<bf16_t, 256, 8, 256>(a, st);       // fragment reads in two bursts per MFMA interval
        case 512: return launch_pipe<bf16_t, 256, 8, 512>(a, st);       // barrier arrival before the last four MFMAs
        case 4096: return launch_pipe<bf16_t, 256, 8, 4096>(a, st);     // non-temporal output stores
        case 1024: return launch_pipe<bf16_t, 256, 8, 1024>(a, st);     // no epilogue
        case 2048: return launch_pipe<bf16_t, 256, 8, 2048>(a, st);     // epilogue without global stores / skip loads
        case 8192: return launch_pipe<bf16_t, 256, 8, 8192>(a, st);     // epilogue arithmetic, no global stores
        default: break;
    }
#endif
    return launch_pipe<bf16_t, 256, 8, 0>(a, st);
}

// <128, 8>: the same kernel with 128 output channels per workgroup (choose_variant 9: layers with 128 ... 511 pixel tiles)
int launch_conv_pipe_half(const storm_conv_args& a, hipStream_t st) {
    if (a.dtype == STORM_F16) return launch_pipe<half_t, 128, 8, 0>(a, st);
#if defined(STORM_PROFILING)
    if (switches().conv_ablate == 64) return launch_pipe<bf16_t, 128, 8, 64>(a, st);
#endif
    return launch_pipe<bf16_t, 128, 8, 0>(a, st);
}

// ---- grouped launch (ragged micro-batches of one stream: BASELINE.json configs[4]) ---------------------------------------------
// The micro-batches of a ragged stream run the same network on 1 - 3 rows each: its deep levels are a few pixel tiles per launch, i.e.
// launches that last as long as one workgroup's chain of phases with most of the chip idle (the small-call regime).  Every op of the
// path is per image, so P micro-batches can share ONE launch of a layer: the tiles of all problems in one persistent walk, the
// problem's parameter block looked up per tile.  A tile computes exactly what it computes in its own launch (same chunk order).
long long conv_pipe_group_prepare(const storm_conv_args* a, int P, PipeParams* table, GroupTile* tiles, long long max_tiles) {
    if (P < 1) return -1;
    long long n = 0;
    for (int g = 0; g < P; ++g) {
        if (!build_pipe_params(a[g], table[g], KC)) return -1;
        // one layer: same K loop and weights in every problem (the descriptors differ in tensor addresses and extents only)
        if (a[g].outC != a[0].outC || a[g].Cout != a[0].Cout || a[g].dtype != a[0].dtype || table[g].nchunks != table[0].nchunks ||
            table[g].nchunks9 != table[0].nchunks9 || a[g].out_f32 != a[0].out_f32 || a[g].H >= 65536 || a[g].W >= 65536) return -1;
        for (int r = 0; r < 4; ++r) if (table[g].wrun[r].w != table[0].wrun[r].w) return -1;
        const int tiles_x = cdiv(a[g].W, TILE_W), tiles_y = cdiv(a[g].H, 8);
        for (int b = 0; b < a[g].B; ++b)
            for (int ty = 0; ty < tiles_y; ++ty)
                for (int tx = 0; tx < tiles_x; ++tx) {
                    if (n >= max_tiles) return -1;
                    GroupTile& t = tiles[n++];
                    t.problem = (unsigned)g; t.b = (unsigned)b;
                    t.yx = (unsigned)(ty * 8) | ((unsigned)(tx * TILE_W) << 16);
                    t.tile = (unsigned)((b * tiles_y + ty) * tiles_x + tx);
                }
    }
    return n;
}

template <typename T, int BN>
static int launch_group(const PipeParams* dev_table, const GroupTile* dev_tiles, long long ntiles, int outC, hipStream_t st) {
    typedef PCfg<BN, 8> Cfg;
    auto kern = conv_pipe_group_kernel<T, BN, 8>;
    static bool attr_set = false;
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    const int n_ct = cdiv(outC, BN);
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long vblocks = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(vblocks > 0 && vblocks < (1LL << 31), "storm_conv (group): grid %lld out of range", vblocks);
    const long long resident = (device_cus() + 7) / 8 * 8;
    const long long grid = vblocks < resident ? vblocks : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), Cfg::LDS_BYTES, st, dev_table, dev_tiles, n_ct, tiles_per_xcd, (int)ntiles, (int)vblocks);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_conv_pipe_group(const PipeParams* dev_table, const GroupTile* dev_tiles, long long ntiles, int outC, int bn, int dtype, hipStream_t st) {
    STORM_CHECK(dev_table && dev_tiles && ntiles > 0 && (bn == 256 || bn == 128), "storm_conv (group): bad arguments");
    if (dtype == STORM_F16) return bn == 256 ? launch_group<half_t, 256>(dev_table, dev_tiles, ntiles, outC, st) : launch_group<half_t, 128>(dev_table, dev_tiles, ntiles, outC, st);
    STORM_CHECK(dtype == STORM_BF16, "storm_conv (group): 16-bit operands only");
    return bn == 256 ? launch_group<bf16_t, 256>(dev_table, dev_tiles, ntiles, outC, st) : launch_group<bf16_t, 128>(dev_table, dev_tiles, ntiles, outC, st);
}

// ---- split-K (few-tile layers) ---------------------------------------------------------------------------------------------
// ncsnpplarge at configs[3] runs 44 of its 99 3x3 launches on 16 x 64 ... 4 x 16 pixel images (x 8 utterances): 64 ... 16 workgroups of
// the 128-cout tile, each walking the layer's whole K loop - 72 or 144 phases of ~0.36 us that nothing shortens but a split of K.
// Measured before it was built (tools/probe_splitk.py, profiles/r04_probe_splitk.txt: the slices emulated by the existing kernel over
// B * S images of Cin / S channels with fp32 output, the combine as a torch.sum over the slabs): 512 -> 256 @ 8 x 8 x 32 62.7 -> 23.3 us,
// 256 -> 256 @ 8 x 4 x 16 32.0 -> 17.5 us, 256 -> 256 @ 8 x 16 x 64 38.2 -> 31.0 us; nothing at 256 workgroups (16 x 32 x 64).
// Two launches, no cross-workgroup hand-off inside a launch: the slices write fp32 slabs [slice][B][H][W][outC] (the launch above
// in its SPLIT instantiation), then ONE combine pass sums them in slice order (bit-reproducible), applies the epilogue of
// conv_epilogue.h - (sum + skip) * scale + (bias + temb bias) * scale, rounded to the activation type - and writes the GroupNorm
// partials in the 8 x 32 tile layout every conv kernel uses.  32 pixels x 8 channel octets per workgroup: a wave reads whole 256-byte
// rows of a slab, the statistics are reduced over the pixels in a fixed order (lane exchanges, then the four waves through LDS).
template <typename T>
__global__ __launch_bounds__(256)
void splitk_combine_kernel(const float* __restrict__ slabs, const int S, const long long slab_stride, const int H, const int W, const int outC,
                           const int Cout, const float* __restrict__ bias, const float* __restrict__ tbias, const int tbias_stride,
                           const T* __restrict__ skip, const long long skip_bstride, const float scale, T* __restrict__ out,
                           const long long out_bstride, float* __restrict__ gn_part, const int tiles_x, const int tiles_y8) {
    __shared__ float red[4][64][2];
    const int t8 = blockIdx.x, tpi = tiles_x * tiles_y8;
    const int b = t8 / tpi, trem = t8 - b * tpi;
    const int ty0 = (trem / tiles_x) * TILE_H, tx0 = (trem % tiles_x) * TILE_W;
    const int tid = threadIdx.x, oct = tid & 7, px = tid >> 3, wave = tid >> 6;
    const int co = blockIdx.y * 64 + oct * 8, gx = tx0 + px;
    const bool ok = co < outC && gx < W;
    float badd[8], gsum[8], gsq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        badd[e] = 0.f; gsum[e] = 0.f; gsq[e] = 0.f;
        if (ok && co + e < Cout) {
            if (bias) badd[e] += bias[co + e];
            if (tbias) badd[e] += tbias[(long long)b * tbias_stride + co + e];
        }
        badd[e] *= scale;
    }
    const long long img = (long long)H * W * outC;
    const float* const sl = slabs + (long long)b * img;
    // a launch this small is a chain of memory round trips unless the loads are in flight together: two pixel rows per step, all
    // slabs of both rows (and the skip operand) issued before the first sum; rows past the image read the tile's first valid
    // address and are dropped
    constexpr int RB = 2;
#pragma unroll 1
    for (int r0 = 0; r0 < TILE_H; r0 += RB) {
        bool valid[RB];
        long long o[RB];
        float v[RB][8], sk[RB][8];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const int gy = ty0 + r0 + j;
            valid[j] = ok && gy < H;
            o[j] = valid[j] ? ((long long)gy * W + gx) * outC + co : 0;
            load8(sl + o[j], v[j]);
        }
        if (S == 4) {                                       // (the common cases unrolled: every load issued up front)
            float u[RB][3][8];
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int s = 1; s < 4; ++s) load8(sl + s * slab_stride + o[j], u[j][s - 1]);
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int s = 1; s < 4; ++s)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[j][e] += u[j][s - 1][e];
        } else {
            for (int s = 1; s < S; ++s) {                   // fixed order: slice 0, 1, ...
                float u[RB][8];
#pragma unroll
                for (int j = 0; j < RB; ++j) load8(sl + s * slab_stride + o[j], u[j]);
#pragma unroll
                for (int j = 0; j < RB; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[j][e] += u[j][e];
            }
        }
        if (skip != nullptr) {
#pragma unroll
            for (int j = 0; j < RB; ++j) load8(skip + (long long)b * skip_bstride + o[j], sk[j]);
#pragma unroll
            for (int j = 0; j < RB; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[j][e] += sk[j][e];
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            if (!valid[j]) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[j][e] = fmaf(v[j][e], scale, badd[e]);
                gsum[e] += v[j][e];
                gsq[e] = fmaf(v[j][e], v[j][e], gsq[e]);
            }
            store8(out + (long long)b * out_bstride + o[j], v[j]);
        }
    }
    if (gn_part == nullptr) return;
#pragma unroll
    for (int off = 8; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) { gsum[e] += __shfl_xor(gsum[e], off, 64); gsq[e] += __shfl_xor(gsq[e], off, 64); }
    if ((tid & 63) < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[wave][oct * 8 + e][0] = gsum[e]; red[wave][oct * 8 + e][1] = gsq[e]; }
    }
    __syncthreads();
    if (tid < 64 && blockIdx.y * 64 + tid < outC) {
        const float s0 = ((red[0][tid][0] + red[1][tid][0]) + red[2][tid][0]) + red[3][tid][0];
        const float s1 = ((red[0][tid][1] + red[1][tid][1]) + red[2][tid][1]) + red[3][tid][1];
        float* dst = gn_part + ((long long)t8 * outC + blockIdx.y * 64 + tid) * 2;
        dst[0] = s0; dst[1] = s1;
    }
}

// K slices for a layer, 0 = no split.  Two clauses:
//  (1) the IMAGE rule looks at one image only - its 128-cout tiles (pixel tiles x cout tiles <= 8: images up to 16 x 64 / 32 x 32 pixels at
//      256 couts) - never at the batch size.  At configs[3]'s 8 utterances per GPU that is <= 64 workgroups unsplit; at 128 ... 256
//      workgroups (the 32 x 64 level at batch 16) the split measured slower.  4 slices, 2 when there are fewer than four 64-channel chunks:
//      8 measured behind 4 wherever both apply (profiles/r04f_probe_splitk_s.txt; unsplit -> 4 slices): 256 -> 256 @ 8 x 16 x 64 39.2 -> 37.9 us,
//      512 -> 256 67.9 -> 43.8; @ 8 x 8 x 32 36.7 -> 25.1 and 63.7 -> 32.2; @ 8 x 4 x 16 32.2 -> 25.3 and 55.3 -> 26.5.
//  (2) the SMALL-CALL rule (round 5; STORM_SPLITK_SMALL=0 switches it off) looks at the call: at most 64 unsplit workgroups in the WHOLE
//      launch.  One utterance per call is the reference's own operating point (enhancement.py:66-72) and a ragged stream's tail batches hold
//      one to three rows: there the 32 x 64 level of NCSN++ is 16 workgroups per utterance and its twelve launches were 19 % of a batch-1
//      evaluation at 52 - 57 TFLOP/s (profiles/r05a_ops_b1.json), the 64 x 128 level (64 workgroups) 17 %.
// A split changes the fp32 summation order of K (slices summed in slice order: bit-reproducible, no atomics), so: under clause (1) alone an
// utterance comes out the same bits alone, in any batch and on any rank (test_batch_independence_and_determinism); clause (2) is - like
// the 256- / 128-cout tile and conv_pipe128 selections of choose_variant - a throughput decision on the launch, and across ITS threshold
// (a one-utterance call against the same utterance in a batch of sixteen) a row agrees to the rounding of its 16-bit activations, not
// bit for bit; run-to-run and rank-to-rank results of the same call stay bit-identical.
int conv_splitk_slices(const storm_conv_args& a) {
    if (!conv_pipe_supports(a) || a.outC <= 128 || a.out_f32) return 0;
    const int n9 = cdiv(a.seg[0].Ca, KC) + (a.seg[0].Cb ? cdiv(a.seg[0].Cb, KC) : 0);
    const int forced = switches().splitk;                    // (A/B hook)
    if (forced == 1) return 0;
    if (forced >= 2) return forced <= n9 ? forced : 0;
    const long long per_image = (long long)cdiv(a.W, TILE_W) * cdiv(a.H, TILE_H) * cdiv(a.outC, 128);
    if (per_image > 8 && (switches().splitk_small == 0 || switches().batch_invariant != 0 || per_image * a.B > 64)) return 0;
    return n9 >= 4 ? 4 : n9 >= 2 ? 2 : 0;
}
long long conv_splitk_bytes(const storm_conv_args& a, int slices) {
    return slices < 2 ? 0 : (long long)slices * a.B * a.H * a.W * a.outC * 4;
}

template <typename T>
static int launch_splitk(const storm_conv_args& a, const int S, hipStream_t st) {
    constexpr int BN = 128, TH = 8;
    typedef PCfg<BN, TH> Cfg;
    auto kern = conv_pipe_splitk_kernel<T, BN, TH>;
    static bool attr_set = false;
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
        attr_set = true;
    }
    PipeParams prm;
    STORM_CHECK(build_pipe_params(a, prm, KC), "storm_conv: convolution outside the pipelined kernel's coverage");
    STORM_CHECK(S >= 2 && S <= prm.nchunks9 && a.splitk_ws != nullptr && a.splitk_ws_bytes >= conv_splitk_bytes(a, S) && !a.out_f32,
                "storm_conv: split-K launch with %d slices of %d nine-tap chunks, workspace %lld bytes", S, prm.nchunks9, a.splitk_ws_bytes);
    const long long img = (long long)a.H * a.W * a.outC;
    prm.out = a.splitk_ws; prm.out_f32 = 1; prm.out_bstride = img;                // slab s = "batch items" s * B ...
    prm.bias = nullptr; prm.tbias = nullptr; prm.skip = nullptr; prm.scale = 1.0f; prm.gn_part = nullptr;
    const int tiles_x = cdiv(a.W, TILE_W), tiles_y = cdiv(a.H, TH);
    const int tiles_per_img = tiles_x * tiles_y;
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int n_ct = cdiv(a.outC, BN);
    prm.kslices = S; prm.split_nct = n_ct;
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long vblocks = 8LL * tiles_per_xcd * n_ct * S;
    STORM_CHECK(vblocks > 0 && vblocks < (1LL << 31) && ntiles < (1LL << 31), "storm_conv: split-K grid %lld out of range", vblocks);
    const long long resident = (device_cus() + 7) / 8 * 8;
    const long long grid = vblocks < resident ? vblocks : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), Cfg::LDS_BYTES, st, prm, n_ct * S, tiles_per_xcd, (int)ntiles,
                       tiles_x, tiles_per_img, (int)vblocks);
    STORM_LAUNCH_CHECK();
    hipLaunchKernelGGL(splitk_combine_kernel<T>, dim3((unsigned)ntiles, (unsigned)cdiv(a.outC, 64)), dim3(256), 0, st,
                       static_cast<const float*>(a.splitk_ws), S, (long long)a.B * img, a.H, a.W, a.outC, a.Cout, a.bias, a.tbias,
                       a.tbias_stride, static_cast<const T*>(a.skip), a.skip_bstride, a.scale, static_cast<T*>(a.out), a.out_bstride,
                       a.gn_part, tiles_x, tiles_y);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_conv_pipe_splitk(const storm_conv_args& a, int slices, hipStream_t st) {
    if (a.dtype == STORM_F16) return launch_splitk<half_t>(a, slices, st);
    return launch_splitk<bf16_t>(a, slices, st);
}

const char* conv_pipe_kernel_name(int dtype, bool half_tile) {
    if (half_tile) return dtype == STORM_F16 ? "storm::conv_pipe_kernel<storm::half_t, 128, 8, 0>" : "storm::conv_pipe_kernel<storm::bf16_t, 128, 8, 0>";
    return dtype == STORM_F16 ? "storm::conv_pipe_kernel<storm::half_t, 256, 8, 0>" : "storm::conv_pipe_kernel<storm::bf16_t, 256, 8, 0>";
}

}  // namespace storm
