// Producer / consumer 3x3 implicit-GEMM convolution for layers with <= 128 output channels (bf16 / fp16 operands): the kernel
// behind the 128-cout 3x3 layers of NCSN++ (layers.py:119-126 ddpm_conv3x3 with the fused pieces of layerspp.py:242-274
// listed in include/storm_hip.h).  Same math, arguments, K order (64-channel chunk, tap, 16-channel k-group) and epilogue
// arithmetic as conv_igemm.hip / conv_pipe.hip: the stored activations are bit-identical to theirs.
//
// Why another structure.  With 128 output channels every non-MFMA cost of a tile weighs twice as much per MFMA as in the
// 256-cout kernel (patch bytes, fused GroupNorm transform, epilogue, weight traffic), and in the ping-pong kernels those costs
// are ADDITIVE: every wave owns accumulators, so whatever a wave does besides MFMAs - waiting for DMA, transforming the patch,
// storing a tile - stops its share of the matrix pipe (conv_igemm 0.48 matrix-pipe-busy, conv_pipe128 0.44 with the fused
// transform, 0.63 without; profiles/r02_*).  Here the roles are separate WAVES:
//   * waves 0-3, one per SIMD ("consumers"): 64 couts x (4 x 32) pixels each = 8 accumulator tiles of v_mfma_f32_32x32x16;
//     their instruction stream is MFMAs, pixel-fragment reads from LDS and weight-fragment loads - nothing else.  Weights
//     never pass through LDS: an A fragment (32 couts x 16 channels) is exactly one 16-byte-per-lane buffer load from the packed
//     [tap][cout][cin] matrix (L2 / L1 resident: 16 KiB per tap and chunk, shared by the two waves of a cout half), issued three
//     k-groups ahead into a four-deep register ring.  No weight ring in LDS, no DMA instructions for weights, no per-phase
//     synchronisation: consumers and producers meet at ONE barrier per 64-channel chunk (18 x 16 MFMAs per wave).
//   * waves 4-7, the second wave of every SIMD ("producers"): fetch the next chunk's haloed patch (all of its 1-KiB pieces in
//     flight at once, `buffer_load_dwordx4 ... lds`), apply the fused GroupNorm + SiLU in place, and run the previous tile's
//     epilogue - bias / temb bias / skip / scale, statistics, 16-byte stores - out of an fp32 staging area the consumers dump
//     their accumulators into.  Their VALU / LDS / VMEM work issues beside the consumers' MFMAs on the same SIMD.
//   * tile epilogue: a consumer writes two of its four pixel rows to staging, the producers drain them (the only time the
//     matrix pipe waits: two passes of ~0.7 k cycles per tile), it writes the other two and starts the next tile, whose patch
//     and first weight fragments are already there; the producers drain the second half under the next tile's first chunk.
//     A producer takes a 32-cout slice of all eight pixel rows, so a tile's per-channel statistics complete inside one wave.
// LDS (152 KiB, one workgroup per CU): two 43-KiB patch buffers, two (scale, shift) tables, 64 KiB of staging (two passes of
// four consumer waves).  Barriers per tile: one per chunk + three around the epilogue hand-off; every wave executes all of them.
#include <cstring>
#include "conv_pipe_common.h"

namespace storm {
using namespace cidx;

namespace pc {
using namespace pipe;

constexpr int BN = 128, TH = 8, KC = 64, PIXB = 128;          // couts x pixel rows per workgroup; channels / bytes per pixel and chunk
constexpr int PW = TILE_W + 2;
constexpr int THREADS = 512, NPROD = 4;
constexpr int WM = 2, WN = 4;                                 // consumer wave tile: 2 x 32 couts, 4 pixel rows of 32
constexpr int NPIX = (TH + 2) * PW;
constexpr int PPIECES = (NPIX + 7) / 8;                       // 1-KiB DMA pieces (8 pixels) of a haloed patch: 43
constexpr int PATCH_BYTES = PPIECES * 1024;
constexpr int CPIECES = TH * 4;                               // pieces of a compact (one-tap) image: 32
constexpr int NSLOT = (PPIECES + NPROD - 1) / NPROD;          // haloed pieces per producer wave: 11
constexpr int NSLOT1 = CPIECES / NPROD;                       // compact pieces per producer wave: 8
constexpr int OFF_SS = 2 * PATCH_BYTES;                       // two 1-KiB (scale, shift) tables
constexpr int OFF_STAGE = OFF_SS + 2048;
constexpr int WSTAGE = 32 * WM * 128;                         // one pass of one consumer wave: fp32 [32 px][64 couts] = 8 KiB
constexpr int PASS_BYTES = 4 * WSTAGE;
constexpr int LDS_BYTES = OFF_STAGE + 2 * PASS_BYTES;
constexpr int AHEAD = 3, RING = 4;                            // weight fragments: k-groups of lookahead / register ring depth
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(PATCH_BYTES % 256 == 0, "k-group XOR must stay inside the slot field");
static_assert(NSLOT * NPROD - PPIECES < NPROD && CPIECES % NPROD == 0 && NSLOT1 <= NSLOT, "piece slots");

// patch image: as conv_pipe.hip (pixel row of 128 B, 16-B slots XOR-swizzled by the pixel COLUMN)
STORM_HD int p_swz(int px, int slot) { return (slot ^ ((px >> 1) & 7)) << 4; }

struct Tile { int tile, b, ty0, tx0, cout0; };

}  // namespace pc
using namespace pc;

// ABL (profiling build only, work-skipping instantiations for the cycle budget): 1 = the weight-fragment loads read 1 KiB of
// CONTIGUOUS bytes per instruction (wrong values: what a fragment-major weight layout would cost the vector memory path),
// 2 = no pixel-fragment reads, 4 = no weight-fragment loads, 8 = no patch DMA / transform, 16 = no epilogue, 32 = no MFMAs
template <typename T, int ABL = 0>
__global__ __launch_bounds__(pc::THREADS, 2)
void conv_pc_kernel(const PipeParams a, const int n_ct, const int tiles_per_xcd, const int ntiles, const int tiles_x,
                    const int tiles_per_img, const int total_vblocks) {
    typedef typename Mma<T>::Frag Frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PipeArgPtr ap = pipe_args(a);

    // persistent workgroups, at most one per CU, walking the XCD-aware virtual block ids (conv_pipe.hip)
    auto next_vb = [&](int v) {
        while (v < total_vblocks && block_map(v, n_ct, tiles_per_xcd).tile >= ntiles) v += gridDim.x;   // (padding ids of the XCD map)
        return v;
    };
    auto decode = [&](int v) {
        const BlockMap bm = block_map(v, n_ct, tiles_per_xcd);
        Tile t;
        t.tile = bm.tile;
        t.b = bm.tile / tiles_per_img;
        const int trem = bm.tile - t.b * tiles_per_img;
        t.ty0 = (trem / tiles_x) * TH;
        t.tx0 = (trem % tiles_x) * TILE_W;
        t.cout0 = bm.ct * BN;
        return t;
    };
    int vb = next_vb(blockIdx.x);
    if (vb >= total_vblocks) return;
    const int imgH = pin(ap->H), imgW = pin(ap->W);
    const int nchunks = pin(ap->nchunks), n9 = pin(ap->nchunks9);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int w4 = wave & 3;

    if (wave < 4) {
        // ======================================= consumers ===========================================================
        prio(1);
        const int wm = w4 & 1, wn = w4 >> 1;                    // cout half (64) / pixel-row half (4 rows)
        f32x16 acc[WM][WN];
        Frag ar[RING][WM];                                      // weight fragments of k-groups s .. s + 3
        Frag fb[2][WN];                                         // pixel fragments of k-groups s, s + 1
        // pixel fragments: this lane's pixel of row ni = 0 under tap (0, 0) + the k-group-0 swizzle term of tap column dx
        int pbase[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) pbase[d] = ((wn * WN) * PW + (lane & 31)) * PIXB + p_swz((lane & 31) + d, lane >> 5);
        const int cdelta = wn * WN * (PW - TILE_W) * PIXB;      // haloed row index - compact row index of this wave's pixels

        // weight-fragment stream: the A operand of (chunk, tap, k-group kg, cout tile mi) is ONE buffer load of 16 B per lane
        // at voff[mi] + 32 kg (row = cout, lane half = 8 channels) + the scalar offset of (chunk, tap)
        struct ACtx { BufRsrc buf; uint32_t voff[WM]; int soff0, tapbytes; };
        auto make_actx = [&](int ci, const Tile& t, bool valid) {
            const ChunkDesc& d = ap->chunk[ci < nchunks ? ci : nchunks];
            const WRunDesc& W = ap->wrun[d.wrun];
            ACtx c;
            c.buf = make_buf(reinterpret_cast<const void*>(W.w), valid ? W.bytes : 0u);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                const int row = t.cout0 + wm * (WM * 32) + mi * 32 + (lane & 31);     // rows past the matrix: zeros (never stored)
                c.voff[mi] = row < W.rows ? (uint32_t)(row * W.CinP2 + (lane >> 5) * 16) : BUF_OOB;
                if (ABL & 1) c.voff[mi] = (uint32_t)((t.cout0 + wm * (WM * 32) + mi * 32) * W.CinP2 + lane * 16);
            }
            c.soff0 = d.w_soff; c.tapbytes = W.tapbytes;
            return c;
        };
        auto a_load = [&](const ACtx& c, auto tap_, auto kg_, auto slot_) {
            constexpr int tap = decltype(tap_)::value, kg = decltype(kg_)::value, slot = decltype(slot_)::value;
            if (ABL & 4) return;
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) {
                const uint4 v = buf_load16(c.buf, c.voff[mi] + 32u * kg, (uint32_t)(c.soff0 + tap * c.tapbytes));
                ar[slot][mi] = *reinterpret_cast<const Frag*>(&v);
            }
        };
        auto read_b = [&](Frag& f, int pb, auto kg_, auto poff_, auto prow_, auto ni_) {
            constexpr int kg = decltype(kg_)::value, POFF = decltype(poff_)::value, PROW = decltype(prow_)::value, ni = decltype(ni_)::value;
            if (ABL & 2) return;
            f = *reinterpret_cast<const Frag*>(smem + (pb ^ (kg << 5)) + POFF + ni * PROW);
        };
        typedef IC<PW * PIXB> Prow9; typedef IC<TILE_W * PIXB> Prow1;

        ACtx cur, nx;
        // k-group step S of a chunk with NT taps (4 k-groups per tap): 8 MFMAs; the weight fragments of step S + 3 are requested
        // first, the pixel fragments of step S + 1 are read one per MFMA gap
        auto step = [&](auto s_, auto nt_) {
            constexpr int S = decltype(s_)::value, NT = decltype(nt_)::value, NS = 4 * NT;
            constexpr int U = S + AHEAD;
            if constexpr (U < NS) a_load(cur, IC<U / 4>{}, IC<U % 4>{}, IC<U % RING>{});
            else a_load(nx, IC<0>{}, IC<U - NS>{}, IC<U % RING>{});
            __builtin_amdgcn_sched_barrier(0);
            constexpr int S1 = S + 1, TP1 = S1 / 4, KG1 = S1 % 4;
            constexpr int DX1 = NT == 9 ? TP1 % 3 : 0;
            constexpr int POFF1 = NT == 9 ? ((TP1 / 3) * PW + DX1) * PIXB : 0;
            typedef std::conditional_t<NT == 9, Prow9, Prow1> Prow;
            const int pb1 = NT == 9 ? pbase[DX1] : pbase[0] - cdelta;
            Frag (&fa)[WM] = ar[S % RING];
            Frag (&fc)[WN] = fb[S & 1];
            Frag (&fn)[WN] = fb[S1 & 1];
            auto mma = [&](auto i_) { constexpr int i = decltype(i_)::value; if (!(ABL & 32)) Mma<T>::run(fa[i / WN], fc[i % WN], acc[i / WN][i % WN]); };
            auto rd = [&](auto ni_) { if constexpr (S1 < NS) read_b(fn[decltype(ni_)::value], pb1, IC<KG1>{}, IC<POFF1>{}, Prow{}, ni_); };
            if constexpr ((ABL & 64) != 0 && S > 0 && S % 2 == 0) raw_barrier();      // (profiling: what a barrier per phase would cost)
            mma(IC<0>{}); rd(IC<0>{}); __builtin_amdgcn_sched_barrier(0);
            mma(IC<1>{}); rd(IC<1>{}); __builtin_amdgcn_sched_barrier(0);
            mma(IC<2>{}); rd(IC<2>{}); __builtin_amdgcn_sched_barrier(0);
            mma(IC<3>{}); rd(IC<3>{}); __builtin_amdgcn_sched_barrier(0);
            mma(IC<4>{}); mma(IC<5>{}); __builtin_amdgcn_sched_barrier(0);
            mma(IC<6>{}); mma(IC<7>{}); __builtin_amdgcn_sched_barrier(0);
        };
        auto first_b = [&](auto nt_) {                          // pixel fragments of step 0 (after the chunk barrier)
            constexpr int NT = decltype(nt_)::value;
            typedef std::conditional_t<NT == 9, Prow9, Prow1> Prow;
            const int pb = NT == 9 ? pbase[0] : pbase[0] - cdelta;
            read_b(fb[0][0], pb, IC<0>{}, IC<0>{}, Prow{}, IC<0>{}); read_b(fb[0][1], pb, IC<0>{}, IC<0>{}, Prow{}, IC<1>{});
            read_b(fb[0][2], pb, IC<0>{}, IC<0>{}, Prow{}, IC<2>{}); read_b(fb[0][3], pb, IC<0>{}, IC<0>{}, Prow{}, IC<3>{});
        };
        // accumulators of pixel row `pass` -> this wave's staging block of pass parity `pass & 1` (layout: conv_igemm.hip's stage_off)
        auto stage_pass = [&](auto pass_) {
            constexpr int pass = decltype(pass_)::value;
            if (ABL & 16) return;
            char* const stage = smem + OFF_STAGE + (pass & 1) * PASS_BYTES + w4 * WSTAGE;
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& c = acc[mi][pass];
                    *reinterpret_cast<float4*>(stage + stage_off<WM>(lane & 31, stage_wslot(lane, mi, g))) =
                        make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                }
        };

        Tile t = decode(vb);
        cur = make_actx(0, t, true);
        a_load(cur, IC<0>{}, IC<0>{}, IC<0>{}); a_load(cur, IC<0>{}, IC<1>{}, IC<1>{}); a_load(cur, IC<0>{}, IC<2>{}, IC<2>{});
        int par = 0;
        while (true) {
            const int nvb = next_vb(vb + gridDim.x);
            const bool has_next = nvb < total_vblocks;
            Tile tn = t;
            if (has_next) tn = decode(nvb);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
            auto chunk_change = [&](int ci) {                   // after chunk ci: the other patch buffer, the next weight context
                par ^= 1;
                const int dlt = par ? PATCH_BYTES : -PATCH_BYTES;
#pragma unroll
                for (int d = 0; d < 3; ++d) pbase[d] += dlt;
                cur = nx;
            };
            auto next_ctx = [&](int ci) {                       // the weight context of the chunk after ci (or the next tile's first)
                relaunder(ap);
                nx = ci + 1 < nchunks ? make_actx(ci + 1, t, true) : make_actx(0, tn, has_next);
            };
            int ci = 0;
            for (; ci < n9; ++ci) {
                next_ctx(ci);
                raw_barrier();                                  // chunk barrier: this chunk's patch is in buffer `par`
                first_b(IC<9>{});
                static_for<36>([&](auto s) { step(s, IC<9>{}); });
                chunk_change(ci);
            }
            for (; ci < nchunks; ++ci) {
                next_ctx(ci);
                raw_barrier();
                first_b(IC<1>{});
                static_for<4>([&](auto s) { step(s, IC<1>{}); });
                chunk_change(ci);
            }
            // ---- epilogue hand-off: two pixel rows -> staging, the producers drain them, the other two, next tile --------------
            raw_barrier();                                      // E0: the producers are done with the previous tile's staging
            stage_pass(IC<0>{}); stage_pass(IC<1>{});
            raw_barrier();                                      // E1: passes 0, 1 staged
            raw_barrier();                                      // E2: passes 0, 1 drained
            stage_pass(IC<2>{}); stage_pass(IC<3>{});
            if (!has_next) break;
            vb = nvb; t = tn;
        }
        raw_barrier();                                          // F: passes 2, 3 of the last tile staged
        return;
    }

    // =========================================== producers ==============================================================
    const int qi = w4;                                          // this wave's 32-cout slice of the tile; its patch pieces k = qi + 4 i
    // R[i] = haloed patch piece qi + 4 i of the tile whose chunks are being fetched: (pixel index << 3) | logical 16-B slot
    // that lands in this lane's physical slot, or -1 (padding / past the patch: hardware zero fill)
    uint32_t R[NSLOT];
    Tile it = decode(vb);                                       // the tile whose chunks are being FETCHED
    auto patch_table = [&]() {
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int k = qi + NPROD * i;
            const int row = k * 8 + (lane >> 3);
            const int py = row / PW, px = row - py * PW;
            const int slot = (lane & 7) ^ ((px >> 1) & 7);
            const int gy = it.ty0 + py - 1, gx = it.tx0 + px - 1;
            const bool ok = k < PPIECES && row < NPIX && gy >= 0 && gy < imgH && gx >= 0 && gx < imgW;
            R[i] = ok ? (uint32_t)(((gy * imgW + gx) << 3) | slot) : 0xffffffffu;
        }
    };
    // fetch state: the chunk being fetched
    u32x4 f_srd, f_ss_srd;
    int f_C2 = 0, f_cbeg2 = 0, f_cvalid = 0, f_ntaps = 9, f_gn = 0, f_silu = 0;
    auto load_desc = [&](int ci) {
        relaunder(ap);
        const ChunkDesc& d = ap->chunk[ci];
        f_srd = make_srd(reinterpret_cast<const char*>(d.src + (unsigned long long)it.b * d.bstride), d.src_bytes);
        f_gn = d.ss != 0ull;
        f_ss_srd = make_srd(reinterpret_cast<const char*>(f_gn ? d.ss + (unsigned long long)it.b * d.ss_bstride : d.src),
                            f_gn ? (uint32_t)d.cvalid * 8u : 0u);
        f_C2 = d.C2; f_cbeg2 = d.cbeg2; f_cvalid = d.cvalid; f_ntaps = d.ntaps; f_silu = d.silu;
    };
    // the whole patch of chunk ci of tile `it` -> buffer `into`: every piece of this wave in flight at once, then - in issue
    // order - wait, and apply the fused GroupNorm (+ SiLU) in place by the lane that fetched the 16 bytes
    auto fetch_chunk = [&](int ci, int into) {
        if (ABL & 8) return;
        load_desc(ci);
        vm_wait<0>();                                           // (stores of an epilogue pass issued before: counted waits start clean)
        char* const pbuf = smem + into * PATCH_BYTES;
        if (f_ntaps == 9) {
            // every producer fetches the (identical) table, so that its own vmcnt orders it before its transforms
            dma16(f_ss_srd, (uint32_t)lane * 16u, 0u, smem + OFF_SS + into * 1024, lane);
#pragma unroll
            for (int i = 0; i < NSLOT; ++i) {
                const int k = qi + NPROD * i;
                if (k < PPIECES) {
                    const uint32_t v = R[i];
                    const bool ok = (int)v >= 0 && (int)(v & 7u) * 8 < f_cvalid;
                    dma16(f_srd, ok ? mad24(v >> 3, (uint32_t)f_C2, (v & 7u) * 16u) : OOB, (uint32_t)f_cbeg2, pbuf + k * 1024, lane);
                } else {
                    dma16(f_ss_srd, (uint32_t)lane * 16u, 0u, smem + OFF_SS + into * 1024, lane);   // surplus slot: the table again (uniform VMEM count)
                }
            }
            static_for<NSLOT>([&](auto i_) {
                constexpr int i = decltype(i_)::value;
                vm_wait<NSLOT - 1 - i>();
                const int k = qi + NPROD * i;
                const uint32_t v = R[i];
                if (f_gn && k < PPIECES && (int)v >= 0 && (int)(v & 7u) * 8 < f_cvalid) {
                    uint4* const q = reinterpret_cast<uint4*>(pbuf + k * 1024 + lane * 16);
                    float ss[16];
                    load_ss<8>(reinterpret_cast<const float*>(smem + OFF_SS + into * 1024), (int)(v & 7u), ss);
                    *q = gn_act_slot(*q, ss, f_silu, (T*)nullptr);
                }
            });
        } else {
            // compact image (one-tap chunk: TH x 32 pixels, no halo, never transformed): piece k = pixel row k >> 2, columns 8 (k & 3) ..
#pragma unroll
            for (int i = 0; i < NSLOT1; ++i) {
                const int k = qi + NPROD * i;
                const int trow = k >> 2, n = (k & 3) * 8 + (lane >> 3);
                const int slot = (lane & 7) ^ ((n >> 1) & 7);
                const int gy = it.ty0 + trow, gx = it.tx0 + n;
                const bool ok = gy < imgH && gx < imgW && slot * 8 < f_cvalid;
                dma16(f_srd, ok ? mad24((uint32_t)(gy * imgW + gx), (uint32_t)f_C2, (uint32_t)slot * 16u) : OOB, (uint32_t)f_cbeg2,
                      pbuf + k * 1024, lane);
            }
            vm_wait<0>();
        }
    };

    // ---- epilogue: drain staged passes of tile `dr` - this wave's 32 couts (slice qi) of both pixel-row halves ------------------
    Tile dr = it;
    f32x2 gsum2[4], gsq2[4];
    auto drain_begin = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) { gsum2[i] = f32x2{0.f, 0.f}; gsq2[i] = f32x2{0.f, 0.f}; }
    };
    // passes P0, P0 + 1 (pixel rows P0, P0 + 1 of each half): out = (acc + bias + temb bias + skip) * scale, evaluated as
    // (acc [+ skip]) * scale + (bias * scale) in packed fma exactly as conv_pipe.hip / conv_igemm.hip do
    auto drain = [&](int P0) {
        if (ABL & 16) return;
        relaunder(ap);
        const int outC = pin(ap->outC), out_f32 = pin(ap->out_f32);
        const bool has_skip = ap->skip != nullptr;
        char* const out_b = as_global(reinterpret_cast<unsigned long long>(ap->out) +
                                      (unsigned long long)((long long)dr.b * ap->out_bstride * (out_f32 ? 4 : (int)sizeof(T))));
        const int c8 = lane & 3, l16 = lane >> 2;               // cout octet of the slice / pixel (of 16 per iteration) of this lane
        const int co = dr.cout0 + qi * 32 + c8 * 8;
        float badd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) badd[e] = 0.f;
        if (co + 8 <= ap->Cout) {
            if (ap->bias) { float bb[8]; load8(ap->bias + co, bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) badd[e] += bb[e]; }
            if (ap->tbias) { float bb[8]; load8(ap->tbias + (long long)dr.b * ap->tbias_stride + co, bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) badd[e] += bb[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (co + e < ap->Cout) {
                    if (ap->bias) badd[e] += ap->bias[co + e];
                    if (ap->tbias) badd[e] += ap->tbias[(long long)dr.b * ap->tbias_stride + co + e];
                }
        }
        const bool co_ok = co < outC;
        const T* const skip_b = reinterpret_cast<const T*>(ap->skip) + (long long)dr.b * ap->skip_bstride;
        const f32x2 scale2 = {ap->scale, ap->scale};
        f32x2 badd2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) badd2[i] = f32x2{badd[2 * i] * ap->scale, badd[2 * i + 1] * ap->scale};
        const int mi = qi & 1;                                  // the slice is cout tile mi of consumer waves (wm = qi >> 1, wn = 0, 1)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pass = P0 + pp;
#pragma unroll
            for (int wn = 0; wn < 2; ++wn) {
                const char* const stage = smem + OFF_STAGE + (pass & 1) * PASS_BYTES + ((qi >> 1) + 2 * wn) * WSTAGE;
                const int gy = dr.ty0 + wn * WN + pass;
                if (gy >= imgH) continue;
                const uint32_t o_row = (uint32_t)(gy * imgW) * (uint32_t)outC;
#pragma unroll
                for (int itn = 0; itn < 2; ++itn) {
                    const int row = itn * 16 + l16;             // pixel of the staged row
                    const float4 v0 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, mi * 8 + 2 * c8));
                    const float4 v1 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, mi * 8 + 2 * c8 + 1));
                    f32x2 v2[4] = {f32x2{v0.x, v0.y}, f32x2{v0.z, v0.w}, f32x2{v1.x, v1.y}, f32x2{v1.z, v1.w}};
                    const int gx = dr.tx0 + row;
                    if (gx < imgW && co_ok) {
                        const uint32_t o = o_row + (uint32_t)gx * (uint32_t)outC + (uint32_t)co;
                        if (has_skip) {
                            alignas(16) T sk[8];
                            *reinterpret_cast<uint4*>(sk) = *reinterpret_cast<const uint4*>(skip_b + o);
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
                        if (out_f32) store8(reinterpret_cast<float*>(out_b) + o, v);
                        else store8(reinterpret_cast<T*>(out_b) + o, v);
                    }
                }
            }
        }
    };
    // the tile's per-channel (sum, sum of squares): lanes of one cout octet are 4 apart
    auto drain_end = [&]() {
        relaunder(ap);
        if (ap->gn_part == nullptr) return;
        float gs[8], gq[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { gs[2 * i] = gsum2[i].x; gs[2 * i + 1] = gsum2[i].y; gq[2 * i] = gsq2[i].x; gq[2 * i + 1] = gsq2[i].y; }
#pragma unroll
        for (int off = 4; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) { gs[e] += __shfl_xor(gs[e], off, 64); gq[e] += __shfl_xor(gq[e], off, 64); }
        if (lane < 4) {
            const int co = dr.cout0 + qi * 32 + lane * 8;
            float* dst = ap->gn_part + ((long long)dr.tile * ap->outC + co) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (co + e < ap->outC) { dst[2 * e] = gs[e]; dst[2 * e + 1] = gq[e]; }
        }
    };

    // ---- kernel prologue: the first tile's first patch -------------------------------------------------------------------
    patch_table();
    fetch_chunk(0, 0);
    int par = 0;
    bool pending = false;                                       // passes 2, 3 of tile `dr` wait in staging
    while (true) {
        const int nvb = next_vb(vb + gridDim.x);
        const bool has_next = nvb < total_vblocks;
        const Tile cur_tile = decode(vb);
        for (int ci = 0; ci < nchunks; ++ci) {
            raw_barrier();                                      // chunk barrier: buffer par ^ 1 is free, (ci == 0:) passes 2, 3 are staged
            if (ci == 0 && pending) { drain(2); drain_end(); pending = false; }
            if (ci + 1 < nchunks) fetch_chunk(ci + 1, par ^ 1);
            else if (has_next) { it = decode(nvb); patch_table(); fetch_chunk(0, par ^ 1); }
            par ^= 1;
            if (ABL & 64) { const int nb = ci < n9 ? 17 : 1; for (int k = 0; k < nb; ++k) raw_barrier(); }
        }
        raw_barrier();                                          // E0
        raw_barrier();                                          // E1: passes 0, 1 staged
        dr = cur_tile;
        drain_begin();
        drain(0);
        raw_barrier();                                          // E2: passes 0, 1 drained
        pending = true;
        if (!has_next) break;
        vb = nvb;
    }
    raw_barrier();                                              // F
    drain(2); drain_end();
}

// ---- host side ---------------------------------------------------------------------------------------------------
bool conv_pc_supports(const storm_conv_args& a) {
    if (a.outC > pc::BN) return false;
    PipeParams p;
    return pipe::build_pipe_params(a, p, pc::KC);
}

template <typename T, int ABL = 0>
static int launch_pc(const storm_conv_args& a, hipStream_t st) {
    auto kern = conv_pc_kernel<T, ABL>;
    static bool attr_set = false;                       // per instantiation; benign race (idempotent)
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, pc::LDS_BYTES));
        attr_set = true;
    }
    PipeParams prm;
    STORM_CHECK(a.outC <= pc::BN && pipe::build_pipe_params(a, prm, pc::KC), "storm_conv: convolution outside the producer / consumer kernel's coverage");
    const int tiles_x = cdiv(a.W, TILE_W);
    const int tiles_per_img = tiles_x * cdiv(a.H, pc::TH);
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int n_ct = 1;
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long vblocks = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(vblocks > 0 && vblocks < (1LL << 31), "storm_conv: grid %lld out of range", vblocks);
    const long long resident = (device_cus() + 7) / 8 * 8;             // one workgroup per CU; a multiple of 8
    const long long grid = vblocks < resident ? vblocks : resident;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pc::THREADS), pc::LDS_BYTES, st, prm, n_ct, tiles_per_xcd, (int)ntiles,
                       tiles_x, tiles_per_img, (int)vblocks);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_conv_pc(const storm_conv_args& a, hipStream_t st) {
#if defined(STORM_PROFILING)
    if (a.dtype == STORM_BF16)
        switch (switches().conv_ablate) {
            case 1: return launch_pc<bf16_t, 1>(a, st);
            case 2: return launch_pc<bf16_t, 2>(a, st);
            case 4: return launch_pc<bf16_t, 4>(a, st);
            case 6: return launch_pc<bf16_t, 6>(a, st);       // MFMAs + patch pipeline + epilogue
            case 8: return launch_pc<bf16_t, 8>(a, st);
            case 14: return launch_pc<bf16_t, 14>(a, st);     // MFMAs + barriers + epilogue
            case 16: return launch_pc<bf16_t, 16>(a, st);
            case 30: return launch_pc<bf16_t, 30>(a, st);     // MFMAs + barriers only
            case 32: return launch_pc<bf16_t, 32>(a, st);
            case 68: return launch_pc<bf16_t, 68>(a, st);     // no weight loads, a barrier per phase
            case 78: return launch_pc<bf16_t, 78>(a, st);     // MFMAs + epilogue, a barrier per phase
            default: break;
        }
#endif
    return a.dtype == STORM_F16 ? launch_pc<half_t>(a, st) : launch_pc<bf16_t>(a, st);
}

const char* conv_pc_kernel_name(int dtype) {
    return dtype == STORM_F16 ? "storm::conv_pc_kernel<storm::half_t>" : "storm::conv_pc_kernel<storm::bf16_t>";
}

}  // namespace storm
