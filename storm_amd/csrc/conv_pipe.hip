// Software-pipelined 3x3 implicit-GEMM convolution for the wide layers (bf16, >128 output channels).
//
// Same math, arguments and epilogue as conv_igemm.hip (see there and include/storm_hip.h); what differs is the
// staging pipeline, built for ONE 8-wave workgroup per CU (2 waves / SIMD, 256 registers each):
//   * tile: 256 output channels x (8 x 32) pixels; wave grid 4 (cout) x 2 (pixel rows), 64 x 128 per wave
//     = 8 accumulator tiles of v_mfma_f32_32x32x16_bf16.
//   * weights never touch registers: every "phase" (half a tap of one 128-byte K-chunk: 256 rows x 64 B = 16 KiB)
//     is copied global -> LDS by two global_load_lds_dwordx4 per wave into a 4-slot ring, issued THREE phases
//     before use.  The LDS image is lane-linear per instruction, so the bank swizzle is applied to the per-lane
//     SOURCE address (and again on the fragment read).
//   * the haloed 10 x 34 pixel patch of a K-chunk is double buffered; the next chunk's patch is fetched to
//     registers by inline-asm global loads issued under the MFMAs of the current chunk's first tap, gets the
//     fused GroupNorm-apply + SiLU (scale / shift table fetched to LDS the same way) and is written to the other
//     buffer one or two phases later.
//   * all VMEM of the main loop is inline asm, so the counted `s_waitcnt vmcnt(N)` below are the only waits:
//     loads stay in flight across the one raw s_barrier per phase (a compiler-visible load would drain the queue
//     with vmcnt(0) at every barrier).  Counting rule: N = number of VMEM instructions this wave issued AFTER
//     the one that must have landed (loads return in order).
//   * fragments are read one k-group ahead of the MFMAs that use them (two register sets), across the barrier
//     too, so a wave's MFMA stream does not stop for LDS latency.
//
// Per phase q (two k-groups, 16 MFMAs per wave):
//     vmcnt(N): own pieces of phase q+1's weights landed | s_barrier: everyone's landed, reads of q-1 done
//     [first tap of a chunk: issue next chunk's patch loads]  issue weights of phase q+3 -> ring slot (q+3)&3
//     read frag set 1 <- k-group 1 of q | MFMA set 0 | read set 0 <- k-group 0 of q+1 | MFMA set 1
#include <cstdlib>
#include <cstring>
#include "conv_params.h"

namespace storm {
using namespace cidx;

namespace pipe {
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int WM = 2, WN = 4, WAVES_M = 4, WAVES_N = 2, NWAVES = 8, THREADS = 512, BN = 256;
constexpr int PW = Geo<9>::PW, NPIX = Geo<9>::NPIX;
constexpr int PATCH_ROWS = (NPIX + 7) / 8 * 8;                // 344: whole 8-row (1 KiB) LDS-DMA pieces
constexpr int PATCH_BYTES = PATCH_ROWS * PIX_BYTES;           // 44032
constexpr int PPIECES = PATCH_ROWS / 8;                       // 43 wave-instructions cover a patch
constexpr int PU = (PPIECES + NWAVES - 1) / NWAVES;           // pieces per wave (6; the surplus re-issues a piece)
constexpr int WROW = 64;                                      // bytes per weight row and phase
constexpr int WPHASE_BYTES = BN * WROW;                       // 16 KiB
constexpr int RING = 4;
constexpr int OFF_RING = 2 * PATCH_BYTES;
constexpr int OFF_SS = OFF_RING + RING * WPHASE_BYTES;
constexpr int SS_BYTES = 1024;                                // one wave-instruction: 64 channels x (scale, shift) + pad
constexpr int MAIN_BYTES = OFF_SS + 2 * SS_BYTES;             // 155648
constexpr int PR = 2;                                         // pixel rows (of 32 px) staged per epilogue pass and wave
constexpr int STAGE_BYTES = NWAVES * 32 * PR * WM * 128;      // 128 KiB
constexpr int LDS_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
constexpr int NP = PU + 1;                                    // VMEM instructions of one patch issue (+ the table)
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// LDS byte offset of 16-B slot s (0..3) of row `row` of a weight phase tile.  16 lanes of a ds_read_b128 group
// (distinct rows mod 16... see MI355X LDS notes) hit 16 distinct 16-B bank groups.
STORM_HD int w_off(int row, int s) { return row * WROW + ((s ^ ((row >> 2) & 3)) << 4); }

// ---- asynchronous memory primitives (inline asm on the device; synchronous on the host simulator) ----------
// 16 B per lane, global (uniform base + 32-bit lane offset) -> LDS at (uniform lds_wave + 16 * lane).
__device__ __forceinline__ void glds16(const void* base, uint32_t voff, char* lds_wave, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)lane;
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_wave);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(la) : "memory");
#else
    memcpy(lds_wave + 16 * lane, static_cast<const char*>(base) + voff, 16);
#endif
}
template <int N> __device__ __forceinline__ void vm_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#elif defined(STORM_HOST_SIM)
    simrt::wave_rendezvous();          // simulator lanes are not in lockstep: every lane's copy is done past this point
#endif
}
// workgroup barrier WITHOUT a vmcnt drain: LDS traffic of this wave retired (lgkmcnt), loads keep flying
__device__ __forceinline__ void raw_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}
__device__ __forceinline__ void prio(int p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (p) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#else
    (void)p;
#endif
}

}  // namespace pipe
using namespace pipe;

template <int ABL>
__global__ __launch_bounds__(pipe::THREADS, 2)
void conv_pipe_kernel(const ConvParams a, const int n_ct, const int tiles_per_xcd,
                      const int ntiles, const int tiles_x, const int tiles_per_img) {
    typedef bf16_t T;
    typedef bf16x8 Frag;
    constexpr int KC = 64;                                      // channels per K-chunk (128 B)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const BlockMap bm = block_map(blockIdx.x, n_ct, tiles_per_xcd);
    if (bm.tile >= ntiles) return;
    const int b = bm.tile / tiles_per_img;
    const int trem = bm.tile - b * tiles_per_img;
    const int ty0 = (trem / tiles_x) * TILE_H;
    const int tx0 = (trem % tiles_x) * TILE_W;
    const int cout0 = bm.ct * BN;

    const int tid = threadIdx.x, lane = tid & 63;
#if defined(__HIP_DEVICE_COMPILE__)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#else
    const int wave = tid >> 6;
#endif
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long* const trace_rec = (ABL & 64) && a.trace ? a.trace + ((long long)blockIdx.x * NWAVES + wave) * TRACE_SLOTS : nullptr;
    auto stamp = [&](int idx) {                  // profiling instantiation only (tools/conv_trace.py)
        if ((ABL & 64) && trace_rec && idx < 496) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) trace_rec[idx] = t;
        }
    };
    auto stamp_tail = [&](int idx) {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((ABL & 64) && trace_rec) { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (lane == 0) trace_rec[idx] = t; }
#endif
    };
    if ((ABL & 64) && trace_rec && lane == 0)
        trace_rec[0] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
#else
    auto stamp = [&](int) {};
    auto stamp_tail = [&](int) {};
#endif
    stamp(1);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // ---- K-chunk descriptors (wave-uniform) -----------------------------------------------------
    struct Chunk {
        const T* src; const float* gn_ss;
        int C, cbeg, cvalid, ntaps, gn_silu;
    };
    auto get_chunk = [&](int r, int ch) {
        const ConvRun& R = a.run[r];
        Chunk c;
        c.src = reinterpret_cast<const T*>(R.src) + (long long)b * R.src_bstride;
        c.C = R.C; c.cbeg = R.c0 + ch * KC; c.cvalid = min(KC, R.cn - ch * KC);
        c.ntaps = R.ntaps;
        c.gn_ss = R.gn_ss ? R.gn_ss + 2 * ((long long)b * R.gn_C + R.wc0 + ch * KC) : nullptr;
        c.gn_silu = R.gn_silu;
        return c;
    };
    auto chunks_of = [&](int r) { return (a.run[r].cn + KC - 1) / KC; };
    const int nruns = a.nruns;

    // ---- weight stream: cursor over (run, chunk, tap, half), three phases ahead of the MFMAs -----
    int w_r = 0, w_ch = 0, w_tp = 0, w_h = 0;
    const T* w_run; int w_tapstride, w_CinP, w_klim, w_rows, w_ntaps, w_nch;
    auto w_enter_run = [&](int r) {
        const ConvRun& R = a.run[r];
        w_run = reinterpret_cast<const T*>(R.w) + (long long)b * R.w_bstride + R.wc0;
        w_tapstride = (int)R.w_tapstride; w_CinP = R.CinP; w_klim = R.CinP - R.wc0; w_rows = R.w_rows;
        w_ntaps = R.ntaps; w_nch = (R.cn + KC - 1) / KC;
    };
    w_enter_run(0);
    int grow[2], gk8[2];                       // this lane's row / logical k offset (elements) inside a phase tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2);
        grow[j] = cout0 + row;
        gk8[j] = ((lane & 3) ^ ((row >> 2) & 3)) * 8;
    }
    auto w_issue = [&](int q) {                // weights of the cursor's phase -> ring slot q & 3; then advance
        const T* tapbase = w_run + (long long)w_tp * w_tapstride;
        const int k0 = w_ch * KC + w_h * 32;
        char* dst = smem + OFF_RING + (q & (RING - 1)) * WPHASE_BYTES + wave * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = min(grow[j], w_rows - 1);             // rows past the matrix: any valid row (never stored)
            const int k = k0 + gk8[j];
            const uint32_t voff = (uint32_t)(co * w_CinP + (k < w_klim ? k : 0)) * 2u;   // k past the row: finite data x zero patch
            glds16(tapbase, voff, dst + j * 1024, lane);
        }
        // advance; past the end the cursor stays on the last phase (harmless re-load into a free slot)
        int h = w_h ^ 1, tp = w_tp, ch = w_ch, r = w_r;
        if (h == 0) {
            ++tp;
            if (tp == w_ntaps) {
                tp = 0; ++ch;
                if (ch == w_nch) { ch = 0; ++r; }
            }
        }
        if (r < nruns) {
            if (r != w_r) w_enter_run(r);
            w_h = h; w_tp = tp; w_ch = ch; w_r = r;
        }
    };

    // ---- patch staging: issue (LDS-DMA, raw activations straight into the next patch buffer, swizzle on the
    //      source address) ... commit (in place, every lane fixes up the 16-B units it fetched itself: zeros for
    //      the padding halo / channels past the run, GroupNorm affine + SiLU when fused).  No staging registers:
    //      nothing asynchronous ever targets a VGPR, so the compiler cannot touch data that has not landed.
    uint32_t pmask = 0;                        // bit i: unit i of this lane is real input (inside the image, channel valid)
    auto patch_issue = [&](const Chunk& c, int parity) {
        char* dst = smem + parity * PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < PU; ++i) {
            int k = wave + i * NWAVES;                           // piece: patch rows 8k .. 8k+7
            if (k >= PPIECES) k -= NWAVES;                       // surplus slot: same piece again (keeps the VMEM count uniform)
            const int row = k * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);      // logical 16-B slot that lands in physical slot lane & 7
            const int py = row / PW, px = row - py * PW;
            const int gy = ty0 + py - 1, gx = tx0 + px - 1;
            const bool ok = row < NPIX && slot * 8 < c.cvalid && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const uint32_t voff = ok ? (uint32_t)((gy * a.W + gx) * c.C + c.cbeg + slot * 8) * 2u : 0u;
            glds16(c.src, voff, dst + k * 1024, lane);
            pmask = ok ? (pmask | (1u << i)) : (pmask & ~(1u << i));
        }
        // (scale, shift) of the chunk's channels -> LDS table; lanes past the chunk re-read channel 0
        const bool has = c.gn_ss != nullptr;
        const void* tb = has ? static_cast<const void*>(c.gn_ss) : static_cast<const void*>(c.src);
        const uint32_t toff = (has && 2 * lane < c.cvalid) ? (uint32_t)lane * 16u : 0u;
        glds16(tb, toff, smem + OFF_SS + parity * SS_BYTES, lane);
    };
    auto patch_commit = [&](const Chunk& c, int parity) {
        char* dst = smem + parity * PATCH_BYTES;
        const bool gn = c.gn_ss != nullptr;
#pragma unroll
        for (int i = 0; i < PU; ++i) {
            const int k = wave + i * NWAVES;
            if (k < PPIECES) {
                const int row = k * 8 + (lane >> 3);
                const int slot = (lane & 7) ^ ((row >> 1) & 7);
                uint4* const q = reinterpret_cast<uint4*>(dst + k * 1024 + lane * 16);
                const bool ok = (pmask >> i) & 1u;
                if (!ok) {
                    *q = make_uint4(0u, 0u, 0u, 0u);
                } else if (gn) {
                    float ss[16];
                    const float* t = reinterpret_cast<const float*>(smem + OFF_SS + parity * SS_BYTES) + 16 * slot;
#pragma unroll
                    for (int j = 0; j < 16; j += 4) {
                        const float4 t4 = *reinterpret_cast<const float4*>(t + j);
                        ss[j] = t4.x; ss[j + 1] = t4.y; ss[j + 2] = t4.z; ss[j + 3] = t4.w;
                    }
                    *q = gn_act_slot(*q, ss, c.gn_silu, (T*)nullptr);
                }
            }
        }
    };

    // ---- fragment reads ----------------------------------------------------------------------------
    int aoff[WM];                              // weight-tile offsets of this lane's rows, k-group 0 of a phase
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) aoff[mi] = w_off((wm * WM + mi) * 32 + (lane & 31), lane >> 5);
    int pbase[WN];                             // patch offsets for the tap being read, k-group 0 of the chunk
    auto set_tap = [&](int parity, int dy, int dx) {
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
            pbase[ni] = parity * PATCH_BYTES + lds_off(patch_pixel<9>(lane, wn * WN + ni, dy, dx), lane >> 5);
    };
    // k-group kg (0..3) of the chunk = k-group (kg & 1) of ring phase q
    auto read_frags = [&](Frag (&fa)[WM], Frag (&fb)[WN], int q, int kg) {
        const char* wb = smem + OFF_RING + (q & (RING - 1)) * WPHASE_BYTES;
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const Frag*>(wb + (aoff[mi] ^ ((kg & 1) << 5)));
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[ni] = *reinterpret_cast<const Frag*>(smem + (pbase[ni] ^ (kg << 5)));
    };
    auto mma = [&](const Frag (&fa)[WM], const Frag (&fb)[WN]) {
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) Mma<T>::run(fa[mi], fb[ni], acc[mi][ni]);
    };
    auto tap_offsets = [&](int ntaps, int tp, int& dy, int& dx) {
        if (ntaps == 9) { dy = tp / 3; dx = tp - dy * 3; } else { dy = 1; dx = 1; }
    };

    // ---- prologue: first patch, first two weight phases -------------------------------------------
    int r = 0, ch = 0, nch_r = chunks_of(0), ci = 0;
    Chunk cur = get_chunk(0, 0);
    w_issue(0); w_issue(1);
    patch_issue(cur, 0);
    vm_wait<0>();
    patch_commit(cur, 0);
    raw_barrier();
    // Ping-pong: waves 4-7 (the second wave of every SIMD) run one barrier interval behind waves 0-3, so one
    // group's MFMA interval coincides with the other's staging interval (waits, LDS-DMA issue, fragment reads,
    // patch commit) and the matrix pipe of a SIMD always has one wave feeding it.
    const int grp = (ABL & 2) ? 0 : wave >> 2;        // (ABL & 2: profiling variant without the stagger)
    if (grp == 1) raw_barrier();
    stamp(2);
    Frag fa0[WM], fb0[WN], fa1[WM], fb1[WN];

    // ---- main loop: one iteration = one phase P = (chunk, tap, half): staging interval S(P), MFMA interval C(P) ----
    //   S(P): [commit the patch fetched two phases ago]  issue weights of phase P+2 -> slot (P+2)&3
    //         [second phase of a chunk: issue the next chunk's patch LDS-DMA]  read k-group 0 of P
    //         vmcnt: own share of phase P+1 landed | barrier
    //   C(P): read k-group 1 | 16 MFMAs | barrier
    // LDS lifetimes (intervals counted in barriers; group 1 lags by one): phase P's slot is read in intervals
    // 2P..2P+2, slot (P+2)&3 = (P-2)&3 was last read in interval 2P-2 -> free in S(P).  The patch buffer of the
    // next chunk was last read in the interval of S(P0) itself (by the lagging group) -> DMA into it from S(P0+1) on.
    int P = 0, tp = 0, h = 0, since_issue = 99, step = 0;       // since_issue: phases since the last patch issue
    bool has_nc, commit_pending = false; Chunk nxt = cur;
    {
        int nr = r, nc = ch + 1;
        if (nc == nch_r) { nc = 0; ++nr; }
        has_nc = nr < nruns;
        nxt = get_chunk(has_nc ? nr : r, has_nc ? nc : ch);
    }
    while (true) {
        const int ntaps = cur.ntaps, par = ci & 1;
        // ---------------- S(P) ----------------
        stamp(4 + 8 * step);
        if (commit_pending && since_issue == 2) {               // 9-tap chunk: patch issued two phases ago
            vm_wait<2>();
            patch_commit(nxt, par ^ 1);
            commit_pending = false;
        }
        w_issue(P + 2);
        // The other patch buffer was read until the lagging group's C(P0 - 1), which shares its interval with the
        // leading group's S(P0): the LDS-DMA into it may start in the chunk's SECOND phase.
        const bool issue_now = tp == 0 && h == 1 && has_nc;
        if (issue_now) { patch_issue(nxt, par ^ 1); since_issue = 0; commit_pending = ntaps != 1; }
        stamp(5 + 8 * step);
        if (h == 0) { int dy, dx; tap_offsets(ntaps, tp, dy, dx); set_tap(par, dy, dx); }
        read_frags(fa0, fb0, P, 2 * h);
        if (issue_now && ntaps == 1) {                          // two-phase chunk: the next phase already reads it
            vm_wait<0>();
            patch_commit(nxt, par ^ 1);
        }
        stamp(6 + 8 * step);
        if (since_issue <= 1) vm_wait<2 + NP>(); else vm_wait<2>();
        stamp(7 + 8 * step);
        raw_barrier();
        // ---------------- C(P) ----------------
        stamp(8 + 8 * step);
        read_frags(fa1, fb1, P, 2 * h + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ABL & 1)) prio(1);
        mma(fa0, fb0);
        if (ABL & 64) { __builtin_amdgcn_sched_barrier(0); stamp(9 + 8 * step); __builtin_amdgcn_sched_barrier(0); }
        mma(fa1, fb1);
        if (!(ABL & 1)) prio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp(10 + 8 * step);
        raw_barrier();
        stamp(11 + 8 * step);
        // ---------------- advance ----------------
        ++P; ++since_issue; ++step;
        h ^= 1;
        if (h == 0) {
            ++tp;
            if (tp == ntaps) {
                if (!has_nc) break;
                tp = 0;
                int nr = r, nc = ch + 1;
                if (nc == nch_r) { nc = 0; ++nr; }
                if (nr != r) nch_r = chunks_of(nr);
                cur = nxt; r = nr; ch = nc; ++ci;
                nr = r; nc = ch + 1;
                if (nc == nch_r) { nc = 0; ++nr; }
                has_nc = nr < nruns;
                nxt = get_chunk(has_nc ? nr : r, has_nc ? nc : ch);
            }
        }
    }
    if (grp == 0) raw_barrier();                        // balance the stagger: every wave has executed the same barriers

    // ---- epilogue: LDS transpose -> (bias, temb bias, skip, scale) -> wide stores (as conv_igemm.hip) ------
    stamp_tail(500);
    vm_wait<0>();                                       // trailing ring re-loads landed: LDS is free to reuse
    raw_barrier();
    stamp_tail(501);
    constexpr int SROWS = 32 * PR;
    char* const stage = smem + wave * (SROWS * WM * 128);
    constexpr int LPR = WM * 4;                 // lanes per staged row (8 couts each)
    constexpr int RPI = 64 / LPR;               // rows per read iteration
    const int skipC = a.outC;
    const int c8 = lane % LPR;
    const int co = cout0 + wm * WM * 32 + c8 * 8;
    float badd[8];
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
    float gsum[8], gsq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gsum[e] = 0.f; gsq[e] = 0.f; }
#pragma unroll
    for (int pass = 0; pass < WN / PR; ++pass) {
        if (pass > 0) wave_sync();
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
        wave_sync();
#pragma unroll 4
        for (int it = 0; it < SROWS / RPI; ++it) {
            const int row = it * RPI + lane / LPR;
            const float4 v0 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, 2 * c8));
            const float4 v1 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, 2 * c8 + 1));
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const int trow = wn * WN + pass * PR + (row >> 5), n = row & 31;
            const int gy = ty0 + trow, gx = tx0 + n;
            const bool ok = gy < a.H && gx < a.W;
            const int pix = gy * a.W + gx;
            if (ok && co_ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += badd[e];
                if (a.skip) {
                    float sk[8];
                    load8(skip_b + (uint32_t)(pix * skipC + co), sk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += sk[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { v[e] *= a.scale; gsum[e] += v[e]; gsq[e] = fmaf(v[e], v[e], gsq[e]); }
                const uint32_t o = (uint32_t)(pix * a.outC + co);
                if (a.out_f32) store8(reinterpret_cast<float*>(a.out) + (long long)b * a.out_bstride + o, v);
                else store8(reinterpret_cast<T*>(a.out) + (long long)b * a.out_bstride + o, v);
            }
        }
    }
    stamp_tail(502);
    if (ABL & 64) { vm_wait<0>(); stamp_tail(503); }
    if (a.gn_part != nullptr) {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) { gsum[e] += __shfl_xor(gsum[e], off, 64); gsq[e] += __shfl_xor(gsq[e], off, 64); }
        __syncthreads();
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

bool conv_pipe_supports(const storm_conv_args& a) {
    if (a.dtype != STORM_BF16 || a.nseg < 1 || a.seg[0].ntaps != 9) return false;
    for (int s = 0; s < a.nseg; ++s)
        if ((a.seg[s].w_tapstride >> 31) != 0) return false;
    return true;
}

int launch_conv_pipe(const storm_conv_args& a, hipStream_t st) {
    using namespace pipe;
    const char* abl_env = getenv("STORM_CONV_ABLATE");
    const int abl = abl_env ? atoi(abl_env) : 0;                 // profiling instantiations (tools/conv_trace.py, A/B probes)
    const bool traced = (abl & 64) != 0;
    auto kern = conv_pipe_kernel<0>;
    int ki = 0;
    switch (abl) {
        case 1: kern = conv_pipe_kernel<1>; ki = 1; break;       // no s_setprio
        case 2: kern = conv_pipe_kernel<2>; ki = 2; break;       // no stagger
        case 64: kern = conv_pipe_kernel<64>; ki = 3; break;     // wave timeline stamps
        case 65: kern = conv_pipe_kernel<65>; ki = 4; break;
        case 66: kern = conv_pipe_kernel<66>; ki = 5; break;
        default: break;
    }
    static bool attr_set[6] = {false, false, false, false, false, false};
    if (!attr_set[ki]) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set[ki] = true;
    }
    const int tiles_x = cdiv(a.W, TILE_W);
    const int tiles_per_img = tiles_x * cdiv(a.H, TILE_H);
    const long long ntiles = (long long)a.B * tiles_per_img;
    const int n_ct = cdiv(a.outC, BN);
    const int tiles_per_xcd = cdiv(ntiles, 8);
    const long long grid = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(grid > 0 && grid < (1LL << 31), "storm_conv: grid %lld out of range", grid);
    ConvParams prm = make_params(a);
    if (traced) {
        const char* tp = getenv("STORM_CONV_TRACE_PTR");
        prm.trace = tp ? reinterpret_cast<unsigned long long*>(strtoull(tp, nullptr, 0)) : nullptr;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), LDS_BYTES, st, prm, n_ct, tiles_per_xcd, (int)ntiles,
                       tiles_x, tiles_per_img);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

}  // namespace storm
