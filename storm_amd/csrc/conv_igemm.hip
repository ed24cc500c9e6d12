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
//     [BN][chunk] stream per tap through a 2-deep LDS ring, prefetched into registers during
//     the MFMAs of the previous tap (one barrier per tap).
//   * LDS rows are XOR-swizzled (conv_index.h) so every ds_read_b128 group is conflict free.
//   * ~75 KB LDS and <=256 VGPR per workgroup -> 2 workgroups / CU so one group's staging
//     overlaps the other's MFMA phase; block ids are mapped so each XCD's L2 sees a contiguous
//     run of pixel tiles and both cout halves of a tile.
//   * epilogue: accumulators go through an LDS transpose so that global stores (and the
//     skip / bias reads) are 16-32 B per lane, full 128-B lines per 4-8 lanes.
//   * bf16 operands: v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  fp32 operands:
//     v_mfma_f32_32x32x2_f32 (exact fp32, used by the parity path).
#include "common.h"
#include "conv_index.h"

namespace storm {
using namespace cidx;

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        // the 4 floats of a slot are 4 k-positions; pairing (a[r], b[r]) keeps A and B consistent
#pragma unroll
        for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], c, 0, 0, 0);
    }
};

template <int TAPS, int WM, int WAVES_M>
struct ConvCfg {
    static constexpr int WAVES_N = 4 / WAVES_M;
    static constexpr int WN = TILE_H / WAVES_N;          // pixel tile rows (of 32 px) per wave
    static constexpr int BN = WAVES_M * WM * 32;          // output channels per workgroup
    static constexpr int NPIX = Geo<TAPS>::NPIX;
    static constexpr int PATCH_BYTES = NPIX * PIX_BYTES;
    static constexpr int WBUF_BYTES = BN * PIX_BYTES;
    static constexpr int STAGE_BYTES = 4 * 64 * WM * 128;
    static constexpr int MAIN_BYTES = PATCH_BYTES + 2 * WBUF_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
    static constexpr int PU = (NPIX * 8 + THREADS - 1) / THREADS;   // patch 16-B units per thread
    static constexpr int WU = BN * 8 / THREADS;                      // weight units per thread
    static_assert(WN % 2 == 0, "epilogue stages two pixel rows per pass");
    static_assert(BN * 8 % THREADS == 0, "");
};

struct SegInfo {           // per-segment scalars derived once per block
    int Cin, nchunks, ntaps;
};

template <typename T, int TAPS, int WM, int WAVES_M>
__global__ __launch_bounds__(256, 2)
void conv_igemm_kernel(const storm_conv_args a, const int n_ct, const int tiles_per_xcd,
                       const int ntiles, const int tiles_x, const int tiles_per_img) {
    typedef ConvCfg<TAPS, WM, WAVES_M> Cfg;
    typedef typename Mma<T>::Frag Frag;
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int KC = 8 * PER16;              // channels per K-chunk (128 B)
    constexpr int KG = 2 * PER16;              // channels per k-group (one fragment slot pair)
    constexpr int WN = Cfg::WN, BN = Cfg::BN, PW = Geo<TAPS>::PW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch = smem;
    char* const wbuf = smem + Cfg::PATCH_BYTES;

    const BlockMap bm = block_map(blockIdx.x, n_ct, tiles_per_xcd);
    if (bm.tile >= ntiles) return;
    const int b = bm.tile / tiles_per_img;
    const int trem = bm.tile - b * tiles_per_img;
    const int ty0 = (TAPS == 9) ? (trem / tiles_x) * TILE_H : 0;
    const int tx0 = (TAPS == 9) ? (trem % tiles_x) * TILE_W : 0;
    const long long npix = (long long)a.H * a.W;
    const long long lin0 = (long long)trem * (TILE_H * TILE_W);     // TAPS==1: linear pixel base
    const int cout0 = bm.ct * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;

    SegInfo si[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        si[s].Cin = a.seg[s].Ca + a.seg[s].Cb;
        si[s].nchunks = (si[s].Cin + KC - 1) / KC;
        si[s].ntaps = a.seg[s].ntaps;
    }
    const int nseg = a.nseg;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // ---- loaders -------------------------------------------------------------------------
    uint4 wreg[Cfg::WU];
    auto load_w = [&](int sg, int ch, int tp) {
        const storm_conv_seg& S = a.seg[sg];
        const T* wbase = reinterpret_cast<const T*>(S.w) + (long long)b * S.w_bstride +
                         (long long)tp * S.w_tapstride;
#pragma unroll
        for (int i = 0; i < Cfg::WU; ++i) {
            const int u = tid + i * THREADS;
            const int row = u >> 3, slot = u & 7;
            const int co = cout0 + row, c = ch * KC + slot * PER16;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (co < S.w_rows && c < S.CinP)
                v = *reinterpret_cast<const uint4*>(wbase + (long long)co * S.CinP + c);
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
    auto load_patch = [&](int sg, int ch) {
        const storm_conv_seg& S = a.seg[sg];
        const T* pa = reinterpret_cast<const T*>(S.src_a) + (long long)b * S.bstride_a;
        const T* pb = reinterpret_cast<const T*>(S.src_b) + (long long)b * S.bstride_b;
        const int Ca = S.Ca, Cb = S.Cb, Cin = Ca + Cb;
        uint4 preg[Cfg::PU];
#pragma unroll
        for (int i = 0; i < Cfg::PU; ++i) {
            const int u = tid + i * THREADS;
            const int p = u >> 3, slot = u & 7;
            const int c = ch * KC + slot * PER16;
            long long pix;
            bool ok = (u < Cfg::NPIX * 8) && (c < Cin);
            if (TAPS == 9) {
                const int py = p / PW, px = p - py * PW;
                const int gy = ty0 + py - 1, gx = tx0 + px - 1;
                ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                pix = (long long)gy * a.W + gx;
            } else {
                pix = lin0 + p;
                ok = ok && pix < npix;
            }
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ok) {
                const T* src = (c < Ca) ? (pa + pix * Ca + c) : (pb + pix * Cb + (c - Ca));
                v = *reinterpret_cast<const uint4*>(src);
            }
            preg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < Cfg::PU; ++i) {
            const int u = tid + i * THREADS;
            if (u < Cfg::NPIX * 8) *reinterpret_cast<uint4*>(patch + lds_off(u >> 3, u & 7)) = preg[i];
        }
    };

    // ---- main loop over (segment, K-chunk, tap) steps -------------------------------------
    int sg = 0, ch = 0, tp = 0, step = 0;
    load_w(0, 0, 0);
    store_w(0);
    while (sg < nseg) {
        if (tp == 0) {
            __syncthreads();                 // every wave finished reading the previous patch
            load_patch(sg, ch);
        }
        // next step's coordinates
        int nsg = sg, nch = ch, ntp = tp + 1;
        if (ntp == si[sg].ntaps) { ntp = 0; ++nch; if (nch == si[sg].nchunks) { nch = 0; ++nsg; } }
        const bool has_next = nsg < nseg;
        if (has_next) load_w(nsg, nch, ntp);      // global loads fly during the MFMAs below
        __syncthreads();                          // patch + wbuf[step&1] visible

        {
            const char* wb = wbuf + (step & 1) * Cfg::WBUF_BYTES;
            int dy = 0, dx = 0;
            if (TAPS == 9) {
                if (si[sg].ntaps == 9) { dy = tp / 3; dx = tp - dy * 3; } else { dy = 1; dx = 1; }
            }
            const int rem = si[sg].Cin - ch * KC;
            const int nk = rem >= KC ? 4 : (rem + KG - 1) / KG;
            int prow[WN], arow[WM];
#pragma unroll
            for (int ni = 0; ni < WN; ++ni) prow[ni] = patch_pixel<TAPS>(lane, wn * WN + ni, dy, dx);
#pragma unroll
            for (int mi = 0; mi < WM; ++mi) arow[mi] = (wm * WM + mi) * 32 + (lane & 31);
            auto kgroup = [&](int j) {
                const int slot = frag_slot(lane, j);
                Frag fa[WM], fb[WN];
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
                    fa[mi] = *reinterpret_cast<const Frag*>(wb + lds_off(arow[mi], slot));
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                    fb[ni] = *reinterpret_cast<const Frag*>(patch + lds_off(prow[ni], slot));
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < WN; ++ni) Mma<T>::run(fa[mi], fb[ni], acc[mi][ni]);
            };
            if (nk == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) kgroup(j);
            } else {
#pragma unroll 1
                for (int j = 0; j < nk; ++j) kgroup(j);
            }
        }
        if (has_next) store_w((step + 1) & 1);    // that buffer was last read in step-1 (barrier passed)
        sg = nsg; ch = nch; tp = ntp; ++step;
    }

    // ---- epilogue: LDS transpose -> (bias, temb bias, skip, scale) -> wide stores ---------
    __syncthreads();
    char* const stage = smem + wave * (64 * WM * 128);
    constexpr int LPR = WM * 4;                 // lanes per staged row (8 couts each)
    constexpr int RPI = 64 / LPR;               // rows per read iteration
    const int skipC = a.outC;
#pragma unroll
    for (int pass = 0; pass < WN / 2; ++pass) {
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = nn * 32 + (lane & 31);
                    const f32x16& c = acc[mi][pass * 2 + nn];
                    *reinterpret_cast<float4*>(stage + stage_off<WM>(row, stage_wslot(lane, mi, g))) =
                        make_float4(c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
                }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < LPR; ++it) {
            const int row = it * RPI + lane / LPR, c8 = lane % LPR;
            const float4 v0 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, 2 * c8));
            const float4 v1 = *reinterpret_cast<const float4*>(stage + stage_off<WM>(row, 2 * c8 + 1));
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const int trow = wn * WN + pass * 2 + (row >> 5), n = row & 31;
            long long pix;
            bool ok;
            if (TAPS == 9) {
                const int gy = ty0 + trow, gx = tx0 + n;
                ok = gy < a.H && gx < a.W;
                pix = (long long)gy * a.W + gx;
            } else {
                pix = lin0 + trow * TILE_W + n;
                ok = pix < npix;
            }
            const int co = cout0 + wm * WM * 32 + c8 * 8;
            if (ok && co < a.outC) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float add = 0.0f;
                    if (co + e < a.Cout) {
                        if (a.bias) add += a.bias[co + e];
                        if (a.tbias) add += a.tbias[(long long)b * a.tbias_stride + co + e];
                    }
                    v[e] += add;
                }
                if (a.skip) {
                    float sk[8];
                    load8(reinterpret_cast<const T*>(a.skip) + (long long)b * a.skip_bstride + pix * skipC + co, sk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += sk[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= a.scale;
                const long long o = (long long)b * a.out_bstride + pix * a.outC + co;
                if (a.out_f32) store8(reinterpret_cast<float*>(a.out) + o, v);
                else store8(reinterpret_cast<T*>(a.out) + o, v);
            }
        }
        if (pass + 1 < WN / 2) __syncthreads();
    }
}

template <typename T, int TAPS, int WM, int WAVES_M>
static int launch_conv(const storm_conv_args& a, hipStream_t st) {
    typedef ConvCfg<TAPS, WM, WAVES_M> Cfg;
    auto kern = conv_igemm_kernel<T, TAPS, WM, WAVES_M>;
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
    const long long grid = 8LL * tiles_per_xcd * n_ct;
    STORM_CHECK(grid > 0 && grid < (1LL << 31), "storm_conv: grid %lld out of range", grid);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(THREADS), Cfg::LDS_BYTES, st, a, n_ct,
                       tiles_per_xcd, (int)ntiles, tiles_x, tiles_per_img);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

template <typename T>
static int dispatch_conv(const storm_conv_args& a, hipStream_t st) {
    bool any9 = false;
    for (int s = 0; s < a.nseg; ++s) any9 = any9 || a.seg[s].ntaps == 9;
    const bool small = a.outC <= 32;
    if (any9) return small ? launch_conv<T, 9, 1, 1>(a, st) : launch_conv<T, 9, 2, 2>(a, st);
    return small ? launch_conv<T, 1, 1, 1>(a, st) : launch_conv<T, 1, 2, 2>(a, st);
}

}  // namespace storm

extern "C" int storm_conv(const storm_conv_args* ap, storm_stream_t s) {
    using namespace storm;
    STORM_CHECK(ap != nullptr, "storm_conv: null args");
    const storm_conv_args& a = *ap;
    STORM_CHECK(a.nseg >= 1 && a.nseg <= 2, "storm_conv: nseg=%d", a.nseg);
    STORM_CHECK(a.B > 0 && a.H > 0 && a.W > 0, "storm_conv: bad shape B=%d H=%d W=%d", a.B, a.H, a.W);
    STORM_CHECK(a.outC > 0 && a.outC % 8 == 0 && a.Cout <= a.outC, "storm_conv: outC=%d Cout=%d", a.outC, a.Cout);
    STORM_CHECK(a.out != nullptr, "storm_conv: null out");
    const int per16 = a.dtype == STORM_BF16 ? 8 : 4;
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
    if (a.dtype == STORM_F32) return dispatch_conv<float>(a, st);
    STORM_CHECK(false, "storm_conv: dtype %d", a.dtype);
}
