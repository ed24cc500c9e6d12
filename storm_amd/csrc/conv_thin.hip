// Convolutions over an 8-channel input (bf16 / fp16): the stem of NCSN++ (ncsnpp.py:183, conv3x3 over the packed [x.re, x.im,
// y.re, y.im, ...] planes) and the input-skip 1x1 convolutions of `Combine` (layerspp.py:44-59: conv1x1(pyramid_downsample(input))
// + h).  Same math, arguments, statistics-partial layout and epilogue arithmetic as conv_igemm.hip, which ran them as padded
// 64-channel K-chunks (a 10 x 34 x 128-byte patch staged through LDS for 16 useful bytes per pixel, 36 MFMA steps for 5 useful
// ones): 1.4 - 2.3 TB/s on layers whose time is their OUTPUT (128 - 256 channels written for 8 read).
//
// With 8 input channels a pixel IS one 16-byte MFMA operand slot, so nothing needs staging:
//   * B fragment (32 pixels x 16 k) = two taps of 32 consecutive pixels: ONE 16-byte-per-lane buffer load straight from the
//     NHWC tensor (the 32 lanes of a half read 512 contiguous bytes; padding pixels are out-of-range offsets = hardware zeros);
//   * A fragments (all taps of 128 output channels: 5 k-steps x 4 cout tiles, or 1 x 4) live in registers for the whole tile;
//   * a wave computes 32 pixels x 128 couts (4 accumulator tiles, 20 MFMAs for a 3x3), stages them through its private LDS block
//     and stores 16 bytes per lane, 256 contiguous bytes per pixel; the per-tile statistics partials are reduced across the
//     workgroup's four waves (8 x 32-pixel tile for 3x3, 256 linear pixels for 1x1: the layouts storm_conv_tiles promises).
#include <cstring>
#include "conv_pipe_common.h"

namespace storm {
using namespace cidx;

namespace thin {
using pipe::IC; using pipe::static_for;
constexpr int THREADS = 256, BN = 128;                      // couts per pass (a workgroup loops over outC / 128 passes)
constexpr int WSTAGE = 32 * 4 * 128;                        // one wave's staging block: fp32 [32 px][128 couts] = 16 KiB
constexpr int LDS_BYTES = 4 * WSTAGE + 4 * BN * 2 * 4;      // + statistics scratch [4 waves][128][2]
struct Params {
    const void* src; const void* w; void* out; const float* bias; const void* skip; float* gn_part;
    long long src_bstride, out_bstride, skip_bstride;       // elements
    int B, H, W, outC, Cout, w_rows, tiles_x, tiles_per_img;
    float scale; int nt;                                    // nt: non-temporal output stores (the stem's 537 MB are read next from HBM anyway)
};
}  // namespace thin

template <typename T, int TAPS>
__device__ __forceinline__ void conv_thin_body(const thin::Params& p, const int tile) {
    using namespace thin;
    typedef typename Mma<T>::Frag Frag;
    constexpr int KS = (TAPS + 1) / 2;                      // k-steps: two taps (2 x 8 channels) each
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = uniform(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int b = tile / p.tiles_per_img, trem = tile - b * p.tiles_per_img;
    const int ty0 = TAPS == 9 ? (trem / p.tiles_x) * TILE_H : 0, tx0 = TAPS == 9 ? (trem % p.tiles_x) * TILE_W : 0;
    const long long lin0 = (long long)trem * (TILE_H * TILE_W);            // 1x1: linear pixel base of the tile
    const long long npix = (long long)p.H * p.W;
    const BufRsrc sbuf = make_buf(reinterpret_cast<const T*>(p.src) + (long long)b * p.src_bstride, (uint32_t)(npix * 16));
    const BufRsrc wbuf = make_buf(p.w, (uint32_t)((long long)TAPS * p.w_rows * 32));       // [tap][rows][16] 16-bit elements
    char* const stage = smem + wave * WSTAGE;
    float* const red = reinterpret_cast<float*>(smem + 4 * WSTAGE);
    const int c16 = lane & 15, l4 = lane >> 4;              // epilogue: cout octet of the pass / pixel (of 4 per iteration)
    T* const out_b = reinterpret_cast<T*>(p.out) + (long long)b * p.out_bstride;
    const T* const skip_b = reinterpret_cast<const T*>(p.skip) + (long long)b * p.skip_bstride;
    const f32x2 scale2 = {p.scale, p.scale};

    for (int cout0 = 0; cout0 < p.outC; cout0 += BN) {
        // ---- weight fragments of this pass: lane (m, h) holds the 8 channels of tap 2 s + h for cout tile mi --------------
        Frag af[KS][4];
        static_for<KS>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            const int tap = 2 * s + h;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int row = cout0 + mi * 32 + n;
                const uint4 v = buf_load16(wbuf, tap < TAPS && row < p.w_rows ? (uint32_t)((tap * p.w_rows + row) * 32) : BUF_OOB, 0u);
                af[s][mi] = *reinterpret_cast<const Frag*>(&v);
            }
        });
        const int co = cout0 + c16 * 8;
        float badd[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) badd[e] = (p.bias && co + e < p.Cout) ? p.bias[co + e] : 0.f;
        f32x2 badd2[4], gsum2[4], gsq2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { badd2[i] = f32x2{badd[2 * i] * p.scale, badd[2 * i + 1] * p.scale}; gsum2[i] = f32x2{0.f, 0.f}; gsq2[i] = f32x2{0.f, 0.f}; }
        const bool co_ok = co < p.outC;

#pragma unroll 1
        for (int pt = 2 * wave; pt < 2 * wave + 2; ++pt) {  // this wave's two 32-pixel rows of the tile
            // ---- pixel fragments: tap 2 s + h of pixel n, straight from global memory ------------------------------------
            Frag bf[KS];
            static_for<KS>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
                const int tap = 2 * s + h;
                uint32_t off = BUF_OOB;
                if (TAPS == 9) {
                    const int gy = ty0 + pt + tap / 3 - 1, gx = tx0 + n + tap % 3 - 1;
                    if (tap < 9 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) off = (uint32_t)((gy * p.W + gx) * 16);
                } else {
                    const long long q = lin0 + pt * 32 + n;
                    if (tap == 0 && q < npix) off = (uint32_t)(q * 16);
                }
                const uint4 v = buf_load16(sbuf, off, 0u);
                bf[s] = *reinterpret_cast<const Frag*>(&v);
            });
            f32x16 acc[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
            static_for<KS>([&](auto s_) {
                constexpr int s = decltype(s_)::value;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) Mma<T>::run(af[s][mi], bf[s], acc[mi]);
            });
            // ---- epilogue: LDS transpose -> (bias, skip, scale) -> 16-byte stores (conv_igemm.hip's arithmetic) ----------
            wave_sync();                                    // this wave's reads of the previous row are done
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(stage + stage_off<4>(n, stage_wslot(lane, mi, g))) =
                        make_float4(acc[mi][4 * g], acc[mi][4 * g + 1], acc[mi][4 * g + 2], acc[mi][4 * g + 3]);
            wave_sync();
            long long o_base; int nvalid;                   // element offset of the row's first pixel; valid pixels of the 32
            if (TAPS == 9) {
                const int gy = ty0 + pt;
                o_base = ((long long)gy * p.W + tx0) * p.outC;
                nvalid = gy < p.H ? min(32, p.W - tx0) : 0;
            } else {
                const long long q0 = lin0 + pt * 32;
                o_base = q0 * p.outC;
                nvalid = (int)(q0 < npix ? (npix - q0 < 32 ? npix - q0 : 32) : 0);
            }
#pragma unroll 2
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + l4;
                const float4 v0 = *reinterpret_cast<const float4*>(stage + stage_off<4>(row, 2 * c16));
                const float4 v1 = *reinterpret_cast<const float4*>(stage + stage_off<4>(row, 2 * c16 + 1));
                f32x2 v2[4] = {f32x2{v0.x, v0.y}, f32x2{v0.z, v0.w}, f32x2{v1.x, v1.y}, f32x2{v1.z, v1.w}};
                if (row < nvalid && co_ok) {
                    const long long o = o_base + (long long)row * p.outC + co;
                    if (p.skip) {
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
                    if (p.nt) store16_nt(out_b + o, make_uint4(pack2(v[0], v[1], (T*)nullptr), pack2(v[2], v[3], (T*)nullptr),
                                                                pack2(v[4], v[5], (T*)nullptr), pack2(v[6], v[7], (T*)nullptr)));
                    else store8(out_b + o, v);
                }
            }
        }
        // ---- statistics partials of this tile and pass: lanes of one cout octet are 16 apart, then the four waves ------------
        if (p.gn_part != nullptr) {
            float gs[8], gq[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { gs[2 * i] = gsum2[i].x; gs[2 * i + 1] = gsum2[i].y; gq[2 * i] = gsq2[i].x; gq[2 * i + 1] = gsq2[i].y; }
#pragma unroll
            for (int off = 16; off < 64; off <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { gs[e] += __shfl_xor(gs[e], off, 64); gq[e] += __shfl_xor(gq[e], off, 64); }
            __syncthreads();                                // (the previous pass' combine is done reading `red`)
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { red[(wave * BN + lane * 8 + e) * 2] = gs[e]; red[(wave * BN + lane * 8 + e) * 2 + 1] = gq[e]; }
            }
            __syncthreads();
            if (tid < BN && cout0 + tid < p.outC) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s0 += red[(w * BN + tid) * 2]; s1 += red[(w * BN + tid) * 2 + 1]; }
                float* dst = p.gn_part + ((long long)tile * p.outC + cout0 + tid) * 2;
                dst[0] = s0; dst[1] = s1;
            }
        }
    }
}

template <typename T, int TAPS>
__global__ __launch_bounds__(thin::THREADS, 2)
void conv_thin_kernel(const thin::Params p) {
    conv_thin_body<T, TAPS>(p, (int)blockIdx.x);
}
// The tiles of several problems of ONE layer (ragged micro-batches of a stream) in one launch: problem g owns the workgroups
// [first[g], first[g + 1]); its Params come from a device table.  A tile is computed by the code of its own launch.
template <typename T, int TAPS>
__global__ __launch_bounds__(thin::THREADS, 2)
void conv_thin_group_kernel(const thin::Params* __restrict__ gtab, const int* __restrict__ first, const int P) {
    int g = 0;
    while (g + 1 < P && (int)blockIdx.x >= first[g + 1]) ++g;           // (uniform: P <= a few dozen)
    const thin::Params p = gtab[g];
    conv_thin_body<T, TAPS>(p, (int)blockIdx.x - first[g]);
}

// ---- host side ---------------------------------------------------------------------------------------------------
bool conv_thin_supports(const storm_conv_args& a) {
    if (a.dtype != STORM_BF16 && a.dtype != STORM_F16) return false;
    if (a.nseg != 1 || a.out_f32 || a.tbias != nullptr) return false;
    const storm_conv_seg& g = a.seg[0];
    if (g.Ca != 8 || g.Cb != 0 || g.CinP != 16 || g.gn_ss != nullptr || g.w_bstride != 0) return false;
    if ((g.ntaps != 9 && g.ntaps != 1) || g.w_tapstride != (long long)g.w_rows * g.CinP) return false;
    if (a.outC % 8 != 0 || a.outC < 64) return false;                  // (narrow outputs: the generic small-tile kernel)
    const long long img = (long long)a.H * a.W;
    return img * 16 < (1LL << 31) && (long long)g.ntaps * g.w_rows * 32 < (1LL << 31) && img * a.outC < (1LL << 31);
}

static long long thin_params(const storm_conv_args& a, thin::Params& p) {
    memset(&p, 0, sizeof(p));
    const storm_conv_seg& g = a.seg[0];
    p.src = g.src_a; p.w = g.w; p.out = a.out; p.bias = a.bias; p.skip = a.skip; p.gn_part = a.gn_part;
    p.src_bstride = g.bstride_a; p.out_bstride = a.out_bstride; p.skip_bstride = a.skip_bstride;
    p.nt = (switches().gn_nt >> 2) & 1;
    p.B = a.B; p.H = a.H; p.W = a.W; p.outC = a.outC; p.Cout = a.Cout; p.w_rows = g.w_rows; p.scale = a.scale;
    p.tiles_x = cdiv(a.W, TILE_W);
    p.tiles_per_img = g.ntaps == 9 ? p.tiles_x * cdiv(a.H, TILE_H) : cdiv((long long)a.H * a.W, TILE_H * TILE_W);
    return (long long)a.B * p.tiles_per_img;
}

// grouped launch (conv_params.h): the host image = P Params, then first[P + 1] (the workgroup ranges); returns the workgroup count or -1
long long conv_thin_group_bytes(int P) { return ((long long)P * (long long)sizeof(thin::Params) + (P + 1) * 4 + 255) / 256 * 256; }
long long conv_thin_group_prepare(const storm_conv_args* a, int P, void* image) {
    thin::Params* tab = static_cast<thin::Params*>(image);
    int* first = reinterpret_cast<int*>(static_cast<char*>(image) + (long long)P * (long long)sizeof(thin::Params));
    long long n = 0;
    for (int g = 0; g < P; ++g) {
        if (!conv_thin_supports(a[g])) return -1;
        const storm_conv_seg &s0 = a[0].seg[0], &sg = a[g].seg[0];
        if (sg.w != s0.w || sg.ntaps != s0.ntaps || a[g].outC != a[0].outC || a[g].Cout != a[0].Cout || a[g].bias != a[0].bias || a[g].dtype != a[0].dtype ||
            a[g].scale != a[0].scale || (a[g].skip != nullptr) != (a[0].skip != nullptr) || (a[g].gn_part != nullptr) != (a[0].gn_part != nullptr)) return -1;
        first[g] = (int)n;
        n += thin_params(a[g], tab[g]);
        if (n >= (1LL << 31)) return -1;
    }
    first[P] = (int)n;
    return n;
}
template <typename T, int TAPS>
static int launch_thin_group(const void* dev_image, int P, long long ntiles, hipStream_t st) {
    auto kern = conv_thin_group_kernel<T, TAPS>;
    static bool attr_set = false;
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, thin::LDS_BYTES));
        attr_set = true;
    }
    const thin::Params* tab = static_cast<const thin::Params*>(dev_image);
    const int* first = reinterpret_cast<const int*>(static_cast<const char*>(dev_image) + (long long)P * (long long)sizeof(thin::Params));
    hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(thin::THREADS), thin::LDS_BYTES, st, tab, first, P);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}
int launch_conv_thin_group(const void* dev_image, int P, long long ntiles, int ntaps, int dtype, hipStream_t st) {
    STORM_CHECK(dev_image && P >= 1 && ntiles > 0, "storm_conv (thin group): bad arguments");
    const bool nine = ntaps == 9;
    if (dtype == STORM_F16) return nine ? launch_thin_group<half_t, 9>(dev_image, P, ntiles, st) : launch_thin_group<half_t, 1>(dev_image, P, ntiles, st);
    return nine ? launch_thin_group<bf16_t, 9>(dev_image, P, ntiles, st) : launch_thin_group<bf16_t, 1>(dev_image, P, ntiles, st);
}

template <typename T, int TAPS>
static int launch_thin(const storm_conv_args& a, hipStream_t st) {
    auto kern = conv_thin_kernel<T, TAPS>;
    static bool attr_set = false;                       // per instantiation; benign race (idempotent)
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, thin::LDS_BYTES));
        attr_set = true;
    }
    thin::Params p;
    const long long ntiles = thin_params(a, p);
    STORM_CHECK(ntiles > 0 && ntiles < (1LL << 31), "storm_conv: grid %lld out of range", ntiles);
    hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(thin::THREADS), thin::LDS_BYTES, st, p);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_conv_thin(const storm_conv_args& a, hipStream_t st) {
    const bool nine = a.seg[0].ntaps == 9;
    if (a.dtype == STORM_F16) return nine ? launch_thin<half_t, 9>(a, st) : launch_thin<half_t, 1>(a, st);
    return nine ? launch_thin<bf16_t, 9>(a, st) : launch_thin<bf16_t, 1>(a, st);
}

const char* conv_thin_kernel_name(int dtype, int ntaps) {
    if (dtype == STORM_F16) return ntaps == 9 ? "storm::conv_thin_kernel<storm::half_t, 9>" : "storm::conv_thin_kernel<storm::half_t, 1>";
    return ntaps == 9 ? "storm::conv_thin_kernel<storm::bf16_t, 9>" : "storm::conv_thin_kernel<storm::bf16_t, 1>";
}

}  // namespace storm
