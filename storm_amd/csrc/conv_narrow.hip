// 3x3 convolutions to at most four output channels (bf16 / fp16): the output pyramid of NCSN++ (ncsnpp.py:389-410, progressive
// "output_skip": conv3x3(act(GroupNorm(h))) -> `channels` planes at every level), with the GroupNorm-apply + SiLU fused into the
// operand load like every other conv here.  Same math and arguments as conv_igemm.hip, whose 32-cout tile ran these layers at
// 1.2 - 1.7 TB/s: 36 MFMA steps and nine pixel-fragment reads per 32-channel chunk for four useful output columns.
//
// With so few outputs the nine taps fit the OTHER matrix dimension: Z[q][tap][co] = sum_c w[tap][co][c] act(x[q][c]) is ONE 1x1
// GEMM with 36 rows (9 taps x 4 couts, two 32-row MFMA tiles) over the haloed pixel region, and
// out[y][x][co] = bias[co] + sum_tap Z[(y + dy, x + dx)][tap][co] is a nine-point gather of 16-byte vectors from LDS.  So
//   * every input pixel is fetched, normalised and multiplied ONCE (not once per tap): a lane loads 64 contiguous bytes of its
//     pixel per 64 channels straight into registers - no patch staging; the k order of the MFMA steps is chosen so that those
//     64 bytes are four consecutive k-steps of the lane's half (lane (n, h), step 4 G + j: channels 64 G + 32 h + 8 j ...), and
//     the weight image in LDS is laid out by that order (one conflict-free 16-byte read per tile and step);
//   * the accumulator of a 32-pixel fragment (lane (n, h): rows 8 g + 4 h + i = tap 2 g + h, cout i of pixel n) goes to LDS as
//     four + one 16-byte writes, 144 bytes per pixel; row stride 36 floats keeps writes and gathers bank-conflict free;
//   * one 8-wave workgroup per CU walks tiles of 20 x 32 output pixels (22 x 34 = 748 region pixels = 24 fragments, three per wave:
//     two waves per SIMD, because a single wave issues a VALU instruction only every other slot - tools/ubench/valu_rate); the raw
//     pixels of the NEXT fragment - across tiles too - are in flight while the current one is normalised and multiplied.
// What is left is the GroupNorm + SiLU arithmetic itself (two quarter-rate transcendentals per element, 1.2 x for the halo):
// the kernel is VALU-bound at about the HBM time of its input.
#include <cstring>
#include "conv_pipe_common.h"

namespace storm {

namespace narrow {
constexpr int THREADS = 512, NWAVES = THREADS / 64, TH = 20, TW = 32, RH = TH + 2, RW = TW + 2, NPX = RH * RW;
constexpr int NFRAG = (NPX + 31) / 32, FPW = NFRAG / NWAVES; // fragments per tile / per wave
constexpr int ZROW = 36;                                     // floats per region pixel: [9 taps][4 couts]
constexpr int Z_BYTES = NFRAG * 32 * ZROW * 4;
constexpr int WROWS = 40;                                    // rows per channel octet of the weight image: 36 + 4 zero rows
static_assert(NFRAG % NWAVES == 0, "every wave walks the same number of fragments");
struct Params {
    const void* src; const void* w; void* out; const float* bias; const float* gn_ss;
    long long src_bstride, out_bstride, w_tapstride;         // elements
    int B, H, W, C, Cout, CinP, silu, tiles_x, tiles_per_img, ntiles;
};
__host__ __device__ inline int lds_bytes(int C) { return Z_BYTES + (C / 8) * WROWS * 16 + C * 2 * 4; }
}  // namespace narrow

// CG = C / 64: 64-byte pieces per lane and pixel; SILU: the activation behind the fused GroupNorm affine (compile time: as a run-time
// flag it is if-converted into a select per value)
// GROUP: the tiles of several problems of ONE layer (ragged micro-batches of a stream, conv_params.h: GroupTile) in one persistent walk - the
// problem's Params come from a device table (`p` is then problem 0: the weights, bias and channel counts every problem shares)
template <typename T, int CG, bool SILU, bool GROUP>
__device__ __forceinline__ void conv_narrow_body(const narrow::Params& p, const narrow::Params* __restrict__ gtab, const pipe::GroupTile* __restrict__ glist, const int ntiles_all) {
    using namespace narrow;
    typedef typename Mma<T>::Frag Frag;
    constexpr int C = CG * 64, NG8 = C / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const Z = reinterpret_cast<float*>(smem);
    char* const Wl = smem + Z_BYTES;                         // [NG8][WROWS][16 B]
    float* const SS = reinterpret_cast<float*>(Wl + NG8 * WROWS * 16);   // [NG8][2][8]: this batch item's (scale, shift) table
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const bool gn = p.gn_ss != nullptr;

    // ---- weight image: row m = 4 tap + co of channel octet g8; rows of missing couts and rows 36 .. 39 are zero -------------------
    {
        const T* const w = reinterpret_cast<const T*>(p.w);
        for (int i = tid; i < NG8 * WROWS; i += THREADS) {
            const int g8 = i / WROWS, m = i - g8 * WROWS;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (m < 36 && (m & 3) < p.Cout) v = *reinterpret_cast<const uint4*>(w + (long long)(m >> 2) * p.w_tapstride + (long long)(m & 3) * p.CinP + g8 * 8);
            *reinterpret_cast<uint4*>(Wl + i * 16) = v;
        }
    }
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < p.Cout) bias4[e] = p.bias[e];
    }

    // ---- a fragment's raw pixels: lane (n, h) = region pixel 32 f + n, channels 64 G + 32 h ... + 32 of every G --------------------
    struct Raw { uint4 q[CG][4]; int qi; bool valid; };
    // tile -> (problem's parameters, batch item, first row, first column)
    struct At { const Params* q; int b, ty0, tx0, key; };
    auto locate = [&](int tile) {
        At a;
        if constexpr (GROUP) {
            const pipe::GroupTile te = glist[tile];
            a.q = gtab + te.problem; a.b = (int)te.b; a.ty0 = (int)(te.yx & 0xffffu); a.tx0 = (int)(te.yx >> 16); a.key = (int)(te.problem << 16 | te.b);
        } else {
            a.q = &p; a.b = tile / p.tiles_per_img;
            const int trem = tile - a.b * p.tiles_per_img;
            a.ty0 = (trem / p.tiles_x) * TH; a.tx0 = (trem % p.tiles_x) * TW; a.key = a.b;
        }
        return a;
    };
    auto fetch = [&](int tile, int f, Raw& r) {
        const At at = locate(tile);
        const Params& p = *at.q;
        const int b = at.b, ty0 = at.ty0, tx0 = at.tx0;
        const int qi = f * 32 + n, ry = qi / RW, rx = qi - ry * RW;
        const int y = ty0 - 1 + ry, x = tx0 - 1 + rx;
        r.qi = qi;
        r.valid = qi < NPX && y >= 0 && y < p.H && x >= 0 && x < p.W;
        if (r.valid) {
            const T* const px = reinterpret_cast<const T*>(p.src) + (long long)b * p.src_bstride + ((long long)y * p.W + x) * C + h * 32;
#pragma unroll
            for (int G = 0; G < CG; ++G)
#pragma unroll
                for (int j = 0; j < 4; ++j) r.q[G][j] = *reinterpret_cast<const uint4*>(px + G * 64 + j * 8);
        } else {
#pragma unroll
            for (int G = 0; G < CG; ++G)
#pragma unroll
                for (int j = 0; j < 4; ++j) r.q[G][j] = make_uint4(0u, 0u, 0u, 0u);
        }
    };

    const int ntiles = GROUP ? ntiles_all : p.ntiles;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    Raw cur, nxt;
    nxt.qi = 0; nxt.valid = false;
    fetch(tile, wave, cur);
    int ss_b = -1;
    for (;;) {
        const At at = locate(tile);
        const Params& p = *at.q;
        const int b = at.b, ty0 = at.ty0, tx0 = at.tx0;
        if (gn && at.key != ss_b) {                          // (uniform) this batch item's affine table
            const float4* const g = reinterpret_cast<const float4*>(p.gn_ss + (long long)b * C * 2);
            for (int i = tid; i < C * 2 / 4; i += THREADS) reinterpret_cast<float4*>(SS)[i] = g[i];
            ss_b = at.key;
        }
        __syncthreads();                                     // weight image / table visible; the previous tile's gather is done with Z
        const int next_tile = tile + (int)gridDim.x;
#pragma unroll 1
        for (int k = 0; k < FPW; ++k) {
            if (k + 1 < FPW) fetch(tile, wave + NWAVES * (k + 1), nxt);
            else if (next_tile < ntiles) fetch(next_tile, wave, nxt);
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int G = 0; G < CG; ++G)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int g8 = G * 8 + h * 4 + j;
                    uint4 v = cur.q[G][j];
                    if (gn) {
                        float ss[16];
                        load_ss<8>(SS, g8, ss);
                        v = gn_act_slot_t<SILU>(v, ss, (T*)nullptr);
                        if (!cur.valid) v = make_uint4(0u, 0u, 0u, 0u);      // (zero padding applies to the ACTIVATED tensor)
                    }
                    Frag bf, a0, a1;
                    memcpy(&bf, &v, 16);
                    const char* const wrow = Wl + g8 * (WROWS * 16);
                    a0 = *reinterpret_cast<const Frag*>(wrow + n * 16);
                    a1 = *reinterpret_cast<const Frag*>(wrow + (n < WROWS - 32 ? 32 + n : WROWS - 1) * 16);
                    Mma<T>::run(a0, bf, acc0);
                    Mma<T>::run(a1, bf, acc1);
                }
            float* const zq = Z + cur.qi * ZROW;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(zq + (2 * g + h) * 4) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
            if (h == 0) *reinterpret_cast<float4*>(zq + 8 * 4) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
            cur = nxt;
        }
        __syncthreads();
        // ---- gather: out[y][x][co] = bias[co] + sum over the nine taps of Z[(y + dy, x + dx)][tap][co] ----------------------------
        T* const out_b = reinterpret_cast<T*>(p.out) + (long long)b * p.out_bstride;
#pragma unroll
        for (int pp = tid; pp < TH * TW; pp += THREADS) {
            const int oy = pp / TW, ox = pp - oy * TW;
            const int y = ty0 + oy, x = tx0 + ox;
            if (y < p.H && x < p.W) {
                float s[8] = {bias4[0], bias4[1], bias4[2], bias4[3], 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float4 z = *reinterpret_cast<const float4*>(Z + ((oy + tap / 3) * RW + ox + tap % 3) * ZROW + tap * 4);
                    s[0] += z.x; s[1] += z.y; s[2] += z.z; s[3] += z.w;
                }
                store8(out_b + ((long long)y * p.W + x) * 8, s);
            }
        }
        if (next_tile >= ntiles) break;
        tile = next_tile;
    }
}

template <typename T, int CG, bool SILU>
__global__ __launch_bounds__(narrow::THREADS, 1)
void conv_narrow_kernel(const narrow::Params p) {
    conv_narrow_body<T, CG, SILU, false>(p, nullptr, nullptr, 0);
}
template <typename T, int CG, bool SILU>
__global__ __launch_bounds__(narrow::THREADS, 1)
void conv_narrow_group_kernel(const narrow::Params* __restrict__ gtab, const pipe::GroupTile* __restrict__ glist, const int ntiles) {
    conv_narrow_body<T, CG, SILU, true>(gtab[0], gtab, glist, ntiles);
}

// ---- host side ---------------------------------------------------------------------------------------------------
bool conv_narrow_supports(const storm_conv_args& a) {
    if (a.dtype != STORM_BF16 && a.dtype != STORM_F16) return false;
    if (a.nseg != 1 || a.out_f32 || a.tbias != nullptr || a.skip != nullptr || a.gn_part != nullptr) return false;
    if (a.outC != 8 || a.Cout < 1 || a.Cout > 4 || a.scale != 1.0f) return false;
    const storm_conv_seg& g = a.seg[0];
    if (g.ntaps != 9 || g.Cb != 0 || (g.Ca != 128 && g.Ca != 256) || g.CinP < g.Ca || g.w_bstride != 0 || g.w_rows < a.Cout) return false;
    return (long long)a.H * a.W * g.Ca < (1LL << 31) && a.H >= 1 && a.W >= 1;
}

static long long narrow_params(const storm_conv_args& a, narrow::Params& p) {
    memset(&p, 0, sizeof(p));
    const storm_conv_seg& g = a.seg[0];
    p.src = g.src_a; p.w = g.w; p.out = a.out; p.bias = a.bias; p.gn_ss = g.gn_ss; p.silu = g.gn_silu;
    p.src_bstride = g.bstride_a; p.out_bstride = a.out_bstride; p.w_tapstride = g.w_tapstride;
    p.B = a.B; p.H = a.H; p.W = a.W; p.C = g.Ca; p.Cout = a.Cout; p.CinP = g.CinP;
    p.tiles_x = cdiv(a.W, narrow::TW);
    p.tiles_per_img = p.tiles_x * cdiv(a.H, narrow::TH);
    return (long long)a.B * p.tiles_per_img;
}

// grouped launch (common.h): host images of the table / tile list; -1 = the problems are not one layer of this kernel
long long conv_narrow_group_bytes(int P) { return ((long long)P * (long long)sizeof(narrow::Params) + 255) / 256 * 256; }
long long conv_narrow_group_tiles(const storm_conv_args& a) { return (long long)a.B * cdiv(a.W, narrow::TW) * cdiv(a.H, narrow::TH); }
long long conv_narrow_group_prepare(const storm_conv_args* a, int P, void* table, pipe::GroupTile* tiles, long long max_tiles) {
    narrow::Params* tab = static_cast<narrow::Params*>(table);
    long long n = 0;
    for (int g = 0; g < P; ++g) {
        if (!conv_narrow_supports(a[g])) return -1;
        const storm_conv_seg &s0 = a[0].seg[0], &sg = a[g].seg[0];
        if (sg.w != s0.w || sg.Ca != s0.Ca || a[g].Cout != a[0].Cout || a[g].bias != a[0].bias || a[g].dtype != a[0].dtype ||
            (sg.gn_ss != nullptr) != (s0.gn_ss != nullptr) || sg.gn_silu != s0.gn_silu || a[g].H >= 65536 || a[g].W >= 65536) return -1;
        narrow_params(a[g], tab[g]);
        const int tiles_x = tab[g].tiles_x, tiles_y = cdiv(a[g].H, narrow::TH);
        for (int b = 0; b < a[g].B; ++b)
            for (int ty = 0; ty < tiles_y; ++ty)
                for (int tx = 0; tx < tiles_x; ++tx) {
                    if (n >= max_tiles) return -1;
                    pipe::GroupTile& t = tiles[n++];
                    t.problem = (unsigned)g; t.b = (unsigned)b; t.yx = (unsigned)(ty * narrow::TH) | ((unsigned)(tx * narrow::TW) << 16); t.tile = 0;
                }
    }
    return n;
}

template <typename T, int CG, bool SILU>
static int launch_narrow_group(const void* dev_table, const pipe::GroupTile* dev_tiles, long long ntiles, hipStream_t st) {
    auto kern = conv_narrow_group_kernel<T, CG, SILU>;
    const int lds = narrow::lds_bytes(CG * 64);
    static bool attr_set = false;
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(narrow::THREADS), lds, st, static_cast<const narrow::Params*>(dev_table), dev_tiles, (int)ntiles);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}
int launch_conv_narrow_group(const storm_conv_args& a0, const void* dev_table, const pipe::GroupTile* dev_tiles, long long ntiles, hipStream_t st) {
    STORM_CHECK(dev_table && dev_tiles && ntiles > 0 && ntiles < (1LL << 31), "storm_conv (narrow group): bad arguments");
    const bool c256 = a0.seg[0].Ca == 256, silu = a0.seg[0].gn_ss != nullptr && a0.seg[0].gn_silu != 0;
#define STORM_NG(T_, CG_) (silu ? launch_narrow_group<T_, CG_, true>(dev_table, dev_tiles, ntiles, st) : launch_narrow_group<T_, CG_, false>(dev_table, dev_tiles, ntiles, st))
    if (a0.dtype == STORM_F16) return c256 ? STORM_NG(half_t, 4) : STORM_NG(half_t, 2);
    return c256 ? STORM_NG(bf16_t, 4) : STORM_NG(bf16_t, 2);
#undef STORM_NG
}

template <typename T, int CG, bool SILU>
static int launch_narrow(const storm_conv_args& a, hipStream_t st) {
    auto kern = conv_narrow_kernel<T, CG, SILU>;
    const int lds = narrow::lds_bytes(CG * 64);
    static bool attr_set = false;                       // per instantiation; benign race (idempotent)
    if (!attr_set) {
        STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    narrow::Params p;
    const long long ntiles = narrow_params(a, p);
    STORM_CHECK(ntiles > 0 && ntiles < (1LL << 31), "storm_conv: grid %lld out of range", ntiles);
    p.ntiles = (int)ntiles;
    const int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(narrow::THREADS), lds, st, p);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

template <typename T, int CG>
static int launch_narrow_t(const storm_conv_args& a, hipStream_t st) {
    const bool silu = a.seg[0].gn_ss != nullptr && a.seg[0].gn_silu != 0;        // (without a fused GroupNorm operand the flag is unused)
    return silu ? launch_narrow<T, CG, true>(a, st) : launch_narrow<T, CG, false>(a, st);
}

int launch_conv_narrow(const storm_conv_args& a, hipStream_t st) {
    const bool c256 = a.seg[0].Ca == 256;
    if (a.dtype == STORM_F16) return c256 ? launch_narrow_t<half_t, 4>(a, st) : launch_narrow_t<half_t, 2>(a, st);
    return c256 ? launch_narrow_t<bf16_t, 4>(a, st) : launch_narrow_t<bf16_t, 2>(a, st);
}

const char* conv_narrow_kernel_name(const storm_conv_args& a) {
    static thread_local char buf[96];
    const bool silu = a.seg[0].gn_ss != nullptr && a.seg[0].gn_silu != 0;
    snprintf(buf, sizeof(buf), "storm::conv_narrow_kernel<storm::%s, %d, %s>", a.dtype == STORM_F16 ? "half_t" : "bf16_t", a.seg[0].Ca / 64,
             silu ? "true" : "false");
    return buf;
}

}  // namespace storm
