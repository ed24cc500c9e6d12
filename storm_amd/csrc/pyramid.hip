// The two 8-channel pyramids of NCSN++ (progressive "input_skip" / "output_skip", ncsnpp.py:352-355 and 389-410), each as ONE launch.
//
// The reference walks them level by level between the blocks of the U-Net:
//   input_pyramid  = pyramid_downsample(input_pyramid)            (FIR x2 down, up_or_down_sampling.py:230-257) before every Combine,
//   pyramid        = pyramid_upsample(pyramid) + conv3x3(act(GN(h)))  (FIR x2 up, :195-228) after every level of the decoder,
// and round 1-5 of this engine did the same: one fir_kernel launch per level (six per evaluation of `ncsnpp`, twelve of `ncsnpplarge`),
// each a few microseconds of work over 8-channel tensors behind a full launch.  But neither chain depends on the U-Net's activations for
// more than its `+ ph` terms: the input pyramid is a function of the network input alone, and the output pyramid is
//   p_0 = ph_0 + up(ph_1 + up(ph_2 + up(ph_3)))
// over the four narrow convolutions' outputs, consumed only by the output head.  So
//   * storm_input_pyramid packs the complex inputs (storm_pack_input) AND filters every level down in one launch at the start of the
//     evaluation (up to three FIR steps per launch: a workgroup owns a tile of the coarsest level and recomputes the haloed regions of the
//     finer ones in LDS - 1.75 x the reads of a 33-MB tensor);
//   * storm_output_pyramid runs the whole up chain AND the output head (storm_output_head) in one launch at its end: a workgroup owns a
//     32 x 64-pixel tile of the finest level and computes the regions of the coarser levels it needs (18 x 34, 12 x 20, ... pixels) in LDS.
// Every value goes through the arithmetic of the kernels it replaces (the same fmaf chain over the same taps, every level rounded to the
// storage type), so the results are theirs bit for bit (tests/test_ops.py::test_pyramids_equal_their_chains).
#include <cstring>
#include "common.h"

namespace storm {
namespace pyr {
constexpr int THREADS = 1024;                  // 16 waves: the levels are short dependent phases - the parallelism has to come from the width
constexpr int MAXD = 3;                      // FIR x2 down steps per launch
constexpr int MAXL = 8;                      // levels of the output pyramid
constexpr int OTH = 32, OTW = 64;            // finest-level tile of the output pyramid

struct InParams {
    const float* in[3]; int n_in;            // PACK: the complex inputs [B][H][W] (re, im)
    void* lvl[MAXD + 1];                     // level 0 .. nd: [B][H >> k][W >> k][8]
    int nd, B, H, W;                         // FIR steps of this launch; the level-0 image
    int th, tw, tiles_y, tiles_x;            // tile of the coarsest level per workgroup
};
struct OutParams {
    const void* ph[MAXL]; int nl;            // ph[k]: [B][H >> k][W >> k][8], finest first
    const float* t; const float* W; const float* bias; float* out;
    int cin, B, H, Wd, tiles_y, tiles_x; float sign;
};
}  // namespace pyr

// the network input of one pixel: x -> 2x - 1 on (re, im) of every complex input, zero padded to 8 channels (ncsnpp.py:289-296, 321-323)
__device__ __forceinline__ void pack_pixel(const float* const (&in)[3], int n_in, long long i, float (&v)[8]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < n_in) {
            const float2 z = reinterpret_cast<const float2*>(in[k])[i];
            v[2 * k] = 2.0f * z.x - 1.0f;
            v[2 * k + 1] = 2.0f * z.y - 1.0f;
        } else {
            v[2 * k] = 0.f; v[2 * k + 1] = 0.f;
        }
    }
    v[6] = 0.f; v[7] = 0.f;
}
// a value as the storage type holds it (what the level-by-level kernels wrote and read back)
template <typename T> __device__ __forceinline__ void round8(float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        alignas(16) T tmp[8];
        store8(tmp, v);
        load8(tmp, v);
    }
}

// ---- input pyramid: level 0 = pack (PACK) or the given tensor, level k = FIR x2 down of level k - 1 (fir_kernel<T, 2>: k = [1,3,3,1] / 8 per
// axis over 2o - 1 .. 2o + 2, zero boundary, taps accumulated in (i, j) order) ----------------------------------------------------------------
template <typename T, bool PACK>
__global__ __launch_bounds__(pyr::THREADS)
void input_pyramid_kernel(const pyr::InParams p) {
    using namespace pyr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* const lds = reinterpret_cast<T*>(smem);
    const int tid = threadIdx.x, nd = p.nd;
    int wg = blockIdx.x;
    const int tx = wg % p.tiles_x; wg /= p.tiles_x;
    const int ty = wg % p.tiles_y, b = wg / p.tiles_y;
    int r0[MAXD + 1], c0[MAXD + 1], nr[MAXD + 1], nc[MAXD + 1], off[MAXD + 1];
    r0[nd] = ty * p.th; c0[nd] = tx * p.tw; nr[nd] = p.th; nc[nd] = p.tw;
    for (int k = nd - 1; k >= 0; --k) { r0[k] = 2 * r0[k + 1] - 1; c0[k] = 2 * c0[k + 1] - 1; nr[k] = 2 * nr[k + 1] + 2; nc[k] = 2 * nc[k + 1] + 2; }
    off[0] = 0;
    for (int k = 0; k < nd; ++k) off[k + 1] = off[k] + nr[k] * nc[k];
    // level 0: the haloed region into LDS (zeros outside the image); PACK also writes the pixels this workgroup owns
    {
        const int H = p.H, W = p.W;
        const int oy0 = (ty * p.th) << nd, ox0 = (tx * p.tw) << nd, oy1 = oy0 + (p.th << nd), ox1 = ox0 + (p.tw << nd);
        T* const g0 = static_cast<T*>(p.lvl[0]) + (long long)b * H * W * 8;
        // (every load of a thread's pixels is issued before the first store: one memory round trip per U pixels, not per pixel)
        constexpr int U = 4;
        const int n0 = nr[0] * nc[0];
        for (int q0 = tid; q0 < n0; q0 += U * THREADS) {
            float v[U][8];
            long long gpx[U];
            bool own[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * THREADS;
                const int ry = q / nc[0], rx = q - ry * nc[0];
                const int gy = r0[0] + ry, gx = c0[0] + rx;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
                own[u] = false; gpx[u] = 0;
                if (q < n0 && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    gpx[u] = (long long)gy * W + gx;
                    if constexpr (PACK) {
                        pack_pixel(p.in, p.n_in, (long long)b * H * W + gpx[u], v[u]);
                        own[u] = gy >= oy0 && gy < oy1 && gx >= ox0 && gx < ox1;
                    } else {
                        load8(g0 + gpx[u] * 8, v[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * THREADS;
                if (q >= n0) continue;
                if (PACK && own[u]) store8(g0 + gpx[u] * 8, v[u]);
                store8(lds + (long long)q * 8, v[u]);
            }
        }
    }
    __syncthreads();
    const float kf[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    for (int k = 1; k <= nd; ++k) {
        const int H = p.H >> k, W = p.W >> k;
        const int oy0 = (ty * p.th) << (nd - k), ox0 = (tx * p.tw) << (nd - k), oy1 = oy0 + (p.th << (nd - k)), ox1 = ox0 + (p.tw << (nd - k));
        T* const gk = static_cast<T*>(p.lvl[k]) + (long long)b * H * W * 8;
        const T* const src = lds + (long long)off[k - 1] * 8;
        const int ncs = nc[k - 1];
        for (int q = tid; q < nr[k] * nc[k]; q += THREADS) {
            const int ry = q / nc[k], rx = q - ry * nc[k];
            const int gy = r0[k] + ry, gx = c0[k] + rx;
            const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (inside) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x[8];
                        load8(src + ((long long)(2 * ry + i) * ncs + 2 * rx + j) * 8, x);
                        const float w = kf[i] * kf[j];
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, x[e], acc[e]);
                    }
                if (gy >= oy0 && gy < oy1 && gx >= ox0 && gx < ox1) store8(gk + ((long long)gy * W + gx) * 8, acc);
            }
            if (k < nd) store8(lds + ((long long)off[k] + q) * 8, acc);
        }
        __syncthreads();
    }
}

// ---- output pyramid + head: p_{nl-1} = ph_{nl-1}; p_k = T(up(p_{k+1}) + ph_k) (fir_kernel<T, 1> with its add operand: out[2i + a] =
// 9/16, 3/16, 3/16, 1/16 over (i, i -/+ 1) per axis, zero boundary); out = sign (W . (p_0 / t_b) + bias) (output_head_kernel) --------------------
template <typename T>
__global__ __launch_bounds__(pyr::THREADS)
void output_pyramid_kernel(const pyr::OutParams p) {
    using namespace pyr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const lds = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, nl = p.nl;
    int wg = blockIdx.x;
    const int tx = wg % p.tiles_x; wg /= p.tiles_x;
    const int ty = wg % p.tiles_y, b = wg / p.tiles_y;
    int r0[MAXL], c0[MAXL], nr[MAXL], nc[MAXL], off[MAXL + 1];
    r0[0] = ty * OTH; c0[0] = tx * OTW; nr[0] = OTH; nc[0] = OTW;
    off[0] = 0; off[1] = 0;
    for (int k = 1; k < nl; ++k) {
        r0[k] = (r0[k - 1] >> 1) - 1; c0[k] = (c0[k - 1] >> 1) - 1;
        nr[k] = ((r0[k - 1] + nr[k - 1] - 1) >> 1) + 1 - r0[k] + 1; nc[k] = ((c0[k - 1] + nc[k - 1] - 1) >> 1) + 1 - c0[k] + 1;
        off[k + 1] = off[k] + nr[k] * nc[k];
    }
    // the four-tap up-sampling of level k + 1 (in LDS) at pixel (gy, gx) of level k, plus ph_k, as the storage type holds it
    auto level_value = [&](int k, int gy, int gx, int Wk, float (&v)[8]) {
        const T* const phk = static_cast<const T*>(p.ph[k]) + (long long)b * (p.H >> k) * Wk * 8;
        if (k == nl - 1) { load8(phk + ((long long)gy * Wk + gx) * 8, v); return; }
        const float* const src = lds + (long long)off[k + 1] * 8;
        const int iy = gy >> 1, ix = gx >> 1;
        const int ny = (gy & 1) ? iy + 1 : iy - 1, nx = (gx & 1) ? ix + 1 : ix - 1;
        const int ncs = nc[k + 1], ry = iy - r0[k + 1], rx = ix - c0[k + 1], qy = ny - r0[k + 1], qx = nx - c0[k + 1];
        const float* const a00 = src + ((long long)ry * ncs + rx) * 8;
        const float* const a01 = src + ((long long)ry * ncs + qx) * 8;
        const float* const a10 = src + ((long long)qy * ncs + rx) * 8;
        const float* const a11 = src + ((long long)qy * ncs + qx) * 8;
        float add[8];
        load8(phk + ((long long)gy * Wk + gx) * 8, add);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float s = fmaf(0.5625f, a00[e], 0.f);
            s = fmaf(0.1875f, a01[e], s);
            s = fmaf(0.1875f, a10[e], s);
            s = fmaf(0.0625f, a11[e], s);
            v[e] = s + add[e];
        }
        round8<T>(v);
    };
    for (int k = nl - 1; k >= 1; --k) {
        const int Hk = p.H >> k, Wk = p.Wd >> k;
        for (int q = tid; q < nr[k] * nc[k]; q += THREADS) {
            const int ry = q / nc[k], rx = q - ry * nc[k];
            const int gy = r0[k] + ry, gx = c0[k] + rx;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < Hk && gx >= 0 && gx < Wk) level_value(k, gy, gx, Wk, v);
            float* const d = lds + ((long long)off[k] + q) * 8;
            *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        __syncthreads();
    }
    const float tb = p.t ? p.t[b] : 1.0f;
    const int cin = p.cin;
    float2* const out_b = reinterpret_cast<float2*>(p.out) + (long long)b * p.H * p.Wd;
    // (both pixels of a thread are computed before either is stored: their loads share one round trip)
    constexpr int PPT = OTH * OTW / THREADS;
    static_assert(OTH * OTW % THREADS == 0, "whole pixels per thread");
    float v[PPT][8];
    bool ok[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int q = tid + j * THREADS;
        const int gy = r0[0] + q / OTW, gx = c0[0] + q % OTW;
        ok[j] = gy < p.H && gx < p.Wd;
        if (ok[j]) level_value(0, gy, gx, p.Wd, v[j]);
    }
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        if (!ok[j]) continue;
        const int q = tid + j * THREADS;
        const int gy = r0[0] + q / OTW, gx = c0[0] + q % OTW;
        float o0 = 0.f, o1 = 0.f;
        for (int c = 0; c < cin; ++c) {
            const float h = p.t ? v[j][c] / tb : v[j][c];
            o0 = fmaf(p.W[c], h, o0);
            o1 = fmaf(p.W[cin + c], h, o1);
        }
        out_b[(long long)gy * p.Wd + gx] = make_float2(p.sign * (o0 + p.bias[0]), p.sign * (o1 + p.bias[1]));
    }
}

}  // namespace storm

using namespace storm;

// levels[k] = [B][F >> k][T >> k][8] for k = 0 .. n_levels - 1.  cplx_in != NULL: level 0 is WRITTEN (the packed inputs, storm_pack_input);
// cplx_in == NULL: level 0 is read (the continuation of a pyramid deeper than three steps).  At most pyr::MAXD + 1 levels per call.
extern "C" int storm_input_pyramid(const float* const* cplx_in, int n_in, void* const* levels, int n_levels, int B, int F, int T, int dtype,
                                   storm_stream_t s) {
    STORM_CHECK(levels && n_levels >= 1 && n_levels <= pyr::MAXD + 1 && B > 0 && F > 0 && T > 0, "storm_input_pyramid: bad arguments (n_levels=%d)", n_levels);
    STORM_CHECK(cplx_in == nullptr || (n_in >= 1 && n_in <= 3), "storm_input_pyramid: n_in=%d", n_in);
    const int nd = n_levels - 1;
    STORM_CHECK(F % (1 << nd) == 0 && T % (1 << nd) == 0, "storm_input_pyramid: %d x %d is not divisible by 2^%d", F, T, nd);
    pyr::InParams p;
    memset(&p, 0, sizeof(p));
    for (int k = 0; k < n_levels; ++k) { STORM_CHECK(levels[k], "storm_input_pyramid: level %d is NULL", k); p.lvl[k] = levels[k]; }
    if (cplx_in) for (int i = 0; i < n_in; ++i) { STORM_CHECK(cplx_in[i], "storm_input_pyramid: null input %d", i); p.in[i] = cplx_in[i]; }
    p.n_in = cplx_in ? n_in : 0; p.nd = nd; p.B = B; p.H = F; p.W = T;
    // the coarsest level's tile: 4 x 8 pixels behind three steps, 8 x 16 behind two, 16 x 32 behind one (a level-0 region of <= 46 x 78 pixels)
    p.th = 32 >> nd; p.tw = 64 >> nd;
    const int Hc = F >> nd, Wc = T >> nd;
    p.tiles_y = cdiv(Hc, p.th); p.tiles_x = cdiv(Wc, p.tw);
    long long px = 0;
    { int nr = p.th, nc = p.tw; for (int k = nd - 1; k >= 0; --k) { nr = 2 * nr + 2; nc = 2 * nc + 2; px += (long long)nr * nc; } }
    if (nd == 0) px = (long long)p.th * p.tw;                  // (pack only: the region is the tile)
    const long long grid = (long long)B * p.tiles_y * p.tiles_x;
    STORM_CHECK(grid < (1LL << 31), "storm_input_pyramid: grid %lld out of range", grid);
    hipStream_t st = (hipStream_t)s;
#define STORM_IP(T_) do { const int lds = (int)(px * 8 * (long long)sizeof(T_)); \
        if (cplx_in) { auto kern = input_pyramid_kernel<T_, true>; static bool set_ = false; \
            if (!set_) { STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set_ = true; } \
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pyr::THREADS), lds, st, p); } \
        else { auto kern = input_pyramid_kernel<T_, false>; static bool set_ = false; \
            if (!set_) { STORM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set_ = true; } \
            hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pyr::THREADS), lds, st, p); } } while (0)
    if (dtype == STORM_BF16) STORM_IP(bf16_t);
    else if (dtype == STORM_F16) STORM_IP(half_t);
    else if (dtype == STORM_F32) STORM_IP(float);
    else STORM_CHECK(false, "storm_input_pyramid: dtype %d", dtype);
#undef STORM_IP
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

// ph[k] = [B][F >> k][T >> k][8] (the narrow convolutions' outputs, finest first), n_levels <= 8; the rest as storm_output_head.
extern "C" int storm_output_pyramid(const void* const* ph, int n_levels, const float* t, const float* W, const float* bias, int cin,
                                    float* out_cplx, int B, int F, int T, int negate, int dtype, storm_stream_t s) {
    STORM_CHECK(ph && n_levels >= 1 && n_levels <= pyr::MAXL && W && bias && out_cplx && cin >= 1 && cin <= 8 && B > 0 && F > 0 && T > 0,
                "storm_output_pyramid: bad arguments (n_levels=%d cin=%d)", n_levels, cin);
    STORM_CHECK(F % (1 << (n_levels - 1)) == 0 && T % (1 << (n_levels - 1)) == 0, "storm_output_pyramid: %d x %d is not divisible by 2^%d", F, T, n_levels - 1);
    pyr::OutParams p;
    memset(&p, 0, sizeof(p));
    for (int k = 0; k < n_levels; ++k) { STORM_CHECK(ph[k], "storm_output_pyramid: level %d is NULL", k); p.ph[k] = ph[k]; }
    p.nl = n_levels; p.t = t; p.W = W; p.bias = bias; p.out = out_cplx; p.cin = cin; p.B = B; p.H = F; p.Wd = T;
    p.sign = negate ? -1.0f : 1.0f;
    p.tiles_y = cdiv(F, pyr::OTH); p.tiles_x = cdiv(T, pyr::OTW);
    long long px = 0;
    { int nr = pyr::OTH, nc = pyr::OTW; for (int k = 1; k < n_levels; ++k) { nr = (nr + 1) / 2 + 3; nc = (nc + 1) / 2 + 3; px += (long long)nr * nc; } }
    const int lds = (int)(px * 8 * 4);
    const long long grid = (long long)B * p.tiles_y * p.tiles_x;
    STORM_CHECK(grid < (1LL << 31) && lds <= 64 * 1024, "storm_output_pyramid: grid %lld / LDS %d out of range", grid, lds);
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) hipLaunchKernelGGL((output_pyramid_kernel<bf16_t>), dim3((unsigned)grid), dim3(pyr::THREADS), lds, st, p);
    else if (dtype == STORM_F16) hipLaunchKernelGGL((output_pyramid_kernel<half_t>), dim3((unsigned)grid), dim3(pyr::THREADS), lds, st, p);
    else if (dtype == STORM_F32) hipLaunchKernelGGL((output_pyramid_kernel<float>), dim3((unsigned)grid), dim3(pyr::THREADS), lds, st, p);
    else STORM_CHECK(false, "storm_output_pyramid: dtype %d", dtype);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}
