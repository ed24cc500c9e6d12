// GroupNorm (+SiLU) and FIR x2 resampling on NHWC activations — the HBM-bound family.
//
// Replaces nn.GroupNorm(eps=1e-6)+nn.SiLU and upsample_2d/downsample_2d (upfirdn2d CUDA
// kernel modes 3/5) of the reference; see include/storm_hip.h for the file:line map.
// Design: every thread owns a fixed octet of channels (16 B bf16 / 32 B fp32 per access,
// a pixel's channels are contiguous so a wave reads whole 128-B lines), statistics are
// reduced per thread in fp32 over <=256 pixels, then in fp64 across threads / workgroups
// (one fp64 atomicAdd per (batch, group) per workgroup).  The resampling variants fuse
// GN-apply + SiLU + FIR of BOTH the activated and the raw tensor (BigGAN block) in one pass.
#include <cstdlib>
#include "common.h"

namespace storm {

constexpr int GN_MAX_C = 1024;

struct GnGeom { int C8, PL, NT; };
static inline GnGeom gn_geom(int C) {
    GnGeom g; g.C8 = C / 8; g.PL = 256 / g.C8; if (g.PL < 1) g.PL = 1; g.NT = g.C8 * g.PL; return g;
}

// Pixels each thread walks: aim at >= ~4096 workgroups so small feature maps are not latency bound
// (a thread that loops 256 dependent loads takes ~80 us whatever the tensor size).
static inline int pixels_per_thread(long long total_pixels, int PL, int max_per_thread) {
    long long per = total_pixels / ((long long)PL * 4096);
    if (per < 2) per = 2;
    if (per > max_per_thread) per = max_per_thread;
    return (int)per;
}

template <typename T>
__device__ __forceinline__ void load_cat8(const T* xa, int Ca, const T* xb, int Cb, long long pix, int c,
                                          float (&v)[8]) {
    if (c < Ca) load8(xa + pix * Ca + c, v);
    else load8(xb + pix * Cb + (c - Ca), v);
}

template <typename T>
__global__ void gn_stats_kernel(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb,
                                int HW, int G, double* __restrict__ stats, int ppb, int C8, int PL) {
    __shared__ float red[256 * 16];
    __shared__ double chs[GN_MAX_C * 2];
    const int C = Ca + Cb, b = blockIdx.y, tid = threadIdx.x;
    const int oct = tid % C8, pl = tid / C8, c = oct * 8;
    const long long base = (long long)b * HW;
    const int p0 = blockIdx.x * ppb, p1 = min(HW, p0 + ppb);
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    for (int p = p0 + pl; p < p1; p += PL) {
        float v[8];
        load_cat8(xa, Ca, xb, Cb, base + p, c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += v[e]; ss[e] = fmaf(v[e], v[e], ss[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = s[e]; red[tid * 16 + 8 + e] = ss[e]; }
    __syncthreads();
    if (tid < C8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            double a = 0.0, q = 0.0;
            for (int k = 0; k < PL; ++k) { a += (double)red[(k * C8 + tid) * 16 + e]; q += (double)red[(k * C8 + tid) * 16 + 8 + e]; }
            chs[(tid * 8 + e) * 2] = a; chs[(tid * 8 + e) * 2 + 1] = q;
        }
    }
    __syncthreads();
    const int gs = C / G;
    for (int g = tid; g < G; g += blockDim.x) {
        double a = 0.0, q = 0.0;
        for (int k = 0; k < gs; ++k) { a += chs[(g * gs + k) * 2]; q += chs[(g * gs + k) * 2 + 1]; }
        atomicAdd(&stats[((long long)b * G + g) * 2], a);
        atomicAdd(&stats[((long long)b * G + g) * 2 + 1], q);
    }
}

// Finalise fused statistics: sum the per-tile fp32 partials of a conv epilogue into [B][G][2] fp64.
__global__ __launch_bounds__(256)
void gn_finalize_kernel(const float* __restrict__ pa, int Ca, int tiles_a, const float* __restrict__ pb,
                        int Cb, int tiles_b, int G, double* __restrict__ stats, long long count,
                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                        float* __restrict__ ss) {
    // one workgroup (4 waves) per (group, batch item): the group's gs channels of a tile are contiguous (gs float2), so a
    // thread walks (tile, channel) pairs with the channel fastest; fp64 partial sums, fixed reduction order (deterministic)
    __shared__ double red[2][4];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int C = Ca + Cb, gs = C / G;
    double s0 = 0.0, s1 = 0.0;
    const int c_lo = g * gs, c_hi = c_lo + gs;
    for (int part = 0; part < 2; ++part) {
        const float* p = part == 0 ? pa : pb;
        const int Cs = part == 0 ? Ca : Cb, nt = part == 0 ? tiles_a : tiles_b, off = part == 0 ? 0 : Ca;
        const int lo = max(c_lo, off) - off, hi = min(c_hi, off + Cs) - off;       // this group's channels inside the part
        const int w = hi - lo;
        if (p == nullptr || w <= 0) continue;
        const float2* q = reinterpret_cast<const float2*>(p) + (long long)b * nt * Cs + lo;
        for (int i = tid; i < nt * w; i += 256) {
            const int t = i / w, k = i - t * w;
            const float2 v = q[(long long)t * Cs + k];
            s0 += (double)v.x; s1 += (double)v.y;
        }
    }
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s0; red[1][tid >> 6] = s1; }
    __syncthreads();
    s0 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s1 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    if (tid == 0) { stats[((long long)b * G + g) * 2] = s0; stats[((long long)b * G + g) * 2 + 1] = s1; }
    if (ss != nullptr) {
        const double n = (double)gs * (double)count;
        const double m = s0 / n;
        double var = s1 / n - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        for (int k = tid; k < gs; k += 256) {
            const int c = g * gs + k;
            const float sc = rstd * gamma[c];
            float* const o = ss + ((long long)b * C + (c & ~7)) * 2 + (c & 7);     // [C/8][2][8]: 8 scales, then 8 shifts
            o[0] = sc;
            o[8] = beta[c] - (float)m * sc;
        }
    }
}

// FIR taps: down: k = [1,3,3,1]/8 per axis over input 2o-1..2o+2; up: out[2i+a] = 3/4 x[i] + 1/4 x[i -/+ 1].
struct GnParams { float mean[8], a[8], beta[8]; };

template <typename T, bool ACT_PATH>
__device__ __forceinline__ void fetch8(const T* xa, int Ca, const T* xb, int Cb, long long base, int H, int W,
                                       int iy, int ix, int c, const GnParams& gp, int silu, float w, float (&acc)[8]) {
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) return;       // zero boundary
    float v[8];
    load_cat8(xa, Ca, xb, Cb, base + (long long)iy * W + ix, c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float y = v[e];
        if (ACT_PATH) {
            y = (y - gp.mean[e]) * gp.a[e] + gp.beta[e];
            if (silu) y = silu_f(y);
        }
        acc[e] = fmaf(w, y, acc[e]);
    }
}

// Returns the (optionally GN+SiLU transformed) FIR-resampled octet at output pixel (oy, ox).
template <typename T, int RESAMPLE, bool ACT_PATH>
__device__ __forceinline__ void gather8(const T* xa, int Ca, const T* xb, int Cb, long long base, int H, int W,
                                        int oy, int ox, int c, const GnParams& gp, int silu, float (&out)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = 0.f;
    if (RESAMPLE == 0) {
        fetch8<T, ACT_PATH>(xa, Ca, xb, Cb, base, H, W, oy, ox, c, gp, silu, 1.0f, out);
    } else if (RESAMPLE == 1) {          // up x2
        const int iy = oy >> 1, ix = ox >> 1;
        const int ny = (oy & 1) ? iy + 1 : iy - 1, nx = (ox & 1) ? ix + 1 : ix - 1;
        fetch8<T, ACT_PATH>(xa, Ca, xb, Cb, base, H, W, iy, ix, c, gp, silu, 0.5625f, out);
        fetch8<T, ACT_PATH>(xa, Ca, xb, Cb, base, H, W, iy, nx, c, gp, silu, 0.1875f, out);
        fetch8<T, ACT_PATH>(xa, Ca, xb, Cb, base, H, W, ny, ix, c, gp, silu, 0.1875f, out);
        fetch8<T, ACT_PATH>(xa, Ca, xb, Cb, base, H, W, ny, nx, c, gp, silu, 0.0625f, out);
    } else {                              // down x2
        const float k[4] = {0.125f, 0.375f, 0.375f, 0.125f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fetch8<T, ACT_PATH>(xa, Ca, xb, Cb, base, H, W, 2 * oy - 1 + i, 2 * ox - 1 + j, c, gp, silu, k[i] * k[j], out);
    }
}

template <typename T, int RESAMPLE>
__global__ void gn_apply_kernel(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb,
                                int H, int W, int G, const double* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                int silu, T* __restrict__ out_act, T* __restrict__ out_raw,
                                int ppb, int C8, int PL) {
    const int C = Ca + Cb, b = blockIdx.y, tid = threadIdx.x;
    const int oct = tid % C8, pl = tid / C8, c = oct * 8;
    const int OH = RESAMPLE == 1 ? 2 * H : (RESAMPLE == 2 ? H / 2 : H);
    const int OW = RESAMPLE == 1 ? 2 * W : (RESAMPLE == 2 ? W / 2 : W);
    const int gs = C / G;
    const double n = (double)gs * H * W;
    GnParams gp;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c + e) / gs;
        const double m = stats[((long long)b * G + g) * 2] / n;
        double var = stats[((long long)b * G + g) * 2 + 1] / n - m * m;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        gp.mean[e] = (float)m;
        gp.a[e] = rstd * gamma[c + e];
        gp.beta[e] = beta[c + e];
    }
    const long long ibase = (long long)b * H * W, obase = (long long)b * OH * OW;
    const int OHW = OH * OW;
    const int p0 = blockIdx.x * ppb, p1 = min(OHW, p0 + ppb);
    for (int p = p0 + pl; p < p1; p += PL) {
        const int oy = p / OW, ox = p - oy * OW;
        float v[8];
        gather8<T, RESAMPLE, true>(xa, Ca, xb, Cb, ibase, H, W, oy, ox, c, gp, silu, v);
        store8(out_act + (obase + p) * C + c, v);
        if (RESAMPLE != 0 && out_raw != nullptr) {
            gather8<T, RESAMPLE, false>(xa, Ca, xb, Cb, ibase, H, W, oy, ox, c, gp, 0, v);
            store8(out_raw + (obase + p) * C + c, v);
        }
    }
}

// Fused GN-apply + SiLU + FIR x2 (up or down) of BOTH the activated and the raw tensor, LDS tiled:
// a workgroup stages a 10x18-pixel input tile (8x16 core + 1-pixel halo) of one 128-byte channel
// group, applying the normalisation + SiLU ONCE per input element, then filters from LDS.
// Output tile: 4x8 pixels (down) / 16x32 pixels (up).
// Persistent workgroups with a register prefetch: a workgroup walks a contiguous run of tiles (so its GroupNorm table rarely
// changes and the halo columns it shares with its previous tile are L2 / L1 hits); the NEXT tile's
// global loads are issued as soon as the current tile's registers have been written to LDS, so they fly under the
// current tile's filter + stores (with one tile per workgroup the kernel was bound by the load -> transform -> store
// latency chain at 3 workgroups per CU: 1.7 TB/s down / 3.3 TB/s up).
constexpr int RS_IH = 10, RS_IW = 18, RS_NPIX = RS_IH * RS_IW;
// LDS position of staged pixel p (128 B each).  Down-sampling: the lanes of one ds_read_b128 group read the pixels p, p + 2,
// p + 4, p + 6 (four output columns, stride two), 64 B each - in a linear image those are 256 B apart, the same banks (every
// filter read 2-way conflicted); swapping the last two pixels of every group of four puts p + 2 on the other 128-byte half.
template <int RESAMPLE> __host__ __device__ inline int rs_pos(int p) { return RESAMPLE == 2 ? (p ^ ((p >> 1) & 1)) : p; }
static_assert(RS_NPIX % 4 == 0, "rs_pos permutes inside groups of four pixels");
template <typename T, int RESAMPLE>
__global__ __launch_bounds__(256)
void gn_apply_resample_kernel(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb,
                              int H, int W, int G, const double* __restrict__ stats,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                              int silu, T* __restrict__ out_act, T* __restrict__ out_raw, int tiles_x, int tiles_y, int ncg,
                              int total_tiles) {
    constexpr int PER16 = Elem<T>::PER16;           // elements per 16-byte slot
    constexpr int CG = 8 * PER16;                   // channels per workgroup (128 B per pixel)
    __shared__ __attribute__((aligned(16))) char tile[2 * RS_NPIX * 128];
    __shared__ float gtab[2 * CG];
    char* const t_act = tile;
    char* const t_raw = tile + RS_NPIX * 128;
    const int C = Ca + Cb, tid = threadIdx.x;
    const int slot = tid & 7;
    const int OH = RESAMPLE == 1 ? 2 * H : H / 2, OW = RESAMPLE == 1 ? 2 * W : W / 2;
    const int gs = C / G;
    constexpr int NU = (RS_NPIX * 8 + 255) / 256;
    constexpr int TOH = RESAMPLE == 1 ? 16 : 4, TOW = RESAMPLE == 1 ? 32 : 8;

    struct TileId { int b, cg, ty, tx; };
    auto decode = [&](int t) {                       // tx fastest, then ty, channel group, batch item
        TileId d;
        d.tx = t % tiles_x; t /= tiles_x;
        d.ty = t % tiles_y; t /= tiles_y;
        d.cg = t % ncg; d.b = t / ncg;
        return d;
    };
    uint4 rawv[NU];
    unsigned okmask = 0;
    // all global loads of a tile are issued back to back (one memory round trip), zeros outside the image
    auto load_tile = [&](const TileId& d) {
        const int c = d.cg * CG + slot * PER16;
        const bool cvalid = c < C;
        const int iy0 = d.ty * 8 - 1, ix0 = d.tx * 16 - 1;
        const long long ibase = (long long)d.b * H * W;
        okmask = 0;
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = tid + k * 256;
            const int p = u >> 3;                        // (u & 7) == slot because 256 % 8 == 0
            const int py = p / RS_IW, px = p - py * RS_IW;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = (u < RS_NPIX * 8) && cvalid && iy >= 0 && iy < H && ix >= 0 && ix < W;
            rawv[k] = make_uint4(0u, 0u, 0u, 0u);
            if (ok) {
                const long long pix = ibase + (long long)iy * W + ix;
                const T* src = (c < Ca) ? (xa + pix * Ca + c) : (xb + pix * Cb + (c - Ca));
                rawv[k] = *reinterpret_cast<const uint4*>(src);
                okmask |= 1u << k;
            }
        }
    };

    const int per = (total_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    int t = blockIdx.x * per;
    const int t_end = min(total_tiles, t + per);
    if (t >= t_end) return;
    TileId cur = decode(t);
    load_tile(cur);
    int tab_b = -1, tab_cg = -1;
    while (true) {
        // GN parameters of the tile's CG channels (y = x * scale + shift): one thread per channel does the fp64 statistics
        // math, only when the (batch item, channel group) changes; the previous tile's filter is done reading LDS
        __syncthreads();
        if (cur.b != tab_b || cur.cg != tab_cg) {
            if (tid < CG) {
                const int cc = cur.cg * CG + tid;
                float sc = 0.f, sh = 0.f;
                if (cc < C) {
                    const double n = (double)gs * H * W;
                    const int g = cc / gs;
                    const double m = stats[((long long)cur.b * G + g) * 2] / n;
                    double var = stats[((long long)cur.b * G + g) * 2 + 1] / n - m * m;
                    if (var < 0.0) var = 0.0;
                    const float pm = (float)m;
                    sc = (float)(1.0 / sqrt(var + (double)eps)) * gamma[cc];
                    sh = beta[cc] - pm * sc;
                }
                gtab[2 * tid] = sc; gtab[2 * tid + 1] = sh;
            }
            tab_b = cur.b; tab_cg = cur.cg;
            __syncthreads();
        }
        // ---- stage: raw + activated input tile -> LDS ----
        float pa[PER16], pb[PER16];
#pragma unroll
        for (int e = 0; e < PER16; ++e) { pa[e] = gtab[2 * (slot * PER16 + e)]; pb[e] = gtab[2 * (slot * PER16 + e) + 1]; }
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = tid + k * 256;
            if (u >= RS_NPIX * 8) continue;
            const int p = u >> 3;
            alignas(16) T raw[PER16];
            alignas(16) T act[PER16];
            *reinterpret_cast<uint4*>(raw) = rawv[k];
            if (okmask & (1u << k)) {
                if constexpr (sizeof(T) == 2) {
                    uint32_t aw[4];
#pragma unroll
                    for (int e = 0; e < PER16; e += 2) {
                        f32x2 y = __builtin_elementwise_fma(f32x2{to_f32(raw[e]), to_f32(raw[e + 1])}, f32x2{pa[e], pa[e + 1]}, f32x2{pb[e], pb[e + 1]});
                        if (silu) y = silu2(y);
                        aw[e / 2] = pack2(y.x, y.y, (T*)nullptr);
                    }
                    *reinterpret_cast<uint4*>(act) = make_uint4(aw[0], aw[1], aw[2], aw[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < PER16; ++e) {
                        float y = fmaf(to_f32(raw[e]), pa[e], pb[e]);
                        if (silu) y = silu_f(y);
                        from_f32(act[e], y);
                    }
                }
            } else {
                *reinterpret_cast<uint4*>(act) = make_uint4(0u, 0u, 0u, 0u);
            }
            *reinterpret_cast<uint4*>(t_raw + rs_pos<RESAMPLE>(p) * 128 + slot * 16) = rawv[k];
            *reinterpret_cast<uint4*>(t_act + rs_pos<RESAMPLE>(p) * 128 + slot * 16) = *reinterpret_cast<const uint4*>(act);
        }
        // ---- the next tile's loads fly under this tile's filter + stores ----
        const TileId me = cur;
        const int tn = t + 1;
        const bool has_next = tn < t_end;
        if (has_next) { cur = decode(tn); load_tile(cur); }
        __syncthreads();
        // ---- filter from LDS ----
        const int c = me.cg * CG + slot * PER16;
        if (c < C) {
            const long long obase = (long long)me.b * OH * OW;
            for (int u = tid; u < TOH * TOW * 8; u += 256) {
                const int q = u >> 3;
                const int oy_l = q / TOW, ox_l = q - oy_l * TOW;
                const int oy = me.ty * TOH + oy_l, ox = me.tx * TOW + ox_l;
                if (oy >= OH || ox >= OW) continue;
                float va[PER16], vr[PER16];
#pragma unroll
                for (int e = 0; e < PER16; ++e) { va[e] = 0.f; vr[e] = 0.f; }
                auto tap = [&](int py, int px, float wgt) {
                    const uint4 qa = *reinterpret_cast<const uint4*>(t_act + rs_pos<RESAMPLE>(py * RS_IW + px) * 128 + slot * 16);
                    const uint4 qr = *reinterpret_cast<const uint4*>(t_raw + rs_pos<RESAMPLE>(py * RS_IW + px) * 128 + slot * 16);
                    if constexpr (sizeof(T) == 2) {
                        // 16-bit data: one v_dot2c per channel and tap on the packed dwords (no unpack; exact, see dot2_acc)
                        uint32_t wl = tap_weight_bits(wgt, (T*)nullptr), wh = wl << 16;
                        // the weights must reach v_dot2c in REGISTERS: as a 32-bit literal of a packed-16-bit operand only the
                        // low half is honoured (measured on gfx950: the (0, w) literal acted as (0, 0))
                        keep_rw(wl); keep_rw(wh);
                        const uint32_t a[4] = {qa.x, qa.y, qa.z, qa.w}, r[4] = {qr.x, qr.y, qr.z, qr.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            va[2 * i] = dot2_acc(a[i], wl, va[2 * i], (T*)nullptr); va[2 * i + 1] = dot2_acc(a[i], wh, va[2 * i + 1], (T*)nullptr);
                            vr[2 * i] = dot2_acc(r[i], wl, vr[2 * i], (T*)nullptr); vr[2 * i + 1] = dot2_acc(r[i], wh, vr[2 * i + 1], (T*)nullptr);
                        }
                    } else {
                        alignas(16) T ra[PER16];
                        alignas(16) T rr[PER16];
                        *reinterpret_cast<uint4*>(ra) = qa;
                        *reinterpret_cast<uint4*>(rr) = qr;
#pragma unroll
                        for (int e = 0; e < PER16; ++e) { va[e] = fmaf(wgt, to_f32(ra[e]), va[e]); vr[e] = fmaf(wgt, to_f32(rr[e]), vr[e]); }
                    }
                };
                if (RESAMPLE == 1) {
                    // input pixel (oy>>1, ox>>1) sits at tile coords (+1, +1) relative to the core origin
                    const int py = (oy_l >> 1) + 1, px = (ox_l >> 1) + 1;
                    const int ny = (oy_l & 1) ? py + 1 : py - 1, nx = (ox_l & 1) ? px + 1 : px - 1;
                    tap(py, px, 0.5625f); tap(py, nx, 0.1875f); tap(ny, px, 0.1875f); tap(ny, nx, 0.0625f);
                } else {
                    // one filter row (4 taps x 2 tensors) in flight at a time: unrolling all 16 taps costs > 200 registers
                    // and the occupancy this kernel lives on
#pragma unroll 1
                    for (int i = 0; i < 4; ++i) {
                        const int py = 2 * oy_l + i, px = 2 * ox_l;
                        if (i == 0 || i == 3) { tap(py, px, 0.015625f); tap(py, px + 1, 0.046875f); tap(py, px + 2, 0.046875f); tap(py, px + 3, 0.015625f); }
                        else { tap(py, px, 0.046875f); tap(py, px + 1, 0.140625f); tap(py, px + 2, 0.140625f); tap(py, px + 3, 0.046875f); }
                    }
                }
                const long long o = (obase + (long long)oy * OW + ox) * C + c;
                if constexpr (sizeof(T) == 2) {
                    *reinterpret_cast<uint4*>(out_act + o) = make_uint4(pack2(va[0], va[1], (T*)nullptr), pack2(va[2], va[3], (T*)nullptr),
                                                                        pack2(va[4], va[5], (T*)nullptr), pack2(va[6], va[7], (T*)nullptr));
                    if (out_raw) *reinterpret_cast<uint4*>(out_raw + o) = make_uint4(pack2(vr[0], vr[1], (T*)nullptr), pack2(vr[2], vr[3], (T*)nullptr),
                                                                                     pack2(vr[4], vr[5], (T*)nullptr), pack2(vr[6], vr[7], (T*)nullptr));
                } else {
                    alignas(16) T oa[PER16];
                    alignas(16) T orr[PER16];
#pragma unroll
                    for (int e = 0; e < PER16; ++e) { from_f32(oa[e], va[e]); from_f32(orr[e], vr[e]); }
                    *reinterpret_cast<uint4*>(out_act + o) = *reinterpret_cast<const uint4*>(oa);
                    if (out_raw) *reinterpret_cast<uint4*>(out_raw + o) = *reinterpret_cast<const uint4*>(orr);
                }
            }
        }
        if (!has_next) break;
        t = tn;
    }
}

template <typename T, int RESAMPLE>
__global__ void fir_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ out,
                           int H, int W, int C, int ppb, int C8, int PL) {
    const int b = blockIdx.y, tid = threadIdx.x;
    const int oct = tid % C8, pl = tid / C8, c = oct * 8;
    const int OH = RESAMPLE == 1 ? 2 * H : H / 2, OW = RESAMPLE == 1 ? 2 * W : W / 2;
    const long long ibase = (long long)b * H * W, obase = (long long)b * OH * OW;
    const int OHW = OH * OW;
    const int p0 = blockIdx.x * ppb, p1 = min(OHW, p0 + ppb);
    GnParams gp;   // unused (raw path)
    for (int p = p0 + pl; p < p1; p += PL) {
        const int oy = p / OW, ox = p - oy * OW;
        float v[8];
        gather8<T, RESAMPLE, false>(x, C, (const T*)nullptr, 0, ibase, H, W, oy, ox, c, gp, 0, v);
        if (add != nullptr) {
            float a[8];
            load8(add + (obase + p) * C + c, a);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a[e];
        }
        store8(out + (obase + p) * C + c, v);
    }
}

template <typename T>
static int gn_stats_t(const void* xa, int Ca, const void* xb, int Cb, int B, int HW, int G, double* stats,
                      hipStream_t st) {
    const GnGeom g = gn_geom(Ca + Cb);
    const int per_thread = pixels_per_thread((long long)B * HW, g.PL, 256);
    int ppb = g.PL * per_thread;
    const int nblk = cdiv(HW, ppb);
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nblk, B), dim3(g.NT), 0, st, (const T*)xa, Ca, (const T*)xb, Cb,
                       HW, G, stats, ppb, g.C8, g.PL);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

template <typename T, int R>
static int gn_apply_t(const void* xa, int Ca, const void* xb, int Cb, int B, int H, int W, int G,
                      const double* stats, const float* gamma, const float* beta, float eps, int silu,
                      void* out_act, void* out_raw, hipStream_t st) {
    const GnGeom g = gn_geom(Ca + Cb);
    if (R != 0) {
        constexpr int CG = 8 * Elem<T>::PER16;
        const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, 8), ncg = cdiv(Ca + Cb, CG);
        const long long total = (long long)tiles_x * tiles_y * ncg * B;
        STORM_CHECK(total > 0 && total < (1LL << 31), "storm_gn_apply: %lld tiles out of range", total);
        // persistent: 3 workgroups fit a CU (46 KiB of LDS each); a few per slot so that the tail stays short
        const long long cap = switches().resample_wgs > 0 ? switches().resample_wgs : 256LL * 3 * 4;   // (test hook: cap the persistent grid)
        const long long grid = total < cap ? total : cap;
        hipLaunchKernelGGL((gn_apply_resample_kernel<T, R == 0 ? 1 : R>), dim3((unsigned)grid), dim3(256), 0, st,
                           (const T*)xa, Ca, (const T*)xb, Cb, H, W, G, stats, gamma, beta, eps, silu, (T*)out_act,
                           (T*)out_raw, tiles_x, tiles_y, ncg, (int)total);
        STORM_LAUNCH_CHECK();
        return STORM_OK;
    }
    const int OHW = H * W;
    const int per_thread = pixels_per_thread((long long)B * OHW, g.PL, 32);
    int ppb = g.PL * per_thread;
    const int nblk = cdiv(OHW, ppb);
    hipLaunchKernelGGL((gn_apply_kernel<T, 0>), dim3(nblk, B), dim3(g.NT), 0, st, (const T*)xa, Ca, (const T*)xb,
                       Cb, H, W, G, stats, gamma, beta, eps, silu, (T*)out_act, (T*)out_raw, ppb, g.C8, g.PL);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

template <typename T, int R>
static int fir_t(const void* x, const void* add, void* out, int B, int H, int W, int C, hipStream_t st) {
    const GnGeom g = gn_geom(C);
    const int OHW = R == 1 ? 4 * H * W : H * W / 4;
    int ppb = g.PL * pixels_per_thread((long long)B * OHW, g.PL, 32);
    const int nblk = cdiv(OHW, ppb);
    hipLaunchKernelGGL((fir_kernel<T, R>), dim3(nblk, B), dim3(g.NT), 0, st, (const T*)x, (const T*)add, (T*)out,
                       H, W, C, ppb, g.C8, g.PL);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

}  // namespace storm

using namespace storm;

static int check_c(const char* who, int Ca, int Cb, int groups) {
    const int C = Ca + Cb;
    STORM_CHECK(Ca > 0 && Ca % 8 == 0 && Cb >= 0 && Cb % 8 == 0, "%s: channels must be multiples of 8 (Ca=%d Cb=%d)", who, Ca, Cb);
    STORM_CHECK(C <= GN_MAX_C, "%s: C=%d > %d unsupported", who, C, GN_MAX_C);
    STORM_CHECK(groups > 0 && C % groups == 0, "%s: C=%d not divisible by groups=%d", who, C, groups);
    return STORM_OK;
}

extern "C" int storm_gn_stats(const void* xa, int Ca, const void* xb, int Cb, int B, int HW, int groups,
                              double* stats, int dtype, storm_stream_t s) {
    if (int e = check_c("storm_gn_stats", Ca, Cb, groups)) return e;
    STORM_CHECK(xa && stats && B > 0 && HW > 0, "storm_gn_stats: bad arguments");
    STORM_CHECK((Cb == 0) == (xb == nullptr), "storm_gn_stats: xb / Cb mismatch");
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) return gn_stats_t<bf16_t>(xa, Ca, xb, Cb, B, HW, groups, stats, st);
    if (dtype == STORM_F16) return gn_stats_t<half_t>(xa, Ca, xb, Cb, B, HW, groups, stats, st);
    if (dtype == STORM_F32) return gn_stats_t<float>(xa, Ca, xb, Cb, B, HW, groups, stats, st);
    STORM_CHECK(false, "storm_gn_stats: dtype %d", dtype);
}

extern "C" int storm_gn_finalize(const float* part_a, int Ca, int tiles_a, const float* part_b, int Cb, int tiles_b,
                                 int B, int groups, double* stats, storm_stream_t s) {
    if (int e = check_c("storm_gn_finalize", Ca, Cb, groups)) return e;
    STORM_CHECK(part_a && stats && B > 0 && tiles_a > 0, "storm_gn_finalize: bad arguments");
    STORM_CHECK((Cb == 0) == (part_b == nullptr), "storm_gn_finalize: part_b / Cb mismatch");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, (hipStream_t)s, part_a, Ca, tiles_a, part_b, Cb,
                       tiles_b, groups, stats, 0LL, (const float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_gn_finalize_ss(const float* part_a, int Ca, int tiles_a, const float* part_b, int Cb, int tiles_b,
                                    int B, int groups, long long count, const float* gamma, const float* beta, float eps,
                                    double* stats, float* ss, storm_stream_t s) {
    if (int e = check_c("storm_gn_finalize_ss", Ca, Cb, groups)) return e;
    STORM_CHECK(part_a && stats && ss && gamma && beta && B > 0 && tiles_a > 0 && count > 0, "storm_gn_finalize_ss: bad arguments");
    STORM_CHECK((Cb == 0) == (part_b == nullptr), "storm_gn_finalize_ss: part_b / Cb mismatch");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(256), 0, (hipStream_t)s, part_a, Ca, tiles_a, part_b, Cb,
                       tiles_b, groups, stats, count, gamma, beta, eps, ss);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

extern "C" int storm_gn_apply(const void* xa, int Ca, const void* xb, int Cb, int B, int H, int W, int groups,
                              const double* stats, const float* gamma, const float* beta, float eps, int silu,
                              int resample, void* out_act, void* out_raw, int dtype, storm_stream_t s) {
    if (int e = check_c("storm_gn_apply", Ca, Cb, groups)) return e;
    STORM_CHECK(xa && stats && gamma && beta && out_act, "storm_gn_apply: null pointer");
    STORM_CHECK((Cb == 0) == (xb == nullptr), "storm_gn_apply: xb / Cb mismatch");
    STORM_CHECK(resample >= 0 && resample <= 2, "storm_gn_apply: resample=%d", resample);
    STORM_CHECK(resample != 2 || (H % 2 == 0 && W % 2 == 0), "storm_gn_apply: FIR down needs even H, W");
    hipStream_t st = (hipStream_t)s;
#define STORM_GN_DISPATCH(T)                                                                                   \
    switch (resample) {                                                                                        \
        case 0: return gn_apply_t<T, 0>(xa, Ca, xb, Cb, B, H, W, groups, stats, gamma, beta, eps, silu, out_act, out_raw, st); \
        case 1: return gn_apply_t<T, 1>(xa, Ca, xb, Cb, B, H, W, groups, stats, gamma, beta, eps, silu, out_act, out_raw, st); \
        default: return gn_apply_t<T, 2>(xa, Ca, xb, Cb, B, H, W, groups, stats, gamma, beta, eps, silu, out_act, out_raw, st); \
    }
    if (dtype == STORM_BF16) { STORM_GN_DISPATCH(bf16_t) }
    if (dtype == STORM_F16) { STORM_GN_DISPATCH(half_t) }
    if (dtype == STORM_F32) { STORM_GN_DISPATCH(float) }
#undef STORM_GN_DISPATCH
    STORM_CHECK(false, "storm_gn_apply: dtype %d", dtype);
}

extern "C" int storm_fir_up2(const void* x, const void* add, void* out, int B, int H, int W, int C, int dtype,
                             storm_stream_t s) {
    STORM_CHECK(x && out && C > 0 && C % 8 == 0 && C <= GN_MAX_C, "storm_fir_up2: bad arguments (C=%d)", C);
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) return fir_t<bf16_t, 1>(x, add, out, B, H, W, C, st);
    if (dtype == STORM_F16) return fir_t<half_t, 1>(x, add, out, B, H, W, C, st);
    if (dtype == STORM_F32) return fir_t<float, 1>(x, add, out, B, H, W, C, st);
    STORM_CHECK(false, "storm_fir_up2: dtype %d", dtype);
}

extern "C" int storm_fir_down2(const void* x, void* out, int B, int H, int W, int C, int dtype, storm_stream_t s) {
    STORM_CHECK(x && out && C > 0 && C % 8 == 0 && C <= GN_MAX_C, "storm_fir_down2: bad arguments (C=%d)", C);
    STORM_CHECK(H % 2 == 0 && W % 2 == 0, "storm_fir_down2: needs even H, W");
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) return fir_t<bf16_t, 2>(x, nullptr, out, B, H, W, C, st);
    if (dtype == STORM_F16) return fir_t<half_t, 2>(x, nullptr, out, B, H, W, C, st);
    if (dtype == STORM_F32) return fir_t<float, 2>(x, nullptr, out, B, H, W, C, st);
    STORM_CHECK(false, "storm_fir_down2: dtype %d", dtype);
}
