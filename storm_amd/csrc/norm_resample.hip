// GroupNorm (+SiLU) and FIR x2 resampling on NHWC activations — the HBM-bound family.
//
// Replaces nn.GroupNorm(eps=1e-6)+nn.SiLU and upsample_2d/downsample_2d (upfirdn2d CUDA
// kernel modes 3/5) of the reference; see include/storm_hip.h for the file:line map.
// Design: every thread owns a fixed octet of channels (16 B bf16 / 32 B fp32 per access,
// a pixel's channels are contiguous so a wave reads whole 128-B lines), statistics are
// reduced per thread in fp32 over <=256 pixels, then in fp64 across threads / workgroups
// (one fp64 atomicAdd per (batch, group) per workgroup).  The resampling variants fuse
// GN-apply + SiLU + FIR of BOTH the activated and the raw tensor (BigGAN block) in one pass.
#include <algorithm>
#include <cstdio>
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
__device__ __forceinline__
void gn_finalize_body(const float* __restrict__ pa, int Ca, int tiles_a, const float* __restrict__ pb,
                      int Cb, int tiles_b, int G, double* __restrict__ stats, long long count,
                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                      float* __restrict__ ss, const int g, const int b) {
    // one workgroup (4 waves) per (group, batch item): the group's gs channels of a tile are contiguous (gs float2), so a
    // thread walks (tile, channel) pairs with the channel fastest; fp64 partial sums, fixed reduction order (deterministic)
    __shared__ double red[2][4];
    const int tid = threadIdx.x;
    const int C = Ca + Cb, gs = C / G;
    double s0 = 0.0, s1 = 0.0;
    const int c_lo = g * gs, c_hi = c_lo + gs;
    // the affine parameters of this group's channels are fetched NOW, beside the partials (round 5: fetched where they are used - behind the
    // reduction and its barrier - they were one more exposed round trip at the end of each of the 45 launches of an evaluation)
    float gam[4], bet[4];                                    // (gs <= 1024 channels per group: four per thread at most)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = tid + 256 * u;
        const bool live = ss != nullptr && k < gs;
        gam[u] = live ? gamma[c_lo + k] : 0.f;
        bet[u] = live ? beta[c_lo + k] : 0.f;
    }
    for (int part = 0; part < 2; ++part) {
        const float* p = part == 0 ? pa : pb;
        const int Cs = part == 0 ? Ca : Cb, nt = part == 0 ? tiles_a : tiles_b, off = part == 0 ? 0 : Ca;
        const int lo = max(c_lo, off) - off, hi = min(c_hi, off + Cs) - off;       // this group's channels inside the part
        const int w = hi - lo;
        if (p == nullptr || w <= 0) continue;
        const float2* q = reinterpret_cast<const float2*>(p) + (long long)b * nt * Cs + lo;
        // (tile, channel) pairs tid, tid + 256, ...: ONE division per thread, then (t, k) advance by (256 / w, 256 % w) with a carry
        const int dt = 256 / w, dk = 256 - dt * w;
        int t = tid / w, k = tid - t * w;
        const int n = nt * w;
        for (int i = tid; i < n; i += 256 * 8) {          // eight loads in flight per thread (one at a time, the loop was a chain of round trips)
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u] = (i + 256 * u < n) ? q[(long long)t * Cs + k] : make_float2(0.f, 0.f);
                t += dt; k += dk;
                if (k >= w) { k -= w; ++t; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += (double)v[u].x; s1 += (double)v[u].y; }
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
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = tid + 256 * u;
            if (k < gs) {
                const int c = g * gs + k;
                const float sc = rstd * gam[u];
                float* const o = ss + ((long long)b * C + (c & ~7)) * 2 + (c & 7);     // [C/8][2][8]: 8 scales, then 8 shifts
                o[0] = sc;
                o[8] = bet[u] - (float)m * sc;
            }
        }
    }
}

__global__ __launch_bounds__(256)
void gn_finalize_kernel(const float* __restrict__ pa, int Ca, int tiles_a, const float* __restrict__ pb,
                        int Cb, int tiles_b, int G, double* __restrict__ stats, long long count,
                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                        float* __restrict__ ss) {
    gn_finalize_body(pa, Ca, tiles_a, pb, Cb, tiles_b, G, stats, count, gamma, beta, eps, ss, blockIdx.x, blockIdx.y);
}
// The same for the items of SEVERAL problems in one launch (grouped evaluation of a ragged stream's micro-batches, common.h): item =
// (problem, batch item) from a host-built list, the problem's arguments from a device table.  An item's sums are the sums of its own launch.
__global__ __launch_bounds__(256)
void gn_finalize_group_kernel(const GnFinProblem* __restrict__ tab, const GnFinItem* __restrict__ items, int G) {
    const GnFinItem it = items[blockIdx.y];
    const GnFinProblem& q = tab[it.problem];
    gn_finalize_body(q.pa, q.Ca, q.tiles_a, q.pb, q.Cb, q.tiles_b, G, q.stats, q.count, q.gamma, q.beta, q.eps, q.ss, blockIdx.x, it.b);
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

// Fused GN-apply + SiLU + FIR x2 DOWN of both the activated and the raw tensor, register sliding window (no LDS tile, no
// barriers in the loop).  The separable filter k = [1,3,3,1] / 8 per axis, out[oy][ox] = sum_i k_i (sum_j k_j x[2 oy - 1 + i][2 ox - 1 + j]):
// a thread owns one 16-byte channel slot of one output column and walks DOWN a strip of output rows; per output row it loads
// the four taps of two new input rows (16 B each: the loaded slot feeds both tensors), normalises + activates them once,
// filters each row horizontally with v_dot2c on the packed 16-bit data, and combines with the carry of the two rows it shares
// with the previous output row (out[oy] = carry + k2 h[2 oy + 1] + k3 h[2 oy + 2]; carry' = k0 h[2 oy + 1] + k1 h[2 oy + 2]).
// Workgroup = 8 slots (128 B per pixel: coalesced loads and stores) x 32 output columns; ~80 registers, so 5-6 waves per SIMD
// keep 8 KiB of loads in flight each - the LDS-tiled predecessor (10 x 18-pixel tiles staged once, 3 workgroups per CU, two
// barriers per 32 output pixels; git show 9dfa6dc:storm_amd/csrc/norm_resample.hip) ran at 1.7 TB/s with its waves parked half
// of the time (profiles/r03a_pmc_summary.txt); this one measures 2.8 TB/s.
constexpr int DN_ROWS = 16;                         // output rows per strip
// NS = 16-byte slots of a pixel per workgroup (256 / NS output columns).  Round 3 used 8 everywhere (one 128-byte line per pixel and
// wave access); with NS = 32 a wave reads / writes 512 contiguous bytes per pixel - whole DRAM bursts of one page instead of 128-byte
// pieces 1 KiB apart that other workgroups complete at another time (round 4: the write stream of the up-sampling kernel was at 4.1 TB/s
// where a plain fill reaches 6.9, profiles/r04f_hbm_probe.txt).
// SILU: the activation as a compile-time choice (as a run-time flag it is if-converted: both results computed, a select per value)
template <typename T, bool SILU, int NS>
__device__ __forceinline__ void gn_apply_down_body(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb,
                          int H, int W, int G, const double* __restrict__ stats,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                          T* __restrict__ out_act, T* __restrict__ out_raw, int ncg, int nstrips, int nt_stores, const int by, const int bx) {
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int CG = NS * PER16;                  // channels per workgroup
    constexpr int DN_COLS = 256 / NS;               // output columns per workgroup
    __shared__ float gtab[2 * CG];
    const int C = Ca + Cb, tid = threadIdx.x;
    const int slot = tid % NS, col = tid / NS;
    const int OH = H / 2, OW = W / 2;
    int t = by;                             // (channel group, strip, batch item)
    const int cg = t % ncg; t /= ncg;
    const int strip = t % nstrips, b = t / nstrips;
    const int gs = C / G;
    if (tid < CG) {                                 // (scale, shift) of this workgroup's channels: y = x * sc + sh
        const int cc = cg * CG + tid;
        float sc = 0.f, sh = 0.f;
        if (cc < C) {
            const double n = (double)gs * H * W;
            const int g = cc / gs;
            const double m = stats[((long long)b * G + g) * 2] / n;
            double var = stats[((long long)b * G + g) * 2 + 1] / n - m * m;
            if (var < 0.0) var = 0.0;
            const float pm = (float)m;
            sc = (float)(1.0 / sqrt(var + (double)eps)) * gamma[cc];
            sh = beta[cc] - pm * sc;
        }
        gtab[2 * tid] = sc; gtab[2 * tid + 1] = sh;
    }
    __syncthreads();
    const int c = cg * CG + slot * PER16;
    const int ox = bx * DN_COLS + col;
    if (c >= C || ox >= OW) return;
    float pa[PER16], pb[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) { pa[e] = gtab[2 * (slot * PER16 + e)]; pb[e] = gtab[2 * (slot * PER16 + e) + 1]; }
    const T* const src = (c < Ca) ? xa + c : xb + (c - Ca);
    const int cs = (c < Ca) ? Ca : Cb;              // channel stride of the source this slot lives in
    const long long ibase = (long long)b * H * W;
    const long long obase = (long long)b * OH * OW;
    const int ix0 = 2 * ox - 1;
    const bool x_in[4] = {ix0 >= 0, true, ix0 + 2 < W, ix0 + 3 < W};     // (ix0 + 1 = 2 ox < W always)

    // the horizontally filtered input row iy, activated (hA) and raw (hR): zero rows / taps outside the image
    auto hrow = [&](int iy, float (&hA)[PER16], float (&hR)[PER16]) {
#pragma unroll
        for (int e = 0; e < PER16; ++e) { hA[e] = 0.f; hR[e] = 0.f; }
        if (iy < 0 || iy >= H) return;
        const T* const row = src + (ibase + (long long)iy * W) * cs;
        uint4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            q[j] = x_in[j] ? *reinterpret_cast<const uint4*>(row + (long long)(ix0 + j) * cs) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wgt = (j == 0 || j == 3) ? 0.125f : 0.375f;
            alignas(16) T raw[PER16];
            *reinterpret_cast<uint4*>(raw) = q[j];
            if constexpr (sizeof(T) == 2) {
                uint32_t aw[4] = {0u, 0u, 0u, 0u};
                if (x_in[j]) {
#pragma unroll
                    for (int e = 0; e < PER16; e += 2) {
                        f32x2 y = __builtin_elementwise_fma(f32x2{to_f32(raw[e]), to_f32(raw[e + 1])}, f32x2{pa[e], pa[e + 1]}, f32x2{pb[e], pb[e + 1]});
                        if (SILU) y = silu2(y);
                        aw[e / 2] = pack2(y.x, y.y, (T*)nullptr);
                    }
                }
                uint32_t wl = tap_weight_bits(wgt, (T*)nullptr), wh = wl << 16;
                keep_rw(wl); keep_rw(wh);            // (packed 16-bit operands of v_dot2c must come from registers: see above)
                const uint32_t r[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hA[2 * i] = dot2_acc(aw[i], wl, hA[2 * i], (T*)nullptr); hA[2 * i + 1] = dot2_acc(aw[i], wh, hA[2 * i + 1], (T*)nullptr);
                    hR[2 * i] = dot2_acc(r[i], wl, hR[2 * i], (T*)nullptr); hR[2 * i + 1] = dot2_acc(r[i], wh, hR[2 * i + 1], (T*)nullptr);
                }
            } else {
#pragma unroll
                for (int e = 0; e < PER16; ++e) {
                    const float xr = to_f32(raw[e]);
                    float y = 0.f;
                    if (x_in[j]) {
                        y = fmaf(xr, pa[e], pb[e]);
                        if (SILU) y = silu_f(y);
                        T ya; from_f32(ya, y); y = to_f32(ya);      // (the activated tensor is rounded to T before it is filtered)
                    }
                    hA[e] = fmaf(wgt, y, hA[e]); hR[e] = fmaf(wgt, xr, hR[e]);
                }
            }
        }
    };
    const int dn_rows = (OH + nstrips - 1) / nstrips;
    const int oy0 = strip * dn_rows, oy1 = min(OH, oy0 + dn_rows);
    float cA[PER16], cR[PER16];                     // carry: k0 h[2 oy - 1] + k1 h[2 oy]
    {
        float h0A[PER16], h0R[PER16], h1A[PER16], h1R[PER16];
        hrow(2 * oy0 - 1, h0A, h0R);
        hrow(2 * oy0, h1A, h1R);
#pragma unroll
        for (int e = 0; e < PER16; ++e) { cA[e] = fmaf(0.375f, h1A[e], 0.125f * h0A[e]); cR[e] = fmaf(0.375f, h1R[e], 0.125f * h0R[e]); }
    }
    for (int oy = oy0; oy < oy1; ++oy) {
        float h2A[PER16], h2R[PER16], h3A[PER16], h3R[PER16];
        hrow(2 * oy + 1, h2A, h2R);
        hrow(2 * oy + 2, h3A, h3R);
        float va[PER16], vr[PER16];
#pragma unroll
        for (int e = 0; e < PER16; ++e) {
            va[e] = fmaf(0.125f, h3A[e], fmaf(0.375f, h2A[e], cA[e]));
            vr[e] = fmaf(0.125f, h3R[e], fmaf(0.375f, h2R[e], cR[e]));
            cA[e] = fmaf(0.375f, h3A[e], 0.125f * h2A[e]);
            cR[e] = fmaf(0.375f, h3R[e], 0.125f * h2R[e]);
        }
        const long long o = (obase + (long long)oy * OW + ox) * C + c;
        if constexpr (sizeof(T) == 2) {
            const uint4 qa = make_uint4(pack2(va[0], va[1], (T*)nullptr), pack2(va[2], va[3], (T*)nullptr),
                                        pack2(va[4], va[5], (T*)nullptr), pack2(va[6], va[7], (T*)nullptr));
            const uint4 qr = make_uint4(pack2(vr[0], vr[1], (T*)nullptr), pack2(vr[2], vr[3], (T*)nullptr),
                                        pack2(vr[4], vr[5], (T*)nullptr), pack2(vr[6], vr[7], (T*)nullptr));
            if (nt_stores) { store16_nt(out_act + o, qa); if (out_raw) store16_nt(out_raw + o, qr); }
            else { *reinterpret_cast<uint4*>(out_act + o) = qa; if (out_raw) *reinterpret_cast<uint4*>(out_raw + o) = qr; }
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
template <typename T, bool SILU, int NS>
__global__ __launch_bounds__(256)
void gn_apply_down_kernel(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb, int H, int W, int G, const double* __restrict__ stats,
          const float* __restrict__ gamma, const float* __restrict__ beta, float eps, T* __restrict__ out_act, T* __restrict__ out_raw,
          int ncg, int nstrips, int nt_stores) {
    gn_apply_down_body<T, SILU, NS>(xa, Ca, xb, Cb, H, W, G, stats, gamma, beta, eps, out_act, out_raw, ncg, nstrips, nt_stores, blockIdx.y, blockIdx.x);
}
// the strips of SEVERAL problems in one launch (grouped evaluation of a ragged stream's micro-batches, common.h): blockIdx.y = an item
// (problem, the y index of the problem's own launch) of a host-built list; a strip is computed by the code of its own launch
template <typename T, bool SILU, int NS>
__global__ __launch_bounds__(256)
void gn_apply_down_group_kernel(const GnApplyProblem* __restrict__ tab, const GnFinItem* __restrict__ items, int Ca, int Cb, int G,
           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int nt_stores) {
    const GnFinItem it = items[blockIdx.y];
    const GnApplyProblem& q = tab[it.problem];
    if ((int)blockIdx.x >= q.cols) return;
    gn_apply_down_body<T, SILU, NS>(static_cast<const T*>(q.xa), Ca, static_cast<const T*>(q.xb), Cb, q.H, q.W, G, q.stats, gamma, beta, eps,
                      static_cast<T*>(q.out_act), static_cast<T*>(q.out_raw), q.ncg, q.nstrips, nt_stores, it.b, blockIdx.x);
}

// Round 5: the down-sampling kernel with the ACTIVATION shared between neighbouring threads.  Timed without its SiLU the kernel above runs at the
// memory time (169 of 280 us at the bench batch's level-0 launch, tools/debug/probe_gn_silu.py): it is bound by the activation - two quarter-rate
// transcendentals per element - and evaluates it TWICE per input pixel, because the four taps of neighbouring output columns overlap by two.  Here a
// thread still loads its four taps (the raw tensor's filter needs them) but normalises + activates only the two centre columns it owns (2 ox,
// 2 ox + 1), hands them to its neighbours through LDS and takes its outer taps from theirs; the first / last column of a workgroup activates its
// outer tap itself (one masked pass in the two edge waves).  5 activation passes per output row and wave on average instead of 8; one barrier per
// output row (double-buffered exchange); every activated value is computed by the same arithmetic from the same inputs whoever computes it, and
// each output keeps its own tap order: the same bits (test_groupnorm_fir_fused).  Wider register windows instead (two output columns per thread)
// paid the saved instructions back in occupancy: tools/experiments/gn_down_two_columns.patch.
template <typename T, bool SILU, int NS>
__device__ __forceinline__ void gn_apply_down_share_body(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb,
                                int H, int W, int G, const double* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                T* __restrict__ out_act, T* __restrict__ out_raw, int ncg, int nstrips, int nt_stores, const int by, const int bx) {
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int CG = NS * PER16;                  // channels per workgroup
    constexpr int DN_COLS = 256 / NS;               // output columns per workgroup
    constexpr int LC = 2 * DN_COLS + 2;             // local input columns 0 .. LC - 1 = taps 2 col + j (0 and LC - 1: the edge threads' own)
    __shared__ float gtab[2 * CG];
    __shared__ uint4 xact[2][2][LC][NS];            // [exchange buffer][row of the pair][local input column][slot]: activated, in T
    const int C = Ca + Cb, tid = threadIdx.x;
    const int slot = tid % NS, col = tid / NS;
    const int OH = H / 2, OW = W / 2;
    int t = by;                             // (channel group, strip, batch item)
    const int cg = t % ncg; t /= ncg;
    const int strip = t % nstrips, b = t / nstrips;
    const int gs = C / G;
    if (tid < CG) {                                 // (scale, shift) of this workgroup's channels: y = x * sc + sh
        const int cc = cg * CG + tid;
        float sc = 0.f, sh = 0.f;
        if (cc < C) {
            const double n = (double)gs * H * W;
            const int g = cc / gs;
            const double m = stats[((long long)b * G + g) * 2] / n;
            double var = stats[((long long)b * G + g) * 2 + 1] / n - m * m;
            if (var < 0.0) var = 0.0;
            const float pm = (float)m;
            sc = (float)(1.0 / sqrt(var + (double)eps)) * gamma[cc];
            sh = beta[cc] - pm * sc;
        }
        gtab[2 * tid] = sc; gtab[2 * tid + 1] = sh;
    }
    __syncthreads();
    const int c = cg * CG + slot * PER16;
    const int ox = bx * DN_COLS + col;
    const bool chan = c < C;                        // (a slot past a ragged last channel group loads nothing - but every thread takes part in the barriers)
    const bool live = chan && ox < OW;              // (a column past the image may still own input columns a live neighbour filters)
    float pa[PER16], pb[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) { pa[e] = gtab[2 * (slot * PER16 + e)]; pb[e] = gtab[2 * (slot * PER16 + e) + 1]; }
    const int cl = chan ? c : 0;
    const T* const src = (cl < Ca) ? xa + cl : xb + (cl - Ca);
    const int cs = (cl < Ca) ? Ca : Cb;             // channel stride of the source this slot lives in
    const long long ibase = (long long)b * H * W;
    const long long obase = (long long)b * OH * OW;
    const int ix0 = 2 * ox - 1;
    bool x_in[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x_in[j] = chan && ix0 + j >= 0 && ix0 + j < W;
    const bool first = col == 0, last = col == DN_COLS - 1;

    auto activate = [&](const uint4 qv, bool ok) -> uint4 {          // GroupNorm affine (+ SiLU) of one 16-byte slot, rounded to T
        uint4 r = make_uint4(0u, 0u, 0u, 0u);
        if (!ok) return r;
        alignas(16) T raw[PER16];
        *reinterpret_cast<uint4*>(raw) = qv;
        if constexpr (sizeof(T) == 2) {
            uint32_t aw[4];
#pragma unroll
            for (int e = 0; e < PER16; e += 2) {
                f32x2 y = __builtin_elementwise_fma(f32x2{to_f32(raw[e]), to_f32(raw[e + 1])}, f32x2{pa[e], pa[e + 1]}, f32x2{pb[e], pb[e + 1]});
                if (SILU) y = silu2(y);
                aw[e / 2] = pack2(y.x, y.y, (T*)nullptr);
            }
            r = make_uint4(aw[0], aw[1], aw[2], aw[3]);
        } else {
            alignas(16) T ya[PER16];
#pragma unroll
            for (int e = 0; e < PER16; ++e) {
                float y = fmaf(to_f32(raw[e]), pa[e], pb[e]);
                if (SILU) y = silu_f(y);
                from_f32(ya[e], y);                 // (the activated tensor is rounded to T before it is filtered)
            }
            r = *reinterpret_cast<const uint4*>(ya);
        }
        return r;
    };
    // raw taps of input row iy (zeros outside the image)
    auto load_row = [&](int iy, uint4 (&q)[4]) {
        const bool rowok = iy >= 0 && iy < H;
        const T* const row = src + (ibase + (long long)(rowok ? iy : 0) * W) * cs;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            q[j] = rowok && x_in[j] ? *reinterpret_cast<const uint4*>(row + (long long)(ix0 + j) * cs) : make_uint4(0u, 0u, 0u, 0u);
    };
    // own activations of a row: the centre taps (to LDS for the neighbours) and, in the first / last column, the outer tap
    auto own_acts = [&](int iy, const uint4 (&q)[4], uint4 (&a)[4], int buf, int r) {
        const bool rowok = iy >= 0 && iy < H;       // (uniform in the workgroup)
        a[0] = a[1] = a[2] = a[3] = make_uint4(0u, 0u, 0u, 0u);
        if (!rowok) return;
        a[1] = activate(q[1], x_in[1]);
        a[2] = activate(q[2], x_in[2]);
        xact[buf][r][2 * col + 1][slot] = a[1];
        xact[buf][r][2 * col + 2][slot] = a[2];
        if (first || last) {
            const uint4 e = activate(first ? q[0] : q[3], first ? x_in[0] : x_in[3]);
            if (first) a[0] = e; else a[3] = e;
            if (first && last) a[3] = activate(q[3], x_in[3]);       // (one column per workgroup: never with NS <= 32, kept for completeness)
        }
    };
    // the neighbours' activations of the outer taps (after the barrier)
    auto nb_acts = [&](int iy, uint4 (&a)[4], int buf, int r) {
        if (iy < 0 || iy >= H) return;
        if (!first) a[0] = xact[buf][r][2 * col][slot];
        if (!last) a[3] = xact[buf][r][2 * col + 3][slot];
    };
    // horizontal filter of one row: activated (hA) and raw (hR), taps in the order 0 .. 3
    auto hfilter = [&](const uint4 (&a)[4], const uint4 (&q)[4], float (&hA)[PER16], float (&hR)[PER16]) {
#pragma unroll
        for (int e = 0; e < PER16; ++e) { hA[e] = 0.f; hR[e] = 0.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float wgt = (j == 0 || j == 3) ? 0.125f : 0.375f;
            if constexpr (sizeof(T) == 2) {
                uint32_t wl = tap_weight_bits(wgt, (T*)nullptr), wh = wl << 16;
                keep_rw(wl); keep_rw(wh);            // (packed 16-bit operands of v_dot2c must come from registers)
                const uint32_t aw[4] = {a[j].x, a[j].y, a[j].z, a[j].w}, rw[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hA[2 * i] = dot2_acc(aw[i], wl, hA[2 * i], (T*)nullptr); hA[2 * i + 1] = dot2_acc(aw[i], wh, hA[2 * i + 1], (T*)nullptr);
                    hR[2 * i] = dot2_acc(rw[i], wl, hR[2 * i], (T*)nullptr); hR[2 * i + 1] = dot2_acc(rw[i], wh, hR[2 * i + 1], (T*)nullptr);
                }
            } else {
                alignas(16) T ya[PER16];
                alignas(16) T xr[PER16];
                *reinterpret_cast<uint4*>(ya) = a[j];
                *reinterpret_cast<uint4*>(xr) = q[j];
#pragma unroll
                for (int e = 0; e < PER16; ++e) { hA[e] = fmaf(wgt, to_f32(ya[e]), hA[e]); hR[e] = fmaf(wgt, to_f32(xr[e]), hR[e]); }
            }
        }
    };
    const int dn_rows = (OH + nstrips - 1) / nstrips;
    const int oy0 = strip * dn_rows, oy1 = min(OH, oy0 + dn_rows);
    float cA[PER16], cR[PER16];                     // carry: k0 h[2 oy - 1] + k1 h[2 oy]
    int buf = 1;
    {
        uint4 q0[4], q1[4], a0[4], a1[4];
        load_row(2 * oy0 - 1, q0);
        load_row(2 * oy0, q1);
        own_acts(2 * oy0 - 1, q0, a0, buf, 0);
        own_acts(2 * oy0, q1, a1, buf, 1);
        __syncthreads();
        nb_acts(2 * oy0 - 1, a0, buf, 0);
        nb_acts(2 * oy0, a1, buf, 1);
        float h0A[PER16], h0R[PER16], h1A[PER16], h1R[PER16];
        hfilter(a0, q0, h0A, h0R);
        hfilter(a1, q1, h1A, h1R);
#pragma unroll
        for (int e = 0; e < PER16; ++e) { cA[e] = fmaf(0.375f, h1A[e], 0.125f * h0A[e]); cR[e] = fmaf(0.375f, h1R[e], 0.125f * h0R[e]); }
    }
    for (int oy = oy0; oy < oy1; ++oy) {
        buf ^= 1;
        uint4 q2[4], q3[4], a2[4], a3[4];
        load_row(2 * oy + 1, q2);
        load_row(2 * oy + 2, q3);
        own_acts(2 * oy + 1, q2, a2, buf, 0);
        own_acts(2 * oy + 2, q3, a3, buf, 1);
        __syncthreads();
        nb_acts(2 * oy + 1, a2, buf, 0);
        nb_acts(2 * oy + 2, a3, buf, 1);
        float h2A[PER16], h2R[PER16], h3A[PER16], h3R[PER16];
        hfilter(a2, q2, h2A, h2R);
        hfilter(a3, q3, h3A, h3R);
        float va[PER16], vr[PER16];
#pragma unroll
        for (int e = 0; e < PER16; ++e) {
            va[e] = fmaf(0.125f, h3A[e], fmaf(0.375f, h2A[e], cA[e]));
            vr[e] = fmaf(0.125f, h3R[e], fmaf(0.375f, h2R[e], cR[e]));
            cA[e] = fmaf(0.375f, h3A[e], 0.125f * h2A[e]);
            cR[e] = fmaf(0.375f, h3R[e], 0.125f * h2R[e]);
        }
        if (!live) continue;
        const long long o = (obase + (long long)oy * OW + ox) * C + c;
        if constexpr (sizeof(T) == 2) {
            const uint4 qa = make_uint4(pack2(va[0], va[1], (T*)nullptr), pack2(va[2], va[3], (T*)nullptr),
                                        pack2(va[4], va[5], (T*)nullptr), pack2(va[6], va[7], (T*)nullptr));
            const uint4 qr = make_uint4(pack2(vr[0], vr[1], (T*)nullptr), pack2(vr[2], vr[3], (T*)nullptr),
                                        pack2(vr[4], vr[5], (T*)nullptr), pack2(vr[6], vr[7], (T*)nullptr));
            if (nt_stores) { store16_nt(out_act + o, qa); if (out_raw) store16_nt(out_raw + o, qr); }
            else { *reinterpret_cast<uint4*>(out_act + o) = qa; if (out_raw) *reinterpret_cast<uint4*>(out_raw + o) = qr; }
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
template <typename T, bool SILU, int NS>
__global__ __launch_bounds__(256)
void gn_apply_down_share_kernel(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb, int H, int W, int G, const double* __restrict__ stats,
          const float* __restrict__ gamma, const float* __restrict__ beta, float eps, T* __restrict__ out_act, T* __restrict__ out_raw,
          int ncg, int nstrips, int nt_stores) {
    gn_apply_down_share_body<T, SILU, NS>(xa, Ca, xb, Cb, H, W, G, stats, gamma, beta, eps, out_act, out_raw, ncg, nstrips, nt_stores, blockIdx.y, blockIdx.x);
}
// the strips of SEVERAL problems in one launch (grouped evaluation of a ragged stream's micro-batches, common.h): blockIdx.y = an item
// (problem, the y index of the problem's own launch) of a host-built list; a strip is computed by the code of its own launch
template <typename T, bool SILU, int NS>
__global__ __launch_bounds__(256)
void gn_apply_down_share_group_kernel(const GnApplyProblem* __restrict__ tab, const GnFinItem* __restrict__ items, int Ca, int Cb, int G,
           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int nt_stores) {
    const GnFinItem it = items[blockIdx.y];
    const GnApplyProblem& q = tab[it.problem];
    if ((int)blockIdx.x >= q.cols) return;
    gn_apply_down_share_body<T, SILU, NS>(static_cast<const T*>(q.xa), Ca, static_cast<const T*>(q.xb), Cb, q.H, q.W, G, q.stats, gamma, beta, eps,
                      static_cast<T*>(q.out_act), static_cast<T*>(q.out_raw), q.ncg, q.nstrips, nt_stores, it.b, blockIdx.x);
}

// The same for x2 UP: out[2i] = 3/4 x[i] + 1/4 x[i - 1], out[2i + 1] = 3/4 x[i] + 1/4 x[i + 1] per axis (k = [1,3,3,1], gain 2 per
// axis, zero boundary).  A thread owns one 16-byte channel slot of one INPUT column and walks down a strip of input rows: per
// input row three loads (left, centre, right), each normalised + activated once, give the two horizontally up-sampled columns
// (2 ix, 2 ix + 1) of both tensors; every pair of consecutive filtered rows (h[i], h[i + 1]) emits the output rows 2 i + 1 and
// 2 i + 2.  The kernel is bound by its stores (two tensors of four times the input size: 8 x 16 B per thread and input row):
// 4.1 TB/s, exactly what the LDS-tiled predecessor reached - a write-dominated stream does not get the copy rate on this part.
constexpr int UP_ROWS = 16;                         // input rows per strip
template <typename T, bool SILU, int NS>
__device__ __forceinline__ void gn_apply_up_body(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb,
                        int H, int W, int G, const double* __restrict__ stats,
                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                        T* __restrict__ out_act, T* __restrict__ out_raw, int ncg, int nstrips, int nt_stores, const int by, const int bx) {
    constexpr int PER16 = Elem<T>::PER16;
    constexpr int CG = NS * PER16;
    constexpr int UP_COLS = 256 / NS;               // input columns per workgroup
    __shared__ float gtab[2 * CG];
    const int C = Ca + Cb, tid = threadIdx.x;
    const int slot = tid % NS, col = tid / NS;
    const int OH = 2 * H, OW = 2 * W;
    int t = by;
    const int cg = t % ncg; t /= ncg;
    const int strip = t % nstrips, b = t / nstrips;
    const int gs = C / G;
    if (tid < CG) {
        const int cc = cg * CG + tid;
        float sc = 0.f, sh = 0.f;
        if (cc < C) {
            const double n = (double)gs * H * W;
            const int g = cc / gs;
            const double m = stats[((long long)b * G + g) * 2] / n;
            double var = stats[((long long)b * G + g) * 2 + 1] / n - m * m;
            if (var < 0.0) var = 0.0;
            const float pm = (float)m;
            sc = (float)(1.0 / sqrt(var + (double)eps)) * gamma[cc];
            sh = beta[cc] - pm * sc;
        }
        gtab[2 * tid] = sc; gtab[2 * tid + 1] = sh;
    }
    __syncthreads();
    const int c = cg * CG + slot * PER16;
    const int ix = bx * UP_COLS + col;
    if (c >= C || ix >= W) return;
    float pa[PER16], pb[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) { pa[e] = gtab[2 * (slot * PER16 + e)]; pb[e] = gtab[2 * (slot * PER16 + e) + 1]; }
    const T* const src = (c < Ca) ? xa + c : xb + (c - Ca);
    const int cs = (c < Ca) ? Ca : Cb;
    const long long ibase = (long long)b * H * W;
    const long long obase = (long long)b * OH * OW;
    const bool x_in[3] = {ix > 0, true, ix + 1 < W};
    struct HRow { float eA[PER16], oA[PER16], eR[PER16], oR[PER16]; };    // columns 2 ix (even) / 2 ix + 1 (odd), activated / raw
    auto hrow = [&](int iy, HRow& h) {
#pragma unroll
        for (int e = 0; e < PER16; ++e) { h.eA[e] = 0.f; h.oA[e] = 0.f; h.eR[e] = 0.f; h.oR[e] = 0.f; }
        if (iy < 0 || iy >= H) return;
        const T* const row = src + (ibase + (long long)iy * W) * cs;
        uint4 q[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            q[j] = x_in[j] ? *reinterpret_cast<const uint4*>(row + (long long)(ix - 1 + j) * cs) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // weights of this tap in the even / odd output column: left (1/4, 0), centre (3/4, 3/4), right (0, 1/4)
            const float we = j == 0 ? 0.25f : (j == 1 ? 0.75f : 0.f), wo = j == 0 ? 0.f : (j == 1 ? 0.75f : 0.25f);
            alignas(16) T raw[PER16];
            *reinterpret_cast<uint4*>(raw) = q[j];
            if constexpr (sizeof(T) == 2) {
                uint32_t aw[4] = {0u, 0u, 0u, 0u};
                if (x_in[j]) {
#pragma unroll
                    for (int e = 0; e < PER16; e += 2) {
                        f32x2 y = __builtin_elementwise_fma(f32x2{to_f32(raw[e]), to_f32(raw[e + 1])}, f32x2{pa[e], pa[e + 1]}, f32x2{pb[e], pb[e + 1]});
                        if (SILU) y = silu2(y);
                        aw[e / 2] = pack2(y.x, y.y, (T*)nullptr);
                    }
                }
                const uint32_t r[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
                if (j != 2) {
                    uint32_t wl = tap_weight_bits(we, (T*)nullptr), wh = wl << 16;
                    keep_rw(wl); keep_rw(wh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h.eA[2 * i] = dot2_acc(aw[i], wl, h.eA[2 * i], (T*)nullptr); h.eA[2 * i + 1] = dot2_acc(aw[i], wh, h.eA[2 * i + 1], (T*)nullptr);
                        h.eR[2 * i] = dot2_acc(r[i], wl, h.eR[2 * i], (T*)nullptr); h.eR[2 * i + 1] = dot2_acc(r[i], wh, h.eR[2 * i + 1], (T*)nullptr);
                    }
                }
                if (j != 0) {
                    uint32_t wl = tap_weight_bits(wo, (T*)nullptr), wh = wl << 16;
                    keep_rw(wl); keep_rw(wh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        h.oA[2 * i] = dot2_acc(aw[i], wl, h.oA[2 * i], (T*)nullptr); h.oA[2 * i + 1] = dot2_acc(aw[i], wh, h.oA[2 * i + 1], (T*)nullptr);
                        h.oR[2 * i] = dot2_acc(r[i], wl, h.oR[2 * i], (T*)nullptr); h.oR[2 * i + 1] = dot2_acc(r[i], wh, h.oR[2 * i + 1], (T*)nullptr);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < PER16; ++e) {
                    const float xr = to_f32(raw[e]);
                    float y = 0.f;
                    if (x_in[j]) {
                        y = fmaf(xr, pa[e], pb[e]);
                        if (SILU) y = silu_f(y);
                        T ya; from_f32(ya, y); y = to_f32(ya);
                    }
                    if (j != 2) { h.eA[e] = fmaf(we, y, h.eA[e]); h.eR[e] = fmaf(we, xr, h.eR[e]); }
                    if (j != 0) { h.oA[e] = fmaf(wo, y, h.oA[e]); h.oR[e] = fmaf(wo, xr, h.oR[e]); }
                }
            }
        }
    };
    auto store_row = [&](int oy, const HRow& a, float wa_, const HRow& bq, float wb_) {     // out[oy] = wa_ a + wb_ bq, both columns
        const long long o0 = (obase + (long long)oy * OW + 2 * ix) * C + c;
        float v[4][PER16];
#pragma unroll
        for (int e = 0; e < PER16; ++e) {
            v[0][e] = fmaf(wb_, bq.eA[e], wa_ * a.eA[e]); v[1][e] = fmaf(wb_, bq.oA[e], wa_ * a.oA[e]);
            v[2][e] = fmaf(wb_, bq.eR[e], wa_ * a.eR[e]); v[3][e] = fmaf(wb_, bq.oR[e], wa_ * a.oR[e]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            T* const dst = (k < 2 ? out_act : out_raw);
            if (dst == nullptr) continue;
            alignas(16) T ov[PER16];
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint4*>(ov) = make_uint4(pack2(v[k][0], v[k][1], (T*)nullptr), pack2(v[k][2], v[k][3], (T*)nullptr),
                                                           pack2(v[k][4], v[k][5], (T*)nullptr), pack2(v[k][6], v[k][7], (T*)nullptr));
            } else {
#pragma unroll
                for (int e = 0; e < PER16; ++e) from_f32(ov[e], v[k][e]);
            }
            if (nt_stores) store16_nt(dst + o0 + (k & 1) * C, *reinterpret_cast<const uint4*>(ov));
            else *reinterpret_cast<uint4*>(dst + o0 + (k & 1) * C) = *reinterpret_cast<const uint4*>(ov);
        }
    };
    const int up_rows = (H + nstrips - 1) / nstrips;
    const int iy0 = strip * up_rows, iy1 = min(H, iy0 + up_rows);
    HRow h0, h1;
    hrow(iy0 - 1, h0);
    for (int i = iy0 - 1; i < iy1; ++i) {            // the pair (h[i], h[i + 1]) emits the output rows 2 i + 1 and 2 i + 2
        hrow(i + 1, h1);
        if (i >= iy0) store_row(2 * i + 1, h0, 0.75f, h1, 0.25f);
        if (i + 1 < iy1) store_row(2 * i + 2, h1, 0.75f, h0, 0.25f);
        h0 = h1;
    }
}
template <typename T, bool SILU, int NS>
__global__ __launch_bounds__(256)
void gn_apply_up_kernel(const T* __restrict__ xa, int Ca, const T* __restrict__ xb, int Cb, int H, int W, int G, const double* __restrict__ stats,
          const float* __restrict__ gamma, const float* __restrict__ beta, float eps, T* __restrict__ out_act, T* __restrict__ out_raw,
          int ncg, int nstrips, int nt_stores) {
    gn_apply_up_body<T, SILU, NS>(xa, Ca, xb, Cb, H, W, G, stats, gamma, beta, eps, out_act, out_raw, ncg, nstrips, nt_stores, blockIdx.y, blockIdx.x);
}
// the strips of SEVERAL problems in one launch (grouped evaluation of a ragged stream's micro-batches, common.h): blockIdx.y = an item
// (problem, the y index of the problem's own launch) of a host-built list; a strip is computed by the code of its own launch
template <typename T, bool SILU, int NS>
__global__ __launch_bounds__(256)
void gn_apply_up_group_kernel(const GnApplyProblem* __restrict__ tab, const GnFinItem* __restrict__ items, int Ca, int Cb, int G,
           const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int nt_stores) {
    const GnFinItem it = items[blockIdx.y];
    const GnApplyProblem& q = tab[it.problem];
    if ((int)blockIdx.x >= q.cols) return;
    gn_apply_up_body<T, SILU, NS>(static_cast<const T*>(q.xa), Ca, static_cast<const T*>(q.xb), Cb, q.H, q.W, G, q.stats, gamma, beta, eps,
                      static_cast<T*>(q.out_act), static_cast<T*>(q.out_raw), q.ncg, q.nstrips, nt_stores, it.b, blockIdx.x);
}

template <typename T, int RESAMPLE>
__device__ __forceinline__ void fir_body(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ out,
                                         int H, int W, int C, int ppb, int C8, int PL, const int b) {
    const int tid = threadIdx.x;
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

template <typename T, int RESAMPLE>
__global__ void fir_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ out,
                           int H, int W, int C, int ppb, int C8, int PL) {
    fir_body<T, RESAMPLE>(x, add, out, H, W, C, ppb, C8, PL, blockIdx.y);
}
// the same for the batch items of several problems in one launch (grouped evaluation, common.h): grid.x = the largest problem's blocks
template <typename T, int RESAMPLE>
__global__ void fir_group_kernel(const FirProblem* __restrict__ tab, const GnFinItem* __restrict__ items, int C, int C8, int PL) {
    const GnFinItem it = items[blockIdx.y];
    const FirProblem& q = tab[it.problem];
    fir_body<T, RESAMPLE>(static_cast<const T*>(q.x), static_cast<const T*>(q.add), static_cast<T*>(q.out), q.H, q.W, C, q.ppb, C8, PL, it.b);
}

template <typename T>
static int gn_stats_t(const void* xa, int Ca, const void* xb, int Cb, int B, int HW, int G, double* stats,
                      hipStream_t st) {
    const GnGeom g = gn_geom(Ca + Cb);
    const int per_thread = pixels_per_thread((long long)(switches().batch_invariant ? 1 : B) * HW, g.PL, 256);   // (a thread's fp32 partial spans `per_thread` pixels)
    int ppb = g.PL * per_thread;
    const int nblk = cdiv(HW, ppb);
    hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(nblk, B), dim3(g.NT), 0, st, (const T*)xa, Ca, (const T*)xb, Cb,
                       HW, G, stats, ppb, g.C8, g.PL);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

// Rows per strip of the two resampling kernels: 16 where that already gives every CU four workgroups (the bench batch: thousands), halved
// down to 4 for small calls.  One utterance per call (the reference's own operating point, enhancement.py:66-72) at 16 rows is 128
// workgroups for the level-0 down-sampling launch - one wave per SIMD on half the chip, a launch that lasts as long as ONE strip's chain
// of row loads (41 us for 33 MB, profiles/r05a_b1_eager_rocprofv3_kernel_stats.csv); short strips re-read 3 halo rows per strip (4 rows:
// 1.4 x the reads) and finish sooner.  A strip boundary does not change an output's arithmetic (each output row is the same four-tap
// combination of the same filtered rows), so the choice is invisible in the results (test_groupnorm_fir_fused, STORM_GN_ROWS sweep).
static int strip_rows(int rows_total, long long wgs_per_strip, int dflt) {
    if (switches().gn_rows > 0) return switches().gn_rows;
    int r = dflt;
    while (r > 4 && wgs_per_strip * cdiv(rows_total, r) < 4LL * device_cus()) r >>= 1;
    return r;
}

// The launch of the down-sampling kernel: channel groups, strips, and which of the two kernels.  The activation shared through LDS
// (gn_apply_down_share_kernel) for full launches - at least four workgroups per CU; the barrier-free kernel for small calls (a chain of
// latencies, not of instructions: measured slower there, profiles/r05_probe_gn_down_share.txt).  STORM_GN_DOWN_SHARE: 0 = this rule, 1 = never,
// 2 = always (A/B, tests)
struct DownPlan { int ncg, nstrips; long long gy; bool share; };
static DownPlan down_plan(int C, int B, int H, int W, int per16, int NSr) {
    const int OH = H / 2, OW = W / 2;
    DownPlan d;
    d.ncg = cdiv(C, NSr * per16);
    d.nstrips = cdiv(OH, strip_rows(OH, (long long)cdiv(OW, 256 / NSr) * d.ncg * B, DN_ROWS));
    d.gy = (long long)d.ncg * d.nstrips * B;
    const int shsw = switches().gn_down_share;
    d.share = shsw == 2 || (shsw == 0 && (long long)cdiv(OW, 256 / NSr) * d.gy >= 4LL * device_cus());
    return d;
}

template <typename T, int R>
static int gn_apply_t(const void* xa, int Ca, const void* xb, int Cb, int B, int H, int W, int G,
                      const double* stats, const float* gamma, const float* beta, float eps, int silu,
                      void* out_act, void* out_raw, hipStream_t st) {
    const GnGeom g = gn_geom(Ca + Cb);
    // slots of a pixel per workgroup: the widest of 32 / 16 / 8 that tiles the channel count (256 channels of 16-bit data: 32)
    const int slots = (Ca + Cb) / Elem<T>::PER16;
    const int NSr = switches().gn_wide == 0 ? 8 : (slots % 32 == 0 ? 32 : slots % 16 == 0 ? 16 : 8);
    if (R == 2) {
        const int OH = H / 2, OW = W / 2;
        const DownPlan dp = down_plan(Ca + Cb, B, H, W, Elem<T>::PER16, NSr);
        const int ncg = dp.ncg, nstrips = dp.nstrips;
        const long long gy = dp.gy;
        STORM_CHECK(OH > 0 && OW > 0 && gy < 65536, "storm_gn_apply: down-sampling grid %lld out of range", gy);
        const bool share = dp.share;
#define STORM_GN_DOWN1(KERN_, SILU_, NS_) hipLaunchKernelGGL((KERN_<T, SILU_, NS_>), dim3(cdiv(OW, 256 / NS_), (unsigned)gy), dim3(256), 0, st, \
                           (const T*)xa, Ca, (const T*)xb, Cb, H, W, G, stats, gamma, beta, eps, (T*)out_act, (T*)out_raw, ncg, nstrips, (switches().gn_nt >> 1) & 1)
#define STORM_GN_DOWN(SILU_, NS_) do { if (share) STORM_GN_DOWN1(gn_apply_down_share_kernel, SILU_, NS_); else STORM_GN_DOWN1(gn_apply_down_kernel, SILU_, NS_); } while (0)
        if (silu) { if (NSr == 32) STORM_GN_DOWN(true, 32); else if (NSr == 16) STORM_GN_DOWN(true, 16); else STORM_GN_DOWN(true, 8); }
        else { if (NSr == 32) STORM_GN_DOWN(false, 32); else if (NSr == 16) STORM_GN_DOWN(false, 16); else STORM_GN_DOWN(false, 8); }
#undef STORM_GN_DOWN
#undef STORM_GN_DOWN1
        STORM_LAUNCH_CHECK();
        return STORM_OK;
    }
    if (R == 1) {
        const int ncg = cdiv(Ca + Cb, NSr * Elem<T>::PER16), nstrips = cdiv(H, strip_rows(H, (long long)cdiv(W, 256 / NSr) * ncg * B, UP_ROWS));
        const long long gy = (long long)ncg * nstrips * B;
        STORM_CHECK(gy < 65536, "storm_gn_apply: up-sampling grid %lld out of range", gy);
#define STORM_GN_UP(SILU_, NS_) hipLaunchKernelGGL((gn_apply_up_kernel<T, SILU_, NS_>), dim3(cdiv(W, 256 / NS_), (unsigned)gy), dim3(256), 0, st, \
                           (const T*)xa, Ca, (const T*)xb, Cb, H, W, G, stats, gamma, beta, eps, (T*)out_act, (T*)out_raw, ncg, nstrips, switches().gn_nt & 1)
        if (silu) { if (NSr == 32) STORM_GN_UP(true, 32); else if (NSr == 16) STORM_GN_UP(true, 16); else STORM_GN_UP(true, 8); }
        else { if (NSr == 32) STORM_GN_UP(false, 32); else if (NSr == 16) STORM_GN_UP(false, 16); else STORM_GN_UP(false, 8); }
#undef STORM_GN_UP
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

// The reference's ONE native op with its own argument list (op/upfirdn2d.cpp:12-22, op/upfirdn2d_kernel.cu:107-369): N planes [H][W]
// (the reference reshapes [B,C,H,W] to [B*C,H,W,1]), zero-stuffing by (up_x, up_y), padding / cropping by the four pads, a true convolution
// with kernel[kh][kw] (upfirdn2d_native flips it, op/upfirdn2d.py:183-184) and decimation by (down_x, down_y).  One thread per output
// element gathers its taps (zero-stuffed positions are skipped, not multiplied); fp32 accumulation in tap order.  The network never
// calls this - its FIR steps are fused into gn_apply_up / gn_apply_down / fir_kernel on NHWC tensors - it is the drop-in for the seam.
template <typename T>
__global__ void upfirdn2d_planes_kernel(const T* __restrict__ x, const float* __restrict__ k, T* __restrict__ out, int H, int W, int OH, int OW,
                                        int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_y0) {
    const long long plane = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= OH * OW) return;
    const int oy = p / OW, ox = p - oy * OW;
    const T* xp = x + plane * H * W;
    float acc = 0.0f;
    for (int ky = 0; ky < kh; ++ky) {
        const int uy = oy * down_y + ky - pad_y0;          // row of the zero-stuffed image
        if (uy < 0 || uy % up_y != 0 || uy / up_y >= H) continue;
        for (int kx = 0; kx < kw; ++kx) {
            const int ux = ox * down_x + kx - pad_x0;
            if (ux < 0 || ux % up_x != 0 || ux / up_x >= W) continue;
            acc += to_f32(xp[(long long)(uy / up_y) * W + ux / up_x]) * k[(kh - 1 - ky) * kw + (kw - 1 - kx)];
        }
    }
    from_f32(out[plane * OH * OW + p], acc);
}

template <typename T>
static int upfirdn2d_t(const void* x, const float* k, void* out, int N, int H, int W, int OH, int OW, int kh, int kw, int up_x, int up_y,
                       int down_x, int down_y, int pad_x0, int pad_y0, hipStream_t st) {
    hipLaunchKernelGGL((upfirdn2d_planes_kernel<T>), dim3(cdiv((long long)OH * OW, 256), N), dim3(256), 0, st, (const T*)x, k, (T*)out, H, W, OH, OW,
                       kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0);
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

// grouped FIR x2 (8-channel pyramids): fills one problem's table entry and returns its block count
template <int R>
static int fir_problem(const void* x, const void* add, void* out, int B, int H, int W, int C, FirProblem& q) {
    const GnGeom g = gn_geom(C);
    const int OHW = R == 1 ? 4 * H * W : H * W / 4;
    q.x = x; q.add = add; q.out = out; q.H = H; q.W = W;
    q.ppb = g.PL * pixels_per_thread((long long)B * OHW, g.PL, 32);
    return cdiv(OHW, q.ppb);
}
int fir_group_problem(int resample, const void* x, const void* add, void* out, int B, int H, int W, int C, FirProblem& q) {
    return resample == 1 ? fir_problem<1>(x, add, out, B, H, W, C, q) : fir_problem<2>(x, add, out, B, H, W, C, q);
}
// ---- grouped GroupNorm-apply + SiLU + FIR x2 (common.h) -----------------------------------------------------------------------------------
static int gn_group_ns(int C, int dtype) {
    const int per16 = dtype == STORM_F32 ? 4 : 8, slots = C / per16;
    return switches().gn_wide == 0 ? 8 : (slots % 32 == 0 ? 32 : slots % 16 == 0 ? 16 : 8);
}
bool gn_apply_group_plan(int resample, int C, int P, const int* B, const int* H, const int* W, int dtype, GnApplyGroupPlan& plan) {
    if ((dtype != STORM_BF16 && dtype != STORM_F16) || (resample != 1 && resample != 2) || C % 8 != 0 || C > GN_MAX_C) return false;
    const int NSr = gn_group_ns(C, dtype), ncg = cdiv(C, NSr * 8);
    auto rows_of = [&](int g) { return resample == 1 ? H[g] : H[g] / 2; };
    auto cols_of = [&](int g) { return cdiv(resample == 1 ? W[g] : W[g] / 2, 256 / NSr); };
    auto wgs = [&](int r) { long long n = 0; for (int g = 0; g < P; ++g) n += (long long)cols_of(g) * ncg * B[g] * cdiv(rows_of(g), r); return n; };
    int r = resample == 1 ? UP_ROWS : DN_ROWS;
    if (switches().gn_rows > 0) r = switches().gn_rows;
    else while (r > 4 && wgs(r) < 4LL * device_cus()) r >>= 1;      // strip_rows' rule on the GROUP's workgroups
    plan.rows_per_strip = r;
    const int shsw = switches().gn_down_share;
    plan.share = resample == 2 && (shsw == 2 || (shsw == 0 && wgs(r) >= 4LL * device_cus()));
    plan.max_cols = 0; plan.items = 0;
    for (int g = 0; g < P; ++g) {
        if (rows_of(g) < 1 || (resample == 2 && (H[g] % 2 || W[g] % 2))) return false;
        plan.max_cols = std::max(plan.max_cols, cols_of(g));
        plan.items += (long long)ncg * cdiv(rows_of(g), r) * B[g];
    }
    return plan.items > 0 && plan.items < 65536;
}
long long gn_apply_group_problem(int resample, int C, int B, int dtype, const GnApplyGroupPlan& plan, int g, GnApplyProblem& q, GnFinItem* items) {
    const int NSr = gn_group_ns(C, dtype);
    q.ncg = cdiv(C, NSr * 8);
    q.nstrips = cdiv(resample == 1 ? q.H : q.H / 2, plan.rows_per_strip);
    q.cols = cdiv(resample == 1 ? q.W : q.W / 2, 256 / NSr);
    const long long gy = (long long)q.ncg * q.nstrips * B;
    for (long long y = 0; y < gy; ++y) { items[y].problem = g; items[y].b = (int)y; }
    return gy;
}
int launch_gn_apply_group(int resample, const GnApplyProblem* dev_tab, const void* dev_items, const GnApplyGroupPlan& plan, int Ca, int Cb, int G,
                          const float* gamma, const float* beta, float eps, int dtype, hipStream_t st) {
    STORM_CHECK(dev_tab && dev_items && gamma && beta && plan.items > 0 && plan.items < 65536 && plan.max_cols > 0, "storm_gn_apply (group): bad arguments");
    const int NSr = gn_group_ns(Ca + Cb, dtype);
    const GnFinItem* it = static_cast<const GnFinItem*>(dev_items);
    const int nt = resample == 1 ? (switches().gn_nt & 1) : ((switches().gn_nt >> 1) & 1);
#define STORM_GAG1(KERN_, T_, NS_) hipLaunchKernelGGL((KERN_<T_, true, NS_>), dim3((unsigned)plan.max_cols, (unsigned)plan.items), dim3(256), 0, st, dev_tab, it, Ca, Cb, G, gamma, beta, eps, nt)
#define STORM_GAG(KERN_, T_) do { if (NSr == 32) STORM_GAG1(KERN_, T_, 32); else if (NSr == 16) STORM_GAG1(KERN_, T_, 16); else STORM_GAG1(KERN_, T_, 8); } while (0)
#define STORM_GAGT(T_) do { if (resample == 1) STORM_GAG(gn_apply_up_group_kernel, T_); else if (plan.share) STORM_GAG(gn_apply_down_share_group_kernel, T_); \
                            else STORM_GAG(gn_apply_down_group_kernel, T_); } while (0)
    if (dtype == STORM_BF16) STORM_GAGT(bf16_t);
    else if (dtype == STORM_F16) STORM_GAGT(half_t);
    else STORM_CHECK(false, "storm_gn_apply (group): dtype %d", dtype);
#undef STORM_GAGT
#undef STORM_GAG
#undef STORM_GAG1
    STORM_LAUNCH_CHECK();
    return STORM_OK;
}

int launch_fir_group(int resample, const FirProblem* dev_tab, const void* dev_items, int n_items, int max_blocks, int C, int dtype, hipStream_t st) {
    STORM_CHECK(dev_tab && dev_items && n_items > 0 && n_items < 65536 && max_blocks > 0 && C % 8 == 0 && C <= GN_MAX_C, "storm_fir (group): bad arguments");
    const GnGeom g = gn_geom(C);
    const GnFinItem* it = static_cast<const GnFinItem*>(dev_items);
#define STORM_FIRG(T_, R_) hipLaunchKernelGGL((fir_group_kernel<T_, R_>), dim3(max_blocks, n_items), dim3(g.NT), 0, st, dev_tab, it, C, g.C8, g.PL)
    if (dtype == STORM_BF16) { if (resample == 1) STORM_FIRG(bf16_t, 1); else STORM_FIRG(bf16_t, 2); }
    else if (dtype == STORM_F16) { if (resample == 1) STORM_FIRG(half_t, 1); else STORM_FIRG(half_t, 2); }
    else { if (resample == 1) STORM_FIRG(float, 1); else STORM_FIRG(float, 2); }
#undef STORM_FIRG
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

int storm::launch_gn_finalize_group(const GnFinProblem* dev_tab, const void* dev_items, int n_items, int groups, hipStream_t st) {
    STORM_CHECK(dev_tab && dev_items && n_items > 0 && n_items < 65536 && groups > 0, "storm_gn_finalize (group): bad arguments");
    hipLaunchKernelGGL(gn_finalize_group_kernel, dim3(groups, n_items), dim3(256), 0, st, dev_tab, static_cast<const GnFinItem*>(dev_items), groups);
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

// name (as rocprofv3 prints it, without the argument list) of the kernel storm_gn_apply launches for these arguments; static storage
extern "C" const char* storm_gn_apply_kernel_name(int C, int B, int H, int W, int silu, int resample, int dtype) {
    static thread_local char name[160];
    const char* const tn = dtype == STORM_F32 ? "float" : dtype == STORM_F16 ? "storm::half_t" : "storm::bf16_t";
    const int per16 = dtype == STORM_F32 ? 4 : 8;
    if (resample == 0) { snprintf(name, sizeof(name), "storm::gn_apply_kernel<%s, 0>", tn); return name; }
    const int slots = C / per16;
    const int NSr = switches().gn_wide == 0 ? 8 : (slots % 32 == 0 ? 32 : slots % 16 == 0 ? 16 : 8);
    const bool share = resample == 2 && down_plan(C, B, H, W, per16, NSr).share;
    snprintf(name, sizeof(name), "storm::gn_apply_%s_kernel<%s, %s, %d>", resample == 1 ? "up" : share ? "down_share" : "down", tn, silu ? "true" : "false", NSr);
    return name;
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

extern "C" int storm_upfirdn2d(const void* input, const float* kernel, void* out, int N, int H, int W, int kh, int kw, int up_x, int up_y,
                               int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, int dtype, storm_stream_t s) {
    STORM_CHECK(input && kernel && out && N > 0 && H > 0 && W > 0 && kh > 0 && kw > 0, "storm_upfirdn2d: bad arguments (N=%d H=%d W=%d kernel %dx%d)", N, H, W, kh, kw);
    STORM_CHECK(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "storm_upfirdn2d: up / down factors must be positive");
    STORM_CHECK(N < 65536, "storm_upfirdn2d: %d planes exceed the launch grid (split the call)", N);
    const long long OH = storm_upfirdn2d_out_size(H, up_y, down_y, pad_y0, pad_y1, kh), OW = storm_upfirdn2d_out_size(W, up_x, down_x, pad_x0, pad_x1, kw);
    STORM_CHECK(OH > 0 && OW > 0 && OH * OW < (1LL << 31), "storm_upfirdn2d: empty or oversized output %lld x %lld", OH, OW);
    hipStream_t st = (hipStream_t)s;
    if (dtype == STORM_BF16) return upfirdn2d_t<bf16_t>(input, kernel, out, N, H, W, (int)OH, (int)OW, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, st);
    if (dtype == STORM_F16) return upfirdn2d_t<half_t>(input, kernel, out, N, H, W, (int)OH, (int)OW, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, st);
    if (dtype == STORM_F32) return upfirdn2d_t<float>(input, kernel, out, N, H, W, (int)OH, (int)OW, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0, st);
    set_error("storm_upfirdn2d: dtype %d", dtype);
    return STORM_ERR_UNSUPPORTED;
}

extern "C" long long storm_upfirdn2d_out_size(int in, int up, int down, int pad0, int pad1, int ktaps) {
    if (up <= 0 || down <= 0 || ktaps <= 0) return -1;
    const long long span = (long long)in * up + pad0 + pad1 - ktaps;       // upfirdn2d_kernel.cu:226-227 / op/upfirdn2d.py:197-198
    return span < 0 ? 0 : span / down + 1;
}
