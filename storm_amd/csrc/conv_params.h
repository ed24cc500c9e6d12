// Shared between the convolution kernels (conv_igemm.hip, conv_pipe.hip): the kernel-side parameter block,
// MFMA wrappers and the fused GroupNorm-apply slot transform.
#pragma once
#include <cstring>
#include "common.h"
#include "conv_index.h"

namespace storm {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<half_t> {
    typedef f16x8 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
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

// 16-byte global load at (wave-uniform base pointer) + (32-bit per-lane byte offset): lowers to the
// saddr + voffset addressing form, so no 64-bit per-lane address is ever kept (or spilled).
__device__ __forceinline__ uint4 ld16(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st16(void* base, uint32_t byte_off, uint4 v) {
    *reinterpret_cast<uint4*>(static_cast<char*>(base) + byte_off) = v;
}



// y = act(x * scale + shift) on the 16-byte slot held in `v` (GroupNorm-apply + SiLU fused into the operand load: the
// normalised tensor is never written to HBM).  ss = the slot's channels' scales in ss[0 .. PER16), shifts in ss[8 .. 8 + PER16)
// (the table layout of storm_gn_finalize_ss), so channel pairs are adjacent registers and the affine, the exponent scaling,
// the +1 and the final product are packed fp32 operations.  The transform costs 9-14 % of a fused convolution's time, and that is
// the extra LDS pass, not this arithmetic: without the two transcendentals per element the kernels are 0-2 % faster
// (profiles/r03_ubench.txt).
__device__ __forceinline__ f32x2 gn_affine2(f32x2 x, const float (&ss)[16], int i) {
    return __builtin_elementwise_fma(x, f32x2{ss[2 * i], ss[2 * i + 1]}, f32x2{ss[8 + 2 * i], ss[8 + 2 * i + 1]});
}
__device__ __forceinline__ uint4 gn_act_slot(uint4 v, const float (&ss)[16], int silu, bf16_t*) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 y = gn_affine2(f32x2{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)}, ss, i);
        if (silu) y = silu2(y);
        w[i] = pack_bf16x2(y.x, y.y);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 gn_act_slot(uint4 v, const float (&ss)[16], int silu, half_t*) {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x2 y = gn_affine2(f32x2{f16_bits_to_f32((uint16_t)(w[i] & 0xffffu)), f16_bits_to_f32((uint16_t)(w[i] >> 16))}, ss, i);
        if (silu) y = silu2(y);
        w[i] = pack_f16x2(y.x, y.y);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint4 gn_act_slot(uint4 v, const float (&ss)[16], int silu, float*) {
    float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = fmaf(x[i], ss[i], ss[8 + i]);
        if (silu) x[i] = silu_f(x[i]);
    }
    return make_uint4(__float_as_uint(x[0]), __float_as_uint(x[1]), __float_as_uint(x[2]), __float_as_uint(x[3]));
}
// the same with the activation as a compile-time choice: callers that branch ONCE per patch (uniformly) instead of selecting per
// channel pair (as a run-time flag the compiler if-converts the SiLU: both results computed, eight v_cndmask per slot)
template <bool SILU, typename T>
__device__ __forceinline__ uint4 gn_act_slot_t(uint4 v, const float (&ss)[16], T* tag) {
    if constexpr (sizeof(T) == 4) return gn_act_slot(v, ss, SILU ? 1 : 0, tag);
    else {
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x2 x;
            if constexpr (Elem<T>::DT == STORM_BF16) x = f32x2{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
            else x = f32x2{f16_bits_to_f32((uint16_t)(w[i] & 0xffffu)), f16_bits_to_f32((uint16_t)(w[i] >> 16))};
            f32x2 y = gn_affine2(x, ss, i);
            if (SILU) y = silu2(y);
            w[i] = pack2(y.x, y.y, (T*)nullptr);
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// The PER16 scales / shifts of 16-byte slot `slot` of a chunk from its (scale, shift) table (global memory or its LDS image):
// 16-bit operands: slot = one channel octet = 16 consecutive floats; fp32: half an octet.
template <int PER16>
__device__ __forceinline__ void load_ss(const float* table, int slot, float (&ss)[16]) {
    const float* q = table + (PER16 == 8 ? 16 * slot : 16 * (slot >> 1) + 4 * (slot & 1));
#pragma unroll
    for (int j = 0; j < PER16; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(q + j), b = *reinterpret_cast<const float4*>(q + 8 + j);
        ss[j] = a.x; ss[j + 1] = a.y; ss[j + 2] = a.z; ss[j + 3] = a.w;
        ss[8 + j] = b.x; ss[9 + j] = b.y; ss[10 + j] = b.z; ss[11 + j] = b.w;
    }
}


// Kernel-side view of storm_conv_args: the K dimension as up to four single-source "runs"
// (a segment reading cat[xa, xb] becomes two runs), so the inner loops never select a source per
// element and every run field is a scalar loaded once per run.
struct ConvRun {
    const void* src; const void* w;
    long long src_bstride, w_bstride, w_tapstride;
    int C;          // channel stride of src
    int c0, cn;     // channels [c0, c0+cn) of src ...
    int wc0;        // ... multiply weight columns [wc0, wc0+cn)
    int CinP, w_rows, ntaps;
    int gn_silu;    // SiLU after the fused GroupNorm affine
    const float* gn_ss;   // optional fused GroupNorm apply on load: [B][gn_C][2] (scale, shift); channel
    int gn_C, pad_;       //   index of element (c) of this run = wc0 + c
};
struct ConvParams {
    ConvRun run[4];
    int nruns, B, H, W;
    void* out; int outC, Cout; long long out_bstride;
    const float* bias; const float* tbias; int tbias_stride, out_f32;
    const void* skip; long long skip_bstride; float scale; int pad_;
    float* gn_part;   // optional [B][tiles_per_img][outC][2] per-tile (sum, sumsq) of the stored output
    unsigned long long* trace;   // profiling instantiation (ABL & 64) only: [vblock][wave][TRACE_SLOTS] s_memtime stamps
};
constexpr int TRACE_SLOTS = 512;

static inline ConvParams make_params(const storm_conv_args& a) {
    ConvParams p;
    memset(&p, 0, sizeof(p));
    int n = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const storm_conv_seg& g = a.seg[s];
        for (int part = 0; part < 2; ++part) {
            if (part == 1 && g.Cb == 0) break;
            ConvRun& r = p.run[n++];
            r.src = part == 0 ? g.src_a : g.src_b;
            r.src_bstride = part == 0 ? g.bstride_a : g.bstride_b;
            r.C = part == 0 ? g.Ca : g.Cb;
            r.c0 = 0; r.cn = r.C;
            r.wc0 = part == 0 ? 0 : g.Ca;
            r.w = g.w; r.w_bstride = g.w_bstride; r.w_tapstride = g.w_tapstride;
            r.CinP = g.CinP; r.w_rows = g.w_rows; r.ntaps = g.ntaps;
            r.gn_ss = g.gn_ss; r.gn_C = g.Ca + g.Cb; r.gn_silu = g.gn_silu;
        }
    }
    p.nruns = n; p.B = a.B; p.H = a.H; p.W = a.W;
    p.out = a.out; p.outC = a.outC; p.Cout = a.Cout; p.out_bstride = a.out_bstride;
    p.bias = a.bias; p.tbias = a.tbias; p.tbias_stride = a.tbias_stride; p.out_f32 = a.out_f32;
    p.skip = a.skip; p.skip_bstride = a.skip_bstride; p.scale = a.scale;
    p.gn_part = a.gn_part;
    p.trace = nullptr;
    return p;
}

// ---- conv_pipe.hip: the K loop as a host-built list of chunk descriptors -----------------------------------
// One chunk = 64 channels (128 B per pixel; conv_pipe128.hip: 32 channels, 64 B) of one run under all its taps.  Everything that does not depend on the
// workgroup's batch index is computed on the host, so the kernel's chunk boundary is two scalar loads and a few
// scalar adds - no per-tap or per-chunk address arithmetic is left in the pipelined loop.
namespace pipe {
constexpr int MAX_CHUNKS = 36;
struct ChunkDesc {                    // 64 bytes
    unsigned long long src;           // the run's source tensor (batch 0)
    unsigned long long bstride;       // its batch stride in bytes
    unsigned long long ss;            // (scale, shift) pairs of this chunk's channels for batch 0 (fused GroupNorm), or 0
    unsigned int ss_bstride;          // bytes
    unsigned int src_bytes;           // one batch image in bytes (= num_records; 0 in the terminator: every read is zero)
    int C2, cbeg2;                    // pixel stride / this chunk's first channel, in bytes
    int cvalid, ntaps;                // channels of this chunk (<= 64 / 32); 9 or 1
    int silu, wrun;                   // SiLU after the fused affine; index of the weight run
    int w_soff, new_wrun;             // byte offset of the chunk's first weight column inside a weight row; run changes here
};
struct WRunDesc {                     // 32 bytes: the weight matrix of a run, [tap][row][CinP] bf16
    unsigned long long w;             // first element of column wc0
    unsigned int bytes;               // num_records
    int CinP2, rows, tapbytes, pad0_, pad1_;
};
struct PipeParams {
    ChunkDesc chunk[MAX_CHUNKS + 1];  // [nchunks] = terminator (prefetch target of the last chunk)
    WRunDesc wrun[4];
    int nchunks, nchunks9, B, H, W;   // nchunks9: the leading nine-tap chunks (one-tap chunks follow)
    int kslices;                      // split-K instantiation only: K slices per (pixel tile, cout tile), each on its own workgroup
    void* out; int outC, Cout; long long out_bstride;
    const float* bias; const float* tbias; int tbias_stride, out_f32;
    const void* skip; long long skip_bstride; float scale;
    int split_nct;                    // split-K instantiation only: cout tiles of the layer (the kernel's n_ct argument = split_nct * kslices)
    float* gn_part;
    unsigned long long* trace;        // profiling build (-DSTORM_PROFILING) only
};
// Grouped launch (conv_pipe.hip, GROUP instantiation): ONE launch over the pixel tiles of several problems - the same layer (weights, K loop)
// on activation tensors of different batch sizes / widths in different buffers: the ragged micro-batches of a stream (BASELINE.json configs[4]).
// A problem = one PipeParams in a device table; a pixel tile = one entry of a device list built on the host.
struct GroupTile { unsigned int problem, b, yx, tile; };   // yx = ty0 | tx0 << 16; tile = index inside its problem (GroupNorm partials)
}  // namespace pipe

// defined in conv_pipe.hip: software-pipelined 3x3 kernel (bf16 / fp16 operands), 256 output channels per workgroup
bool conv_pipe_supports(const storm_conv_args& a);
int launch_conv_pipe(const storm_conv_args& a, hipStream_t st);
int launch_conv_pipe_half(const storm_conv_args& a, hipStream_t st);      // 128 output channels per workgroup (few pixel tiles)
// split-K (choose_variant 10): the 128-cout tile with the nine-tap chunks of the K loop cut into `slices` contiguous ranges, one workgroup
// per (pixel tile, cout tile, slice) writing an fp32 slab into a.splitk_ws, then one combine launch (fixed summation order: slice 0, 1, ...)
// that applies bias / temb bias / skip / scale, rounds, and writes the GroupNorm partials.  conv_splitk_slices: 0 = no split for this layer.
int conv_splitk_slices(const storm_conv_args& a);
long long conv_splitk_bytes(const storm_conv_args& a, int slices);
int launch_conv_pipe_splitk(const storm_conv_args& a, int slices, hipStream_t st);
// grouped launch of P problems that run the SAME layer (conv_pipe_supports each; same weights / channels / taps, own tensors, B, W):
// conv_pipe_group_prepare fills the host images of the parameter table [P] and the tile list (no device work) and returns the tile
// count (or -1: not groupable); launch_conv_pipe_group runs them from their DEVICE copies.  bn = 256 or 128 couts per workgroup.
long long conv_pipe_group_prepare(const storm_conv_args* a, int P, pipe::PipeParams* table, pipe::GroupTile* tiles, long long max_tiles);
int launch_conv_pipe_group(const pipe::PipeParams* dev_table, const pipe::GroupTile* dev_tiles, long long ntiles, int outC, int bn, int dtype, hipStream_t st);
const char* conv_pipe_kernel_name(int dtype, bool half_tile);
// defined in conv_pipe128.hip: the same pipeline for layers with <= 128 output channels (128 couts x 512 pixels per workgroup)
bool conv_pipe128_supports(const storm_conv_args& a);
int launch_conv_pipe128(const storm_conv_args& a, hipStream_t st);
const char* conv_pipe128_kernel_name(int dtype);
// defined in conv_duo.hip: <= 128 output channels, 128 couts x 256 pixels per 4-wave workgroup, two workgroups per CU, one instruction
// stream and one barrier per phase (staging and the fused GroupNorm transform in the MFMA gaps)
bool conv_duo_supports(const storm_conv_args& a);
int launch_conv_duo(const storm_conv_args& a, hipStream_t st);
const char* conv_duo_kernel_name(int dtype);

// defined in conv_thin.hip: convolutions over an 8-channel input (the stem, the input-skip 1x1s): operands straight from global
// memory, no staging
bool conv_thin_supports(const storm_conv_args& a);
int launch_conv_thin(const storm_conv_args& a, hipStream_t st);
const char* conv_thin_kernel_name(int dtype, int ntaps);
// conv_thin.hip, grouped (the stem / the input-skip 1x1s of several problems in one launch): image = P thin::Params + first[P + 1]
long long conv_thin_group_bytes(int P);
long long conv_thin_group_prepare(const storm_conv_args* a, int P, void* image);
int launch_conv_thin_group(const void* dev_image, int P, long long ntiles, int ntaps, int dtype, hipStream_t st);
// conv_narrow.hip, grouped (the output pyramid's convolutions of several problems in one launch): table = P narrow::Params, tiles of 20 x 32 pixels
long long conv_narrow_group_bytes(int P);
long long conv_narrow_group_tiles(const storm_conv_args& a);
long long conv_narrow_group_prepare(const storm_conv_args* a, int P, void* table, pipe::GroupTile* tiles, long long max_tiles);
int launch_conv_narrow_group(const storm_conv_args& a0, const void* dev_table, const pipe::GroupTile* dev_tiles, long long ntiles, hipStream_t st);
// defined in conv_narrow.hip: 3x3 convolutions to <= 4 output channels (the output pyramid): one 36-row 1x1 GEMM + a nine-point gather
bool conv_narrow_supports(const storm_conv_args& a);
int launch_conv_narrow(const storm_conv_args& a, hipStream_t st);
const char* conv_narrow_kernel_name(const storm_conv_args& a);

}  // namespace storm
