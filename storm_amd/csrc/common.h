// Shared device/host helpers for libstorm_hip (gfx950 / CDNA4 only).
#pragma once
#ifdef STORM_HOST_SIM
#include "hip_host_shim.h"   // tests/sim: lane-accurate CPU simulation (test infrastructure)
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include "../../include/storm_hip.h"

namespace storm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

struct bf16_t { uint16_t v; };   // storage type for bf16 activations / weights
struct half_t { uint16_t v; };   // storage type for fp16 activations / weights (BASELINE.json configs[4])
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// ---- error reporting (storm_last_error) ----
void set_error(const char* fmt, ...);

// A/B and test switches of the launchers.  Production never sets them: the table is filled ONCE, when the library is first
// used, from the environment variables of the same names (so `STORM_CONV_VARIANT=0 python tools/...` still works), and after
// that only storm_set_switch() - the test / tool hook exported next to the ABI - changes it.  A launch reads plain ints.
struct Switches {
    int conv_variant = -1;      // STORM_CONV_VARIANT: force a conv kernel family (see choose_variant), -1 = the dispatcher's choice
    int conv_pipe128 = 1;       // STORM_CONV_PIPE128: 0 = keep the <= 128-cout 3x3 layers on conv_igemm
    int conv_cus = 0;           // STORM_CONV_CUS: pretend the device has this many CUs (persistent tile walks in tests), 0 = ask the device
    int conv_persist = 0;       // STORM_CONV_PERSIST (profiling build)
    int conv_dma = 1;           // STORM_CONV_DMA (profiling build): 0 = register staging in conv_igemm's 128-cout kernel
    int conv_ablate = 0;        // STORM_CONV_ABLATE (profiling build): work-skipping instantiations
    int gn_rows = 0;            // STORM_GN_ROWS (experiment): rows per strip of the GroupNorm + FIR kernels (0 = 16)
    int gn_nt = 5;              // STORM_GN_NT: non-temporal output stores - bit 0 gn_apply_up, bit 1 gn_apply_down (no gain: off), bit 2 conv_thin (A/B: profiles/r04_gnexp.txt)
    int gn_down_share = 0;      // STORM_GN_DOWN_SHARE: gn_apply_down with the activation shared between neighbouring threads through LDS - 0 = the launcher's rule (full launches), 1 = never, 2 = always (A/B, tests)
    int gn_wide = 1;            // STORM_GN_WIDE: 0 = the GroupNorm + FIR kernels with 8 slots (128 B) of a pixel per workgroup (A/B)
    int splitk = 0;             // STORM_SPLITK: 0 = the dispatcher's K slices for few-tile 3x3 layers, 1 = never split, 2 / 4 / 8 = that many (A/B)
    int attn_split = 0;         // STORM_ATTN_SPLIT: 0 = attn_splits' rule (key ranges for calls that leave most CUs idle), 1 = never, 2 / 4 / 8 = that many (tests, A/B)
    int conv_table = 1;         // STORM_CONV_TABLE: 0 = ignore the measured dispatch table (conv_dispatch_table.h), the rule ladder alone decides (the tuner's baseline)
    int splitk_small = 1;       // STORM_SPLITK_SMALL: 0 = split K by the per-image rule only (bit-identical rows across batch sizes), 1 = also for launches of <= 64 workgroups (conv_splitk_slices)
    int batch_invariant = 0;    // STORM_BATCH_INVARIANT: 1 = every launch decision that changes a summation order is taken per IMAGE (as for a one-image call): the kernel
                                //   ladder of choose_variant without the batch-ranged table, no small-call K split, no key-range split of the attention - an utterance then
                                //   comes out the same bits alone and in any batch (serving with dynamic batching; costs the batch-aware selections)
    int graph = -1;             // STORM_GRAPH: HIP-graph replay of storm_ncsnpp_forward - -1 = the handle's mode (storm_ncsnpp_set_graph), 0 = never, 1 = always (A/B)
    unsigned long long conv_trace_ptr = 0;   // STORM_CONV_TRACE_PTR (profiling build): device buffer of tools/conv_trace.py
};
Switches& switches();
// bumped by every storm_set_switch: recorded HIP graphs bake in the kernel selection of the moment (conv variant, split-K, attention split,
// GroupNorm launch geometry ...), so a graph set recorded under another epoch is dropped and re-recorded (ncsnpp_graph.hip::forward_replay)
unsigned long long switch_epoch();
int device_cus();               // CU count of the current device (cached), or switches().conv_cus
#define STORM_CHECK(cond, ...) do { if (!(cond)) { storm::set_error(__VA_ARGS__); return STORM_ERR_INVALID; } } while (0)
#define STORM_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    storm::set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return STORM_ERR_HIP; } } while (0)
#define STORM_LAUNCH_CHECK() STORM_HIP(hipGetLastError())

// ---- bf16 <-> f32 (round-to-nearest-even, NaN quieted) ----
__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {
    union { float f; uint32_t u; } c; c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_bits_to_f32(uint16_t h) {
    union { float f; uint32_t u; } c; c.u = ((uint32_t)h) << 16;
    return c.f;
}

// fp16 <-> f32 (round-to-nearest-even; the compiler's _Float16 conversions: v_cvt_f16_f32 / v_cvt_f32_f16 on the device)
__host__ __device__ inline uint16_t f32_to_f16_bits(float f) {
    const _Float16 h = (_Float16)f;
    uint16_t u; __builtin_memcpy(&u, &h, 2);
    return u;
}
__host__ __device__ inline float f16_bits_to_f32(uint16_t u) {
    _Float16 h; __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
__device__ inline uint32_t pack_f16x2(float lo, float hi) {
    return (uint32_t)f32_to_f16_bits(lo) | ((uint32_t)f32_to_f16_bits(hi) << 16);
}
}  // namespace storm
#include "hw.h"          // every platform-dependent primitive (gfx950 / host pass / test simulator) lives there
namespace storm {

__device__ inline float fast_silu(float y) { return y * hw_rcp(1.0f + hw_exp2(-1.44269504088896341f * y)); }
// the same on a channel pair: everything but the two transcendentals per element is a packed fp32 operation
__device__ __forceinline__ f32x2 silu2(f32x2 y) {
    const f32x2 a = y * f32x2{-1.44269504088896341f, -1.44269504088896341f};
#if defined(STORM_PROFILING) && defined(STORM_EXP_NOTRANS)
    // experiment build (never the product): the same packed arithmetic WITHOUT the four transcendentals - what they cost inside
    // the convolutions (profiles/r03_ubench.txt).  The result is not a SiLU.
    const f32x2 d = a + f32x2{1.0f, 1.0f};
    return y * d;
#else
    const f32x2 d = f32x2{hw_exp2(a.x), hw_exp2(a.y)} + f32x2{1.0f, 1.0f};
    return y * f32x2{hw_rcp(d.x), hw_rcp(d.y)};
#endif
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi, bf16_t*) { return pack_bf16x2(lo, hi); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi, half_t*) { return pack_f16x2(lo, hi); }
// eight values as one 16-byte vector of T (16-bit T only; the fp32 instantiation exists for generic code that never stores it)
template <typename T> __device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) return make_uint4(pack2(v[0], v[1], (T*)nullptr), pack2(v[2], v[3], (T*)nullptr),
                                                    pack2(v[4], v[5], (T*)nullptr), pack2(v[6], v[7], (T*)nullptr));
    else return make_uint4(0u, 0u, 0u, 0u);
}
__device__ __forceinline__ uint32_t tap_weight_bits(float w, bf16_t*) { return f32_to_bf16_bits(w); }   // (exact for the FIR taps)
__device__ __forceinline__ uint32_t tap_weight_bits(float w, half_t*) { return f32_to_f16_bits(w); }

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PER16 = 4;                 // elements per 16-byte slot
    static constexpr int DT = STORM_F32;
};
template <> struct Elem<bf16_t> {
    static constexpr int PER16 = 8;
    static constexpr int DT = STORM_BF16;
};
template <> struct Elem<half_t> {
    static constexpr int PER16 = 8;
    static constexpr int DT = STORM_F16;
};

// Load / store 8 consecutive elements as fp32 (addresses 16-B aligned for bf16, 32-B for f32).
__device__ inline void load8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ inline void load8(const bf16_t* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ inline void load8(const half_t* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = f16_bits_to_f32((uint16_t)(w[i] & 0xffffu));
        v[2 * i + 1] = f16_bits_to_f32((uint16_t)(w[i] >> 16));
    }
}
// Eight elements as they sit in memory (16 B of 16-bit data, 32 B of fp32): fetched early, converted where they are used
template <typename T> struct Raw8 { uint4 q; };
template <> struct Raw8<float> { float4 a, b; };
template <typename T> __device__ __forceinline__ void fetch8(const T* p, Raw8<T>& r) { r.q = *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void fetch8(const float* p, Raw8<float>& r) {
    r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4);
}
__device__ __forceinline__ void unpack8(const Raw8<bf16_t>& r, float (&v)[8]) {
    const uint32_t w[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ void unpack8(const Raw8<half_t>& r, float (&v)[8]) {
    const uint32_t w[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = f16_bits_to_f32((uint16_t)(w[i] & 0xffffu)); v[2 * i + 1] = f16_bits_to_f32((uint16_t)(w[i] >> 16)); }
}
__device__ __forceinline__ void unpack8(const Raw8<float>& r, float (&v)[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
}
__device__ inline void store8(half_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_f16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ inline void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// 16 bytes that will not be read again before they leave the caches (streaming outputs far larger than the L2): non-temporal
typedef uint32_t u32x4_nt_t __attribute__((ext_vector_type(4)));
__device__ inline void store16_nt(void* p, uint4 q) {
    u32x4_nt_t val = {q.x, q.y, q.z, q.w};
    __builtin_nontemporal_store(val, reinterpret_cast<u32x4_nt_t*>(p));
}
__device__ inline void store8(bf16_t* p, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ inline float to_f32(float x) { return x; }
__device__ inline float to_f32(bf16_t x) { return bf16_bits_to_f32(x.v); }
__device__ inline void from_f32(float& d, float x) { d = x; }
__device__ inline void from_f32(bf16_t& d, float x) { d.v = f32_to_bf16_bits(x); }
__device__ inline float to_f32(half_t x) { return f16_bits_to_f32(x.v); }
__device__ inline void from_f32(half_t& d, float x) { d.v = f32_to_f16_bits(x); }

__device__ inline float silu_f(float x) { return x / (1.0f + expf(-x)); }

// wave64 all-reduce (sum / max) through DPP-lowered shuffles
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}


inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- grouped evaluation of P op lists of ONE network on P problems (program.hip; used by storm_ncsnpp_forward_group) -----------------------
// The ragged micro-batches of a stream run the same op sequence on tensors of different (B, T).  Op k of every problem is launched together
// where a grouped kernel exists for it (16-bit 3x3 convolutions of the conv_pipe family: one launch over all problems' pixel tiles),
// one after the other otherwise.  The tables of the grouped launches (absolute device pointers) live in a caller-owned device blob.
struct GroupOp { int k, outC, bn, kind; long long table_off, tiles_off, ntiles; const void* aux; float faux; int pad2_; const void* aux2; int x[4]; };   // kind 0: 3x3 convolution (conv_pipe), 1: GroupNorm finalize, 2: 3x3 convolution to <= 4 planes (conv_narrow; outC = input channels, bn = fused GroupNorm / SiLU flags), 3: convolution over an 8-channel input (conv_thin; outC = problems, bn = taps), 4: FIR x2 of the 8-channel pyramids (outC = channels, bn = blocks << 2 | resample), 5: fused attention (outC = channels), 6: GroupNorm-apply + SiLU + FIR x2 (outC = Ca, bn = Cb, x = {groups, resample, rows per strip << 1 | shared-activation kernel, widest problem's column blocks}, aux / aux2 = gamma / beta, faux = eps)
// one problem of a grouped GroupNorm finalize (norm_resample.hip): the arguments of its own storm_gn_finalize(_ss) call
struct GnFinProblem { const float* pa; const float* pb; double* stats; const float* gamma; const float* beta; float* ss; long long count;
                      int Ca, tiles_a, Cb, tiles_b; float eps; int pad_; };
struct GnFinItem { int problem, b; };
// one problem / one workgroup of a grouped fused attention (attention.hip): the arguments of the problem's own storm_attention call
struct AttnProblem { const void* q; const void* k; const void* vT; void* out; long long q_bs, k_bs, v_bs, o_bs; int L, ldv; };
struct AttnItem { int problem, b, qblock, pad_; };
int attn_query_blocks(int L);
int launch_attention_group(const AttnProblem* dev_tab, const AttnItem* dev_items, int n_items, const float* bias, int C, float scale, int dtype, hipStream_t st);
// one problem of a grouped FIR x2 launch (the 8-channel pyramids: fir_kernel): its own storm_fir_up2 / _down2 arguments + pixels per block
struct FirProblem { const void* x; const void* add; void* out; int H, W, ppb, pad_; };
int fir_group_problem(int resample, const void* x, const void* add, void* out, int B, int H, int W, int C, FirProblem& q);   // returns the problem's blocks
int launch_fir_group(int resample, const FirProblem* dev_tab, const void* dev_items, int n_items, int max_blocks, int C, int dtype, hipStream_t st);
int launch_gn_finalize_group(const GnFinProblem* dev_tab, const void* dev_items, int n_items, int groups, hipStream_t st);
// one problem of a grouped GroupNorm-apply + SiLU + FIR x2 launch (norm_resample.hip: gn_apply_up / gn_apply_down(_share)): the arguments of its own
// storm_gn_apply call + the geometry of the strips; items = (problem, y index of the problem's own launch) as GnFinItem
struct GnApplyProblem { const void* xa; const void* xb; const double* stats; void* out_act; void* out_raw; int H, W, ncg, nstrips, cols, pad_; };
struct GnApplyGroupPlan { int rows_per_strip, share, max_cols; long long items; };
// the strips of the GROUP (any strip length / either down-sampling kernel gives the same bits: chosen by the group's workgroup count); 16-bit, SiLU only
bool gn_apply_group_plan(int resample, int C, int P, const int* B, const int* H, const int* W, int dtype, GnApplyGroupPlan& plan);
// fills problem g's geometry (pointers are the caller's) and appends its items; returns the items written
long long gn_apply_group_problem(int resample, int C, int B, int dtype, const GnApplyGroupPlan& plan, int g, GnApplyProblem& q, GnFinItem* items);
int launch_gn_apply_group(int resample, const GnApplyProblem* dev_tab, const void* dev_items, const GnApplyGroupPlan& plan, int Ca, int Cb, int G,
                          const float* gamma, const float* beta, float eps, int dtype, hipStream_t st);
// bytes of the device blob for these shapes (a bound: every groupable op with all its tiles), 0 = nothing groups
long long program_group_blob_bytes(const storm_op* const* ops, int n_ops, int P, int dtype);
// fill the host image of the blob (tables hold pointers resolved against bufs[p]) and the list of grouped ops; returns their count or < 0
// (stable_bufs: buffers [0, stable_bufs) keep their addresses from call to call - the workspace and the weight arena; an op that references any
//  other buffer - the caller's input / time / output tensors - is never grouped, so the tables depend on those addresses only)
int program_group_build(const storm_op* const* ops, int n_ops, void* const* const* bufs, int n_bufs, int P, int dtype, char* host_blob,
                        long long blob_bytes, GroupOp* gops, int max_gops, int stable_bufs);
// run ops [0, n_ops) of the P problems: grouped ops from the DEVICE copy of the blob, every other op problem by problem; the last op
// (output head) with `negate`
int program_run_group(const storm_op* const* ops, int n_ops, void* const* const* bufs, int n_bufs, int P, int dtype, const char* dev_blob,
                      const GroupOp* gops, int n_gops, int negate, storm_stream_t s);

}  // namespace storm
