// Helpers shared by the chunk-unrolled LDS-DMA convolution kernels (conv_pipe.hip: 256 output channels x 256 pixels;
// conv_pipe128.hip: 128 output channels x 512 pixels).
#pragma once
#include <type_traits>
#include <utility>
#include "conv_params.h"

namespace storm { namespace pipe {

constexpr int WROW = 64;                                      // bytes per weight row and phase (two k-groups of 16)
constexpr uint32_t OOB = BUF_OOB;                             // per-lane offset that is out of range of every buffer here

// LDS byte offset of 16-B slot s (0..3) of row `row` of a weight phase tile: the 16 lanes of a ds_read_b128
// group hit 16 distinct 16-B bank groups.
STORM_HD int w_off(int row, int s) { return row * WROW + ((s ^ ((row >> 2) & 3)) << 4); }

template <int N> using IC = std::integral_constant<int, N>;
template <typename F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(IC<Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// keep a wave-uniform value in an SGPR (stops re-materialisation from the kernarg segment inside the loop)
__device__ __forceinline__ int pin(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(x));
#endif
    return x;
}
__device__ __forceinline__ int uniform(int x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(x);
#else
    return x;
#endif
}

// a * b + c for a, b < 2^24 (pixel indices, per-pixel byte strides): ONE 32-bit instruction (v_mad_u32_u24).  A plain 32-bit
// product goes through v_mad_u64_u32 and a 64-bit register pair - which, spilled, put a scratch reload + s_waitcnt vmcnt(0)
// into the pipelined loop of conv_pipe.hip (found in round 2: once per chunk, draining every in-flight patch piece).
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b) + c;
#else
    return a * b + c;
#endif
}

// The parameter block of the chunk-descriptor kernels, read through the kernarg segment (it is the kernel's first argument, at
// offset 0; the constant address space keeps every field access a scalar load and nothing of the 2.6-KiB block is copied).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const PipeParams __attribute__((address_space(4)))* PipeArgPtr;
__device__ __forceinline__ PipeArgPtr pipe_args(const PipeParams&) { return (PipeArgPtr)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ void relaunder(PipeArgPtr& p) { asm volatile("" : "+s"(p)); }     // (stops hoisting of its scalar loads)
template <typename V> __device__ __forceinline__ void launder(V& v) { asm volatile("" : "+v"(v)); }   // opaque per use: derived values are not kept live
__device__ __forceinline__ char* as_global(unsigned long long u) { asm volatile("" : "+s"(u)); typedef __attribute__((address_space(1))) char G; return (char*)(G*)u; }
#else
typedef const PipeParams* PipeArgPtr;                       // (host simulation / the host pass of the device build)
__device__ __forceinline__ PipeArgPtr pipe_args(const PipeParams& a) { return &a; }
__device__ __forceinline__ void relaunder(PipeArgPtr&) {}
template <typename V> __device__ __forceinline__ void launder(V&) {}
__device__ __forceinline__ char* as_global(unsigned long long u) { return reinterpret_cast<char*>(u); }
#endif

// The K loop as chunk descriptors of `kc` channels each (conv_pipe.hip; kc = 64 or 32).  Returns false when the convolution
// is outside what the pipelined kernels cover.
bool build_pipe_params(const storm_conv_args& a, PipeParams& p, int kc);

}}  // namespace storm::pipe
