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

// The parameter block of the chunk-descriptor kernels, read in place through the kernarg segment (hw.h: kernarg_of)
typedef KArg<PipeParams>::Ptr PipeArgPtr;
__device__ __forceinline__ PipeArgPtr pipe_args(const PipeParams& a) { return kernarg_of(a); }

// The K loop as chunk descriptors of `kc` channels each (conv_pipe.hip; kc = 64 or 32).  Returns false when the convolution
// is outside what the pipelined kernels cover.
bool build_pipe_params(const storm_conv_args& a, PipeParams& p, int kc);

}}  // namespace storm::pipe
