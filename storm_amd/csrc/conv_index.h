// Index math of the implicit-GEMM convolution kernel (conv_igemm.hip), kept in plain
// host+device inline functions so the host-side lane-level simulator
// (tests/sim/sim_conv.cpp) exercises exactly the code the kernel runs.
#pragma once

#ifdef __HIPCC__
#define STORM_HD __host__ __device__ inline
#else
#define STORM_HD inline
#endif

namespace storm { namespace cidx {

constexpr int TILE_H = 8;          // output tile: 8 rows x 32 pixels = 256 pixels
constexpr int TILE_W = 32;
constexpr int PIX_BYTES = 128;     // one K-chunk of one pixel / weight row in LDS: 8 slots x 16 B

template <int TAPS> struct Geo;
template <> struct Geo<9> { static constexpr int PH = TILE_H + 2, PW = TILE_W + 2, NPIX = PH * PW; };
template <> struct Geo<1> { static constexpr int PH = TILE_H, PW = TILE_W, NPIX = PH * PW; };

// LDS byte offset of 16-B slot `slot` (0..7) of row `row` (a patch pixel or a weight row).
// XOR swizzle: 16 consecutive rows read at one slot hit 16 distinct 16-B bank groups.
STORM_HD int lds_off(int row, int slot) { return row * PIX_BYTES + ((slot ^ ((row >> 1) & 7)) << 4); }

// Patch image (activations): pixel `pix` = py * PW + px of the staged patch; its slots are swizzled by the COLUMN
// px, so 16 consecutive px of a fragment read still hit 16 distinct bank groups (row parity == px parity, PW even)
// while a tap's row offset / a wave's pixel rows become plain byte offsets (instruction immediates).
STORM_HD int patch_off(int pix, int px, int slot) { return pix * PIX_BYTES + ((slot ^ ((px >> 1) & 7)) << 4); }

// MFMA 32x32 accumulator layout (all dtypes): lane holds column (lane & 31) and, for
// register r in [0,16), row  (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
STORM_HD int acc_row(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Operand fragments: lane supplies row/col (lane & 31) and the 16-B slot 2*kgroup + (lane >> 5).
STORM_HD int frag_slot(int lane, int kgroup) { return 2 * kgroup + (lane >> 5); }

// patch pixel index read by `lane` for pixel-tile row `trow` (0..7) under tap (dy,dx)
template <int TAPS> STORM_HD int patch_pixel(int lane, int trow, int dy, int dx) {
    return (trow + dy) * Geo<TAPS>::PW + (lane & 31) + dx;
}

// Epilogue staging: fp32 [64 rows][WM*32 couts] per wave, 16-B slots XOR-swizzled by row.
template <int WM> STORM_HD int stage_off(int row, int slot) {
    return row * (WM * 128) + ((slot ^ (row & (WM * 8 - 1))) << 4);
}
// slot written by `lane` for cout-tile mi, register group g (registers 4g..4g+3 = 4 consecutive couts)
STORM_HD int stage_wslot(int lane, int mi, int g) { return mi * 8 + 2 * g + (lane >> 5); }

// Block id -> (tile, cout tile).  Each XCD (block id & 7, observed dispatch rule) owns a
// contiguous range of pixel tiles so halo rows / both cout halves share its private L2.
struct BlockMap { int tile, ct; };
STORM_HD BlockMap block_map(int bid, int n_ct, int tiles_per_xcd) {
    const int xcd = bid & 7, k = bid >> 3;
    BlockMap m; m.ct = k % n_ct; m.tile = xcd * tiles_per_xcd + k / n_ct; return m;
}

}}  // namespace storm::cidx
