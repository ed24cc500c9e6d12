/*
 * libstorm_hip — C ABI of the MI355X-native (gfx950) StoRM reverse-SDE sampling engine.
 *
 * Drop-in boundary for the hot path  ScoreModel.enhance -> get_pc_sampler ->
 * predictor/corrector -> OUVE SDE -> NCSN++ forward  (+ STFT/iSTFT front/back end).
 * The reference (sp-uhh/storm) has no FFI for this path except one pybind op
 * (sgmse/backbones/ncsnpp_utils/op/upfirdn2d.cpp:12-22); its extension points are
 * Python.  Each entry point below therefore names the reference
 * Python/ATen/CUDA interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *  - every function returns 0 (STORM_OK) or a negative error code; the message is
 *    available from storm_last_error() (thread-local);
 *  - nothing allocates: the caller owns every buffer (activations, packed weights,
 *    workspace); all pointers are DEVICE pointers unless a parameter says "host";
 *  - kernels are enqueued on the given HIP stream (pass torch's current stream),
 *    no host synchronisation inside (mirrors upfirdn2d_kernel.cu:213-215);
 *  - activations are NHWC ("[B][F][T][C]", C a multiple of 8) in fp32 or bf16
 *    (dtype argument); the complex spectrogram [B,1,F,T] complex64 of the reference
 *    is bit-identical to NHWC fp32 with C=2 (re, im interleaved);
 *  - thread-compatible: calls from different host threads must use different streams
 *    and buffers.
 */
#ifndef STORM_HIP_H
#define STORM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* storm_stream_t;              /* hipStream_t */

enum { STORM_OK = 0, STORM_ERR_INVALID = -1, STORM_ERR_HIP = -2, STORM_ERR_UNSUPPORTED = -3 };
enum { STORM_F32 = 0, STORM_BF16 = 1, STORM_F16 = 2 };    /* activation / operand dtype (fp32 accumulation always) */

const char* storm_last_error(void);
/* Bumped whenever a public struct changes size or layout (2: storm_conv_args gained splitk_ws / splitk_ws_bytes, storm_op 13 pointer
 * slots).  A host compiled against another header must not call in: check storm_abi_version() == STORM_ABI_VERSION and
 * storm_abi_struct_bytes(0 / 1) == sizeof(storm_conv_args) / sizeof(storm_op) once at load time.  Callers zero-initialise
 * storm_conv_args (memset) before filling it: every optional pointer is "absent" as NULL. */
#define STORM_ABI_VERSION 2
int storm_abi_version(void);
long long storm_abi_struct_bytes(int which);   /* 0: storm_conv_args, 1: storm_op, 2: storm_conv_seg, 3: storm_ncsnpp_config */
/* Test / tool hook, not part of the drop-in surface: the launchers' A/B switches (forced kernel family, pretend-small device for
 * persistent tile walks, ...) are a table filled once from the environment variables of the same names when the library is
 * first used; these two calls read / change an entry afterwards (names: STORM_CONV_VARIANT, STORM_CONV_PIPE128,
 * STORM_CONV_CUS; profiling build also STORM_CONV_PERSIST / _DMA / _ABLATE / _TRACE_PTR).  Production code
 * never calls them and a launch never reads the environment - with ONE exception that is a serving mode, not a test hook:
 * STORM_BATCH_INVARIANT = 1 takes every launch decision that changes a summation order (conv tile by rounds of workgroups,
 * small-call K split, key ranges of the attention) for one image whatever the call holds, so that a row's result does not
 * depend - bit for bit - on what it is batched with (storm_amd.set_batch_invariant). */
int storm_set_switch(const char* name, long long value);
long long storm_get_switch(const char* name);
/* device name / CU count of the current device, for logs; host out-buffers */
int storm_device_info(char* name, int name_len, int* n_cu, size_t* hbm_bytes);

/* ------------------------------------------------------------------------------------------
 * Weight repacking (device -> device).  Replaces nothing in the reference: it is the
 * engine's own weight layout, built once per model from the reference's state_dict tensors.
 *   conv:   src fp32 [Cout][Cin][KH][KW] (nn.Conv2d.weight, layers.py:100-126)
 *           -> dst [KH*KW][CoutP][CinP] (ci contiguous), zero padded
 *   matrix: src fp32 [rows][cols]; transpose!=0 reads src as [cols][rows]
 *           (layers.NIN.W is [Cin][Cout], layers.py:548-557) -> dst [CoutP][CinP]
 * ------------------------------------------------------------------------------------------ */
int storm_pack_conv_weight(const float* src, void* dst, int Cout, int Cin, int ntaps,
                           int CoutP, int CinP, int dtype, storm_stream_t s);
int storm_pack_matrix(const float* src, void* dst, int Cout, int Cin, int transpose,
                      int CoutP, int CinP, int dtype, storm_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the matrix cores (MFMA 32x32x16 bf16 / 32x32x2 f32).
 * Replaces nn.Conv2d 3x3/1x1 (layers.py:100-126, called from layerspp.py:225-235,260,266,269,
 * ncsnpp.py:183,240,252,108), layers.NIN (layers.py:548-557) and the attention einsums
 * (layerspp.py:82,86) — all are "NT" contractions  out[pix][co] = sum_k X[pix][k] W[co][k].
 *
 * Up to two K-segments accumulate into the same output tile (e.g. Conv_1 3x3 over h plus the
 * Conv_2 1x1 shortcut over x of a BigGAN block, layerspp.py:266-274); a segment may read the
 * channel-concatenation of two tensors (torch.cat([h, hs.pop()], 1), ncsnpp.py:381).
 * Epilogue: out = (acc + bias[co] + tbias[b][co] + skip[pix][co]) * scale.
 * ------------------------------------------------------------------------------------------ */
typedef struct storm_conv_seg {
    const void* src_a;          /* NHWC [B][H][W][Ca]                                   */
    const void* src_b;          /* optional second tensor [B][H][W][Cb] (concat) or NULL  */
    int Ca, Cb;
    long long bstride_a;        /* elements between batches of src_a (H*W*Ca)             */
    long long bstride_b;
    const void* w;              /* packed [ntaps][w_rows_padded][CinP]                    */
    int CinP;                   /* row stride of w (elements)                             */
    int w_rows;                 /* rows of w that may be read (>= Cout valid)             */
    int ntaps;                  /* 9 (3x3, pad 1) or 1                                    */
    long long w_bstride;        /* elements between per-batch weight matrices, 0 = shared */
    long long w_tapstride;      /* elements between taps of w                             */
    const float* gn_ss;         /* optional: fused GroupNorm apply on load: the fp32 (scale, shift) table storm_gn_finalize_ss
                                   writes, [B][(Ca+Cb)/8][2][8] (per 8 channels: their 8 scales, then their 8 shifts);
                                   the conv consumes act(x*scale+shift) (zero padded) instead of x   */
    int gn_silu;                /* apply SiLU after the affine                            */
} storm_conv_seg;

typedef struct storm_conv_args {
    storm_conv_seg seg[2];
    int nseg;
    int B, H, W;                /* 1x1-only calls may pass H=1, W=npix                    */
    void* out;                  /* NHWC [B][H][W][outC]                                   */
    int outC;                   /* channel count of out (multiple of 8)                   */
    int Cout;                   /* valid output channels (<= outC); the rest are written 0+skip */
    long long out_bstride;
    const float* bias;          /* [Cout] or NULL                                         */
    const float* tbias;         /* [B][tbias_stride] per-batch per-channel bias or NULL   */
    int tbias_stride;
    const void* skip;           /* NHWC like out (same dtype as activations) or NULL      */
    long long skip_bstride;
    float scale;
    int out_f32;                /* !=0: write fp32 whatever the operand dtype (scores)    */
    int dtype;                  /* STORM_F32 / STORM_BF16 operands + activations          */
    float* gn_part;             /* optional fused GroupNorm statistics: per-tile (sum, sumsq) of
                                   the output, fp32 [B][storm_conv_tiles()][outC][2]; NULL = off */
    void* splitk_ws;            /* optional scratch of storm_conv_splitk_bytes() bytes (16-byte aligned): lets a 3x3 layer with
                                   so few pixel tiles that most CUs would idle split its K loop over workgroups (fp32 slabs,
                                   summed in a fixed order: bit-reproducible).  NULL / too small = the unsplit kernel       */
    long long splitk_ws_bytes;
} storm_conv_args;

int storm_conv(const storm_conv_args* a, storm_stream_t s);
/* The same for P problems that are ONE layer (same weights, channels, taps; own tensors, batch sizes and widths) in ONE launch - the
 * ragged micro-batches of a stream (BASELINE.json configs[4]).  16-bit 3x3 convolutions of the pipelined kernel only
 * (STORM_ERR_UNSUPPORTED otherwise: run them one by one).  A pixel tile computes exactly what storm_conv computes for it with the
 * same tile (bn = 256 or 128 output channels per workgroup; 0 = chosen by the group's tile count); K is never split.  blob: device
 * scratch >= storm_conv_group_blob_bytes (tables of the launch; this convenience entry fills it with a synchronous copy). */
long long storm_conv_group_blob_bytes(const storm_conv_args* a, int P);
int storm_conv_group(const storm_conv_args* a, int P, void* blob, long long blob_bytes, int bn, storm_stream_t s);
/* scratch bytes with which storm_conv would split K for this call; 0 = it would not (most layers).  No counterpart in the
 * reference: a scheduling aid for ddpm_conv3x3 (layers.py:119-126) on the 16 x 64 ... 4 x 16 pixel levels of ncsnpplarge
 * (ncsnpp.py:460-470), where one launch is otherwise a serial K loop on a few workgroups. */
long long storm_conv_splitk_bytes(const storm_conv_args* a);
/* number of pixel tiles per batch item the kernel will use for this call (size of gn_part) */
int storm_conv_tiles(const storm_conv_args* a);
/* name (as rocprofv3 prints it) of the kernel storm_conv launches for these arguments: lets a profiler or
 * bench.py's roofline name the kernel the launcher really picked; static storage */
const char* storm_conv_kernel_name(const storm_conv_args* a);

/* ------------------------------------------------------------------------------------------
 * Fused single-head attention (flash style: online softmax, the [L][L] scores never leave the chip).  Replaces the two
 * einsums and F.softmax of AttnBlockpp.forward (layerspp.py:82-86):
 *   out[b][i][:] = sum_j softmax_j(scale * <q[b][i], k[b][j]>) v[b][j][:] + bias[:]
 * q, k, out: [B][L][C] (C contiguous, = NHWC activations of the NIN projections, layerspp.py:78-80); vT: [B][C][ldv]
 * (v transposed, row stride ldv >= L, a multiple of 8, zero past L); bias = the NIN_2 (v) bias, which passes through the
 * softmax-weighted sum unchanged (rows of the weights sum to one), or NULL.  bf16 / fp16 (P and the output rounded to the
 * operand type) and fp32 (exact-fp32 MFMA: the parity path), C in {32, 64, 128, 256}, any L; other cases return
 * STORM_ERR_UNSUPPORTED (storm_attention_supported tells beforehand) and run as GEMM + storm_softmax_rows.
 * ------------------------------------------------------------------------------------------ */
int storm_attention_supported(int C, int dtype);
int storm_attention(const void* q, const void* k, const void* vT, const float* bias, void* out, int B, int L, int C, int ldv,
                    long long q_bstride, long long k_bstride, long long vT_bstride, long long out_bstride, float scale,
                    int dtype, storm_stream_t s);
/* The same with caller scratch: a call whose query blocks leave most CUs idle (one to four utterances) splits the KEY loop into 2 - 8
 * ranges on as many workgroups and merges them in a second launch (fixed order, no atomics).  storm_attention_scratch_bytes = the scratch
 * that call wants (0: it runs unsplit; fp32 never splits); scratch NULL or too small = unsplit.  The merged result differs from the
 * unsplit one by fp32 rounding of the merge (layerspp.py:82-86 computes one softmax over all keys; so does the merge, exactly, in fp32). */
long long storm_attention_scratch_bytes(int B, int L, int C, int dtype);
int storm_attention_ws(const void* q, const void* k, const void* vT, const float* bias, void* out, int B, int L, int C, int ldv,
                       long long q_bstride, long long k_bstride, long long vT_bstride, long long out_bstride, float scale, int dtype,
                       void* scratch, long long scratch_bytes, storm_stream_t s);
/* The same for P problems of one layer in ONE launch (ragged micro-batches of a stream: their own batch sizes and sequence lengths; 16-bit
 * operands; the key loop is never split: the group fills the chip).  q / k / out: P pointers to contiguous [B_p][L_p][C], vT: [B_p][C][ldv_p].
 * A query block computes what the unsplit storm_attention computes for it.  blob: device scratch >= storm_attention_group_blob_bytes. */
long long storm_attention_group_blob_bytes(const int* B, const int* L, int P);
int storm_attention_group(const void* const* q, const void* const* k, const void* const* vT, void* const* out, const int* B, const int* L,
                          const int* ldv, int P, const float* bias, int C, float scale, int dtype, void* blob, long long blob_bytes,
                          storm_stream_t s);

/* ------------------------------------------------------------------------------------------
 * GroupNorm(min(C/4,32) groups, eps) [+ SiLU] [+ FIR x2 up / down of BOTH the activated and
 * the raw tensor].  Replaces nn.GroupNorm(eps=1e-6) + nn.SiLU (layerspp.py:219,231,243,264;
 * ncsnpp.py:238,250,392,406; layerspp.py:67,77) and, fused, upsample_2d/downsample_2d on h
 * and x inside ResnetBlockBigGANpp.forward (layerspp.py:245-255).
 * The input may be the channel concat of two tensors.  stats: [B][G][2] doubles (sum, sumsq),
 * must be zero before storm_gn_stats.
 * ------------------------------------------------------------------------------------------ */
int storm_gn_stats(const void* xa, int Ca, const void* xb, int Cb, int B, int HW,
                   int groups, double* stats, int dtype, storm_stream_t s);
/* stats from the per-tile partials a producing storm_conv left in gn_part (no pass over the
 * tensor): channel c < Ca comes from part_a [B][tiles_a][Ca][2], else part_b [B][tiles_b][Cb][2]. */
int storm_gn_finalize(const float* part_a, int Ca, int tiles_a, const float* part_b, int Cb, int tiles_b,
                      int B, int groups, double* stats, storm_stream_t s);
/* Same, and also the per-channel affine of the normalisation for consumers that fuse the apply:
 * scale[c] = rstd*gamma[c], shift[c] = beta[c] - mean*rstd*gamma[c], stored as ss [B][C/8][2][8] (the 8 scales of a channel
 * octet, then its 8 shifts: the order the kernels' packed fp32 math reads them in); count = elements per channel (H*W). */
int storm_gn_finalize_ss(const float* part_a, int Ca, int tiles_a, const float* part_b, int Cb, int tiles_b,
                         int B, int groups, long long count, const float* gamma, const float* beta, float eps,
                         double* stats, float* ss, storm_stream_t s);
/* resample: 0 none, 1 FIR up x2, 2 FIR down x2.  out_act gets act(GN(x)) (resampled),
 * out_raw (may be NULL; required non-NULL only if wanted) gets the resampled raw concat. */
int storm_gn_apply(const void* xa, int Ca, const void* xb, int Cb, int B, int H, int W,
                   int groups, const double* stats, const float* gamma, const float* beta,
                   float eps, int silu, int resample, void* out_act, void* out_raw,
                   int dtype, storm_stream_t s);
/* name (as rocprofv3 prints it, without the argument list) of the kernel storm_gn_apply launches for these arguments (C = Ca + Cb):
 * lets a profiler / bench.py's per-op table name what the launcher really picked; static storage */
const char* storm_gn_apply_kernel_name(int C, int B, int H, int W, int silu, int resample, int dtype);

/* FIR resampling with taps [1,3,3,1] (zero boundary), optional "+ add" on the output.
 * Replaces upsample_2d / downsample_2d -> upfirdn2d (up_or_down_sampling.py:195-257,
 * op/upfirdn2d.py:145-156, op/upfirdn2d_kernel.cu:107-207 modes 3 and 5).               */
int storm_fir_up2(const void* x, const void* add, void* out, int B, int H, int W, int C,
                  int dtype, storm_stream_t s);
int storm_fir_down2(const void* x, void* out, int B, int H, int W, int C,
                    int dtype, storm_stream_t s);
/* The reference's one native-op ABI with its own argument list:
 *   upfirdn2d(input[N,H,W,1], kernel[kh,kw], up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1) -> [N,outH,outW,1]
 * (op/upfirdn2d.cpp:12-22 -> upfirdn2d_op, op/upfirdn2d_kernel.cu:209-369; the CPU form is upfirdn2d_native, op/upfirdn2d.py:159-200).
 * input / out: N contiguous planes (the reference reshapes [B,C,H,W] to [B*C,H,W,1]: minor = 1) of `dtype`; kernel: fp32 device taps
 * (the reference always builds it as fp32, up_or_down_sampling.py:223,256); any up / down factors, pads (negative = crop) and kernel
 * size.  out must hold N * outH * outW elements with outH / outW = storm_upfirdn2d_out_size(...) (the caller allocates: at::empty in
 * upfirdn2d_kernel.cu:242-243).  Errors as everywhere: negative code + storm_last_error (TORCH_CHECK in the reference). */
int storm_upfirdn2d(const void* input, const float* kernel, void* out, int N, int H, int W, int kh, int kw,
                    int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                    int dtype, storm_stream_t s);
/* (in * up + pad0 + pad1 - ktaps) / down + 1 (upfirdn2d_kernel.cu:226-227); host arithmetic, no device work */
long long storm_upfirdn2d_out_size(int in, int up, int down, int pad0, int pad1, int ktaps);

/* Row softmax of fp32 scores [rows][ld] (first L columns valid) -> probabilities (activation
 * dtype, same row stride, padding columns written as 0).
 * Replaces F.softmax(w, dim=-1) in AttnBlockpp.forward (layerspp.py:84).                 */
int storm_softmax_rows(const float* scores, void* probs, long long rows, int L, int ld,
                       int dtype, storm_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Network input / time embedding / output head.
 * ------------------------------------------------------------------------------------------ */
/* complex [B,F,T] tensors (n_in of them) -> NHWC [B][F][T][8] = 2*(re,im..)-1, zero padded.
 * Replaces the real/imag channel packing + "x = 2x - 1" (ncsnpp.py:289-296, 321-323).     */
int storm_pack_input(const float* const* cplx_in /* host array of n_in device ptrs */, int n_in,
                     void* out, int B, int F, int T, int dtype, storm_stream_t s);
/* t[B] -> act_temb[B][4nf] = SiLU(Linear2(SiLU(Linear1(GFP(log t)))))
 * Replaces GaussianFourierProjection + the 2-layer MLP (layerspp.py:39-41, ncsnpp.py:298-317)
 * plus the SiLU that every block applies to temb (layerspp.py:263).                        */
int storm_time_embedding(const float* t, const float* gfp_W, const float* W1, const float* b1,
                         const float* W2, const float* b2, float* act_temb, int B, int nf,
                         storm_stream_t s);
/* out[B][N] = W[N][K] . act_temb[b] + bias[N]  — all blocks' Dense_0 at once (layerspp.py:262-263) */
int storm_dense(const float* x, const float* W, const float* bias, float* out, int B, int N, int K,
                storm_stream_t s);
/* pyramid NHWC [B][F][T][8] -> complex [B,F,T]:  sign * (W[2][cin] . (p / t_b) + b)
 * Replaces "h / used_sigmas", output_layer and view_as_complex (ncsnpp.py:441-449) and the
 * negation in ScoreModel.forward (model.py:131-132) when negate!=0.                        */
int storm_output_head(const void* pyr, const float* t /* NULL: no division */, const float* W,
                      const float* bias, int cin, float* out_cplx, int B, int F, int T,
                      int negate, int dtype, storm_stream_t s);
/* The progressive input pyramid in ONE launch (csrc/pyramid.hip): levels[0] = the packed inputs (exactly storm_pack_input),
 * levels[k] = FIR x2 down of levels[k - 1] (exactly storm_fir_down2), k < n_levels <= 4; levels[k]: NHWC [B][F >> k][T >> k][8].
 * cplx_in == NULL: levels[0] is READ instead of written (the continuation of a pyramid deeper than three steps: ncsnpplarge has six).
 * Replaces the packing above and pyramid_downsample before every Combine (ncsnpp.py:352-355; up_or_down_sampling.py:230-257): the input
 * pyramid depends on the network input alone, so it is built once, ahead of the U-Net.                                            */
int storm_input_pyramid(const float* const* cplx_in /* host array of n_in device ptrs, or NULL */, int n_in,
                        void* const* levels /* host array of n_levels device ptrs */, int n_levels, int B, int F, int T,
                        int dtype, storm_stream_t s);
/* The progressive output pyramid and the head in ONE launch: p_{L-1} = ph[L-1]; p_k = up2(p_{k+1}) + ph[k] (exactly storm_fir_up2 with
 * its add operand, every level rounded to the storage type); out = storm_output_head(p_0, ...).  ph[k]: NHWC [B][F >> k][T >> k][8], the
 * outputs of the pyramid's 3x3 convolutions, finest first, n_levels <= 8.  Replaces pyramid_upsample + "pyramid = pyramid + pyramid_h"
 * after every decoder level (ncsnpp.py:389-410) and the head (ncsnpp.py:441-449).                                                   */
int storm_output_pyramid(const void* const* ph /* host array of n_levels device ptrs */, int n_levels, const float* t /* NULL: no division */,
                         const float* W, const float* bias, int cin, float* out_cplx, int B, int F, int T, int negate, int dtype,
                         storm_stream_t s);

/* ------------------------------------------------------------------------------------------
 * OUVE SDE steps on the complex64 state [B, n] (n = F*T complex per batch item).
 * Replace OUVESDE.prior_sampling/_std/sde, SDE.discretize, RSDE.discretize (sdes.py:73-90,
 * 147-157, 200-237), ReverseDiffusionPredictor / EulerMaruyamaPredictor.update_fn
 * (predictors.py:46-69), AnnealedLangevinDynamics / LangevinCorrector.update_fn
 * (correctors.py:45-93).  z == NULL: standard complex normal noise (variance 1/2 per
 * component, like torch.randn_like on a complex tensor) is generated in-kernel with
 * Philox4x32-10 from (seed, offset).
 * ------------------------------------------------------------------------------------------ */
typedef struct storm_ouve { float theta, sigma_min, sigma_max; int N; } storm_ouve;

int storm_ouve_prior(const float* y, const float* z, float* x, int B, long long n, storm_ouve p,
                     uint64_t seed, uint64_t offset, storm_stream_t s);
int storm_ouve_ald_step(float* x, float* x_mean, const float* score, const float* z, const float* t,
                        int B, long long n, storm_ouve p, float snr,
                        uint64_t seed, uint64_t offset, storm_stream_t s);
/* predictor: kind 0 = reverse_diffusion, 1 = euler_maruyama; noise_free!=0 skips "+ G z" */
int storm_ouve_predictor_step(float* x, float* x_mean, const float* score, const float* y,
                              const float* z, const float* t, int B, long long n, storm_ouve p,
                              int kind, int noise_free, uint64_t seed, uint64_t offset,
                              storm_stream_t s);
/* per-batch L2 norms of complex tensors: out[b] = ||v_b||  (correctors.py:53-54) */
int storm_batch_l2norm(const float* v, float* out, int B, long long n, storm_stream_t s);
/* Langevin corrector step; z must be given or generated beforehand (its norm is needed):
 * step = 2 (snr * ||z|| / ||score||)^2 with the norms taken as
 *   mode 0: means over the B rows (correctors.py:53-55 for the batch handed to the sampler),
 *   mode 1: row b's own norms (B independent batch-1 calls: what the reference CLI computes per file),
 *   mode 2: score_norms[0] / z_norms[0] = means over a larger batch (all-reduced over the ranks of a sharded run). */
int storm_langevin_step(float* x, float* x_mean, const float* score, const float* z,
                        const float* score_norms, const float* z_norms, int B, long long n, float snr, int mode,
                        storm_stream_t s);
/* Scale-invariant SDR in dB of B (clean s, estimate s_hat) waveform pairs, fp32 [B][>= n] with row strides (util/other.py:82-94:
 * si_sdr with eps = 0, si_sdr_torch with eps = 1e-10; used by the evaluation loop, util/inference.py:20-72) */
int storm_si_sdr(const float* s, const float* s_hat, float* out, int B, long long n, long long stride_s, long long stride_hat,
                 float eps, storm_stream_t st);
/* drift of the probability-flow ODE, out = theta (y - x) - 1/2 g(t)^2 score  (sdes.py:92-121 with probability_flow, :203-207) */
int storm_ouve_pf_drift(float* out, const float* x, const float* y, const float* score, const float* t, int B,
                        long long n, storm_ouve p, storm_stream_t s);
/* the same with g(t_b) given per row (device fp32 [B]; the ODE sampler computes it with the reference's own fp32 formula,
 * sdes.py:203-207, so the right-hand side matches the reference's to the last bit) */
int storm_ouve_pf_drift_g(float* out, const float* x, const float* y, const float* score, const float* g_rows, int B,
                          long long n, float theta, storm_stream_t s);
/* ---- coefficient-table forms for any SDE  dx = a(t) (y - x) dt + g(t) dw: the second SDE the reference registers, OUVPSDE
 * (sdes.py:255-326: a = 1/2 stiffness beta(t), g = sqrt(beta(t)), beta(t) = beta_min + t (beta_max - beta_min)).  a_rows / g_rows /
 * std_rows = the per-row values at t_b (device fp32 [B]); the host side forms them with the reference's own fp32 expressions
 * (sdes.py:286-301), the kernels do the state-sized work of OUVPSDE.prior_sampling (:306-310), SDE.discretize (:73-90),
 * RSDE.discretize / rsde_parts (:123-157) and the predictors' update_fn (predictors.py:46-69) exactly as the storm_ouve_* ones do.
 * kind / noise_free / z == NULL as in storm_ouve_predictor_step; N = the SDE's discretisation steps (dt = 1 / N). */
int storm_sde_prior_rows(const float* y, const float* z, float* x, const float* std_rows, int B, long long n,
                         uint64_t seed, uint64_t offset, storm_stream_t s);
int storm_sde_predictor_step_rows(float* x, float* x_mean, const float* score, const float* y, const float* z,
                                  const float* a_rows, const float* g_rows, int B, long long n, int N, int kind,
                                  int noise_free, uint64_t seed, uint64_t offset, storm_stream_t s);
/* out = a_b (y - x) - 1/2 g_b^2 score: the probability-flow right-hand side (sdes.py:92-145 with probability_flow=True) */
int storm_sde_pf_drift_rows(float* out, const float* x, const float* y, const float* score, const float* a_rows,
                            const float* g_rows, int B, long long n, storm_stream_t s);
/* ---- probability-flow ODE sampler (sampling/__init__.py:71-141 hands the state to scipy's RK45 on the host) ----
 * out = x + h * sum_{j<n_terms} coef[j] K[j]   (one Runge-Kutta stage; K = host array of device pointers, <= 7) */
int storm_rk_combine(float* out, const float* x, const float* const* K, const float* coef, int n_terms, float h,
                     long long n_complex, storm_stream_t s);
/* out[0] = sum_c |v_c|^2 / (atol + max(|xa_c|, |xb_c|) rtol)^2 over the complex elements (scipy's scaled norms, squared
 * and un-averaged): v = h * sum coef[j] K[j] (n_terms > 0: the embedded error), K[0] (n_terms = -1) or K[0] - K[1]
 * (n_terms = -2).  xb may be NULL.  scratch: >= 2048 doubles; deterministic (fixed summation order). */
int storm_rk_scaled_sumsq(double* out, double* scratch, int scratch_len, const float* xa, const float* xb,
                          const float* const* K, const float* coef, int n_terms, float h, float atol, float rtol,
                          long long n_complex, storm_stream_t s);
/* Per-row forms for a batch of B independent utterances that advance with their OWN step sizes (the reference calls solve_ivp
 * once per utterance: model.py:224-244, minibatch = 1).  scipy keeps its state in complex128 and only the right-hand side
 * runs in fp32 (sampling/__init__.py:119-123): x / out64 / xa / xb are complex128 [B][n_complex_row], the stages K complex64,
 * coefficients and step sizes fp64, so step decisions follow scipy's to 1e-16.  h_rows: HOST array of B step sizes
 * (B <= STORM_RK_MAX_ROWS, passed in the kernel arguments).
 *   combine_rows:      out[b] = x[b] + h_rows[b] * sum_j coef[j] K[j][b]  -> out64 (complex128) and / or out32 (complex64)
 *   scaled_sumsq_rows: out[b] = row b's sum of |v|^2 / (atol + max(|xa|, |xb|) rtol)^2; v as in storm_rk_scaled_sumsq
 *                      (n_terms = -3: v = xa itself); h_rows NULL = 1.  The partial-sum geometry depends on the row length
 *                      only, so a row's value does not depend on the batch around it; scratch: >= STORM_RK_ROW_BLOCKS * B doubles
 *   copy_rows:         dst[b] = src[b] for the rows with row_mask[b] != 0 (HOST mask): the accepted rows of a step */
enum { STORM_RK_MAX_ROWS = 128, STORM_RK_ROW_BLOCKS = 256 };
int storm_rk_combine_rows(double* out64, float* out32, const double* x, const float* const* K, const double* coef, int n_terms,
                          const double* h_rows, int B, long long n_complex_row, storm_stream_t s);
int storm_rk_scaled_sumsq_rows(double* out, double* scratch, long long scratch_len, const double* xa, const double* xb,
                               const float* const* K, const double* coef, int n_terms, const double* h_rows, double atol,
                               double rtol, int B, long long n_complex_row, storm_stream_t s);
int storm_copy_rows(void* dst, const void* src, const int* row_mask, int B, long long row_bytes, storm_stream_t s);
/* fills z[B*n] complex with standard complex normal noise (Philox) */
int storm_complex_randn(float* z, long long n_complex, uint64_t seed, uint64_t offset,
                        storm_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Spectral front / back end.  Replace torch.stft / torch.istft as called by
 * SpecsDataModule.stft/istft (data_module.py:195-223: n_fft, hop, periodic Hann, center=True,
 * reflect padding), spec_fwd / spec_back (data_module.py:182-193), pad_spec (util/other.py:
 * 102-109) and the peak normalisation in enhance() (model.py:282-284, 302).
 * ------------------------------------------------------------------------------------------ */
/* out = spec_fwd(in) (inverse==0) or spec_back(in) (inverse!=0) on n complex64 values
 * (data_module.py:182-193): |z|^e e^{j angle z} * factor  /  its inverse.                  */
int storm_spec_transform(const float* in, float* out, long long n_complex, float spec_factor,
                         float spec_abs_exponent, int inverse, storm_stream_t s);
/* Ragged batches (micro-batching of utterances of different lengths that share ONE padded frame count, i.e.
 * roundup(1 + L_b / hop, 64) equal for all rows): the three front / back end calls take row_len = device int32 [B] with
 * every row's own sample count (NULL = all rows L); wav buffers are [B][L] with L = the longest row, zero filled past a
 * row's length.  Row b then equals the single-utterance call with L_b (the reference pads per utterance,
 * util/other.py:102-109, enhancement.py:66-72).
 * peak[b] = max |wav[b][:L_b]|  (model.py:283) */
int storm_peak_abs(const float* wav, float* peak, int B, long long L, long long stride, const int* row_len,
                   storm_stream_t s);
/* wav [B][L] (row stride `stride`) -> spec complex [B][F][Tpad]; frames >= n_frames are zero.
 * spec = spec_fwd(stft(wav / peak[b])); peak may be NULL (no normalisation).
 * twiddle: fp32 [n_fft][2] = (cos, sin)(2 pi k / n_fft); window: fp32 [n_fft].            */
int storm_stft(const float* wav, const float* peak, float* spec, const float* window,
               const float* twiddle, int B, long long L, long long stride, int n_fft, int hop,
               int n_frames, int Tpad, float spec_factor, float spec_abs_exponent, const int* row_len,
               storm_stream_t s);
/* spec complex [B][F][T] (ALL T frames are inverted, as to_audio does, model.py:258-259,301)
 * -> wav [B][L] = istft(spec_back(spec), length=L) * peak[b].
 * frames: workspace fp32 [B][T][n_fft].                                                    */
int storm_istft(const float* spec, const float* peak, float* wav, float* frames,
                const float* window, const float* twiddle, int B, int T, long long L,
                long long stride, int n_fft, int hop, float spec_factor, float spec_abs_exponent,
                const int* row_len, storm_stream_t s);

/* ------------------------------------------------------------------------------------------
 * Program interpreter: runs a whole NCSN++ forward (or any op list) with one host call.
 * The op list is planned on the host side (storm_amd/backbones/plan.py) once per
 * (model, B, F, T, dtype); pointers are offsets into the caller-owned buffers in `bufs`.
 * Replaces the Python op-by-op dispatch of NCSNpp.forward (ncsnpp.py:281-450).
 * ------------------------------------------------------------------------------------------ */
enum {
    STORM_OP_MEMSET = 0, STORM_OP_PACK_INPUT = 1, STORM_OP_TEMB = 2, STORM_OP_DENSE = 3,
    STORM_OP_CONV = 4, STORM_OP_GN_STATS = 5, STORM_OP_GN_APPLY = 6, STORM_OP_FIR_UP = 7,
    STORM_OP_FIR_DOWN = 8, STORM_OP_SOFTMAX = 9, STORM_OP_OUTPUT_HEAD = 10, STORM_OP_GN_FINALIZE = 11,
    STORM_OP_ATTENTION = 12,
    STORM_OP_INPUT_PYRAMID = 13,    /* p[0..2] complex inputs (or none: level 0 is read), p[3..6] levels; i = n_in, B, F, T, n_levels */
    STORM_OP_OUTPUT_PYRAMID = 14    /* p[0..7] ph (finest first), p[8] t, p[9] W, p[10] bias, p[11] out; i = cin, B, F, T, negate, n_levels */
};
#define STORM_OP_NPTR 13
#define STORM_OP_NINT 24
#define STORM_OP_NFLT 4
typedef struct storm_ref { int32_t buf; int32_t pad_; int64_t off; } storm_ref; /* buf<0: NULL; byte offset */
typedef struct storm_op {
    int32_t code;
    int32_t pad_;
    storm_ref p[STORM_OP_NPTR];
    int64_t i[STORM_OP_NINT];
    float f[STORM_OP_NFLT];
} storm_op;

/* ops: host array; bufs: host array of n_bufs device base pointers.                       */
int storm_program_run(const storm_op* ops, int n_ops, void* const* bufs, int n_bufs, int dtype,
                      storm_stream_t s);
/* Same, but brackets every op with HIP events on stream `s`, synchronises, and writes the
 * elapsed milliseconds of each op to the HOST array ms[n_ops] (bench.py's roofline leg).    */
int storm_program_run_timed(const storm_op* ops, int n_ops, void* const* bufs, int n_bufs,
                            int dtype, storm_stream_t s, float* ms);
/* name of the kernel op k of a program launches (STORM_OP_CONV ops; "" otherwise), see storm_conv_kernel_name */
const char* storm_program_kernel_name(const storm_op* ops, int k, int dtype);

/* ------------------------------------------------------------------------------------------
 * Whole-network entry points: NCSN++ as ONE object of the C ABI (a host in any language builds it from the reference's
 * state_dict tensors and evaluates the score with one call; the Python class storm_amd.backbones.NCSNpp is a thin caller).
 * Replaces NCSNpp.__init__ / load_state_dict (ncsnpp.py:38-273) and NCSNpp.forward (ncsnpp.py:281-450) for the
 * configuration family of the StoRM hot path (BigGAN blocks, FIR resampling, output_skip / input_skip pyramids, Fourier
 * time embedding, swish).
 * ------------------------------------------------------------------------------------------ */
typedef struct storm_ncsnpp_config {
    int nf;                     /* base width (128)                                                  */
    int n_levels;               /* len(ch_mult)                                                      */
    int ch_mult[8];             /* (1, 2, 2, 2); ncsnpplarge (1, 1, 2, 2, 2, 2, 2)                    */
    int num_res_blocks;
    int n_attn;
    int attn_resolutions[4];    /* frequency heights that get an AttnBlockpp ((0,) = bottleneck only)  */
    int image_size;             /* 256                                                               */
    int input_channels;         /* real channels of the score net input: 4 (x, y) or 6 (x, y, y_den)   */
    int discriminative;         /* predictive denoiser: 2 channels, no time conditioning, no 1/t       */
} storm_ncsnpp_config;
typedef struct storm_ncsnpp storm_ncsnpp;

/* the reference state_dict of this configuration: count, then (name, shape) of tensor i in the reference's order */
int storm_ncsnpp_num_tensors(const storm_ncsnpp_config* cfg);
int storm_ncsnpp_tensor_info(const storm_ncsnpp_config* cfg, int i, char* name, int name_len, int* ndim, long long* shape4);
long long storm_ncsnpp_arena_bytes(const storm_ncsnpp_config* cfg, int dtype);
/* weights[i] = device pointer to fp32 tensor i of the state_dict (contiguous).  Packs them (stream s) into the engine's
 * layout: arena = caller-owned device buffer of storm_ncsnpp_arena_bytes() or NULL (the handle allocates one). */
int storm_ncsnpp_create(const storm_ncsnpp_config* cfg, const void* const* weights, int n_weights, int dtype, void* arena,
                        storm_stream_t s, storm_ncsnpp** out);
void storm_ncsnpp_destroy(storm_ncsnpp* h);
/* A/B switches of the planner (all on by default): GroupNorm statistics from conv epilogues, GroupNorm apply in conv operand
 * loads, fused attention kernel */
int storm_ncsnpp_set_fusion(storm_ncsnpp* h, int fuse_stats, int fuse_apply, int fused_attention);
/* HIP-graph replay of this handle's evaluations (the reference's own operating point is ONE utterance per
 * call, enhancement.py:66-72, where an evaluation is ~120 short launches): mode 0 = eager launches, 1 = replay, -1 (default) = the
 * library's rule (today: eager - measured on MI355X the launch loop keeps the queue full at every batch size, profiles/r05a_*).  The first call per (shape, workspace address) runs eagerly, the second records the runs
 * of ops that touch only the workspace and the weights, later calls are one hipGraphLaunch per run + the three ops that read the
 * caller's tensors (input packing, time embedding, output head).  A replayed evaluation executes exactly the eager kernels with the
 * eager arguments: results are bit-identical.  A workspace passed to a graph-mode call must stay allocated while the handle lives
 * or until the same address is passed again (the recorded kernels point into it). */
int storm_ncsnpp_set_graph(storm_ncsnpp* h, int mode);
long long storm_ncsnpp_graph_launches(storm_ncsnpp* h);    /* hipGraphLaunch calls this handle has made (diagnostics: 0 = everything ran eagerly) */
/* bytes of scratch one forward at (B, F, T) needs (liveness-planned; 5.2 GB at B = 16, 256 x 512, bf16); -1 on error */
long long storm_ncsnpp_workspace_bytes(storm_ncsnpp* h, int B, int F, int T);
/* out[b] = dnn(cat[parts...], t) (negate != 0: its negative = the score, model.py:131-132).  parts: n device pointers to
 * complex64 [B][F][T]; t fp32 [B] (NULL when discriminative); out complex64 [B][F][T]; ws: >= workspace_bytes, pure scratch
 * (nothing is carried from one call to the next: one buffer sized for the largest shape serves every shape).
 * Threading: a handle may be shared by host threads that call with DIFFERENT streams and workspaces - the per-shape plan
 * cache is locked and bounded (64 shapes, least recently used out), `negate` is a per-call argument, the weights are read
 * only; storm_ncsnpp_set_fusion / _destroy must not race with calls. */
int storm_ncsnpp_forward(storm_ncsnpp* h, const void* const* parts, int n_parts, const float* t, void* out, void* ws,
                         long long ws_bytes, int B, int F, int T, int negate, storm_stream_t s);
/* GROUPED evaluation - P micro-batches of different (B_p, T_p) of one stream in ONE call (BASELINE.json configs[4]: 2 - 10 s utterances
 * bucketed by padded frame count are 2 - 3 rows per micro-batch; the reference processes one file per call, enhancement.py:66-72).  All
 * problems run the same op sequence; op k of every problem shares one launch where a grouped kernel exists (today: the 16-bit 3x3
 * convolutions with > 128 output channels - one persistent walk over all problems' pixel tiles), and runs problem by problem otherwise.
 * Every row computes what its own micro-batch's storm_ncsnpp_forward computes in the kernels that serve it (the tile choice and the K split
 * of few-tile layers follow the GROUP's tile count, so a 16-bit row agrees with its own call to the rounding of its activations; fp32: no
 * grouped kernel, bit-identical).  parts: P * n_parts pointers, problem-major; t / out: P pointers; ws >= _group_workspace_bytes(same list). */
long long storm_ncsnpp_group_workspace_bytes(storm_ncsnpp* h, int P, const int* B, const int* T, int F);
int storm_ncsnpp_forward_group(storm_ncsnpp* h, int P, const int* B, const int* T, int F, const void* const* parts, int n_parts,
                               const float* const* t, void* const* out, void* ws, long long ws_bytes, int negate, storm_stream_t s);
long long storm_ncsnpp_group_launches(storm_ncsnpp* h);    /* grouped kernel launches this handle has made (diagnostics: 0 = every op ran problem by problem) */
/* the planned op list of (B, F, T) (owned by the handle) for storm_program_run_timed / storm_program_kernel_name, the packed
 * arena, and the algorithmic FLOPs of one forward */
int storm_ncsnpp_program(storm_ncsnpp* h, int B, int F, int T, const storm_op** ops, int* n_ops, long long* flops);
/* a program handed out above stays valid (pinned) until the handle is destroyed or the caller releases it: */
int storm_ncsnpp_release_program(storm_ncsnpp* h, const storm_op* ops);
const void* storm_ncsnpp_arena(storm_ncsnpp* h);

#ifdef __cplusplus
}
#endif
#endif /* STORM_HIP_H */
