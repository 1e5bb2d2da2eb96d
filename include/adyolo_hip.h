/*
 * adyolo_hip.h -- C ABI of libadyolo_hip.so: the MI355X (gfx950) hot path of AD-YOLO.
 *
 * The reference (sadPororo/AD-YOLO) has NO FFI: its plugin boundary is Python class dispatch
 * (src/wrapper.py:26-50, :70-85) and every arithmetic step is an implicit ATen / NumPy / librosa
 * call.  This header is therefore the boundary the build defines: each entry point replaces the
 * implicit library call(s) cited next to it.  All pointers are DEVICE pointers (HBM) unless a name
 * ends in _host; sizes are plain ints; `stream` is a hipStream_t passed as void* (NULL = default
 * stream).  Activations are float32, channels-last:  x[n][h][w][c]  with h = time frame, w = mel bin.
 * Every function returns 0 on success, a negative ADYOLO_E* code on bad arguments, or a positive
 * hipError_t; adyolo_last_error() returns a static description of the last failure on this thread.
 * No function allocates device memory or synchronises the device: workspaces are passed in.
 */
#ifndef ADYOLO_HIP_H
#define ADYOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADYOLO_ABI_VERSION 1
#define ADYOLO_EINVAL (-1)   /* bad shape / alignment / null pointer */
#define ADYOLO_ENOSUP (-2)   /* shape outside what the kernels are built for */

int         adyolo_abi_version(void);
const char *adyolo_last_error(void);
/* number of float32 words a workspace must hold, for the calls that take one */

/* ------------------------------------------------------------------------------------------------
 * K1  feature extraction: STFT(n_fft=1200, hop=600, periodic Hann, reflect-centre) -> log-mel (4 ch)
 *     + mel-scale FOA intensity vector (3 ch) -> z-score.
 *     replaces src/datasets.py:252-292 (librosa.core.stft :255, mel GEMM :264/:275, power_to_db :265,
 *     scaler :289-290) and the tensorise/concat step :158-160.
 *   audio   [B][n_samples][4] float32, already int16/32768 + 1e-8 (datasets.py:147), channels W,Y,Z,X
 *   clip_offset  NULL, or [B] int64 sample offsets into `audio`: "virtual clip" b is then the n_samples samples starting
 *           at clip_offset[b] -- the 20 s / 1 s-stride training chunks of src/preprocess.py:13-84 (chunking), computed
 *           from the whole recording in place; every virtual clip gets its own reflect padding (np.pad at the chunk
 *           start, preprocess.py writes the chunk as its own file) and its own top_db reference (chan_max row b)
 *   twiddle [2388][2]  rows 0..1199: tw[n] = exp(-2 pi i n/1200) (re,im) (the periodic Hann window is generated in the
 *           kernel); rows 1200 + (k-1) 120 + s = tw[(s k) mod 1200] and rows 2280 + (k-1) 12 + j = tw[10 j k] for k = 1..9,
 *           s < 120, j < 12: the same entries in the order the lanes of transform passes 1 and 2 read them
 *   mel_w   [n_mel_w <= 1200] float32 non-zero weights of the 64 (contiguous, triangular) mel filters, filter
 *           after filter; the filters are cut into n_chunks (<= 224) pieces of a few bins for load balance:
 *           chunk_mel[i] = filter index (non-decreasing, every filter 0..63 present), chunk_start[i] = first FFT bin, chunk_len[i], chunk_off[i] = offset in mel_w
 *   scaler_mean/scaler_rstd [7][64]: (x-mean)*rstd per (feature channel, mel bin)
 *   out     layout 0: [B][7][T][64]  (reference order, datasets.py:160)
 *           layout 1: [B][T][64][8]  (channels-last, 8th channel zero) -- what the encoder consumes
 *   chan_max [B][4] float32 workspace (per clip/channel max of the un-clipped log-mel, for top_db=80)
 *   T = n_samples / 600 (n_samples must be a multiple of 600).
 * Two launches: adyolo_feat_stft_mel (writes IV final, log-mel un-clipped + chan_max) then
 * adyolo_feat_finish (top_db clip relative to chan_max + z-score of the 4 log-mel channels).
 * ---------------------------------------------------------------------------------------------- */
int adyolo_feat_stft_mel(const float *audio, const int64_t *clip_offset, const float *twiddle,
                         const int32_t *chunk_mel, const int32_t *chunk_start, const int32_t *chunk_len,
                         const int32_t *chunk_off, const float *mel_w, int n_chunks, int n_mel_w,
                         const float *scaler_mean, const float *scaler_rstd, float *out, float *chan_max, int B,
                         int n_samples, int layout, void *stream);
int adyolo_feat_finish(float *out, const float *chan_max, const float *scaler_mean,
                       const float *scaler_rstd, int B, int T, int layout, void *stream);

/* K1m  GCC-PHAT features of 4-channel MIC-format audio (BASELINE config 5: "DCASE2022 MIC (GCC-PHAT features)").  NOT in
 *     the reference (FOA is hard-coded: src/datasets.py:36-37,55; the --feature switch is commented out, src/main.py:40):
 *     PARITY UNPINNED.  Definition: the DCASE2022 SELD baseline's (the repository README.md:156 credits for the metrics):
 *     per microphone pair m < n, R = conj(X_m) X_n, cc = irfft(exp(i angle(R))), feature = concat(cc[-32:], cc[:32]);
 *     X = the STFT of K1.  The four log-mel channels of the MIC feature set are adyolo_feat_stft_mel on the same audio.
 *   audio [B][n_samples][4] float32; out: channels-last pixels of pix_stride floats, [B][T][64 lag bins][pix_stride];
 *   the six pair channels (0,1) (0,2) (0,3) (1,2) (1,3) (2,3) are written at ch0 .. ch0+5 as (cc - mean) * rstd with
 *   scaler_mean / scaler_rstd [6][64]. */
int adyolo_feat_gcc_phat(const float *audio, const int64_t *clip_offset /*or NULL*/, const float *twiddle,
                         const float *scaler_mean, const float *scaler_rstd, float *out, int B, int n_samples,
                         int pix_stride, int ch0, void *stream);
/* [B][C][H][W] (C<=8) -> [B][H][W][8] zero padded; entry of WrapperModel.forward (wrapper.py:52-57) */
int adyolo_nchw_to_nhwc8(const float *x, float *y, int B, int C, int H, int W, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K2  3x3 convolution, stride 1, pad 1, as an fp32-MFMA implicit GEMM (v_mfma_f32_32x32x2_f32).
 *     replaces nn.Conv2d forward/backward at src/models/backbones/resnet.py:16,18,142.
 *   x [N][H][W][Cin]; y [N][H][W][Cout];  Cin in {8, multiples of 32}, Cout multiple of 32.
 *   wpk   packed weights [Cout][9][Cin]  (tap = ky*3+kx, Cin fastest) -- see adyolo_pack_w3x3
 *   bias  [Cout] or NULL;  addend [N][H][W][Cout] or NULL (added before the optional ReLU)
 *   y = relu?( conv(x', w) + bias + addend' )   with the optional fusions
 *     x' = x*in_scale[c] + in_shift[c] on in-image pixels (BatchNorm affine of the producer; zero padding stays 0)
 *     addend' = addend * (addend_mask > 0)  (residual gradient  de * (e > 0)  formed on the fly)
 *     mask_bits: bit 0 -- addend_mask points to ReLU-mask BITS (the uint64 words adyolo_se_tail_fwd writes, see K3b)
 *       instead of a float tensor; bit 1 -- the same for stat_mask.  Bits move 1/32 of the bytes.
 *     stats [2][tiles][Cout]: per 256-pixel patch, per channel sum of y and either sum of y^2 (stat_aux == NULL: the
 *       BatchNorm statistics / SE squeeze of the consumer, finished by adyolo_bn_stats_tiles) or sum of
 *       y * (stat_aux - stat_mean[c]) * stat_invstd[c] (y is a gradient, stat_aux the BatchNorm input at the same
 *       positions: the two sums of BatchNorm's backward, finished by adyolo_bn_bwd_tiles) -- no separate read pass.
 * data-gradient = the same call with the dgrad packing (wpk_dgrad, Cin<->Cout swapped).
 * weight-gradient: adyolo_conv3x3_wgrad accumulates into `slabs` ([n_slabs][Cout][9][CinP] float32,
 * n_slabs = adyolo_conv3x3_wgrad_slabs(...)) then reduces them into dw in the reference layout
 * [Cout][Cin_real][3][3] (Cin_real <= Cin: 7 for the stem whose activations are padded to 8).
 * ---------------------------------------------------------------------------------------------- */
int adyolo_pack_w3x3(const float *w /*[Cout][Cin_real][3][3]*/, float *wpk_fwd /*[Cout][9][Cin]*/,
                     float *wpk_dgrad /*[Cin][9][Cout] or NULL*/, int Cout, int Cin_real, int Cin,
                     void *stream);
int adyolo_conv3x3_tiles(int N, int H, int W);   /* number of 256-pixel patches = rows of `stats` */
int adyolo_conv3x3_fwd(const float *x, const float *wpk, const float *bias, const float *addend,
                       const float *addend_mask, const float *in_scale, const float *in_shift, float *y,
                       float *stats, const float *stat_aux, const float *stat_mean, const float *stat_invstd,
                       const float *stat_mask, int N, int H, int W, int Cin, int Cout, int relu, int mask_bits, void *stream);
/*   stat_mask (optional, needs stats): the two per-patch sums are taken of y * (stat_mask > 0) instead of y -- with
 *   stat_aux = c and stat_mask = e of the block BELOW, the data-gradient launch that produces de also produces the
 *   per-sample sums of the SE / BatchNorm-2 backward (finished by adyolo_se_tail_bwd_tiles), y itself is unaffected. */
/* K2w  the same operator as Winograd F(2x2,3x3) (2.25x fewer matrix FLOPs; fp32 throughout, results agree with the
 *      direct form to ~1e-6 relative).  u_fwd / u_dgrad: the transformed filters G g G^T in MFMA-fragment order
 *      [16][Cout/32][Cin/8][64][4] (forward) / [16][Cin/32][Cout/8][64][4] (data-gradient, taps flipped, channels
 *      transposed); Cin and Cout multiples of 32.  adyolo_wino_fwd takes the arguments of adyolo_conv3x3_fwd with
 *      `u` in place of `wpk`; its `stats` rows are 8x16-pixel patches: adyolo_wino_tiles(N,H,W) of them. */
int adyolo_wino_pack_w(const float *w /*[Cout][Cin_real][3][3]*/, float *u_fwd /*or NULL*/, float *u_dgrad /*or NULL*/,
                       int Cout, int Cin_real, int Cin, void *stream);
/* the same for EVERY 3x3 filter of a model in one launch (the packed filters change once per optimizer step, not per layer
 * call: 64 pack launches per SE-ResNet34 train step become one).  table (device) = n rows of 8 int64:
 * {w, u_fwd, u_dgrad or 0, Cout, Cin_real, Cin, unused, unused}; max_cout / max_cin = the largest channel counts in the table. */
int adyolo_wino_pack_many(const int64_t *table, int n, int max_cout, int max_cin, void *stream);
int adyolo_wino_tiles(int N, int H, int W);
int adyolo_wino_fwd(const float *x, const float *u, const float *bias, const float *addend,
                    const float *addend_mask, const float *in_scale, const float *in_shift, float *y, float *stats,
                    const float *stat_aux, const float *stat_mean, const float *stat_invstd, const float *stat_mask,
                    int N, int H, int W, int Cin, int Cout, int relu, int mask_bits, void *stream);
/* K2w4  the same operator as Winograd F(4x4,3x3) (csrc/wino4.hip; round 4): 36 multiplies per 4x4 output tile and channel pair,
 *      1.78x fewer matrix instructions than K2w, still on the exact-fp32 MFMA.  Interpolation points 0, +-3/4, +-3/2, infinity;
 *      error against a float64 convolution ~1e-6 of the output's absmax (K2w: 2e-7).  Replaces nn.Conv2d forward / data-gradient
 *      at src/models/backbones/resnet.py:16,18 for Cout a multiple of 64 and Cin a multiple of 32 (<= 512).
 *      u_fwd / u_dgrad: G g G^T in MFMA-fragment order [36][Cout/32][Cin/8][64][4] (forward) / [36][Cin/32][Cout/8][64][4]
 *      (data-gradient, taps flipped, channels transposed); position order: see csrc/wino4.hip.  adyolo_wino4_fwd takes the
 *      arguments of adyolo_conv3x3_fwd with `u` in place of `wpk`; its `stats` rows are 32x16-pixel (maps narrower than 32
 *      pixels) or 16x32-pixel patches: adyolo_wino4_tiles(N,H,W) of them.  adyolo_wino4_pack_many: the table of
 *      adyolo_wino_pack_many (columns 6, 7 unused). */
int adyolo_wino4_pack_w(const float *w /*[Cout][Cin_real][3][3]*/, float *u_fwd /*or NULL*/, float *u_dgrad /*or NULL*/,
                        int Cout, int Cin_real, int Cin, void *stream);
int adyolo_wino4_pack_many(const int64_t *table, int n, int max_cout, int max_cin, void *stream);
int adyolo_wino4_tiles(int N, int H, int W);
int adyolo_wino4_fwd(const float *x, const float *u, const float *bias, const float *addend, const float *addend_mask,
                     const float *in_scale, const float *in_shift, float *y, float *stats, const float *stat_aux,
                     const float *stat_mean, const float *stat_invstd, const float *stat_mask, int N, int H, int W, int Cin,
                     int Cout, int relu, int mask_bits, void *stream);
/*      Round 5: adyolo_wino4_fwd launches the PERSISTENT form of the kernel (csrc/wino4p.hpp: one workgroup per CU walks the patches,
 *      the staging pipeline runs across patch boundaries, raw accumulators are exchanged and transformed on the reader side) for
 *      the operand combinations of the SE-ResNet block (no bias, masks as bits, Cout / 64 in {1, 2, 4, 8}) and the
 *      one-patch-per-workgroup kernel otherwise or with ADYOLO_W4_PERSIST=0; same results within fp32 rounding.
 *      adyolo_wino4_last_form(): which one the last call launched (1 one-patch, 2 persistent, 0 none yet) -- for reporting. */
int adyolo_wino4_last_form(void);
/*      adyolo_reload_switches(): the library reads ADYOLO_W4_PERSIST / ADYOLO_W4_NARROW from the environment ONCE (at the first
 *      adyolo_wino4_fwd) and keeps them; this re-reads them (ops.reload_thresholds() calls it: one switch table for the host
 *      code and the library).  Returns the table as bits: 1 persistent kernel on, 2 narrow patches on. */
int adyolo_reload_switches(void);
/* K2w4w (round 5, csrc/wino4w.hip): the weight gradient in the Winograd F(4x4,3x3) domain,
 *      dw = G^T [ sum over 4x4 output tiles (B^T d B) (.) (A e A^T) ] G  (36 multiplies per 16 outputs and channel pair: 9/36 of the
 *      direct form's matrix FLOPs, 1.78x fewer MFMAs than adyolo_wino_wgrad; interpolation points of K2w4; error against a float64
 *      weight gradient ~2e-6 of its absmax, tools/wino4w/numerics.py).  Same operator and argument meaning as adyolo_wino_wgrad
 *      (nn.Conv2d backward-weights, src/models/backbones/resnet.py:16,18; x optionally seen through a per-channel affine with
 *      zero padding).  Shapes: Cin % 32 == 0, Cout % 32 == 0, W % 16 == 0, H % 4 == 0, each tensor below 2 GiB --
 *      adyolo_wino4_wgrad_slabs returns the number of [36][Cin][Cout] float32 slabs the launch needs, or 0 when the shape is
 *      not supported (callers then use adyolo_wino_wgrad).  dw: reference layout [Cout][Cin_real][3][3]. */
int adyolo_wino4_wgrad_slabs(int N, int H, int W, int Cin, int Cout);
int adyolo_wino4_wgrad(const float *x, const float *dy, const float *in_scale /*or NULL*/, const float *in_shift /*or NULL*/,
                       float *slabs, float *du /*[36][Cin][Cout] scratch*/, float *dw, int N, int H, int W, int Cin, int Cin_real,
                       int Cout, void *stream);
/* Winograd weight-gradient: dw = G^T [ sum_tiles (B^T d B)(.)(A e A^T) ] G.  slabs: [n_slabs][16][Cin][Cout] float32 with
 * n_slabs = adyolo_wino_wgrad_slabs(...); du: [16][Cin][Cout] scratch; dw: reference layout [Cout][Cin_real][3][3]. */
int adyolo_wino_wgrad_slabs(int N, int H, int W, int Cin, int Cout);
int adyolo_wino_wgrad(const float *x, const float *dy, const float *in_scale, const float *in_shift, float *slabs,
                      float *du, float *dw, int N, int H, int W, int Cin, int Cin_real, int Cout, void *stream);
int adyolo_conv3x3_wgrad_slabs(int N, int H, int W, int Cin, int Cout);
int adyolo_conv3x3_wgrad(const float *x, const float *dy, const float *in_scale, const float *in_shift,
                         float *slabs, float *dw, int N, int H, int W, int Cin, int Cin_real, int Cout,
                         void *stream);

/* ------------------------------------------------------------------------------------------------
 * K7  dense GEMM on fp32 MFMA:  C[m][n] = sum_k opA(m,k) * opB(n,k) (+ bias[n])
 *     opA(m,k) = transA ? A[k*lda+m] : A[m*lda+k];   opB(n,k) = transB ? B[k*ldb+n] : B[n*ldb+k]
 *     replaces nn.Linear (linearheads.py:95-98, resnet.py:96-98), the 1x1 downsample conv
 *     (resnet.py:160-162), the GRU input projections (resnet.py:153) and their backward passes.
 *   Leading dimensions must be multiples of 4, and so must each operand's contiguous axis (K for a
 *   k-major operand, M resp. N for a transposed one) -- 16-byte vector loads run along it.
 *   splits > 1: K is cut into `splits` ranges, partial products go to `slabs` ([splits][M][N]) and are
 *   summed deterministically into C (bias added once).  accumulate != 0: C += result.
 * ---------------------------------------------------------------------------------------------- */
int adyolo_gemm(const float *A, const float *B, const float *bias, float *C, float *slabs, int M,
                int N, int K, int lda, int ldb, int ldc, int transA, int transB, int splits,
                int accumulate, void *stream);
/* batched variant (attention, resnet_conformer.py:57-85): problem (o, i), o < outer, i < inner, uses the operand
 * bases A + o*oA + i*iA etc. (strides in floats, multiples of 4);  C = alpha * product (+ C when accumulate). */
int adyolo_gemm_batched(const float *A, const float *B, float *C, int M, int N, int K, int lda, int ldb, int ldc,
                        int transA, int transB, int outer, int inner, long oA, long iA, long oB, long iB, long oC,
                        long iC, float alpha, int accumulate, void *stream);
/* out[c] (+)= sum_r A[r*lda + c], deterministic two-stage; partial: [1024][C] workspace */
int adyolo_colsum(const float *A, float *out, float *partial, int R, int C, int lda, int accumulate,
                  void *stream);

/* ------------------------------------------------------------------------------------------------
 * K3  BatchNorm2d (train-mode batch statistics / eval-mode running statistics), channels-last.
 *     replaces nn.BatchNorm2d at resnet.py:17,19,144,163 (momentum 0.1, eps 1e-5).
 *   adyolo_bn_stats: per-sample channel sums  ssum[N][C]  (these are also the SE squeeze, K3b) and
 *     mean/invstd [C] of the whole batch; updates running_mean / running_var (unbiased) in place when
 *     they are non-NULL.  partial: workspace of 4*1024*C floats.
 *   adyolo_bn_scale_shift: scale = gamma*invstd, shift = beta - mean*scale   (train)
 *     or from running stats when mean == NULL is not allowed: pass running_mean/invstd computed by
 *     adyolo_bn_eval_stats.
 *   adyolo_affine_nhwc: y = x*scale[c] + shift[c]
 *   adyolo_bn_bwd_reduce: sdy[c] = sum dy, sdyx[c] = sum dy * xhat   (partial: 2*1024*C floats)
 *   adyolo_bn_bwd_apply:  dx = gamma*invstd*(dy - sdy/R - xhat*sdyx/R) [* (x > 0) when relu_mask]
 *     and dgamma += sdyx, dbeta += sdy when those pointers are non-NULL.
 * ---------------------------------------------------------------------------------------------- */
int adyolo_bn_stats(const float *x, float *ssum, float *mean, float *invstd, float *running_mean,
                    float *running_var, float *partial, int N, int HW, int C, float momentum,
                    float eps, void *stream);
/* same as adyolo_bn_stats but from the per-patch sums a convolution epilogue wrote (tiles = G*N, patch index
 * n*G + g); partial: workspace of 2*1024*C floats.  With gamma / beta / scale / shift (all or none) the same finishing
 * launch also writes scale = gamma*invstd, shift = beta - mean*scale (what adyolo_bn_scale_shift would compute). */
int adyolo_bn_stats_tiles(const float *tile_stats, float *ssum, float *mean, float *invstd, float *running_mean,
                          float *running_var, const float *gamma, const float *beta, float *scale, float *shift,
                          float *partial, int N, int G, int HW, int C, float momentum, float eps, void *stream);
/* the two halves of adyolo_bn_stats_tiles as separate calls (per-patch sums -> per-sample sums ps0 / ps1 [N][C];  per-sample
 * sums -> mean / invstd / running statistics / scale / shift over N samples of HW positions).  Exact data parallelism
 * all-gathers the per-sample sums of all ranks between the two, so N ranks compute the statistics of the concatenated
 * batch bit-identically to one device. */
int adyolo_bn_persample(const float *tile_stats, float *ps0, float *ps1, int N, int G, int C, void *stream);
int adyolo_bn_finish(const float *ps0, const float *ps1, float *mean, float *invstd, float *running_mean,
                     float *running_var, const float *gamma, const float *beta, float *scale, float *shift, int N,
                     int HW, int C, float momentum, float eps, void *stream);
int adyolo_bn_eval_stats(const float *running_mean, const float *running_var, float *mean,
                         float *invstd, int C, float eps, void *stream);
int adyolo_bn_scale_shift(const float *gamma, const float *beta, const float *mean,
                          const float *invstd, float *scale, float *shift, int C, void *stream);
int adyolo_affine_nhwc(const float *x, const float *scale, const float *shift, float *y, long rows,
                       int C, void *stream);
int adyolo_bn_bwd_reduce(const float *dy, const float *x, const float *mean, const float *invstd,
                         float *sdy, float *sdyx, float *partial, long rows, int C, void *stream);
/* sdy / sdyx from the per-patch sums a data-gradient convolution wrote (stat_aux mode) */
int adyolo_bn_bwd_tiles(const float *tile_stats, float *sdy, float *sdyx, float *partial /*[2][256][C]*/, int tiles, int C,
                        void *stream);
int adyolo_bn_bwd_apply(const float *dy, const float *x, const float *gamma, const float *mean,
                        const float *invstd, const float *sdy, const float *sdyx, float *dx,
                        float *dgamma, float *dbeta, float *dx_colsum /*or NULL: [C] channel sums of dx*/,
                        float *colsum_partial /*[8192][C] workspace when dx_colsum*/, long rows, int C, int relu_mask,
                        float count_scale /*1, or the number of equal data-parallel micro-batches sdy / sdyx were summed over*/,
                        void *stream);

/* ------------------------------------------------------------------------------------------------
 * K3b squeeze-excite + residual tail of SEBasicBlock (resnet.py:38-47, SELayer :91-106), fused:
 *     d = c*scale + shift (bn2);  s = sigmoid(W2 relu(W1 mean_hw(d) + b1) + b2);  e = relu(d*s + r)
 *   adyolo_se_fc_fwd: pooled[n][c] = scale*ssum/HW + shift;  hid = relu(W1 pooled + b1) [N][Cr];
 *                     s = sigmoid(W2 hid + b2) [N][C]
 *   adyolo_se_tail_fwd: e = relu((c*scale+shift)*s[n][c] + r)
 *   adyolo_se_tail_bwd_reduce: g = de*(e>0);  sg[n][c] = sum_hw g;  sgx[n][c] = sum_hw g*xhat
 *   adyolo_se_fc_bwd: from sg, sgx: dpool[n][c] and, packed as [db2 C | dW2 C*Cr | db1 Cr | dW1 Cr*C | sdd C | sddx C]
 *                     (P words, one workgroup per sample + a deterministic column sum over the batch), the FC
 *                     gradients and the batch sums bn2's backward needs: sdd[c] = sum dd (= dbeta2),
 *                     sddx[c] = sum dd*xhat (= dgamma2)
 *   adyolo_se_tail_bwd_apply: dc = scale*( g*s + dpool/HW - sdd/R - xhat*sddx/R ),  dr = g (dr may be NULL: the
 *                     identity-shortcut gradient is then formed inside the dgrad epilogue via addend_mask)
 * ---------------------------------------------------------------------------------------------- */
int adyolo_se_fc_fwd(const float *ssum, const float *scale, const float *shift, const float *w1,
                     const float *b1, const float *w2, const float *b2, float *pooled, float *hid,
                     float *s, int N, int HW, int C, int Cr, void *stream);
/* mask (optional): the ReLU mask (e > 0) as bits -- float4 i (4 consecutive channels) of the flattened tensor owns bit
 * (i & 63) of the 64-bit words mask[(i >> 6) * 4 + k], k = component; adyolo_relu_mask_words(N, HW, C) words (0 = shape
 * not supported: HW*C/4 must be a multiple of 64).  The backward passes then read the bits (1/32 of the bytes) instead
 * of e; with mask given, e may be NULL there. */
long adyolo_relu_mask_words(int N, int HW, int C);
/* r_scale / r_shift (both or neither): e = relu((c*scale+shift)*s + (r*r_scale + r_shift)) -- the downsample branch's
 * BatchNorm (resnet.py:160-163) applied while the shortcut is read, so bn(conv1x1(x)) is never written */
int adyolo_se_tail_fwd(const float *c, const float *r, const float *scale, const float *shift,
                       const float *s, const float *r_scale /*or NULL*/, const float *r_shift /*or NULL*/, float *e,
                       uint64_t *mask /*or NULL*/, int N, int HW, int C, void *stream);
/* The tail of the LAST block in front of a pooled stage boundary (the next SEBasicBlock starts with nn.AvgPool2d(2, 2):
 * resnet.py:29,40 / _make_layer :158-164): pooled [N][H/2][W/2][C] = avgpool2(e) and the ReLU-mask bits of e; e itself is not
 * written (only the pooling reads it in the forward pass, the backward passes read the bits).  adyolo_se_tail_fwd_pool_ok: 1
 * when the shape is taken (C/4 a power of two <= 32, H and W even, W*C/4 a multiple of 64), else use adyolo_se_tail_fwd +
 * adyolo_avgpool2_fwd.  Same values as those two calls, bit for bit. */
int adyolo_se_tail_fwd_pool_ok(int H, int W, int C);
int adyolo_se_tail_fwd_pool(const float *c, const float *r, const float *scale, const float *shift, const float *s,
                            const float *r_scale /*or NULL*/, const float *r_shift /*or NULL*/, float *pooled,
                            uint64_t *mask /*or NULL*/, int N, int H, int W, int C, void *stream);
int adyolo_se_tail_bwd_reduce(const float *de, const float *e, const uint64_t *mask /*or NULL*/, const float *c,
                              const float *mean, const float *invstd, float *sg, float *sgx, float *partial, int N,
                              int HW, int C, void *stream);
/* sg, sgx from per-patch sums [2][N*G][C] of a convolution epilogue run with stat_mask = e, stat_aux = c */
int adyolo_se_tail_bwd_tiles(const float *tile_stats, float *sg, float *sgx, int N, int G, int C, void *stream);
long adyolo_se_fc_bwd_words(int C, int Cr);   /* P = 2*C*Cr + Cr + 3*C */
int adyolo_se_fc_bwd(const float *sg, const float *sgx, const float *ssum, const float *gamma,
                     const float *beta, const float *mean, const float *invstd, const float *pooled,
                     const float *hid, const float *s, const float *w1, const float *w2, float *dpool,
                     float *part /*[N][P]*/, float *packed /*[P]*/, float *colsum_ws /*[1024][P]*/, int N,
                     int HW, int C, int Cr, void *stream);
/* The two passes for a block whose output was the pooled tensor (adyolo_se_tail_fwd_pool): dpooled [N][H/2][W/2][C] is the
 * gradient of avgpool2(e); de = 0.25 * dpooled spread over 2 x 2 pixels is formed on the fly (torch.nn.AvgPool2d's backward,
 * resnet.py:29) and written to de_out when given -- the identity shortcut's share of the gradient, read by conv1's
 * data-gradient epilogue -- so no adyolo_avgpool2_bwd launch and no full-size de is read by either pass.  Same values. */
int adyolo_se_tail_bwd_reduce_pooled(const float *dpooled, const uint64_t *mask, const float *c, const float *mean,
                                     const float *invstd, float *sg, float *sgx, float *partial, int N, int H, int W, int C,
                                     void *stream);
int adyolo_se_tail_bwd_apply_pooled(const float *dpooled, const uint64_t *mask, const float *c, const float *gamma,
                                    const float *mean, const float *invstd, const float *s, const float *dpool,
                                    const float *sdd, const float *sddx, float *dc, float *dr /*or NULL*/,
                                    float *de_out /*or NULL*/, int N, int H, int W, int C, float count_scale, void *stream);
int adyolo_se_tail_bwd_apply(const float *de, const float *e, const uint64_t *mask /*or NULL*/, const float *c,
                             const float *gamma, const float *mean, const float *invstd, const float *s,
                             const float *dpool, const float *sdd, const float *sddx, float *dc,
                             float *dr, int N, int HW, int C, float count_scale /*as in adyolo_bn_bwd_apply*/, void *stream);

/* K4  AvgPool2d(2,2) (resnet.py:13,27-29), channels-last; H and W even.  bwd: dx = dy/4 broadcast (+= if accumulate) */
int adyolo_avgpool2_fwd(const float *x, float *y, int N, int H, int W, int C, void *stream);
int adyolo_avgpool2_bwd(const float *dy, float *dx, int N, int H, int W, int C, void *stream);
/* y = a + b (elementwise, n float32);  adyolo_scale_dev: y = a * scalar_dev[0] (scalar read on device: no host sync) */
int adyolo_add(const float *a, const float *b, float *y, long n, void *stream);
int adyolo_scale_dev(const float *a, const float *scalar_dev, float *y, long n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K5  self-attention pooling over the 16 mel positions (resnet.py:109-123)
 *   x [R][F][C] (R = B*T'), w [C], b [1] -> y [R][C], attn [R][F]    (F <= 32, C == 256)
 *   bwd: dx [R][F][C]; dw_partial [nblk][C+1] workspace (last column = db), reduced into dw[C], db[1]
 * ---------------------------------------------------------------------------------------------- */
int adyolo_sap_fwd(const float *x, const float *w, const float *b, float *y, float *attn, int R,
                   int F, int C, void *stream);
int adyolo_sap_bwd(const float *dy, const float *x, const float *w, const float *attn, float *dx,
                   float *dw, float *db, float *partial, int R, int F, int C, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K6  one bidirectional GRU layer, hidden 128 (nn.GRU at resnet.py:153,195; gate order r,z,n)
 *   gx  [B][T][2][384]  = x W_ih^T + b_ih for both directions (from adyolo_gemm)
 *   whh [2][384][128], bhh [2][384]
 *   out [B][T][256] (forward hidden | backward hidden);  gates [B][T][2][4][128] saves r,z,n,hn;
 *   hprev [B][T][2][128] saves the hidden state entering each step (gates/hprev may be NULL in eval)
 *   bwd: dout [B][T][256] -> dgx [B][T][2][384], dgh [B][T][2][384]
 *        (dW_ih = dgx^T x, dW_hh = dgh^T hprev, db = column sums, dx = dgx W_ih: adyolo_gemm/colsum)
 * K6b LayerNorm(256) + tanh (resnet.py:154,196-197): y = tanh(LN(x));
 *     bwd: dx, and dgamma/dbeta through `partial` ([nblk][2*C]) reduced into dgamma/dbeta (accumulated)
 * dropout mask for the inter-layer dropout (p = 0.3, train only): mask[i] = keep ? 1/(1-p) : 0 from a
 * counter-based generator (seed, offset); adyolo_mul applies it.
 * ---------------------------------------------------------------------------------------------- */
int adyolo_gru_fwd(const float *gx, const float *whh, const float *bhh, float *out, float *gates,
                   float *hprev, int B, int T, void *stream);
int adyolo_gru_bwd(const float *dout, const float *gates, const float *hprev, const float *whh,
                   float *dgx, float *dgh, int B, int T, void *stream);
int adyolo_ln_tanh_fwd(const float *x, const float *gamma, const float *beta, float *y, long R,
                       int C, float eps, void *stream);
int adyolo_ln_tanh_bwd(const float *dy, const float *x, const float *y, const float *gamma,
                       float *dx, float *dgamma, float *dbeta, float *partial, long R, int C,
                       float eps, void *stream);
int adyolo_dropout_mask(float *mask, long n, float p, uint64_t seed, uint64_t offset, void *stream);
/* y[i] = x[i] * mask[i] with the mask values adyolo_dropout_mask writes for (seed, offset), generated on the fly (forward:
 * x -> y; backward: the same call on the incoming gradient); n a multiple of 4.  Replaces nn.Dropout /
 * nn.GRU(dropout=) (reference resnet.py:153, resnet_conformer.py:46-47,199-206). */
int adyolo_dropout_apply(const float *x, float *y, long n, float p, uint64_t seed, uint64_t offset, void *stream);
/* same, with a device-side running offset added to `offset` (offset_dev may be NULL): a launch recorded in a hipGraph
 * draws a fresh part of the stream at every replay; adyolo_counter_add advances the counter at the end of a step. */
int adyolo_dropout_apply_dev(const float *x, float *y, long n, float p, uint64_t seed, uint64_t offset,
                             const uint64_t *offset_dev, void *stream);
int adyolo_counter_add(uint64_t *counter, uint64_t inc, void *stream);
int adyolo_mul(const float *a, const float *b, float *y, long n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K8  AD-YOLO loss, forward + backward in one pass over the logits (src/models/loss.py:189-251)
 *   logit  [B*T][Gaz*Gel][A][C+3]  (channel order obj, cls x C, u, v)
 *   target [M][7] float32 (b, frame, Gi, Gj, cls, U, V) -- datasets.py:164-184
 *   thr[3] responsibility thresholds in degrees (train_unify), gains[4] = angular, object, nonobj, class
 *   ws: workspace of adyolo_loss_workspace_words(...) 32-bit words (zeroed by the call itself)
 *   loss [1] float32;  dlogit same shape as logit or NULL (eval).  grad_scale multiplies dlogit.
 *   dist [M][A] (optional, may be NULL): angular distances D, for tests.
 * ---------------------------------------------------------------------------------------------- */
long adyolo_loss_workspace_words(int BT, int G, int A, int M);
int  adyolo_loss_fwd_bwd(const float *logit, const float *target, float *ws, float *loss,
                         float *dlogit, float *dist, int B, int T, int Gaz, int Gel, int A, int C,
                         int M, const float *thr_host, const float *gains_host, float grid_az,
                         float grid_el, float g_overlap, float grad_scale, void *stream);
/* the same in two phases (phases bit 0: workspace reset + assignment, bit 1: pass over the logits + final sum).  Exact data
 * parallelism all-reduces the first four 32-bit words of the workspace (distinct positives per threshold, responsible pairs)
 * between the phases and passes na_total = anchors of the whole batch (0: this call's own), so that every loss term is
 * normalised as on one device over the concatenated batch (loss.py:236-243). */
int  adyolo_loss_phase(const float *logit, const float *target, float *ws, float *loss,
                       float *dlogit, float *dist, int B, int T, int Gaz, int Gel, int A, int C,
                       int M, const float *thr_host, const float *gains_host, float grid_az,
                       float grid_el, float g_overlap, float grad_scale, int phases, long na_total, void *stream);

/* K8b inference decode (LabelPostProcessor.get_yolo_output, src/datasets.py:752-771): per anchor
 *   out = [sigmoid(obj), sigmoid(cls_c)*sigmoid(obj) x C, U deg in [-180,180), V deg in [-90, 90-1e-7]];
 *   thresholding and the (tiny, data-dependent) NMS stay on the host (ad-yolo_amd/postprocess.py). */
int adyolo_yolo_decode(const float *logit, float *out, long n_frames, int Gaz, int Gel, int A, int C,
                       float grid_az, float grid_el, float g_overlap, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K10 the other heads / losses behind the reference's --loss switch (src/main.py:43)
 *   adyolo_act_fwd/bwd : y[r][c] = c < n_sigmoid_cols ? sigmoid(x) : tanh(x)   (linearheads.py:44-47,65,83)
 *   adyolo_seddoa_loss : w_bce * mean BCE(out[:, :nsed], tgt[:, :nsed]) + w_mse * mean((out[:, nsed:] * m - tgt[:, nsed:])^2),
 *                        m = tgt activity of the column's class when `masked` (SEDDOAloss loss.py:32-54: w = 1, 1000;
 *                        ACCDOAloss loss.py:57-67: nsed = 0, w_mse = 1).  partial: 2*1024 floats.  dout may be NULL.
 *   adyolo_adpit_loss  : ADPITloss (loss.py:70-153): out [rows][9][C], tgt [rows][6][4][C]; 13-permutation min-MSE.
 *                        partial: 1024 floats.
 * ---------------------------------------------------------------------------------------------- */
int adyolo_act_fwd(const float *x, float *y, long rows, int cols, int n_sigmoid_cols, void *stream);
int adyolo_act_bwd(const float *dy, const float *y, float *dx, long rows, int cols, int n_sigmoid_cols,
                   void *stream);
int adyolo_seddoa_loss(const float *out, const float *tgt, float *loss, float *dout, float *partial, long rows,
                       int cols, int nsed, int masked, float w_bce, float w_mse, void *stream);
int adyolo_adpit_loss(const float *out, const float *tgt, float *loss, float *dout, float *partial, long rows,
                      int C, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline around K1.
 *   adyolo_pcm16_to_f32: staged WAV samples int16 -> float, x / 32768 + 1e-8 (src/datasets.py:105, src/preprocess.py:104)
 *   adyolo_mask_ranges:  SpecAug (src/utils/augmentations.py:6-33) on feat [B][T][F][C]: per sample zero frames [t0,t1) and
 *                        mel bins [f0,f1); ranges [B][4] int32 = {t0, t1, f0, f1}, an empty range masks nothing
 *   adyolo_colstats:     per-column sum, sum of squares, max, min of a [rows][cols] fp32 matrix -> out [4][cols] float64
 *                        (train-set scaler statistics, src/preprocess.py:86-130); partial = [4][1024][cols] fp32 scratch
 * ---------------------------------------------------------------------------------------------- */
int adyolo_pcm16_to_f32(const int16_t *pcm, float *out, long n, void *stream);
int adyolo_mask_ranges(float *feat, const int32_t *ranges, int B, int T, int F, int C, void *stream);
int adyolo_colstats(const float *a, float *partial, double *out, long rows, int cols, void *stream);

/* General strided convolution (channels-last) as an implicit GEMM on the fp32 MFMA, no column buffer (replaces nn.Conv2d
 * 7x7 s(1,2) at reference resnet_conformer.py:347 and torchvision BasicBlock's 3x3 / 1x1 s(1,2) at :353-393; forward,
 * data-gradient and weight-gradient).  Cin, Cout multiples of 4.  Kp = roundup4(KH*KW*Cin), Kq = roundup4(KH*KW*Cout).
 *   mode 0: src = x [N][H][W][Cin],     other = wk  [Cout][Kp] (k = (kh*KW+kw)*Cin + ci),  out = y  [N][Ho][Wo][Cout]
 *   mode 1: src = dy [N][Ho][Wo][Cout], other = wkT [Cin][Kq]  (k = (kh*KW+kw)*Cout + co), out = dx [N][H][W][Cin]
 *   mode 2: src = x,                    other = dy,                                         out = dwk [Cout][Kp]
 * splits (mode 2 only): split-K over the N*Ho*Wo output pixels, slabs = [splits][Cout][Kp] workspace (deterministic sum). */
int adyolo_conv_gemm(const float *src, const float *other, float *out, float *slabs, int mode, int N, int H, int W,
                     int Cin, int Cout, int KH, int KW, int SH, int SW, int PH, int PW, int splits, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K9  ResNet-Conformer encoder pieces (src/models/backbones/resnet_conformer.py), channels-last fp32
 *   pack_wk : filter layout of adyolo_conv_gemm; rows of `wk` are (kh, kw, c)-ordered and padded to a multiple of 4 floats
 *       (Kp).  Used for the 7x7 s(1,2) stem (:347) and the torchvision BasicBlock 3x3 / 1x1 s(1,2) convolutions (:353-393).
 *   maxpool3 : MaxPool2d(3, stride (1,2), padding 1) (:350); arg = arg-max tap per output (uint8); bwd atomically
 *       adds into a ZEROED dx.
 *   affine_relu, relu_bwd, axpby : BN->ReLU of BasicBlock, residual mixing a*x + b*z (:98)
 *   swish (:142-150), glu over the channel axis (:167), dwconv3: depthwise Conv1d k=3, dilation d, padding d (:169)
 *       (flip = 1 gives the data gradient), softmax rows with a pre-scale (:73-76), avgpool1d(k) * fac (:288-294),
 *   ln : LayerNorm(256).
 * ---------------------------------------------------------------------------------------------- */
int adyolo_pack_wk(float *w, float *wk, int Cout, int Cin, int KH, int KW, int to_packed, void *stream);
int adyolo_maxpool3_fwd(const float *x, float *y, unsigned char *arg, int N, int H, int W, int C, void *stream);
int adyolo_maxpool3_bwd(const float *dy, const unsigned char *arg, float *dx_zeroed, int N, int H, int W, int C,
                        void *stream);
int adyolo_affine_relu_nhwc(const float *x, const float *scale, const float *shift, float *y, long rows, int C,
                            void *stream);
int adyolo_relu_bwd(const float *dy, const float *y, float *dx, long n, void *stream);
int adyolo_axpby(const float *x, const float *z, float *y, float a, float b, long n, void *stream);
/* y = a * dropout(x) + b * z in one pass (z NULL: y = a * dropout(x), the gradient w.r.t. x); mask = the stateless stream of
 * adyolo_dropout_apply(_dev) -- ResidualConnectionModule (resnet_conformer.py:98) on a sub-module that ends in nn.Dropout
 * (:209 FeedForwardModule, :178 ConformerConvModule, :272-274 attention branch); same rounding as the two separate calls. */
int adyolo_dropout_axpby(const float *x, const float *z /*or NULL*/, float *y, long n, float p, uint64_t seed, uint64_t offset,
                         const uint64_t *offset_dev /*or NULL*/, float a, float b, void *stream);
int adyolo_swish_fwd(const float *x, float *y, long n, void *stream);
int adyolo_swish_bwd(const float *dy, const float *x, float *dx, long n, void *stream);
int adyolo_glu_fwd(const float *x, float *y, long rows, int C, void *stream);
int adyolo_glu_bwd(const float *dy, const float *x, float *dx, long rows, int C, void *stream);
int adyolo_dwconv3_fwd(const float *x, const float *w, const float *bias, float *y, int B, int T, int C,
                       int dilation, int flip, void *stream);
int adyolo_dwconv3_wgrad(const float *dy, const float *x, float *dw, float *db, float *partial, float *colsum_ws,
                         int B, int T, int C, int dilation, void *stream);
int adyolo_softmax_fwd(const float *s, float *p, long rows, int L, float scale, void *stream);
int adyolo_softmax_bwd(const float *dp, const float *p, float *ds, long rows, int L, float scale, void *stream);
int adyolo_avgpool1d_fwd(const float *x, float *y, int B, int T, int C, int k, float fac, void *stream);
int adyolo_avgpool1d_bwd(const float *dy, float *dx, int B, int T, int C, int k, float fac, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K9w  1-D Winograd F(4, 3) along the time axis for the stride-1 3x3 convolutions of the ResNet-Conformer's deep stages, whose
 *      maps are 1 bin wide (or 2 bins, folded into the channels): there torchvision BasicBlock's convolution (reference
 *      resnet_conformer.py:353-393) IS a 3 x 1 one.  H % 4 == 0, C % 4 == 0; T = H / 4 tiles per sample.
 *   wino1d_in     : x  [N][H][C] -> V [6][N T][C]  (B^T over the rows 4 t - 1 .. 4 t + 4; rows outside the sample are zeros)
 *   wino1d_out    : M  [6][N T][C] -> y [N][H][C]  (A^T)
 *   wino1d_dy     : dy [N][H][C] -> E [6][N T][C]  (A: the output gradient in the transform domain, for the weight gradient)
 *   wino1d_filter : mode 0  U[p][co][ci] = sum_k G[p][k] w[co][ci][k]        (forward)
 *                   mode 1  U[p][ci][co] = sum_k G[p][k] w[co][ci][2 - k]    (data gradient)
 *                   mode 2  w[co][ci][k] = sum_p G[p][k] U[p][co][ci]        (weight gradient from dU; writes w)
 *   The six position GEMMs between them are ONE adyolo_gemm_batched launch each way (M[p] = V[p] U[p]^T, dU[p] = E[p]^T V[p]).
 * ---------------------------------------------------------------------------------------------- */
int adyolo_wino1d_in(const float *x, float *V, int N, int H, int C, void *stream);
int adyolo_wino1d_out(const float *M, float *y, int N, int H, int C, void *stream);
int adyolo_wino1d_dy(const float *dy, float *E, int N, int H, int C, void *stream);
int adyolo_wino1d_filter(float *w, float *U, int Cout, int Cin, int mode, void *stream);
int adyolo_ln_fwd(const float *x, const float *gamma, const float *beta, float *y, long R, int C, float eps,
                  void *stream);
int adyolo_ln_bwd(const float *dy, const float *x, const float *gamma, float *dx, float *dgamma, float *dbeta,
                  float *partial, long R, int C, float eps, void *stream);

/* FOA rotation augmentation on raw audio (src/utils/augmentations.py:81-96): audio/out [B][n_samples][4] (W,Y,Z,X),
 * cfg [B][4] = {sign_y, sign_z, sign_x, swap_xy}.  Label angles are remapped on the host (augmentations.RotationAug). */
int adyolo_foa_rotate(const float *audio, float *out, const float *cfg, int B, long n_samples, void *stream);

/* K11 fused Adam over one flat parameter buffer (torch.optim.Adam at src/train.py:31,55; no amsgrad) */
int adyolo_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n,
                     float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     float grad_scale, void *stream);
/* same update with the step counter ON THE DEVICE (one uint64, incremented by the call itself; bc_dev = 2 floats of
 * scratch for the bias corrections): nothing in the argument list changes from step to step, so the whole train step can
 * be recorded once in a hipGraph and replayed (train.TrainStep(graph=True)). */
int adyolo_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, uint64_t *step_dev, float *bc_dev,
                         float grad_scale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K9a multi-head self-attention core, flash style on the exact-fp32 matrix cores (csrc/attention.hip).
 *     replaces the energy / softmax / dropout / context products of MultiHeadAttention.forward
 *     (src/models/backbones/resnet_conformer.py:57-85); the (B, H, T, T) scores are never written to HBM.
 *   q, k, v, ctx, dq, dk, dv, dctx: [B][T][H*D] float32, head h in columns h*D .. h*D+D-1;  D must be 64
 *   ctx = dropout(softmax(scale * q k^T), p) v    per (batch, head);  dropout_p = 0: none (eval)
 *   lse2 [B][H][T]: per query row log2(sum_k exp(scale * q k)) (written by fwd when non-NULL, read by bwd)
 *   delta [B][H][T]: workspace of bwd (sum_dv dctx * ctx)
 *   seed: 32-bit seed of the stateless dropout hash keep(b, h, query, key); forward and backward of one call must use
 *         the same (dropout_p, seed).  adyolo_attn_dropout_mask writes that mask ([B][H][T][T], values 0 or 1/(1-p)).
 * ---------------------------------------------------------------------------------------------- */
/*   seed_dev (round 4; may be NULL): when non-NULL the kernels read the seed from this device word instead of `seed` -- a recorded
 *         step replays with a seed derived on the device by adyolo_seed32_dev(seed64, offset, offset_dev, out): out[0] = the 32-bit
 *         value the host-side stream (rng.DropoutStream.seed32) computes for (seed64, offset + *offset_dev). */
int adyolo_attn_fwd(const float *q, const float *k, const float *v, float *ctx, float *lse2, int B, int T, int H, int D,
                    float scale, float dropout_p, uint32_t seed, const uint32_t *seed_dev, void *stream);
int adyolo_attn_bwd(const float *q, const float *k, const float *v, const float *ctx, const float *dctx,
                    const float *lse2, float *delta, float *dq, float *dk, float *dv, int B, int T, int H, int D,
                    float scale, float dropout_p, uint32_t seed, const uint32_t *seed_dev, void *stream);
int adyolo_seed32_dev(uint64_t seed, uint64_t offset, const int64_t *offset_dev /*or NULL*/, uint32_t *out, void *stream);
int adyolo_attn_dropout_mask(float *mask, int B, int T, int H, float dropout_p, uint32_t seed, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ADYOLO_HIP_H */
