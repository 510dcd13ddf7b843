/*
 * tatt_hip.h -- C ABI of libtatt_hip.so, the MI355X (gfx950) kernels of the TATT super-resolution
 * hot path (TSRN backbone + TP interpreter + STN/TPS sampler, the TBSRN variant, the training-step losses / optimiser and the
 * CRNN text-prior generator in front of the path; forward and backward).
 *
 * Every entry point takes plain DEVICE pointers, sizes/strides and a hipStream_t; none allocates,
 * synchronises or touches host memory, so a call sequence can be captured in a hipGraph.  Return
 * value: 0 on success, otherwise a hipError_t (or 1 for an unsupported shape).  All tensors are fp32.
 * Feature maps are token-major ("NHWC": row = (b,h,w), contiguous channel axis) unless strides say
 * otherwise.  Scratch ("ws"/"part") sizes are stated per function.
 *
 * The reference (mjq11302010044/TATT) is pure PyTorch, so there is no FFI upstream; each function
 * names the torch.nn call site it replaces (file:line in the reference).  The host-side mirror of the
 * reference's nn.Module surface (tatt_amd/tsrn.py, tbsrn.py, crnn.py, train.py) is the only caller.
 *
 * Activation codes: 0 none, 1 relu, 2 mish (x*tanh(softplus(x)), model/tsrn.py:1056-1064), 3 tanh.
 */
#ifndef TATT_HIP_H
#define TATT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* ---- GEMM / convolution (v_mfma_f32_32x32x2_f32, exact fp32) --------------------------------------- */

/* C[z] = act(alpha * (A[z] @ B[z] + bias[z])) + beta * C[z],  A(i,r) = A[i*sam + r*sak] for r < K1 (or all r
 * when A2 == NULL), A2[i*sa2m + (r-K1)*sa2k] otherwise; B(r,j) = B[r*sbk + j*sbn]; C[i*scm + j*scn].
 * splitk > 1: ws >= Z*splitk*M*N floats.  rowsum != NULL (Z == 1): rowsum[i] = alpha * sum_r A(i,r) is produced in the same pass
 * (a bias gradient riding along with the weight-gradient GEMM); with split-K ws needs splitk*M more floats.
 * Replaces nn.Linear (model/tsrn.py:170; model/transformer_v2.py:455-457,788-790; model/stn_head.py:50,53),
 * the 1x1 nn.Conv2d of GruBlock (model/tsrn.py:1071), nn.GRU input projections (model/tsrn.py:1072;
 * model/transformer_v2.py:177) and the packed in/out projections of nn.MultiheadAttention
 * (model/transformer_v2.py:453,786), forward and backward. */
int tatt_gemm(const float* A, long sam, long sak, const float* A2, long sa2m, long sa2k, int K1,
              const float* B, long sbk, long sbn, const float* bias, float* C, long scm, long scn,
              int M, int N, int K, int Z, long bsA, long bsA2, long bsB, long bsC, long bsBias,
              float alpha, float beta, int act, int splitk, float* ws, float* rowsum, hipStream_t st);

/* y[pixel*ldy + co] = act(conv(x, w) + bias) + beta*y; stride 1, 'same' padding; x[n*xsn + h*xsh + w*xsw + c*xsc];
 * wpacked = [KH][KW][Cin][Cout] from tatt_repack_conv_weight.  Also computes the data gradient when given
 * dY as x and the mode-1 packed filter.  splitk > 1 (ws >= splitk*Bn*H*W*Cout floats) for small-M / deep-K shapes.
 * Replaces nn.Conv2d (model/tsrn.py:597,612,877,885,1043,623;
 * model/stn_head.py:15). */
int tatt_conv2d_fwd(const float* x, long xsn, long xsh, long xsw, long xsc, const float* wpacked,
                    const float* bias, float* y, long ldy, int Bn, int H, int W, int Cin, int Cout,
                    int KH, int KW, int act, float beta, int splitk, float* ws, hipStream_t st);

/* tatt_conv2d_fwd without bias / activation / beta that leaves the split contraction UNSUMMED: ws receives *splits partial maps
 * [s][pixel][Cout] (*splits <= splitk; 1 = the finished map) for a consumer that adds them while it loads them
 * (tatt_stn_bn_pool_fwd_parts / tatt_stn_bn_pool_bwd_parts: one launch less per link of the STN head's dependent chain,
 * model/stn_head.py:15).  ws >= max(splitk, 1)*Bn*H*W*Cout floats; `splits` is a HOST pointer, written before the call returns. */
int tatt_conv2d_fwd_partials(const float* x, long xsn, long xsh, long xsw, long xsc, const float* wpacked, int Bn, int H,
                             int W, int Cin, int Cout, int KH, int KW, int splitk, float* ws, int* splits, hipStream_t st);

/* dw_oihw[co][ci][kh][kw] = sum_pixels x[pixel+(kh,kw)][ci] * dy[pixel*lddy + co] + beta*dw; ws >= splitk*KH*KW*Cin*Cout floats */
int tatt_conv2d_wgrad(const float* x, long xsn, long xsh, long xsw, long xsc, const float* dy,
                      long lddy, float* dw_oihw, int Bn, int H, int W, int Cin, int Cout, int KH,
                      int KW, float beta, int splitk, float* ws, hipStream_t st);

/* OIHW nn.Conv2d weight -> GEMM operand; mode 0: [KH][KW][Cin][Cout]; mode 1: [KH][KW][Cout][Cin], taps flipped (data gradient);
 * mode 2: [KH][KW][Cout][Cin]; mode 3: [KH][KW][Cin][Cout], taps flipped -- forward / data-gradient filters with the
 * contraction axis contiguous, for tatt_conv3_c64_fwd_t; modes 6 / 7: the same two 3x3 filters (64 contraction channels) in the
 * per-lane register order of tatt_conv3_c64_fwd_ws16 (4 / 5: retired);
 * modes 8 / 9: the Toeplitz-expanded 9x9 filter of tatt_conv9_c64_to_c4_mfma in its MFMA fragment order (out needs 110,592 floats);
 * modes 10 / 11: the split-bf16 (hi / lo) forward / data-gradient operand of tatt_conv3_c64_fwd_sb (3x3, channel counts multiples
 * of 64; Cout*Cin*9 32-bit words of two bf16, one chunk per 64 contraction channels);
 * modes 12 / 13: the Toeplitz filter of modes 8 / 9 as split-bf16 hi / lo operand fragments of tatt_conv9_c64_to_c4_sb (110,592 words) */
int tatt_repack_conv_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW,
                            int mode, hipStream_t st);
/* number of 32-bit words the packed layout `mode` of a (Cout, Cin, KH, KW) filter occupies = the size of `out` above; -1 for an
 * unknown mode.  Host-only (no launch): the one place the layout sizes are defined. */
int tatt_repack_words(int Cout, int Cin, int KH, int KW, int mode);
/* the same for n filters in one launch (all packed layouts of a model, refreshed once per optimiser step): ws / outs are HOST arrays
 * of n device pointers, dims a host array of n x 5 ints (Cout, Cin, KH, KW, mode) */
int tatt_repack_conv_weight_batch(const float* const* ws, float* const* outs, const int* dims, int n, hipStream_t st);

/* Weight gradients of one GruBlock in one pass over the tokens (split-bf16 MFMA, hi hi + hi lo + lo hi): per-group partials of
 * dW' (192 x K) = dgi^T [x | xb] with the row sums of dgi (= db') and dW_hh (192 x 64) = dgh^T hprev with the row sums of dgh
 * (= db_hh); reference model/tsrn.py:1075-1084 (GruBlock: 1x1 conv composed with nn.GRU's input projection).  dgi, dgh (M, 192),
 * x, xb, hprev (M, 64) contiguous, xb may be null (K = 64, else 128), M % 32 == 0.  G = persistent work-groups = partial slabs,
 * 1 <= G <= min(M / 32, 256): ws1 >= G*192*K + G*192 floats, ws2 >= G*192*64 + G*192; sum them with tatt_splitk_reduce(ws1, dWp, 192, K, G, 0, 0, 0, dbp, 192) and
 * tatt_splitk_reduce(ws2, dWhh, 192, 64, G, 0, 0, 0, dbhh, 192). */
int tatt_gru_wgrad_sb(const float* dgi, const float* dgh, const float* x, const float* xb, const float* hprev,
                      float* ws1, float* ws2, int M, int G, hipStream_t st);

/* C = sum over S partial (M x N) slabs (+ beta*C); remap_cin > 0: row i = tap*remap_cin + ci, col j = co is scattered to
 * the OIHW filter layout dW[co][ci][tap].  vec (may be NULL): additionally vec[i] = sum over S partial vectors of vec_len
 * floats stored behind the S slabs (the bias-gradient partials of tatt_conv3_c64_wgrad_partial) */
int tatt_splitk_reduce(const float* partial, float* C, int M, int N, int S, int remap_cin, int remap_taps,
                       float beta, float* vec, int vec_len, hipStream_t st);
/* on != 0: split-K reductions issued ON STREAM st from now on (tatt_gemm / tatt_conv2d_wgrad with splitk > 1, tatt_splitk_reduce)
 * are only REGISTERED -- their partial slabs must stay allocated -- and are summed by one launch per 36 entries at
 * tatt_reduce_flush(st) or tatt_reduce_defer(0, st): the 70 small reduction launches behind the weight-gradient GEMMs of a training
 * step become 6.  Results are undefined until the flush.  Accumulating reductions (beta != 0) flush and run immediately.
 * This is the ONLY host-side state the library keeps: one registration table per stream, guarded by a mutex -- callers that drive
 * different streams (several trainers, several host threads) do not interact; calls on ONE stream must come from one thread at a
 * time, as for any stream-ordered API. */
int tatt_reduce_defer(int on, hipStream_t st);
int tatt_reduce_flush(hipStream_t st);

/* Specialised 3x3 convolution, Cin/Cout/W multiples of 64, NHWC contiguous: the SRB / block7 / up-sampler convs
 * (model/tsrn.py:877,885,612,1043), forward and data gradient.  y = act(conv + bias) + beta*y.  Filter packed [9][Cout][Cin]
 * (repack mode 2; mode 3 for the data gradient); the filter slice of each tap is staged through LDS, both MFMA operands are
 * read with 16-byte LDS loads along the contraction axis.  Used when Cin != 64. */
int tatt_conv3_c64_fwd_t(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W, int Cin,
                         int Cout, int act, float beta, hipStream_t st);
/* weight-stationary 3x3 convolution for 64 input channels in exact fp32 (reference nn.Conv2d(64, Cout, 3, padding=1):
 * model/tsrn.py:877,885,612,1043 and their data gradients): a persistent work-group keeps the filter in registers, wave = 32 pixels
 * x 16 output channels and the whole 9 x 64 contraction on v_mfma_f32_16x16x4_f32; wl = tatt_repack_conv_weight mode 6 (forward)
 * / mode 7 (data gradient of a 64-output-channel convolution) */
int tatt_conv3_c64_fwd_ws16(const float* x, const float* wl, const float* bias, float* y, int B, int H, int W,
                            int Cout, int act, float beta, hipStream_t st);
/* tatt_conv3_c64_fwd_ws16 with BatchNorm folded in on either side (reference model/tsrn.py:877-886: conv -> bn -> mish -> conv -> bn):
 * in_scale / in_shift (64 floats, nullable): input pixels pass through in_act(x * in_scale[c] + in_shift[c]) while the halo is
 * staged (the producer's BatchNorm + activation; zero padding pads the transformed map); stats (nullable; Cout == 64, act none,
 * beta 0): [min(256, B*H*W/64)][2][64] doubles, per-work-group sum and sum of squares of the output per channel = the stage-1
 * partials tatt_bn_stats_finish turns into mean / rstd / running statistics. */
int tatt_conv3_c64_fwd_ws16_bn(const float* x, const float* wl, const float* bias, float* y, int B, int H, int W,
                               int Cout, int act, float beta, const float* in_scale, const float* in_shift, int in_act,
                               double* stats, hipStream_t st);
/* The same 3x3 convolution on the bf16 matrix cores by operand splitting: a = hi + lo (hi = bf16(a), lo = bf16(a - hi)),
 * a*b ~ hi hi + hi lo + lo hi accumulated in fp32 (the dropped lo*lo term is 2^-16 relative; measured effect on the network:
 * profiles/r03_split_bf16_probe.txt).  x holds cin_total >= 64 channels per pixel; the 64-channel slice starting at ci0 is
 * contracted with the matching chunk of a mode-10 / mode-11 (or 14 / 15: tatt_conv3_sb_packing) packed filter (chunk c starts c*Cout*576 words in); wider inputs are
 * chunked by the caller with beta = 1.  BatchNorm folding arguments as tatt_conv3_c64_fwd_ws16_bn. */
int tatt_conv3_c64_fwd_sb(const float* x, int cin_total, int ci0, const float* wl, const float* bias, float* y, int B, int H,
                          int W, int Cout, int act, float beta, const float* in_scale, const float* in_shift, int in_act,
                          double* stats, hipStream_t st);
/* The data-gradient convolution (64 -> 64 channels, wl = mode-11 packed filter) of a conv -> bn -> act -> conv -> bn chain (reference
 * model/tsrn.py:877-886, backward) with the BatchNorm backward folded in on both sides:
 *   input side (x2 != NULL): the gradient entering is x * in_scale + x2 * in_scale2 + in_shift per channel -- the BatchNorm backward
 *     dy = a du + b y + c of the layer above with (a, b, c) from tatt_bn_bwd_finish, x = du, x2 = that BatchNorm's input y;
 *   output side (ep_x != NULL): y = (convolution) * ep_act'(ep_gamma xhat + ep_beta), xhat = (ep_x - ep_mean) ep_rstd -- the gradient
 *     through the activation that follows the layer below's BatchNorm -- and stats [min(256, B*H*W/64)][2][64] doubles = the
 *     per-work-group sums of y and y * xhat, i.e. the stage-1 partials of THAT BatchNorm's backward (tatt_bn_bwd_finish sums them). */
int tatt_conv3_c64_dgrad_bn_sb(const float* x, const float* x2, const float* in_scale, const float* in_scale2,
                               const float* in_shift, const float* wl, float* y, int B, int H, int W, const float* ep_x,
                               const float* ep_mean, const float* ep_rstd, const float* ep_gamma, const float* ep_beta,
                               int ep_act, double* stats, hipStream_t st);
/* Which kernel the two entries above launch, and the filter packing it wants.  tatt_conv3_sb_generation (test / A-B hook; returns the
 * previous setting, other values only query): 4 (default) = 4 x 16-pixel tiles, four MFMA waves (one per SIMD, each owning 32 output
 * channels over half the contraction, v_mfma_f32_32x32x16_bf16) with four staging waves beside them (round 6); 3 = the same without
 * staging waves (one 512-register wave per SIMD does everything); 1 = the 64-pixel row tiles of rounds 3-5.  Generation 4 takes
 * H % 4 == 0, any W (the last tile column of a ragged map is cut), act none / ReLU, no tanh; generation 3 H % 4 == 0, W % 16 == 0, no
 * output activation; everything else runs generation 1 (W % 64 == 0).  tatt_conv3_sb_packing:
 * the tatt_repack_conv_weight mode of the FORWARD filter for a call with these arguments -- 10 (generation 1) or 14 (generations
 * 3 / 4); the data-gradient packing is that + 1. */
int tatt_conv3_sb_generation(int gen);
int tatt_conv3_sb_packing(int B, int H, int W, int cin_total, int Cout, int act, int ep_act);
/* weight-gradient partials part[G][9*Cin][Cout] (G persistent work-groups, G <= B*H*W/64) and, if pdb != NULL, bias-gradient
 * partials pdb[G][Cout] (the column sums of dy the kernel streams anyway; nn.Conv2d's bias gradient); finish with
 * tatt_splitk_reduce(part, dw_oihw, 9*Cin, Cout, G, Cin, 9, beta, db, Cout) where pdb = part + G*9*Cin*Cout */
int tatt_conv3_c64_wgrad_partial(const float* x, const float* dy, float* part, float* pdb, int B, int H, int W,
                                 int Cin, int Cout, int G, hipStream_t st);
/* the same partials on the bf16 matrix cores by operand splitting (hi hi + hi lo + lo hi, fp32 accumulation; csrc/conv3w.hip): same
 * arguments, same outputs up to 2^-16 relative per product */
int tatt_conv3_c64_wgrad_partial_sb(const float* x, const float* dy, float* part, float* pdb, int B, int H, int W,
                                    int Cin, int Cout, int G, hipStream_t st);
/* Test / A-B hook: 2 (default) = 4 x 16-pixel tiles, transposing LDS reads, staging waves beside MFMA waves (round 6; H % 4 == 0,
 * any W: the last tile column of a ragged map is cut; G <= B * (H / 4) * ceil(W / 16)), 1 = the 64-pixel row segments of rounds 3-5
 * (also what H % 4 != 0 runs; W % 64 == 0).  Returns the previous setting. */
int tatt_conv3_wgrad_sb_generation(int gen);

/* 9x9 convolution 64 -> 4 channels (fp32 vector ALU; filter through the scalar cache): the final reconstruction conv
 * (model/tsrn.py:623) and, with the mode-1 packed filter, the data gradient of block1 (model/tsrn.py:597).
 * x (B,H,W,C) NHWC, C % 16 == 0, H % 8 == 0, W % 32 == 0; wpacked [81][C][4]; y (B,H,W,4) */
int tatt_conv9_c64_to_c4(const float* x, const float* wpacked, const float* bias, float* y, int B, int H, int W,
                         int C, hipStream_t st);
/* the same convolution on v_mfma_f32_16x16x4_f32: tile columns = (4 neighbouring pixels x 4 output channels), Toeplitz-expanded
 * filter wt (110,592 floats) from tatt_repack_conv_weight mode 8 (forward, reference model/tsrn.py:623) / mode 9 (data gradient of
 * block1's 4->64 convolution, :597); x (B,H,W,64) NHWC contiguous, H % 4 == 0, W % 64 == 0; y (B,H,W,4) */
int tatt_conv9_c64_to_c4_mfma(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                              hipStream_t st);
/* the same Toeplitz convolution on v_mfma_f32_16x16x32_bf16 with every fp32 operand split a = hi + lo (hi hi + hi lo + lo hi, fp32
 * accumulation: 2^-16 relative per product, the arithmetic of tatt_conv3_c64_fwd_sb); wt from tatt_repack_conv_weight mode 12
 * (forward, model/tsrn.py:623) / mode 13 (data gradient of block1's 4->64 convolution, :597); geometry as the _mfma entry */
int tatt_conv9_c64_to_c4_sb(const float* x, const float* wt, const float* bias, float* y, int B, int H, int W,
                            hipStream_t st);
/* 9x9 convolution from 4 channels to 64 (weight-stationary MFMA kernel, k = the 4 input channels): y = act(conv(in, wp) + bias),
 * wp [81][4][64] = tatt_repack_conv_weight mode 0 of block1's filter (reference model/tsrn.py:597) or mode 1 of the 64->4
 * reconstruction filter (its data gradient, :623); in (B,H,W,4), out (B,H,W,64) NHWC contiguous, H % 4 == 0, W % 64 == 0 */
int tatt_conv9_c4_to_c64(const float* in, const float* wp, const float* bias, float* out, int B, int H, int W, int act,
                         hipStream_t st);
/* the same convolution on v_mfma_f32_16x16x32_bf16 with split operands (k = 8 taps x 4 input channels; hi hi + hi lo + lo hi, fp32
 * accumulation: 2^-16 relative per product); same arguments and the same fp32 packed filter (the kernel splits it itself) */
int tatt_conv9_c4_to_c64_sb(const float* in, const float* wp, const float* bias, float* out, int B, int H, int W, int act,
                            hipStream_t st);
/* dw (4,64,9,9) = sum_px x[px+tap][ci] * dy[px][co] on v_mfma_f32_16x16x4_f32 (rows = input channels, columns = (tap, output
 * channel) read Toeplitz-fashion from the dy tile; weight gradient of reference model/tsrn.py:623); H % 4 == 0, W % 64 == 0;
 * part >= min(B*(H/4)*(W/64), 256) * 64 * 336 floats */
int tatt_conv9_c64_c4_wgrad(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                            hipStream_t st);
/* x (B,H,W,4), dy (B,H,W,64) -> dw (64,4,9,9): the same kernel with the two tensors' roles exchanged (weight gradient of block1's
 * 4->64 convolution, reference model/tsrn.py:597); same constraints and workspace */
int tatt_conv9_c4_c64_wgrad(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                            hipStream_t st);
/* the two 9x9 weight gradients above on v_mfma_f32_16x16x32_bf16 with split operands (the contraction runs over pixels: both operands
 * are gathered 8 pixels per lane from the fp32 images and split hi + lo in registers; hi hi + hi lo + lo hi, fp32 accumulation);
 * same arguments, constraints and workspace */
int tatt_conv9_c64_c4_wgrad_sb(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                               hipStream_t st);
int tatt_conv9_c4_c64_wgrad_sb(const float* x, const float* dy, float* dw, float* part, int B, int H, int W,
                               hipStream_t st);

/* ---- reductions / normalisation ------------------------------------------------------------------- */

/* out[c] = scale * sum_m X[m*ld + c] + beta*out[c]  (bias gradients); ws >= 256*C doubles */
int tatt_colsum(const float* X, long ld, int M, int C, float* out, float scale, float beta,
                double* ws, hipStream_t st);

/* train-mode nn.BatchNorm statistics over the M rows (model/tsrn.py:878,886,613; model/stn_head.py:19,51):
 * mean, rstd = 1/sqrt(biased var + eps); running stats updated in place (unbiased var, momentum) when non-NULL.
 * ws >= 256*2*C doubles */
int tatt_bn_stats(const float* X, long ld, int M, int C, float eps, float momentum, float* mean,
                  float* rstd, float* running_mean, float* running_var, double* ws, hipStream_t st);
/* stage 2 of tatt_bn_stats alone: part [G][2][C] doubles (per-block sum, sum of squares; e.g. from tatt_conv3_c64_fwd_ws16_bn) ->
 * mean, rstd, running statistics; scale / shift (nullable): gamma * rstd and beta - mean * gamma * rstd, the folded affine map */
int tatt_bn_stats_finish(const double* part, int G, int C, int M, float eps, float momentum, const float* gamma,
                         const float* beta, float* mean, float* rstd, float* running_mean, float* running_var, float* scale,
                         float* shift, hipStream_t st);
/* eval mode: rstd = 1/sqrt(running_var + eps) */
int tatt_bn_rstd(const float* var, float* rstd, int C, float eps, hipStream_t st);
/* Y = act((X - mean) * rstd * gamma + beta) */
int tatt_bn_apply(const float* X, long ldx, float* Y, long ldy, int M, int C, const float* mean,
                  const float* rstd, const float* gamma, const float* beta, int act, hipStream_t st);
/* backward of tatt_bn_apply (+ batch statistics when training): dX, dgamma, dbeta; sums: 2*C floats scratch;
 * ws >= 256*2*C doubles */
int tatt_bn_bwd(const float* X, long ldx, const float* dY, long lddy, float* dX, long lddx, int M, int C,
                const float* mean, const float* rstd, const float* gamma, const float* beta, int act,
                int training, float* dgamma, float* dbeta, float* sums, double* ws, hipStream_t st);
/* The same backward in pieces (round 4): stage-1 partials alone (part: tatt_bn_bwd_groups(M) x 2 x C doubles: sums of
 * du = dY act'(gamma xhat + beta) and du xhat) -- or produced by tatt_conv3_c64_dgrad_bn_sb's epilogue; */
int tatt_bn_bwd_groups(int M);
int tatt_bn_bwd_partials(const float* X, long ldx, const float* dY, long lddy, int M, int C, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, int act, double* part, hipStream_t st);
/* ... their sum over G groups -> dgamma, dbeta and coef[3][C] = (a, b, c) of dX = a du + b X + c (= gamma rstd (du - mean(du) -
 * xhat mean(du xhat))), which a consumer applies while staging (tatt_conv3_c64_dgrad_bn_sb) ... */
int tatt_bn_bwd_finish(const double* part, int G, int C, int M, const float* mean, const float* rstd, const float* gamma,
                       float* dgamma, float* dbeta, float* coef, hipStream_t st);
/* ... or this kernel materialises: dX = a dU + b X + c (M, C contiguous, C % 4 == 0) */
int tatt_bn_bwd_affine(const float* X, const float* dU, float* dX, int M, int C, const float* coef, hipStream_t st);

/* Y = LayerNorm(A + Bres) * gamma + beta over the last axis (C <= 256), stats[M][2] = (mean, 1/denominator).
 * mode 0 = nn.LayerNorm (biased variance, eps inside the sqrt); mode 1 = the TBSRN variant's own LayerNorm
 * (model/tbsrn.py:23-36: unbiased std, eps added to the std).
 * nn.LayerNorm + the residual add in front of it (model/transformer_v2.py:478-483,826-832,380-387).
 * pdrop > 0: Bres goes through nn.Dropout(pdrop) first (mask = tatt_dropout's for the same seed word, site and flat index):
 * LayerNorm(A + Dropout(Bres)) in one pass. */
int tatt_ln_fwd(const float* A, const float* Bres, float* Y, float* stats, int M, int C,
                const float* gamma, const float* beta, float eps, int mode, float pdrop,
                const unsigned long long* seed, unsigned site, hipStream_t st);
/* dX = d(A + Dropout(Bres)) (the gradient of A); dB (pdrop > 0 only) = the gradient of Bres = Dropout'(dX);
 * part >= ceil(M/64)*2*C floats; ws >= 256*2*C doubles */
int tatt_ln_bwd(const float* A, const float* Bres, const float* dY, const float* stats, float* dX, float* dB, int M,
                int C, const float* gamma, float* dgamma, float* dbeta, float* part, double* ws, float eps, int mode,
                float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st);

/* ---- element-wise ------------------------------------------------------------------------------------ */

/* nn.PReLU() with one shared slope (model/tsrn.py:598,173) */
int tatt_prelu_fwd(const float* x, float* y, const float* alpha, long n, hipStream_t st);
/* part >= ceil(n/256) floats: per-block partial d alpha (sum with tatt_colsum(part, 1, G, 1, ...)) */
int tatt_prelu_bwd(const float* x, const float* dy, float* dx, const float* alpha, long n, float* part,
                   hipStream_t st);
int tatt_act_fwd(const float* x, float* y, long n, int act, hipStream_t st);
/* dx = dy * act'(.)  -- ref is the pre-activation, or the OUTPUT when from_output (relu/tanh only) */
int tatt_act_bwd(const float* ref, const float* dy, float* dx, long n, int act, int from_output,
                 hipStream_t st);
/* y = alpha*a + beta*b (b may be NULL) */
int tatt_axpby(const float* a, const float* b, float* y, float alpha, float beta, long n, hipStream_t st);
/* y = ((s0 + s1) + s2) + ... : n <= 8 contiguous, 16-byte aligned tensors of numel floats (srcs: HOST array of n device pointers);
 * the gradient of a tensor with several consumers, which autograd sums with a chain of at::add */
int tatt_add_n(const float* const* srcs, int n, float* y, long numel, hipStream_t st);
/* y[m,:] = a[m,:] + b[m % period,:]  (positional embedding broadcast over the batch) */
int tatt_add_rowbcast(const float* a, const float* b, float* y, long rows, int C, long period,
                      hipStream_t st);
/* nn.PixelShuffle(2) + activation on NHWC maps: out[b,2h+i,2w+j,c] = act(in[b,h,w,4c+2i+j]) (model/tsrn.py:1045-1052) */
int tatt_pixel_shuffle_fwd(const float* in, float* out, int B, int H, int W, int C, int act,
                           hipStream_t st);
int tatt_pixel_shuffle_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C,
                           int act, hipStream_t st);
/* nn.MaxPool2d(kernel (kh,kw), stride (sh,sw), padding (ph,pw)) on NHWC maps (model/stn_head.py:36-44; model/crnn/crnn.py:57-69);
 * out is (B, (H+2ph-kh)/sh+1, (W+2pw-kw)/sw+1, C); the gradient goes to the first maximum of each window */
int tatt_maxpool_fwd(const float* in, float* out, int B, int H, int W, int C, int kh, int kw, int sh, int sw,
                     int ph, int pw, hipStream_t st);
int tatt_maxpool_bwd(const float* in, const float* dout, float* din, int B, int H, int W, int C, int kh,
                     int kw, int sh, int sw, int ph, int pw, hipStream_t st);
/* nn.Dropout(p): y = keep ? x/(1-p) : 0 with a counter-based mask keyed by (*seed, site, index); the same call
 * with dy as x is the backward (model/transformer_v2.py:27,456,461-462,789,795-797) */
int tatt_dropout(const float* x, float* y, long n, float p, const unsigned long long* seed, unsigned site,
                 hipStream_t st);
/* advance the device-resident seed word and (snap != NULL) copy the new value to `snap`: one call per TRAINING FORWARD; every
   dropout site of that forward and of its backward reads the snapshot (graph-replay safe; several forwards may be in flight) */
int tatt_bump_seed(unsigned long long* seed, unsigned long long* snap, hipStream_t st);
/* out[0] = 100 MHz wall clock at this point of the stream (a one-thread launch): timelines inside a replayed hipGraph (tooling) */
int tatt_stamp(unsigned long long* out, hipStream_t st);
/* dst[i0*d0+i1*d1+i2*d2+i3*d3] = src[i0*s0+...] + beta*dst  (layout changes, parameter gathers) */
int tatt_copy4d(const float* src, float* dst, int n0, int n1, int n2, int n3, long s0, long s1, long s2,
                long s3, long d0, long d1, long d2, long d3, float beta, hipStream_t st);

/* ---- optimiser (flat buffers; step-varying scalars live in device memory => hipGraph-replayable) --------- */

/* out[0] = ||g||_2 ; ws >= 1024 doubles  (torch.nn.utils.clip_grad_norm_, interfaces/super_resolution.py:1083-1084) */
int tatt_l2norm(const float* g, long n, float* out, double* ws, hipStream_t st);
/* p -= Adam(g * gscale * min(1, max_norm/(gnorm*gscale + 1e-6))); gnorm: device float; step: device int64, 1-based
 * (torch.optim.Adam(lr, betas=(0.5,0.999)), interfaces/base.py:527, config/super_resolution.yaml:26-29) */
int tatt_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                   float eps, const float* gnorm, float max_norm, float gscale, const long long* step,
                   hipStream_t st);

/* ---- GRU recurrences --------------------------------------------------------------------------------- */

/* Bidirectional GRU recurrence, hidden 32 (nn.GRU(64,32,bidirectional), model/tsrn.py:1072) given the input
 * projection gi[tok][192] = [fwd r,z,n | rev r,z,n]; out[tok][64] = [fwd h | rev h].
 * token(s,t) = (s / s_in)*stride_hi + (s % s_in)*stride_lo + t*stride_t on the NHWC token grid: vertical scan
 * (gru1): s_in=W, stride_hi=H*W, stride_lo=1, stride_t=W, T=H; horizontal (gru2): s_in=1, stride_hi=W, stride_t=1, T=W. */
int tatt_gru32_fwd(const float* gi, const float* whh_f, const float* bhh_f, const float* whh_r,
                   const float* bhh_r, float* out, float* gates, int nseq, int T, int s_in, long stride_hi,
                   long stride_lo, long stride_t, hipStream_t st);
/* gates (may be NULL: inference): [tok][256] = per direction {r, z, n, W_hn h + b_hn} x 32, what autograd keeps of
 * torch.nn.GRU's forward for its backward.
 * BPTT from those: writes dgi[tok][192], dgh[tok][192] (recurrent-side gate gradients) and hprev[tok][64] */
int tatt_gru32_bwd(const float* gates, const float* out, const float* dout, const float* whh_f,
                   const float* whh_r, float* dgi, float* dgh, float* hprev, int nseq, int T, int s_in,
                   long stride_hi, long stride_lo, long stride_t, hipStream_t st);

/* Second generation of the two recurrences above (round 4; same reference lines, same arguments, same results up to the order of
 * fp32 summation): one WAVE per (sequence, direction), two lanes per hidden unit, halves joined by v_permlane32_swap. */
int tatt_gru32_fwd2(const float* gi, const float* whh_f, const float* bhh_f, const float* whh_r,
                    const float* bhh_r, float* out, float* gates, int nseq, int T, int s_in, long stride_hi,
                    long stride_lo, long stride_t, hipStream_t st);
/* frag == NULL: dgi, dgh, hprev as tatt_gru32_bwd.  frag != NULL (T % 8 == 0 and nseq * T / 8 % 4 == 0, else returns 1): dgi as
 * before; instead of dgh / hprev (unused, may be NULL) the operands of the weight-gradient pass in bf16 hi / lo MFMA fragment
 * order, nseq * T / 32 K-steps of 10240 floats: [K-step][slot 0..19][hi, lo][lane][4 dwords], slots d*8 + {r0 r1 z0 z1 n0 n1 gn0
 * gn1}, 16 + d*2 + {0, 1} = h_{t-1}; consumed by tatt_gru_wgrad_frag */
int tatt_gru32_bwd2(const float* gates, const float* out, const float* dout, const float* whh_f,
                    const float* whh_r, float* dgi, float* dgh, float* hprev, float* frag, int nseq, int T, int s_in,
                    long stride_hi, long stride_lo, long stride_t, hipStream_t st);
/* Weight gradients of one GruBlock (as tatt_gru_wgrad_sb; reference model/tsrn.py:1075-1084) from that fragment stream and the
 * token-major inputs x (M, 64) | xb (M, 64 or NULL) under the same sequence geometry.  1 <= G <= min(nseq * T / 32, 256);
 * ws1 >= G*192*K + G*192 floats (K = 128 with xb, else 64), ws2 >= G*192*32 + G*192; finish with
 * tatt_splitk_reduce(ws1, dWp, 192, K, G, 0, 0, 0, dbp, 192) and tatt_splitk_reduce(ws2, dWhh_c, 192, 32, G, 0, 0, 0, dbhh, 192)
 * where dWhh_c (192, 32) = [dW_hh forward; dW_hh reverse] */
int tatt_gru_wgrad_frag(const float* frag, const float* x, const float* xb, float* ws1, float* ws2, int nseq, int T,
                        int s_in, long stride_hi, long stride_lo, long stride_t, int G, hipStream_t st);

/* backward step, fused form: dh = dhseq_next + dhcarry + dgh_cur @ whh, then the gate part of the NEXT step on the same tile:
 * dgi_acc += input-side gate grads, dgh_next = recurrent-side gate grads, dhcarry = dh*z  (whhT = whh transposed) */
int tatt_qgru_bwd_fused(const float* dgh_cur0, const float* dgh_cur1, const float* whhT0, const float* whhT1,
                        const float* dhseq_next0, const float* dhseq_next1, const float* gsave_next0,
                        const float* gsave_next1, const float* hprev_next0, const float* hprev_next1,
                        float* dhcarry0, float* dhcarry1, float* dgi_acc0, float* dgi_acc1, float* dgh_next0,
                        float* dgh_next1, int Wb, int HID, hipStream_t st);

/* GruBlock glue: compose the 1x1 conv (Wc (64,K), bc) with the GRU input projections into Wp (192,K) = [wih_f; wih_r] Wc,
 * bp (192) = [wih_f; wih_r] bc + [bih_f; bih_r]  (reference GruBlock.forward, model/tsrn.py:1075-1081) */
int tatt_gru_compose(const float* wih_f, const float* wih_r, const float* bih_f, const float* bih_r,
                     const float* Wc, const float* bc, float* Wp, float* bp, int K, hipStream_t st);
/* tatt_gru_compose for n GruBlocks in one launch (a generator composes all of its blocks once per forward): ptrs = HOST array of
 * n x 8 device pointers (wih_f, wih_r, bih_f, bih_r, Wc, bc, Wp, bp), Ks = HOST array of the n conv input widths */
int tatt_gru_compose_batch(const float* const* ptrs, const int* Ks, int n, hipStream_t st);
/* ... and map the gradients of the composed projection back: dwih_d (96,64) = dWp_d Wc^T + dbp_d bc^T,
 * dWc (64,K) = sum_d wih_d^T dWp_d, dbc (64) = sum_d wih_d^T dbp_d; dwhh_d (96,32) = d-th diagonal block of dWhh (192,64) */
int tatt_gru_tail(const float* dWp, const float* dbp, const float* Wc, const float* bc, const float* wih_f,
                  const float* wih_r, float* dwih_f, float* dwih_r, float* dWc, float* dbc, int K,
                  const float* dWhh, float* dwhh_f, float* dwhh_r, hipStream_t st);

/* tatt_gru_tail with dWhh given compact: dWhh_c (192, 32) = [dW_hh forward; dW_hh reverse] (tatt_gru_wgrad_frag) */
int tatt_gru_tail_c(const float* dWp, const float* dbp, const float* Wc, const float* bc, const float* wih_f,
                    const float* wih_r, float* dwih_f, float* dwih_r, float* dWc, float* dbc, int K,
                    const float* dWhh_c, float* dwhh_f, float* dwhh_r, hipStream_t st);

/* One time step of the query-embedding GRU (nn.GRU(64*H, 32*H, bidirectional), model/transformer_v2.py:177,218;
 * time axis = sample axis), both directions: gi* (Wb,3*HID) incl. b_ih, whh* (3*HID,HID),
 * hprev* (Wb,HID) or NULL for h=0, hnew* (Wb,HID), gsave* (4,Wb,HID) = r,z,n,(W_hn h+b_hn) or NULL. */
int tatt_qgru_fwd_step(const float* gi0, const float* gi1, const float* whh0, const float* whh1,
                       const float* bhh0, const float* bhh1, const float* hprev0, const float* hprev1,
                       float* hnew0, float* hnew1, float* gsave0, float* gsave1, int Wb, int HID,
                       hipStream_t st);
/* backward step, gate part: dh = dhseq + dhcarry (dhcarry ignored when first); dgi_acc (+)= input-side gate grads;
 * dgh = recurrent-side gate grads; dhcarry = dh*z */
int tatt_qgru_bwd_gates(const float* dhseq0, const float* dhseq1, const float* gsave0, const float* gsave1,
                        const float* hprev0, const float* hprev1, float* dhcarry0, float* dhcarry1,
                        float* dgi_acc0, float* dgi_acc1, float* dgh0, float* dgh1, int Wb, int HID, int first,
                        hipStream_t st);
/* backward step, matmul part: dhcarry (Wb,HID) += dgh (Wb,3*HID) @ whh (3*HID,HID); whhT* = whh transposed (HID,3*HID) */
int tatt_qgru_bwd_mm(const float* dgh0, const float* dgh1, const float* whhT0, const float* whhT1,
                     float* dhcarry0, float* dhcarry1, int Wb, int HID, hipStream_t st);

/* The backward recurrence above as ONE persistent launch: fused steps s0 .. s1-1 of the T-1 (step s = tatt_qgru_bwd_fused with the
 * forward direction at time T-1-s and the reverse direction at time s; model/transformer_v2.py:201-221 backward).  A work-group keeps
 * its tile's W_hh^T slice, dhcarry and dgi accumulators in registers; dgh crosses work-groups per step through write-through (sc1)
 * stores and per-(row block, direction) flag words -- no grid-wide barrier.  dgh* (T, Wb, 3*HID): slot T-1 (dir 0) / 0 (dir 1) filled
 * by tatt_qgru_bwd_gates(first = 1), which also initialises dhcarry* and dgi_acc*; hbuf0 + t*Wb*HID = h_prev of time t (dir 0),
 * hbuf1 + (t+1)*Wb*HID = h_prev of time t (dir 1); gsave* (T, 4, Wb, HID); dhseq* (T, Wb, HID).  sync: 1024 words (zeroed here when
 * s0 == 0); sync[1023] != 0 afterwards: a wall-clock-bounded spin expired (results invalid).  Returns 1 for geometries it does not
 * take (HID != 512, Wb % 16 != 0, more than 256 work-groups): use the per-step entry points then.  whhT*: W_hh^T (HID, 3*HID) with
 * transposed != 0, W_hh itself (3*HID, HID) with 0 (no transposition launch needed: the slices are read once per launch).
 * xch* non-NULL (workspace of T*Wb*3*HID floats per direction): the recurrent product runs on the bf16 matrix cores with split operands
 * (hi hi + hi lo + lo hi, fp32 accumulation) and the exchanged tensor travels in operand form; NULL: exact fp32 products. */
int tatt_qgru_bwd_chain(float* dgh0, float* dgh1, const float* whhT0, const float* whhT1, const float* dhseq0,
                        const float* dhseq1, const float* gsave0, const float* gsave1, const float* hbuf0,
                        const float* hbuf1, float* dhcarry0, float* dhcarry1, float* dgi_acc0, float* dgi_acc1,
                        unsigned* sync, int T, int Wb, int HID, int s0, int s1, int transposed, float* xch0, float* xch1,
                        hipStream_t st);

/* Recurrent weight gradient of the query GRU after the backward recurrence (the gradient nn.GRU's autograd accumulates over the time
 * steps, model/transformer_v2.py:201-221 backward): for both directions d in one launch dW_d (N x K) = A_d^T B_d and db_d (N) = column
 * sums of A_d, with A_d = dgh of direction d (M = T*Wb tokens, N = 3*HID) and B_d = h_prev of the same tokens (M, K = HID), contiguous;
 * split-bf16 products (hi hi + hi lo + lo hi, fp32 accumulation) on the bf16 matrix cores.  M % 32 == 0, N % 128 == 0, K % 128 == 0;
 * the contraction is split S ways (1 <= S <= M / 32); ws_d >= S*N*K + S*N floats.  Leaves the per-split partials: finish each
 * direction with tatt_splitk_reduce(ws_d, dW_d, N, K, S, 0, 0, 0, db_d, N).  Returns 1 / 2 for geometries it does not take. */
int tatt_qgru_wgrad_sb(const float* A0, const float* A1, const float* B0, const float* B1, float* ws0, float* ws1, int M, int N,
                       int K, int S, hipStream_t st);

/* The forward recurrence (T calls of tatt_qgru_fwd_step) as one persistent launch, time steps s0 .. s1-1 (direction 0 at time s,
 * direction 1 at time T-1-s), same hand-off as tatt_qgru_bwd_chain with h as the exchanged tensor.  hbuf* (T+1, Wb, HID): h of time t
 * at slot t+1 (direction 0; slot 0 zero) / slot t (direction 1; slot T zero), zero slots filled by the caller; gsave* (T, 4, Wb, HID)
 * or NULL; gi* (Wb, 3*HID) incl. b_ih, or NULL: the launch computes the projection itself from x (Wb, IN), wih* (3*HID, IN), bih*
 * (IN % 1024 == 0); q (T, 2*HID/C, Wb, C) or NULL: h also in the (sample, H, W, C) layout of the query embedding.
 * xch* non-NULL (workspace of (T+1)*Wb*HID floats per direction, s0 == 0): split-bf16 form as for tatt_qgru_bwd_chain.
 * model/transformer_v2.py:201-221. */
int tatt_qgru_fwd_chain(const float* gi0, const float* gi1, const float* whh0, const float* whh1, const float* bhh0,
                        const float* bhh1, float* hbuf0, float* hbuf1, float* gsave0, float* gsave1, unsigned* sync,
                        int T, int Wb, int HID, int s0, int s1, const float* x, const float* wih0, const float* wih1,
                        const float* bih0, const float* bih1, int IN, float* q, int C, float* xch0, float* xch1, hipStream_t st);

/* ---- STN head glue (model/stn_head.py:25-106): everything between two of its convolutions as ONE launch ------------------------
 * The launches below synchronise their (<= 128) work-groups INSIDE the launch: partial sums leave through write-through stores, each
 * work-group raises a flag word and waits for the others'.  `sync` is a 256-word buffer per call site, zeroed ONCE by the caller when
 * it is allocated and then reused (word 254 = epoch, word 255 = error: a wall-clock-bounded spin expired, results invalid); launches
 * that share a sync buffer must not overlap. */

/* A (B,H/ph,W/pw,C) = maxpool_{ph x pw}(relu(batchnorm_train(X))), X (B,H,W,C) contiguous; mean, rstd (C) out; running statistics
 * updated (NULL: not); part >= 128*2*C doubles.  nn.BatchNorm2d + nn.ReLU + nn.MaxPool2d of stn_head.py:9-15,33-49.  ph, pw in {1,2},
 * C in {32, 64, 128, 256} and at most 128 * 4 * 1024 / C pool windows, else 1. */
int tatt_stn_bn_pool_fwd(const float* X, float* A, const float* gamma, const float* beta, float* mean, float* rstd,
                         float* running_mean, float* running_var, double* part, unsigned* sync, int B, int H, int W, int C,
                         int ph, int pw, float eps, float momentum, hipStream_t st);
/* the same with the producing convolution's split contraction folded in: Xparts = S partial maps (S,B,H,W,C) as
 * tatt_conv2d_fwd_partials leaves them; X = their sum in slab order + bias (NULL: none; tatt_splitk_reduce adds the same terms four-way
 * interleaved: last-bit differences) is written to Xout (the backward reads it) and normalised + pooled into A */
int tatt_stn_bn_pool_fwd_parts(const float* Xparts, int S, const float* bias, float* Xout, float* A, const float* gamma,
                               const float* beta, float* mean, float* rstd, float* running_mean, float* running_var,
                               double* part, unsigned* sync, int B, int H, int W, int C, int ph, int pw, float eps,
                               float momentum, hipStream_t st);
/* its backward: dA -> dX (gradient w.r.t. X), dgamma, dbeta, dbias = column sums of dX (the producing convolution's bias gradient;
 * NULL: skipped).  The pooled gradient goes to the first maximum of its window (tatt_maxpool_bwd's rule).  part >= 128*3*C doubles. */
int tatt_stn_bn_pool_bwd(const float* X, const float* dA, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, float* dX, float* dgamma, float* dbeta, float* dbias, double* part,
                         unsigned* sync, int B, int H, int W, int C, int ph, int pw, hipStream_t st);
/* tatt_stn_bn_pool_bwd with dA given as S partial maps (S,B,H/ph,W/pw,C) of the data-gradient convolution behind it (summed in
 * slab order while they are loaded; S = 1: the map itself) */
int tatt_stn_bn_pool_bwd_parts(const float* X, const float* dAparts, int S, const float* gamma, const float* beta,
                               const float* mean, const float* rstd, float* dX, float* dgamma, float* dbeta, float* dbias,
                               double* part, unsigned* sync, int B, int H, int W, int C, int ph, int pw, hipStream_t st);
/* x.view(B,-1) of the (B,256,1,2) map [given NHWC: A6 (B,2,256)] -> Linear(512,512) -> BatchNorm1d(train) -> ReLU -> x0.1 ->
 * Linear(512,NO): U (B,512) = first Linear's output, S (B,512) = the second Linear's input, ctrl (B,NO); W1 (512,512), W2 (NO,512)
 * row-major [out][in]; part >= 32*B*NO floats.  stn_head.py:51-58,96-106.  B <= 64, NO % 4 == 0, NO <= 64, else 1. */
int tatt_stn_fc_fwd(const float* A6, const float* W1, const float* b1, const float* g1, const float* be1, float* rm1,
                    float* rv1, const float* W2, const float* b2, float* U, float* mean1, float* rstd1, float* S,
                    float* ctrl, float* part, unsigned* sync, int B, int NO, float eps, float momentum, hipStream_t st);
/* its backward incl. every parameter gradient: dW2 (NO,512), db2, dgamma1, dbeta1, dW1 (512,512), db1, dA6 (B,2,256); dU (B,512)
 * is workspace (the tensor the work-groups exchange). */
int tatt_stn_fc_bwd(const float* dctrl, const float* W2, const float* S, const float* U, const float* mean1,
                    const float* rstd1, const float* g1, const float* W1, const float* A6, float* dW2, float* db2,
                    float* dg1, float* dbe1, float* dW1, float* db1, float* dU, float* dA6, unsigned* sync, int B, int NO,
                    hipStream_t st);

/* ---- attention core ------------------------------------------------------------------------------------ */

/* ctx = dropout(softmax(Q K^T)) V per head (E=64, 4 heads, S <= 32), wavg = head-mean of the dropped
 * probabilities (may be NULL).  Q (B,Lq,64) already projected and scaled; K,V (B,S,64).
 * Core of nn.MultiheadAttention (model/transformer_v2.py:472-474,821-824). */
int tatt_attn_fwd(const float* Q, const float* K, const float* V, float* ctx, float* wavg, int B, int Lq,
                  int S, float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st);
/* part >= B*ceil(Lq/64)*2*S*64 floats; dwavg may be NULL */
int tatt_attn_bwd(const float* Q, const float* K, const float* V, const float* dctx, const float* dwavg,
                  float* dQ, float* dK, float* dV, float* part, int B, int Lq, int S, float pdrop,
                  const unsigned long long* seed, unsigned site, hipStream_t st);

/* Row softmax over materialised attention scores (TBSRN FeatureEnhancer self-attention, model/tbsrn.py:130-151):
 * S (rows x L, L <= 4096) is overwritten by softmax(S); Pd (nullable) receives dropout(softmax(S)) */
int tatt_softmax_rows_fwd(float* S, float* Pd, long rows, int L, float pdrop, const unsigned long long* seed,
                          unsigned site, hipStream_t st);
/* dP (gradient w.r.t. the dropped probabilities) is overwritten by the gradient w.r.t. the scores */
int tatt_softmax_rows_bwd(const float* P, float* dP, long rows, int L, float pdrop, const unsigned long long* seed,
                          unsigned site, hipStream_t st);

/* ---- token-matrix projections on the bf16 matrix cores (csrc/tokgemm.hip) ------------------------------------------ */

/* Y (M x N) = [X1 | X2] (M x K) W^T + bias by operand splitting (hi + lo bf16, three products, fp32 accumulation; same arithmetic as
 * tatt_conv3_c64_fwd_sb): the GRU input projections of the GruBlocks (reference model/tsrn.py:1075-1084) and their data gradients.
 * Wp = packed operand from tatt_tokgemm_pack (N*K 32-bit words); X1 (M, K1), X2 (M, K - K1) (NULL when K1 == K); output columns
 * [0, N1) go to Y1 (M, N1), the rest to Y2 (M, N - N1) (NULL when N1 == N).  M a multiple of 64; (N, K) one of (192, 128), (192, 64),
 * (128, 192), (64, 192), (64, 64), (64, 128). */
int tatt_tokgemm_sb(const float* X1, const float* X2, int K1, const float* Wp, const float* bias, float* Y1, float* Y2, int N1,
                    int M, int N, int K, hipStream_t st);
/* The same with an epilogue: act = 1 (ReLU) after the bias; accum != 0: Y += instead of Y = (sums of data gradients into one map).
 * Also takes (N, K) = (128, 128) and (128, 64): the TBSRN FeatureEnhancer projections (reference model/tbsrn.py:77-151). */
int tatt_tokgemm_sb_ex(const float* X1, const float* X2, int K1, const float* Wp, const float* bias, float* Y1, float* Y2, int N1,
                       int M, int N, int K, int act, int accum, hipStream_t st);
/* The epilogues of a position-wise feed-forward w_2(Dropout(relu(w_1 x))) (reference PositionwiseFeedForward, model/tbsrn.py:154-164):
 * Y = Dropout_pdrop(act(X Wp^T + bias)) with the mask tatt_dropout draws for Y's flat index (forward of w_1), or, with `gate`,
 * Y = (X Wp^T) * gate_scale where gate > 0, else 0 (data gradient of w_2 gated by the saved forward output F: F > 0 exactly where the
 * unit was active and kept, gate_scale = 1 / (1 - pdrop)).  One source, one destination; shapes of tatt_tokgemm_sb_ex. */
int tatt_tokgemm_sb_ffn(const float* X, const float* Wp, const float* bias, float* Y, int M, int N, int K, int act, float pdrop,
                        const unsigned long long* seed, unsigned site, const float* gate, float gate_scale, hipStream_t st);
/* Y = X Wp^T + bias + addend: addend (M, N) contiguous and left intact, Y must not alias it; (N, K) within 128 x 128.  The sum of a
 * data gradient and the gradient a residual connection carries (TBSRN FeatureEnhancer sub-layers) without an element-wise launch. */
int tatt_tokgemm_sb_add(const float* X, const float* Wp, const float* bias, const float* addend, float* Y, int M, int N, int K,
                        hipStream_t st);
/* trans = 0: w(n, k) = W[n*ldw + k] (y = x W^T);  trans = 1: w(n, k) = W[k*ldw + n] (dx = dy W).  out: N*K words */
int tatt_tokgemm_pack(const float* W, float* out, int N, int K, int ldw, int trans, hipStream_t st);
/* n packs in one launch: ptrs = HOST array of n x 2 device pointers (W, out), dims = HOST array of n x 4 ints (N, K, ldw, trans) */
int tatt_tokgemm_pack_batch(const float* const* ptrs, const int* dims, int n, hipStream_t st);

/* Weight gradient of a token projection y = x W^T + b on the bf16 matrix cores with split operands (csrc/tokwgrad.hip): per-split
 * partials of dW (N x K) = A^T B and db (N) = column sums of A for A = dY (M, N), B = X (M, K) contiguous, M % 32 == 0, N, K in
 * {64, 128}, 1 <= S <= M / 32 (every split owns at least one 32-token chunk); ws >= S*N*K + S*N floats.  Finish with
 * tatt_splitk_reduce(ws, dW, N, K, S, 0, 0, 0, db, N).  The nn.Linear weight gradients of the TBSRN FeatureEnhancer (reference
 * model/tbsrn.py:77-164). */
int tatt_tok_wgrad_sb(const float* A, const float* B, float* ws, int M, int N, int K, int S, hipStream_t st);

/* ---- score-free self-attention of the TBSRN FeatureEnhancer (csrc/sattn.hip) ---------------------------------- */

/* O (B,P,E) = dropout_{pdrop}(softmax(Q K^T * scale)) V per head of 32 channels (E = 32 h, P a multiple of 64): reference
 * `attention` / MultiHeadedAttention (model/tbsrn.py:96-151) without materialising the (B,h,P,P) scores; lse (B,h,P) receives the
 * log-sum-exp of the scaled scores per query (the only thing the backward needs besides Q, K, V, O).  Dropout masks = those of
 * tatt_softmax_rows_fwd for the same seed word and site (flat index ((b h + head) P + q) P + key). */
int tatt_sattn_fwd(const float* Q, const float* K, const float* V, float* O, float* lse, int B, int P, int h, float scale,
                   float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st);
/* gradients of the same (probabilities recomputed tile by tile, no atomics: deterministic); Dws: workspace of B*h*P floats */
int tatt_sattn_bwd(const float* Q, const float* K, const float* V, const float* O, const float* lse, const float* dO,
                   float* dQ, float* dK, float* dV, float* Dws, int B, int P, int h, float scale, float pdrop,
                   const unsigned long long* seed, unsigned site, hipStream_t st);
/* The same pair with the dropout keep decisions handed from the forward to the backward as bits instead of being recomputed in all three
 * kernels (the counter hash is 19 of the ~30 VALU issue slots a score costs): `bits` = B h P P / 32 words the forward fills (layout:
 * csrc/sattn2.hip) and the backward of the same call reads; null = recompute (= the entries above).  Only the split-bf16 kernels use
 * them; whatever runs, a backward must be given what its forward was given.  Same masks, same results as the entries above. */
int tatt_sattn_fwd_bits(const float* Q, const float* K, const float* V, float* O, float* lse, unsigned* bits, int B, int P, int h,
                        float scale, float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st);
int tatt_sattn_bwd_bits(const float* Q, const float* K, const float* V, const float* O, const float* lse, const float* dO,
                        const unsigned* bits, float* dQ, float* dK, float* dV, float* Dws, int B, int P, int h, float scale,
                        float pdrop, const unsigned long long* seed, unsigned site, hipStream_t st);
/* Test / A-B hook: 2 (default) = the split-bf16 kernels (csrc/sattn2.hip: three v_mfma_f32_32x32x16_bf16 products of hi / lo halves per
 * fp32 product; P % 128 == 0, B h P^2 < 4e9), 1 = the exact-fp32 MFMA kernels (also what other geometries run).  Returns the previous
 * setting. */
int tatt_sattn_generation(int gen);

/* ---- one TP-interpreter transformer layer as ONE kernel (csrc/tplayer.hip) --------------------------------- */

/* Reference TransformerDecoderLayer_TP.forward_post (model/transformer_v2.py:806-833: cross-attention over the S <= 32 projected
 * keys / values of the text prior -> +residual -> LayerNorm -> FFN -> +residual -> LayerNorm; the self-attention is commented
 * out upstream) and TransformerEncoderLayer.forward_post (:470-484), E = 64, 4 heads, dim_ff = 64:
 *   q    = ((x + qpos) in_w[0:64]^T + in_b[0:64]) / 4          qpos: (B,L,64) [qbs = L*64] or (L,64) broadcast [qbs = 0]
 *   ctx  = dropout_{p_attn}(softmax(q K^T)) V per head,        wavg (B,L,S; nullable) = head mean of the dropped probabilities
 *   x1   = LN_A(x + dropout_{p_res}(ctx out_w^T + out_b))
 *   xout = LN_B(x1 + dropout_{p_res}(w2 dropout_{p_ffn}(relu(w1 x1 + b1)) + b2))                        (nullable)
 *   fin  = fin_scale * (LN_F(x) [if fin_both] + LN_F(xout))   when lnF_w != NULL: the stacked final norms of
 *          TransformerDecoder.forward (:380-390) averaged as TPInterpreter does (model/tsrn.py:218)       (nullable)
 * K, V: (B,S,64) = the key / value rows of the packed in-projection applied beforehand.  Dropout sites site0 .. site0+3 draw the
 * masks of tatt_attn_fwd / tatt_ln_fwd / tatt_dropout / tatt_ln_fwd for the same seed word and flat element index. */
int tatt_tplayer_fwd(const float* x, const float* qpos, long qbs, const float* K, const float* V, const float* in_w,
                     const float* in_b, const float* out_w, const float* out_b, const float* w1, const float* b1,
                     const float* w2, const float* b2, const float* lnA_w, const float* lnA_b, const float* lnB_w,
                     const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale, int fin_both, float* xout,
                     float* fin, float* wavg, int B, int L, int S, float p_attn, float p_res, float p_ffn,
                     const unsigned long long* seed, unsigned site0, float eps, hipStream_t st);
/* Backward of the layer, recomputed from x (the forward saves nothing else): upstream gradients dxout (of xout), dfin (of fin;
 * required when lnF_w != NULL), dwavg (of wavg) -- each nullable; dqacc (nullable) is added to dqpos (the next layer's dqpos).
 * Writes dx (B,L,64), dqpos (B,L,64; nullable), and partial records: kvpart (dK / dV per work-group and sample) and ppart
 * (parameter gradients per work-group) -- sizes from tatt_tplayer_geom -- to be summed by the two reducers. */
int tatt_tplayer_bwd(const float* x, const float* qpos, long qbs, const float* K, const float* V, const float* in_w,
                     const float* in_b, const float* out_w, const float* out_b, const float* w1, const float* b1,
                     const float* w2, const float* b2, const float* lnA_w, const float* lnA_b, const float* lnB_w,
                     const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale, int fin_both,
                     const float* dxout, const float* dfin, const float* dwavg, const float* dqacc, float* dx,
                     float* dqpos, float* kvpart, float* ppart, int B, int L, int S, float p_attn, float p_res,
                     float p_ffn, const unsigned long long* seed, unsigned site0, float eps, hipStream_t st);
/* host-side: out[0] = work-groups, out[1] = dK/dV records per work-group, out[2] = floats of kvpart, out[3] = floats of ppart */
int tatt_tplayer_geom(int B, int L, int* out);
/* dK, dV (B,S,64) = sums of the kvpart records */
int tatt_tplayer_reduce_kv(const float* kvpart, float* dK, float* dV, int B, int L, int S, hipStream_t st);
/* parameter gradients = sums of the ppart records: d_in_w receives 64x64 (the query rows of the packed in-projection), d_in_b 64;
 * any destination may be NULL; betaF = 1 accumulates into d_lnF_w / d_lnF_b */
int tatt_tplayer_reduce_params(const float* ppart, int B, int L, float* d_in_w, float* d_in_b, float* d_out_w, float* d_out_b,
                               float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_lnA_w, float* d_lnA_b,
                               float* d_lnB_w, float* d_lnB_b, float* d_lnF_w, float* d_lnF_b, float betaF, hipStream_t st);

/* the same reduction over G records (the second-generation backward below has its own work-group count) */
int tatt_tplayer_reduce_params_g(const float* ppart, int G, float* d_in_w, float* d_in_b, float* d_out_w, float* d_out_b,
                                 float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_lnA_w, float* d_lnA_b,
                                 float* d_lnB_w, float* d_lnB_b, float* d_lnF_w, float* d_lnF_b, float betaF, hipStream_t st);

/* ---- second-generation backward of the layer (csrc/tplayer2.hip): split-bf16 products on the bf16 matrix cores, a wave owns 16
 * tokens and chains every product in registers (transposed MFMA orientation), weight gradients shared by the four waves of a
 * work-group.  Same inputs, outputs and dropout masks as tatt_tplayer_bwd; takes L % 64 == 0, S <= 32 (tatt_tplayer2_geom says). */
/* host-side: out[0] = 1 if the geometry is taken, out[1] = work-groups, out[2] = floats of kvpart, out[3] = floats of ppart,
 * out[4] = ints of kvflags, out[5] = 32-bit words of wimg, out[6] = 32-bit words of kvf, out[7] = floats of wimg32, out[8] = floats of
 * kvf32 */
int tatt_tplayer2_geom(int B, int L, int S, int* out);
/* packed operands of one layer, one launch: wimg = the four 64x64 matrices (in_w: query rows of the packed in-projection) as MFMA A
 * fragments, forward and transposed, bf16 hi / lo, and kvf = K, V (B,S,64) as the four fragment forms of the attention products (the
 * backward's split-bf16 operands); wimg32 / kvf32 = the forward's exact-fp32 operands (one float per lane and v_mfma_f32_16x16x4_f32 step) */
int tatt_tplayer2_prep(const float* in_w, const float* out_w, const float* w1, const float* w2, const float* K, const float* V,
                       unsigned* wimg, unsigned* kvf, float* wimg32, float* kvf32, int B, int S, hipStream_t st);
/* the same packing with the key / value projections folded in, ONE launch for nl = 1 or 2 layers over one memory: K = (mem + pos) Wk^T + bk and
 * V = mem Wv^T + bv (Wk / Wv = rows 64..127 / 128..191 of the layer's packed in-projection) are computed per (layer, sample) in fp32 and leave
 * only as the fragment forms above; kin (B,S,64; nullable) receives mem + pos (the backward's weight gradients read it).  pos: (B,S,64) with
 * pos_bs = S*64, or (S,64) with pos_bs = 0.  in_w ... kvf32 are HOST arrays of nl device pointers. */
int tatt_tplayer2_kvprep(const float* mem, const float* pos, long pos_bs, const float* const* in_w, const float* const* in_b,
                         const float* const* out_w, const float* const* w1, const float* const* w2, unsigned* const* wimg,
                         unsigned* const* kvf, float* const* wimg32, float* const* kvf32, float* kin, int B, int S, int nl,
                         hipStream_t st);
/* the layer's forward in the same organisation (training mode), exact fp32 products (v_mfma_f32_16x16x4_f32): arguments as tatt_tplayer_fwd
 * with the matrices / K / V replaced by wimg32 / kvf32; hmask (B*L 64-bit words, nullable) receives the relu-and-kept bits of the FFN's hidden layer -- word [16-token tile][channel
 * block][r], bit = lane -- which tatt_tplayer2_bwd reads instead of re-deciding the relu (a recomputation that differs from the forward in
 * the last bits flips a relu whose pre-activation is within ~1e-5 of zero: O(1) error on that token's gradients) */
int tatt_tplayer2_fwd(const float* x, const float* qpos, long qbs, const float* wimg32, const float* kvf32, const float* in_b,
                      const float* out_b, const float* b1, const float* b2, const float* lnA_w, const float* lnA_b,
                      const float* lnB_w, const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale,
                      int fin_both, float* xout, float* fin, float* wavg, unsigned long long* hmask, int B, int L, int S,
                      float p_attn, float p_res, float p_ffn, const unsigned long long* seed, unsigned site0, float eps,
                      hipStream_t st);
/* arguments as tatt_tplayer_bwd with the matrices / K / V replaced by their packed forms; kvflags: which sample the dK / dV
 * records of a work-group belong to (read by tatt_tplayer2_reduce_kv); ppart is summed by tatt_tplayer_reduce_params_g;
 * hmask: the bits tatt_tplayer2_fwd left (nullable: the relu is then decided by the recomputation) */
int tatt_tplayer2_bwd(const float* x, const float* qpos, long qbs, const unsigned* wimg, const unsigned* kvf, const float* in_b,
                      const float* out_b, const float* b1, const float* b2, const float* lnA_w, const float* lnA_b,
                      const float* lnB_w, const float* lnB_b, const float* lnF_w, const float* lnF_b, float fin_scale,
                      int fin_both, const float* dxout, const float* dfin, const float* dwavg, const float* dqacc, float* dx,
                      float* dqpos, float* kvpart, float* ppart, int* kvflags, const unsigned long long* hmask, int B, int L, int S,
                      float p_attn, float p_res, float p_ffn, const unsigned long long* seed, unsigned site0, float eps,
                      hipStream_t st);
int tatt_tplayer2_reduce_kv(const float* kvpart, const int* kvflags, float* dK, float* dV, int B, int L, int S, hipStream_t st);

/* ---- launches that synchronise their work-groups in flight: residency and the sticky error word -------------------------------- */

/* The persistent query-GRU recurrences (tatt_qgru_fwd_chain / _bwd_chain) and the STN-head launches (tatt_stn_*) exchange data between
 * work-groups INSIDE a launch: their whole grid must be resident at once.  These report, for the CURRENT device, how many work-groups of
 * each kernel fit (hipOccupancyMaxActiveBlocksPerMultiprocessor x the CUs the process sees), so that a caller on a partitioned or
 * CU-masked device takes the per-step / operator-chain path instead:
 *   tatt_qgru_chain_capacity: out[0] forward split-bf16, out[1] forward fp32, out[2] backward split-bf16, out[3] backward fp32
 *                             (a launch needs (Wb / 16) * 2 * (HID / 16) work-groups);
 *   tatt_stn_capacity:        out[0] the map launches (need <= 128), out[1] the fully connected launches (need 32). */
int tatt_qgru_chain_capacity(int* out);
int tatt_stn_capacity(int* out);
/* Every wait inside those launches is bounded by the wall clock (2 s); one that expires raises the launch's own error word AND ORs a code
 * (1: query GRU, 2: STN head) into the device's sticky word -- one zero-initialised 32-bit word in device memory registered here, which the
 * library never resets (NULL unregisters). */
int tatt_set_sticky(unsigned* word);
/* One-thread launch that traps (GPU exception: the process dies) if the sticky word is non-zero; no-op without a registered word.  Issued
 * in front of the optimiser kernels it keeps gradients of a launch that gave up waiting from ever reaching the weights. */
int tatt_sync_guard(hipStream_t st);

/* diagnostic: `groups` work-groups of 512 threads (lds_bytes of LDS each, <= 64 KB) that stay resident for `ticks` of the 100 MHz wall
 * clock (<= 1 s) and do nothing else -- a stand-in for a collective's channel kernels in the residency tests; sink: any device word */
int tatt_cu_holder(int groups, long ticks, int lds_bytes, unsigned* sink, hipStream_t st);

/* ---- small launches that keep the replayed step free of framework kernels ------------------------------------------------------------ */
/* dst[offs[k] .. + ns[k]) <- srcs[k] (zeros where srcs[k] == NULL) for k < count: one bucket of the flat gradient buffer gathered from
 * the parameters' gradient tensors in ONE launch.  srcs / offs / ns are HOST arrays; the table travels in the kernel arguments. */
int tatt_gather_grads(const float* const* srcs, const long* offs, const int* ns, int count, float* dst, hipStream_t st);
/* v[i] += 1 for n 64-bit counters */
int tatt_inc_i64(long long* v, int n, hipStream_t st);
/* y[0 .. n) = 0 */
int tatt_zero_f32(float* y, long n, hipStream_t st);

/* ---- TPS rectification ---------------------------------------------------------------------------------- */

/* src[b,p,:] = repr[p,:] @ (inv @ [ctrl[b]; pad])  (model/tps_spatial_transformer.py:103-105); N ctrl points, P pixels */
int tatt_tps_grid_fwd(const float* ctrl, const float* inv, const float* pad, const float* repr, float* src,
                      int B, int N, int P, hipStream_t st);
int tatt_tps_grid_bwd(const float* dsrc, const float* inv, const float* repr, float* dctrl, int B, int N,
                      int P, hipStream_t st);
/* out (B,H,W,C) = F.grid_sample(x, 2*clamp(src,0,1)-1) bilinear / zeros / align_corners=False
 * (model/tps_spatial_transformer.py:107-111,11); x[b*xsn + c*xsc + h*xsh + w*xsw] */
int tatt_grid_sample_fwd(const float* x, long xsn, long xsc, long xsh, long xsw, const float* src, float* out,
                         int B, int C, int H, int W, hipStream_t st);
/* gradient w.r.t. src (through the clamp); the LR image carries no gradient */
int tatt_grid_sample_bwd(const float* x, long xsn, long xsc, long xsh, long xsw, const float* src,
                         const float* dout, float* dsrc, int B, int C, int H, int W, hipStream_t st);

/* ImageLoss(gradient=True, loss_weight=[w0, w1]).forward (reference loss/image_loss.py:19-34,50-58): per-sample
 * loss[b] = w0*mean((sr-hr)^2) + w1*mean|gradient_map(sr[:3]) - gradient_map(hr[:3])|; when loss_mean != NULL also
 * loss_mean[0] = scale * mean_b loss[b] (interfaces/super_resolution.py:889-894 uses scale 100).  sr / hr are (B,C,H,W) images
 * addressed by element strides (n, c, h, w). */
int tatt_image_loss_fwd(const float* sr, long s_n, long s_c, long s_h, long s_w, const float* hr, long h_n, long h_c,
                        long h_h, long h_w, float* loss, float* loss_mean, float scale, int B, int C, int H, int W,
                        float w0, float w1, hipStream_t st);
/* its gradient w.r.t. sr (dsr has sr's strides); exactly one of gper (B per-sample upstream gradients) and gscalar (upstream
 * gradient of the scaled batch mean, 1 element) is non-NULL */
int tatt_image_loss_bwd(const float* sr, long s_n, long s_c, long s_h, long s_w, const float* hr, long h_n, long h_c,
                        long h_h, long h_w, const float* gper, const float* gscalar, float scale, float* dsr,
                        int B, int C, int H, int W, float w0, float w1, hipStream_t st);

/* CRNN text-prior generator.  Bidirectional nn.LSTM (model/crnn/crnn.py:5-26), one launch per time step for
 * both directions; time-major (T, Bt, *) row-major tensors; gi (T,Bt,8H) = x W_ih^T + b_ih for [fwd | rev] x gates (i,f,g,o);
 * out (T,Bt,2H) = [h_fwd | h_rev] doubles as the h_{t-1} operand; cseq (2,T,Bt,H), gsave (2,T,Bt,4,H) are kept for the
 * backward.  H multiple of 256.  Step s handles time s (fwd) / T-1-s (rev). */
int tatt_lstm_fwd_step(const float* gi, const float* whh_f, const float* whh_r, const float* bhh_f, const float* bhh_r,
                       float* out, float* cseq, float* gsave, int T, int Bt, int H, int s, hipStream_t st);
/* backward step s = T-1..0: dgates (2,T,Bt,4H) receives the pre-activation gate gradients of time(s); whhT_d = W_hh_d^T (H,4H);
 * dccarry (2,Bt,H) scratch carried between steps */
int tatt_lstm_bwd_step(const float* dout, const float* whhT_f, const float* whhT_r, const float* cseq, const float* gsave,
                       float* dgates, float* dccarry, int T, int Bt, int H, int s, hipStream_t st);
/* out (B,OH,OW) = 0.299 R + 0.587 G + 0.114 B of F.interpolate(img[:, :3], (OH,OW), mode='bicubic') (interfaces/base.py:797-815);
 * img (B,C>=3,H,W) by element strides */
int tatt_bicubic_luma(const float* img, long sn, long sc, long sh, long sw, float* out, int B, int H, int W, int OH,
                      int OW, hipStream_t st);

/* SemanticLoss(pred, gt) = mean|gt - pred| + mean (gt+1e-20)(log(gt+1e-20) - log(pred+1e-20)) over n elements (reference
 * loss/semantic_loss.py:21-38: the student / teacher prior distillation loss); out[0] scalar; backward w.r.t. pred */
int tatt_semantic_loss_fwd(const float* pred, const float* gt, long n, float* out, hipStream_t st);
int tatt_semantic_loss_bwd(const float* pred, const float* gt, const float* gout, long n, float* dpred, hipStream_t st);
/* calculate_psnr (reference utils/ssim_psnr.py:9-15) of two (B,C,H,W) images in [0,1] given by element strides, first 3 channels */
int tatt_psnr(const float* a, long a_n, long a_c, long a_h, long a_w, const float* b, long b_n, long b_c, long b_h, long b_w,
              float* out, int B, int C, int H, int W, hipStream_t st);

/* ---- SSIM / TRI_SSIM losses and the rotation augmentation of the shipped recipe (train_TATT.sh: --tssim_loss
 * --rotate_train=5) -------------------------------------------------------------------------------------------------------- */
/* out[n] = mean over (C,H,W) of the SSIM map of sample n -- reference utils/ssim_psnr.py:76-97 (_ssim, x3 == NULL) or :99-129
 * (_tri_ssim); 11x11 Gaussian window (sigma 1.5, :28-37), zero padding, one depth-wise filter per channel.  Images (B,C,H,W) are
 * given by element strides.  ws: B*C*ceil(H/16)*ceil(W/64) doubles. */
int tatt_ssim_fwd(const float* x1, long a_n, long a_c, long a_h, long a_w, const float* x2, long b_n, long b_c, long b_h,
                  long b_w, const float* x3, long c_n, long c_c, long c_h, long c_w, int B, int C, int H, int W,
                  float* out, double* ws, hipStream_t st);
/* gradients of sum_n gs[n] * out[n] w.r.t. the images (contiguous (B,C,H,W) outputs; any of dx1/dx2/dx3 may be NULL).
 * maps: workspace of B*C*5*H*W floats. */
int tatt_ssim_bwd(const float* x1, long a_n, long a_c, long a_h, long a_w, const float* x2, long b_n, long b_c, long b_h,
                  long b_w, const float* x3, long c_n, long c_c, long c_h, long c_w, int B, int C, int H, int W,
                  const float* gs, float* maps, float* dx1, float* dx2, float* dx3, hipStream_t st);
/* out (B,C,H,W contiguous) = F.grid_sample(x, F.affine_grid(theta (B,2,3), x.shape)) -- bilinear, zeros padding,
 * align_corners=False: the resampling of torch_distortion / TextSR.torch_rotate_img (reference model/__init__.py:4-29,
 * interfaces/super_resolution.py:126-157).  x by element strides. */
int tatt_affine_sample_fwd(const float* x, long xsn, long xsc, long xsh, long xsw, const float* theta, float* out, int B,
                           int C, int H, int W, hipStream_t st);
/* gradient w.r.t. the image (deterministic gather; dout, dimg contiguous (B,C,H,W), C <= 4) */
int tatt_affine_sample_bwd(const float* theta, const float* dout, float* dimg, int B, int C, int H, int W,
                           hipStream_t st);

#ifdef __cplusplus
}
#endif
#endif /* TATT_HIP_H */
