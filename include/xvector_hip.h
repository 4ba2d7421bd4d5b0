/*
 * xvector_hip.h -- C ABI of libxvector_hip.so: the MI355X (gfx950) x-vector extraction hot path.
 *
 * This is the drop-in boundary for ONE path of BUTSpeechFIT/x-vector-kaldi-tf: the forward pass that
 * Model.make_embedding (local/tf/models.py:356-432) evaluates once per utterance chunk through
 * sess.run(embed_layer-0/scores) (local/tf/models.py:414).  The reference has no native code and no
 * FFI; what it binds for this path are the stock TensorFlow ops listed next to each entry point
 * below.  A maintainer replaces the sess.run call by these calls (see INTEGRATION.md for the ctypes
 * stub); the Python twin in x-vector-kaldi-tf_amd/local/tf/models.py does exactly that.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (e.g. a torch-ROCm tensor's data_ptr);
 *    the library never allocates, frees or synchronises; work is enqueued on `stream`
 *    (hipStream_t passed as void*; NULL = the null stream).
 *  - return value: 0 on success, otherwise a hipError_t / negative argument-error code;
 *    xv_last_error() returns a thread-local message.  No exceptions cross the ABI.
 *  - arithmetic: three GEMM arithmetics behind the same fp32-in / fp32-out contraction, all accumulating in fp32:
 *      *_f32      exact IEEE fp32 (f32-input MFMA: exact fp32 products) -- the reference's own arithmetic;
 *      *_bf16x3   every operand split x = hi + lo into two bf16, a product formed as hi*hi + hi*lo + lo*hi on the bf16 matrix
 *                 cores (16x the f32-MFMA rate; the dropped lo*lo term is ~2^-16 relative): ~5e-6 relative L2 on the x-vector;
 *      *_f16bf8   hi = fp16(x), the two cross terms through ONE block-scaled e5m2 MFMA at twice the 16-bit rate (csrc/xv_split8.h):
 *                 ~1.3e-5.  This is what the host side runs BY DEFAULT for the hidden frame-level layers (layer 0 and the segment
 *                 FCs stay bf16x3) -- per checkpoint, after an accuracy probe admitted it (xvector_amd/engine.py select_model;
 *                 DESIGN.md 5.1); a model whose f16bf8 results drift from bf16x3 runs as bf16x3, then as f32.
 *    All are inside the 1e-4 parity bar against the fp64 oracle.  Pooling, epilogues and the chunk average are fp32 in all three.
 *
 * Ragged batch layout ("packed rows with gaps")
 *    A batch of utterance chunks is ONE row-major matrix x[R, C].  Chunk b owns rows
 *    [row_start[b], row_start[b]+row_len[b]).  Between consecutive chunks (and before the first /
 *    after the last) the caller leaves >= G all-zero "gap" rows, G = max over layers of
 *    (K-1)*dilation/2.  Because the gaps are zero, a temporal tap that reaches past a chunk's own
 *    first/last frame reads zeros -- exactly tf.nn.conv1d's per-utterance SAME padding at batch 1
 *    (local/tf/models.py:60,410) -- with no per-tap bounds logic in the kernel.  row_valid[R]
 *    (1 = frame, 0 = gap) lets each layer's epilogue write zeros back into the gap rows, so the
 *    invariant holds for the next layer.  Rows outside [0,R) read as zero and are never written.
 */
#ifndef XVECTOR_HIP_H_
#define XVECTOR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XV_ACT_NONE 0
#define XV_ACT_RELU 1   /* tf.nn.relu            local/tf/models.py:64   */
#define XV_ACT_LRELU 2  /* tf.nn.leaky_relu      local/tf/models.py:912  (alpha[0]) */
#define XV_ACT_PRELU 3  /* tf_block.prelu        local/tf/tf_block.py:38-47 (alpha[Cout]) */

#define XV_ERR_BAD_ARG (-1)
#define XV_ERR_UNSUPPORTED (-2)

/* Library / ABI version (increments whenever an entry point is added or changed; currently 23). */
int xv_version(void);
/* Thread-local description of the last non-zero return. */
const char *xv_last_error(void);
/* Process-wide launch tuning (no TF counterpart; the analogue of the session's ConfigProto knobs,
 * local/tf/models.py:361-363).  Results do not depend on any of these.
 *   XV_TUNE_TILE_ROWS  rows per workgroup tile of the bf16x3 / f16bf8 GEMMs with split-format input: 128 (4 waves, two
 *                      workgroups per CU), 256 (8 waves, one per CU), 512 / 1024 (f16bf8: the 256 x 256 tile on the 32 x 32 /
 *                      16 x 16 MFMA shapes where the shape allows it, else as 0; bf16x3: as 256), 0 = built-in choice.
 *   XV_TUNE_FP32_GEMM  form of the exact-fp32 GEMM where several exist (bit-identical results): 1 = register-staged (tdnn_gemm_kernel),
 *                      2 = fed by LDS-DMA on 32-channel slabs (tdnn_gemm_dma_kernel), 3 = as 2 with the K = 1 layers on 16-channel
 *                      slabs, three workgroups per CU (tdnn_gemm_k1_kernel), 0 = built-in choice (3 unless XV_FP32_K1=0 / XV_FP32_DMA=0).
 *   XV_TUNE_XCD_COLUMNS  the 256 x 256-tile f16bf8 GEMM on two column tiles (Cout = 512): 1 = XCDs 0-3 work on column tile 0 and
 *                      XCDs 4-7 on tile 1 (each L2 holds one tile's weights, operand rows are fetched by two XCDs), 0 = every XCD
 *                      works on both column tiles of a contiguous run of row tiles (built-in).
 *   XV_TUNE_FIRST_TILES  16-frame tiles per wave of the first-layer kernel (xv_tdnn_first_*): 1..4096, 0 = spread the rows evenly
 *                      over one workgroup per CU (tests use small values to walk through several groups of tiles on small inputs). */
#define XV_TUNE_TILE_ROWS 1
#define XV_TUNE_FIRST_TILES 2
#define XV_TUNE_FP32_GEMM 3
#define XV_TUNE_XCD_COLUMNS 4
int xv_set_tuning(int key, int value);

/* One-off weight re-layout.  TF stores a conv kernel as w[K, Cin, Cout] == row-major [K*Cin, Cout]
 * (local/tf/models.py:56-57) and an FC weight as w[In, Out] (local/tf/models.py:83); the GEMM kernel
 * wants each output channel's reduction vector contiguous: wp[Cout][K*Cin].  kred = K*Cin (or In). */
int xv_pack_weights_f32(const float *w, int kred, int cout, float *wp, void *stream);

/* Fold tf.nn.batch_normalization's inference form (local/tf/tf_block.py:25-26, epsilon 1e-3 from
 * tf_block.py:9) into a per-channel affine:  scale = gamma*rsqrt(var+eps), shift = beta - mean*scale. */
int xv_fold_bn_f32(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                   int c, float *scale, float *shift, void *stream);

/* Frame-level TDNN layer over a whole ragged batch.  Replaces, per layer,
 *   tf.nn.conv1d(stride 1, SAME) / tf.nn.convolution(dilation_rate)  local/tf/models.py:60, :579-580
 *   + tf.nn.bias_add (:61) + relu|leaky_relu|prelu (:64, :912, tf_block.py:38-47)
 *   + batch_norm_wrapper eval branch (tf_block.py:25-26).
 *     z[r,o]  = bias[o] + sum_{k<K} sum_{c<Cin} x[r + (k-(K-1)/2)*dilation, c] * w[k,c,o]
 *     y[r,o]  = row_valid[r] ? act(z)*bn_scale[o] + bn_shift[o] : 0
 *   x[R,Cin] with row stride ldx; wp = xv_pack_weights_f32(w) i.e. [Cout][K*Cin];
 *   bn_scale/bn_shift may be NULL (identity); act_alpha: [1] for LRELU, [Cout] for PRELU;
 *   row_valid may be NULL (all rows valid); y_preact (row stride ldy) may be NULL, else receives z.
 *   K odd, (K-1)*dilation <= 8.  Implicit-im2col GEMM on v_mfma_f32_32x32x2_f32. */
int xv_tdnn_layer_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias,
                      const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha,
                      int K, int dilation, int cout, const uint8_t *row_valid, float *y, int ldy,
                      float *y_preact, void *stream);

/* The FIRST frame-level layer (models.py:54-67 on the feature rows) as a K = 1 GEMM over OVERLAPPING rows.  The packed feature rows
 * x[R, ldx] lie back to back (ldx = the feature dimension padded to a multiple of 4, padding columns zero), so the K-tap window
 * of frame r IS the contiguous span of K * ldx floats that starts at x + (r - (K-1)/2) * ldx: no im2col, no taps -- the exact-fp32
 * GEMM kernels read "row r" from there (DMA-fed: rows before the buffer or past its end read as zeros through the buffer
 * descriptor's range check).  wp = xv_pack_weights_rows_f32(w[K,Cin,Cout]): [Cout][roundup32(K * ldx)] floats with w[k][c][o] at
 * column k * ldx + c and zeros elsewhere (xv_packed_weights_rows_f32_floats; 0 = unsupported: K odd, Cin <= ldx, ldx % 4 == 0).
 * Same products as xv_tdnn_layer_f32, summed in another order (tap-major): not bit-identical to it; used by the "fp32tc" path.
 * x, y and the per-column parameters 16-byte aligned; gap rows must be zero as everywhere (dilation 1 only). */
size_t xv_packed_weights_rows_f32_floats(int K, int cin, int ldx, int cout);
int xv_pack_weights_rows_f32(const float *w, int K, int cin, int ldx, int cout, float *wp, void *stream);
int xv_tdnn_layer_rows_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                           const float *bn_shift, int act_kind, const float *act_alpha, int K, int cout, const uint8_t *row_valid,
                           float *y, int ldy, void *stream);

/* ---- "fp32tc": the wide-context layers with fewer multiplications (Toom-Cook F(2, K) over time; csrc/xv_toom.hip) -------------
 * Same layer as xv_tdnn_layer_f32 (models.py:54-67) for K in {3, 5, 7} (the default topology's layers 1 and 2, models.py:28 -- 74 % of
 * the network's multiplications; the dilated class's K = 3 layers, models.py:545-548,579-585): two output rows are formed from K + 1
 * products of TRANSFORMED input rows and TRANSFORMED taps instead of 2 K products (0.67 / 0.60 / 0.57 of the MFMA work), exact fp32
 * products, fp32 accumulation.  Not bit-identical to xv_tdnn_layer_f32 (another rounding, same accuracy class), hence its own
 * entry points.
 *   wp = xv_pack_weights_toom_f32(w[K,Cin,Cout]): [Cout][(K+1)*Cin] floats (xv_packed_weights_toom_f32_floats; 0 = unsupported);
 *   xv_toom_supported: K in {3,5,7}, dilation 1..8, Cin % 32 == 0, Cout % 4 == 0; the layer call additionally needs 16-byte aligned
 *   x / y rows and per-column parameters (else XV_ERR_UNSUPPORTED: use xv_tdnn_layer_f32).
 * xv_tdnn_layer_toom_dilated_f32: tf.nn.convolution(dilation_rate = d) (models.py:579-585) as d independent undilated problems over
 * the rows sub, sub + d, sub + 2 d, ... (one launch); its row pairs are rows (r, r + d) with floor(r / d) even.
 * Row pairs sit on fixed global rows: chunks should start on multiples of 2 d rows (then a chunk's bits do not depend on its
 * neighbours); xv_tdnn_layer_toom_f32 is the d = 1 call. */
int xv_toom_supported(int K, int dilation, int cin, int cout);
size_t xv_packed_weights_toom_f32_floats(int K, int cin, int cout);
int xv_pack_weights_toom_f32(const float *w, int K, int cin, int cout, float *wp, void *stream);
int xv_tdnn_layer_toom_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                           const float *bn_shift, int act_kind, const float *act_alpha, int K, int cout, const uint8_t *row_valid,
                           float *y, int ldy, void *stream);
int xv_tdnn_layer_toom_dilated_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                                   const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                                   const uint8_t *row_valid, float *y, int ldy, void *stream);

/* ---- bf16x3 split-precision twins (same contraction, fp32-class accuracy, bf16 matrix cores) ----------------
 *
 * Tensor formats for xv_tdnn_layer_bf16x3:
 *   XV_FMT_F32    row-major fp32, row stride ld (Cin, ld multiples of 4, 16-byte aligned).
 *   XV_FMT_SPLIT  "split" activations: for row r and 32-channel slab s, 128 bytes at
 *                 ((r * ceil(C/32) + s) * 128): 8 slots of 16 B; logical slot t = plane*4 + (k>>3) holds channels
 *                 s*32 + 8*(k>>3) .. +7 of plane hi (0) / lo (1) as bf16; it is stored at physical slot
 *                 t ^ ((r>>1)&7).  value = float(hi) + float(lo).  Same bytes/element as fp32.  The halo tile of a
 *                 layer is then a linear global->LDS DMA.  A split INPUT buffer must be readable (and finite, e.g.
 *                 zero) for rows [-XV_SPLIT_PAD_BEFORE, R + XV_SPLIT_PAD_AFTER) around the pointer to row 0;
 *                 channels >= C inside the last slab are written as zero by the producing layer.
 *   xv_split_row_bytes(C) = ceil(C/32)*128.  xv_split_encode_f32 / xv_split_decode_f32 convert fp32 rows <-> split
 *   (tooling and tests; the layers read/write the format directly).
 * Weights: xv_pack_weights_bf16x3 turns TF's w[K,Cin,Cout] into xv_packed_weights_bf16x3_bytes(K,Cin,Cout) bytes of
 *   16 KB tiles, one per (128-column tile, 32-channel slab, tap) in K-loop order, each [hi 128x64 B][lo 128x64 B]. */
#define XV_FMT_F32 0
#define XV_FMT_SPLIT 1
#define XV_SPLIT_PAD_BEFORE 8
#define XV_SPLIT_PAD_AFTER 264
size_t xv_packed_weights_bf16x3_bytes(int K, int cin, int cout);
int xv_pack_weights_bf16x3(const float *w, int K, int cin, int cout, void *wt, void *stream);
/* The tiles of n layers in ONE launch, each in one or both orientations (the training step re-packs every weight after every
 * optimizer step): wt_fwd[i] = xv_pack_weights_bf16x3 of w[i][K, cin, cout] with cin padded by zero rows to cin_pad[i]
 * (xv_packed_weights_bf16x3_bytes(K, cin_pad, cout) bytes); wt_bwd[i] = the operand of the input-gradient GEMM,
 * w'[k, o, c] = w[K-1-k, c, o] as [K, cout, cin_pad] (xv_packed_weights_bf16x3_bytes(K, cout, cin_pad) bytes) -- what
 * tf.gradients builds for conv1d / xw_plus_b (local/tf/models.py:112).  Either destination of a layer may be NULL. */
int xv_pack_weights_bf16x3_many(int n, const float *const *w, const int32_t *K, const int32_t *cin, const int32_t *cin_pad,
                                const int32_t *cout, void *const *wt_fwd, void *const *wt_bwd, void *stream);
size_t xv_split_row_bytes(int channels);
int xv_split_encode_f32(const float *x, int64_t R, int c, int ldx, void *xs, void *stream);
int xv_split_decode_f32(const void *xs, int64_t R, int c, float *x, int ldx, void *stream);
/* Same semantics as xv_tdnn_layer_f32; x / y in the given formats (ldx / ldy used for XV_FMT_F32 only); y may be
 * NULL when only y_preact (always fp32 rows, stride ldpre) is wanted. */
int xv_tdnn_layer_bf16x3(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                         const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation,
                         int cout, const uint8_t *row_valid, void *y, int y_format, int ldy, float *y_preact, int ldpre,
                         void *stream);
/* The same layer with fp32 rows out, and the column sums of that output for free: the epilogue also leaves, per 128-row tile,
 * [sum_rows y | sum_rows y * sum_r] in double in `workspace` (xv_col_sums_workspace_bytes(R, cout) bytes; merged by
 * xv_col_sums_merge_f32 / xv_bn_act_backward_parts_f32).  The training step calls it as the input-gradient GEMM of a layer
 * (tf.gradients of conv1d, local/tf/models.py:112): y = dL/dh of the layer below, sum_r = that layer's activation output,
 * and the two sums are what its BN backward starts from -- no separate pass over y.  cout % 8 == 0. */
int xv_tdnn_layer_bf16x3_sums(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                              const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K,
                              int dilation, int cout, const uint8_t *row_valid, float *y, int ldy, const float *sum_r,
                              int ld_sum_r, void *workspace, void *stream);
/* xv_tdnn_layer_bf16x3 (fp32 rows out, optional y_preact) whose epilogue also leaves, per 128-row tile, [sum_rows y | sum_rows y^2]
 * in double in `workspace` (xv_col_sums_workspace_bytes(R, cout) bytes): the batch moments tf.layers.batch_normalization(training=
 * True) takes of the layer's activation output (local/tf/models.py:66-68) without a pass over it -- xv_bn_moments_fold_f32 turns
 * them into mean / var / the folded affine.  cout % 8 == 0. */
int xv_tdnn_layer_bf16x3_moments(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                                 const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K,
                                 int dilation, int cout, const uint8_t *row_valid, float *y, int ldy, float *y_preact, int ldpre,
                                 void *workspace, void *stream);
/* Last frame-level layer fused with the first half of statistics pooling (models.py:66-76 / 482-486 in one pass):
 * the same GEMM as xv_tdnn_layer_bf16x3, but instead of storing y[R, Cout] the epilogue reduces every block of 8
 * consecutive rows (global rows 8i..8i+7, valid rows only) to per-channel (mean, M2 = sum (v-mean)^2) and writes
 *     block_stats[ceil(R/8)][2][Cout]  fp32   (xv_block_stats_bytes(R, Cout) bytes, 16-byte aligned)
 * -- a quarter of the bytes of y, and y is never re-read.  Every chunk must START ON A ROW THAT IS A MULTIPLE OF 8 (gap
 * rows pad up to it), so that a block never mixes two chunks and the result does not depend on batch composition.
 * xv_stats_pool_blocks_f32 merges the blocks of each chunk in order (fp64) into out[B, 2*Cout] = [mean | sqrt(var+eps)],
 * the same quantity as xv_stats_pool_f32; a chunk whose row_start is not a multiple of 8 yields NaN. */
size_t xv_block_stats_bytes(int64_t R, int cout);
int xv_tdnn_layer_pool_bf16x3(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                              const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K,
                              int dilation, int cout, const uint8_t *row_valid, float *block_stats, void *stream);
int xv_stats_pool_blocks_f32(const float *block_stats, int c, const int32_t *row_start, const int32_t *row_len, int nchunks,
                             float eps, float *out, void *stream);
/* The same fused last layer in the exact-fp32 arithmetic (xv_tdnn_layer_f32's GEMM with the block-statistics epilogue: the
 * 4 * Cout bytes per frame of y are neither written nor re-read).  x / wp as for xv_tdnn_layer_f32; needs Cout % 4 == 0 and
 * 16-byte aligned bias / bn_scale / bn_shift / per-channel act_alpha. */
int xv_tdnn_layer_pool_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                           const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                           const uint8_t *row_valid, float *block_stats, void *stream);

/* The FIRST frame-level layer (frame_level_info_layer-0: conv1d over the MFCC rows, models.py:54-67) as a kernel built around
 * its output stream: the im2col operand of a wave's 16 frames is formed in registers straight from the fp32 feature rows, the
 * weights of 128 output channels stay in LDS for a strip of 512 frames, and the epilogue writes XV_FMT_SPLIT rows directly from
 * the accumulators.  Same contraction and epilogue as xv_tdnn_layer_bf16x3 with x in XV_FMT_F32 and y in XV_FMT_SPLIT.
 * x[R, ldx]: rows of ceil8(Cin) floats (columns >= Cin zero), ldx % 8 == 0, 32-byte aligned; wt = xv_pack_first_bf16x3(w[K,Cin,Cout]).
 * Supported: K odd, K*ceil8(Cin) <= 128, (K-1)*dilation <= 8, Cout % 32 == 0, Cout <= 512 (xv_packed_first_bf16x3_bytes
 * returns 0 otherwise; the general kernel takes those shapes). */
size_t xv_packed_first_bf16x3_bytes(int K, int cin, int cout);
int xv_pack_first_bf16x3(const float *w, int K, int cin, int cout, void *wt, void *stream);
int xv_tdnn_first_bf16x3(const float *x, int64_t R, int cin, int ldx, const void *wt, const float *bias, const float *bn_scale,
                         const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                         const uint8_t *row_valid, void *y, void *stream);

/* The last TWO frame-level layers and the first half of statistics pooling in one launch, for topologies whose last two
 * layers have no temporal context (kernel size 1: models.py:28 [5,5,7,1,1], :545 [5,3,3,1,1]):
 *     h = bn1(act(x . w1 + b1))          frame_level_info_layer-(n-2)   models.py:54-67
 *     y = bn2(act(h . w2 + b2))          frame_level_info_layer-(n-1)
 *     block_stats = per-8-row (mean, M2) of y, as xv_tdnn_layer_pool_bf16x3 writes them (merge: xv_stats_pool_blocks_f32)
 * h never exists in memory: a wave keeps its 16 frames x Cmid of h in registers, produced by the first GEMM directly in
 * the operand layout of the second (v_mfma_f32_16x16x32_bf16, bf16x3 arithmetic).  x: XV_FMT_SPLIT rows (same padding
 * contract as xv_tdnn_layer_bf16x3).  wt = xv_pack_pair_bf16x3(w1[Cin,Cmid], w2[Cmid,Cout]) (TF's [in, out] order;
 * xv_packed_pair_bf16x3_bytes bytes, 0 = unsupported shape).  Supported: Cmid == 512, Cin % 32 == 0, Cout % 64 == 0,
 * Cout <= 2048 (else XV_ERR_UNSUPPORTED: run the two layers separately).  Chunks must start on rows that are multiples of
 * 8, as for xv_tdnn_layer_pool_bf16x3.  act_alpha1 / act_alpha2: [1] for LRELU, [Cmid] / [Cout] for PRELU. */
size_t xv_packed_pair_bf16x3_bytes(int cin, int cmid, int cout);
int xv_pack_pair_bf16x3(const float *w1, const float *w2, int cin, int cmid, int cout, void *wt, void *stream);
int xv_tdnn_pair_pool_bf16x3(const void *x, int64_t R, int cin, int cmid, int cout, const void *wt, const float *bias1,
                             const float *bn_scale1, const float *bn_shift1, const float *act_alpha1, const float *bias2,
                             const float *bn_scale2, const float *bn_shift2, const float *act_alpha2, int act_kind,
                             const uint8_t *row_valid, float *block_stats, void *stream);

/* ---- f16bf8 split-precision twins of the hidden frame-level layers --------------------------------------------------
 *
 * The same layer (models.py:54-76) with every product formed as
 *     x*w = xh*wh + 2^-11 (xl8*wh8 + xh8*wl8)      xh = fp16(x), xl8 = bf8(2^11 (x - xh)), xh8 = bf8(x)   (bf8 = e5m2)
 * i.e. one v_mfma_f32_32x32x16_f16 and one block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 holds both cross terms
 * of a 32-channel slab; the 2^-11 is its E8M0 scale operand) per three bf16 MFMAs of the bf16x3 arithmetic; fp32
 * accumulation, fp32 epilogue.  Accuracy on the full network: 1.1e-5 relative L2 against fp64 (bf16x3: 5e-6).
 *   XV_FMT_SPLIT8  the XV_FMT_SPLIT geometry (128 bytes per row and 32-channel slab, same padding contract, same
 *                  xv_split_row_bytes, physical slot = logical slot ^ ((r>>1)&7)) with logical slot g (0..3) = fp16 hi of
 *                  channels 8g..8g+7 and slot 4+g = [8 x xl8 | 8 x xh8] of the same channels.
 * Range: values are clamped to +-57344 (largest finite e5m2; fp16 ends at 65504) when a layer WRITES this format; every
 * writer takes `status` (device int32, may be NULL) and ORs bit 0 into it when it had to clamp -- the caller then repeats
 * the batch in the bf16x3 arithmetic (fp32 range).  Activations of a BN-normalised network are O(1..100).
 * Weights: xv_pack_weights_f16bf8 -> xv_packed_weights_f16bf8_bytes(K,Cin,Cout) bytes of 16 KB tiles in the order of
 *   xv_pack_weights_bf16x3, each [fp16 plane 128x64 B][8-bit plane 128x64 B, slot g = 8 x wh8 | 8 x wl8]; |w| > 57344 clamps.
 * xv_tdnn_layer_f16bf8: x in XV_FMT_SPLIT8; y in XV_FMT_F32 (row stride ldy), XV_FMT_SPLIT (to feed a bf16x3 consumer such
 *   as xv_tdnn_pair_pool_bf16x3) or XV_FMT_SPLIT8; K in {1,3,5,7}, 2 <= (K-1)*dilation <= 8 for K > 1.
 * xv_tdnn_layer_pool_f16bf8: the xv_tdnn_layer_pool_bf16x3 contract (block statistics instead of y).
 * xv_tdnn_first_f16bf8: xv_tdnn_first_bf16x3 (bf16x3 arithmetic on the fp32 feature rows, same packed weights) writing
 *   XV_FMT_SPLIT8 rows.  xv_split8_encode_f32 / xv_split8_decode_f32: tooling and tests (decode returns hi + xl8 / 2^11). */
#define XV_FMT_SPLIT8 2
size_t xv_packed_weights_f16bf8_bytes(int K, int cin, int cout);
int xv_pack_weights_f16bf8(const float *w, int K, int cin, int cout, void *wt, void *stream);
int xv_split8_encode_f32(const float *x, int64_t R, int c, int ldx, void *xs, int32_t *status, void *stream);
int xv_split8_decode_f32(const void *xs, int64_t R, int c, float *x, int ldx, void *stream);
int xv_tdnn_layer_f16bf8(const void *x, int64_t R, int cin, const void *wt, const float *bias, const float *bn_scale,
                         const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                         const uint8_t *row_valid, void *y, int y_format, int ldy, int32_t *status, void *stream);
int xv_tdnn_layer_pool_f16bf8(const void *x, int64_t R, int cin, const void *wt, const float *bias, const float *bn_scale,
                              const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                              const uint8_t *row_valid, float *block_stats, void *stream);
int xv_tdnn_first_f16bf8(const float *x, int64_t R, int cin, int ldx, const void *wt, const float *bias, const float *bn_scale,
                         const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                         const uint8_t *row_valid, void *y, int32_t *status, void *stream);

/* xv_tdnn_pair_pool_bf16x3 in the f16bf8 arithmetic: x in XV_FMT_SPLIT8, wt = xv_pack_pair_f16bf8(w1, w2)
 * (xv_packed_pair_f16bf8_bytes bytes, 0 = unsupported: Cmid == 512, Cin % 32 == 0, Cout % 64 == 0, Cout <= 4096), same
 * block_stats contract.  A pair of waves shares 32 frames: each wave keeps HALF of the intermediate's channels in registers
 * (so it streams half of both layers' weights through its fragment reads) and the two partial sums of the second GEMM meet in
 * LDS.  `status` (may be NULL): bit 0 is ORed in when the intermediate activation had to be clamped to +-57344. */
size_t xv_packed_pair_f16bf8_bytes(int cin, int cmid, int cout);
int xv_pack_pair_f16bf8(const float *w1, const float *w2, int cin, int cmid, int cout, void *wt, void *stream);
int xv_tdnn_pair_pool_f16bf8(const void *x, int64_t R, int cin, int cmid, int cout, const void *wt, const float *bias1,
                             const float *bn_scale1, const float *bn_shift1, const float *act_alpha1, const float *bias2,
                             const float *bn_scale2, const float *bn_shift2, const float *act_alpha2, int act_kind,
                             const uint8_t *row_valid, float *block_stats, int32_t *status, void *stream);

/* xv_fc_f32 twin: fp32 rows in, fp32 rows out; wt = xv_pack_weights_bf16x3(w, 1, In, Out). */
int xv_fc_bf16x3(const float *x, int nrows, int in_dim, const void *wt, const float *bias, const float *bn_scale,
                 const float *bn_shift, int act_kind, const float *act_alpha, int out_dim, float *y, float *y_preact, void *stream);

/* Statistics pooling.  Replaces tf.nn.moments(h, 1) + tf.sqrt(var + 1e-5) + tf.concat
 * (local/tf/models.py:16,75-76):  out[b] = [ mean_t h[t,:]  ||  sqrt(mean_t (h-mean)^2 + eps) ]
 * over the rows of chunk b.  h[R,C] row stride ldh, C % 4 == 0; out[B, 2C].
 * Chunks longer than `split_rows` rows are reduced by several workgroups; `workspace` must then hold
 * xv_stats_pool_workspace_bytes(C, B, max_len, split_rows) bytes (may be NULL when that is 0). */
size_t xv_stats_pool_workspace_bytes(int c, int nchunks, int max_len, int split_rows);
int xv_stats_pool_f32(const float *h, int64_t ldh, int c, const int32_t *row_start, const int32_t *row_len,
                      int nchunks, int max_len, int split_rows, float eps, float *out, void *workspace,
                      void *stream);

/* Segment-level affine.  Replaces tf.nn.xw_plus_b (local/tf/models.py:86) and, for the path to
 * embed_layer-1, the relu + batch_norm between the two (local/tf/models.py:88-89):
 *     z = x[B,In] . w + bias ;  y_preact <- z (the x-vector when this is embed_layer-0) ;
 *     y <- act(z)*bn_scale + bn_shift   (skipped when y == NULL)
 *   wp = xv_pack_weights_f32(w[In,Out]) i.e. [Out][In]. */
int xv_fc_f32(const float *x, int nrows, int in_dim, const float *wp, const float *bias, const float *bn_scale,
              const float *bn_shift, int act_kind, const float *act_alpha, int out_dim, float *y,
              float *y_preact, void *stream);

/* xv_fc_f32 for a SKINNY problem (a training minibatch's segment level: 64 rows x 3072 -> 512): with <= 128 rows there are only a few
 * output tiles, and each would walk all In / 32 slabs one after the other.  The slabs are dealt to groups of workgroups that write
 * raw partial sums to `workspace` (xv_fc_splitk_workspace_bytes bytes; 0 = the shape is not skinny and the call IS xv_fc_f32), a
 * second kernel adds the groups in order (deterministic) and applies bias / activation / BN.  Same arguments as xv_fc_f32. */
size_t xv_fc_splitk_workspace_bytes(int nrows, int in_dim, int out_dim);
int xv_fc_splitk_f32(const float *x, int nrows, int in_dim, const float *wp, const float *bias, const float *bn_scale,
                     const float *bn_shift, int act_kind, const float *act_alpha, int out_dim, float *y, float *y_preact, void *workspace,
                     void *stream);

/* Length-weighted average of chunk embeddings.  Replaces the NumPy lines local/tf/models.py:398,
 * 418-421 with the same float32 operation order (product, sum in chunk order, one division):
 *   out[u] = ( sum_{i in [seg_start[u], seg_start[u+1])} float(chunk_len[i]) * e[i] ) / float(sum len). */
int xv_chunk_average_f32(const float *e, const int32_t *seg_start, const int32_t *chunk_len, int nutts, int dim,
                         float *out, void *stream);

/* ---- training step (reference Model.train_one_iteration, local/tf/models.py:216-305) ----------
 * The forward GEMMs and the input-gradient GEMMs are xv_tdnn_layer_f32 / xv_fc_f32 (dgrad = the same kernel with the
 * taps flipped and Cin/Cout swapped in the weights); the entry points below add the rest.  fp32 storage, reductions
 * accumulate in fp64.  Workspaces are caller-provided device buffers. */

/* Per-chunk (mean, BIASED variance) over the rows of each chunk: out[B, 2C] = [mean || var].  Same kernel and arguments
 * as xv_stats_pool_f32 without the sqrt(var+eps).  With xv_merge_moments_f32 this is tf.nn.moments over all frames of
 * the minibatch (local/tf/tf_block.py:19). */
int xv_chunk_moments_f32(const float *h, int64_t ldh, int c, const int32_t *row_start, const int32_t *row_len, int nchunks,
                         int max_len, int split_rows, float *out, void *workspace, void *stream);
int xv_merge_moments_f32(const float *chunk_mean_var, const int32_t *row_len, int nchunks, int c, float *mean, float *var,
                         void *stream);
/* y[r,:] = row_valid[r] ? x[r,:]*scale + shift : 0   (batch-norm with batch statistics folded by xv_fold_bn_f32). */
int xv_rows_affine_f32(const float *x, int ldx, int64_t R, int c, const float *scale, const float *shift,
                       const uint8_t *row_valid, float *y, int ldy, void *stream);
/* The same with a second copy of y in the bf16 split activation format (XV_FMT_SPLIT; y_split = row 0 of a buffer with the format's
 * padding rows, c % 32 == 0; NULL = xv_rows_affine_f32): the training step's K = 1 layers then take the DMA-fed GEMM. */
int xv_rows_affine_split_f32(const float *x, int ldx, int64_t R, int c, const float *scale, const float *shift,
                             const uint8_t *row_valid, float *y, int ldy, void *y_split, void *stream);
/* Weight gradient of a TDNN/FC layer: dw[k,ci,co] = sum_r x[r + (k-(K-1)/2)*dilation, ci] * dz[r, co]  (TF layout
 * [K,Cin,Cout]; rows outside [0,R) read as zero; gap rows of x and dz are zero by contract). */
size_t xv_wgrad_workspace_bytes(int64_t R, int cin, int cout, int K);
int xv_wgrad_f32(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation,
                 float *dw, void *workspace, void *stream);
/* The same gradient in the bf16x3 arithmetic of xv_tdnn_layer_bf16x3 (operands split hi + lo while they are staged, three bf16
 * MFMAs per product, fp32 accumulate; ~5e-6 relative): what `--train-precision bf16x3` runs.  Same arguments, same workspace,
 * same deterministic split merge; falls back to xv_wgrad_f32 when R * ld * 4 >= 2^31 (32-bit buffer offsets). */
int xv_wgrad_bf16x3(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation,
                    float *dw, void *workspace, void *stream);
/* xv_wgrad_bf16x3 that also leaves db[co] = sum_r dz[r, co], the bias gradient of the same layer (models.py:61 under minimize()): the
 * workgroups of tap 0 / input tile 0 sum the dz rows they stream anyway (fp32 inside a 16-row step, double across steps and row
 * splits, merged in split order: deterministic) -- no separate pass over dz (xv_col_sums_f32).  workspace: xv_wgrad_bias_workspace_bytes
 * (always required).  No fp32 fallback: XV_ERR_UNSUPPORTED when R * ld * 4 >= 2^31. */
size_t xv_wgrad_bias_workspace_bytes(int64_t R, int cin, int cout, int K);
int xv_wgrad_bias_bf16x3(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation,
                         float *dw, float *db, void *workspace, void *stream);
/* sum_a[c] = sum_r a[r,c];  sum_ab[c] = sum_r a[r,c]*b[r,c]  (b, sum_ab may be NULL). */
size_t xv_col_sums_workspace_bytes(int64_t R, int c);
int xv_col_sums_f32(const float *a, int lda, const float *b, int ldb, int64_t R, int c, float *sum_a, float *sum_ab,
                    void *workspace, void *stream);
/* Batch-norm backward through the activation (closed form): given dh = dL/d(BN output), r = activation output, the
 * batch statistics and sum_dh = sum_r dh, sum_dh_r = sum_r dh*r, writes dgamma, dbeta and dz = dL/d(pre-activation)
 * (gap rows zero).  coef_ws: 3*c floats.  act_kind: NONE / RELU / LRELU(act_alpha). */
int xv_bn_act_backward_f32(const float *dh, const float *r, int ld, int64_t R, int c, const float *sum_dh,
                           const float *sum_dh_r, const float *mean, const float *var, const float *gamma, float eps,
                           float n_frames, int act_kind, float act_alpha, const uint8_t *row_valid, float *dgamma,
                           float *dbeta, float *coef_ws, float *dz, void *stream);
/* The same with a second copy of dz in the bf16 split activation format (dz_split as y_split above; NULL = the plain form). */
int xv_bn_act_backward_split_f32(const float *dh, const float *r, int ld, int64_t R, int c, const float *sum_dh,
                                 const float *sum_dh_r, const float *mean, const float *var, const float *gamma, float eps,
                                 float n_frames, int act_kind, float act_alpha, const uint8_t *row_valid, float *dgamma,
                                 float *dbeta, float *coef_ws, float *dz, void *dz_split, void *stream);
/* The same from the PARTIAL column sums a producer left in `sums_workspace` (xv_col_sums_workspace_bytes(R, c) bytes: per 128 rows
 * [sum dh | sum dh*r] in double -- written by xv_tdnn_layer_bf16x3_sums, the input-gradient GEMM that produced dh): merge and
 * coefficients in one launch, no pass over dh for the sums.  tf.gradients of tf.layers.batch_normalization(training=True),
 * local/tf/models.py:66-68,109-113. */
int xv_bn_act_backward_parts_f32(const float *dh, const float *r, int ld, int64_t R, int c, const void *sums_workspace,
                                 const float *mean, const float *var, const float *gamma, float eps, float n_frames,
                                 int act_kind, float act_alpha, const uint8_t *row_valid, float *dgamma, float *dbeta,
                                 float *coef_ws, float *dz, void *dz_split, void *stream);
/* Batch moments + BN fold from the partial sums of xv_tdnn_layer_bf16x3_moments, one launch: mean = S1 / n_frames,
 * var = S2 / n_frames - mean^2 (biased: tf.nn.moments), scale = gamma / sqrt(var + eps), shift = beta - mean * scale (the
 * operations of xv_fold_bn_f32).  Replaces xv_chunk_moments_f32 + xv_merge_moments_f32 + xv_fold_bn_f32 of a training forward. */
int xv_bn_moments_fold_f32(const void *sums_workspace, int64_t R, int c, float n_frames, const float *gamma, const float *beta,
                           float eps, float *mean, float *var, float *scale, float *shift, void *stream);
/* Merge of such partial sums alone: sum_a[c], sum_ab[c] (sum_ab may be NULL) -- what xv_col_sums_f32 returns for the same rows. */
int xv_col_sums_merge_f32(const void *sums_workspace, int64_t R, int c, float *sum_a, float *sum_ab, void *stream);
/* Backward of [statistics pooling -> BN -> activation] of the LAST frame-level layer in two launches: the gradient that reaches
 * h = BN(r) comes from the pooling alone (local/tf/models.py:75-76), so the BN backward's column sums follow from per-chunk numbers
 * (pooled = [mu | sig], dpooled, chunk_moments = xv_chunk_moments_f32 of r: [mean | biased var] per chunk) and dh is formed on the
 * fly, never stored.  Writes dgamma, dbeta, dz (gap rows and rows outside every chunk zero) and optionally dz in the split format. */
int xv_pool_bn_act_backward_f32(const float *h, const float *r, int ld, int c, const int32_t *row_start, const int32_t *row_len,
                                int nchunks, int64_t R, const float *pooled, const float *dpooled, const float *chunk_moments,
                                const float *mean, const float *var, const float *gamma, float eps, float n_frames,
                                int act_kind, float act_alpha, float *dgamma, float *dbeta, float *coef_ws, float *dz,
                                void *dz_split, void *stream);
/* tf.layers.batch_normalization(training=True) of a SMALL matrix (1 .. 1024 rows, every row a sample: the segment-level layers of
 * a training step, local/tf/models.py:88-91) in one launch each way.  Forward: batch mean / biased variance into mean, var, and
 * y = x * scale + shift with the fold of xv_fold_bn_f32 (what xv_chunk_moments_f32 + xv_merge_moments_f32 + xv_fold_bn_f32 +
 * xv_rows_affine_f32 do in four).  Backward: dgamma, dbeta and dz through the activation (NONE / RELU / LRELU), as xv_col_sums_f32
 * + xv_bn_act_backward_f32. */
int xv_bn_small_forward_f32(const float *x, int ldx, int nrows, int c, const float *gamma, const float *beta, float eps, float *mean,
                            float *var, float *y, int ldy, void *stream);
int xv_bn_small_backward_f32(const float *dh, const float *r, int ld, int nrows, int c, const float *mean, const float *var,
                             const float *gamma, float eps, int act_kind, float act_alpha, float *dgamma, float *dbeta, float *dz,
                             void *stream);
/* Gradient of statistics pooling: dh[t,c] = dmu[c]/T + dsig[c]*(h[t,c]-mu[c])/(T*sig[c]); dh gap rows are zeroed. */
int xv_pool_backward_f32(const float *h, int ldh, int c, const int32_t *row_start, const int32_t *row_len, int nchunks,
                         int64_t R, const float *pooled, const float *dpooled, float *dh, void *stream);
/* tf.nn.softmax_cross_entropy_with_logits + reduce_mean + accuracy (local/tf/models.py:106-117):
 * loss_acc[0] = mean loss, loss_acc[1] = accuracy; dlogits (may be NULL) = (softmax - onehot)/nrows.  row_ws: 2*nrows. */
int xv_softmax_ce_f32(const float *logits, const int32_t *labels, int nrows, int nclasses, float *loss_acc, float *row_ws,
                      float *dlogits, void *stream);
/* tf.train.AdamOptimizer dense update with lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) computed by the caller. */
int xv_adam_f32(float *param, const float *grad, float *m, float *v, int64_t n, float lr_t, float beta1, float beta2, float eps,
                void *stream);
/* y += a*x and out[0] = sum x^2: the L2 penalty of the ModelL2Loss* classes, beta*(0.1|1)*tf.nn.l2_loss(w)
 * (local/tf/models.py:811-842): gradient g += beta*coef*w, loss += beta*coef*sumsq/2.  xv_sumsq_f32 reduces in two
 * stages (fp64 partials in `workspace`, xv_sumsq_workspace_bytes(n) bytes, 8-byte aligned, added in a fixed order). */
int xv_axpy_f32(float *y, const float *x, float a, int64_t n, void *stream);
size_t xv_sumsq_workspace_bytes(int64_t n);
int xv_sumsq_f32(const float *x, int64_t n, float *out, void *workspace, void *stream);
/* tf.nn.dropout(x, keep_prob) in place on x[R, C] (models.py:70-72, 92-94): element (r, c) is kept (and scaled by
 * 1/keep_prob) iff the top 32 bits of splitmix64(seed ^ 0x9E3779B97F4A7C15*(r*C + c + 1)) < keep_prob*2^32, else zeroed.
 * Stateless: the same call on the gradient buffer is the backward pass.  keep_prob == 1 is a no-op. */
int xv_dropout_f32(float *x, int ldx, int64_t R, int c, uint64_t seed, float keep_prob, void *stream);
/* Index arrays of a training minibatch in the packed-rows layout (B chunks of T frames, `gap` zero rows in front of, between and
 * behind them; rows >= gap + B (T + gap)): row_start[B], row_len[B], row_valid[rows] -- generated on the device (a host copy of
 * them is a host synchronisation, and a run meets a new length in most of its first few hundred steps). */
int xv_minibatch_layout(int B, int T, int gap, int64_t rows, int32_t *row_start, int32_t *row_len, uint8_t *row_valid, void *stream);
/* A training minibatch src[B, T, F] (float16 when src_is_f16, else float32; DEVICE pointer like everything else) -> the packed
 * rows-with-gaps matrix dst[rows, in_dim] of the header comment: chunk b at rows gap + b*(T+gap), all other rows and the columns
 * F..in_dim-1 zero; rows >= gap + B*(T+gap).  Replaces the feed_dict hand-over of models.py:255-262 (the egs hold float16). */
int xv_pack_minibatch_f32(const void *src, int src_is_f16, int B, int T, int F, int gap, int in_dim, float *dst, int64_t rows,
                          void *stream);

/* PReLU backward (tf_block.py:38-47): on entry dr = dL/d(activation output), z = pre-activation; on exit
 * dr = dL/dz = dr * (z > 0 ? 1 : alpha[c]) and z = dr_in * min(z, 0), whose column sums (xv_col_sums_f32) are dL/dalpha. */
int xv_prelu_backward_f32(float *dr, float *z, int ld, int64_t R, int c, const float *alpha, void *stream);

/* moving = moving*decay + batch*(1-decay)   (local/tf/tf_block.py:20-21). */
int xv_ema_f32(float *moving, const float *batch, int n, float decay, void *stream);

/* ---- additive-margin softmax head (build-defined: BASELINE configs[4] asks for it, the reference has no margin head) ----
 * logits[b, j] = scale * (cos(x_b, w_j) - margin * [j == label_b]),  cos = <x/||x||, w_j/||w_j||>   (Wang et al. 2018).
 * xv_l2_normalize_rows_f32: y = x / max(||x||, 1e-12) per row, norm[r] = ||x_r||;  _backward: dx = (dy - y<y,dy>) / norm;
 * xv_am_margin_f32 applies margin and scale in place to the cosine matrix (which xv_fc_* computes from the normalised
 * operands); the loss is xv_softmax_ce_f32 on the result. */
int xv_l2_normalize_rows_f32(const float *x, int ldx, int nrows, int c, float *y, int ldy, float *norm, void *stream);
int xv_l2_normalize_backward_f32(const float *dy, const float *y, const float *norm, int nrows, int c, float *dx, void *stream);
int xv_am_margin_f32(float *cosines, const int32_t *labels, int nrows, int nclasses, float scale, float margin, void *stream);

/* ---- self-attentive statistics pooling (ModelL2LossWithoutDropoutLReluAttention, local/tf/models.py:1036-1052) -------
 * The last frame-level layer has 2*C channels, h = [h1 | h2].  With u = h1.W + b (one more K=1 xv_tdnn_layer_* call):
 *   xv_attention_scores_f32   scores[r] = sum_c v[c]*tanh(u[r,c])            (models.py:1045-1046; fp64 row sums);
 *                             nonlin (optional, [R, ldn]) receives tanh(u) for the training backward
 *   xv_attention_softmax_f32  att[rows of chunk b] = softmax(scores[rows of chunk b])   (models.py:1046)
 *   xv_attention_pool_f32     out[b] = [ m | sqrt(max(q,0) + eps) ],  m = sum_t att[t] h2[t,:],  q = sum_t att[t] h2[t,:]^2 - m^2
 *                             (models.py:1048-1050; products and sums in fp64).  Chunks longer than split_rows rows are
 *                             reduced by several workgroups through `workspace`
 *                             (xv_attention_pool_workspace_bytes bytes, 8-byte aligned; may be NULL when that is 0).
 * Gap rows of scores / att are never read or written. */
int xv_attention_scores_f32(const float *u, int64_t ldu, int64_t R, int c, const float *v, float *scores, float *nonlin,
                            int64_t ldn, void *stream);
int xv_attention_softmax_f32(const float *scores, const int32_t *row_start, const int32_t *row_len, int nchunks, float *att,
                             void *stream);
size_t xv_attention_pool_workspace_bytes(int c, int nchunks, int max_len, int split_rows);
int xv_attention_pool_f32(const float *h, int64_t ldh, int c, const float *att, const int32_t *row_start, const int32_t *row_len,
                          int nchunks, int max_len, int split_rows, float eps, float *out, void *workspace, void *stream);
/* Backward of the three steps above (training step of the attention class):
 *   xv_attention_pool_backward_f32     dh[t,c] = att[t] (g1[c] + 2 h[t,c] g2[c]),  datt[t] = sum_c h[t,c] g1[c] + h[t,c]^2 g2[c]
 *                                      with g2 = dsd/(2 sd), g1 = dm - 2 m g2, [m | sd] = pooled, [dm | dsd] = dpooled
 *   xv_attention_softmax_backward_f32  dscores[t] = att[t] (datt[t] - sum_tau att[tau] datt[tau])   per chunk
 *   xv_attention_scores_backward_f32   du[r,c] = dscores[r] v[c] (1 - n[r,c]^2);  nonlin (= n = tanh(u)) is OVERWRITTEN with
 *                                      dscores[r] n[r,c], whose column sums are dv (xv_col_sums_f32); rows outside every
 *                                      chunk must carry dscores = 0. */
int xv_attention_pool_backward_f32(const float *h, int64_t ldh, int c, const float *att, const int32_t *row_start,
                                   const int32_t *row_len, int nchunks, int max_len, const float *pooled, const float *dpooled,
                                   float *dh, int64_t lddh, float *datt, void *stream);
int xv_attention_softmax_backward_f32(const float *att, const float *datt, const int32_t *row_start, const int32_t *row_len,
                                      int nchunks, float *dscores, void *stream);
int xv_attention_scores_backward_f32(float *nonlin, int64_t ldn, const float *dscores, const float *v, int64_t R, int c, float *du,
                                     int64_t lddu, void *stream);

/* ---- feature front-end -------------------------------------------------------------------------------
 * Sliding-window cepstral mean normalisation + VAD frame selection, i.e. what
 *   apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 ... | select-voiced-frames ...
 * (local/tf/extract_xvectors.sh:68) do in front of extract_embedding.py, as one scatter kernel.
 *   x[sum T, F]  raw features of n_utts utterances back to back (utt_start[u], utt_len[u] in rows; max_len = max utt_len)
 *   dst_row[t]   per INPUT row: row of y that receives the normalised frame, or -1 to drop it (unvoiced / not in a chunk)
 *   y[., ldy]    destination (e.g. the packed batch buffer); only columns [0, F) of the addressed rows are written
 * Window rule and arithmetic (double-precision window sums) follow Kaldi's SlidingWindowCmn; min_window only matters
 * when center == 0 (Kaldi default 100). */
int xv_cmn_sliding_scatter_f32(const float *x, int ldx, int feat_dim, const int32_t *utt_start, const int32_t *utt_len,
                               int n_utts, int max_len, int cmn_window, int center, int min_window, const int32_t *dst_row,
                               float *y, int ldy, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XVECTOR_HIP_H_ */
