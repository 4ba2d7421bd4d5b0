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
 *  - arithmetic: the *_f32 GEMM entry points are exact IEEE fp32 (f32-input MFMA: exact fp32 products, fp32
 *    accumulation).  The *_bf16x3 twins compute the same fp32-in / fp32-out contraction with every operand
 *    split x = hi + lo into two bf16 and the product accumulated in fp32 as hi*hi + hi*lo + lo*hi on the
 *    bf16 matrix cores (16x the f32-MFMA rate; the dropped lo*lo term is ~2^-16 relative): measured
 *    ~3e-6 relative L2 on the x-vector against the fp64 oracle, inside the 1e-4 parity bar.  Pooling,
 *    epilogues and the chunk average are fp32 in both.
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

/* Library / ABI version (increments on any signature change). */
int xv_version(void);
/* Thread-local description of the last non-zero return. */
const char *xv_last_error(void);

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

/* bf16x3 split-precision twins of xv_pack_weights_f32 / xv_tdnn_layer_f32 / xv_fc_f32 (same semantics, same
 * operands except the weights, which are pre-split into two bf16 planes wp_hi/wp_lo[Cout][K*Cin] by
 * xv_pack_weights_bf16x3; activations stay fp32 in HBM and are split while staged).  Cin % 8 == 0. */
int xv_pack_weights_bf16x3(const float *w, int kred, int cout, uint16_t *wp_hi, uint16_t *wp_lo, void *stream);
int xv_tdnn_layer_bf16x3(const float *x, int64_t R, int cin, int ldx, const uint16_t *wp_hi, const uint16_t *wp_lo,
                         const float *bias, const float *bn_scale, const float *bn_shift, int act_kind,
                         const float *act_alpha, int K, int dilation, int cout, const uint8_t *row_valid, float *y, int ldy,
                         float *y_preact, void *stream);
int xv_fc_bf16x3(const float *x, int nrows, int in_dim, const uint16_t *wp_hi, const uint16_t *wp_lo, const float *bias,
                 const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int out_dim, float *y,
                 float *y_preact, void *stream);

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

/* Length-weighted average of chunk embeddings.  Replaces the NumPy lines local/tf/models.py:398,
 * 418-421 with the same float32 operation order (product, sum in chunk order, one division):
 *   out[u] = ( sum_{i in [seg_start[u], seg_start[u+1])} float(chunk_len[i]) * e[i] ) / float(sum len). */
int xv_chunk_average_f32(const float *e, const int32_t *seg_start, const int32_t *chunk_len, int nutts, int dim,
                         float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XVECTOR_HIP_H_ */
