/*
 * xv_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Type-generic body of the CPU oracle.  Included twice by xv_oracle.c with
 *   XV_T = float  / XV_SUFFIX(x) = x##_f32
 *   XV_T = double / XV_SUFFIX(x) = x##_f64
 * Every function restates one step of the reference's x-vector forward graph
 * (reference = BUTSpeechFIT/x-vector-kaldi-tf, paths relative to its root).
 *
 * The arithmetic of the reference lives in TensorFlow 1.x (un-vendored, version
 * unpinned, absent from this image), so these functions follow the TF op
 * DEFINITIONS at the reference's call sites; how they are WIRED is checked
 * against the reference's graph code executed (header of xv_oracle.c):
 *   conv1d / convolution  local/tf/models.py:60, :476, :579-580
 *   bias_add              local/tf/models.py:61
 *   relu / leaky / prelu  local/tf/models.py:64, :912 ; local/tf/tf_block.py:38-47
 *   batch_norm eval       local/tf/tf_block.py:9-16,25-28 (epsilon default 1e-3)
 *   moments + sqrt+concat local/tf/models.py:16,75-76    (VAR2STD_EPSILON 1e-5)
 *   xw_plus_b             local/tf/models.py:86
 */

/* One frame-level layer for ONE utterance (batch 1, as the reference runs it,
 * local/tf/models.py:410-414): SAME zero padding at this utterance's own edges,
 * cross-correlation (no kernel flip), then +b -> activation -> BN(eval).
 *   x[T,Cin] row-major, w[K,Cin,Cout] (TF kernel layout, models.py:56-57),
 *   y[T,Cout].
 * SAME padding: pad_total=(K-1)*dil, left=pad_total/2 (all K odd -> symmetric).
 * BN eval (tf.nn.batch_normalization): y = r*s + (beta - mean*s), s = gamma*rsqrt(var+eps).
 * act_kind: 0 none, 1 relu, 2 leaky-relu(alpha[0]), 3 prelu(alpha[c]).
 * bn==0 skips the BN affine.
 */
void XV_SUFFIX(xv_oracle_tdnn_layer)(const XV_T *x, int T, int Cin,
                                     const XV_T *w, const XV_T *b,
                                     const XV_T *gamma, const XV_T *beta,
                                     const XV_T *mean, const XV_T *var,
                                     double bn_eps, int bn,
                                     int act_kind, const XV_T *alpha,
                                     int K, int dil, int Cout, XV_T *y)
{
    const int left = ((K - 1) * dil) / 2;
    XV_T *scale = (XV_T *)malloc(sizeof(XV_T) * (size_t)Cout);
    XV_T *shift = (XV_T *)malloc(sizeof(XV_T) * (size_t)Cout);
    for (int o = 0; o < Cout; ++o) {
        if (bn) {
            XV_T s = gamma[o] / (XV_T)sqrt((double)var[o] + bn_eps);
            scale[o] = s;
            shift[o] = beta[o] - mean[o] * s;
        } else {
            scale[o] = (XV_T)1;
            shift[o] = (XV_T)0;
        }
    }
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; ++t) {
        XV_T *yt = y + (size_t)t * Cout;
        for (int o = 0; o < Cout; ++o) yt[o] = b[o];
        for (int k = 0; k < K; ++k) {
            const int ts = t - left + k * dil;
            if (ts < 0 || ts >= T) continue;           /* zero padding */
            const XV_T *xs = x + (size_t)ts * Cin;
            const XV_T *wk = w + (size_t)k * Cin * Cout;
            for (int c = 0; c < Cin; ++c) {
                const XV_T xv = xs[c];
                const XV_T *wr = wk + (size_t)c * Cout;
                for (int o = 0; o < Cout; ++o) yt[o] += xv * wr[o];
            }
        }
        for (int o = 0; o < Cout; ++o) {
            XV_T z = yt[o], r;
            switch (act_kind) {
            case 1: r = z > 0 ? z : (XV_T)0; break;
            case 2: r = z > 0 ? z : alpha[0] * z; break;           /* tf.nn.leaky_relu = max(a*z, z) */
            case 3: r = (z > 0 ? z : (XV_T)0) + alpha[o] * (z < 0 ? z : (XV_T)0); break;
            default: r = z; break;
            }
            yt[o] = r * scale[o] + shift[o];
        }
    }
    free(scale);
    free(shift);
}

/* Statistics pooling for ONE utterance: tf.nn.moments(h, axis=time) is the
 * population variance about the mean (two-pass definition); output is
 * [mean || sqrt(var + eps)], local/tf/models.py:75-76. */
void XV_SUFFIX(xv_oracle_stats_pool)(const XV_T *h, int T, int C, double eps, XV_T *out)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        XV_T s = 0;
        for (int t = 0; t < T; ++t) s += h[(size_t)t * C + c];
        const XV_T mu = s / (XV_T)T;
        XV_T v = 0;
        for (int t = 0; t < T; ++t) {
            const XV_T d = h[(size_t)t * C + c] - mu;
            v += d * d;
        }
        v /= (XV_T)T;
        out[c] = mu;
        out[C + c] = (XV_T)sqrt((double)v + eps);
    }
}

/* Segment-level affine tf.nn.xw_plus_b: y[B,Out] = x[B,In] . w[In,Out] + b,
 * local/tf/models.py:86. */
void XV_SUFFIX(xv_oracle_fc)(const XV_T *x, int B, int In, const XV_T *w, const XV_T *b,
                             int Out, XV_T *y)
{
#pragma omp parallel for schedule(static)
    for (int r = 0; r < B; ++r) {
        XV_T *yr = y + (size_t)r * Out;
        for (int o = 0; o < Out; ++o) yr[o] = b[o];
        for (int i = 0; i < In; ++i) {
            const XV_T xv = x[(size_t)r * In + i];
            const XV_T *wr = w + (size_t)i * Out;
            for (int o = 0; o < Out; ++o) yr[o] += xv * wr[o];
        }
    }
}

/* relu|leaky|prelu followed by BN(eval) on a [B,C] matrix: the block between
 * embed_layer-0/scores and embed_layer-1 (local/tf/models.py:88-89). */
void XV_SUFFIX(xv_oracle_act_bn)(const XV_T *x, int B, int C,
                                 const XV_T *gamma, const XV_T *beta,
                                 const XV_T *mean, const XV_T *var, double bn_eps,
                                 int act_kind, const XV_T *alpha, XV_T *y)
{
    for (int r = 0; r < B; ++r)
        for (int o = 0; o < C; ++o) {
            XV_T z = x[(size_t)r * C + o], a;
            switch (act_kind) {
            case 1: a = z > 0 ? z : (XV_T)0; break;
            case 2: a = z > 0 ? z : alpha[0] * z; break;
            case 3: a = (z > 0 ? z : (XV_T)0) + alpha[o] * (z < 0 ? z : (XV_T)0); break;
            default: a = z; break;
            }
            const XV_T s = gamma[o] / (XV_T)sqrt((double)var[o] + bn_eps);
            y[(size_t)r * C + o] = a * s + (beta[o] - mean[o] * s);
        }
}
