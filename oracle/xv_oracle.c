/*
 * xv_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's x-vector extraction hot path
 * (BUTSpeechFIT/x-vector-kaldi-tf: local/tf/models.py:50-94 forward graph,
 * local/tf/tf_block.py:9-28 batch-norm, local/tf/models.py:356-432
 * make_embedding driver).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.  The product
 * path (x-vector-kaldi-tf_amd/) never calls it and fails loudly without its HIP
 * library.
 *
 * PARITY STATUS: pinned in two layers (tests/test_oracle.py, fixtures written by
 * tests/golden/make_golden.py in the build container):
 *   GRAPH WIRING -- reference-executed.  Every build_model of the reference's
 *   local/tf/models.py (+ tf_block.py), its load_model, make_embedding,
 *   train_one_iteration and eval run UNMODIFIED under tests/golden/numpy_tf1.py
 *   (a NumPy float64 evaluator registered as `tensorflow`); what those graphs
 *   return for embedding[0] / [1], the pooled vector and every layer's output is
 *   tests/golden/forward_refgraph.npz, and this oracle agrees with it to < 1e-12
 *   on all 8 classes.  Op order, both epsilons, SAME / dilation arguments, the
 *   variable names, which tensor is the x-vector are therefore the reference's
 *   code's decisions, not a reading of it.
 *   OP NUMERICS -- by cited definition.  TensorFlow 1.x itself (un-vendored,
 *   version-unpinned README.md:28-32, not installable here) never runs: each op
 *   in numpy_tf1.py is the documented TF definition, checked on its own by
 *   tests/test_numpy_tf1.py (index-arithmetic convolution, finite differences).
 *   Control flow / ark framing: the reference's own Python, byte for byte.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define XV_T float
#define XV_SUFFIX(n) n##_f32
#include "xv_oracle_impl.h"
#undef XV_T
#undef XV_SUFFIX

#define XV_T double
#define XV_SUFFIX(n) n##_f64
#include "xv_oracle_impl.h"
#undef XV_T
#undef XV_SUFFIX

/* Chunking rule of Model.make_embedding, local/tf/models.py:377-407.
 * Returns the number of chunks that are RUN (those >= min_chunk), or -1 when the
 * utterance is rejected (T==0 or T<min_chunk: models.py:378-387 -> nothing is
 * written for the key).  starts/lens must hold ceil(T/cs) entries at most. */
int xv_oracle_chunk_plan(int T, int min_chunk, int chunk, int *starts, int *lens, int cap)
{
    if (T == 0 || T < min_chunk) return -1;
    int cs = chunk;
    if (T < chunk) cs = T;                 /* models.py:389-392 */
    else if (chunk == -1) cs = T;          /* models.py:393-394 */
    int num_chunks = (int)ceil((double)T / (double)cs);   /* models.py:396 */
    int n = 0;
    for (int i = 0; i < num_chunks; ++i) {
        int rem = T - i * cs;
        int len = cs < rem ? cs : rem;     /* models.py:405 (NB: tail chunk is NOT shifted back) */
        if (len < min_chunk) continue;     /* models.py:406-407 */
        if (n < cap) { starts[n] = i * cs; lens[n] = len; }
        ++n;
    }
    return n;
}

/* Length-weighted chunk average exactly as NumPy evaluates models.py:398,418-421
 * in float32:  avg = 0; avg += len_i * e_i (float32 product, float32 sum, in
 * chunk order); avg /= sum(len) .  One rounding per op, no FMA contraction. */
void xv_oracle_chunk_average_f32(const float *e, const int *lens, int nchunks, int dim, float *out)
{
    double tot = 0.0;
    for (int d = 0; d < dim; ++d) out[d] = 0.0f;
    for (int i = 0; i < nchunks; ++i) {
        const float wgt = (float)lens[i];
        for (int d = 0; d < dim; ++d) {
            volatile float p = wgt * e[(size_t)i * dim + d];
            volatile float s = out[d] + p;
            out[d] = s;
        }
        tot += (double)lens[i];
    }
    const float ft = (float)tot;
    for (int d = 0; d < dim; ++d) out[d] = out[d] / ft;
}

void xv_oracle_chunk_average_f64(const double *e, const int *lens, int nchunks, int dim, double *out)
{
    double tot = 0.0;
    for (int d = 0; d < dim; ++d) out[d] = 0.0;
    for (int i = 0; i < nchunks; ++i) {
        for (int d = 0; d < dim; ++d) out[d] += (double)lens[i] * e[(size_t)i * dim + d];
        tot += (double)lens[i];
    }
    for (int d = 0; d < dim; ++d) out[d] /= tot;
}

int xv_oracle_version(void) { return 1; }
