// xv_attention.hip -- self-attentive statistics pooling of ModelL2LossWithoutDropoutLReluAttention on the MI355X.
//
// The reference (local/tf/models.py:1036-1052) widens the last frame-level layer to 6*512 channels, splits its output
// into h1 | h2 (1536 each) and pools h2 with weights computed from h1:
//     n[t,:]  = tanh(h1[t,:] . W + b)                         tf.einsum('ijk,kl->ijl') + bias_add + tanh      (:1045)
//     s[t]    = n[t,:] . v                                    tf.einsum('ijk,k->ij')                          (:1046)
//     a[:]    = softmax_t(s)                                  tf.nn.softmax over the frames of the utterance  (:1046)
//     m       = sum_t a[t] h2[t,:]                            (:1048)
//     q       = sum_t a[t] h2[t,:]^2 - m^2                    (:1049)
//     pooled  = [ m | sqrt(q + 1e-5) ]                        (:1050)
// The GEMM h1.W + b is one more K=1 layer of the TDNN GEMM kernels (xv_tdnn_layer_*); this file holds the HBM-bound
// rest: the row scores, the per-chunk softmax, the weighted moments (and their backward for the training step).
//
// Numerics: products and sums of the weighted moments are carried in fp64 (the vector fp64 rate equals the fp32 rate on
// CDNA4 and the kernel is HBM-bound), so q = S2 - m^2 has no cancellation problem; q is clamped at 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "xvector_hip.h"

extern "C" void xv_internal_set_error(const char *msg);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

int att_fail(int code, const char *msg)
{
    xv_internal_set_error(msg);
    return code;
}

int att_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    xv_internal_set_error(buf);
    return (int)e;
}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// tanh(x) = 1 - 2/(e^{2x} + 1) on the hardware exp / rcp: ~1e-7 ABSOLUTE error (the rounding level of an fp32 result near
// 1; what matters for sum_c v_c tanh(u_c) and for 1 - n^2 in the backward), exact limits (e -> inf gives 1, e -> 0 gives
// -1), 5 instructions.  The library tanhf (branches, IEEE division, relative accuracy for tiny arguments) made the scores
// kernel compute-bound at 47 % of the HBM peak.
__device__ __forceinline__ float tanh_fast(float x)
{
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f);
}

// One wave64 per row: scores[r] = sum_c v[c] * tanh(u[r,c]); a lane takes float4 number lane, lane+64, ... of the row
// (<= 24 fp32 terms per lane for C = 1536), the 64 lane sums are added in fp64.
__global__ __launch_bounds__(256) void attention_scores_kernel(const float *__restrict__ u, long ldu, long R, int C,
                                                               const float *__restrict__ v, float *__restrict__ scores,
                                                               float *__restrict__ nl, long ldn)
{
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float *row = u + r * ldu;
    float part = 0.0f;
    int c = lane * 4;
    for (; c + 256 < C; c += 512) {                 // two independent 16-byte loads in flight per lane
        const f32x4 x0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(row + c));
        const f32x4 x1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(row + c + 256));
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(v + c), w1 = *reinterpret_cast<const f32x4 *>(v + c + 256);
        f32x4 t0, t1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t0[i] = tanh_fast(x0[i]);
            t1[i] = tanh_fast(x1[i]);
            part = fmaf(w0[i], t0[i], part);
            part = fmaf(w1[i], t1[i], part);
        }
        if (nl) {                                   // training keeps tanh(u) for the backward
            *reinterpret_cast<f32x4 *>(nl + r * ldn + c) = t0;
            *reinterpret_cast<f32x4 *>(nl + r * ldn + c + 256) = t1;
        }
    }
    if (c < C) {
        const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(row + c));
        const f32x4 w = *reinterpret_cast<const f32x4 *>(v + c);
        f32x4 t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t[i] = tanh_fast(x[i]);
            part = fmaf(w[i], t[i], part);
        }
        if (nl) *reinterpret_cast<f32x4 *>(nl + r * ldn + c) = t;
    }
    const double acc = wave_sum((double)part);
    if (lane == 0) scores[r] = (float)acc;
}

// One workgroup per chunk: att[rows of the chunk] = softmax(scores[rows of the chunk])  (max-shifted, sum in fp64).
__global__ __launch_bounds__(256) void attention_softmax_kernel(const float *__restrict__ scores,
                                                                const int *__restrict__ row_start,
                                                                const int *__restrict__ row_len, float *__restrict__ att)
{
    __shared__ float red_f[4];
    __shared__ double red_d[4];
    const int b = blockIdx.x;
    const int len = row_len[b];
    if (len <= 0) return;
    const float *s = scores + row_start[b];
    float *a = att + row_start[b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -__builtin_inff();
    for (int t = tid; t < len; t += 256) m = fmaxf(m, s[t]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if (lane == 0) red_f[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    double z = 0.0;
    for (int t = tid; t < len; t += 256) z += (double)expf(s[t] - m);
    z = wave_sum(z);
    if (lane == 0) red_d[wave] = z;
    __syncthreads();
    z = (red_d[0] + red_d[1]) + (red_d[2] + red_d[3]);
    const float zf = (float)z;
    for (int t = tid; t < len; t += 256) a[t] = expf(s[t] - m) / zf;
}

constexpr int APOOL_UNROLL = 8;

// Weighted moments.  Same decomposition as stats_pool_kernel: one wave64 per (chunk b, time split sp, 64-channel group),
// lane = (phase = lane>>4: which of 4 interleaved rows, cg = lane&15: which float4 of the 64 channels).
__global__ __launch_bounds__(256) void attention_pool_kernel(const float *__restrict__ h, long ldh, int C,
                                                             const float *__restrict__ att,
                                                             const int *__restrict__ row_start,
                                                             const int *__restrict__ row_len, int split_rows, int max_splits,
                                                             float eps, float *__restrict__ out, double *__restrict__ partial)
{
    const int b = blockIdx.z, sp = blockIdx.y;
    const int len = row_len[b];
    const int begin = sp * split_rows;
    if (begin >= len) return;
    const int n_rows = min(split_rows, len - begin);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int phase = lane >> 4;
    const int c = (blockIdx.x * 4 + wave) * 64 + (lane & 15) * 4;
    if (c >= C) return;
    const size_t r0 = (size_t)row_start[b] + begin;
    const float *base = h + r0 * ldh + c;
    const float *wts = att + r0;

    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    int r = phase;
    for (; r + 4 * (APOOL_UNROLL - 1) < n_rows; r += 4 * APOOL_UNROLL) {
        f32x4 v[APOOL_UNROLL];
        float a[APOOL_UNROLL];
#pragma unroll
        for (int i = 0; i < APOOL_UNROLL; ++i) {
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (size_t)(r + 4 * i) * ldh));
            a[i] = wts[r + 4 * i];
        }
#pragma unroll
        for (int i = 0; i < APOOL_UNROLL; ++i) {
            const double w = (double)a[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double x = (double)v[i][k];
                s1[k] = fma(w, x, s1[k]);
                s2[k] = fma(w, x * x, s2[k]);
            }
        }
    }
    for (; r < n_rows; r += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(base + (size_t)r * ldh);
        const double w = (double)wts[r];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double x = (double)v[k];
            s1[k] = fma(w, x, s1[k]);
            s2[k] = fma(w, x * x, s2[k]);
        }
    }
    // the 4 row phases of the wave: fixed pairing (16, then 32) -> deterministic
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s1[k] += __shfl_xor(s1[k], off, 64);
            s2[k] += __shfl_xor(s2[k], off, 64);
        }
    }
    if (phase != 0) return;
    if (max_splits == 1) {
        f32x4 mo, so;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mo[k] = (float)s1[k];
            so[k] = sqrtf((float)fmax(s2[k] - s1[k] * s1[k], 0.0) + eps);
        }
        float *o = out + (size_t)b * 2 * C;
        *reinterpret_cast<f32x4 *>(o + c) = mo;
        *reinterpret_cast<f32x4 *>(o + C + c) = so;
    } else {
        double *pm = partial + ((size_t)b * max_splits + sp) * 2 * C;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pm[c + k] = s1[k];
            pm[C + c + k] = s2[k];
        }
    }
}

__global__ void attention_pool_merge_kernel(const double *__restrict__ partial, int C, const int *__restrict__ row_len,
                                            int split_rows, int max_splits, float eps, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int len = row_len[b];
    double s1 = 0.0, s2 = 0.0;
    for (int sp = 0; sp * split_rows < len; ++sp) {
        const double *pm = partial + ((size_t)b * max_splits + sp) * 2 * C;
        s1 += pm[c];
        s2 += pm[C + c];
    }
    out[(size_t)b * 2 * C + c] = (float)s1;
    out[(size_t)b * 2 * C + C + c] = sqrtf((float)fmax(s2 - s1 * s1, 0.0) + eps);
}


// ---- backward (training step) -----------------------------------------------------------------------------------------
// pooled = [m | sd], sd = sqrt(S2 - m^2 + eps), m = sum_t a_t x_t, S2 = sum_t a_t x_t^2.  With g2 = dsd/(2 sd) and
// g1 = dm - 2 m g2:   dx[t,c] = a_t (g1[c] + 2 x[t,c] g2[c]),   da_t = sum_c x[t,c] g1[c] + x[t,c]^2 g2[c].
// One workgroup per (chunk, block of APB_ROWS rows): g1, g2 of the chunk are formed once per workgroup in LDS (2*C floats,
// one division per channel instead of one per element), then each of the 4 waves takes every 4th row of the block.
constexpr int APB_ROWS = 32;

__global__ __launch_bounds__(256) void attention_pool_backward_kernel(const float *__restrict__ h, long ldh, int C,
                                                                      const float *__restrict__ att,
                                                                      const int *__restrict__ row_start,
                                                                      const int *__restrict__ row_len,
                                                                      const float *__restrict__ pooled,
                                                                      const float *__restrict__ dpooled,
                                                                      float *__restrict__ dh, long lddh, float *__restrict__ datt)
{
    extern __shared__ float g[];                     // g1[C] | g2[C]
    const int b = blockIdx.y;
    const int len = row_len[b];
    const int t0 = blockIdx.x * APB_ROWS;
    if (t0 >= len) return;
    const float *pm = pooled + (size_t)b * 2 * C, *pd = dpooled + (size_t)b * 2 * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g2 = pd[C + c] / (2.0f * pm[C + c]);
        g[C + c] = g2;
        g[c] = pd[c] - 2.0f * pm[c] * g2;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int t1 = min(t0 + APB_ROWS, len);
    for (int t = t0 + (threadIdx.x >> 6); t < t1; t += 4) {
        const size_t r = (size_t)row_start[b] + t;
        const float a = att[r];
        double acc = 0.0;
        for (int c = lane * 4; c < C; c += 256) {
            const f32x4 x = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(h + r * ldh + c));
            const f32x4 g1 = *reinterpret_cast<const f32x4 *>(g + c), g2 = *reinterpret_cast<const f32x4 *>(g + C + c);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i] = a * (g1[i] + 2.0f * x[i] * g2[i]);
                acc += (double)x[i] * ((double)g1[i] + (double)x[i] * (double)g2[i]);
            }
            __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(dh + r * lddh + c));
        }
        acc = wave_sum(acc);
        if (lane == 0) datt[r] = (float)acc;
    }
}

// a = softmax(s) over the rows of a chunk:  ds_t = a_t (da_t - sum_tau a_tau da_tau).  One workgroup per chunk.
__global__ __launch_bounds__(256) void attention_softmax_backward_kernel(const float *__restrict__ att,
                                                                         const float *__restrict__ datt,
                                                                         const int *__restrict__ row_start,
                                                                         const int *__restrict__ row_len,
                                                                         float *__restrict__ dscores)
{
    __shared__ double red[4];
    const int b = blockIdx.x;
    const int len = row_len[b];
    if (len <= 0) return;
    const size_t r0 = row_start[b];
    const int tid = threadIdx.x;
    double dot = 0.0;
    for (int t = tid; t < len; t += 256) dot += (double)att[r0 + t] * (double)datt[r0 + t];
    dot = wave_sum(dot);
    if ((tid & 63) == 0) red[tid >> 6] = dot;
    __syncthreads();
    dot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int t = tid; t < len; t += 256) dscores[r0 + t] = (float)((double)att[r0 + t] * ((double)datt[r0 + t] - dot));
}

// s_r = sum_c v_c n_rc, n = tanh(u):  du_rc = ds_r v_c (1 - n_rc^2);  nonlin is overwritten with ds_r n_rc, whose column sums
// are dv (xv_col_sums_f32).  One wave64 per row.
__global__ __launch_bounds__(256) void attention_scores_backward_kernel(float *__restrict__ nl, long ldn,
                                                                        const float *__restrict__ dscores,
                                                                        const float *__restrict__ v, long R, int C,
                                                                        float *__restrict__ du, long lddu)
{
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float ds = dscores[r];
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 n = *reinterpret_cast<const f32x4 *>(nl + r * ldn + c);
        const f32x4 w = *reinterpret_cast<const f32x4 *>(v + c);
        f32x4 o, g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[i] = ds * w[i] * (1.0f - n[i] * n[i]);
            g[i] = ds * n[i];
        }
        *reinterpret_cast<f32x4 *>(du + r * lddu + c) = o;
        *reinterpret_cast<f32x4 *>(nl + r * ldn + c) = g;
    }
}

}  // namespace

extern "C" {

int xv_attention_scores_f32(const float *u, int64_t ldu, int64_t R, int c, const float *v, float *scores, float *nonlin,
                            int64_t ldn, void *stream)
{
    if (R <= 0) return 0;
    if (!u || !v || !scores || c <= 0 || (c & 3) || (ldu & 3) || ldu < c || (((uintptr_t)u) & 15) || (((uintptr_t)v) & 15))
        return att_fail(XV_ERR_BAD_ARG, "attention_scores: C, ldu must be multiples of 4 and u/v 16-byte aligned");
    if (nonlin && ((ldn & 3) || ldn < c || (((uintptr_t)nonlin) & 15)))
        return att_fail(XV_ERR_BAD_ARG, "attention_scores: bad nonlin buffer");
    hipLaunchKernelGGL(attention_scores_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, u, (long)ldu,
                       (long)R, c, v, scores, nonlin, (long)ldn);
    return att_check("attention_scores_kernel");
}

int xv_attention_softmax_f32(const float *scores, const int32_t *row_start, const int32_t *row_len, int nchunks, float *att,
                             void *stream)
{
    if (nchunks <= 0) return 0;
    if (!scores || !row_start || !row_len || !att) return att_fail(XV_ERR_BAD_ARG, "attention_softmax: NULL pointer");
    hipLaunchKernelGGL(attention_softmax_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, scores, row_start, row_len, att);
    return att_check("attention_softmax_kernel");
}

size_t xv_attention_pool_workspace_bytes(int c, int nchunks, int max_len, int split_rows)
{
    if (split_rows <= 0 || max_len <= split_rows) return 0;
    const size_t splits = ((size_t)max_len + split_rows - 1) / split_rows;
    return (size_t)nchunks * splits * 2 * (size_t)c * sizeof(double);
}

int xv_attention_pool_f32(const float *h, int64_t ldh, int c, const float *att, const int32_t *row_start, const int32_t *row_len,
                          int nchunks, int max_len, int split_rows, float eps, float *out, void *workspace, void *stream)
{
    if (nchunks <= 0) return 0;
    if (!h || !att || !row_start || !row_len || !out) return att_fail(XV_ERR_BAD_ARG, "attention_pool: NULL pointer");
    if (c <= 0 || (c & 3) || (ldh & 3) || ldh < c || (((uintptr_t)h) & 15) || (((uintptr_t)out) & 15))
        return att_fail(XV_ERR_BAD_ARG, "attention_pool: C, ldh must be multiples of 4 and h/out 16-byte aligned");
    if (split_rows <= 0 || max_len <= 0) return att_fail(XV_ERR_BAD_ARG, "attention_pool: split_rows/max_len must be > 0");
    const int max_splits = (max_len + split_rows - 1) / split_rows;
    if (max_splits > 1 && (!workspace || (((uintptr_t)workspace) & 7)))
        return att_fail(XV_ERR_BAD_ARG, "attention_pool: 8-byte aligned workspace required for split chunks");
    if (max_splits > 65535) return att_fail(XV_ERR_UNSUPPORTED, "attention_pool: too many splits");
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < nchunks; b0 += 65535) {              // grid.z <= 65535
        const int nb = nchunks - b0 < 65535 ? nchunks - b0 : 65535;
        double *ws = (double *)workspace + (size_t)b0 * max_splits * 2 * c;
        hipLaunchKernelGGL(attention_pool_kernel, dim3((c + 255) / 256, max_splits, nb), dim3(256), 0, st, h, (long)ldh, c, att,
                           row_start + b0, row_len + b0, split_rows, max_splits, eps, out + (size_t)b0 * 2 * c, ws);
        int rc = att_check("attention_pool_kernel");
        if (rc) return rc;
        if (max_splits > 1) {
            hipLaunchKernelGGL(attention_pool_merge_kernel, dim3((c + 255) / 256, nb), dim3(256), 0, st, ws, c, row_len + b0,
                               split_rows, max_splits, eps, out + (size_t)b0 * 2 * c);
            rc = att_check("attention_pool_merge_kernel");
            if (rc) return rc;
        }
    }
    return 0;
}

int xv_attention_pool_backward_f32(const float *h, int64_t ldh, int c, const float *att, const int32_t *row_start,
                                   const int32_t *row_len, int nchunks, int max_len, const float *pooled, const float *dpooled,
                                   float *dh, int64_t lddh, float *datt, void *stream)
{
    if (nchunks <= 0 || max_len <= 0) return 0;
    if (!h || !att || !row_start || !row_len || !pooled || !dpooled || !dh || !datt)
        return att_fail(XV_ERR_BAD_ARG, "attention_pool_backward: NULL pointer");
    if (c <= 0 || (c & 3) || (ldh & 3) || (lddh & 3) || ldh < c || lddh < c || (((uintptr_t)h) & 15) || (((uintptr_t)dh) & 15) ||
        (((uintptr_t)pooled) & 15) || (((uintptr_t)dpooled) & 15))
        return att_fail(XV_ERR_BAD_ARG, "attention_pool_backward: C, ld must be multiples of 4 and buffers 16-byte aligned");
    if (c > 8192) return att_fail(XV_ERR_UNSUPPORTED, "attention_pool_backward: more than 8192 channels");      // 64 KB of LDS
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < nchunks; b0 += 65535) {
        const int nb = nchunks - b0 < 65535 ? nchunks - b0 : 65535;
        hipLaunchKernelGGL(attention_pool_backward_kernel, dim3((max_len + APB_ROWS - 1) / APB_ROWS, nb), dim3(256),
                           2 * (size_t)c * sizeof(float), st, h, (long)ldh, c, att,
                           row_start + b0, row_len + b0, pooled + (size_t)b0 * 2 * c, dpooled + (size_t)b0 * 2 * c, dh, (long)lddh,
                           datt);
        int rc = att_check("attention_pool_backward_kernel");
        if (rc) return rc;
    }
    return 0;
}

int xv_attention_softmax_backward_f32(const float *att, const float *datt, const int32_t *row_start, const int32_t *row_len,
                                      int nchunks, float *dscores, void *stream)
{
    if (nchunks <= 0) return 0;
    if (!att || !datt || !row_start || !row_len || !dscores) return att_fail(XV_ERR_BAD_ARG, "attention_softmax_backward: NULL pointer");
    hipLaunchKernelGGL(attention_softmax_backward_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, att, datt, row_start,
                       row_len, dscores);
    return att_check("attention_softmax_backward_kernel");
}

int xv_attention_scores_backward_f32(float *nonlin, int64_t ldn, const float *dscores, const float *v, int64_t R, int c, float *du,
                                     int64_t lddu, void *stream)
{
    if (R <= 0) return 0;
    if (!nonlin || !dscores || !v || !du || c <= 0 || (c & 3) || (ldn & 3) || (lddu & 3) || ldn < c || lddu < c ||
        (((uintptr_t)nonlin) & 15) || (((uintptr_t)du) & 15) || (((uintptr_t)v) & 15))
        return att_fail(XV_ERR_BAD_ARG, "attention_scores_backward: C, ld must be multiples of 4 and buffers 16-byte aligned");
    hipLaunchKernelGGL(attention_scores_backward_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, nonlin,
                       (long)ldn, dscores, v, (long)R, c, du, (long)lddu);
    return att_check("attention_scores_backward_kernel");
}

}  // extern "C"
