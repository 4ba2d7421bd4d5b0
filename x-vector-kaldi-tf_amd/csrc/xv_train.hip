// xv_train.hip -- gfx950 kernels for the TRAINING step (reference: Model.train_one_iteration,
// local/tf/models.py:216-305, graph local/tf/models.py:466-534, BN train branch local/tf/tf_block.py:18-23).
//
// The two dense GEMMs of a layer's forward and input-gradient reuse the inference kernel (xv_tdnn_layer_f32: dgrad is
// the same implicit-im2col GEMM with tap-flipped, transposed weights).  This file adds what training needs on top:
//   wgrad_kernel          dW[k,c,o] = sum_r x[r+(k-(K-1)/2)d, c] * dz[r,o]   (fp32 MFMA, reduction over rows, split + ordered merge)
//   col_sums_kernel       per-channel sum_r a, sum_r a*b with fp64 accumulators (db, BN-backward statistics)
//   merge_moments_kernel  per-chunk (mean, var) -> batch (mean, biased var)   (tf.nn.moments over all frames)
//   rows_affine_kernel    y = valid ? x*scale + shift : 0                      (BN with batch statistics)
//   bn_coeffs / bn_act_backward  closed-form BN backward through the activation
//   pool_backward_kernel  gradient of [mean || sqrt(var+eps)] w.r.t. every frame
//   softmax_ce_kernel     loss / accuracy / dlogits of tf.nn.softmax_cross_entropy_with_logits + reduce_mean
//   adam_kernel, ema_kernel
// Everything is fp32 storage; reductions accumulate in fp64.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "xvector_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" void xv_internal_set_error(const char *msg);

namespace {

int tfail(int code, const char *msg)
{
    xv_internal_set_error(msg);          // shared with xv_last_error() (xv_kernels.hip)
    return code;
}
int tcheck(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    xv_internal_set_error(buf);
    return (int)e;
}

// ------------------------------------------------------------------------------------------------
// wgrad: D[cin 128][cout 128] per (tap, tile) accumulated over a range of rows
// ------------------------------------------------------------------------------------------------
constexpr int WT = 128;      // tile edge (channels)
constexpr int WR = 32;       // rows per step
constexpr int WLD = WT + 4;  // LDS row stride (floats): the two 32-lane halves of a ds_read_b32 land 4 banks apart

struct WgradParams {
    const float *x;
    const float *dz;
    long R;
    int cin, ldx, cout, lddz, K, dil;
    int n_ct, n_ot;      // tiles along cin / cout
    long rows_per_split;
    float *out;          // [nsplit][K][cin][cout] (or dw itself when nsplit == 1)
};

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) float As[WR * WLD];
    __shared__ __attribute__((aligned(16))) float Bs[WR * WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    int t = blockIdx.x;
    const int ot = t % p.n_ot; t /= p.n_ot;
    const int ct = t % p.n_ct; t /= p.n_ct;
    const int k = t;
    const int c0 = ct * WT, o0 = ot * WT;
    const long shift = (long)(k - (p.K - 1) / 2) * p.dil;
    const long r_begin = (long)blockIdx.y * p.rows_per_split;
    const long r_end = min(p.R, r_begin + p.rows_per_split);

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    const bool vec = !(p.cin & 3) && !(p.ldx & 3) && !(p.cout & 3) && !(p.lddz & 3);
    f32x4 va[4], vb[4];
    // 32 rows x 128 channels of x (shifted by the tap) and of dz -> registers (prefetched one step ahead)
    auto load = [&](long r0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            const int rr = f >> 5, q = f & 31;           // row, float4 index
            const long ra = r0 + rr + shift, rb = r0 + rr;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
            const int ca = c0 + q * 4, cb = o0 + q * 4;
            if (rb < r_end) {
                if (vec) {
                    if (ra >= 0 && ra < p.R && ca < p.cin) a = *reinterpret_cast<const f32x4 *>(p.x + (size_t)ra * p.ldx + ca);
                    if (cb < p.cout) b = *reinterpret_cast<const f32x4 *>(p.dz + (size_t)rb * p.lddz + cb);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (ra >= 0 && ra < p.R && ca + i < p.cin) a[i] = p.x[(size_t)ra * p.ldx + ca + i];
                        if (cb + i < p.cout) b[i] = p.dz[(size_t)rb * p.lddz + cb + i];
                    }
                }
            }
            va[j] = a;
            vb[j] = b;
        }
    };
    load(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += WR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = tid + 256 * j;
            *reinterpret_cast<f32x4 *>(As + (f >> 5) * WLD + (f & 31) * 4) = va[j];
            *reinterpret_cast<f32x4 *>(Bs + (f >> 5) * WLD + (f & 31) * 4) = vb[j];
        }
        __syncthreads();
        if (r0 + WR < r_end) load(r0 + WR);              // in flight under the 64 MFMAs below
        const float *ap = As + (lane >> 5) * WLD + wi * 64 + (lane & 31);
        const float *bp = Bs + (lane >> 5) * WLD + wj * 64 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < WR / 2; ++kk) {
            const float a0 = ap[kk * 2 * WLD], a1 = ap[kk * 2 * WLD + 32];
            const float b0 = bp[kk * 2 * WLD], b1 = bp[kk * 2 * WLD + 32];
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
        __syncthreads();
    }
    // D: col = lane&31 (cout), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cin)
    float *out = p.out + ((size_t)blockIdx.y * p.K + k) * (size_t)p.cin * p.cout;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
            const f32x16 &a = bi == 0 ? (bj == 0 ? acc00 : acc01) : (bj == 0 ? acc10 : acc11);
            const int o = o0 + wj * 64 + bj * 32 + (lane & 31);
            if (o >= p.cout) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int c = c0 + wi * 64 + bi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (c < p.cin) out[(size_t)c * p.cout + o] = a[reg];
            }
        }
}

__global__ void sum_splits_kernel(const float *__restrict__ part, size_t n, int nsplit, float *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int j = 0; j < nsplit; ++j) s += part[(size_t)j * n + i];      // fixed order: deterministic
    out[i] = s;
}

// ------------------------------------------------------------------------------------------------
// column sums with fp64 accumulation: out_a[c] = sum_r a[r,c], out_ab[c] = sum_r a[r,c]*b[r,c]
// ------------------------------------------------------------------------------------------------
constexpr long CS_ROWS = 128;      // rows per split: R = 19.6k frames -> ~150 splits x C/256 blocks

// block = 4 waves; a wave covers 256 channels (float4 per lane) of one row per load; waves take interleaved rows
__global__ __launch_bounds__(256) void col_sums_kernel(const float *__restrict__ a, const float *__restrict__ b, long R, int C,
                                                       int lda, int ldb, long rows_per_split, double *__restrict__ part)
{
    __shared__ double sa[4][256], sb[4][256];
    const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + lane * 4;
    const long r0 = (long)blockIdx.y * rows_per_split, r1 = min(R, r0 + rows_per_split);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sab[4] = {0.0, 0.0, 0.0, 0.0};
    const bool vec = c + 4 <= C && !(lda & 3) && !(ldb & 3);
    if (c < C)
        for (long r = r0 + ty; r < r1; r += 4) {
            float av[4], bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (vec) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(a + (size_t)r * lda + c);
                av[0] = t[0]; av[1] = t[1]; av[2] = t[2]; av[3] = t[3];
                if (b) {
                    const f32x4 u = *reinterpret_cast<const f32x4 *>(b + (size_t)r * ldb + c);
                    bv[0] = u[0]; bv[1] = u[1]; bv[2] = u[2]; bv[3] = u[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    av[i] = c + i < C ? a[(size_t)r * lda + c + i] : 0.f;
                    if (b) bv[i] = c + i < C ? b[(size_t)r * ldb + c + i] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] += av[i];
                sab[i] += (double)av[i] * (double)bv[i];
            }
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sa[ty][lane * 4 + i] = s[i];
        sb[ty][lane * 4 + i] = sab[i];
    }
    __syncthreads();
    const int cc = blockIdx.x * 256 + threadIdx.x;
    if (cc < C) {
        double *o = part + (size_t)blockIdx.y * 2 * C;
        const int t = threadIdx.x;
        o[cc] = sa[0][t] + sa[1][t] + sa[2][t] + sa[3][t];
        o[C + cc] = sb[0][t] + sb[1][t] + sb[2][t] + sb[3][t];
    }
}

// 64 channels x 4 split groups per block: group g sums the splits j = g, g+4, ... in order, the four group sums are then
// added in group order (fixed order -> deterministic); 4x the parallelism and a quarter of the dependent loads of one
// thread per channel.
// 64 channels x 16 groups of splits per workgroup: a launch has only C / 64 workgroups, so what a thread does serially is the
// whole run time (150 splits over 4 groups were 38 dependent-latency loads = 11 us for a few kilobytes; 16 groups: 10)
constexpr int CSM_GROUPS = 16;
__global__ __launch_bounds__(64 * CSM_GROUPS) void col_sums_merge_kernel(const double *__restrict__ part, int C, int nsplit,
                                                                         float *__restrict__ out_a, float *__restrict__ out_ab)
{
    __shared__ double sh[2][CSM_GROUPS][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, sab = 0.0;
    if (c < C) {
        for (int j = g; j < nsplit; j += CSM_GROUPS) {
            s += part[(size_t)j * 2 * C + c];
            sab += part[(size_t)j * 2 * C + C + c];
        }
    }
    sh[0][g][cl] = s;
    sh[1][g][cl] = sab;
    __syncthreads();
    if (g == 0 && c < C) {
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int k = 0; k < CSM_GROUPS; ++k) {               // fixed order: deterministic
            ta += sh[0][k][cl];
            tb += sh[1][k][cl];
        }
        out_a[c] = (float)ta;
        if (out_ab) out_ab[c] = (float)tb;
    }
}

// Batch moments of a forward layer from the partial sums its GEMM left behind (xv_tdnn_layer_bf16x3_moments: per 128-row tile
// [sum y | sum y^2] in double), and the BN fold that follows, in one launch: mean = S1 / N, var = S2 / N - mean^2 (biased, what
// tf.nn.moments returns), scale = gamma / sqrt(var + eps), shift = beta - mean * scale -- the last two with the operations and the
// rounding order of fold_bn_kernel (xv_kernels.hip), so that a checkpoint's eval-mode fold and this one share their bits for equal
// moments.  Replaces chunk_moments + merge_moments + fold_bn (three launches and a pass over the activations) for the layers whose
// forward GEMM is the bf16x3 one.
__global__ __launch_bounds__(64 * CSM_GROUPS) void moments_fold_kernel(const double *__restrict__ part, int C, int nsplit, float n_frames,
                                                                       const float *gamma, const float *beta, float eps, float *mean,
                                                                       float *var, float *scale, float *shift)
{
#pragma clang fp contract(off)
    __shared__ double sh[2][CSM_GROUPS][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, sq = 0.0;
    if (c < C) {
        for (int j = g; j < nsplit; j += CSM_GROUPS) {
            s += part[(size_t)j * 2 * C + c];
            sq += part[(size_t)j * 2 * C + C + c];
        }
    }
    sh[0][g][cl] = s;
    sh[1][g][cl] = sq;
    __syncthreads();
    if (g != 0 || c >= C) return;
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int k = 0; k < CSM_GROUPS; ++k) {                   // fixed order: deterministic
        ta += sh[0][k][cl];
        tb += sh[1][k][cl];
    }
    const double N = n_frames;
    const double m = ta / N;
    const double v = fmax(tb / N - m * m, 0.0);
    const float mf = (float)m, vf = (float)v;
    mean[c] = mf;
    var[c] = vf;
    const float sc = gamma[c] * (1.0f / sqrtf(vf + eps));
    float ms = mf * sc;
    asm volatile("" : "+v"(ms));
    scale[c] = sc;
    shift[c] = beta[c] - ms;
}

// per-chunk (mean, biased var) -> moments over all frames of all chunks, fp64, two passes over the chunk table:
// mu = sum_b n_b m_b / N, then var = sum_b n_b (v_b + (m_b - mu)^2) / N -- the definition, no cancellation.  64 channels x 16 groups of
// chunks per workgroup (a serial Chan merge of 64 chunks with its fp64 divisions was 17 us on the handful of workgroups a launch has).
constexpr int MM_GROUPS = 16;
__global__ __launch_bounds__(64 * MM_GROUPS) void merge_moments_kernel(const float *__restrict__ cm, const int *__restrict__ row_len,
                                                                       int nchunks, int C, float *__restrict__ mean, float *__restrict__ var)
{
    __shared__ double sh[2][MM_GROUPS][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    double n = 0.0, s1 = 0.0;
    for (int b = g; b < nchunks; b += MM_GROUPS) {
        const double m = (double)row_len[b];
        if (m > 0.0 && ok) {
            n += m;
            s1 += m * (double)cm[(size_t)b * 2 * C + c];
        }
    }
    sh[0][g][cl] = n;
    sh[1][g][cl] = s1;
    __syncthreads();
    double N = 0.0, S1 = 0.0;
#pragma unroll
    for (int k = 0; k < MM_GROUPS; ++k) {                    // fixed order: deterministic
        N += sh[0][k][cl];
        S1 += sh[1][k][cl];
    }
    const double mu = N > 0.0 ? S1 / N : 0.0;
    __syncthreads();
    double m2 = 0.0;
    for (int b = g; b < nchunks; b += MM_GROUPS) {
        const double m = (double)row_len[b];
        if (m > 0.0 && ok) {
            const double d = (double)cm[(size_t)b * 2 * C + c] - mu;
            m2 += m * ((double)cm[(size_t)b * 2 * C + C + c] + d * d);
        }
    }
    sh[0][g][cl] = m2;
    __syncthreads();
    if (g == 0 && ok) {
        double M2 = 0.0;
#pragma unroll
        for (int k = 0; k < MM_GROUPS; ++k) M2 += sh[0][k][cl];
        mean[c] = (float)mu;
        var[c] = (float)(N > 0.0 ? M2 / N : 0.0);
    }
}

// (one workgroup row = 256 threads x 4 channels; rows along blockIdx.y with a stride loop: no 64-bit division per element)
// SPLIT: the same values once more in the bf16 split activation format (include/xvector_hip.h: per row and 32-channel slab 128 bytes =
// 4 hi slots + 4 lo slots of 8 bf16, slot t at t ^ ((row >> 1) & 7)) -- a thread's 4 channels are half a slot of either plane.  The
// K = 1 layers of the training step read it: the DMA-fed GEMM instead of the one that splits fp32 rows while staging them.
typedef __bf16 tbf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_split4(uint8_t *ys, long r, int C, int c, const f32x4 &o)
{
    tbf16x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hi[i] = (__bf16)o[i];
        lo[i] = (__bf16)(o[i] - (float)hi[i]);
    }
    const int q = (c & 31) >> 2, sw = (int)(r >> 1) & 7;
    uint8_t *row = ys + ((size_t)r * (C >> 5) + (c >> 5)) * 128 + (q & 1) * 8;
    *reinterpret_cast<tbf16x4 *>(row + (((q >> 1)) ^ sw) * 16) = hi;
    *reinterpret_cast<tbf16x4 *>(row + ((4 + (q >> 1)) ^ sw) * 16) = lo;
}

template <bool VEC, bool SPLIT = false>
__global__ __launch_bounds__(256) void rows_affine_kernel(const float *__restrict__ x, long R, int C, int ldx, const float *__restrict__ scale,
                                                          const float *__restrict__ shift, const uint8_t *__restrict__ valid,
                                                          float *__restrict__ y, int ldy, uint8_t *__restrict__ ys = nullptr)
{
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    float sc[4], sh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sc[i] = c + i < C ? scale[c + i] : 0.f;
        sh[i] = c + i < C ? shift[c + i] : 0.f;
    }
    for (long r = blockIdx.y; r < R; r += gridDim.y) {
        const bool keep = !valid || valid[r];
        if constexpr (VEC) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (size_t)r * ldx + c);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = keep ? v[i] * sc[i] + sh[i] : 0.f;
            *reinterpret_cast<f32x4 *>(y + (size_t)r * ldy + c) = o;
            if constexpr (SPLIT) store_split4(ys, r, C, c, o);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c + i < C) y[(size_t)r * ldy + c + i] = keep ? x[(size_t)r * ldx + c + i] * sc[i] + sh[i] : 0.f;
        }
    }
}

// tf.nn.dropout(h, keep_prob) (models.py:70-72,92-94) with a stateless mask: element (row, col) of call `seed` is kept iff
// the top 32 bits of splitmix64(seed ^ golden*(row*C + col + 1)) are below keep_prob*2^32; kept elements are scaled by
// 1/keep_prob.  The same call on the gradient is the backward pass -- no mask is stored.
__device__ __forceinline__ uint32_t dropout_bits(uint64_t seed, uint64_t idx)
{
    uint64_t x = seed ^ ((idx + 1) * 0x9E3779B97F4A7C15ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}

__global__ void dropout_kernel(float *__restrict__ x, long R, int C, int ld, uint64_t seed, uint32_t threshold, float inv_keep)
{
    const size_t n = (size_t)R * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long row = (long)(i / C);
        const int c = (int)(i - (size_t)row * C);
        const size_t o = (size_t)row * ld + c;
        x[o] = dropout_bits(seed, i) < threshold ? x[o] * inv_keep : 0.f;
    }
}

// PReLU backward (tf_block.py:38-47: max(0,z) + alpha[c]*min(0,z)): given dr = dL/d(act output) and the pre-activation z,
//   dz = dr * (z > 0 ? 1 : alpha[c])   (written over dr),   t = dr * min(z, 0)   (written over z; its column sums are dalpha)
__global__ void prelu_backward_kernel(float *__restrict__ dr, float *__restrict__ z, long R, int C, int ld,
                                      const float *__restrict__ alpha)
{
    const size_t n = (size_t)R * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long row = (long)(i / C);
        const int c = (int)(i - (size_t)row * C);
        const size_t o = (size_t)row * ld + c;
        const float zv = z[o], g = dr[o];
        dr[o] = zv > 0.f ? g : alpha[c] * g;
        z[o] = g * fminf(zv, 0.f);
    }
}

// BN backward through the activation, closed form.  With xhat = (r-mean)*rstd, N frames:
//   dbeta = sum dh ; dgamma = sum dh*xhat = rstd*(sum dh*r - mean*sum dh)
//   dr = gamma*rstd*(dh - dbeta/N - xhat*dgamma/N) = A*dh + B*r + Cc
__global__ void bn_coeffs_kernel(const float *sum_dh, const float *sum_dh_r, const float *mean, const float *var,
                                 const float *gamma, float eps, float n_frames, int C, float *dgamma, float *dbeta, float *coefA,
                                 float *coefB, float *coefC)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double rstd = 1.0 / sqrt((double)var[c] + (double)eps);
    const double db = sum_dh[c];
    const double dg = rstd * ((double)sum_dh_r[c] - (double)mean[c] * db);
    const double g = gamma[c], N = n_frames;
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
    coefA[c] = (float)(g * rstd);
    coefB[c] = (float)(-g * rstd * rstd * dg / N);
    coefC[c] = (float)(-g * rstd * db / N + g * rstd * rstd * (double)mean[c] * dg / N);
}

// col_sums_merge_kernel and bn_coeffs_kernel in one launch: the partial column sums of (dh, dh * r) -- written per 128-row tile by the
// input-gradient GEMM that produced dh (xv_tdnn_layer_bf16x3_sums) or by col_sums_kernel -- are merged in the same fixed order and the
// thread that holds a channel's two sums goes on to the BN-backward coefficients.
__global__ __launch_bounds__(64 * CSM_GROUPS) void col_sums_merge_coeffs_kernel(const double *__restrict__ part, int C, int nsplit,
                                                                                const float *mean, const float *var, const float *gamma,
                                                                                float eps, float n_frames, float *dgamma, float *dbeta,
                                                                                float *coefA, float *coefB, float *coefC)
{
    __shared__ double sh[2][CSM_GROUPS][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0, sab = 0.0;
    if (c < C) {
        for (int j = g; j < nsplit; j += CSM_GROUPS) {
            s += part[(size_t)j * 2 * C + c];
            sab += part[(size_t)j * 2 * C + C + c];
        }
    }
    sh[0][g][cl] = s;
    sh[1][g][cl] = sab;
    __syncthreads();
    if (g == 0 && c < C) {
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int k = 0; k < CSM_GROUPS; ++k) {
            ta += sh[0][k][cl];
            tb += sh[1][k][cl];
        }
        // (the sums pass through fp32 exactly as they do between col_sums_merge_kernel and bn_coeffs_kernel: same bits either way)
        const double db = (double)(float)ta, sdr = (double)(float)tb;
        const double rstd = 1.0 / sqrt((double)var[c] + (double)eps);
        const double dg = rstd * (sdr - (double)mean[c] * db);
        const double gm = gamma[c], N = n_frames;
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
        coefA[c] = (float)(gm * rstd);
        coefB[c] = (float)(-gm * rstd * rstd * dg / N);
        coefC[c] = (float)(-gm * rstd * db / N + gm * rstd * rstd * (double)mean[c] * dg / N);
    }
}

__global__ void bn_act_backward_kernel(const float *__restrict__ dh, const float *__restrict__ r, long R, int C, int ld,
                                       const float *__restrict__ coefA, const float *__restrict__ coefB,
                                       const float *__restrict__ coefC, int act, float alpha, const uint8_t *__restrict__ valid,
                                       float *__restrict__ dz)
{
    const size_t n = (size_t)R * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long row = (long)(i / C);
        const int c = (int)(i - (size_t)row * C);
        const size_t o = (size_t)row * ld + c;
        const float rv = r[o];
        float dr = coefA[c] * dh[o] + coefB[c] * rv + coefC[c];
        // activation derivative from the activation OUTPUT (sign(r) == sign(z) for relu / leaky relu with alpha > 0)
        if (act == XV_ACT_RELU) dr = rv > 0.f ? dr : 0.f;
        else if (act == XV_ACT_LRELU) dr = rv > 0.f ? dr : alpha * dr;
        dz[o] = (!valid || valid[row]) ? dr : 0.f;
    }
}

// The same per 4 channels (one workgroup row = 256 threads x 4 channels, rows along blockIdx.y: no 64-bit division per element), with an
// optional copy of dz in the bf16 split format (the input-gradient GEMM of a K = 1 layer reads that)
template <bool SPLIT>
__global__ __launch_bounds__(256) void bn_act_backward_vec_kernel(const float *__restrict__ dh, const float *__restrict__ r, long R, int C, int ld,
                                                                  const float *__restrict__ coefA, const float *__restrict__ coefB,
                                                                  const float *__restrict__ coefC, int act, float alpha,
                                                                  const uint8_t *__restrict__ valid, float *__restrict__ dz,
                                                                  uint8_t *__restrict__ dzs)
{
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= C) return;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(coefA + c), b = *reinterpret_cast<const f32x4 *>(coefB + c),
                k = *reinterpret_cast<const f32x4 *>(coefC + c);
    for (long row = blockIdx.y; row < R; row += gridDim.y) {
        const size_t o = (size_t)row * ld + c;
        const f32x4 rv = *reinterpret_cast<const f32x4 *>(r + o), g = *reinterpret_cast<const f32x4 *>(dh + o);
        const bool keep = !valid || valid[row];
        f32x4 out;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float dr = a[i] * g[i] + b[i] * rv[i] + k[i];
            if (act == XV_ACT_RELU) dr = rv[i] > 0.f ? dr : 0.f;
            else if (act == XV_ACT_LRELU) dr = rv[i] > 0.f ? dr : alpha * dr;
            out[i] = keep ? dr : 0.f;
        }
        *reinterpret_cast<f32x4 *>(dz + o) = out;
        if constexpr (SPLIT) store_split4(dzs, row, C, c, out);
    }
}

// d[mean || sqrt(var+eps)] / d h[t,c]:  dmu/T + dsig*(h-mu)/(T*sig)
__global__ void pool_backward_kernel(const float *__restrict__ h, int ldh, int C, const int *__restrict__ row_start,
                                     const int *__restrict__ row_len, const float *__restrict__ pooled,
                                     const float *__restrict__ dpooled, float *__restrict__ dh)
{
    const int b = blockIdx.y;
    const int T = row_len[b];
    const size_t base = (size_t)row_start[b];
    const float invT = 1.0f / (float)T;
    const float *mu = pooled + (size_t)b * 2 * C, *sig = mu + C;
    const float *dmu = dpooled + (size_t)b * 2 * C, *dsig = dmu + C;
    const size_t n = (size_t)T * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / C), c = (int)(i - (size_t)t * C);
        const size_t o = (base + t) * ldh + c;
        dh[o] = dmu[c] * invT + dsig[c] * (h[o] - mu[c]) * invT / sig[c];
    }
}

// ------------------------------------------------------------------------------------------------
// The last frame-level layer's backward chain in two launches instead of six (memset, pool_backward, col_sums x 2, bn_coeffs,
// bn_act_backward): the gradient that reaches h = BN(r) of that layer comes from the statistics pooling alone,
//   dh[t,c] = dmu_b[c]/T + dsig_b[c] (h[t,c] - mu_b[c]) / (T sig_b[c])            (chunk b, T frames),
// so the two column sums the BN backward needs follow from per-CHUNK numbers the forward pass already holds -- with h = s r + shift,
// s = gamma rstd, and (m_b, v_b) the chunk moments of r (xv_chunk_moments_f32):
//   sum_t dh        = sum_b dmu_b                            (the second term sums to zero inside a chunk)
//   sum_t dh r      = sum_b [ dmu_b m_b + dsig_b s v_b / sig_b ]    (sum_t (h - mu_b) r = s T v_b)
// -- no pass over the [R, C] matrices -- and dh itself never has to exist: the element-wise kernel forms it on the fly.
// 64 channels x CSM_GROUPS groups of chunks per workgroup, the group sums added in group order (as col_sums_merge_kernel: a thread
// per channel walking all chunks alone was 37 us of dependent loads for a few kilobytes)
__global__ __launch_bounds__(64 * CSM_GROUPS) void pool_bn_coeffs_kernel(const float *__restrict__ pooled, const float *__restrict__ dpooled,
                                                                         const float *__restrict__ cm, int nchunks, const float *mean,
                                                                         const float *var, const float *gamma, float eps, float n_frames,
                                                                         int C, float *dgamma, float *dbeta, float *coefA, float *coefB,
                                                                         float *coefC)
{
    __shared__ double sh[2][CSM_GROUPS][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double rstd = 0.0, g = 0.0, s = 0.0;
    if (c < C) {
        rstd = 1.0 / sqrt((double)var[c] + (double)eps);
        g = gamma[c];
        s = g * rstd;
    }
    double db = 0.0, sdr = 0.0;
    if (c < C)
        for (int b = grp; b < nchunks; b += CSM_GROUPS) {
            const size_t o = (size_t)b * 2 * C + c;
            const double dmu = dpooled[o], dsig = dpooled[o + C];
            db += dmu;
            sdr += dmu * (double)cm[o] + dsig * s * (double)cm[o + C] / (double)pooled[o + C];
        }
    sh[0][grp][cl] = db;
    sh[1][grp][cl] = sdr;
    __syncthreads();
    if (grp != 0 || c >= C) return;
    db = sdr = 0.0;
#pragma unroll
    for (int k = 0; k < CSM_GROUPS; ++k) {                 // fixed order: deterministic
        db += sh[0][k][cl];
        sdr += sh[1][k][cl];
    }
    const double N = n_frames;
    const double dg = rstd * (sdr - (double)mean[c] * db);
    dgamma[c] = (float)dg;
    dbeta[c] = (float)db;
    coefA[c] = (float)(g * rstd);
    coefB[c] = (float)(-g * rstd * rstd * dg / N);
    coefC[c] = (float)(-g * rstd * db / N + g * rstd * rstd * (double)mean[c] * dg / N);
}

// grid (C / 512, chunks, row groups); 128 threads x 4 channels.  A workgroup column also zeroes the gap rows that follow its
// chunk (and, for chunk 0, the rows in front of it): the input-gradient GEMM reads them as halo.
template <bool SPLIT>
__global__ __launch_bounds__(128) void pool_bn_act_backward_kernel(const float *__restrict__ h, const float *__restrict__ r, int ld, int C,
                                                                   const int *__restrict__ row_start, const int *__restrict__ row_len,
                                                                   int nchunks, long R, const float *__restrict__ pooled,
                                                                   const float *__restrict__ dpooled, const float *__restrict__ coefA,
                                                                   const float *__restrict__ coefB, const float *__restrict__ coefC, int act,
                                                                   float alpha, float *__restrict__ dz, uint8_t *__restrict__ dzs)
{
    const int c = (blockIdx.x * 128 + threadIdx.x) * 4;
    if (c >= C) return;
    const int b = blockIdx.y;
    const long first = row_start[b], T = row_len[b];
    const long lo = b == 0 ? 0 : first, hi = b + 1 < nchunks ? (long)row_start[b + 1] : R;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(coefA + c), kb = *reinterpret_cast<const f32x4 *>(coefB + c),
                kc = *reinterpret_cast<const f32x4 *>(coefC + c);
    const size_t po = (size_t)b * 2 * C + c;
    const f32x4 mu = *reinterpret_cast<const f32x4 *>(pooled + po), sig = *reinterpret_cast<const f32x4 *>(pooled + po + C),
                dmu = *reinterpret_cast<const f32x4 *>(dpooled + po), dsig = *reinterpret_cast<const f32x4 *>(dpooled + po + C);
    const float invT = 1.0f / (float)T;
    f32x4 g0, g1;                                           // dh = g0 + g1 (h - mu): the expression of pool_backward_kernel, per channel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        g0[i] = dmu[i] * invT;
        g1[i] = dsig[i] * invT / sig[i];
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (long row = lo + blockIdx.z; row < hi; row += gridDim.z) {
        const size_t o = (size_t)row * ld + c;
        f32x4 out = zero;
        if (row >= first && row < first + T) {
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(h + o), rv = *reinterpret_cast<const f32x4 *>(r + o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dh = g0[i] + g1[i] * (hv[i] - mu[i]);
                float dr = a[i] * dh + kb[i] * rv[i] + kc[i];
                if (act == XV_ACT_RELU) dr = rv[i] > 0.f ? dr : 0.f;
                else if (act == XV_ACT_LRELU) dr = rv[i] > 0.f ? dr : alpha * dr;
                out[i] = dr;
            }
        }
        *reinterpret_cast<f32x4 *>(dz + o) = out;
        if constexpr (SPLIT) store_split4(dzs, row, C, c, out);
    }
}

// ------------------------------------------------------------------------------------------------
// Batch normalisation of a SMALL matrix (the segment level of a training step: 64 rows) in ONE launch each way.  The general path is
// chunk moments -> merge -> fold -> affine forward and column sums -> merge -> coefficients -> element-wise backward: four dependent
// launches of ~4.7 us each for a few kilobytes of work, twice per embedding layer, in the middle of the step where nothing overlaps
// them.  64 channels x 16 row groups per workgroup; a group's rows are r = g, g + 16, ...; sums in double, merged in group order.
// ------------------------------------------------------------------------------------------------
constexpr int BNS_GROUPS = 16;
constexpr int BNS_MAX_ROWS = 1024;

__global__ __launch_bounds__(64 * BNS_GROUPS) void bn_small_forward_kernel(const float *__restrict__ x, int ldx, int R, int C,
                                                                           const float *gamma, const float *beta, float eps,
                                                                           float *mean_out, float *var_out, float *__restrict__ y, int ldy)
{
#pragma clang fp contract(off)
    __shared__ double sh[BNS_GROUPS][64];
    __shared__ float bc[2][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    double s = 0.0;
    if (ok)
        for (int r = g; r < R; r += BNS_GROUPS) s += (double)x[(size_t)r * ldx + c];
    sh[g][cl] = s;
    __syncthreads();
    double m = 0.0;
#pragma unroll
    for (int k = 0; k < BNS_GROUPS; ++k) m += sh[k][cl];
    m /= (double)R;
    __syncthreads();
    double q = 0.0;
    if (ok)
        for (int r = g; r < R; r += BNS_GROUPS) {
            const double d = (double)x[(size_t)r * ldx + c] - m;
            q += d * d;
        }
    sh[g][cl] = q;
    __syncthreads();
    if (g == 0 && ok) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < BNS_GROUPS; ++k) v += sh[k][cl];
        const float mf = (float)m, vf = (float)(v / (double)R);
        mean_out[c] = mf;
        var_out[c] = vf;
        // (the operations and rounding order of fold_bn_kernel)
        const float sc = gamma[c] * (1.0f / sqrtf(vf + eps));
        float ms = mf * sc;
        asm volatile("" : "+v"(ms));
        bc[0][cl] = sc;
        bc[1][cl] = beta[c] - ms;
    }
    __syncthreads();
    if (!ok) return;
    const float sc = bc[0][cl], sf = bc[1][cl];
    for (int r = g; r < R; r += BNS_GROUPS) {
        y[(size_t)r * ldy + c] = __builtin_fmaf(x[(size_t)r * ldx + c], sc, sf);       // (rows_affine_kernel's contracted x * scale + shift)
    }
}

__global__ __launch_bounds__(64 * BNS_GROUPS) void bn_small_backward_kernel(const float *__restrict__ dh, const float *__restrict__ r, int ld,
                                                                            int R, int C, const float *mean, const float *var,
                                                                            const float *gamma, float eps, int act, float alpha,
                                                                            float *dgamma, float *dbeta, float *__restrict__ dz)
{
    __shared__ double sh[2][BNS_GROUPS][64];
    __shared__ float bc[3][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    double s1 = 0.0, s2 = 0.0;
    if (ok)
        for (int i = g; i < R; i += BNS_GROUPS) {
            const double a = dh[(size_t)i * ld + c];
            s1 += a;
            s2 += a * (double)r[(size_t)i * ld + c];
        }
    sh[0][g][cl] = s1;
    sh[1][g][cl] = s2;
    __syncthreads();
    if (g == 0 && ok) {
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int k = 0; k < BNS_GROUPS; ++k) {
            ta += sh[0][k][cl];
            tb += sh[1][k][cl];
        }
        const double db = (double)(float)ta, sdr = (double)(float)tb;      // (through fp32 as between col_sums and bn_coeffs)
        const double rstd = 1.0 / sqrt((double)var[c] + (double)eps);
        const double dg = rstd * (sdr - (double)mean[c] * db);
        const double gm = gamma[c], N = R;
        dgamma[c] = (float)dg;
        dbeta[c] = (float)db;
        bc[0][cl] = (float)(gm * rstd);
        bc[1][cl] = (float)(-gm * rstd * rstd * dg / N);
        bc[2][cl] = (float)(-gm * rstd * db / N + gm * rstd * rstd * (double)mean[c] * dg / N);
    }
    __syncthreads();
    if (!ok) return;
    const float A = bc[0][cl], B = bc[1][cl], K = bc[2][cl];
    for (int i = g; i < R; i += BNS_GROUPS) {
        const size_t o = (size_t)i * ld + c;
        const float rv = r[o];
        float dr = A * dh[o] + B * rv + K;
        if (act == XV_ACT_RELU) dr = rv > 0.f ? dr : 0.f;
        else if (act == XV_ACT_LRELU) dr = rv > 0.f ? dr : alpha * dr;
        dz[o] = dr;
    }
}

// one wave64 per row of logits
__global__ __launch_bounds__(64) void softmax_ce_kernel(const float *__restrict__ logits, const int *__restrict__ labels, int B,
                                                        int N, float *__restrict__ row_loss, float *__restrict__ row_correct,
                                                        float *__restrict__ dlogits)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *z = logits + (size_t)b * N;
    float mx = -INFINITY;
    int arg = 0;
    for (int j = lane; j < N; j += 64)
        if (z[j] > mx) { mx = z[j]; arg = j; }
    for (int off = 32; off; off >>= 1) {
        const float om = __shfl_xor(mx, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }        // first maximum, like tf.argmax
    }
    double s = 0.0;
    for (int j = lane; j < N; j += 64) s += exp((double)z[j] - (double)mx);
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off, 64);
    const int lab = labels[b];
    const double lse = log(s) + (double)mx;
    if (lane == 0) {
        row_loss[b] = (float)(lse - (double)z[lab]);
        row_correct[b] = arg == lab ? 1.f : 0.f;
    }
    if (dlogits) {
        const double invB = 1.0 / (double)B;
        for (int j = lane; j < N; j += 64) {
            const double pj = exp((double)z[j] - lse);
            dlogits[(size_t)b * N + j] = (float)((pj - (j == lab ? 1.0 : 0.0)) * invB);
        }
    }
}

// ---- additive-margin softmax head (build-defined: BASELINE configs[4] names it, the reference has no margin head) ----
// one wave64 per row: y = x / max(||x||, 1e-12), norm = ||x|| (fp64 accumulate)
__global__ __launch_bounds__(64) void l2_normalize_rows_kernel(const float *__restrict__ x, int ldx, int C, float *__restrict__ y,
                                                               int ldy, float *__restrict__ norm)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    const float *xr = x + (size_t)r * ldx;
    double ss = 0.0;
    for (int j = lane; j < C; j += 64) ss += (double)xr[j] * (double)xr[j];
    for (int off = 32; off; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const double n = sqrt(ss);
    const float inv = (float)(1.0 / fmax(n, 1e-12));
    for (int j = lane; j < C; j += 64) y[(size_t)r * ldy + j] = xr[j] * inv;
    if (lane == 0) norm[r] = (float)n;
}

// dx = (dy - y * <y, dy>) / max(norm, 1e-12)    for y = x / ||x||
__global__ __launch_bounds__(64) void l2_normalize_backward_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                                   const float *__restrict__ norm, int C, float *__restrict__ dx)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    const float *g = dy + (size_t)r * C, *yr = y + (size_t)r * C;
    double dot = 0.0;
    for (int j = lane; j < C; j += 64) dot += (double)g[j] * (double)yr[j];
    for (int off = 32; off; off >>= 1) dot += __shfl_xor(dot, off, 64);
    const double inv = 1.0 / fmax((double)norm[r], 1e-12);
    for (int j = lane; j < C; j += 64) dx[(size_t)r * C + j] = (float)(((double)g[j] - (double)yr[j] * dot) * inv);
}

// z[b, j] = scale * (cos[b, j] - margin * [j == label[b]])   in place
__global__ void am_margin_kernel(float *__restrict__ z, const int *__restrict__ labels, int B, int N, float scale, float margin)
{
    const size_t n = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / N), j = (int)(i - (size_t)b * N);
        z[i] = scale * (z[i] - (j == labels[b] ? margin : 0.f));
    }
}

__global__ void mean2_kernel(const float *a, const float *b, int n, float *out)      // out[0]=mean(a), out[1]=mean(b), in order
{
    if (threadIdx.x || blockIdx.x) return;
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < n; ++i) { sa += a[i]; sb += b[i]; }
    out[0] = (float)(sa / n);
    out[1] = (float)(sb / n);
}

// tf.train.AdamOptimizer._apply_dense:  m,v EMAs; var -= lr_t * m / (sqrt(v) + eps),  lr_t = lr*sqrt(1-b2^t)/(1-b1^t)
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
                            size_t n, float lr_t, float b1, float b2, float eps)
{
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

__global__ void ema_kernel(float *__restrict__ moving, const float *__restrict__ batch, int n, float decay)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) moving[i] = moving[i] * decay + batch[i] * (1.0f - decay);
}

__global__ void axpy_kernel(float *__restrict__ y, const float *__restrict__ x, float a, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}

// sum x^2: stage 1 = one fp64 partial per workgroup (grid-stride float4 reads), stage 2 = the partials added in index
// order by one wave (deterministic).  A single 1024-thread workgroup took 88 us on the 1.5 M-element embed_layer-0/w.
constexpr int SUMSQ_MAX_BLOCKS = 256;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *__restrict__ x, size_t n, double *__restrict__ partial)
{
    __shared__ double sh[4];
    double s = 0.0;
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * 256;
    const bool vec = (((uintptr_t)x) & 15) == 0;
    if (vec) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const float4 v = reinterpret_cast<const float4 *>(x)[i];
            s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    }
    for (size_t i = (vec ? n4 * 4 : 0) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s += (double)x[i] * (double)x[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(64) void sumsq_final_kernel(const double *__restrict__ partial, int nb, float *__restrict__ out)
{
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nb; ++i) s += partial[i];
        out[0] = (float)s;
    }
}

// A minibatch [B, T, F] (fp16 or fp32, as the egs loader hands it over: examples_io.py:165,176 store float16) -> the packed fp32
// rows with gaps the TDNN kernels read: chunk b at rows gap + b (T + gap) .. + T, every other row and the padding columns zero.
// (Round 3: this was host work -- np.zeros, astype, a strided assignment and a pageable copy, 0.7 ms in front of every step with
// the GPU idle; now the raw bytes go up from a pinned buffer and one small kernel converts and scatters.)
template <typename T_IN>
__global__ void pack_minibatch_kernel(const T_IN *__restrict__ src, int B, int T, int F, int gap, int in_dim, long rows,
                                      float *__restrict__ dst)
{
    const size_t n = (size_t)rows * in_dim;
    const int slot = T + gap;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const long r = (long)(i / (unsigned)in_dim);
        const int c = (int)(i - (size_t)r * in_dim);
        float v = 0.f;
        if (r >= gap && c < F) {
            const long q = r - gap;
            const long b = q / slot;
            const int t = (int)(q - b * slot);
            if (b < B && t < T) v = (float)src[((size_t)b * T + t) * F + c];
        }
        dst[i] = v;
    }
}

inline unsigned gs_blocks(size_t n) { return (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096); }

}  // namespace

extern "C" {

// Row splits of a wgrad launch: at least ceil(R/4096), then as many more as still fit into the same number of rounds of
// 512 resident workgroups (256 CUs x 2) -- a 19 k-row minibatch with 112 (tap, tile) pairs ran 560 workgroups = 1.09
// rounds before, i.e. the chip half empty for the second round; 9 splits make it 1008 = 1.97 rounds.  Splits never get
// shorter than 512 rows, and the merge order stays fixed (deterministic).
static long wgrad_splits(int64_t R, int cin, int cout, int K)
{
    if (R <= 4096) return 1;
    const long tiles = (long)K * ((cin + WT - 1) / WT) * ((cout + WT - 1) / WT);
    const long s0 = (R + 4095) / 4096;
    const long rounds = (tiles * s0 + 511) / 512;
    long s = (512 * rounds) / tiles;
    const long smax = (R + 511) / 512;
    if (s > smax) s = smax;
    return s < s0 ? s0 : s;
}

size_t xv_wgrad_workspace_bytes(int64_t R, int cin, int cout, int K)
{
    const long splits = wgrad_splits(R, cin, cout, K);
    return splits <= 1 ? 0 : (size_t)splits * K * cin * (size_t)cout * sizeof(float);
}

int xv_wgrad_f32(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation, float *dw,
                 void *workspace, void *stream)
{
    if (!x || !dz || !dw || R <= 0 || cin <= 0 || cout <= 0 || K <= 0 || !(K & 1) || dilation <= 0)
        return tfail(XV_ERR_BAD_ARG, "wgrad: bad argument");
    WgradParams p{};
    p.x = x; p.dz = dz; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.cout = cout; p.lddz = lddz; p.K = K; p.dil = dilation;
    p.n_ct = (cin + WT - 1) / WT; p.n_ot = (cout + WT - 1) / WT;
    const long splits = wgrad_splits(R, cin, cout, K);
    p.rows_per_split = ((R + splits - 1) / splits + WR - 1) / WR * WR;
    if (splits > 1 && !workspace) return tfail(XV_ERR_BAD_ARG, "wgrad: workspace required");
    p.out = splits > 1 ? (float *)workspace : dw;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)(K * p.n_ct * p.n_ot), (unsigned)splits), dim3(256), 0, st, p);
    int rc = tcheck("wgrad_kernel");
    if (rc) return rc;
    if (splits > 1) {
        const size_t n = (size_t)K * cin * cout;
        hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float *)workspace, n,
                           (int)splits, dw);
        rc = tcheck("sum_splits_kernel");
    }
    return rc;
}

size_t xv_col_sums_workspace_bytes(int64_t R, int c) { return (size_t)((R + CS_ROWS - 1) / CS_ROWS) * 2 * (size_t)c * sizeof(double); }

int xv_col_sums_f32(const float *a, int lda, const float *b, int ldb, int64_t R, int c, float *sum_a, float *sum_ab, void *workspace,
                    void *stream)
{
    if (!a || !sum_a || !workspace || R <= 0 || c <= 0 || (b && !sum_ab)) return tfail(XV_ERR_BAD_ARG, "col_sums: bad argument");
    const int splits = (int)((R + CS_ROWS - 1) / CS_ROWS);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(col_sums_kernel, dim3((c + 255) / 256, splits), dim3(256), 0, st, a, b, (long)R, c, lda, ldb, CS_ROWS,
                       (double *)workspace);
    int rc = tcheck("col_sums_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(col_sums_merge_kernel, dim3((c + 63) / 64), dim3(64 * CSM_GROUPS), 0, st, (const double *)workspace, c, splits, sum_a,
                       b ? sum_ab : nullptr);
    return tcheck("col_sums_merge_kernel");
}

int xv_col_sums_merge_f32(const void *workspace, int64_t R, int c, float *sum_a, float *sum_ab, void *stream)
{
    if (!workspace || !sum_a || R <= 0 || c <= 0) return tfail(XV_ERR_BAD_ARG, "col_sums_merge: bad argument");
    const int splits = (int)((R + CS_ROWS - 1) / CS_ROWS);
    hipLaunchKernelGGL(col_sums_merge_kernel, dim3((c + 63) / 64), dim3(64 * CSM_GROUPS), 0, (hipStream_t)stream, (const double *)workspace, c,
                       splits, sum_a, sum_ab);
    return tcheck("col_sums_merge_kernel");
}

int xv_bn_moments_fold_f32(const void *sums_workspace, int64_t R, int c, float n_frames, const float *gamma, const float *beta, float eps,
                           float *mean, float *var, float *scale, float *shift, void *stream)
{
    if (!sums_workspace || !gamma || !beta || !mean || !var || !scale || !shift || R <= 0 || c <= 0 || !(n_frames > 0.f))
        return tfail(XV_ERR_BAD_ARG, "bn_moments_fold: bad argument");
    const int splits = (int)((R + CS_ROWS - 1) / CS_ROWS);
    hipLaunchKernelGGL(moments_fold_kernel, dim3((c + 63) / 64), dim3(64 * CSM_GROUPS), 0, (hipStream_t)stream, (const double *)sums_workspace,
                       c, splits, n_frames, gamma, beta, eps, mean, var, scale, shift);
    return tcheck("moments_fold_kernel");
}

int xv_merge_moments_f32(const float *chunk_mean_var, const int32_t *row_len, int nchunks, int c, float *mean, float *var, void *stream)
{
    if (!chunk_mean_var || !row_len || !mean || !var || nchunks <= 0 || c <= 0) return tfail(XV_ERR_BAD_ARG, "merge_moments: bad argument");
    hipLaunchKernelGGL(merge_moments_kernel, dim3((c + 63) / 64), dim3(64 * MM_GROUPS), 0, (hipStream_t)stream, chunk_mean_var, row_len,
                       nchunks, c, mean, var);
    return tcheck("merge_moments_kernel");
}

int xv_rows_affine_f32(const float *x, int ldx, int64_t R, int c, const float *scale, const float *shift, const uint8_t *row_valid,
                       float *y, int ldy, void *stream)
{
    return xv_rows_affine_split_f32(x, ldx, R, c, scale, shift, row_valid, y, ldy, nullptr, stream);
}

int xv_rows_affine_split_f32(const float *x, int ldx, int64_t R, int c, const float *scale, const float *shift, const uint8_t *row_valid,
                             float *y, int ldy, void *y_split, void *stream)
{
    if (!x || !y || !scale || !shift || R <= 0 || c <= 0) return tfail(XV_ERR_BAD_ARG, "rows_affine: bad argument");
    const bool vec = !(c & 3) && !(ldx & 3) && !(ldy & 3) && !(((uintptr_t)x | (uintptr_t)y) & 15);
    const dim3 grid((unsigned)((c + 1023) / 1024), (unsigned)(R < 4096 ? R : 4096));
    if (y_split) {
        if (!vec || (c & 31) || (((uintptr_t)y_split) & 15))
            return tfail(XV_ERR_UNSUPPORTED, "rows_affine: the split copy needs c % 32 == 0 and 16-byte aligned fp32 rows");
        hipLaunchKernelGGL((rows_affine_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, x, (long)R, c, ldx, scale, shift, row_valid,
                           y, ldy, (uint8_t *)y_split);
        return tcheck("rows_affine_kernel");
    }
    if (vec) hipLaunchKernelGGL(rows_affine_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)R, c, ldx, scale, shift, row_valid, y, ldy);
    else hipLaunchKernelGGL(rows_affine_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, (long)R, c, ldx, scale, shift, row_valid, y, ldy);
    return tcheck("rows_affine_kernel");
}

int xv_bn_act_backward_f32(const float *dh, const float *r, int ld, int64_t R, int c, const float *sum_dh, const float *sum_dh_r,
                           const float *mean, const float *var, const float *gamma, float eps, float n_frames, int act_kind,
                           float act_alpha, const uint8_t *row_valid, float *dgamma, float *dbeta, float *coef_ws, float *dz,
                           void *stream)
{
    return xv_bn_act_backward_split_f32(dh, r, ld, R, c, sum_dh, sum_dh_r, mean, var, gamma, eps, n_frames, act_kind, act_alpha, row_valid,
                                        dgamma, dbeta, coef_ws, dz, nullptr, stream);
}

static int bn_act_backward_tail(const float *dh, const float *r, int ld, int64_t R, int c, int act_kind, float act_alpha,
                                const uint8_t *row_valid, float *coef_ws, float *dz, void *dz_split, hipStream_t st)
{
    const bool vec = !(c & 3) && !(ld & 3) && !(((uintptr_t)dh | (uintptr_t)r | (uintptr_t)dz | (uintptr_t)coef_ws) & 15);
    if (dz_split && (!vec || (c & 31) || (((uintptr_t)dz_split) & 15)))
        return tfail(XV_ERR_UNSUPPORTED, "bn_act_backward: the split copy needs c % 32 == 0 and 16-byte aligned fp32 rows");
    if (vec) {
        const dim3 grid((unsigned)((c + 1023) / 1024), (unsigned)(R < 4096 ? R : 4096));
        if (dz_split)
            hipLaunchKernelGGL(bn_act_backward_vec_kernel<true>, grid, dim3(256), 0, st, dh, r, (long)R, c, ld, coef_ws, coef_ws + c,
                               coef_ws + 2 * c, act_kind, act_alpha, row_valid, dz, (uint8_t *)dz_split);
        else
            hipLaunchKernelGGL(bn_act_backward_vec_kernel<false>, grid, dim3(256), 0, st, dh, r, (long)R, c, ld, coef_ws, coef_ws + c,
                               coef_ws + 2 * c, act_kind, act_alpha, row_valid, dz, (uint8_t *)nullptr);
        return tcheck("bn_act_backward_vec_kernel");
    }
    hipLaunchKernelGGL(bn_act_backward_kernel, dim3(gs_blocks((size_t)R * c)), dim3(256), 0, st, dh, r, (long)R, c, ld, coef_ws,
                       coef_ws + c, coef_ws + 2 * c, act_kind, act_alpha, row_valid, dz);
    return tcheck("bn_act_backward_kernel");
}

int xv_bn_act_backward_split_f32(const float *dh, const float *r, int ld, int64_t R, int c, const float *sum_dh, const float *sum_dh_r,
                                 const float *mean, const float *var, const float *gamma, float eps, float n_frames, int act_kind,
                                 float act_alpha, const uint8_t *row_valid, float *dgamma, float *dbeta, float *coef_ws, float *dz,
                                 void *dz_split, void *stream)
{
    if (!dh || !r || !sum_dh || !sum_dh_r || !mean || !var || !gamma || !dgamma || !dbeta || !coef_ws || !dz || R <= 0 || c <= 0)
        return tfail(XV_ERR_BAD_ARG, "bn_act_backward: bad argument");
    if (act_kind == XV_ACT_PRELU) return tfail(XV_ERR_UNSUPPORTED, "bn_act_backward: PReLU training is not implemented");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_coeffs_kernel, dim3((c + 255) / 256), dim3(256), 0, st, sum_dh, sum_dh_r, mean, var, gamma, eps, n_frames, c,
                       dgamma, dbeta, coef_ws, coef_ws + c, coef_ws + 2 * c);
    int rc = tcheck("bn_coeffs_kernel");
    if (rc) return rc;
    return bn_act_backward_tail(dh, r, ld, R, c, act_kind, act_alpha, row_valid, coef_ws, dz, dz_split, st);
}

int xv_bn_act_backward_parts_f32(const float *dh, const float *r, int ld, int64_t R, int c, const void *sums_workspace, const float *mean,
                                 const float *var, const float *gamma, float eps, float n_frames, int act_kind, float act_alpha,
                                 const uint8_t *row_valid, float *dgamma, float *dbeta, float *coef_ws, float *dz, void *dz_split,
                                 void *stream)
{
    if (!dh || !r || !sums_workspace || !mean || !var || !gamma || !dgamma || !dbeta || !coef_ws || !dz || R <= 0 || c <= 0)
        return tfail(XV_ERR_BAD_ARG, "bn_act_backward_parts: bad argument");
    if (act_kind == XV_ACT_PRELU) return tfail(XV_ERR_UNSUPPORTED, "bn_act_backward: PReLU training is not implemented");
    hipStream_t st = (hipStream_t)stream;
    const int splits = (int)((R + CS_ROWS - 1) / CS_ROWS);
    hipLaunchKernelGGL(col_sums_merge_coeffs_kernel, dim3((c + 63) / 64), dim3(64 * CSM_GROUPS), 0, st, (const double *)sums_workspace, c,
                       splits, mean, var, gamma, eps, n_frames, dgamma, dbeta, coef_ws, coef_ws + c, coef_ws + 2 * c);
    int rc = tcheck("col_sums_merge_coeffs_kernel");
    if (rc) return rc;
    return bn_act_backward_tail(dh, r, ld, R, c, act_kind, act_alpha, row_valid, coef_ws, dz, dz_split, st);
}

int xv_bn_small_forward_f32(const float *x, int ldx, int nrows, int c, const float *gamma, const float *beta, float eps, float *mean,
                            float *var, float *y, int ldy, void *stream)
{
    if (!x || !gamma || !beta || !mean || !var || !y || nrows <= 0 || nrows > BNS_MAX_ROWS || c <= 0 || ldx < c || ldy < c)
        return tfail(XV_ERR_BAD_ARG, "bn_small_forward: bad argument (1 .. 1024 rows)");
    hipLaunchKernelGGL(bn_small_forward_kernel, dim3((c + 63) / 64), dim3(64 * BNS_GROUPS), 0, (hipStream_t)stream, x, ldx, nrows, c, gamma, beta,
                       eps, mean, var, y, ldy);
    return tcheck("bn_small_forward_kernel");
}

int xv_bn_small_backward_f32(const float *dh, const float *r, int ld, int nrows, int c, const float *mean, const float *var,
                             const float *gamma, float eps, int act_kind, float act_alpha, float *dgamma, float *dbeta, float *dz,
                             void *stream)
{
    if (!dh || !r || !mean || !var || !gamma || !dgamma || !dbeta || !dz || nrows <= 0 || nrows > BNS_MAX_ROWS || c <= 0 || ld < c)
        return tfail(XV_ERR_BAD_ARG, "bn_small_backward: bad argument (1 .. 1024 rows)");
    if (act_kind == XV_ACT_PRELU) return tfail(XV_ERR_UNSUPPORTED, "bn_small_backward: PReLU training is not implemented");
    hipLaunchKernelGGL(bn_small_backward_kernel, dim3((c + 63) / 64), dim3(64 * BNS_GROUPS), 0, (hipStream_t)stream, dh, r, ld, nrows, c, mean,
                       var, gamma, eps, act_kind, act_alpha, dgamma, dbeta, dz);
    return tcheck("bn_small_backward_kernel");
}

int xv_pool_bn_act_backward_f32(const float *h, const float *r, int ld, int c, const int32_t *row_start, const int32_t *row_len,
                                int nchunks, int64_t R, const float *pooled, const float *dpooled, const float *chunk_moments,
                                const float *mean, const float *var, const float *gamma, float eps, float n_frames, int act_kind,
                                float act_alpha, float *dgamma, float *dbeta, float *coef_ws, float *dz, void *dz_split, void *stream)
{
    if (!h || !r || !row_start || !row_len || !pooled || !dpooled || !chunk_moments || !mean || !var || !gamma || !dgamma || !dbeta ||
        !coef_ws || !dz || nchunks <= 0 || nchunks > 65535 || R <= 0 || c <= 0)
        return tfail(XV_ERR_BAD_ARG, "pool_bn_act_backward: bad argument");
    if (act_kind == XV_ACT_PRELU) return tfail(XV_ERR_UNSUPPORTED, "pool_bn_act_backward: PReLU training is not implemented");
    if ((c & 3) || (ld & 3) || (((uintptr_t)h | (uintptr_t)r | (uintptr_t)dz | (uintptr_t)coef_ws | (uintptr_t)pooled | (uintptr_t)dpooled) & 15))
        return tfail(XV_ERR_UNSUPPORTED, "pool_bn_act_backward: needs c % 4 == 0 and 16-byte aligned rows");
    if (dz_split && ((c & 31) || (((uintptr_t)dz_split) & 15)))
        return tfail(XV_ERR_UNSUPPORTED, "pool_bn_act_backward: the split copy needs c % 32 == 0");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pool_bn_coeffs_kernel, dim3((c + 63) / 64), dim3(64 * CSM_GROUPS), 0, st, pooled, dpooled, chunk_moments, nchunks, mean, var, gamma,
                       eps, n_frames, c, dgamma, dbeta, coef_ws, coef_ws + c, coef_ws + 2 * c);
    int rc = tcheck("pool_bn_coeffs_kernel");
    if (rc) return rc;
    const int zs = nchunks >= 2048 ? 1 : (2048 + nchunks - 1) / nchunks < 16 ? (2048 + nchunks - 1) / nchunks : 16;
    const dim3 grid((unsigned)((c + 511) / 512), (unsigned)nchunks, (unsigned)zs);
    if (dz_split)
        hipLaunchKernelGGL(pool_bn_act_backward_kernel<true>, grid, dim3(128), 0, st, h, r, ld, c, row_start, row_len, nchunks, (long)R, pooled,
                           dpooled, coef_ws, coef_ws + c, coef_ws + 2 * c, act_kind, act_alpha, dz, (uint8_t *)dz_split);
    else
        hipLaunchKernelGGL(pool_bn_act_backward_kernel<false>, grid, dim3(128), 0, st, h, r, ld, c, row_start, row_len, nchunks, (long)R, pooled,
                           dpooled, coef_ws, coef_ws + c, coef_ws + 2 * c, act_kind, act_alpha, dz, (uint8_t *)nullptr);
    return tcheck("pool_bn_act_backward_kernel");
}

int xv_pool_backward_f32(const float *h, int ldh, int c, const int32_t *row_start, const int32_t *row_len, int nchunks, int64_t R,
                         const float *pooled, const float *dpooled, float *dh, void *stream)
{
    if (!h || !row_start || !row_len || !pooled || !dpooled || !dh || nchunks <= 0 || c <= 0) return tfail(XV_ERR_BAD_ARG, "pool_backward: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dh, 0, (size_t)R * ldh * sizeof(float), st);      // gap rows carry no gradient
    if (e != hipSuccess) return tfail((int)e, "pool_backward: memset failed");
    for (int b0 = 0; b0 < nchunks; b0 += 65535) {
        const int nb = nchunks - b0 < 65535 ? nchunks - b0 : 65535;
        hipLaunchKernelGGL(pool_backward_kernel, dim3(64, nb), dim3(256), 0, st, h, ldh, c, row_start + b0, row_len + b0,
                           pooled + (size_t)b0 * 2 * c, dpooled + (size_t)b0 * 2 * c, dh);
        int rc = tcheck("pool_backward_kernel");
        if (rc) return rc;
    }
    return 0;
}

int xv_softmax_ce_f32(const float *logits, const int32_t *labels, int nrows, int nclasses, float *loss_acc, float *row_ws,
                      float *dlogits, void *stream)
{
    if (!logits || !labels || !loss_acc || !row_ws || nrows <= 0 || nclasses <= 0) return tfail(XV_ERR_BAD_ARG, "softmax_ce: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(nrows), dim3(64), 0, st, logits, labels, nrows, nclasses, row_ws, row_ws + nrows, dlogits);
    int rc = tcheck("softmax_ce_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(mean2_kernel, dim3(1), dim3(64), 0, st, (const float *)row_ws, (const float *)(row_ws + nrows), nrows, loss_acc);
    return tcheck("mean2_kernel");
}

int xv_adam_f32(float *param, const float *grad, float *m, float *v, int64_t n, float lr_t, float beta1, float beta2, float eps,
                void *stream)
{
    if (!param || !grad || !m || !v || n <= 0) return tfail(XV_ERR_BAD_ARG, "adam: bad argument");
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, (size_t)n,
                       lr_t, beta1, beta2, eps);
    return tcheck("adam_kernel");
}

int xv_axpy_f32(float *y, const float *x, float a, int64_t n, void *stream)
{
    if (!y || !x || n <= 0) return tfail(XV_ERR_BAD_ARG, "axpy: bad argument");
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, x, a, (size_t)n);
    return tcheck("axpy_kernel");
}

static int sumsq_blocks(int64_t n)
{
    const int64_t b = (n + 8191) / 8192;
    return (int)(b < 1 ? 1 : (b > SUMSQ_MAX_BLOCKS ? SUMSQ_MAX_BLOCKS : b));
}

size_t xv_sumsq_workspace_bytes(int64_t n) { return (size_t)sumsq_blocks(n) * sizeof(double); }

int xv_sumsq_f32(const float *x, int64_t n, float *out, void *workspace, void *stream)
{
    if (!x || !out || n <= 0 || !workspace || (((uintptr_t)workspace) & 7)) return tfail(XV_ERR_BAD_ARG, "sumsq: bad argument");
    const int nb = sumsq_blocks(n);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, (double *)workspace);
    int rc = tcheck("sumsq_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double *)workspace, nb, out);
    return tcheck("sumsq_final_kernel");
}

// the three index arrays of a minibatch of B equal chunks of T frames with `gap` zero rows in front of, between and behind them
__global__ void minibatch_layout_kernel(int B, int T, int gap, long rows, int *__restrict__ row_start, int *__restrict__ row_len,
                                        uint8_t *__restrict__ row_valid)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = T + gap;
    if (i < B) {
        row_start[i] = gap + (int)i * slot;
        row_len[i] = T;
    }
    if (i < rows) {
        const long j = i - gap;
        row_valid[i] = (j >= 0 && j % slot < T && j / slot < B) ? 1 : 0;
    }
}

int xv_minibatch_layout(int B, int T, int gap, int64_t rows, int32_t *row_start, int32_t *row_len, uint8_t *row_valid, void *stream)
{
    if (!row_start || !row_len || !row_valid || B <= 0 || T <= 0 || gap < 0 || rows < (int64_t)gap + (int64_t)B * (T + gap) || rows < B)
        return tfail(XV_ERR_BAD_ARG, "minibatch_layout: bad argument");
    hipLaunchKernelGGL(minibatch_layout_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, T, gap, (long)rows,
                       row_start, row_len, row_valid);
    return tcheck("minibatch_layout_kernel");
}

int xv_pack_minibatch_f32(const void *src, int src_is_f16, int B, int T, int F, int gap, int in_dim, float *dst, int64_t rows,
                          void *stream)
{
    if (!src || !dst || B <= 0 || T <= 0 || F <= 0 || gap < 0 || in_dim < F || rows < (int64_t)gap + (int64_t)B * (T + gap))
        return tfail(XV_ERR_BAD_ARG, "pack_minibatch: bad argument");
    const size_t n = (size_t)rows * in_dim;
    const dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 4096));
    if (src_is_f16)
        hipLaunchKernelGGL(pack_minibatch_kernel<_Float16>, grid, dim3(256), 0, (hipStream_t)stream, (const _Float16 *)src, B, T, F, gap,
                           in_dim, (long)rows, dst);
    else
        hipLaunchKernelGGL(pack_minibatch_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float *)src, B, T, F, gap, in_dim,
                           (long)rows, dst);
    return tcheck("pack_minibatch_kernel");
}

int xv_dropout_f32(float *x, int ldx, int64_t R, int c, uint64_t seed, float keep_prob, void *stream)
{
    if (R <= 0 || c <= 0) return 0;
    if (!x || ldx < c) return tfail(XV_ERR_BAD_ARG, "dropout: bad argument");
    if (!(keep_prob > 0.f) || keep_prob > 1.f) return tfail(XV_ERR_BAD_ARG, "dropout: keep_prob must be in (0, 1]");
    if (keep_prob == 1.f) return 0;
    const double thr = (double)keep_prob * 4294967296.0;
    const size_t n = (size_t)R * c;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0, (hipStream_t)stream, x,
                       (long)R, c, ldx, seed, (uint32_t)std::min(thr, 4294967295.0), 1.f / keep_prob);
    return tcheck("dropout_kernel");
}

int xv_prelu_backward_f32(float *dr, float *z, int ld, int64_t R, int c, const float *alpha, void *stream)
{
    if (R <= 0 || c <= 0) return 0;
    if (!dr || !z || !alpha || ld < c) return tfail(XV_ERR_BAD_ARG, "prelu_backward: bad argument");
    const size_t n = (size_t)R * c;
    hipLaunchKernelGGL(prelu_backward_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0,
                       (hipStream_t)stream, dr, z, (long)R, c, ld, alpha);
    return tcheck("prelu_backward_kernel");
}

int xv_l2_normalize_rows_f32(const float *x, int ldx, int nrows, int c, float *y, int ldy, float *norm, void *stream)
{
    if (nrows <= 0) return 0;
    if (!x || !y || !norm || c <= 0 || ldx < c || ldy < c) return tfail(XV_ERR_BAD_ARG, "l2_normalize_rows: bad argument");
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3(nrows), dim3(64), 0, (hipStream_t)stream, x, ldx, c, y, ldy, norm);
    return tcheck("l2_normalize_rows_kernel");
}

int xv_l2_normalize_backward_f32(const float *dy, const float *y, const float *norm, int nrows, int c, float *dx, void *stream)
{
    if (nrows <= 0) return 0;
    if (!dy || !y || !norm || !dx || c <= 0) return tfail(XV_ERR_BAD_ARG, "l2_normalize_backward: bad argument");
    hipLaunchKernelGGL(l2_normalize_backward_kernel, dim3(nrows), dim3(64), 0, (hipStream_t)stream, dy, y, norm, c, dx);
    return tcheck("l2_normalize_backward_kernel");
}

int xv_am_margin_f32(float *cosines, const int32_t *labels, int nrows, int nclasses, float scale, float margin, void *stream)
{
    if (nrows <= 0 || nclasses <= 0) return 0;
    if (!cosines || !labels) return tfail(XV_ERR_BAD_ARG, "am_margin: bad argument");
    const size_t n = (size_t)nrows * nclasses;
    hipLaunchKernelGGL(am_margin_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       cosines, labels, nrows, nclasses, scale, margin);
    return tcheck("am_margin_kernel");
}

int xv_ema_f32(float *moving, const float *batch, int n, float decay, void *stream)
{
    if (!moving || !batch || n <= 0) return tfail(XV_ERR_BAD_ARG, "ema: bad argument");
    hipLaunchKernelGGL(ema_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, moving, batch, n, decay);
    return tcheck("ema_kernel");
}

}  // extern "C"
