// xv_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of libxvector_hip.so.
//
// Hot path of BUTSpeechFIT/x-vector-kaldi-tf's extraction (local/tf/models.py:50-94 evaluated by
// local/tf/models.py:414), written for CDNA4 from scratch:
//   tdnn_gemm_kernel   implicit-im2col GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), 128x128 tile,
//                      4 wave64 per workgroup (2x2 waves, 2x2 MFMA tiles each), the dilated temporal
//                      context window staged ONCE per channel slab in LDS and re-used by all K taps,
//                      double-buffered LDS with register prefetch, fused bias/act/BN/gap-mask epilogue
//   stats_pool_kernel  HBM-bound mean/std reduction: one wave64 per (chunk, split, 64 channels),
//                      16 B/lane loads, blocked two-pass + Chan merges, wave shuffle combine
//   chunk_average, pack_weights, fold_bn   small helpers
// See include/xvector_hip.h for the ABI contract and DESIGN.md for the layout / roofline notes.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "xvector_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int hip_fail(hipError_t e, const char *where)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}

// ------------------------------------------------------------------------------------------------
// TDNN / FC GEMM
// ------------------------------------------------------------------------------------------------
constexpr int BM = 128;        // frames per workgroup tile
constexpr int BN = 128;        // output channels per workgroup tile
constexpr int BK = 32;         // input channels per LDS stage
constexpr int LDS_LD = 36;     // padded LDS row (floats): 144 B keeps ds_read_b128 conflict-free
constexpr int MAX_SPAN = 8;    // (K-1)*dilation limit -> A tile holds BM+8 rows
constexpr int A_ROWS = BM + MAX_SPAN;
constexpr int NT = 256;

struct GemmParams {
    const float *x;
    long R;
    int cin, ldx;
    const float *wp;
    int kred;
    const float *bias, *scale, *shift;
    int act;
    const float *alpha;
    int K, dil, cout;
    const uint8_t *valid;
    float *y;
    int ldy;
    float *ypre;
    float *blk;         // POOL epilogue (no y / ypre): per-8-row-block (mean, M2) planes [ceil(R/8)][2][cout], as Gemm3Params::blk
    int n_mt, n_nt;
    int vec_out;        // outputs take 16-byte stores: cout % 4 == 0, ldy % 4 == 0, y / ypre / per-column parameters 16-byte aligned
    int k_splits;       // split-K (xv_fc_splitk_f32, register-staged kernel only): the slabs are dealt to k_splits groups of workgroups, group ks
    float *part;        //   writes its raw partial sums to part + ks * part_stride ([R, cout] fp32 rows); a second kernel adds them up in order
    long part_stride;
    int lead;           // "rows" form (xv_tdnn_layer_rows_f32): K = 1 over rows that OVERLAP -- virtual row r = the cin floats from
                        // x + (r - lead) * ldx on, cin > ldx; reads are bounded by the END OF THE BUFFER (R * ldx floats), not by the row
};

constexpr size_t GEMM_LDS_BYTES = (size_t)(2 * A_ROWS * LDS_LD + 2 * BN * LDS_LD) * sizeof(float) + BM;

__device__ __forceinline__ float apply_act(float z, int act, float a)
{
    switch (act) {
    case XV_ACT_RELU: return fmaxf(z, 0.0f);
    case XV_ACT_LRELU: return z > 0.0f ? z : a * z;
    case XV_ACT_PRELU: return fmaxf(z, 0.0f) + a * fminf(z, 0.0f);
    default: return z;
    }
}

// The same with the kind as a compile-time constant: identical expressions, hence identical bits -- but no switch per ELEMENT
// (the row-wise epilogue below used to evaluate one: ~6 scalar compares / branches around every 4 VALU instructions, nothing
// packed; beside an fp32 MFMA a co-resident workgroup pays for every one of them, cf. csrc/xv_toom.hip)
template <int ACT>
__device__ __forceinline__ float act_t(float z, float a)
{
    if constexpr (ACT == XV_ACT_RELU) return fmaxf(z, 0.0f);
    else if constexpr (ACT == XV_ACT_LRELU) return z > 0.0f ? z : a * z;
    else if constexpr (ACT == XV_ACT_PRELU) return fmaxf(z, 0.0f) + a * fminf(z, 0.0f);
    else return z;
}

// Fused epilogue shared by the fp32 and the bf16x3 GEMM kernels.
// D layout of a 32x32 MFMA tile: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
template <int WROWS>      // rows of the tile one wave owns: 64 (two 32-row MFMA blocks) or 32 (one)
__device__ __forceinline__ void gemm_epilogue(const GemmParams &p, const uint8_t *Ms, long m0, int n0, int wr, int wc,
                                              int lane, const f32x16 &acc00, const f32x16 &acc01, const f32x16 &acc10,
                                              const f32x16 &acc11)
{
    const int colb = n0 + wc * 64 + (lane & 31);
    const int rowb = wr * WROWS + 4 * (lane >> 5);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int gc = colb + cb * 32;
        if (gc >= p.cout) continue;
        const float bias = p.bias ? p.bias[gc] : 0.f;
        const float sc = p.scale ? p.scale[gc] : 1.f;
        const float sh = p.shift ? p.shift[gc] : 0.f;
        const float al = (p.act == XV_ACT_LRELU) ? p.alpha[0] : (p.act == XV_ACT_PRELU ? p.alpha[gc] : 0.f);
#pragma unroll
        for (int rb = 0; rb < WROWS / 32; ++rb) {
            const f32x16 &a = (rb == 0) ? (cb == 0 ? acc00 : acc01) : (cb == 0 ? acc10 : acc11);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = rowb + rb * 32 + (reg & 3) + 8 * (reg >> 2);
                const long gr = m0 + lr;
                if (gr >= p.R) continue;
                const float z = a[reg] + bias;
                if (p.ypre) p.ypre[(size_t)gr * p.ldy + gc] = z;
                if (p.y) {
                    float v = apply_act(z, p.act, al) * sc + sh;
                    if (!Ms[lr]) v = 0.f;
                    __builtin_nontemporal_store(v, &p.y[(size_t)gr * p.ldy + gc]);   // streamed once: no L2 write-allocate
                }
            }
        }
    }
}

// The same epilogue with the output rows written as they lie in memory (round 3): the accumulators go through an fp32 tile in LDS
// (the operand buffers are dead by then: 128 x 132 floats of their 76 KB), and a thread then owns 4 consecutive columns of a row
// -- 32 lanes cover 512 contiguous bytes of an output row with one 16-byte store each, where the register-direct form above
// writes two 128-byte segments per 4-byte store instruction (a timing-only build without those stores ran the exact-fp32 step
// 3.8 % faster).  Element for element the same arithmetic: results are bit-identical to gemm_epilogue.
template <int BMT, int ACT>
__device__ __forceinline__ void gemm_epilogue_rows_body(const GemmParams &p, const float *T, const uint8_t *Ms, long m0, int n0, int tid)
{
    constexpr int TLD = BN + 4;
    const int cg = tid & 31, rp = tid >> 5;              // 4 columns; rows rp, rp + 8, ...
    const int gc = n0 + cg * 4;
    if (gc >= p.cout) return;                             // (cout % 4 == 0: a column group is inside or outside as a whole)
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = p.bias ? *reinterpret_cast<const f32x4 *>(p.bias + gc) : zero;
    const f32x4 sc = p.scale ? *reinterpret_cast<const f32x4 *>(p.scale + gc) : one;
    const f32x4 sh = p.shift ? *reinterpret_cast<const f32x4 *>(p.shift + gc) : zero;
    f32x4 al = zero;
    if constexpr (ACT == XV_ACT_LRELU) al = (f32x4){p.alpha[0], p.alpha[0], p.alpha[0], p.alpha[0]};
    else if constexpr (ACT == XV_ACT_PRELU) al = *reinterpret_cast<const f32x4 *>(p.alpha + gc);
    if (p.blk) {
        // POOL: the layer output is not stored; every 8-row block of the tile is reduced to per-channel (mean, M2) of its valid
        // rows, shifted by the block's first row (as the POOL epilogue of tdnn_gemm_bf16x3_kernel: same planes, same finalize).
        // Thread = (4 channels, blocks rp and rp + 8).  Explicit fma: the two unrolled instances must round alike, a block's
        // statistics may not depend on where it sits in the tile.
#pragma unroll
        for (int bb = 0; bb < (BMT / 8 + 7) / 8; ++bb) {
            const int blk = rp + 8 * bb;
            if (blk >= BMT / 8 || m0 + blk * 8 >= p.R) continue;          // (a 32-row pass has four blocks: rp < 4)
            f32x4 tv[8];
            float keep[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                tv[j] = *reinterpret_cast<const f32x4 *>(T + (blk * 8 + j) * TLD + cg * 4);
                keep[j] = Ms[blk * 8 + j] ? 1.f : 0.f;
            }
            f32x4 v0 = zero, s1 = zero, s2 = zero;
            float n = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                n += keep[j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = __builtin_fmaf(act_t<ACT>(tv[j][i] + bias[i], al[i]), sc[i], sh[i]);
                    if (j == 0) v0[i] = v;
                    else {
                        const float d = keep[j] != 0.f ? v - v0[i] : 0.f;       // (a select: a row past R may hold anything)
                        s1[i] += d;
                        s2[i] = __builtin_fmaf(d, d, s2[i]);
                    }
                }
            }
            // row 0 of a block is valid whenever any row is (chunks start on block boundaries, gaps follow the frames)
            const float rn = n > 0.f ? 1.f / n : 0.f;
            f32x4 mean, m2;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = s1[i] * rn;
                mean[i] = n > 0.f ? v0[i] + t : 0.f;
                m2[i] = fmaxf(__builtin_fmaf(-t, s1[i], s2[i]), 0.f);
            }
            float *o = p.blk + ((size_t)((m0 >> 3) + blk) * 2) * p.cout + gc;
            *reinterpret_cast<f32x4 *>(o) = mean;
            *reinterpret_cast<f32x4 *>(o + p.cout) = m2;
        }
        return;
    }
#pragma unroll 4
    for (int j = 0; j < BMT / 8; ++j) {
        const int lr = rp + 8 * j;
        const long gr = m0 + lr;
        if (gr >= p.R) continue;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(T + lr * TLD + cg * 4);
        f32x4 z, v;
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = a[i] + bias[i];
        if (p.ypre) *reinterpret_cast<f32x4 *>(p.ypre + (size_t)gr * p.ldy + gc) = z;
        if (p.y) {
            const bool keep = Ms[lr] != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = act_t<ACT>(z[i], al[i]) * sc[i] + sh[i];
                v[i] = keep ? t : 0.f;
            }
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p.y + (size_t)gr * p.ldy + gc));   // streamed once: no L2 write-allocate
        }
    }
}

template <int BMT>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmParams &p, float *T, const uint8_t *Ms, long m0, int n0, int wr, int wc,
                                                   int tid, const f32x16 &acc00, const f32x16 &acc01, const f32x16 &acc10,
                                                   const f32x16 &acc11)
{
    constexpr int WROWS = BMT / 2;
    constexpr int TLD = BN + 4;
    const int lane = tid & 63;
    {
        const int col = wc * 64 + (lane & 31);
        const int rowb = wr * WROWS + 4 * (lane >> 5);
#pragma unroll
        for (int rb = 0; rb < WROWS / 32; ++rb)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = rowb + rb * 32 + (reg & 3) + 8 * (reg >> 2);
                T[lr * TLD + col] = rb == 0 ? acc00[reg] : acc10[reg];
                T[lr * TLD + col + 32] = rb == 0 ? acc01[reg] : acc11[reg];
            }
    }
    __syncthreads();
    switch (p.act) {                                      // (uniform: one switch per thread, not one per element)
    case XV_ACT_RELU: gemm_epilogue_rows_body<BMT, XV_ACT_RELU>(p, T, Ms, m0, n0, tid); break;
    case XV_ACT_LRELU: gemm_epilogue_rows_body<BMT, XV_ACT_LRELU>(p, T, Ms, m0, n0, tid); break;
    case XV_ACT_PRELU: gemm_epilogue_rows_body<BMT, XV_ACT_PRELU>(p, T, Ms, m0, n0, tid); break;
    default: gemm_epilogue_rows_body<BMT, XV_ACT_NONE>(p, T, Ms, m0, n0, tid); break;
    }
}

// VEC: Cin % 4 == 0 and 16-B aligned rows -> dwordx4 staging loads; otherwise dword loads (the
// 23-dim MFCC input layer).
// BMT: rows per workgroup tile, 128 or 64.  The 64-row form (each wave one 32-row MFMA block x two column blocks) exists
// for small problems -- a training minibatch of 19 k rows makes 608 128-row tiles on Cout = 512, i.e. 1.19 rounds of the
// 512 resident workgroups; 1216 half-size tiles are 2.4 rounds of half the length.
template <bool VEC, int BMT>
__global__ __launch_bounds__(NT, 2) void tdnn_gemm_kernel(const GemmParams p)
{
    constexpr int WROWS = BMT / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                                 // [2][A_ROWS][LDS_LD]
    float *Bs = smem + 2 * A_ROWS * LDS_LD;           // [2][BN][LDS_LD]
    uint8_t *Ms = (uint8_t *)(Bs + 2 * BN * LDS_LD);  // [BM] row-valid bytes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: hardware places block b on XCD b%8; give each XCD a contiguous run of
    // logical tiles so that the n_nt tiles sharing one A panel sit on one L2 (bijective form).
    const int nwg = p.n_mt * p.n_nt;
    int bid = blockIdx.x, ks = 0;
    if (p.k_splits > 1) {                              // split-K: blocks [ks * nwg, (ks + 1) * nwg) work on slab group ks
        ks = bid / nwg;
        bid -= ks * nwg;
    }
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = wg / p.n_nt, nt = wg - mt * p.n_nt;
    const long m0 = (long)mt * BMT;
    const int n0 = nt * BN;

    const int span = (p.K - 1) * p.dil;
    const int left = (span >> 1) + p.lead;
    const int rowsA = BMT + span;
    const int all_chunks = (p.cin + BK - 1) / BK;
    const int per_split = p.k_splits > 1 ? (all_chunks + p.k_splits - 1) / p.k_splits : all_chunks;
    const int c_lo = ks * per_split;
    const int n_chunks = (c_lo + per_split < all_chunks ? c_lo + per_split : all_chunks) - c_lo;      // (the launcher leaves no group empty)
    const int n_stages = n_chunks * p.K;
    const long flat_end = p.R * p.ldx;                 // (rows form: what the DMA-fed kernel's buffer descriptor checks)

    if (tid < BMT) {
        const long gr = m0 + tid;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    }

    constexpr int B_REGS = VEC ? 4 : 16;
    constexpr int A_REGS = VEC ? ((BMT + MAX_SPAN) * 8 + NT - 1) / NT : ((BMT + MAX_SPAN) * 32 + NT - 1) / NT;
    typedef typename std::conditional<VEC, f32x4, float>::type stage_t;
    stage_t breg[B_REGS];
    stage_t areg[A_REGS];

    auto load_b = [&](int chunk, int tap) {
        const int c0 = chunk * BK;
#pragma unroll
        for (int j = 0; j < B_REGS; ++j) {
            const int f = tid + NT * j;
            if constexpr (VEC) {
                const int col = f >> 3, qq = f & 7;
                const int gcol = n0 + col, c = c0 + qq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gcol < p.cout && c < p.cin)
                    v = *reinterpret_cast<const f32x4 *>(p.wp + (size_t)gcol * p.kred + (size_t)tap * p.cin + c);
                breg[j] = v;
            } else {
                const int col = f >> 5, cc = f & 31;
                const int gcol = n0 + col, c = c0 + cc;
                float v = 0.f;
                if (gcol < p.cout && c < p.cin) v = p.wp[(size_t)gcol * p.kred + (size_t)tap * p.cin + c];
                breg[j] = v;
            }
        }
    };
    auto load_a = [&](int chunk) {
        const int c0 = chunk * BK;
#pragma unroll
        for (int j = 0; j < A_REGS; ++j) {
            const int f = tid + NT * j;
            if constexpr (VEC) {
                const int lr = f >> 3, qq = f & 7;
                const long gr = m0 - left + lr;
                const int c = c0 + qq * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (lr < rowsA && gr >= 0 && c < p.cin && (p.lead ? gr * p.ldx + c + 4 <= flat_end : gr < p.R))
                    v = *reinterpret_cast<const f32x4 *>(p.x + (size_t)gr * p.ldx + c);
                areg[j] = v;
            } else {
                const int lr = f >> 5, cc = f & 31;
                const long gr = m0 - left + lr;
                const int c = c0 + cc;
                float v = 0.f;
                if (lr < rowsA && gr >= 0 && gr < p.R && c < p.cin) v = p.x[(size_t)gr * p.ldx + c];
                areg[j] = v;
            }
        }
    };
    auto store_b = [&](int buf) {
        float *dst = Bs + buf * (BN * LDS_LD);
#pragma unroll
        for (int j = 0; j < B_REGS; ++j) {
            const int f = tid + NT * j;
            if constexpr (VEC)
                *reinterpret_cast<f32x4 *>(dst + (f >> 3) * LDS_LD + (f & 7) * 4) = breg[j];
            else
                dst[(f >> 5) * LDS_LD + (f & 31)] = breg[j];
        }
    };
    auto store_a = [&](int buf) {
        float *dst = As + buf * (A_ROWS * LDS_LD);
#pragma unroll
        for (int j = 0; j < A_REGS; ++j) {
            const int f = tid + NT * j;
            if constexpr (VEC) {
                const int lr = f >> 3;
                if (lr < BMT + MAX_SPAN) *reinterpret_cast<f32x4 *>(dst + lr * LDS_LD + (f & 7) * 4) = areg[j];
            } else {
                const int lr = f >> 5;
                if (lr < BMT + MAX_SPAN) dst[lr * LDS_LD + (f & 31)] = areg[j];
            }
        }
    };

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

    // prologue: stage 0
    load_a(c_lo);
    load_b(c_lo, 0);
    store_a(c_lo & 1);
    store_b(0);
    __syncthreads();

    int chunk = c_lo, tap = 0;
    for (int s = 0; s < n_stages; ++s) {
        int nchunk = chunk, ntap = tap + 1;
        if (ntap == p.K) { ntap = 0; nchunk = chunk + 1; }
        const bool has_next = (s + 1) < n_stages;
        const bool new_a = has_next && (ntap == 0);
        if (has_next) {
            load_b(nchunk, ntap);
            if (new_a) load_a(nchunk);
        }

        const float *Ab = As + (chunk & 1) * (A_ROWS * LDS_LD) +
                          (wr * WROWS + (lane & 31) + tap * p.dil) * LDS_LD + (lane >> 5) * 4;
        const float *Bb = Bs + (s & 1) * (BN * LDS_LD) + (wc * 64 + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(Ab + kk * 8);
            f32x4 a1 = a0;
            if constexpr (BMT == 128) a1 = *reinterpret_cast<const f32x4 *>(Ab + 32 * LDS_LD + kk * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(Bb + kk * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(Bb + 32 * LDS_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc01, 0, 0, 0);
                if constexpr (BMT == 128) {
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc10, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc11, 0, 0, 0);
                }
            }
        }

        if (has_next) {
            store_b((s + 1) & 1);
            if (new_a) store_a(nchunk & 1);
        }
        __syncthreads();
        chunk = nchunk;
        tap = ntap;
    }

    if (p.k_splits > 1) {
        // raw partial sums of this slab group as fp32 rows: the ordinary epilogue with nothing to add (bias, activation and BN
        // belong to the kernel that adds the groups up)
        GemmParams qp = p;
        qp.bias = qp.scale = qp.shift = nullptr;
        qp.act = XV_ACT_NONE;
        qp.y = nullptr;
        qp.blk = nullptr;
        qp.ypre = p.part + (size_t)ks * p.part_stride;
        qp.ldy = p.cout;
        if (p.vec_out) gemm_epilogue_rows<BMT>(qp, smem, Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11);
        else gemm_epilogue<WROWS>(qp, Ms, m0, n0, wr, wc, lane, acc00, acc01, acc10, acc11);
        return;
    }
    if (p.vec_out) gemm_epilogue_rows<BMT>(p, smem, Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11);     // (the loop ended with a barrier)
    else gemm_epilogue<WROWS>(p, Ms, m0, n0, wr, wc, lane, acc00, acc01, acc10, acc11);
}

// y = act(bias + sum_s part[s]) * scale + shift, ypre = bias + sum: the groups of a split-K launch added in group order (deterministic)
__global__ void splitk_reduce_kernel(const float *__restrict__ part, long part_stride, int nsplit, int R, int cout, const float *bias,
                                     const float *scale, const float *shift, int act, const float *alpha, float *y, int ldy, float *ypre,
                                     int ldpre)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * cout) return;
    const int r = (int)(i / cout), c = (int)(i - (size_t)r * cout);
    float z = bias ? bias[c] : 0.f;
    for (int k = 0; k < nsplit; ++k) z += part[(size_t)k * part_stride + i];
    if (ypre) ypre[(size_t)r * ldpre + c] = z;
    if (y) {
        const float a = act == XV_ACT_LRELU ? alpha[0] : act == XV_ACT_PRELU ? alpha[c] : 0.f;
        y[(size_t)r * ldy + c] = apply_act(z, act, a) * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16x3 split-precision GEMM (v2: DMA-fed).
//
// Every fp32 operand is split x = hi + lo (hi = bf16(x), lo = bf16(x - hi)); products are accumulated in
// fp32 as lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 (the lo*lo term, ~2^-16 relative, is dropped).
//
// Operand formats (all produced on the GPU, see include/xvector_hip.h):
//  * weights: xv_pack_weights_bf16x3 writes one 16 KB tile per (column tile, channel slab, tap) in exactly the
//    LDS image order ([hi: 128 cols x 64 B][lo: 128 x 64 B], 16-B slots XOR-swizzled by (col>>2)&3), tiles
//    ordered as the K-loop walks them -> a stage's B operand is ONE linear 16 KB global->LDS DMA.
//  * activations between layers: "split" format -- per row and 32-channel slab 128 B = [4 hi slots | 4 lo slots]
//    of 8 bf16, slots XOR-swizzled by (row>>1)&7 -> the (128+(K-1)d)-row halo tile of a slab is a linear DMA of
//    128 B per row, re-used by all K taps; same bytes per element as fp32.
//  * the first layer / the segment FC read plain fp32 rows and split them while staging (FP32 A mode).
// Mainloop per stage and wave: 4 B-DMA + ~1 A-DMA instructions, 16 ds_read_b128, 24 MFMAs, one barrier.
// Epilogue: accumulators -> LDS (fp32 tile) -> bias/act/BN/gap-mask -> 16-byte NON-TEMPORAL stores (fp32 rows or
// split): the outputs are 0.27-0.8 GB streams, and plain stores (L2 write-allocate) measured 7-16 % slower.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int SROW = 128;                         // bytes per (row, 32-channel slab): LDS A row and HBM split row-slab
constexpr int B3_PLANE = BN * 64;                 // 8192
constexpr int B3_BYTES = 2 * B3_PLANE;            // 16384: [hi tile][lo tile]
constexpr int T_LD = BN + 4;                      // epilogue fp32 tile row (floats)
// Workgroup geometry of the bf16x3 kernel: WM x 2 waves, each wave a 64 x 64 sub-tile (2 x 2 MFMA tiles) -> tile of
// WM*64 rows x 128 columns.  WM = 2: 4 waves, 128-row tile, 69.7 KB of LDS, two workgroups per CU.  WM = 4: 8 waves,
// 256-row tile, one workgroup per CU (same waves per SIMD): a weight tile feeds twice the MFMAs, i.e. half the
// global->LDS weight bytes per FLOP.  LDS: [operand buffers | epilogue fp32 tile (aliased)] [row mask] [epilogue params].
constexpr int gemm3_oper_bytes(int wm) { return 2 * (wm * 64 + MAX_SPAN) * SROW + 2 * B3_BYTES; }
constexpr int gemm3_tile_bytes(int wm) { return wm * 64 * T_LD * 4; }
constexpr int gemm3_mask_off(int wm) { return gemm3_oper_bytes(wm) > gemm3_tile_bytes(wm) ? gemm3_oper_bytes(wm) : gemm3_tile_bytes(wm); }
constexpr size_t gemm3_lds_bytes(int wm) { return (size_t)gemm3_mask_off(wm) + wm * 64 + 4 * BN * sizeof(float); }

// f(integral_constant<int, T>) for T = T0 .. KT-1, unrolled at compile time
template <int T, int KT, class F>
__device__ __forceinline__ void for_taps(F &f)
{
    if constexpr (T < KT) {
        f(std::integral_constant<int, T>{});
        for_taps<T + 1, KT>(f);
    }
}

struct Gemm3Params {
    const void *x;        // fp32 rows (x_split == 0) or split buffer (row 0 of it)
    int x_split;
    long R;
    int cin, ldx, xchunks;
    const uint8_t *wt;    // tiled bf16x3 weights
    const float *bias, *scale, *shift;
    int act;
    const float *alpha;
    int K, dil, cout;
    const uint8_t *valid;
    void *y;              // fp32 rows or split buffer (may be NULL when only ypre is wanted)
    int y_split, ldy, ychunks;
    float *ypre;          // fp32 rows, stride ldpre (optional)
    int ldpre;
    float *blk;           // POOL epilogue: per-8-row-block (mean, M2) planes [ceil(R/8)][2][cout]
    int n_mt, n_nt, n_chunks;
    // column sums of the fp32 output (training: the BN backward of the layer BELOW needs sum y and sum y * r over the rows, y = the
    // input gradient this launch produces): per row tile partials cs_part[mt][{sum y, sum y r}][cout] in double, merged in order by
    // xv_col_sums_merge_f32.  cs_r: the other factor, fp32 rows of stride cs_ldr; NULL = the output itself (sum y, sum y^2: the
    // batch moments BN(training) takes of a forward layer's activation output, accumulated in double).  cs_part NULL = off.
    const float *cs_r;
    int cs_ldr;
    double *cs_part;
};

#define XV_GLDS16_OFF(gptr, lptr, imm)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                    \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, imm, 0)
#define XV_GLDS16(gptr, lptr)                                                                                   \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                    \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// KT: kernel size K of the split-input path as a compile-time constant (1, 3, 5, 7; 0 = the fp32-input path, runtime K).
// The stage loop of that path is unrolled over the K taps of a slab, so everything that depends on the tap -- fragment
// row offsets, the A-halo DMA schedule -- is computed once, and an iteration carries ~15 integer instructions next to
// its 24 MFMAs instead of ~85 (each one beside an MFMA costs issue slots AND clock on this power-limited loop).
// A-halo DMA pieces (8 rows = 1 KB each): the (<=17)-piece halo tile of the NEXT slab is spread over taps 0..K-2 of
// the current slab, PW pieces per wave per tap (K=1: 4 -- every stage loads its own slab --, K=3: 3, K=5: 2, K=7: 1).
// POOL: the layer output is not stored; the epilogue reduces every 8-row block of the tile to per-channel (mean, M2)
// for the statistics pooling that follows the last frame-level layer (see stats_pool_blocks_kernel).
// S16 (split input, K > 1, an even number of slabs): the same tile and the same bytes through LDS on v_mfma_f32_16x16x32_bf16 -- a
// dot product twice as long per instruction, quarter-size accumulator tiles: fewer joules per product on the power-limited pipe
// (tools/experiments/shape_probe.hip).  A wave holds ALL 16 fragments of a stage (4 row tiles + 4 column tiles, hi and lo: 64
// VGPRs) and reads the next stage's 16 behind the 48 MFMAs of the current one: two fragment sets + 64 accumulators = 192 VGPRs.
template <bool SPLIT_A, int KT, bool POOL, int WM, bool S16 = false>
__global__ __launch_bounds__(WM * 128, (S16 && WM == 4) ? 1 : 2) void tdnn_gemm_bf16x3_kernel(const Gemm3Params p)
{
    static_assert(!S16 || (SPLIT_A && KT > 1), "the 16 x 16 form exists for split input and K > 1");
    constexpr int NW = 2 * WM;                         // waves per workgroup
    constexpr int NT = NW * 64;                        // threads (shadows the file-scope constant of the fp32 kernel)
    constexpr int BM = WM * 64;                        // rows per workgroup tile
    constexpr int A_ROWS = BM + MAX_SPAN;
    constexpr int A3_BYTES = A_ROWS * SROW;
    constexpr int BP = 16 / NW;                        // 1 KB pieces of a 16 KB weight tile per wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Abuf = lds;                                  // [2][A_ROWS][128 B]
    char *Bbuf = lds + 2 * A3_BYTES;                   // [2][hi 8 KB | lo 8 KB]
    uint8_t *Ms = reinterpret_cast<uint8_t *>(lds + gemm3_mask_off(WM));
    float *Ps = reinterpret_cast<float *>(lds + gemm3_mask_off(WM) + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order.  Hardware places block b on XCD b%8; every XCD gets a contiguous run of logical tile ids
    // L = mt*n_nt + nt (bijective chunking), column tiles fastest, so the n_nt tiles sharing an A panel run close
    // together on one L2.  (Measured alternatives that did NOT help: column-tile-major order to pin one weight panel
    // in L2 (-3 %), persistent workgroups (-4 %), de-synchronised start delays (0 %), s_setprio around the MFMAs (0 %).)
    const int nwg = p.n_mt * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = wg / p.n_nt, nt = wg - mt * p.n_nt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int span = (p.K - 1) * p.dil;
    const int left = span >> 1;
    const int rowsA = BM + span;
    const int n_stages = p.n_chunks * p.K;
    const int goff = (int)((m0 - left) & 15);          // LDS row lr <-> global row gr: (gr & 15) == (lr + goff) & 15

    if (tid < BM) {
        const long gr = m0 + tid;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    }
    // epilogue parameters of the tile's 128 columns, staged now so that their latency hides behind the main loop:
    // [bias | BN scale | BN shift | alpha], alpha such that act(z) = max(z,0) + alpha*min(z,0) for none (1), relu (0), prelu.
    // Columns beyond Cout (ragged last tile) get scale = shift = 0 so they come out as exact zeros.
    if (tid >= BM && tid < BM + BN) {
        const int c = tid - BM, gc = n0 + c;
        const bool ok = gc < p.cout;
        Ps[c] = (ok && p.bias) ? p.bias[gc] : 0.f;
        Ps[BN + c] = ok ? (p.scale ? p.scale[gc] : 1.f) : 0.f;
        Ps[2 * BN + c] = (ok && p.shift) ? p.shift[gc] : 0.f;
        Ps[3 * BN + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.alpha[0]
                       : (p.act == XV_ACT_PRELU && ok) ? p.alpha[gc] : 0.f;
    }

    // ---- B: one 16 KB tile per stage, 4 x 1 KB DMA pieces per wave ---------------------------------------
    const uint8_t *bsrc = p.wt + (size_t)nt * n_stages * B3_BYTES + wave * (BP * 1024) + lane * 16;
    auto dma_b = [&](int buf) {
        char *dst = Bbuf + buf * B3_BYTES + wave * (BP * 1024);
#pragma unroll
        for (int j = 0; j < BP; ++j) XV_GLDS16(bsrc + j * 1024, dst + j * 1024);
        bsrc += B3_BYTES;
    };

    // ---- A (split input): halo tile = rowsA rows x 128 B, DMA pieces of 8 rows --------------------------
    const size_t xrow_bytes = (size_t)p.xchunks * SROW;
    const uint8_t *asrc = nullptr;
    if constexpr (SPLIT_A)
        asrc = reinterpret_cast<const uint8_t *>(p.x) + (m0 - left + wave * 8 + (lane >> 3)) * (long)xrow_bytes + (lane & 7) * 16;
    const int n_pieces = (rowsA + 7) >> 3;             // <= BM/8 + 1
    auto dma_a = [&](int buf) {
        char *dst = Abuf + buf * A3_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int piece = wave + NW * j;
            if (piece < n_pieces) XV_GLDS16(asrc + (size_t)(8 * NW * j) * xrow_bytes, dst + j * (NW * 1024));
        }
        asrc += SROW;                                   // next 32-channel slab
    };

    // ---- A (fp32 input): load rows, split to hi/lo while writing the same LDS image -----------------------
    f32x4 areg[5];
    auto load_a = [&](int chunk) {
        const float *xf = reinterpret_cast<const float *>(p.x);
        const int c0 = chunk * BK;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int f = tid + NT * j;
            const int lr = f >> 3, qq = f & 7;
            const long gr = m0 - left + lr;
            const int c = c0 + qq * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (lr < rowsA && gr >= 0 && gr < p.R && c < p.cin) v = *reinterpret_cast<const f32x4 *>(xf + (size_t)gr * p.ldx + c);
            areg[j] = v;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int f = tid + NT * j;
            const int lr = f >> 3, qq = f & 7;
            if (lr < A_ROWS) {
                bf16x4 hi, lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hi[i] = (__bf16)areg[j][i];
                    lo[i] = (__bf16)(areg[j][i] - (float)hi[i]);
                }
                const int sw = ((lr + goff) & 15) >> 1;
                char *row = Abuf + buf * A3_BYTES + lr * SROW + (qq & 1) * 8;
                *reinterpret_cast<bf16x4 *>(row + (((qq >> 1)) ^ sw) * 16) = hi;
                *reinterpret_cast<bf16x4 *>(row + ((4 + (qq >> 1)) ^ sw) * 16) = lo;
            }
        }
    };

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};

    // fragment addressing (B is stage-invariant up to the buffer toggle)
    const int arow0 = wr * 64 + (lane & 31);
    const int brow = wc * 64 + (lane & 31);
    const int kh = lane >> 5;
    const int boff0 = brow * 64, boff1 = (brow + 32) * 64;
    const int bsw0 = (brow >> 2) & 3, bsw1 = ((brow + 32) >> 2) & 3;

    struct Frags {            // fragments of one k-step (16 of the 32 channels of a stage): 8 x 4 VGPRs
        bf16x8 ah0, al0, ah1, al1, bh0, bl0, bh1, bl1;
    };
    auto load_frags = [&](Frags &F, int st, int ch, int tp, int ks) {
        const char *Ab = Abuf + (ch & 1) * A3_BYTES;
        const char *Bb = Bbuf + (st & 1) * B3_BYTES;
        const int lr0 = arow0 + tp * p.dil, lr1 = lr0 + 32;
        const int sw0 = ((lr0 + goff) & 15) >> 1, sw1 = ((lr1 + goff) & 15) >> 1;
        const char *a0 = Ab + lr0 * SROW, *a1 = Ab + lr1 * SROW;
        const int t = ks * 2 + kh;
        F.al0 = *reinterpret_cast<const bf16x8 *>(a0 + (((t + 4) ^ sw0) << 4));
        F.bh0 = *reinterpret_cast<const bf16x8 *>(Bb + boff0 + ((t ^ bsw0) << 4));
        F.bh1 = *reinterpret_cast<const bf16x8 *>(Bb + boff1 + ((t ^ bsw1) << 4));
        F.al1 = *reinterpret_cast<const bf16x8 *>(a1 + (((t + 4) ^ sw1) << 4));
        F.ah0 = *reinterpret_cast<const bf16x8 *>(a0 + ((t ^ sw0) << 4));
        F.bl0 = *reinterpret_cast<const bf16x8 *>(Bb + B3_PLANE + boff0 + ((t ^ bsw0) << 4));
        F.bl1 = *reinterpret_cast<const bf16x8 *>(Bb + B3_PLANE + boff1 + ((t ^ bsw1) << 4));
        F.ah1 = *reinterpret_cast<const bf16x8 *>(a1 + ((t ^ sw1) << 4));
    };
    auto mma = [&](const Frags &F) {
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.al0, F.bh0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.al0, F.bh1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.al1, F.bh0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.al1, F.bh1, acc11, 0, 0, 0);
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah0, F.bl0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah0, F.bl1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah1, F.bl0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah1, F.bl1, acc11, 0, 0, 0);
        acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah0, F.bh0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah0, F.bh1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah1, F.bh0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(F.ah1, F.bh1, acc11, 0, 0, 0);
    };

    // Register-level software pipeline at k-step granularity (two 32-VGPR fragment sets F, G):
    //   iteration s:  G <- LDS(stage s, k-step 1) | 12 MFMAs on F (stage s, k-step 0)
    //                 barrier B(s)   [stage s fully read by everybody; stage s+1 landed]
    //                 DMA(stage s+2) into the buffers of stage s | F <- LDS(stage s+1, k-step 0) | 12 MFMAs on G
    // Each MFMA group hides the LDS latency of the other set's reads; a DMA has a full stage to land.
    int c1 = 0, t1 = 0;                       // (chunk, tap) of stage s+1
    auto advance = [&](int &c, int &t) { if (++t == p.K) { t = 0; ++c; } };

    dma_b(0);
    if constexpr (SPLIT_A) dma_a(0);
    else { load_a(0); store_a(0); }
    advance(c1, t1);
    if (n_stages > 1) {
        dma_b(1);
        if (t1 == 0) {
            if constexpr (SPLIT_A) dma_a(1);
            else { load_a(1); store_a(1); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frags F = {}, G = {};
    if constexpr (!S16) load_frags(F, 0, 0, 0, 0);

    int c0 = 0, t0 = 0;                       // (chunk, tap) of stage s
    int c2 = c1, t2 = t1;                     // (chunk, tap) of stage s+2
    advance(c2, t2);
    f32x4 acc16[4][4];                        // S16: row tile i, column tile j of the wave's 64 x 64
    if constexpr (S16) {
        constexpr int NP = BM / 8 + 1;
        constexpr int DT = KT - 1;
        constexpr int NS = (NP + NW - 1) / NW;
        constexpr int PW = (NS + DT - 1) / DT;
        auto slots_of = [](int t) constexpr { return t < DT ? (NS + DT - 1 - t) / DT : 0; };
        auto slot_base = [](int t) constexpr { int b = 0; for (int u = 0; u < t; ++u) b += (NS + DT - 1 - u) / DT; return b; };
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // lane (row / column l & 15, k block l >> 4 = the 16-byte slot of the 32-channel slab).  A ds_read_b128 is served in four
        // groups of 16 lanes that mix two k blocks ({0-3, 12-15, 20-27}, ...): with tile row i = lane & 15 read where it lies, the
        // slot swizzles of the operand formats (made for 32-row fragments) collide two-way (SQ_LDS_BANK_CONFLICT 221 M cycles per
        // K = 7 launch against 1 M for the 32 x 32 form).  MFMA tile row / column i is therefore ROW16(i) / COL16(i) of the 16 --
        // permutations under which every lane group touches 16 different 16-byte chunks for every tap offset (found by
        // search over the bank model of MI355X_MICROARCH.md); the accumulators go back through the same maps.
        const int kb = lane >> 4;
        auto ROW16 = [](int i) { return (int)((0x48c67dbf391502eaull >> (4 * i)) & 15); };
        auto COL16 = [](int i) { return (int)((0xfedc76543210ba98ull >> (4 * i)) & 15); };
        int pa16[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int lr0 = wr * 64 + ROW16(lane & 15) + t * p.dil;
            pa16[t] = lr0 * SROW + (((((lr0 + goff) & 15) >> 1) ^ kb) << 4);      // (+ 16 i rows: the same swizzle; lo plane: ^ 64)
        }
        const int col16 = wc * 64 + COL16(lane & 15);
        const int pb16 = 2 * A3_BYTES + col16 * 64 + ((kb ^ ((col16 >> 2) & 3)) << 4);   // (+ 16 j columns: + 1024, the same swizzle)
        const uint32_t rowstep = 8u * (uint32_t)xrow_bytes;
        uint32_t ag_off[DT][PW], al_off[DT][PW];
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                int piece = (slot_base(t) + j) * NW + wave;
                piece = piece < NP ? piece : NP - 1;
                ag_off[t][j] = (uint32_t)piece * rowstep;
                al_off[t][j] = (uint32_t)piece * 1024u;
            }
        const uint8_t *bnext = bsrc;
        if (n_stages <= 2) bnext = bsrc - B3_BYTES;
        const uint8_t *abase = reinterpret_cast<const uint8_t *>(p.x) + (m0 - left + (lane >> 3)) * (long)xrow_bytes + (lane & 7) * 16;
        struct Set16 { bf16x8 ah[4], al[4], bh[4], bl[4]; };
        auto load16 = [&](Set16 &X, int abase_off, int bbase_off) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                X.ah[i] = *reinterpret_cast<const bf16x8 *>(lds + abase_off + i * 16 * SROW);
                X.al[i] = *reinterpret_cast<const bf16x8 *>(lds + (abase_off ^ 64) + i * 16 * SROW);
                X.bh[i] = *reinterpret_cast<const bf16x8 *>(lds + bbase_off + i * 1024);
                X.bl[i] = *reinterpret_cast<const bf16x8 *>(lds + bbase_off + B3_PLANE + i * 1024);
            }
        };
        auto mma16 = [&](const Set16 &X) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X.al[i], X.bh[j], acc16[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X.ah[i], X.bl[j], acc16[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X.ah[i], X.bh[j], acc16[i][j], 0, 0, 0);
        };
        Set16 F16, G16;
        load16(F16, pa16[0], pb16);
        int s = 0;
        for (int c = 0; c < p.n_chunks; c += 2) {         // two slabs per trip: the fragment sets swap roles every stage, K is odd
            auto stage = [&](auto UU) {
                constexpr int u = decltype(UU)::value;
                constexpr int t = u % KT;
                const int cc = c + u / KT;
                const int abuf = (cc & 1) * A3_BYTES;
                const int cn = (cc + 1 < p.n_chunks) ? cc + 1 : p.n_chunks - 1;
                const uint8_t *anext = abase + (size_t)cn * SROW;
                char *adst_n = Abuf + (cn & 1) * A3_BYTES;
                const int bbuf = (s & 1) * B3_BYTES;
                // stage s is in registers (everybody's reads of it have returned), stage s+1 has landed
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                {
                    char *dst = Bbuf + bbuf + wave * (BP * 1024);
                    XV_GLDS16_OFF(bnext, dst, 0);
                    XV_GLDS16_OFF(bnext, dst, 1024);
                    if constexpr (BP == 4) {
                        XV_GLDS16_OFF(bnext, dst, 2048);
                        XV_GLDS16_OFF(bnext, dst, 3072);
                    }
                    bnext += (s + 3 < n_stages) ? B3_BYTES : 0;
                }
                if constexpr (t < KT - 1) {
#pragma unroll
                    for (int j = 0; j < slots_of(t); ++j) XV_GLDS16(anext + ag_off[t][j], adst_n + al_off[t][j]);
                }
                const int a_next = (t + 1 < KT) ? pa16[(t + 1) % KT] + abuf : pa16[0] + (A3_BYTES - abuf);
                if constexpr ((u & 1) == 0) { load16(G16, a_next, pb16 + (B3_BYTES - bbuf)); mma16(F16); }
                else { load16(F16, a_next, pb16 + (B3_BYTES - bbuf)); mma16(G16); }
                constexpr int NV = BP + slots_of(t);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (LDS-DMA piece)
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 48 - 32 - NV, 0);
                __builtin_amdgcn_sched_barrier(0);
                ++s;
            };
            for_taps<0, 2 * KT>(stage);
        }
    } else if constexpr (SPLIT_A) {
        // Straight-line iteration body (no branches): DMA is issued unconditionally (clamped at the tail, where it
        // rewrites identical bytes), so that sched_group_barrier can interleave every memory instruction with the
        // MFMAs of the same wave: the wave overlaps its own memory issue instead of relying on the co-resident block.
        constexpr int NP = KT == 1 ? BM / 8 : BM / 8 + 1;   // pieces of a halo tile: BM + (K-1)*dil rows, (K-1)*dil in 2..8
        constexpr int DT = KT == 1 ? 1 : KT - 1;            // taps that carry A pieces
        // a "slot" = one piece per wave; the NS slots a halo tile needs are dealt to the taps as evenly as possible,
        // early taps first (K=7: 1,1,1,1,1,0  K=5: 2,1,1,1  K=3: 3,2  K=1: 4), so that at most NW-1 pieces per slab are
        // clamped duplicates
        constexpr int NS = (NP + NW - 1) / NW;
        constexpr int PW = (NS + DT - 1) / DT;              // most slots any tap carries
        auto slots_of = [](int t) constexpr { return t < DT ? (NS + DT - 1 - t) / DT : 0; };
        auto slot_base = [](int t) constexpr { int b = 0; for (int u = 0; u < t; ++u) b += (NS + DT - 1 - u) / DT; return b; };
        // per-tap, per-lane fragment row offset in A buffer 0 with the slot swizzle and the lane's k-half folded in:
        // the 16-B slot T of a row sits at ((T ^ sw) << 4), T = ks*2 + kh (+4 for lo)  ->  pa ^ (ks << 5) ^ (lo << 6)
        int pa[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int lr0 = arow0 + t * p.dil;
            pa[t] = lr0 * SROW + (((((lr0 + goff) & 15) >> 1) ^ kh) << 4);
        }
        // B fragments: (col + 32) has the same swizzle, the lo plane is +B3_PLANE -> immediates; per k-step one base
        int pb[2];
        pb[0] = 2 * A3_BYTES + boff0 + ((kh ^ bsw0) << 4);
        pb[1] = 2 * A3_BYTES + boff0 + (((2 + kh) ^ bsw0) << 4);
        // A-halo DMA schedule of this wave: byte offset of piece (t, j) in the split buffer / in an LDS A buffer
        const uint32_t rowstep = 8u * (uint32_t)xrow_bytes;
        uint32_t ag_off[DT][PW], al_off[DT][PW];
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                int piece = (slot_base(t) + j) * NW + wave;      // slots beyond slots_of(t) are never issued
                piece = piece < NP ? piece : NP - 1;
                ag_off[t][j] = (uint32_t)piece * rowstep;
                al_off[t][j] = (uint32_t)piece * 1024u;
            }
        const uint8_t *bnext = bsrc;                         // tile of stage min(s+2, n_stages-1)
        if (n_stages <= 2) bnext = bsrc - B3_BYTES;
        const uint8_t *abase = reinterpret_cast<const uint8_t *>(p.x) + (m0 - left + (lane >> 3)) * (long)xrow_bytes + (lane & 7) * 16;
        auto load_a_frags = [&](Frags &X, int base, int ks) {       // base = pa[t] (+ buffer offset), ks compile-time
            const char *a = lds + (base ^ (ks << 5));
            const char *al = lds + (base ^ (ks << 5) ^ 64);
            X.al0 = *reinterpret_cast<const bf16x8 *>(al);
            X.al1 = *reinterpret_cast<const bf16x8 *>(al + 32 * SROW);
            X.ah0 = *reinterpret_cast<const bf16x8 *>(a);
            X.ah1 = *reinterpret_cast<const bf16x8 *>(a + 32 * SROW);
        };
        auto load_b_frags = [&](Frags &X, int base) {               // base = pb[ks] + stage buffer offset
            const char *b = lds + base;
            X.bh0 = *reinterpret_cast<const bf16x8 *>(b);
            X.bh1 = *reinterpret_cast<const bf16x8 *>(b + 32 * 64);
            X.bl0 = *reinterpret_cast<const bf16x8 *>(b + B3_PLANE);
            X.bl1 = *reinterpret_cast<const bf16x8 *>(b + B3_PLANE + 32 * 64);
        };
        int s = 0;
        for (int c = 0; c < p.n_chunks; ++c) {
            const int abuf = (c & 1) * A3_BYTES;                     // A buffer of slab c / of slab c+1
            const int abuf_n = A3_BYTES - abuf;
            // slab whose halo is loaded while slab c is consumed (K > 1), clamped at the tail
            const int cn = (c + 1 < p.n_chunks) ? c + 1 : p.n_chunks - 1;
            const uint8_t *anext = abase + (size_t)cn * SROW;
            char *adst_n = Abuf + (cn & 1) * A3_BYTES;
            auto tap = [&](auto TT) {
                constexpr int t = decltype(TT)::value;
                const int bbuf = (s & 1) * B3_BYTES;
                // ---- phase 1: G <- LDS(stage s, k-step 1) interleaved with the 12 MFMAs on F -----------------------
                load_a_frags(G, pa[t] + abuf, 1);
                load_b_frags(G, pb[1] + bbuf);
                mma(F);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                  // B(s): stage s fully read by everybody, stage s+1 landed
                // ---- phase 2: DMA(s+2), F <- LDS(stage s+1, k-step 0), 12 MFMAs on G --------------------------------
                {
                    char *dst = Bbuf + bbuf + wave * (BP * 1024);    // one M0; the immediate advances source AND destination
                    XV_GLDS16_OFF(bnext, dst, 0);
                    XV_GLDS16_OFF(bnext, dst, 1024);
                    if constexpr (BP == 4) {
                        XV_GLDS16_OFF(bnext, dst, 2048);
                        XV_GLDS16_OFF(bnext, dst, 3072);
                    }
                    bnext += (s + 3 < n_stages) ? B3_BYTES : 0;
                }
                if constexpr (KT == 1) {
                    // every stage is its own slab: all 16 pieces of slab min(s+2, last) now, into the buffer of slab s
                    const int ca = (s + 2 < p.n_chunks) ? s + 2 : p.n_chunks - 1;
                    const uint8_t *ag = abase + (size_t)ca * SROW;
                    char *adst = Abuf + (ca & 1) * A3_BYTES;
#pragma unroll
                    for (int j = 0; j < PW; ++j) XV_GLDS16(ag + ag_off[0][j], adst + al_off[0][j]);
                } else if constexpr (t < KT - 1) {
#pragma unroll
                    for (int j = 0; j < slots_of(t); ++j) XV_GLDS16(anext + ag_off[t][j], adst_n + al_off[t][j]);
                }
                if constexpr (t + 1 < KT) load_a_frags(F, pa[t + 1] + abuf, 0);
                else load_a_frags(F, pa[0] + abuf_n, 0);             // first tap of the next slab (tail: harmless read)
                load_b_frags(F, pb[0] + (B3_BYTES - bbuf));
                mma(G);
                constexpr int NV = BP + (KT == 1 ? PW : slots_of(t));
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (LDS-DMA piece)
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 DS reads
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 12 - (4 + NV) > 0 ? 12 - (4 + NV) : 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                ++s;
            };
            for_taps<0, KT>(tap);
        }
    } else {
        for (int s = 0; s < n_stages; ++s) {
            load_frags(G, s, c0, t0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(F);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                  // B(s)
            const bool dma2 = (s + 2) < n_stages;
            const bool newa2 = dma2 && (t2 == 0);
            if (dma2) {
                dma_b(s & 1);
                if (newa2) load_a(c2);
            }
            if (s + 1 < n_stages) load_frags(F, s + 1, c1, t1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma(G);
            __builtin_amdgcn_sched_barrier(0);
            if (newa2) store_a(c2 & 1);
            c0 = c1; t0 = t1;
            advance(c1, t1);
            advance(c2, t2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: accumulators -> LDS fp32 tile (the operand buffers are dead after the last barrier) ------
    float *T = reinterpret_cast<float *>(lds);
    if constexpr (S16) {
        const int col = wc * 64 + (int)((0xfedc76543210ba98ull >> (4 * (lane & 15))) & 15);                 // COL16
        int rows4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) rows4[e] = wr * 64 + (int)((0x48c67dbf391502eaull >> (4 * (4 * (lane >> 4) + e))) & 15);   // ROW16
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(rows4[e] + 16 * i) * T_LD + col + 16 * j] = acc16[i][j][e];
    } else {
        const int col = wc * 64 + (lane & 31);
        const int rowb = wr * 64 + 4 * (lane >> 5);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int rr = rowb + (reg & 3) + 8 * (reg >> 2);
            T[rr * T_LD + col] = acc00[reg];
            T[rr * T_LD + col + 32] = acc01[reg];
            T[(rr + 32) * T_LD + col] = acc10[reg];
            T[(rr + 32) * T_LD + col + 32] = acc11[reg];
        }
    }
    __syncthreads();

    const int cg = tid & 15;                            // 8-channel group of the 128-column tile
    const int gc0 = n0 + cg * 8;
    float bias[8], sc[8], sh[8], al[8];
    {
        const f32x4 *P4 = reinterpret_cast<const f32x4 *>(Ps) + cg * 2;
        const f32x4 q0 = P4[0], q1 = P4[1], q2 = P4[BN / 4], q3 = P4[BN / 4 + 1], q4 = P4[2 * BN / 4], q5 = P4[2 * BN / 4 + 1],
                    q6 = P4[3 * BN / 4], q7 = P4[3 * BN / 4 + 1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bias[i] = q0[i]; bias[4 + i] = q1[i];
            sc[i] = q2[i]; sc[4 + i] = q3[i];
            sh[i] = q4[i]; sh[4 + i] = q5[i];
            al[i] = q6[i]; al[4 + i] = q7[i];
        }
    }
    const bool full = gc0 + 8 <= p.cout;
    const bool lrelu = p.act == XV_ACT_LRELU;           // tf.nn.leaky_relu is max(alpha*z, z) for ANY alpha
    auto activate = [&](float z, float a) { return lrelu ? fmaxf(a * z, z) : fmaxf(z, 0.f) + a * fminf(z, 0.f); };
    if constexpr (POOL) {
        // thread = (8-row block tid>>4 of the tile, 8 channels): statistics of the block's valid rows, shifted by the
        // block's first row so that s2 - s1^2/n does not cancel.  Blocks are aligned to global row multiples of 8;
        // callers start every chunk on such a row, so a block never mixes two chunks and its statistics do not depend
        // on where the chunk sits in the batch.
        const int blk = tid >> 4;
        if (m0 + blk * 8 >= p.R) return;
        float v0[8], s1[8], s2[8];
        float n = 0.f;
        f32x4 tv[8][2];
        float keep[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lr = blk * 8 + j;
            tv[j][0] = *reinterpret_cast<const f32x4 *>(T + lr * T_LD + cg * 8);
            tv[j][1] = *reinterpret_cast<const f32x4 *>(T + lr * T_LD + cg * 8 + 4);
            keep[j] = Ms[lr] ? 1.f : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        auto rows = [&](auto MODE) {                   // 0: max(z,0) + alpha*min(z,0)   1: leaky max(alpha*z, z)   2: plain ReLU
            constexpr int mode = decltype(MODE)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                n += keep[j];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float z = (i < 4 ? tv[j][0][i] : tv[j][1][i - 4]) + bias[i];
                    const float a = mode == 1 ? fmaxf(al[i] * z, z) : mode == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + al[i] * fminf(z, 0.f);
                    const float v = a * sc[i] + sh[i];
                    if (j == 0) { v0[i] = v; s1[i] = 0.f; s2[i] = 0.f; }
                    else {
                        const float d = keep[j] != 0.f ? v - v0[i] : 0.f;      // (a select: a row past R may hold anything, NaN * 0 is NaN)
                        s1[i] += d;
                        s2[i] += d * d;
                    }
                }
            }
        };
        if (lrelu) rows(std::integral_constant<int, 1>{});
        else if (p.act == XV_ACT_RELU) rows(std::integral_constant<int, 2>{});
        else rows(std::integral_constant<int, 0>{});
        // row 0 of a block is valid whenever any row is (chunks start on block boundaries, gaps follow the frames)
        const float rn = n > 0.f ? 1.f / n : 0.f;
        float mean[8], m2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mean[i] = n > 0.f ? v0[i] + s1[i] * rn : 0.f;
            m2[i] = fmaxf(s2[i] - s1[i] * s1[i] * rn, 0.f);
        }
        float *o = p.blk + ((size_t)((m0 >> 3) + blk) * 2) * p.cout + gc0;
        if (full && !(p.cout & 3)) {
            *reinterpret_cast<f32x4 *>(o) = (f32x4){mean[0], mean[1], mean[2], mean[3]};
            *reinterpret_cast<f32x4 *>(o + 4) = (f32x4){mean[4], mean[5], mean[6], mean[7]};
            *reinterpret_cast<f32x4 *>(o + p.cout) = (f32x4){m2[0], m2[1], m2[2], m2[3]};
            *reinterpret_cast<f32x4 *>(o + p.cout + 4) = (f32x4){m2[4], m2[5], m2[6], m2[7]};
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (gc0 + i < p.cout) { o[i] = mean[i]; o[p.cout + i] = m2[i]; }
        }
        return;
    }
    if (p.y && p.y_split && !p.ypre && n0 + BN <= p.cout) {
        // fast path (hidden layers): full-width tile into the split format, straight-line.  Rows >= R of the last tile
        // land in the buffer's zero padding (XV_SPLIT_PAD_AFTER >= BM) and are written as zeros (keep == 0).
        const int ch = gc0 >> 5, slot = cg & 3;
        char *ybase = reinterpret_cast<char *>(p.y) + (size_t)ch * SROW;
        const size_t yrow = (size_t)p.ychunks * SROW;
        // all 16 LDS reads first: the LDS pipe is kept busy by the co-resident workgroup's main loop, so a round trip costs
        // ~1 us under load -- pay it once, not once per row
        f32x4 tv[8][2];
        float keep[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lr = (tid >> 4) + (NT / 16) * j;
            tv[j][0] = *reinterpret_cast<const f32x4 *>(T + lr * T_LD + cg * 8);
            tv[j][1] = *reinterpret_cast<const f32x4 *>(T + lr * T_LD + cg * 8 + 4);
            keep[j] = Ms[lr] ? 1.f : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        // every instruction here competes with the co-resident workgroup's MFMA stream for issue slots (an epilogue takes
        // 5-15 us of wall time for ~500 VALU instructions), so the common cases are specialised at compile time: plain ReLU
        // (no alpha term) and threads none of whose 8 rows is a gap row (no mask multiply; ~99 % of threads)
        auto rows = [&](auto MODE, auto MASKED) {
            constexpr int mode = decltype(MODE)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long gr = m0 + (tid >> 4) + (NT / 16) * j;
                bf16x8 hi, lo;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float z = (i < 4 ? tv[j][0][i] : tv[j][1][i - 4]) + bias[i];
                    const float a = mode == 1 ? fmaxf(al[i] * z, z) : mode == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + al[i] * fminf(z, 0.f);
                    float v = a * sc[i] + sh[i];
                    if constexpr (decltype(MASKED)::value) v *= keep[j];
                    hi[i] = (__bf16)v;
                    lo[i] = (__bf16)(v - (float)hi[i]);
                }
                const int sw = (int)(gr >> 1) & 7;
                char *row = ybase + (size_t)gr * yrow;
                __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8 *>(row + ((slot ^ sw) << 4)));
                __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8 *>(row + (((4 + slot) ^ sw) << 4)));
            }
        };
        const bool masked = keep[0] * keep[1] * keep[2] * keep[3] * keep[4] * keep[5] * keep[6] * keep[7] == 0.f;
        auto run = [&](auto MODE) {
            if (masked) rows(MODE, std::true_type{});
            else rows(MODE, std::false_type{});
        };
        if (lrelu) run(std::integral_constant<int, 1>{});
        else if (p.act == XV_ACT_RELU) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 0>{});
        return;
    }
    // column sums (training, see Gemm3Params::cs_part): the rows of the other factor are fetched before anything else so that
    // their latency is paid once; cout % 8 == 0 is the launcher's condition, so a column group is inside or outside as a whole
    const bool sums = p.cs_part != nullptr;             // (uniform)
    const bool sums_self = sums && p.cs_r == nullptr;   // (uniform)
    f32x4 rq[8][2];
    float cs1[8], cs2[8];
    double ds1[8], ds2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        cs1[i] = cs2[i] = 0.f;
        ds1[i] = ds2[i] = 0.0;
    }
    if (sums && !sums_self) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long gr = m0 + (tid >> 4) + (NT / 16) * j;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            rq[j][0] = rq[j][1] = zero4;
            if (gr < p.R && full) {
                const float *rr = p.cs_r + (size_t)gr * p.cs_ldr + gc0;
                rq[j][0] = *reinterpret_cast<const f32x4 *>(rr);
                rq[j][1] = *reinterpret_cast<const f32x4 *>(rr + 4);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int lr = (tid >> 4) + (NT / 16) * j;
        const long gr = m0 + lr;
        if (gr >= p.R) continue;
        const f32x4 t0 = *reinterpret_cast<const f32x4 *>(T + lr * T_LD + cg * 8);
        const f32x4 t1 = *reinterpret_cast<const f32x4 *>(T + lr * T_LD + cg * 8 + 4);
        float z[8], v[8];
        const float keep = Ms[lr] ? 1.f : 0.f;          // gap rows: one multiply per element instead of a select
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            z[i] = (i < 4 ? t0[i] : t1[i - 4]) + bias[i];
            v[i] = (activate(z[i], al[i]) * sc[i] + sh[i]) * keep;
        }
        if (sums_self) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const double d = (double)v[i];
                ds1[i] += d;
                ds2[i] = __builtin_fma(d, d, ds2[i]);
            }
        } else if (sums) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                cs1[i] += v[i];
                cs2[i] = __builtin_fmaf(v[i], i < 4 ? rq[j][0][i] : rq[j][1][i - 4], cs2[i]);
            }
        }
        if (p.ypre) {
            float *o = p.ypre + (size_t)gr * p.ldpre + gc0;
            if (full && !(p.ldpre & 3)) {
                *reinterpret_cast<f32x4 *>(o) = (f32x4){z[0], z[1], z[2], z[3]};
                *reinterpret_cast<f32x4 *>(o + 4) = (f32x4){z[4], z[5], z[6], z[7]};
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (gc0 + i < p.cout) o[i] = z[i];
            }
        }
        if (p.y) {
            if (p.y_split) {
                const int ch = gc0 >> 5;
                if (ch < p.ychunks) {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        hi[i] = (__bf16)v[i];
                        lo[i] = (__bf16)(v[i] - (float)hi[i]);
                    }
                    const int sw = (int)(gr >> 1) & 7;
                    const int slot = cg & 3;
                    char *row = reinterpret_cast<char *>(p.y) + ((size_t)gr * p.ychunks + ch) * SROW;
                    __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8 *>(row + ((slot ^ sw) << 4)));
                    __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8 *>(row + (((4 + slot) ^ sw) << 4)));
                }
            } else {
                float *o = reinterpret_cast<float *>(p.y) + (size_t)gr * p.ldy + gc0;
                if (full && !(p.ldy & 3)) {
                    __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(o));
                    __builtin_nontemporal_store((f32x4){v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4 *>(o + 4));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (gc0 + i < p.cout) o[i] = v[i];
                }
            }
        }
    }
    if (sums) {
        // a thread holds 8 rows x 8 columns (fp32); the NT / 16 row groups of a column are added in double, in group order
        constexpr int NG = NT / 16;
        __syncthreads();                                // every thread has read its rows of T
        double *D = reinterpret_cast<double *>(lds);    // [2][NG][BN]
        const int g = tid >> 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            D[(0 * NG + g) * BN + cg * 8 + i] = sums_self ? ds1[i] : (double)cs1[i];
            D[(1 * NG + g) * BN + cg * 8 + i] = sums_self ? ds2[i] : (double)cs2[i];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid >> 7, col = tid & (BN - 1);
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < NG; ++k) a += D[(which * NG + k) * BN + col];
            if (n0 + col < p.cout) p.cs_part[((size_t)mt * 2 + which) * p.cout + n0 + col] = a;
        }
    }
}

typedef void (*gemm3_fn)(const Gemm3Params);
struct Gemm3Kernel {
    int kt;       // 0: fp32-row input (runtime K), else the compile-time kernel size of the split-input path
    bool pool;
    int wm;
    gemm3_fn fn;
    bool s16 = false;
};
#define XV_G3(SPLIT, KT, POOL, WM) {KT, POOL, WM, tdnn_gemm_bf16x3_kernel<SPLIT, KT, POOL, WM>}
const Gemm3Kernel GEMM3_KERNELS[] = {
    XV_G3(false, 0, false, 2), XV_G3(false, 0, true, 2),
    XV_G3(true, 1, false, 2), XV_G3(true, 3, false, 2), XV_G3(true, 5, false, 2), XV_G3(true, 7, false, 2),
    XV_G3(true, 1, true, 2),  XV_G3(true, 3, true, 2),  XV_G3(true, 5, true, 2),  XV_G3(true, 7, true, 2),
    XV_G3(true, 1, false, 4), XV_G3(true, 3, false, 4), XV_G3(true, 5, false, 4), XV_G3(true, 7, false, 4),
    XV_G3(true, 1, true, 4),  XV_G3(true, 3, true, 4),  XV_G3(true, 5, true, 4),  XV_G3(true, 7, true, 4),
    // the 16 x 16 MFMA form (split input, K > 1)
#define XV_G3S(KT, POOL, WM) {KT, POOL, WM, tdnn_gemm_bf16x3_kernel<true, KT, POOL, WM, true>, true}
    XV_G3S(3, false, 2), XV_G3S(5, false, 2), XV_G3S(7, false, 2), XV_G3S(3, true, 2), XV_G3S(5, true, 2), XV_G3S(7, true, 2),
    XV_G3S(3, false, 4), XV_G3S(5, false, 4), XV_G3S(7, false, 4), XV_G3S(3, true, 4), XV_G3S(5, true, 4), XV_G3S(7, true, 4),
#undef XV_G3S
};
#undef XV_G3
const Gemm3Kernel *find_gemm3(int kt, bool pool, int wm, bool s16 = false)
{
    for (const Gemm3Kernel &e : GEMM3_KERNELS)
        if (e.kt == kt && e.pool == pool && e.wm == wm && e.s16 == s16) return &e;
    return nullptr;
}

// tuning knobs (xv_set_tuning); 0 = built-in choice
std::atomic<int> g_tile_rows{0};

int launch_gemm3(const Gemm3Params &p0, hipStream_t st)
{
    Gemm3Params p = p0;
    if (p.R <= 0 || p.cout <= 0) return 0;
    if (p.cin <= 0 || p.K <= 0 || (p.K & 1) == 0 || p.dil <= 0) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: K must be odd, dims > 0");
    if ((p.K - 1) * p.dil > MAX_SPAN) return fail(XV_ERR_UNSUPPORTED, "tdnn_bf16x3: (K-1)*dilation > 8 unsupported");
    if ((p.act == XV_ACT_LRELU || p.act == XV_ACT_PRELU) && !p.alpha) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: act_alpha is NULL");
    p.n_chunks = (p.cin + BK - 1) / BK;
    if (p.x_split) {
        p.xchunks = p.n_chunks;
        if (((uintptr_t)p.x) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: split input must be 16-byte aligned");
    } else {
        if (p.ldx < p.cin) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: ldx < cin");
        if ((p.cin & 3) || (p.ldx & 3) || (((uintptr_t)p.x) & 15))
            return fail(XV_ERR_UNSUPPORTED, "tdnn_bf16x3: fp32 input needs Cin and ldx multiples of 4 and a 16-byte aligned pointer");
    }
    if (p.y) {
        if (p.y_split) {
            p.ychunks = (p.cout + 31) / 32;
            if (((uintptr_t)p.y) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: split output must be 16-byte aligned");
        } else if (p.ldy < p.cout || (((uintptr_t)p.y) & 15)) {
            return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: ldy < cout or output not 16-byte aligned");
        }
    }
    if (p.ypre && (p.ldpre < p.cout || (((uintptr_t)p.ypre) & 15))) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: bad y_preact");
    if (((uintptr_t)p.wt) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: packed weights must be 16-byte aligned");
    p.n_nt = (p.cout + BN - 1) / BN;
    int kt = 0;
    if (p.x_split) {
        kt = p.K;
        const int span = (p.K - 1) * p.dil;
        if ((kt != 1 && kt != 3 && kt != 5 && kt != 7) || (kt > 1 && (span < 2 || span > MAX_SPAN)))
            return fail(XV_ERR_UNSUPPORTED, "tdnn_bf16x3: split-format input supports K in {1,3,5,7} with (K-1)*dilation <= 8");
    }
    // workgroup tile: 256 rows (8 waves, one workgroup per CU) for the wide-context layers when the input is in the split
    // format and there are enough rows to fill the chip with such tiles, else 128 rows (4 waves, two per CU);
    // xv_set_tuning(XV_TUNE_TILE_ROWS) overrides
    // the 16 x 16 MFMA form where it exists: split input, K > 1, an even number of 32-channel slabs (XV_BF16X3_S16=0: off)
    static const bool s16_on = !(std::getenv("XV_BF16X3_S16") != nullptr && std::getenv("XV_BF16X3_S16")[0] == '0');
    const bool s16 = s16_on && p.x_split && kt > 1 && (p.n_chunks & 1) == 0;
    int wm = 2;
    if (p.x_split) {
        const int want = g_tile_rows.load(std::memory_order_relaxed);
        // measured on 262144-row batches (tools/layer_bench.py, profiles/r02a_layer_tile.txt): K = 7 +2.7 %, K = 5 +1.2 %,
        // K = 1 -1.5 ... -3.5 % (with one workgroup per CU the prologue and epilogue of a 16-stage tile are exposed).  The
        // 16 x 16 form is faster on 128-row tiles (K = 5 1.306 against 1.357 ms, K = 7 1.759 against 1.773)
        const bool big_enough = ((p.R + 255) / 256) * p.n_nt >= 512;          // two rounds of 256 CUs
        if (want == 256 || (want == 0 && p.K >= 5 && big_enough && !s16)) wm = 4;
    }
    if (p.cs_part) wm = 2;                                   // one partial per 128 rows: the split xv_col_sums_merge_f32 walks
    p.n_mt = (int)((p.R + wm * 64 - 1) / (wm * 64));
    const Gemm3Kernel *k = find_gemm3(kt, p.blk != nullptr, wm, s16);
    if (!k) return fail(XV_ERR_UNSUPPORTED, "tdnn_bf16x3: no kernel for this configuration");
    // the dynamic-LDS opt-in is per device and idempotent: one bit per device id, set after the first successful pass
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (const Gemm3Kernel &e : GEMM3_KERNELS) {
            hipError_t err = hipFuncSetAttribute((const void *)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm3_lds_bytes(e.wm));
            if (err != hipSuccess) return hip_fail(err, "hipFuncSetAttribute");
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(k->fn, dim3((unsigned)(p.n_mt * p.n_nt)), dim3(wm * 128), gemm3_lds_bytes(wm), st, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "tdnn_gemm_bf16x3_kernel launch");
}

// ------------------------------------------------------------------------------------------------
// Exact-fp32 GEMM, DMA-fed (round 4): the arithmetic of tdnn_gemm_kernel -- the same v_mfma_f32_32x32x2_f32 sequence on the same
// fragments in the same order, hence bit-identical results -- in the form of the 16-bit kernels:
//  * both operands go global -> LDS by buffer_load ... lds straight from the fp32 rows x[R, Cin] and the packed weights wp[Cout, K Cin]
//    as they lie (no staging registers, no ds_write, no exec-masked blocks of bounds logic: rows and columns outside the matrices
//    come back as zeros from the buffer descriptors' range check);
//  * an LDS row is 128 bytes (32 channels), unpadded; the 16-byte slot s of row r holds channel group s ^ ((r >> 1) & 7) -- the
//    swizzle is applied on the GLOBAL side of the DMA (lane (row, slot) of an 8-row piece fetches group slot ^ swz), fragment reads
//    are conflict-free for every tap offset;
//  * K is a template constant, the stage loop is unrolled over the taps of a slab; ONE barrier per stage, placed in front of the last
//    quarter of the stage's MFMAs: behind it the DMA of stage s + 2 goes out and the first fragments of stage s + 1 are read under the
//    remaining 16 MFMAs (two fragment sets, k-group granularity) -- no wave starts a stage with an empty pipe.
// Needs Cin % 32 == 0, 16-byte aligned rows, matrices below 2^31 bytes, the row-wise epilogue (else tdnn_gemm_kernel).
// ------------------------------------------------------------------------------------------------
#define XV_BLDS16(rsrc, lptr, voff, soff, imm)                                                                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lptr), 16, voff, soff, imm, 0)
constexpr int XV_RSRC_FLAGS = 0x00020000;              // raw buffer, 32-bit data format (gfx9 family dword 3)
constexpr int F_SROW = 128;
constexpr int F_A_BYTES = (BM + MAX_SPAN) * F_SROW;    // 17408
constexpr int F_B_BYTES = BN * F_SROW;                 // 16384
constexpr int F_OPER = 2 * F_A_BYTES + 2 * F_B_BYTES;  // 67584 = the epilogue's 128 x 132 fp32 tile
constexpr size_t F_LDS_BYTES = (size_t)F_OPER + BM;

template <int I, int N, class F>
__device__ __forceinline__ void f_static_for(F &f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        f_static_for<I + 1, N>(f);
    }
}

template <int KT>
__global__ __launch_bounds__(NT, 2) void tdnn_gemm_dma_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char flds[];
    char *Abuf = flds;                                 // [2][BM + 8 rows][128 B]
    char *Bbuf = flds + 2 * F_A_BYTES;                 // [2][BN cols][128 B]
    uint8_t *Ms = reinterpret_cast<uint8_t *>(flds + F_OPER);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int nwg = p.n_mt * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = wg / p.n_nt, nt = wg - mt * p.n_nt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int span = (KT - 1) * p.dil;
    const int left = (span >> 1) + p.lead;
    const int n_chunks = p.cin / BK;
    const int n_stages = n_chunks * KT;

    if (tid < BM) {
        const long gr = m0 + tid;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    }

    // ---- DMA: piece pc = 8 rows (columns) x 128 bytes; wave w moves pieces w, w + 4, ...: their parity is the wave's, and with it
    // bit 2 of (row >> 1) & 7 -- the swizzle of the row a lane moves is a per-lane constant.  The row part of an address sits in
    // the VGPR offset (what the range check looks at), the channel / tap part in the scalar offset.
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)(p.R * p.ldx * 4), XV_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wp), 0, (int)((long)p.cout * p.kred * 4), XV_RSRC_FLAGS);
    const int slotb = ((lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7)) << 4;
    const int arow_bytes = p.ldx * 4, brow_bytes = p.kred * 4;
    const int va0 = (int)(m0 - left + 8 * wave + (lane >> 3)) * arow_bytes + slotb;        // piece `wave`; piece wave + 4 j: + 32 j rows
    const int vb0 = (n0 + 8 * wave + (lane >> 3)) * brow_bytes + slotb;
    constexpr int NP = KT == 1 ? BM / 8 : BM / 8 + 1;   // pieces of a halo tile (BM + span rows, span <= 8)
    auto dma_b = [&](int stage, int buf) {              // the weight tile of (slab, tap) = stage, four pieces per wave
        const int st = stage < n_stages ? stage : n_stages - 1;
        const int c = st / KT, t = st - c * KT;
        const int so = (t * p.cin + c * BK) * 4;
        char *dst = Bbuf + buf * F_B_BYTES + wave * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) XV_BLDS16(brs, dst + j * 4096, vb0 + j * 32 * brow_bytes, so, 0);
    };
    auto dma_a_piece = [&](int chunk, int j) {          // piece wave + 4 j of slab `chunk` (clamped: the tail rewrites identical bytes)
        const int c = chunk < n_chunks ? chunk : n_chunks - 1;
        XV_BLDS16(ars, Abuf + (c & 1) * F_A_BYTES + (wave + 4 * j) * 1024, va0 + j * 32 * arow_bytes, c * BK * 4, 0);
    };
    auto dma_a_all = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_a_piece(chunk, j);
        if (NP > 16 && wave == 0) dma_a_piece(chunk, 4);
    };
    dma_b(0, 0);
    dma_b(1, 1);
    dma_a_all(0);
    if (KT == 1) dma_a_all(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    struct Fr { f32x4 a0, a1, b0, b1; };
    // fragment addresses: lane (row l & 31, k half kh = l >> 5) reads channels 8 kk + 4 kh .. + 3 = slot 2 kk + kh of its row
    const int kh = lane >> 5;
    int pa[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int lr0 = wr * 64 + (lane & 31) + t * p.dil;
        pa[t] = lr0 * F_SROW + ((((lr0 >> 1) & 7) ^ kh) << 4);          // + 32 rows: + 4096, the same swizzle; k group kk: ^ (kk << 5)
    }
    const int bcol = wc * 64 + (lane & 31);
    const int pb = 2 * F_A_BYTES + bcol * F_SROW + ((((bcol >> 1) & 7) ^ kh) << 4);
    auto load = [&](Fr &X, int abase, int bbase, int kk) {
        X.a0 = *reinterpret_cast<const f32x4 *>(flds + (abase ^ (kk << 5)));
        X.a1 = *reinterpret_cast<const f32x4 *>(flds + (abase ^ (kk << 5)) + 32 * F_SROW);
        X.b0 = *reinterpret_cast<const f32x4 *>(flds + (bbase ^ (kk << 5)));
        X.b1 = *reinterpret_cast<const f32x4 *>(flds + (bbase ^ (kk << 5)) + 32 * F_SROW);
    };
    auto mma = [&](const Fr &X) {                        // (the order of tdnn_gemm_kernel: results are bit-identical)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a0[j], X.b0[j], acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a0[j], X.b1[j], acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a1[j], X.b0[j], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a1[j], X.b1[j], acc11, 0, 0, 0);
        }
    };
    auto pin = [&]() {                                   // 16 MFMAs, the four fragment reads behind the first four
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // halo pieces of the NEXT slab, dealt over taps 0 .. K-2 of the current one (K > 1): five one-piece-per-wave slots
    constexpr int DT = KT == 1 ? 1 : KT - 1;
    constexpr int NS = KT == 1 ? 4 : 5;
    auto slots_of = [](int t) constexpr { return t < DT ? (NS + DT - 1 - t) / DT : 0; };
    auto slot_base = [](int t) constexpr { int b = 0; for (int u = 0; u < t; ++u) b += (NS + DT - 1 - u) / DT; return b; };

    Fr F, G;
    load(F, pa[0], pb, 0);
    int s = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const int abuf = (c & 1) * F_A_BYTES;
        auto tap = [&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int ab = pa[t] + abuf, bb = pb + (s & 1) * F_B_BYTES;
            load(G, ab, bb, 1);
            mma(F);
            pin();
            load(F, ab, bb, 2);
            mma(G);
            pin();
            load(G, ab, bb, 3);
            mma(F);
            pin();
            // every fragment of stage s is in registers (the reads of k group 3 went out 1024 MFMA cycles ago), stage s + 1 has landed
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma_b(s + 2, s & 1);
            if constexpr (KT == 1) {
                dma_a_all(s + 2);
            } else if constexpr (t < DT) {
#pragma unroll
                for (int j = 0; j < slots_of(t); ++j) {
                    const int jj = slot_base(t) + j;
                    if (jj < 4 || wave == 0) dma_a_piece(c + 1, jj);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t + 1 < KT) load(F, pa[t + 1] + abuf, pb + ((s + 1) & 1) * F_B_BYTES, 0);
            else load(F, pa[0] + (F_A_BYTES - abuf), pb + ((s + 1) & 1) * F_B_BYTES, 0);      // first tap of the next slab (tail: harmless)
            mma(G);
            pin();
            ++s;
        };
        f_static_for<0, KT>(tap);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the tail's clamped pieces must have landed before the tile below reuses the LDS)
    __syncthreads();
    gemm_epilogue_rows<BM>(p, reinterpret_cast<float *>(flds), Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11);
}

// ------------------------------------------------------------------------------------------------
// The K = 1 layers of the exact-fp32 path on 16-CHANNEL slabs (round 6): the same MFMA sequence on the same fragments in the same
// order as tdnn_gemm_dma_kernel<1> (bit-identical), but an LDS row is 64 bytes, a stage 8 + 8 KB, and the epilogue goes through a
// 64-row tile in two passes -- 34 KB of LDS per workgroup instead of 68, THREE workgroups per CU instead of two (four fit the LDS; at 128 VGPRs the epilogue spills).  A K = 1 tile is
// 16 (here 32) stages and pays ~2.1 of the old stages at its boundary (epilogue, then the first DMA's latency; fitted from K = 1 / 5 / 7:
// DESIGN 9.6) with ONE other workgroup on the CU to cover it; a persistent launch changed nothing (profiles/r06_fp32_persistent.txt),
// two other workgroups do: 0.85 -> 0.89-0.91 of the fp32-MFMA peak.
//  * piece = 16 rows x 64 bytes; lane l of a piece moves row l >> 2, 16-byte slot l & 3; physical slot = logical ^ ((row >> 2) & 3),
//    applied on the global side of the DMA: the 16 lanes of a ds_read_b128 group (16 consecutive rows, one logical slot) hit 16
//    different bank quads;
//  * stage = 2 k groups = 32 MFMAs per wave, one barrier per stage, the DMA of stage s + 2 behind it.
// ------------------------------------------------------------------------------------------------
constexpr int G_BK = 16;
constexpr int G_SROW = 64;
constexpr int G_A_BYTES = BM * G_SROW;                 // 8192
constexpr int G_B_BYTES = BN * G_SROW;                 // 8192
constexpr int G_TROWS = 64;                            // the epilogue's tile: two passes of 64 rows (33.8 KB; three or four workgroups per CU)
constexpr int G_TILE = G_TROWS * (BN + 4) * 4;
constexpr int G_OPER = 2 * G_A_BYTES + 2 * G_B_BYTES;  // 32768
constexpr int G_MS = G_TILE > G_OPER ? G_TILE : G_OPER;
constexpr size_t G_LDS_BYTES = (size_t)G_MS + BM;

template <int ACT>
__device__ __forceinline__ void k1_epilogue(const GemmParams &p, char *lds, const uint8_t *Ms, long m0, int n0, int wr, int wc, int tid,
                                            const f32x16 &acc00, const f32x16 &acc01, const f32x16 &acc10, const f32x16 &acc11)
{
    constexpr int TLD = BN + 4;
    float *T = reinterpret_cast<float *>(lds);
    const int lane = tid & 63;
    const int col = wc * 64 + (lane & 31);
#pragma unroll
    for (int pass = 0; pass < BM / G_TROWS; ++pass) {
        if (wr == pass) {                                  // the waves that own rows [64 pass, 64 pass + 64)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = 4 * (lane >> 5) + (reg & 3) + 8 * (reg >> 2);
                T[lr * TLD + col] = acc00[reg];
                T[lr * TLD + col + 32] = acc01[reg];
                T[(lr + 32) * TLD + col] = acc10[reg];
                T[(lr + 32) * TLD + col + 32] = acc11[reg];
            }
        }
        __syncthreads();
        gemm_epilogue_rows_body<G_TROWS, ACT>(p, T, Ms + G_TROWS * pass, m0 + G_TROWS * pass, n0, tid);
        __syncthreads();
    }
}

template <int OCC>                                     // workgroups per CU the register budget is cut for (3: 168 VGPRs, no spills; 4: 128, the epilogue spills)
__global__ __launch_bounds__(NT, OCC) void tdnn_gemm_k1_kernel(const GemmParams p)
{
    extern __shared__ __attribute__((aligned(16))) char glds[];
    char *Abuf = glds;                                 // [2][BM rows][64 B]
    char *Bbuf = glds + 2 * G_A_BYTES;                 // [2][BN cols][64 B]
    uint8_t *Ms = reinterpret_cast<uint8_t *>(glds + G_MS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int nwg = p.n_mt * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = wg / p.n_nt, nt = wg - mt * p.n_nt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    const int n_stages = p.cin / G_BK;

    if (tid < BM) {
        const long gr = m0 + tid;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    }

    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)(p.R * p.ldx * 4), XV_RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wp), 0, (int)((long)p.cout * p.kred * 4), XV_RSRC_FLAGS);
    // piece pc = 16 rows (columns); wave w moves pieces w and w + 4 of each operand and stage
    const int slotb = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const int arow_bytes = p.ldx * 4, brow_bytes = p.kred * 4;
    const int va0 = (int)(m0 - p.lead + 16 * wave + (lane >> 2)) * arow_bytes + slotb;
    const int vb0 = (n0 + 16 * wave + (lane >> 2)) * brow_bytes + slotb;
    auto dma = [&](int stage, int buf) {
        const int st = stage < n_stages ? stage : n_stages - 1;          // (the tail rewrites identical bytes)
        const int so = st * G_BK * 4;
        char *da = Abuf + buf * G_A_BYTES + wave * 1024, *db = Bbuf + buf * G_B_BYTES + wave * 1024;
        XV_BLDS16(ars, da, va0, so, 0);
        XV_BLDS16(ars, da + 4096, va0 + 64 * arow_bytes, so, 0);
        XV_BLDS16(brs, db, vb0, so, 0);
        XV_BLDS16(brs, db + 4096, vb0 + 64 * brow_bytes, so, 0);
    };
    dma(0, 0);
    dma(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    struct Fr { f32x4 a0, a1, b0, b1; };
    // lane (row l & 31, k half kh = l >> 5) reads channels 8 kk + 4 kh .. + 3 of the slab = logical slot 2 kk + kh of its row
    const int kh = lane >> 5;
    const int ar = wr * 64 + (lane & 31), bc = wc * 64 + (lane & 31);
    const int pa = ar * G_SROW + ((kh ^ ((ar >> 2) & 3)) << 4);          // + 32 rows: + 2048, the same swizzle; k group 1: ^ 32
    const int pb = 2 * G_A_BYTES + bc * G_SROW + ((kh ^ ((bc >> 2) & 3)) << 4);
    auto load = [&](Fr &X, int buf, int kk) {
        const int ab = (pa ^ (kk << 5)) + buf * G_A_BYTES, bb = (pb ^ (kk << 5)) + buf * G_B_BYTES;
        X.a0 = *reinterpret_cast<const f32x4 *>(glds + ab);
        X.a1 = *reinterpret_cast<const f32x4 *>(glds + ab + 32 * G_SROW);
        X.b0 = *reinterpret_cast<const f32x4 *>(glds + bb);
        X.b1 = *reinterpret_cast<const f32x4 *>(glds + bb + 32 * G_SROW);
    };
    auto mma = [&](const Fr &X) {                        // (the order of tdnn_gemm_kernel / tdnn_gemm_dma_kernel: results are bit-identical)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a0[j], X.b0[j], acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a0[j], X.b1[j], acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a1[j], X.b0[j], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(X.a1[j], X.b1[j], acc11, 0, 0, 0);
        }
    };
    auto pin = [&]() {                                   // 16 MFMAs, the four fragment reads behind the first four
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    Fr F, G;
    load(F, 0, 0);
    for (int s = 0; s < n_stages; ++s) {
        load(G, s & 1, 1);
        mma(F);
        pin();
        // both fragment sets of stage s are in registers, stage s + 1 has landed.  (A third stage buffer -- the DMA two stages ahead,
        // counted waits -- was measured: +-0, profiles/r06_fp32_k1_slab16.txt.)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        dma(s + 2, s & 1);
        __builtin_amdgcn_sched_barrier(0);
        load(F, (s + 1) & 1, 0);                         // (tail: harmless)
        mma(G);
        pin();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the tail's clamped pieces must have landed before the tile below reuses the LDS)
    __syncthreads();
    switch (p.act) {                                      // (uniform: one switch per thread, not one per element)
    case XV_ACT_RELU: k1_epilogue<XV_ACT_RELU>(p, glds, Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11); break;
    case XV_ACT_LRELU: k1_epilogue<XV_ACT_LRELU>(p, glds, Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11); break;
    case XV_ACT_PRELU: k1_epilogue<XV_ACT_PRELU>(p, glds, Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11); break;
    default: k1_epilogue<XV_ACT_NONE>(p, glds, Ms, m0, n0, wr, wc, tid, acc00, acc01, acc10, acc11); break;
    }
}

constexpr bool FP32_K1_DEFAULT = true;                // (profiles/r06_fp32_k1_slab16.txt: 1.03 -> 0.98 ms, 3.06 -> 2.90 ms; XV_FP32_K1=0 turns it off)
std::atomic<int> g_fp32_form{0};

int launch_gemm(const GemmParams &p0, hipStream_t st)
{
    GemmParams p = p0;
    if (p.R <= 0 || p.cout <= 0) return 0;
    if (p.cin <= 0 || p.K <= 0 || (p.K & 1) == 0 || p.dil <= 0) return fail(XV_ERR_BAD_ARG, "tdnn: K must be odd, dims > 0");
    if ((p.K - 1) * p.dil > MAX_SPAN) return fail(XV_ERR_UNSUPPORTED, "tdnn: (K-1)*dilation > 8 unsupported");
    if ((p.lead == 0 && p.ldx < p.cin) || p.ldy < p.cout) return fail(XV_ERR_BAD_ARG, "tdnn: leading dimension too small");
    if ((p.act == XV_ACT_LRELU || p.act == XV_ACT_PRELU) && !p.alpha) return fail(XV_ERR_BAD_ARG, "tdnn: act_alpha is NULL");
    p.kred = p.K * p.cin;
    p.n_nt = (p.cout + BN - 1) / BN;
    // 64-row tiles when 128-row tiles would leave the chip with fewer than 1.5 rounds of resident workgroups
    const bool small = ((p.R + BM - 1) / BM) * p.n_nt < 768;
    const int bmt = small ? 64 : BM;
    p.n_mt = (int)((p.R + bmt - 1) / bmt);
    const bool vec = (p.cin % 4 == 0) && (p.ldx % 4 == 0) && (((uintptr_t)p.x) % 16 == 0) && (((uintptr_t)p.wp) % 16 == 0);
    if (p.lead && (!vec || p.K != 1 || p.ypre)) return fail(XV_ERR_UNSUPPORTED, "tdnn_rows: needs 16-byte aligned rows of a multiple of 4 floats");
    const uintptr_t out_bits = (uintptr_t)p.y | (uintptr_t)p.ypre | (uintptr_t)p.bias | (uintptr_t)p.scale | (uintptr_t)p.shift |
                               (p.act == XV_ACT_PRELU ? (uintptr_t)p.alpha : 0);
    p.vec_out = (p.cout % 4 == 0) && (p.ldy % 4 == 0) && (out_bits % 16 == 0) && (p.blk || std::getenv("XV_FP32_SCALAR_EPILOGUE") == nullptr);
    if (p.blk && !p.vec_out) return fail(XV_ERR_UNSUPPORTED, "tdnn_pool: needs cout % 4 == 0 and 16-byte aligned per-column parameters");
    typedef void (*kern_t)(const GemmParams);
    const kern_t all[] = {tdnn_gemm_kernel<true, 128>, tdnn_gemm_kernel<false, 128>, tdnn_gemm_kernel<true, 64>,
                          tdnn_gemm_kernel<false, 64>};
    const kern_t dma_all[] = {tdnn_gemm_dma_kernel<1>, tdnn_gemm_dma_kernel<3>, tdnn_gemm_dma_kernel<5>, tdnn_gemm_dma_kernel<7>};
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (kern_t k : all) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        }
        for (kern_t k : dma_all) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        }
        for (kern_t k : {tdnn_gemm_k1_kernel<3>, tdnn_gemm_k1_kernel<4>}) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G_LDS_BYTES);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const dim3 grid((unsigned)(p.n_mt * p.n_nt));
    // the DMA-fed form (128-row tiles): whole 32-channel slabs, 16-byte aligned rows, byte offsets that fit the descriptors' 32 bits
    static const bool dma_env = !(std::getenv("XV_FP32_DMA") != nullptr && std::getenv("XV_FP32_DMA")[0] == '0');
    const int form = g_fp32_form.load(std::memory_order_relaxed);       // XV_TUNE_FP32_GEMM: 0 built-in, 1 register-staged, 2 DMA-fed, 3 DMA-fed with the K = 1 layers on 16-channel slabs
    const bool dma_on = form >= 2 || (form == 0 && dma_env);
    static const int k1_env = std::getenv("XV_FP32_K1") ? atoi(std::getenv("XV_FP32_K1")) : -1;
    const bool k1_on = form == 3 || (form == 0 && (k1_env < 0 ? FP32_K1_DEFAULT : k1_env != 0));
    const bool dma_ok = dma_on && !small && p.k_splits <= 1 && vec && p.vec_out && (p.cin % BK) == 0 && (p.K == 1 || p.K == 3 || p.K == 5 || p.K == 7) &&
                        (p.R + BM + MAX_SPAN) * (long)p.ldx * 4 < (1l << 31) && (long)(p.cout + BN) * p.kred * 4 < (1l << 31);
    if (dma_ok && k1_on && p.K == 1 && p.cin >= 2 * G_BK) {
        if (k1_env == 4) hipLaunchKernelGGL(tdnn_gemm_k1_kernel<4>, grid, dim3(NT), G_LDS_BYTES, st, p);
        else hipLaunchKernelGGL(tdnn_gemm_k1_kernel<3>, grid, dim3(NT), G_LDS_BYTES, st, p);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : hip_fail(e, "tdnn_gemm_k1_kernel launch");
    }
    if (dma_ok) {
        const kern_t dk = p.K == 1 ? tdnn_gemm_dma_kernel<1> : p.K == 3 ? tdnn_gemm_dma_kernel<3> : p.K == 5 ? tdnn_gemm_dma_kernel<5>
                                                                                                             : tdnn_gemm_dma_kernel<7>;
        hipLaunchKernelGGL(dk, grid, dim3(NT), F_LDS_BYTES, st, p);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : hip_fail(e, "tdnn_gemm_dma_kernel launch");
    }
    if (p.k_splits > 1) {
        hipLaunchKernelGGL(all[(small ? 2 : 0) + (vec ? 0 : 1)], dim3((unsigned)(p.n_mt * p.n_nt * p.k_splits)), dim3(NT), GEMM_LDS_BYTES, st, p);
        hipError_t e2 = hipGetLastError();
        return e2 == hipSuccess ? 0 : hip_fail(e2, "tdnn_gemm_kernel (split-K) launch");
    }
    hipLaunchKernelGGL(all[(small ? 2 : 0) + (vec ? 0 : 1)], grid, dim3(NT), GEMM_LDS_BYTES, st, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "tdnn_gemm_kernel launch");
}

// ------------------------------------------------------------------------------------------------
// statistics pooling
// ------------------------------------------------------------------------------------------------
struct Stat4 {
    f32x4 mean, m2;
    float n;
};

// merge a block of `m` values per channel given as (block mean, block M2) into the running stats
__device__ __forceinline__ void chan_merge(Stat4 &s, const f32x4 bmean, const f32x4 bm2, float m)
{
    const float nn = s.n + m;
    if (nn > 0.f) {
        const float w = m / nn;
        const f32x4 d = bmean - s.mean;
        s.mean += d * w;
        s.m2 += bm2 + d * d * (s.n * w);
        s.n = nn;
    }
}

constexpr int POOL_UNROLL = 8;

// One wave64 per (chunk b, time split sp, 64-channel group).  lane = (phase = lane>>4 : which of 4
// interleaved rows, cg = lane&15 : which float4 of the 64 channels).  A wave instruction therefore
// reads 4 rows x 256 contiguous bytes; a 4-wave workgroup covers a 256-channel slab.
__global__ __launch_bounds__(256) void stats_pool_kernel(const float *__restrict__ h, long ldh, int C,
                                                         const int *__restrict__ row_start,
                                                         const int *__restrict__ row_len, int split_rows,
                                                         int max_splits, float eps, float *__restrict__ out,
                                                         float *__restrict__ partial, int raw)
{
    const int b = blockIdx.z, sp = blockIdx.y;
    const int len = row_len[b];
    const int begin = sp * split_rows;
    if (begin >= len) return;
    const int n_rows = min(split_rows, len - begin);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int phase = lane >> 4;
    const int c = (blockIdx.x * 4 + wave) * 64 + (lane & 15) * 4;
    if (c >= C) return;          // C % 4 == 0: whole float4 in or out (lanes of other phases agree)
    const float *base = h + ((size_t)row_start[b] + begin) * ldh + c;

    Stat4 s;
    s.mean = (f32x4){0.f, 0.f, 0.f, 0.f};
    s.m2 = s.mean;
    s.n = 0.f;

    int r = phase;
    // full blocks: 8 rows per lane (rows r, r+4, ..., r+28)
    for (; r + 4 * (POOL_UNROLL - 1) < n_rows; r += 4 * POOL_UNROLL) {
        f32x4 v[POOL_UNROLL];
#pragma unroll
        for (int i = 0; i < POOL_UNROLL; ++i)
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (size_t)(r + 4 * i) * ldh));
        // block statistics about v[0] (shifted): exact for constant channels, no cancellation
        f32x4 d[POOL_UNROLL];
        f32x4 sumd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 1; i < POOL_UNROLL; ++i) {
            d[i] = v[i] - v[0];
            sumd += d[i];
        }
        const f32x4 md = sumd * (1.0f / POOL_UNROLL);
        const f32x4 bm = v[0] + md;
        f32x4 m2 = md * md;                 // element 0: (0 - md)^2
#pragma unroll
        for (int i = 1; i < POOL_UNROLL; ++i) {
            const f32x4 e = d[i] - md;
            m2 += e * e;
        }
        chan_merge(s, bm, m2, (float)POOL_UNROLL);
    }
    // tail: fewer than 8 rows left for this lane
    if (r < n_rows) {
        f32x4 v[POOL_UNROLL];
        int m = 0;
#pragma unroll
        for (int i = 0; i < POOL_UNROLL; ++i) {
            const bool ok = (r + 4 * i) < n_rows;
            v[i] = ok ? *reinterpret_cast<const f32x4 *>(base + (size_t)(r + 4 * i) * ldh) : (f32x4){0.f, 0.f, 0.f, 0.f};
            m += ok ? 1 : 0;
        }
        const float fm = (float)m;          // m >= 1: v[0] is always a real row
        f32x4 sumd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 1; i < POOL_UNROLL; ++i)
            if ((r + 4 * i) < n_rows) sumd += v[i] - v[0];
        const f32x4 md = sumd / fm;
        const f32x4 bm = v[0] + md;
        f32x4 m2 = md * md;
#pragma unroll
        for (int i = 1; i < POOL_UNROLL; ++i) {
            const f32x4 e = (v[i] - v[0]) - md;
            if ((r + 4 * i) < n_rows) m2 += e * e;
        }
        chan_merge(s, bm, m2, fm);
    }

    // combine the 4 row phases of the wave: shuffle-xor 16 then 32 (Chan merge of (n, mean, M2))
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        Stat4 o;
        o.n = __shfl_xor(s.n, off, 64);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o.mean[i] = __shfl_xor(s.mean[i], off, 64);
            o.m2[i] = __shfl_xor(s.m2[i], off, 64);
        }
        chan_merge(s, o.mean, o.m2, o.n);
    }
    if (phase != 0) return;
    if (max_splits == 1) {
        const f32x4 var = s.m2 / s.n;
        f32x4 sd;
#pragma unroll
        for (int i = 0; i < 4; ++i) sd[i] = raw ? var[i] : sqrtf(var[i] + eps);      // raw: (mean, biased variance)
        float *o = out + (size_t)b * 2 * C;
        *reinterpret_cast<f32x4 *>(o + c) = s.mean;
        *reinterpret_cast<f32x4 *>(o + C + c) = sd;
    } else {
        float *pm = partial + ((size_t)b * max_splits + sp) * 2 * C;
        *reinterpret_cast<f32x4 *>(pm + c) = s.mean;
        *reinterpret_cast<f32x4 *>(pm + C + c) = s.m2;
    }
}

// second stage for split chunks: merge the per-split (mean, M2) in split order, finalize
__global__ void stats_pool_merge_kernel(const float *__restrict__ partial, int C, const int *__restrict__ row_len,
                                        int split_rows, int max_splits, float eps, float *__restrict__ out, int raw)
{
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int len = row_len[b];
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int sp = 0; sp * split_rows < len; ++sp) {
        const float m = (float)min(split_rows, len - sp * split_rows);
        const float *pm = partial + ((size_t)b * max_splits + sp) * 2 * C;
        const float bm = pm[c], bm2 = pm[C + c];
        const float nn = n + m;
        const float w = m / nn;
        const float d = bm - mean;
        mean += d * w;
        m2 += bm2 + d * d * (n * w);
        n = nn;
    }
    out[(size_t)b * 2 * C + c] = mean;
    out[(size_t)b * 2 * C + C + c] = raw ? m2 / n : sqrtf(m2 / n + eps);
}

// finalize for the POOL epilogue of the bf16x3 GEMM: chunk b = the 8-row blocks row_start[b]/8 ... in order (all full but
// the last); per channel  mean = sum n_i*mean_i / N,  var = sum(M2_i + n_i*mean_i^2)/N - mean^2  in fp64 (the inputs are
// fp32, so the subtraction loses nothing that matters), out = [mean | sqrt(var + eps)].  A chunk that does not start on a
// multiple of 8 rows was not reduced block-wise by the epilogue: its outputs are set to NaN.
__global__ void stats_pool_blocks_kernel(const float *__restrict__ blk, int C, const int *__restrict__ row_start,
                                         const int *__restrict__ row_len, float eps, float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int rs = row_start[b], len = row_len[b];
    float *o = out + (size_t)b * 2 * C;
    if ((rs & 7) || len <= 0) {
        o[c] = __builtin_nanf("");
        o[C + c] = __builtin_nanf("");
        return;
    }
    const float *pb = blk + (size_t)(rs >> 3) * 2 * C + c;
    const int nb = (len + 7) >> 3;
    double S = 0.0, Q = 0.0;
#pragma unroll 4
    for (int i = 0; i < nb; ++i) {
        const double m = (double)__builtin_nontemporal_load(pb + (size_t)i * 2 * C);
        const double m2 = (double)__builtin_nontemporal_load(pb + (size_t)i * 2 * C + C);
        const double n = (double)min(8, len - 8 * i);
        S += n * m;
        Q += m2 + n * m * m;
    }
    const double mean = S / (double)len;
    const double var = fmax(Q / (double)len - mean * mean, 0.0);
    o[c] = (float)mean;
    o[C + c] = sqrtf((float)var + eps);
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
// float32 op order of NumPy in local/tf/models.py:418-421: p = len*e (rounded), acc += p (rounded),
// acc /= total.  __fmul_rn/__fadd_rn/__fdiv_rn forbid FMA contraction.
__global__ void chunk_average_kernel(const float *__restrict__ e, const int *__restrict__ seg_start,
                                     const int *__restrict__ chunk_len, int dim, float *__restrict__ out)
{
#pragma clang fp contract(off)      // hipcc contracts a*b+c into fma by default; NumPy does not
    const int u = blockIdx.y;
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= dim) return;
    const int s0 = seg_start[u], s1 = seg_start[u + 1];
    float acc = 0.f;
    double tot = 0.0;
    for (int i = s0; i < s1; ++i) {
        const float w = (float)chunk_len[i];
        float prod = w * e[(size_t)i * dim + d];
        asm volatile("" : "+v"(prod));      // opaque to the optimiser: product is rounded before the add
        acc = acc + prod;
        tot += (double)chunk_len[i];
    }
    out[(size_t)u * dim + d] = acc / (float)tot;      // IEEE-correct fp32 division (hipcc default)
}

__global__ void pack_weights_kernel(const float *__restrict__ w, int kred, int cout, float *__restrict__ wp)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)kred * cout) return;
    const int n = (int)(i / kred), k = (int)(i - (size_t)n * kred);
    wp[i] = w[(size_t)k * cout + n];
}

// w[K, cin, cout] -> wp[cout][kpad] for the rows form: column k * ldx + c holds w[k][c][o] (c < cin), every other column zero
__global__ void pack_weights_rows_kernel(const float *__restrict__ w, int K, int cin, int ldx, int cout, int kpad, float *__restrict__ wp)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)kpad * cout) return;
    const int o = (int)(i / kpad), col = (int)(i - (size_t)o * kpad);
    const int k = col / ldx, c = col - k * ldx;
    wp[i] = (k < K && c < cin) ? w[((size_t)k * cin + c) * cout + o] : 0.f;
}

// w[K, cin, cout] fp32 -> tiled bf16x3 weights: tile (nt, chunk, tap) = 16 KB [hi 128x64B][lo 128x64B], slots swizzled
__global__ void pack_weights_bf16x3_kernel(const float *__restrict__ w, int K, int cin, int cout, int n_chunks,
                                           uint8_t *__restrict__ wt, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (tile, col, k) element
    if (i >= total) return;
    const int k = (int)(i & 31);
    const int n = (int)((i >> 5) & 127);
    const size_t tile = i >> 12;
    const int tap = (int)(tile % K);
    const int chunk = (int)((tile / K) % n_chunks);
    const int nt = (int)(tile / ((size_t)K * n_chunks));
    const int c = chunk * 32 + k, gn = nt * 128 + n;
    const float x = (c < cin && gn < cout) ? w[((size_t)tap * cin + c) * cout + gn] : 0.f;
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    uint8_t *t = wt + tile * B3_BYTES + n * 64 + (((k >> 3) ^ ((n >> 2) & 3)) << 4) + (k & 7) * 2;
    *reinterpret_cast<uint16_t *>(t) = __builtin_bit_cast(uint16_t, hi);
    *reinterpret_cast<uint16_t *>(t + B3_PLANE) = __builtin_bit_cast(uint16_t, lo);
}

// The same tiles for MANY layers in one launch, each in one or both of two orientations (the training step re-packs every
// weight after every optimizer step: sixteen launches and as many torch flip / permute / cat kernels per step before this):
//   forward   wt_fwd = pack(w[K, cin_pad, cout])                                    (columns cin .. cin_pad-1 read as zero)
//   backward  wt_bwd = pack(w'[K, cout, cin_pad]),  w'[k, o, c] = w[K-1-k, c, o]    (the input-gradient GEMM's operand)
struct PackJob {
    const float *w;
    uint8_t *wt;
    int K, cin_src, cin, cout, cout_src;      // logical [K, cin, cout] of the tiles; the source is w[K, cin_src, cout_src]
    int transposed;
    unsigned first_block;
    unsigned long long total;
};
constexpr int PACK_MAX_JOBS = 24;
struct PackJobs {
    int n;
    PackJob j[PACK_MAX_JOBS];
};

__global__ void pack_weights_bf16x3_many_kernel(const PackJobs jobs)
{
    int ji = 0;
#pragma unroll 1
    for (int t = 1; t < jobs.n; ++t)
        if (blockIdx.x >= jobs.j[t].first_block) ji = t;
    const PackJob &J = jobs.j[ji];
    // one thread per 16-byte slot (8 channels of one column): the four slots of a column's 64-byte row are neighbouring lanes, so a wave
    // writes 1 KB contiguous per plane; the transposed orientation also READS 32 contiguous bytes per thread
    const size_t i = (size_t)(blockIdx.x - J.first_block) * blockDim.x + threadIdx.x;
    if (i >= J.total) return;
    const int n_chunks = (J.cin + BK - 1) / BK;
    const int k8 = (int)(i & 3);
    const int n = (int)((i >> 2) & 127);
    const size_t tile = i >> 9;
    const int tap = (int)(tile % J.K);
    const int chunk = (int)((tile / J.K) % n_chunks);
    const int nt = (int)(tile / ((size_t)J.K * n_chunks));
    const int c0 = chunk * 32 + k8 * 8, gn = nt * 128 + n;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = 0.f;
    if (gn < J.cout) {
        if (!J.transposed) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + e < J.cin_src) x[e] = J.w[((size_t)tap * J.cin_src + c0 + e) * J.cout_src + gn];
        } else if (gn < J.cin_src) {                       // logical input channel c = source column, output gn = source row
            const float *row = J.w + ((size_t)(J.K - 1 - tap) * J.cin_src + gn) * J.cout_src;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + e < J.cin) x[e] = row[c0 + e];
        }
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (__bf16)x[e];
        lo[e] = (__bf16)(x[e] - (float)hi[e]);
    }
    uint8_t *t = J.wt + tile * B3_BYTES + n * 64 + ((k8 ^ ((n >> 2) & 3)) << 4);
    *reinterpret_cast<bf16x8 *>(t) = hi;
    *reinterpret_cast<bf16x8 *>(t + B3_PLANE) = lo;
}

// fp32 rows -> split format (test / tooling helper; the layers write the format themselves)
__global__ void split_encode_kernel(const float *__restrict__ x, long R, int c, int ldx, uint8_t *__restrict__ xs, int chunks)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * chunks * 32) return;
    const int k = (int)(i & 31);
    const int ch = (int)((i >> 5) % chunks);
    const long r = (long)(i / ((size_t)32 * chunks));
    const int cc = ch * 32 + k;
    const float v = cc < c ? x[(size_t)r * ldx + cc] : 0.f;
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    const int sw = (int)(r >> 1) & 7;
    uint8_t *row = xs + ((size_t)r * chunks + ch) * SROW + (k & 7) * 2;
    *reinterpret_cast<uint16_t *>(row + (((k >> 3) ^ sw) << 4)) = __builtin_bit_cast(uint16_t, hi);
    *reinterpret_cast<uint16_t *>(row + (((4 + (k >> 3)) ^ sw) << 4)) = __builtin_bit_cast(uint16_t, lo);
}

__global__ void split_decode_kernel(const uint8_t *__restrict__ xs, long R, int c, int chunks, float *__restrict__ x, int ldx)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * c) return;
    const long r = (long)(i / c);
    const int cc = (int)(i - (size_t)r * c);
    const int ch = cc >> 5, k = cc & 31;
    const int sw = (int)(r >> 1) & 7;
    const uint8_t *row = xs + ((size_t)r * chunks + ch) * SROW + (k & 7) * 2;
    const uint16_t h = *reinterpret_cast<const uint16_t *>(row + (((k >> 3) ^ sw) << 4));
    const uint16_t l = *reinterpret_cast<const uint16_t *>(row + (((4 + (k >> 3)) ^ sw) << 4));
    x[(size_t)r * ldx + cc] = __builtin_bit_cast(float, (uint32_t)h << 16) + __builtin_bit_cast(float, (uint32_t)l << 16);
}

__global__ void fold_bn_kernel(const float *gamma, const float *beta, const float *mean, const float *var, float eps,
                               int c, float *scale, float *shift)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float s = gamma[i] * (1.0f / sqrtf(var[i] + eps));
    float ms = mean[i] * s;
    asm volatile("" : "+v"(ms));
    scale[i] = s;
    shift[i] = beta[i] - ms;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, what);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

void xv_internal_gemm8_tile_rows(int value);      // xv_gemm8.hip
void xv_internal_gemm8_xcd_columns(int value);    // xv_gemm8.hip
void xv_internal_first_tiles(int tiles);          // xv_first.hip

int xv_version(void) { return 23; }

int xv_set_tuning(int key, int value)
{
    switch (key) {
    case XV_TUNE_TILE_ROWS:
        if (value != 0 && value != 128 && value != 256 && value != 512 && value != 1024)
            return fail(XV_ERR_BAD_ARG, "xv_set_tuning: tile rows must be 0, 128, 256, 512 or 1024");
        g_tile_rows.store(value >= 512 ? 256 : value, std::memory_order_relaxed);
        xv_internal_gemm8_tile_rows(value);
        return 0;
    case XV_TUNE_FIRST_TILES:
        if (value < 0 || value > 4096) return fail(XV_ERR_BAD_ARG, "xv_set_tuning: first-layer tiles per wave must be 0 .. 4096");
        xv_internal_first_tiles(value);
        return 0;
    case XV_TUNE_FP32_GEMM:
        if (value < 0 || value > 3) return fail(XV_ERR_BAD_ARG, "xv_set_tuning: fp32 GEMM form must be 0, 1, 2 or 3");
        g_fp32_form.store(value, std::memory_order_relaxed);
        return 0;
    case XV_TUNE_XCD_COLUMNS:
        if (value < 0 || value > 1) return fail(XV_ERR_BAD_ARG, "xv_set_tuning: XCD column placement must be 0 or 1");
        xv_internal_gemm8_xcd_columns(value);
        return 0;
    default:
        return fail(XV_ERR_BAD_ARG, "xv_set_tuning: unknown key");
    }
}

const char *xv_last_error(void) { return g_err; }

int xv_pack_weights_f32(const float *w, int kred, int cout, float *wp, void *stream)
{
    if (!w || !wp || kred <= 0 || cout <= 0) return fail(XV_ERR_BAD_ARG, "pack_weights: bad argument");
    const size_t n = (size_t)kred * cout;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, kred, cout, wp);
    return check_launch("pack_weights_kernel");
}

int xv_fold_bn_f32(const float *gamma, const float *beta, const float *mean, const float *var, float eps, int c,
                   float *scale, float *shift, void *stream)
{
    if (!gamma || !beta || !mean || !var || !scale || !shift || c <= 0) return fail(XV_ERR_BAD_ARG, "fold_bn: bad argument");
    hipLaunchKernelGGL(fold_bn_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, c, scale, shift);
    return check_launch("fold_bn_kernel");
}

int xv_tdnn_layer_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias,
                      const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K,
                      int dilation, int cout, const uint8_t *row_valid, float *y, int ldy, float *y_preact,
                      void *stream)
{
    if (!x || !wp || (!y && !y_preact)) return fail(XV_ERR_BAD_ARG, "tdnn: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn: unknown act_kind");
    GemmParams p{};
    p.x = x; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.wp = wp;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid; p.y = y; p.ldy = ldy; p.ypre = y_preact;
    return launch_gemm(p, (hipStream_t)stream);
}

size_t xv_packed_weights_rows_f32_floats(int K, int cin, int ldx, int cout)
{
    if (K <= 0 || (K & 1) == 0 || cin <= 0 || ldx < cin || (ldx & 3) || cout <= 0) return 0;
    return (size_t)cout * (((size_t)K * ldx + BK - 1) / BK * BK);
}

int xv_pack_weights_rows_f32(const float *w, int K, int cin, int ldx, int cout, float *wp, void *stream)
{
    const size_t n = xv_packed_weights_rows_f32_floats(K, cin, ldx, cout);
    if (!w || !wp || n == 0) return fail(XV_ERR_BAD_ARG, "pack_weights_rows: K odd, cin <= ldx, ldx % 4 == 0");
    hipLaunchKernelGGL(pack_weights_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, K, cin, ldx, cout,
                       (int)(n / cout), wp);
    return check_launch("pack_weights_rows_kernel");
}

int xv_tdnn_layer_rows_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                           const float *bn_shift, int act_kind, const float *act_alpha, int K, int cout, const uint8_t *row_valid,
                           float *y, int ldy, void *stream)
{
    if (!x || !wp || !y) return fail(XV_ERR_BAD_ARG, "tdnn_rows: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_rows: unknown act_kind");
    const size_t n = xv_packed_weights_rows_f32_floats(K, cin, ldx, cout);
    if (n == 0) return fail(XV_ERR_BAD_ARG, "tdnn_rows: K odd, cin <= ldx, ldx % 4 == 0");
    GemmParams p{};
    p.x = x; p.R = (long)R; p.cin = (int)(n / cout); p.ldx = ldx; p.wp = wp;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = 1; p.dil = 1; p.cout = cout; p.valid = row_valid; p.y = y; p.ldy = ldy; p.lead = (K - 1) / 2;
    return launch_gemm(p, (hipStream_t)stream);
}

int xv_tdnn_layer_pool_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                           const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                           const uint8_t *row_valid, float *block_stats, void *stream)
{
    if (!x || !wp || !block_stats) return fail(XV_ERR_BAD_ARG, "tdnn_pool: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_pool: unknown act_kind");
    if (((uintptr_t)block_stats) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_pool: block_stats must be 16-byte aligned");
    GemmParams p{};
    p.x = x; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.wp = wp;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid; p.ldy = cout; p.blk = block_stats;
    return launch_gemm(p, (hipStream_t)stream);
}

int xv_fc_f32(const float *x, int nrows, int in_dim, const float *wp, const float *bias, const float *bn_scale,
              const float *bn_shift, int act_kind, const float *act_alpha, int out_dim, float *y, float *y_preact,
              void *stream)
{
    return xv_tdnn_layer_f32(x, nrows, in_dim, in_dim, wp, bias, bn_scale, bn_shift, act_kind, act_alpha, 1, 1,
                             out_dim, nullptr, y, out_dim, y_preact, stream);
}

// Split-K plan of a skinny FC: nrows <= 128 (one or two 64-row tiles) and so many slabs that the few tiles would walk them one
// after the other (embed_layer-0 of a 64-chunk training minibatch: 96 slabs on 4 workgroups = 168 us for 0.2 GFLOP).
static int splitk_groups(int nrows, int in_dim, int out_dim)
{
    if (nrows <= 0 || nrows > 128 || in_dim < 16 * BK) return 1;
    const int chunks = (in_dim + BK - 1) / BK;
    const int tiles = ((nrows + 63) / 64) * ((out_dim + BN - 1) / BN);
    int per = (chunks * tiles + 255) / 256;             // ~ one workgroup per CU
    if (per < 2) per = 2;
    return (chunks + per - 1) / per;
}

size_t xv_fc_splitk_workspace_bytes(int nrows, int in_dim, int out_dim)
{
    const int g = splitk_groups(nrows, in_dim, out_dim);
    return g > 1 ? (size_t)g * nrows * out_dim * sizeof(float) : 0;
}

int xv_fc_splitk_f32(const float *x, int nrows, int in_dim, const float *wp, const float *bias, const float *bn_scale,
                     const float *bn_shift, int act_kind, const float *act_alpha, int out_dim, float *y, float *y_preact, void *workspace,
                     void *stream)
{
    const int g = splitk_groups(nrows, in_dim, out_dim);
    if (g <= 1) return xv_fc_f32(x, nrows, in_dim, wp, bias, bn_scale, bn_shift, act_kind, act_alpha, out_dim, y, y_preact, stream);
    if (!x || !wp || (!y && !y_preact) || !workspace) return fail(XV_ERR_BAD_ARG, "fc_splitk: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "fc_splitk: unknown act_kind");
    if ((act_kind == XV_ACT_LRELU || act_kind == XV_ACT_PRELU) && !act_alpha) return fail(XV_ERR_BAD_ARG, "fc_splitk: act_alpha is NULL");
    GemmParams p{};
    p.x = x; p.R = nrows; p.cin = in_dim; p.ldx = in_dim; p.wp = wp;
    p.act = XV_ACT_NONE; p.K = 1; p.dil = 1; p.cout = out_dim; p.ldy = out_dim;
    p.ypre = (float *)workspace;                        // (launch_gemm wants an output; the split-K epilogue redirects it per group)
    p.k_splits = g; p.part = (float *)workspace; p.part_stride = (long)nrows * out_dim;
    int rc = launch_gemm(p, (hipStream_t)stream);
    if (rc) return rc;
    const size_t n = (size_t)nrows * out_dim;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float *)workspace,
                       (long)nrows * out_dim, g, nrows, out_dim, bias, bn_scale, bn_shift, act_kind, act_alpha, y, out_dim, y_preact, out_dim);
    return check_launch("splitk_reduce_kernel");
}


size_t xv_packed_weights_bf16x3_bytes(int K, int cin, int cout)
{
    if (K <= 0 || cin <= 0 || cout <= 0) return 0;
    return (size_t)((cout + BN - 1) / BN) * ((cin + BK - 1) / BK) * K * B3_BYTES;
}

int xv_pack_weights_bf16x3(const float *w, int K, int cin, int cout, void *wt, void *stream)
{
    if (!w || !wt || K <= 0 || cin <= 0 || cout <= 0) return fail(XV_ERR_BAD_ARG, "pack_weights_bf16x3: bad argument");
    const int n_chunks = (cin + BK - 1) / BK;
    const size_t total = xv_packed_weights_bf16x3_bytes(K, cin, cout) / 4;      // 4 bytes (hi+lo) per element
    hipLaunchKernelGGL(pack_weights_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, K,
                       cin, cout, n_chunks, (uint8_t *)wt, total);
    return check_launch("pack_weights_bf16x3_kernel");
}

int xv_pack_weights_bf16x3_many(int n, const float *const *w, const int32_t *K, const int32_t *cin, const int32_t *cin_pad,
                                const int32_t *cout, void *const *wt_fwd, void *const *wt_bwd, void *stream)
{
    if (n <= 0) return 0;
    if (!w || !K || !cin || !cin_pad || !cout || !wt_fwd || !wt_bwd) return fail(XV_ERR_BAD_ARG, "pack_weights_bf16x3_many: NULL pointer");
    PackJobs jobs{};
    unsigned blocks = 0;
    auto flush = [&]() -> int {
        if (jobs.n == 0) return 0;
        hipLaunchKernelGGL(pack_weights_bf16x3_many_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs);
        jobs.n = 0;
        blocks = 0;
        return check_launch("pack_weights_bf16x3_many_kernel");
    };
    for (int i = 0; i < n; ++i) {
        if (!w[i] || K[i] <= 0 || cin[i] <= 0 || cin_pad[i] < cin[i] || cout[i] <= 0)
            return fail(XV_ERR_BAD_ARG, "pack_weights_bf16x3_many: bad layer shape");
        for (int dir = 0; dir < 2; ++dir) {
            void *dst = dir ? wt_bwd[i] : wt_fwd[i];
            if (!dst) continue;
            if (((uintptr_t)dst) & 15) return fail(XV_ERR_BAD_ARG, "pack_weights_bf16x3_many: destinations must be 16-byte aligned");
            if (jobs.n == PACK_MAX_JOBS)
                if (int e = flush()) return e;
            PackJob &J = jobs.j[jobs.n++];
            J.w = w[i]; J.wt = (uint8_t *)dst; J.K = K[i]; J.cin_src = cin[i]; J.cout_src = cout[i]; J.transposed = dir;
            J.cin = dir ? cout[i] : cin_pad[i];
            J.cout = dir ? cin_pad[i] : cout[i];
            J.total = xv_packed_weights_bf16x3_bytes(J.K, J.cin, J.cout) / 32;     // one thread per 16-byte slot of each plane
            J.first_block = blocks;
            blocks += (unsigned)((J.total + 255) / 256);
        }
    }
    return flush();
}

size_t xv_split_row_bytes(int channels) { return channels <= 0 ? 0 : (size_t)((channels + 31) / 32) * SROW; }

int xv_split_encode_f32(const float *x, int64_t R, int c, int ldx, void *xs, void *stream)
{
    if (R <= 0) return 0;
    if (!x || !xs || c <= 0 || ldx < c) return fail(XV_ERR_BAD_ARG, "split_encode: bad argument");
    const int chunks = (c + 31) / 32;
    const size_t n = (size_t)R * chunks * 32;
    hipLaunchKernelGGL(split_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)R, c, ldx,
                       (uint8_t *)xs, chunks);
    return check_launch("split_encode_kernel");
}

int xv_split_decode_f32(const void *xs, int64_t R, int c, float *x, int ldx, void *stream)
{
    if (R <= 0) return 0;
    if (!x || !xs || c <= 0 || ldx < c) return fail(XV_ERR_BAD_ARG, "split_decode: bad argument");
    const size_t n = (size_t)R * c;
    hipLaunchKernelGGL(split_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)xs, (long)R, c, (c + 31) / 32, x, ldx);
    return check_launch("split_decode_kernel");
}

int xv_tdnn_layer_bf16x3(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                         const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation,
                         int cout, const uint8_t *row_valid, void *y, int y_format, int ldy, float *y_preact, int ldpre,
                         void *stream)
{
    if (!x || !wt || (!y && !y_preact)) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: unknown act_kind");
    if ((x_format != XV_FMT_F32 && x_format != XV_FMT_SPLIT) || (y_format != XV_FMT_F32 && y_format != XV_FMT_SPLIT))
        return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3: unknown tensor format");
    Gemm3Params p{};
    p.x = x; p.x_split = x_format == XV_FMT_SPLIT; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.wt = (const uint8_t *)wt;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid;
    p.y = y; p.y_split = y_format == XV_FMT_SPLIT; p.ldy = ldy; p.ypre = y_preact; p.ldpre = ldpre;
    return launch_gemm3(p, (hipStream_t)stream);
}

int xv_tdnn_layer_bf16x3_sums(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                              const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation,
                              int cout, const uint8_t *row_valid, float *y, int ldy, const float *sum_r, int ld_sum_r, void *workspace,
                              void *stream)
{
    if (!x || !wt || !y || !sum_r || !workspace) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3_sums: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3_sums: unknown act_kind");
    if (x_format != XV_FMT_F32 && x_format != XV_FMT_SPLIT) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3_sums: unknown tensor format");
    if ((cout & 7) || (ldy & 3) || (ld_sum_r & 3) || ld_sum_r < cout || (((uintptr_t)sum_r) & 15) || (((uintptr_t)workspace) & 7))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_bf16x3_sums: needs cout % 8 == 0, row strides % 4 == 0 and aligned pointers");
    Gemm3Params p{};
    p.x = x; p.x_split = x_format == XV_FMT_SPLIT; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.wt = (const uint8_t *)wt;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid;
    p.y = y; p.y_split = 0; p.ldy = ldy;
    p.cs_r = sum_r; p.cs_ldr = ld_sum_r; p.cs_part = (double *)workspace;
    return launch_gemm3(p, (hipStream_t)stream);
}

int xv_tdnn_layer_bf16x3_moments(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                                 const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation,
                                 int cout, const uint8_t *row_valid, float *y, int ldy, float *y_preact, int ldpre, void *workspace,
                                 void *stream)
{
    if (!x || !wt || !y || !workspace) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3_moments: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3_moments: unknown act_kind");
    if (x_format != XV_FMT_F32 && x_format != XV_FMT_SPLIT) return fail(XV_ERR_BAD_ARG, "tdnn_bf16x3_moments: unknown tensor format");
    if ((cout & 7) || (ldy & 3) || (((uintptr_t)workspace) & 7))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_bf16x3_moments: needs cout % 8 == 0, ldy % 4 == 0 and an 8-byte aligned workspace");
    Gemm3Params p{};
    p.x = x; p.x_split = x_format == XV_FMT_SPLIT; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.wt = (const uint8_t *)wt;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid;
    p.y = y; p.y_split = 0; p.ldy = ldy; p.ypre = y_preact; p.ldpre = ldpre;
    p.cs_r = nullptr; p.cs_part = (double *)workspace;
    return launch_gemm3(p, (hipStream_t)stream);
}

size_t xv_block_stats_bytes(int64_t R, int cout)
{
    if (R <= 0 || cout <= 0) return 0;
    return (size_t)((R + 7) / 8) * 2 * (size_t)cout * sizeof(float);
}

int xv_tdnn_layer_pool_bf16x3(const void *x, int x_format, int64_t R, int cin, int ldx, const void *wt, const float *bias,
                              const float *bn_scale, const float *bn_shift, int act_kind, const float *act_alpha, int K,
                              int dilation, int cout, const uint8_t *row_valid, float *block_stats, void *stream)
{
    if (!x || !wt || !block_stats) return fail(XV_ERR_BAD_ARG, "tdnn_pool_bf16x3: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_pool_bf16x3: unknown act_kind");
    if (x_format != XV_FMT_F32 && x_format != XV_FMT_SPLIT) return fail(XV_ERR_BAD_ARG, "tdnn_pool_bf16x3: unknown tensor format");
    if (((uintptr_t)block_stats) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_pool_bf16x3: block_stats must be 16-byte aligned");
    Gemm3Params p{};
    p.x = x; p.x_split = x_format == XV_FMT_SPLIT; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.wt = (const uint8_t *)wt;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid;
    p.blk = block_stats;
    return launch_gemm3(p, (hipStream_t)stream);
}

int xv_stats_pool_blocks_f32(const float *block_stats, int c, const int32_t *row_start, const int32_t *row_len, int nchunks,
                             float eps, float *out, void *stream)
{
    if (nchunks <= 0) return 0;
    if (!block_stats || !row_start || !row_len || !out || c <= 0) return fail(XV_ERR_BAD_ARG, "stats_pool_blocks: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int b0 = 0; b0 < nchunks; b0 += 65535) {
        const int nb = min(65535, nchunks - b0);
        hipLaunchKernelGGL(stats_pool_blocks_kernel, dim3((c + 255) / 256, nb), dim3(256), 0, st, block_stats, c, row_start + b0,
                           row_len + b0, eps, out + (size_t)b0 * 2 * c);
        int rc = check_launch("stats_pool_blocks_kernel");
        if (rc) return rc;
    }
    return 0;
}

int xv_fc_bf16x3(const float *x, int nrows, int in_dim, const void *wt, const float *bias, const float *bn_scale,
                 const float *bn_shift, int act_kind, const float *act_alpha, int out_dim, float *y, float *y_preact, void *stream)
{
    return xv_tdnn_layer_bf16x3(x, XV_FMT_F32, nrows, in_dim, in_dim, wt, bias, bn_scale, bn_shift, act_kind, act_alpha, 1, 1,
                                out_dim, nullptr, y, XV_FMT_F32, out_dim, y_preact, out_dim, stream);
}

size_t xv_stats_pool_workspace_bytes(int c, int nchunks, int max_len, int split_rows)
{
    if (split_rows <= 0 || max_len <= split_rows) return 0;
    const size_t splits = ((size_t)max_len + split_rows - 1) / split_rows;
    return (size_t)nchunks * splits * 2 * (size_t)c * sizeof(float);
}

static int stats_pool_impl(const float *h, int64_t ldh, int c, const int32_t *row_start, const int32_t *row_len, int nchunks,
                           int max_len, int split_rows, float eps, float *out, void *workspace, void *stream, int raw)
{
    if (nchunks <= 0) return 0;
    if (!h || !row_start || !row_len || !out) return fail(XV_ERR_BAD_ARG, "stats_pool: NULL pointer");
    if (c <= 0 || (c & 3) || (ldh & 3) || (((uintptr_t)h) & 15) || (((uintptr_t)out) & 15))
        return fail(XV_ERR_BAD_ARG, "stats_pool: C, ldh must be multiples of 4 and h/out 16-byte aligned");
    if (split_rows <= 0 || max_len <= 0) return fail(XV_ERR_BAD_ARG, "stats_pool: split_rows/max_len must be > 0");
    const int max_splits = (max_len + split_rows - 1) / split_rows;
    if (max_splits > 1 && !workspace) return fail(XV_ERR_BAD_ARG, "stats_pool: workspace required for split chunks");
    if (max_splits > 65535) return fail(XV_ERR_UNSUPPORTED, "stats_pool: too many splits");
    hipStream_t st = (hipStream_t)stream;
    // grid.z is limited to 65535: loop over slices of chunks
    for (int b0 = 0; b0 < nchunks; b0 += 65535) {
        const int nb = min(65535, nchunks - b0);
        const dim3 grid((c + 255) / 256, max_splits, nb);
        hipLaunchKernelGGL(stats_pool_kernel, grid, dim3(256), 0, st, h, (long)ldh, c, row_start + b0, row_len + b0,
                           split_rows, max_splits, eps, out + (size_t)b0 * 2 * c,
                           (float *)workspace + (size_t)b0 * max_splits * 2 * c, raw);
        int rc = check_launch("stats_pool_kernel");
        if (rc) return rc;
        if (max_splits > 1) {
            hipLaunchKernelGGL(stats_pool_merge_kernel, dim3((c + 255) / 256, nb), dim3(256), 0, st,
                               (const float *)workspace + (size_t)b0 * max_splits * 2 * c, c, row_len + b0, split_rows,
                               max_splits, eps, out + (size_t)b0 * 2 * c, raw);
            rc = check_launch("stats_pool_merge_kernel");
            if (rc) return rc;
        }
    }
    return 0;
}

int xv_stats_pool_f32(const float *h, int64_t ldh, int c, const int32_t *row_start, const int32_t *row_len, int nchunks,
                      int max_len, int split_rows, float eps, float *out, void *workspace, void *stream)
{
    return stats_pool_impl(h, ldh, c, row_start, row_len, nchunks, max_len, split_rows, eps, out, workspace, stream, 0);
}

int xv_chunk_moments_f32(const float *h, int64_t ldh, int c, const int32_t *row_start, const int32_t *row_len, int nchunks,
                         int max_len, int split_rows, float *out, void *workspace, void *stream)
{
    return stats_pool_impl(h, ldh, c, row_start, row_len, nchunks, max_len, split_rows, 0.f, out, workspace, stream, 1);
}

void xv_internal_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }

int xv_chunk_average_f32(const float *e, const int32_t *seg_start, const int32_t *chunk_len, int nutts, int dim, float *out,
                         void *stream)
{
    if (nutts <= 0) return 0;
    if (!e || !seg_start || !chunk_len || !out || dim <= 0) return fail(XV_ERR_BAD_ARG, "chunk_average: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int u0 = 0; u0 < nutts; u0 += 65535) {
        const int nu = min(65535, nutts - u0);
        hipLaunchKernelGGL(chunk_average_kernel, dim3((dim + 255) / 256, nu), dim3(256), 0, st, e, seg_start + u0, chunk_len,
                           dim, out + (size_t)u0 * dim);
        int rc = check_launch("chunk_average_kernel");
        if (rc) return rc;
    }
    return 0;
}

}  // extern "C"
