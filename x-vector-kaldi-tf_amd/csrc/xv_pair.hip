// xv_pair.hip -- the two context-free (K = 1) frame-level layers and the first half of statistics pooling as ONE gfx950
// kernel: frame_level_info_layer-3 -> frame_level_info_layer-4 -> per-8-row block statistics
// (local/tf/models.py:54-76 / 470-486 for the last two layers of the default topology, kernel sizes [5,5,7,1,1]).
//
// Why a kernel of its own.  With K = 1 a GEMM tile has no tap re-use: the LDS-staged kernel of xv_kernels.hip moves one
// activation tile AND one weight tile per 24 MFMAs and spends a third of a tile's life in prologue/epilogue; layer 3
// writes 0.54 GB per 262144-row batch that layer 4 reads back 12 times from L2.  Here the intermediate never leaves the
// register file:
//   * a wave owns 16 frames.  Phase 1 computes H^T = W1^T . X^T on v_mfma_f32_16x16x32_bf16 with the WEIGHTS as the A
//     operand and the frames as the B operand, so the accumulator of channel tile T holds, in lane (frame f = lane&15,
//     group G = lane>>4), channels 16T+4G .. 16T+4G+3 of frame f -- after bias/activation/BN and the hi/lo split, the
//     accumulators of tiles 2u and 2u+1 ARE the A operand (16 frames x 32 "k") of k-step u of the second GEMM, with the
//     k order (e -> channel 32u + 16(e>>2) + 4G + (e&3)) that the packed weights of layer 4 simply follow.  No LDS round
//     trip, no store, no reload: 512 channels x 16 frames x (hi + lo) = 128 VGPRs.
//   * Phase 2 computes Y = H . W2 for 64 columns at a time (4 accumulators of 16 frames x 16 columns), and its epilogue
//     reduces the wave's two 8-row blocks to per-channel (mean, M2) exactly like the POOL epilogue of
//     tdnn_gemm_bf16x3_kernel (same block_stats layout, merged by stats_pool_blocks_kernel).
//   * the only LDS traffic is the weight stream: both layers' weights are packed once, in MFMA-fragment order (1 KB per
//     16x32 fragment, hi and lo planes), into 32 KB stages in the exact order the kernel consumes them; 8 waves fetch a
//     stage with 4 x 1 KB global->LDS DMAs each into a ring of 3 stages (DMA issued three stages ahead, counted
//     s_waitcnt vmcnt, one s_barrier per stage); every fragment read is a conflict-free linear ds_read_b128.
//   * the frames operand X is gathered by DMA straight into fragment order (per wave 16 rows x 32 channels x hi/lo per
//     k-step) from the split-format activation buffer of the previous layer.
// Arithmetic is the bf16x3 scheme of xv_kernels.hip (x = hi + lo, products lo*hi + hi*lo + hi*hi, fp32 accumulate).
// Per 128-frame workgroup: 4.25 MB global->LDS (the unfused pair: 8 MB), no activation store, no activation reload.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "xvector_hip.h"

extern "C" void xv_internal_set_error(const char *msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

int fail(int code, const char *msg)
{
    xv_internal_set_error(msg);
    return code;
}

constexpr int CMID = 512;                  // width of the intermediate layer (register budget: CMID/4 VGPRs of operands)
constexpr int PR_WAVES = 8;
constexpr int PR_ROWS = 16 * PR_WAVES;     // frames per workgroup
constexpr int PR_STAGE = 32768;            // bytes per weight stage: 32 fragments of 1 KB
constexpr int PR_RING = 3;
constexpr int PR_X_OFF = PR_RING * PR_STAGE;               // per wave [hi 1 KB | lo 1 KB] frames fragment of one k-step
constexpr int PR_P1_OFF = PR_X_OFF + PR_WAVES * 2048;      // [bias | scale | shift | alpha][CMID] of the first layer
constexpr int PR_P2_OFF = PR_P1_OFF + 4 * CMID * 4;        // float4 {bias, scale, shift, alpha} per output column
constexpr int SROW = 128;                  // bytes per (row, 32-channel slab) of the split activation format

struct PairParams {
    const uint8_t *x;          // split-format input, row 0
    long R;
    int n_ks;                  // cin / 32
    int cout, n_ct;            // n_ct = cout / 64
    const uint8_t *wt;         // packed stages: 2*n_ks stages of layer 1, then 4*n_ct stages of layer 2
    const float *b1, *sc1, *sh1, *al1;
    const float *b2, *sc2, *sh2, *al2;
    int act;
    const uint8_t *valid;
    float *blk;                // [ceil(R/8)][2][cout]
    long n_blocks;
};

#define XV_GLDS16_OFF(gptr, lptr, imm)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                    \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, imm, 0)

struct Frags {                 // the 4 weight fragments (hi and lo plane) of one sub-step: 8 x 4 VGPRs
    bf16x8 hi[4], lo[4];
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// MODE 0: max(z,0) + alpha*min(z,0) (identity with alpha = 1, PReLU with per-channel alpha)   1: tf.nn.leaky_relu =
// max(alpha*z, z)   2: plain ReLU.  A compile-time choice: the epilogues below are straight-line code.
template <int MODE>
__device__ __forceinline__ float act_fn(float z, float a)
{
    return MODE == 1 ? fmaxf(a * z, z) : MODE == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + a * fminf(z, 0.f);
}

template <int MODE>
__global__ __launch_bounds__(PR_WAVES * 64, 2) void tdnn_pair_pool_kernel(const PairParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = lane >> 4, li = lane & 15;
    const long m0 = (long)blockIdx.x * PR_ROWS;
    const long row0 = m0 + 16 * wave;                 // the wave's 16 frames; a multiple of 8: two pooling blocks

    // ---- before any DMA is in flight: row validity of this lane's 4 frames (4G .. 4G+3), epilogue parameters -> LDS ----
    float keep[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long gr = row0 + 4 * G + r;
        keep[r] = (gr < p.R && (!p.valid || p.valid[gr])) ? 1.f : 0.f;
    }
    float nblk = keep[0] + keep[1] + keep[2] + keep[3];
    nblk += __shfl_xor(nblk, 16, 64);                 // valid frames of this lane's 8-row block (groups G, G^1)
    const float rn = nblk > 0.f ? 1.f / nblk : 0.f;
    {
        float *P1 = reinterpret_cast<float *>(lds + PR_P1_OFF);
        for (int c = tid; c < CMID; c += PR_WAVES * 64) {
            P1[c] = p.b1 ? p.b1[c] : 0.f;
            P1[CMID + c] = p.sc1 ? p.sc1[c] : 1.f;
            P1[2 * CMID + c] = p.sh1 ? p.sh1[c] : 0.f;
            P1[3 * CMID + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.al1[0] : p.act == XV_ACT_PRELU ? p.al1[c] : 0.f;
        }
        f32x4 *P2 = reinterpret_cast<f32x4 *>(lds + PR_P2_OFF);
        for (int c = tid; c < p.cout; c += PR_WAVES * 64) {
            f32x4 v;
            v[0] = p.b2 ? p.b2[c] : 0.f;
            v[1] = p.sc2 ? p.sc2[c] : 1.f;
            v[2] = p.sh2 ? p.sh2[c] : 0.f;
            v[3] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.al2[0] : p.act == XV_ACT_PRELU ? p.al2[c] : 0.f;
            P2[c] = v;
        }
    }
    __syncthreads();

    // ---- DMA streams -----------------------------------------------------------------------------------------------
    // weights: stage after stage; every wave moves 4 KB of a stage (4 pieces, one address register, one M0)
    const uint8_t *wsrc = p.wt + wave * 4096 + lane * 16;
    int wleft = 2 * p.n_ks + 4 * p.n_ct;               // stages not yet issued (the tail re-issues the last stage)
    auto issue_w = [&](int slot_off) {
        char *dst = lds + slot_off + wave * 4096;
        XV_GLDS16_OFF(wsrc, dst, 0);
        XV_GLDS16_OFF(wsrc, dst, 1024);
        XV_GLDS16_OFF(wsrc, dst, 2048);
        XV_GLDS16_OFF(wsrc, dst, 3072);
        const bool more = wleft > 1;
        wsrc += more ? PR_STAGE : 0;
        wleft -= more ? 1 : 0;
    };
    // frames: lane (li, G) of k-step ks fetches the 16-B slots G (hi) and 4+G (lo) of row row0+li, slab ks -- the DMA
    // lands them lane-linear, i.e. in B-operand fragment order.  Physical slot = logical ^ ((row>>1)&7).
    const int sw = (int)((row0 + li) >> 1) & 7;
    const size_t xrow_bytes = (size_t)p.n_ks * SROW;
    const uint8_t *xh = p.x + (row0 + li) * (long)xrow_bytes + ((G ^ sw) << 4);
    const uint8_t *xl = p.x + (row0 + li) * (long)xrow_bytes + (((4 + G) ^ sw) << 4);
    int xleft = p.n_ks;
    char *xdst = lds + PR_X_OFF + wave * 2048;
    auto issue_x = [&]() {
        XV_GLDS16_OFF(xh, xdst, 0);
        XV_GLDS16_OFF(xl, xdst + 1024, 0);          // (an immediate offset would move the SOURCE address as well)
        const bool more = xleft > 1;
        xh += more ? SROW : 0;
        xl += more ? SROW : 0;
        xleft -= more ? 1 : 0;
    };

    // fragment reads: fragment f of the stage in ring slot `slot` is 1 KB at slot + f*1024, lane-linear
    const char *fbase = lds + lane * 16;
    auto load_frags = [&](Frags &F, int slot, int sub) {
        const char *b = fbase + slot + sub * 8192;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            F.hi[c] = *reinterpret_cast<const bf16x8 *>(b + c * 2048);
            F.lo[c] = *reinterpret_cast<const bf16x8 *>(b + c * 2048 + 1024);
        }
    };
    auto next_slot = [](int slot) { return slot + PR_STAGE == PR_RING * PR_STAGE ? 0 : slot + PR_STAGE; };
    // B(s): this wave's fragment reads of stage s are complete (its slot may be overwritten), its DMA pieces of stage
    // s+1 (and the frames fragment that goes with it) have landed -- issued two barriers ago, only the 4 pieces of stage
    // s+2 may still be in flight -- and after the barrier the same holds for every wave.
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // MFMA / LDS-read interleave of one sub-step (12 MFMAs, 8 ds_read_b128, optionally DMA pieces)
    auto pin = [&](auto NVMEM) {
        constexpr int nv = decltype(NVMEM)::value;
#pragma unroll
        for (int i = 0; i < nv; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read (LDS-DMA piece)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 12 - nv) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12 - nv - 8 > 0 ? 12 - nv - 8 : 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue ----------------------------------------------------------------------------------------------------
    issue_x();
    issue_w(0);
    issue_w(PR_STAGE);
    issue_w(2 * PR_STAGE);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // frames fragment 0 and stage 0
    __builtin_amdgcn_s_barrier();
    Frags F, Gf;
    load_frags(F, 0, 0);
    bf16x8 xfh = *reinterpret_cast<const bf16x8 *>(lds + PR_X_OFF + wave * 2048 + lane * 16);
    bf16x8 xfl = *reinterpret_cast<const bf16x8 *>(lds + PR_X_OFF + wave * 2048 + 1024 + lane * 16);
    int slot = 0;

    // ---- phase 1: H^T[channel][frame], 32 channel tiles x 4 accumulator VGPRs ---------------------------------------
    f32x4 acc[CMID / 16];
#pragma unroll
    for (int t = 0; t < CMID / 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma1 = [&](const Frags &W, auto T0) {             // tiles T0 .. T0+3: A = weight fragment, B = frames fragment
        constexpr int t0 = decltype(T0)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.lo[c], xfh, acc[t0 + c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.hi[c], xfl, acc[t0 + c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t0 + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.hi[c], xfh, acc[t0 + c], 0, 0, 0);
    };
    for (int ks = 0; ks < p.n_ks; ++ks) {
        auto half = [&](auto HALF) {                       // one stage = one k-step x 16 channel tiles
            constexpr int h = decltype(HALF)::value;
            load_frags(Gf, slot, 1);
            mma1(F, std::integral_constant<int, 16 * h>{});
            pin(std::integral_constant<int, 0>{});
            load_frags(F, slot, 2);
            mma1(Gf, std::integral_constant<int, 16 * h + 4>{});
            pin(std::integral_constant<int, 0>{});
            load_frags(Gf, slot, 3);
            mma1(F, std::integral_constant<int, 16 * h + 8>{});
            pin(std::integral_constant<int, 0>{});
            stage_barrier();
            if constexpr (h == 0) issue_x();               // the frames fragment of k-step ks+1 (the slot was read after B(2ks-1))
            issue_w(slot);
            slot = next_slot(slot);
            load_frags(F, slot, 0);
            if constexpr (h == 0) {
                mma1(Gf, std::integral_constant<int, 16 * h + 12>{});
                pin(std::integral_constant<int, 6>{});
            } else {
                const bf16x8 nh = *reinterpret_cast<const bf16x8 *>(lds + PR_X_OFF + wave * 2048 + lane * 16);
                const bf16x8 nl = *reinterpret_cast<const bf16x8 *>(lds + PR_X_OFF + wave * 2048 + 1024 + lane * 16);
                mma1(Gf, std::integral_constant<int, 16 * h + 12>{});
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_barrier(0);
                xfh = nh;
                xfl = nl;
            }
        };
        half(std::integral_constant<int, 0>{});
        half(std::integral_constant<int, 1>{});
    }

    // ---- accumulators -> A operands of the second GEMM: bias, activation, BN, hi/lo split ----------------------------
    bf16x8 Hh[CMID / 32], Hl[CMID / 32];
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        auto conv = [&](auto U) {
            constexpr int u = decltype(U)::value;
            // k-step u's parameter reads must not start before k-step u-1 is converted: the optimiser would otherwise
            // hoist all 128 reads to the top and spill them.  An opaque zero that depends on the previous result does it.
            int dep = 0;
            if constexpr (u > 0) {
                const u32x4 a = __builtin_bit_cast(u32x4, Hh[u - 1]), b = __builtin_bit_cast(u32x4, Hl[u - 1]);
                asm volatile("" : "+v"(dep) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
            }
            const f32x4 *P1 = reinterpret_cast<const f32x4 *>(lds + PR_P1_OFF + dep);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int c4 = (32 * u + 16 * half + 4 * G) >> 2;                      // float4 index of the lane's 4 channels
                const f32x4 b = P1[c4], s = P1[CMID / 4 + c4], o = P1[2 * CMID / 4 + c4], a = P1[3 * CMID / 4 + c4];
                const f32x4 t = acc[2 * u + half];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = act_fn<MODE>(t[e] + b[e], a[e]) * s[e] + o[e];
                    const __bf16 hi = (__bf16)v;
                    Hh[u][4 * half + e] = hi;
                    Hl[u][4 * half + e] = (__bf16)(v - (float)hi);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, CMID / 32>(conv);
    }
    load_frags(F, slot, 0);        // (again: the copy read inside the loop is dropped so that the conversion has the registers)

    // ---- phase 2: Y[frame][column] for 64 columns at a time, pooled on the spot ------------------------------------
    const f32x4 *P2 = reinterpret_cast<const f32x4 *>(lds + PR_P2_OFF);
    const long blk_row = (row0 >> 3) + (G >> 1);                    // this lane's 8-row block
    const bool blk_ok = blk_row < p.n_blocks;
    float *blk_out = p.blk + (size_t)(blk_ok ? blk_row : 0) * 2 * p.cout + ((G & 1) ? p.cout : 0) + li;
    for (int ct = 0; ct < p.n_ct; ++ct) {
        f32x4 y[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) y[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto mma2 = [&](const Frags &W, auto U) {           // k-step u: A = H fragment (registers), B = weight fragment
            constexpr int u = decltype(U)::value;
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Hl[u], W.hi[c], y[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Hh[u], W.lo[c], y[c], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 4; ++c) y[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Hh[u], W.hi[c], y[c], 0, 0, 0);
        };
        auto quarter = [&](auto Q) {                        // one stage = 4 k-steps x 64 columns
            constexpr int q = decltype(Q)::value;
            load_frags(Gf, slot, 1);
            mma2(F, std::integral_constant<int, 4 * q>{});
            pin(std::integral_constant<int, 0>{});
            load_frags(F, slot, 2);
            mma2(Gf, std::integral_constant<int, 4 * q + 1>{});
            pin(std::integral_constant<int, 0>{});
            load_frags(Gf, slot, 3);
            mma2(F, std::integral_constant<int, 4 * q + 2>{});
            pin(std::integral_constant<int, 0>{});
            stage_barrier();
            issue_w(slot);
            slot = next_slot(slot);
            load_frags(F, slot, 0);
            mma2(Gf, std::integral_constant<int, 4 * q + 3>{});
            pin(std::integral_constant<int, 4>{});
        };
        static_for<0, CMID / 128>(quarter);

        // pooling epilogue: lane (column li of tile c, group G) holds frames 4G .. 4G+3; groups G and G^1 make one 8-row
        // block.  Statistics shifted by the block's first row (no cancellation), as in the POOL epilogue of the GEMM kernel.
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = ct * 64 + c * 16 + li;
            const f32x4 prm = P2[col];
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = act_fn<MODE>(y[c][r] + prm[0], prm[3]) * prm[1] + prm[2];
            const float v0 = __shfl(v[0], lane & ~16, 64);           // first row of the block (held by the even group)
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = keep[r] != 0.f ? v[r] - v0 : 0.f;         // (a select: a row past R may hold anything, NaN * 0 is NaN)
                s1 += d;
                s2 += d * d;
            }
            s1 += __shfl_xor(s1, 16, 64);
            s2 += __shfl_xor(s2, 16, 64);
            const float mean = nblk > 0.f ? v0 + s1 * rn : 0.f;
            const float m2 = fmaxf(s2 - s1 * s1 * rn, 0.f);
            if (blk_ok) blk_out[ct * 64 + c * 16] = (G & 1) ? m2 : mean;      // even group: mean plane, odd group: M2 plane
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// w1[cin][CMID], w2[CMID][cout] (fp32, TF's [in, out] order) -> stages in consumption order (see the kernel):
//   layer 1, stage 2*ks + h, fragment (t, plane):  lane (i, g), element e  =  w1[32ks + 8g + e][16(16h + t) + i]
//   layer 2, stage 4*ct + q, fragment (uu, c, plane): lane (j, g), element e = w2[32(4q+uu) + 16(e>>2) + 4g + (e&3)][64ct + 16c + j]
__global__ void pack_pair_kernel(const float *__restrict__ w1, const float *__restrict__ w2, int n_ks, int cout, int n_ct,
                                 uint8_t *__restrict__ wt, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (stage, fragment pair, lane, e)
    if (i >= total) return;
    const int e = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    const int fp = (int)((i >> 9) & 15);          // fragment pair within the stage
    const long stage = (long)(i >> 13);
    const int j = lane & 15, g = lane >> 4;
    float x;
    if (stage < 2L * n_ks) {
        const int ks = (int)(stage >> 1), h = (int)(stage & 1);
        x = w1[(size_t)(32 * ks + 8 * g + e) * CMID + 16 * (16 * h + fp) + j];
    } else {
        const long s2 = stage - 2L * n_ks;
        const int ct = (int)(s2 >> 2), q = (int)(s2 & 3);
        const int uu = fp >> 2, c = fp & 3;
        const int ch = 32 * (4 * q + uu) + 16 * (e >> 2) + 4 * g + (e & 3);
        x = w2[(size_t)ch * cout + 64 * ct + 16 * c + j];
    }
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    uint8_t *t = wt + (size_t)stage * PR_STAGE + (size_t)fp * 2048 + lane * 16 + e * 2;
    *reinterpret_cast<uint16_t *>(t) = __builtin_bit_cast(uint16_t, hi);
    *reinterpret_cast<uint16_t *>(t + 1024) = __builtin_bit_cast(uint16_t, lo);
}

bool pair_shape_ok(int cin, int cmid, int cout) { return cmid == CMID && cin > 0 && (cin & 31) == 0 && cout > 0 && (cout & 63) == 0 && cout <= 2048; }

}  // namespace

extern "C" {

size_t xv_packed_pair_bf16x3_bytes(int cin, int cmid, int cout)
{
    if (!pair_shape_ok(cin, cmid, cout)) return 0;
    return (size_t)(2 * (cin / 32) + 4 * (cout / 64)) * PR_STAGE;
}

int xv_pack_pair_bf16x3(const float *w1, const float *w2, int cin, int cmid, int cout, void *wt, void *stream)
{
    if (!w1 || !w2 || !wt) return fail(XV_ERR_BAD_ARG, "pack_pair_bf16x3: NULL pointer");
    if (!pair_shape_ok(cin, cmid, cout))
        return fail(XV_ERR_UNSUPPORTED, "pack_pair_bf16x3: needs cmid == 512, cin % 32 == 0, cout % 64 == 0, cout <= 2048");
    const size_t total = xv_packed_pair_bf16x3_bytes(cin, cmid, cout) / 4;
    hipLaunchKernelGGL(pack_pair_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1, w2, cin / 32,
                       cout, cout / 64, (uint8_t *)wt, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

int xv_tdnn_pair_pool_bf16x3(const void *x, int64_t R, int cin, int cmid, int cout, const void *wt, const float *bias1,
                             const float *bn_scale1, const float *bn_shift1, const float *act_alpha1, const float *bias2,
                             const float *bn_scale2, const float *bn_shift2, const float *act_alpha2, int act_kind,
                             const uint8_t *row_valid, float *block_stats, void *stream)
{
    if (R <= 0) return 0;
    if (!x || !wt || !block_stats) return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_bf16x3: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_bf16x3: unknown act_kind");
    if ((act_kind == XV_ACT_LRELU || act_kind == XV_ACT_PRELU) && (!act_alpha1 || !act_alpha2))
        return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_bf16x3: act_alpha is NULL");
    if (!pair_shape_ok(cin, cmid, cout))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_pair_pool_bf16x3: needs cmid == 512, cin % 32 == 0, cout % 64 == 0, cout <= 2048");
    if ((((uintptr_t)x) | ((uintptr_t)wt) | ((uintptr_t)block_stats)) & 15)
        return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_bf16x3: x, wt and block_stats must be 16-byte aligned");
    PairParams p{};
    p.x = (const uint8_t *)x; p.R = (long)R; p.n_ks = cin / 32; p.cout = cout; p.n_ct = cout / 64; p.wt = (const uint8_t *)wt;
    p.b1 = bias1; p.sc1 = bn_scale1; p.sh1 = bn_shift1; p.al1 = act_alpha1;
    p.b2 = bias2; p.sc2 = bn_scale2; p.sh2 = bn_shift2; p.al2 = act_alpha2;
    p.act = act_kind; p.valid = row_valid; p.blk = block_stats; p.n_blocks = (long)((R + 7) / 8);
    const size_t lds_bytes = (size_t)PR_P2_OFF + (size_t)cout * 16;
    typedef void (*kern_t)(const PairParams);
    const kern_t kerns[3] = {tdnn_pair_pool_kernel<0>, tdnn_pair_pool_kernel<1>, tdnn_pair_pool_kernel<2>};
    static std::atomic<unsigned long long> attr_done{0};      // dynamic-LDS opt-in: per device, idempotent
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (kern_t k : kerns) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, PR_P2_OFF + 2048 * 16);
            if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int mode = act_kind == XV_ACT_LRELU ? 1 : act_kind == XV_ACT_RELU ? 2 : 0;
    hipLaunchKernelGGL(kerns[mode], dim3((unsigned)((R + PR_ROWS - 1) / PR_ROWS)), dim3(PR_WAVES * 64), lds_bytes,
                       (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

}  // extern "C"
