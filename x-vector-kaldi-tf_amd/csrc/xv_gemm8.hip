// xv_gemm8.hip -- the hidden frame-level layers (tf.nn.conv1d 'SAME' + bias + activation + BN-eval, local/tf/models.py:54-76)
// in the "f16bf8" arithmetic: x*w = xh*wh (fp16 MFMA) + 2^-11 (xl8*wh8 + xh8*wl8) (ONE block-scaled 8-bit MFMA whose K = 64
// carries both cross terms of a 32-channel slab) -- 128 MFMA passes per stage and wave where the bf16x3 kernel of
// xv_kernels.hip spends 192.  Representation, accuracy and range rules: xv_split8.h.
//
// Everything around the arithmetic is the design of tdnn_gemm_bf16x3_kernel, unchanged on purpose:
// implicit im2col over a (BM + (K-1)d)-row halo tile that all K taps of a 32-channel slab re-use, operands fed by direct
// global->LDS DMA from pre-formatted buffers (128 bytes per row-slab, 16 KB weight tile per stage, XOR-swizzled 16-byte
// slots), WM x 2 waves of 64 x 64 sub-tiles, XCD-aware tile order, a register-level software pipeline with one barrier per
// stage, the epilogue through an fp32 LDS tile.  Per stage and wave:
//     phase 1:  8 fp16 MFMAs (32x32x16: 2 k-steps x 2 x 2 tiles) on set F   | 8 ds_read_b128: set G = 8-bit fragments of stage s
//     barrier
//     phase 2:  4 scaled 8-bit MFMAs (32x32x64) on set G | DMA of stage s+2 | 8 ds_read_b128: set F = fp16 fragments of stage s+1
// 8-bit fragment of lane (row = lane & 31, half = lane >> 5): slots 4+2*half and 5+2*half of the row-slab = channels
// 16*half .. 16*half+15 as [8 x l8 | 8 x h8] twice -- and the weight tile holds [8 x h8 | 8 x l8] at the same positions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "xvector_hip.h"
#include "xv_split8.h"

extern "C" void xv_internal_set_error(const char *msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

int fail(int code, const char *msg)
{
    xv_internal_set_error(msg);
    return code;
}

int hip_fail(hipError_t e, const char *where)
{
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", where, hipGetErrorString(e));
    xv_internal_set_error(buf);
    return (int)e;
}

constexpr int BN = 128;                 // output channels per workgroup tile
constexpr int BK = 32;                  // input channels per stage
constexpr int MAX_SPAN = 8;             // (K-1)*dilation limit
constexpr int SROW = 128;               // bytes per (row, 32-channel slab)
constexpr int B_PLANE = BN * 64;        // 8192: fp16 plane / 8-bit plane of a weight tile
constexpr int B_BYTES = 2 * B_PLANE;    // 16384
constexpr int T_LD = BN + 4;            // epilogue fp32 tile row (floats)

// (A ring of THREE weight tiles with the DMA issued three stages ahead and a counted s_waitcnt vmcnt was measured on the
// 256-row form: no gain -- DMA latency is not what parks the waves -- so two buffers and vmcnt(0) it is.)
constexpr int g8_ring(int, int) { return 2; }           // weight tiles in LDS
constexpr int g8_oper_bytes(int kt, int wm) { return 2 * (wm * 64 + MAX_SPAN) * SROW + g8_ring(kt, wm) * B_BYTES; }
constexpr int g8_tile_bytes(int wm) { return wm * 64 * T_LD * 4; }
constexpr int g8_mask_off(int kt, int wm) { return g8_oper_bytes(kt, wm) > g8_tile_bytes(wm) ? g8_oper_bytes(kt, wm) : g8_tile_bytes(wm); }
constexpr size_t g8_lds_bytes(int kt, int wm) { return (size_t)g8_mask_off(kt, wm) + wm * 64 + 4 * BN * sizeof(float); }

template <int K, class F0, class... Fs>
__device__ __forceinline__ void call_kth(F0 &&f0, Fs &&...fs)
{
    if constexpr (K == 0) f0();
    else call_kth<K - 1>(fs...);
}

template <int T, int KT, class F>
__device__ __forceinline__ void for_taps(F &f)
{
    if constexpr (T < KT) {
        f(std::integral_constant<int, T>{});
        for_taps<T + 1, KT>(f);
    }
}

struct Gemm8Params {
    const uint8_t *x;     // split8 buffer, row 0
    long R;
    int cin, xchunks;
    const uint8_t *wt;    // tiled f16bf8 weights
    const float *bias, *scale, *shift;
    int act;
    const float *alpha;
    int K, dil, cout;
    const uint8_t *valid;
    void *y;              // fp32 rows, bf16 split buffer or split8 buffer
    int y_format, ldy, ychunks;
    float *blk;           // POOL: per-8-row-block (mean, M2) planes
    int *status;          // bit 0 is set when a split8 output had to be clamped (may be NULL)
    int n_mt, n_nt, n_chunks;
    int colmap;           // wide16, two column tiles: 1 = XCDs 0-3 work on column tile 0, XCDs 4-7 on tile 1 (XV_TUNE_XCD_COLUMNS)
};

#define XV_GLDS16_OFF(gptr, lptr, imm)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                    \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, imm, 0)
#define XV_GLDS16(gptr, lptr) XV_GLDS16_OFF(gptr, lptr, 0)
// MUBUF form: 16 bytes per lane from buffer rsrc at voff (per lane) + soff (wave-uniform) + imm to LDS lptr + imm + 16 * lane
#define XV_BLDS16(rsrc, lptr, voff, soff, imm)                                                                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lptr), 16, voff, soff, imm, 0)
#define XV_BLDS16_X4(rsrc, lptr, voff, soff, imm)                                                               \
    do {                                                                                                        \
        XV_BLDS16(rsrc, lptr, voff, soff, (imm));                                                               \
        XV_BLDS16(rsrc, lptr, voff, soff, (imm) + 1024);                                                        \
        XV_BLDS16(rsrc, lptr, voff, soff, (imm) + 2048);                                                        \
        XV_BLDS16(rsrc, lptr, voff, soff, (imm) + 3072);                                                        \
    } while (0)
constexpr int XV_RSRC_FLAGS = 0x00020000;              // raw buffer, 32-bit data format (gfx9 family dword 3)
#define XV_GLDS16_X4(gptr, lptr, imm)                                                                           \
    do {                                                                                                        \
        XV_GLDS16_OFF(gptr, lptr, (imm));                                                                       \
        XV_GLDS16_OFF(gptr, lptr, (imm) + 1024);                                                                \
        XV_GLDS16_OFF(gptr, lptr, (imm) + 2048);                                                                \
        XV_GLDS16_OFF(gptr, lptr, (imm) + 3072);                                                                \
    } while (0)

template <int KT, bool POOL, int WM>
__global__ __launch_bounds__(WM * 128, 2) void tdnn_gemm_f16bf8_kernel(const Gemm8Params p)
{
    constexpr int NW = 2 * WM;
    constexpr int NT = NW * 64;
    constexpr int BM = WM * 64;
    constexpr int A_ROWS = BM + MAX_SPAN;
    constexpr int A_BYTES = A_ROWS * SROW;
    constexpr int BP = 16 / NW;                        // 1 KB pieces of a weight tile per wave
    constexpr int RING = g8_ring(KT, WM);              // weight tiles in LDS
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Abuf = lds;                                  // [2][A_ROWS][128 B]
    char *Bbuf = lds + 2 * A_BYTES;                    // [RING][fp16 plane 8 KB | 8-bit plane 8 KB]
    uint8_t *Ms = reinterpret_cast<uint8_t *>(lds + g8_mask_off(KT, WM));
    float *Ps = reinterpret_cast<float *>(lds + g8_mask_off(KT, WM) + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // XCD-aware tile order: every XCD gets a contiguous run of logical tile ids, column tiles fastest
    const int nwg = p.n_mt * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = wg / p.n_nt, nt = wg - mt * p.n_nt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    const int span = (KT - 1) * p.dil;
    const int left = span >> 1;
    const int n_stages = p.n_chunks * KT;
    const int goff = (int)((m0 - left) & 15);          // LDS row lr <-> global row gr: (gr & 15) == (lr + goff) & 15

    if (tid < BM) {
        const long gr = m0 + tid;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    }
    if (tid >= BM && tid < BM + BN) {                  // [bias | BN scale | BN shift | alpha] of the tile's columns
        const int c = tid - BM, gc = n0 + c;
        const bool ok = gc < p.cout;
        Ps[c] = (ok && p.bias) ? p.bias[gc] : 0.f;
        Ps[BN + c] = ok ? (p.scale ? p.scale[gc] : 1.f) : 0.f;
        Ps[2 * BN + c] = (ok && p.shift) ? p.shift[gc] : 0.f;
        Ps[3 * BN + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.alpha[0]
                       : (p.act == XV_ACT_PRELU && ok) ? p.alpha[gc] : 0.f;
    }

    const size_t xrow_bytes = (size_t)p.xchunks * SROW;
    const int arow0 = wr * 64 + (lane & 31);
    const int brow = wc * 64 + (lane & 31);
    const int kh = lane >> 5;
    const int boff0 = brow * 64;
    const int bsw0 = (brow >> 2) & 3;                  // (brow + 32) has the same swizzle

    // ---- prologue: the first RING weight tiles and the first halo tile(s) ------------------------------------------
    const uint8_t *bbase = p.wt + (size_t)nt * n_stages * B_BYTES + wave * (BP * 1024) + lane * 16;
    auto dma_b = [&](int stage, int buf) {
        const uint8_t *src = bbase + (size_t)(stage < n_stages ? stage : n_stages - 1) * B_BYTES;
        char *dst = Bbuf + buf * B_BYTES + wave * (BP * 1024);
#pragma unroll
        for (int j = 0; j < BP; ++j) XV_GLDS16(src + j * 1024, dst + j * 1024);
    };
    const uint8_t *abase = p.x + (m0 - left + (lane >> 3)) * (long)xrow_bytes + (lane & 7) * 16;
    constexpr int NP = KT == 1 ? BM / 8 : BM / 8 + 1;   // 8-row (1 KB) pieces of a halo tile
    auto dma_a_slab = [&](int chunk) {                  // whole halo tile of one slab (prologue only)
        char *dst = Abuf + (chunk & 1) * A_BYTES;
        for (int piece = wave; piece < NP; piece += NW)
            XV_GLDS16(abase + (size_t)chunk * SROW + (size_t)piece * 8 * xrow_bytes, dst + piece * 1024);
    };
#pragma unroll
    for (int j = 0; j < RING; ++j) dma_b(j, j);
    dma_a_slab(0);
    if (KT == 1 && n_stages > 1) dma_a_slab(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    struct FragsH {           // fp16 fragments of both k-steps of a stage: 8 x 4 VGPRs
        xv_f16x8 a0[2], a1[2], b0[2], b1[2];
    };
    struct FragsX {           // 8-bit fragments of a stage: 4 x 8 VGPRs
        xv_i32x8 a0, a1, b0, b1;
    };
    int scale_a = XV_SPLIT8_E8M0, scale_b = 127;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));        // keep them in VGPRs (a literal would be read as an fp32 constant)

    // per-tap, per-lane fragment addresses in A buffer 0.  Slot T of a row sits at ((T ^ sw) << 4):
    //   fp16, k-step ks:  T = 2 ks + kh        ->  pa ^ (ks << 5)
    //   8-bit, part e:    T = 4 + 2 kh + e     ->  px ^ (e << 4)
    int pa[KT], px[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int lr0 = arow0 + t * p.dil;
        const int sw = ((lr0 + goff) & 15) >> 1;
        pa[t] = lr0 * SROW + ((sw ^ kh) << 4);
        px[t] = lr0 * SROW + ((sw ^ (4 + 2 * kh)) << 4);
    }
    int pb[2];
    pb[0] = 2 * A_BYTES + boff0 + ((kh ^ bsw0) << 4);
    pb[1] = 2 * A_BYTES + boff0 + (((2 + kh) ^ bsw0) << 4);
    const int pbx = 2 * A_BYTES + B_PLANE + boff0 + (((2 * kh) ^ bsw0) << 4);

    auto load_h = [&](FragsH &X, int abase_, int bbase) {        // abase_ = pa[t] + A buffer offset, bbase = B buffer offset
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char *a = lds + (abase_ ^ (ks << 5));
            const char *b = lds + pb[ks] + bbase;
            X.a0[ks] = *reinterpret_cast<const xv_f16x8 *>(a);
            X.a1[ks] = *reinterpret_cast<const xv_f16x8 *>(a + 32 * SROW);
            X.b0[ks] = *reinterpret_cast<const xv_f16x8 *>(b);
            X.b1[ks] = *reinterpret_cast<const xv_f16x8 *>(b + 32 * 64);
        }
    };
    auto cat = [](xv_i32x4 u, xv_i32x4 v) { return __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7); };
    auto load_x = [&](FragsX &X, int abase_, int bbase) {        // abase_ = px[t] + A buffer offset
        const char *a = lds + abase_, *a2 = lds + (abase_ ^ 16);
        const char *b = lds + pbx + bbase, *b2 = lds + ((pbx + bbase) ^ 16);
        X.a0 = cat(*reinterpret_cast<const xv_i32x4 *>(a), *reinterpret_cast<const xv_i32x4 *>(a2));
        X.a1 = cat(*reinterpret_cast<const xv_i32x4 *>(a + 32 * SROW), *reinterpret_cast<const xv_i32x4 *>(a2 + 32 * SROW));
        X.b0 = cat(*reinterpret_cast<const xv_i32x4 *>(b), *reinterpret_cast<const xv_i32x4 *>(b2));
        X.b1 = cat(*reinterpret_cast<const xv_i32x4 *>(b + 32 * 64), *reinterpret_cast<const xv_i32x4 *>(b2 + 32 * 64));
    };
    auto mma_h = [&](const FragsH &X) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(X.a0[ks], X.b0[ks], acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(X.a0[ks], X.b1[ks], acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(X.a1[ks], X.b0[ks], acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(X.a1[ks], X.b1[ks], acc11, 0, 0, 0);
        }
    };
    auto mma_x = [&](const FragsX &X) {                          // cbsz = blgp = 1: both operands bf8 (e5m2)
        acc00 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(X.a0, X.b0, acc00, 1, 1, 0, scale_a, 0, scale_b);
        acc01 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(X.a0, X.b1, acc01, 1, 1, 0, scale_a, 0, scale_b);
        acc10 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(X.a1, X.b0, acc10, 1, 1, 0, scale_a, 0, scale_b);
        acc11 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(X.a1, X.b1, acc11, 1, 1, 0, scale_a, 0, scale_b);
    };

    // A-halo DMA schedule (as in tdnn_gemm_bf16x3_kernel): the NS one-piece-per-wave slots of the NEXT slab's halo tile are
    // dealt to taps 0..K-2 of the current slab as evenly as possible, early taps first
    constexpr int DT = KT == 1 ? 1 : KT - 1;
    constexpr int NS = (NP + NW - 1) / NW;
    constexpr int PW = (NS + DT - 1) / DT;
    auto slots_of = [](int t) constexpr { return t < DT ? (NS + DT - 1 - t) / DT : 0; };
    auto slot_base = [](int t) constexpr { int b = 0; for (int u = 0; u < t; ++u) b += (NS + DT - 1 - u) / DT; return b; };
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
    const uint8_t *bnext = bbase + (size_t)(RING < n_stages ? RING : n_stages - 1) * B_BYTES;     // tile of stage min(s+RING, last)
    int bcur = 0, bnxt = B_BYTES;                        // ring offsets of the weight tiles of stages s and s+1

    FragsH F;
    FragsX G;
    load_h(F, pa[0], 0);

    int s = 0;
    for (int c = 0; c < p.n_chunks; ++c) {
        const int abuf = (c & 1) * A_BYTES;
        const int abuf_n = A_BYTES - abuf;
        const int cn = (c + 1 < p.n_chunks) ? c + 1 : p.n_chunks - 1;
        const uint8_t *anext = abase + (size_t)cn * SROW;
        char *adst_n = Abuf + (cn & 1) * A_BYTES;
        auto tap = [&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int bbuf = bcur;
            // ---- phase 1: G <- 8-bit fragments of stage s, interleaved with the 8 fp16 MFMAs on F -------------------
            load_x(G, px[t] + abuf, bbuf);
            mma_h(F);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                  // stage s fully read by everybody, stage s+1 landed
            // ---- phase 2: DMA(s+2), F <- fp16 fragments of stage s+1, 4 scaled 8-bit MFMAs on G -----------------------
            {
                char *dst = Bbuf + bbuf + wave * (BP * 1024);    // one M0; the immediate advances source AND destination
                XV_GLDS16_OFF(bnext, dst, 0);
                XV_GLDS16_OFF(bnext, dst, 1024);
                if constexpr (BP == 4) {
                    XV_GLDS16_OFF(bnext, dst, 2048);
                    XV_GLDS16_OFF(bnext, dst, 3072);
                }
                bnext += (s + RING + 1 < n_stages) ? B_BYTES : 0;
            }
            if constexpr (KT == 1) {
                const int ca = (s + 2 < p.n_chunks) ? s + 2 : p.n_chunks - 1;
                const uint8_t *ag = abase + (size_t)ca * SROW;
                char *adst = Abuf + (ca & 1) * A_BYTES;
#pragma unroll
                for (int j = 0; j < PW; ++j) XV_GLDS16(ag + ag_off[0][j], adst + al_off[0][j]);
            } else if constexpr (t < DT) {
#pragma unroll
                for (int j = 0; j < slots_of(t); ++j) XV_GLDS16(anext + ag_off[t][j], adst_n + al_off[t][j]);
            }
            if constexpr (t + 1 < KT) load_h(F, pa[t + 1] + abuf, bnxt);
            else load_h(F, pa[0] + abuf_n, bnxt);                 // first tap of the next slab (tail: harmless read)
            mma_x(G);
            constexpr int NV = BP + (KT == 1 ? PW : slots_of(t));
            // per scaled MFMA (64 cycles): its share of the LDS-DMA pieces and 2 DS reads
#define XV_G8_GROUP(i)                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
            if constexpr ((NV + 3 - (i)) / 4 > 0) __builtin_amdgcn_sched_group_barrier(0x020, (NV + 3 - (i)) / 4, 0); \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            XV_G8_GROUP(0) XV_G8_GROUP(1) XV_G8_GROUP(2) XV_G8_GROUP(3)
#undef XV_G8_GROUP
            __builtin_amdgcn_sched_barrier(0);
            ++s;
            bcur = bnxt;
            bnxt = bnxt + B_BYTES == RING * B_BYTES ? 0 : bnxt + B_BYTES;
        };
        for_taps<0, KT>(tap);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- epilogue: accumulators -> LDS fp32 tile (the operand buffers are dead after the last barrier) ------
    float *T = reinterpret_cast<float *>(lds);
    {
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
    const bool lrelu = p.act == XV_ACT_LRELU;           // tf.nn.leaky_relu is max(alpha*z, z) for ANY alpha
    auto act3 = [&](auto MODE, float z, float a) {     // 0: max(z,0) + alpha*min(z,0)   1: leaky max(alpha*z, z)   2: plain ReLU
        constexpr int mode = decltype(MODE)::value;
        return mode == 1 ? fmaxf(a * z, z) : mode == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + a * fminf(z, 0.f);
    };
    auto by_mode = [&](auto &&f) {
        if (lrelu) f(std::integral_constant<int, 1>{});
        else if (p.act == XV_ACT_RELU) f(std::integral_constant<int, 2>{});
        else f(std::integral_constant<int, 0>{});
    };
    if constexpr (POOL) {
        // thread = (8-row block of the tile, 8 channels): (mean, M2) of the block's valid rows, shifted by its first row
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
        by_mode([&](auto MODE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                n += keep[j];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float z = (i < 4 ? tv[j][0][i] : tv[j][1][i - 4]) + bias[i];
                    const float v = act3(MODE, z, al[i]) * sc[i] + sh[i];
                    if (j == 0) { v0[i] = v; s1[i] = 0.f; s2[i] = 0.f; }
                    else {
                        const float d = keep[j] != 0.f ? v - v0[i] : 0.f;      // (a select: a row past R may hold NaN, and NaN * 0 is NaN)
                        s1[i] += d;
                        s2[i] += d * d;
                    }
                }
            }
        });
        const float rn = n > 0.f ? 1.f / n : 0.f;
        float mean[8], m2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mean[i] = n > 0.f ? v0[i] + s1[i] * rn : 0.f;
            m2[i] = fmaxf(s2[i] - s1[i] * s1[i] * rn, 0.f);
        }
        float *o = p.blk + ((size_t)((m0 >> 3) + blk) * 2) * p.cout + gc0;
        if (gc0 + 8 <= p.cout && !(p.cout & 3)) {
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
    } else {
        // all 16 LDS reads first (a round trip costs ~1 us under load -- pay it once, not once per row)
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
        if (p.y_format != XV_FMT_F32 && n0 + BN <= p.cout) {
            // hidden layers: full-width tile into a split format.  Rows >= R of the last tile land in the buffer's zero
            // padding (XV_SPLIT_PAD_AFTER >= BM) and are written as zeros (keep == 0).
            const int ch = gc0 >> 5, slot = cg & 3;
            char *ybase = reinterpret_cast<char *>(p.y) + (size_t)ch * SROW;
            const size_t yrow = (size_t)p.ychunks * SROW;
            float amax = 0.f;
            auto rows = [&](auto MODE, auto Y8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const long gr = m0 + (tid >> 4) + (NT / 16) * j;
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float z = (i < 4 ? tv[j][0][i] : tv[j][1][i - 4]) + bias[i];
                        v[i] = keep[j] != 0.f ? act3(MODE, z, al[i]) * sc[i] + sh[i] : 0.f;
                    }
                    const int sw = (int)(gr >> 1) & 7;
                    char *row = ybase + (size_t)gr * yrow;
                    if constexpr (decltype(Y8)::value) {
                        xv_f16x8 hi;
                        xv_i32x4 x8;
                        xv_split8_encode8<true>(v, hi, x8, amax);
                        __builtin_nontemporal_store(hi, reinterpret_cast<xv_f16x8 *>(row + ((slot ^ sw) << 4)));
                        __builtin_nontemporal_store(x8, reinterpret_cast<xv_i32x4 *>(row + (((4 + slot) ^ sw) << 4)));
                    } else {
                        bf16x8 hi, lo;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            hi[i] = (__bf16)v[i];
                            lo[i] = (__bf16)(v[i] - (float)hi[i]);
                        }
                        __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8 *>(row + ((slot ^ sw) << 4)));
                        __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8 *>(row + (((4 + slot) ^ sw) << 4)));
                    }
                }
            };
            if (p.y_format == XV_FMT_SPLIT8) {
                by_mode([&](auto MODE) { rows(MODE, std::true_type{}); });
                if (amax > XV_SPLIT8_MAX && p.status) atomicOr(p.status, 1);
            } else {
                by_mode([&](auto MODE) { rows(MODE, std::false_type{}); });
            }
            return;
        }
        // general path: fp32 rows, or a ragged last column tile of a split output
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lr = (tid >> 4) + (NT / 16) * j;
            const long gr = m0 + lr;
            if (gr >= p.R) continue;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float z = (i < 4 ? tv[j][0][i] : tv[j][1][i - 4]) + bias[i];
                const float a = lrelu ? fmaxf(al[i] * z, z) : fmaxf(z, 0.f) + al[i] * fminf(z, 0.f);
                v[i] = keep[j] != 0.f ? a * sc[i] + sh[i] : 0.f;
            }
            if (p.y_format == XV_FMT_F32) {
                float *o = reinterpret_cast<float *>(p.y) + (size_t)gr * p.ldy + gc0;
                if (gc0 + 8 <= p.cout && !(p.ldy & 3)) {
                    __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(o));
                    __builtin_nontemporal_store((f32x4){v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4 *>(o + 4));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (gc0 + i < p.cout) o[i] = v[i];
                }
            } else {
                const int ch = gc0 >> 5;
                if (ch >= p.ychunks) continue;
                const int sw = (int)(gr >> 1) & 7, slot = cg & 3;
                char *row = reinterpret_cast<char *>(p.y) + ((size_t)gr * p.ychunks + ch) * SROW;
                if (p.y_format == XV_FMT_SPLIT8) {
                    xv_f16x8 hi;
                    xv_i32x4 x8;
                    xv_split8_encode8<true>(v, hi, x8, amax);
                    *reinterpret_cast<xv_f16x8 *>(row + ((slot ^ sw) << 4)) = hi;
                    *reinterpret_cast<xv_i32x4 *>(row + (((4 + slot) ^ sw) << 4)) = x8;
                } else {
                    bf16x8 hi, lo;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        hi[i] = (__bf16)v[i];
                        lo[i] = (__bf16)(v[i] - (float)hi[i]);
                    }
                    *reinterpret_cast<bf16x8 *>(row + ((slot ^ sw) << 4)) = hi;
                    *reinterpret_cast<bf16x8 *>(row + (((4 + slot) ^ sw) << 4)) = lo;
                }
            }
        }
        if (amax > XV_SPLIT8_MAX && p.status) atomicOr(p.status, 1);
    }
}

// ------------------------------------------------------------------------------------------------
// The same layer on a 256 x 256 workgroup tile: 8 waves as 2 x 4, each a 128 x 64 sub-tile (4 x 2 MFMA tiles, 128
// accumulator registers).  With 64 x 64 sub-tiles this arithmetic needs 147 LDS bytes per MFMA cycle -- more than the 128
// the LDS delivers (halving the fragment reads made the kernel 14 % faster); 4 x 2 sub-tiles need 24 fragment reads per
// 1024 MFMA cycles instead of 16 per 512, and a 32 KB weight stage feeds twice the MFMAs: 113 bytes per cycle.
// The fragment sets are pipelined in four steps per stage so that at most four 16-register sets are live next to the
// accumulators (t / b = row tiles 0,1 / 2,3 of the wave; H = fp16 fragments of both k-steps, X = 8-bit fragments):
//     step 1   8 fp16 MFMAs (top,    BH)   | read AHb(s), BX(s)
//     step 2   8 fp16 MFMAs (bottom, BH)   | read AXt(s)
//     barrier  [every B fragment of stage s is in registers; stage s+1 has landed]
//     step 3   4 scaled MFMAs (top,    BX) | DMA of stage s+2 | read AXb(s), BH(s+1)
//     step 4   4 scaled MFMAs (bottom, BX) | read AHt(s+1)
// A fragments may be read after the barrier because a halo buffer is only rewritten one slab later (K > 1 only; the K = 1
// layers keep the narrow kernel).  Cout % 256 == 0, split-format or POOL output (the launcher falls back otherwise).
// The epilogue goes through the LDS in two halves of 128 rows (the fp32 tile of a half is exactly the operand area).
// ------------------------------------------------------------------------------------------------
constexpr int W_BM = 256, W_BN = 256, W_TLD = W_BN + 4;
constexpr int W_A_BYTES = (W_BM + MAX_SPAN) * SROW;        // 33792
constexpr int W_B_BYTES = 2 * B_BYTES;                     // 32768: two 128-column weight tiles
constexpr int W_OPER = 2 * W_A_BYTES + 2 * W_B_BYTES;      // 133120
constexpr int W_TILE = 128 * W_TLD * 4;                    // 133120
constexpr int W_MASK_OFF = W_OPER > W_TILE ? W_OPER : W_TILE;
constexpr size_t W_LDS_BYTES = (size_t)W_MASK_OFF + W_BM + 4 * W_BN * sizeof(float);

// Epilogue of the 256 x 256 kernels (both MFMA shapes), 128 rows at a time through an fp32 tile that re-uses the operand area:
// write_tile(T) puts the calling wave's 128 x 64 accumulators into the tile of its half.
template <bool POOL, class WriteTile>
__device__ __forceinline__ void wide_epilogue(const Gemm8Params &p, char *lds, const uint8_t *Ms, const float *Ps, long m0, int n0,
                                              int tid, int wr, WriteTile &&write_tile)
{
    float *T = reinterpret_cast<float *>(lds);
    const int cg = tid & 31;                            // 8-channel group of the 256-column tile
    const int gc0 = n0 + cg * 8;
    float bias[8], sc[8], sh[8], al[8];
    {
        const f32x4 *P4 = reinterpret_cast<const f32x4 *>(Ps) + cg * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bias[i] = P4[0][i]; bias[4 + i] = P4[1][i];
            sc[i] = P4[W_BN / 4][i]; sc[4 + i] = P4[W_BN / 4 + 1][i];
            sh[i] = P4[2 * W_BN / 4][i]; sh[4 + i] = P4[2 * W_BN / 4 + 1][i];
            al[i] = P4[3 * W_BN / 4][i]; al[4 + i] = P4[3 * W_BN / 4 + 1][i];
        }
    }
    const bool lrelu = p.act == XV_ACT_LRELU;
    auto act3 = [&](auto MODE, float z, float a) {
        constexpr int mode = decltype(MODE)::value;
        return mode == 1 ? fmaxf(a * z, z) : mode == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + a * fminf(z, 0.f);
    };
    auto by_mode = [&](auto &&f) {
        if (lrelu) f(std::integral_constant<int, 1>{});
        else if (p.act == XV_ACT_RELU) f(std::integral_constant<int, 2>{});
        else f(std::integral_constant<int, 0>{});
    };
    float amax = 0.f;
    f32x4 tv[8][2];
    float keep[8];
    // thread -> 8 rows x 8 channels of a half: POOL: the 8 rows of block tid >> 5; else rows (tid >> 5) + 16 j
    auto read_tile = [&](int h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lr = POOL ? (tid >> 5) * 8 + j : (tid >> 5) + 16 * j;
            tv[j][0] = *reinterpret_cast<const f32x4 *>(T + lr * W_TLD + cg * 8);
            tv[j][1] = *reinterpret_cast<const f32x4 *>(T + lr * W_TLD + cg * 8 + 4);
            keep[j] = Ms[h * 128 + lr] ? 1.f : 0.f;
        }
    };
    auto process = [&](int h) {
        const long mh = m0 + h * 128;
        if constexpr (POOL) {
            const int blk = tid >> 5;                   // 16 blocks of 8 rows
            if (mh + blk * 8 >= p.R) return;
            float v0[8], s1[8], s2[8];
            float n = 0.f;
            by_mode([&](auto MODE) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    n += keep[j];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float z = (i < 4 ? tv[j][0][i] : tv[j][1][i - 4]) + bias[i];
                        const float v = act3(MODE, z, al[i]) * sc[i] + sh[i];
                        if (j == 0) { v0[i] = v; s1[i] = 0.f; s2[i] = 0.f; }
                        else {
                            const float d = keep[j] != 0.f ? v - v0[i] : 0.f;      // (a select: a row past R may hold NaN, and NaN * 0 is NaN)
                            s1[i] += d;
                            s2[i] += d * d;
                        }
                    }
                }
            });
            const float rn = n > 0.f ? 1.f / n : 0.f;
            f32x4 mean[2], m2[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                mean[i >> 2][i & 3] = n > 0.f ? v0[i] + s1[i] * rn : 0.f;
                m2[i >> 2][i & 3] = fmaxf(s2[i] - s1[i] * s1[i] * rn, 0.f);
            }
            float *o = p.blk + ((size_t)((mh >> 3) + blk) * 2) * p.cout + gc0;
            *reinterpret_cast<f32x4 *>(o) = mean[0];
            *reinterpret_cast<f32x4 *>(o + 4) = mean[1];
            *reinterpret_cast<f32x4 *>(o + p.cout) = m2[0];
            *reinterpret_cast<f32x4 *>(o + p.cout + 4) = m2[1];
        } else {
            const int ch = gc0 >> 5, slot = cg & 3;
            char *ybase = reinterpret_cast<char *>(p.y) + (size_t)ch * SROW;
            const size_t yrow = (size_t)p.ychunks * SROW;
            auto rows = [&](auto MODE, auto Y8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const long gr = mh + (tid >> 5) + 16 * j;
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float z = (i < 4 ? tv[j][0][i] : tv[j][1][i - 4]) + bias[i];
                        v[i] = keep[j] != 0.f ? act3(MODE, z, al[i]) * sc[i] + sh[i] : 0.f;
                    }
                    const int sw = (int)(gr >> 1) & 7;
                    char *row = ybase + (size_t)gr * yrow;
                    if constexpr (decltype(Y8)::value) {
                        xv_f16x8 hi;
                        xv_i32x4 x8;
                        xv_split8_encode8<true>(v, hi, x8, amax);
                        *reinterpret_cast<xv_f16x8 *>(row + ((slot ^ sw) << 4)) = hi;              // (plain, not non-temporal: see DESIGN 3.1e)
                        *reinterpret_cast<xv_i32x4 *>(row + (((4 + slot) ^ sw) << 4)) = x8;
                    } else {
                        bf16x8 hi, lo;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            hi[i] = (__bf16)v[i];
                            lo[i] = (__bf16)(v[i] - (float)hi[i]);
                        }
                        __builtin_nontemporal_store(hi, reinterpret_cast<bf16x8 *>(row + ((slot ^ sw) << 4)));
                        __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8 *>(row + (((4 + slot) ^ sw) << 4)));
                    }
                }
            };
            if (p.y_format == XV_FMT_SPLIT8) by_mode([&](auto MODE) { rows(MODE, std::true_type{}); });
            else by_mode([&](auto MODE) { rows(MODE, std::false_type{}); });
        }
    };
    // upper half through the tile; the lower half's accumulators go into the tile as soon as the upper half has been read
    // into registers, i.e. BEFORE the arithmetic of the upper half (128 accumulators + 64 tile values + the arithmetic of
    // an epilogue do not fit the register file)
    __syncthreads();                                    // operand buffers are dead
    if (wr == 0) write_tile(T);
    __syncthreads();
    read_tile(0);
    __syncthreads();
    if (wr == 1) write_tile(T);
    __builtin_amdgcn_sched_barrier(0);
    process(0);
    __syncthreads();
    read_tile(1);
    __builtin_amdgcn_sched_barrier(0);
    process(1);
    if constexpr (!POOL)
        if (amax > XV_SPLIT8_MAX && p.status) atomicOr(p.status, 1);
}


template <int KT, bool POOL>
__global__ __launch_bounds__(512, 2) void tdnn_gemm_f16bf8_wide_kernel(const Gemm8Params p)
{
    static_assert(KT > 1, "the wide kernel reads A fragments after the stage barrier: K > 1 only");
    constexpr int NW = 8;
    constexpr int BP = 4;                              // 1 KB pieces of a 32 KB weight stage per wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Abuf = lds;
    char *Bbuf = lds + 2 * W_A_BYTES;
    uint8_t *Ms = reinterpret_cast<uint8_t *>(lds + W_MASK_OFF);
    float *Ps = reinterpret_cast<float *>(lds + W_MASK_OFF + W_BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nwg = p.n_mt * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int mt = wg / p.n_nt, nt = wg - mt * p.n_nt;
    const long m0 = (long)mt * W_BM;
    const int n0 = nt * W_BN;

    const int span = (KT - 1) * p.dil;
    const int left = span >> 1;
    const int n_stages = p.n_chunks * KT;
    const int goff = (int)((m0 - left) & 15);

    if (tid < W_BM) {
        const long gr = m0 + tid;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    } else {
        const int c = tid - W_BM, gc = n0 + c;           // Cout % 256 == 0: every column exists
        Ps[c] = p.bias ? p.bias[gc] : 0.f;
        Ps[W_BN + c] = p.scale ? p.scale[gc] : 1.f;
        Ps[2 * W_BN + c] = p.shift ? p.shift[gc] : 0.f;
        Ps[3 * W_BN + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.alpha[0] : p.act == XV_ACT_PRELU ? p.alpha[gc] : 0.f;
    }

    const size_t xrow_bytes = (size_t)p.xchunks * SROW;
    const int arow0 = wr * 128 + (lane & 31);
    const int bcol = wc * 64 + (lane & 31);            // column of the 256-column stage; + 32 stays inside its 128-column tile
    const int kh = lane >> 5;
    const int boff0 = (bcol >> 7) * B_BYTES + (bcol & 127) * 64;
    const int bsw0 = ((bcol & 127) >> 2) & 3;

    // weight stage = the tiles of column tiles 2 nt and 2 nt + 1; wave w moves pieces 4w .. 4w+3 of its 32 KB
    const uint8_t *bbase = p.wt + ((size_t)(2 * nt + (wave >> 2)) * n_stages) * B_BYTES + (wave & 3) * 4096 + lane * 16;
    const int bdst = (wave >> 2) * B_BYTES + (wave & 3) * 4096;
    const uint8_t *abase = p.x + (m0 - left + (lane >> 3)) * (long)xrow_bytes + (lane & 7) * 16;
    constexpr int NP = W_BM / 8 + 1;
    {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint8_t *src = bbase + (size_t)(j < n_stages ? j : n_stages - 1) * B_BYTES;
            char *dst = Bbuf + j * W_B_BYTES + bdst;
#pragma unroll
            for (int i = 0; i < BP; ++i) XV_GLDS16(src + i * 1024, dst + i * 1024);
        }
        for (int piece = wave; piece < NP; piece += NW) XV_GLDS16(abase + (size_t)piece * 8 * xrow_bytes, Abuf + piece * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    struct SetH { xv_f16x8 f0[2], f1[2]; };            // two MFMA tiles x two k-steps
    struct SetX { xv_i32x8 f0, f1; };
    int scale_a = XV_SPLIT8_E8M0, scale_b = 127;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));

    int pa[KT], px[KT];                                // row tile 0 of the wave; tile i is + i * 32 rows (same swizzle)
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int lr0 = arow0 + t * p.dil;
        const int sw = ((lr0 + goff) & 15) >> 1;
        pa[t] = lr0 * SROW + ((sw ^ kh) << 4);
        px[t] = lr0 * SROW + ((sw ^ (4 + 2 * kh)) << 4);
    }
    const int pb0 = 2 * W_A_BYTES + boff0 + ((kh ^ bsw0) << 4);
    const int pb1 = 2 * W_A_BYTES + boff0 + (((2 + kh) ^ bsw0) << 4);
    const int pbx = 2 * W_A_BYTES + B_PLANE + boff0 + (((2 * kh) ^ bsw0) << 4);

    auto load_ah = [&](SetH &X, int base) {             // base = pa[t] + A buffer offset + (0 | 64 rows)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char *a = lds + (base ^ (ks << 5));
            X.f0[ks] = *reinterpret_cast<const xv_f16x8 *>(a);
            X.f1[ks] = *reinterpret_cast<const xv_f16x8 *>(a + 32 * SROW);
        }
    };
    auto load_bh = [&](SetH &X, int bbase_) {
        X.f0[0] = *reinterpret_cast<const xv_f16x8 *>(lds + pb0 + bbase_);
        X.f1[0] = *reinterpret_cast<const xv_f16x8 *>(lds + pb0 + bbase_ + 32 * 64);
        X.f0[1] = *reinterpret_cast<const xv_f16x8 *>(lds + pb1 + bbase_);
        X.f1[1] = *reinterpret_cast<const xv_f16x8 *>(lds + pb1 + bbase_ + 32 * 64);
    };
    auto cat = [](xv_i32x4 u, xv_i32x4 v) { return __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7); };
    auto load_ax = [&](SetX &X, int base) {             // base = px[t] + A buffer offset + (0 | 64 rows)
        const char *a = lds + base, *a2 = lds + (base ^ 16);
        X.f0 = cat(*reinterpret_cast<const xv_i32x4 *>(a), *reinterpret_cast<const xv_i32x4 *>(a2));
        X.f1 = cat(*reinterpret_cast<const xv_i32x4 *>(a + 32 * SROW), *reinterpret_cast<const xv_i32x4 *>(a2 + 32 * SROW));
    };
    auto load_bx = [&](SetX &X, int bbase_) {
        const char *b = lds + pbx + bbase_, *b2 = lds + ((pbx + bbase_) ^ 16);
        X.f0 = cat(*reinterpret_cast<const xv_i32x4 *>(b), *reinterpret_cast<const xv_i32x4 *>(b2));
        X.f1 = cat(*reinterpret_cast<const xv_i32x4 *>(b + 32 * 64), *reinterpret_cast<const xv_i32x4 *>(b2 + 32 * 64));
    };
    auto mma_h = [&](const SetH &A, const SetH &B, int i0) {      // row tiles i0, i0+1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            acc[i0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.f0[ks], B.f0[ks], acc[i0][0], 0, 0, 0);
            acc[i0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.f0[ks], B.f1[ks], acc[i0][1], 0, 0, 0);
            acc[i0 + 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.f1[ks], B.f0[ks], acc[i0 + 1][0], 0, 0, 0);
            acc[i0 + 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.f1[ks], B.f1[ks], acc[i0 + 1][1], 0, 0, 0);
        }
    };
    auto mma_x = [&](const SetX &A, const SetX &B, int i0) {
        acc[i0][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A.f0, B.f0, acc[i0][0], 1, 1, 0, scale_a, 0, scale_b);
        acc[i0][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A.f0, B.f1, acc[i0][1], 1, 1, 0, scale_a, 0, scale_b);
        acc[i0 + 1][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A.f1, B.f0, acc[i0 + 1][0], 1, 1, 0, scale_a, 0, scale_b);
        acc[i0 + 1][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A.f1, B.f1, acc[i0 + 1][1], 1, 1, 0, scale_a, 0, scale_b);
    };

    // In the main loop only waves 0-3 -- one per SIMD (a workgroup's waves go to the SIMDs cyclically: w and w + 4 share one) --
    // issue DMA.  An LDS-DMA instruction holds a wave's issue for 60-180 cycles; when all eight waves issue their pieces behind
    // the same barrier, both waves of every SIMD stand still together and the MFMA pipe with them.  With one issuing wave per
    // SIMD the other one keeps the pipe busy, and the issuing wave catches up while its partner waits at the next barrier.
    constexpr int NI = 4;                              // issuing waves
    constexpr int DT = KT - 1;
    constexpr int NS = (NP + NI - 1) / NI;
    constexpr int PW = (NS + DT - 1) / DT;
    auto slots_of = [](int t) constexpr { return t < DT ? (NS + DT - 1 - t) / DT : 0; };
    auto slot_base = [](int t) constexpr { int b = 0; for (int u = 0; u < t; ++u) b += (NS + DT - 1 - u) / DT; return b; };
    const uint32_t rowstep = 8u * (uint32_t)xrow_bytes;
    uint32_t ag_off[DT][PW], al_off[DT][PW];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            int piece = (slot_base(t) + j) * NI + wave;
            piece = piece < NP ? piece : NP - 1;
            ag_off[t][j] = (uint32_t)piece * rowstep;
            al_off[t][j] = (uint32_t)piece * 1024u;
        }
    // (main loop: wave w < 4 moves pieces 4w .. 4w+3 of BOTH 16 KB weight tiles of a stage)
    const uint8_t *bnext = p.wt + ((size_t)(2 * nt) * n_stages) * B_BYTES + (wave & 3) * 4096 + lane * 16 +
                           (size_t)(2 < n_stages ? 2 : n_stages - 1) * B_BYTES;
    const size_t btile = (size_t)n_stages * B_BYTES;   // from column tile 2 nt to 2 nt + 1

    constexpr int HALF = 64 * SROW;                    // row tiles 2,3 of the wave
    SetH AHt, AHb, BH;
    SetX AXt, AXb, BX;
    load_ah(AHt, pa[0]);
    load_bh(BH, 0);

    int s = 0;
    for (int c = 0; c < p.n_chunks; ++c) {
        const int abuf = (c & 1) * W_A_BYTES;
        const int abuf_n = W_A_BYTES - abuf;
        const int cn = (c + 1 < p.n_chunks) ? c + 1 : p.n_chunks - 1;
        const uint8_t *anext = abase + (size_t)cn * SROW;
        char *adst_n = Abuf + (cn & 1) * W_A_BYTES;
        auto tap = [&](auto TT) {
            constexpr int t = decltype(TT)::value;
            const int bbuf = (s & 1) * W_B_BYTES;
            // ---- step 1 ----
            load_ah(AHb, pa[t] + abuf + HALF);
            load_bx(BX, bbuf);
            mma_h(AHt, BH, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 2 ----
            load_ax(AXt, px[t] + abuf);
            mma_h(AHb, BH, 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                  // every B fragment of stage s is in registers; stage s+1 has landed
            // ---- step 3 ----
            if (wave < NI) {
                char *dst = Bbuf + bbuf + (wave & 3) * 4096;
                XV_GLDS16_OFF(bnext, dst, 0);
                XV_GLDS16_OFF(bnext, dst, 1024);
                XV_GLDS16_OFF(bnext, dst, 2048);
                XV_GLDS16_OFF(bnext, dst, 3072);
                XV_GLDS16_OFF(bnext + btile, dst + B_BYTES, 0);
                XV_GLDS16_OFF(bnext + btile, dst + B_BYTES, 1024);
                XV_GLDS16_OFF(bnext + btile, dst + B_BYTES, 2048);
                XV_GLDS16_OFF(bnext + btile, dst + B_BYTES, 3072);
                if constexpr (t < DT) {
#pragma unroll
                    for (int j = 0; j < slots_of(t); ++j) XV_GLDS16(anext + ag_off[t][j], adst_n + al_off[t][j]);
                }
            }
            bnext += (s + 3 < n_stages) ? B_BYTES : 0;
            load_ax(AXb, px[t] + abuf + HALF);
            load_bh(BH, W_B_BYTES - bbuf);
            mma_x(AXt, BX, 0);
            constexpr int NV = 0;                       // (the DMA block above is a region of its own now)
#define XV_G8W_GROUP(i)                                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      \
            if constexpr ((NV + 3 - (i)) / 4 > 0) __builtin_amdgcn_sched_group_barrier(0x020, (NV + 3 - (i)) / 4, 0); \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            XV_G8W_GROUP(0) XV_G8W_GROUP(1) XV_G8W_GROUP(2) XV_G8W_GROUP(3)
#undef XV_G8W_GROUP
            __builtin_amdgcn_sched_barrier(0);
            // ---- step 4 ----
            if constexpr (t + 1 < KT) load_ah(AHt, pa[t + 1] + abuf);
            else load_ah(AHt, pa[0] + abuf_n);               // first tap of the next slab (tail: harmless read)
            mma_x(AXb, BX, 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            ++s;
        };
        for_taps<0, KT>(tap);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue, 128 rows at a time ---------------------------------------------------------------------------------
    auto write_tile = [&](float *T) {                   // this wave's 128 x 64 accumulators -> the fp32 tile of its half
        const int col = wc * 64 + (lane & 31);
        const int rowb = 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rr = i * 32 + rowb + (reg & 3) + 8 * (reg >> 2);
                T[rr * W_TLD + col] = acc[i][0][reg];
                T[rr * W_TLD + col + 32] = acc[i][1][reg];
            }
    };
    wide_epilogue<POOL>(p, lds, Ms, Ps, m0, n0, tid, wr, write_tile);
}

// ------------------------------------------------------------------------------------------------
// The 256 x 256 tile on the 16 x 16 MFMA shapes (round 4): v_mfma_f32_16x16x32_f16 + v_mfma_scale_f32_16x16x128_f8f6f4.
// On the power-limited chip a product costs ~8 % fewer joules on these shapes (tools/experiments/shape_probe.hip).  Same tile,
// same operand formats, the same bytes through LDS, the same 128 accumulator registers per wave (8 x 4 tiles of 16 x 16).
//   * fp16: a lane holds slot kb = lane >> 4 of row / column lane & 15 -- a 32-channel slab is ONE k-step.
//   * 8-bit: K = 128 is four 32-byte K blocks = the cross terms of TWO (slab, tap) items.  K block kb: item kb >> 1 of the pair,
//     slots 4 + (kb & 1) and 6 + (kb & 1) of the row-slab (channels 8c..8c+7 and 16+8c..16+8c+7, c = kb & 1; the weight tile holds
//     [h8 | l8] at the same positions).  Lanes 0-31 of an A / B fragment therefore read item 0's rows / weight tile, lanes 32-63
//     item 1's -- a per-lane address, nothing else.  Items are paired in stage order (slab-major, tap-minor); K is odd, so the
//     middle pair of two slabs straddles the slab boundary and the loop is unrolled over two slabs = K pairs.
//   * MFMA tile row i is row ROW16(i) of the 16, column j is COL16(j) (the permutations found for the bf16x3 S16 form: every
//     lane group of a ds_read_b128 touches 16 different 16-byte chunks for every tap offset; the 8-bit slots 4+c / 6+c differ
//     from the fp16 slots only by a constant XOR, which keeps that property).
//   * LDS: two halo buffers as before; the weight area (64 KB) is four 16 KB regions H0 | H1 | X0 | X1 = the fp16 / 8-bit planes
//     of the even / odd item of a pair (each [column tile 2nt: 8 KB][2nt+1: 8 KB]), fetched from the unchanged 16 KB weight
//     tiles.  All B fragments of a pair live in registers (BH 16, BX 32), A fragments are streamed in quarters of the wave's
//     128 rows (two sets of AH 8 / AX 16 registers).
// Per pair and wave, 2048 MFMA cycles, two barriers:
//     phase A   32 fp16 MFMAs (item 0, quarters 0-3)   | read BX(pair), AH quarters, AX q0, q1
//     barrier 1 [BX in registers, H1 of this pair landed]         -> DMA X0, X1, H0 of the next pair
//     phase B   16 scaled MFMAs (quarters 0, 1)        | read BH(item 1), AX q2, AH(item 1) q0, q1
//     phase C   32 fp16 MFMAs (item 1)                 | read AH quarters, AX q3
//     barrier 2 [X, H0 of the next pair landed]                   -> DMA H1 of the next pair, halo pieces
//     phase D   16 scaled MFMAs (quarters 2, 3)        | read BH, AH q0 of the next pair's item 0
// Needs an even number of slabs, K in {3, 5, 7}, Cout % 256 == 0, split-format or POOL output (else the 32 x 32 form).
// ------------------------------------------------------------------------------------------------
constexpr int W16_H = 2 * W_A_BYTES;                 // H0 (even item), H1 = + B_BYTES
constexpr int W16_X = 2 * W_A_BYTES + 2 * B_BYTES;   // X0, X1 = + B_BYTES

template <int KT, bool POOL>
__global__ __launch_bounds__(512, 2) void tdnn_gemm_f16bf8_wide16_kernel(const Gemm8Params p)
{
    static_assert(KT == 3 || KT == 5 || KT == 7, "pairs of (slab, tap) items over two slabs: K odd");
    constexpr int NW = 8;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Abuf = lds;
    uint8_t *Ms = reinterpret_cast<uint8_t *>(lds + W_MASK_OFF);
    float *Ps = reinterpret_cast<float *>(lds + W_MASK_OFF + W_BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int nwg = p.n_mt * p.n_nt;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q_ = nwg >> 3, r_ = nwg & 7;
    const int wg = (xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_) + idx;
    int mt = wg / p.n_nt, nt = wg - mt * p.n_nt;
    if (p.colmap) {
        // XCD-aware column placement (two column tiles): an XCD's L2 then holds ONE column tile's weights (3.7 MB of the K = 7
        // layer's 7.3 MB) and every row tile's operand rows are fetched by two XCDs.  Slot (column tile, quarter x4 of the row tiles)
        // = XCD 4 nt + x4; the <= 6 blocks the round-robin deals to other XCDs than the slots need take the slots' last tiles.
        const int q4 = p.n_mt >> 2, r4 = p.n_mt & 3;
        int x4 = xcd & 3;
        nt = xcd >> 2;
        if (idx >= q4) { nt = xcd / r4; x4 = xcd - nt * r4; }
        mt = x4 * q4 + (x4 < r4 ? x4 : r4) + idx;
    }
    const long m0 = (long)mt * W_BM;
    const int n0 = nt * W_BN;

    const int span = (KT - 1) * p.dil;
    const int left = span >> 1;
    const int n_stages = p.n_chunks * KT;              // (slab, tap) items; even
    const int n_pairs = n_stages >> 1;
    const int goff = (int)((m0 - left) & 15);

    if (tid < W_BM) {
        const long gr = m0 + tid;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    } else {
        const int c = tid - W_BM, gc = n0 + c;
        Ps[c] = p.bias ? p.bias[gc] : 0.f;
        Ps[W_BN + c] = p.scale ? p.scale[gc] : 1.f;
        Ps[2 * W_BN + c] = p.shift ? p.shift[gc] : 0.f;
        Ps[3 * W_BN + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.alpha[0] : p.act == XV_ACT_PRELU ? p.alpha[gc] : 0.f;
    }

    const size_t xrow_bytes = (size_t)p.xchunks * SROW;
    const int kb = lane >> 4, kc = kb & 1, kp = kb >> 1;
    const int row16 = (int)((0x48c67dbf391502eaull >> (4 * (lane & 15))) & 15);        // ROW16
    const int col16 = (int)((0xfedc76543210ba98ull >> (4 * (lane & 15))) & 15);        // COL16

    // ---- DMA streams (MUBUF buffer_load ... lds: an SGPR descriptor + ONE per-lane offset register per stream; the FLAT form needs a
    //      64-bit per-lane pointer and 64-bit VALU adds per piece).  A 16 KB weight tile (column tile 2nt + h, item u) =
    //      [fp16 plane: pieces 0-7][8-bit plane: 8-15].  EVERY wave moves the same share of a pair, one piece at a time between
    //      its MFMAs (an LDS-DMA instruction holds a wave's issue for 60-180 cycles: issued in bursts behind the barriers, the two
    //      waves of a SIMD stood still together and the MFMA pipe with them -- 13 % of the loop at full clock):
    //        8-bit planes (4 per pair: item e x column tile h): wave w moves pieces 4 (w >> 2) .. + 3 of plane e = w & 1, h = (w >> 1) & 1;
    //        fp16 planes (2 per item): pieces 2 hidx, 2 hidx + 1 of plane h, hidx = (w & 1) + 2 (w >> 2);  halo piece 8 slot + w.
    const int dh = (wave >> 1) & 1;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(p.wt) + ((size_t)(2 * nt + dh) * n_stages) * B_BYTES, 0, 0x7ffffff0, XV_RSRC_FLAGS);
    const int wvoff = lane * 16;
    const int hidx = (wave & 1) + 2 * (wave >> 2);
    const int xs_rel = (wave & 1) * B_BYTES + B_PLANE + (wave >> 2) * 4096;        // source offsets inside a pair's 32 KB
    const int hs_rel = hidx * 2048;
    char *const xdst = lds + W16_X + (wave & 1) * B_BYTES + dh * B_PLANE + (wave >> 2) * 4096;
    char *const hdst = lds + W16_H + dh * B_PLANE + hidx * 2048;
    int wsoff = 0;                                     // the pair in flight (items 2P, 2P+1)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(p.x) + (m0 - left) * (long)xrow_bytes, 0, 0x7ffffff0, XV_RSRC_FLAGS);
    const int xvoff = (lane >> 3) * (int)xrow_bytes + (lane & 7) * 16;     // (264 rows of a workgroup: far below 2^31 bytes)
    constexpr int NP = W_BM / 8 + 1;
    {   // prologue: pair 0 and the halo tile of slab 0
        XV_BLDS16_X4(wrs, xdst, wvoff, wsoff + xs_rel, 0);
        XV_BLDS16(wrs, hdst, wvoff, wsoff + hs_rel, 0);
        XV_BLDS16(wrs, hdst, wvoff, wsoff + hs_rel, 1024);
        XV_BLDS16(wrs, hdst + B_BYTES, wvoff, wsoff + hs_rel + B_BYTES, 0);
        XV_BLDS16(wrs, hdst + B_BYTES, wvoff, wsoff + hs_rel + B_BYTES, 1024);
        for (int piece = wave; piece < NP; piece += NW) XV_BLDS16(xrs, Abuf + piece * 1024, xvoff, piece * 8 * (int)xrow_bytes, 0);
    }
    wsoff += (n_pairs > 1) ? 2 * B_BYTES : 0;          // -> pair 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int scale_a = XV_SPLIT8_E8M0, scale_b = 127;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));

    // per-lane fragment addresses, computed per pair (held in registers they would be 2 K + 2 K values -- spills):
    //   fp16, tap t:  row wr*128 + ROW16 + t*dil, slot kb                   (+ 16 i rows: + 2048, the same swizzle)
    //   8-bit, pair:  lanes of K blocks 0,1 read the pair's first item, those of 2,3 the second: slot 4 + kc = the fp16 address
    //                 ^ ((4 ^ 2 kp) << 4); ^ 32 for slot 6 + kc
    const int rowbase = wr * 128 + row16;
    const int kb16 = kb << 4;
    const int xorc = (4 ^ (2 * kp)) << 4;
    auto addr_h = [&](int tdil, int rb) {               // rb: an opaque copy of rowbase made inside the loop (else "rowbase + t dil"
        const int lr0 = rb + tdil;                      // is hoisted out of the loop for every tap: K registers)
        return (lr0 << 7) + ((((lr0 + goff) & 14) << 3) ^ kb16);
    };
    auto addr_x = [&](int a0, int a1, int j) {          // pair j of the period from the fp16 addresses of its two items
        return (kp ? a1 + ((2 * j + 1) / KT) * W_A_BYTES : a0 + ((2 * j) / KT) * W_A_BYTES) ^ xorc;
    };
    const int colin = (wc & 1) * 64 + col16;
    const int bsw = (colin >> 2) & 3;
    const int pbh = W16_H + (wc >> 1) * B_PLANE + colin * 64 + ((kb ^ bsw) << 4);                 // + (u & 1) * B_BYTES, + 1024 j
    const int pbx = W16_X + kp * B_BYTES + (wc >> 1) * B_PLANE + colin * 64 + ((kc ^ bsw) << 4);  // + 1024 j

    // Fragment registers: BH 16, BX 32 (the pair's weights), AH 2 x 8, AX 16 (quarters of the wave's 128 rows) = 80.  One AX set is
    // enough: X jobs never follow each other (two H jobs lie between them and cover the refill), H jobs come in twos.
    xv_f16x8 AH[2][2], BH[4];
    xv_i32x4 AXl[2], AXh[2], BXl[4], BXh[4];             // a 32-byte 8-bit fragment = two 16-byte reads
    auto cat = [](xv_i32x4 u, xv_i32x4 v) { return __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7); };
    // The MFMAs are TIED inline asm (dst = srcC).  The compiler has no tied form of the 16 x 16 MFMAs (only the shapes with more
    // than four passes have one): left to the builtins its register allocator let the 32 accumulator tiles wander -- 82 % of the
    // MFMAs wrote their tile somewhere else, 500 copies per period, 130-190 registers spilled inside the loop.  Nothing in
    // the loop needs the hazard recogniser (which does not look into asm): every MFMA operand comes from ds_read (waited for
    // by s_waitcnt, which the compiler still inserts per register), from an MFMA at least 8 MFMAs back, or from a constant.
    // asm volatile statements also keep their order among themselves and against memory operations, so the interleave of MFMAs,
    // fragment reads and LDS-DMA below IS the source order (no sched_group_barrier needed, none possible).
    auto mfma_h = [&](f32x4 &c, const xv_f16x8 &a, const xv_f16x8 &b) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    };
    auto mfma_x = [&](f32x4 &c, const xv_i32x8 &a, const xv_i32x8 &b) {
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:1 blgp:1"
                     : "+v"(c) : "v"(a), "v"(b), "v"(scale_a), "v"(scale_b));
    };
    // job = 8 MFMAs of a quarter (two row tiles x four column tiles), interleaved one to one with up to 8 fragment reads
    auto job_h = [&](auto SET, auto Q, auto &&...loads) {
        constexpr int set = decltype(SET)::value, q = decltype(Q)::value;
        auto step = [&](auto KK) {
            constexpr int k = decltype(KK)::value;
            mfma_h(acc[2 * q + (k & 1)][k >> 1], AH[set][k & 1], BH[k >> 1]);
            if constexpr (k < (int)sizeof...(loads)) call_kth<k>(loads...);
        };
        for_taps<0, 8>(step);
    };
    auto job_x = [&](auto Q, auto &&...loads) {               // column tile outer: BX of tiles 2, 3 may still be on its way
        constexpr int q = decltype(Q)::value;
        auto step = [&](auto KK) {
            constexpr int k = decltype(KK)::value;
            mfma_x(acc[2 * q + (k & 1)][k >> 1], cat(AXl[k & 1], AXh[k & 1]), cat(BXl[k >> 1], BXh[k >> 1]));
            if constexpr (k < (int)sizeof...(loads)) call_kth<k>(loads...);
        };
        for_taps<0, 8>(step);
    };
    // single 16-byte fragment reads
#define XV_LD(dst, T, ptr, off) dst = *reinterpret_cast<const T *>((ptr) + (off))
    // (every address is a per-lane BASE POINTER plus a compile-time offset that the ds_read carries in its offset field; with the
    // offset added to an integer first the compiler hoisted "base + 1024 j" out of the loop into registers of their own)
    auto ld_ah = [&](auto SET, auto R, const char *ah, int buf, int q) {
        return [&, ah, buf, q] { XV_LD(AH[decltype(SET)::value][decltype(R)::value], xv_f16x8, ah, buf * W_A_BYTES + q * 32 * SROW + decltype(R)::value * 16 * SROW); };
    };
    auto ld_axl = [&](auto R, const char *px, int q) {
        return [&, px, q] { XV_LD(AXl[decltype(R)::value], xv_i32x4, px, q * 32 * SROW + decltype(R)::value * 16 * SROW); };
    };
    auto ld_axh = [&](auto R, const char *px2, int q) {
        return [&, px2, q] { XV_LD(AXh[decltype(R)::value], xv_i32x4, px2, q * 32 * SROW + decltype(R)::value * 16 * SROW); };
    };
    // (opaque: else the compiler splits off the region base -- 0x10800 and up, too large for an offset field -- and keeps
    // "lane part + region + 1024 j" in a register per j)
    int pbh_o = pbh, pbx_o = pbx, pbx2_o = pbx ^ 32;
    asm volatile("" : "+v"(pbh_o), "+v"(pbx_o), "+v"(pbx2_o));
    const char *const pBH = lds + pbh_o, *const pBX = lds + pbx_o, *const pBX2 = lds + pbx2_o;
    auto ld_bh = [&](auto J, int par) { return [&, par] { XV_LD(BH[decltype(J)::value], xv_f16x8, pBH, par * B_BYTES + decltype(J)::value * 1024); }; };
    auto ld_bxl = [&](auto J) { return [&] { XV_LD(BXl[decltype(J)::value], xv_i32x4, pBX, decltype(J)::value * 1024); }; };
    auto ld_bxh = [&](auto J) { return [&] { XV_LD(BXh[decltype(J)::value], xv_i32x4, pBX2, decltype(J)::value * 1024); }; };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;

    // halo pieces: the next ODD slab goes out in DMA slots 0 .. K-3 of the period (slot 2j: behind barrier 1 of pair j, 2j+1:
    // behind barrier 2), the next EVEN slab in slots K+1 .. 2K-2 -- between the last read of the buffer's old slab and the
    // barrier in front of the first read of the new one.  NS one-piece-per-wave slots dealt over K-2 DMA slots, all waves.
    constexpr int NI = 8;
    constexpr int DT = KT - 2;
    constexpr int NS = (NP + NI - 1) / NI;
    constexpr int PW = (NS + DT - 1) / DT;
    auto slots_of = [](int t) constexpr { return t < DT ? (NS + DT - 1 - t) / DT : 0; };
    auto slot_base = [](int t) constexpr { int b = 0; for (int u = 0; u < t; ++u) b += (NS + DT - 1 - u) / DT; return b; };
    const int rowstep = 8 * (int)xrow_bytes;
    int ag_off[DT][PW], al_off[DT][PW];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            int piece = (slot_base(t) + j) * NI + wave;
            piece = piece < NP ? piece : NP - 1;
            ag_off[t][j] = piece * rowstep;
            al_off[t][j] = piece * 1024;
        }

    // Jobs of pair p (H = 8 fp16 MFMAs of item u0 / u1 on a quarter, X = 8 scaled MFMAs of the pair on a quarter), with the reads
    // issued behind them (a set is refilled right behind the job that consumed it, one or two jobs before its next use):
    //   J1  H u0 q0 | BX tiles 0,1; AX <- q0                     J7  H u1 q0 | AX <- q2
    //   J2  H u0 q1 | BX tiles 2,3; AH[0] <- u0 q2               J8  H u1 q1 | AH[0] <- u1 q2
    //   barrier 1: BX(p) read, H1(p) landed -> DMA X(p+1), H0(p+1)   barrier 2: X(p+1), H0(p+1) landed -> DMA H1(p+1), halo pieces
    //   J3  X q0    | AH[1] <- u0 q3                              J9  X q2    | AH[1] <- u1 q3
    //   J4  H u0 q2 | AX <- q1                                    J10 H u1 q2 | AX <- q3
    //   J5  H u0 q3 | AH[0] <- u1 q0                              J11 X q3    | AH[0] <- u0' q0
    //   J6  X q1    | BH <- u1; AH[1] <- u1 q1                    J12 H u1 q3 | BH <- u0' tile by tile; AH[1] <- u0' q1
    // EVERY quarter accumulates a pair as  H u0, X, H u1: an output row's bits must not depend on where in a tile the row lies
    // (a chunk's result is independent of its batch neighbours).  That fixes the order more than it seems: the refill
    // of BH between the two items wants an X job to hide behind, and an X job in that place lies between H u0 and H u1 of its quarter --
    // so all four must, and none is left to cover the refill of BH for the NEXT pair: J12 refills BH tile by tile instead.
    const char *ah0 = lds + addr_h(0, rowbase);        // fp16 address of the current pair's first item
    const char *px, *px2;                              // 8-bit addresses of the current pair (slots 4 + kc / 6 + kc)
    {   // what J11, J12 of a pair "-1" would have read
        const int x0 = addr_x(addr_h(0, rowbase), addr_h((1 % KT) * p.dil, rowbase), 0);
        px = lds + x0;
        px2 = lds + (x0 ^ 32);
        ld_bh(I0{}, 0)(); ld_bh(I1{}, 0)(); ld_bh(I2{}, 0)(); ld_bh(I3{}, 0)();
        ld_ah(I0{}, I0{}, ah0, 0, 0)(); ld_ah(I0{}, I1{}, ah0, 0, 0)();
        ld_ah(I1{}, I0{}, ah0, 0, 1)(); ld_ah(I1{}, I1{}, ah0, 0, 1)();
    }

    int pairs_left = n_pairs - 2;                      // pairs behind the one wsoff points at
    for (int c = 0; c < p.n_chunks; c += 2) {
        const int ce = (c + 2 < p.n_chunks) ? c + 2 : p.n_chunks - 1;
        const int a_odd = (c + 1) * SROW, a_even = ce * SROW;
        char *d_odd = Abuf + W_A_BYTES;
        char *d_even = Abuf + (ce & 1) * W_A_BYTES;
        auto halo = [&](auto D) {                       // DMA slot d of the period
            constexpr int d = decltype(D)::value;
            if constexpr (d <= KT - 3) {
#pragma unroll
                for (int j = 0; j < slots_of(d); ++j) {
                    // (the scalar offset through a local: with the array element as the builtin's argument the HOST pass of hipcc
                    // silently drops the kernel's launch stub -- the library then fails to load with an undefined symbol)
                    const int so = a_odd + ag_off[d][j];
                    XV_BLDS16(xrs, d_odd + al_off[d][j], xvoff, so, 0);
                }
            } else if constexpr (d >= KT + 1 && d <= 2 * KT - 2) {
#pragma unroll
                for (int j = 0; j < slots_of(d - (KT + 1)); ++j) {
                    const int so = a_even + ag_off[d - (KT + 1)][j];
                    XV_BLDS16(xrs, d_even + al_off[d - (KT + 1)][j], xvoff, so, 0);
                }
            }
        };
        auto pair = [&](auto JJ) {
            constexpr int j = decltype(JJ)::value;
            constexpr int u0 = 2 * j, u1 = 2 * j + 1;
            constexpr int un = (2 * j + 2) % (2 * KT), un1 = (2 * j + 3) % (2 * KT);     // the next pair's items (next period: the same offsets)
            constexpr int b0 = u0 / KT, b1 = u1 / KT, bn = un / KT;                      // halo buffers
            int lz = rowbase;
            asm volatile("" : "+v"(lz));
            auto none = [] {};
            // this wave's DMA pieces of the next pair (X planes and H0 behind barrier 1, H1 and the halo pieces behind barrier 2)
            auto dx = [&](auto I) { return [&] { XV_BLDS16(wrs, xdst, wvoff, wsoff + xs_rel, decltype(I)::value * 1024); }; };
            auto dh0 = [&](auto I) { return [&] { XV_BLDS16(wrs, hdst, wvoff, wsoff + hs_rel, decltype(I)::value * 1024); }; };
            auto dh1 = [&](auto I) { return [&] { XV_BLDS16(wrs, hdst + B_BYTES, wvoff, wsoff + hs_rel + B_BYTES, decltype(I)::value * 1024); }; };
            auto halo1 = [&] { halo(std::integral_constant<int, 2 * j>{}); };
            auto halo2 = [&] { halo(std::integral_constant<int, 2 * j + 1>{}); };
            job_h(I0{}, I0{}, ld_bxl(I0{}), ld_bxh(I0{}), ld_bxl(I1{}), ld_bxh(I1{}),
                  ld_axl(I0{}, px, 0), ld_axh(I0{}, px2, 0), ld_axl(I1{}, px, 0), ld_axh(I1{}, px2, 0));                                  // J1
            job_h(I1{}, I1{}, ld_bxl(I2{}), ld_bxh(I2{}), ld_bxl(I3{}), ld_bxh(I3{}), ld_ah(I0{}, I0{}, ah0, b0, 2), ld_ah(I0{}, I1{}, ah0, b0, 2));  // J2
            // barrier 1.  This wave's DMA pieces have landed; its BX reads are complete (LDS operations return in order and the two
            // newest are the AH reads of J2) -- the X regions may be overwritten.  No full drain: the AH reads stay in flight.
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            job_x(I0{}, ld_ah(I1{}, I0{}, ah0, b0, 3), ld_ah(I1{}, I1{}, ah0, b0, 3), dx(I0{}), none, dx(I1{}), none, dx(I2{}), none);    // J3
            const char *ah1 = lds + addr_h((u1 % KT) * p.dil, lz);       // (ah0 is dead from here on)
            job_h(I0{}, I2{}, ld_axl(I0{}, px, 1), ld_axh(I0{}, px2, 1), ld_axl(I1{}, px, 1), ld_axh(I1{}, px2, 1), none, dx(I3{}), none, dh0(I0{}));  // J4
            job_h(I1{}, I3{}, ld_ah(I0{}, I0{}, ah1, b1, 0), ld_ah(I0{}, I1{}, ah1, b1, 0), none, dh0(I1{}), none, halo1);                // J5
            job_x(I1{}, ld_bh(I0{}, 1), ld_bh(I1{}, 1), ld_bh(I2{}, 1), ld_bh(I3{}, 1), ld_ah(I1{}, I0{}, ah1, b1, 1), ld_ah(I1{}, I1{}, ah1, b1, 1));  // J6
            job_h(I0{}, I0{}, ld_axl(I0{}, px, 2), ld_axh(I0{}, px2, 2), ld_axl(I1{}, px, 2), ld_axh(I1{}, px2, 2));                      // J7
            job_h(I1{}, I1{}, ld_ah(I0{}, I0{}, ah1, b1, 2), ld_ah(I0{}, I1{}, ah1, b1, 2));                                               // J8
            // barrier 2.  Nothing read since barrier 1 is rewritten behind it that an MFMA has not consumed already (BH of u1: J7).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            job_x(I2{}, ld_ah(I1{}, I0{}, ah1, b1, 3), ld_ah(I1{}, I1{}, ah1, b1, 3), dh1(I0{}), none, dh1(I1{}), none, halo2);           // J9
            wsoff += pairs_left > 0 ? 2 * B_BYTES : 0;
            --pairs_left;
            job_h(I0{}, I2{}, ld_axl(I0{}, px, 3), ld_axh(I0{}, px2, 3), ld_axl(I1{}, px, 3), ld_axh(I1{}, px2, 3));                      // J10
            // the next pair's addresses (ah1, px, px2 are dead from here on)
            const int ahn_i = addr_h((un % KT) * p.dil, lz);
            const char *ahn = lds + ahn_i;
            const int xn = addr_x(ahn_i, addr_h((un1 % KT) * p.dil, lz), (j + 1) % KT);
            job_x(I3{}, ld_ah(I0{}, I0{}, ahn, bn, 0), ld_ah(I0{}, I1{}, ahn, bn, 0));                                                     // J11
            // J12: the weights of column tile t are dead behind its two MFMAs -- BH is refilled tile by tile (no X job can cover this
            // refill: see the note on the order above), AH[1] behind the last use of each of its halves
            job_h(I1{}, I3{}, none, ld_bh(I0{}, 0), none, ld_bh(I1{}, 0), none, ld_bh(I2{}, 0), ld_ah(I1{}, I0{}, ahn, bn, 1),
                  [&] { ld_bh(I3{}, 0)(); ld_ah(I1{}, I1{}, ahn, bn, 1)(); });                                                              // J12
            ah0 = ahn;
            px = lds + xn;
            px2 = lds + (xn ^ 32);
        };
        for_taps<0, KT>(pair);
    }
#undef XV_LD
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");       // (the last MFMAs' results: the asm is opaque to the hazard recogniser)

    // ---- epilogue ----  (lane constants recomputed from an opaque copy of the lane id: not kept in registers across the loop)
    int lane2 = lane;
    asm volatile("" : "+v"(lane2));
    int rows4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rows4[e] = (int)((0x48c67dbf391502eaull >> (4 * (4 * (lane2 >> 4) + e))) & 15);
    const int col16e = (int)((0xfedc76543210ba98ull >> (4 * (lane2 & 15))) & 15);
    auto write_tile = [&](float *T) {                   // result of tile (i, j): lane = column COL16(lane & 15), rows ROW16(4 kb + e)
        const int col = wc * 64 + col16e;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) T[(16 * i + rows4[e]) * W_TLD + col + 16 * j] = acc[i][j][e];
    };
    wide_epilogue<POOL>(p, lds, Ms, Ps, m0, n0, tid, wr, write_tile);
}

typedef void (*gemm8_fn)(const Gemm8Params);
struct Gemm8Kernel {
    int kt;
    bool pool;
    int wm;
    gemm8_fn fn;
};
#define XV_G8(KT, POOL, WM) {KT, POOL, WM, tdnn_gemm_f16bf8_kernel<KT, POOL, WM>}
const Gemm8Kernel GEMM8_KERNELS[] = {
    XV_G8(1, false, 2), XV_G8(3, false, 2), XV_G8(5, false, 2), XV_G8(7, false, 2),
    XV_G8(1, true, 2),  XV_G8(3, true, 2),  XV_G8(5, true, 2),  XV_G8(7, true, 2),
    XV_G8(1, false, 4), XV_G8(3, false, 4), XV_G8(5, false, 4), XV_G8(7, false, 4),
    XV_G8(1, true, 4),  XV_G8(3, true, 4),  XV_G8(5, true, 4),  XV_G8(7, true, 4),
    // wm == 8: the 256 x 256 tile (tdnn_gemm_f16bf8_wide_kernel)
    {3, false, 8, tdnn_gemm_f16bf8_wide_kernel<3, false>}, {5, false, 8, tdnn_gemm_f16bf8_wide_kernel<5, false>},
    {7, false, 8, tdnn_gemm_f16bf8_wide_kernel<7, false>},
    {3, true, 8, tdnn_gemm_f16bf8_wide_kernel<3, true>},   {5, true, 8, tdnn_gemm_f16bf8_wide_kernel<5, true>},
    {7, true, 8, tdnn_gemm_f16bf8_wide_kernel<7, true>},
    // wm == 16: the 256 x 256 tile on the 16 x 16 MFMA shapes (tdnn_gemm_f16bf8_wide16_kernel)
    {3, false, 16, tdnn_gemm_f16bf8_wide16_kernel<3, false>}, {5, false, 16, tdnn_gemm_f16bf8_wide16_kernel<5, false>},
    {7, false, 16, tdnn_gemm_f16bf8_wide16_kernel<7, false>},
    {3, true, 16, tdnn_gemm_f16bf8_wide16_kernel<3, true>},   {5, true, 16, tdnn_gemm_f16bf8_wide16_kernel<5, true>},
    {7, true, 16, tdnn_gemm_f16bf8_wide16_kernel<7, true>},
};
#undef XV_G8
size_t g8_kernel_lds(const Gemm8Kernel &e) { return e.wm >= 8 ? W_LDS_BYTES : g8_lds_bytes(e.kt, e.wm); }

std::atomic<int> g_tile_rows8{0};
std::atomic<int> g_wide16{1};          // XV_F16BF8_S16=0: the built-in choice keeps the 32 x 32 form of the 256 x 256 tile
std::atomic<int> g_xcd_columns{0};     // XV_TUNE_XCD_COLUMNS

int launch_gemm8(const Gemm8Params &p0, hipStream_t st)
{
    Gemm8Params p = p0;
    if (p.R <= 0 || p.cout <= 0) return 0;
    static const bool env_once = [] {
        const char *e = std::getenv("XV_F16BF8_S16");
        if (e && e[0] == '0') g_wide16.store(0, std::memory_order_relaxed);
        const char *x = std::getenv("XV_XCD_COLUMNS");                 // (counter runs: the knob of XV_TUNE_XCD_COLUMNS from outside)
        if (x && (x[0] == '0' || x[0] == '1')) g_xcd_columns.store(x[0] - '0', std::memory_order_relaxed);
        return true;
    }();
    (void)env_once;
    if (p.cin <= 0 || p.dil <= 0) return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: dims > 0");
    const int span = (p.K - 1) * p.dil;
    if ((p.K != 1 && p.K != 3 && p.K != 5 && p.K != 7) || (p.K > 1 && (span < 2 || span > MAX_SPAN)))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_f16bf8: supports K in {1,3,5,7} with (K-1)*dilation <= 8");
    if ((p.act == XV_ACT_LRELU || p.act == XV_ACT_PRELU) && !p.alpha) return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: act_alpha is NULL");
    if ((((uintptr_t)p.x) | ((uintptr_t)p.wt)) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: input and packed weights must be 16-byte aligned");
    p.n_chunks = (p.cin + BK - 1) / BK;
    p.xchunks = p.n_chunks;
    if (p.y) {
        if (((uintptr_t)p.y) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: output must be 16-byte aligned");
        if (p.y_format == XV_FMT_F32) {
            if (p.ldy < p.cout) return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: ldy < cout");
        } else {
            p.ychunks = (p.cout + 31) / 32;
        }
    }
    p.n_nt = (p.cout + BN - 1) / BN;
    // tile: 128 x 128 (4 waves, two workgroups per CU), 256 x 128 (8 waves) or 256 x 256 (8 waves of 128 x 64; K > 1,
    // Cout % 256 == 0, split-format or POOL output; on the 16 x 16 MFMA shapes where the number of slabs is even, else on the
    // 32 x 32 ones).  XV_TUNE_TILE_ROWS: 128 / 256 force the first two, 512 the 32 x 32 form of the third, 1024 the 16 x 16 form.
    int wm = 2;
    {
        const int want = g_tile_rows8.load(std::memory_order_relaxed);
        const bool wide_ok = p.K > 1 && (p.cout & 255) == 0 && (p.blk != nullptr || p.y_format != XV_FMT_F32);
        const bool big_enough = ((p.R + 255) / 256) * p.n_nt >= 512;
        // the 16 x 16 MFMA form of the 256 x 256 tile pairs (slab, tap) items over two slabs: an even number of slabs
        const bool w16_ok = wide_ok && (p.n_chunks & 1) == 0;
        const bool wide_pays = ((p.R + 255) / 256) * (p.cout / 256) >= 512;
        // The 16 x 16 form adds an element's products in another order than the 32 x 32 tiles (which agree among themselves bit for
        // bit), so a shape that can take it ALWAYS takes it, whatever the number of rows: an utterance's x-vector must not depend on
        // how many utterances share its batch -- on the sharding of a job over ranks, say (tests/test_gpu_eight_ranks.py: 8 ranks
        // write the single process's bytes).  A batch too small to fill the chip twice with 256 x 256 tiles loses a few per cent.
        if (w16_ok && (want == 1024 || (want == 0 && g_wide16.load(std::memory_order_relaxed)))) wm = 16;
        else if (wide_ok && (want == 512 || want == 1024 || (want == 0 && wide_pays))) wm = 8;
        else if (want == 256 || (want == 0 && p.K >= 5 && big_enough)) wm = 4;
    }
    if (wm >= 8) {
        p.n_nt = p.cout / W_BN;
        p.n_mt = (int)((p.R + W_BM - 1) / W_BM);
    } else {
        p.n_mt = (int)((p.R + wm * 64 - 1) / (wm * 64));
    }
    const Gemm8Kernel *k = nullptr;
    for (const Gemm8Kernel &e : GEMM8_KERNELS)
        if (e.kt == p.K && e.pool == (p.blk != nullptr) && e.wm == wm) k = &e;
    if (!k) return fail(XV_ERR_UNSUPPORTED, "tdnn_f16bf8: no kernel for this configuration");
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (const Gemm8Kernel &e : GEMM8_KERNELS) {
            hipError_t err = hipFuncSetAttribute((const void *)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g8_kernel_lds(e));
            if (err != hipSuccess) return hip_fail(err, "hipFuncSetAttribute");
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    p.colmap = (wm == 16 && p.n_nt == 2 && p.n_mt >= 8 && g_xcd_columns.load(std::memory_order_relaxed)) ? 1 : 0;
    hipLaunchKernelGGL(k->fn, dim3((unsigned)(p.n_mt * p.n_nt)), dim3(wm >= 8 ? 512 : wm * 128), g8_kernel_lds(*k), st, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "tdnn_gemm_f16bf8_kernel launch");
}

// w[K, cin, cout] fp32 -> tiled f16bf8 weights: tile (nt, chunk, tap) = 16 KB [fp16 plane 128 x 64 B][8-bit plane 128 x 64 B]
__global__ void pack_weights_f16bf8_kernel(const float *__restrict__ w, int K, int cin, int cout, int n_chunks,
                                           uint8_t *__restrict__ wt, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (tile, col, 8-channel group)
    if (i >= total) return;
    const int g = (int)(i & 3);
    const int n = (int)((i >> 2) & 127);
    const size_t tile = i >> 9;
    const int tap = (int)(tile % K);
    const int chunk = (int)((tile / K) % n_chunks);
    const int nt = (int)(tile / ((size_t)K * n_chunks));
    const int gn = nt * 128 + n;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = chunk * 32 + g * 8 + e;
        v[e] = (c < cin && gn < cout) ? w[((size_t)tap * cin + c) * cout + gn] : 0.f;
    }
    xv_f16x8 hi;
    xv_i32x4 x8;
    float amax = 0.f;
    xv_split8_encode8<false>(v, hi, x8, amax);
    uint8_t *t = wt + tile * B_BYTES + n * 64 + ((g ^ ((n >> 2) & 3)) << 4);
    *reinterpret_cast<xv_f16x8 *>(t) = hi;
    *reinterpret_cast<xv_i32x4 *>(t + B_PLANE) = x8;
}

// fp32 rows -> split8 (tests / tooling; the layers write the format themselves)
__global__ void split8_encode_kernel(const float *__restrict__ x, long R, int c, int ldx, uint8_t *__restrict__ xs, int chunks,
                                     int *status)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (row, slab, 8-channel group)
    if (i >= (size_t)R * chunks * 4) return;
    const int g = (int)(i & 3);
    const int ch = (int)((i >> 2) % chunks);
    const long r = (long)(i / ((size_t)4 * chunks));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cc = ch * 32 + g * 8 + e;
        v[e] = cc < c ? x[(size_t)r * ldx + cc] : 0.f;
    }
    xv_f16x8 hi;
    xv_i32x4 x8;
    float amax = 0.f;
    xv_split8_encode8<true>(v, hi, x8, amax);
    const int sw = (int)(r >> 1) & 7;
    uint8_t *row = xs + ((size_t)r * chunks + ch) * SROW;
    *reinterpret_cast<xv_f16x8 *>(row + ((g ^ sw) << 4)) = hi;
    *reinterpret_cast<xv_i32x4 *>(row + (((4 + g) ^ sw) << 4)) = x8;
    if (amax > XV_SPLIT8_MAX && status) atomicOr(status, 1);
}

__global__ void split8_decode_kernel(const uint8_t *__restrict__ xs, long R, int c, int chunks, float *__restrict__ x, int ldx)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * c) return;
    const long r = (long)(i / c);
    const int cc = (int)(i - (size_t)r * c);
    const int ch = cc >> 5, k = cc & 31;
    const int sw = (int)(r >> 1) & 7;
    const uint8_t *row = xs + ((size_t)r * chunks + ch) * SROW;
    const _Float16 h = *reinterpret_cast<const _Float16 *>(row + (((k >> 3) ^ sw) << 4) + (k & 7) * 2);
    const uint8_t l = row[(((4 + (k >> 3)) ^ sw) << 4) + (k & 7)];
    x[(size_t)r * ldx + cc] = (float)h + xv_bf8_to_float(l) * (1.f / XV_SPLIT8_LO_SCALE);
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, what);
}

}  // namespace

extern "C" {

void xv_internal_gemm8_tile_rows(int value) { g_tile_rows8.store(value, std::memory_order_relaxed); }
void xv_internal_gemm8_xcd_columns(int value) { g_xcd_columns.store(value, std::memory_order_relaxed); }

size_t xv_packed_weights_f16bf8_bytes(int K, int cin, int cout)
{
    if (K <= 0 || cin <= 0 || cout <= 0) return 0;
    return (size_t)((cout + BN - 1) / BN) * ((cin + BK - 1) / BK) * K * B_BYTES;
}

int xv_pack_weights_f16bf8(const float *w, int K, int cin, int cout, void *wt, void *stream)
{
    if (!w || !wt || K <= 0 || cin <= 0 || cout <= 0) return fail(XV_ERR_BAD_ARG, "pack_weights_f16bf8: bad argument");
    if (((uintptr_t)wt) & 15) return fail(XV_ERR_BAD_ARG, "pack_weights_f16bf8: wt must be 16-byte aligned");
    const int n_chunks = (cin + BK - 1) / BK;
    const size_t total = xv_packed_weights_f16bf8_bytes(K, cin, cout) / 32;     // one thread per 8-channel group (16 + 16 bytes)
    hipLaunchKernelGGL(pack_weights_f16bf8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, K,
                       cin, cout, n_chunks, (uint8_t *)wt, total);
    return check_launch("pack_weights_f16bf8_kernel");
}

int xv_split8_encode_f32(const float *x, int64_t R, int c, int ldx, void *xs, int32_t *status, void *stream)
{
    if (R <= 0) return 0;
    if (!x || !xs || c <= 0 || ldx < c) return fail(XV_ERR_BAD_ARG, "split8_encode: bad argument");
    if (((uintptr_t)xs) & 15) return fail(XV_ERR_BAD_ARG, "split8_encode: xs must be 16-byte aligned");
    const int chunks = (c + 31) / 32;
    const size_t n = (size_t)R * chunks * 4;
    hipLaunchKernelGGL(split8_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (long)R, c, ldx,
                       (uint8_t *)xs, chunks, (int *)status);
    return check_launch("split8_encode_kernel");
}

int xv_split8_decode_f32(const void *xs, int64_t R, int c, float *x, int ldx, void *stream)
{
    if (R <= 0) return 0;
    if (!x || !xs || c <= 0 || ldx < c) return fail(XV_ERR_BAD_ARG, "split8_decode: bad argument");
    const size_t n = (size_t)R * c;
    hipLaunchKernelGGL(split8_decode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)xs, (long)R, c, (c + 31) / 32, x, ldx);
    return check_launch("split8_decode_kernel");
}

int xv_tdnn_layer_f16bf8(const void *x, int64_t R, int cin, const void *wt, const float *bias, const float *bn_scale,
                         const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                         const uint8_t *row_valid, void *y, int y_format, int ldy, int32_t *status, void *stream)
{
    if (!x || !wt || !y) return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: unknown act_kind");
    if (y_format != XV_FMT_F32 && y_format != XV_FMT_SPLIT && y_format != XV_FMT_SPLIT8)
        return fail(XV_ERR_BAD_ARG, "tdnn_f16bf8: unknown tensor format");
    Gemm8Params p{};
    p.x = (const uint8_t *)x; p.R = (long)R; p.cin = cin; p.wt = (const uint8_t *)wt;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid;
    p.y = y; p.y_format = y_format; p.ldy = ldy; p.status = (int *)status;
    return launch_gemm8(p, (hipStream_t)stream);
}

int xv_tdnn_layer_pool_f16bf8(const void *x, int64_t R, int cin, const void *wt, const float *bias, const float *bn_scale,
                              const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                              const uint8_t *row_valid, float *block_stats, void *stream)
{
    if (!x || !wt || !block_stats) return fail(XV_ERR_BAD_ARG, "tdnn_pool_f16bf8: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_pool_f16bf8: unknown act_kind");
    if (((uintptr_t)block_stats) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_pool_f16bf8: block_stats must be 16-byte aligned");
    Gemm8Params p{};
    p.x = (const uint8_t *)x; p.R = (long)R; p.cin = cin; p.wt = (const uint8_t *)wt;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha;
    p.K = K; p.dil = dilation; p.cout = cout; p.valid = row_valid;
    p.blk = block_stats; p.y_format = XV_FMT_F32;
    return launch_gemm8(p, (hipStream_t)stream);
}

}  // extern "C"
