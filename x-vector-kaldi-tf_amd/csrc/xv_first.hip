// xv_first.hip -- frame_level_info_layer-0 as a kernel of its own (local/tf/models.py:54-67 with the 23 MFCC dimensions as input).
//
// The first TDNN layer is 1.4 % of the FLOPs but wrote 4 % of the time in the general LDS-staged GEMM: its whole reduction is
// K*Cin = 5*24 = 120 deep -- four MFMA k-steps -- so a 128x128 tile is prologue, five short stages and epilogue, repeated for
// each of the four column tiles with the fp32 -> hi/lo conversion of the same input rows done four times.  What bounds the
// layer is its OUTPUT: 2 KB per frame in the split activation format, 0.54 GB per 262144-row batch.  This kernel is laid out
// around that:
//   * a wave owns 16 frames; its im2col operand (16 frames x 128 k, k = tap*24 + channel) is built ONCE in registers straight
//     from the fp32 feature rows (8 x 32-byte loads per lane, L2 hits; rows outside [0, R) read as zero) and split to hi/lo;
//   * the weights of 128 output channels (8 tiles x 4 k-steps x hi/lo fragments = 64 KB, packed in MFMA-fragment order) are
//     brought into LDS by DMA once per pass and re-used for a strip of 4 x 128 frames; four passes cover 512 channels, and two
//     workgroups share a CU, so one computes while the other waits for its weights;
//   * the product is formed transposed (A = weight fragment, B = frames) on v_mfma_f32_16x16x32_bf16, so that a lane holds 4
//     rows of one frame per tile; the packed weights order the rows of a tile PAIR so that a lane's 4 + 4 rows are 8 consecutive
//     channels -- exactly one 16-byte hi slot and one 16-byte lo slot of the split format, no cross-lane exchange -- and the
//     epilogue (bias, activation, BN, gap-row mask, hi/lo split) stores them with non-temporal 16-byte stores, no LDS round trip.
// Arithmetic: bf16x3 as everywhere (hi*hi + hi*lo + lo*hi, fp32 accumulate).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "xvector_hip.h"
#include "xv_split8.h"

extern "C" void xv_internal_set_error(const char *msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

int fail(int code, const char *msg)
{
    xv_internal_set_error(msg);
    return code;
}

constexpr int FR_WAVES = 8;
constexpr int FR_ROWS = 16 * FR_WAVES;      // frames per row tile
constexpr int FR_STRIP = 4;                 // row tiles per workgroup (weights are loaded once per pass and strip)
constexpr int FR_NKS = 4;                   // k-steps of 32: K * ceil8(Cin) <= 128
constexpr int FR_PASS_COLS = 128;           // output channels whose weights are resident at a time (64 KB: two workgroups per CU)
constexpr int FR_W_BYTES = (FR_PASS_COLS / 16) * FR_NKS * 2048;      // 64 KB
constexpr int FR_P_OFF = FR_W_BYTES;        // [bias | scale | shift | alpha][cout <= 512]
constexpr int FR_MAX_COUT = 512;
constexpr size_t FR_LDS_BYTES = FR_P_OFF + 4 * FR_MAX_COUT * 4;
constexpr int SROW = 128;

struct FirstParams {
    const float *x;
    long R;
    int ldx, kc;               // kc = ceil8(cin): im2col index k = tap*kc + channel
    int K, dil, cout;
    const uint8_t *wt;         // fragments: [pass][tile 0..15][k-step][hi 1 KB | lo 1 KB]
    const float *bias, *scale, *shift, *alpha;
    int act;
    const uint8_t *valid;
    uint8_t *y;                // split-format output, row 0
    int ychunks;
    int *status;               // Y8: bit 0 set when a value had to be clamped (may be NULL)
};

#define XV_GLDS16_OFF(gptr, lptr, imm)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                    \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, imm, 0)

template <int MODE>
__device__ __forceinline__ float act_fn(float z, float a)
{
    return MODE == 1 ? fmaxf(a * z, z) : MODE == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + a * fminf(z, 0.f);
}

// Y8: write XV_FMT_SPLIT8 rows (fp16 hi + bf8 cross bytes, xv_split8.h) instead of the bf16 hi/lo planes
template <int MODE, bool Y8>
__global__ __launch_bounds__(FR_WAVES * 64, 4) void tdnn_first_kernel(const FirstParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = lane >> 4, f = lane & 15;
    const long strip0 = (long)blockIdx.x * (FR_STRIP * FR_ROWS);
    const int left = ((p.K - 1) * p.dil) >> 1;

    // epilogue parameters -> LDS (before any DMA is in flight)
    {
        float *P = reinterpret_cast<float *>(lds + FR_P_OFF);
        for (int c = tid; c < p.cout; c += FR_WAVES * 64) {
            P[c] = p.bias ? p.bias[c] : 0.f;
            P[FR_MAX_COUT + c] = p.scale ? p.scale[c] : 1.f;
            P[2 * FR_MAX_COUT + c] = p.shift ? p.shift[c] : 0.f;
            P[3 * FR_MAX_COUT + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.alpha[0] : p.act == XV_ACT_PRELU ? p.alpha[c] : 0.f;
        }
    }
    const f32x4 *P4 = reinterpret_cast<const f32x4 *>(lds + FR_P_OFF);
    // this lane's im2col slots: k-step u, group G -> k = 32u + 8G .. +7 = (tap, channels c0 .. c0+7)
    int tap_off[FR_NKS], c0[FR_NKS];
    bool live[FR_NKS];
#pragma unroll
    for (int u = 0; u < FR_NKS; ++u) {
        const int k = 32 * u + 8 * G;
        const int tap = k / p.kc;
        live[u] = tap < p.K;
        tap_off[u] = (tap - 0) * p.dil - left;
        c0[u] = k - tap * p.kc;
    }
    const size_t yrow = (size_t)p.ychunks * SROW;
    float amax = 0.f;
    const int n_pass = (p.cout + FR_PASS_COLS - 1) / FR_PASS_COLS;

    for (int pass = 0; pass < n_pass; ++pass) {
        const int pass_cols = min(FR_PASS_COLS, p.cout - pass * FR_PASS_COLS);
        const int n_pairs = pass_cols >> 5;                       // tile pairs (32 channels) of this pass
        // ---- weights of this pass: n_pairs * 2 tiles * 4 k-steps * 2 KB, lane-linear fragments, 1 KB DMA pieces ----------
        __syncthreads();                                          // everybody is done with the previous pass's weights
        {
            const int pieces = n_pairs * 2 * FR_NKS * 2;
            const uint8_t *src = p.wt + (size_t)pass * FR_W_BYTES + lane * 16;
            for (int pc = wave; pc < pieces; pc += FR_WAVES) XV_GLDS16_OFF(src + (size_t)pc * 1024, lds + pc * 1024, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        for (int rt = 0; rt < FR_STRIP; ++rt) {
            const long row0 = strip0 + (long)rt * FR_ROWS + 16 * wave;       // this wave's 16 frames
            if (row0 >= p.R) break;                                            // (wave-uniform)
            const long row = row0 + f;
            // ---- B operand: the wave's frames as im2col fragments, fp32 -> hi/lo ------------------------------------
            bf16x8 xh[FR_NKS], xl[FR_NKS];
#pragma unroll
            for (int u = 0; u < FR_NKS; ++u) {
                const long r = row + tap_off[u];
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
                if (live[u] && r >= 0 && r < p.R) {
                    const float *src = p.x + (size_t)r * p.ldx + c0[u];
                    a = *reinterpret_cast<const f32x4 *>(src);
                    b = *reinterpret_cast<const f32x4 *>(src + 4);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const __bf16 ha = (__bf16)a[e], hb = (__bf16)b[e];
                    xh[u][e] = ha; xh[u][4 + e] = hb;
                    xl[u][e] = (__bf16)(a[e] - (float)ha); xl[u][4 + e] = (__bf16)(b[e] - (float)hb);
                }
            }
            const float keep = (row < p.R && (!p.valid || p.valid[row])) ? 1.f : 0.f;
            const int sw = (int)(row >> 1) & 7;
            uint8_t *yr = p.y + (size_t)row * yrow;

            for (int tp = 0; tp < n_pairs; ++tp) {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
                const char *w0 = lds + (size_t)(tp * 2) * (FR_NKS * 2048) + lane * 16;
                const char *w1 = w0 + FR_NKS * 2048;
#pragma unroll
                for (int u = 0; u < FR_NKS; ++u) {
                    const bf16x8 h0 = *reinterpret_cast<const bf16x8 *>(w0 + u * 2048);
                    const bf16x8 l0 = *reinterpret_cast<const bf16x8 *>(w0 + u * 2048 + 1024);
                    const bf16x8 h1 = *reinterpret_cast<const bf16x8 *>(w1 + u * 2048);
                    const bf16x8 l1 = *reinterpret_cast<const bf16x8 *>(w1 + u * 2048 + 1024);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l0, xh[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(l1, xh[u], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xl[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xl[u], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h0, xh[u], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, xh[u], acc1, 0, 0, 0);
                }
                // lane (frame f, group G) holds rows 4G..4G+3 of both tiles; the packed weights order the rows of a tile pair so
                // that these are channels 8G..8G+3 (tile 0) and 8G+4..8G+7 (tile 1) of the pair's 32: one 16-byte slot of the
                // output format per lane, with no exchange between lanes
                const f32x4 lo4 = acc0;                           // channels cb .. cb+3
                const f32x4 hi4 = acc1;                           // channels cb+4 .. cb+7
                const int cb = pass * FR_PASS_COLS + tp * 32 + 8 * G;
                const int c4 = cb >> 2;
                const f32x4 b0 = P4[c4], b1 = P4[c4 + 1], s0 = P4[FR_MAX_COUT / 4 + c4], s1 = P4[FR_MAX_COUT / 4 + c4 + 1],
                            o0 = P4[2 * FR_MAX_COUT / 4 + c4], o1 = P4[2 * FR_MAX_COUT / 4 + c4 + 1],
                            a0 = P4[3 * FR_MAX_COUT / 4 + c4], a1 = P4[3 * FR_MAX_COUT / 4 + c4 + 1];
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = keep != 0.f ? act_fn<MODE>(lo4[e] + b0[e], a0[e]) * s0[e] + o0[e] : 0.f;
                    v[4 + e] = keep != 0.f ? act_fn<MODE>(hi4[e] + b1[e], a1[e]) * s1[e] + o1[e] : 0.f;
                }
                const int slab = cb >> 5, t = (cb & 31) >> 3;
                uint8_t *slabp = yr + (size_t)slab * SROW;
                if constexpr (Y8) {
                    xv_f16x8 vh;
                    xv_i32x4 vx;
                    xv_split8_encode8<true>(v, vh, vx, amax);
                    if (row < p.R) {
                        __builtin_nontemporal_store(vh, reinterpret_cast<xv_f16x8 *>(slabp + ((t ^ sw) << 4)));
                        __builtin_nontemporal_store(vx, reinterpret_cast<xv_i32x4 *>(slabp + (((4 + t) ^ sw) << 4)));
                    }
                } else {
                    bf16x8 vh, vl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const __bf16 h = (__bf16)v[e];
                        vh[e] = h;
                        vl[e] = (__bf16)(v[e] - (float)h);
                    }
                    if (row < p.R) {
                        __builtin_nontemporal_store(vh, reinterpret_cast<bf16x8 *>(slabp + ((t ^ sw) << 4)));
                        __builtin_nontemporal_store(vl, reinterpret_cast<bf16x8 *>(slabp + (((4 + t) ^ sw) << 4)));
                    }
                }
            }
        }
    }
    if constexpr (Y8)
        if (amax > XV_SPLIT8_MAX && p.status) atomicOr(p.status, 1);
}

// w[K][cin][cout] (TF order) -> per pass of FR_PASS_COLS output channels: [tile][k-step 0..3][hi 1 KB | lo 1 KB], fragment
// element (lane (i, g), e) = w[tap][c][col] with k = 32u + 8g + e = tap*kc + c (zero beyond K*kc / cin); row i of tile 2p + t is
// column col = pass*FR_PASS_COLS + 32p + 8(i>>2) + 4t + (i&3): the accumulator rows 4G..4G+3 of the pair's two tiles are the
// consecutive channels 8G..8G+7 (see the kernel's epilogue)
__global__ void pack_first_kernel(const float *__restrict__ w, int K, int cin, int kc, int cout, uint8_t *__restrict__ wt, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (pass, tile, k-step, lane, e)
    if (i >= total) return;
    const int e = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    constexpr int TILES = FR_PASS_COLS / 16;                            // 8: a power of two
    const int u = (int)((i >> 9) & 3);
    const int tile = (int)((i >> 11) & (TILES - 1));
    const int pass = (int)(i / (size_t)(512 * FR_NKS * TILES));
    const int k = 32 * u + 8 * (lane >> 4) + e;
    const int tap = k / kc, c = k - tap * kc;
    const int i_row = lane & 15;
    const int col = pass * FR_PASS_COLS + 32 * (tile >> 1) + 8 * (i_row >> 2) + 4 * (tile & 1) + (i_row & 3);
    const float x = (tap < K && c < cin && col < cout) ? w[((size_t)tap * cin + c) * cout + col] : 0.f;
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    uint8_t *t = wt + (size_t)pass * FR_W_BYTES + (size_t)(tile * FR_NKS + u) * 2048 + lane * 16 + e * 2;
    *reinterpret_cast<uint16_t *>(t) = __builtin_bit_cast(uint16_t, hi);
    *reinterpret_cast<uint16_t *>(t + 1024) = __builtin_bit_cast(uint16_t, lo);
}

bool first_shape_ok(int K, int cin, int cout)
{
    const int kc = (cin + 7) / 8 * 8;
    return K > 0 && (K & 1) && cin > 0 && K * kc <= 32 * FR_NKS && cout > 0 && (cout & 31) == 0 && cout <= FR_MAX_COUT;
}

}  // namespace

extern "C" {

size_t xv_packed_first_bf16x3_bytes(int K, int cin, int cout)
{
    if (!first_shape_ok(K, cin, cout)) return 0;
    return (size_t)((cout + FR_PASS_COLS - 1) / FR_PASS_COLS) * FR_W_BYTES;
}

int xv_pack_first_bf16x3(const float *w, int K, int cin, int cout, void *wt, void *stream)
{
    if (!w || !wt) return fail(XV_ERR_BAD_ARG, "pack_first_bf16x3: NULL pointer");
    if (!first_shape_ok(K, cin, cout))
        return fail(XV_ERR_UNSUPPORTED, "pack_first_bf16x3: needs K odd, K*ceil8(cin) <= 128, cout % 32 == 0, cout <= 512");
    const size_t total = xv_packed_first_bf16x3_bytes(K, cin, cout) / 4;
    hipLaunchKernelGGL(pack_first_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, K, cin,
                       (cin + 7) / 8 * 8, cout, (uint8_t *)wt, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

static int first_launch(const float *x, int64_t R, int cin, int ldx, const void *wt, const float *bias, const float *bn_scale,
                        const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                        const uint8_t *row_valid, void *y, bool y8, int32_t *status, void *stream)
{
    if (R <= 0) return 0;
    if (!x || !wt || !y) return fail(XV_ERR_BAD_ARG, "tdnn_first: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_first: unknown act_kind");
    if ((act_kind == XV_ACT_LRELU || act_kind == XV_ACT_PRELU) && !act_alpha) return fail(XV_ERR_BAD_ARG, "tdnn_first: act_alpha is NULL");
    const int kc = (cin + 7) / 8 * 8;
    if (!first_shape_ok(K, cin, cout) || dilation <= 0 || (K - 1) * dilation > 8)
        return fail(XV_ERR_UNSUPPORTED, "tdnn_first: needs K odd, K*ceil8(cin) <= 128, (K-1)*dilation <= 8, cout % 32 == 0, cout <= 512");
    if (ldx < kc || (ldx & 7) || (((uintptr_t)x) & 31))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_first: rows must hold ceil8(cin) floats (padding columns zero), ldx % 8 == 0, x 32-byte aligned");
    if ((((uintptr_t)wt) | ((uintptr_t)y)) & 15) return fail(XV_ERR_BAD_ARG, "tdnn_first: wt and y must be 16-byte aligned");
    FirstParams p{};
    p.x = x; p.R = (long)R; p.ldx = ldx; p.kc = kc; p.K = K; p.dil = dilation; p.cout = cout; p.wt = (const uint8_t *)wt;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.alpha = act_alpha; p.act = act_kind; p.valid = row_valid;
    p.y = (uint8_t *)y; p.ychunks = cout / 32; p.status = (int *)status;
    typedef void (*kern_t)(const FirstParams);
    const kern_t kerns[6] = {tdnn_first_kernel<0, false>, tdnn_first_kernel<1, false>, tdnn_first_kernel<2, false>,
                             tdnn_first_kernel<0, true>,  tdnn_first_kernel<1, true>,  tdnn_first_kernel<2, true>};
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (kern_t k : kerns) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FR_LDS_BYTES);
            if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int mode = act_kind == XV_ACT_LRELU ? 1 : act_kind == XV_ACT_RELU ? 2 : 0;
    const long strip = (long)FR_STRIP * FR_ROWS;
    hipLaunchKernelGGL(kerns[mode + (y8 ? 3 : 0)], dim3((unsigned)((R + strip - 1) / strip)), dim3(FR_WAVES * 64), FR_LDS_BYTES,
                       (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

int xv_tdnn_first_bf16x3(const float *x, int64_t R, int cin, int ldx, const void *wt, const float *bias, const float *bn_scale,
                         const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                         const uint8_t *row_valid, void *y, void *stream)
{
    return first_launch(x, R, cin, ldx, wt, bias, bn_scale, bn_shift, act_kind, act_alpha, K, dilation, cout, row_valid, y, false,
                        nullptr, stream);
}

int xv_tdnn_first_f16bf8(const float *x, int64_t R, int cin, int ldx, const void *wt, const float *bias, const float *bn_scale,
                         const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                         const uint8_t *row_valid, void *y, int32_t *status, void *stream)
{
    return first_launch(x, R, cin, ldx, wt, bias, bn_scale, bn_shift, act_kind, act_alpha, K, dilation, cout, row_valid, y, true,
                        status, stream);
}

}  // extern "C"
