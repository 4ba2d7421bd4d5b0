// xv_first.hip -- frame_level_info_layer-0 as a kernel of its own (local/tf/models.py:54-67 with the 23 MFCC dimensions as input).
//
// The first TDNN layer is 1.4 % of the FLOPs but wrote 4 % of the time in the general LDS-staged GEMM: its whole reduction is
// K*Cin = 5*24 = 120 deep -- four MFMA k-steps -- so a 128x128 tile is prologue, five short stages and epilogue, repeated for
// each of the four column tiles with the fp32 -> hi/lo conversion of the same input rows done four times.  What bounds the
// layer is its OUTPUT: 2 KB per frame in the split activation format, 0.54 GB per 262144-row batch.  This kernel is laid out
// around that:
//   * one workgroup per CU, eight waves; every wave owns a contiguous run of 16-frame tiles and works on three at a time: their
//     im2col operands (16 frames x 128 k, k = tap*24 + channel) are built ONCE in registers straight from the fp32 feature rows
//     (8 x 32-byte loads per lane; rows outside [0, R) read as zero), split to hi/lo, and stay there for all 512 channels;
//   * the weights of 128 output channels (8 tiles x 4 k-steps x hi/lo fragments = 64 KB, packed in MFMA-fragment order) alternate
//     between two LDS buffers, fetched by DMA one step (= one 128-channel pass of one group of tiles) ahead; the wait for them is a
//     COUNTED vmcnt -- a wave's stores of the step in between need not have retired;
//   * the weight fragments and epilogue parameters of a tile pair (32 channels) are read from LDS once for the wave's three
//     tiles (a third of the fragment reads of the earlier one-tile-at-a-time form);
//   * the product is formed transposed (A = weight fragment, B = frames) on v_mfma_f32_16x16x32_bf16, so that a lane holds 4
//     rows of one frame per tile; the packed weights order the rows of a tile PAIR so that a lane's 4 + 4 rows are 8 consecutive
//     channels -- exactly one 16-byte hi slot and one 16-byte lo slot of the split format, no cross-lane exchange -- and the
//     epilogue (bias as the accumulators' start value, activation, BN, gap-row mask, hi/lo split) stores them with non-temporal
//     16-byte buffer stores (rows past the end: dropped by the descriptor's range check), no LDS round trip;
//   * every output row is 2 KB, so workgroups that all write the same 128-byte slab of their rows at the same moment send the
//     whole chip's stores to the same few memory channels: the workgroups start at different passes and the waves of a workgroup
//     at different tile pairs (measured: 0.184 -> 0.156 ms).
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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

std::atomic<int> g_first_tiles{0};        // xv_internal_first_tiles(): tiles per wave, 0 = spread over the CUs

int fail(int code, const char *msg)
{
    xv_internal_set_error(msg);
    return code;
}

constexpr int FR_WAVES = 8;                // one workgroup per CU, two waves per SIMD: 256 VGPRs each
constexpr int FR_TILES = 3;                // 16-frame tiles a wave works on at a time: a tile pair's weights and parameters are read once for the three
constexpr int FR_NKS = 4;                   // k-steps of 32: K * ceil8(Cin) <= 128
constexpr int FR_PASS_COLS = 128;           // output channels of one pass: their weights are 64 KB, two such buffers alternate
constexpr int FR_W_BYTES = (FR_PASS_COLS / 16) * FR_NKS * 2048;      // 64 KB
constexpr int FR_P_OFF = 2 * FR_W_BYTES;    // [bias | scale | shift | alpha][cout <= 512]
constexpr int FR_MAX_COUT = 512;
constexpr size_t FR_LDS_BYTES = FR_P_OFF + 4 * FR_MAX_COUT * 4;
constexpr int SROW = 128;
static_assert(FR_TILES == 3, "tdnn_first_kernel dispatches first_block<.., 1 | 2 | 3>");

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
    long row_base, rows_here;  // this launch writes rows [row_base, row_base + rows_here) (a buffer descriptor addresses < 4 GB of y)
    long n_tiles;              // 16-frame tiles of this launch: ceil(rows_here / 16)
    int tpw;                   // tiles per wave: wave w of workgroup b owns tiles [(8b + w) tpw, +tpw)
    int *status;               // Y8: bit 0 set when a value had to be clamped (may be NULL)
};

#define XV_GLDS16_OFF(gptr, lptr, imm)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                    \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, imm, 0)

template <int MODE>
__device__ __forceinline__ float act_fn(float z, float a)
{
    // relu as ONE instruction: fmaxf() first canonicalises an MFMA result (v_max z, z); the signed-integer maximum of the bit pattern
    // with 0 is the same function (negative floats are negative integers; -0 and negative NaNs become +0).  (Not inline assembly:
    // the compiler's hazard recogniser does not see an asm statement that reads an MFMA result.)
    if constexpr (MODE == 2) return __builtin_bit_cast(float, max(__builtin_bit_cast(int, z), 0));
    return MODE == 1 ? fmaxf(a * z, z) : MODE == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + a * fminf(z, 0.f);
}

// c - (float)h for the fp16 in the low / high half of a register: ONE v_fma_mix_f32 (the compiler's own form is a conversion plus a
// subtraction); exact, like the subtraction
template <int HALF>
__device__ __forceinline__ float sub_f16_half(float c, int hpair)
{
    float r;
    if constexpr (HALF == 0)
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(c));
    else
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpair), "v"(c));
    return r;
}

constexpr int FR_STORE_AUX = 2;                        // nt

constexpr int FR_RSRC_FLAGS = 0x00020000;            // raw buffer, 32-bit data format (gfx9 family dword 3)

// One tile pair (32 output channels) of NT 16-frame tiles: 24 MFMAs per tile, then activation / BN / gap-row mask / split encoding
// and two 16-byte stores per lane, without a branch (rows past the end are dropped by the buffer descriptor's range check).
// MASK: the group has a gap row (zeroed after the arithmetic); CLAMP: clamp to the fp16 / bf8 range first -- the caller starts
// without and repeats the pair with it once the running maximum says so.
template <int MODE, bool Y8, int NT, bool MASK, bool CLAMP>
__device__ __forceinline__ void first_block(const bf16x8 (&xh)[FR_TILES][FR_NKS], const bf16x8 (&xl)[FR_TILES][FR_NKS],
                                            const bf16x8 (&wh0)[FR_NKS], const bf16x8 (&wl0)[FR_NKS], const bf16x8 (&wh1)[FR_NKS],
                                            const bf16x8 (&wl1)[FR_NKS], const f32x4 b0, const f32x4 b1, const f32x4 s0, const f32x4 s1,
                                            const f32x4 o0, const f32x4 o1, const f32x4 a0, const f32x4 a1, const int (&keepm)[FR_TILES],
                                            const unsigned (&hioff)[FR_TILES], const int slab_off, const __amdgpu_buffer_rsrc_t yrs,
                                            float &amax)
{
    f32x4 m0[NT], m1[NT];
    auto mfma_tile = [&](int j) {
        m0[j] = b0; m1[j] = b1;                                   // the bias is where the sums start
#pragma unroll
        for (int u = 0; u < FR_NKS; ++u) {
            m0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl0[u], xh[j][u], m0[j], 0, 0, 0);
            m1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl1[u], xh[j][u], m1[j], 0, 0, 0);
            m0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0[u], xl[j][u], m0[j], 0, 0, 0);
            m1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1[u], xl[j][u], m1[j], 0, 0, 0);
            m0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0[u], xh[j][u], m0[j], 0, 0, 0);
            m1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1[u], xh[j][u], m1[j], 0, 0, 0);
        }
    };
    auto epilogue = [&](int j) {
        // lane (frame f, group G) holds rows 4G..4G+3 of both tiles; the packed weights order the rows of a tile pair so that these
        // are channels 8G..8G+3 (tile 0) and 8G+4..8G+7 (tile 1) of the pair's 32: one 16-byte slot of the output format per lane
        const f32x4 z0 = m0[j], z1 = m1[j];
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {                          // activation, then BN as packed FMAs
            const f32x2 r0 = {act_fn<MODE>(z0[e], a0[e]), act_fn<MODE>(z0[e + 1], a0[e + 1])};
            const f32x2 r1 = {act_fn<MODE>(z1[e], a1[e]), act_fn<MODE>(z1[e + 1], a1[e + 1])};
            const f32x2 y0 = __builtin_elementwise_fma(r0, (f32x2){s0[e], s0[e + 1]}, (f32x2){o0[e], o0[e + 1]});
            const f32x2 y1 = __builtin_elementwise_fma(r1, (f32x2){s1[e], s1[e + 1]}, (f32x2){o1[e], o1[e + 1]});
            v[e] = y0[0]; v[e + 1] = y0[1]; v[4 + e] = y1[0]; v[5 + e] = y1[1];
        }
        if constexpr (Y8) {
            // xv_split8_encode8's results; the running maximum and the clamp come before the gap-row mask (the values are known
            // to be canonical there: one v_max3 per pair, no v_max x, x in front of it)
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            xv_i32x4 vh, vx;
            float lo[8];
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                amax = fmaxf(amax, fmaxf(fabsf(v[i]), fabsf(v[i + 1])));
                float ca = v[i], cb = v[i + 1];
                if constexpr (CLAMP) {
                    ca = __builtin_amdgcn_fmed3f(ca, -XV_SPLIT8_MAX, XV_SPLIT8_MAX);
                    cb = __builtin_amdgcn_fmed3f(cb, -XV_SPLIT8_MAX, XV_SPLIT8_MAX);
                }
                if constexpr (MASK) {
                    ca = __builtin_bit_cast(float, __builtin_bit_cast(int, ca) & keepm[j]);
                    cb = __builtin_bit_cast(float, __builtin_bit_cast(int, cb) & keepm[j]);
                }
                const f16x2 h = {(_Float16)ca, (_Float16)cb};
                vh[i >> 1] = __builtin_bit_cast(int, h);
                lo[i] = sub_f16_half<0>(ca, vh[i >> 1]);
                lo[i + 1] = sub_f16_half<1>(cb, vh[i >> 1]);
                v[i] = ca; v[i + 1] = cb;
            }
            constexpr float inv = 1.f / XV_SPLIT8_LO_SCALE;
            s16x2 t0 = {0, 0}, t1 = {0, 0};
            t0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t0, lo[0], lo[1], inv, false);
            t0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t0, lo[2], lo[3], inv, true);
            t1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t1, lo[4], lo[5], inv, false);
            t1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t1, lo[6], lo[7], inv, true);
            int h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[0], v[1], 0, false);
            h0 = __builtin_amdgcn_cvt_pk_bf8_f32(v[2], v[3], h0, true);
            int h1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[4], v[5], 0, false);
            h1 = __builtin_amdgcn_cvt_pk_bf8_f32(v[6], v[7], h1, true);
            vx = (xv_i32x4){__builtin_bit_cast(int, t0), __builtin_bit_cast(int, t1), h0, h1};
            __builtin_amdgcn_raw_buffer_store_b128(vh, yrs, hioff[j], slab_off, FR_STORE_AUX);             // slot g ...
            __builtin_amdgcn_raw_buffer_store_b128(vx, yrs, hioff[j] ^ 64u, slab_off, FR_STORE_AUX);       // ... and slot 4 + g, same swizzle
        } else {
            bf16x8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float m = MASK ? __builtin_bit_cast(float, __builtin_bit_cast(int, v[e]) & keepm[j]) : v[e];
                const __bf16 h = (__bf16)m;
                vh[e] = h;
                vl[e] = (__bf16)(m - (float)h);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xv_i32x4, vh), yrs, hioff[j], slab_off, FR_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xv_i32x4, vl), yrs, hioff[j] ^ 64u, slab_off, FR_STORE_AUX);
        }
    };
    mfma_tile(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 1; j < NT; ++j) {
        mfma_tile(j);
        epilogue(j - 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    epilogue(NT - 1);
    __builtin_amdgcn_sched_barrier(0);
}

// Y8: write XV_FMT_SPLIT8 rows (fp16 hi + bf8 cross bytes, xv_split8.h) instead of the bf16 hi/lo planes
template <int MODE, bool Y8>
__global__ __launch_bounds__(FR_WAVES * 64) void tdnn_first_kernel(const FirstParams p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = lane >> 4, f = lane & 15;
    const long tile0 = ((long)blockIdx.x * FR_WAVES + wave) * p.tpw;
    const int left = ((p.K - 1) * p.dil) >> 1;
    const int n_pass = (p.cout + FR_PASS_COLS - 1) / FR_PASS_COLS;
    const int n_groups = (p.tpw + FR_TILES - 1) / FR_TILES;
    const int steps = n_groups * n_pass;                         // a step = one pass of one group of tiles; the same for every wave
    const unsigned yrow = (unsigned)p.ychunks * SROW;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)p.row_base * yrow, 0, (int)(unsigned)(p.rows_here * yrow), FR_RSRC_FLAGS);

    // weights of pass `ps` -> buffer `buf`: n_pairs * 2 tiles * 4 k-steps * 2 KB of lane-linear fragments, 1 KB DMA pieces
    auto fetch_weights = [&](int ps, int buf) {
        const int pieces = (min(FR_PASS_COLS, p.cout - ps * FR_PASS_COLS) >> 5) * 2 * FR_NKS * 2;
        const uint8_t *src = p.wt + (size_t)ps * FR_W_BYTES + lane * 16;
        char *dst = lds + buf * FR_W_BYTES;
        for (int pc = wave; pc < pieces; pc += FR_WAVES) XV_GLDS16_OFF(src + (size_t)pc * 1024, dst + pc * 1024, 0);
    };
    // Every output row is 2 KB: workgroups that all write the same 128-byte slab of their rows at the same time send the whole
    // chip's stores to the same few memory channels.  The workgroups therefore start at different passes (and the waves of a
    // workgroup at different tile pairs of a pass, below).
    const int rot = (int)(blockIdx.x % (unsigned)n_pass);
    fetch_weights(rot, 0);
    // epilogue parameters -> LDS (the first step's barrier publishes them)
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

    bf16x8 xh[FR_TILES][FR_NKS], xl[FR_TILES][FR_NKS];   // B operands of the current group: im2col fragments, fp32 -> hi/lo
    int keepm[FR_TILES];                      // all ones: the row is a frame; zero: gap row, written as zeros
    unsigned hioff[FR_TILES];                 // byte offset of the lane's hi slot in slab 0 of its row, in this launch's part of y
                                              // (rows past the end: an offset the buffer descriptor's range check drops)
    bool group_masked = false, clamping = false;   // (wave-uniform) a gap row in the group; a value beyond the fp16 / bf8 range was seen
    int n_live = 0;                           // (wave-uniform) tiles of the current group that exist: the first n_live
    float amax = 0.f;
    int pass = rot, group = 0, in_group = 0;   // in_group: passes of the current group done so far
    int stores_behind_fetch = -1;             // global stores this wave has issued since its last weight fetch (-1: unknown)

    for (int s = 0; s < steps; ++s) {
        if (in_group == 0) {
            // ---- a new group of tiles: B operands ONCE for all 512 channels.  lane (f, G), k-step u: k = 32u + 8G .. +7 = (tap, channels c0 .. c0+7)
            const long gtile = tile0 + (long)group * FR_TILES;
            const long lrow0 = gtile * 16;                         // first row of the group, relative to row_base
            n_live = 0;
#pragma unroll
            for (int j = 0; j < FR_TILES; ++j) {
                const bool ex = group * FR_TILES + j < p.tpw && gtile + j < p.n_tiles;
                n_live += ex ? 1 : 0;
                const long lrow = lrow0 + 16 * j + f, row = p.row_base + lrow;
                const bool in = ex && lrow < p.rows_here;
                keepm[j] = (in && (!p.valid || p.valid[row])) ? -1 : 0;
                hioff[j] = in ? (unsigned)lrow * yrow + (unsigned)((G ^ ((int)(row >> 1) & 7)) << 4) : 0x80000000u;
            }
#pragma unroll
            for (int u = 0; u < FR_NKS; ++u) {
                const int k = 32 * u + 8 * G;
                const int tap = k / p.kc;
                const bool live = tap < p.K;
                const int tap_off = tap * p.dil - left;
                const int c0 = k - tap * p.kc;
#pragma unroll
                for (int j = 0; j < FR_TILES; ++j) {
                    const long r = p.row_base + lrow0 + 16 * j + f + tap_off;
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
                    if (live && j < n_live && r >= 0 && r < p.R) {
                        const float *src = p.x + (size_t)r * p.ldx + c0;
                        a = *reinterpret_cast<const f32x4 *>(src);
                        b = *reinterpret_cast<const f32x4 *>(src + 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const __bf16 ha = (__bf16)a[e], hb = (__bf16)b[e];
                        xh[j][u][e] = ha; xh[j][u][4 + e] = hb;
                        xl[j][u][e] = (__bf16)(a[e] - (float)ha); xl[j][u][4 + e] = (__bf16)(b[e] - (float)hb);
                    }
                }
            }
            group_masked = __builtin_amdgcn_ballot_w64(keepm[0] == 0 || keepm[1] == 0 || keepm[2] == 0) != 0;
            stores_behind_fetch = -1;
        }
        // ---- this step's weights have landed (fetched a step ahead), everybody is done with the other buffer --------------------
        // A wave's vector-memory operations retire in order, stores included: waiting for the fetch means waiting for everything
        // issued before it, NOT for the stores of the previous step that were issued behind it -- when their number is known.
        if (stores_behind_fetch == 2 * 4 * FR_TILES) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (stores_behind_fetch == 2 * 4 * (FR_TILES - 1)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int n_pairs = min(FR_PASS_COLS, p.cout - pass * FR_PASS_COLS) >> 5;      // tile pairs (32 channels) of this pass
        if (s + 1 < steps) fetch_weights(pass + 1 == n_pass ? 0 : pass + 1, (s + 1) & 1);
        stores_behind_fetch = 0;
        const char *wbuf = lds + (s & 1) * FR_W_BYTES + lane * 16;

        if (n_live > 0) {
            for (int t = 0; t < n_pairs; ++t) {
                const int tp = (t + wave) % n_pairs;
                // weights and parameters of this tile pair: registers, for the wave's three tiles
                bf16x8 wh0[FR_NKS], wl0[FR_NKS], wh1[FR_NKS], wl1[FR_NKS];
                const char *w0 = wbuf + (size_t)(tp * 2) * (FR_NKS * 2048);
#pragma unroll
                for (int u = 0; u < FR_NKS; ++u) {
                    wh0[u] = *reinterpret_cast<const bf16x8 *>(w0 + u * 2048);
                    wl0[u] = *reinterpret_cast<const bf16x8 *>(w0 + u * 2048 + 1024);
                    wh1[u] = *reinterpret_cast<const bf16x8 *>(w0 + FR_NKS * 2048 + u * 2048);
                    wl1[u] = *reinterpret_cast<const bf16x8 *>(w0 + FR_NKS * 2048 + u * 2048 + 1024);
                }
                const int cb = pass * FR_PASS_COLS + tp * 32 + 8 * G;
                const int c4 = cb >> 2;
                const f32x4 b0 = P4[c4], b1 = P4[c4 + 1], s0 = P4[FR_MAX_COUT / 4 + c4], s1 = P4[FR_MAX_COUT / 4 + c4 + 1],
                            o0 = P4[2 * FR_MAX_COUT / 4 + c4], o1 = P4[2 * FR_MAX_COUT / 4 + c4 + 1];
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
                if constexpr (MODE != 2) { a0 = P4[3 * FR_MAX_COUT / 4 + c4]; a1 = P4[3 * FR_MAX_COUT / 4 + c4 + 1]; }
                const int slab_off = (pass * (FR_PASS_COLS / 32) + tp) * SROW;       // (wave-uniform: the store's scalar offset)
                // fast form first: no gap-row mask when the group has no gap row, no clamp; a value beyond the range (absurd for a
                // BN-normalised network; the caller repeats the batch in bf16x3 anyway) sends the wave through the clamping form
                // from then on -- the pair is done again, its stores simply land twice
                auto run = [&](auto nt, auto mask, auto clamp) {
                    first_block<MODE, Y8, decltype(nt)::value, decltype(mask)::value, decltype(clamp)::value>(
                        xh, xl, wh0, wl0, wh1, wl1, b0, b1, s0, s1, o0, o1, a0, a1, keepm, hioff, slab_off, yrs, amax);
                };
                auto run_nt = [&](auto mask, auto clamp) {
                    if (n_live == 3) run(std::integral_constant<int, 3>{}, mask, clamp);
                    else if (n_live == 2) run(std::integral_constant<int, 2>{}, mask, clamp);
                    else run(std::integral_constant<int, 1>{}, mask, clamp);
                };
                if (Y8 && clamping) run_nt(std::true_type{}, std::true_type{});
                else {
                    if (group_masked) run_nt(std::true_type{}, std::false_type{});
                    else run_nt(std::false_type{}, std::false_type{});
                    if (Y8 && __builtin_amdgcn_ballot_w64(!(amax <= XV_SPLIT8_MAX)) != 0) {
                        clamping = true;
                        run_nt(std::true_type{}, std::true_type{});
                        stores_behind_fetch = -1000;              // (the count is off now: the next wait drains)
                    }
                }
                stores_behind_fetch += 2 * n_live;
            }
        }
        if (++pass == n_pass) pass = 0;
        if (++in_group == n_pass) { in_group = 0; ++group; }
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

void xv_internal_first_tiles(int tiles) { g_first_tiles.store(tiles, std::memory_order_relaxed); }

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
    // one workgroup per CU, each wave a contiguous run of 16-frame tiles (at least one): the weights cycle through LDS once per
    // FR_TILES tiles of every wave, and nobody ends a round early
    static std::atomic<int> n_cu{0};
    int cus = n_cu.load(std::memory_order_relaxed);
    if (cus <= 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu.store(cus, std::memory_order_relaxed);
    }
    const int forced = g_first_tiles.load(std::memory_order_relaxed);
    const long waves = (long)cus * FR_WAVES;
    constexpr long MAX_ROWS = 1L << 19;                            // x 2 KB per row = 1 GB: one buffer descriptor of y, offsets < 2^31
    for (long r0 = 0; r0 < p.R; r0 += MAX_ROWS) {
        p.row_base = r0;
        p.rows_here = p.R - r0 < MAX_ROWS ? p.R - r0 : MAX_ROWS;
        p.n_tiles = (p.rows_here + 15) / 16;
        p.tpw = forced > 0 ? forced : (int)((p.n_tiles + waves - 1) / waves);
        const long per_wg = (long)p.tpw * FR_WAVES;
        hipLaunchKernelGGL(kerns[mode + (y8 ? 3 : 0)], dim3((unsigned)((p.n_tiles + per_wg - 1) / per_wg)), dim3(FR_WAVES * 64), FR_LDS_BYTES,
                           (hipStream_t)stream, p);
    }
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
