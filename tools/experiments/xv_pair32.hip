// xv_pair32.hip -- EXPERIMENT: the pair kernel of xv_pair.hip with 32 frames per wave (v_mfma_f32_32x32x16_bf16, 4 waves per
// workgroup, one wave per SIMD, 512 registers per lane).  A weight fragment then feeds 32 frames instead of 16, which halves
// the LDS read rate that bounds the 16-frame form (2 ds_read_b128 per 3 MFMAs of 16 cycles, 8 waves reading the same
// fragments).  Same algorithm, same packed-stage idea; entry points xv_x_* are NOT part of the ABI header (tools only) until
// the variant has proven itself.
//
//   phase 1  H^T[channel][frame] = W1^T . X^T : A = weight fragment (32 channels x 16 k), B = frames fragment (32 frames x 16 k);
//            D tile T: lane (frame f = lane&31, half kh = lane>>5), register r: channel 32T + (r&3) + 8(r>>2) + 4kh
//   convert  registers 0..7 / 8..15 of tile T = the A operand of k-steps 2T / 2T+1 of the second GEMM, k order
//            e -> channel 32T + 16(u&1) + (e&3) + 8(e>>2) + 4kh  (the packed weights of the second layer follow it)
//   phase 2  Y[frame][column] = H . W2 for 64 columns at a time (2 accumulators of 32 x 32), pooled on the spot:
//            D lane (column = lane&31, kh), register r: frame (r&3) + 8(r>>2) + 4kh -> block r>>2, rows 4kh + (r&3) of it
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "xvector_hip.h"

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

constexpr int CMID = 512;
constexpr int P3_WAVES = 4;
constexpr int P3_ROWS = 32 * P3_WAVES;
constexpr int P3_STAGE = 32768;            // 16 (hi, lo) fragment pairs of 1 KB + 1 KB
constexpr int P3_RING = 3;
constexpr int P3_X_OFF = P3_RING * P3_STAGE;                 // per wave 2 slots x [hi 1 KB | lo 1 KB]
constexpr int P3_P1_OFF = P3_X_OFF + P3_WAVES * 4096;
constexpr int P3_P2_OFF = P3_P1_OFF + 4 * CMID * 4;
constexpr int SROW = 128;

struct Pair32Params {
    const uint8_t *x;
    long R;
    int n_ks;                  // cin / 16 : k-steps (= stages) of the first GEMM
    int cout, n_ct;            // n_ct = cout / 64
    const uint8_t *wt;         // n_ks stages of layer 1, then 4*n_ct stages of layer 2
    const float *b1, *sc1, *sh1, *al1;
    const float *b2, *sc2, *sh2, *al2;
    int act;
    const uint8_t *valid;
    float *blk;
    long n_blocks;
};

#define XV_GLDS16_OFF(gptr, lptr, imm)                                                                          \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                    \
                                     (__attribute__((address_space(3))) void *)(lptr), 16, imm, 0)

struct Frags {
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

template <int MODE>
__device__ __forceinline__ float act_fn(float z, float a)
{
    return MODE == 1 ? fmaxf(a * z, z) : MODE == 2 ? fmaxf(z, 0.f) : fmaxf(z, 0.f) + a * fminf(z, 0.f);
}

template <int MODE>
__global__ __launch_bounds__(P3_WAVES * 64, 1) void tdnn_pair_pool32_kernel(const Pair32Params p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = lane >> 5, fl = lane & 31;
    const long row0 = (long)blockIdx.x * P3_ROWS + 32 * wave;        // this wave's 32 frames = 4 pooling blocks

    // row validity of the frames this lane meets in the pooling epilogue: block b = r>>2, row 8b + 4kh + (r&3)
    float keep[16], nblk[4], rn[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float n = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long gr = row0 + 8 * b + 4 * kh + q;
            keep[4 * b + q] = (gr < p.R && (!p.valid || p.valid[gr])) ? 1.f : 0.f;
            n += keep[4 * b + q];
        }
        n += __shfl_xor(n, 32, 64);
        nblk[b] = n;
        rn[b] = n > 0.f ? 1.f / n : 0.f;
    }
    {
        float *P1 = reinterpret_cast<float *>(lds + P3_P1_OFF);
        for (int c = tid; c < CMID; c += P3_WAVES * 64) {
            P1[c] = p.b1 ? p.b1[c] : 0.f;
            P1[CMID + c] = p.sc1 ? p.sc1[c] : 1.f;
            P1[2 * CMID + c] = p.sh1 ? p.sh1[c] : 0.f;
            P1[3 * CMID + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.al1[0] : p.act == XV_ACT_PRELU ? p.al1[c] : 0.f;
        }
        f32x4 *P2 = reinterpret_cast<f32x4 *>(lds + P3_P2_OFF);
        for (int c = tid; c < p.cout; c += P3_WAVES * 64) {
            f32x4 v;
            v[0] = p.b2 ? p.b2[c] : 0.f;
            v[1] = p.sc2 ? p.sc2[c] : 1.f;
            v[2] = p.sh2 ? p.sh2[c] : 0.f;
            v[3] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.al2[0] : p.act == XV_ACT_PRELU ? p.al2[c] : 0.f;
            P2[c] = v;
        }
    }
    __syncthreads();

    // ---- DMA streams: a wave moves 8 KB of every 32 KB stage --------------------------------------------------------
    const uint8_t *wsrc = p.wt + wave * 8192 + lane * 16;
    int wleft = p.n_ks + 4 * p.n_ct;
    auto issue_w = [&](int slot_off) {
        char *dst = lds + slot_off + wave * 8192;
        XV_GLDS16_OFF(wsrc, dst, 0);                          // (the immediate offset field ends at 4095)
        XV_GLDS16_OFF(wsrc, dst, 1024);
        XV_GLDS16_OFF(wsrc, dst, 2048);
        XV_GLDS16_OFF(wsrc, dst, 3072);
        XV_GLDS16_OFF(wsrc + 4096, dst + 4096, 0);
        XV_GLDS16_OFF(wsrc + 4096, dst + 4096, 1024);
        XV_GLDS16_OFF(wsrc + 4096, dst + 4096, 2048);
        XV_GLDS16_OFF(wsrc + 4096, dst + 4096, 3072);
        const bool more = wleft > 1;
        wsrc += more ? P3_STAGE : 0;
        wleft -= more ? 1 : 0;
    };
    // frames fragment of k-step s (16 channels): lane (frame, kh) fetches the 16-byte slot t = 2(s&1) + kh of slab s>>1 (hi)
    // and 4 + t (lo); physical slot = logical ^ ((row>>1)&7).  Two slots per wave, k-step s lands in slot s&1.
    const int sw = (int)((row0 + fl) >> 1) & 7;
    const size_t xrow_bytes = (size_t)(p.n_ks >> 1) * SROW;
    const uint8_t *xrow = p.x + (row0 + fl) * (long)xrow_bytes;
    int xs = 0;                                                       // next k-step to fetch (clamped at the last)
    auto issue_x = [&]() {
        const int s = xs < p.n_ks ? xs : p.n_ks - 1;
        const uint8_t *slab = xrow + (size_t)(s >> 1) * SROW;
        const int t = 2 * (s & 1) + kh;
        char *dst = lds + P3_X_OFF + wave * 4096 + (xs & 1) * 2048;
        XV_GLDS16_OFF(slab + ((t ^ sw) << 4), dst, 0);
        XV_GLDS16_OFF(slab + (((4 + t) ^ sw) << 4), dst + 1024, 0);
        ++xs;
    };
    const char *fbase = lds + lane * 16;
    auto load_frags = [&](Frags &F, int slot, int sub) {
        const char *b = fbase + slot + sub * 8192;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            F.hi[c] = *reinterpret_cast<const bf16x8 *>(b + c * 2048);
            F.lo[c] = *reinterpret_cast<const bf16x8 *>(b + c * 2048 + 1024);
        }
    };
    auto next_slot = [](int slot) { return slot + P3_STAGE == P3_RING * P3_STAGE ? 0 : slot + P3_STAGE; };
    // B(t): my fragment reads of stage t are complete, my DMA pieces of stage t+1 and of the frames fragment t+1 have landed
    // (only the 8 pieces of stage t+2 -- issued last, after the frames pieces -- may still be in flight)
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto pin = [&](auto NVMEM) {
        constexpr int nv = decltype(NVMEM)::value;
#pragma unroll
        for (int i = 0; i < nv; ++i) {
            if (i < 12) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < 12 - nv) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12 - nv - 8 > 0 ? 12 - nv - 8 : 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue -------------------------------------------------------------------------------------------------
    issue_x();                      // k-step 0
    issue_w(0);
    issue_x();                      // k-step 1
    issue_w(P3_STAGE);
    issue_w(2 * P3_STAGE);
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");               // frames 0 and stage 0
    __builtin_amdgcn_s_barrier();
    Frags F, Gf;
    load_frags(F, 0, 0);
    const char *xfrag = lds + P3_X_OFF + wave * 4096 + lane * 16;
    bf16x8 xfh = *reinterpret_cast<const bf16x8 *>(xfrag);
    bf16x8 xfl = *reinterpret_cast<const bf16x8 *>(xfrag + 1024);
    int slot = 0;

    // ---- phase 1: one stage per k-step, 16 channel tiles of 32 x 32 ----------------------------------------------------
    f32x16 acc[CMID / 32];
#pragma unroll
    for (int t = 0; t < CMID / 32; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto mma1 = [&](const Frags &W, auto T0) {
        constexpr int t0 = decltype(T0)::value;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.lo[c], xfh, acc[t0 + c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.hi[c], xfl, acc[t0 + c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.hi[c], xfh, acc[t0 + c], 0, 0, 0);
    };
    for (int s = 0; s < p.n_ks; ++s) {
        load_frags(Gf, slot, 1);
        mma1(F, std::integral_constant<int, 0>{});
        pin(std::integral_constant<int, 0>{});
        load_frags(F, slot, 2);
        mma1(Gf, std::integral_constant<int, 4>{});
        pin(std::integral_constant<int, 0>{});
        load_frags(Gf, slot, 3);
        mma1(F, std::integral_constant<int, 8>{});
        pin(std::integral_constant<int, 0>{});
        stage_barrier();
        issue_x();                                                 // k-step s+2 into the slot k-step s was read from
        issue_w(slot);
        slot = next_slot(slot);
        load_frags(F, slot, 0);
        const char *nx = xfrag + ((s + 1) & 1) * 2048;
        const bf16x8 nh = *reinterpret_cast<const bf16x8 *>(nx);
        const bf16x8 nl = *reinterpret_cast<const bf16x8 *>(nx + 1024);
        mma1(Gf, std::integral_constant<int, 12>{});
        __builtin_amdgcn_sched_barrier(0);
        xfh = nh;
        xfl = nl;
    }

    // ---- accumulators -> A operands of the second GEMM ------------------------------------------------------------------
    bf16x8 Hh[CMID / 16], Hl[CMID / 16];
    {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        auto conv = [&](auto TT) {
            constexpr int T = decltype(TT)::value;
            int dep = 0;
            if constexpr (T > 0) {
                const u32x4 a = __builtin_bit_cast(u32x4, Hh[2 * T - 1]), b = __builtin_bit_cast(u32x4, Hl[2 * T - 1]);
                const u32x4 c = __builtin_bit_cast(u32x4, Hh[2 * T - 2]), d = __builtin_bit_cast(u32x4, Hl[2 * T - 2]);
                asm volatile("" : "+v"(dep) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]),
                             "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
            }
            const f32x4 *P1 = reinterpret_cast<const f32x4 *>(lds + P3_P1_OFF + dep);
#pragma unroll
            for (int q = 0; q < 4; ++q) {                           // registers 4q .. 4q+3: channels 32T + 8q + 4kh .. +3
                const int c4 = (32 * T + 8 * q + 4 * kh) >> 2;
                const f32x4 b = P1[c4], sc = P1[CMID / 4 + c4], o = P1[2 * CMID / 4 + c4], a = P1[3 * CMID / 4 + c4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = act_fn<MODE>(acc[T][4 * q + e] + b[e], a[e]) * sc[e] + o[e];
                    const __bf16 hi = (__bf16)v;
                    Hh[2 * T + (q >> 1)][4 * (q & 1) + e] = hi;
                    Hl[2 * T + (q >> 1)][4 * (q & 1) + e] = (__bf16)(v - (float)hi);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, CMID / 32>(conv);
    }
    load_frags(F, slot, 0);

    // ---- phase 2: 64 columns at a time, 4 stages of 8 k-steps each --------------------------------------------------------
    const f32x4 *P2 = reinterpret_cast<const f32x4 *>(lds + P3_P2_OFF);
    for (int ct = 0; ct < p.n_ct; ++ct) {
        f32x16 y0, y1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { y0[r] = 0.f; y1[r] = 0.f; }
        // sub-step = 2 k-steps x 2 column tiles: fragments [k-step][tile] = hi/lo[2*kk + tile]
        auto mma2 = [&](const Frags &W, auto U0) {
            constexpr int u0 = decltype(U0)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Hl[u0 + kk], W.hi[2 * kk], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Hl[u0 + kk], W.hi[2 * kk + 1], y1, 0, 0, 0);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Hh[u0 + kk], W.lo[2 * kk], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Hh[u0 + kk], W.lo[2 * kk + 1], y1, 0, 0, 0);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Hh[u0 + kk], W.hi[2 * kk], y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Hh[u0 + kk], W.hi[2 * kk + 1], y1, 0, 0, 0);
            }
        };
        auto quarter = [&](auto Q) {
            constexpr int q = decltype(Q)::value;                    // k-steps 8q .. 8q+7
            load_frags(Gf, slot, 1);
            mma2(F, std::integral_constant<int, 8 * q>{});
            pin(std::integral_constant<int, 0>{});
            load_frags(F, slot, 2);
            mma2(Gf, std::integral_constant<int, 8 * q + 2>{});
            pin(std::integral_constant<int, 0>{});
            load_frags(Gf, slot, 3);
            mma2(F, std::integral_constant<int, 8 * q + 4>{});
            pin(std::integral_constant<int, 0>{});
            stage_barrier();
            issue_w(slot);
            slot = next_slot(slot);
            load_frags(F, slot, 0);
            mma2(Gf, std::integral_constant<int, 8 * q + 6>{});
            pin(std::integral_constant<int, 8>{});
        };
        static_for<0, 4>(quarter);

        // pooling epilogue: lane (column, kh) holds rows 4kh + (r&3) of block r>>2; the partner lane^32 the other four
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x16 &y = c == 0 ? y0 : y1;
            const int col = ct * 64 + c * 32 + fl;
            const f32x4 prm = P2[col];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = act_fn<MODE>(y[4 * b + q] + prm[0], prm[3]) * prm[1] + prm[2];
                const float v0 = __shfl(v[0], fl, 64);               // first row of the block: lane half 0
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = (v[q] - v0) * keep[4 * b + q];
                    s1 += d;
                    s2 += d * d;
                }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                const float mean = nblk[b] > 0.f ? v0 + s1 * rn[b] : 0.f;
                const float m2 = fmaxf(s2 - s1 * s1 * rn[b], 0.f);
                const long br = (row0 >> 3) + b;
                if (br < p.n_blocks) p.blk[(size_t)br * 2 * p.cout + (kh ? p.cout : 0) + col] = kh ? m2 : mean;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// layer 1, stage s (k-step of 16), fragment pair T (0..15), lane (i, kh), element e:  w1[16s + 8kh + e][32T + i]
// layer 2, stage 4*ct + q, fragment pair f = 4*sub + 2*kk + tile (sub 0..3, kk 0..1, tile 0..1), k-step u = 8q + 2*sub + kk:
//          w2[32(u>>1) + 16(u&1) + (e&3) + 8(e>>2) + 4kh][64ct + 32tile + i]
__global__ void pack_pair32_kernel(const float *__restrict__ w1, const float *__restrict__ w2, int n_ks, int cout, uint8_t *__restrict__ wt,
                                   size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    const int fp = (int)((i >> 9) & 15);
    const long stage = (long)(i >> 13);
    const int j = lane & 31, kh = lane >> 5;
    float x;
    if (stage < n_ks) {
        x = w1[(size_t)(16 * stage + 8 * kh + e) * CMID + 32 * fp + j];
    } else {
        const long s2 = stage - n_ks;
        const int ct = (int)(s2 >> 2), q = (int)(s2 & 3);
        const int sub = fp >> 2, kk = (fp >> 1) & 1, tile = fp & 1;
        const int u = 8 * q + 2 * sub + kk;
        const int ch = 32 * (u >> 1) + 16 * (u & 1) + (e & 3) + 8 * (e >> 2) + 4 * kh;
        x = w2[(size_t)ch * cout + 64 * ct + 32 * tile + j];
    }
    const __bf16 hi = (__bf16)x;
    const __bf16 lo = (__bf16)(x - (float)hi);
    uint8_t *t = wt + (size_t)stage * P3_STAGE + (size_t)fp * 2048 + lane * 16 + e * 2;
    *reinterpret_cast<uint16_t *>(t) = __builtin_bit_cast(uint16_t, hi);
    *reinterpret_cast<uint16_t *>(t + 1024) = __builtin_bit_cast(uint16_t, lo);
}

}  // namespace

extern "C" {

size_t xv_x_packed_pair32_bytes(int cin, int cout) { return (size_t)(cin / 16 + 4 * (cout / 64)) * P3_STAGE; }

int xv_x_pack_pair32(const float *w1, const float *w2, int cin, int cout, void *wt, void *stream)
{
    if (!w1 || !w2 || !wt || (cin & 31) || (cout & 63) || cout > 2048) return fail(XV_ERR_BAD_ARG, "pack_pair32: bad argument");
    const size_t total = xv_x_packed_pair32_bytes(cin, cout) / 4;
    hipLaunchKernelGGL(pack_pair32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1, w2, cin / 16, cout,
                       (uint8_t *)wt, total);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

int xv_x_tdnn_pair_pool32(const void *x, int64_t R, int cin, int cout, const void *wt, const float *bias1, const float *bn_scale1,
                          const float *bn_shift1, const float *act_alpha1, const float *bias2, const float *bn_scale2,
                          const float *bn_shift2, const float *act_alpha2, int act_kind, const uint8_t *row_valid, float *block_stats,
                          void *stream)
{
    if (R <= 0) return 0;
    if (!x || !wt || !block_stats || (cin & 31) || (cout & 63) || cout > 2048) return fail(XV_ERR_BAD_ARG, "pair_pool32: bad argument");
    Pair32Params p{};
    p.x = (const uint8_t *)x; p.R = (long)R; p.n_ks = cin / 16; p.cout = cout; p.n_ct = cout / 64; p.wt = (const uint8_t *)wt;
    p.b1 = bias1; p.sc1 = bn_scale1; p.sh1 = bn_shift1; p.al1 = act_alpha1;
    p.b2 = bias2; p.sc2 = bn_scale2; p.sh2 = bn_shift2; p.al2 = act_alpha2;
    p.act = act_kind; p.valid = row_valid; p.blk = block_stats; p.n_blocks = (long)((R + 7) / 8);
    const size_t lds_bytes = (size_t)P3_P2_OFF + (size_t)cout * 16;
    typedef void (*kern_t)(const Pair32Params);
    const kern_t kerns[3] = {tdnn_pair_pool32_kernel<0>, tdnn_pair_pool32_kernel<1>, tdnn_pair_pool32_kernel<2>};
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (kern_t k : kerns) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, P3_P2_OFF + 2048 * 16);
            if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int mode = act_kind == XV_ACT_LRELU ? 1 : act_kind == XV_ACT_RELU ? 2 : 0;
    hipLaunchKernelGGL(kerns[mode], dim3((unsigned)((R + P3_ROWS - 1) / P3_ROWS)), dim3(P3_WAVES * 64), lds_bytes, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
