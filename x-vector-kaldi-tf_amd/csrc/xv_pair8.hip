// xv_pair8.hip -- the two context-free (K = 1) frame-level layers and the first half of statistics pooling as ONE kernel
// (frame_level_info_layer-3 -> -4 -> per-8-row block statistics; local/tf/models.py:54-76 / 470-486), in the f16bf8
// arithmetic of xv_split8.h and with the reduction of the second GEMM split over a PAIR of waves.
//
// Why a second form of xv_pair.hip.  That kernel keeps the intermediate activation of 16 frames per wave in registers and
// is bound by the LDS: every wave streams ALL weights of both layers through its fragment reads (36 MB of LDS traffic per
// 128-frame workgroup = its whole run time).  Here
//   * a PAIR of waves owns 32 frames; wave h of the pair computes channels [256 h, 256 h + 256) of the intermediate H for
//     all 32 frames (v_mfma_*_32x32*: H^T tile = 32 channels x 32 frames) and keeps them in registers (128 VGPRs, as
//     before) -- so it only ever reads HALF of the first layer's weights;
//   * in the second GEMM wave h contracts over ITS 256 channels only (half of the second layer's weights).  Of a 32-frame x
//     64-column tile it FINISHES the 32 columns of half h (KEEP) and hands the partner its partial sums of the other 32 (RED):
//     the two halves are accumulated one after the other, so RED is complete in the middle of a tile and goes to LDS there
//     (ds_write_addtid_b32, 4 KB per wave), while KEEP never leaves the registers -- eight v_permlane32_swap give a lane the
//     sixteen rows of its column that make up two whole 8-row pooling blocks, and it pools them (one ds_read of the partner's
//     partial sum per two rows) in the shadow of the next tile's first two stages.  Round 4; before, both halves went through
//     LDS and back.
// LDS traffic per workgroup: 16 MB of fragment reads + 4 MB of weight DMA instead of 32 + 4.
//   * products are formed as in xv_gemm8.hip: one v_mfma_f32_32x32x16_f16 on the fp16 parts + one
//     v_mfma_scale_f32_32x32x64_f8f6f4 whose K = 64 holds [xl8 . wh8 | xh8 . wl8] of a 32-channel slab, i.e. 32 MFMA passes
//     per 32x32x32 block where bf16x3 spends 48.
// The accumulator of an H tile holds, in lane (frame f = lane & 31, hh = lane >> 5), the 16 channels c(r) = (r&3) + 8(r>>2)
// + 4 hh -- after bias / activation / BN and the split8 encoding exactly the A operand (32 frames x its K share) of the second
// GEMM for that 32-channel slab: fp16 k-step ks takes r = 8 ks .. 8 ks + 7, the scaled MFMA's K-block hh takes
// [l8 r0..7 | h8 r0..7 | l8 r8..15 | h8 r8..15]; the packed weights of layer 4 follow that order.  No LDS round trip.
// Weight stream: 32 KB stages [half 0: 4 units][half 1: 4 units], unit = [fp16 fragment k-step 0 | k-step 1 | 8-bit
// fragment, two 1 KB planes], lane-linear 16-byte fragments; ring of three.  ONE barrier per stage, between its third and
// fourth unit (the fourth unit's fragments are in registers by then): behind it the slot is refilled with stage s+3, so a
// DMA has two full stages to land and the barrier's wait is a counted vmcnt(4), never a drain.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "xvector_hip.h"
#include "xv_split8.h"

extern "C" void xv_internal_set_error(const char *msg);

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

int fail(int code, const char *msg)
{
    xv_internal_set_error(msg);
    return code;
}

constexpr int CMID = 512;                  // width of the intermediate layer
constexpr int P8_WAVES = 8;
constexpr int P8_ROWS = 128;               // frames per workgroup: 4 pairs x 32
constexpr int P8_STAGE = 32768;
constexpr int P8_RING = 3;
constexpr int P8_X_OFF = P8_RING * P8_STAGE;           // phase 1: per pair the 4 KB frames fragments of one slab
constexpr int P8_RED_OFF = P8_X_OFF;                   // phase 2: per wave the 4 KB half tile for the partner, tiles of even parity ...
constexpr int P8_KEEP_OFF = P8_X_OFF + P8_WAVES * 4096; // ... and of odd parity (over the dead first-layer parameters)
constexpr int P8_P1_OFF = P8_X_OFF + P8_WAVES * 4096;  // [bias | scale | shift | alpha][CMID] of the first layer
constexpr size_t P8_LDS_BYTES = P8_KEEP_OFF + P8_WAVES * 4096;      // 160 KB: everything a CU has
constexpr int SROW = 128;

struct Pair8Params {
    const uint8_t *x;          // split8 input, row 0
    long R;
    int n_ks;                  // cin / 32
    int cout, n_ct;            // n_ct = cout / 64
    const uint8_t *wt;         // packed stages: 2*n_ks of layer 1, then 4*n_ct of layer 2
    const float *b1, *sc1, *sh1, *al1;
    const float *b2, *sc2, *sh2, *al2;
    int act;
    const uint8_t *valid;
    float *blk;                // [ceil(R/8)][2][cout]
    long n_blocks;
    int *status;               // bit 0: the intermediate left the fp16 range (clamped)
};

// One 1 KB LDS-DMA piece: 16 bytes per lane from buffer `rsrc` at byte voff (per lane) + soff (wave-uniform) + imm to LDS
// lptr + imm + 16 * lane.  The MUBUF form on purpose: the FLAT-encoded global_load_lds marks "a flat access is pending" in the
// compiler's wait-count bookkeeping for as long as any piece is in flight -- here always -- and every s_waitcnt it inserts for an
// LDS read then degrades to lgkmcnt(0): a step could not start its first MFMA before ALL fragment reads of the previous step
// had returned.
#define XV_BLDS16(rsrc, lptr, voff, soff, imm)                                                                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lptr), 16, voff, soff, imm, 0)
constexpr int XV_RSRC_FLAGS = 0x00020000;              // raw buffer, 32-bit data format (gfx9 family dword 3)

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
__global__ __launch_bounds__(P8_WAVES * 64, 2) void tdnn_pair_pool_f16bf8_kernel(const Pair8Params p)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = wave >> 1, hf = wave & 1;
    const int hh = lane >> 5, li = lane & 31;
    const long row0 = (long)blockIdx.x * P8_ROWS + 32 * pair;     // the pair's 32 frames (a multiple of 8: four pooling blocks)

    // ---- DMA streams -------------------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(p.wt), 0, 0x7ffffff0, XV_RSRC_FLAGS);
    const int wvoff = wave * 4096 + lane * 16;
    int wsoff = 0;
    int wleft = 2 * p.n_ks + 4 * p.n_ct;               // stages not yet issued (the tail re-issues the last stage)
    int fill = 0;                                      // ring slot being refilled
    // One piece per call: an LDS-DMA instruction holds the wave's issue for 60-180 cycles, so the four pieces of a stage go
    // behind four different MFMAs (piece 0 in the step behind the barrier that vacated the slot, 1-3 in the next three steps)
    // instead of in one burst that only a single MFMA covers.
    auto w_piece = [&](auto K) {
        constexpr int k = decltype(K)::value;
        XV_BLDS16(wrs, lds + fill + wave * 4096, wvoff, wsoff, k * 1024);
        if constexpr (k == 3) {
            const bool more = wleft > 1;
            wsoff += more ? P8_STAGE : 0;
            wleft -= more ? 1 : 0;
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;
    typedef std::integral_constant<int, 3> I3;
    auto issue_w = [&](int slot_off) {                 // (prologue only)
        fill = slot_off;
        w_piece(I0{}); w_piece(I1{}); w_piece(I2{}); w_piece(I3{});
    };
    // frames of the pair, slab ks: [fp16 k-step 0 | fp16 k-step 1 | 8-bit plane 0 | plane 1], 1 KB each, lane (f, hh) = 16 bytes:
    // logical slot 2 ks16 + hh (fp16) / 4 + 2 hh + e (8-bit plane e) of row row0 + f; physical slot = logical ^ ((row >> 1) & 7).
    // Wave 0 of the pair fetches the two fp16 pieces, wave 1 the two 8-bit planes.
    const int sw = (int)((row0 + li) >> 1) & 7;
    const size_t xrow_bytes = (size_t)p.n_ks * SROW;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(p.x) + (size_t)blockIdx.x * P8_ROWS * xrow_bytes, 0, 0x7ffffff0, XV_RSRC_FLAGS);
    const int xrow = (32 * pair + li) * (int)xrow_bytes;            // (the workgroup's 128 rows: < 2^31 bytes for any n_ks)
    const int xa = xrow + (((hf ? 4 + 2 * hh : hh) ^ sw) << 4);
    const int xb = xrow + (((hf ? 5 + 2 * hh : 2 + hh) ^ sw) << 4);
    int xsoff = 0;
    int xleft = p.n_ks;
    char *xbase = lds + P8_X_OFF + pair * 4096;
    auto issue_x = [&]() {
        XV_BLDS16(xrs, xbase + hf * 2048, xa, xsoff, 0);
        XV_BLDS16(xrs, xbase + hf * 2048 + 1024, xb, xsoff, 0);
        const bool more = xleft > 1;
        xsoff += more ? SROW : 0;
        xleft -= more ? 1 : 0;
    };
    struct XFrag {
        xv_f16x8 h0, h1;
        xv_i32x8 x;
    };
    auto cat = [](xv_i32x4 u, xv_i32x4 v) { return __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7); };
    auto load_xfrag = [&](XFrag &X) {
        const char *b = xbase + lane * 16;
        X.h0 = *reinterpret_cast<const xv_f16x8 *>(b);
        X.h1 = *reinterpret_cast<const xv_f16x8 *>(b + 1024);
        X.x = cat(*reinterpret_cast<const xv_i32x4 *>(b + 2048), *reinterpret_cast<const xv_i32x4 *>(b + 3072));
    };

    // Fragment sets (ping-pong): fp16 fragments of a unit in Hh0/Hh1[set], its 8-bit fragment in Mx[set].  The MFMAs of a unit
    // are issued SKEWED -- fp16 k-step 0 of unit j, the 8-bit MFMA of unit j-1, fp16 k-step 1 of unit j -- so that two MFMAs
    // into the same accumulator are never back to back (a wave has one accumulator per unit; a dependent MFMA would wait
    // for the pipeline to drain), and the 8-bit fragment of unit j is therefore read one step later than its fp16 fragments.
    xv_f16x8 Hh0[2], Hh1[2];
    xv_i32x8 Mx[2];
    Mx[1] = (xv_i32x8){0, 0, 0, 0, 0, 0, 0, 0};            // the "previous unit" of the very first step: adds zero
    const char *fbase = lds + hf * 16384 + lane * 16;       // this wave's half of a stage
    auto load_h = [&](auto SET, int slot, int unit) {
        constexpr int set = decltype(SET)::value;
        const char *b = fbase + slot + unit * 4096;
        Hh0[set] = *reinterpret_cast<const xv_f16x8 *>(b);
        Hh1[set] = *reinterpret_cast<const xv_f16x8 *>(b + 1024);
    };
    auto load_m = [&](auto SET, int slot, int unit) {
        constexpr int set = decltype(SET)::value;
        const char *b = fbase + slot + unit * 4096;
        Mx[set] = cat(*reinterpret_cast<const xv_i32x4 *>(b + 2048), *reinterpret_cast<const xv_i32x4 *>(b + 3072));
    };
    auto next_slot = [](int slot) { return slot + P8_STAGE == P8_RING * P8_STAGE ? 0 : slot + P8_STAGE; };
    // B(s), between steps 2 and 3 of stage s.  Before it this wave waits until (a) its fragment reads of stage s are complete
    // -- the fragments of the stage's last unit are in registers -- and (b) at most 4 of its vector-memory operations are in
    // flight.  The newest operations are the refill in progress (stage s+2: piece 0 issued behind B(s-1), pieces 1-3 in steps
    // 0-2 of this stage) and, between them, statistics stores; loads complete in order, so a piece of the refill before
    // (stage s+1) still in flight would put at least five operations in flight: (b) implies that stage s+1 and the frames issued
    // with it have landed (stores in the count only make the wait stricter).  After the barrier the same holds for every
    // wave, and the slot of stage s is refilled from here on.
    auto stage_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    // Issue order of a step: MFMA 1 | the ND fragment reads (ds_read_b128; they have the rest of the step to return) | MFMA 2
    // (the 64-cycle one) | the step's LDS-DMA pieces | MFMA 3 | pooling arithmetic.  Hard scheduling barriers between the parts:
    // the LDS-DMA pieces are not matched by the VMEM group masks.
    auto pin_a = [&](auto NDS) {
        constexpr int nd = decltype(NDS)::value;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (nd > 0) __builtin_amdgcn_sched_group_barrier(0x100, nd, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto pin_b = [&]() { __builtin_amdgcn_sched_barrier(0); };
    auto pin_c = [&]() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    int scale_a = XV_SPLIT8_E8M0, scale_b = 127;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));

    // ---- prologue --------------------------------------------------------------------------------------------------------------
    issue_x();
    issue_w(0);
    issue_w(P8_STAGE);
    fill = 2 * P8_STAGE;
    w_piece(I0{});                                             // (pieces 1-3 of stage 2 follow in steps 0-2 of stage 0)
    // ---- while the first stages are on their way: row validity of this lane's frames, first-layer parameters -> LDS --------
    // (lane (li, hh) pools rows 16hh .. 16hh+15 of the pair's 32 frames, see phase 2; the 16 validity bytes in ONE load --
    // sixteen dependent byte loads were most of this prologue)
    bool frame_ok = row0 + li < p.R;                           // the frame whose channels this lane converts
    uint32_t rows_mask = 0;
    {
        const long g0 = row0 + 16 * hh;
        uint8_t vb[16];
        if (!p.valid) {
#pragma unroll
            for (int k = 0; k < 16; ++k) vb[k] = 1;
        } else if (g0 + 16 <= p.R) {
            __builtin_memcpy(vb, p.valid + g0, 16);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) vb[k] = g0 + k < p.R ? p.valid[g0 + k] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (g0 + k < p.R && vb[k]) rows_mask |= 1u << k;
        if (p.valid && frame_ok) frame_ok = p.valid[row0 + li] != 0;
        float *P1 = reinterpret_cast<float *>(lds + P8_P1_OFF);
        for (int c = tid; c < CMID; c += P8_WAVES * 64) {
            P1[c] = p.b1 ? p.b1[c] : 0.f;
            P1[CMID + c] = p.sc1 ? p.sc1[c] : 1.f;
            P1[2 * CMID + c] = p.sh1 ? p.sh1[c] : 0.f;
            P1[3 * CMID + c] = p.act == XV_ACT_NONE ? 1.f : p.act == XV_ACT_LRELU ? p.al1[0] : p.act == XV_ACT_PRELU ? p.al1[c] : 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(5)" ::: "memory");           // frames of slab 0 and stage 0 (the loads above were behind them)
    __syncthreads();
    XFrag X;
    load_xfrag(X);
    load_h(I0{}, 0, 0);
    int slot = 0;

    // ---- phase 1: H^T[channel][frame] for this wave's 256 channels: 8 tiles of 32 channels x 32 frames ---------------------------
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f32x16){0};
    xv_i32x8 Xprev = X.x;                                      // the 8-bit frames fragment the pending MFMA of the previous slab needs
    auto f16a = [&](auto SET, auto T, const xv_f16x8 &xf, bool second) {   // A = weight fragment, B = frames fragment
        constexpr int set = decltype(SET)::value, t = decltype(T)::value;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(second ? Hh1[set] : Hh0[set], xf, acc[t], 0, 0, 0);
    };
    auto mxa = [&](auto SET, auto T, const xv_i32x8 &xf) {
        constexpr int set = decltype(SET)::value, t = decltype(T)::value;
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Mx[set], xf, acc[t], 1, 1, 0, scale_b, 0, scale_a);
    };
    for (int ks = 0; ks < p.n_ks; ++ks) {
        auto stage1 = [&](auto Q) {                        // stage (ks, q): tiles 4q .. 4q+3, one per step
            constexpr int q = decltype(Q)::value;
            typedef std::integral_constant<int, 4 * q> T0;
            typedef std::integral_constant<int, 4 * q + 1> T1;
            typedef std::integral_constant<int, 4 * q + 2> T2;
            typedef std::integral_constant<int, 4 * q + 3> T3;
            typedef std::integral_constant<int, q == 0 ? 7 : 3> TP;          // tile of the pending 8-bit MFMA
            typedef std::integral_constant<int, 4> N4;
            typedef std::integral_constant<int, 6> N6;
            // step 0
            load_h(I1{}, slot, 1);
            load_m(I0{}, slot, 0);
            f16a(I0{}, T0{}, X.h0, false);
            mxa(I1{}, TP{}, q == 0 ? Xprev : X.x);
            pin_a(N4{});
            w_piece(I1{});
            pin_b();
            f16a(I0{}, T0{}, X.h1, true);
            pin_c();
            // step 1
            load_h(I0{}, slot, 2);
            load_m(I1{}, slot, 1);
            f16a(I1{}, T1{}, X.h0, false);
            mxa(I0{}, T0{}, X.x);
            pin_a(N4{});
            w_piece(I2{});
            pin_b();
            f16a(I1{}, T1{}, X.h1, true);
            pin_c();
            // step 2 (also the 8-bit fragment of unit 3: everything of this slot is read before the barrier)
            load_h(I1{}, slot, 3);
            load_m(I0{}, slot, 2);
            f16a(I0{}, T2{}, X.h0, false);
            mxa(I1{}, T1{}, X.x);
            load_m(I1{}, slot, 3);      // (a new value: the read itself is scheduled with the others, behind the first MFMA)
            pin_a(N6{});
            w_piece(I3{});
            pin_b();
            f16a(I0{}, T2{}, X.h1, true);
            pin_c();
            stage_barrier();
            fill = slot;
            slot = next_slot(slot);
            // step 3 (the fp16 fragments of the next stage's first unit come from the next ring slot)
            load_h(I0{}, slot, 0);
            if constexpr (q == 1) {
                XFrag N;
                load_xfrag(N);                             // frames of slab ks+1: issued two barriers ago
                f16a(I1{}, T3{}, X.h0, false);
                mxa(I0{}, T2{}, X.x);
                pin_a(N6{});
                w_piece(I0{});
                pin_b();
                f16a(I1{}, T3{}, X.h1, true);
                pin_c();
                Xprev = X.x;
                X = N;
            } else {
                f16a(I1{}, T3{}, X.h0, false);
                mxa(I0{}, T2{}, X.x);
                pin_a(I2{});
                issue_x();                                 // the frames of slab ks+1 (everybody holds slab ks in registers)
                w_piece(I0{});
                pin_b();
                f16a(I1{}, T3{}, X.h1, true);
                pin_c();
            }
        };
        stage1(I0{});
        stage1(I1{});
    }
    mxa(I1{}, std::integral_constant<int, 7>{}, Xprev);     // the pending 8-bit MFMA of the last unit

    // ---- accumulators -> A operands of the second GEMM: bias, activation, BN, split8 encoding ----------------------------------------
    xv_f16x8 Hf0[8], Hf1[8];
    xv_i32x8 Hx[8];
    float amax = 0.f;
    {
        auto conv = [&](auto U) {
            constexpr int u = decltype(U)::value;
            // tile u's parameter reads must not start before tile u-1 is converted (the optimiser would hoist all of them and
            // spill): an opaque zero that depends on the previous result
            int dep = 0;
            if constexpr (u > 0) {
                const xv_i32x8 a = Hx[u - 1];
                asm volatile("" : "+v"(dep) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
            }
            const f32x4 *P1 = reinterpret_cast<const f32x4 *>(lds + P8_P1_OFF + dep);
            const f32x16 t = acc[u];
            xv_i32x4 x8[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {                                  // r = 8g .. 8g+7: channels (r&3) + 8(r>>2) + 4hh
                float v[8];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c4 = (256 * hf + 32 * u + 8 * (2 * g + j) + 4 * hh) >> 2;
                    const f32x4 b = P1[c4], s = P1[CMID / 4 + c4], o = P1[2 * CMID / 4 + c4], a = P1[3 * CMID / 4 + c4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * j + e] = act_fn<MODE>(t[8 * g + 4 * j + e] + b[e], a[e]) * s[e] + o[e];
                }
                xv_f16x8 hi;
                xv_split8_encode8<true>(v, hi, x8[g], amax);
                if (g == 0) Hf0[u] = hi;
                else Hf1[u] = hi;
            }
            Hx[u] = cat(x8[0], x8[1]);
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, 8>(conv);
    }
    // (a lane holds ONE frame's channels: frames past R or in a gap may hold anything -- stale bytes of a recycled buffer --
    // and their H never reaches an output, so they must not raise the flag either)
    if (amax > XV_SPLIT8_MAX && frame_ok && p.status) atomicOr(p.status, 1);
    load_h(I0{}, slot, 0);        // (again: the copy read inside the loop is dropped so that the conversion has the registers)

    // ---- phase 2: partial Y[frame][column] over this wave's 256 channels, 64 columns at a time -----------------------------------------
    // Of a 32-frame x 64-column tile wave hf FINISHES the 32 columns 32 hf .. 32 hf + 31 (its KEEP half) and contributes a partial sum to
    // the partner's 32 (its RED half).  The two halves are not accumulated side by side but one after the other: stages 0-1 of a column
    // tile run the wave's eight slabs against the RED half's weights, stages 2-3 against the KEEP half's (the packed stream of a half is
    // ordered that way, pack_pair8_kernel).  So
    //   * the RED half is complete in the MIDDLE of a tile: it goes to LDS in the first two steps of stage 2 (sixteen ds_write_b32,
    //     element (register r, lane l) at float r*64 + l; two buffers, by the parity of the tile) and is visible to the partner from the
    //     tile's last barrier on;
    //   * the KEEP half is complete at the END of a tile and never leaves the registers: eight v_permlane32_swap turn "lane (column, hh)
    //     holds rows (r&3) + 8(r>>2) + 4hh" into "lane (column, hh) holds rows 16hh .. 16hh+15" -- the two 8-row blocks 2hh, 2hh+1 of its
    //     column COMPLETELY -- and the lane pools them during stages 0-1 of the NEXT tile, two rows per step (one ds_read_b32 of the
    //     partner's partial sum and ~10 VALU instructions per row, in the shadow of the step's MFMAs), while those stages accumulate
    //     into the RED half's registers; at stage 2 the KEEP registers are free again.  No extra accumulator registers, a quarter
    //     of the earlier form's exchange traffic (it wrote BOTH halves to LDS and read both back).
    // row 16hh + k of the half-tile = accumulator register 8hh + 4(k>>3) + (k&3) of lane li + 32((k>>2)&1)
    const float *pool_red = reinterpret_cast<const float *>(lds + P8_RED_OFF + (wave ^ 1) * 4096) + 8 * hh * 64 + li;
    constexpr int RED_BUF_FLOATS = P8_WAVES * 1024;        // the second buffer lies over the dead first-layer parameters
    const bool lrelu = p.act == XV_ACT_LRELU, prelu = p.act == XV_ACT_PRELU;
    // epilogue parameters of the column this lane pools, fetched one column tile ahead (a global load at the point of use
    // would stall the wave for a microsecond per column tile)
    f32x4 prm = {0.f, 1.f, 0.f, 0.f}, prm_next = prm;      // {bias, scale, shift, alpha}
    auto fetch_params = [&](int ct) {
        const int col = ct * 64 + 32 * hf + li;
        f32x4 v;
        v[0] = p.b2 ? p.b2[col] : 0.f;
        v[1] = p.sc2 ? p.sc2[col] : 1.f;
        v[2] = p.sh2 ? p.sh2[col] : 0.f;
        v[3] = p.act == XV_ACT_NONE ? 1.f : lrelu ? p.al2[0] : prelu ? p.al2[col] : 0.f;
        return v;
    };
    // block statistics of this workgroup: blocks row0_wg/8 .. +15, [block][mean | M2][cout] floats
    const int blk_row_bytes = 2 * p.cout * 4;
    const long blk0 = (long)blockIdx.x * (P8_ROWS / 8);
    const long blk_left = p.n_blocks - blk0;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
        p.blk + (size_t)blk0 * 2 * p.cout, 0, (int)(blk_left <= 0 ? 0 : (blk_left < P8_ROWS / 8 ? blk_left : P8_ROWS / 8) * blk_row_bytes), XV_RSRC_FLAGS);
    const int blk_voff = (4 * pair + 2 * hh) * blk_row_bytes + (32 * hf + li) * 4;
    f32x16 yR, yK;                                         // the two halves of the partial tile, by ROLE (column half: RED 1-hf, KEEP hf)
    yK = (f32x16){0};                                      // ("the tile before the first": pooled like any other, its stores are dropped)
    const float *pool_cur = pool_red;                      // the partner's RED half of the tile being pooled
    // two rows of it per step, read ONE STEP AHEAD (pn) of the step that pools them (pb): the wait in front of the pooling
    // arithmetic then is for the previous step's reads, not a drain of the fragment reads just issued
    float pb[2] = {0.f, 0.f}, pn[2] = {0.f, 0.f};
    auto pool_read = [&](auto K, auto I, const float *from) {
        constexpr int k = decltype(K)::value, i = decltype(I)::value;
        constexpr int off = (4 * (k >> 3) + (k & 3)) * 64 + 32 * ((k >> 2) & 1);
        pn[i] = from[off];
    };
    // Row k of column tile ct.  Statistics are taken of the ACTIVATION, shifted by the block's first row; the BatchNorm that
    // follows it is affine per column and is applied to the block's (mean, M2) when they are stored -- mean' = scale * mean +
    // shift, M2' = scale^2 * M2 -- instead of to every element.  (The two-wide form of this body, v_pk_add_f32 / v_pk_fma_f32 on
    // row pairs, was measured 1 % SLOWER than one row at a time.)
    float pv0 = 0.f, ps1 = 0.f, ps2 = 0.f;                 // running statistics of the block being pooled
    auto pool_row = [&](auto K, auto I, int ct) {
        constexpr int k = decltype(K)::value, i = decltype(I)::value;
        // after the swap: row k sits in register (k&3) + 4(k>>3), + 8 for the rows whose source lane was in the upper half
        constexpr int reg = (k & 3) + 4 * (k >> 3) + 8 * ((k >> 2) & 1);
        // Every product-sum below is an EXPLICIT fma: left to the optimiser, the sixteen instances of this body contract (or pack
        // into v_pk_mul / v_pk_add) differently, and a block's statistics would depend on whether it sits at an even or an odd
        // 8-row position -- an utterance's x-vector must not depend on where in the batch it lies.
        float v = act_fn<MODE>(yK[reg] + pb[i] + prm[0], prm[3]);
        asm volatile("" : "+v"(v));              // (computed for every lane: masked rows must not turn into a branch around the reads)
        if constexpr ((k & 7) == 0) {
            pv0 = v;
            ps1 = 0.f;
            ps2 = 0.f;
        } else {
            const float d = ((rows_mask >> k) & 1u) ? v - pv0 : 0.f;
            ps1 += d;
            ps2 = __builtin_fmaf(d, d, ps2);
        }
        if constexpr ((k & 7) == 7) {
            const float n = (float)__builtin_popcount((rows_mask >> (k - 7)) & 255u);
            const float rn = n > 0.f ? 1.f / n : 0.f;
            const float mean_a = __builtin_fmaf(ps1, rn, pv0);
            float sq = ps1 * ps1;
            asm volatile("" : "+v"(sq));
            const float m2_a = fmaxf(__builtin_fmaf(-sq, rn, ps2), 0.f);
            const float mean = n > 0.f ? __builtin_fmaf(mean_a, prm[1], prm[2]) : 0.f;
            float s2 = prm[1] * prm[1];
            asm volatile("" : "+v"(s2));
            const float m2 = m2_a * s2;
            // (blocks past n_blocks and the non-existent column tile -1 fall outside the descriptor and are dropped by the
            // range check: no branch that would split the step's scheduling region)
            const int o = ct >= 0 ? blk_voff + (k >> 3) * blk_row_bytes + ct * 256 : -1;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mean), brs, o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, m2), brs, o, p.cout * 4, 0);
        }
    };
    // Inline asm on purpose: this compiler's __builtin_amdgcn_permlane32_swap hands back the FIRST result for both halves of its
    // pair when both are used (seen in the ISA: sixteen rows pooled from eight registers).  The hazard recogniser does not look
    // into asm, so the distance to the MFMA that wrote the KEEP half is kept by construction: the swaps sit behind the first
    // MFMA of the next tile and its LDS-DMA piece (>= 100 cycles after the 64-cycle MFMA in question has left the pipe), the
    // tail waits explicitly.
    auto swap_keep = [&](auto TAIL) {                      // rows (r&3) + 8(r>>2) + 4hh  ->  rows 16hh .. 16hh+15 (see pool_row)
        if constexpr (decltype(TAIL)::value) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        else asm volatile("s_nop 7");
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float a = yK[r], b = yK[r + 8];
            asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
            yK[r] = a;
            yK[r + 8] = b;
        }
    };
    // Rows of the PREVIOUS column tile pooled in step j of stage q: 2 in every step of stages 0 and 1 (rows 8q + 2j, + 1).  The
    // partner wrote them in stage 2 of that tile (before B(., 2)); the buffer is written again two tiles later.
    // (step (3, 3) reads the first two rows of the tile that is just being finished: the partner wrote them before B(., 2))
    constexpr auto pool_ahead = [](int q, int j) { return q == 0 || (q == 1 && j < 3) || (q == 3 && j == 3); };
    auto pool_load = [&](auto Q, auto J, int ct) {
        constexpr int q = decltype(Q)::value, j = decltype(J)::value;
        if constexpr (q < 2) {
            pb[0] = pn[0];
            pb[1] = pn[1];
        }
        if constexpr (q == 3 && j == 3) {
            const float *nxt = pool_red + (ct & 1) * RED_BUF_FLOATS;
            pool_read(I0{}, I0{}, nxt);
            pool_read(I1{}, I1{}, nxt);
        } else if constexpr (pool_ahead(q, j)) {
            pool_read(std::integral_constant<int, 8 * q + 2 * j + 2>{}, I0{}, pool_cur);
            pool_read(std::integral_constant<int, 8 * q + 2 * j + 3>{}, I1{}, pool_cur);
        }
    };
    auto pool_step = [&](auto Q, auto J, int ct) {
        constexpr int q = decltype(Q)::value, j = decltype(J)::value;
        // (column tile -1 does not exist: its rows are read from whatever the buffers hold and never stored -- no branch
        // that would split the scheduling region of the step)
        if constexpr (q < 2) {
            pool_row(std::integral_constant<int, 8 * q + 2 * j>{}, I0{}, ct - 1);
            pool_row(std::integral_constant<int, 8 * q + 2 * j + 1>{}, I1{}, ct - 1);
        }
    };
    // Registers 8 half .. 8 half + 7 of the RED half -> LDS, element (register r, lane l) at float r*64 + l of this wave's slot:
    // ds_write_addtid_b32 (address = M0 + offset + 4 * lane: no address register, 2 LDS cycles per instruction where
    // ds_write_b32 takes 4).  M0 also carries the LDS address of the compiler's LDS-DMA instructions: saved and restored here.
    const unsigned red_base = (unsigned)(size_t)(lds + P8_RED_OFF + wave * 4096);
    auto red_write = [&](auto HALF, int ct) {
        constexpr int h = decltype(HALF)::value;
        const unsigned base = __builtin_amdgcn_readfirstlane(red_base + (unsigned)(ct & 1) * (RED_BUF_FLOATS * 4) + h * 2048);
        unsigned keep_m0;
        const f32x16 &y = yR;
        const float r0 = y[8 * h], r1 = y[8 * h + 1], r2 = y[8 * h + 2], r3 = y[8 * h + 3], r4 = y[8 * h + 4], r5 = y[8 * h + 5],
                    r6 = y[8 * h + 6], r7 = y[8 * h + 7];
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\t"
                     "ds_write_addtid_b32 %1 offset:0\n\tds_write_addtid_b32 %2 offset:256\n\t"
                     "ds_write_addtid_b32 %3 offset:512\n\tds_write_addtid_b32 %4 offset:768\n\t"
                     "ds_write_addtid_b32 %5 offset:1024\n\tds_write_addtid_b32 %6 offset:1280\n\t"
                     "ds_write_addtid_b32 %7 offset:1536\n\tds_write_addtid_b32 %8 offset:1792\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0)
                     : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(r4), "v"(r5), "v"(r6), "v"(r7), "s"(base)
                     : "memory");
    };
    prm_next = fetch_params(0);
    for (int ct = 0; ct < p.n_ct; ++ct) {
        auto f16b = [&](auto SET, auto U, auto ROLE, auto FIRST, bool second) {     // A = H fragment (registers), B = weight fragment
            constexpr int set = decltype(SET)::value, u = decltype(U)::value, role = decltype(ROLE)::value;
            constexpr bool first = decltype(FIRST)::value;
            const xv_f16x8 a = second ? Hf1[u] : Hf0[u], b = second ? Hh1[set] : Hh0[set];
            if constexpr (role == 0) yR = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, first ? (f32x16){0} : yR, 0, 0, 0);
            else yK = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, first ? (f32x16){0} : yK, 0, 0, 0);
        };
        auto mxb = [&](auto SET, auto U, auto ROLE) {
            constexpr int set = decltype(SET)::value, u = decltype(U)::value, role = decltype(ROLE)::value;
            if constexpr (role == 0) yR = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Hx[u], Mx[set], yR, 1, 1, 0, scale_a, 0, scale_b);
            else yK = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(Hx[u], Mx[set], yK, 1, 1, 0, scale_a, 0, scale_b);
        };
        auto quarter = [&](auto Q) {                        // stage (ct, q): slabs 4(q&1) .. 4(q&1)+3 of half-tile q >> 1 (0: RED, 1: KEEP)
            constexpr int q = decltype(Q)::value;
            typedef std::integral_constant<int, (q >> 1)> RL;
            typedef std::integral_constant<int, 4 * (q & 1)> U0;
            typedef std::integral_constant<int, 4 * (q & 1) + 1> U1;
            typedef std::integral_constant<int, 4 * (q & 1) + 2> U2;
            typedef std::integral_constant<int, 4 * (q & 1) + 3> U3;
            typedef std::integral_constant<int, 3> UP;        // slab of the 8-bit MFMA pending from the stage before (q odd only)
            typedef std::integral_constant<bool, (q & 1) == 0> FIRST;
            typedef std::integral_constant<bool, false> NO;
            if constexpr (q == 0) {
                prm = prm_next;                             // the parameters of column tile ct-1, fetched a whole tile ago
                prm_next = fetch_params(ct);
                pool_cur = pool_red + ((ct + 1) & 1) * RED_BUF_FLOATS;
            }
            // step 0: slab 4(q&1)
            load_h(I1{}, slot, 1);
            load_m(I0{}, slot, 0);
            f16b(I0{}, U0{}, RL{}, FIRST{}, false);
            if constexpr ((q & 1) != 0) mxb(I1{}, UP{}, RL{});
            pool_load(Q, I0{}, ct);
            if constexpr ((q & 1) != 0) pin_a(std::integral_constant<int, 4 + (pool_ahead(q, 0) ? 1 : 0)>{});
            else { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 4 + (pool_ahead(q, 0) ? 1 : 0), 0); __builtin_amdgcn_sched_barrier(0); }
            w_piece(I1{});
            pin_b();
            if constexpr (q == 0) swap_keep(NO{});          // (the KEEP half of tile ct-1, complete since the end of its stage 3)
            f16b(I0{}, U0{}, RL{}, NO{}, true);
            pool_step(Q, I0{}, ct);
            if constexpr (q == 2) red_write(I0{}, ct);      // the RED half of this tile (complete since the end of stage 1) -> LDS
            pin_c();
            // step 1
            load_h(I0{}, slot, 2);
            load_m(I1{}, slot, 1);
            f16b(I1{}, U1{}, RL{}, NO{}, false);
            mxb(I0{}, U0{}, RL{});
            pool_load(Q, I1{}, ct);
            pin_a(std::integral_constant<int, 4 + (pool_ahead(q, 1) ? 1 : 0)>{});
            w_piece(I2{});
            pin_b();
            f16b(I1{}, U1{}, RL{}, NO{}, true);
            pool_step(Q, I1{}, ct);
            if constexpr (q == 2) red_write(I1{}, ct);
            pin_c();
            // step 2; also the 8-bit fragment of unit 3
            load_h(I1{}, slot, 3);
            load_m(I0{}, slot, 2);
            f16b(I0{}, U2{}, RL{}, NO{}, false);
            mxb(I1{}, U1{}, RL{});
            load_m(I1{}, slot, 3);
            pool_load(Q, I2{}, ct);
            pin_a(std::integral_constant<int, 6 + (pool_ahead(q, 2) ? 1 : 0)>{});
            w_piece(I3{});
            pin_b();
            f16b(I0{}, U2{}, RL{}, NO{}, true);
            pool_step(Q, I2{}, ct);
            pin_c();
            stage_barrier();
            fill = slot;
            slot = next_slot(slot);
            // step 3
            load_h(I0{}, slot, 0);
            f16b(I1{}, U3{}, RL{}, NO{}, false);
            mxb(I0{}, U2{}, RL{});
            pool_load(Q, I3{}, ct);
            pin_a(std::integral_constant<int, 2 + (pool_ahead(q, 3) ? 1 : 0)>{});
            w_piece(I0{});
            pin_b();
            f16b(I1{}, U3{}, RL{}, NO{}, true);
            pool_step(Q, I3{}, ct);
            pin_c();
            if constexpr ((q & 1) != 0) mxb(I1{}, U3{}, RL{});   // the pending 8-bit MFMA of the half-tile's last slab
        };
        static_for<0, 4>(quarter);
    }
    // the last column tile: its RED halves were written before B(last, 2); pool it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    prm = prm_next;
    pool_cur = pool_red + ((p.n_ct + 1) & 1) * RED_BUF_FLOATS;
    swap_keep(std::integral_constant<bool, true>{});
    {
        auto tail = [&](auto H) {
            constexpr int k = 2 * decltype(H)::value;
            pool_read(std::integral_constant<int, k>{}, I0{}, pool_cur);
            pool_read(std::integral_constant<int, k + 1>{}, I1{}, pool_cur);
            pb[0] = pn[0];
            pb[1] = pn[1];
            pool_row(std::integral_constant<int, k>{}, I0{}, p.n_ct - 1);
            pool_row(std::integral_constant<int, k + 1>{}, I1{}, p.n_ct - 1);
        };
        static_for<0, 8>(tail);
    }
}

// w1[cin][CMID], w2[CMID][cout] (fp32, TF's [in, out] order) -> stages in consumption order (see the kernel).  One thread per
// (stage, half, unit, lane): its three 16/32-byte fragments.
//   layer 1, stage 2 ks + q, half h, unit j (tile T = 4q + j): output channel o = 256 h + 32 T + (lane & 31), kh = lane >> 5
//       fp16 k-step s: w1[32 ks + 16 s + 8 kh + e][o];  8-bit: channels 32 ks + 16 kh + {0..7 | 8..15}
//   layer 2, stage 4 ct + q, half h, unit j (slab u = 4 (q & 1) + j, column tile c = 1 - h for q < 2, h for q >= 2): column n = 64 ct + 32 c + (lane & 31)
//       channel of (r, kh) = 256 h + 32 u + (r&3) + 8 (r>>2) + 4 kh;  fp16 k-step s: r = 8 s + e;  8-bit: r = {0..7 | 8..15}
__global__ void pack_pair8_kernel(const float *__restrict__ w1, const float *__restrict__ w2, int n_ks, int cout, int n_ct,
                                  uint8_t *__restrict__ wt, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int lane = (int)(i & 63);
    const int unit = (int)((i >> 6) & 3);
    const int h = (int)((i >> 8) & 1);
    const long stage = (long)(i >> 9);
    const int n = lane & 31, kh = lane >> 5;
    float v[16];
    if (stage < 2L * n_ks) {
        const int ks = (int)(stage >> 1), q = (int)(stage & 1);
        const int o = 256 * h + 32 * (4 * q + unit) + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = w1[(size_t)(32 * ks + 16 * (r >> 3) + 8 * kh + (r & 7)) * CMID + o];
        // (the 8-bit fragment wants channels 16 kh + 0..15 = the same 16 values in another order: see below)
        float w8[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) w8[r] = w1[(size_t)(32 * ks + 16 * kh + r) * CMID + o];
        xv_f16x8 h0, h1, dummy;
        xv_i32x4 x0, x1, xd;
        float amax = 0.f;
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = v[e]; b[e] = v[8 + e]; }
        xv_split8_encode8<false>(a, h0, xd, amax);
        xv_split8_encode8<false>(b, h1, xd, amax);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = w8[e]; b[e] = w8[8 + e]; }
        xv_split8_encode8<false>(a, dummy, x0, amax);
        xv_split8_encode8<false>(b, dummy, x1, amax);
        uint8_t *t = wt + (size_t)stage * P8_STAGE + h * 16384 + unit * 4096 + lane * 16;
        *reinterpret_cast<xv_f16x8 *>(t) = h0;
        *reinterpret_cast<xv_f16x8 *>(t + 1024) = h1;
        *reinterpret_cast<xv_i32x4 *>(t + 2048) = x0;
        *reinterpret_cast<xv_i32x4 *>(t + 3072) = x1;
    } else {
        const long s2 = stage - 2L * n_ks;
        const int ct = (int)(s2 >> 2), q = (int)(s2 & 3);
        const int u = 4 * (q & 1) + unit, c = (q >> 1) ? h : 1 - h;   // stages 0-1: the half the PARTNER finishes, 2-3: this wave's own
        const int col = 64 * ct + 32 * c + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = w2[(size_t)(256 * h + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * kh) * cout + col];
        xv_f16x8 h0, h1;
        xv_i32x4 x0, x1;
        float amax = 0.f;
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = v[e]; b[e] = v[8 + e]; }
        xv_split8_encode8<false>(a, h0, x0, amax);
        xv_split8_encode8<false>(b, h1, x1, amax);
        uint8_t *t = wt + (size_t)stage * P8_STAGE + h * 16384 + unit * 4096 + lane * 16;
        *reinterpret_cast<xv_f16x8 *>(t) = h0;
        *reinterpret_cast<xv_f16x8 *>(t + 1024) = h1;
        *reinterpret_cast<xv_i32x4 *>(t + 2048) = x0;
        *reinterpret_cast<xv_i32x4 *>(t + 3072) = x1;
    }
}

bool pair8_shape_ok(int cin, int cmid, int cout) { return cmid == CMID && cin > 0 && (cin & 31) == 0 && cout > 0 && (cout & 63) == 0 && cout <= 4096; }

}  // namespace

extern "C" {

size_t xv_packed_pair_f16bf8_bytes(int cin, int cmid, int cout)
{
    if (!pair8_shape_ok(cin, cmid, cout)) return 0;
    return (size_t)(2 * (cin / 32) + 4 * (cout / 64)) * P8_STAGE;
}

int xv_pack_pair_f16bf8(const float *w1, const float *w2, int cin, int cmid, int cout, void *wt, void *stream)
{
    if (!w1 || !w2 || !wt) return fail(XV_ERR_BAD_ARG, "pack_pair_f16bf8: NULL pointer");
    if (!pair8_shape_ok(cin, cmid, cout))
        return fail(XV_ERR_UNSUPPORTED, "pack_pair_f16bf8: needs cmid == 512, cin % 32 == 0, cout % 64 == 0, cout <= 4096");
    if (((uintptr_t)wt) & 15) return fail(XV_ERR_BAD_ARG, "pack_pair_f16bf8: wt must be 16-byte aligned");
    const size_t total = xv_packed_pair_f16bf8_bytes(cin, cmid, cout) / 64;      // one thread per 64 packed bytes
    hipLaunchKernelGGL(pack_pair8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w1, w2, cin / 32,
                       cout, cout / 64, (uint8_t *)wt, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

int xv_tdnn_pair_pool_f16bf8(const void *x, int64_t R, int cin, int cmid, int cout, const void *wt, const float *bias1,
                             const float *bn_scale1, const float *bn_shift1, const float *act_alpha1, const float *bias2,
                             const float *bn_scale2, const float *bn_shift2, const float *act_alpha2, int act_kind,
                             const uint8_t *row_valid, float *block_stats, int32_t *status, void *stream)
{
    if (R <= 0) return 0;
    if (!x || !wt || !block_stats) return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_f16bf8: NULL pointer");
    if (act_kind < XV_ACT_NONE || act_kind > XV_ACT_PRELU) return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_f16bf8: unknown act_kind");
    if ((act_kind == XV_ACT_LRELU || act_kind == XV_ACT_PRELU) && (!act_alpha1 || !act_alpha2))
        return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_f16bf8: act_alpha is NULL");
    if (!pair8_shape_ok(cin, cmid, cout))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_pair_pool_f16bf8: needs cmid == 512, cin % 32 == 0, cout % 64 == 0, cout <= 4096");
    if ((((uintptr_t)x) | ((uintptr_t)wt) | ((uintptr_t)block_stats)) & 15)
        return fail(XV_ERR_BAD_ARG, "tdnn_pair_pool_f16bf8: x, wt and block_stats must be 16-byte aligned");
    Pair8Params p{};
    p.x = (const uint8_t *)x; p.R = (long)R; p.n_ks = cin / 32; p.cout = cout; p.n_ct = cout / 64; p.wt = (const uint8_t *)wt;
    p.b1 = bias1; p.sc1 = bn_scale1; p.sh1 = bn_shift1; p.al1 = act_alpha1;
    p.b2 = bias2; p.sc2 = bn_scale2; p.sh2 = bn_shift2; p.al2 = act_alpha2;
    p.act = act_kind; p.valid = row_valid; p.blk = block_stats; p.n_blocks = (long)((R + 7) / 8); p.status = (int *)status;
    typedef void (*kern_t)(const Pair8Params);
    const kern_t kerns[3] = {tdnn_pair_pool_f16bf8_kernel<0>, tdnn_pair_pool_f16bf8_kernel<1>, tdnn_pair_pool_f16bf8_kernel<2>};
    static std::atomic<unsigned long long> attr_done{0};      // dynamic-LDS opt-in: per device, idempotent
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (kern_t k : kerns) {
            hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P8_LDS_BYTES);
            if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int mode = act_kind == XV_ACT_LRELU ? 1 : act_kind == XV_ACT_RELU ? 2 : 0;
    hipLaunchKernelGGL(kerns[mode], dim3((unsigned)((R + P8_ROWS - 1) / P8_ROWS)), dim3(P8_WAVES * 64), P8_LDS_BYTES,
                       (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, hipGetErrorString(e));
    return 0;
}

}  // extern "C"
