// xv_toom.hip -- the wide-context frame-level layers (tf.nn.conv1d 'SAME' + bias + activation + BN-eval, local/tf/models.py:54-67,
// kernel sizes 5 and 7 of models.py:28) with FEWER MULTIPLICATIONS than the K-tap contraction states: Toom-Cook F(2, K) over time.
//
// Two consecutive output rows (2P, 2P+1) of a K-tap correlation read K + 1 input rows d_0 .. d_K = x[2P - (K-1)/2 ...]; instead of
// 2 K row-by-matrix products they are
//        out_q = sum_j AT[q][j] * ( V_j . U_j ),      V_j = sum_i BT[j][i] d_i     (K + 1 transformed rows, formed here on the fly)
//                                                    U_j = sum_k G[j][k]  w[k]    (K + 1 transformed taps, formed once at load)
// i.e. K + 1 products per row PAIR: 6 instead of 10 (K = 5), 8 instead of 14 (K = 7) -- 0.60 / 0.57 of the MFMA work of the direct
// form (tdnn_gemm_dma_kernel in xv_kernels.hip), exact fp32 products and fp32 accumulation as there.  Evaluation points 0, +-1, +-2
// (, +-1/2), infinity; matrices generated and checked in exact rationals by tools/experiments/toomcook_gen.py.  The result is NOT
// bit-identical to the direct form (another rounding; measured 5-8e-7 relative L2 per layer against fp64 where the direct form's
// 2560 / 3584-term fp32 chain has 9e-7 - 1.1e-6, 8.3e-7 against 8.6e-7 on the x-vector, and the smaller error on every one of 22
// trained / trained-like / hostile checkpoints, profiles/r05_fp32tc_accuracy_sweep.txt), which is why this is a separate arithmetic
// ("fp32tc") with its own entry point.
//
// Kernel = the DMA-fed fp32 GEMM's design with the taps replaced by the transformed products:
//  * tile 128 output rows (64 row pairs) x 128 columns, 4 waves (2 x 2), a wave owns 32 pairs x 64 columns;
//  * the halo tile of a 32-channel slab (128 + K - 1 rows) goes global -> LDS by buffer_load ... lds, EVEN and ODD rows into two
//    regions so that the 32 lanes of a fragment read (rows 2P + i for consecutive P) walk consecutive 128-byte LDS rows: the
//    conflict-free XOR-swizzled pattern of the direct kernel;
//  * stage = (slab, product): the wave reads the <= 6 raw fragments the product's row of BT needs, forms V_j with 2 .. 5 packed
//    fp32 operations per channel pair and accumulates V_j . U_j over the slab;
//  * one barrier per stage, weights double-buffered, the next slab's halo pieces dealt over the first stages of the current one.
// What shapes the inner loop is a measurement (tools/experiments/f32_mfma_valu_probe.hip): beside v_mfma_f32_32x32x2_f32 a VALU
// instruction of ANY kind is not hidden -- it costs ~4 of the MFMA pipe's cycles, plus ~10 more when it sits alone between two
// MFMAs (ds_read / s_nop are free) -- so the kernel counts VALU instructions:
//  * the products of the points 0 and infinity reach ONE output row each (AT has a single non-zero in their columns) and accumulate
//    straight into that row's tile; the others go to a temporary tile pair (two pairs, alternating by stage) that is folded into
//    both output rows with packed adds / FMAs under the MFMAs of the NEXT stage; the last temporary of a slab is folded in two
//    halves, each under the direct stage that does not own the half's output row (no MFMA ever waits for a fold or vice versa);
//  * rows of BT are scaled (the inverse goes into G) so that every transform is a short chain of packed FMAs without a leading
//    multiplication, the stage order puts the weight-tile parity and the halo-buffer parity into immediates (the only address
//    arithmetic left is one XOR per raw fragment), and every step is VALU first, then the next step's ds_reads, then 8 MFMAs
//    back to back.
// Row pairs are aligned to EVEN global rows; chunks start on even rows (the host lays batches out with align 8), so a chunk's bits
// do not depend on its batch neighbours.  Gap rows are zero and masked as everywhere else.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <type_traits>

#include "xvector_hip.h"

extern "C" void xv_internal_set_error(const char *msg);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// ---- F(2, K) matrices (tools/experiments/toomcook_gen.py prints BT / G / AT and checks them in exact rationals).  Here per
// product j: the scale SC[j] its row of BT is divided by (G's row is multiplied by it), AT[1][j] (AT[0][j] is 1 except for the
// point at infinity), and the stage order: the product of point 0 (output row 2P only), the product of infinity (row 2P + 1
// only), then the +- pairs.  The transforms themselves are written out in xform() below.
template <int KT>
struct Toom;

template <>
struct Toom<3> {                                  // points 0, 1, -1, inf: Winograd's F(2, 3)
    static constexpr int J = 4;
    // BT = {{1,0,-1,0}, {0,1,1,0}, {0,-1,1,0}, {0,-1,0,1}} (the last row negated against the textbook form: AT[1][3] = +1)
    static constexpr int ORDER[4] = {0, 3, 1, 2};
    static constexpr double SC[4] = {1, 1, 1, 1};
    static constexpr float A1[4] = {0, 1, -1, 1};
    static constexpr unsigned NEED[4] = {0x5, 0x6, 0x6, 0xa};
    static constexpr double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
};

template <>
struct Toom<5> {                                  // points 0, 1, -1, 2, -2, inf
    static constexpr int J = 6;
    // BT = {{4,0,-5,0,1,0}, {0,-4,-4,1,1,0}, {0,4,-4,-1,1,0}, {0,-2,-1,2,1,0}, {0,2,-1,-2,1,0}, {0,4,0,-5,0,1}}
    static constexpr int ORDER[6] = {0, 5, 1, 2, 3, 4};
    static constexpr double SC[6] = {1, 1, 1, 1, 1, 1};
    static constexpr float A1[6] = {0, 1, -1, 2, -2, 1};
    static constexpr unsigned NEED[6] = {0x15, 0x1e, 0x1e, 0x1e, 0x1e, 0x2a};     // bit i: the product reads input row i
    static constexpr double G[6][5] = {{1. / 4, 0, 0, 0, 0},
                                       {-1. / 6, -1. / 6, -1. / 6, -1. / 6, -1. / 6},
                                       {-1. / 6, 1. / 6, -1. / 6, 1. / 6, -1. / 6},
                                       {1. / 24, 1. / 12, 1. / 6, 1. / 3, 2. / 3},
                                       {1. / 24, -1. / 12, 1. / 6, -1. / 3, 2. / 3},
                                       {0, 0, 0, 0, 1}};
};

template <>
struct Toom<7> {                                  // points 0, 1, -1, 2, -2, 1/2, -1/2, inf
    static constexpr int J = 8;
    // BT = {{-4,0,21,0,-21,0,4,0}, {0,4,4,-17,-17,4,4,0}, {0,-4,4,17,-17,-4,4,0}, {0,2,1,-10,-5,8,4,0}, {0,-2,1,10,-5,-8,4,0},
    //       {0,4,8,-5,-10,1,2,0}, {0,-4,8,5,-10,-1,2,0}, {0,-4,0,21,0,-21,0,4}}; rows 0, 1, 2, 7 are used divided by 4, row 6 negated
    static constexpr int ORDER[8] = {0, 7, 1, 2, 3, 4, 5, 6};
    static constexpr double SC[8] = {4, 4, 4, 1, 1, 1, -1, 4};
    static constexpr float A1[8] = {0, 1, -1, 2, -2, 0.5f, -0.5f, 1};
    static constexpr unsigned NEED[8] = {0x55, 0x7e, 0x7e, 0x7e, 0x7e, 0x7e, 0x7e, 0xaa};
    static constexpr double G[8][7] = {{-1. / 4, 0, 0, 0, 0, 0, 0},
                                       {-1. / 18, -1. / 18, -1. / 18, -1. / 18, -1. / 18, -1. / 18, -1. / 18},
                                       {-1. / 18, 1. / 18, -1. / 18, 1. / 18, -1. / 18, 1. / 18, -1. / 18},
                                       {1. / 360, 1. / 180, 1. / 90, 1. / 45, 2. / 45, 4. / 45, 8. / 45},
                                       {1. / 360, -1. / 180, 1. / 90, -1. / 45, 2. / 45, -4. / 45, 8. / 45},
                                       {16. / 45, 8. / 45, 4. / 45, 2. / 45, 1. / 45, 1. / 90, 1. / 180},
                                       {16. / 45, -8. / 45, 4. / 45, -2. / 45, 1. / 45, -1. / 90, 1. / 180},
                                       {0, 0, 0, 0, 0, 0, 1. / 4}};
};

constexpr int BM = 128;                      // output rows per workgroup tile (64 row pairs)
constexpr int BN = 128;                      // output channels per workgroup tile
constexpr int BK = 32;                       // input channels per slab
constexpr int NT = 256;
constexpr int SROW = 128;                    // bytes per (row, slab)
constexpr int REGION_ROWS = 72;              // LDS rows per parity region: 9 pieces of 8 (64 + K/2 + 1 <= 68 are read)
constexpr int A_PIECES = 18;
constexpr int A_BYTES = A_PIECES * 1024;     // 18432
constexpr int B_BYTES = BN * SROW;           // 16384
constexpr int OPER = 2 * A_BYTES + 2 * B_BYTES;   // 69632
constexpr int TLD = BN + 4;                  // epilogue fp32 tile row (floats)
static_assert(BM * TLD * 4 <= OPER, "the epilogue tile must fit the operand buffers");
constexpr size_t LDS_BYTES = (size_t)OPER + BM;
constexpr int RSRC_FLAGS = 0x00020000;       // raw buffer, 32-bit data format (gfx9 family dword 3)

#define XV_BLDS16(rsrc, lptr, voff, soff, imm)                                                                  \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lptr), 16, voff, soff, imm, 0)

struct ToomParams {
    const float *x;
    long R;
    int cin, ldx;
    const float *wp;            // [cout][J * cin]: stage q's transformed tap U_ORDER[q][c][o] at wp[o][q * cin + c]
    int kred;                   // J * cin
    const float *bias, *scale, *shift;
    int act;
    const float *alpha;
    int cout;
    const uint8_t *valid;
    float *y;
    int ldy;
    int n_mt, n_nt;
    int dil;                    // dilation d: the launch runs d independent undilated problems, rows sub, sub + d, sub + 2 d, ...
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// the activation with its kind as a compile-time constant (the same expressions as apply_act of xv_kernels.hip: same bits)
template <int ACT>
__device__ __forceinline__ float act_t(float z, float a)
{
    if constexpr (ACT == XV_ACT_RELU) return fmaxf(z, 0.0f);
    else if constexpr (ACT == XV_ACT_LRELU) return z > 0.0f ? z : a * z;
    else if constexpr (ACT == XV_ACT_PRELU) return fmaxf(z, 0.0f) + a * fminf(z, 0.0f);
    else return z;
}

__device__ __forceinline__ f32x4 fma4(float a, f32x4 x, f32x4 y)
{
    return __builtin_elementwise_fma((f32x4){a, a, a, a}, x, y);
}

// x - y as a packed FMA: a vector fsub is selected as four scalar v_sub_f32 (and fma(-1, y, x) is canonicalised back into one), so
// the -1 comes in as a value the compiler cannot see through.  Beside an fp32 MFMA the instruction count is what matters.
__device__ __forceinline__ f32x4 sub4(f32x4 x, f32x4 y, float m1)
{
    return fma4(m1, y, x);
}

// V_j (divided by SC[j]) from the raw input rows d[0 .. K] of a row pair: one 4-channel fragment.
template <int KT, int j>
__device__ __forceinline__ f32x4 xform(const f32x4 (&d)[KT + 1], float m1)
{
    if constexpr (KT == 3) {
        if constexpr (j == 0) return sub4(d[0], d[2], m1);
        if constexpr (j == 1) return d[1] + d[2];
        if constexpr (j == 2) return sub4(d[2], d[1], m1);
        if constexpr (j == 3) return sub4(d[3], d[1], m1);
    } else if constexpr (KT == 5) {
        if constexpr (j == 0) return fma4(4.f, d[0], fma4(-5.f, d[2], d[4]));
        if constexpr (j == 1) return fma4(-4.f, d[2], d[4]) + fma4(-4.f, d[1], d[3]);
        if constexpr (j == 2) return sub4(fma4(-4.f, d[2], d[4]), fma4(-4.f, d[1], d[3]), m1);
        if constexpr (j == 3) return fma4(2.f, sub4(d[3], d[1], m1), sub4(d[4], d[2], m1));
        if constexpr (j == 4) return fma4(-2.f, sub4(d[3], d[1], m1), sub4(d[4], d[2], m1));
        if constexpr (j == 5) return fma4(4.f, d[1], fma4(-5.f, d[3], d[5]));
    } else {
        if constexpr (j == 0) return fma4(5.25f, sub4(d[2], d[4], m1), sub4(d[6], d[0], m1));
        if constexpr (j == 1) return fma4(-4.25f, d[4], d[2] + d[6]) + fma4(-4.25f, d[3], d[1] + d[5]);
        if constexpr (j == 2) return sub4(fma4(-4.25f, d[4], d[2] + d[6]), fma4(-4.25f, d[3], d[1] + d[5]), m1);
        if constexpr (j == 3) return fma4(2.f, fma4(4.f, d[5], fma4(-5.f, d[3], d[1])), fma4(4.f, d[6], fma4(-5.f, d[4], d[2])));
        if constexpr (j == 4) return fma4(-2.f, fma4(4.f, d[5], fma4(-5.f, d[3], d[1])), fma4(4.f, d[6], fma4(-5.f, d[4], d[2])));
        if constexpr (j == 5) return fma4(2.f, fma4(4.f, d[2], fma4(-5.f, d[4], d[6])), fma4(4.f, d[1], fma4(-5.f, d[3], d[5])));
        if constexpr (j == 6) return fma4(-2.f, fma4(4.f, d[2], fma4(-5.f, d[4], d[6])), fma4(4.f, d[1], fma4(-5.f, d[3], d[5])));
        if constexpr (j == 7) return fma4(5.25f, sub4(d[3], d[5], m1), sub4(d[7], d[1], m1));
    }
}

// The two products of a +- pair of points read the same rows and share the even / odd parts of their transforms: both V at once
// (va for product j, vb for product j + 1), 4 / 6 packed operations per channel pair instead of 6 / 10.
template <int KT, int j>
__device__ __forceinline__ void xform_pair(const f32x4 (&d)[KT + 1], float m1, f32x4 &va, f32x4 &vb)
{
    if constexpr (KT == 3) {
        static_assert(j == 1, "pair: (1, 2)");
        va = d[1] + d[2]; vb = sub4(d[2], d[1], m1);
    } else if constexpr (KT == 5 && j == 1) {
        const f32x4 e = fma4(-4.f, d[2], d[4]), o = fma4(-4.f, d[1], d[3]);
        va = e + o; vb = sub4(e, o, m1);
    } else if constexpr (KT == 5 && j == 3) {
        const f32x4 e = sub4(d[4], d[2], m1), o = sub4(d[3], d[1], m1);
        va = fma4(2.f, o, e); vb = fma4(-2.f, o, e);
    } else if constexpr (KT == 7 && j == 1) {
        const f32x4 e = fma4(-4.25f, d[4], d[2] + d[6]), o = fma4(-4.25f, d[3], d[1] + d[5]);
        va = e + o; vb = sub4(e, o, m1);
    } else if constexpr (KT == 7 && j == 3) {
        const f32x4 e = fma4(4.f, d[6], fma4(-5.f, d[4], d[2])), o = fma4(4.f, d[5], fma4(-5.f, d[3], d[1]));
        va = fma4(2.f, o, e); vb = fma4(-2.f, o, e);
    } else {
        static_assert(KT == 7 && j == 5, "pairs: (1, 2), (3, 4), (5, 6)");
        const f32x4 e = fma4(4.f, d[2], fma4(-5.f, d[4], d[6])), o = fma4(4.f, d[1], fma4(-5.f, d[3], d[5]));
        va = fma4(2.f, e, o); vb = fma4(-2.f, e, o);
    }
}

// the rows of the finished tile: bias, activation, affine, row mask, one 16-B streamed store per (row, 4 columns).  The
// activation kind is a template constant (one uniform switch per thread in the caller, none per element)
template <int ACT>
__device__ __forceinline__ void toom_store_rows(const ToomParams &p, const float *T, const uint8_t *Ms, long m0, int sub, int n0, int tid)
{
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
#pragma unroll 4
    for (int j = 0; j < BM / 8; ++j) {
        const int lr = rp + 8 * j;
        const long gr = sub + (m0 + lr) * p.dil;
        if (gr >= p.R) continue;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(T + lr * TLD + cg * 4);
        const bool keep = Ms[lr] != 0;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = act_t<ACT>(a[i] + bias[i], al[i]) * sc[i] + sh[i];
            v[i] = keep ? t : 0.f;
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p.y + (size_t)gr * p.ldy + gc));   // streamed once: no L2 write-allocate
    }
}

template <int KT>
__global__ __launch_bounds__(NT, 2) void tdnn_gemm_toom_kernel(const ToomParams p)
{
    using TC = Toom<KT>;
    constexpr int J = TC::J;                           // products per row pair = input rows per row pair
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Abuf = lds;                                  // [2][2 regions][72 rows][128 B]
    char *Bbuf = lds + 2 * A_BYTES;                    // [2][BN cols][128 B]
    uint8_t *Ms = reinterpret_cast<uint8_t *>(lds + OPER);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    // A dilated layer (tap offsets 0, +-d, ...) is d independent UNDILATED problems: sub-problem `sub` owns the rows sub, sub + d,
    // sub + 2 d, ... of x and of y -- its row n is row sub + n d, its leading dimensions d ldx / d ldy.  Row pairs are pairs of a
    // sub-problem's consecutive rows (rows r, r + d of the matrix); the host lays chunks out on multiples of 2 d rows, so the pairing
    // of a chunk's rows does not depend on where in the batch it lies.  m0 below counts a sub-problem's rows.
    const int nwg = p.n_mt * p.n_nt * p.dil;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int ms = wg / p.n_nt, nt = wg - ms * p.n_nt;
    const int mt = ms / p.dil, sub = ms - mt * p.dil;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;

    constexpr int left = (KT - 1) / 2;
    const int n_chunks = p.cin / BK;
    const int n_stages = n_chunks * J;

    if (tid < BM) {
        const long gr = sub + (m0 + tid) * p.dil;
        Ms[tid] = (gr < p.R) ? (p.valid ? p.valid[gr] : (uint8_t)1) : (uint8_t)0;
    }

    // ---- DMA.  Piece q = 8 LDS rows x 128 bytes; q < 9: even tile rows 16 q + 2 j8, q >= 9: odd tile rows 16 (q - 9) + 1 + 2 j8
    // (tile row lr = global row m0 - left + lr).  Wave w moves pieces w, w + 4, ...: their parity is the wave's, and with it bit 2
    // of (LDS row >> 1) & 7 -- the swizzle of the row a lane moves is a per-lane constant.  Row part of an address in the VGPR
    // offset (range-checked: rows outside [0, R) come back as zeros), slab part in the scalar offset.
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x) + (size_t)sub * p.ldx, 0,
                                                                         (int)((p.R - sub) * p.ldx * 4), RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wp), 0, (int)((long)p.cout * p.kred * 4), RSRC_FLAGS);
    const int slotb = ((lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7)) << 4;
    const int arow_bytes = p.dil * p.ldx * 4, brow_bytes = p.kred * 4;
    const int va0 = (int)(m0 - left + 2 * (lane >> 3)) * arow_bytes + slotb;
    const int vb0 = (n0 + 8 * wave + (lane >> 3)) * brow_bytes + slotb;
    auto dma_b = [&](int stage, int buf) {              // the weight tile of (slab, product) = stage, four pieces per wave
        const int st = stage < n_stages ? stage : n_stages - 1;
        const int c = st / J, j = st - c * J;
        const int so = (j * p.cin + c * BK) * 4;
        char *dst = Bbuf + buf * B_BYTES + wave * 1024;
#pragma unroll
        for (int t = 0; t < 4; ++t) XV_BLDS16(brs, dst + t * 4096, vb0 + t * 32 * brow_bytes, so, 0);
    };
    auto dma_a_piece = [&](int chunk, int t) {          // piece wave + 4 t of slab `chunk` (clamped: the tail rewrites identical bytes)
        const int c = chunk < n_chunks ? chunk : n_chunks - 1;
        const int q = wave + 4 * t;
        const int rows = q < 9 ? 16 * q : 16 * (q - 9) + 1;
        XV_BLDS16(ars, Abuf + (c & 1) * A_BYTES + q * 1024, va0 + rows * arow_bytes, c * BK * 4, 0);
    };
    auto dma_a_slot = [&](int chunk, int t) {           // slot t of 5: every wave one piece, the last slot waves 0 and 1 only
        if (t < 4 || wave < 2) dma_a_piece(chunk, t);
    };
    dma_b(0, 0);
    dma_b(1, 1);
#pragma unroll
    for (int t = 0; t < 5; ++t) dma_a_slot(0, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- fragments.  Lane (pair pr = lane & 31, k half kh = lane >> 5) reads channels 8 kk + 4 kh .. + 3 = slot 2 kk + kh of a row.
    // Input i of the wave's pair P = wr * 32 + pr: tile row 2 P + i = LDS row (i & 1) * 72 + P + (i >> 1).
    const int kh = lane >> 5;
    int pa[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        const int L = (i & 1) * REGION_ROWS + wr * 32 + (lane & 31) + (i >> 1);
        pa[i] = L * SROW + ((((L >> 1) & 7) ^ kh) << 4);
    }
    const int bcol = wc * 64 + (lane & 31);
    int pbk[4];                                              // weight fragments of k group kk: + buffer and column-block immediates
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) pbk[kk] = (2 * A_BYTES + bcol * SROW + ((((bcol >> 1) & 7) ^ kh) << 4)) ^ (kk << 5);

    float m1 = -1.f;                                         // (see sub4)
    asm volatile("" : "+s"(m1));
    f32x4 raw[J];                                            // the raw fragments of one k group (only the rows the product reads)
    f32x4 vb[4];                                             // V of the second product of a +- pair, four k groups
    struct Bf { f32x4 b0, b1; };
    Bf bf[2];
    auto load_raw = [&](auto JJ, auto CP, int kk) {          // product j, halo buffer CP, k group kk
        constexpr int j = decltype(JJ)::value;
        constexpr int off = decltype(CP)::value * A_BYTES;
        auto one = [&](auto II) {
            constexpr int i = decltype(II)::value;
            if constexpr ((TC::NEED[j] >> i) & 1) raw[i] = *reinterpret_cast<const f32x4 *>(lds + (pa[i] ^ (kk << 5)) + off);
        };
        static_for<0, J>(one);
    };
    auto load_b = [&](Bf &X, auto BP, int kk) {
        constexpr int off = decltype(BP)::value * B_BYTES;
        X.b0 = *reinterpret_cast<const f32x4 *>(lds + pbk[kk] + off);
        X.b1 = *reinterpret_cast<const f32x4 *>(lds + pbk[kk] + off + 32 * SROW);
    };

    f32x16 o00 = {0}, o01 = {0}, o10 = {0}, o11 = {0};      // output rows 2P (o0x) and 2P + 1 (o1x), column blocks 0 / 1
    f32x16 tA0 = {0}, tA1 = {0}, tB0 = {0}, tB1 = {0};      // V_j . U_j of a +- product's stage, column blocks 0 / 1
    const f32x16 zero16 = {0};

    auto mma8 = [&](const f32x4 a, const Bf &X, f32x16 &t0, f32x16 &t1, bool fresh) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], X.b0[e], (fresh && e == 0) ? zero16 : t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], X.b1[e], (fresh && e == 0) ? zero16 : t1, 0, 0, 0);
        }
    };
    auto add_to = [&](f32x16 &o, const f32x16 &t) { o += t; };                                   // 8 v_pk_add_f32
    auto fma_to = [&](auto JJ, f32x16 &o, const f32x16 &t) {                                     // o += AT[1][j] t
        constexpr float a = TC::A1[decltype(JJ)::value];
        if constexpr (a == 1.f) o += t;
        else if constexpr (a == -1.f) o = __builtin_elementwise_fma((f32x16){m1, m1, m1, m1, m1, m1, m1, m1, m1, m1, m1, m1, m1, m1, m1, m1}, t, o);
        else o = __builtin_elementwise_fma((f32x16){a, a, a, a, a, a, a, a, a, a, a, a, a, a, a, a}, t, o);
    };
    // (a fold is plain arithmetic on SSA values: without an anchor in the chain of side effects instruction selection emits it
    // at the end of the loop body and every stage's temporaries stay live; the anchors also keep each piece in its step)
    auto anchor = [&](f32x16 &a) { asm volatile("" : "+v"(a)); };
    auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
    auto order_step = [&]() {                                // VALU first, then the next step's fragment reads, then 8 MFMAs back to back
        __builtin_amdgcn_sched_group_barrier(0x002, 96, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    };
    // (hipcc's pre-emit peephole UNPACKS v_pk_*_f32 that follow an MFMA within its latency into two scalar instructions, to co-issue
    // them under the MFMA; under an fp32 MFMA nothing co-issues, the unpacked pair just costs twice.  Its scan stops at an
    // instruction that names the MFMA's destination: an empty asm on the step's last accumulator is that instruction.)
    auto end_step = [&](f32x16 &t) { asm volatile("" : "+v"(t)); };
    using JLAST = std::integral_constant<int, J - 2>;        // the product whose temporary (tB) outlives its slab

    int s = 0;
    auto slab = [&](auto CP, int c) {
        constexpr int cp = decltype(CP)::value;
        auto stage = [&](auto QQ) {
            constexpr int q = decltype(QQ)::value;
            constexpr int j = TC::ORDER[q];
            using JJ = std::integral_constant<int, j>;
            using BP = std::integral_constant<int, q & 1>;
            constexpr bool direct = q < 2;
            f32x16 &t0 = q == 0 ? o00 : q == 1 ? o10 : (q & 1) ? tB0 : tA0;
            f32x16 &t1 = q == 0 ? o01 : q == 1 ? o11 : (q & 1) ? tB1 : tA1;
            f32x16 &u0 = (q & 1) ? tA0 : tB0, &u1 = (q & 1) ? tA1 : tB1;       // the previous +- stage's products (q >= 3)
            using JP = std::integral_constant<int, TC::ORDER[q >= 3 ? q - 1 : 2]>;
            // stages 2, 3 / 4, 5 / 6, 7 hold the two products of a +- pair: the first forms both V (the second's into vb), the second
            // reads no raw fragments at all
            constexpr bool pair_a = q >= 2 && !(q & 1), pair_b = q >= 2 && (q & 1);
            auto form = [&](int kk) -> f32x4 {
                if constexpr (pair_b) return vb[kk];
                else if constexpr (pair_a) {
                    f32x4 a;
                    xform_pair<KT, j>(raw, m1, a, vb[kk]);
                    return a;
                } else return xform<KT, j>(raw, m1);
            };
            // ---- step 0
            f32x4 v = form(0);
            if constexpr (!pair_b) load_raw(JJ{}, CP, 1);
            load_b(bf[1], BP{}, 1);
            mma8(v, bf[0], t0, t1, !direct);
            order_step();
            end_step(t1);
            fence();
            // ---- step 1 (+ the folds into column block 0)
            v = form(1);
            if constexpr (q == 0) { fma_to(JLAST{}, o10, tB0); anchor(o10); }
            if constexpr (q == 1) { add_to(o00, tB0); anchor(o00); }
            if constexpr (q >= 3) { add_to(o00, u0); fma_to(JP{}, o10, u0); anchor(o00); anchor(o10); }
            if constexpr (!pair_b) load_raw(JJ{}, CP, 2);
            load_b(bf[0], BP{}, 2);
            mma8(v, bf[1], t0, t1, false);
            order_step();
            end_step(t1);
            fence();
            // ---- step 2 (+ the folds into column block 1)
            v = form(2);
            if constexpr (q == 0) { fma_to(JLAST{}, o11, tB1); anchor(o11); }
            if constexpr (q == 1) { add_to(o01, tB1); anchor(o01); }
            if constexpr (q >= 3) { add_to(o01, u1); fma_to(JP{}, o11, u1); anchor(o01); anchor(o11); }
            if constexpr (!pair_b) load_raw(JJ{}, CP, 3);
            load_b(bf[1], BP{}, 3);
            mma8(v, bf[0], t0, t1, false);
            order_step();
            end_step(t1);
            fence();
            // ---- step 3: every fragment of stage s is in registers, stage s + 1 has landed
            v = form(3);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma_b(s + 2, q & 1);
            if constexpr (q < 5) dma_a_slot(c + 1, q);
            if constexpr (J < 5 && q == J - 1) {                        // (F(2, 3): four stages for the five slots)
#pragma unroll
                for (int t = J; t < 5; ++t) dma_a_slot(c + 1, t);
            }
            fence();
            if constexpr (pair_a) { /* the next stage's V are in vb */ }
            else if constexpr (q + 1 < J) load_raw(std::integral_constant<int, TC::ORDER[(q + 1) % J]>{}, CP, 0);
            else load_raw(std::integral_constant<int, TC::ORDER[0]>{}, std::integral_constant<int, 1 - cp>{}, 0);
            load_b(bf[0], std::integral_constant<int, (q + 1) & 1>{}, 0);
            mma8(v, bf[1], t0, t1, false);
            end_step(t1);
            fence();
            ++s;
        };
        static_for<0, J>(stage);
    };
    load_raw(std::integral_constant<int, TC::ORDER[0]>{}, std::integral_constant<int, 0>{}, 0);
    load_b(bf[0], std::integral_constant<int, 0>{}, 0);
    for (int c = 0; c < n_chunks; c += 2) {
        slab(std::integral_constant<int, 0>{}, c);
        if (c + 1 < n_chunks) slab(std::integral_constant<int, 1>{}, c + 1);
    }
    fma_to(JLAST{}, o10, tB0);
    fma_to(JLAST{}, o11, tB1);
    add_to(o00, tB0);
    add_to(o01, tB1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the tail's clamped pieces must have landed before the tile below reuses the LDS)
    __syncthreads();

    // ---- epilogue through an fp32 tile in LDS: D layout of a 32x32 MFMA tile: col = lane & 31, pair = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float *T = reinterpret_cast<float *>(lds);
    {
        const int col = wc * 64 + (lane & 31);
        const int rowb = wr * 64 + 8 * (lane >> 5);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = rowb + 2 * ((reg & 3) + 8 * (reg >> 2));
            T[lr * TLD + col] = o00[reg];
            T[lr * TLD + col + 32] = o01[reg];
            T[(lr + 1) * TLD + col] = o10[reg];
            T[(lr + 1) * TLD + col + 32] = o11[reg];
        }
    }
    __syncthreads();
    switch (p.act) {
    case XV_ACT_RELU: toom_store_rows<XV_ACT_RELU>(p, T, Ms, m0, sub, n0, tid); break;
    case XV_ACT_LRELU: toom_store_rows<XV_ACT_LRELU>(p, T, Ms, m0, sub, n0, tid); break;
    case XV_ACT_PRELU: toom_store_rows<XV_ACT_PRELU>(p, T, Ms, m0, sub, n0, tid); break;
    default: toom_store_rows<XV_ACT_NONE>(p, T, Ms, m0, sub, n0, tid); break;
    }
}

// U_j[c][o] = SC[j] sum_k G[j][k] w[k][c][o] in double, rounded once, laid out as the GEMM's B operand in STAGE order:
// wp[o][q * cin + c] = U_ORDER[q][c][o]
template <int KT>
__global__ void pack_weights_toom_kernel(const float *__restrict__ w, int cin, int cout, float *__restrict__ wp)
{
    using TC = Toom<KT>;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)cin * cout) return;
    const int c = (int)(i / cout), o = (int)(i - (size_t)c * cout);
    double g[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) g[k] = (double)w[((size_t)k * cin + c) * cout + o];
#pragma unroll
    for (int q = 0; q < TC::J; ++q) {
        const int j = TC::ORDER[q];
        double u = 0.0;
#pragma unroll
        for (int k = 0; k < KT; ++k) u += TC::G[j][k] * g[k];
        wp[(size_t)o * (TC::J * cin) + (size_t)q * cin + c] = (float)(TC::SC[j] * u);
    }
}

}  // namespace

extern "C" {

int xv_toom_supported(int K, int dilation, int cin, int cout)
{
    return (K == 3 || K == 5 || K == 7) && dilation >= 1 && dilation <= 8 && cin > 0 && cin % BK == 0 && cout > 0 && cout % 4 == 0;
}

size_t xv_packed_weights_toom_f32_floats(int K, int cin, int cout)
{
    if (!xv_toom_supported(K, 1, cin, cout)) return 0;
    return (size_t)(K + 1) * cin * cout;
}

int xv_pack_weights_toom_f32(const float *w, int K, int cin, int cout, float *wp, void *stream)
{
    if (!w || !wp) return fail(XV_ERR_BAD_ARG, "pack_weights_toom: NULL argument");
    if (!xv_toom_supported(K, 1, cin, cout)) return fail(XV_ERR_UNSUPPORTED, "pack_weights_toom: needs K in {3, 5, 7}, Cin % 32 == 0, Cout % 4 == 0");
    const size_t n = (size_t)cin * cout;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (K == 3) hipLaunchKernelGGL(pack_weights_toom_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, w, cin, cout, wp);
    else if (K == 5) hipLaunchKernelGGL(pack_weights_toom_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, w, cin, cout, wp);
    else hipLaunchKernelGGL(pack_weights_toom_kernel<7>, grid, dim3(256), 0, (hipStream_t)stream, w, cin, cout, wp);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "pack_weights_toom_kernel launch");
}

int xv_tdnn_layer_toom_dilated_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                                   const float *bn_shift, int act_kind, const float *act_alpha, int K, int dilation, int cout,
                                   const uint8_t *row_valid, float *y, int ldy, void *stream)
{
    if (!x || !wp || !y) return fail(XV_ERR_BAD_ARG, "tdnn_toom: NULL argument");
    if (R <= 0) return 0;
    if (!xv_toom_supported(K, dilation, cin, cout))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_toom: needs K in {3, 5, 7}, dilation 1 .. 8, Cin % 32 == 0, Cout % 4 == 0");
    if (ldx < cin || ldy < cout) return fail(XV_ERR_BAD_ARG, "tdnn_toom: leading dimension too small");
    if ((act_kind == XV_ACT_LRELU || act_kind == XV_ACT_PRELU) && !act_alpha) return fail(XV_ERR_BAD_ARG, "tdnn_toom: act_alpha is NULL");
    const uintptr_t bits = (uintptr_t)x | (uintptr_t)wp | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)bn_scale | (uintptr_t)bn_shift |
                           (act_kind == XV_ACT_PRELU ? (uintptr_t)act_alpha : 0);
    if (bits % 16 != 0 || ldx % 4 != 0 || ldy % 4 != 0) return fail(XV_ERR_UNSUPPORTED, "tdnn_toom: needs 16-byte aligned rows and per-column parameters");
    ToomParams p;
    p.x = x; p.R = R; p.cin = cin; p.ldx = ldx; p.wp = wp; p.kred = (K + 1) * cin;
    p.bias = bias; p.scale = bn_scale; p.shift = bn_shift; p.act = act_kind; p.alpha = act_alpha; p.cout = cout;
    p.valid = row_valid; p.y = y; p.ldy = ldy;
    p.dil = dilation;
    const long rows_sub = (R + dilation - 1) / dilation;            // rows of the largest sub-problem (sub = 0)
    p.n_mt = (int)((rows_sub + BM - 1) / BM);
    p.n_nt = (cout + BN - 1) / BN;
    if ((R + (long)(BM + 8) * dilation) * (long)ldx * 4 >= (1l << 31) || (long)(cout + BN) * p.kred * 4 >= (1l << 31))
        return fail(XV_ERR_UNSUPPORTED, "tdnn_toom: matrices must stay below 2^31 bytes (32-bit buffer offsets)");
    typedef void (*kern_t)(const ToomParams);
    const kern_t k = K == 3 ? tdnn_gemm_toom_kernel<3> : K == 5 ? tdnn_gemm_toom_kernel<5> : tdnn_gemm_toom_kernel<7>;
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        for (kern_t kk : {tdnn_gemm_toom_kernel<3>, tdnn_gemm_toom_kernel<5>, tdnn_gemm_toom_kernel<7>}) {
            hipError_t e = hipFuncSetAttribute((const void *)kk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
        }
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)(p.n_mt * p.n_nt * dilation)), dim3(NT), LDS_BYTES, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "tdnn_gemm_toom_kernel launch");
}

int xv_tdnn_layer_toom_f32(const float *x, int64_t R, int cin, int ldx, const float *wp, const float *bias, const float *bn_scale,
                           const float *bn_shift, int act_kind, const float *act_alpha, int K, int cout, const uint8_t *row_valid,
                           float *y, int ldy, void *stream)
{
    return xv_tdnn_layer_toom_dilated_f32(x, R, cin, ldx, wp, bias, bn_scale, bn_shift, act_kind, act_alpha, K, 1, cout, row_valid, y, ldy,
                                          stream);
}

}  // extern "C"
