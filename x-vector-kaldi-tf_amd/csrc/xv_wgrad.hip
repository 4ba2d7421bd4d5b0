// xv_wgrad.hip -- weight gradient of a TDNN / FC layer in the bf16x3 arithmetic (training step):
//
//     dW[k, ci, co] = sum_r x[r + (k - (K-1)/2) d, ci] * dz[r, co]                       (the conv1d filter gradient TF derives for
//                                                                                          local/tf/models.py:60 inside minimize(), :109-113)
//
// Same contract as xv_wgrad_f32 (xv_train.hip): TF layout [K, Cin, Cout], rows outside [0, R) read as zero, the row range is
// cut into splits whose partial tiles are merged in a fixed order (deterministic).  What differs is the arithmetic: the
// exact-fp32 kernel is bound by v_mfma_f32_32x32x2_f32 (157 TF peak; 26 % of a training step), here every fp32 operand is
// split hi + lo (bf16 each) and a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the
// arithmetic of the forward / input-gradient GEMMs of --train-precision bf16x3 (xv_kernels.hip), ~5e-6 relative.
//
// The reduction runs over ROWS, so both MFMA operands are needed "transposed": a lane of the A (cin) or B (cout) fragment holds
// 8 consecutive rows of ONE channel, while memory holds a row's channels side by side.  The transposition happens on the way
// into LDS and costs no extra pass:
//   * thread t owns channel t & 127 of the 128-channel tile and rows 16 (t >> 7) .. + 15 of the 32-row step: sixteen dword loads
//     per operand (a wave reads 256 contiguous bytes per row: coalesced), through buffer descriptors -- rows before 0 and
//     past R fall outside the descriptor and come back as zero, no branch per element;
//   * its 16 values become 16 hi + 16 lo bf16 = 4 x 16 bytes, written with ds_write_b128 into a [channel][row] image
//     (80-byte channel stride: both the writes' 8-lane groups and the reads' 16-lane groups touch distinct 16-byte slots);
//   * a fragment is then ONE ds_read_b128 per (32-channel tile, k-step, plane): 16 reads for the 24 MFMAs of a step and wave.
// Workgroup = 4 waves as 2 x 2, 128 x 128 tile, 64 x 64 per wave; next step's 32 loads per thread are in flight under the MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xvector_hip.h"

extern "C" void xv_internal_set_error(const char *msg);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int WT = 128;           // tile edge (channels)
constexpr int WR = 32;            // rows per step = two MFMA k-steps
constexpr int CH_STRIDE = 80;     // bytes per channel in the LDS image: 32 rows x 2 bytes + 16 (see above)
constexpr int PLANE = WT * CH_STRIDE;                  // 10240: one operand, one of hi / lo
constexpr int RSRC_FLAGS = 0x00020000;                 // raw buffer, 32-bit data format (gfx9 family dword 3)

struct WgradParams {
    const float *x;
    const float *dz;
    long R;
    int cin, ldx, cout, lddz, K, dil;
    int n_ct, n_ot;      // tiles along cin / cout
    long rows_per_split;
    float *out;          // [nsplit][K][cin][cout] (or dw itself when nsplit == 1)
    double *db_part;     // BIAS: [nsplit][cout] partial column sums of dz (the bias gradient), written by the workgroups with k == 0, ct == 0
};

// BIAS: the workgroups of tap 0 / input tile 0 also leave the column sums of the dz rows they stream anyway -- db = sum_r dz[r, :], the bias
// gradient of the same layer (models.py:61 bias_add under minimize()) -- per row split, in double: the separate pass over dz that
// xv_col_sums_f32 made for it was 4 % of a training step (tools/experiments/train_bias_sums_ablation.py).
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void wgrad_bf16x3_kernel(const WgradParams p)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * PLANE];       // [x hi | x lo | dz hi | dz lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    int t = blockIdx.x;
    const int ot = t % p.n_ot; t /= p.n_ot;
    const int ct = t % p.n_ct; t /= p.n_ct;
    const int k = t;
    const int c0 = ct * WT, o0 = ot * WT;
    const long shift = (long)(k - (p.K - 1) / 2) * p.dil;
    const long r_begin = (long)blockIdx.y * p.rows_per_split;
    const long r_end = r_begin + p.rows_per_split < p.R ? r_begin + p.rows_per_split : p.R;

    // operands through buffer descriptors of exactly R rows: a row index outside [0, R) -- before the first row the byte offset
    // wraps to > 2^31 -- is out of range and loads as zero (the host checks R * ld * 4 < 2^31)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x), 0, (int)(p.R * p.ldx * 4), RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.dz), 0, (int)(p.R * p.lddz * 4), RSRC_FLAGS);
    const int ch = tid & 127, half = tid >> 7;
    const bool xc_ok = c0 + ch < p.cin, zc_ok = o0 + ch < p.cout;      // a ragged last tile: the column would alias the next row
    const int xcol = (c0 + ch) * 4, zcol = (o0 + ch) * 4;
    const int xrow_bytes = p.ldx * 4, zrow_bytes = p.lddz * 4;

    float vx[16], vz[16];
    auto load = [&](long r0) {
        const long rb = r0 + 16 * half;
        const int xo = (int)((rb + shift) * xrow_bytes) + xcol;
        const int zo = (int)(rb * zrow_bytes) + zcol;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            vx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo + i * xrow_bytes, 0, 0));
            vz[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrs, zo + i * zrow_bytes, 0, 0));
        }
    };
    auto stage = [&](const float (&v)[16], bool ok, char *plane_hi) {     // 16 rows of one channel -> [hi | lo] images
        bf16x8 hi[2], lo[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float f = ok ? v[i] : 0.f;
            const __bf16 h = (__bf16)f;
            hi[i >> 3][i & 7] = h;
            lo[i >> 3][i & 7] = (__bf16)(f - (float)h);
        }
        char *d = plane_hi + ch * CH_STRIDE + 32 * half;
        *reinterpret_cast<bf16x8 *>(d) = hi[0];
        *reinterpret_cast<bf16x8 *>(d + 16) = hi[1];
        *reinterpret_cast<bf16x8 *>(d + PLANE) = lo[0];
        *reinterpret_cast<bf16x8 *>(d + PLANE + 16) = lo[1];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
    // fragment of tile i, k-step ks: channel wi*64 + 32 i + (lane & 31), rows 16 ks + 8 (lane >> 5) .. + 7
    const char *afrag = lds + (wi * 64 + (lane & 31)) * CH_STRIDE + 16 * (lane >> 5);
    const char *bfrag = lds + 2 * PLANE + (wj * 64 + (lane & 31)) * CH_STRIDE + 16 * (lane >> 5);

    const bool sums = BIAS && k == 0 && ct == 0;                        // (uniform per workgroup)
    double bsum = 0.0;
    load(r_begin);
    for (long r0 = r_begin; r0 < r_end; r0 += WR) {
        if (BIAS && sums) {                                             // 16 rows of this thread's channel: fp32 inside the step, double across steps
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += vz[i];
            bsum += (double)t;
        }
        stage(vx, xc_ok, lds);
        stage(vz, zc_ok, lds + 2 * PLANE);
        __syncthreads();
        if (r0 + WR < r_end) load(r0 + WR);              // in flight under the 24 MFMAs below
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const bf16x8 *>(afrag + i * 32 * CH_STRIDE + ks * 32);
                al[i] = *reinterpret_cast<const bf16x8 *>(afrag + i * 32 * CH_STRIDE + ks * 32 + PLANE);
                bh[i] = *reinterpret_cast<const bf16x8 *>(bfrag + i * 32 * CH_STRIDE + ks * 32);
                bl[i] = *reinterpret_cast<const bf16x8 *>(bfrag + i * 32 * CH_STRIDE + ks * 32 + PLANE);
            }
            // small terms first, as in the forward kernel (xv_kernels.hip): lo*hi + hi*lo + hi*hi
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    if (BIAS && sums) {                                                 // the two row halves of a channel, in a fixed order
        double *sh = reinterpret_cast<double *>(lds);                   // (the operand images are dead: the loop ended on a barrier)
        if (half) sh[ch] = bsum;
        __syncthreads();
        if (!half && zc_ok) p.db_part[(size_t)blockIdx.y * p.cout + o0 + ch] = bsum + sh[ch];
    }
    // D: col = lane & 31 (cout), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (cin)
    float *out = p.out + ((size_t)blockIdx.y * p.K + k) * (size_t)p.cin * p.cout;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj) {
            const int o = o0 + wj * 64 + bj * 32 + (lane & 31);
            if (o >= p.cout) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int c = c0 + wi * 64 + bi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                if (c < p.cin) out[(size_t)c * p.cout + o] = acc[bi][bj][reg];
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

__global__ void bias_splits_kernel(const double *__restrict__ part, int cout, int nsplit, float *__restrict__ db)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cout) return;
    double s = 0.0;
    for (int j = 0; j < nsplit; ++j) s += part[(size_t)j * cout + c];   // fixed order: deterministic
    db[c] = (float)s;
}

}  // namespace

extern "C" {

size_t xv_wgrad_workspace_bytes(int64_t R, int cin, int cout, int K);       // (xv_train.hip: the split rule is shared)

static int wgrad_bf16x3_impl(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation, float *dw,
                             float *db, void *workspace, void *stream);

int xv_wgrad_bf16x3(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation, float *dw,
                    void *workspace, void *stream)
{
    return wgrad_bf16x3_impl(x, ldx, dz, lddz, R, cin, cout, K, dilation, dw, nullptr, workspace, stream);
}

size_t xv_wgrad_bias_workspace_bytes(int64_t R, int cin, int cout, int K)
{
    if (R <= 0 || cin <= 0 || cout <= 0 || K <= 0) return 0;
    const size_t base = xv_wgrad_workspace_bytes(R, cin, cout, K);
    const size_t splits = base ? base / ((size_t)K * cin * (size_t)cout * sizeof(float)) : 1;
    return (base + 7) / 8 * 8 + splits * (size_t)cout * sizeof(double);
}

int xv_wgrad_bias_bf16x3(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation, float *dw,
                         float *db, void *workspace, void *stream)
{
    if (!db || !workspace) {
        xv_internal_set_error("wgrad_bias_bf16x3: db and the workspace of xv_wgrad_bias_workspace_bytes are required");
        return XV_ERR_BAD_ARG;
    }
    return wgrad_bf16x3_impl(x, ldx, dz, lddz, R, cin, cout, K, dilation, dw, db, workspace, stream);
}

static int wgrad_bf16x3_impl(const float *x, int ldx, const float *dz, int lddz, int64_t R, int cin, int cout, int K, int dilation, float *dw,
                             float *db, void *workspace, void *stream)
{
    if (!x || !dz || !dw || R <= 0 || cin <= 0 || cout <= 0 || K <= 0 || !(K & 1) || dilation <= 0 || ldx < cin || lddz < cout) {
        xv_internal_set_error("wgrad_bf16x3: bad argument");
        return XV_ERR_BAD_ARG;
    }
    if ((double)R * ldx * 4 >= 2147483648.0 || (double)R * lddz * 4 >= 2147483648.0) {   // 32-bit buffer offsets
        if (db) {
            xv_internal_set_error("wgrad_bias_bf16x3: matrices must stay below 2^31 bytes (use xv_wgrad_bf16x3 + xv_col_sums_f32)");
            return XV_ERR_UNSUPPORTED;
        }
        return xv_wgrad_f32(x, ldx, dz, lddz, R, cin, cout, K, dilation, dw, workspace, stream);
    }
    WgradParams p{};
    p.x = x; p.dz = dz; p.R = (long)R; p.cin = cin; p.ldx = ldx; p.cout = cout; p.lddz = lddz; p.K = K; p.dil = dilation;
    p.n_ct = (cin + WT - 1) / WT; p.n_ot = (cout + WT - 1) / WT;
    const size_t ws_bytes = xv_wgrad_workspace_bytes(R, cin, cout, K);
    const long splits = ws_bytes ? (long)(ws_bytes / ((size_t)K * cin * (size_t)cout * sizeof(float))) : 1;
    p.rows_per_split = ((R + splits - 1) / splits + WR - 1) / WR * WR;
    if (splits > 1 && !workspace) {
        xv_internal_set_error("wgrad_bf16x3: workspace required");
        return XV_ERR_BAD_ARG;
    }
    p.out = splits > 1 ? (float *)workspace : dw;
    p.db_part = db ? reinterpret_cast<double *>((char *)workspace + (ws_bytes + 7) / 8 * 8) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (db) hipLaunchKernelGGL(wgrad_bf16x3_kernel<true>, dim3((unsigned)(K * p.n_ct * p.n_ot), (unsigned)splits), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(wgrad_bf16x3_kernel<false>, dim3((unsigned)(K * p.n_ct * p.n_ot), (unsigned)splits), dim3(256), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && db) {
        hipLaunchKernelGGL(bias_splits_kernel, dim3((unsigned)((cout + 255) / 256)), dim3(256), 0, st, (const double *)p.db_part, cout, (int)splits, db);
        e = hipGetLastError();
    }
    if (e == hipSuccess && splits > 1) {
        const size_t n = (size_t)K * cin * cout;
        hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float *)workspace, n,
                           (int)splits, dw);
        e = hipGetLastError();
    }
    if (e != hipSuccess) {
        xv_internal_set_error(hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

}  // extern "C"
