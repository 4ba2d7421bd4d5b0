// PROBE for the next step on the power-limited GEMMs: the operand / result layout of the 16 x 16 MFMA shapes the
// f16bf8 kernels would move to -- v_mfma_f32_16x16x32_f16 and v_mfma_scale_f32_16x16x128_f8f6f4 with bf8 (e5m2) operands -- checked
// against a host reference, and whether the block scale of the scaled instruction is taken per lane (i.e. per row and 32-value K block).
// mx_probe.hip is the same check for the 32 x 32 shapes.  RESULT (MI355X, end of round 3): H1 holds EXACTLY for both instructions --
// operands and result -- with no scaling, with uniform scales and with a scale that differs per ROW (trial 3); a scale that differs per K
// block of a row (trial 2: lanes l, l + 16, l + 32, l + 48 passing different bytes) does NOT follow H1 (max |D - ref| 34.7 of 46.2): the
// instruction does not take the four K blocks' scales from the four lanes that hold them.  The f16bf8 kernels pass one constant per
// operand, so the shapes can be used as they are; a per-block scale would need the ISA's rule first.
//   hipcc --offload-arch=gfx950 -O3 -o mx16_probe mx16_probe.hip && ./mx16_probe
// Hypotheses (H1), as the 32 x 32 shapes suggest:  A: lane l = row (l & 15), K block (l >> 4) -- 8 halves k = 8 (l >> 4) + j for f16,
// 32 bytes k = 32 (l >> 4) + j for the scaled form;  B: lane l = column (l & 15), the same K blocks;  D (4 floats per lane): column
// (l & 15), rows 4 (l >> 4) + i;  scale: byte 0 of the lane's scale register applies to that lane's 32 values.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static float bf8_to_float(uint8_t b)
{
    const int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
    float v;
    if (e == 0) v = ldexpf((float)m, -16);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf(1.f + m / 4.f, e - 15);
    return s ? -v : v;
}

// A[16][128] bytes (row-major), B[128][16] bytes (k-major), D[16][16] floats; sa[64] / sb[64]: the scale byte every lane passes
__global__ void layout8_kernel(const uint8_t *A, const uint8_t *B, float *D, const int *sa, const int *sb)
{
    const int lane = threadIdx.x, rc = lane & 15, kb = lane >> 4;
    i32x8 a, b;
    const int *ap = reinterpret_cast<const int *>(A + rc * 128 + kb * 32);
    uint8_t bb[32];
    for (int j = 0; j < 32; ++j) bb[j] = B[(kb * 32 + j) * 16 + rc];
    for (int j = 0; j < 8; ++j) {
        a[j] = ap[j];
        b[j] = bb[4 * j] | (bb[4 * j + 1] << 8) | (bb[4 * j + 2] << 16) | (bb[4 * j + 3] << 24);
    }
    int va = sa[lane], vb = sb[lane];
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, 0, va, 0, vb);      // cbsz = blgp = 1: bf8 x bf8
    for (int i = 0; i < 4; ++i) D[(4 * kb + i) * 16 + rc] = c[i];
}

// A[16][32] halves, B[32][16] halves
__global__ void layout16_kernel(const _Float16 *A, const _Float16 *B, float *D)
{
    const int lane = threadIdx.x, rc = lane & 15, kb = lane >> 4;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[rc * 32 + kb * 8 + j];
        b[j] = B[(kb * 8 + j) * 16 + rc];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * kb + i) * 16 + rc] = c[i];
}

int main()
{
    srand(1);
    const uint8_t vals[] = {0x00, 0x3c, 0xbc, 0x40, 0xc0, 0x38, 0xb8, 0x44, 0x34, 0xb4, 0x3e, 0xbe};   // 0, +-1, +-2, +-.5, 4, +-.25, +-1.5
    std::vector<uint8_t> A(16 * 128), B(128 * 16);
    for (auto &v : A) v = vals[rand() % 12];
    for (auto &v : B) v = vals[rand() % 12];
    uint8_t *dA, *dB; float *dD; int *dsa, *dsb;
    CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dD, 256 * 4)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    for (int trial = 0; trial < 4; ++trial) {
        // 0: no scaling  1: uniform scales  2: A's scale differs per K block (lane >> 4)  3: A's scale differs per row (lane & 15)
        std::vector<int> sa(64), sb(64);
        for (int l = 0; l < 64; ++l) {
            sa[l] = trial == 0 ? 127 : trial == 1 ? 115 : trial == 2 ? 124 + (l >> 4) : 120 + (l & 15);
            sb[l] = trial == 1 ? 130 : 127;
        }
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        layout8_kernel<<<1, 64>>>(dA, dB, dD, dsa, dsb);
        std::vector<float> D(256);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        double worst = 0, ref_max = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int k = 0; k < 128; ++k)      // scale of A's value (i, k): the lane that holds it = row i, K block k / 32; of B's: column j
                    ref += (double)bf8_to_float(A[i * 128 + k]) * ldexp(1.0, sa[i + 16 * (k / 32)] - 127) *
                           (double)bf8_to_float(B[k * 16 + j]) * ldexp(1.0, sb[j + 16 * (k / 32)] - 127);
                worst = fmax(worst, fabs(ref - D[i * 16 + j]));
                ref_max = fmax(ref_max, fabs(ref));
            }
        printf("16x16x128 bf8, trial %d: max |D - ref| = %g (max |ref| %g)  %s\n", trial, worst, ref_max, worst == 0 ? "EXACT" : "MISMATCH");
    }
    std::vector<_Float16> Ah(16 * 32), Bh(32 * 16);
    for (auto &v : Ah) v = (_Float16)((rand() % 17 - 8) * 0.25f);
    for (auto &v : Bh) v = (_Float16)((rand() % 17 - 8) * 0.5f);
    _Float16 *dAh, *dBh;
    CK(hipMalloc(&dAh, Ah.size() * 2)); CK(hipMalloc(&dBh, Bh.size() * 2));
    CK(hipMemcpy(dAh, Ah.data(), Ah.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBh, Bh.data(), Bh.size() * 2, hipMemcpyHostToDevice));
    layout16_kernel<<<1, 64>>>(dAh, dBh, dD);
    std::vector<float> D(256);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0;
            for (int k = 0; k < 32; ++k) ref += (double)(float)Ah[i * 32 + k] * (double)(float)Bh[k * 16 + j];
            worst = fmax(worst, fabs(ref - D[i * 16 + j]));
        }
    printf("16x16x32 f16: max |D - ref| = %g  %s\n", worst, worst == 0 ? "EXACT" : "MISMATCH");
    return 0;
}
