// EXPERIMENT: (1) operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with bf8 (e5m2) operands, (2) MFMA-only throughput of the
// shipped bf16x3 pattern (24 x bf16 32x32x16 per 32-channel stage and wave) against the candidate "f16 + bf8 cross terms"
// pattern (8 x f16 32x32x16 + 4 x MX 32x32x64) with register-resident operands.
//   hipcc --offload-arch=gfx950 -O3 -o mx_probe mx_probe.hip && ./mx_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

// ---- (1) layout: hypothesis H1 -- lane l holds row/col (l & 31), K block (l >> 5), byte j of its 32 bytes = k 32*(l>>5) + j
__global__ void layout_kernel(const uint8_t *A, const uint8_t *B, float *D, int sa, int sb)
{
    const int lane = threadIdx.x;
    i32x8 a, b;
    const int *ap = reinterpret_cast<const int *>(A + (lane & 31) * 64 + (lane >> 5) * 32);   // A[row][k]
    uint8_t bb[32];
    for (int j = 0; j < 32; ++j) bb[j] = B[((lane >> 5) * 32 + j) * 32 + (lane & 31)];       // B[k][col]
    for (int j = 0; j < 8; ++j) {
        a[j] = ap[j];
        b[j] = bb[4 * j] | (bb[4 * j + 1] << 8) | (bb[4 * j + 2] << 16) | (bb[4 * j + 3] << 24);
    }
    int va = sa, vb = sb;
    asm volatile("" : "+v"(va), "+v"(vb));
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, va, 0, vb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[r];
}

// ---- (2) throughput ------------------------------------------------------------------------------------------------
template <int MODE>   // 0: 24 bf16   1: 24 f16   2: 8 f16 + 4 MX bf8   3: 12 MX only   7: 8 f16 + 4 MX fp6 (e2m3)   8: 12 MX fp6 only
__global__ __launch_bounds__(512, 1) void rate_kernel(const uint8_t *src, float *out, int iters)
{
    const int tid = threadIdx.x;
    const uint8_t *s = src + (size_t)(tid & 63) * 512 + (tid >> 6) * 32768;
    i32x8 r[8];
    for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const i32x8 *>(s + 32 * j);
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    int va = 115, vb = 127;
    asm volatile("" : "+v"(va), "+v"(vb));
    for (int it = 0; it < (MODE >= 4 && MODE < 7 ? 0 : iters); ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                bf16x8 a0, a1, b0, b1;
                __builtin_memcpy(&a0, &r[u & 7], 16); __builtin_memcpy(&a1, reinterpret_cast<char *>(&r[(u + 1) & 7]) + 16, 16);
                __builtin_memcpy(&b0, &r[(u + 2) & 7], 16); __builtin_memcpy(&b1, reinterpret_cast<char *>(&r[(u + 3) & 7]) + 16, 16);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                f16x8 a0, a1, b0, b1;
                __builtin_memcpy(&a0, &r[u & 7], 16); __builtin_memcpy(&a1, reinterpret_cast<char *>(&r[(u + 1) & 7]) + 16, 16);
                __builtin_memcpy(&b0, &r[(u + 2) & 7], 16); __builtin_memcpy(&b1, reinterpret_cast<char *>(&r[(u + 3) & 7]) + 16, 16);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
            }
        } else if constexpr (MODE >= 7) {
            if constexpr (MODE == 7) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f16x8 a0, a1, b0, b1;
                    __builtin_memcpy(&a0, &r[u], 16); __builtin_memcpy(&a1, reinterpret_cast<char *>(&r[u + 1]) + 16, 16);
                    __builtin_memcpy(&b0, &r[u + 2], 16); __builtin_memcpy(&b1, reinterpret_cast<char *>(&r[u + 3]) + 16, 16);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < (MODE == 7 ? 1 : 3); ++u) {
                acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4 + u], r[6], acc[0], 2, 2, 0, va, 0, vb);
                acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4 + u], r[7], acc[1], 2, 2, 0, va, 0, vb);
                acc[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[5 - u / 2], r[6], acc[2], 2, 2, 0, va, 0, vb);
                acc[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[5 - u / 2], r[7], acc[3], 2, 2, 0, va, 0, vb);
            }
        } else {
            if constexpr (MODE == 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f16x8 a0, a1, b0, b1;
                    __builtin_memcpy(&a0, &r[u], 16); __builtin_memcpy(&a1, reinterpret_cast<char *>(&r[u + 1]) + 16, 16);
                    __builtin_memcpy(&b0, &r[u + 2], 16); __builtin_memcpy(&b1, reinterpret_cast<char *>(&r[u + 3]) + 16, 16);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < (MODE == 2 ? 1 : 3); ++u) {
                acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4 + u], r[6], acc[0], 1, 1, 0, va, 0, vb);
                acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4 + u], r[7], acc[1], 1, 1, 0, va, 0, vb);
                acc[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[5 - u / 2], r[6], acc[2], 1, 1, 0, va, 0, vb);
                acc[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[5 - u / 2], r[7], acc[3], 1, 1, 0, va, 0, vb);
            }
        }
    }
    if constexpr (MODE == 4 || MODE == 5 || MODE == 6) {
        f16x8 a0, a1, b0, b1;
        __builtin_memcpy(&a0, &r[0], 16); __builtin_memcpy(&a1, reinterpret_cast<char *>(&r[1]) + 16, 16);
        __builtin_memcpy(&b0, &r[2], 16); __builtin_memcpy(&b1, reinterpret_cast<char *>(&r[3]) + 16, 16);
        for (int it = 0; it < iters; ++it) {
            if constexpr (MODE == 4) {              // 3 MFMAs of a unit back to back into ONE accumulator, 4 units
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[u], 0, 0, 0);
                    acc[u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4], r[6], acc[u], 1, 1, 0, va, 0, vb);
                }
            } else if constexpr (MODE == 5) {       // skewed: h0(u), x(u-1), h1(u), accumulators alternate between TWO
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[u & 1], 0, 0, 0);
                    acc[(u + 1) & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4], r[6], acc[(u + 1) & 1], 1, 1, 0, va, 0, vb);
                    acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[u & 1], 0, 0, 0);
                }
            } else {                                // skewed, accumulators rotate over FOUR (phase 1 of the pair kernel)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[u], 0, 0, 0);
                    acc[(u + 3) & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4], r[6], acc[(u + 3) & 3], 1, 1, 0, va, 0, vb);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[u], 0, 0, 0);
                }
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
    if (t == 12345.678f) out[tid] = t;
}

template <int MODE>
static double run_rate(const uint8_t *src, float *out, int iters, const char *name, double passes_per_iter)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    rate_kernel<MODE><<<256, 512>>>(src, out, iters / 10);
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        for (int k = 0; k < 4; ++k) rate_kernel<MODE><<<256, 512>>>(src, out, iters);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms / 4 < best) best = ms / 4;
        printf("  %-34s rep %d: %.3f ms per launch\n", name, rep, ms / 4);
    }
    // one wave issues passes_per_iter MFMA passes (4 cycles each) per iteration; 2 waves per SIMD
    const double cyc = 2.0 * iters * passes_per_iter * 4.0;
    printf("%-36s %.3f ms  -> MFMA pipe busy at 2.4 GHz nominal: %.1f %%   (stages/s per CU-wave pair: %.3e)\n", name, best,
           100.0 * cyc / (best * 1e-3 * 2.4e9), iters / (best * 1e-3));
    return best;
}

int main()
{
    // ---- layout check ----
    std::vector<uint8_t> A(32 * 64), B(64 * 32);
    srand(1);
    const uint8_t vals[] = {0x00, 0x3c, 0xbc, 0x40, 0xc0, 0x38, 0xb8, 0x44, 0x34, 0xb4, 0x3e, 0xbe};   // 0, +-1, +-2, +-.5, 4, +-.25, +-1.5
    for (auto &v : A) v = vals[rand() % 12];
    for (auto &v : B) v = vals[rand() % 12];
    uint8_t *dA, *dB; float *dD;
    CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dD, 32 * 32 * 4));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    for (int trial = 0; trial < 3; ++trial) {
        const int sa = trial == 0 ? 127 : trial == 1 ? 115 : 127, sb = trial == 2 ? 130 : 127;
        layout_kernel<<<1, 64>>>(dA, dB, dD, sa, sb);
        std::vector<float> D(32 * 32);
        CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
        const double scale = ldexp(1.0, (sa - 127) + (sb - 127));
        double worst = 0, ref_max = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)bf8_to_float(A[i * 64 + k]) * bf8_to_float(B[k * 32 + j]);
                ref *= scale;
                worst = fmax(worst, fabs(ref - D[i * 32 + j]));
                ref_max = fmax(ref_max, fabs(ref));
            }
        printf("layout H1, scale_a %d scale_b %d: max |D - ref| = %g (max |ref| %g)  %s\n", sa, sb, worst, ref_max, worst == 0 ? "EXACT" : "MISMATCH");
    }

    // ---- throughput ----
    const size_t nb = 8 * 32768;
    std::vector<uint8_t> S(nb);
    for (size_t i = 0; i < nb; i += 2) {           // random f16/bf16-ish words with moderate exponents, also valid bf8 bytes
        const int e = 12 + rand() % 6;
        S[i] = rand() & 0xff;
        S[i + 1] = (uint8_t)(((rand() & 1) << 7) | (e << 2) | (rand() & 3));
    }
    uint8_t *dS; float *dO;
    CK(hipMalloc(&dS, nb)); CK(hipMalloc(&dO, 4096));
    CK(hipMemcpy(dS, S.data(), nb, hipMemcpyHostToDevice));
    const int iters = 40000;
    run_rate<0>(dS, dO, iters, "24 x bf16 32x32x16 (shipped)", 24 * 8);
    run_rate<1>(dS, dO, iters, "24 x f16 32x32x16", 24 * 8);
    run_rate<2>(dS, dO, iters, "8 x f16 + 4 x MX bf8 32x32x64", 8 * 8 + 4 * 16);
    run_rate<3>(dS, dO, iters, "12 x MX bf8 32x32x64", 12 * 16);
    run_rate<4>(dS, dO, iters, "4 x [f16,f16,MX] one acc per unit", 8 * 8 + 4 * 16);
    run_rate<5>(dS, dO, iters, "skewed, two accumulators", 8 * 8 + 4 * 16);
    run_rate<6>(dS, dO, iters, "skewed, four accumulators", 8 * 8 + 4 * 16);
    run_rate<7>(dS, dO, iters, "8 x f16 + 4 x MX fp6 32x32x64", 8 * 8 + 4 * 8);
    run_rate<8>(dS, dO, iters, "12 x MX fp6 32x32x64", 12 * 8);
    run_rate<2>(dS, dO, iters, "8 x f16 + 4 x MX bf8 (again)", 8 * 8 + 4 * 16);
    run_rate<0>(dS, dO, iters, "24 x bf16 32x32x16 (again)", 24 * 8);
    return 0;
}
