// EXPERIMENT: does the MFMA tile shape change the ENERGY of the f16 + bf8 product (the chip is power-limited in this mix, so the
// time of a register-only loop on random operands is a proxy for joules per product)?  Per iteration and wave, equal FLOPs:
//   mode 0:  8 x f16 32x32x16 + 4 x MX bf8 32x32x64      (the shipped stage of the 256 x 256 kernel)
//   mode 1: 16 x f16 16x16x32 + 8 x MX bf8 16x16x128     (quarter-size accumulators, twice the operand registers per FLOP)
//   mode 2 / 3: the f16 halves alone;  mode 4 / 5: the MX halves alone
//   mode 6: as 5 with K blocks 2 and 3 of both operands zero -- what a 16x16x128 instruction costs when only one (tap, slab) item
//           fills it (the padding an odd number of taps needs);  mode 7: as 5 with only the A operand's blocks 2, 3 zero
//   hipcc --offload-arch=gfx950 -O3 -o shape_probe shape_probe.hip && ./shape_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void rate_kernel(const uint8_t *src, float *out, int iters)
{
    const int tid = threadIdx.x;
    const uint8_t *s = src + (size_t)(tid & 63) * 512 + (tid >> 6) * 32768;
    i32x8 r[8];
    for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const i32x8 *>(s + 32 * j);
    f32x16 big[4] = {{0}, {0}, {0}, {0}};
    f32x4 sm[16];
    for (int i = 0; i < 16; ++i) sm[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int va = 115, vb = 127;
    asm volatile("" : "+v"(va), "+v"(vb));
    f16x8 h[8];
    for (int j = 0; j < 8; ++j) __builtin_memcpy(&h[j], reinterpret_cast<char *>(&r[j]) + (j & 1) * 16, 16);
    // operands whose K blocks 2, 3 (lanes 32..63 of a 16x16x128 operand) are zero
    i32x8 z[4];
    for (int j = 0; j < 4; ++j) z[j] = (tid & 32) ? (i32x8){0, 0, 0, 0, 0, 0, 0, 0} : r[4 + j];
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                big[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[u], h[u + 2], big[0], 0, 0, 0);
                big[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[u], h[u + 3], big[1], 0, 0, 0);
                big[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[u + 1], h[u + 2], big[2], 0, 0, 0);
                big[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h[u + 1], h[u + 3], big[3], 0, 0, 0);
            }
        }
        if constexpr (MODE == 0 || MODE == 4) {
            big[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4], r[6], big[0], 1, 1, 0, va, 0, vb);
            big[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[4], r[7], big[1], 1, 1, 0, va, 0, vb);
            big[2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[5], r[6], big[2], 1, 1, 0, va, 0, vb);
            big[3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(r[5], r[7], big[3], 1, 1, 0, va, 0, vb);
        }
        if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
                sm[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[u & 3], h[4 + (u >> 2)], sm[u], 0, 0, 0);
        }
        if constexpr (MODE == 1 || MODE == 5) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                sm[u + 4] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(r[4 + (u & 1)], r[6 + ((u >> 1) & 1)], sm[u + 4], 1, 1, 0, va, 0, vb);
        }
        if constexpr (MODE == 6 || MODE == 7) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                sm[u + 4] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(z[u & 1], MODE == 6 ? z[2 + ((u >> 1) & 1)] : r[6 + ((u >> 1) & 1)], sm[u + 4], 1, 1, 0, va, 0, vb);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += big[0][i] + big[1][i] + big[2][i] + big[3][i] + sm[i][0] + sm[i][1] + sm[i][2] + sm[i][3];
    if (t == 12345.678f) out[tid] = t;
}

template <int MODE>
static double run_rate(const uint8_t *src, float *out, int iters, const char *name)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    rate_kernel<MODE><<<256, 512>>>(src, out, iters / 10);
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a));
        for (int k = 0; k < 4; ++k) rate_kernel<MODE><<<256, 512>>>(src, out, iters);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms / 4 < best) best = ms / 4;
    }
    printf("%-44s %.3f ms\n", name, best);
    return best;
}

int main(int argc, char **argv)
{
    const bool zero = argc > 1;
    const size_t nb = 8 * 32768;
    std::vector<uint8_t> S(nb);
    srand(1);
    for (size_t i = 0; i < nb; i += 2) {
        const int e = 12 + rand() % 6;
        S[i] = zero ? 0 : rand() & 0xff;
        S[i + 1] = zero ? 0 : (uint8_t)(((rand() & 1) << 7) | (e << 2) | (rand() & 3));
    }
    uint8_t *dS; float *dO;
    CK(hipMalloc(&dS, nb)); CK(hipMalloc(&dO, 4096));
    CK(hipMemcpy(dS, S.data(), nb, hipMemcpyHostToDevice));
    const int iters = 40000;
    printf("%s operands\n", zero ? "all-zero" : "random");
    for (int rnd = 0; rnd < 2; ++rnd) {
        run_rate<0>(dS, dO, iters, "8 x f16 32x32x16 + 4 x MX 32x32x64");
        run_rate<1>(dS, dO, iters, "16 x f16 16x16x32 + 8 x MX 16x16x128");
        run_rate<2>(dS, dO, iters, "8 x f16 32x32x16");
        run_rate<3>(dS, dO, iters, "16 x f16 16x16x32");
        run_rate<4>(dS, dO, iters, "4 x MX 32x32x64");
        run_rate<5>(dS, dO, iters, "8 x MX 16x16x128");
        run_rate<6>(dS, dO, iters, "8 x MX 16x16x128, K blocks 2,3 zero (A, B)");
        run_rate<7>(dS, dO, iters, "8 x MX 16x16x128, K blocks 2,3 zero (A)");
    }
    return 0;
}
