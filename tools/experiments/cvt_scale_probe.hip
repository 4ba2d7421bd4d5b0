// EXPERIMENT: does v_cvt_scalef32_pk_bf8_f32 (gfx950) equal v_cvt_pk_bf8_f32 of the pre-scaled values -- i.e. can the "* 2^11" of
// the split8 encoder ride on the conversion?   hipcc --offload-arch=gfx950 -O3 -o cvt_scale_probe cvt_scale_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, int n, uint32_t *ref, uint32_t *s_mul, uint32_t *s_div)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    ref[i] = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(a * 2048.f, b * 2048.f, 0, false) & 0xffffu;
    s16x2 z = {0, 0};
    s16x2 r1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(z, a, b, 2048.f, false);
    s16x2 r2 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(z, a, b, 1.f / 2048.f, false);
    s_mul[i] = (uint16_t)r1[0];
    s_div[i] = (uint16_t)r2[0];
}
int main()
{
    const int n = 1 << 16;
    float *hx = (float *)malloc(n * 4);
    srand(3);
    for (int i = 0; i < n; ++i) {
        const int e = rand() % 40 - 30;
        hx[i] = ldexpf((rand() / (float)RAND_MAX) * 2.f - 1.f, e);
    }
    hx[0] = 0.f; hx[1] = -0.f; hx[2] = 28.f; hx[3] = 1e9f; hx[4] = 1e-30f; hx[5] = 27.99f;
    float *dx; uint32_t *d0, *d1, *d2;
    hipMalloc(&dx, n * 4); hipMalloc(&d0, n * 2); hipMalloc(&d1, n * 2); hipMalloc(&d2, n * 2);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    k<<<n / 2 / 256, 256>>>(dx, n, d0, d1, d2);
    uint32_t *h0 = (uint32_t *)malloc(n * 2), *h1 = (uint32_t *)malloc(n * 2), *h2 = (uint32_t *)malloc(n * 2);
    hipMemcpy(h0, d0, n * 2, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 2, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 2, hipMemcpyDeviceToHost);
    int m1 = 0, m2 = 0;
    for (int i = 0; i < n / 2; ++i) { m1 += h0[i] != h1[i]; m2 += h0[i] != h2[i]; }
    printf("pairs %d: scale operand 2048 differs in %d, scale operand 1/2048 differs in %d\n", n / 2, m1, m2);
    for (int i = 0; i < 4; ++i) printf("  x = %g, %g: ref %04x  scale=2048 %04x  scale=1/2048 %04x\n", hx[2 * i], hx[2 * i + 1], h0[i], h1[i], h2[i]);
    return 0;
}
