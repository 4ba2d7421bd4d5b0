// EXPERIMENT: what v_permlane32_swap returns (gfx950).   hipcc --offload-arch=gfx950 -O3 -o permlane_probe permlane_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *out)
{
    const unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    asm volatile("" : "+v"(a), "+v"(b));
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
    unsigned c = 300 + lane, d = c;
    asm volatile("" : "+v"(c), "+v"(d));
    const u32x2 q = __builtin_amdgcn_permlane32_swap(c, d, false, false);
    out[128 + lane] = q[0];
    out[192 + lane] = q[1];
}
int main()
{
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int v = 0; v < 4; ++v) {
        printf("%s:", v == 0 ? "swap(a=100+l, b=200+l)[0]" : v == 1 ? "swap(a, b)[1]            " : v == 2 ? "swap(c=300+l, c)[0]      " : "swap(c, c)[1]            ");
        for (int l = 0; l < 64; l += 8) printf(" l%d=%u", l, h[v * 64 + l]);
        printf("\n");
    }
    return 0;
}
