// EXPERIMENT: issue cost of the conversion instructions the split8 encoder uses, against plain VALU (gfx950).  One wave per SIMD runs a
// dependent-free loop of N copies of one instruction; cycles per instruction = s_memtime delta / count.
//   hipcc --offload-arch=gfx950 -O3 -o cvt_rate_probe cvt_rate_probe.hip && ./cvt_rate_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP8(x) x x x x x x x x
#define BODY(INS) REP8(REP8(INS))
#define KERNEL(NAME, INS)                                                                                              \
    __global__ void NAME(unsigned long long *out, float seed)                                                         \
    {                                                                                                                  \
        float a = seed + threadIdx.x, b = seed * 0.5f, c = 1.5f;                                                       \
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;                                                                            \
        unsigned long long t0 = __builtin_readcyclecounter();                                                          \
        for (int i = 0; i < 64; ++i) asm volatile(BODY(INS) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c)); \
        unsigned long long t1 = __builtin_readcyclecounter();                                                          \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                               \
        if (r0 + r1 + r2 + r3 == 12345) out[1] = 0;                                                                     \
    }
KERNEL(k_add, "v_add_f32 %0, %4, %5\n v_add_f32 %1, %4, %6\n v_add_f32 %2, %5, %6\n v_add_f32 %3, %4, %4\n")
KERNEL(k_med3, "v_med3_f32 %0, %4, %5, %6\n v_med3_f32 %1, %4, %6, %5\n v_med3_f32 %2, %5, %6, %4\n v_med3_f32 %3, %4, %4, %6\n")
KERNEL(k_cvt_f16, "v_cvt_f16_f32 %0, %4\n v_cvt_f16_f32 %1, %5\n v_cvt_f16_f32 %2, %6\n v_cvt_f16_f32 %3, %4\n")
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %4\n v_cvt_f32_f16 %1, %5\n v_cvt_f32_f16 %2, %6\n v_cvt_f32_f16 %3, %4\n")
KERNEL(k_cvt_pk_f16, "v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %6\n v_cvt_pk_f16_f32 %2, %6, %4\n v_cvt_pk_f16_f32 %3, %4, %4\n")
KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %4, %5\n v_cvt_pkrtz_f16_f32 %1, %5, %6\n v_cvt_pkrtz_f16_f32 %2, %6, %4\n v_cvt_pkrtz_f16_f32 %3, %4, %4\n")
KERNEL(k_cvt_pk_bf8, "v_cvt_pk_bf8_f32 %0, %4, %5\n v_cvt_pk_bf8_f32 %1, %5, %6\n v_cvt_pk_bf8_f32 %2, %6, %4\n v_cvt_pk_bf8_f32 %3, %4, %4\n")
KERNEL(k_cvt_scale_bf8, "v_cvt_scalef32_pk_bf8_f32 %0, %4, %5, %6\n v_cvt_scalef32_pk_bf8_f32 %1, %5, %6, %6\n v_cvt_scalef32_pk_bf8_f32 %2, %6, %4, %6\n v_cvt_scalef32_pk_bf8_f32 %3, %4, %4, %6\n")
KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %4, %5, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %5, %6, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %6, %4, %5 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %4, %6 op_sel_hi:[1,0,0]\n")
KERNEL(k_and_add3, "v_and_b32 %0, %4, %5\n v_add3_u32 %1, %4, %5, %6\n v_bfe_u32 %2, %4, 13, 1\n v_and_b32 %3, %5, %6\n")
int main()
{
    unsigned long long *d, h[4];
    hipMalloc(&d, sizeof(h));
#define RUN(NAME)                                                                                  \
    for (int waves = 1; waves <= 2; ++waves) {                                                     \
        hipMemset(d, 0, sizeof(h));                                                                \
        NAME<<<1, 256 * waves>>>(d, 1.25f);                                                        \
        NAME<<<1, 256 * waves>>>(d, 1.25f);                                                        \
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);                                         \
        printf("%-18s %d wave(s)/SIMD: %.2f cycles per instruction and wave (counter ticks / %d)\n", #NAME, waves, (double)h[0] / (64.0 * 64 * 4), 64 * 64 * 4); \
    }
    RUN(k_add) RUN(k_med3) RUN(k_cvt_f16) RUN(k_cvt_f32_f16) RUN(k_cvt_pk_f16) RUN(k_cvt_pkrtz) RUN(k_cvt_pk_bf8) RUN(k_cvt_scale_bf8) RUN(k_fma_mix) RUN(k_and_add3)
    return 0;
}
