// What does a VALU instruction cost beside v_mfma_f32_32x32x2_f32?  (Design question of csrc/xv_toom.hip: the transformed form
// needs ~1-4 fp32 VALU operations per MFMA.)  One workgroup per CU x 4 or 8 waves (1 or 2 per SIMD); each wave runs LOOPS
// iterations of { 4 MFMAs on 4 accumulators, each followed by N filler instructions of one kind }.  Prints cycles per MFMA.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/experiments/f32_mfma_valu_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int N>
__device__ __forceinline__ void fillers(float (&f)[8], int (&g)[8], float a)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(f[i & 7]) : "v"(a));
        if (KIND == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(f[i & 7]) : "v"(a));
        if (KIND == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(g[i & 7]) : "v"(g[(i + 1) & 7]));
        if (KIND == 3) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(g[i & 7]) : "v"(g[(i + 1) & 7]));
        if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(g[i & 7]) : "v"(g[(i + 1) & 7]));
        if (KIND == 5) asm volatile("s_nop 0");
        if (KIND == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(*reinterpret_cast<f32x4 *>(&f[(i & 1) * 4])) : "v"(g[0] & 0x3ff0) : "memory");
        if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(*reinterpret_cast<f32x2 *>(&f[(i & 3) * 2])) : "v"(*reinterpret_cast<f32x2 *>(&f[6])));
    }
}

template <int KIND, int N, bool CL>
__global__ __launch_bounds__(512, 1) void probe16(float *out, long long *cyc, int loops)
{
    f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float f[8];
    int g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 0.001f + i; g[i] = threadIdx.x + i; }
    float a = threadIdx.x * 1e-3f;
    f32x4 av = {a, a, a, a}, bv = {1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    for (int l = 0; l < loops; ++l) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(av), "v"(bv));
        if (!CL) fillers<KIND, N>(f, g, a);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(av), "v"(bv));
        if (!CL) fillers<KIND, N>(f, g, a);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(av), "v"(bv));
        if (!CL) fillers<KIND, N>(f, g, a);
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(av), "v"(bv));
        fillers<KIND, CL ? 4 * N : N>(f, g, a);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + g[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int N, bool CL>
__global__ __launch_bounds__(512, 1) void probe(float *out, long long *cyc, int loops)
{
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float f[8];
    int g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x * 0.001f + i; g[i] = threadIdx.x + i; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int l = 0; l < loops; ++l) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
        if (!CL) fillers<KIND, N>(f, g, a);
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
        if (!CL) fillers<KIND, N>(f, g, a);
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c2) : "v"(a), "v"(b));
        if (!CL) fillers<KIND, N>(f, g, a);
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c3) : "v"(a), "v"(b));
        fillers<KIND, CL ? 4 * N : N>(f, g, a);
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + g[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int N, bool CL = false, bool B16 = false>
void run(const char *name, int threads)
{
    const int blocks = 256, loops = 2000;
    float *out;
    long long *cyc;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto kern = B16 ? probe16<KIND, N, CL> : probe<KIND, N, CL>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, loops);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)loops * 4 * (threads / 256);
    // wall clock at an assumed 2.4 GHz and the achieved fraction of the fp32 MFMA peak (64 cycles per MFMA per SIMD)
    if (B16) {
        printf("bf16 16x16x32 + %-10s N=%2d %s waves/SIMD=%d: %.3f ms -> %.2f ns per MFMA per SIMD (nominal 8 passes = 16 cyc @2.4 GHz = 6.7 ns)\n", name, N,
               CL ? "clustered x4" : "spread      ", threads / 256, ms, ms * 1e6 / mfma_per_simd);
        hipFree(out); hipFree(cyc);
        return;
    }
    printf("%-10s N=%2d %s waves/SIMD=%d: %.3f ms  -> %.1f ns per MFMA per SIMD (64 cyc @2.4 GHz = 26.7 ns): MFMA pipe %.3f busy\n", name, N, CL ? "clustered x4" : "spread      ", threads / 256,
           ms, ms * 1e6 / mfma_per_simd, 26.67 / (ms * 1e6 / mfma_per_simd));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int threads : {256, 512}) {
        run<5, 0>("none", threads);
        run<0, 1>("v_fma_f32", threads);  run<0, 2>("v_fma_f32", threads);  run<0, 4>("v_fma_f32", threads);  run<0, 8>("v_fma_f32", threads);
        run<0, 1, true>("v_fma_f32", threads);  run<0, 2, true>("v_fma_f32", threads);  run<0, 4, true>("v_fma_f32", threads);  run<0, 8, true>("v_fma_f32", threads);
        run<7, 1>("v_pk_fma", threads);  run<7, 2>("v_pk_fma", threads);  run<7, 4>("v_pk_fma", threads); run<7, 4, true>("v_pk_fma", threads);
        run<6, 1>("ds_read128", threads);  run<6, 2>("ds_read128", threads);  run<6, 2, true>("ds_read128", threads);
        run<5, 8>("s_nop", threads);
    }
    for (int threads : {256, 512}) {
        run<5, 0, false, true>("none", threads);
        run<0, 1, false, true>("v_fma_f32", threads); run<0, 2, false, true>("v_fma_f32", threads); run<0, 4, false, true>("v_fma_f32", threads);
        run<0, 1, true, true>("v_fma_f32", threads); run<0, 2, true, true>("v_fma_f32", threads); run<0, 4, true, true>("v_fma_f32", threads);
        run<7, 2, true, true>("v_pk_fma", threads);
    }
    return 0;
}
