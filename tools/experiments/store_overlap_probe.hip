// EXPERIMENT: does a workgroup's output stream (256 KB per tile, as in the 256 x 256 f16bf8 kernel) overlap with compute?
//   A  2048 workgroups, each: MFMA loop of ~T us, then 256 KB of stores, then it ends (the shipped structure)
//   B  256 persistent workgroups, each 8 x { MFMA loop, 256 KB of stores } -- the stores of tile i may drain under tile i+1
//   C  as A without the stores, D as B without the stores (the compute alone)
//   hipcc --offload-arch=gfx950 -O3 -o store_overlap_probe store_overlap_probe.hip && ./store_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void tile(char *y, int bx, int iters, bool stores, float *sink)
{
    const int tid = threadIdx.x;
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(tid * 0.001f + i); b[i] = (_Float16)(1.f + i * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u], 0, 0, 0);
    }
    i32x4 v = {tid, (int)acc[0][0], (int)acc[1][1], (int)acc[2][2] + (int)acc[3][3]};
    if (stores) {
        const long m0 = (long)(bx >> 1) * 256;
        const int n0 = (bx & 1) * 1024;
        const int cg = tid & 31, slab = cg >> 2, slot = cg & 3;
#pragma unroll 1
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long gr = m0 + h * 128 + (tid >> 5) + 16 * j;
                const int sw = (int)(gr >> 1) & 7;
                char *row = y + gr * 2048 + n0 + slab * 128;
                *reinterpret_cast<i32x4 *>(row + ((slot ^ sw) << 4)) = v;
                *reinterpret_cast<i32x4 *>(row + (((4 + slot) ^ sw) << 4)) = v;
            }
    } else if (v[1] == 123456789) sink[tid] = 1.f;
}

__global__ __launch_bounds__(512) void k_oneshot(char *y, int iters, int stores, float *sink) { tile(y, blockIdx.x, iters, stores, sink); }
__global__ __launch_bounds__(512) void k_persistent(char *y, int iters, int stores, float *sink)
{
    for (int t = 0; t < 8; ++t) tile(y, blockIdx.x + 256 * t, iters, stores, sink);
}

static float timeit(bool persistent, char *y, int iters, int stores, float *sink)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        if (persistent) k_persistent<<<256, 512>>>(y, iters, stores, sink); else k_oneshot<<<2048, 512>>>(y, iters, stores, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main()
{
    char *y; float *sink;
    CK(hipMalloc(&y, (size_t)262144 * 2048 + 4096)); CK(hipMalloc(&sink, 4096));
    for (int iters : {200, 800, 1600}) {          // MFMAs per wave: 4 * iters; 2 waves per SIMD
        const float A = timeit(false, y, iters, 1, sink), B = timeit(true, y, iters, 1, sink);
        const float C = timeit(false, y, iters, 0, sink), D = timeit(true, y, iters, 0, sink);
        printf("iters %4d: one-shot %.3f ms (compute alone %.3f -> stores cost %.3f)   persistent %.3f ms (compute alone %.3f -> stores cost %.3f)\n",
               iters, A, C, A - C, B, D, B - D);
    }
    return 0;
}
