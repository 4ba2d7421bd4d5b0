// EXPERIMENT: what a CU's store path delivers for the epilogue pattern of the 256 x 256 f16bf8 kernel (xv_gemm8.hip) against a
// row-contiguous pattern.  One workgroup (512 threads) writes a 256-row x 1 KB block of a [rows][2 KB] array, like one output
// tile; 2048 workgroups.   hipcc --offload-arch=gfx950 -O3 -o store_probe store_probe.hip && ./store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: the shipped epilogue: thread = (8-channel group cg = tid & 31, row (tid >> 5) + 16 j); two stores per row: hi slot and
//         cross slot of the group's 128-byte slab -> a wave instruction writes 2 rows x 8 slabs x 64 bytes
// MODE 1: a wave instruction writes ONE row's 1 KB: lane = 16-byte slot
// MODE 2: as 0 with non-temporal stores     MODE 3: as 1 with non-temporal stores
template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(char *y, int spin, int tiles)
{
    const int tid = threadIdx.x;
  for (int tile = 0; tile < tiles; ++tile) {
    const int bx = blockIdx.x * tiles + tile;
    const long m0 = (long)(bx >> 1) * 256;
    const int n0 = (bx & 1) * 1024;              // byte offset of the 256-column half in the 2 KB row
    i32x4 v = {tid, spin, tid * 3, 7};
    for (int k = 0; k < spin; ++k) v = v * 3 + 1;        // (a little arithmetic in front, as in the real epilogue)
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if constexpr (MODE == 0 || MODE == 2) {
            const int cg = tid & 31, slab = cg >> 2, slot = cg & 3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long gr = m0 + h * 128 + (tid >> 5) + 16 * j;
                const int sw = (int)(gr >> 1) & 7;
                char *row = y + gr * 2048 + n0 + slab * 128;
                if (MODE == 0) {
                    *reinterpret_cast<i32x4 *>(row + ((slot ^ sw) << 4)) = v;
                    *reinterpret_cast<i32x4 *>(row + (((4 + slot) ^ sw) << 4)) = v;
                } else {
                    __builtin_nontemporal_store(v, reinterpret_cast<i32x4 *>(row + ((slot ^ sw) << 4)));
                    __builtin_nontemporal_store(v, reinterpret_cast<i32x4 *>(row + (((4 + slot) ^ sw) << 4)));
                }
            }
        } else {
            const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const long gr = m0 + h * 128 + wave + 8 * j;
                char *p = y + gr * 2048 + n0 + lane * 16;
                if (MODE == 1) *reinterpret_cast<i32x4 *>(p) = v;
                else __builtin_nontemporal_store(v, reinterpret_cast<i32x4 *>(p));
            }
        }
    }
  }
}

template <int MODE>
static void run(char *y, const char *name, int grid = 2048)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    store_kernel<MODE><<<grid, 512>>>(y, 4, 2048 / grid);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(a));
        for (int k = 0; k < 4; ++k) store_kernel<MODE><<<grid, 512>>>(y, 4, 2048 / grid);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms / 4 < best) best = ms / 4;
    }
    const double bytes = 2048.0 * 256 * 1024;
    const int cus = grid < 256 ? grid : 256;
    printf("%-44s grid %4d: %.3f ms  %.2f TB/s  = %.1f B/clk per busy CU at 2.4 GHz\n", name, grid, best, bytes / best / 1e9, bytes / (best * 1e-3) / cus / 2.4e9);
}

int main()
{
    char *y;
    CK(hipMalloc(&y, (size_t)262144 * 2048 + 4096));
    run<0>(y, "2 rows x 8 slabs x 64 B per instr (shipped)");
    run<1>(y, "1 row x 1 KB per instr");
    run<2>(y, "shipped pattern, non-temporal");
    run<3>(y, "1 KB per instr, non-temporal");
    run<0>(y, "shipped (again)");
    run<0>(y, "shipped", 256);
    run<0>(y, "shipped", 128);
    run<0>(y, "shipped", 64);
    run<0>(y, "shipped", 32);
    run<1>(y, "1 KB per instr", 64);
    run<1>(y, "1 KB per instr", 32);
    return 0;
}
