// How fast can the chip absorb the GEMM's output alone?  Every workgroup writes 256 x 256 bf16 tiles of a [51200, 2048] map in
// the epilogue's shape (8 rows x 128 B per wave instruction), nothing else.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void store_tiles(uint4* out, int tiles_n, int ntiles, long ld16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        const long row0 = (long)tm * 256 + (wave >> 2) * 128, col16 = (long)tn * 32 + (wave & 3) * 8;
        const uint4 v = make_uint4(t, lane, wave, 1);
#pragma unroll
        for (int it = 0; it < 16; ++it) out[(row0 + it * 8 + (lane >> 3)) * ld16 + col16 + (lane & 7)] = v;
    }
}
int main() {
    const int M = 51200, N = 2048;
    uint4* out; hipMalloc(&out, (size_t)M * N * 2);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {256, 512, 1600}) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_tiles, dim3(grid), dim3(512), 0, 0, out, N / 256, (M / 256) * (N / 256), (long)N / 8);
        hipEventRecord(a);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_tiles, dim3(grid), dim3(512), 0, 0, out, N / 256, (M / 256) * (N / 256), (long)N / 8);
        hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("grid %4d: %.1f us per 210 MB = %.2f TB/s\n", grid, ms * 50, (double)M * N * 2 / (ms / 20 * 1e-3) / 1e12);
    }
    return 0;
}
