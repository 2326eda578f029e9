// Two questions before the MSDeformAttn backward is partitioned by XCD (stand-alone, no torch):
//  1. does s_getreg_b32 HW_REG_XCC_ID give the XCD a workgroup runs on, and is it blockIdx % 8?
//  2. rate of float atomics executed in the XCD's OWN L2 (workgroup-scope: no sc1 bit) on cells that only this XCD touches,
//     against the memory-side (agent-scope) atomics, same access shape as tools/micro/atomic_rate.hip
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/xcc_atomics.hip -o tools/micro/bin/xcc_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }  // hwreg(HW_REG_XCC_ID, 0, 4)

__global__ void probe(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// mode 0: agent-scope atomics, any cell.  mode 1: workgroup-scope atomics, cell drawn from the slice of the XCD the wave runs on
template <int MODE>
__global__ __launch_bounds__(256) void scatter(float* g, int rows, int updates, int cells) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const int x = xcc_id();
    const int per = cells / 8;
    for (int u = 0; u < updates; ++u) {
        uint32_t cell = hash(wave * 977 + u);
        if (MODE == 0) {
            cell %= cells;
            __hip_atomic_fetch_add(g + (size_t)cell * 64 + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            cell = x * per + cell % per;
            __hip_atomic_fetch_add(g + (size_t)cell * 64 + lane, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

__global__ void checksum(const float* g, size_t n, double* out) {
    double s = 0;
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i];
    atomicAdd(out, s);
}

int main() {
    int* d; hipMalloc(&d, 4096 * 4);
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d);
    int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int match = 0, hist[16] = {0};
    for (int i = 0; i < 4096; ++i) { match += h[i] == (i & 7); hist[h[i] & 15]++; }
    printf("XCC_ID == blockIdx %% 8 for %d of 4096 workgroups; histogram:", match);
    for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
    printf("\n");
    const int rows = 38400;
    for (int cells : {32 * 8500 * 8, 32 * 8500 * 8 / 5, 8 * 600}) {  // whole B = 32 map, the ~20 % a step touches, a contended few
        float* g; hipMalloc(&g, (size_t)cells * 64 * 4);
        double* cs; hipMalloc(&cs, 8);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemset(g, 0, (size_t)cells * 64 * 4);
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(scatter<0>, dim3(rows / 4), dim3(256), 0, 0, g, rows, 64, cells);
                else hipLaunchKernelGGL(scatter<1>, dim3(rows / 4), dim3(256), 0, 0, g, rows, 64, cells);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            hipMemset(cs, 0, 8);
            hipLaunchKernelGGL(checksum, dim3(1024), dim3(256), 0, 0, g, (size_t)cells * 64, cs);
            double hs; hipMemcpy(&hs, cs, 8, hipMemcpyDeviceToHost);
            printf("cells %8d  %s: %7.1f us  (%5.0f G channel-adds/s)  sum %.0f (expected %.0f)\n", cells,
                   mode == 0 ? "agent scope (memory side)" : "workgroup scope, own XCD  ", best * 1e3,
                   (double)rows * 64 * 64 / (best * 1e-3) / 1e9, hs, (double)rows * 64 * 64);
        }
        hipFree(g);
    }
    printf("status %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
