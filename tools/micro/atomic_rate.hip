// Micro-benchmark (stand-alone, no torch): chip-wide rate of memory-side float atomics vs packed-bf16 atomics with the
// access shape of the MSDeformAttn backward (a wave adds one contiguous pixel-head row at a pseudo-random pixel).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_rate.hip -o /tmp/atomic_rate && /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }

// mode 0: 64 lanes x f32 (256 B per update); mode 1: 32 lanes x packed bf16x2 (128 B per update), two updates per wave-instr;
// mode 2: 32 lanes x u64 integer add (two 32-bit fixed-point channels per lane, 256 B per update), two updates per wave-instr;
// mode 3: 64 lanes x u32 integer add (256 B per update)
template <int MODE>
__global__ __launch_bounds__(256) void scatter(float* gf, uint32_t* gh, int rows, int updates, int cells) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    for (int u = 0; u < updates; ++u) {
        if (MODE == 0) {
            const uint32_t cell = hash(wave * 977 + u) % cells;
            atomicAdd(gf + (size_t)cell * 64 + lane, 1.0f);
        } else if (MODE == 2) {
            const uint32_t cell = hash(wave * 977 + 2 * u + (lane >> 5)) % cells;
            __hip_atomic_fetch_add((unsigned long long*)gf + (size_t)cell * 32 + (lane & 31), 0x0000000100000001ull,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 3) {
            const uint32_t cell = hash(wave * 977 + u) % cells;
            __hip_atomic_fetch_add((unsigned int*)gf + (size_t)cell * 64 + lane, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const uint32_t cell = hash(wave * 977 + 2 * u + (lane >> 5)) % cells;
            typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
            v2bf v = {(__bf16)1.0f, (__bf16)1.0f};
            __builtin_amdgcn_global_atomic_fadd_v2bf16((v2bf __attribute__((address_space(1)))*)(gh + (size_t)cell * 32 + (lane & 31)), v);
        }
    }
}

int main() {
    const int rows = 38400, cells = 32 * 8500 * 8;  // B = 32 geometry
    float* gf; uint32_t* gh;
    hipMalloc(&gf, (size_t)cells * 64 * 4); hipMalloc(&gh, (size_t)cells * 32 * 4);
    hipMemset(gf, 0, (size_t)cells * 64 * 4); hipMemset(gh, 0, (size_t)cells * 32 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[4] = {"f32 atomics        ", "packed bf16 atomics", "u64 int atomics    ", "u32 int atomics    "};
    for (int mode = 0; mode < 4; ++mode) {
        const int updates = (mode == 0 || mode == 3) ? 64 : 32;  // same number of (pixel, head) row updates: 64 per row
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(scatter<0>, dim3(rows / 4), dim3(256), 0, 0, gf, gh, rows, updates, cells);
            else if (mode == 1) hipLaunchKernelGGL(scatter<1>, dim3(rows / 4), dim3(256), 0, 0, gf, gh, rows, updates, cells);
            else if (mode == 2) hipLaunchKernelGGL(scatter<2>, dim3(rows / 4), dim3(256), 0, 0, gf, gh, rows, updates, cells);
            else hipLaunchKernelGGL(scatter<3>, dim3(rows / 4), dim3(256), 0, 0, gf, gh, rows, updates, cells);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep == 2) printf("%s: %.1f us for %d row updates (%.0f G channel-adds/s)\n", names[mode],
                                 ms * 1e3, rows * 64, (double)rows * 64 * 64 / (ms * 1e-3) / 1e9);
        }
    }
    printf("status %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
