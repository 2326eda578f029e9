// What the matrix pipe sustains on this part: 256 x N workgroups of four waves run nothing but dependent-free chains of
// v_mfma_f32_16x16x32_bf16 on register operands (no memory, no LDS) for a few hundred microseconds.  Reports the achieved PFLOP/s, the
// cycles the waves counted (s_memtime) and the wall time between HIP events: the ratio is the shader clock under sustained matrix
// load, which is what every "fraction of 2.5 PFLOP/s" in DESIGN.md has to be read against.
//     hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/micro/mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int ACCS>
__global__ __launch_bounds__(256) void burn(int iters, float* out, unsigned long long* cycles) {
    v4f acc[ACCS];
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v8bf a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x & 3); b[i] = (__bf16)1.0f; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    // (inline asm with the accumulators pinned to AGPRs: through the builtin hipcc interleaves accumulator copies and s_nops)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ACCS; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACCS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 4000, accs = 16, grid = 256 * wgs_per_cu;
            burn<16><<<grid, 256>>>(100, out, cyc);  // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0);
            burn<16><<<grid, 256>>>(iters, out, cyc);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double flops = 2.0 * 16 * 16 * 32 * (double)iters * accs * 4 /*waves*/ * grid;
            printf("%d workgroup(s) of 4 waves per CU: %.3f ms, %.3f PFLOP/s; wave 0 counted %llu cycles = %.0f MHz of the counter; %d MFMAs per wave -> %.2f counter cycles per MFMA\n",
                   wgs_per_cu, ms, flops / ms / 1e12, c, c / (ms * 1e3), iters * accs, (double)c / (iters * accs));
        }
    }
    return 0;
}
