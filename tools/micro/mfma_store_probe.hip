// Can ONE CU's matrix work hide under its own HBM stores?  256 workgroups of eight waves (one per CU): waves 0-3 (one per SIMD) run a
// chain of v_mfma_f32_16x16x32_bf16 on register operands with RANDOM bit patterns (the clock under load depends on the data), waves 4-7
// stream 16-byte non-temporal row stores (1 KB per wave instruction, each wave its own 128-byte-row-aligned region of a 1 GiB buffer) from registers.
// Three launches, same code, roles switched by arguments: matrix only, stores only, both.  If both == max(matrix, stores) a kernel can be built
// whose epilogue stores run under the next tile's K loop; if both == matrix + stores no schedule will.
//     hipcc --offload-arch=gfx950 -O3 -o mfma_store_probe tools/micro/mfma_store_probe.hip && ./mfma_store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// loads: 0 = the matrix waves touch no memory; n > 0 = per 16 MFMAs they also issue n 1-KB LDS-DMA transfers (global_load_lds, 16 bytes per
// lane) from a 64 KB L2-resident panel per workgroup and wait for all but the last two groups -- the operand traffic of a real K loop
__global__ __launch_bounds__(512) void probe(int mfma_iters, int store_iters, char* buf, size_t bytes_per_wave, float* out, int loads,
                                             const char* panel) {
    __shared__ __attribute__((aligned(1024))) char lds[32768];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (mfma_iters == 0) return;
        const char* src = panel + (size_t)(blockIdx.x & 63) * 65536 + wave * 16384 + lane * 16;
        v4f acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
        unsigned s = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x;
        unsigned r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; r[i] = (s & 0x807f807fu) | 0x3f003f00u; }  // +-[0.5, 1) both halves
        const v8bf a = __builtin_bit_cast(v8bf, u32x4{r[0], r[1], r[2], r[3]}), b = __builtin_bit_cast(v8bf, u32x4{r[4], r[5], r[6], r[7]});
        for (int it = 0; it < mfma_iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            if (loads) {
                for (int k = 0; k < loads; ++k)
                    __builtin_amdgcn_global_load_lds((gptr_t)(src + ((it * loads + k) & 15) * 1024), (lptr_t)(lds + wave * 8192 + (k & 7) * 1024), 16, 0, 0);
                if (loads == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (t == 123.456f) out[0] = t;
    } else {
        if (store_iters == 0) return;
        char* base = buf + ((size_t)blockIdx.x * 4 + (wave - 4)) * bytes_per_wave + lane * 16;
        const u32x4 v = {(unsigned)lane, (unsigned)wave, 0x3f803f80u, 0x40004000u};
        const size_t span = bytes_per_wave;
        size_t off = 0;
        for (int it = 0; it < store_iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + off));
                off += 1024;
                if (off >= span) off = 0;
            }
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // at most 32 stores in flight per wave, like a tile's epilogue
        }
    }
}

int main() {
    const size_t bytes_per_wave = 1u << 20;  // 1 MiB per wave, 1 GiB in all: far beyond L2 + Infinity Cache
    char* buf; float* out;
    hipMalloc(&buf, bytes_per_wave * 1024); hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    char* panel; hipMalloc(&panel, 64 * 65536); hipMemset(panel, 0x3c, 64 * 65536);
    int loads = 0;
    auto run = [&](int mi, int si) {
        probe<<<256, 512>>>(mi / 8 + 1, si / 8 + 1, buf, bytes_per_wave, out, loads, panel);
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            probe<<<256, 512>>>(mi, si, buf, bytes_per_wave, out, loads, panel);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    for (int pass = 0; pass < 6; ++pass) {
        const int scale = 1 + (pass & 1);
        loads = pass < 2 ? 0 : (pass < 4 ? 2 : 4);
        const int mi = 3000 * scale;            // 16 MFMAs per iteration and wave
        const double flops = 2.0 * 16 * 16 * 32 * 16.0 * mi * 4 * 256;
        const float tm = run(mi, 0);
        // stores sized to take about as long as the matrix part: 8 KB per iteration and wave
        int si = 8000 * scale;
        float ts = run(0, si);
        si = (int)(si * (tm / ts));
        ts = run(0, si);
        const double bytes = 8192.0 * si * 4 * 256;
        const float tb = run(mi, si);
        printf("DMA loads per 16 MFMAs %d | matrix only %.3f ms (%.2f PFLOP/s) | stores only %.3f ms (%.2f TB/s) | both %.3f ms | max %.3f sum %.3f -> both / sum = %.2f\n",
               loads, tm, flops / tm / 1e12, ts, bytes / ts / 1e9, tb, tm > ts ? tm : ts, tm + ts, tb / (tm + ts));
    }
    return 0;
}
