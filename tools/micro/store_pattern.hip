// Micro-benchmark (stand-alone): writing a dense [B, S, M, 64] bf16 map (278 MB at B = 32, S = 8 500, M = 8) in the two orders
// the gather-form MSDeformAttn backward can produce it:
//   mode 0  "segment major": a wave owns 16 consecutive cells of ONE (image, head): 16 stores of 128 B, 1 KB apart (6 KB in the
//           stacked layout); consecutive waves continue the same (image, head)
//   mode 1  "pixel major":   a wave owns 2 consecutive pixels, all 8 heads: 16 stores of 128 B that tile 2 KB contiguously
//   mode 2  reference: every lane 16 B, fully coalesced 1 KB per wave instruction
//   hipcc --offload-arch=gfx950 -O3 tools/micro/store_pattern.hip -o tools/micro/bin/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void fill(unsigned short* out, int B, int S, int M, long pix_el, long nwaves) {
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (wid >= nwaves) return;
    if (MODE == 0) {
        const int chunks = (S + 15) / 16;
        const int seg = (int)(wid / chunks), s0 = (int)(wid % chunks) * 16;
        const int b = seg / M, m = seg % M;
        unsigned short* dst = out + ((size_t)b * S + s0) * pix_el + (size_t)m * 64 + lane;
        for (int c = 0; c < 16 && s0 + c < S; ++c) dst[(size_t)c * pix_el] = (unsigned short)c;
    } else if (MODE == 1) {
        const int pairs = (S + 1) / 2;
        const int b = (int)(wid / pairs), s0 = (int)(wid % pairs) * 2;
        for (int c = 0; c < 2 && s0 + c < S; ++c)
            for (int m = 0; m < M; ++m) out[((size_t)b * S + s0 + c) * pix_el + (size_t)m * 64 + lane] = (unsigned short)m;
    } else {
        uint4* o = reinterpret_cast<uint4*>(out) + wid * 128 + lane;
        o[0] = make_uint4(1, 2, 3, 4);
        o[64] = make_uint4(1, 2, 3, 4);
    }
}

int main() {
    const int B = 32, S = 8500, M = 8;
    for (int layers = 1; layers <= 6; layers += 5) {
        const long pix_el = (long)layers * M * 64;
        unsigned short* buf;
        const size_t bytes = (size_t)B * S * pix_el * 2;
        hipMalloc(&buf, bytes);
        hipMemset(buf, 0, bytes);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 2 && layers > 1) continue;
            const long nwaves = mode == 0 ? (long)B * M * ((S + 15) / 16) : mode == 1 ? (long)B * ((S + 1) / 2) : (long)B * S * M * 128 / 2048;
            const unsigned grid = (unsigned)((nwaves + 3) / 4);
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(grid), dim3(256), 0, 0, buf, B, S, M, pix_el, nwaves);
                else if (mode == 1) hipLaunchKernelGGL(fill<1>, dim3(grid), dim3(256), 0, 0, buf, B, S, M, pix_el, nwaves);
                else hipLaunchKernelGGL(fill<2>, dim3(grid), dim3(256), 0, 0, buf, B, S, M, pix_el, nwaves);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (it > 0 && ms < best) best = ms;
            }
            printf("layers in the map %d  mode %d  %.1f us  %.2f TB/s of the %.0f MB slice\n", layers, mode, best * 1e3,
                   (double)B * S * M * 128 / (best * 1e-3) / 1e12, (double)B * S * M * 128 / 1e6);
        }
        hipFree(buf);
    }
    return 0;
}
