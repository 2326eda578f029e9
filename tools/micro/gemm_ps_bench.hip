// Stand-alone check + timing of the persistent stream GEMM (tools/micro/gemm_ps.hip) next to the per-tile and ping-pong kernels of
// gemm.hip, no torch:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGRIT_GEMM_STAMPS tools/micro/gemm_ps_bench.hip -o tools/micro/bin/gemm_ps_bench
//   tools/micro/bin/gemm_ps_bench            (all Swin shapes: correctness on sampled rows against a naive fp32 kernel, us per launch,
//                                             per-phase s_memtime sums of the stream kernel)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/grit_hip.h"
#include "../../grit_amd/csrc/gemm_math.h"
#include "../../grit_amd/csrc/gemm_launchers.h"
namespace v45 {
#include "../../grit_amd/csrc/gemm.hip"
}
#include "gemm_ps.hip"
#include "../../grit_amd/csrc/gemm_w4.hip"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void ref_rows(const __bf16* A, const __bf16* B, const __bf16* bias, const int* rows, int nrows, int N, int K, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
    if (n >= N || r >= nrows) return;
    const __bf16* a = A + (size_t)rows[r] * K;
    const __bf16* b = B + (size_t)n * K;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)a[k] * (float)b[k];
    out[(size_t)r * N + n] = s + (bias ? (float)bias[n] : 0.f);
}

static float bf2f(unsigned short h) { union { unsigned u; float f; } c; c.u = (unsigned)h << 16; return c.f; }

int main(int argc, char** argv) {
    struct Shape { int M, N, K; };
    std::vector<Shape> shapes = {{51200, 2048, 512}, {51200, 1536, 512}, {51200, 512, 512}, {51200, 512, 2048}, {51200, 512, 1536},
                                 {204800, 1024, 256}, {204800, 768, 256}, {12800, 4096, 1024}, {12800, 1024, 4096},
                                 {51100, 2048, 512}, {4800, 512, 512}};
    if (argc > 3) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}};
    const int iters = 20;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long* stamps; hipMalloc(&stamps, (size_t)256 * 8 * 16 * 8);
    for (const Shape& sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K;
        std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K), hbias(N);
        srand(M + N + K);
        // full-range sign, values ~ +-[0.5, 1) / sqrt-ish scale so that sums stay O(1)
        for (auto& v : ha) v = (unsigned short)(0x3f00 + (rand() & 0x7f) + ((rand() & 1) << 15));
        for (auto& v : hb) v = (unsigned short)(0x3c00 + (rand() & 0x7f) + ((rand() & 1) << 15));
        for (auto& v : hbias) v = (unsigned short)(0x3e00 + (rand() & 0x7f) + ((rand() & 1) << 15));
        void *A, *B, *C, *C2, *aux, *bias; float* ref; int* drows;
        hipMalloc(&A, ha.size() * 2); hipMalloc(&B, hb.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&C2, (size_t)M * N * 2);
        hipMalloc(&aux, (size_t)M * N * 2); hipMalloc(&bias, N * 2);
        hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(bias, hbias.data(), N * 2, hipMemcpyHostToDevice);
        // sampled rows: first and last tile rows, a few in the middle
        std::vector<int> rows;
        for (int r = 0; r < 256 && r < M; ++r) rows.push_back(r);
        for (int r = std::max(0, M - 300); r < M; ++r) rows.push_back(r);
        for (int t = 0; t < 256; ++t) rows.push_back((int)(((long long)rand() * 7919 + t * 104729LL) % M));
        const int nrows = (int)rows.size();
        hipMalloc(&drows, nrows * 4); hipMemcpy(drows, rows.data(), nrows * 4, hipMemcpyHostToDevice);
        hipMalloc(&ref, (size_t)nrows * N * 4);
        ref_rows<<<dim3((N + 255) / 256, nrows), 256>>>((const __bf16*)A, (const __bf16*)B, (const __bf16*)bias, drows, nrows, N, K, ref);
        std::vector<float> href((size_t)nrows * N);
        hipMemcpy(href.data(), ref, href.size() * 4, hipMemcpyDeviceToHost);

        auto ps = [&](int epi, void* out, unsigned long long* st) {
            return grit_detail::gemm_ps_launch(A, K, B, K, out, N, M, N, K, epi, bias, epi == GRIT_GEMM_BIAS_GELU ? aux : nullptr, N,
                                               epi == GRIT_GEMM_BIAS_GELU ? 3 : 0, 0, st);
        };
        auto old = [&](int variant, int epi, void* out) {
            return v45::grit_gemm_bf16_nt(A, K, B, K, out, N, M, N, K, epi, bias, epi == GRIT_GEMM_BIAS_GELU ? aux : nullptr, N, nullptr,
                                          variant, 0);
        };
        auto check = [&](void* out, bool gelu, const char* what) {
            std::vector<unsigned short> h((size_t)M * N);
            hipMemcpy(h.data(), out, h.size() * 2, hipMemcpyDeviceToHost);
            double worst = 0, scale = 0;
            long bad = 0;
            for (int r = 0; r < nrows; ++r)
                for (int n = 0; n < N; ++n) {
                    float want = href[(size_t)r * N + n];
                    if (gelu) {
                        // GELU of the bf16-rounded pre-activation (what the stream kernel computes); the per-tile kernels take the
                        // fp32 value: differences of one bf16 ulp of the input are inside the tolerance below
                        want = 0.5f * want * (1.f + erff(want * 0.70710678f));
                    }
                    const float got = bf2f(h[(size_t)rows[r] * N + n]);
                    const double err = fabs((double)got - want), tol = 0.02 * fabs(want) + 0.02;
                    worst = std::max(worst, err); scale = std::max(scale, (double)fabs(want));
                    if (!(err <= tol)) ++bad;
                }
            printf("    %-28s max err %.4f (scale %.2f) bad %ld of %ld\n", what, worst, scale, bad, (long)nrows * N);
            return bad == 0;
        };
        auto time_it = [&](auto fn) {
            for (int i = 0; i < 3; ++i) fn();
            hipEventRecord(e0); for (int i = 0; i < iters; ++i) fn(); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms * 1000.f / iters;
        };
        printf("M %d N %d K %d  (%.1f GFLOP)\n", M, N, K, 2.0 * M * N * K * 1e-9);
        const bool can_ps = N % 256 == 0 && M >= 256;
        if (can_ps) {
            hipMemset(C, 0xff, (size_t)M * N * 2);
            int st = ps(GRIT_GEMM_BIAS, C, nullptr); hipDeviceSynchronize();
            if (st != 0 || hipGetLastError() != hipSuccess) { printf("    stream kernel launch failed %d\n", st); continue; }
            check(C, false, "stream bias");
            hipMemset(C2, 0xff, (size_t)M * N * 2); hipMemset(aux, 0xff, (size_t)M * N * 2);
            ps(GRIT_GEMM_BIAS_GELU, C2, nullptr); hipDeviceSynchronize();
            check(aux, false, "stream gelu: pre-activation");
            check(C2, true, "stream gelu: activation");
            // bit-for-bit against the per-tile kernel on the bias epilogue (same products, same k order per accumulator)
            old(4, GRIT_GEMM_BIAS, C2); hipDeviceSynchronize();
            std::vector<unsigned short> h1((size_t)M * N), h2((size_t)M * N);
            hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), C2, h2.size() * 2, hipMemcpyDeviceToHost);
            long diff = 0;
            for (size_t i = 0; i < h1.size(); ++i) diff += h1[i] != h2[i];
            printf("    stream vs per-tile kernel (bias): %ld of %zu elements differ\n", diff, h1.size());
        }
        const double gf = 2.0 * M * N * K * 1e-9;
        float t;
        const bool can_w4 = N % 256 == 0 && K % 64 == 0 && M >= 256;
        float* cs_w4; float* cs_v4;
        const int slabs_w4 = 2 * ((M + 255) / 256), slabs_v4 = (M + 127) / 128;
        hipMalloc(&cs_w4, (size_t)slabs_w4 * N * 4); hipMalloc(&cs_v4, (size_t)slabs_v4 * N * 4);
        auto w4 = [&](int epi, void* out, void* auxp) {
            return grit_detail::gemm_w4_launch(A, K, B, K, out, N, M, N, K, epi, bias, auxp, N, cs_w4, epi == GRIT_GEMM_BIAS ? 0 : 15, 0);
        };
        if (can_w4) {
            hipMemset(C, 0xff, (size_t)M * N * 2);
            int st = w4(GRIT_GEMM_BIAS, C, nullptr); hipDeviceSynchronize();
            if (st != 0 || hipGetLastError() != hipSuccess) printf("    w4 launch failed %d\n", st);
            else {
                check(C, false, "w4 bias");
                old(4, GRIT_GEMM_BIAS, C2); hipDeviceSynchronize();
                std::vector<unsigned short> h1((size_t)M * N), h2((size_t)M * N);
                hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), C2, h2.size() * 2, hipMemcpyDeviceToHost);
                long diff = 0;
                for (size_t i = 0; i < h1.size(); ++i) diff += h1[i] != h2[i];
                printf("    w4 vs per-tile kernel (bias): %ld of %zu elements differ\n", diff, h1.size());
                // GELU pair
                hipMemset(C2, 0xff, (size_t)M * N * 2); hipMemset(aux, 0xff, (size_t)M * N * 2);
                w4(GRIT_GEMM_BIAS_GELU, C2, aux); hipDeviceSynchronize();
                check(aux, false, "w4 gelu: pre-activation");
                check(C2, true, "w4 gelu: activation");
                // GELU' x accumulator + column sums against the per-tile kernel (aux = the pre-activation just written)
                hipMemset(C, 0xff, (size_t)M * N * 2);
                w4(GRIT_GEMM_DGELU, C, aux); hipDeviceSynchronize();
                v45::grit_gemm_bf16_nt(A, K, B, K, C2, N, M, N, K, GRIT_GEMM_DGELU, nullptr, aux, N, cs_v4, 4, 0); hipDeviceSynchronize();
                hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), C2, h2.size() * 2, hipMemcpyDeviceToHost);
                double worst = 0; long bad = 0;
                for (size_t i = 0; i < h1.size(); ++i) {
                    const double a = bf2f(h1[i]), b = bf2f(h2[i]), e = fabs(a - b);
                    worst = std::max(worst, e);
                    if (!(e <= 0.02 * fabs(b) + 0.01)) ++bad;
                }
                std::vector<float> ca((size_t)slabs_w4 * N), cb((size_t)slabs_v4 * N);
                hipMemcpy(ca.data(), cs_w4, ca.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(cb.data(), cs_v4, cb.size() * 4, hipMemcpyDeviceToHost);
                double cworst = 0, cscale = 0;
                for (int n = 0; n < N; ++n) {
                    double sa = 0, sb = 0;
                    for (int r = 0; r < slabs_w4; ++r) sa += ca[(size_t)r * N + n];
                    for (int r = 0; r < slabs_v4; ++r) sb += cb[(size_t)r * N + n];
                    cworst = std::max(cworst, fabs(sa - sb)); cscale = std::max(cscale, fabs(sb));
                }
                printf("    w4 dgelu vs per-tile kernel: max err %.4f, bad %ld of %zu; column sums max err %.4f (scale %.1f)\n", worst, bad,
                       h1.size(), cworst, cscale);
                t = time_it([&] { w4(GRIT_GEMM_BIAS, C, nullptr); }); printf("    w4 bias        %7.1f us  %.2f PF\n", t, gf / t * 1e-6);
                t = time_it([&] { w4(GRIT_GEMM_BIAS_GELU, C, C2); }); printf("    w4 gelu+aux    %7.1f us\n", t);
                t = time_it([&] { w4(GRIT_GEMM_DGELU, C, aux); }); printf("    w4 dgelu       %7.1f us\n", t);
                t = time_it([&] { v45::grit_gemm_bf16_nt(A, K, B, K, C2, N, M, N, K, GRIT_GEMM_DGELU, nullptr, aux, N, cs_v4, 4, 0); });
                printf("    v4 dgelu       %7.1f us\n", t);
            }
        }
        hipFree(cs_w4); hipFree(cs_v4);
        if (N % 256 == 0 && K % 64 == 0) { t = time_it([&] { old(4, GRIT_GEMM_BIAS, C); }); printf("    v4 bias        %7.1f us  %.2f PF\n", t, gf / t * 1e-6); }
        if (N % 256 == 0) { t = time_it([&] { old(5, GRIT_GEMM_BIAS, C); }); printf("    v5 bias        %7.1f us  %.2f PF\n", t, gf / t * 1e-6); }
        if (can_ps) { t = time_it([&] { ps(GRIT_GEMM_BIAS, C, nullptr); }); printf("    stream bias    %7.1f us  %.2f PF\n", t, gf / t * 1e-6); }
        if (N % 256 == 0 && K % 64 == 0) { t = time_it([&] { old(4, GRIT_GEMM_BIAS_GELU, C); }); printf("    v4 gelu+aux    %7.1f us\n", t); }
        if (can_ps) { t = time_it([&] { ps(GRIT_GEMM_BIAS_GELU, C, nullptr); }); printf("    stream gelu+aux%7.1f us\n", t); }
#ifdef GRIT_GEMM_STAMPS
        if (N % 256 == 0) {  // the ping-pong kernel's phases on the same box (its epilogue is the fifth figure, per K step)
            hipMemset(stamps, 0, (size_t)256 * 8 * 16 * 8);
            v45::GemmArgs a;
            a.row_scale = nullptr; a.rows_per_sample = 0;
            a.A = (const __bf16*)A; a.lda = K; a.B = (const __bf16*)B; a.ldb = K; a.C = (__bf16*)C; a.ldc = N; a.bias = (const __bf16*)bias;
            a.aux = nullptr; a.ldaux = N; a.colsum = nullptr; a.M = M; a.N = N; a.K = K; a.tiles_m = a.tiles_n = 0; a.nt_aux = 0;
            a.stamps = stamps;
            v45::launch_pp(a, GRIT_GEMM_BIAS, 0); hipDeviceSynchronize();
            std::vector<unsigned long long> h((size_t)256 * 8 * 16);
            hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
            for (int grp = 0; grp < 2; ++grp) {
                printf("    v5 waves %d-%d, cycles per K step:", 4 * grp, 4 * grp + 3);
                for (int s = 0; s < 5; ++s) {
                    std::vector<double> d;
                    for (int wg = 0; wg < 256; ++wg)
                        for (int w = 4 * grp; w < 4 * grp + 4; ++w) {
                            const unsigned long long* o = &h[((size_t)wg * 8 + w) * 16];
                            if (o[13]) d.push_back((double)o[8 + s] / (double)o[13]);
                        }
                    if (d.empty()) continue;
                    std::sort(d.begin(), d.end());
                    printf("  %s %.0f", s == 0 ? "L" : s == 1 ? "b1" : s == 2 ? "M" : s == 3 ? "b2" : "epi", d[d.size() / 2]);
                }
                printf("\n");
            }
        }
#endif
        if (can_ps) {
            hipMemset(stamps, 0, (size_t)256 * 8 * 16 * 8);
            ps(GRIT_GEMM_BIAS, C, stamps); hipDeviceSynchronize();
            std::vector<unsigned long long> h((size_t)256 * 8 * 16);
            hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
            const char* pn[] = {"L (DMA issue + fragment reads)", "barrier after L", "M (MFMAs + drain + wait)", "barrier after M", "tile end (pack)"};
            for (int grp = 0; grp < 2; ++grp) {
                printf("    waves %d-%d, cycles per K step:", 4 * grp, 4 * grp + 3);
                for (int s = 0; s < 5; ++s) {
                    std::vector<double> d;
                    for (int wg = 0; wg < 256; ++wg)
                        for (int w = 4 * grp; w < 4 * grp + 4; ++w) {
                            const unsigned long long* o = &h[((size_t)wg * 8 + w) * 16];
                            if (o[13]) d.push_back((double)o[8 + s] / (double)o[13]);
                        }
                    if (d.empty()) continue;
                    std::sort(d.begin(), d.end());
                    printf("  %s %.0f", s == 0 ? "L" : s == 1 ? "b1" : s == 2 ? "M" : s == 3 ? "b2" : "out", d[d.size() / 2]);
                }
                printf("\n");
            }
            (void)pn;
        }
        hipFree(A); hipFree(B); hipFree(C); hipFree(C2); hipFree(aux); hipFree(bias); hipFree(ref); hipFree(drows);
    }
    return 0;
}
