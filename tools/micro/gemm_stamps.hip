// Diagnostic build of grit_amd/csrc/gemm.hip with s_memtime stamps: where a workgroup of the GEMM spends its cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGRIT_GEMM_STAMPS tools/micro/gemm_stamps.hip -o /tmp/gemm_stamps && /tmp/gemm_stamps [variant] [epilogue]
// Stamps (per wave): 0 start, 1 prologue DMAs issued, 2/3/4 past the barrier of K tile 0/1/2, 5 main loop done, 6 past the
// epilogue barrier, 7 end.  Prints medians over all waves of the cycle differences.
#include "../../grit_amd/csrc/gemm.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 4, epi = argc > 2 ? atoi(argv[2]) : 1;
    const int M = 51200, N = 2048, K = 512;
    std::vector<unsigned short> ha((size_t)M * K), hb((size_t)N * K), hbias(N);
    srand(1);
    for (auto& v : ha) v = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));  // ~ +-[0.008, 0.03]
    for (auto& v : hb) v = (unsigned short)(0x3c00 + (rand() & 0x1ff) + ((rand() & 1) << 15));
    for (auto& v : hbias) v = 0x3c00;
    void *A, *B, *C, *bias, *aux; float* cs; unsigned long long* stamps;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&B, hb.size() * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&aux, (size_t)M * N * 2);
    hipMalloc(&bias, N * 2); hipMalloc(&cs, (size_t)(M / 128) * N * 4);
    const int waves_total = 3200 * 8;
    hipMalloc(&stamps, (size_t)waves_total * 16 * 8);
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(bias, hbias.data(), N * 2, hipMemcpyHostToDevice); hipMemset(aux, 0, (size_t)M * N * 2);
    GemmArgs a;
    a.row_scale = nullptr; a.rows_per_sample = 0;
    a.A = (const __bf16*)A; a.lda = K; a.B = (const __bf16*)B; a.ldb = K; a.C = (__bf16*)C; a.ldc = N; a.bias = (const __bf16*)bias;
    a.aux = (__bf16*)aux; a.ldaux = N; a.colsum = cs; a.M = M; a.N = N; a.K = K; a.tiles_m = a.tiles_n = 0; a.stamps = nullptr;
    auto run = [&](unsigned long long* st) {
        a.stamps = st;
        switch (variant) {
            case 1: return launch<256, 128, 32, 2, 2, 3>(a, epi, 0);
            case 4: return launch<256, 256, 64, 2, 4, 2>(a, epi, 0);
            case 5: return launch_pp(a, epi, 0);
            default: return 1;
        }
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) run(nullptr);
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) run(nullptr); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d epilogue %d: %.1f us per launch (no stamps)\n", variant, epi, ms * 100);
    if (variant == 5) {  // persistent kernel: per-wave sums over its whole stream of K steps
        hipMemset(stamps, 0, (size_t)waves_total * 16 * 8);
        run(stamps); hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)waves_total * 16);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        const char* pn[] = {"L phase (DMA issue + fragment reads)", "barrier after L", "M phase (32 MFMAs + vmcnt)", "barrier after M", "epilogues"};
        for (int grp = 0; grp < 2; ++grp) {
            printf("  waves %d-%d, cycles per K step (32 deep); epilogue: per tile\n", 4 * grp, 4 * grp + 3);
            for (int s = 0; s < 5; ++s) {
                std::vector<double> d;
                for (int wg = 0; wg < 256; ++wg)
                    for (int w = 4 * grp; w < 4 * grp + 4; ++w) {
                        const unsigned long long* o = &h[((size_t)wg * 8 + w) * 16];
                        if (o[13]) d.push_back((double)o[8 + s] / (s == 4 ? (double)o[13] / (K / 32) : (double)o[13]));
                    }
                if (d.empty()) continue;
                std::sort(d.begin(), d.end());
                printf("    %-40s median %7.0f  p10 %7.0f  p90 %7.0f\n", pn[s], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
            }
        }
        return 0;
    }
    hipMemset(stamps, 0, (size_t)waves_total * 16 * 8);
    run(stamps); hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)waves_total * 16);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"prologue issue", "first tile landed (barrier 0)", "K tile 0", "K tile 1", "K tiles 2..end", "epilogue barrier", "epilogue"};
    for (int s = 0; s < 7; ++s) {
        std::vector<double> d;
        for (int w = 0; w < waves_total; ++w) {
            const unsigned long long a0 = h[(size_t)w * 16 + s], a1 = h[(size_t)w * 16 + s + 1];
            if (a0 && a1 && a1 >= a0) d.push_back((double)(a1 - a0));
        }
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("  %-32s median %8.0f  p10 %8.0f  p90 %8.0f cycles  (%zu waves)\n", names[s], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10], d.size());
    }
    const char* pn[] = {"sum over K tiles >= 1: vmcnt wait", "  barrier wait", "  fragment reads + MFMAs"};
    for (int s = 0; s < 3; ++s) {
        std::vector<double> d;
        for (int w = 0; w < waves_total; ++w) if (h[(size_t)w * 16]) d.push_back((double)h[(size_t)w * 16 + 8 + s]);
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        printf("  %-36s median %8.0f  p10 %8.0f  p90 %8.0f cycles\n", pn[s], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    std::vector<double> tot;
    for (int w = 0; w < waves_total; ++w) { const unsigned long long a0 = h[(size_t)w * 16], a1 = h[(size_t)w * 16 + 7]; if (a0 && a1) tot.push_back((double)(a1 - a0)); }
    std::sort(tot.begin(), tot.end());
    if (!tot.empty()) printf("  %-32s median %8.0f cycles\n", "whole workgroup", tot[tot.size() / 2]);
    return 0;
}
