// Internal launchers shared by the translation units behind grit_gemm_bf16_nt (gemm.hip dispatches on `variant`).
#pragma once
namespace grit_detail {
// four waves, 128 x 128 wave tiles (gemm_w4.hip): variant 7
int gemm_w4_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, float* colsum, int nt, void* stream);
}
