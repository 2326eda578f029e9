// Internal launcher of the persistent stream GEMM (gemm_ps.hip); the C ABI entry is grit_gemm_bf16_nt, variant 6 (gemm.hip).
#pragma once
namespace grit_detail {
int gemm_ps_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, int nt, void* stream, unsigned long long* stamps);
// four waves, 128 x 128 wave tiles (gemm_w4.hip): variant 7
int gemm_w4_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, float* colsum, int nt, void* stream);
}
