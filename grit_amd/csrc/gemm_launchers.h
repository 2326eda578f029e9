// Internal launchers shared by the translation units behind grit_gemm_bf16_nt (gemm.hip dispatches on `variant`).
#pragma once
namespace grit_detail {
// four waves, 128 x 128 wave tiles (gemm_w4.hip): variant 7
// tile_rows: 256, 224 or 0 = gemm_w4_tile_rows(M, N); row_scale / rows_per_sample: GRIT_GEMM_BIAS_RES only
int gemm_w4_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, float* colsum, int nt, void* stream, int tile_rows = 256,
                   const float* row_scale = nullptr, int rows_per_sample = 0);
int gemm_w4_tile_rows(int M, int N);
}
