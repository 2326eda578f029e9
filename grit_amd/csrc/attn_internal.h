// Internal (not part of the C ABI): MFMA bf16 attention launchers of attn_mfma.hip, tried first by
// grit_attn_{fwd,bwd}_bf16 (attn.hip).  They return GRIT_ERR_UNSUPPORTED when the shape / alignment does not
// fit (Tq, Nk <= 160, head_dim 64, 16-byte aligned rows), in which case the fp32 VALU kernels run.  Both paths
// store the row log-sum-exp in natural-log units, so a forward of one kind pairs with a backward of the other.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int grit_attn_mfma_fwd(const void* q, long ldq, long bsq, const void* k, long ldk, long bsk, const void* v, long ldv,
                       long bsv, const uint8_t* mask, long msb, long msq, int B, int H, int Tq, int Nk, int D, float scale,
                       float drop_p, unsigned long long seed, const unsigned long long* seed_dev, void* out, float* lse,
                       hipStream_t st);
int grit_attn_mfma_bwd(const void* q, long ldq, long bsq, const void* k, long ldk, long bsk, const void* v, long ldv,
                       long bsv, const uint8_t* mask, long msb, long msq, const void* out, const void* dout,
                       const float* lse, int B, int H, int Tq, int Nk, int D, float scale, float drop_p,
                       unsigned long long seed, const unsigned long long* seed_dev, void* dq, void* dk, void* dv,
                       hipStream_t st);
