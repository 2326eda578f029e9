/*
 * grit_hip.h -- C ABI of libgrit_hip.so, the MI355X (gfx950) kernels behind the GRIT captioning hot path.
 *
 * Every entry point takes plain device pointers + sizes + a hipStream_t (passed as void*), allocates nothing,
 * keeps no state between calls and is re-entrant (forward is called from the Python main thread, backward
 * from the autograd engine thread).  The caller sets the device, owns every buffer and passes the stream the
 * tensors were produced on.  Return value: 0 on success, otherwise one of the GRIT_ERR_* codes below
 * (grit_status_string() gives the text; the Python shim turns it into RuntimeError).
 *
 * Reference interfaces replaced (paths into davidnvq/grit):
 *   grit_msda_fwd_*      <- ms_deform_attn_forward   models/ops/src/ms_deform_attn.h:20-40, vision.cpp:14
 *                           (CUDA body models/ops/src/cuda/ms_deform_attn_cuda.cu:20-80,
 *                            kernel ms_deform_im2col_cuda.cuh:237-299)
 *   grit_msda_bwd_*      <- ms_deform_attn_backward  models/ops/src/ms_deform_attn.h:42-62, vision.cpp:15
 *                           (CUDA body ms_deform_attn_cuda.cu:83-153, kernel ms_deform_im2col_cuda.cuh:406-510)
 *   grit_winattn_*       <- WindowAttention.forward core + roll/partition/reverse around it,
 *                           models/common/swin_model.py:155-186, 244-300, 424-441 (no native ancestor)
 *   grit_attn_*          <- Attention.forward core (QK^T/sqrt(d_k), masked_fill(-inf), softmax, .V),
 *                           models/common/attention.py:51-88 (no native ancestor); also serves
 *                           nn.MultiheadAttention inside models/detection/det_module.py:330-333
 */
#ifndef GRIT_HIP_H
#define GRIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIT_ABI_VERSION 1

#define GRIT_OK 0
#define GRIT_ERR_BAD_ARG 1      /* null pointer, non-positive dimension, overflow of 32-bit index math      */
#define GRIT_ERR_UNSUPPORTED 2  /* shape outside what the kernels implement (stated per function)           */
#define GRIT_ERR_LAUNCH 3       /* hipGetLastError() != hipSuccess after the launch                         */

/* Library ABI version; bumped whenever a signature below changes. */
int grit_abi_version(void);
const char* grit_status_string(int status);

/* ------------------------------------------------------------------------------------------------------
 * Multi-scale deformable attention.
 *
 *   value           [B, S, M, D]        contiguous; S = sum_l H_l*W_l
 *   spatial_shapes  [L, 2] int64 (H, W) DEVICE memory, as in the reference
 *   level_start     [L]    int64        DEVICE memory
 *   loc             [B, Lq, M, L, P, 2] (x, y) normalised to [0,1]
 *   attn_w          [B, Lq, M, L, P]
 *   out             [B, Lq, M*D]        fully overwritten (need not be zeroed)
 *
 * out[b,q,m,:] = sum_{l,p} attn_w * bilinear(value_l[b,:,m,:], (x*W_l-0.5, y*H_l-0.5)), zero padding,
 * a point contributes only if -1 < h < H_l and -1 < w < W_l  (ms_deform_im2col_cuda.cuh:288).
 * Any D >= 1; D == 64 and D == 32 take the wide-load fast path.  The reference's im2col_step
 * argument has no counterpart here: chunking is internal and B is unrestricted.
 * ------------------------------------------------------------------------------------------------------ */
int grit_msda_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const float* loc, const float* attn_w,
                      int B, int S, int M, int D, int L, int Lq, int P,
                      float* out, void* stream);
int grit_msda_fwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const double* loc, const double* attn_w,
                      int B, int S, int M, int D, int L, int Lq, int P,
                      double* out, void* stream);

/*   grad_out    [B, Lq, M*D]
 *   grad_value  like value   -- accumulated with float atomics: the CALLER MUST ZERO IT (reference:
 *                               at::zeros_like, ms_deform_attn_cuda.cu:121)
 *   grad_loc    like loc     -- fully overwritten
 *   grad_attn_w like attn_w  -- fully overwritten
 * Summation order of grad_value is not deterministic (atomics), exactly as in the reference. */
int grit_msda_bwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const float* loc, const float* attn_w, const float* grad_out,
                      int B, int S, int M, int D, int L, int Lq, int P,
                      float* grad_value, float* grad_loc, float* grad_attn_w, void* stream);
int grit_msda_bwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const double* loc, const double* attn_w, const double* grad_out,
                      int B, int S, int M, int D, int L, int Lq, int P,
                      double* grad_value, double* grad_loc, double* grad_attn_w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRIT_HIP_H */
