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
 *   grit_layernorm_*, grit_add_layernorm_*
 *                        <- nn.LayerNorm of the Swin blocks (models/common/swin_model.py:229,233,315,495) and the
 *                           "projection -> dropout / drop-path -> residual -> LayerNorm" tails around it
 *                           (swin_model.py:289-298, det_module.py:313-349, attention.py:166-184, pos_embed.py:44-48)
 *   grit_relbias_*       <- relative_position_bias_table[relative_position_index] gather, swin_model.py:168-171
 *   grit_groupnorm_tokens_* <- input_proj's GroupNorm(32, 512) + flatten/cat, models/caption/detector.py:28-33,58,
 *                           models/detection/det_module.py:172-175
 *   grit_colsum, grit_slab_sum <- bias / weight gradient reductions of nn.Linear's backward on the long token maps
 *   grit_adam_flat       <- the two torch.optim.Adam of build_optimizers, engine/caption_engine.py:18-73
 *   grit_resample_taps_bicubic, grit_image_batch_fwd
 *                        <- the image side of the batch contract: PIL resize(BICUBIC) of MaxWHResize / MinMaxResize
 *                           (datasets/caption/transforms/utils.py:4-45), ToTensor + Normalize
 *                           (datasets/caption/transforms/__init__.py:6-32), zero padding + mask of
 *                           nested_tensor_from_tensor_list (engine/utils.py:278-295)
 *   grit_gemm_bf16_nt    <- nn.Linear + nn.GELU of Mlp (models/common/swin_model.py:31-37) and their autograd backward:
 *                           fc1 + bias + exact GELU in one pass; fc2's input gradient x GELU' + fc1's bias gradient in one pass
 *   grit_topk_rows_f32   <- Transformer.select (models/caption/transformer.py:184-188): sort of beam x vocabulary candidates
 *   grit_beam_step_f32   <- the body of Transformer.iter after the word log-probabilities (models/caption/transformer.py:208-240):
 *                           finished-beam masking, candidate scores, selection, beam / word split, score / mask / log-prob gathers
 *   grit_decode_step_inputs <- CaptionGenerator.get_seq_inputs in stateful mode + the word / position embedding sum
 *                           (models/caption/cap_generator.py:116-137,148)
 *   grit_kv_append       <- running_keys / running_values of the stateful self-attention (models/common/attention.py:166-181) and
 *                           their per-step re-gather by the surviving beam (models/caption/transformer.py:229)
 *   grit_gate_pack, grit_gate_fuse
 *                        <- the sigmoid-gated merge of the two cross-attentions at inference, ParallelAttentionLayer.forward
 *                           (models/caption/cap_generator.py:44-56): masks, concatenations, sigmoids, products, sum, scale
 * (none of the last twelve groups has a native ancestor in the reference: they replace chains of torch / PIL ops)
 */
#ifndef GRIT_HIP_H
#define GRIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIT_ABI_VERSION 43

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

/* bf16 value maps (D = 64, L*P <= 32): value / out / grad_out are bf16, loc / attn_w and every gradient stay f32.
 * grad_value is f32 [B,S,M,64], zeroed by the caller, accumulated with float atomics.  Other shapes (L*P > 16 for
 * backward): GRIT_ERR_UNSUPPORTED (the caller converts to f32 and uses the entry points above). */
int grit_msda_fwd_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const float* loc,
                       const float* attn_w, int B, int S, int M, int D, int L, int Lq, int P, void* out, void* stream);
int grit_msda_bwd_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const float* loc,
                       const float* attn_w, const void* grad_out, int B, int S, int M, int D, int L, int Lq, int P,
                       float* grad_value, float* grad_loc, float* grad_attn_w, void* stream);
/* Same backward with grad_value ACCUMULATED IN bf16 by packed atomics (two channels per 32-bit memory-side atomic: twice
 * the rate of the f32 atomics that bound grit_msda_bwd_bf16, no f32 staging map, no cast).  grad_value [B, S, M, D] bf16,
 * zeroed by the caller.  Accumulation order and rounding are those of torch's own bf16 scatter / grid_sample backward;
 * use grit_msda_bwd_bf16 when f32 accumulation of the value gradient is required.  D = 64, L*P <= 16. */
int grit_msda_bwd_bf16acc(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const float* loc,
                          const float* attn_w, const void* grad_out, int B, int S, int M, int D, int L, int Lq, int P,
                          void* grad_value, float* grad_loc, float* grad_attn_w, void* stream);
/* The same two kernels on a value map whose pixels are `pixel_stride` elements apart (>= M*D, a multiple of 8; image b
 * starts at b * S * pixel_stride): the deformable decoder projects the flat feature map for ALL its layers with one GEMM,
 * value_proj of models/ops/modules/ms_deform_attn.py:93 x 6 layers of models/detection/det_module.py:274-349 -> one
 * [B, S, layers, M, D] tensor, and each layer samples its slice in place.  grad_value has the same layout (one zero-filled
 * buffer that all layers' backward kernels add into), so the projection's input gradient is ONE GEMM over K = layers*M*D. */
int grit_msda_fwd_bf16_strided(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                               const int64_t* level_start, const float* loc, const float* attn_w, int B, int S, int M,
                               int D, int L, int Lq, int P, void* out, void* stream);
int grit_msda_bwd_bf16acc_strided(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                                  const int64_t* level_start, const float* loc, const float* attn_w, const void* grad_out,
                                  int B, int S, int M, int D, int L, int Lq, int P, void* grad_value, float* grad_loc,
                                  float* grad_attn_w, void* stream);
/* bf16 maps, value gradient ACCUMULATED IN f32 (the precision of the reference's atomicAdd on its fp32 grad_value,
 * ms_deform_im2col_cuda.cuh:125-152) and rounded to bf16 once: the default of the training step.  `stage` [B, S, M, 64] f32 and
 * `cell_flags` [B, S, M] u8 are scratch that must ARRIVE ZEROED and is RETURNED ZEROED (the second launch of the call visits the
 * cells the scatter touched -- flagged by the first -- rounds them into grad_value and clears stage and flags), so one
 * allocation serves every layer and every step.  grad_value: bf16 map with `pixel_stride` elements between pixels, zero-filled
 * by the caller (untouched cells are not written).  Same row walk, same-cell merges and grad_loc / grad_attn_w arithmetic as
 * grit_msda_bwd_bf16acc_strided.  D = 64, L*P <= 16. */
int grit_msda_bwd_bf16_staged(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                              const int64_t* level_start, const float* loc, const float* attn_w, const void* grad_out,
                              int B, int S, int M, int D, int L, int Lq, int P, float* stage, unsigned char* cell_flags,
                              void* grad_value, float* grad_loc, float* grad_attn_w, void* stream);

/* bf16 maps, value gradient in GATHER form: the default of the training step.  No atomics on memory -- one workgroup per
 * (image, head) keeps that pair's whole problem in the LDS of its CU: the Lq rows of grad_out, the Lq*L*P*4 corner contributions
 * binned by cell (counting sort: LDS counters, LDS scan, LDS records {query, attention weight x bilinear weight}); every cell then
 * sums its contributions w_i * grad_out[q_i, :] in f32 registers and is rounded to bf16 ONCE.  f32 accumulation like the
 * reference's atomicAdd (ms_deform_im2col_cuda.cuh:125-152) without its rate limit (the memory-side atomic units of gfx950 do
 * ~323 G f32 adds/s: ~460 us per launch at GRIT's shapes; this path: profiles/r03/msda_bwd_methods.txt).
 * grad_value: bf16 map with `pixel_stride` elements between pixels (a multiple of 8, base 16-byte aligned); EVERY cell of the
 * [B, S, M, 64] slice is written (zeros where nothing sampled), so the caller does not zero-fill it.  grad_loc / grad_attn_w: the
 * row walk of grit_msda_bwd_bf16acc_strided with its scatter compiled out.  D = 64, L*P <= 16, Lq*L*P <= 4096 and
 * 4 (S + 1) + 32 Lq L P + 128 Lq + 64 <= 160 KB of LDS (GRIT: 130 KB); grit_msda_bwd_sorted_supported says GRIT_OK /
 * GRIT_ERR_UNSUPPORTED for a shape -- callers fall back to grit_msda_bwd_bf16_staged. */
int grit_msda_bwd_sorted_supported(int B, int S, int M, int L, int Lq, int P);
int grit_msda_bwd_bf16_sorted(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                              const int64_t* level_start, const float* loc, const float* attn_w, const void* grad_out,
                              int B, int S, int M, int D, int L, int Lq, int P, void* grad_value, float* grad_loc,
                              float* grad_attn_w, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Scaled-dot attention core, fp32 arithmetic, head_dim D = 64 (also the 150-query
 * self-attention of A5).  f32 and bf16 storage variants.
 *
 *   q      [B, Tq, H, 64]   row stride ldq, batch stride bsq (elements) -- a slice of a packed projection is fine
 *   k, v   [B, Nk, H, 64]   row strides ldk / ldv, batch strides bsk / bsv;  Nk <= 256
 *   mask   uint8, nonzero = masked (-inf before softmax), or NULL.  Element (b, q, j) is read at
 *          mask[b*mask_sb + q*mask_sq + j]; pass stride 0 to broadcast over batch and/or query
 *          (reference masks are [B,1,T,T], [B,1,1,Nk]: attention.py:79-80, cap_generator.py:126-136)
 *   out    [B, Tq, H*64] contiguous;   lse [B, H, Tq] row log-sum-exp of the scaled, masked scores
 *   P = softmax(scale * q k^T masked);  dropout_p > 0 drops entries of P with a counter hash of `seed`
 *   (the backward call must pass the same seed);  out = P_drop v.  The effective seed is `seed ^ *seed_dev` when
 *   seed_dev (device uint64, may be NULL) is given: a seed that lives in device memory can be refreshed by a captured
 *   RNG kernel, so a hipGraph replay of the step draws a new mask each time.
 *   A fully masked row yields NaN, as torch.softmax over all -inf does in the reference.
 * Backward overwrites dq [B,Tq,H,64], dk, dv [B,Nk,H,64] (contiguous); no atomics to global memory.
 * ------------------------------------------------------------------------------------------------------ */
int grit_attn_fwd_f32(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk,
                      const void* v, int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                      int B, int H, int Tq, int Nk, int D, float scale, float dropout_p, uint64_t seed,
                      const uint64_t* seed_dev, void* out, float* lse, void* stream);
int grit_attn_fwd_bf16(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk,
                       const void* v, int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                       int B, int H, int Tq, int Nk, int D, float scale, float dropout_p, uint64_t seed,
                       const uint64_t* seed_dev, void* out, float* lse, void* stream);
int grit_attn_bwd_f32(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk,
                      const void* v, int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                      const void* out, const void* dout, const float* lse,
                      int B, int H, int Tq, int Nk, int D, float scale, float dropout_p, uint64_t seed,
                      const uint64_t* seed_dev, void* dq, void* dk, void* dv, void* stream);
int grit_attn_bwd_bf16(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk,
                       const void* v, int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq,
                       const void* out, const void* dout, const float* lse,
                       int B, int H, int Tq, int Nk, int D, float scale, float dropout_p, uint64_t seed,
                       const uint64_t* seed_dev, void* dq, void* dk, void* dv, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Swin (shifted-)window attention, window 12 (N = 144), head_dim 32, bf16 storage / MFMA, fp32 softmax
 * (SURVEY 8 row A6).  Operates on the map in TOKEN ORDER; pad / roll / partition / reverse / crop of
 * swin_model.py:257-293 and the shift mask of :424-441 are done by address arithmetic inside the kernel.
 *
 *   qkv       bf16 [B, H*W, 3*C]   output of the qkv Linear on the un-partitioned map; channel = which*C + head*32 + d
 *   rel_bias  f32  [nH, 144, 144]  relative_position_bias_table gathered by relative_position_index (:168-171)
 *   pad_qkv   bf16 [3*C]           q/k/v of a window-padding token (= the qkv Linear bias)
 *   mask      f32  [n_mask_windows, 144, 144] additive, or NULL.  NULL + shift > 0 -> the kernel derives the
 *             0 / -100 shift mask from (H, W, shift) itself.  Non-NULL replaces it (WindowAttention.forward(x, mask)
 *             call form); window i uses mask[i % n_mask_windows].
 *   out       bf16 [B, H*W, C]     heads concatenated (channel = head*32 + d), padded positions are not written
 *   lse       f32  [B*nWh*nWw, nH, 144]  per-row log2-sum-exp2 of the logits, consumed by the backward call
 *   scale     q scaling (head_dim^-0.5);  C = num_heads*32;  0 <= shift < 12;  window must be 12.
 * Backward: dqkv bf16 like qkv (fully overwritten); drel_bias f32 like rel_bias and dpad f32 [3*C] are
 * ACCUMULATED with float atomics -- the caller zeroes them.
 * Two kernels behind each entry point, same results: the DMA-staged ones (default; operands of the next window by global_load_lds into
 * double-buffered LDS tiles, 160 KB of dynamic LDS for the backward) and the register-staged ones (environment
 * GRIT_WINATTN_FWD_DMA=0 / GRIT_WINATTN_BWD_DMA=0, read once per process; a non-NULL mask always takes the register-staged forward).
 * ------------------------------------------------------------------------------------------------------ */
int grit_winattn_fwd_bf16(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                          int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                          void* out, float* lse, void* stream);
int grit_winattn_bwd_bf16(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                          const void* out, const void* dout, const float* lse,
                          int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                          void* dqkv, float* drel_bias, float* dpad, void* stream);
/* The forward with the drop-path factors of the branch the attention belongs to (training): the windows of images with
 * row_scale[b] == 0 are not computed, out and lse receive zeros for them -- valid only because the caller multiplies the branch by the
 * factors afterwards and hands the SAME factors to grit_winattn_bwd_bf16_rows.  Applies under the conditions listed there (both
 * DMA-staged kernels in use); otherwise every window is computed. */
int grit_winattn_fwd_bf16_rows(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                               int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                               void* out, float* lse, const float* row_scale, void* stream);
/* The backward with the drop-path factors of the attention branch (reference timm DropPath, models/common/swin_model.py:289-298):
 * row_scale[b] == 0 promises that dout is zero for image b.  Its windows contribute nothing to any output: they are not computed, dqkv
 * receives zeros for its tokens, the workgroups share the windows of the kept images.  row_scale == NULL, an explicit mask, B > 64,
 * a register-staged variant selected (GRIT_WINATTN_FWD_DMA=0 / GRIT_WINATTN_BWD_DMA=0) or GRIT_WINATTN_ROW_SKIP=0: every window is computed
 * (same results). */
int grit_winattn_bwd_bf16_rows(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                               const void* out, const void* dout, const float* lse,
                               int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                               void* dqkv, float* drel_bias, float* dpad, const float* row_scale, void* stream);

/* The same two entry points in fp32 storage and plain fp32 arithmetic (fmaf chains, no matrix cores; lse in natural-log
 * units): the parity path for fp32 weights -- the reference's WindowAttention is fp32 (swin_model.py:155-186).  Not tuned for
 * throughput; a forward of one precision pairs only with the backward of the same precision. */
int grit_winattn_fwd_f32(const float* qkv, const float* rel_bias, const float* pad_qkv, const float* mask, int n_mask_windows,
                         int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                         float* out, float* lse, void* stream);
int grit_winattn_bwd_f32(const float* qkv, const float* rel_bias, const float* pad_qkv, const float* mask, int n_mask_windows,
                         const float* out, const float* dout, const float* lse, int B, int H, int W, int C, int num_heads,
                         int window, int shift, float scale, float* dqkv, float* drel_bias, float* dpad, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension for the Swin token maps (nn.LayerNorm at swin_model.py:229,233,315,495).
 *   x, y, dy, dx  [rows, C] contiguous, f32 (x_is_bf16 = 0) or bf16 (1);  C in {128, 256, 512, 1024, 2048, 4096}
 *   weight, bias  [C], f32 or bf16 (w_is_bf16); the combination f32 x with bf16 weight is not provided
 *   mean, rstd    [rows] f32, written by forward, read by backward
 *   dweight/dbias [GRIT_LN_BWD_PARTIALS, C] f32 per-workgroup partial sums: rows [0, min(ceil(rows / rows_per_block),
 *                 GRIT_LN_BWD_PARTIALS)) are overwritten and only those may be summed (grit_slab_sum); no zero fill needed
 * Statistics and arithmetic are fp32; eps as in torch.nn.functional.layer_norm.
 * ------------------------------------------------------------------------------------------------------ */
#define GRIT_LN_BWD_PARTIALS 1024
int grit_layernorm_fwd(const void* x, const void* weight, const void* bias, int rows, int C, float eps, int x_is_bf16,
                       int w_is_bf16, void* y, float* mean, float* rstd, void* stream);
int grit_layernorm_bwd(const void* x, const void* weight, const void* dy, const float* mean, const float* rstd, int rows,
                       int C, int x_is_bf16, int w_is_bf16, void* dx, float* dweight, float* dbias, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Relative-position bias of window attention (WindowAttention.forward, swin_model.py:168-171):
 *   bias[h][p] = (float) table[index[p]][h]          table [n_rows, num_heads] f32 / bf16, index [n_pos] int64
 * and its gradient  dtable[r][h] = sum over the positions p with index[p] == r of dbias[h][p].  The positions are
 * passed sorted by table row: order [n_pos] int32 (a permutation of 0..n_pos-1, stable sort of index) and
 * offsets [n_rows + 1] int32 (row r owns order[offsets[r] .. offsets[r+1])).  dtable is fully overwritten.
 * ------------------------------------------------------------------------------------------------------ */
int grit_relbias_fwd(const void* table, const int64_t* index, int n_rows, int num_heads, int n_pos, int table_is_bf16,
                     float* bias, void* stream);
/* Up to GRIT_RELBIAS_GROUP_MAX forward gathers in ONE launch (the 24 Swin blocks of a step: same arithmetic per job as grit_relbias_fwd). */
#define GRIT_RELBIAS_GROUP_MAX 32
typedef struct grit_relbias_job {
    const void* table;
    const int64_t* index;
    float* bias;
    int n_rows, num_heads, n_pos, table_is_bf16;
} grit_relbias_job;
int grit_relbias_fwd_grouped(const grit_relbias_job* jobs, int n_jobs, void* stream);
/* ... and up to GRIT_RELBIAS_GROUP_MAX table gradients in ONE launch (same arithmetic per job as grit_relbias_bwd). */
typedef struct grit_relbias_bwd_job {
    const float* dbias;
    const int32_t* order;
    const int32_t* offsets;
    void* dtable;
    int n_rows, num_heads, n_pos, table_is_bf16;
} grit_relbias_bwd_job;
int grit_relbias_bwd_grouped(const grit_relbias_bwd_job* jobs, int n_jobs, void* stream);
int grit_relbias_bwd(const float* dbias, const int32_t* order, const int32_t* offsets, int n_rows, int num_heads, int n_pos,
                     int table_is_bf16, void* dtable, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Residual connection + the LayerNorm that follows it, one pass (SwinTransformerBlock.forward, swin_model.py:289-298:
 * `x = shortcut + self.drop_path(x)` then `self.norm2(x)`; likewise the MLP residual and the next block's norm1):
 *   sum_out = round(shortcut + row_scale[row / rows_per_sample] * branch)     (row_scale NULL: plain add)
 *   y       = LayerNorm(sum_out) with the statistics of the rounded sum, i.e. bit-identical to the unfused pair
 * Backward: dx = dres + dLayerNorm(dy)   (dres = gradient reaching sum_out through the skip path, may be NULL) is the
 * gradient of `shortcut`; with row_scale the gradient of `branch`, dbranch = row_scale[row / rows_per_sample] * dx, is
 * written too (row_scale and dbranch are given together or both NULL: the branch gradient is then dx itself).
 * drop_p > 0 additionally applies element dropout to the branch before the residual add (the nn.Dropout between a
 * projection and `LayerNorm(x + dropout(proj))` in the post-norm decoder layers: det_module.py:313-349, attention.py:
 * 166-184, pos_embed.py:44-48): keep mask = counter hash of (seed read from seed_dev, element index), regenerated in the
 * backward (same drop_p / seed_dev), which then requires dbranch.
 * dbranch_colsum (optional, C <= 1024): [GRIT_LN_BWD_PARTIALS, C] f32 per-workgroup partial column sums of the branch
 * gradient as stored -- the bias gradient of the Linear that produced `branch` (attn.proj / mlp.fc2), for free.
 * Other arguments as in grit_layernorm_{fwd,bwd}.
 * ------------------------------------------------------------------------------------------------------ */
/* Patch embedding of the Swin backbone in one pass (reference models/common/swin_model.py:336-365, PatchEmbed: Conv2d(3, C, kernel 4,
 * stride 4) -> flatten(2).transpose(1, 2) -> LayerNorm(C)): out[b, hh * W/4 + ww, :] = LayerNorm(conv(img)[b, :, hh, ww] + bias),
 * bf16.  img [B, 3, H, W] fp32 (img_is_bf16 = 0: rounded to bf16 on load, as the unfused cast does) or bf16; weight [C, 48] bf16 =
 * the conv weight flattened in its own (c, kh, kw) order; bias, gamma, beta [C] bf16; C in {96, 128, 192}; H % 4 == 0, W % 64 == 0
 * (GRIT_ERR_UNSUPPORTED otherwise: the caller keeps its GEMM path); 16-byte aligned bases.  The conv output is rounded to bf16
 * before the statistics, like the tensor the unfused GEMM stores.  Forward only: PatchEmbed is frozen in GRIT (frozen_stages = 2). */
int grit_patch_embed_ln_fwd(const void* img, int img_is_bf16, int B, int H, int W, int C, const void* weight, const void* bias,
                            const void* gamma, const void* beta, float eps, void* out, void* stream);

/* LayerNorm of the PATCH-MERGED map without the map (reference models/common/swin_model.py:279-288, PatchMerging.forward: 2 x 2
 * neighbourhood concat in the order (0,0), (1,0), (0,1), (1,1) -> LayerNorm(4 Cs)): x is the token map [B, H, W, Cs] (H, W even, Cs a
 * multiple of 8, 4 Cs in the supported widths); row r of the [B * H/2 * W/2, 4 Cs] view is gathered on load.  y, mean, rstd as in
 * grit_layernorm_fwd on that view.  Backward: dy [rows, 4 Cs] contiguous, dx is written straight in the token-map layout
 * [B, H, W, Cs] (every element exactly once).  Replaces the permute + reshape copy and the scatter of its gradient. */
int grit_merge_layernorm_fwd(const void* x, int B, int H, int W, int Cs, const void* weight, const void* bias, float eps,
                             int x_is_bf16, int w_is_bf16, void* y, float* mean, float* rstd, void* stream);
int grit_merge_layernorm_bwd(const void* x, int B, int H, int W, int Cs, const void* weight, const void* dy, const float* mean,
                             const float* rstd, int x_is_bf16, int w_is_bf16, void* dx, float* dweight, float* dbias,
                             void* stream);
int grit_add_layernorm_fwd(const void* shortcut, const void* branch, const float* row_scale, int rows_per_sample,
                           float drop_p, const uint64_t* seed_dev, const void* weight, const void* bias, int rows, int C,
                           float eps, int x_is_bf16, int w_is_bf16, void* sum_out, void* y, float* mean, float* rstd,
                           void* stream);
int grit_add_layernorm_bwd(const void* x, const void* weight, const void* dy, const void* dres, const float* mean,
                           const float* rstd, const float* row_scale, int rows_per_sample, float drop_p,
                           const uint64_t* seed_dev, int rows, int C, int x_is_bf16, int w_is_bf16, void* dx, void* dbranch,
                           float* dweight, float* dbias, float* dbranch_colsum, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * GroupNorm on token-major maps (nn.GroupNorm(32, hidden_dim) after the 1x1 input projection of every feature level,
 * reference models/caption/detector.py:28-33,58; statistics over (H*W, C/G) per image and group, biased variance).
 *   x [B, T, C]: image b at x + b * x_bstride (elements), rows of C channels contiguous; C in {256, 512}, (C/G) % 8 == 0
 *   y            same geometry with its own batch stride: a level can be written straight into its slice of the flat
 *                [B, sum_l T_l, C] map that the deformable-attention decoder consumes
 *   mean, rstd   [B, G] f32, written by forward, read by backward
 *   workspace    f32, forward: B * GRIT_GN_CHUNKS * 2 * G, backward: B * GRIT_GN_CHUNKS * 2 * C elements (fully overwritten)
 * Backward: dx [B, T, C] contiguous; dweight / dbias [C] in the weight dtype.  f32 x with bf16 weight is not provided.
 * ------------------------------------------------------------------------------------------------------ */
#define GRIT_GN_CHUNKS 16
int grit_groupnorm_tokens_fwd(const void* x, long x_bstride, const void* weight, const void* bias, int B, int T, int C, int G,
                              float eps, int x_is_bf16, int w_is_bf16, void* y, long y_bstride, float* mean, float* rstd,
                              float* workspace, void* stream);
int grit_groupnorm_tokens_bwd(const void* x, long x_bstride, const void* dy, long dy_bstride, const void* weight,
                              const float* mean, const float* rstd, int B, int T, int C, int G, int x_is_bf16, int w_is_bf16,
                              void* dx, void* dweight, void* dbias, float* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Column sums (bias gradient of nn.Linear: db = sum over rows of dY).  x [M, N] row-major, f32 or bf16, N % 8 == 0.
 * partial [slabs, N] f32 is fully overwritten: row s holds the sums over row slab s; the caller adds the slabs.
 * ------------------------------------------------------------------------------------------------------ */
#define GRIT_COLSUM_MAX_SLABS 256
int grit_colsum(const void* x, int M, int N, int x_is_bf16, int slabs, float* partial, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Second stage of the slab-wise reductions of the backward pass: out[g][i] = cast(sum_s partial[g][s][i]).
 * Replaces the `partial.sum(0).to(dtype)` pairs that follow grit_colsum, the split-M weight-gradient GEMM
 * (dW of nn.Linear, autograd's mm at the sites of swin_model.py:75-76,139,148) and grit_layernorm_bwd.
 *   partial  f32, group g at partial + g * group_stride, slab s of a group at + s * n      (n % 4 == 0, 16-byte aligned)
 *   out      [groups, n] f32 (out_is_bf16 = 0) or bf16 (1)
 * ------------------------------------------------------------------------------------------------------ */
int grit_slab_sum(const float* partial, int groups, long group_stride, int slabs, long n, void* out, int out_is_bf16,
                  void* stream);

/* Up to GRIT_SLAB_GROUP_MAX slab sums in ONE launch (same arithmetic per job as grit_slab_sum, job table passed by value): the
 * reductions a backward node owes -- split-M weight gradients, bias-gradient column sums, LayerNorm dgamma / dbeta -- done together
 * instead of as dependent launches of a few microseconds each.  bf16 outputs need 8-byte, f32 outputs 16-byte alignment. */
#define GRIT_SLAB_GROUP_MAX 48
typedef struct grit_slab_job {
    const float* partial;   /* group g at partial + g * group_stride, slab s of a group at + s * n */
    long group_stride;
    int groups;
    int slabs;
    long n;
    void* out;              /* [groups, n] */
    int out_is_bf16;
    const float* extra;     /* NULL, or [groups, n] f32 (16-byte aligned): one more term of every sum, added last (ABI 40: the q / k / v
                             * rows of the window-padding tokens are the qkv bias -- grit_winattn_bwd's d(pad) joins the bias gradient's sum) */
} grit_slab_job;
int grit_slab_sum_grouped(const grit_slab_job* jobs, int n_jobs, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Weight + bias gradient of nn.Linear on SHORT token maps (the decoders and the grid net, M = 640 .. 4 800 rows; autograd of
 * the Linear layers of models/detection/det_module.py:313-349, models/caption/grid_net.py:9-42, models/common/attention.py:51-88,
 * models/common/pos_embed.py:34-48):     dW[n, k] = sum_m dY[m, n] * X[m, k]        db[n] = sum_m dY[m, n]
 * dY [M, N], X [M, K] bf16 row-major with leading dimensions ldy / ldx (elements, multiples of 8; 16-byte aligned bases);
 * N % 64 == 0, K % 64 == 0.  splits = grit_wgrad_small_splits(M, N, K) (0: shape not supported):
 *   splits == 1  (M <= 4 800): dW_out is the finished gradient, bf16 [N, K], db_out bf16 [N] (or NULL: not wanted) -- ONE launch
 *                replaces the library's transposed GEMM (24-47 us for these 2.5 GFLOP problems), the column-sum kernel and the
 *                reduction launch behind it;
 *   splits  > 1  dW_out = f32 partials [splits, N, K], db_out = f32 partials [splits, N] (or NULL), fully overwritten; the
 *                caller sums over the splits (grit_slab_sum / grit_slab_sum_grouped).
 * ------------------------------------------------------------------------------------------------------ */
int grit_wgrad_small_splits(int M, int N, int K);
int grit_wgrad_small(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, void* dW_out,
                     void* db_out, void* stream);
/* Up to GRIT_WGRAD_GROUP_MAX such problems in ONE launch (job table by value): the deferred weight / bias gradients of a decoder's
 * short-map Linears (grit_amd/ops/linear.py DeferredWgrads).  Every job writes f32 partials dW_partial [splits, N, K] and, when
 * db_partial != NULL, [splits, N], splits = grit_wgrad_group_splits(M); the caller reduces them (grit_slab_sum_grouped). */
#define GRIT_WGRAD_GROUP_MAX 32
typedef struct grit_wgrad_job {
    const void* dY; long ldy;
    const void* X; long ldx;
    int M, N, K, splits;
    float* dW_partial;
    float* db_partial;
    /* grit_wgrad_tn_grouped only (grit_wgrad_small_grouped ignores them): per-sample factors of the branch whose gradient dY is -- see
     * grit_wgrad_tn_rows.  NULL / 0: every row is processed. */
    const float* row_scale;
    int rows_per_sample;
} grit_wgrad_job;
int grit_wgrad_group_splits(int M);
int grit_wgrad_small_grouped(const grit_wgrad_job* jobs, int n_jobs, void* stream);
/* The same job table through the long-map kernel (grit_wgrad_tn's 256 x 256 tiles, one launch for all jobs): for jobs with
 * N % 256 == 0, K % 256 == 0, M % 32 == 0 (grit_wgrad_tn_group_ok), db_partial [splits, N] (column sums per slice) or NULL,
 * splits = any number of row slices such that every slice owns at least one step (64 rows when M % 64 == 0, else 32; slice s =
 * rows [s * ceil(steps / splits) * step_rows, ...)) -- the caller picks it so that the tiles of all jobs together fill the chip with
 * LONG loops (thirty M = 4 800 problems are 128 tiles x 2 slices of 38 steps).  dW_partial [splits, N, K] fp32 as above.  When every
 * job has M % 64 == 0 the launch runs the four-wave kernel (wgrad_tn4_256_grouped), else the eight-wave one; in both the workgroups
 * of one XCD (block % 8) take a contiguous band of a job's (slice, tile) pairs. */
int grit_wgrad_tn_group_ok(int M, int N, int K);
int grit_wgrad_tn_grouped(const grit_wgrad_job* jobs, int n_jobs, void* stream);
/* Column sums of up to GRIT_COLSUM_GROUP_MAX bf16 matrices x [M, N] (leading dimension ld) in one launch:
 * partial[slabs, N] fp32 per job, slab s = rows [s * ceil(M / slabs), ...); N % 8 == 0, ld % 8 == 0, 16-byte aligned bases. */
#define GRIT_COLSUM_GROUP_MAX 32
typedef struct grit_colsum_job {
    const void* x; long ld;
    int M, N, slabs;
    float* partial;
} grit_colsum_job;
int grit_colsum_grouped(const grit_colsum_job* jobs, int n_jobs, void* stream);

/* Transposed copies dst [cols, rows] of up to GRIT_TRANSPOSE_GROUP_MAX bf16 matrices src [rows, cols] (contiguous) in one launch:
 * the K-contiguous B operands grit_gemm_bf16_nt needs for an input gradient (fc2.weight^T of every Swin Mlp,
 * models/common/swin_model.py:31-37: autograd's Linear backward leaves that transpose to its GEMM) made once per optimizer step
 * instead of once per block inside backward.  rows, cols multiples of 64, 16-byte aligned bases. */
#define GRIT_TRANSPOSE_GROUP_MAX 32
typedef struct grit_transpose_job {
    const void* src; void* dst;
    int rows, cols;
} grit_transpose_job;
int grit_transpose_bf16_grouped(const grit_transpose_job* jobs, int n_jobs, void* stream);

/* Weight gradient of a Linear on a LONG token map (the Swin blocks): partial[s, N, K] (fp32) = dY[rows of slice s]^T . X[rows of
 * slice s] for S = grit_wgrad_tn_splits(M, N, K) row slices (chosen so that tiles x S fills the chip with one 256 x 256-tile
 * workgroup per CU); the caller sums the slices (grit_slab_sum_grouped).  The contraction runs over the slow index of both
 * operands; the transpose happens in the LDS reads (ds_read_b64_tr_b16 on both MFMA operands, grit_amd/csrc/wgrad_tn.hip).
 * Replaces the batched library GEMM over row slices of autograd's Linear backward (models/common/swin_model.py:26-35, 147-149).
 * N, K multiples of 256, M a multiple of 32 (grit_wgrad_tn_splits returns 0 otherwise: use the library), 16-byte aligned bases,
 * leading dimensions multiples of 8.  Slices are whole steps of 64 rows when M % 64 == 0 (the four-wave kernel: 128 x 128 wave
 * tiles, AGPR accumulators; GRIT_WGRAD_TN_W4=0 selects the eight-wave kernel for A/B runs), of 32 rows otherwise (eight waves).
 * db_partial != NULL: [S, N] fp32 column sums of dY per slice (the bias gradient) as a by-product of the workgroups of k-tile 0
 * (four waves: one extra MFMA per dY^T fragment against a block of ones; eight waves: v_dot2 on the fragments + LDS adds). */
int grit_wgrad_tn_splits(int M, int N, int K);
int grit_wgrad_tn(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, float* partial,
                  float* db_partial, void* stream);
/* The same with the drop-path factors of the branch whose gradient dY is (reference timm DropPath, models/common/swin_model.py:289-298:
 * per-sample keep w.p. 1 - p): row_scale[b] == 0 promises that rows [b * rows_per_sample, (b + 1) * rows_per_sample) of dY are exact
 * zeros.  Those rows add nothing to dW, so their 64-row steps are never loaded and the S slices share the LIVE steps equally (a slice is
 * a run of live steps instead of a fixed row range: the slice partials differ from grit_wgrad_tn's, their sum only by fp32 summation
 * order).  Applies on the four-wave kernel when rows_per_sample % 64 == 0 and M / rows_per_sample is 2..64; otherwise -- and with
 * GRIT_WGRAD_ROW_SKIP=0 -- every row is processed, which is equally correct.  X may hold anything in skipped rows (never read). */
int grit_wgrad_tn_rows(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, float* partial,
                       float* db_partial, const float* row_scale, int rows_per_sample, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Adam step on one flat range of the fp32-master / bf16-compute training state (torch.optim.Adam as configured by the
 * reference's build_optimizers, engine/caption_engine.py:18-73: amsgrad off, weight decay 0).
 *   param, exp_avg, exp_avg_sq  f32 [n], updated in place;  grad [n] bf16 (grad_is_bf16) or f32, multiplied by grad_scale
 *   compute_bf16                bf16 [n] copy of the updated parameters, or NULL
 *   bias_correction1 = 1 - beta1^t,  bias_correction2_sqrt = sqrt(1 - beta2^t)   (t = 1-based step count)
 * n % 4 == 0; f32 pointers 16-byte, bf16 pointers 8-byte aligned.
 * ------------------------------------------------------------------------------------------------------ */
int grit_adam_flat(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq, void* compute_bf16,
                   long n, float lr, float beta1, float beta2, float eps, float bias_correction1,
                   float bias_correction2_sqrt, float grad_scale, void* stream);
/* The same step with its two per-step scalars read from DEVICE memory at run time: hyper[0] = lr / bias_correction1,
 * hyper[1] = 1 / bias_correction2_sqrt (8-byte aligned).  For a training step captured in a HIP graph (grit_amd/engine/graph_step.py):
 * the host rewrites hyper before every replay, so the learning-rate schedule and the step count advance without a re-capture. */
int grit_adam_flat_dev(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq, void* compute_bf16,
                       long n, float beta1, float beta2, float eps, float grad_scale, const float* hyper, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Decoded RGB images -> the model's input batch: bicubic resize with Pillow's 8-bit arithmetic
 * (bit-identical to Image.resize(size, Image.BICUBIC) on an RGB image), x/255, (x - mean)/std, zero padding to the
 * batch maximum and the padding mask -- transforms/utils.py:4-45, transforms/__init__.py:6-32, engine/utils.py:278-295.
 *
 * grit_resample_taps_bicubic (host code, no GPU work): the tap table of one axis.
 *   bounds [out_size][2] = (first source index, tap count);  taps [out_size][ksize] 22-bit fixed point, zero past the count
 *   returns ksize (> 0); with bounds == taps == NULL only ksize is computed; -GRIT_ERR_BAD_ARG on bad sizes / capacity
 *
 * grit_image_batch_fwd (two launches):
 *   src     uint8 blob holding the images, each [src_h, src_w, 3] contiguous at byte offset src_off; 4-byte aligned and
 *           readable for GRIT_IMAGE_SRC_PAD bytes past the last image (the row pass fetches whole dwords)
 *   desc    [batch][GRIT_IMAGE_DESC_FIELDS] int64 (device): src_off, src_h, src_w, dst_h, dst_w, kx, ky, xb_off, xt_off,
 *           yb_off, yt_off, tmp_off -- k* = ksize of the x / y table, *b_off / *t_off = offsets (in int32 elements) of the
 *           bounds / taps of that axis inside `tables` (even), tmp_off = byte offset (multiple of 4) of this image's slot
 *           in tmp: src_h rows of GRIT_IMAGE_TMP_PITCH(dst_w) bytes
 *   tables  int32 blob (device) of the tap tables;  tmp uint8 scratch, 4-byte aligned, sum of the slots
 *   lut     [3][256] f32: lut[c][v] = (v / 255 - mean[c]) / std[c], computed by the caller
 *   out     [batch, 3, out_h, out_w] f32 and mask [batch, out_h, out_w] uint8 (1 on padding), both fully overwritten
 *   max_src_h = max src_h, max_dst_w = max dst_w (<= out_w), max_kx = max kx over the batch; grid extents <= 65535
 * ------------------------------------------------------------------------------------------------------ */
#define GRIT_IMAGE_DESC_FIELDS 12
#define GRIT_IMAGE_SRC_PAD 64
#define GRIT_IMAGE_TMP_PITCH(dst_w) ((3 * (dst_w) + 3) & ~3)
int grit_resample_taps_bicubic(int in_size, int out_size, int32_t* bounds, int32_t* taps, long taps_capacity);
int grit_image_batch_fwd(const uint8_t* src, const int64_t* desc, const int32_t* tables, uint8_t* tmp, const float* lut,
                         int batch, int max_src_h, int max_dst_w, int max_kx, int out_h, int out_w, float* out,
                         uint8_t* mask, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * bf16 MFMA GEMM with fused epilogues (Swin Mlp, models/common/swin_model.py:31-37, on the [B*H*W, C] token maps).
 *
 *   C[M, N] = epilogue(A[M, K] . B[N, K]^T)    A, B, C, aux, bias bf16; fp32 accumulation; row-major with leading
 *                                              dimensions lda / ldb / ldc / ldaux in elements (multiples of 8)
 *   GRIT_GEMM_NONE       C = acc
 *   GRIT_GEMM_BIAS       C = acc + bias[n]
 *   GRIT_GEMM_BIAS_GELU  aux = acc + bias[n] (written when aux != NULL: the pre-activation the backward needs);
 *                        C = gelu(acc + bias[n]) in fp32 on the unrounded sum, GELU as x * sigmoid(x * P(x^2)) fitted to the
 *                        erf form (|error| <= 2.6e-5 absolute, below bf16 resolution; tools/micro/fit_gelu.py)
 *   GRIT_GEMM_DGELU      C = acc * gelu'(aux[m, n]) (aux read; the exact derivative of the same expression); colsum[s, n] = sum over rows [128 s, 128 s + 128) of C
 *                        (fp32, before the bf16 rounding of C), s < ceil(M / GRIT_GEMM_COLSUM_ROWS): fully overwritten, to be summed over s
 *                        with grit_slab_sum (the bias gradient of the Linear that produced aux)
 * Needs N % 128 == 0, K % 32 == 0, 16-byte aligned bases; M is free.  variant 0 = tile configuration chosen from the
 * shape; 1..4 = explicit eight-wave configurations, 5 = persistent ping-pong, 7 = persistent FOUR-wave kernel with 128 x 128 wave tiles (gemm_w4.hip: N % 256 == 0,
 * K % 64 == 0, M >= 256, operands below 2 GiB; what the long-map forward / input-gradient GEMMs run where it beats the library).
 * 10..13 = the eight-wave kernel's template on SHORT-map tiles (64 x 128 x 64 three slots, 128 x 128 x 64 two slots, 64 x 64 x 64 three slots (N % 64 == 0
 * suffices), 64 x 128 x 32 four slots; two or three workgroups per CU) for maps of 640 .. 12 800 rows, NONE / BIAS / BIAS_GELU only: measured against
 * the library in profiles/r06/small_gemm.txt (1-2.5 us ahead per call at N, K <= 512; not wired into the decoders: 0.3 ms per step at stake).
 * The NONE / BIAS results are bit-identical across variants (same products, same k order per accumulator).  The GELU epilogues of
 * variant 7 start from the bf16-ROUNDED pre-activation / gradient (what an unfused Linear -> GELU pair computes), those of variants
 * 1..5 from the fp32 values: equal up to that one rounding.  Variant 7 writes its GRIT_GEMM_DGELU column sums as 2 * ceil(M / 256) rows (one per
 * 128-row wave block of its 256-row tiles; every row written).
 * ------------------------------------------------------------------------------------------------------ */
#define GRIT_GEMM_NONE 0
#define GRIT_GEMM_BIAS 1
#define GRIT_GEMM_BIAS_GELU 2
#define GRIT_GEMM_DGELU 3
#define GRIT_GEMM_BIAS_RES 4   /* grit_gemm_bf16_nt_res only */
#define GRIT_GEMM_BIAS_RELU_DROP 5   /* grit_gemm_bf16_nt_relu only */
#define GRIT_GEMM_DRELU 6            /* grit_gemm_bf16_nt_relu only */
#define GRIT_GEMM_COLSUM_ROWS 128
int grit_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                      int epilogue, const void* bias, void* aux, long ldaux, float* colsum, int variant, void* stream);
/* variant 9 = variant 7 with the tile height chosen by shape: 256 rows, or 224 where that needs fewer 16-row block rows per CU
 * (the N = C products of the Swin blocks -- proj, fc2 forward, fc1 / qkv input gradients: 400 / 800 / 200 tiles of 256 rows fill the
 * 256 CUs 1.56 / 3.1 / 0.78 times, 458 / 915 / 232 tiles of 224 rows 1.79 / 3.6 / 0.91 times: 12.5 % less time, same rounds).  Same
 * products in the same k order: results bit-identical to variant 7.  grit_gemm_w4_tile_rows(M, N) = the height variant 9 runs for a
 * shape on the current device; its GRIT_GEMM_DGELU column sums are 2 * ceil(M / height) rows. */
int grit_gemm_w4_tile_rows(int M, int N);
/* The output projection of a Swin branch WITH its residual connection (models/common/swin_model.py:289-298:
 * x = shortcut + drop_path(proj(...)) / x = x + drop_path(mlp(...))):
 *     C[m, n] = residual[m, n] + row_scale[m / rows_per_sample] * bf16(acc[m, n] + bias[n])
 * -- the branch rounded to bf16 as an unfused Linear stores it, product and sum in fp32 (not fused), one rounding: bit for bit what
 * grit_add_layernorm_fwd computes as its `sum_out` from a stored branch.  row_scale NULL = 1 (then rows_per_sample is ignored);
 * otherwise rows_per_sample >= 256.  The branch map is never written or re-read.  N % 256 == 0 and K % 64 == 0: variant 7's kernel and
 * shape limits; other N % 128 == 0, K % 32 == 0 (the 128 output columns of the stage-0 map): the per-tile kernel on 256 x 128 tiles. */
int grit_gemm_bf16_nt_res(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                          const void* bias, const void* residual, long ldres, const float* row_scale, int rows_per_sample,
                          void* stream);
/* The position-wise FFNs of the two decoders and the grid net (models/detection/det_module.py:302-304, models/common/pos_embed.py:44-48:
 * Linear -> ReLU -> dropout -> Linear) on the short-map tiles (variant 12's 64 x 64 x 64; N % 64 == 0, K % 64 == 0), ReLU + dropout in the
 * epilogues instead of a launch each way:
 *   GRIT_GEMM_BIAS_RELU_DROP   C = dropout(relu(bf16(A B^T + bias)), drop_p): the first Linear's forward; keep factors from the hash of
 *                              grit_relu_dropout_fwd over the element index m * N + n (ldc == N), seed read from seed_dev when drop_p > 0;
 *                              bit for bit Linear (bf16 result) followed by grit_relu_dropout_fwd.  aux unused.
 *   GRIT_GEMM_DRELU            C = aux > 0 ? bf16(A B^T) / (1 - drop_p) : 0, aux [M, N] = the forward's C (positive exactly where the unit
 *                              was active and kept): the second Linear's input gradient with the backward of ReLU + dropout applied;
 *                              bit for bit the stored bf16 product followed by grit_relu_dropout_bwd.  bias, seed_dev unused. */
int grit_gemm_bf16_nt_relu(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                           const void* bias, const void* aux, long ldaux, float drop_p, const uint64_t* seed_dev, void* stream);
/* The fused Mlp GEMMs with the per-sample drop-path factors of the Swin blocks at hand (models/common/swin_model.py:289-298:
 * x = shortcut + drop_path(mlp(norm2(x))); a dropped sample's branch contributes nothing forward and receives an exactly zero
 * gradient): row_scale [ceil(M / rows_per_sample)] f32 on the device; a 256-row tile that lies inside ONE sample with factor 0 is
 * not computed.  GRIT_GEMM_DGELU (backward: the rows of A are already zero): C and its column sums are written as
 * zeros -- the same results as grit_gemm_bf16_nt.  GRIT_GEMM_BIAS_GELU (forward): C and aux are written as zeros instead of the values
 * nobody will use (the caller multiplies the branch by the same factor 0; the saved tensors meet zero gradients).  Eight-wave
 * variants (0..4); row_scale NULL = no skipping. */
int grit_gemm_bf16_nt_rows(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                           const void* bias, void* aux, long ldaux, float* colsum, const float* row_scale, int rows_per_sample,
                           int variant, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Beam-search candidate selection (reference models/caption/transformer.py:184-188: descending sort of the flattened
 * [beam * vocabulary] candidates, first `beam` kept).  x [rows, n] float32 with row stride ld (elements); the k <= 8 best of
 * every row, best first: idx_out [rows, k] int64 (position in the row), val_out [rows, k].  Equal values: lower index first;
 * NaN ranks above every number (torch's order).  Safe to capture in a HIP graph (no workspace, no host state).
 * ------------------------------------------------------------------------------------------------------ */
int grit_topk_rows_f32(const float* x, long ld, int rows, int n, int k, int64_t* idx_out, float* val_out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * One beam-search step after the decoder (reference models/caption/transformer.py:208-240, `iter`), for B images with
 * cur_beam live beams each and a vocabulary of V words.  logp [B * cur_beam, V] float32 word log-probabilities (row stride ld),
 * seq_logprob [B * cur_beam] running scores, seq_mask [B * cur_beam] 1 while a beam is alive, prev_words [B * cur_beam] int64
 * the words chosen at the previous step (both ignored when first_step != 0: every beam alive).
 *     alive      = seq_mask * (prev_words != eos)
 *     candidate  = alive ? seq_logprob + logp : (word == 0 ? seq_logprob : -999)
 *     the k <= 8 best candidates of each image over (beam, word), best first; equal scores by ascending beam * V + word
 * Outputs, all [B, k]: sel_beam / sel_word int64, new_seq_logprob (the candidate score), new_seq_mask (alive of the source beam),
 * picked_logprob = logp[sel_beam][sel_word] * alive.  Bit-identical to the reference's composed arithmetic.
 * workspace: grit_beam_step_workspace(B, cur_beam, k) bytes of device memory, no initialisation needed.  Limits: cur_beam * k <= 64
 * (16 at first_step), B <= 65535.  Two launches; safe to capture in a HIP graph.
 * ------------------------------------------------------------------------------------------------------ */
long grit_beam_step_workspace(int B, int cur_beam, int k);
int grit_beam_step_f32(const float* logp, long ld, const float* seq_logprob, const float* seq_mask, const int64_t* prev_words,
                       int eos, int first_step, int B, int cur_beam, int V, int k, void* workspace, long workspace_bytes,
                       int64_t* sel_beam, int64_t* sel_word, float* new_seq_logprob, float* new_seq_mask, float* picked_logprob,
                       void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Gated merge of the grid / region cross-attentions of a caption decoder layer, inference only (reference
 * models/caption/cap_generator.py:44-56).  All tensors [rows, d] row-major of one dtype (is_bf16: bfloat16, else float32),
 * d % 8 == 0, 16-byte aligned; mask_pad [rows] of the same dtype (1 = real token, 0 = padding).
 *   grit_gate_pack: X [2 * rows, 2 * d] with X[r] = (self_att[r], enc1[r] * mask_pad[r]), X[rows + r] = (self_att[r], enc2[r] * mask_pad[r])
 *                   -- the two inputs of fc_alpha1, stacked so that ONE GEMM produces both gate pre-activations.
 *   grit_gate_fuse: gates [2 * rows, d] = fc_alpha1(X);  out[r] = ((enc1*m * sigmoid(gates[r]) + enc2*m * sigmoid(gates[rows + r]))
 *                   * (1 / divisor)) * m, every step rounded to the tensor dtype as the composed torch form rounds it.
 * ------------------------------------------------------------------------------------------------------ */
int grit_gate_pack(const void* self_att, const void* enc1, const void* enc2, const void* mask_pad, int rows, int d, int is_bf16,
                   void* X, void* stream);
int grit_gate_fuse(const void* enc1, const void* enc2, const void* gates, const void* mask_pad, int rows, int d, float divisor,
                   int is_bf16, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Element-wise glue of the decoders in the TRAINING step, one launch each instead of a chain of 4-14 torch launches at the ~6 us
 * dependent-launch floor (grit_amd/csrc/glue.hip).  fp32 arithmetic in the reference's operation order, outputs rounded once.
 *
 * grit_msda_geometry_fwd / _bwd -- MSDeformAttn.forward's query-side arithmetic, models/ops/modules/ms_deform_attn.py:97-113:
 *   offsets [rows, M, L, P, 2], logits [rows, M, L*P] (bf16 or f32: the outputs of sampling_offsets / attention_weights; rows =
 *   B * Lq), ref [rows, L, ref_dim] f32 (ref_dim 4: box cx, cy, w, h -> loc = ref_xy + offset / P * wh * 0.5; ref_dim 2: point ->
 *   loc = ref_xy + offset / (W_l, H_l), spatial_shapes [L, 2] int64 (H, W)), L*P <= 64
 *     -> loc [rows, M, L, P, 2] f32, attn_w [rows, M, L*P] f32 = softmax(logits) over the L*P points.
 *   backward: grad_offsets / grad_logits (bf16 or f32) from grad_loc, grad_attn_w, attn_w; ref receives no gradient (the
 *   reference detaches it: det_module.py:52).
 * grit_box_refine -- DetectionModule.bbox_refine, det_module.py:40-53: out [rows, 4] f32 = sigmoid(delta + inverse_sigmoid(ref))
 *   (ref_dim 4) or sigmoid(cat(delta[:2] + inverse_sigmoid(ref), delta[2:])) (ref_dim 2); delta [rows, 4] bf16 / f32.
 * grit_relu_dropout_fwd / _bwd -- dropout(relu(x)) of the position-wise FFNs (det_module.py:302-304, pos_embed.py:44-48), n
 *   elements; the keep mask is a counter hash of (element index, *seed_dev) regenerated in the backward (dx = dy * keep / (1 - p)
 *   where x > 0); p = 0: plain ReLU and its gradient.
 * grit_gate_bwd_a / _b -- backward of the gated merge of models/caption/cap_generator.py:44-56 around the fc_alpha1 GEMMs (forward:
 *   grit_gate_pack, GEMM, grit_gate_fuse): a: d_gates [2 rows, d] from d_out, enc1, enc2, gates, mask_pad; b: d_self, d_enc1,
 *   d_enc2 [rows, d] from d_out, gates, mask_pad and dX [2 rows, 2 d] = d_gates @ W.  One dtype (bf16 / f32) for all tensors.
 * ------------------------------------------------------------------------------------------------------ */
int grit_msda_geometry_fwd(const void* offsets, const void* logits, int in_is_bf16, const float* ref, int ref_dim,
                           const int64_t* spatial_shapes, long rows, int M, int L, int P, float* loc, float* attn_w, void* stream);
int grit_msda_geometry_bwd(const float* grad_loc, const float* grad_attn_w, const float* attn_w, const float* ref, int ref_dim,
                           const int64_t* spatial_shapes, long rows, int M, int L, int P, int out_is_bf16, void* grad_offsets,
                           void* grad_logits, void* stream);
int grit_box_refine(const void* delta, int delta_is_bf16, const float* ref, int ref_dim, long rows, float* out, void* stream);
int grit_relu_dropout_fwd(const void* x, long n, float p, const uint64_t* seed_dev, int is_bf16, void* y, void* stream);
int grit_relu_dropout_bwd(const void* x, const void* dy, long n, float p, const uint64_t* seed_dev, int is_bf16, void* dx, void* stream);
int grit_gate_bwd_a(const void* d_out, const void* enc1, const void* enc2, const void* gates, const void* mask_pad, long rows, int d,
                    float divisor, int is_bf16, void* d_gates, void* stream);
int grit_gate_bwd_b(const void* d_out, const void* gates, const void* mask_pad, const void* dX, long rows, int d, float divisor,
                    int is_bf16, void* d_self, void* d_enc1, void* d_enc2, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Key / value cache of step-wise decoding under beam search (reference models/common/attention.py:166-181 appends with torch.cat,
 * models/caption/transformer.py:229 re-gathers every state by the surviving beam): for B images with `beam` surviving beams each,
 *     out[(b, j)][0 .. t_old) = old[(b, src_beam[b][j])][0 .. t_old)      (old holds cur_beam histories per image)
 *     out[(b, j)][t_old]      = new[(b, j)]
 * for keys and values in one launch.  old_* [B * cur_beam, t_old, row_bytes] contiguous (ignored when t_old == 0), new_* one row per
 * (b, j) at new_row_stride_bytes (a slice of the fused q/k/v projection), out_* [B * beam, t_old + 1, row_bytes] contiguous.
 * src_beam [B * beam] int64 (index of the source beam within the image) or NULL = identity (then cur_beam must equal beam).
 * Byte copies: any dtype; row_bytes, strides and bases 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------ */
int grit_kv_append(const void* old_k, const void* old_v, const int64_t* src_beam, int B, int cur_beam, int beam, int t_old,
                   int row_bytes, const void* new_k, const void* new_v, long new_row_stride_bytes, void* out_k, void* out_v,
                   void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Inputs of one step of step-wise decoding (reference models/caption/cap_generator.py:116-137 get_seq_inputs in stateful mode and the
 * embedding sum of :148).  tokens [rows] int64; word_emb [vocab, d], pos_emb [n_pos, d] (is_bf16: bfloat16, else float32);
 * running_seq [rows] int64 step counters, advanced in place; old_mask [rows, t_old] bytes (1 = masked key), ignored when t_old == 0.
 *     pos = ++running_seq[r];  x[r] = word_emb[tokens[r]] + pos_emb[pos];  mask_pad[r] = tokens[r] != pad_idx (as 1 / 0 in the table dtype);
 *     new_mask[r] = (old_mask[r], tokens[r] == pad_idx)            [rows, t_old + 1]
 * The caller checks that the step count stays inside pos_emb (the kernel clamps instead of faulting).
 * ------------------------------------------------------------------------------------------------------ */
int grit_decode_step_inputs(const int64_t* tokens, int64_t pad_idx, const void* word_emb, int vocab, const void* pos_emb, int n_pos,
                            int d, int is_bf16, int64_t* running_seq, const uint8_t* old_mask, int t_old, int rows, void* x,
                            void* mask_pad, uint8_t* new_mask, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRIT_HIP_H */
