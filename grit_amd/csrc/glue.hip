// Element-wise glue of the two decoders in the TRAINING step, fused (gfx950).  None of these kernels moves more than a few
// megabytes: each replaces a chain of 4-14 torch element-wise launches that cost ~6 us apiece at the dependent-launch floor
// (profiles/r03/steady_*: ~500 such launches per step, the decoder phase is ~13 ms of GPU time for 2 % of the step's flops).
//
//   msda_geometry_{fwd,bwd}   MSDeformAttn.forward's query-side arithmetic (reference models/ops/modules/ms_deform_attn.py:97-113):
//                             offsets.view / softmax(weights) / reference point + offset / P * wh * 0.5 (4-d refs) or + offset /
//                             (W_l, H_l) (2-d refs), and its gradient: 7 + 9 launches -> 1 + 1
//   box_refine                sigmoid(delta + inverse_sigmoid(ref)) of DetectionModule.bbox_refine (models/detection/det_module.py:
//                             40-53; no gradient: the result is detached there): 9 launches -> 1
//   relu_dropout_{fwd,bwd}    dropout(relu(x)) of the position-wise FFNs (det_module.py:302-304, models/common/pos_embed.py:44-48)
//                             with the keep mask regenerated from a device seed in the backward: 2 + 2 launches -> 1 + 1
//   gate_bwd_{a,b}            backward of the sigmoid-gated merge of the two cross-attentions (models/caption/cap_generator.py:44-56)
//                             around ONE fc_alpha1 GEMM pair (forward: grit_gate_pack / GEMM / grit_gate_fuse of decoder.hip):
//                             ~25 launches -> 2 + the GEMMs
// Arithmetic in fp32 on the loaded values, in the reference's operation order; outputs rounded once to their dtype.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<__bf16>(const __bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<__bf16>(__bf16* p, float v) { *p = (__bf16)v; }

// ---------------------------------------------------------------------------------------------------------------------
// MSDeformAttn query-side geometry.  One group of G lanes (G = 16, 32 or 64 >= L*P) per (row r = b*Lq + q, head m); lane = point.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int G>
__global__ __launch_bounds__(256)
void msda_geometry_fwd(const T* __restrict__ offsets, const T* __restrict__ logits, const float* __restrict__ ref, int ref_dim,
                       const int64_t* __restrict__ shapes, long rows_heads, int M, int L, int P, float* __restrict__ loc,
                       float* __restrict__ aw) {
    const int LP = L * P;
    const long grp = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const int j = threadIdx.x & (G - 1);
    if (grp >= rows_heads) return;
    const long r = grp / M;
    const bool live = j < LP;
    const int l = live ? j / P : 0;
    float logit = live ? ldf<T>(logits + grp * LP + j) : -INFINITY;
    float mx = logit;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float e = live ? __expf(logit - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (!live) return;
    aw[grp * LP + j] = e / sum;
    const float ox = ldf<T>(offsets + (grp * LP + j) * 2), oy = ldf<T>(offsets + (grp * LP + j) * 2 + 1);
    const float* rp = ref + (r * L + l) * ref_dim;
    float x, y;
    if (ref_dim == 4) {  // reference box (cx, cy, w, h): ref_xy + offset / P * wh * 0.5
        x = rp[0] + ox / (float)P * rp[2] * 0.5f;
        y = rp[1] + oy / (float)P * rp[3] * 0.5f;
    } else {             // reference point: ref_xy + offset / (W_l, H_l)
        x = rp[0] + ox / (float)shapes[2 * l + 1];
        y = rp[1] + oy / (float)shapes[2 * l];
    }
    loc[(grp * LP + j) * 2] = x;
    loc[(grp * LP + j) * 2 + 1] = y;
}

template <typename T, int G>
__global__ __launch_bounds__(256)
void msda_geometry_bwd(const float* __restrict__ grad_loc, const float* __restrict__ grad_aw, const float* __restrict__ aw,
                       const float* __restrict__ ref, int ref_dim, const int64_t* __restrict__ shapes, long rows_heads, int M, int L,
                       int P, T* __restrict__ grad_offsets, T* __restrict__ grad_logits) {
    const int LP = L * P;
    const long grp = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const int j = threadIdx.x & (G - 1);
    if (grp >= rows_heads) return;
    const long r = grp / M;
    const bool live = j < LP;
    const int l = live ? j / P : 0;
    const float a = live ? aw[grp * LP + j] : 0.f, ga = live ? grad_aw[grp * LP + j] : 0.f;
    float dot = a * ga;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (!live) return;
    stf<T>(grad_logits + grp * LP + j, a * (ga - dot));  // softmax backward
    const float gx = grad_loc[(grp * LP + j) * 2], gy = grad_loc[(grp * LP + j) * 2 + 1];
    const float* rp = ref + (r * L + l) * ref_dim;
    float dx, dy;
    if (ref_dim == 4) {
        dx = gx * 0.5f * rp[2] / (float)P;
        dy = gy * 0.5f * rp[3] / (float)P;
    } else {
        dx = gx / (float)shapes[2 * l + 1];
        dy = gy / (float)shapes[2 * l];
    }
    stf<T>(grad_offsets + (grp * LP + j) * 2, dx);
    stf<T>(grad_offsets + (grp * LP + j) * 2 + 1, dy);
}

// ---------------------------------------------------------------------------------------------------------------------
// Box refinement: new = sigmoid(delta + inverse_sigmoid(ref)); inverse_sigmoid(x) = log(clamp(x, 0, 1).clamp(min eps) /
// (1 - clamp(x, 0, 1)).clamp(min eps)), eps = 1e-5 (utils/misc.py inverse_sigmoid).  2-d references refine the centre only.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float inv_sigmoid(float x) {
    x = fminf(fmaxf(x, 0.f), 1.f);
    return logf(fmaxf(x, 1e-5f) / fmaxf(1.f - x, 1e-5f));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256)
void box_refine(const T* __restrict__ delta, const float* __restrict__ ref, int ref_dim, long rows, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * 4) return;
    const long r = i >> 2;
    const int c = (int)(i & 3);
    float v = ldf<T>(delta + i);
    if (c < ref_dim) v += inv_sigmoid(ref[r * ref_dim + c]);
    out[i] = sigmoidf(v);
}

// ---------------------------------------------------------------------------------------------------------------------
// dropout(relu(x)): keep factor from the counter hash of layernorm.hip / attn.hip (murmur3 finaliser over element index + seed)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ull;
    unsigned int x = (unsigned int)(z ^ (z >> 32));
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256)
void relu_dropout(const T* __restrict__ x, const T* __restrict__ dy, long n, float p, const unsigned long long* __restrict__ seed_dev,
                  T* __restrict__ out) {
    const unsigned long long seed = p > 0.f ? *seed_dev : 0ull;
    const float inv_keep = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = ldf<T>(x + i);
        const float k = p > 0.f ? keep_scale(seed, (unsigned long long)i, p, inv_keep) : 1.0f;
        if (BWD) stf<T>(out + i, v > 0.f ? ldf<T>(dy + i) * k : 0.f);
        else stf<T>(out + i, v > 0.f ? v * k : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the gated merge.  Forward (decoder.hip): X = [(s, e1); (s, e2)], G = fc(X), out = ((e1 s(G1) + e2 s(G2)) * c) * m with
// e_i = enc_i * m, c = 1 / sqrt 2, s = sigmoid.  Given d_out:
//   a:  t = d_out * m * c;  dG1 = t * e1 * s(G1) (1 - s(G1)),  dG2 likewise                      -> dG [2R, d] (input of the GEMMs)
//   b:  dX = dG @ W  [2R, 2d];  d_self = dX[r, :d] + dX[R + r, :d];  d_enc_i = (t * s(G_i) + dX[(i-1) R + r, d:]) * m
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void gate_bwd_a(const T* __restrict__ d_out, const T* __restrict__ enc1, const T* __restrict__ enc2, const T* __restrict__ G,
                const T* __restrict__ mask_pad, long rows, int d, float c, T* __restrict__ dG) {
    const long n = rows * d;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / d;
        const float m = ldf<T>(mask_pad + r);
        const float t = ldf<T>(d_out + i) * m * c;
        const float s1 = sigmoidf(ldf<T>(G + i)), s2 = sigmoidf(ldf<T>(G + n + i));
        stf<T>(dG + i, t * (ldf<T>(enc1 + i) * m) * s1 * (1.f - s1));
        stf<T>(dG + n + i, t * (ldf<T>(enc2 + i) * m) * s2 * (1.f - s2));
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void gate_bwd_b(const T* __restrict__ d_out, const T* __restrict__ G, const T* __restrict__ mask_pad, const T* __restrict__ dX,
                long rows, int d, float c, T* __restrict__ d_self, T* __restrict__ d_enc1, T* __restrict__ d_enc2) {
    const long n = rows * d;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / d;
        const int col = (int)(i - r * d);
        const float m = ldf<T>(mask_pad + r);
        const float t = ldf<T>(d_out + i) * m * c;
        const float s1 = sigmoidf(ldf<T>(G + i)), s2 = sigmoidf(ldf<T>(G + n + i));
        const T* xa = dX + r * 2 * d + col;
        const T* xb = dX + (rows + r) * 2 * d + col;
        stf<T>(d_self + i, ldf<T>(xa) + ldf<T>(xb));
        stf<T>(d_enc1 + i, (t * s1 + ldf<T>(xa + d)) * m);
        stf<T>(d_enc2 + i, (t * s2 + ldf<T>(xb + d)) * m);
    }
}

int blocks_for(long n, int cap = 4096) {
    const long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b < cap ? b : cap));
}

}  // namespace

extern "C" {

int grit_msda_geometry_fwd(const void* offsets, const void* logits, int in_is_bf16, const float* ref, int ref_dim,
                           const int64_t* spatial_shapes, long rows, int M, int L, int P, float* loc, float* attn_w, void* stream) {
    if (!offsets || !logits || !ref || !spatial_shapes || !loc || !attn_w || rows <= 0 || M <= 0 || L <= 0 || P <= 0)
        return GRIT_ERR_BAD_ARG;
    if ((ref_dim != 2 && ref_dim != 4) || L * P > 64) return GRIT_ERR_UNSUPPORTED;
    const int LP = L * P, G = LP <= 16 ? 16 : (LP <= 32 ? 32 : 64);
    const long rh = rows * M;
    const dim3 grid((unsigned)((rh * G + 255) / 256)), block(256);
#define GRIT_GEO_FWD(T_, G_)                                                                                                   \
    hipLaunchKernelGGL((msda_geometry_fwd<T_, G_>), grid, block, 0, (hipStream_t)stream, (const T_*)offsets, (const T_*)logits, ref, \
                       ref_dim, spatial_shapes, rh, M, L, P, loc, attn_w)
    if (in_is_bf16) { if (G == 16) GRIT_GEO_FWD(__bf16, 16); else if (G == 32) GRIT_GEO_FWD(__bf16, 32); else GRIT_GEO_FWD(__bf16, 64); }
    else { if (G == 16) GRIT_GEO_FWD(float, 16); else if (G == 32) GRIT_GEO_FWD(float, 32); else GRIT_GEO_FWD(float, 64); }
#undef GRIT_GEO_FWD
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_msda_geometry_bwd(const float* grad_loc, const float* grad_attn_w, const float* attn_w, const float* ref, int ref_dim,
                           const int64_t* spatial_shapes, long rows, int M, int L, int P, int out_is_bf16, void* grad_offsets,
                           void* grad_logits, void* stream) {
    if (!grad_loc || !grad_attn_w || !attn_w || !ref || !spatial_shapes || !grad_offsets || !grad_logits || rows <= 0 || M <= 0 ||
        L <= 0 || P <= 0)
        return GRIT_ERR_BAD_ARG;
    if ((ref_dim != 2 && ref_dim != 4) || L * P > 64) return GRIT_ERR_UNSUPPORTED;
    const int LP = L * P, G = LP <= 16 ? 16 : (LP <= 32 ? 32 : 64);
    const long rh = rows * M;
    const dim3 grid((unsigned)((rh * G + 255) / 256)), block(256);
#define GRIT_GEO_BWD(T_, G_)                                                                                                   \
    hipLaunchKernelGGL((msda_geometry_bwd<T_, G_>), grid, block, 0, (hipStream_t)stream, grad_loc, grad_attn_w, attn_w, ref, ref_dim, \
                       spatial_shapes, rh, M, L, P, (T_*)grad_offsets, (T_*)grad_logits)
    if (out_is_bf16) { if (G == 16) GRIT_GEO_BWD(__bf16, 16); else if (G == 32) GRIT_GEO_BWD(__bf16, 32); else GRIT_GEO_BWD(__bf16, 64); }
    else { if (G == 16) GRIT_GEO_BWD(float, 16); else if (G == 32) GRIT_GEO_BWD(float, 32); else GRIT_GEO_BWD(float, 64); }
#undef GRIT_GEO_BWD
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_box_refine(const void* delta, int delta_is_bf16, const float* ref, int ref_dim, long rows, float* out, void* stream) {
    if (!delta || !ref || !out || rows <= 0) return GRIT_ERR_BAD_ARG;
    if (ref_dim != 2 && ref_dim != 4) return GRIT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((rows * 4 + 255) / 256)), block(256);
    if (delta_is_bf16) hipLaunchKernelGGL(box_refine<__bf16>, grid, block, 0, (hipStream_t)stream, (const __bf16*)delta, ref, ref_dim, rows, out);
    else hipLaunchKernelGGL(box_refine<float>, grid, block, 0, (hipStream_t)stream, (const float*)delta, ref, ref_dim, rows, out);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_relu_dropout_fwd(const void* x, long n, float p, const uint64_t* seed_dev, int is_bf16, void* y, void* stream) {
    if (!x || !y || n <= 0 || p < 0.f || p >= 1.f || (p > 0.f && !seed_dev)) return GRIT_ERR_BAD_ARG;
    const dim3 grid(blocks_for(n)), block(256);
    if (is_bf16)
        hipLaunchKernelGGL((relu_dropout<__bf16, false>), grid, block, 0, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)nullptr, n,
                           p, (const unsigned long long*)seed_dev, (__bf16*)y);
    else
        hipLaunchKernelGGL((relu_dropout<float, false>), grid, block, 0, (hipStream_t)stream, (const float*)x, (const float*)nullptr, n, p,
                           (const unsigned long long*)seed_dev, (float*)y);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_relu_dropout_bwd(const void* x, const void* dy, long n, float p, const uint64_t* seed_dev, int is_bf16, void* dx, void* stream) {
    if (!x || !dy || !dx || n <= 0 || p < 0.f || p >= 1.f || (p > 0.f && !seed_dev)) return GRIT_ERR_BAD_ARG;
    const dim3 grid(blocks_for(n)), block(256);
    if (is_bf16)
        hipLaunchKernelGGL((relu_dropout<__bf16, true>), grid, block, 0, (hipStream_t)stream, (const __bf16*)x, (const __bf16*)dy, n, p,
                           (const unsigned long long*)seed_dev, (__bf16*)dx);
    else
        hipLaunchKernelGGL((relu_dropout<float, true>), grid, block, 0, (hipStream_t)stream, (const float*)x, (const float*)dy, n, p,
                           (const unsigned long long*)seed_dev, (float*)dx);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_gate_bwd_a(const void* d_out, const void* enc1, const void* enc2, const void* gates, const void* mask_pad, long rows, int d,
                    float divisor, int is_bf16, void* d_gates, void* stream) {
    if (!d_out || !enc1 || !enc2 || !gates || !mask_pad || !d_gates || rows <= 0 || d <= 0 || !(divisor != 0.f)) return GRIT_ERR_BAD_ARG;
    const dim3 grid(blocks_for(rows * d)), block(256);
    const float c = 1.0f / divisor;
    if (is_bf16)
        hipLaunchKernelGGL(gate_bwd_a<__bf16>, grid, block, 0, (hipStream_t)stream, (const __bf16*)d_out, (const __bf16*)enc1,
                           (const __bf16*)enc2, (const __bf16*)gates, (const __bf16*)mask_pad, rows, d, c, (__bf16*)d_gates);
    else
        hipLaunchKernelGGL(gate_bwd_a<float>, grid, block, 0, (hipStream_t)stream, (const float*)d_out, (const float*)enc1,
                           (const float*)enc2, (const float*)gates, (const float*)mask_pad, rows, d, c, (float*)d_gates);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_gate_bwd_b(const void* d_out, const void* gates, const void* mask_pad, const void* dX, long rows, int d, float divisor,
                    int is_bf16, void* d_self, void* d_enc1, void* d_enc2, void* stream) {
    if (!d_out || !gates || !mask_pad || !dX || !d_self || !d_enc1 || !d_enc2 || rows <= 0 || d <= 0 || !(divisor != 0.f))
        return GRIT_ERR_BAD_ARG;
    const dim3 grid(blocks_for(rows * d)), block(256);
    const float c = 1.0f / divisor;
    if (is_bf16)
        hipLaunchKernelGGL(gate_bwd_b<__bf16>, grid, block, 0, (hipStream_t)stream, (const __bf16*)d_out, (const __bf16*)gates,
                           (const __bf16*)mask_pad, (const __bf16*)dX, rows, d, c, (__bf16*)d_self, (__bf16*)d_enc1, (__bf16*)d_enc2);
    else
        hipLaunchKernelGGL(gate_bwd_b<float>, grid, block, 0, (hipStream_t)stream, (const float*)d_out, (const float*)gates,
                           (const float*)mask_pad, (const float*)dX, rows, d, c, (float*)d_self, (float*)d_enc1, (float*)d_enc2);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // extern "C"
