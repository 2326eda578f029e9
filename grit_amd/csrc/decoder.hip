// Element-wise fusion of the caption decoder's gated cross-attention merge at inference (davidnvq/grit
// models/caption/cap_generator.py:44-56, ParallelAttentionLayer.forward):
//
//     enc1 = vis_att1(...) * mask_pad;  enc2 = vis_att2(...) * mask_pad
//     gate1 = sigmoid(fc_alpha1(cat[self_att, enc1]));  gate2 = sigmoid(fc_alpha1(cat[self_att, enc2]))
//     fused = (enc1 * gate1 + enc2 * gate2) / sqrt(2);   ff_in = fused * mask_pad
//
// Fourteen launches per layer in the composed form (two masks, two cats, two GEMMs, two sigmoids, two products, sum, scale,
// mask); here gate_pack writes both GEMM inputs stacked ([2R, 2d]: ONE GEMM with fc_alpha1), gate_fuse does the rest.  Every
// intermediate is rounded to the tensor dtype where the composed form rounds it (one torch kernel = one rounding) and products
// and sums are kept un-contracted: in bf16 the result is the composed form's bit for bit (tests/test_gate_gpu.py), in fp32 to the
// last place of exp / divide; the scale is the multiplication by 1.0f / float(sqrt(2)) torch performs for a division by a
// Python scalar.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

template <typename T> struct Vec;
template <> struct Vec<float> { static constexpr int n = 4; };
template <> struct Vec<__bf16> { static constexpr int n = 8; };

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__bf16>(__bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __bf16 from_f<__bf16>(float v) { return (__bf16)v; }  // round to nearest even

// The rounding a separate torch kernel would apply to its output.  For bf16 it is spelled on the bit pattern (round to nearest
// even): a float -> __bf16 -> float cast pair is NOT a rounding under clang's default -fbfloat16-excess-precision=fast, which
// keeps the float (measured: 21 % of the outputs off by up to a few bf16 ulps where the two products cancel).
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<__bf16>(float v) {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return v;  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
}

template <typename T>
__global__ __launch_bounds__(256)
void gate_pack(const T* __restrict__ self_att, const T* __restrict__ enc1, const T* __restrict__ enc2,
               const T* __restrict__ mask_pad, int R, int d, T* __restrict__ X) {
    constexpr int n = Vec<T>::n;
    const int per_row = d / n;
    const long total = (long)R * per_row;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per_row), c = (int)(i - (long)r * per_row) * n;
        const float m = to_f<T>(mask_pad[r]);
        T s[n], a[n], b[n];
        *reinterpret_cast<uint4*>(s) = *reinterpret_cast<const uint4*>(self_att + (size_t)r * d + c);
        *reinterpret_cast<uint4*>(a) = *reinterpret_cast<const uint4*>(enc1 + (size_t)r * d + c);
        *reinterpret_cast<uint4*>(b) = *reinterpret_cast<const uint4*>(enc2 + (size_t)r * d + c);
#pragma unroll
        for (int e = 0; e < n; ++e) {
            a[e] = from_f<T>(rnd<T>(__fmul_rn(to_f<T>(a[e]), m)));
            b[e] = from_f<T>(rnd<T>(__fmul_rn(to_f<T>(b[e]), m)));
        }
        T* x1 = X + (size_t)r * 2 * d;
        T* x2 = X + ((size_t)R + r) * 2 * d;
        *reinterpret_cast<uint4*>(x1 + c) = *reinterpret_cast<uint4*>(s);
        *reinterpret_cast<uint4*>(x1 + d + c) = *reinterpret_cast<uint4*>(a);
        *reinterpret_cast<uint4*>(x2 + c) = *reinterpret_cast<uint4*>(s);
        *reinterpret_cast<uint4*>(x2 + d + c) = *reinterpret_cast<uint4*>(b);
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void gate_fuse(const T* __restrict__ enc1, const T* __restrict__ enc2, const T* __restrict__ G, const T* __restrict__ mask_pad,
               int R, int d, float inv_scale, T* __restrict__ out) {
#pragma clang fp contract(off)  // fp32: a product and the following sum are two torch kernels, never one fma
    constexpr int n = Vec<T>::n;
    const int per_row = d / n;
    const long total = (long)R * per_row;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per_row), c = (int)(i - (long)r * per_row) * n;
        const float m = to_f<T>(mask_pad[r]);
        T a[n], b[n], g1[n], g2[n], o[n];
        *reinterpret_cast<uint4*>(a) = *reinterpret_cast<const uint4*>(enc1 + (size_t)r * d + c);
        *reinterpret_cast<uint4*>(b) = *reinterpret_cast<const uint4*>(enc2 + (size_t)r * d + c);
        *reinterpret_cast<uint4*>(g1) = *reinterpret_cast<const uint4*>(G + (size_t)r * d + c);
        *reinterpret_cast<uint4*>(g2) = *reinterpret_cast<const uint4*>(G + ((size_t)R + r) * d + c);
#pragma unroll
        for (int e = 0; e < n; ++e) {
            const float e1 = rnd<T>(__fmul_rn(to_f<T>(a[e]), m)), e2 = rnd<T>(__fmul_rn(to_f<T>(b[e]), m));
            const float s1 = rnd<T>(1.0f / (1.0f + expf(-to_f<T>(g1[e])))), s2 = rnd<T>(1.0f / (1.0f + expf(-to_f<T>(g2[e]))));
            const float p1 = rnd<T>(__fmul_rn(e1, s1)), p2 = rnd<T>(__fmul_rn(e2, s2));
            const float sum = rnd<T>(__fadd_rn(p1, p2));
            const float scaled = rnd<T>(__fmul_rn(sum, inv_scale));
            o[e] = from_f<T>(rnd<T>(__fmul_rn(scaled, m)));
        }
        *reinterpret_cast<uint4*>(out + (size_t)r * d + c) = *reinterpret_cast<uint4*>(o);
    }
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Key / value cache of step-wise decoding under beam search: the surviving beams take over the history of their source beam and
// the new token's projected key / value is appended -- the re-gather of `running_keys` / `running_values` by the selected beam
// (transformer.py:229 through containers.py apply_to_states) and the two torch.cat of attention.py:166-181, four launches per
// layer and step, as one copy kernel.  out[(b, j)][0 .. t) = old[(b, src[b][j])][0 .. t),  out[(b, j)][t] = new[(b, j)].
__global__ __launch_bounds__(256)
void kv_append(const uint4* __restrict__ old_k, const uint4* __restrict__ old_v, const int64_t* __restrict__ src, int cur, int beam,
               int t_old, int row16, const uint4* __restrict__ new_k, const uint4* __restrict__ new_v, long new_stride16,
               long rows, uint4* __restrict__ out_k, uint4* __restrict__ out_v) {
    const long per_row = (long)(t_old + 1) * row16;
    const long total = rows * per_row;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < 2 * total; i += (long)gridDim.x * blockDim.x) {
        const bool is_v = i >= total;
        const long e = is_v ? i - total : i;
        const long r = e / per_row, within = e - r * per_row;
        const int t = (int)(within / row16), c = (int)(within - (long)t * row16);
        uint4 val;
        if (t < t_old) {
            const long b = r / beam;
            const long from = src ? b * cur + src[r] : r;
            val = (is_v ? old_v : old_k)[(from * t_old + t) * row16 + c];
        } else {
            val = (is_v ? new_v : new_k)[r * new_stride16 + c];
        }
        (is_v ? out_v : out_k)[e] = val;
    }
}

// Inputs of one step of step-wise decoding (davidnvq/grit models/caption/cap_generator.py:116-137, get_seq_inputs in stateful mode +
// the two embedding lookups of :148): for every row r with token tok,
//     pos = ++running_seq[r];   x[r] = word_emb[tok] + pos_emb[pos];   mask_pad[r] = tok != pad
//     new_mask[r] = (old_mask[r][0 .. t_old), tok == pad)                       -- the self-attention key mask grows by one column
// Fourteen launches in the composed form (eq, not, float, triu of ones, or, cat, add_, two embeddings, add, casts), one here.
template <typename T>
__global__ __launch_bounds__(64)
void decode_step_inputs(const int64_t* __restrict__ tokens, int64_t pad_idx, const T* __restrict__ word_emb, int vocab,
                        const T* __restrict__ pos_emb, int n_pos, int d, int64_t* __restrict__ running_seq,
                        const uint8_t* __restrict__ old_mask, int t_old, T* __restrict__ x, T* __restrict__ mask_pad,
                        uint8_t* __restrict__ new_mask) {
#pragma clang fp contract(off)
    const int r = blockIdx.x, tid = threadIdx.x;
    const int64_t tok = tokens[r];
    int64_t pos = running_seq[r] + 1;
    __builtin_amdgcn_s_barrier();  // every lane has read the counter before lane 0 advances it (one wave per row)
    if (tid == 0) {
        running_seq[r] = pos;
        mask_pad[r] = from_f<T>(tok != pad_idx ? 1.0f : 0.0f);
        new_mask[(size_t)r * (t_old + 1) + t_old] = tok == pad_idx ? 1 : 0;
    }
    for (int i = tid; i < t_old; i += 64) new_mask[(size_t)r * (t_old + 1) + i] = old_mask[(size_t)r * t_old + i];
    if (pos >= n_pos) pos = n_pos - 1;  // the host checks the step count against the table; never read out of bounds
    // a token id outside the table (torch's embedding would raise) must never become an out-of-bounds read: such a row reads
    // row 0; ids come from this library's own beam step, so this is a guard, not a code path
    const int64_t tok_in = (tok >= 0 && tok < vocab) ? tok : 0;
    const T* w = word_emb + (size_t)tok_in * d;
    const T* p = pos_emb + (size_t)pos * d;
    for (int c = tid; c < d; c += 64) x[(size_t)r * d + c] = from_f<T>(rnd<T>(to_f<T>(w[c]) + to_f<T>(p[c])));
}

}  // namespace

extern "C" int grit_decode_step_inputs(const int64_t* tokens, int64_t pad_idx, const void* word_emb, int vocab, const void* pos_emb,
                                       int n_pos, int d, int is_bf16, int64_t* running_seq, const uint8_t* old_mask, int t_old,
                                       int rows, void* x, void* mask_pad, uint8_t* new_mask, void* stream) {
    if (!tokens || !word_emb || !pos_emb || !running_seq || !x || !mask_pad || !new_mask || rows <= 0 || d <= 0 || vocab <= 0 ||
        n_pos <= 0 || t_old < 0 || (t_old > 0 && !old_mask))
        return GRIT_ERR_BAD_ARG;
    if (is_bf16)
        hipLaunchKernelGGL(decode_step_inputs<__bf16>, dim3(rows), dim3(64), 0, (hipStream_t)stream, tokens, pad_idx,
                           (const __bf16*)word_emb, vocab, (const __bf16*)pos_emb, n_pos, d, running_seq, old_mask, t_old, (__bf16*)x,
                           (__bf16*)mask_pad, new_mask);
    else
        hipLaunchKernelGGL(decode_step_inputs<float>, dim3(rows), dim3(64), 0, (hipStream_t)stream, tokens, pad_idx,
                           (const float*)word_emb, vocab, (const float*)pos_emb, n_pos, d, running_seq, old_mask, t_old, (float*)x,
                           (float*)mask_pad, new_mask);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_gate_pack(const void* self_att, const void* enc1, const void* enc2, const void* mask_pad, int rows, int d,
                              int is_bf16, void* X, void* stream) {
    if (!self_att || !enc1 || !enc2 || !mask_pad || !X || rows <= 0 || d <= 0) return GRIT_ERR_BAD_ARG;
    if (d % 8 || !aligned16(self_att) || !aligned16(enc1) || !aligned16(enc2) || !aligned16(X)) return GRIT_ERR_UNSUPPORTED;
    const long total = (long)rows * (d / (is_bf16 ? 8 : 4));
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (is_bf16)
        hipLaunchKernelGGL(gate_pack<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)self_att, (const __bf16*)enc1, (const __bf16*)enc2,
                           (const __bf16*)mask_pad, rows, d, (__bf16*)X);
    else
        hipLaunchKernelGGL(gate_pack<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)self_att,
                           (const float*)enc1, (const float*)enc2, (const float*)mask_pad, rows, d, (float*)X);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_gate_fuse(const void* enc1, const void* enc2, const void* gates, const void* mask_pad, int rows, int d,
                              float divisor, int is_bf16, void* out, void* stream) {
    if (!enc1 || !enc2 || !gates || !mask_pad || !out || rows <= 0 || d <= 0 || !(divisor != 0.f)) return GRIT_ERR_BAD_ARG;
    if (d % 8 || !aligned16(enc1) || !aligned16(enc2) || !aligned16(gates) || !aligned16(out)) return GRIT_ERR_UNSUPPORTED;
    const float inv = 1.0f / divisor;
    const long total = (long)rows * (d / (is_bf16 ? 8 : 4));
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (is_bf16)
        hipLaunchKernelGGL(gate_fuse<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)enc1, (const __bf16*)enc2, (const __bf16*)gates,
                           (const __bf16*)mask_pad, rows, d, inv, (__bf16*)out);
    else
        hipLaunchKernelGGL(gate_fuse<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)enc1,
                           (const float*)enc2, (const float*)gates, (const float*)mask_pad, rows, d, inv, (float*)out);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_kv_append(const void* old_k, const void* old_v, const int64_t* src_beam, int B, int cur_beam, int beam, int t_old,
                              int row_bytes, const void* new_k, const void* new_v, long new_row_stride_bytes, void* out_k,
                              void* out_v, void* stream) {
    if (!new_k || !new_v || !out_k || !out_v || B <= 0 || cur_beam <= 0 || beam <= 0 || t_old < 0 || row_bytes <= 0)
        return GRIT_ERR_BAD_ARG;
    if (t_old > 0 && (!old_k || !old_v)) return GRIT_ERR_BAD_ARG;
    if (!src_beam && cur_beam != beam) return GRIT_ERR_BAD_ARG;
    if (row_bytes % 16 || new_row_stride_bytes % 16 || !aligned16(old_k) || !aligned16(old_v) || !aligned16(new_k) ||
        !aligned16(new_v) || !aligned16(out_k) || !aligned16(out_v))
        return GRIT_ERR_UNSUPPORTED;
    const long rows = (long)B * beam;
    const long total = 2 * rows * (t_old + 1) * (row_bytes / 16);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(kv_append, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)old_k, (const uint4*)old_v, src_beam,
                       cur_beam, beam, t_old, row_bytes / 16, (const uint4*)new_k, (const uint4*)new_v, new_row_stride_bytes / 16,
                       rows, (uint4*)out_k, (uint4*)out_v);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
