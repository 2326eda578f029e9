// LayerNorm forward / backward for the Swin token maps on gfx950 (rows = B*H*W tokens, C in {128..4096}).
//
// torch's bf16 LayerNorm kernels reach ~0.9 TB/s on these shapes on MI355X (profiles/r01: 18 ms of a 138 ms step);
// the op is a pure HBM stream (read x, write y; backward read x, dy, write dx).  Here a row is spread over
// C/8 lanes (<= 64) with one 16-byte load per lane per 512-channel chunk, several rows per wavefront when C < 512,
// statistics in fp32 registers (two-pass on the register copy), wave-shuffle reductions, and -- backward -- dgamma /
// dbeta accumulated in registers across a persistent row loop and written as one partial row per workgroup
// (grit_slab_sum folds them).  Both directions optionally fuse the residual connection that precedes the norm in a
// Swin block (x = shortcut + drop_path(branch); y = LN(x)): forward forms, stores and normalises x in one pass, backward
// adds the gradient arriving at x through the skip path to the LayerNorm input gradient.
// Semantics: torch.nn.functional.layer_norm over the last dimension (reference modules use nn.LayerNorm:
// models/common/swin_model.py:229,233,315).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8], bool nt = false) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
        if (nt) {
            __builtin_nontemporal_store(a, reinterpret_cast<v4f*>(p));
            __builtin_nontemporal_store(b, reinterpret_cast<v4f*>(p + 4));
        } else {
            *reinterpret_cast<v4f*>(p) = a;
            *reinterpret_cast<v4f*>(p + 4) = b;
        }
    }
};
template <> struct Vec8<__hip_bfloat16> {
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8]) {
#ifdef GRIT_LN_NT_LOADS
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
        const uint32_t w[4] = {u[0], u[1], u[2], u[3]};
#else
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[8], bool nt = false) {
        typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
        v8bf o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
        // nt (wave-uniform, chosen per launch): maps of tens to hundreds of MB are written once and read by a later kernel from
        // HBM anyway; streamed past L2 they leave the neighbouring GEMMs' operand panels resident (63.5 -> 62.9 ms per training
        // step, A/B on one box).  Small maps (decoders, beam search) stay cached for their consumer.
        if (nt)
            __builtin_nontemporal_store(o, reinterpret_cast<v8bf*>(p));
        else
            *reinterpret_cast<v8bf*>(p) = o;
    }
};

template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<__hip_bfloat16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

// counter-based dropout keep factor (same construction as the attention kernels, attn.hip): murmur3 finaliser over
// (element index + seed mix); forward and backward regenerate the mask from the seed, nothing is stored
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ull;
    unsigned int x = (unsigned int)(z ^ (z >> 32));
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

// Sum over the LPR lanes of a row.  Not as __shfl_xor steps: each of those is a ds_bpermute round trip through the LDS pipe with
// a full lgkmcnt wait, twelve of them on the dependent chain load -> mean -> variance -> store of a row, and a wave of these
// kernels is nothing but that chain.  Within a DPP row (16 lanes) four row rotations (plain VALU); across the rows of a wave the
// four row sums are read into scalars.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
    if constexpr (LPR >= 16) {
#define GRIT_ROW_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, false))
        v += GRIT_ROW_ROR(v, 8); v += GRIT_ROW_ROR(v, 4); v += GRIT_ROW_ROR(v, 2); v += GRIT_ROW_ROR(v, 1);
#undef GRIT_ROW_ROR
        if constexpr (LPR == 16) return v;
        const int iv = __builtin_bit_cast(int, v);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
        if constexpr (LPR == 64) return (r0 + r1) + (r2 + r3);
        return (threadIdx.x & 32) ? r2 + r3 : r0 + r1;
    } else {
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
}

// Patch-merging view (reference models/common/swin_model.py:279-287, PatchMerging.forward): row r of the [B * H/2 * W/2, 4 Cs] map the
// LayerNorm sees is the concatenation of the four pixels (2 hh + dy, 2 ww + dx), (dy, dx) = (0,0), (1,0), (0,1), (1,1), of the
// [B, H, W, Cs] token map.  With Cs > 0 the kernels address x (forward, backward) and dx (backward) THROUGH that view: the
// permute + reshape copy of the map (and the scatter of its gradient) never exists.  H, W even; Cs a multiple of 8.
struct MergeView {
    int H = 0, W = 0, Cs = 0;
};

// element offset of channel chunk `col` (a multiple of 8, < 4 Cs) of merged row r
__device__ __forceinline__ size_t merged_offset(const MergeView& mv, int r, int col) {
    const int Wh = mv.W >> 1, per_img = (mv.H >> 1) * Wh;
    const int b = r / per_img, rem = r - b * per_img, hh = rem / Wh, ww = rem - hh * Wh;
    const int q = col / mv.Cs, cc = col - q * mv.Cs;
    return ((size_t)(b * mv.H + 2 * hh + (q & 1)) * mv.W + 2 * ww + (q >> 1)) * mv.Cs + cc;
}

// LPR lanes per row, CH chunks of 8 channels per lane: C = LPR * 8 * CH
template <typename T, typename WT, int LPR, int CH>
__global__ __launch_bounds__(256)
void ln_fwd(const T* __restrict__ x, const WT* __restrict__ w, const WT* __restrict__ b, int rows, float eps,
            T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
            const T* __restrict__ branch, const float* __restrict__ row_scale, int rows_per_sample, T* __restrict__ sum_out,
            float drop_p, const unsigned long long* __restrict__ seed_dev, bool nt, MergeView mv) {
    constexpr int C = LPR * 8 * CH, R = 64 / LPR;
    const unsigned long long seed = (branch && drop_p > 0.f) ? *seed_dev : 0ull;
    const float inv_keep = 1.0f / (1.0f - drop_p);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR;
    const int row = (blockIdx.x * 4 + wave) * R + lane / LPR;
    const int rc = min(row, rows - 1);
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        Vec8<T>::load(x + (mv.Cs ? merged_offset(mv, rc, (c * LPR + sub) * 8) : (size_t)rc * C + (c * LPR + sub) * 8), v[c]);
        if (branch) {  // x <- shortcut + scale * branch, rounded to T exactly as the unfused add / addcmul would store it
            float br[8];
            Vec8<T>::load(branch + (size_t)rc * C + (c * LPR + sub) * 8, br);
            const float sc = row_scale ? row_scale[rc / rows_per_sample] : 1.0f;
            if (drop_p > 0.f) {  // element dropout of the branch (nn.Dropout between the projection and the residual)
                const unsigned long long e0 = (unsigned long long)rc * C + (c * LPR + sub) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) br[i] = round_to<T>(br[i] * keep_scale(seed, e0 + i, drop_p, inv_keep));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) v[c][i] = round_to<T>(__fadd_rn(v[c][i], __fmul_rn(br[i], sc)));
            if (row < rows) Vec8<T>::store(sum_out + (size_t)row * C + (c * LPR + sub) * 8, v[c], nt);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[c][i];
    }
    const float mu = row_sum<LPR>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mu; q = fmaf(d, d, q); }
    const float rs = rsqrtf(row_sum<LPR>(q) * (1.0f / C) + eps);
    if (row < rows) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            float wv[8], bv[8], o[8];
            Vec8<WT>::load(w + (c * LPR + sub) * 8, wv);
            Vec8<WT>::load(b + (c * LPR + sub) * 8, bv);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf((v[c][i] - mu) * rs, wv[i], bv[i]);
            Vec8<T>::store(y + (size_t)row * C + (c * LPR + sub) * 8, o, nt);
        }
        if (sub == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <typename T, typename WT, int LPR, int CH>
__global__ __launch_bounds__(256)
void ln_bwd(const T* __restrict__ x, const WT* __restrict__ w, const T* __restrict__ dy, const float* __restrict__ mean,
            const float* __restrict__ rstd, int rows, T* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
            const T* __restrict__ dres, const float* __restrict__ row_scale, int rows_per_sample, T* __restrict__ dbranch,
            float* __restrict__ dsum, float drop_p, const unsigned long long* __restrict__ seed_dev, bool nt, MergeView mv) {
    constexpr int C = LPR * 8 * CH, R = 64 / LPR;
    const unsigned long long seed = (dbranch && drop_p > 0.f) ? *seed_dev : 0ull;
    const float inv_keep = 1.0f / (1.0f - drop_p);
    // column sums of the branch gradient (= bias gradient of the Linear that produced the branch) ride along for the
    // block widths (C <= 1024); the wide merging norms never fuse a residual and keep their registers
    constexpr bool kBranchSum = CH <= 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR;
    float wv[CH][8], aw[CH][8], ab[CH][8], ad[kBranchSum ? CH : 1][8];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        Vec8<WT>::load(w + (c * LPR + sub) * 8, wv[c]);
#pragma unroll
        for (int i = 0; i < 8; ++i) aw[c][i] = ab[c][i] = 0.f;
        if constexpr (kBranchSum) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ad[c][i] = 0.f;
        }
    }
    const int rows_per_pass = gridDim.x * 4 * R;
    for (int base = (blockIdx.x * 4 + wave) * R; base < rows; base += rows_per_pass) {  // wave-uniform trip count
        const int row = base + lane / LPR;
        const bool live = row < rows;
        const int rc = live ? row : rows - 1;
        const float mu = mean[rc], rs = rstd[rc];
        float xh[CH][8], g[CH][8], skip[CH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            Vec8<T>::load(x + (mv.Cs ? merged_offset(mv, rc, (c * LPR + sub) * 8) : (size_t)rc * C + (c * LPR + sub) * 8), xh[c]);
            Vec8<T>::load(dy + (size_t)rc * C + (c * LPR + sub) * 8, g[c]);
            if (dres) {
                Vec8<T>::load(dres + (size_t)rc * C + (c * LPR + sub) * 8, skip[c]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) skip[c][i] = 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xh[c][i] = (xh[c][i] - mu) * rs;
                const float gw = g[c][i] * wv[c][i];
                s1 += gw;
                s2 = fmaf(gw, xh[c][i], s2);
            }
        }
        s1 = row_sum<LPR>(s1) * (1.0f / C);
        s2 = row_sum<LPR>(s2) * (1.0f / C);
        if (live) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    o[i] = rs * (g[c][i] * wv[c][i] - s1 - xh[c][i] * s2) + skip[c][i];
                    aw[c][i] = fmaf(g[c][i], xh[c][i], aw[c][i]);
                    ab[c][i] += g[c][i];
                }
                Vec8<T>::store(dx + (mv.Cs ? merged_offset(mv, row, (c * LPR + sub) * 8) : (size_t)row * C + (c * LPR + sub) * 8), o, nt);
                if (dbranch) {  // gradient of the branch: the rounded dx times the sample's drop-path factor and/or the
                                // element's dropout keep factor
                    const float sc = row_scale ? row_scale[row / rows_per_sample] : 1.0f;
                    const unsigned long long e0 = (unsigned long long)row * C + (c * LPR + sub) * 8;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        o[i] = round_to<T>(o[i]) * sc;
                        if (drop_p > 0.f) o[i] *= keep_scale(seed, e0 + i, drop_p, inv_keep);
                    }
                    Vec8<T>::store(dbranch + (size_t)row * C + (c * LPR + sub) * 8, o, nt);
                }
                if constexpr (kBranchSum) {
                    if (dsum) {  // what the consumer of the branch gradient reads: the stored (rounded) values
#pragma unroll
                        for (int i = 0; i < 8; ++i) ad[c][i] += round_to<T>(o[i]);
                    }
                }
            }
        }
    }
    // rows handled side by side in one wave hold the same channels: fold them, fold the 4 waves through LDS, and
    // write this workgroup's partial sums (no atomics: thousands of waves on <= 4096 addresses serialise badly)
    __shared__ float part[4][3][512];  // per wave, (dw | db | branch sums), one 512-channel chunk at a time
    const int nsum = (kBranchSum && dsum) ? 3 : 2;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = aw[c][i], bsum = ab[c][i], dsm = kBranchSum ? ad[kBranchSum ? c : 0][i] : 0.f;
#pragma unroll
            for (int o = LPR; o < 64; o <<= 1) {
                a += __shfl_xor(a, o, 64); bsum += __shfl_xor(bsum, o, 64);
                if (kBranchSum) dsm += __shfl_xor(dsm, o, 64);
            }
            if (lane < LPR) { part[wave][0][sub * 8 + i] = a; part[wave][1][sub * 8 + i] = bsum; part[wave][2][sub * 8 + i] = dsm; }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < nsum * LPR * 8; j += 256) {
            const int which = j / (LPR * 8), ch = j % (LPR * 8);
            const float v = part[0][which][ch] + part[1][which][ch] + part[2][which][ch] + part[3][which][ch];
            (which == 0 ? dw : which == 1 ? db : dsum)[(size_t)blockIdx.x * C + c * LPR * 8 + ch] = v;
        }
        __syncthreads();
    }
}

constexpr int kBwdBlocks = GRIT_LN_BWD_PARTIALS;  // persistent backward grid = rows of the partial-sum workspace (GRIT_LN_BWD_PARTIALS)

struct Fused {  // optional residual operands (all null / 0 for the plain LayerNorm)
    const void* branch = nullptr;     // forward: drop-path branch added to x
    const float* row_scale = nullptr; // forward: per-sample drop-path keep / scale factors, or null
    int rows_per_sample = 1;
    void* sum_out = nullptr;          // forward: where x + scale * branch is stored
    const void* dres = nullptr;       // backward: gradient arriving at x through the skip path
    void* dbranch = nullptr;          // backward: where row_scale * dx (the branch gradient) is stored, or null
    float* dsum = nullptr;            // backward: per-workgroup partial column sums of the branch gradient, or null
    float drop_p = 0.f;               // element dropout probability applied to the branch (0: none)
    const unsigned long long* seed_dev = nullptr;  // device-resident dropout seed (read when drop_p > 0)
    MergeView merge;                  // patch-merging view of x (and of dx): grit_merge_layernorm_*
};

template <typename T, typename WT>
int launch(bool fwd, const void* x, const void* w, const void* b_or_dy, const float* mean_in, const float* rstd_in, int rows,
           int C, float eps, void* out, float* o1, float* o2, hipStream_t st, const Fused& fu) {
#define GRIT_LN_CASE(LPR_, CH_)                                                                                        \
    {                                                                                                                  \
        constexpr int R = 64 / LPR_;                                                                                   \
        const int blocks = (rows + 4 * R - 1) / (4 * R);                                                               \
        const bool nt = (size_t)rows * C * sizeof(T) >= ((size_t)16 << 20);                                            \
        if (fwd)                                                                                                       \
            hipLaunchKernelGGL((ln_fwd<T, WT, LPR_, CH_>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const WT*)w,  \
                               (const WT*)b_or_dy, rows, eps, (T*)out, o1, o2, (const T*)fu.branch, fu.row_scale,        \
                               fu.rows_per_sample, (T*)fu.sum_out, fu.drop_p, fu.seed_dev, nt, fu.merge);             \
        else                                                                                                           \
            hipLaunchKernelGGL((ln_bwd<T, WT, LPR_, CH_>), dim3(blocks < kBwdBlocks ? blocks : kBwdBlocks), dim3(256), 0, st, \
                               (const T*)x, (const WT*)w, (const T*)b_or_dy, mean_in, rstd_in, rows, (T*)out, o1, o2,  \
                               (const T*)fu.dres, fu.row_scale, fu.rows_per_sample, (T*)fu.dbranch, fu.dsum, fu.drop_p, fu.seed_dev, nt, fu.merge); \
        return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;                                            \
    }
    switch (C) {
        case 128: GRIT_LN_CASE(16, 1)
        case 256: GRIT_LN_CASE(32, 1)
        case 512: GRIT_LN_CASE(64, 1)
        case 1024: GRIT_LN_CASE(64, 2)
        case 2048: GRIT_LN_CASE(64, 4)
        case 4096: GRIT_LN_CASE(64, 8)
        default: return GRIT_ERR_UNSUPPORTED;
    }
#undef GRIT_LN_CASE
}

int dispatch(bool fwd, const void* x, const void* w, const void* b_or_dy, const float* mean_in, const float* rstd_in,
             int rows, int C, float eps, int x_bf16, int w_bf16, void* out, float* o1, float* o2, hipStream_t st,
             const Fused& fu = Fused()) {
    if (!x || !w || !b_or_dy || !out || !o1 || !o2 || rows <= 0 || C <= 0) return GRIT_ERR_BAD_ARG;
    if (x_bf16 && w_bf16) return launch<__hip_bfloat16, __hip_bfloat16>(fwd, x, w, b_or_dy, mean_in, rstd_in, rows, C, eps, out, o1, o2, st, fu);
    if (x_bf16) return launch<__hip_bfloat16, float>(fwd, x, w, b_or_dy, mean_in, rstd_in, rows, C, eps, out, o1, o2, st, fu);
    if (!w_bf16) return launch<float, float>(fwd, x, w, b_or_dy, mean_in, rstd_in, rows, C, eps, out, o1, o2, st, fu);
    return GRIT_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" {

int grit_layernorm_fwd(const void* x, const void* weight, const void* bias, int rows, int C, float eps, int x_is_bf16,
                       int w_is_bf16, void* y, float* mean, float* rstd, void* stream) {
    return dispatch(true, x, weight, bias, nullptr, nullptr, rows, C, eps, x_is_bf16, w_is_bf16, y, mean, rstd, (hipStream_t)stream);
}

int grit_layernorm_bwd(const void* x, const void* weight, const void* dy, const float* mean, const float* rstd, int rows,
                       int C, int x_is_bf16, int w_is_bf16, void* dx, float* dweight, float* dbias, void* stream) {
    if (!mean || !rstd) return GRIT_ERR_BAD_ARG;
    return dispatch(false, x, weight, dy, mean, rstd, rows, C, 0.f, x_is_bf16, w_is_bf16, dx, dweight, dbias, (hipStream_t)stream);
}

static bool merge_ok(int B, int H, int W, int Cs, int rows, int C) {
    return B > 0 && H > 0 && W > 0 && Cs > 0 && H % 2 == 0 && W % 2 == 0 && Cs % 8 == 0 && C == 4 * Cs &&
           (long long)B * (H / 2) * (W / 2) == rows;
}

int grit_merge_layernorm_fwd(const void* x, int B, int H, int W, int Cs, const void* weight, const void* bias, float eps,
                             int x_is_bf16, int w_is_bf16, void* y, float* mean, float* rstd, void* stream) {
    const long long rows = (long long)B * (H / 2) * (W / 2);
    if (rows <= 0 || rows > 0x7fffffffLL || !merge_ok(B, H, W, Cs, (int)rows, 4 * Cs)) return GRIT_ERR_BAD_ARG;
    Fused fu;
    fu.merge.H = H; fu.merge.W = W; fu.merge.Cs = Cs;
    return dispatch(true, x, weight, bias, nullptr, nullptr, (int)rows, 4 * Cs, eps, x_is_bf16, w_is_bf16, y, mean, rstd,
                    (hipStream_t)stream, fu);
}

int grit_merge_layernorm_bwd(const void* x, int B, int H, int W, int Cs, const void* weight, const void* dy, const float* mean,
                             const float* rstd, int x_is_bf16, int w_is_bf16, void* dx, float* dweight, float* dbias,
                             void* stream) {
    const long long rows = (long long)B * (H / 2) * (W / 2);
    if (!mean || !rstd || rows <= 0 || rows > 0x7fffffffLL || !merge_ok(B, H, W, Cs, (int)rows, 4 * Cs)) return GRIT_ERR_BAD_ARG;
    Fused fu;
    fu.merge.H = H; fu.merge.W = W; fu.merge.Cs = Cs;
    return dispatch(false, x, weight, dy, mean, rstd, (int)rows, 4 * Cs, 0.f, x_is_bf16, w_is_bf16, dx, dweight, dbias,
                    (hipStream_t)stream, fu);
}

int grit_add_layernorm_fwd(const void* shortcut, const void* branch, const float* row_scale, int rows_per_sample,
                           float drop_p, const uint64_t* seed_dev, const void* weight, const void* bias, int rows, int C,
                           float eps, int x_is_bf16, int w_is_bf16, void* sum_out, void* y, float* mean, float* rstd,
                           void* stream) {
    if (!branch || !sum_out || rows_per_sample <= 0 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !seed_dev))
        return GRIT_ERR_BAD_ARG;
    Fused fu;
    fu.branch = branch; fu.row_scale = row_scale; fu.rows_per_sample = rows_per_sample; fu.sum_out = sum_out;
    fu.drop_p = drop_p; fu.seed_dev = (const unsigned long long*)seed_dev;
    return dispatch(true, shortcut, weight, bias, nullptr, nullptr, rows, C, eps, x_is_bf16, w_is_bf16, y, mean, rstd,
                    (hipStream_t)stream, fu);
}

int grit_add_layernorm_bwd(const void* x, const void* weight, const void* dy, const void* dres, const float* mean,
                           const float* rstd, const float* row_scale, int rows_per_sample, float drop_p,
                           const uint64_t* seed_dev, int rows, int C, int x_is_bf16, int w_is_bf16, void* dx, void* dbranch,
                           float* dweight, float* dbias, float* dbranch_colsum, void* stream) {
    if (!mean || !rstd || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !seed_dev) || (row_scale && rows_per_sample <= 0))
        return GRIT_ERR_BAD_ARG;
    if (((row_scale != nullptr) || drop_p > 0.f) != (dbranch != nullptr)) return GRIT_ERR_BAD_ARG;
    if (dbranch_colsum && C > 1024) return GRIT_ERR_UNSUPPORTED;
    Fused fu;
    fu.dres = dres; fu.row_scale = row_scale; fu.rows_per_sample = row_scale ? rows_per_sample : 1; fu.dbranch = dbranch;
    fu.dsum = dbranch_colsum; fu.drop_p = drop_p; fu.seed_dev = (const unsigned long long*)seed_dev;
    return dispatch(false, x, weight, dy, mean, rstd, rows, C, 0.f, x_is_bf16, w_is_bf16, dx, dweight, dbias,
                    (hipStream_t)stream, fu);
}

}  // extern "C"
