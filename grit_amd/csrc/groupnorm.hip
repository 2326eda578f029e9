// GroupNorm over token-major maps: x [B, T, C] (T = H*W tokens, C channels innermost), G groups of C/G channels,
// statistics over (T, C/G) per (image, group) -- torch.nn.GroupNorm on the NCHW view of the same data
// (reference models/caption/detector.py:28-33: Conv2d 1x1 + GroupNorm(32, 512) per feature level).
//
// Why not torch's kernel: it wants NCHW, while everything around it here is token-major (the 1x1 convolution runs as a
// GEMM on tokens, the deformable-attention value map is [B, S, C]); the NCHW round trip costs two full permute copies
// per level and direction plus the concatenation of the levels.  These kernels read the GEMM output as it is and write
// the normalised tokens straight into the level's slice of the flat [B, S, C] map (arbitrary batch stride).
//
// Layout: one lane owns 8 consecutive channels (one 16-byte access), a row is C/8 lanes; a group's C/G channels are a
// whole number of lanes (C/G % 8 == 0).  A workgroup (4 waves) walks a chunk of rows of one image.
//   forward : gn_stats (per-chunk partial sum / sum of squares per group)  ->  gn_apply (folds the chunk partials,
//             writes mean / rstd once, normalises)
//   backward: gn_bwd_stats (per-chunk, per-channel sums of dy and dy*x)  ->  gn_bwd_apply (dx)  +  gn_bwd_params (dgamma, dbeta)
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

constexpr int kChunks = GRIT_GN_CHUNKS;

template <typename T> struct V8;
template <> struct V8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct V8<__hip_bfloat16> {
    static __device__ __forceinline__ void load(const __hip_bfloat16* p, float (&v)[8]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(__hip_bfloat16* p, const float (&v)[8]) {
        typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
        v8bf o;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (__bf16)v[i];
        *reinterpret_cast<v8bf*>(p) = o;
    }
};

struct Rows { int r0, r1; };
__device__ __forceinline__ Rows chunk_rows(int T) {
    const int per = (T + kChunks - 1) / kChunks;
    Rows r;
    r.r0 = min(T, (int)blockIdx.x * per);
    r.r1 = min(T, r.r0 + per);
    return r;
}

// sum over the lanes that hold the same channels at different rows (stride LPR) -- result valid in lanes < LPR
template <int LPR>
__device__ __forceinline__ float fold_rows(float v) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- forward statistics: partial[b][chunk][g] = (sum, sum of squares) over this chunk's rows and the group's channels
template <typename T, int LPR>
__global__ __launch_bounds__(256)
void gn_stats(const T* __restrict__ x, long x_bstride, int Tn, int G, float* __restrict__ partial) {
    constexpr int C = LPR * 8, R = 64 / LPR;
    __shared__ float red[4][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR;
    const int b = blockIdx.y;
    const Rows rr = chunk_rows(Tn);
    const T* xb = x + (size_t)b * x_bstride;
    float s = 0.f, q = 0.f;
    for (int row = rr.r0 + wave * R + lane / LPR; row < rr.r1; row += 4 * R) {
        float v[8];
        V8<T>::load(xb + (size_t)row * C + sub * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s += v[i]; q = fmaf(v[i], v[i], q); }
    }
    s = fold_rows<LPR>(s); q = fold_rows<LPR>(q);
    const int lpg = (C / G) / 8;  // lanes per group (power of two)
    for (int o = 1; o < lpg; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane < LPR) { red[wave][0][lane] = s; red[wave][1][lane] = q; }
    __syncthreads();
    if (threadIdx.x < 2 * G) {
        const int which = threadIdx.x / G, g = threadIdx.x % G, l = g * lpg;
        const float v = red[0][which][l] + red[1][which][l] + red[2][which][l] + red[3][which][l];
        partial[(((size_t)b * kChunks + blockIdx.x) * 2 + which) * G + g] = v;
    }
}

// ---- forward apply (also folds the chunk partials; block (0, b) publishes mean / rstd for the backward)
template <typename T, typename WT, int LPR>
__global__ __launch_bounds__(256)
void gn_apply(const T* __restrict__ x, long x_bstride, const WT* __restrict__ w, const WT* __restrict__ bias, int Tn, int G,
              float eps, const float* __restrict__ partial, T* __restrict__ y, long y_bstride, float* __restrict__ mean,
              float* __restrict__ rstd) {
    constexpr int C = LPR * 8, R = 64 / LPR;
    __shared__ float mu_s[64], rs_s[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR;
    const int b = blockIdx.y;
    if (threadIdx.x < G) {
        float s = 0.f, q = 0.f;
        for (int c = 0; c < kChunks; ++c) {
            s += partial[(((size_t)b * kChunks + c) * 2 + 0) * G + threadIdx.x];
            q += partial[(((size_t)b * kChunks + c) * 2 + 1) * G + threadIdx.x];
        }
        const float n = (float)Tn * (float)(C / G);
        const float mu = s / n;
        const float var = fmaxf(q / n - mu * mu, 0.f);
        const float rs = rsqrtf(var + eps);
        mu_s[threadIdx.x] = mu; rs_s[threadIdx.x] = rs;
        if (blockIdx.x == 0) { mean[b * G + threadIdx.x] = mu; rstd[b * G + threadIdx.x] = rs; }
    }
    __syncthreads();
    const int g = (sub * 8) / (C / G);
    const float mu = mu_s[g], rs = rs_s[g];
    float wv[8], bv[8];
    V8<WT>::load(w + sub * 8, wv);
    V8<WT>::load(bias + sub * 8, bv);
#pragma unroll
    for (int i = 0; i < 8; ++i) { wv[i] *= rs; bv[i] = fmaf(-mu, wv[i], bv[i]); }  // y = x * (rs*gamma) + (beta - mu*rs*gamma)
    const Rows rr = chunk_rows(Tn);
    const T* xb = x + (size_t)b * x_bstride;
    T* yb = y + (size_t)b * y_bstride;
    for (int row = rr.r0 + wave * R + lane / LPR; row < rr.r1; row += 4 * R) {
        float v[8];
        V8<T>::load(xb + (size_t)row * C + sub * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], wv[i], bv[i]);
        V8<T>::store(yb + (size_t)row * C + sub * 8, v);
    }
}

// ---- backward statistics: partial[b][chunk][0][c] = sum_rows dy*x, [1][c] = sum_rows dy
template <typename T, int LPR>
__global__ __launch_bounds__(256)
void gn_bwd_stats(const T* __restrict__ x, long x_bstride, const T* __restrict__ dy, long dy_bstride, int Tn,
                  float* __restrict__ partial) {
    constexpr int C = LPR * 8, R = 64 / LPR;
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR;
    const int b = blockIdx.y;
    const Rows rr = chunk_rows(Tn);
    const T* xb = x + (size_t)b * x_bstride;
    const T* gb = dy + (size_t)b * dy_bstride;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int row = rr.r0 + wave * R + lane / LPR; row < rr.r1; row += 4 * R) {
        float v[8], g[8];
        V8<T>::load(xb + (size_t)row * C + sub * 8, v);
        V8<T>::load(gb + (size_t)row * C + sub * 8, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = fmaf(g[i], v[i], a[i]); s[i] += g[i]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float av = fold_rows<LPR>(a[i]), sv = fold_rows<LPR>(s[i]);
        if (lane < LPR) { red[wave][0][sub * 8 + i] = av; red[wave][1][sub * 8 + i] = sv; }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += 256) {
        const int which = j / C, c = j % C;
        partial[(((size_t)b * kChunks + blockIdx.x) * 2 + which) * C + c] =
            red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
}

// ---- backward apply: dx = rs*gamma*dy - rs*(db_g + xhat*ds_g)/n   with ds_g = sum_c gamma_c*rs*(A_c - mu*B_c), db_g = sum_c gamma_c*B_c
template <typename T, typename WT, int LPR>
__global__ __launch_bounds__(256)
void gn_bwd_apply(const T* __restrict__ x, long x_bstride, const T* __restrict__ dy, long dy_bstride,
                  const WT* __restrict__ w, const float* __restrict__ mean, const float* __restrict__ rstd, int Tn, int G,
                  const float* __restrict__ partial, T* __restrict__ dx) {
    constexpr int C = LPR * 8, R = 64 / LPR;
    __shared__ float AB[2][C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR;
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < 2 * C; j += 256) {
        float v = 0.f;
        for (int c = 0; c < kChunks; ++c) v += partial[((size_t)b * kChunks + c) * 2 * C + j];
        AB[j / C][j % C] = v;
    }
    __syncthreads();
    const int cpg = C / G, g = (sub * 8) / cpg, lpg = cpg / 8;
    const float mu = mean[b * G + g], rs = rstd[b * G + g];
    float wv[8];
    V8<WT>::load(w + sub * 8, wv);
    float ds = 0.f, db = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float A = AB[0][sub * 8 + i], Bv = AB[1][sub * 8 + i];
        ds = fmaf(wv[i], A - mu * Bv, ds);
        db = fmaf(wv[i], Bv, db);
    }
    for (int o = 1; o < lpg; o <<= 1) { ds += __shfl_xor(ds, o, 64); db += __shfl_xor(db, o, 64); }
    ds *= rs;  // sum over the group of gamma*dy*xhat
    const float inv_n = 1.0f / ((float)Tn * (float)cpg);
    // dx = c1*dy + c2*x + c3 with c1 = rs*gamma, c2 = -rs*rs*ds/n, c3 = -c2*mu - rs*db/n
    const float c2 = -rs * rs * ds * inv_n, c3 = -c2 * mu - rs * db * inv_n;
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] *= rs;
    const Rows rr = chunk_rows(Tn);
    const T* xb = x + (size_t)b * x_bstride;
    const T* gb = dy + (size_t)b * dy_bstride;
    T* ob = dx + (size_t)b * Tn * C;
    for (int row = rr.r0 + wave * R + lane / LPR; row < rr.r1; row += 4 * R) {
        float v[8], gg[8];
        V8<T>::load(xb + (size_t)row * C + sub * 8, v);
        V8<T>::load(gb + (size_t)row * C + sub * 8, gg);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaf(gg[i], wv[i], fmaf(v[i], c2, c3));
        V8<T>::store(ob + (size_t)row * C + sub * 8, v);
    }
}

// ---- parameter gradients: dgamma_c = sum_b rs_bg*(A_bc - mu_bg*B_bc), dbeta_c = sum_b B_bc
template <typename WT>
__global__ __launch_bounds__(256)
void gn_bwd_params(const float* __restrict__ partial, const float* __restrict__ mean, const float* __restrict__ rstd, int B,
                   int C, int G, WT* __restrict__ dgamma, WT* __restrict__ dbeta) {
    // 64 channels per workgroup, the images spread over the 4 waves (a single thread walking B x chunks partials is a
    // 1 000-deep dependent-latency chain)
    __shared__ float red[4][2][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float dg = 0.f, dbv = 0.f;
    if (c < C) {
        const int g = c / (C / G);
        for (int b = part; b < B; b += 4) {
            float A = 0.f, Bv = 0.f;
#pragma unroll
            for (int k = 0; k < kChunks; ++k) {
                A += partial[(((size_t)b * kChunks + k) * 2 + 0) * C + c];
                Bv += partial[(((size_t)b * kChunks + k) * 2 + 1) * C + c];
            }
            dg = fmaf(rstd[b * G + g], A - mean[b * G + g] * Bv, dg);
            dbv += Bv;
        }
    }
    red[part][0][cl] = dg; red[part][1][cl] = dbv;
    __syncthreads();
    if (part == 0 && c < C) {
        dg = red[0][0][cl] + red[1][0][cl] + red[2][0][cl] + red[3][0][cl];
        dbv = red[0][1][cl] + red[1][1][cl] + red[2][1][cl] + red[3][1][cl];
        if constexpr (sizeof(WT) == 4) { dgamma[c] = dg; dbeta[c] = dbv; }
        else { dgamma[c] = __float2bfloat16(dg); dbeta[c] = __float2bfloat16(dbv); }
    }
}

bool shape_ok(int B, int T, int C, int G) {
    return B > 0 && T > 0 && G > 0 && G <= 64 && (C == 256 || C == 512) && C % G == 0 && (C / G) % 8 == 0 &&
           (((C / G) / 8) & ((C / G) / 8 - 1)) == 0 && B <= 65535;
}

}  // namespace

extern "C" {

int grit_groupnorm_tokens_fwd(const void* x, long x_bstride, const void* weight, const void* bias, int B, int T, int C, int G,
                              float eps, int x_is_bf16, int w_is_bf16, void* y, long y_bstride, float* mean, float* rstd,
                              float* workspace, void* stream) {
    if (!x || !weight || !bias || !y || !mean || !rstd || !workspace) return GRIT_ERR_BAD_ARG;
    if (!shape_ok(B, T, C, G) || x_bstride % 8 || y_bstride % 8 || ((uintptr_t)x % 16) || ((uintptr_t)y % 16))
        return GRIT_ERR_UNSUPPORTED;
    if (!x_is_bf16 && w_is_bf16) return GRIT_ERR_UNSUPPORTED;
    const dim3 grid(kChunks, B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define GRIT_GN_FWD(T_, WT_, LPR_)                                                                                          \
    {                                                                                                                       \
        hipLaunchKernelGGL((gn_stats<T_, LPR_>), grid, block, 0, st, (const T_*)x, x_bstride, T, G, workspace);             \
        hipLaunchKernelGGL((gn_apply<T_, WT_, LPR_>), grid, block, 0, st, (const T_*)x, x_bstride, (const WT_*)weight,      \
                           (const WT_*)bias, T, G, eps, workspace, (T_*)y, y_bstride, mean, rstd);                          \
    }
    if (x_is_bf16 && w_is_bf16) { if (C == 512) GRIT_GN_FWD(__hip_bfloat16, __hip_bfloat16, 64) else GRIT_GN_FWD(__hip_bfloat16, __hip_bfloat16, 32) }
    else if (x_is_bf16) { if (C == 512) GRIT_GN_FWD(__hip_bfloat16, float, 64) else GRIT_GN_FWD(__hip_bfloat16, float, 32) }
    else { if (C == 512) GRIT_GN_FWD(float, float, 64) else GRIT_GN_FWD(float, float, 32) }
#undef GRIT_GN_FWD
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_groupnorm_tokens_bwd(const void* x, long x_bstride, const void* dy, long dy_bstride, const void* weight,
                              const float* mean, const float* rstd, int B, int T, int C, int G, int x_is_bf16, int w_is_bf16,
                              void* dx, void* dweight, void* dbias, float* workspace, void* stream) {
    if (!x || !dy || !weight || !mean || !rstd || !dx || !dweight || !dbias || !workspace) return GRIT_ERR_BAD_ARG;
    if (!shape_ok(B, T, C, G) || x_bstride % 8 || dy_bstride % 8 || ((uintptr_t)x % 16) || ((uintptr_t)dy % 16) ||
        ((uintptr_t)dx % 16))
        return GRIT_ERR_UNSUPPORTED;
    if (!x_is_bf16 && w_is_bf16) return GRIT_ERR_UNSUPPORTED;
    const dim3 grid(kChunks, B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define GRIT_GN_BWD(T_, WT_, LPR_)                                                                                          \
    {                                                                                                                       \
        hipLaunchKernelGGL((gn_bwd_stats<T_, LPR_>), grid, block, 0, st, (const T_*)x, x_bstride, (const T_*)dy, dy_bstride, \
                           T, workspace);                                                                                   \
        hipLaunchKernelGGL((gn_bwd_apply<T_, WT_, LPR_>), grid, block, 0, st, (const T_*)x, x_bstride, (const T_*)dy,       \
                           dy_bstride, (const WT_*)weight, mean, rstd, T, G, workspace, (T_*)dx);                           \
        hipLaunchKernelGGL((gn_bwd_params<WT_>), dim3((C + 63) / 64), block, 0, st, workspace, mean, rstd, B, C, G,      \
                           (WT_*)dweight, (WT_*)dbias);                                                                     \
    }
    if (x_is_bf16 && w_is_bf16) { if (C == 512) GRIT_GN_BWD(__hip_bfloat16, __hip_bfloat16, 64) else GRIT_GN_BWD(__hip_bfloat16, __hip_bfloat16, 32) }
    else if (x_is_bf16) { if (C == 512) GRIT_GN_BWD(__hip_bfloat16, float, 64) else GRIT_GN_BWD(__hip_bfloat16, float, 32) }
    else { if (C == 512) GRIT_GN_BWD(float, float, 64) else GRIT_GN_BWD(float, float, 32) }
#undef GRIT_GN_BWD
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // extern "C"
