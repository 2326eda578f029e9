// Fused scaled-dot attention core in fp32 arithmetic for gfx950 (MI355X), head_dim 64.
//
// One launch replaces the chain of davidnvq/grit models/common/attention.py:71-84
//     scores = q @ k^T / sqrt(d_k);  scores.masked_fill(mask, -inf);  softmax;  dropout;  @ v
// (and the same chain inside nn.MultiheadAttention, models/detection/det_module.py:330-333).  The shapes on
// GRIT's path are tiny (Tq <= 150, Nk <= 150, 8 heads x 64): the work is launch/latency bound, so the design
// aims at exact fp32 results (north_star: decoder within 1e-4 fp32, bit-exact beam tokens) with no
// materialised [B,H,Tq,Nk] tensor, not at MFMA peak:
//
//   forward : workgroup = one (batch, head) x a slab of <= 32 query rows.  K is staged once in LDS with a
//             65-float row pitch (lane = key reads are bank-conflict free), V with a 64-float pitch
//             (lane = channel reads are conflict free).  One wavefront owns a query row: scores with
//             lane = key (q[d] arrives as an SGPR via v_readlane), softmax statistics by wave shuffles,
//             output with lane = channel (p[j] broadcast by v_readlane).  Row log-sum-exp is kept for backward.
//   backward: workgroup = one (batch, head), 8 waves.  P is recomputed from the saved log-sum-exp (no max/sum
//             pass), delta = rowsum(dO*O) by a wave reduction, dQ with lane = channel, dK/dV accumulated in
//             LDS with ds_add_f32 (all rows of a head meet in one workgroup, so no global atomics) and
//             written once.
//   dropout : keep-mask from a counter hash of (seed, flat index of P); forward and backward regenerate the
//             same bits.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/grit_hip.h"
#include "attn_internal.h"

namespace {

constexpr int kD = 64;
constexpr int kMaxNk = 256;
constexpr int kChunks = kMaxNk / 64;
constexpr int kRowsPerBlockFwd = 32;

__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ float ld(const __hip_bfloat16* p) { return __bfloat162float(*p); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// counter-based keep decision: murmur3 finaliser over (index ^ seed-mix); u in [0,1) with 24 bits
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ull;
    unsigned int x = (unsigned int)(z ^ (z >> 32));
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

template <typename T>
__global__ __launch_bounds__(256)
void attn_fwd_kernel(const T* __restrict__ q, long ldq, long bsq, const T* __restrict__ k, long ldk, long bsk,
                     const T* __restrict__ v, long ldv, long bsv, const uint8_t* __restrict__ mask, long msb, long msq,
                     int H, int Tq, int Nk, float scale, float drop_p, unsigned long long seed,
                     const unsigned long long* __restrict__ seed_dev, T* __restrict__ out, float* __restrict__ lse) {
    if (seed_dev) seed ^= *seed_dev;  // device-resident seed: graph replays draw fresh masks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                  // [Nk][65]
    float* Vs = smem + (size_t)Nk * 65;  // [Nk][64]
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int row0 = blockIdx.y * kRowsPerBlockFwd;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;

    const T* kb = k + (size_t)b * bsk + (size_t)h * kD;
    const T* vb = v + (size_t)b * bsv + (size_t)h * kD;
    for (int i = threadIdx.x; i < Nk * kD; i += blockDim.x) {
        const int j = i >> 6, d = i & 63;
        Ks[j * 65 + d] = ld(kb + (size_t)j * ldk + d);
        Vs[j * 64 + d] = ld(vb + (size_t)j * ldv + d);
    }
    __syncthreads();

    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int row_end = min(row0 + kRowsPerBlockFwd, Tq);
    for (int r = row0 + wave; r < row_end; r += nwave) {
        const float qd = ld(q + (size_t)b * bsq + (size_t)r * ldq + (size_t)h * kD + lane);
        float s[kChunks];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) s[c] = 0.f;
        // scores: lane = key (j = lane + 64 c)
#pragma unroll
        for (int d = 0; d < kD; ++d) {
            const float qv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qd), d));
#pragma unroll
            for (int c = 0; c < kChunks; ++c) {
                const int j = lane + 64 * c;
                if (c * 64 < Nk) s[c] = fmaf(qv, Ks[min(j, Nk - 1) * 65 + d], s[c]);
            }
        }
        const uint8_t* mrow = mask ? mask + (size_t)b * msb + (size_t)r * msq : nullptr;
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int j = lane + 64 * c;
            float val = s[c] * scale;
            if (j >= Nk) val = -INFINITY;
            else if (mrow && mrow[j]) val = -INFINITY;
            s[c] = val;
            mx = fmaxf(mx, val);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            // a fully masked row gives exp(-inf - -inf) = NaN, like the reference's softmax over all -inf
            const float e = (lane + 64 * c < Nk) ? expf(s[c] - mx) : 0.f;
            s[c] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        const unsigned long long pbase = ((unsigned long long)bh * Tq + r) * (unsigned long long)Nk;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            float pj = s[c] * inv;
            if (drop_p > 0.f) pj *= keep_scale(seed, pbase + lane + 64 * c, drop_p, inv_keep);
            s[c] = pj;
        }
        // output: lane = channel
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int jn = min(64, Nk - 64 * c);
            for (int jj = 0; jj < jn; ++jj) {
                const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s[c]), jj));
                acc = fmaf(pj, Vs[(64 * c + jj) * 64 + lane], acc);
            }
        }
        st(out + ((size_t)b * Tq + r) * ((size_t)H * kD) + (size_t)h * kD + lane, acc);
        if (lane == 0) lse[(size_t)bh * Tq + r] = mx + logf(sum);
    }
}

template <typename T>
__global__ __launch_bounds__(512)
void attn_bwd_kernel(const T* __restrict__ q, long ldq, long bsq, const T* __restrict__ k, long ldk, long bsk,
                     const T* __restrict__ v, long ldv, long bsv, const uint8_t* __restrict__ mask, long msb, long msq,
                     const T* __restrict__ out, const T* __restrict__ dout, const float* __restrict__ lse,
                     int H, int Tq, int Nk, float scale, float drop_p, unsigned long long seed,
                     const unsigned long long* __restrict__ seed_dev, T* __restrict__ dq, T* __restrict__ dk,
                     T* __restrict__ dv) {
    if (seed_dev) seed ^= *seed_dev;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dKs = smem;                          // [Nk][64]
    float* dVs = smem + (size_t)Nk * 64;        // [Nk][64]
    float* scratch = smem + (size_t)Nk * 128;   // per wave: ds[kMaxNk], pd[kMaxNk]
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    float* ds_w = scratch + (size_t)wave * 2 * kMaxNk;
    float* pd_w = ds_w + kMaxNk;

    for (int i = threadIdx.x; i < Nk * 128; i += blockDim.x) smem[i] = 0.f;
    __syncthreads();

    const T* kb = k + (size_t)b * bsk + (size_t)h * kD;
    const T* vb = v + (size_t)b * bsv + (size_t)h * kD;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;

    for (int r = wave; r < Tq; r += nwave) {
        const size_t orow = ((size_t)b * Tq + r) * ((size_t)H * kD) + (size_t)h * kD + lane;
        const float qd = ld(q + (size_t)b * bsq + (size_t)r * ldq + (size_t)h * kD + lane);
        const float god = ld(dout + orow);
        const float od = ld(out + orow);
        const float delta = wave_sum(god * od);
        const float row_lse = lse[(size_t)bh * Tq + r];

        // lane = key: s = q.K[j], dpd = dO.V[j]; each lane walks its own 256-byte K / V row
        float s[kChunks], dp[kChunks];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            s[c] = 0.f; dp[c] = 0.f;
            if (c * 64 < Nk) {
                const int j = min(lane + 64 * c, Nk - 1);
                const T* kr = kb + (size_t)j * ldk;
                const T* vr = vb + (size_t)j * ldv;
#pragma unroll
                for (int d = 0; d < kD; ++d) {
                    const float qv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qd), d));
                    const float gv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(god), d));
                    s[c] = fmaf(qv, ld(kr + d), s[c]);
                    dp[c] = fmaf(gv, ld(vr + d), dp[c]);
                }
            }
        }
        const uint8_t* mrow = mask ? mask + (size_t)b * msb + (size_t)r * msq : nullptr;
        const unsigned long long pbase = ((unsigned long long)bh * Tq + r) * (unsigned long long)Nk;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int j = lane + 64 * c;
            if (j < Nk) {
                float p = expf(s[c] * scale - row_lse);
                if (mrow && mrow[j]) p = 0.f;
                const float m = drop_p > 0.f ? keep_scale(seed, pbase + j, drop_p, inv_keep) : 1.0f;
                ds_w[j] = p * (m * dp[c] - delta) * scale;
                pd_w[j] = p * m;
            }
        }
        // (same wave wrote and reads: LDS ops of one wave are issued in order, a wave barrier is enough)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        float dqd = 0.f;
        for (int j = 0; j < Nk; ++j) {
            const float dsj = ds_w[j], pdj = pd_w[j];
            dqd = fmaf(dsj, ld(kb + (size_t)j * ldk + lane), dqd);
            atomicAdd(&dKs[j * 64 + lane], dsj * qd);
            atomicAdd(&dVs[j * 64 + lane], pdj * god);
        }
        st(dq + ((size_t)b * Tq + r) * ((size_t)H * kD) + (size_t)h * kD + lane, dqd);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Nk * kD; i += blockDim.x) {
        const int j = i >> 6, d = i & 63;
        const size_t o = ((size_t)b * Nk + j) * ((size_t)H * kD) + (size_t)h * kD + d;
        st(dk + o, dKs[i]);
        st(dv + o, dVs[i]);
    }
}

bool args_ok(int B, int H, int Tq, int Nk, int D) { return B > 0 && H > 0 && Tq > 0 && Nk > 0 && D > 0; }

template <typename T>
int launch_fwd(const T* q, long ldq, long bsq, const T* k, long ldk, long bsk, const T* v, long ldv, long bsv,
               const uint8_t* mask, long msb, long msq, int B, int H, int Tq, int Nk, int D, float scale, float drop_p,
               unsigned long long seed, const unsigned long long* seed_dev, T* out, float* lse, hipStream_t st) {
    if (!q || !k || !v || !out || !lse || !args_ok(B, H, Tq, Nk, D)) return GRIT_ERR_BAD_ARG;
    if (D != kD || Nk > kMaxNk || drop_p < 0.f || drop_p >= 1.f) return GRIT_ERR_UNSUPPORTED;
    const size_t lds = (size_t)Nk * (65 + 64) * sizeof(float);
    auto kern = attn_fwd_kernel<T>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GRIT_ERR_LAUNCH;
    const dim3 grid(B * H, (Tq + kRowsPerBlockFwd - 1) / kRowsPerBlockFwd), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, st, q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, mask, msb, msq, H, Tq, Nk,
                       scale, drop_p, seed, seed_dev, out, lse);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

template <typename T>
int launch_bwd(const T* q, long ldq, long bsq, const T* k, long ldk, long bsk, const T* v, long ldv, long bsv,
               const uint8_t* mask, long msb, long msq, const T* out, const T* dout, const float* lse, int B, int H,
               int Tq, int Nk, int D, float scale, float drop_p, unsigned long long seed,
               const unsigned long long* seed_dev, T* dq, T* dk, T* dv, hipStream_t st) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !args_ok(B, H, Tq, Nk, D)) return GRIT_ERR_BAD_ARG;
    if (D != kD || Nk > kMaxNk || drop_p < 0.f || drop_p >= 1.f) return GRIT_ERR_UNSUPPORTED;
    const int threads = 512;
    const size_t lds = ((size_t)Nk * 128 + (size_t)(threads / 64) * 2 * kMaxNk) * sizeof(float);
    auto kern = attn_bwd_kernel<T>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GRIT_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(B * H), dim3(threads), lds, st, q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, mask, msb, msq,
                       out, dout, lse, H, Tq, Nk, scale, drop_p, seed, seed_dev, dq, dk, dv);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // namespace

extern "C" {

#define GRIT_ATTN_FWD(SUF, T)                                                                                       \
    int grit_attn_fwd_##SUF(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk,      \
                            const void* v, int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb,         \
                            int64_t mask_sq, int B, int H, int Tq, int Nk, int D, float scale, float dropout_p,    \
                            uint64_t seed, const uint64_t* seed_dev, void* out, float* lse, void* stream) {        \
        return launch_fwd<T>((const T*)q, ldq, bsq, (const T*)k, ldk, bsk, (const T*)v, ldv, bsv, mask, mask_sb,   \
                             mask_sq, B, H, Tq, Nk, D, scale, dropout_p, seed,                                      \
                             (const unsigned long long*)seed_dev, (T*)out, lse, (hipStream_t)stream);              \
    }
#define GRIT_ATTN_BWD(SUF, T)                                                                                       \
    int grit_attn_bwd_##SUF(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk,      \
                            const void* v, int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb,         \
                            int64_t mask_sq, const void* out, const void* dout, const float* lse, int B, int H,    \
                            int Tq, int Nk, int D, float scale, float dropout_p, uint64_t seed,                    \
                            const uint64_t* seed_dev, void* dq, void* dk, void* dv, void* stream) {                \
        return launch_bwd<T>((const T*)q, ldq, bsq, (const T*)k, ldk, bsk, (const T*)v, ldv, bsv, mask, mask_sb,   \
                             mask_sq, (const T*)out, (const T*)dout, lse, B, H, Tq, Nk, D, scale, dropout_p, seed, \
                             (const unsigned long long*)seed_dev, (T*)dq, (T*)dk, (T*)dv, (hipStream_t)stream);    \
    }

GRIT_ATTN_FWD(f32, float)
GRIT_ATTN_BWD(f32, float)

// bf16 storage: matrix-core kernels (attn_mfma.hip) when the shape fits, else the fp32-arithmetic kernels above
int grit_attn_fwd_bf16(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk, const void* v,
                       int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq, int B, int H,
                       int Tq, int Nk, int D, float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                       void* out, float* lse, void* stream) {
    if (!q || !k || !v || !out || !lse || !args_ok(B, H, Tq, Nk, D)) return GRIT_ERR_BAD_ARG;
    if (dropout_p < 0.f || dropout_p >= 1.f) return GRIT_ERR_UNSUPPORTED;
    const int st = grit_attn_mfma_fwd(q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, mask, mask_sb, mask_sq, B, H, Tq, Nk, D, scale,
                                      dropout_p, seed, (const unsigned long long*)seed_dev, out, lse, (hipStream_t)stream);
    if (st != GRIT_ERR_UNSUPPORTED) return st;
    using T = __hip_bfloat16;
    return launch_fwd<T>((const T*)q, ldq, bsq, (const T*)k, ldk, bsk, (const T*)v, ldv, bsv, mask, mask_sb, mask_sq, B, H,
                         Tq, Nk, D, scale, dropout_p, seed, (const unsigned long long*)seed_dev, (T*)out, lse,
                         (hipStream_t)stream);
}

int grit_attn_bwd_bf16(const void* q, int64_t ldq, int64_t bsq, const void* k, int64_t ldk, int64_t bsk, const void* v,
                       int64_t ldv, int64_t bsv, const uint8_t* mask, int64_t mask_sb, int64_t mask_sq, const void* out,
                       const void* dout, const float* lse, int B, int H, int Tq, int Nk, int D, float scale,
                       float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* dq, void* dk, void* dv,
                       void* stream) {
    if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv || !args_ok(B, H, Tq, Nk, D)) return GRIT_ERR_BAD_ARG;
    if (dropout_p < 0.f || dropout_p >= 1.f) return GRIT_ERR_UNSUPPORTED;
    const int st = grit_attn_mfma_bwd(q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, mask, mask_sb, mask_sq, out, dout, lse, B, H, Tq,
                                      Nk, D, scale, dropout_p, seed, (const unsigned long long*)seed_dev, dq, dk, dv,
                                      (hipStream_t)stream);
    if (st != GRIT_ERR_UNSUPPORTED) return st;
    using T = __hip_bfloat16;
    return launch_bwd<T>((const T*)q, ldq, bsq, (const T*)k, ldk, bsk, (const T*)v, ldv, bsv, mask, mask_sb, mask_sq,
                         (const T*)out, (const T*)dout, lse, B, H, Tq, Nk, D, scale, dropout_p, seed,
                         (const unsigned long long*)seed_dev, (T*)dq, (T*)dk, (T*)dv, (hipStream_t)stream);
}

}  // extern "C"
