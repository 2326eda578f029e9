// bf16-storage attention core on the matrix cores (MFMA 16x16x32 bf16, fp32 softmax) for gfx950, head_dim 64,
// Tq <= 160, Nk <= 160 -- every attention of GRIT's training step (caption decoder T <= 54, grid 100, regions 150,
// 150-query self-attention of the deformable decoder layers).  Same contract as the fp32 kernels in attn.hip
// (davidnvq/grit models/common/attention.py:71-84); grit_attn_{fwd,bwd}_bf16 dispatch here when the shape fits.
//
// Structure (shared with winattn.hip): workgroup = one (batch, head); forward wave w = query tile w, S^T = K Q^T with
// the key on the MFMA row so the softmax reductions are in-register + 2 shuffles and the bf16-packed accumulator is
// directly the B operand of O^T = V^T P^T (V^T fragments via ds_read_b64_tr_b16 from the row-major V tile).
// Backward: phase 1 wave w = key tile w (P from the saved log2-sum-exp2, dP, dS, dV^T, dK^T), dS crosses LDS once
// (transposed), phase 2 wave w = query tile w (dQ^T).  One workgroup owns all rows of its (batch, head): no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grit_hip.h"
#include "attn_internal.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

constexpr int kD = 64, kP = 72, kRows = 160, kT = 10;  // tile pitch 144 B; up to 10 tiles of 16 rows
constexpr int kSP = 168;                                // pitch of the transposed dS tile
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ v8bf ld8(const void* p) { return __builtin_bit_cast(v8bf, *reinterpret_cast<const uint4*>(p)); }

__device__ __forceinline__ v8bf tr_pair(const __bf16* lo, const __bf16* hi) {
    const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)lo);
    const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)hi);
    const v8s r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(v8bf, r);
}

__device__ __forceinline__ v8bf pack8(const v4f& a, const v4f& b) {
    v8bf r;
    r[0] = (__bf16)a[0]; r[1] = (__bf16)a[1]; r[2] = (__bf16)a[2]; r[3] = (__bf16)a[3];
    r[4] = (__bf16)b[0]; r[5] = (__bf16)b[1]; r[6] = (__bf16)b[2]; r[7] = (__bf16)b[3];
    return r;
}

__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ull;  // same hash as attn.hip
    unsigned int x = (unsigned int)(z ^ (z >> 32));
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

// stage rows [0, n) of a [n, 64] bf16 matrix (row stride ld) into an LDS tile of pitch kP, zero rows [n, rows)
__device__ __forceinline__ void stage_tile(__bf16* tile, const __bf16* src, long ld, int n, int rows = kRows) {
    for (int i = threadIdx.x; i < rows * 8; i += blockDim.x) {
        const int row = i >> 3, c = i & 7;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (row < n) val = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c * 8);
        *reinterpret_cast<uint4*>(&tile[row * kP + c * 8]) = val;
    }
}

__global__ __launch_bounds__(640)
void attn_mfma_fwd(const __bf16* __restrict__ q, long ldq, long bsq, const __bf16* __restrict__ k, long ldk, long bsk,
                   const __bf16* __restrict__ v, long ldv, long bsv, const uint8_t* __restrict__ mask, long msb, long msq,
                   int H, int Tq, int Nk, float scale, float drop_p, unsigned long long seed,
                   const unsigned long long* __restrict__ seed_dev, __bf16* __restrict__ out, float* __restrict__ lse2) {
    if (seed_dev) seed ^= *seed_dev;  // device-resident seed: graph replays draw fresh masks
    __shared__ __attribute__((aligned(16))) __bf16 Ks[kRows * kP];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[kRows * kP];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lg = lane >> 4;
    const int trq = l15 >> 2, trp = l15 & 3;
    const int nkt = (Nk + 15) >> 4;

    // only the rows the products below touch: key tiles of 16 for S, k-steps of 32 keys for P V
    const int staged = min(kRows, (Nk + 31) & ~31);
    stage_tile(Ks, k + (size_t)b * bsk + (size_t)h * kD, ldk, Nk, staged);
    stage_tile(Vs, v + (size_t)b * bsv + (size_t)h * kD, ldv, Nk, staged);
    const int row = 16 * w + l15;           // this lane's query
    const int rowc = min(row, Tq - 1);
    const __bf16* qrow = q + (size_t)b * bsq + (size_t)rowc * ldq + (size_t)h * kD + lg * 8;
    const v8bf qf0 = ld8(qrow), qf1 = ld8(qrow + 32);
    __syncthreads();
    if (16 * w >= Tq) return;  // a wave launched only to help staging (few queries, step-wise decoding): no barrier follows

    const float c2 = scale * kLog2e;
    const uint8_t* mrow = mask ? mask + (size_t)b * msb + (size_t)rowc * msq : nullptr;
    v4f acc[kT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < kT; ++kt) {
        acc[kt] = v4f{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (kt < nkt) {
            const __bf16* kr = &Ks[(16 * kt + l15) * kP + lg * 8];
            v4f s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(kr), qf0, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(kr + 32), qf1, s, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * lg + r;
                float t = s[r] * c2;
                if (key >= Nk || (mrow && mrow[key])) t = -INFINITY;
                acc[kt][r] = t;
                m = fmaxf(m, t);
            }
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const unsigned long long pbase = ((unsigned long long)bh * Tq + rowc) * (unsigned long long)Nk;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < kT; ++kt) {
        if (kt < nkt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = __builtin_amdgcn_exp2f(acc[kt][r] - m);  // all -inf row -> NaN, as torch.softmax
                sum += e;
                if (drop_p > 0.f) e *= keep_scale(seed, pbase + 16 * kt + 4 * lg + r, drop_p, inv_keep);
                acc[kt][r] = e;
            }
        } else {
            acc[kt] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    v4f o[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        if (32 * s < Nk) {
            const v8bf pf = pack8(acc[2 * s], acc[2 * s + 1]);
            const __bf16* lo = &Vs[(32 * s + 4 * lg + trq) * kP + 4 * trp];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(lo + 16 * dt, lo + 16 * kP + 16 * dt), pf, o[dt], 0, 0, 0);
        }
    }
    if (row < Tq) {
        __bf16* orow = out + ((size_t)b * Tq + row) * ((size_t)H * kD) + (size_t)h * kD + 4 * lg;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            v4bf a;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = (__bf16)(o[dt][r] * inv);
            *reinterpret_cast<v4bf*>(orow + 16 * dt) = a;
        }
        if (lg == 0) lse2[(size_t)bh * Tq + row] = (m + __builtin_amdgcn_logf(sum)) * kLn2;  // natural-log units
    }
}

__global__ __launch_bounds__(640)
void attn_mfma_bwd(const __bf16* __restrict__ q, long ldq, long bsq, const __bf16* __restrict__ k, long ldk, long bsk,
                   const __bf16* __restrict__ v, long ldv, long bsv, const uint8_t* __restrict__ mask, long msb, long msq,
                   const __bf16* __restrict__ out, const __bf16* __restrict__ dout, const float* __restrict__ lse2,
                   int H, int Tq, int Nk, float scale, float drop_p, unsigned long long seed,
                   const unsigned long long* __restrict__ seed_dev, __bf16* __restrict__ dq, __bf16* __restrict__ dk,
                   __bf16* __restrict__ dv) {
    if (seed_dev) seed ^= *seed_dev;
    __shared__ __attribute__((aligned(16))) __bf16 Qs[kRows * kP];
    __shared__ __attribute__((aligned(16))) __bf16 dOs[kRows * kP];
    __shared__ __attribute__((aligned(16))) __bf16 Ks[kRows * kP];
    __shared__ __attribute__((aligned(16))) __bf16 dSt[kRows * kSP];  // [key][query]
    __shared__ __attribute__((aligned(16))) float lse_s[kRows];
    __shared__ __attribute__((aligned(16))) float delta_s[kRows];
    const int bh = blockIdx.x, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int trq = l15 >> 2, trp = l15 & 3;
    const int nkt = (Nk + 15) >> 4, nqt = (Tq + 15) >> 4;
    const size_t orow_stride = (size_t)H * kD;

    stage_tile(Qs, q + (size_t)b * bsq + (size_t)h * kD, ldq, Tq);
    stage_tile(Ks, k + (size_t)b * bsk + (size_t)h * kD, ldk, Nk);
    // dO tile + delta = rowsum(dO * O): 8 threads per row, 8 channels each
    for (int i = tid; i < kRows * 8; i += blockDim.x) {
        const int row = i >> 3, c = i & 7;
        uint4 g4 = make_uint4(0, 0, 0, 0);
        float part = 0.f;
        if (row < Tq) {
            const size_t off = ((size_t)b * Tq + row) * orow_stride + (size_t)h * kD + c * 8;
            g4 = *reinterpret_cast<const uint4*>(dout + off);
            const v8bf a = __builtin_bit_cast(v8bf, g4), o8 = ld8(out + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) part = fmaf((float)a[e], (float)o8[e], part);
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        *reinterpret_cast<uint4*>(&dOs[row * kP + c * 8]) = g4;
        if (c == 0) delta_s[row] = part;
    }
    for (int i = tid; i < kRows; i += blockDim.x) lse_s[i] = i < Tq ? lse2[(size_t)bh * Tq + i] * kLog2e : 0.f;
    for (int i = tid; i < (kRows - 16 * nkt) * kSP; i += blockDim.x) dSt[16 * nkt * kSP + i] = (__bf16)0.f;
    __syncthreads();

    const float c2 = scale * kLog2e;
    const float inv_keep = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const v4bf z4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};

    // ================= phase 1: wave w = key tile w =================
    if (w < nkt) {
        const int key = 16 * w + l15, keyc = min(key, Nk - 1);
        const __bf16* krow = k + (size_t)b * bsk + (size_t)keyc * ldk + (size_t)h * kD + lg * 8;
        const __bf16* vrow = v + (size_t)b * bsv + (size_t)keyc * ldv + (size_t)h * kD + lg * 8;
        const v8bf kf0 = ld8(krow), kf1 = ld8(krow + 32), vf0 = ld8(vrow), vf1 = ld8(vrow + 32);
        v4bf Pp[kT];
#pragma unroll
        for (int qt = 0; qt < kT; ++qt) {
            Pp[qt] = z4;
            v4bf sp = z4;
            if (qt < nqt) {
                const __bf16* qr = &Qs[(16 * qt + l15) * kP + lg * 8];
                const __bf16* dr = &dOs[(16 * qt + l15) * kP + lg * 8];
                v4f s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(qr), kf0, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(qr + 32), kf1, s, 0, 0, 0);
                v4f dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(dr), vf0, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(dr + 32), vf1, dp, 0, 0, 0);
                const float4 lq = *reinterpret_cast<const float4*>(&lse_s[16 * qt + 4 * lg]);
                const float4 dl = *reinterpret_cast<const float4*>(&delta_s[16 * qt + 4 * lg]);
                const float lqa[4] = {lq.x, lq.y, lq.z, lq.w}, dla[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = 16 * qt + 4 * lg + r;
                    const bool dead = key >= Nk || qi >= Tq || (mask && mask[(size_t)b * msb + (size_t)min(qi, Tq - 1) * msq + keyc]);
                    const float p = dead ? 0.f : __builtin_amdgcn_exp2f(s[r] * c2 - lqa[r]);
                    const float mk = drop_p > 0.f
                        ? keep_scale(seed, ((unsigned long long)bh * Tq + min(qi, Tq - 1)) * (unsigned long long)Nk + keyc, drop_p, inv_keep)
                        : 1.0f;
                    Pp[qt][r] = (__bf16)(p * mk);
                    sp[r] = (__bf16)(p * (mk * dp[r] - dla[r]));
                }
            }
            *reinterpret_cast<v4bf*>(&dSt[(16 * w + l15) * kSP + 16 * qt + 4 * lg]) = sp;
            __builtin_amdgcn_sched_barrier(0);
        }
        v4f dvv[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        v4f dkk[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            if (32 * s5 < Tq) {
                const v4bf pa = Pp[2 * s5], pb = Pp[2 * s5 + 1];
                const v4bf sa = *reinterpret_cast<const v4bf*>(&dSt[(16 * w + l15) * kSP + 32 * s5 + 4 * lg]);
                const v4bf sb = *reinterpret_cast<const v4bf*>(&dSt[(16 * w + l15) * kSP + 32 * s5 + 16 + 4 * lg]);
                const v8bf pf = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]};
                const v8bf sf = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
                const int rr = 32 * s5 + 4 * lg + trq;
                const __bf16* dlo = &dOs[rr * kP + 4 * trp];
                const __bf16* qlo = &Qs[rr * kP + 4 * trp];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dvv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(dlo + 16 * dt, dlo + 16 * kP + 16 * dt), pf, dvv[dt], 0, 0, 0);
                    dkk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(qlo + 16 * dt, qlo + 16 * kP + 16 * dt), sf, dkk[dt], 0, 0, 0);
                }
            }
        }
        if (key < Nk) {
            const size_t off = ((size_t)b * Nk + key) * orow_stride + (size_t)h * kD + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                v4bf a, c;
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = (__bf16)(dkk[dt][r] * scale); c[r] = (__bf16)dvv[dt][r]; }
                *reinterpret_cast<v4bf*>(dk + off + 16 * dt) = a;
                *reinterpret_cast<v4bf*>(dv + off + 16 * dt) = c;
            }
        }
    }
    __syncthreads();

    // ================= phase 2: wave w = query tile w : dQ^T[d][q] = scale * sum_k K^T[d][k] dS^T[k][q]
    if (w < nqt) {
        v4f dqq[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            if (32 * s5 < Nk) {
                const int rr = 32 * s5 + 8 * lg + trq;
                const __bf16* slo = &dSt[rr * kSP + 16 * w + 4 * trp];
                const v8bf sf = tr_pair(slo, slo + 4 * kSP);
                const __bf16* klo = &Ks[rr * kP + 4 * trp];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    dqq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(klo + 16 * dt, klo + 4 * kP + 16 * dt), sf, dqq[dt], 0, 0, 0);
            }
        }
        const int row = 16 * w + l15;
        if (row < Tq) {
            const size_t off = ((size_t)b * Tq + row) * orow_stride + (size_t)h * kD + 4 * lg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                v4bf a;
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = (__bf16)(dqq[dt][r] * scale);
                *reinterpret_cast<v4bf*>(dq + off + 16 * dt) = a;
            }
        }
    }
}

bool fits(const void* q, long ldq, long bsq, const void* k, long ldk, long bsk, const void* v, long ldv, long bsv,
          int Tq, int Nk, int D) {
    auto ok = [](const void* p, long ld, long bs) { return ((uintptr_t)p % 16 == 0) && (ld % 8 == 0) && (bs % 8 == 0); };
    return D == kD && Tq <= kRows && Nk <= kRows && ok(q, ldq, bsq) && ok(k, ldk, bsk) && ok(v, ldv, bsv);
}

}  // namespace

int grit_attn_mfma_fwd(const void* q, long ldq, long bsq, const void* k, long ldk, long bsk, const void* v, long ldv,
                       long bsv, const uint8_t* mask, long msb, long msq, int B, int H, int Tq, int Nk, int D, float scale,
                       float drop_p, unsigned long long seed, const unsigned long long* seed_dev, void* out, float* lse,
                       hipStream_t st) {
    if (!fits(q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, Tq, Nk, D) || (uintptr_t)out % 16) return GRIT_ERR_UNSUPPORTED;
    // at least 4 waves: with one or two query tiles (beam search: 1-8 queries per image) the K / V staging, not the math, is
    // the critical path, and a single wave would issue its 2 x 20 row loads one after the other
    const int waves = max(4, (Tq + 15) / 16);
    hipLaunchKernelGGL(attn_mfma_fwd, dim3(B * H), dim3(64 * waves), 0, st, (const __bf16*)q, ldq, bsq, (const __bf16*)k,
                       ldk, bsk, (const __bf16*)v, ldv, bsv, mask, msb, msq, H, Tq, Nk, scale, drop_p, seed, seed_dev, (__bf16*)out, lse);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_attn_mfma_bwd(const void* q, long ldq, long bsq, const void* k, long ldk, long bsk, const void* v, long ldv,
                       long bsv, const uint8_t* mask, long msb, long msq, const void* out, const void* dout,
                       const float* lse, int B, int H, int Tq, int Nk, int D, float scale, float drop_p,
                       unsigned long long seed, const unsigned long long* seed_dev, void* dq, void* dk, void* dv,
                       hipStream_t st) {
    if (!fits(q, ldq, bsq, k, ldk, bsk, v, ldv, bsv, Tq, Nk, D)) return GRIT_ERR_UNSUPPORTED;
    if ((uintptr_t)out % 16 || (uintptr_t)dout % 16 || (uintptr_t)dq % 16 || (uintptr_t)dk % 16 || (uintptr_t)dv % 16)
        return GRIT_ERR_UNSUPPORTED;
    const int waves = (max(Tq, Nk) + 15) / 16;
    hipLaunchKernelGGL(attn_mfma_bwd, dim3(B * H), dim3(64 * waves), 0, st, (const __bf16*)q, ldq, bsq, (const __bf16*)k,
                       ldk, bsk, (const __bf16*)v, ldv, bsv, mask, msb, msq, (const __bf16*)out, (const __bf16*)dout, lse,
                       H, Tq, Nk, scale, drop_p, seed, seed_dev, (__bf16*)dq, (__bf16*)dk, (__bf16*)dv);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
