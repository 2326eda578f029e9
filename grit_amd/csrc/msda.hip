// Multi-scale deformable attention for gfx950 (MI355X): forward gather + backward scatter.
//
// Semantics follow davidnvq/grit models/ops/src/cuda/ms_deform_im2col_cuda.cuh (forward :237-299 with the
// bilinear rule of :33-84, backward :406-510 with :87-159); the decomposition is CDNA4's own:
//
//   * one 64-lane wavefront owns one output row (b, q, m) -- all L*P sampling points of one head;
//   * forward fast path (D = 64 / 32): a row's D channels are D/4 lanes x float4, so a wave covers
//     64/(D/4) sampling points at once and every corner is one 16-byte load per lane (a whole
//     256-byte pixel-head slice per 16 lanes).  All 4 corners of all points are issued before any
//     is consumed: 16 independent dwordx4 loads in flight per lane hide the L2 / Infinity-Cache / HBM
//     latency of the gather.  Out-of-range corners load a clamped (valid) address and are
//     selected to zero, so there is no divergent branch around a load;
//   * loc / attn_w of the row are read ONCE by the wave (coalesced) and handed to the lanes by
//     ds_bpermute -- the reference re-reads them per channel thread;
//   * the blockIdx -> row map is XCD-aware: workgroups that share an XCD (same blockIdx % 8) walk
//     one contiguous chunk of rows, i.e. one batch element's value map, so its coarse levels stay
//     in that XCD's 4 MiB L2;
//   * backward: lane = channel, so every grad_value update of a corner is ONE wave-wide
//     global_atomic_add_f32 over 256 contiguous bytes (the full-rate shape of the memory-side
//     atomic units); grad_loc / grad_attn_w are wave-shuffle reductions -- no LDS tree, no barrier.
//
// No hipify, no CUDA dual path: this file only builds for gfx950.
#include <hip/hip_runtime.h>
#include "per_device.h"
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grit_hip.h"

namespace {

constexpr int kWave = 64;
constexpr int kRowsPerBlock = 4;  // 4 waves / workgroup, one row each

// Bijective XCD remap (blocks b and b+8 share an XCD): logical id such that one XCD gets a
// contiguous range of logical blocks.  Speed only -- any placement gives the same result.
__device__ __forceinline__ int xcd_logical_block(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// Corner geometry of one sampling point; weights of corners that fall outside are forced to 0 and
// their indices clamped inside the map, so a load through them is always legal.
template <typename T>
struct Corners {
    int o1, o2, o3, o4;  // pixel offsets (h*W + w) of the 4 corners, clamped
    T w1, w2, w3, w4;    // bilinear weights (hh*hw, hh*lw, lh*hw, lh*lw)
    bool k1, k2, k3, k4; // corner inside the map
    T lh, lw, hh, hw;
    bool live;           // point passes the (-1, H) x (-1, W) test
};

template <typename T>
__device__ __forceinline__ Corners<T> make_corners(T x, T y, int H, int W) {
    Corners<T> c;
    const T h_im = y * (T)H - (T)0.5;
    const T w_im = x * (T)W - (T)0.5;
    c.live = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < (T)H) && (w_im < (T)W);
    const T hf = floor(h_im), wf = floor(w_im);
    // clamp before the int conversion so NaN / huge coordinates cannot produce wild indices
    const int h_low = (int)fmin(fmax(hf, (T)-2), (T)H);
    const int w_low = (int)fmin(fmax(wf, (T)-2), (T)W);
    const int h_high = h_low + 1, w_high = w_low + 1;
    c.lh = h_im - hf; c.lw = w_im - wf;
    c.hh = (T)1 - c.lh; c.hw = (T)1 - c.lw;
    const bool hl = h_low >= 0 && h_low <= H - 1, hhv = h_high >= 0 && h_high <= H - 1;
    const bool wl = w_low >= 0 && w_low <= W - 1, whv = w_high >= 0 && w_high <= W - 1;
    c.k1 = c.live && hl && wl;  c.k2 = c.live && hl && whv;
    c.k3 = c.live && hhv && wl; c.k4 = c.live && hhv && whv;
    const int hlc = min(max(h_low, 0), H - 1), hhc = min(max(h_high, 0), H - 1);
    const int wlc = min(max(w_low, 0), W - 1), whc = min(max(w_high, 0), W - 1);
    c.o1 = hlc * W + wlc; c.o2 = hlc * W + whc; c.o3 = hhc * W + wlc; c.o4 = hhc * W + whc;
    c.w1 = c.hh * c.hw; c.w2 = c.hh * c.lw; c.w3 = c.lh * c.hw; c.w4 = c.lh * c.lw;
    return c;
}

// ---------------------------------------------------------------------------------------------------
// Forward, fast path: D = 4*LPP floats, NIT = ceil(L*P / (64/LPP)) point rounds per wave.
// ---------------------------------------------------------------------------------------------------
template <int LPP, int NIT>
__global__ __launch_bounds__(kWave * kRowsPerBlock)
void msda_fwd_vec4(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                   const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                   const float* __restrict__ aw, int S, int M, int L, int Lq, int P,
                   float* __restrict__ out, int nrows, int nblk) {
    constexpr int D = 4 * LPP;
    constexpr int G = kWave / LPP;  // sampling points handled side by side
    const int lane = threadIdx.x & (kWave - 1);
    const int row = xcd_logical_block(blockIdx.x, nblk) * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= nrows) return;  // whole wave leaves together
    const int LP = L * P;
    const int m = row % M;
    const int b = (row / M) / Lq;

    // one coalesced read of the row's 2*LP coordinates and LP weights (LP <= 32 on this path)
    const float locv = lane < 2 * LP ? loc[(size_t)row * 2 * LP + lane] : 0.f;
    const float awv = lane < LP ? aw[(size_t)row * LP + lane] : 0.f;

    const int g = lane / LPP, c4 = lane % LPP;
    const float* vrow = value + (size_t)b * S * M * D + (size_t)m * D + c4 * 4;
    const size_t pix_stride = (size_t)M * D;

    float4 v[NIT][4];
    float cw[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * G + g;
        const bool have = idx < LP;
        const int idc = have ? idx : LP - 1;
        const float x = __shfl(locv, 2 * idc, kWave);
        const float y = __shfl(locv, 2 * idc + 1, kWave);
        const float wt = have ? __shfl(awv, idc, kWave) : 0.f;
        const int l = idc / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const float* vl = vrow + (size_t)lsi[l] * pix_stride;
        const Corners<float> c = make_corners<float>(x, y, H, W);
        v[it][0] = *reinterpret_cast<const float4*>(vl + (size_t)c.o1 * pix_stride);
        v[it][1] = *reinterpret_cast<const float4*>(vl + (size_t)c.o2 * pix_stride);
        v[it][2] = *reinterpret_cast<const float4*>(vl + (size_t)c.o3 * pix_stride);
        v[it][3] = *reinterpret_cast<const float4*>(vl + (size_t)c.o4 * pix_stride);
        cw[it][0] = c.k1 ? c.w1 * wt : 0.f;
        cw[it][1] = c.k2 ? c.w2 * wt : 0.f;
        cw[it][2] = c.k3 ? c.w3 * wt : 0.f;
        cw[it][3] = c.k4 ? c.w4 * wt : 0.f;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // select, not multiply: a non-finite value under a dropped corner must not leak
            const float w = cw[it][k];
            const float4 t = v[it][k];
            acc.x += w != 0.f ? w * t.x : 0.f;
            acc.y += w != 0.f ? w * t.y : 0.f;
            acc.z += w != 0.f ? w * t.z : 0.f;
            acc.w += w != 0.f ? w * t.w : 0.f;
        }
    }
#pragma unroll
    for (int off = LPP; off < kWave; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, kWave);
        acc.y += __shfl_xor(acc.y, off, kWave);
        acc.z += __shfl_xor(acc.z, off, kWave);
        acc.w += __shfl_xor(acc.w, off, kWave);
    }
    if (lane < LPP) *reinterpret_cast<float4*>(out + (size_t)row * D + lane * 4) = acc;
}

// ---------------------------------------------------------------------------------------------------
// Forward, any D / L / P, float or double: lanes stride the channels of the wave's row.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kWave * kRowsPerBlock)
void msda_fwd_generic(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                      const int64_t* __restrict__ lsi, const T* __restrict__ loc,
                      const T* __restrict__ aw, int S, int M, int D, int L, int Lq, int P,
                      T* __restrict__ out, int nrows, int nblk) {
    const int lane = threadIdx.x & (kWave - 1);
    const int row = xcd_logical_block(blockIdx.x, nblk) * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int LP = L * P;
    const int m = row % M;
    const int b = (row / M) / Lq;
    const size_t pix_stride = (size_t)M * D;
    const T* vhead = value + (size_t)b * S * pix_stride + (size_t)m * D;
    const T* lrow = loc + (size_t)row * 2 * LP;
    const T* wrow = aw + (size_t)row * LP;
    for (int c = lane; c < D; c += kWave) {
        T acc = 0;
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            const T* vl = vhead + (size_t)lsi[l] * pix_stride + c;
            for (int p = 0; p < P; ++p) {
                const int idx = l * P + p;
                const Corners<T> k = make_corners<T>(lrow[2 * idx], lrow[2 * idx + 1], H, W);
                if (!k.live) continue;
                const T v1 = k.k1 ? vl[(size_t)k.o1 * pix_stride] : (T)0;
                const T v2 = k.k2 ? vl[(size_t)k.o2 * pix_stride] : (T)0;
                const T v3 = k.k3 ? vl[(size_t)k.o3 * pix_stride] : (T)0;
                const T v4 = k.k4 ? vl[(size_t)k.o4 * pix_stride] : (T)0;
                acc += (k.w1 * v1 + k.w2 * v2 + k.w3 * v3 + k.w4 * v4) * wrow[idx];
            }
        }
        out[(size_t)row * D + c] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------
// Backward, any D / L / P: lane = channel (strided when D > 64).  PU points are loaded together
// (4*PU independent loads per lane) before the atomics of any of them are issued.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add_fast(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_fast(double* p, double v) { unsafeAtomicAdd(p, v); }

template <typename T, int PU>
__global__ __launch_bounds__(kWave * kRowsPerBlock)
void msda_bwd_generic(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                      const int64_t* __restrict__ lsi, const T* __restrict__ loc,
                      const T* __restrict__ aw, const T* __restrict__ grad_out,
                      int S, int M, int D, int L, int Lq, int P,
                      T* __restrict__ grad_value, T* __restrict__ grad_loc, T* __restrict__ grad_aw,
                      int nrows, int nblk) {
    const int lane = threadIdx.x & (kWave - 1);
    const int row = xcd_logical_block(blockIdx.x, nblk) * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int LP = L * P;
    const int m = row % M;
    const int b = (row / M) / Lq;
    const size_t pix_stride = (size_t)M * D;
    const size_t head_off = (size_t)b * S * pix_stride + (size_t)m * D;
    const T* lrow = loc + (size_t)row * 2 * LP;
    const T* wrow = aw + (size_t)row * LP;

    for (int i0 = 0; i0 < LP; i0 += PU) {
        T gx[PU], gy[PU], ga[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) gx[u] = gy[u] = ga[u] = 0;
        for (int c = lane; c < D; c += kWave) {
            const T go = grad_out[(size_t)row * D + c];
            Corners<T> k[PU];
            size_t base[PU];
            T wt[PU], v1[PU], v2[PU], v3[PU], v4[PU];
            int Hs[PU], Ws[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int idx = min(i0 + u, LP - 1);
                const int l = idx / P;
                Hs[u] = (int)shapes[2 * l]; Ws[u] = (int)shapes[2 * l + 1];
                base[u] = head_off + (size_t)lsi[l] * pix_stride + c;
                k[u] = make_corners<T>(lrow[2 * idx], lrow[2 * idx + 1], Hs[u], Ws[u]);
                if (i0 + u >= LP) k[u].live = k[u].k1 = k[u].k2 = k[u].k3 = k[u].k4 = false;
                wt[u] = wrow[idx];
                v1[u] = value[base[u] + (size_t)k[u].o1 * pix_stride];
                v2[u] = value[base[u] + (size_t)k[u].o2 * pix_stride];
                v3[u] = value[base[u] + (size_t)k[u].o3 * pix_stride];
                v4[u] = value[base[u] + (size_t)k[u].o4 * pix_stride];
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const Corners<T>& q = k[u];
                const T tgv = go * wt[u];
                const T a1 = q.k1 ? v1[u] : (T)0, a2 = q.k2 ? v2[u] : (T)0;
                const T a3 = q.k3 ? v3[u] : (T)0, a4 = q.k4 ? v4[u] : (T)0;
                if (q.k1) atomic_add_fast(grad_value + base[u] + (size_t)q.o1 * pix_stride, q.w1 * tgv);
                if (q.k2) atomic_add_fast(grad_value + base[u] + (size_t)q.o2 * pix_stride, q.w2 * tgv);
                if (q.k3) atomic_add_fast(grad_value + base[u] + (size_t)q.o3 * pix_stride, q.w3 * tgv);
                if (q.k4) atomic_add_fast(grad_value + base[u] + (size_t)q.o4 * pix_stride, q.w4 * tgv);
                // d/dh and d/dw of the bilinear form (dropped corners contribute nothing)
                const T gh = -q.hw * a1 - q.lw * a2 + q.hw * a3 + q.lw * a4;
                const T gw = -q.hh * a1 + q.hh * a2 - q.lh * a3 + q.lh * a4;
                const T val = q.w1 * a1 + q.w2 * a2 + q.w3 * a3 + q.w4 * a4;
                ga[u] += go * val;
                gx[u] += (T)Ws[u] * gw * tgv;
                gy[u] += (T)Hs[u] * gh * tgv;
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const T sx = wave_sum(gx[u]), sy = wave_sum(gy[u]), sa = wave_sum(ga[u]);
            if (lane == 0 && i0 + u < LP) {
                grad_aw[(size_t)row * LP + i0 + u] = sa;
                grad_loc[(size_t)row * 2 * LP + 2 * (i0 + u)] = sx;
                grad_loc[(size_t)row * 2 * LP + 2 * (i0 + u) + 1] = sy;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// bf16 value maps (training path: value_proj output stays bf16, no fp32 staging copy).  D = 64.
// Forward: a (pixel, head) slice is 128 bytes = 8 lanes x 16 bytes, so a wave covers 8 sampling points at once.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}

template <int NIT>
__global__ __launch_bounds__(kWave * kRowsPerBlock)
void msda_fwd_bf16_d64(const __hip_bfloat16* __restrict__ value, const int64_t* __restrict__ shapes,
                       const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ aw,
                       int S, int M, int L, int Lq, int P, __hip_bfloat16* __restrict__ out, int nrows, int nblk,
                       int pix_el) {
    constexpr int D = 64, LPP = 8, G = 8;
    const int lane = threadIdx.x & (kWave - 1);
    const int row = xcd_logical_block(blockIdx.x, nblk) * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int LP = L * P;
    const int m = row % M;
    const int b = (row / M) / Lq;
    const float locv = lane < 2 * LP ? loc[(size_t)row * 2 * LP + lane] : 0.f;
    const float awv = lane < LP ? aw[(size_t)row * LP + lane] : 0.f;
    const int g = lane / LPP, c8 = lane % LPP;
    const size_t pix_stride = (size_t)pix_el;
    const __hip_bfloat16* vrow = value + (size_t)b * S * pix_stride + (size_t)m * D + c8 * 8;

    uint4 v[NIT][4];
    float cw[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = it * G + g;
        const bool have = idx < LP;
        const int idc = have ? idx : LP - 1;
        const float x = __shfl(locv, 2 * idc, kWave);
        const float y = __shfl(locv, 2 * idc + 1, kWave);
        const float wt = have ? __shfl(awv, idc, kWave) : 0.f;
        const int l = idc / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const __hip_bfloat16* vl = vrow + (size_t)lsi[l] * pix_stride;
        const Corners<float> c = make_corners<float>(x, y, H, W);
        v[it][0] = *reinterpret_cast<const uint4*>(vl + (size_t)c.o1 * pix_stride);
        v[it][1] = *reinterpret_cast<const uint4*>(vl + (size_t)c.o2 * pix_stride);
        v[it][2] = *reinterpret_cast<const uint4*>(vl + (size_t)c.o3 * pix_stride);
        v[it][3] = *reinterpret_cast<const uint4*>(vl + (size_t)c.o4 * pix_stride);
        cw[it][0] = c.k1 ? c.w1 * wt : 0.f;
        cw[it][1] = c.k2 ? c.w2 * wt : 0.f;
        cw[it][2] = c.k3 ? c.w3 * wt : 0.f;
        cw[it][3] = c.k4 ? c.w4 * wt : 0.f;
    }
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w = cw[it][k];
            float f[8];
            unpack8(v[it][k], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w != 0.f ? w * f[e] : 0.f;
        }
#pragma unroll
    for (int off = LPP; off < kWave; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], off, kWave);
    if (lane < LPP) {
        typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
        v8bf o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)acc[e];
        *reinterpret_cast<v8bf*>(out + (size_t)row * D + lane * 8) = o;
    }
}

// Forward for bf16 maps, D = 64, L*P <= 16 -- the training-step shape -- with the gather latency taken off the
// critical path.  The wave-per-row kernel above is latency x occupancy bound (two dependent HBM hops per wave:
// sampling geometry, then the corners; ~5 us of wave lifetime for < 1 us of work).  Here a wave owns FOUR rows:
//   * geometry: lane j computes point (j & 15) of row r0 + (j >> 4) ONCE (one coalesced float2 + float load for the
//     4 rows' 64 points) and parks {4 corner offsets, 4 corner weights x attention weight} in a wave-private 2 KB LDS
//     tile -- no ds_bpermute traffic, no 8x redundant corner arithmetic;
//   * gather: each 16-lane DPP row serves one output row; its two 8-lane halves walk 8 points each (a 128-byte
//     pixel-head slice = 8 lanes x 16 B), BATCH points (4*BATCH independent 16-byte loads per lane) in flight at a
//     time, addresses = uniform base + 32-bit offsets;
//   * fold: the two halves meet in ONE DPP row_ror:8 add per channel (no LDS shuffles), lanes 0-7 of each row store
//     the 128-byte output row.
template <int BATCH>
__global__ __launch_bounds__(256)
void msda_fwd_bf16_rows4(const __hip_bfloat16* __restrict__ value, const int64_t* __restrict__ shapes,
                         const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ aw,
                         int S, int M, int L, int Lq, int P, __hip_bfloat16* __restrict__ out, int nrows, int nblk,
                         int pix_el) {
    constexpr int D = 64;
    __shared__ uint4 geo[4][64][2];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = (xcd_logical_block(blockIdx.x, nblk) * 4 + wave) * 4;
    if (r0 >= nrows) return;
    const int LP = L * P;
    {
        const int row = r0 + (lane >> 4), p = lane & 15;
        const bool valid = row < nrows && p < LP;
        const int rc = min(row, nrows - 1), pc = min(p, LP - 1);
#ifdef GRIT_MSDA_NT
        typedef float v2f_t __attribute__((ext_vector_type(2)));
        const v2f_t xyv = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(loc + ((size_t)rc * LP + pc) * 2));
        const float2 xy = make_float2(xyv[0], xyv[1]);
        const float wt = valid ? __builtin_nontemporal_load(aw + (size_t)rc * LP + pc) : 0.f;
#else
        const float2 xy = *reinterpret_cast<const float2*>(loc + ((size_t)rc * LP + pc) * 2);
        const float wt = valid ? aw[(size_t)rc * LP + pc] : 0.f;
#endif
        const int l = pc / P;
        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
        const uint32_t m = rc % M, b = (rc / M) / Lq;
        // byte offsets from `value` (pix_el elements per pixel, M*D when the map is dense); the launcher checks that
        // the whole map fits 32 bits
        const uint32_t pix = (uint32_t)pix_el * 2;
        const uint32_t base = (b * (uint32_t)S + (uint32_t)lsi[l]) * pix + m * (D * 2);
        const Corners<float> c = make_corners<float>(xy.x, xy.y, H, W);
        geo[wave][lane][0] = make_uint4(base + c.o1 * pix, base + c.o2 * pix, base + c.o3 * pix, base + c.o4 * pix);
        geo[wave][lane][1] = make_uint4(__float_as_uint(c.k1 ? c.w1 * wt : 0.f), __float_as_uint(c.k2 ? c.w2 * wt : 0.f),
                                        __float_as_uint(c.k3 ? c.w3 * wt : 0.f), __float_as_uint(c.k4 ? c.w4 * wt : 0.f));
    }
    __builtin_amdgcn_wave_barrier();
    const int c8 = lane & 7;
    const uint32_t cb = c8 * 16;
    const uint4(*g)[2] = &geo[wave][(lane & 48) + (lane & 8)];  // this half's 8 points of this row
    const char* vb = reinterpret_cast<const char*>(value);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};  // channels (2j, 2j+1): v_pk_fma_f32
#pragma unroll
    for (int it0 = 0; it0 < 8; it0 += BATCH) {
        uint4 v[BATCH][4];
        float w[BATCH][4];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const uint4 o = g[it0 + i][0], ww = g[it0 + i][1];
            v[i][0] = *reinterpret_cast<const uint4*>(vb + (o.x + cb));
            v[i][1] = *reinterpret_cast<const uint4*>(vb + (o.y + cb));
            v[i][2] = *reinterpret_cast<const uint4*>(vb + (o.z + cb));
            v[i][3] = *reinterpret_cast<const uint4*>(vb + (o.w + cb));
            w[i][0] = __uint_as_float(ww.x); w[i][1] = __uint_as_float(ww.y);
            w[i][2] = __uint_as_float(ww.z); w[i][3] = __uint_as_float(ww.w);
        }
        __builtin_amdgcn_sched_barrier(0);  // every gather of the batch is in flight before the first one is consumed
#pragma unroll
        for (int i = 0; i < BATCH; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // a corner outside the map carries weight 0 and a clamped address: whatever was loaded there (even a
                // NaN) must contribute exactly nothing, as in the reference, which never reads it
                const bool dead = w[i][k] == 0.f;
                const uint32_t d[4] = {dead ? 0u : v[i][k].x, dead ? 0u : v[i][k].y, dead ? 0u : v[i][k].z, dead ? 0u : v[i][k].w};
                const f2 w2 = {w[i][k], w[i][k]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f2 f = {__uint_as_float(d[j] << 16), __uint_as_float(d[j] & 0xffff0000u)};
                    acc[j] = __builtin_elementwise_fma(w2, f, acc[j]);
                }
            }
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // lanes l and l ^ 8 of a 16-lane row: row_ror:8
        r[2 * j] = acc[j].x + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[j].x), 0x128, 0xf, 0xf, true));
        r[2 * j + 1] = acc[j].y + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc[j].y), 0x128, 0xf, 0xf, true));
    }
    const int row = r0 + (lane >> 4);
    if ((lane & 8) == 0 && row < nrows) {
        typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
        v8bf o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)r[e];
#ifdef GRIT_MSDA_NT
        __builtin_nontemporal_store(o, reinterpret_cast<v8bf*>(out + (size_t)row * D + c8 * 8));
#else
        *reinterpret_cast<v8bf*>(out + (size_t)row * D + c8 * 8) = o;
#endif
    }
}

// grad_loc / grad_attn_w for bf16 maps, D = 64, L*P <= 16, in the shape of msda_fwd_bf16_rows4: a wave owns four rows, lane j
// computes point (j & 15) of row r0 + (j >> 4) once and parks the corner offsets in LDS; each 16-lane row serves one (b, q, m)
// row, its two 8-lane halves walk 8 points each with 8 channels (16 B) per lane.  Everything the two gradients need from the
// value map are the four scalars  a_k = <grad_out row, corner k row>  per point (ms_deform_im2col_cuda.cuh:97-158 with the
// channel sum pulled inside): 8 multiply-adds per gathered 16 bytes, then ONE 32-value butterfly over the 8 lanes of a half
// (28 exchanges for four rows; the one-row-per-wave walk above spends 63 per row), after which lane j holds the four dots of
// exactly the point whose geometry it computed.
template <int BATCH>
__global__ __launch_bounds__(256)
void msda_bwd_rows4(const __hip_bfloat16* __restrict__ value, const int64_t* __restrict__ shapes,
                    const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ aw,
                    const __hip_bfloat16* __restrict__ grad_out, int S, int M, int L, int Lq, int P,
                    float* __restrict__ grad_loc, float* __restrict__ grad_aw, int nrows, int nblk, int pix_el) {
    constexpr int D = 64;
    __shared__ uint4 geo[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = (xcd_logical_block(blockIdx.x, nblk) * 4 + wave) * 4;
    if (r0 >= nrows) return;
    const int LP = L * P;
    const int row = r0 + (lane >> 4), p = lane & 15;
    const bool valid = row < nrows && p < LP;
    const int rc = min(row, nrows - 1), pc = min(p, LP - 1);
    const float2 xy = *reinterpret_cast<const float2*>(loc + ((size_t)rc * LP + pc) * 2);
    const float wt = valid ? aw[(size_t)rc * LP + pc] : 0.f;
    const int l = pc / P;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const Corners<float> c = make_corners<float>(xy.x, xy.y, H, W);
    {
        const uint32_t m = rc % M, b = (rc / M) / Lq;
        const uint32_t pix = (uint32_t)pix_el * 2;
        const uint32_t base = (b * (uint32_t)S + (uint32_t)lsi[l]) * pix + m * (D * 2);
        // a dead corner keeps its clamped (legal) address; bit 0 of the byte offset marks it (offsets are multiples of 16)
        geo[wave][lane] = make_uint4((base + c.o1 * pix) | (c.k1 ? 0u : 1u), (base + c.o2 * pix) | (c.k2 ? 0u : 1u),
                                     (base + c.o3 * pix) | (c.k3 ? 0u : 1u), (base + c.o4 * pix) | (c.k4 ? 0u : 1u));
    }
    __builtin_amdgcn_wave_barrier();
    const int c8 = lane & 7;
    const uint32_t cb = c8 * 16;
    const uint4* g = &geo[wave][(lane & 48) + (lane & 8)];  // this half's 8 points of this row
    const char* vb = reinterpret_cast<const char*>(value);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 go2[4];
    {
        const uint4 gq = *reinterpret_cast<const uint4*>(grad_out + (size_t)rc * D + c8 * 8);
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) go2[j] = f2{__uint_as_float(gw[j] << 16), __uint_as_float(gw[j] & 0xffff0000u)};
    }
    float d[32];  // d[4 i + k]: this lane's 8 channels of <grad_out, corner k of this half's point i>
#pragma unroll
    for (int it0 = 0; it0 < 8; it0 += BATCH) {
        uint4 v[BATCH][4];
        uint4 o[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            o[i] = g[it0 + i];
            v[i][0] = *reinterpret_cast<const uint4*>(vb + ((o[i].x & ~1u) + cb));
            v[i][1] = *reinterpret_cast<const uint4*>(vb + ((o[i].y & ~1u) + cb));
            v[i][2] = *reinterpret_cast<const uint4*>(vb + ((o[i].z & ~1u) + cb));
            v[i][3] = *reinterpret_cast<const uint4*>(vb + ((o[i].w & ~1u) + cb));
        }
        __builtin_amdgcn_sched_barrier(0);  // every gather of the batch is in flight before the first one is consumed
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const uint32_t dead[4] = {o[i].x & 1u, o[i].y & 1u, o[i].z & 1u, o[i].w & 1u};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // a corner outside the map: whatever sits at its clamped address (even a NaN) contributes exactly nothing
                const uint32_t w4[4] = {dead[k] ? 0u : v[i][k].x, dead[k] ? 0u : v[i][k].y, dead[k] ? 0u : v[i][k].z,
                                        dead[k] ? 0u : v[i][k].w};
                f2 a = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    a = __builtin_elementwise_fma(go2[j], f2{__uint_as_float(w4[j] << 16), __uint_as_float(w4[j] & 0xffff0000u)}, a);
                d[4 * (it0 + i) + k] = a.x + a.y;
            }
        }
    }
    // 32 values over the 8 lanes of the half: three halving exchanges (lane bit 2 <-> point bit 2, ...), after which lane j of the
    // half holds d[0..3] = the four dots of point j of the half, i.e. of point (lane & 15) of its row
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool hi = lane & 4;
        const float keep = hi ? d[16 + i] : d[i], send = hi ? d[i] : d[16 + i];
        d[i] = keep + __shfl_xor(send, 4, kWave);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool hi = lane & 2;
        const float keep = hi ? d[8 + i] : d[i], send = hi ? d[i] : d[8 + i];
        d[i] = keep + __shfl_xor(send, 2, kWave);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool hi = lane & 1;
        const float keep = hi ? d[4 + i] : d[i], send = hi ? d[i] : d[4 + i];
        d[i] = keep + __shfl_xor(send, 1, kWave);
    }
    if (valid) {
        const float a1 = d[0], a2 = d[1], a3 = d[2], a4 = d[3];  // already zero for dead corners
        const float gh = -c.hw * a1 - c.lw * a2 + c.hw * a3 + c.lw * a4;
        const float gw = -c.hh * a1 + c.hh * a2 - c.lh * a3 + c.lh * a4;
        const float val = c.w1 * a1 + c.w2 * a2 + c.w3 * a3 + c.w4 * a4;
        grad_aw[(size_t)row * LP + p] = val;
        *reinterpret_cast<float2*>(grad_loc + ((size_t)row * LP + p) * 2) = make_float2((float)W * wt * gw, (float)H * wt * gh);
    }
}

__device__ __forceinline__ float ldv(const float* p) { return *p; }
__device__ __forceinline__ float ldv(const __hip_bfloat16* p) { return __bfloat162float(*p); }


// 64-value butterfly: every lane holds v[0..63]; after 6 halving exchanges lane L holds the wave-wide sum of
// v[bitreverse6(L)].  63 shuffles instead of 64 x 6 for separate reductions.
template <int N>
__device__ __forceinline__ void halve_step(float (&v)[64], int lane, int off) {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const bool hi = lane & off;
        const float keep = hi ? v[2 * i + 1] : v[2 * i];
        const float send = hi ? v[2 * i] : v[2 * i + 1];
        v[i] = keep + __shfl_xor(send, off, kWave);
    }
}

// Backward for D = 64, L*P <= 16, value / grad_out in f32 or bf16: one wave per (b, q, m) row, lane = channel.
// (Measured alternatives, see profiles/r01/msda_bwd_ablation.txt: keeping the coarse levels' gradient in LDS with
//  ds_add_f32 is SLOWER than global float atomics on gfx950 -- ~150 cycles per LDS float-atomic wave instruction --
//  and software-pipelining rows at 8 waves/CU loses more from the lower occupancy than it hides.)
template <typename VT>
__global__ __launch_bounds__(kWave * kRowsPerBlock)
void msda_bwd_d64(const VT* __restrict__ value, const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi,
                  const float* __restrict__ loc, const float* __restrict__ aw, const VT* __restrict__ grad_out,
                  int S, int M, int L, int Lq, int P, float* __restrict__ grad_value, float* __restrict__ grad_loc,
                  float* __restrict__ grad_aw, int nrows, int nblk) {
    constexpr int D = 64, kMaxLP = 16;
    const int lane = threadIdx.x & (kWave - 1);
    const int row = xcd_logical_block(blockIdx.x, nblk) * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int LP = L * P;
    const int m = row % M;
    const int b = (row / M) / Lq;
    const int pix_stride = M * D;
    const size_t head_off = (size_t)b * S * pix_stride + (size_t)m * D;
    const VT* vhead = value + head_off + lane;
    float* ghead = grad_value + head_off + lane;

    // geometry ONCE per row, point p on lane p (the reference and the generic kernel redo it on every channel lane);
    // the per-point scalars then reach all lanes through v_readlane (SGPRs: scalar branches, scalar addresses)
    const int pi = min(lane, LP - 1), pl = pi / P;
    const int pH = (int)shapes[2 * pl], pW = (int)shapes[2 * pl + 1], pst = (int)lsi[pl];
    const float go = ldv(grad_out + (size_t)row * D + lane);
    const float px = loc[(size_t)row * 2 * LP + 2 * pi], py = loc[(size_t)row * 2 * LP + 2 * pi + 1];
    const float pwt = lane < LP ? aw[(size_t)row * LP + pi] : 0.f;
    const Corners<float> c = make_corners<float>(px, py, pH, pW);
    const int e1 = pst + c.o1, e2 = pst + c.o2, e3 = pst + c.o3, e4 = pst + c.o4;
    const int flags = lane < LP ? ((c.k1 ? 1 : 0) | (c.k2 ? 2 : 0) | (c.k3 ? 4 : 0) | (c.k4 ? 8 : 0)) : 0;
    const float fW = (float)pW * pwt, fH = (float)pH * pwt;

    float part[64];  // [0,16): grad_attn_w per point, [16,48): grad_loc (x, y) per point, rest zero
#pragma unroll
    for (int i = 0; i < 64; ++i) part[i] = 0.f;
#pragma unroll
    for (int p = 0; p < kMaxLP; ++p) {
        if (p < LP) {
            const int f = __builtin_amdgcn_readlane(flags, p);
            const int s1 = __builtin_amdgcn_readlane(e1, p), s2 = __builtin_amdgcn_readlane(e2, p);
            const int s3 = __builtin_amdgcn_readlane(e3, p), s4 = __builtin_amdgcn_readlane(e4, p);
            const float lh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.lh), p));
            const float lw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.lw), p));
            const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pwt), p));
            const float sW = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fW), p));
            const float sH = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(fH), p));
            const float hh = 1.f - lh, hw = 1.f - lw;
            // clamped indices are always legal: load all four, select by the validity bits
            const float r1 = ldv(vhead + (size_t)s1 * pix_stride), r2 = ldv(vhead + (size_t)s2 * pix_stride);
            const float r3 = ldv(vhead + (size_t)s3 * pix_stride), r4 = ldv(vhead + (size_t)s4 * pix_stride);
            const float a1 = (f & 1) ? r1 : 0.f, a2 = (f & 2) ? r2 : 0.f, a3 = (f & 4) ? r3 : 0.f, a4 = (f & 8) ? r4 : 0.f;
            const float tgv = go * wt;
            if (f & 1) atomic_add_fast(ghead + (size_t)s1 * pix_stride, hh * hw * tgv);
            if (f & 2) atomic_add_fast(ghead + (size_t)s2 * pix_stride, hh * lw * tgv);
            if (f & 4) atomic_add_fast(ghead + (size_t)s3 * pix_stride, lh * hw * tgv);
            if (f & 8) atomic_add_fast(ghead + (size_t)s4 * pix_stride, lh * lw * tgv);
            const float gh = -hw * a1 - lw * a2 + hw * a3 + lw * a4;
            const float gw = -hh * a1 + hh * a2 - lh * a3 + lh * a4;
            const float val = hh * hw * a1 + hh * lw * a2 + lh * hw * a3 + lh * lw * a4;
            part[p] = go * val;
            part[16 + 2 * p] = sW * gw * go;
            part[17 + 2 * p] = sH * gh * go;
        }
    }
    halve_step<64>(part, lane, 32);
    halve_step<32>(part, lane, 16);
    halve_step<16>(part, lane, 8);
    halve_step<8>(part, lane, 4);
    halve_step<4>(part, lane, 2);
    halve_step<2>(part, lane, 1);
    const int j = (int)(__brev((unsigned)lane) >> 26);  // index of the value this lane now owns
    if (j < LP) grad_aw[(size_t)row * LP + j] = part[0];
    else if (j >= 16 && j < 16 + 2 * LP) grad_loc[(size_t)row * 2 * LP + (j - 16)] = part[0];
}

// Backward for bf16 maps with the value gradient accumulated IN bf16 by packed atomics (global_atomic_pk_add_bf16: two
// channels per 32-bit atomic -- measured 655 G channel-adds/s against 323 G for f32 atomics, tools/micro/atomic_rate.hip;
// the f32 kernel above is bound by exactly that rate).  This is what torch's own bf16 scatter / grid_sample backward do
// (atomic adds on the bf16 tensor); against f32 accumulation + one final rounding the relative L2 error of grad_value goes
// from 1.7e-3 to 4.0e-3 on the benchmark geometry (profiles/r01/msda_bwd_bf16_accumulate.txt).  The f32 kernel stays the
// parity path (f32 maps) and can be forced for bf16 maps (GRIT_MSDA_BWD_F32ACC=1).
// Lane = channel PAIR: the two 32-lane halves of the wave walk the even / odd points of the row, so one wave instruction
// gathers or scatters two 128-byte pixel-head rows; grad_loc / grad_w leave through the same 63-shuffle butterfly.
//
// STAGE = true: the same row walk (gathers, geometry, merges, grad_loc / grad_w) with the value gradient accumulated in
// FLOAT32 -- the reference's atomicAdd precision (ms_deform_im2col_cuda.cuh:125-152) -- into a dense staging map
// stage[B, S, M, 64] that arrives zeroed.  The scatter there is lane = channel with wave-uniform point and weight (one
// 256-byte contiguous wave atomic per corner, the full-rate shape of msda_bwd_d64), the weights taken by v_readlane from the
// lane that computed the point.  Every cell a row touches is marked in cell_flags[B, S, M] (plain byte stores of 1: all
// writers write the same value), so msda_stage_flush visits the touched cells only: it rounds them once into the bf16
// gradient map and leaves stage and flags zeroed for the next launch.
// MODE 0: packed-bf16 atomics into grad_value; MODE 1 (STAGE): f32 atomics into the staging map; MODE 2: NO scatter at all --
// the row walk only produces grad_loc / grad_attn_w, the value gradient comes from the gather-form kernels further down
// (msda_bwd_index / msda_bwd_gather).
template <int MODE>
__global__ __launch_bounds__(kWave * kRowsPerBlock)
void msda_bwd_d64_pk(const __hip_bfloat16* __restrict__ value, const int64_t* __restrict__ shapes,
                     const int64_t* __restrict__ lsi, const float* __restrict__ loc, const float* __restrict__ aw,
                     const __hip_bfloat16* __restrict__ grad_out, int S, int M, int L, int Lq, int P,
                     __hip_bfloat16* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_aw,
                     int nrows, int nblk, int images_interleaved, int merge_disabled, int pix_el,
                     float* __restrict__ stage, unsigned char* __restrict__ cell_flags) {
    constexpr int D = 64, kMaxLP = 16;
    constexpr bool STAGE = MODE == 1, PACKED = MODE == 0;
    typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
    typedef v2bf __attribute__((address_space(1))) * gv2bf_ptr;
    const int lane = threadIdx.x & (kWave - 1);
    int blk = xcd_logical_block(blockIdx.x, nblk);
    if (images_interleaved > 1) {
        // consecutive workgroups take the same query block of DIFFERENT images: workgroups in flight together then
        // scatter into different value maps.  A freshly initialised GRIT puts every query's reference point near the
        // image centre, so neighbouring queries of one image pile their atomics onto the same few hundred pixels.
        const int per_image = nblk / images_interleaved;
        blk = (blk % images_interleaved) * per_image + blk / images_interleaved;
    }
    const int row = blk * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int LP = L * P;
    const int m = row % M;
    const int b = (row / M) / Lq;
    const int pix_stride = pix_el;  // elements per pixel: M * D for a dense map
    const bool odd = lane >= 32;            // this half's points: 2i + odd
    const int cp = lane & 31;               // channels 2cp, 2cp + 1
    const size_t head_off = (size_t)b * S * pix_stride + (size_t)m * D + 2 * cp;
    const __hip_bfloat16* vhead = value + head_off;
    __hip_bfloat16* ghead = grad_value + head_off;

    // geometry once per row, point p on lane p
    const int pi = min(lane, LP - 1), pl = pi / P;
    const int pH = (int)shapes[2 * pl], pW = (int)shapes[2 * pl + 1], pst = (int)lsi[pl];
    const uint32_t gob = *reinterpret_cast<const uint32_t*>(grad_out + (size_t)row * D + 2 * cp);
    const float go0 = __uint_as_float(gob << 16), go1 = __uint_as_float(gob & 0xffff0000u);
    const float px = loc[(size_t)row * 2 * LP + 2 * pi], py = loc[(size_t)row * 2 * LP + 2 * pi + 1];
    const float pwt = lane < LP ? aw[(size_t)row * LP + pi] : 0.f;
    const Corners<float> c = make_corners<float>(px, py, pH, pW);
    const int e1 = pst + c.o1, e2 = pst + c.o2, e3 = pst + c.o3, e4 = pst + c.o4;
    const int flags = lane < LP ? ((c.k1 ? 1 : 0) | (c.k2 ? 2 : 0) | (c.k3 ? 4 : 0) | (c.k4 ? 8 : 0)) : 0;
    const float fW = (float)pW * pwt, fH = (float)pH * pwt;
    // STAGE: channel = lane view of this row's output gradient and of the staging map; touched cells flagged by the lane
    // that owns the point
    float go_ch = 0.f;
    float* shead = nullptr;
    if constexpr (STAGE) {
        go_ch = ldv(grad_out + (size_t)row * D + lane);
        shead = stage + (size_t)b * S * M * D + (size_t)m * D + lane;
        if (lane < LP) {
            unsigned char* fl = cell_flags + (size_t)b * S * M + m;
            if (flags & 1) fl[(size_t)e1 * M] = 1;
            if (flags & 2) fl[(size_t)e2 * M] = 1;
            if (flags & 4) fl[(size_t)e3 * M] = 1;
            if (flags & 8) fl[(size_t)e4 * M] = 1;
        }
    }
#define GRIT_RLF(x, p) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (p)))
#define GRIT_STAGE_ADD(f_, s1_, s2_, s3_, s4_, W1_, W2_, W3_, W4_)                                   \
    do {                                                                                             \
        if ((f_) & 1) atomic_add_fast(shead + (size_t)(s1_) * (M * D), (W1_) * go_ch);               \
        if ((f_) & 2) atomic_add_fast(shead + (size_t)(s2_) * (M * D), (W2_) * go_ch);               \
        if ((f_) & 4) atomic_add_fast(shead + (size_t)(s3_) * (M * D), (W3_) * go_ch);               \
        if ((f_) & 8) atomic_add_fast(shead + (size_t)(s4_) * (M * D), (W4_) * go_ch);               \
    } while (0)

    float part[64];  // [0,16): grad_attn_w per point, [16,48): grad_loc (x, y) per point, rest zero
#pragma unroll
    for (int i = 0; i < 64; ++i) part[i] = 0.f;
#define GRIT_PICK_I(x, i) (odd ? __builtin_amdgcn_readlane((x), 2 * (i) + 1) : __builtin_amdgcn_readlane((x), 2 * (i)))
#define GRIT_PICK_F(x, i) __int_as_float(GRIT_PICK_I(__float_as_int(x), i))
    // all gathers of the row first (in the training step the value map is cold: one HBM round trip per row instead of
    // one per point pair), then the arithmetic and the scatters
    uint32_t rr[kMaxLP / 2][4];
#pragma unroll
    for (int i = 0; i < kMaxLP / 2; ++i) {
        if (2 * i < LP) {
            const int s1 = GRIT_PICK_I(e1, i), s2 = GRIT_PICK_I(e2, i), s3 = GRIT_PICK_I(e3, i), s4 = GRIT_PICK_I(e4, i);
            rr[i][0] = *reinterpret_cast<const uint32_t*>(vhead + (size_t)s1 * pix_stride);
            rr[i][1] = *reinterpret_cast<const uint32_t*>(vhead + (size_t)s2 * pix_stride);
            rr[i][2] = *reinterpret_cast<const uint32_t*>(vhead + (size_t)s3 * pix_stride);
            rr[i][3] = *reinterpret_cast<const uint32_t*>(vhead + (size_t)s4 * pix_stride);
        }
    }
    // A freshly initialised model (and any query whose offsets are still small) puts the P points of a level into ONE
    // pixel cell: the four corner updates of the four points then hit the same four addresses.  That case is detected
    // per level with wave-uniform compares and the bilinear weights are summed first -- 4 half-wave atomics per level
    // instead of 8 full-wave ones, and a quarter of the same-address traffic.  Levels whose points differ take the
    // per-point path unchanged.
    const bool mergeable = MODE != 2 && P == 4 && LP == kMaxLP && !merge_disabled;
    bool merged = false;
#pragma unroll
    for (int i = 0; i < kMaxLP / 2; ++i) {
        if (2 * i < LP) {
            if ((i & 1) == 0) {
                merged = mergeable;
                const int p0 = 2 * i;
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    merged = merged && __builtin_amdgcn_readlane(e1, p0) == __builtin_amdgcn_readlane(e1, p0 + k) &&
                             __builtin_amdgcn_readlane(e2, p0) == __builtin_amdgcn_readlane(e2, p0 + k) &&
                             __builtin_amdgcn_readlane(e3, p0) == __builtin_amdgcn_readlane(e3, p0 + k) &&
                             __builtin_amdgcn_readlane(e4, p0) == __builtin_amdgcn_readlane(e4, p0 + k) &&
                             __builtin_amdgcn_readlane(flags, p0) == __builtin_amdgcn_readlane(flags, p0 + k);
                if (merged) {
                    float W1 = 0.f, W2 = 0.f, W3 = 0.f, W4 = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float lh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.lh), p0 + k));
                        const float lw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.lw), p0 + k));
                        const float wt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pwt), p0 + k));
                        const float hh = (1.f - lh) * wt, lhw = lh * wt;
                        W1 += hh * (1.f - lw); W2 += hh * lw; W3 += lhw * (1.f - lw); W4 += lhw * lw;
                    }
                    const int f = __builtin_amdgcn_readlane(flags, p0);
                    if constexpr (STAGE) {
                        GRIT_STAGE_ADD(f, __builtin_amdgcn_readlane(e1, p0), __builtin_amdgcn_readlane(e2, p0),
                                       __builtin_amdgcn_readlane(e3, p0), __builtin_amdgcn_readlane(e4, p0), W1, W2, W3, W4);
                    } else if (PACKED && !odd) {
                        __hip_bfloat16* g1 = ghead + (size_t)__builtin_amdgcn_readlane(e1, p0) * pix_stride;
                        __hip_bfloat16* g2 = ghead + (size_t)__builtin_amdgcn_readlane(e2, p0) * pix_stride;
                        __hip_bfloat16* g3 = ghead + (size_t)__builtin_amdgcn_readlane(e3, p0) * pix_stride;
                        __hip_bfloat16* g4 = ghead + (size_t)__builtin_amdgcn_readlane(e4, p0) * pix_stride;
                        if (f & 1) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)g1, v2bf{(__bf16)(W1 * go0), (__bf16)(W1 * go1)});
                        if (f & 2) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)g2, v2bf{(__bf16)(W2 * go0), (__bf16)(W2 * go1)});
                        if (f & 4) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)g3, v2bf{(__bf16)(W3 * go0), (__bf16)(W3 * go1)});
                        if (f & 8) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)g4, v2bf{(__bf16)(W4 * go0), (__bf16)(W4 * go1)});
                    }
                }
            }
            const int f = (2 * i + 1 < LP || !odd) ? GRIT_PICK_I(flags, i) : 0;
            const int s1 = GRIT_PICK_I(e1, i), s2 = GRIT_PICK_I(e2, i), s3 = GRIT_PICK_I(e3, i), s4 = GRIT_PICK_I(e4, i);
            const float lh = GRIT_PICK_F(c.lh, i), lw = GRIT_PICK_F(c.lw, i), wt = GRIT_PICK_F(pwt, i);
            const float sW = GRIT_PICK_F(fW, i), sH = GRIT_PICK_F(fH, i);
            const float hh = 1.f - lh, hw = 1.f - lw;
            const float t0 = go0 * wt, t1 = go1 * wt;
            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
            if (!merged) {
                // the two points of this step (one per half-wave) in one cell: both halves would add to the same
                // addresses in the same instruction -- every lane can form the sum itself (the halves hold the same
                // channels), the lower half issues it
                const bool pair = mergeable && 2 * i + 1 < LP &&
                                  __builtin_amdgcn_readlane(e1, 2 * i) == __builtin_amdgcn_readlane(e1, 2 * i + 1) &&
                                  __builtin_amdgcn_readlane(e2, 2 * i) == __builtin_amdgcn_readlane(e2, 2 * i + 1) &&
                                  __builtin_amdgcn_readlane(e3, 2 * i) == __builtin_amdgcn_readlane(e3, 2 * i + 1) &&
                                  __builtin_amdgcn_readlane(e4, 2 * i) == __builtin_amdgcn_readlane(e4, 2 * i + 1) &&
                                  __builtin_amdgcn_readlane(flags, 2 * i) == __builtin_amdgcn_readlane(flags, 2 * i + 1);
                if constexpr (STAGE) {
                    // wave-uniform weights of the step's two points (2i on the lower half, 2i + 1 on the upper)
                    const int pa_ = 2 * i, pb_ = 2 * i + 1;
                    const float lhA = GRIT_RLF(c.lh, pa_), lwA = GRIT_RLF(c.lw, pa_), wtA = GRIT_RLF(pwt, pa_);
                    const float hA = (1.f - lhA) * wtA, lA = lhA * wtA;
                    float A1 = hA * (1.f - lwA), A2 = hA * lwA, A3 = lA * (1.f - lwA), A4 = lA * lwA;
                    const int fA = __builtin_amdgcn_readlane(flags, pa_);
                    if (pb_ < LP) {
                        const float lhB = GRIT_RLF(c.lh, pb_), lwB = GRIT_RLF(c.lw, pb_), wtB = GRIT_RLF(pwt, pb_);
                        const float hB = (1.f - lhB) * wtB, lB = lhB * wtB;
                        const float B1 = hB * (1.f - lwB), B2 = hB * lwB, B3 = lB * (1.f - lwB), B4 = lB * lwB;
                        if (pair) {
                            A1 += B1; A2 += B2; A3 += B3; A4 += B4;
                        } else {
                            GRIT_STAGE_ADD(__builtin_amdgcn_readlane(flags, pb_), __builtin_amdgcn_readlane(e1, pb_),
                                           __builtin_amdgcn_readlane(e2, pb_), __builtin_amdgcn_readlane(e3, pb_),
                                           __builtin_amdgcn_readlane(e4, pb_), B1, B2, B3, B4);
                        }
                    }
                    GRIT_STAGE_ADD(fA, __builtin_amdgcn_readlane(e1, pa_), __builtin_amdgcn_readlane(e2, pa_),
                                   __builtin_amdgcn_readlane(e3, pa_), __builtin_amdgcn_readlane(e4, pa_), A1, A2, A3, A4);
                }
                float u1 = w1 * wt, u2 = w2 * wt, u3 = w3 * wt, u4 = w4 * wt;
                bool issue = PACKED;
                if (pair) {
                    // the other half's point: 2i + 1 for the lower half (the only one that issues)
                    const float olh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.lh), 2 * i + 1));
                    const float olw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.lw), 2 * i + 1));
                    const float owt = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pwt), 2 * i + 1));
                    const float ohh = (1.f - olh) * owt, olhw = olh * owt;
                    u1 += ohh * (1.f - olw); u2 += ohh * olw; u3 += olhw * (1.f - olw); u4 += olhw * olw;
                    issue = PACKED && !odd;
                }
                if (PACKED && issue) {
                    if (f & 1) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)(ghead + (size_t)s1 * pix_stride), v2bf{(__bf16)(u1 * go0), (__bf16)(u1 * go1)});
                    if (f & 2) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)(ghead + (size_t)s2 * pix_stride), v2bf{(__bf16)(u2 * go0), (__bf16)(u2 * go1)});
                    if (f & 4) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)(ghead + (size_t)s3 * pix_stride), v2bf{(__bf16)(u3 * go0), (__bf16)(u3 * go1)});
                    if (f & 8) __builtin_amdgcn_global_atomic_fadd_v2bf16((gv2bf_ptr)(ghead + (size_t)s4 * pix_stride), v2bf{(__bf16)(u4 * go0), (__bf16)(u4 * go1)});
                }
            }
            // clamped indices are always legal: all four were loaded, select by the validity bits
            const uint32_t q1 = (f & 1) ? rr[i][0] : 0u, q2 = (f & 2) ? rr[i][1] : 0u, q3 = (f & 4) ? rr[i][2] : 0u, q4 = (f & 8) ? rr[i][3] : 0u;
            const float a1x = __uint_as_float(q1 << 16), a1y = __uint_as_float(q1 & 0xffff0000u);
            const float a2x = __uint_as_float(q2 << 16), a2y = __uint_as_float(q2 & 0xffff0000u);
            const float a3x = __uint_as_float(q3 << 16), a3y = __uint_as_float(q3 & 0xffff0000u);
            const float a4x = __uint_as_float(q4 << 16), a4y = __uint_as_float(q4 & 0xffff0000u);
            // the two channels of this lane folded right away: go . (d/dh, d/dw, value) of the bilinear form
            const float ghx = -hw * a1x - lw * a2x + hw * a3x + lw * a4x, ghy = -hw * a1y - lw * a2y + hw * a3y + lw * a4y;
            const float gwx = -hh * a1x + hh * a2x - lh * a3x + lh * a4x, gwy = -hh * a1y + hh * a2y - lh * a3y + lh * a4y;
            const float vx = w1 * a1x + w2 * a2x + w3 * a3x + w4 * a4x, vy = w1 * a1y + w2 * a2y + w3 * a3y + w4 * a4y;
            const float pa = go0 * vx + go1 * vy;
            const float plx = sW * (gwx * go0 + gwy * go1), ply = sH * (ghx * go0 + ghy * go1);
            part[2 * i] = odd ? 0.f : pa;          part[2 * i + 1] = odd ? pa : 0.f;
            part[16 + 4 * i] = odd ? 0.f : plx;    part[16 + 4 * i + 2] = odd ? plx : 0.f;
            part[17 + 4 * i] = odd ? 0.f : ply;    part[17 + 4 * i + 2] = odd ? ply : 0.f;
        }
    }
#undef GRIT_PICK_I
#undef GRIT_PICK_F
#undef GRIT_RLF
#undef GRIT_STAGE_ADD
    halve_step<64>(part, lane, 32);
    halve_step<32>(part, lane, 16);
    halve_step<16>(part, lane, 8);
    halve_step<8>(part, lane, 4);
    halve_step<4>(part, lane, 2);
    halve_step<2>(part, lane, 1);
    const int j = (int)(__brev((unsigned)lane) >> 26);  // index of the value this lane now owns
    if (j < LP) grad_aw[(size_t)row * LP + j] = part[0];
    else if (j >= 16 && j < 16 + 2 * LP) grad_loc[(size_t)row * 2 * LP + (j - 16)] = part[0];
}

// Second half of the f32-accumulating bf16 backward: every (image, pixel, head) cell whose flag is set gets its 64 staged
// f32 sums rounded ONCE to bf16 and written to its place in the (possibly strided) gradient map; the staged row and the
// flag are cleared, so stage / cell_flags leave the launch as they must enter the next one: all zero.  Untouched cells are
// not read (the gradient map was zero-filled): at GRIT's shapes the gather reaches 20-45 % of the cells.
// A wave owns 64 consecutive cells (lane = cell for the flag read, lane = channel for the rows), four rows in flight.
__global__ __launch_bounds__(256)
void msda_stage_flush(float* __restrict__ stage, unsigned char* __restrict__ cell_flags, __hip_bfloat16* __restrict__ grad_value,
                      long ncells, int S, int M, int pix_el) {
    const int lane = threadIdx.x & 63;
    const long base = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (base >= ncells) return;
    const long cell = base + lane;
    const unsigned char fl = cell < ncells ? cell_flags[cell] : (unsigned char)0;
    unsigned long long mask = __ballot(fl != 0);
    if (fl) cell_flags[cell] = 0;
    while (mask) {
        long c[4];
        float v[4];
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (mask) {
                const int bit = __builtin_ctzll(mask);
                mask &= mask - 1;
                c[k] = base + bit;
                n = k + 1;
            } else {
                c[k] = -1;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) v[k] = stage[c[k] * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) {
                stage[c[k] * 64 + lane] = 0.f;
                const long pix = c[k] / M;  // b * S + pixel
                const int m = (int)(c[k] - pix * M);
                grad_value[pix * pix_el + (long)m * 64 + lane] = __float2bfloat16(v[k]);
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// Gather-form value gradient (bf16 maps, D = 64): no atomics on memory at all.
//
// The scatter of the reference (atomicAdd per corner and channel, ms_deform_im2col_cuda.cuh:125-152) is bound on gfx950 by
// the memory-side atomic units (~323 G f32 adds/s chip-wide, tools/micro/atomic_rate.hip): 157 M channel-adds per launch are
// ~460 us whatever the kernel does around them.  But WHERE every contribution goes is known from the sampling locations
// alone, and per (image, head) there are only Lq*L*P*4 of them (9 600 at GRIT's shapes) over S cells, reading Lq rows of
// grad_out (19 KB): the whole problem of one (image, head) fits the LDS of one CU.  msda_bwd_value, one 1024-thread workgroup
// per (image, head) -- 256 of them at B = 32, M = 8: one per CU --
//   1. loads its Lq rows of grad_out into LDS; bins the (query, point, corner) contributions by cell with LDS counters (the
//      value an LDS atomic returns is the contribution's position inside its cell's run);
//   2. scans the S counters in LDS (exclusive offsets);
//   3. writes the contributions, ordered by cell, into an LDS array {query, attention weight x bilinear weight};
//   4. walks the cells: eight lanes per cell (8 channels each), eight cells per wave instruction; a cell's lanes sum
//      w_i * grad_out[q_i][channels] in f32 registers, round ONCE to bf16 and store their 128-byte row -- every cell, zeros
//      included, so the gradient map needs no zero fill, no f32 staging copy and no flush, and nothing but the final rows
//      ever goes to memory.
// grad_loc / grad_attn_w come from the unchanged row walk with its scatter compiled out (msda_bwd_d64_pk<2>).
// f32 accumulation like the reference's atomicAdd; the order of the terms inside a cell follows the order in which the LDS
// counters were taken (like atomics, not fixed run to run, but the sum is rounded to bf16 once, after the last term).
constexpr int kValueThreads = 1024, kValuePts = 4;   // up to 4 096 (query, point) pairs per (image, head)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf_m __attribute__((ext_vector_type(8)));
typedef short v4s_m __attribute__((ext_vector_type(4)));
typedef short v8s_m __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) v4s_m lds_v4s_m;

__device__ __forceinline__ float bf16_round(float x) {  // x rounded to the nearest bf16 (ties to even), as a float
    return __bfloat162float(__float2bfloat16(x));
}

__host__ __device__ inline size_t value_lds_bytes(int S, int L, int Lq, int P) {
    const size_t counters = ((size_t)S + 2) / 2 * 2 * 4;          // S + 1 offsets, padded to 8 bytes
    return counters + (size_t)Lq * L * P * 4 * 8 + ((size_t)Lq + 1) * 128 + 64;  // + one all-zero row of grad_out
}

__global__ __launch_bounds__(kValueThreads)
void msda_bwd_value(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lsi, const float* __restrict__ loc,
                    const float* __restrict__ aw, const __hip_bfloat16* __restrict__ grad_out,
                    __hip_bfloat16* __restrict__ grad_value, int S, int M, int L, int Lq, int P, int pix_el, int parts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LP = L * P, npts = Lq * LP, cap = npts * 4;
    int* cnt = reinterpret_cast<int*>(smem);                                     // [S + 1] counters, then exclusive offsets
    uint2* recs = reinterpret_cast<uint2*>(smem + ((size_t)S + 2) / 2 * 2 * 4);  // [cap] {query, weight}
    unsigned char* gol = reinterpret_cast<unsigned char*>(recs + cap);           // [Lq][128 B] rows of grad_out
    int* wsum = reinterpret_cast<int*>(gol + ((size_t)Lq + 1) * 128);            // [16] wave totals of the scan
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // parts > 1 (B * M below the CU count, e.g. 16 images x 8 heads on 256 CUs): `parts` workgroups per (image, head); each bins all
    // of the segment's contributions (steps 1 - 3, the cheap part) and takes an equal-cost share of the cell tiles in step 4
    const int seg = blockIdx.x / parts, part = blockIdx.x - seg * parts, b = seg / M, m = seg - b * M;

    // ---- 1. counters to zero, grad_out rows of this (image, head) into LDS, contributions binned
    for (int i = tid; i <= S; i += kValueThreads) cnt[i] = 0;
    if (tid < 32) reinterpret_cast<unsigned*>(gol + (size_t)Lq * 128)[tid] = 0u;  // row Lq: zeros (k slots past a run)
    for (int i = tid; i < Lq * 8; i += kValueThreads) {
        const int q = i >> 3, piece = i & 7;
        *reinterpret_cast<uint4*>(gol + (size_t)q * 128 + piece * 16) =
            *reinterpret_cast<const uint4*>(grad_out + ((size_t)(b * Lq + q) * M + m) * 64 + piece * 8);
    }
    __syncthreads();
    int cell[kValuePts][4], rank[kValuePts][4], qid[kValuePts];
    float wgt[kValuePts][4];
#pragma unroll
    for (int j = 0; j < kValuePts; ++j) {
        const int i = tid + j * kValueThreads;
#pragma unroll
        for (int k = 0; k < 4; ++k) cell[j][k] = -1;
        if (i < npts) {
            const int q = i / LP, p = i - q * LP, l = p / P;
            const size_t row = (size_t)(b * Lq + q) * M + m;
            qid[j] = q;
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1], st = (int)lsi[l];
            const float x = loc[row * 2 * LP + 2 * p], y = loc[row * 2 * LP + 2 * p + 1];
            const float a = aw[row * LP + p];
            const Corners<float> c = make_corners<float>(x, y, H, W);
            if (c.k1) { cell[j][0] = st + c.o1; wgt[j][0] = c.w1 * a; }
            if (c.k2) { cell[j][1] = st + c.o2; wgt[j][1] = c.w2 * a; }
            if (c.k3) { cell[j][2] = st + c.o3; wgt[j][2] = c.w3 * a; }
            if (c.k4) { cell[j][3] = st + c.o4; wgt[j][3] = c.w4 * a; }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cell[j][k] >= 0) rank[j][k] = atomicAdd(&cnt[cell[j][k]], 1);  // position inside the cell's run
        }
    }
    __syncthreads();
    // ---- 2. exclusive scan of the S counters: a contiguous chunk per thread, wave scans of the chunk sums, 16 wave totals
    const int CH = (S + kValueThreads - 1) / kValueThreads, c0 = tid * CH;
    int local = 0;
    for (int k = 0; k < CH; ++k)
        if (c0 + k < S) local += cnt[c0 + k];
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int v = lane < kValueThreads / 64 ? wsum[lane] : 0;
        int iv = v;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int t = __shfl_up(iv, o, 64);
            if (lane >= o) iv += t;
        }
        if (lane < kValueThreads / 64) wsum[lane] = iv - v;  // exclusive
    }
    __syncthreads();
    int run = incl - local + wsum[wave];
    for (int k = 0; k < CH; ++k)
        if (c0 + k < S) {
            const int c = cnt[c0 + k];
            cnt[c0 + k] = run;
            run += c;
        }
    if (tid == kValueThreads - 1) cnt[S] = run;  // the last thread's running total is the segment's record count
    __syncthreads();
    // ---- 3. contributions ordered by cell.  A record is 8 bytes: query (12 bits), cell within its 16-cell tile (4 bits) and the
    // weight as THREE bf16 terms hi + mid + lo (24 mantissa bits: the f32 weight exactly), the form step 4 feeds to the MFMA
#pragma unroll
    for (int j = 0; j < kValuePts; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cell[j][k] >= 0) {
                const float w = wgt[j][k];
                const unsigned hi = __float_as_uint(bf16_round(w)) >> 16;
                const float r1 = w - __uint_as_float(hi << 16);
                const unsigned mid = __float_as_uint(bf16_round(r1)) >> 16;
                const float r2 = r1 - __uint_as_float(mid << 16);
                const unsigned lo = __float_as_uint(bf16_round(r2)) >> 16;
                recs[cnt[cell[j][k]] + rank[j][k]] =
                    make_uint2((unsigned)qid[j] | ((unsigned)(cell[j][k] & 15) << 12) | (hi << 16), mid | (lo << 16));
            }
    __syncthreads();
    // ---- 4. tiles of 16 consecutive cells:  Out^T[64 channels x 16 cells] = G^T[64 x n] . W[n x 16]  on the matrix cores, where
    // the n contributions of the tile are one contiguous run of records, G^T their grad_out rows (bf16, exact) read straight from
    // the LDS rows by transposing reads (every lane addresses the row of ITS record), and W[k][c] = weight of record k if it
    // belongs to cell c, else 0 -- three bf16 terms, three MFMAs into the same f32 accumulator.  No loop over cells, no
    // divergence between hot and empty cells; a wave takes a contiguous range of tiles of equal estimated cost.
    const int l15 = lane & 15, lg = lane >> 4, trq = l15 >> 2, trp = l15 & 3;
    const int T = (S + 15) >> 4, nwave = kValueThreads / 64;
    const int total = cnt[S] + 8 * T;  // cost model: a record ~ 1, a tile's fixed part ~ 8
    auto first_tile_at = [&](int target) {  // smallest t with cnt[16 t] + 8 t >= target (monotone in t)
        int lo = 0, hi = T;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cnt[min(16 * mid, S)] + 8 * mid >= target) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    const int gw = part * nwave + wave, ngw = parts * nwave;  // this wave among the segment's waves
    const int tlo = gw == 0 ? 0 : first_tile_at((int)((long)total * gw / ngw));
    const int thi = gw == ngw - 1 ? T : first_tile_at((int)((long)total * (gw + 1) / ngw));
    __hip_bfloat16* gbase = grad_value + (size_t)b * S * pix_el + (size_t)m * 64 + 4 * lg;
    const unsigned char* grow = gol + 8 * trp;  // + q * 128 + 32 * cb
    for (int t = tlo; t < thi; ++t) {
        const int beg = __builtin_amdgcn_readfirstlane(cnt[16 * t]);
        const int end = __builtin_amdgcn_readfirstlane(cnt[min(16 * t + 16, S)]);
        v4f acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int k0 = beg; k0 < end; k0 += 32) {
            // k slots of this lane (the order the transposing reads define): e < 4: k0 + 4 lg + e,  e >= 4: k0 + 16 + 4 lg + e - 4
            // masked record words first (2 selects per record), then three byte permutes per PAIR of records gather the hi /
            // mid / lo halves into the packed operand registers
            unsigned rx[8], ry[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + 4 * lg + (e & 3) + (e >> 2) * 16;
                const uint2 r = recs[min(k, cap - 1)];
                const bool mine = k < end && ((r.x >> 12) & 15u) == (unsigned)l15;
                rx[e] = mine ? r.x : 0u;
                ry[e] = mine ? r.y : 0u;
            }
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            u4v ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // v_perm_b32(hi word, lo word, selector): selector bytes 0-3 index the low word, 4-7 the high word
                ph[e] = __builtin_amdgcn_perm(rx[2 * e + 1], rx[2 * e], 0x07060302u);  // hi halves: x >> 16
                pm[e] = __builtin_amdgcn_perm(ry[2 * e + 1], ry[2 * e], 0x05040100u);  // mid: y & 0xffff
                pl[e] = __builtin_amdgcn_perm(ry[2 * e + 1], ry[2 * e], 0x07060302u);  // lo: y >> 16
            }
            const v8bf_m bh = __builtin_bit_cast(v8bf_m, ph), bm = __builtin_bit_cast(v8bf_m, pm), bl = __builtin_bit_cast(v8bf_m, pl);
            // rows this lane addresses for the transposing reads: records k0 + 4 lg + trq and + 16 (past the run: the zero row)
            const int ka = k0 + 4 * lg + trq, kb = ka + 16;
            const unsigned qa = ka < end ? (recs[min(ka, cap - 1)].x & 0xfffu) : (unsigned)Lq;
            const unsigned qb = kb < end ? (recs[min(kb, cap - 1)].x & 0xfffu) : (unsigned)Lq;
            const unsigned char* ra = grow + (size_t)qa * 128;
            const unsigned char* rb = grow + (size_t)qb * 128;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const v4s_m x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_m*)(ra + 32 * cb));
                const v4s_m x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_m*)(rb + 32 * cb));
                const v8s_m xs = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                const v8bf_m a = __builtin_bit_cast(v8bf_m, xs);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bh, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bm, acc[cb], 0, 0, 0);
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bl, acc[cb], 0, 0, 0);
            }
        }
        // acc[cb][r] = Out[cell 16 t + l15][channel 16 cb + 4 lg + r]
        const int cellw = 16 * t + l15;
        if (cellw < S) {
            __hip_bfloat16* dst = gbase + (size_t)cellw * pix_el;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                typedef __bf16 v4bf_m __attribute__((ext_vector_type(4)));
                const v4bf_m o = {(__bf16)acc[cb][0], (__bf16)acc[cb][1], (__bf16)acc[cb][2], (__bf16)acc[cb][3]};
                *reinterpret_cast<v4bf_m*>(dst + 16 * cb) = o;
            }
        }
    }
}

bool dims_ok(int B, int S, int M, int D, int L, int Lq, int P) {
    if (B <= 0 || S <= 0 || M <= 0 || D <= 0 || L <= 0 || Lq <= 0 || P <= 0) return false;
    const long long rows = (long long)B * Lq * M;
    return rows < (1LL << 31) - 8 && (long long)L * P < (1 << 20);
}

template <typename T>
int launch_fwd(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* aw,
               int B, int S, int M, int D, int L, int Lq, int P, T* out, hipStream_t st) {
    if (!value || !shapes || !lsi || !loc || !aw || !out) return GRIT_ERR_BAD_ARG;
    if (!dims_ok(B, S, M, D, L, Lq, P)) return GRIT_ERR_BAD_ARG;
    const int nrows = B * Lq * M;
    const int nblk = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    const dim3 grid(nblk), block(kWave * kRowsPerBlock);
    bool done = false;
    if constexpr (sizeof(T) == 4) {
        const int LP = L * P;
        const bool aligned = ((uintptr_t)value % 16 == 0) && ((uintptr_t)out % 16 == 0);
#define GRIT_FWD_CASE(LPP_, NIT_)                                                                    \
    hipLaunchKernelGGL((msda_fwd_vec4<LPP_, NIT_>), grid, block, 0, st, value, shapes, lsi, loc, aw, \
                       S, M, L, Lq, P, out, nrows, nblk);                                            \
    done = true
        if (aligned && D == 64 && LP <= 32) {
            const int nit = (LP + 3) / 4;
            if (nit == 1) { GRIT_FWD_CASE(16, 1); } else if (nit == 2) { GRIT_FWD_CASE(16, 2); }
            else if (nit == 3) { GRIT_FWD_CASE(16, 3); } else if (nit == 4) { GRIT_FWD_CASE(16, 4); }
            else if (nit == 6) { GRIT_FWD_CASE(16, 6); } else if (nit == 8) { GRIT_FWD_CASE(16, 8); }
        } else if (aligned && D == 32 && LP <= 32) {
            const int nit = (LP + 7) / 8;
            if (nit == 1) { GRIT_FWD_CASE(8, 1); } else if (nit == 2) { GRIT_FWD_CASE(8, 2); }
            else if (nit == 3) { GRIT_FWD_CASE(8, 3); } else if (nit == 4) { GRIT_FWD_CASE(8, 4); }
        }
#undef GRIT_FWD_CASE
    }
    if (!done)
        hipLaunchKernelGGL((msda_fwd_generic<T>), grid, block, 0, st, value, shapes, lsi, loc, aw,
                           S, M, D, L, Lq, P, out, nrows, nblk);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

template <typename VT>
int launch_bwd_d64(const VT* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* aw,
                   const VT* go, int B, int S, int M, int L, int Lq, int P, float* gv, float* gl, float* gw,
                   hipStream_t st) {
    const int nrows = B * Lq * M;
    const int nblk = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    hipLaunchKernelGGL((msda_bwd_d64<VT>), dim3(nblk), dim3(kWave * kRowsPerBlock), 0, st, value, shapes, lsi, loc, aw, go,
                       S, M, L, Lq, P, gv, gl, gw, nrows, nblk);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

template <typename T>
int launch_bwd(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc, const T* aw,
               const T* go, int B, int S, int M, int D, int L, int Lq, int P,
               T* gv, T* gl, T* gw, hipStream_t st) {
    if (!value || !shapes || !lsi || !loc || !aw || !go || !gv || !gl || !gw) return GRIT_ERR_BAD_ARG;
    if (!dims_ok(B, S, M, D, L, Lq, P)) return GRIT_ERR_BAD_ARG;
    const int nrows = B * Lq * M;
    const int nblk = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    const dim3 grid(nblk), block(kWave * kRowsPerBlock);
    if constexpr (sizeof(T) == 4) {
        if (D == 64 && L * P <= 16) return launch_bwd_d64<float>(value, shapes, lsi, loc, aw, go, B, S, M, L, Lq, P, gv, gl, gw, st);
    }
    if ((L * P) % 4 == 0)
        hipLaunchKernelGGL((msda_bwd_generic<T, 4>), grid, block, 0, st, value, shapes, lsi, loc, aw, go,
                           S, M, D, L, Lq, P, gv, gl, gw, nrows, nblk);
    else
        hipLaunchKernelGGL((msda_bwd_generic<T, 1>), grid, block, 0, st, value, shapes, lsi, loc, aw, go,
                           S, M, D, L, Lq, P, gv, gl, gw, nrows, nblk);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int grit_msda_fwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const float* loc, const float* attn_w, int B, int S, int M, int D, int L, int Lq,
                      int P, float* out, void* stream) {
    return launch_fwd<float>(value, spatial_shapes, level_start, loc, attn_w, B, S, M, D, L, Lq, P, out,
                             (hipStream_t)stream);
}

int grit_msda_fwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const double* loc, const double* attn_w, int B, int S, int M, int D, int L, int Lq,
                      int P, double* out, void* stream) {
    return launch_fwd<double>(value, spatial_shapes, level_start, loc, attn_w, B, S, M, D, L, Lq, P, out,
                              (hipStream_t)stream);
}

int grit_msda_bwd_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const float* loc, const float* attn_w, const float* grad_out, int B, int S, int M,
                      int D, int L, int Lq, int P, float* grad_value, float* grad_loc, float* grad_attn_w,
                      void* stream) {
    return launch_bwd<float>(value, spatial_shapes, level_start, loc, attn_w, grad_out, B, S, M, D, L, Lq,
                             P, grad_value, grad_loc, grad_attn_w, (hipStream_t)stream);
}

int grit_msda_bwd_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start,
                      const double* loc, const double* attn_w, const double* grad_out, int B, int S, int M,
                      int D, int L, int Lq, int P, double* grad_value, double* grad_loc,
                      double* grad_attn_w, void* stream) {
    return launch_bwd<double>(value, spatial_shapes, level_start, loc, attn_w, grad_out, B, S, M, D, L, Lq,
                              P, grad_value, grad_loc, grad_attn_w, (hipStream_t)stream);
}

int grit_msda_fwd_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const float* loc,
                       const float* attn_w, int B, int S, int M, int D, int L, int Lq, int P, void* out, void* stream) {
    return grit_msda_fwd_bf16_strided(value, (long)M * D, spatial_shapes, level_start, loc, attn_w, B, S, M, D, L, Lq, P, out,
                                      stream);
}

int grit_msda_fwd_bf16_strided(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                               const int64_t* level_start, const float* loc, const float* attn_w, int B, int S, int M,
                               int D, int L, int Lq, int P, void* out, void* stream) {
    if (!value || !spatial_shapes || !level_start || !loc || !attn_w || !out) return GRIT_ERR_BAD_ARG;
    if (!dims_ok(B, S, M, D, L, Lq, P)) return GRIT_ERR_BAD_ARG;
    if (pixel_stride < (long)M * D || pixel_stride % 8 || pixel_stride > 0x3fffffffL) return GRIT_ERR_BAD_ARG;
    const int pix_el = (int)pixel_stride;
    const int LP = L * P;
    if (D != 64 || LP > 32 || ((uintptr_t)value % 16) || ((uintptr_t)out % 16)) return GRIT_ERR_UNSUPPORTED;
    const int nrows = B * Lq * M;
    if (LP <= 16 && (uint64_t)B * S * (uint64_t)pixel_stride * 2 < (1ull << 32)) {  // 4 rows per wave, 16 rows per workgroup
        static const int batch = getenv("GRIT_MSDA_FWD_BATCH") ? atoi(getenv("GRIT_MSDA_FWD_BATCH")) : 2;
        const int nblk16 = (nrows + 15) / 16;
#define GRIT_ROWS4(BATCH_)                                                                                           \
    hipLaunchKernelGGL((msda_fwd_bf16_rows4<BATCH_>), dim3(nblk16), dim3(256), 0, (hipStream_t)stream,               \
                       (const __hip_bfloat16*)value, spatial_shapes, level_start, loc, attn_w, S, M, L, Lq, P,       \
                       (__hip_bfloat16*)out, nrows, nblk16, pix_el)
        if (batch == 0) goto row_per_wave;
        if (batch == 1) GRIT_ROWS4(1); else if (batch == 4) GRIT_ROWS4(4); else if (batch == 8) GRIT_ROWS4(8); else GRIT_ROWS4(2);
#undef GRIT_ROWS4
        return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
    }
row_per_wave:
    const int nblk = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    const dim3 grid(nblk), block(kWave * kRowsPerBlock);
    const int nit = (LP + 7) / 8;
#define GRIT_FWD16(NIT_)                                                                                          \
    hipLaunchKernelGGL((msda_fwd_bf16_d64<NIT_>), grid, block, 0, (hipStream_t)stream, (const __hip_bfloat16*)value, \
                       spatial_shapes, level_start, loc, attn_w, S, M, L, Lq, P, (__hip_bfloat16*)out, nrows, nblk, pix_el)
    if (nit == 1) GRIT_FWD16(1); else if (nit == 2) GRIT_FWD16(2); else if (nit == 3) GRIT_FWD16(3); else GRIT_FWD16(4);
#undef GRIT_FWD16
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_msda_bwd_bf16(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const float* loc,
                       const float* attn_w, const void* grad_out, int B, int S, int M, int D, int L, int Lq, int P,
                       float* grad_value, float* grad_loc, float* grad_attn_w, void* stream) {
    if (!value || !spatial_shapes || !level_start || !loc || !attn_w || !grad_out || !grad_value || !grad_loc || !grad_attn_w)
        return GRIT_ERR_BAD_ARG;
    if (!dims_ok(B, S, M, D, L, Lq, P)) return GRIT_ERR_BAD_ARG;
    if (D != 64 || L * P > 16) return GRIT_ERR_UNSUPPORTED;
    return launch_bwd_d64<__hip_bfloat16>((const __hip_bfloat16*)value, spatial_shapes, level_start, loc, attn_w,
                                            (const __hip_bfloat16*)grad_out, B, S, M, L, Lq, P, grad_value, grad_loc,
                                            grad_attn_w, (hipStream_t)stream);
}

int grit_msda_bwd_bf16acc(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const float* loc,
                          const float* attn_w, const void* grad_out, int B, int S, int M, int D, int L, int Lq, int P,
                          void* grad_value, float* grad_loc, float* grad_attn_w, void* stream) {
    return grit_msda_bwd_bf16acc_strided(value, (long)M * D, spatial_shapes, level_start, loc, attn_w, grad_out, B, S, M, D, L,
                                         Lq, P, grad_value, grad_loc, grad_attn_w, stream);
}

int grit_msda_bwd_bf16acc_strided(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                                  const int64_t* level_start, const float* loc, const float* attn_w, const void* grad_out,
                                  int B, int S, int M, int D, int L, int Lq, int P, void* grad_value, float* grad_loc,
                                  float* grad_attn_w, void* stream) {
    if (pixel_stride < (long)M * D || pixel_stride % 2 || pixel_stride > 0x3fffffffL) return GRIT_ERR_BAD_ARG;
    if (!value || !spatial_shapes || !level_start || !loc || !attn_w || !grad_out || !grad_value || !grad_loc || !grad_attn_w)
        return GRIT_ERR_BAD_ARG;
    if (!dims_ok(B, S, M, D, L, Lq, P)) return GRIT_ERR_BAD_ARG;
    if (D != 64 || L * P > 16 || ((uintptr_t)value % 4) || ((uintptr_t)grad_out % 4) || ((uintptr_t)grad_value % 4))
        return GRIT_ERR_UNSUPPORTED;
    const int nrows = B * Lq * M;
    const int nblk = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    // image-interleaved workgroup order when the rows of an image fill whole workgroups (GRIT_MSDA_BWD_INTERLEAVE=0: off)
    static const bool interleave = !(getenv("GRIT_MSDA_BWD_INTERLEAVE") && atoi(getenv("GRIT_MSDA_BWD_INTERLEAVE")) == 0);
    const int images = (interleave && B > 1 && (Lq * M) % kRowsPerBlock == 0) ? B : 1;
    static const bool no_merge = getenv("GRIT_MSDA_BWD_MERGE") && atoi(getenv("GRIT_MSDA_BWD_MERGE")) == 0;  // A/B knob
    hipLaunchKernelGGL(msda_bwd_d64_pk<0>, dim3(nblk), dim3(kWave * kRowsPerBlock), 0, (hipStream_t)stream,
                       (const __hip_bfloat16*)value, spatial_shapes, level_start, loc, attn_w, (const __hip_bfloat16*)grad_out,
                       S, M, L, Lq, P, (__hip_bfloat16*)grad_value, grad_loc, grad_attn_w, nrows, nblk, images, no_merge ? 1 : 0,
                       (int)pixel_stride, (float*)nullptr, (unsigned char*)nullptr);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_msda_bwd_bf16_staged(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                              const int64_t* level_start, const float* loc, const float* attn_w, const void* grad_out,
                              int B, int S, int M, int D, int L, int Lq, int P, float* stage, unsigned char* cell_flags,
                              void* grad_value, float* grad_loc, float* grad_attn_w, void* stream) {
    if (pixel_stride < (long)M * D || pixel_stride % 2 || pixel_stride > 0x3fffffffL) return GRIT_ERR_BAD_ARG;
    if (!value || !spatial_shapes || !level_start || !loc || !attn_w || !grad_out || !grad_value || !grad_loc || !grad_attn_w ||
        !stage || !cell_flags)
        return GRIT_ERR_BAD_ARG;
    if (!dims_ok(B, S, M, D, L, Lq, P)) return GRIT_ERR_BAD_ARG;
    if (D != 64 || L * P > 16 || ((uintptr_t)value % 4) || ((uintptr_t)grad_out % 4) || ((uintptr_t)grad_value % 2) ||
        ((uintptr_t)stage % 4))
        return GRIT_ERR_UNSUPPORTED;
    const int nrows = B * Lq * M;
    const int nblk = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    static const bool interleave = !(getenv("GRIT_MSDA_BWD_INTERLEAVE") && atoi(getenv("GRIT_MSDA_BWD_INTERLEAVE")) == 0);
    const int images = (interleave && B > 1 && (Lq * M) % kRowsPerBlock == 0) ? B : 1;
    static const bool no_merge = getenv("GRIT_MSDA_BWD_MERGE") && atoi(getenv("GRIT_MSDA_BWD_MERGE")) == 0;
    hipLaunchKernelGGL(msda_bwd_d64_pk<1>, dim3(nblk), dim3(kWave * kRowsPerBlock), 0, (hipStream_t)stream,
                       (const __hip_bfloat16*)value, spatial_shapes, level_start, loc, attn_w, (const __hip_bfloat16*)grad_out,
                       S, M, L, Lq, P, (__hip_bfloat16*)nullptr, grad_loc, grad_attn_w, nrows, nblk, images, no_merge ? 1 : 0,
                       (int)pixel_stride, stage, cell_flags);
    if (hipGetLastError() != hipSuccess) return GRIT_ERR_LAUNCH;
    const long ncells = (long)B * S * M;
    const long nwaves = (ncells + 63) / 64;
    hipLaunchKernelGGL(msda_stage_flush, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, stage, cell_flags,
                       (__hip_bfloat16*)grad_value, ncells, S, M, (int)pixel_stride);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_msda_bwd_sorted_supported(int B, int S, int M, int L, int Lq, int P) {
    if (B <= 0 || S <= 0 || M <= 0 || L <= 0 || Lq <= 0 || P <= 0) return GRIT_ERR_BAD_ARG;
    if ((long)Lq * L * P > (long)kValueThreads * kValuePts || Lq > 4095) return GRIT_ERR_UNSUPPORTED;
    if (value_lds_bytes(S, L, Lq, P) > 160 * 1024) return GRIT_ERR_UNSUPPORTED;
    return GRIT_OK;
}

int grit_msda_bwd_bf16_sorted(const void* value, long pixel_stride, const int64_t* spatial_shapes,
                              const int64_t* level_start, const float* loc, const float* attn_w, const void* grad_out,
                              int B, int S, int M, int D, int L, int Lq, int P, void* grad_value, float* grad_loc,
                              float* grad_attn_w, void* stream) {
    if (pixel_stride < (long)M * D || pixel_stride % 8 || pixel_stride > 0x3fffffffL) return GRIT_ERR_BAD_ARG;
    if (!value || !spatial_shapes || !level_start || !loc || !attn_w || !grad_out || !grad_value || !grad_loc || !grad_attn_w)
        return GRIT_ERR_BAD_ARG;
    if (!dims_ok(B, S, M, D, L, Lq, P)) return GRIT_ERR_BAD_ARG;
    const int ok = grit_msda_bwd_sorted_supported(B, S, M, L, Lq, P);
    if (ok != GRIT_OK) return ok;
    if (D != 64 || L * P > 16 || ((uintptr_t)value % 4) || ((uintptr_t)grad_out % 16) || ((uintptr_t)grad_value % 16))
        return GRIT_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    static grit_detail::PerDevice<bool> lds_attr_set_pd; bool& lds_attr_set = lds_attr_set_pd();  // idempotent attribute
    if (!lds_attr_set) {
        if (hipFuncSetAttribute((const void*)msda_bwd_value, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return GRIT_ERR_LAUNCH;
        lds_attr_set = true;
    }
    // fewer (image, head) segments than CUs: several workgroups per segment (GRIT_MSDA_VALUE_PARTS overrides: A/B)
    static const int parts_env = getenv("GRIT_MSDA_VALUE_PARTS") ? atoi(getenv("GRIT_MSDA_VALUE_PARTS")) : 0;
    int parts = parts_env > 0 ? parts_env : 256 / (B * M);
    parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
    hipLaunchKernelGGL(msda_bwd_value, dim3(B * M * parts), dim3(kValueThreads), value_lds_bytes(S, L, Lq, P), st, spatial_shapes,
                       level_start, loc, attn_w, (const __hip_bfloat16*)grad_out, (__hip_bfloat16*)grad_value, S, M, L, Lq, P,
                       (int)pixel_stride, parts);
    if (hipGetLastError() != hipSuccess) return GRIT_ERR_LAUNCH;
    const int nrows = B * Lq * M;
    static const bool rows1 = getenv("GRIT_MSDA_BWD_ROWS1") && atoi(getenv("GRIT_MSDA_BWD_ROWS1")) != 0;  // A/B: one row per wave
    if (!rows1 && (long)B * S * pixel_stride * 2 < (1LL << 32) && (uintptr_t)value % 16 == 0) {
        const int nblk16 = (nrows + 15) / 16;
        hipLaunchKernelGGL(msda_bwd_rows4<2>, dim3(nblk16), dim3(256), 0, st, (const __hip_bfloat16*)value, spatial_shapes,
                           level_start, loc, attn_w, (const __hip_bfloat16*)grad_out, S, M, L, Lq, P, grad_loc, grad_attn_w,
                           nrows, nblk16, (int)pixel_stride);
        return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
    }
    const int nblk = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    hipLaunchKernelGGL(msda_bwd_d64_pk<2>, dim3(nblk), dim3(kWave * kRowsPerBlock), 0, st, (const __hip_bfloat16*)value,
                       spatial_shapes, level_start, loc, attn_w, (const __hip_bfloat16*)grad_out, S, M, L, Lq, P,
                       (__hip_bfloat16*)nullptr, grad_loc, grad_attn_w, nrows, nblk, 1, 1, (int)pixel_stride, (float*)nullptr,
                       (unsigned char*)nullptr);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // extern "C"
