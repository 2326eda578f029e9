// Fused Swin (shifted-)window attention for gfx950 (MI355X): window 12x12 (N = 144 tokens), head_dim 32.
//
// Replaces, per Swin block, what davidnvq/grit models/common/swin_model.py does with ~12 full-map ops and a
// materialised [B_, nH, 144, 144] score tensor:
//   SwinTransformerBlock.forward :257-293  pad to a multiple of 12, roll(-shift), window_partition,
//                                          window_reverse, roll(+shift), crop
//   BasicLayer.forward :424-441            shift mask (0 / -100 between the 9 regions of the rolled map)
//   WindowAttention.forward :161-183       q*scale @ k^T + relative-position bias + mask, softmax, @ v
// The kernels read q/k/v where the pointwise qkv Linear left them ([B, H*W, 3C], token order) and write the
// result back in token order; padding / roll / partition / reverse / crop are address arithmetic, padded
// tokens are synthesised from `pad_qkv` (= the Linear bias: the reference pads zeros *after* norm1), the shift
// mask is computed from region ids, and nothing N x N ever reaches HBM.
//
// CDNA4 mapping (64-lane waves, MFMA 16x16x32 bf16, fp32 softmax):
//   * workgroup = 9 waves = one (window, head) at a time; persistent over windows of ONE head, so the
//     head's relative-position bias lives in registers (36 floats / lane) for the whole launch;
//   * forward, wave w owns query tile w (16 queries x 144 keys).  S^T = K Q^T is computed with the KEY on the
//     MFMA row: the accumulator then holds, per lane, one query (lane & 15) and 36 of its keys, so the softmax
//     row reductions are 36 in-register ops + 2 cross-lane shuffles, and the bf16-packed accumulator is
//     directly the B operand of O^T = V^T P^T (k-slot order chosen to match; V^T fragments come from the
//     row-major V tile in LDS through ds_read_b64_tr_b16) -- no LDS round trip for P;
//   * backward, phase 1 wave w owns KEY tile w: recomputes P from the saved row log-sum-exp, forms dP, dS,
//     accumulates d(bias) in registers across the windows of the launch, and gets dV^T / dK^T from MFMAs whose
//     B operand is again the packed accumulator; dS goes once through LDS (transposed) so that phase 2, wave w
//     = query tile w, produces dQ^T.  dq/dk/dv of a token are written exactly once (every token belongs to one
//     window): no global atomics except the per-launch flush of d(bias) and of the padded-token gradient.
#include <hip/hip_runtime.h>
#include "per_device.h"
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grit_hip.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// Packed fp32 (two elements per VALU instruction).  hipcc forms v_pk_* from vector expressions only when it feels like
// it (and unpacks them again next to MFMAs), so the *_asm variants spell the instruction out.  They must never read an
// MFMA result directly: the compiler's hazard recogniser does not look inside inline asm, and MFMA -> VALU needs
// software wait states (so does v_exp -> VALU) -- the first consumer of every MFMA accumulator and of every v_exp result
// below is a plain C++ expression.
__device__ __forceinline__ v2f pk_add_asm(v2f a, v2f b) {
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2f pk_sub_asm(v2f a, v2f b) {  // a - b: negate both halves of the second operand
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
typedef __attribute__((address_space(3))) v4s lds_v4s;

#ifdef GRIT_NT_WINATTN
#define GRIT_ST4(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<v4bf*>(ptr))
#else
#define GRIT_ST4(ptr, val) (*reinterpret_cast<v4bf*>(ptr) = (val))
#endif
constexpr int kWs = 12, kN = 144, kHd = 32, kTiles = 9, kThreads = 576;
constexpr int kKP = 40;      // pitch (bf16 elements) of row-read tiles: 80 B, spreads ds_read_b128 over banks
constexpr int kVP = 32;      // pitch of tiles read through ds_read_b64_tr_b16 (64 B rows)
constexpr int kRows = 160;   // tiles are zero-padded to 5 k-steps of 32
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct Geom {
    int B, H, W, C, nH, shift, nWh, nWw, Hp, Wp, T, nWm;
    float scale;
    int xcd_pairs;  // 1: workgroup id -> (head, window group) keeps heads 2j / 2j+1 on one XCD (head_and_group)
    float inv_img, inv_nww;  // 1 / (windows per image), 1 / nWw: window_of
};

// window id -> (image, window row, window column).  Once per (window, head) and wave, on the issue-bound path of both MFMA
// kernels: two 32-bit integer divisions by launch constants are ~50 instructions, the float form is 8 (same-box A/B,
// profiles/r03/winattn_window_index.txt: forward -2..3 %, backward -0.5..1 %, nothing visible in the step).  Exact while
// win < 2^21 (checked on the host): (win + 0.5) / n is at least 0.5 / n from an integer, the float error stays below that.
__device__ __forceinline__ void window_of(int win, const Geom& g, int& b, int& wy, int& wx) {
    b = (int)(((float)win + 0.5f) * g.inv_img);
    const int wrem = win - b * (g.nWh * g.nWw);
    wy = (int)(((float)wrem + 0.5f) * g.inv_nww);
    wx = wrem - wy * g.nWw;
}

// Workgroups are dealt to the 8 XCDs round-robin by id and every XCD has its own L2.  A head's q / k / v slice of a token
// is 64 bytes, so heads 2j and 2j+1 share each 128-byte line: with head = id % nH the two always sit on different XCDs
// and every line of qkv / out / dout is filled into two L2s.  This mapping gives each XCD whole pairs of heads (for 4 and
// 8 heads, two XCDs share a pair and split the window groups).
__device__ __forceinline__ void head_and_group(const Geom& g, int& h, int& grp) {
    const int id = blockIdx.x;
    if (!g.xcd_pairs) { h = id % g.nH; grp = id / g.nH; return; }
    const int x = id & 7, r = id >> 3;
    if (g.nH >= 16) { const int hp = g.nH >> 3; h = x * hp + r % hp; grp = r / hp; }
    else if (g.nH == 8) { h = (x & ~1) + (r & 1); grp = (r >> 1) * 2 + (x & 1); }
    else { h = 2 * ((x >> 1) & 1) + (r & 1); grp = (r >> 1) * 4 + (x & 1) + 2 * (x >> 2); }  // 4 heads
}

// window-local index n of window (wy, wx) -> token index in the H x W map (or -1 for a padding token) and the
// shift-mask region of the position (swin_model.py:424-436 on the rolled map)
__device__ __forceinline__ int token_of(int n, int wy, int wx, const Geom& g, int& region) {
    const int i = n / kWs, j = n - kWs * i;
    const int ys = wy * kWs + i, xs = wx * kWs + j;
    const int rh = ys < g.Hp - kWs ? 0 : (ys < g.Hp - g.shift ? 1 : 2);
    const int rw = xs < g.Wp - kWs ? 0 : (xs < g.Wp - g.shift ? 1 : 2);
    region = 3 * rh + rw;
    int y = ys + g.shift, x = xs + g.shift;
    if (y >= g.Hp) y -= g.Hp;
    if (x >= g.Wp) x -= g.Wp;
    return (y < g.H && x < g.W) ? y * g.W + x : -1;
}

__device__ __forceinline__ uint4 load16(const void* p) { return *reinterpret_cast<const uint4*>(p); }

__device__ __forceinline__ v8bf as_v8bf(uint4 u) { return __builtin_bit_cast(v8bf, u); }

__device__ __forceinline__ v8bf tr_pair(const __bf16* lo, const __bf16* hi) {
    // two transposing reads (4 rows x 16 columns each) -> the 8 k-slots of one A/B fragment
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

__device__ __forceinline__ float xor_max(float v, int o) { return fmaxf(v, __shfl_xor(v, o, 64)); }
// (v_permlane16_swap / v_permlane32_swap instead of the two ds_bpermute steps of the forward's row max / row sum: -1 % at best; not kept)
// sum over the 4 lanes of a quad, as DPP quad permutes
__device__ __forceinline__ float sum_quad(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
    return v;
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads)
void winattn_fwd(const __bf16* __restrict__ qkv, const float* __restrict__ rel_bias, const __bf16* __restrict__ pad_qkv,
                 const float* __restrict__ mask, Geom g, __bf16* __restrict__ out, float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[kN * kKP];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[kRows * kVP];
    __shared__ __attribute__((aligned(16))) uint8_t rid[kRows];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    int h, grp;
    head_and_group(g, h, grp);
    const int ngrp = gridDim.x / g.nH;
    const int NW = g.B * g.nWh * g.nWw;
    const int C3 = 3 * g.C;
    const int hoff = h * kHd;
    const float c2 = g.scale * kLog2e;

    // zero the k-padding rows of V once (rows 144..159 are never written by the window loads)
    for (int i = tid; i < (kRows - kN) * kVP; i += kThreads) Vs[kN * kVP + i] = (__bf16)0.f;
    if (tid < kRows - kN) rid[kN + tid] = 0;

    // this wave's slice of the head's relative-position bias, pre-multiplied by log2(e):
    // b2[kt][r] = bias[h][query 16w + l15][key 16kt + 4lg + r]
    v4f b2[kTiles];
    {
        const float* brow = rel_bias + ((size_t)h * kN + 16 * w + l15) * kN + 4 * lg;
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {
            const float4 t = *reinterpret_cast<const float4*>(brow + 16 * kt);
            b2[kt] = v4f{t.x * kLog2e, t.y * kLog2e, t.z * kLog2e, t.w * kLog2e};
        }
    }

    const int sn = tid >> 2, sc = tid & 3;  // staging role: token sn, 16-byte chunk sc
    // Software pipeline over windows: the global loads of window i+1 (one K chunk, one V chunk, this lane's Q
    // fragment) are issued before the math of window i, so their L2 / HBM latency hides under ~3k cycles of MFMA +
    // softmax instead of stalling the single resident workgroup of the CU at the top of every iteration.
    struct Fetch { uint4 kq, vq, qf; int reg, tq, qreg, wy, wx; size_t img; };
    auto fetch = [&](int win) {
        Fetch f;
        int b;
        window_of(win, g, b, f.wy, f.wx);
        f.img = (size_t)b * g.T;
        const int tk = token_of(sn, f.wy, f.wx, g, f.reg);
        const __bf16* src = tk >= 0 ? qkv + (f.img + tk) * C3 + hoff + sc * 8 : pad_qkv + hoff + sc * 8;
        f.kq = load16(src + g.C);
        f.vq = load16(src + 2 * g.C);
        f.tq = token_of(16 * w + l15, f.wy, f.wx, g, f.qreg);
        const __bf16* qsrc = f.tq >= 0 ? qkv + (f.img + f.tq) * C3 + hoff + lg * 8 : pad_qkv + hoff + lg * 8;
        f.qf = load16(qsrc);
        return f;
    };
    Fetch nxt;
    if (grp < NW) nxt = fetch(grp);
    for (int win = grp; win < NW; win += ngrp) {
        const Fetch cur = nxt;
        const int wy = cur.wy, wx = cur.wx, tq = cur.tq, qreg = cur.qreg;
        const size_t img = cur.img;
        const v8bf qf = as_v8bf(cur.qf);
        __syncthreads();  // previous window's LDS reads are done
        *reinterpret_cast<uint4*>(&Ks[sn * kKP + sc * 8]) = cur.kq;
        *reinterpret_cast<uint4*>(&Vs[sn * kVP + sc * 8]) = cur.vq;
        if (sc == 0) rid[sn] = (uint8_t)cur.reg;
        __syncthreads();
        if (win + ngrp < NW) nxt = fetch(win + ngrp);  // in flight during this window's compute

        // ---- S^T = K Q^T : acc[kt][r] = <q(16w + l15), k(16kt + 4lg + r)>
        v4f acc[kTiles];
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {
            const v8bf kf = as_v8bf(*reinterpret_cast<const uint4*>(&Ks[(16 * kt + l15) * kKP + lg * 8]));
            acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // ---- logits in the log2 domain: t = s*scale*log2e + bias*log2e (+ mask*log2e)
        const bool analytic = (mask == nullptr) && g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {  // two logits per instruction (v_pk_fma_f32 on the halves of the MFMA result quad)
            const v2f lo = __builtin_elementwise_fma(v2f{acc[kt][0], acc[kt][1]}, v2f{c2, c2}, v2f{b2[kt][0], b2[kt][1]});
            const v2f hi = __builtin_elementwise_fma(v2f{acc[kt][2], acc[kt][3]}, v2f{c2, c2}, v2f{b2[kt][2], b2[kt][3]});
            acc[kt] = v4f{lo[0], lo[1], hi[0], hi[1]};
        }
        if (analytic) {
#pragma unroll
            for (int kt = 0; kt < kTiles; ++kt) {
                const uint32_t ids = *reinterpret_cast<const uint32_t*>(&rid[16 * kt + 4 * lg]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((int)((ids >> (8 * r)) & 0xff) != qreg) acc[kt][r] += -100.0f * kLog2e;
            }
        } else if (mask != nullptr) {
            const float* mrow = mask + ((size_t)(win % g.nWm) * kN + 16 * w + l15) * kN + 4 * lg;
#pragma unroll
            for (int kt = 0; kt < kTiles; ++kt) {
                const float4 t = *reinterpret_cast<const float4*>(mrow + 16 * kt);
                acc[kt][0] = fmaf(t.x, kLog2e, acc[kt][0]);
                acc[kt][1] = fmaf(t.y, kLog2e, acc[kt][1]);
                acc[kt][2] = fmaf(t.z, kLog2e, acc[kt][2]);
                acc[kt][3] = fmaf(t.w, kLog2e, acc[kt][3]);
            }
        }
        // ---- softmax over the 144 keys of this lane's query: 36 registers x 4 lane groups
        float m = acc[0][0];
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[kt][r]);
        m = xor_max(xor_max(m, 16), 32);
        v2f sum2 = {0.f, 0.f};
        const v2f m2 = {m, m};
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {
            const v2f d_lo = v2f{acc[kt][0], acc[kt][1]} - m2, d_hi = v2f{acc[kt][2], acc[kt][3]} - m2;
            const v2f e_lo = {__builtin_amdgcn_exp2f(d_lo[0]), __builtin_amdgcn_exp2f(d_lo[1])};
            const v2f e_hi = {__builtin_amdgcn_exp2f(d_hi[0]), __builtin_amdgcn_exp2f(d_hi[1])};
            acc[kt] = v4f{e_lo[0], e_lo[1], e_hi[0], e_hi[1]};
            sum2 += e_lo + e_hi;
        }
        float sum = sum2[0] + sum2[1];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // ---- O^T = V^T P^T; k-slot (lg, j<4) = key 32s + 4lg + j, (lg, j>=4) = key 32s + 16 + 4lg + (j-4)
        v4f o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
        const int trq = l15 >> 2, trp = l15 & 3;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const v8bf pf = s < 4 ? pack8(acc[2 * s], acc[2 * s + 1]) : pack8(acc[8], v4f{0.f, 0.f, 0.f, 0.f});
            const __bf16* lo = &Vs[(32 * s + 4 * lg + trq) * kVP + 4 * trp];
            const __bf16* hi = lo + 16 * kVP;
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(lo, hi), pf, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(lo + 16, hi + 16), pf, o1, 0, 0, 0);
        }
        // o0[r] = O[query 16w + l15][d = 4lg + r], o1[r] = ... d = 16 + 4lg + r
        if (tq >= 0) {
            __bf16* orow = out + (img + tq) * g.C + hoff + 4 * lg;
            v4bf a, c;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (__bf16)(o0[r] * inv); c[r] = (__bf16)(o1[r] * inv); }
            GRIT_ST4(orow, a);
            GRIT_ST4(orow + 16, c);
        }
        if (lg == 0) lse2[((size_t)win * g.nH + h) * kN + 16 * w + l15] = m + __builtin_amdgcn_logf(sum);
    }
}

// DMA-staged forward (the default; GRIT_WINATTN_FWD_DMA=0 or an explicit mask select the register-staged kernel above).  Same
// arithmetic; what changes is who moves the operands and when (s_memtime stamps, profiles/r03/winattn_bwd_stamps.txt: in the
// register-staged kernel the third wave of SIMD 0 needs 2x the first waves' time for its fetch and everybody waits a third of the
// kernel for it):
//   * the Q / K / V tiles of the NEXT window travel global -> LDS by global_load_lds (double-buffered tiles, no staging registers,
//     no LDS writes, ONE barrier per window instead of two), issued as asm by the six waves that do not sit on SIMD 0 -- loader li
//     moves token rows 16 li .. + 15 of the three tiles, rows 96 .. 143 are shared by loader pairs; they also leave the token index
//     and the shift-mask region of every window position in LDS, so no other wave runs token_of;
//   * the 64-byte rows of a DMA image cannot be padded: the 16-byte chunks are permuted on the SOURCE address instead
//     (Q / K, read with ds_read_b128 by 16 rows x one chunk: chunk ^ 3 in rows 8 .. 15 of a 16-row block; V, read with
//     ds_read_b64_tr_b16 by 8 rows x 32 bytes: chunk ^ 2 in rows 4 .. 7 of an 8-row block -- conflict-free by the lane groups of
//     MI355X_MICROARCH.md "LDS").
constexpr unsigned kFTile = kN * 64, kFVTile = kRows * 64;
constexpr unsigned kFOffQ = 0, kFOffK = 2 * kFTile, kFOffV = 4 * kFTile, kFOffTok = kFOffV + 2 * kFVTile, kFOffRid = kFOffTok + 2 * kN * 4;
constexpr size_t kFwdDmaLds = kFOffRid + 2 * kN + 64;  // (+ the kept-image table of the drop-path variant)

// kSkip (round 5): drop path, training forward.  The branch this attention belongs to is multiplied by row_scale[b] afterwards: the
// windows of images with factor 0 are not computed -- out receives zeros for their tokens (finite: 0 * x stays 0 in the residual add),
// their log-sum-exps zeros (the backward that gets the same factors never reads them) -- and the workgroups share the windows of the
// kept images through the live-index map of winattn_bwd_dma<.., true>.  Same window loop, same single exit.
template <bool kSkip = false>
__global__ __launch_bounds__(kThreads)
void winattn_fwd_dma(const __bf16* __restrict__ qkv, const float* __restrict__ rel_bias, const __bf16* __restrict__ pad_qkv, Geom g,
                     __bf16* __restrict__ out, float* __restrict__ lse2, const float* __restrict__ row_scale = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* tok_s = reinterpret_cast<int*>(smem_raw + kFOffTok);       // [2][144]
    uint8_t* rid = smem_raw + kFOffRid;                              // [2][144]
    uint8_t* kept_s = rid + 2 * kN;                                  // [64] kSkip: j-th kept image

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    int h, grp;
    head_and_group(g, h, grp);
    const int ngrp = gridDim.x / g.nH;
    int NW = g.B * g.nWh * g.nWw;  // (kSkip: the windows of the kept images)
    const int C3 = 3 * g.C;
    const int hoff = h * kHd;
    const float c2 = g.scale * kLog2e;
    if constexpr (kSkip) {
        unsigned long long keep = 0ull;  // wave-uniform: bit b = image b is kept; B <= 64 (host)
        for (int b = 0; b < g.B; ++b) keep |= (unsigned long long)(row_scale[b] != 0.f) << b;
        if (tid < g.B && ((keep >> tid) & 1ull)) kept_s[__builtin_popcountll(keep & ((1ull << tid) - 1ull))] = (uint8_t)tid;
        NW = __builtin_popcountll(keep) * g.nWh * g.nWw;
        const int wpi = g.nWh * g.nWw, chunk = tid & 3;  // out: this head's 64-byte slice of a token = 4 threads; 576 = 144 x 4
        for (unsigned long long gone = ~keep & (g.B == 64 ? ~0ull : (1ull << g.B) - 1ull); gone; gone &= gone - 1ull) {
            const int b = __builtin_ctzll(gone);
            for (int t = grp * 144 + (tid >> 2); t < g.T; t += ngrp * 144)
                *reinterpret_cast<uint4*>(out + ((size_t)b * g.T + t) * g.C + hoff + chunk * 8) = make_uint4(0, 0, 0, 0);
            for (int i = grp * kThreads + tid; i < wpi * kN; i += ngrp * kThreads)
                lse2[(((size_t)b * wpi + i / kN) * g.nH + h) * kN + i % kN] = 0.f;
        }
        __syncthreads();  // the table is read by the first prefetch
    }
    auto window_id = [&](int lv) {  // live index -> window id (identity without kSkip)
        if constexpr (kSkip) {
            const int wpi = g.nWh * g.nWw;
            const int q = (int)(((float)lv + 0.5f) * g.inv_img);
            // (readfirstlane: the table entry is the same in every lane -- as a vector value it would turn the scalar address arithmetic of
            // the window's transfers into vector instructions on the issue-bound SIMDs)
            return __builtin_amdgcn_readfirstlane((int)kept_s[q]) * wpi + (lv - q * wpi);
        } else {
            return lv;
        }
    };

    // rows 144 .. 159 of both V buffers: the k-padding of the last PV step, never written by the transfers
    for (int i = tid; i < 2 * (kRows - kN) * 16; i += kThreads) {
        const int b = i / ((kRows - kN) * 16), r = i - b * (kRows - kN) * 16;
        reinterpret_cast<uint32_t*>(smem_raw + kFOffV + b * kFVTile + kN * 64)[r] = 0u;
    }

    v4f b2[kTiles];  // b2[kt][r] = bias[h][query 16w + l15][key 16kt + 4lg + r] * log2(e)
    {
        const float* brow = rel_bias + ((size_t)h * kN + 16 * w + l15) * kN + 4 * lg;
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {
            const float4 t = *reinterpret_cast<const float4*>(brow + 16 * kt);
            b2[kt] = v4f{t.x * kLog2e, t.y * kLog2e, t.z * kLog2e, t.w * kLog2e};
        }
    }
    // the bias loads are the only compiler-visible loads of the kernel: consumed here, so that hipcc's wait for them sits in front
    // of the window loop and not -- as `vmcnt(0)`, i.e. a wait for the transfers -- in front of their first use inside it
#pragma unroll
    for (int kt = 0; kt < kTiles; ++kt) asm volatile("" : "+v"(b2[kt]));

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem_raw;
    auto dma16 = [&](const __bf16* src, unsigned tile_off, int group) {  // rows 16 group .. + 15 of a tile with 64-byte rows
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + tile_off + group * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory", "m0");
    };
    const int li = (w & 3) == 0 ? -1 : (w < 4 ? w - 1 : w - 2);
    auto fetch_rows = [&](int group, int wy, int wx, size_t img, int buf, bool qk_part, bool v_part) {
        int reg;
        int row = 16 * group + (lane >> 2), p = lane & 3;
        asm volatile("" : "+v"(row), "+v"(p));  // per-window values (hoisted they cost registers)
        const int tk = token_of(row, wy, wx, g, reg);
        const __bf16* src = tk >= 0 ? qkv + (img + tk) * C3 + hoff : pad_qkv + hoff;
        if (qk_part) {
            const int c = p ^ ((row & 8) ? 3 : 0);
            dma16(src + c * 8, kFOffQ + buf * kFTile, group);
            dma16(src + g.C + c * 8, kFOffK + buf * kFTile, group);
            if (p == 0) { rid[buf * kN + row] = (uint8_t)reg; tok_s[buf * kN + row] = tk; }
        }
        if (v_part) {
            const int c = p ^ ((row & 4) ? 2 : 0);
            dma16(src + 2 * g.C + c * 8, kFOffV + buf * kFVTile, group);
        }
    };
    struct Next { int wy, wx, win; size_t img; };
    auto prefetch = [&](int win, int buf) {
        Next f;
        int b;
        window_of(win, g, b, f.wy, f.wx);
        f.win = win;
        f.img = (size_t)b * g.T;
        if (li >= 0) {  // wave-uniform
            fetch_rows(li, f.wy, f.wx, f.img, buf, true, true);
            fetch_rows(6 + (li >> 1), f.wy, f.wx, f.img, buf, !(li & 1), (li & 1) != 0);
        }
        return f;
    };

    const int sk = (l15 & 8) ? 3 : 0;
    const int trq = l15 >> 2, trp = l15 & 3;
    Next nxt;
    int cur = 0;
    if (grp < NW) nxt = prefetch(window_id(grp), 0);
    for (int lv = grp; lv < NW; lv += ngrp, cur ^= 1) {
        const int wy = nxt.wy, wx = nxt.wx, win = kSkip ? nxt.win : lv;
        const size_t img = nxt.img;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's transfers (and its stores of the previous window)
        __syncthreads();  // everybody's: the tiles of this window are complete, and nobody reads the other buffers any more
        const __bf16* Qs = reinterpret_cast<const __bf16*>(smem_raw + kFOffQ + cur * kFTile);
        const __bf16* Ks = reinterpret_cast<const __bf16*>(smem_raw + kFOffK + cur * kFTile);
        const __bf16* Vs = reinterpret_cast<const __bf16*>(smem_raw + kFOffV + cur * kFVTile);
        int oFrag = l15 * 32 + ((lg ^ sk) << 3);                         // row read of a 16-row block (+ 16 block * 32)
        int oTrV = (4 * lg + trq) * 32 + ((((trp >> 1) ^ ((lg & 1) << 1))) << 3) + ((trp & 1) << 2);  // transposing reads (+ 32 s * 32)
        int oId = cur * kN;
        asm volatile("" : "+v"(oFrag), "+v"(oTrV), "+v"(oId));
        const int tq = tok_s[oId + 16 * w + l15], qreg = rid[oId + 16 * w + l15];
        const v8bf qf = as_v8bf(*reinterpret_cast<const uint4*>(&Qs[oFrag + 16 * w * 32]));
        if (lv + ngrp < NW) nxt = prefetch(window_id(lv + ngrp), cur ^ 1);  // lands under this window's compute

        // ---- S^T = K Q^T : acc[kt][r] = <q(16w + l15), k(16kt + 4lg + r)>
        v4f acc[kTiles];
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {
            const v8bf kf = as_v8bf(*reinterpret_cast<const uint4*>(&Ks[oFrag + 16 * kt * 32]));
            acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        const bool analytic = g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {
            const v2f lo = __builtin_elementwise_fma(v2f{acc[kt][0], acc[kt][1]}, v2f{c2, c2}, v2f{b2[kt][0], b2[kt][1]});
            const v2f hi = __builtin_elementwise_fma(v2f{acc[kt][2], acc[kt][3]}, v2f{c2, c2}, v2f{b2[kt][2], b2[kt][3]});
            acc[kt] = v4f{lo[0], lo[1], hi[0], hi[1]};
        }
        if (analytic) {
#pragma unroll
            for (int kt = 0; kt < kTiles; ++kt) {
                const uint32_t ids = *reinterpret_cast<const uint32_t*>(&rid[oId + 16 * kt + 4 * lg]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((int)((ids >> (8 * r)) & 0xff) != qreg) acc[kt][r] += -100.0f * kLog2e;
            }
        }
        float m = acc[0][0];
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[kt][r]);
        m = xor_max(xor_max(m, 16), 32);
        v2f sum2 = {0.f, 0.f};
        const v2f m2 = {m, m};
#pragma unroll
        for (int kt = 0; kt < kTiles; ++kt) {
            const v2f d_lo = v2f{acc[kt][0], acc[kt][1]} - m2, d_hi = v2f{acc[kt][2], acc[kt][3]} - m2;
            const v2f e_lo = {__builtin_amdgcn_exp2f(d_lo[0]), __builtin_amdgcn_exp2f(d_lo[1])};
            const v2f e_hi = {__builtin_amdgcn_exp2f(d_hi[0]), __builtin_amdgcn_exp2f(d_hi[1])};
            acc[kt] = v4f{e_lo[0], e_lo[1], e_hi[0], e_hi[1]};
            sum2 += e_lo + e_hi;
        }
        float sum = sum2[0] + sum2[1];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // ---- O^T = V^T P^T; k-slot (lg, j<4) = key 32s + 4lg + j, (lg, j>=4) = key 32s + 16 + 4lg + (j-4)
        v4f o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const v8bf pf = s < 4 ? pack8(acc[2 * s], acc[2 * s + 1]) : pack8(acc[8], v4f{0.f, 0.f, 0.f, 0.f});
            const __bf16* lo = &Vs[oTrV + 32 * s * 32];
            const __bf16* hi = lo + 16 * 32;
            // channels 16 .. 31 are logical chunks 2, 3: position chunk ^ 2 relative to chunks 0, 1
            o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(lo, hi), pf, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_pair(lo + 16 - ((lg & 1) << 5), hi + 16 - ((lg & 1) << 5)), pf, o1, 0, 0, 0);
        }
        if (tq >= 0) {
            __bf16* orow = out + (img + tq) * g.C + hoff + 4 * lg;
            v4bf a, c;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (__bf16)(o0[r] * inv); c[r] = (__bf16)(o1[r] * inv); }
            GRIT_ST4(orow, a);
            GRIT_ST4(orow + 16, c);
        }
        if (lg == 0) lse2[((size_t)win * g.nH + h) * kN + 16 * w + l15] = m + __builtin_amdgcn_logf(sum);
    }
}

// ---------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------
// LDS plan (159 KB, one workgroup per CU): the head's bias slab in fp32, TRANSPOSED [key][query] with a 148-float
// pitch (conflict-free 16-byte reads, shared by all windows of the launch), Q / dO / K tiles [144][32] bf16, the
// transposed dS tile [key][query] bf16 (pitch 148) and the per-row statistics.  Registers hold the d(bias)
// accumulators (36 floats / lane) across the whole launch; everything else is transient, so nothing spills.
constexpr int kTP = 32;    // tile pitch (bf16) of Q / dO / K: 64-byte rows, 144 rows (k-padding handled by selects)
constexpr int kSP = 148;   // pitch (bf16) of the transposed dS tile
constexpr int kBP = 148;   // pitch (float) of the transposed bias slab

// sum over the 16 lanes that share (lane >> 4): the 16 keys / queries of a tile for fixed (channel group, r)
// (DPP row rotations: a row IS those 16 lanes.  As __shfl_xor each step is a ds_bpermute round trip through the LDS pipe plus a
// full lgkmcnt wait -- 24 sums per wave and window wherever a tile holds window-padding tokens, i.e. in 7 of the 16 windows of a
// 40 x 40 map: 11-13 % of the backward kernel there.)
__device__ __forceinline__ float sum16(float v) {
#define GRIT_ROW_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, false))
    v += GRIT_ROW_ROR(v, 8); v += GRIT_ROW_ROR(v, 4); v += GRIT_ROW_ROR(v, 2); v += GRIT_ROW_ROR(v, 1);
#undef GRIT_ROW_ROR
    return v;
}

__device__ __forceinline__ v4s tr_read(const __bf16* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)p); }

__device__ __forceinline__ v8bf join(v4s a, v4s b) {
    const v8s r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(v8bf, r);
}

template <bool kExplicitMask>
__global__ __launch_bounds__(kThreads)
void winattn_bwd(const __bf16* __restrict__ qkv, const float* __restrict__ rel_bias, const __bf16* __restrict__ pad_qkv,
                 const float* __restrict__ mask, Geom g, const __bf16* __restrict__ out, const __bf16* __restrict__ dout,
                 const float* __restrict__ lse2, __bf16* __restrict__ dqkv, float* __restrict__ dbias,
                 float* __restrict__ dpad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* bT = reinterpret_cast<float*>(smem_raw);                         // [144][kBP]  bias^T * log2e
    __bf16* Qs = reinterpret_cast<__bf16*>(bT + kN * kBP);                  // [144][kTP]
    __bf16* dOs = Qs + kN * kTP;
    __bf16* Ks = dOs + kN * kTP;
    __bf16* dSt = Ks + kN * kTP;                                            // [144][kSP]  [key][query]
    float* lse_s = reinterpret_cast<float*>(dSt + kN * kSP);                // [144]
    float* delta_s = lse_s + kN;                                            // [144]
    float* pad_s = delta_s + kN;                                            // [96]
    uint8_t* rid = reinterpret_cast<uint8_t*>(pad_s + 3 * kHd);             // [144]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int trq = l15 >> 2, trp = l15 & 3;
    int h, grp;
    head_and_group(g, h, grp);
    const int ngrp = gridDim.x / g.nH;
    const int NW = g.B * g.nWh * g.nWw;
    const int C3 = 3 * g.C;
    const int hoff = h * kHd;
    const float c2 = g.scale * kLog2e;

    // one-time: bias slab (transposed, log2 domain) and the pad-gradient accumulator
    for (int i = tid; i < kN * kN; i += kThreads) {
        const int qi = i / kN, ki = i - qi * kN;
        bT[ki * kBP + qi] = rel_bias[(size_t)h * kN * kN + i] * kLog2e;
    }
    if (tid < 3 * kHd) pad_s[tid] = 0.f;

    v4f dB[kTiles];  // d(bias)[query 16qt + 4lg + r][key 16w + l15], summed over this workgroup's windows
#pragma unroll
    for (int qt = 0; qt < kTiles; ++qt) dB[qt] = v4f{0.f, 0.f, 0.f, 0.f};

    const v4s z4s = {0, 0, 0, 0};
    const int sn = tid >> 2, sc = tid & 3;
    // Global fetches of a window: the staging chunks (token sn, 16-byte chunk sc of Q / K / dO / O), this lane's K / V
    // fragments (B operands of phase 1) and log-sum-exp.  They are issued one window AHEAD, after phase 1 of the previous
    // window (where the register peak is), so that phase 2 and the gradient stores cover part of their latency.
    struct Fetch { uint4 q_c, k_c, do_c, o_c, kf, vf; float lse_v; int reg, tkk, kreg, wy, wx; size_t img; };
    auto fetch = [&](int win) {
        Fetch f;
        int b;
        window_of(win, g, b, f.wy, f.wx);
        f.img = (size_t)b * g.T;
        const int tk = token_of(sn, f.wy, f.wx, g, f.reg);
        const __bf16* src = tk >= 0 ? qkv + (f.img + tk) * C3 + hoff + sc * 8 : pad_qkv + hoff + sc * 8;
        f.q_c = load16(src);
        f.k_c = load16(src + g.C);
        f.do_c = make_uint4(0, 0, 0, 0); f.o_c = make_uint4(0, 0, 0, 0);
        if (tk >= 0) {
            f.do_c = load16(dout + (f.img + tk) * g.C + hoff + sc * 8);
            f.o_c = load16(out + (f.img + tk) * g.C + hoff + sc * 8);
        }
        f.tkk = token_of(16 * w + l15, f.wy, f.wx, g, f.kreg);  // this lane's key in phase 1 / query in phase 2
        const __bf16* ksrc = f.tkk >= 0 ? qkv + (f.img + f.tkk) * C3 + hoff + lg * 8 : pad_qkv + hoff + lg * 8;
        f.kf = load16(ksrc + g.C);
        f.vf = load16(ksrc + 2 * g.C);
        f.lse_v = 0.f;
        if (tid < kN) f.lse_v = lse2[((size_t)win * g.nH + h) * kN + tid];
        return f;
    };
    Fetch nxt;
    if (grp < NW) nxt = fetch(grp);
    for (int win = grp; win < NW; win += ngrp) {
        const int wy = nxt.wy, wx = nxt.wx, reg = nxt.reg, tkk = nxt.tkk, kreg = nxt.kreg;
        const size_t img = nxt.img;
        const uint4 q_c = nxt.q_c, k_c = nxt.k_c, do_c = nxt.do_c, o_c = nxt.o_c;
        const v8bf kf = as_v8bf(nxt.kf), vf = as_v8bf(nxt.vf);
        const float lse_v = nxt.lse_v;
        float dpart = 0.f;  // delta = rowsum(dO * O): 8 channels per thread, 4 threads per token
        {
            const v8bf a = as_v8bf(do_c), c = as_v8bf(o_c);
#pragma unroll
            for (int e = 0; e < 8; ++e) dpart = fmaf((float)a[e], (float)c[e], dpart);
            dpart = sum_quad(dpart);
        }
        __syncthreads();  // previous window fully consumed (and, first time, the bias slab is complete)
        *reinterpret_cast<uint4*>(&Qs[sn * kTP + sc * 8]) = q_c;
        *reinterpret_cast<uint4*>(&Ks[sn * kTP + sc * 8]) = k_c;
        *reinterpret_cast<uint4*>(&dOs[sn * kTP + sc * 8]) = do_c;
        if (sc == 0) { rid[sn] = (uint8_t)reg; delta_s[sn] = dpart; }
        if (tid < kN) lse_s[tid] = lse_v;
        __syncthreads();

        // Per-lane LDS offsets, made opaque once per window: without this hipcc hoists ~90 loop-invariant LDS
        // addresses out of the window loop, runs out of registers and reloads them from scratch before every read.
        int oRow = l15 * kTP + lg * 8;                   // row reads of Qs / dOs      (+ 16 qt kTP)
        int oTr = (4 * lg + trq) * kTP + 4 * trp;        // transposing reads, phase 1 (+ 32 s kTP)
        int oB = (16 * w + l15) * kBP + 4 * lg;          // bias slab                  (+ 16 qt)
        int oSt = 4 * lg;                                // lse / delta                (+ 16 qt)
        int oW = (16 * w + l15) * kSP + 4 * lg;          // dS^T writes                (+ 16 qt)
        asm volatile("" : "+v"(oRow), "+v"(oTr), "+v"(oB), "+v"(oSt), "+v"(oW));

        // ================= phase 1: wave w = key tile w =================
        // S[q][k] = Q K^T, dP[q][k] = dO V^T on tiles (qt, w); after each PAIR of query tiles (one 32-deep k-step of
        // the products that sum over queries) the packed P / dS feed dV^T += dO^T P and dK^T += Q^T dS at once.
        const bool analytic = !kExplicitMask && g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
        v4f dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;
        v4bf pprev, sprev;
        // software pipeline by one query tile: the S / dP MFMAs of tile qt+1 (and the LDS reads feeding them) are issued
        // before the element-wise work of tile qt, so their latency hides under ~60 VALU instructions instead of parking the
        // wave (PMC: 52 % of the wave cycles were s_waitcnt / barrier waits, 26 % issue stalls on MFMA results)
        auto score_tiles = [&](int qt, v4f& s_out, v4f& dp_out) {
            const v8bf qa = as_v8bf(*reinterpret_cast<const uint4*>(&Qs[oRow + 16 * qt * kTP]));
            const v8bf da = as_v8bf(*reinterpret_cast<const uint4*>(&dOs[oRow + 16 * qt * kTP]));
            s_out = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            dp_out = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        };
        v4f s_cur, dp_cur;
        score_tiles(0, s_cur, dp_cur);
#pragma unroll
        for (int qt = 0; qt < kTiles; ++qt) {
            const v4f s = s_cur, dp = dp_cur;
            if (qt + 1 < kTiles) {  // next tile's MFMAs and LDS reads fly under this tile's element-wise work
                score_tiles(qt + 1, s_cur, dp_cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            // element-wise part, two score elements per instruction: the loop is instruction-issue bound and hipcc does not
            // form packed-fp32 operations from this code by itself, so the fma / sub / mul / add pairs are spelled out
            // (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 on 64-bit register pairs, which the halves of an MFMA result are)
            const v4f bq = *reinterpret_cast<const v4f*>(&bT[oB + 16 * qt]);
            const v4f lq = *reinterpret_cast<const v4f*>(&lse_s[oSt + 16 * qt]);
            const v4f dl = *reinterpret_cast<const v4f*>(&delta_s[oSt + 16 * qt]);
            const v2f c2v = {c2, c2};
            // first consumers of the two MFMA results: compiler-visible
            v2f t_lo = __builtin_elementwise_fma(v2f{s[0], s[1]}, c2v, v2f{bq[0], bq[1]});
            v2f t_hi = __builtin_elementwise_fma(v2f{s[2], s[3]}, c2v, v2f{bq[2], bq[3]});
            const v2f d_lo = v2f{dp[0], dp[1]} - v2f{dl[0], dl[1]}, d_hi = v2f{dp[2], dp[3]} - v2f{dl[2], dl[3]};
            if (kExplicitMask) {
                const float* mrow = mask + ((size_t)(win % g.nWm) * kN + 16 * qt + 4 * lg) * kN + 16 * w + l15;
                t_lo[0] = fmaf(mrow[0], kLog2e, t_lo[0]); t_lo[1] = fmaf(mrow[kN], kLog2e, t_lo[1]);
                t_hi[0] = fmaf(mrow[2 * kN], kLog2e, t_hi[0]); t_hi[1] = fmaf(mrow[3 * kN], kLog2e, t_hi[1]);
            }
            if (analytic) {
                // wave-uniform branch, kept a real branch (the empty asm cannot be speculated): the analytic shift mask is
                // 3 VALU instructions per element and only the last window row / column of shifted blocks needs it
                asm volatile("" ::: "memory");
                const uint32_t ids = *reinterpret_cast<const uint32_t*>(&rid[oSt + 16 * qt]);
                if ((int)(ids & 0xff) != kreg) t_lo[0] += -100.0f * kLog2e;
                if ((int)((ids >> 8) & 0xff) != kreg) t_lo[1] += -100.0f * kLog2e;
                if ((int)((ids >> 16) & 0xff) != kreg) t_hi[0] += -100.0f * kLog2e;
                if ((int)((ids >> 24) & 0xff) != kreg) t_hi[1] += -100.0f * kLog2e;
            }
            const v2f e_lo = pk_sub_asm(t_lo, v2f{lq[0], lq[1]}), e_hi = pk_sub_asm(t_hi, v2f{lq[2], lq[3]});
            const v2f p_lo = {__builtin_amdgcn_exp2f(e_lo[0]), __builtin_amdgcn_exp2f(e_lo[1])};
            const v2f p_hi = {__builtin_amdgcn_exp2f(e_hi[0]), __builtin_amdgcn_exp2f(e_hi[1])};
            // p comes out of the transcendental unit (v_exp_f32): TRANS -> VALU is a software hazard as well, so this product
            // stays a C++ expression too
            const v2f ds_lo = p_lo * d_lo, ds_hi = p_hi * d_hi;
            {
                const v2f a = pk_add_asm(v2f{dB[qt][0], dB[qt][1]}, ds_lo), b = pk_add_asm(v2f{dB[qt][2], dB[qt][3]}, ds_hi);
                dB[qt] = v4f{a[0], a[1], b[0], b[1]};
            }
            const v4bf pp = {(__bf16)p_lo[0], (__bf16)p_lo[1], (__bf16)p_hi[0], (__bf16)p_hi[1]};
            const v4bf sp = {(__bf16)ds_lo[0], (__bf16)ds_lo[1], (__bf16)ds_hi[0], (__bf16)ds_hi[1]};
            *reinterpret_cast<v4bf*>(&dSt[oW + 16 * qt]) = sp;
            if ((qt & 1) || qt == kTiles - 1) {
                const bool single = !(qt & 1);  // last, unpaired tile: upper 16 k-slots are zero
                const v8bf pf = single ? v8bf{pp[0], pp[1], pp[2], pp[3], 0, 0, 0, 0}
                                       : v8bf{pprev[0], pprev[1], pprev[2], pprev[3], pp[0], pp[1], pp[2], pp[3]};
                const v8bf sf = single ? v8bf{sp[0], sp[1], sp[2], sp[3], 0, 0, 0, 0}
                                       : v8bf{sprev[0], sprev[1], sprev[2], sprev[3], sp[0], sp[1], sp[2], sp[3]};
                const int r0 = 16 * (single ? qt : qt - 1);  // queries 32s + 4lg + (0..3) | +16
                const __bf16* dlo = &dOs[oTr + r0 * kTP];
                const __bf16* qlo = &Qs[oTr + r0 * kTP];
                const v4s d_hi0 = single ? z4s : tr_read(dlo + 16 * kTP), d_hi1 = single ? z4s : tr_read(dlo + 16 * kTP + 16);
                const v4s q_hi0 = single ? z4s : tr_read(qlo + 16 * kTP), q_hi1 = single ? z4s : tr_read(qlo + 16 * kTP + 16);
                dv0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(dlo), d_hi0), pf, dv0, 0, 0, 0);
                dv1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(dlo + 16), d_hi1), pf, dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(qlo), q_hi0), sf, dk0, 0, 0, 0);
                dk1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(qlo + 16), q_hi1), sf, dk1, 0, 0, 0);
            }
            pprev = pp;
            sprev = sp;
        }
        if (tkk >= 0) {
            __bf16* base = dqkv + (img + tkk) * C3 + hoff + 4 * lg;
            v4bf a, c, e, f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = (__bf16)(dk0[r] * g.scale); c[r] = (__bf16)(dk1[r] * g.scale);
                e[r] = (__bf16)dv0[r]; f[r] = (__bf16)dv1[r];
            }
            GRIT_ST4(base + g.C, a);
            GRIT_ST4(base + g.C + 16, c);
            GRIT_ST4(base + 2 * g.C, e);
            GRIT_ST4(base + 2 * g.C + 16, f);
        }
        if (__any(tkk < 0)) {  // gradient of window-padding tokens flows to pad_qkv: reduce over the tile first
            const float keep = tkk < 0 ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = sum16(dk0[r] * keep) * g.scale, c = sum16(dk1[r] * keep) * g.scale;
                const float e = sum16(dv0[r] * keep), f = sum16(dv1[r] * keep);
                if (l15 == 0) {
                    atomicAdd(&pad_s[kHd + 4 * lg + r], a);
                    atomicAdd(&pad_s[kHd + 16 + 4 * lg + r], c);
                    atomicAdd(&pad_s[2 * kHd + 4 * lg + r], e);
                    atomicAdd(&pad_s[2 * kHd + 16 + 4 * lg + r], f);
                }
            }
        }
        __syncthreads();  // dSt complete
        if (win + ngrp < NW) nxt = fetch(win + ngrp);  // in flight during phase 2 and the stores

        // ================= phase 2: wave w = query tile w : dQ^T[d][q] = scale * sum_k K^T[d][k] dS^T[k][q]
        // k-slot (lg, j) = key 32s + 8lg + j; the last step covers keys 128..143 only (lanes lg >= 2 contribute zeros)
        v4f dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
        int oS2 = (8 * lg + trq) * kSP + 16 * w + 4 * trp, oK2 = (8 * lg + trq) * kTP + 4 * trp;
        int oS2d = trq * kSP + 16 * w + 4 * trp - 128 * kSP, oK2d = trq * kTP + 4 * trp - 128 * kTP;  // dead lanes, s5 = 4
        asm volatile("" : "+v"(oS2), "+v"(oK2), "+v"(oS2d), "+v"(oK2d));
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            const bool live = (s5 < 4) || (lg < 2);  // dead lanes read a valid address (EXEC stays full), then zero
            const __bf16* slo = &dSt[(live ? oS2 : oS2d) + 32 * s5 * kSP];
            const __bf16* klo = &Ks[(live ? oK2 : oK2d) + 32 * s5 * kTP];
            v8bf sf = join(tr_read(slo), tr_read(slo + 4 * kSP));
            if (!live) sf = v8bf{0, 0, 0, 0, 0, 0, 0, 0};
            dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(klo), tr_read(klo + 4 * kTP)), sf, dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(klo + 16), tr_read(klo + 4 * kTP + 16)), sf, dq1, 0, 0, 0);
        }
        if (tkk >= 0) {
            __bf16* base = dqkv + (img + tkk) * C3 + hoff + 4 * lg;
            v4bf a, c;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (__bf16)(dq0[r] * g.scale); c[r] = (__bf16)(dq1[r] * g.scale); }
            GRIT_ST4(base, a);
            GRIT_ST4(base + 16, c);
        }
        if (__any(tkk < 0)) {
            const float keep = tkk < 0 ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = sum16(dq0[r] * keep) * g.scale, c = sum16(dq1[r] * keep) * g.scale;
                if (l15 == 0) {
                    atomicAdd(&pad_s[4 * lg + r], a);
                    atomicAdd(&pad_s[16 + 4 * lg + r], c);
                }
            }
        }
    }
    // ---- flush the register-resident d(bias) of this workgroup's windows and the padded-token gradient
#pragma unroll
    for (int qt = 0; qt < kTiles; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            atomicAdd(&dbias[((size_t)h * kN + 16 * qt + 4 * lg + r) * kN + 16 * w + l15], dB[qt][r]);
    __syncthreads();
    if (tid < 3 * kHd) {
        const float v = pad_s[tid];
        if (v != 0.f) atomicAdd(&dpad[(tid / kHd) * g.C + hoff + (tid % kHd)], v);
    }
}

// DMA-staged variant (the default; GRIT_WINATTN_BWD_DMA=0 selects the register-staged kernel above): LDS plan, 163 680 of the 163 840 bytes of the CU:
//   bias slab, TRANSPOSED [key][query], as bf16 with a 152-element pitch (conflict-free 8-byte reads per half-wave): 43 776 B.  The
//     slab is the output of grit_relbias_fwd on a bf16 table in the training step, so the conversion is exact there; with an
//     fp32 table it rounds the bias to bf16 inside this kernel only (2^-9 relative on an O(0.1) logit term, below the bf16
//     resolution of P / dS that the products run in);
//   dS^T [key][query] bf16, pitch 148: 42 624 B;  Q / dO / K tiles [144][32] bf16, DOUBLE buffered: 55 296 B;  V and O tiles,
//     single (consumed at the top of their window): 18 432 B;  statistics 3 552 B (the row log-sum-exps, the region ids and the token indices of the
//     144 window positions double buffered: they arrive with the tiles).
constexpr int kBP2 = 152;
constexpr size_t kBwdDmaLds = (size_t)kN * kBP2 * 2 + (size_t)kN * kSP * 2 + 8 * (size_t)kN * kTP * 2 + 3 * kN * 4 + 3 * kHd * 4 + 2 * kN + 2 * kN * 4
                              + 64;  // (+ the kept-sample table of the drop-path variant)
static_assert(kBwdDmaLds <= 160 * 1024, "one workgroup per CU");

// kSkip (round 5): drop path.  row_scale[b] == 0 promises dO == 0 for image b (the branch was multiplied by 0): its windows contribute
// nothing to dq / dk / dv, d(bias) or the padding gradient.  The workgroup then walks the windows of the KEPT images only -- through a
// live-index -> window map, so the window loop itself is the same code with the same single exit (round 4 answered such windows from
// inside the loop: a second exit, 14 more registers in an issue-bound kernel, slower) -- after a store-only pre-loop that writes
// the zeros of the dropped images' dq / dk / dv slices.
template <bool kExplicitMask, bool kSkip = false>
__global__ __launch_bounds__(kThreads)
void winattn_bwd_dma(const __bf16* __restrict__ qkv, const float* __restrict__ rel_bias, const __bf16* __restrict__ pad_qkv,
                 const float* __restrict__ mask, Geom g, const __bf16* __restrict__ out, const __bf16* __restrict__ dout,
                 const float* __restrict__ lse2, __bf16* __restrict__ dqkv, float* __restrict__ dbias,
                 float* __restrict__ dpad, const float* __restrict__ row_scale = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __bf16* bT = reinterpret_cast<__bf16*>(smem_raw);                       // [144][kBP2]  bias^T, bf16 (see header comment)
    __bf16* dSt = bT + kN * kBP2;                                           // [144][kSP]   [key][query]
    __bf16* tiles = dSt + kN * kSP;                                         // Q, dO, K of buffer 0, then of buffer 1: [144][kTP] each
    __bf16* Vs1 = tiles + 6 * kN * kTP;                                     // V and O of the window being STARTED (read at its
    __bf16* Os1 = Vs1 + kN * kTP;                                           //   top only: single buffers)
    float* lse_s = reinterpret_cast<float*>(Os1 + kN * kTP);                // [2][144], buffer = the tile buffer of the window
    float* delta_s = lse_s + 2 * kN;                                        // [144]
    float* pad_s = delta_s + kN;                                            // [96]
    uint8_t* rid = reinterpret_cast<uint8_t*>(pad_s + 3 * kHd);             // [2][144]
    int* tok_s = reinterpret_cast<int*>(rid + 2 * kN);                      // [2][144] token index of a window position, -1: padding
    uint8_t* kept_s = reinterpret_cast<uint8_t*>(tok_s + 2 * kN);           // [64] kSkip: j-th kept image

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int trq = l15 >> 2, trp = l15 & 3;
    int h, grp;
    head_and_group(g, h, grp);
    const int ngrp = gridDim.x / g.nH;
    int NW = g.B * g.nWh * g.nWw;  // (kSkip: the windows of the kept images)
    const int C3 = 3 * g.C;
    const int hoff = h * kHd;
    const float c2 = g.scale * kLog2e;

    // one-time: bias slab (transposed, log2 domain) and the pad-gradient accumulator
    for (int i = tid; i < kN * kN; i += kThreads) {
        const int qi = i / kN, ki = i - qi * kN;
        bT[ki * kBP2 + qi] = (__bf16)rel_bias[(size_t)h * kN * kN + i];
    }
    if (tid < 3 * kHd) pad_s[tid] = 0.f;
    if constexpr (kSkip) {
        unsigned long long keep = 0ull;  // wave-uniform (scalar loads): bit b = image b is kept; B <= 64 (host)
        for (int b = 0; b < g.B; ++b) keep |= (unsigned long long)(row_scale[b] != 0.f) << b;
        if (tid < g.B && ((keep >> tid) & 1ull)) kept_s[__builtin_popcountll(keep & ((1ull << tid) - 1ull))] = (uint8_t)tid;
        NW = __builtin_popcountll(keep) * g.nWh * g.nWw;
        // dq / dk / dv of the dropped images: zeros (this head's three 64-byte slices of every token; 12 threads per token, the
        // workgroups of a head share an image's tokens)
        const int slice = (tid % 12) >> 2, chunk = tid & 3;  // (576 = 48 x 12)
        for (unsigned long long gone = ~keep & (g.B == 64 ? ~0ull : (1ull << g.B) - 1ull); gone; gone &= gone - 1ull) {
            const size_t row0 = (size_t)__builtin_ctzll(gone) * g.T;
            for (int t = grp * 48 + tid / 12; t < g.T; t += ngrp * 48)
                *reinterpret_cast<uint4*>(dqkv + (row0 + t) * C3 + slice * g.C + hoff + chunk * 8) = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();  // the table is read by the first prefetch
    }
    // live index -> window id (kSkip; identity otherwise)
    auto window_id = [&](int lv) {
        if constexpr (kSkip) {
            const int wpi = g.nWh * g.nWw;
            const int q = (int)(((float)lv + 0.5f) * g.inv_img);
            // (readfirstlane: the table entry is the same in every lane -- as a vector value it would turn the scalar address arithmetic of
            // the window's transfers into vector instructions on the issue-bound SIMDs)
            return __builtin_amdgcn_readfirstlane((int)kept_s[q]) * wpi + (lv - q * wpi);
        } else {
            return lv;
        }
    };

    v4f dB[kTiles];  // d(bias)[query 16qt + 4lg + r][key 16w + l15], summed over this workgroup's windows
#pragma unroll
    for (int qt = 0; qt < kTiles; ++qt) dB[qt] = v4f{0.f, 0.f, 0.f, 0.f};

    const v4s z4s = {0, 0, 0, 0};
    const int sn = tid >> 2, sc = tid & 3;
    // Operands of a window travel global -> LDS by DMA (global_load_lds, 16 bytes per lane, lane-linear: one wave-instruction
    // moves 16 token rows = 1 KB of a [144][32] bf16 tile), issued at the TOP of the previous window -- a whole window of
    // compute hides them, and no register holds them meanwhile (the register-staged version could only issue them after phase
    // 1, its register peak: ~20 % of the kernel was exposed load latency, profiles/r01/winattn_bwd_notes.txt).  Window-padding
    // tokens: q / k / v come from pad_qkv, dO / O rows are zero-filled by the owning thread.
    struct Next { int wy, wx; size_t img; };
    // Issued as inline asm: through __builtin_amdgcn_global_load_lds hipcc cannot tell which later LDS reads the transfer may
    // alias (one dynamic LDS block) and puts `s_waitcnt vmcnt(0)` right behind the issue -- the prefetch then overlaps nothing
    // (found in round 3 on the weight-gradient GEMM; it is why this variant measured no better than register staging).  The
    // waits are this kernel's own: `vmcnt(0)` + barrier at the top of the window that consumes the tiles.
    // (LDS destinations as 32-bit offsets from the block's base: generic pointers cost a register pair and a null check each)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem_raw;
    constexpr unsigned kOffTiles = (kN * kBP2 + kN * kSP) * 2, kTileBytes = kN * kTP * 2;
    constexpr unsigned kOffV = kOffTiles + 6 * kTileBytes, kOffO = kOffV + kTileBytes, kOffLse = kOffO + kTileBytes;
    auto dma16 = [&](const __bf16* src, unsigned tile_off, int group) {  // rows 16 group .. 16 group + 15 of a [144][32] tile
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + tile_off + group * 1024);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory", "m0");
    };
    // The row log-sum-exps (4 bytes per lane, threads 0 .. 143) travel the same way.  As a plain load into a register that is
    // carried to the next window they cost a fifth of the kernel: hipcc, which does not see the asm transfers, puts
    // `s_waitcnt vmcnt(0)` in front of the load (the register still has the previous window's load pending in its model) --
    // right behind the five tile transfers just issued, so every window waited for its prefetch to land
    // (s_memtime stamps: 3.5-5.4k of 17.4k cycles per window, profiles/r03/winattn_bwd_stamps.txt).
    auto dma4 = [&](const float* src, unsigned row_off, int piece) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + row_off + piece * 256);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(src) : "memory", "m0");
    };
    // Who issues them: the nine waves sit on four SIMDs, 3-2-2-2, and the kernel is instruction-issue bound on the SIMD that carries
    // waves 0, 4 and 8 (stamps: wave 8 finishes every phase last and everybody waits for it at the next barrier).  The transfers and
    // their address arithmetic (~170 instructions per wave and window) are therefore issued by the six waves of the other three
    // SIMDs: loader li moves token rows 16 li .. 16 li + 15 of all five tiles, and rows 96 .. 143 are shared by loader pairs (the
    // even one takes Q / K / V and the log-sum-exps, the odd one dO / O).
    const int li = (w & 3) == 0 ? -1 : (w < 4 ? w - 1 : w - 2);
    auto fetch_rows = [&](int group, int wy, int wx, size_t img, int buf, uint8_t* rid_n, int* tok_n, bool qkv_part, bool do_part) {
        const unsigned tb_off = kOffTiles + buf * 3 * kTileBytes;
        int reg;
        // 64-byte DMA rows cannot be padded: the 16-byte chunks of a row are permuted on the SOURCE address instead -- position p of
        // row r holds logical chunk p ^ swz(r & 15), swz = (r & 8 ? 3 : 0) ^ (r & 4 ? 2 : 0): conflict-free for the row reads
        // (ds_read_b128, 16 rows x one chunk) and for both transposing read patterns (8 rows x 32 bytes, rows 4 or 8 apart)
        // by the lane groups of MI355X_MICROARCH.md "LDS" (SQ_LDS_BANK_CONFLICT, profiles/r03/winattn_sq_counters*.txt)
        int row = 16 * group + (lane >> 2), col = hoff + ((sc ^ (((lane >> 2) & 8) ? 3 : 0) ^ (((lane >> 2) & 4) ? 2 : 0)) << 3);
        asm volatile("" : "+v"(row), "+v"(col));  // per-window values: hoisted out of the window loop they cost registers the loop does not have
        const int tk = token_of(row, wy, wx, g, reg);
        if (qkv_part) {
            const __bf16* src = tk >= 0 ? qkv + (img + tk) * C3 + col : pad_qkv + col;
            dma16(src, tb_off, group);                          // Q
            dma16(src + g.C, tb_off + 2 * kTileBytes, group);   // K
            dma16(src + 2 * g.C, kOffV, group);                 // V
            if (sc == 0) { rid_n[row] = (uint8_t)reg; tok_n[row] = tk; }
        }
        if (do_part) {
            if (tk >= 0) {
                dma16(dout + (img + tk) * g.C + col, tb_off + kTileBytes, group);   // dO
                dma16(out + (img + tk) * g.C + col, kOffO, group);                  // O
            } else {
                __bf16* tb = tiles + buf * 3 * kN * kTP;
                *reinterpret_cast<uint4*>(&tb[kN * kTP + row * kTP + sc * 8]) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(&Os1[row * kTP + sc * 8]) = make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto prefetch = [&](int win, int buf) {
        Next f;
        int b;
        window_of(win, g, b, f.wy, f.wx);
        f.img = (size_t)b * g.T;
        if (li >= 0) {  // wave-uniform
            uint8_t* rid_n = rid + buf * kN;
            int* tok_n = tok_s + buf * kN;
            fetch_rows(li, f.wy, f.wx, f.img, buf, rid_n, tok_n, true, true);
            fetch_rows(6 + (li >> 1), f.wy, f.wx, f.img, buf, rid_n, tok_n, !(li & 1), (li & 1) != 0);
            if (!(li & 1)) {
                const int piece = li >> 1;
                int t = 64 * piece + lane;
                asm volatile("" : "+v"(t));
                if (t < kN) dma4(lse2 + ((size_t)win * g.nH + h) * kN + t, kOffLse + buf * kN * 4, piece);
            }
        }
        return f;
    };
    Next nxt;
    int cur = 0;
    if (grp < NW) nxt = prefetch(window_id(grp), 0);
    for (int lv = grp; lv < NW; lv += ngrp, cur ^= 1) {
        [[maybe_unused]] const int win = lv;  // (the explicit-mask variant indexes its mask by window id; it never runs with kSkip)
        const int wy = nxt.wy, wx = nxt.wx;
        const size_t img = nxt.img;
        __bf16* Qs = tiles + cur * 3 * kN * kTP;
        __bf16* dOs = Qs + kN * kTP;
        __bf16* Ks = dOs + kN * kTP;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs (and its stores of the previous window) are done
        __syncthreads();  // ... everybody's: the tiles of this window are complete (first time: so is the bias slab)
        float dpart = 0.f;  // delta = rowsum(dO * O): 8 channels per thread, 4 threads per token
        {
            const v8bf a = as_v8bf(*reinterpret_cast<const uint4*>(&dOs[sn * kTP + sc * 8]));
            const v8bf c = as_v8bf(*reinterpret_cast<const uint4*>(&Os1[sn * kTP + sc * 8]));
#pragma unroll
            for (int e = 0; e < 8; ++e) dpart = fmaf((float)a[e], (float)c[e], dpart);
            dpart = sum_quad(dpart);
        }
        if (sc == 0) delta_s[sn] = dpart;
        // this lane's key in phase 1: its token (-1: window padding) and shift-mask region, as the loaders left them
        const int tkk = tok_s[cur * kN + 16 * w + l15], kreg = rid[cur * kN + 16 * w + l15];
        // this lane's K / V fragments (B operands of phase 1): key 16 w + l15, channels 8 lg ..
        const int swr = ((l15 & 8) ? 3 : 0) ^ ((l15 & 4) ? 2 : 0);   // chunk permutation of row l15 of a 16-row block (see fetch_rows)
        const v8bf kf = as_v8bf(*reinterpret_cast<const uint4*>(&Ks[(16 * w + l15) * kTP + ((lg ^ swr) << 3)]));
        const v8bf vf = as_v8bf(*reinterpret_cast<const uint4*>(&Vs1[(16 * w + l15) * kTP + ((lg ^ swr) << 3)]));
        __syncthreads();  // statistics visible; the V / O tiles and the other tile buffer are free for the next window's DMA
        if (lv + ngrp < NW) nxt = prefetch(window_id(lv + ngrp), cur ^ 1);  // lands under phases 1 and 2 of this window

        // Per-lane LDS offsets, made opaque once per window: without this hipcc hoists ~90 loop-invariant LDS
        // addresses out of the window loop, runs out of registers and reloads them from scratch before every read.
        int oRow = l15 * kTP + ((lg ^ swr) << 3);        // row reads of Qs / dOs      (+ 16 qt kTP)
        // transposing reads, phase 1 (+ 32 s kTP): row 4 lg + trq, channels 4 trp .. (logical chunk trp >> 1) and + 16 (chunk + 2)
        const int pa = (trp >> 1) ^ ((lg & 2) ? 3 : 0) ^ ((lg & 1) ? 2 : 0);
        int oTr = (4 * lg + trq) * kTP + (pa << 3) + ((trp & 1) << 2);
        int oTr16 = (4 * lg + trq) * kTP + ((pa ^ 2) << 3) + ((trp & 1) << 2);
        int oB = (16 * w + l15) * kBP2 + 4 * lg;         // bias slab (bf16)           (+ 16 qt)
        int oSt = 4 * lg;                                // delta, region ids          (+ 16 qt)
        int oLse = cur * kN + 4 * lg;                    // lse of this window's buffer (+ 16 qt)
        int oW = (16 * w + l15) * kSP + 4 * lg;          // dS^T writes                (+ 16 qt)
        asm volatile("" : "+v"(oRow), "+v"(oTr), "+v"(oTr16), "+v"(oB), "+v"(oSt), "+v"(oLse), "+v"(oW));

        // ================= phase 1: wave w = key tile w =================
        // S[q][k] = Q K^T, dP[q][k] = dO V^T on tiles (qt, w); after each PAIR of query tiles (one 32-deep k-step of
        // the products that sum over queries) the packed P / dS feed dV^T += dO^T P and dK^T += Q^T dS at once.
        const bool analytic = !kExplicitMask && g.shift > 0 && (wy == g.nWh - 1 || wx == g.nWw - 1);
        v4f dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = dv0, dk0 = dv0, dk1 = dv0;
        v4bf pprev, sprev;
        // software pipeline by one query tile: the S / dP MFMAs of tile qt+1 (and the LDS reads feeding them) are issued
        // before the element-wise work of tile qt, so their latency hides under ~60 VALU instructions instead of parking the
        // wave (PMC: 52 % of the wave cycles were s_waitcnt / barrier waits, 26 % issue stalls on MFMA results)
        auto score_tiles = [&](int qt, v4f& s_out, v4f& dp_out) {
            const v8bf qa = as_v8bf(*reinterpret_cast<const uint4*>(&Qs[oRow + 16 * qt * kTP]));
            const v8bf da = as_v8bf(*reinterpret_cast<const uint4*>(&dOs[oRow + 16 * qt * kTP]));
            s_out = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            dp_out = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da, vf, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        };
        v4f s_cur, dp_cur;
        score_tiles(0, s_cur, dp_cur);
#pragma unroll
        for (int qt = 0; qt < kTiles; ++qt) {
            const v4f s = s_cur, dp = dp_cur;
            if (qt + 1 < kTiles) {  // next tile's MFMAs and LDS reads fly under this tile's element-wise work
                score_tiles(qt + 1, s_cur, dp_cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            // element-wise part, two score elements per instruction: the loop is instruction-issue bound and hipcc does not
            // form packed-fp32 operations from this code by itself, so the fma / sub / mul / add pairs are spelled out
            // (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 on 64-bit register pairs, which the halves of an MFMA result are)
            const uint2 braw = *reinterpret_cast<const uint2*>(&bT[oB + 16 * qt]);  // 4 bf16 biases: exact in f32, scaled here
            const v2f b_lo = v2f{__uint_as_float(braw.x << 16), __uint_as_float(braw.x & 0xffff0000u)} * v2f{kLog2e, kLog2e};
            const v2f b_hi = v2f{__uint_as_float(braw.y << 16), __uint_as_float(braw.y & 0xffff0000u)} * v2f{kLog2e, kLog2e};
            const v4f bq = {b_lo[0], b_lo[1], b_hi[0], b_hi[1]};
            const v4f lq = *reinterpret_cast<const v4f*>(&lse_s[oLse + 16 * qt]);
            const v4f dl = *reinterpret_cast<const v4f*>(&delta_s[oSt + 16 * qt]);
            const v2f c2v = {c2, c2};
            // first consumers of the two MFMA results: compiler-visible
            v2f t_lo = __builtin_elementwise_fma(v2f{s[0], s[1]}, c2v, v2f{bq[0], bq[1]});
            v2f t_hi = __builtin_elementwise_fma(v2f{s[2], s[3]}, c2v, v2f{bq[2], bq[3]});
            const v2f d_lo = v2f{dp[0], dp[1]} - v2f{dl[0], dl[1]}, d_hi = v2f{dp[2], dp[3]} - v2f{dl[2], dl[3]};
            if (kExplicitMask) {
                const float* mrow = mask + ((size_t)(win % g.nWm) * kN + 16 * qt + 4 * lg) * kN + 16 * w + l15;
                t_lo[0] = fmaf(mrow[0], kLog2e, t_lo[0]); t_lo[1] = fmaf(mrow[kN], kLog2e, t_lo[1]);
                t_hi[0] = fmaf(mrow[2 * kN], kLog2e, t_hi[0]); t_hi[1] = fmaf(mrow[3 * kN], kLog2e, t_hi[1]);
            }
            if (analytic) {
                // wave-uniform branch, kept a real branch (the empty asm cannot be speculated): the analytic shift mask is
                // 3 VALU instructions per element and only the last window row / column of shifted blocks needs it
                asm volatile("" ::: "memory");
                const uint32_t ids = *reinterpret_cast<const uint32_t*>(&rid[oLse + 16 * qt]);
                if ((int)(ids & 0xff) != kreg) t_lo[0] += -100.0f * kLog2e;
                if ((int)((ids >> 8) & 0xff) != kreg) t_lo[1] += -100.0f * kLog2e;
                if ((int)((ids >> 16) & 0xff) != kreg) t_hi[0] += -100.0f * kLog2e;
                if ((int)((ids >> 24) & 0xff) != kreg) t_hi[1] += -100.0f * kLog2e;
            }
            const v2f e_lo = pk_sub_asm(t_lo, v2f{lq[0], lq[1]}), e_hi = pk_sub_asm(t_hi, v2f{lq[2], lq[3]});
            const v2f p_lo = {__builtin_amdgcn_exp2f(e_lo[0]), __builtin_amdgcn_exp2f(e_lo[1])};
            const v2f p_hi = {__builtin_amdgcn_exp2f(e_hi[0]), __builtin_amdgcn_exp2f(e_hi[1])};
            // p comes out of the transcendental unit (v_exp_f32): TRANS -> VALU is a software hazard as well, so this product
            // stays a C++ expression too
            const v2f ds_lo = p_lo * d_lo, ds_hi = p_hi * d_hi;
            {
                const v2f a = pk_add_asm(v2f{dB[qt][0], dB[qt][1]}, ds_lo), b = pk_add_asm(v2f{dB[qt][2], dB[qt][3]}, ds_hi);
                dB[qt] = v4f{a[0], a[1], b[0], b[1]};
            }
            const v4bf pp = {(__bf16)p_lo[0], (__bf16)p_lo[1], (__bf16)p_hi[0], (__bf16)p_hi[1]};
            const v4bf sp = {(__bf16)ds_lo[0], (__bf16)ds_lo[1], (__bf16)ds_hi[0], (__bf16)ds_hi[1]};
            *reinterpret_cast<v4bf*>(&dSt[oW + 16 * qt]) = sp;
            if ((qt & 1) || qt == kTiles - 1) {
                const bool single = !(qt & 1);  // last, unpaired tile: upper 16 k-slots are zero
                const v8bf pf = single ? v8bf{pp[0], pp[1], pp[2], pp[3], 0, 0, 0, 0}
                                       : v8bf{pprev[0], pprev[1], pprev[2], pprev[3], pp[0], pp[1], pp[2], pp[3]};
                const v8bf sf = single ? v8bf{sp[0], sp[1], sp[2], sp[3], 0, 0, 0, 0}
                                       : v8bf{sprev[0], sprev[1], sprev[2], sprev[3], sp[0], sp[1], sp[2], sp[3]};
                const int r0 = 16 * (single ? qt : qt - 1);  // queries 32s + 4lg + (0..3) | +16
                const __bf16* dlo = &dOs[oTr + r0 * kTP];
                const __bf16* qlo = &Qs[oTr + r0 * kTP];
                const __bf16* dlo16 = &dOs[oTr16 + r0 * kTP];  // channels + 16 of the same rows
                const __bf16* qlo16 = &Qs[oTr16 + r0 * kTP];
                const v4s d_hi0 = single ? z4s : tr_read(dlo + 16 * kTP), d_hi1 = single ? z4s : tr_read(dlo16 + 16 * kTP);
                const v4s q_hi0 = single ? z4s : tr_read(qlo + 16 * kTP), q_hi1 = single ? z4s : tr_read(qlo16 + 16 * kTP);
                dv0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(dlo), d_hi0), pf, dv0, 0, 0, 0);
                dv1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(dlo16), d_hi1), pf, dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(qlo), q_hi0), sf, dk0, 0, 0, 0);
                dk1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(qlo16), q_hi1), sf, dk1, 0, 0, 0);
            }
            pprev = pp;
            sprev = sp;
        }
        if (tkk >= 0) {
            __bf16* base = dqkv + (img + tkk) * C3 + hoff + 4 * lg;
            v4bf a, c, e, f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a[r] = (__bf16)(dk0[r] * g.scale); c[r] = (__bf16)(dk1[r] * g.scale);
                e[r] = (__bf16)dv0[r]; f[r] = (__bf16)dv1[r];
            }
            GRIT_ST4(base + g.C, a);
            GRIT_ST4(base + g.C + 16, c);
            GRIT_ST4(base + 2 * g.C, e);
            GRIT_ST4(base + 2 * g.C + 16, f);
        }
        if (__any(tkk < 0)) {  // gradient of window-padding tokens flows to pad_qkv: reduce over the tile first
            const float keep = tkk < 0 ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = sum16(dk0[r] * keep) * g.scale, c = sum16(dk1[r] * keep) * g.scale;
                const float e = sum16(dv0[r] * keep), f = sum16(dv1[r] * keep);
                if (l15 == 0) {
                    atomicAdd(&pad_s[kHd + 4 * lg + r], a);
                    atomicAdd(&pad_s[kHd + 16 + 4 * lg + r], c);
                    atomicAdd(&pad_s[2 * kHd + 4 * lg + r], e);
                    atomicAdd(&pad_s[2 * kHd + 16 + 4 * lg + r], f);
                }
            }
        }
        __syncthreads();  // dSt complete

        // ================= phase 2: wave w = query tile w : dQ^T[d][q] = scale * sum_k K^T[d][k] dS^T[k][q]
        // k-slot (lg, j) = key 32s + 8lg + j; the last step covers keys 128..143 only (lanes lg >= 2 contribute zeros)
        // (Handing the nine query tiles to the six waves off SIMD 0 as 18 half-units, three each, was tried: nothing at stages
        // 0 / 1, +4 % at stage 2 -- this phase is bound by its dependent MFMA / LDS chain, not by issue slots.)
        v4f dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = dq0;
        // K rows 8 lg + trq (and + 4) of a 32-key step: position of logical chunk trp >> 1 in row r is pk, in row r + 4 it is pk ^ 2,
        // and channels + 16 (chunk + 2) sit at the other of the two
        const int pk = (trp >> 1) ^ ((lg & 1) ? 3 : 0);
        int oS2 = (8 * lg + trq) * kSP + 16 * w + 4 * trp, oK2 = (8 * lg + trq) * kTP + (pk << 3) + ((trp & 1) << 2);
        int oK2x = (8 * lg + trq) * kTP + ((pk ^ 2) << 3) + ((trp & 1) << 2);
        int oS2d = trq * kSP + 16 * w + 4 * trp - 128 * kSP, oK2d = trq * kTP + 4 * trp - 128 * kTP;  // dead lanes, s5 = 4
        asm volatile("" : "+v"(oS2), "+v"(oK2), "+v"(oK2x), "+v"(oS2d), "+v"(oK2d));
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
            const bool live = (s5 < 4) || (lg < 2);  // dead lanes read a valid address (EXEC stays full), then zero
            const __bf16* slo = &dSt[(live ? oS2 : oS2d) + 32 * s5 * kSP];
            const __bf16* klo = &Ks[(live ? oK2 : oK2d) + 32 * s5 * kTP];
            const __bf16* klx = &Ks[(live ? oK2x : oK2d) + 32 * s5 * kTP];
            v8bf sf = join(tr_read(slo), tr_read(slo + 4 * kSP));
            if (!live) sf = v8bf{0, 0, 0, 0, 0, 0, 0, 0};
            dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(klo), tr_read(klx + 4 * kTP)), sf, dq0, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(join(tr_read(klx), tr_read(klo + 4 * kTP)), sf, dq1, 0, 0, 0);
        }
        if (tkk >= 0) {
            __bf16* base = dqkv + (img + tkk) * C3 + hoff + 4 * lg;
            v4bf a, c;
#pragma unroll
            for (int r = 0; r < 4; ++r) { a[r] = (__bf16)(dq0[r] * g.scale); c[r] = (__bf16)(dq1[r] * g.scale); }
            GRIT_ST4(base, a);
            GRIT_ST4(base + 16, c);
        }
        if (__any(tkk < 0)) {
            const float keep = tkk < 0 ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = sum16(dq0[r] * keep) * g.scale, c = sum16(dq1[r] * keep) * g.scale;
                if (l15 == 0) {
                    atomicAdd(&pad_s[4 * lg + r], a);
                    atomicAdd(&pad_s[16 + 4 * lg + r], c);
                }
            }
        }
    }
    // ---- flush the register-resident d(bias) of this workgroup's windows and the padded-token gradient
#pragma unroll
    for (int qt = 0; qt < kTiles; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            atomicAdd(&dbias[((size_t)h * kN + 16 * qt + 4 * lg + r) * kN + 16 * w + l15], dB[qt][r]);
    __syncthreads();
    if (tid < 3 * kHd) {
        const float v = pad_s[tid];
        if (v != 0.f) atomicAdd(&dpad[(tid / kHd) * g.C + hoff + (tid % kHd)], v);
    }
}

// ---------------------------------------------------------------------------------------------------
// fp32 arithmetic (the parity path: fp32 weights -> the reference's fp32 WindowAttention within 1e-4, swin_model.py:155-186)
// ---------------------------------------------------------------------------------------------------
// Plain fmaf chains, no matrix cores: one workgroup per (window, head), thread = query row (forward, dQ) or key row
// (dK, dV, d(bias)); K / V / Q / dO rows in LDS and read as broadcasts.  Not a throughput path: it runs when the model is
// evaluated or trained in fp32 (tests, config 1 on the device, the --fp32 diagnostic of bench.py).
constexpr int kFP = 33;          // fp32 row pitch (floats): odd -> the per-thread row reads of the backward are conflict-free
constexpr int kF32Threads = 192;

__device__ __forceinline__ float logit_f32(const float* __restrict__ qrow, const float* __restrict__ krow, float scale,
                                           float bias, float maskv) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < kHd; ++d) s = fmaf(qrow[d] * scale, krow[d], s);  // (q * scale) @ k^T, as the reference orders it
    return s + bias + maskv;
}

__global__ __launch_bounds__(kF32Threads)
void winattn_fwd_f32(const float* __restrict__ qkv, const float* __restrict__ rel_bias, const float* __restrict__ pad_qkv,
                     const float* __restrict__ mask, Geom g, float* __restrict__ out, float* __restrict__ lse) {
    __shared__ float Ks[kN * kFP], Vs[kN * kFP];
    __shared__ uint8_t rid[kN];
    const int unit = blockIdx.x, win = unit / g.nH, h = unit - win * g.nH;
    const int b = win / (g.nWh * g.nWw), wrem = win - b * (g.nWh * g.nWw);
    const int wy = wrem / g.nWw, wx = wrem - wy * g.nWw;
    const size_t img = (size_t)b * g.T;
    const int C3 = 3 * g.C, hoff = h * kHd, t = threadIdx.x;
    int reg = 0, tq = -1;
    float q[kHd];
    if (t < kN) {
        tq = token_of(t, wy, wx, g, reg);
        const float* src = tq >= 0 ? qkv + (img + tq) * C3 + hoff : pad_qkv + hoff;
#pragma unroll
        for (int d = 0; d < kHd; ++d) {
            q[d] = src[d];
            Ks[t * kFP + d] = src[g.C + d];
            Vs[t * kFP + d] = src[2 * g.C + d];
        }
        rid[t] = (uint8_t)reg;
    }
    __syncthreads();
    if (t >= kN) return;
    const bool analytic = (mask == nullptr) && g.shift > 0;
    const float* brow = rel_bias + ((size_t)h * kN + t) * kN;
    const float* mrow = mask ? mask + ((size_t)(win % g.nWm) * kN + t) * kN : nullptr;
    // two passes over the keys: row maximum first, then exp / sum / PV (the logits are recomputed: 144 x 32 fmaf)
    float m = -INFINITY;
    for (int k = 0; k < kN; ++k) {
        const float mv = mrow ? mrow[k] : (analytic && rid[k] != reg ? -100.0f : 0.0f);
        m = fmaxf(m, logit_f32(q, &Ks[k * kFP], g.scale, brow[k], mv));
    }
    float o[kHd], sum = 0.f;
#pragma unroll
    for (int d = 0; d < kHd; ++d) o[d] = 0.f;
    for (int k = 0; k < kN; ++k) {
        const float mv = mrow ? mrow[k] : (analytic && rid[k] != reg ? -100.0f : 0.0f);
        const float p = expf(logit_f32(q, &Ks[k * kFP], g.scale, brow[k], mv) - m);
        sum += p;
#pragma unroll
        for (int d = 0; d < kHd; ++d) o[d] = fmaf(p, Vs[k * kFP + d], o[d]);
    }
    const float inv = 1.0f / sum;
    if (tq >= 0) {
        float* orow = out + (img + tq) * g.C + hoff;
#pragma unroll
        for (int d = 0; d < kHd; ++d) orow[d] = o[d] * inv;
    }
    lse[((size_t)win * g.nH + h) * kN + t] = m + logf(sum);  // natural log
}

__global__ __launch_bounds__(kF32Threads)
void winattn_bwd_f32(const float* __restrict__ qkv, const float* __restrict__ rel_bias, const float* __restrict__ pad_qkv,
                     const float* __restrict__ mask, Geom g, const float* __restrict__ out, const float* __restrict__ dout,
                     const float* __restrict__ lse, float* __restrict__ dqkv, float* __restrict__ dbias,
                     float* __restrict__ dpad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];  // 77.5 KB: above the static limit
    float* Qs = reinterpret_cast<float*>(smem_raw);
    float* Ks = Qs + kN * kFP;
    float* Vs = Ks + kN * kFP;
    float* dOs = Vs + kN * kFP;
    float* lse_s = dOs + kN * kFP;
    float* delta_s = lse_s + kN;
    uint8_t* rid = reinterpret_cast<uint8_t*>(delta_s + kN);
    const int unit = blockIdx.x, win = unit / g.nH, h = unit - win * g.nH;
    const int b = win / (g.nWh * g.nWw), wrem = win - b * (g.nWh * g.nWw);
    const int wy = wrem / g.nWw, wx = wrem - wy * g.nWw;
    const size_t img = (size_t)b * g.T;
    const int C3 = 3 * g.C, hoff = h * kHd, t = threadIdx.x;
    int reg = 0, tk = -1;
    if (t < kN) {
        tk = token_of(t, wy, wx, g, reg);
        const float* src = tk >= 0 ? qkv + (img + tk) * C3 + hoff : pad_qkv + hoff;
        float delta = 0.f;
#pragma unroll
        for (int d = 0; d < kHd; ++d) {
            Qs[t * kFP + d] = src[d];
            Ks[t * kFP + d] = src[g.C + d];
            Vs[t * kFP + d] = src[2 * g.C + d];
            const float go = tk >= 0 ? dout[(img + tk) * g.C + hoff + d] : 0.f;  // cropped rows get no gradient
            const float oo = tk >= 0 ? out[(img + tk) * g.C + hoff + d] : 0.f;
            dOs[t * kFP + d] = go;
            delta = fmaf(go, oo, delta);
        }
        rid[t] = (uint8_t)reg;
        lse_s[t] = lse[((size_t)win * g.nH + h) * kN + t];
        delta_s[t] = delta;
    }
    __syncthreads();
    if (t >= kN) return;
    const bool analytic = (mask == nullptr) && g.shift > 0;
    const float* mbase = mask ? mask + (size_t)(win % g.nWm) * kN * kN : nullptr;
    // ---- thread = query t: dQ
    {
        float dq[kHd];
#pragma unroll
        for (int d = 0; d < kHd; ++d) dq[d] = 0.f;
        const float* brow = rel_bias + ((size_t)h * kN + t) * kN;
        for (int k = 0; k < kN; ++k) {
            const float mv = mbase ? mbase[t * kN + k] : (analytic && rid[k] != reg ? -100.0f : 0.0f);
            const float p = expf(logit_f32(&Qs[t * kFP], &Ks[k * kFP], g.scale, brow[k], mv) - lse_s[t]);
            float dp = 0.f;
#pragma unroll
            for (int d = 0; d < kHd; ++d) dp = fmaf(dOs[t * kFP + d], Vs[k * kFP + d], dp);
            const float ds = p * (dp - delta_s[t]);
#pragma unroll
            for (int d = 0; d < kHd; ++d) dq[d] = fmaf(ds, Ks[k * kFP + d], dq[d]);
        }
        if (tk >= 0) {
            float* base = dqkv + (img + tk) * C3 + hoff;
#pragma unroll
            for (int d = 0; d < kHd; ++d) base[d] = dq[d] * g.scale;
        } else {
#pragma unroll
            for (int d = 0; d < kHd; ++d) atomicAdd(&dpad[hoff + d], dq[d] * g.scale);
        }
    }
    // ---- thread = key t: dK, dV, d(bias)[:, t]
    {
        float dk[kHd], dv[kHd];
#pragma unroll
        for (int d = 0; d < kHd; ++d) dk[d] = dv[d] = 0.f;
        for (int qi = 0; qi < kN; ++qi) {
            const float mv = mbase ? mbase[qi * kN + t] : (analytic && (int)rid[qi] != reg ? -100.0f : 0.0f);
            const float bias = rel_bias[((size_t)h * kN + qi) * kN + t];
            const float p = expf(logit_f32(&Qs[qi * kFP], &Ks[t * kFP], g.scale, bias, mv) - lse_s[qi]);
            float dp = 0.f;
#pragma unroll
            for (int d = 0; d < kHd; ++d) dp = fmaf(dOs[qi * kFP + d], Vs[t * kFP + d], dp);
            const float ds = p * (dp - delta_s[qi]);
            if (ds != 0.f) atomicAdd(&dbias[((size_t)h * kN + qi) * kN + t], ds);
#pragma unroll
            for (int d = 0; d < kHd; ++d) {
                dk[d] = fmaf(ds, Qs[qi * kFP + d], dk[d]);
                dv[d] = fmaf(p, dOs[qi * kFP + d], dv[d]);
            }
        }
        if (tk >= 0) {
            float* base = dqkv + (img + tk) * C3 + hoff;
#pragma unroll
            for (int d = 0; d < kHd; ++d) { base[g.C + d] = dk[d] * g.scale; base[2 * g.C + d] = dv[d]; }
        } else {
#pragma unroll
            for (int d = 0; d < kHd; ++d) {
                atomicAdd(&dpad[g.C + hoff + d], dk[d] * g.scale);
                atomicAdd(&dpad[2 * g.C + hoff + d], dv[d]);
            }
        }
    }
}

constexpr size_t kBwdF32Lds = (size_t)4 * kN * kFP * 4 + 2 * kN * 4 + kN;
constexpr size_t kBwdLds = (size_t)kN * kBP * 4 + 3 * (size_t)kN * kTP * 2 + (size_t)kN * kSP * 2 + 2 * kN * 4 + 3 * kHd * 4 + kN;

int check_geom(int B, int H, int W, int C, int nH, int window, int shift) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || nH <= 0) return GRIT_ERR_BAD_ARG;
    if (window != kWs || C != nH * kHd || shift < 0 || shift >= kWs) return GRIT_ERR_UNSUPPORTED;
    if ((long long)B * H * W * 3 * C >= (1LL << 40)) return GRIT_ERR_BAD_ARG;
    if ((long long)B * ((H + kWs - 1) / kWs) * ((W + kWs - 1) / kWs) >= (1LL << 21)) return GRIT_ERR_UNSUPPORTED;  // window_of
    return GRIT_OK;
}

Geom make_geom(int B, int H, int W, int C, int nH, int shift, float scale, int nWm) {
    Geom g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.nH = nH; g.shift = shift;
    g.nWh = (H + kWs - 1) / kWs; g.nWw = (W + kWs - 1) / kWs;
    g.Hp = g.nWh * kWs; g.Wp = g.nWw * kWs; g.T = H * W; g.nWm = nWm > 0 ? nWm : 1; g.scale = scale;
    g.xcd_pairs = 0;
    g.inv_img = 1.0f / (float)(g.nWh * g.nWw); g.inv_nww = 1.0f / (float)g.nWw;
    return g;
}

int grid_blocks(const Geom& g, int target) {
    // persistent: each workgroup serves one head and walks a strided set of windows, one workgroup per CU (both
    // kernels are register / LDS bound to one 9-wave workgroup per CU; measured: 256 beats 512 / 1024 / 2048 by 2-8 %
    // forward and 5-20 % backward, profiles/r01/winattn_bwd_notes.txt), so the bias slice is loaded and the d(bias)
    // accumulators are flushed -- the only contended global atomics -- once per CU.
    const int NW = g.B * g.nWh * g.nWw;
    if (const char* e = getenv("GRIT_WINATTN_BLOCKS")) target = atoi(e);  // tuning knob for tools/bench_kernels.py
    int groups = (target + g.nH - 1) / g.nH;
    if (groups > NW) groups = NW;
    if (groups < 1) groups = 1;
    return groups * g.nH;
}

Geom with_xcd_mapping(Geom g, int blocks) {
    const int groups = blocks / g.nH;
    g.xcd_pairs = (g.nH % 8 == 0 && g.nH >= 16) || (g.nH == 8 && groups % 2 == 0) || (g.nH == 4 && groups % 4 == 0);
    if (getenv("GRIT_WINATTN_PLAIN_MAP")) g.xcd_pairs = 0;  // A/B knob for tools/bench_kernels.py
    return g;
}

}  // namespace

extern "C" {

int grit_winattn_fwd_bf16(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                          int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                          void* out, float* lse, void* stream) {
    return grit_winattn_fwd_bf16_rows(qkv, rel_bias, pad_qkv, mask, n_mask_windows, B, H, W, C, num_heads, window, shift, scale, out, lse,
                                      nullptr, stream);
}

// (one rule for both directions: a forward that skipped an image must meet a backward that skips it too)
static bool winattn_rows_apply(const float* row_scale, const float* mask, int B) {
    static const bool row_skip = !(getenv("GRIT_WINATTN_ROW_SKIP") && atoi(getenv("GRIT_WINATTN_ROW_SKIP")) == 0);
    static const bool fwd_dma = !(getenv("GRIT_WINATTN_FWD_DMA") && atoi(getenv("GRIT_WINATTN_FWD_DMA")) == 0);
    static const bool bwd_dma = !(getenv("GRIT_WINATTN_BWD_DMA") && atoi(getenv("GRIT_WINATTN_BWD_DMA")) == 0);
    return row_scale && !mask && row_skip && fwd_dma && bwd_dma && B >= 2 && B <= 64;
}

int grit_winattn_fwd_bf16_rows(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                               int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                               void* out, float* lse, const float* row_scale, void* stream) {
    if (!qkv || !rel_bias || !pad_qkv || !out || !lse) return GRIT_ERR_BAD_ARG;
    const int st = check_geom(B, H, W, C, num_heads, window, shift);
    if (st != GRIT_OK) return st;
    if (mask && n_mask_windows <= 0) return GRIT_ERR_BAD_ARG;
    const Geom g0 = make_geom(B, H, W, C, num_heads, shift, scale, n_mask_windows);
    const Geom g = with_xcd_mapping(g0, grid_blocks(g0, 256));
    static const bool use_dma = [] { const char* e = getenv("GRIT_WINATTN_FWD_DMA"); return !(e && atoi(e) == 0); }();
    if (use_dma && !mask) {
        static grit_detail::PerDevice<bool> attr_done_pd; bool& attr_done = attr_done_pd();
        if (!attr_done) {
            if (hipFuncSetAttribute((const void*)winattn_fwd_dma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdDmaLds) != hipSuccess ||
                hipFuncSetAttribute((const void*)winattn_fwd_dma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdDmaLds) != hipSuccess)
                return GRIT_ERR_LAUNCH;
            attr_done = true;
        }
        if (winattn_rows_apply(row_scale, mask, B))
            hipLaunchKernelGGL(winattn_fwd_dma<true>, dim3(grid_blocks(g, 256)), dim3(kThreads), kFwdDmaLds, (hipStream_t)stream,
                               (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, g, (__bf16*)out, lse, row_scale);
        else
            hipLaunchKernelGGL(winattn_fwd_dma<false>, dim3(grid_blocks(g, 256)), dim3(kThreads), kFwdDmaLds, (hipStream_t)stream,
                               (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, g, (__bf16*)out, lse, (const float*)nullptr);
        return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(winattn_fwd, dim3(grid_blocks(g, 256)), dim3(kThreads), 0, (hipStream_t)stream,
                       (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, mask, g, (__bf16*)out, lse);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_winattn_bwd_bf16(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                          const void* out, const void* dout, const float* lse, int B, int H, int W, int C, int num_heads,
                          int window, int shift, float scale, void* dqkv, float* drel_bias, float* dpad, void* stream) {
    return grit_winattn_bwd_bf16_rows(qkv, rel_bias, pad_qkv, mask, n_mask_windows, out, dout, lse, B, H, W, C, num_heads, window, shift,
                                      scale, dqkv, drel_bias, dpad, nullptr, stream);
}

int grit_winattn_bwd_bf16_rows(const void* qkv, const float* rel_bias, const void* pad_qkv, const float* mask, int n_mask_windows,
                               const void* out, const void* dout, const float* lse, int B, int H, int W, int C, int num_heads,
                               int window, int shift, float scale, void* dqkv, float* drel_bias, float* dpad, const float* row_scale,
                               void* stream) {
    if (!qkv || !rel_bias || !pad_qkv || !out || !dout || !lse || !dqkv || !drel_bias || !dpad) return GRIT_ERR_BAD_ARG;
    const int st = check_geom(B, H, W, C, num_heads, window, shift);
    if (st != GRIT_OK) return st;
    if (mask && n_mask_windows <= 0) return GRIT_ERR_BAD_ARG;
    const Geom g0 = make_geom(B, H, W, C, num_heads, shift, scale, n_mask_windows);
    const Geom g = with_xcd_mapping(g0, grid_blocks(g0, 256));
    static grit_detail::PerDevice<bool> lds_attr_set_pd; bool& lds_attr_set = lds_attr_set_pd();  // idempotent attribute, racing first calls set the same value
    if (!lds_attr_set) {
        if (hipFuncSetAttribute((const void*)winattn_bwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds) != hipSuccess ||
            hipFuncSetAttribute((const void*)winattn_bwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds) != hipSuccess ||
            hipFuncSetAttribute((const void*)winattn_bwd_dma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdDmaLds) != hipSuccess ||
            hipFuncSetAttribute((const void*)winattn_bwd_dma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdDmaLds) != hipSuccess ||
            hipFuncSetAttribute((const void*)winattn_bwd_dma<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdDmaLds) != hipSuccess)
            return GRIT_ERR_LAUNCH;
        lds_attr_set = true;
    }
    // drop path: the windows of images whose branch was multiplied by 0 are not computed (GRIT_WINATTN_ROW_SKIP=0: A/B switch)
    // The DMA-staged variant is the default since its transfers are issued as asm (before that the compiler's `vmcnt(0)` behind
    // every issue made it 3-9 % SLOWER on stages 2 / 3, profiles/r03/negative_results.txt #3): -3 % on stages 0 / 1, -2 % on
    // stage 2, equal on stage 3, step -0.1 ms (profiles/r03/winattn_dma_asm.txt).  GRIT_WINATTN_BWD_DMA=0: register staging.
    static const bool use_dma = !(getenv("GRIT_WINATTN_BWD_DMA") && atoi(getenv("GRIT_WINATTN_BWD_DMA")) == 0);
    if (use_dma) {
        if (mask)
            hipLaunchKernelGGL(winattn_bwd_dma<true>, dim3(grid_blocks(g, 256)), dim3(kThreads), kBwdDmaLds, (hipStream_t)stream,
                               (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, mask, g, (const __bf16*)out,
                               (const __bf16*)dout, lse, (__bf16*)dqkv, drel_bias, dpad, (const float*)nullptr);
        else if (winattn_rows_apply(row_scale, mask, B))
            hipLaunchKernelGGL((winattn_bwd_dma<false, true>), dim3(grid_blocks(g, 256)), dim3(kThreads), kBwdDmaLds, (hipStream_t)stream,
                               (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, mask, g, (const __bf16*)out,
                               (const __bf16*)dout, lse, (__bf16*)dqkv, drel_bias, dpad, row_scale);
        else
            hipLaunchKernelGGL(winattn_bwd_dma<false>, dim3(grid_blocks(g, 256)), dim3(kThreads), kBwdDmaLds, (hipStream_t)stream,
                               (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, mask, g, (const __bf16*)out,
                               (const __bf16*)dout, lse, (__bf16*)dqkv, drel_bias, dpad, (const float*)nullptr);
        return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
    }
    if (mask)
        hipLaunchKernelGGL(winattn_bwd<true>, dim3(grid_blocks(g, 256)), dim3(kThreads), kBwdLds, (hipStream_t)stream,
                           (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, mask, g, (const __bf16*)out,
                           (const __bf16*)dout, lse, (__bf16*)dqkv, drel_bias, dpad);
    else
        hipLaunchKernelGGL(winattn_bwd<false>, dim3(grid_blocks(g, 256)), dim3(kThreads), kBwdLds, (hipStream_t)stream,
                           (const __bf16*)qkv, rel_bias, (const __bf16*)pad_qkv, mask, g, (const __bf16*)out,
                           (const __bf16*)dout, lse, (__bf16*)dqkv, drel_bias, dpad);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_winattn_fwd_f32(const float* qkv, const float* rel_bias, const float* pad_qkv, const float* mask, int n_mask_windows,
                         int B, int H, int W, int C, int num_heads, int window, int shift, float scale,
                         float* out, float* lse, void* stream) {
    if (!qkv || !rel_bias || !pad_qkv || !out || !lse) return GRIT_ERR_BAD_ARG;
    const int st = check_geom(B, H, W, C, num_heads, window, shift);
    if (st != GRIT_OK) return st;
    if (mask && n_mask_windows <= 0) return GRIT_ERR_BAD_ARG;
    const Geom g = make_geom(B, H, W, C, num_heads, shift, scale, n_mask_windows);
    const long long units = (long long)B * g.nWh * g.nWw * num_heads;
    if (units > 0x7fffffffLL) return GRIT_ERR_BAD_ARG;
    hipLaunchKernelGGL(winattn_fwd_f32, dim3((unsigned)units), dim3(kF32Threads), 0, (hipStream_t)stream, qkv, rel_bias, pad_qkv,
                       mask, g, out, lse);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_winattn_bwd_f32(const float* qkv, const float* rel_bias, const float* pad_qkv, const float* mask, int n_mask_windows,
                         const float* out, const float* dout, const float* lse, int B, int H, int W, int C, int num_heads,
                         int window, int shift, float scale, float* dqkv, float* drel_bias, float* dpad, void* stream) {
    if (!qkv || !rel_bias || !pad_qkv || !out || !dout || !lse || !dqkv || !drel_bias || !dpad) return GRIT_ERR_BAD_ARG;
    const int st = check_geom(B, H, W, C, num_heads, window, shift);
    if (st != GRIT_OK) return st;
    if (mask && n_mask_windows <= 0) return GRIT_ERR_BAD_ARG;
    const Geom g = make_geom(B, H, W, C, num_heads, shift, scale, n_mask_windows);
    const long long units = (long long)B * g.nWh * g.nWw * num_heads;
    if (units > 0x7fffffffLL) return GRIT_ERR_BAD_ARG;
    static grit_detail::PerDevice<bool> f32_attr_set_pd; bool& f32_attr_set = f32_attr_set_pd();  // idempotent attribute, racing first calls set the same value
    if (!f32_attr_set) {
        if (hipFuncSetAttribute((const void*)winattn_bwd_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdF32Lds) != hipSuccess)
            return GRIT_ERR_LAUNCH;
        f32_attr_set = true;
    }
    hipLaunchKernelGGL(winattn_bwd_f32, dim3((unsigned)units), dim3(kF32Threads), kBwdF32Lds, (hipStream_t)stream, qkv, rel_bias,
                       pad_qkv, mask, g, out, dout, lse, dqkv, drel_bias, dpad);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // extern "C"
