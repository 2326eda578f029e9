// Shared pieces of the long-map GEMM kernels (gemm.hip: per-tile and ping-pong variants; gemm_ps.hip: persistent stream with the
// trickled epilogue): vector types, the sigmoid-form GELU of the fused Mlp epilogues, DPP row sums, the LDS chunk permutation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// GELU on the epilogue's VALU budget.  The erf form costs ~18 vector instructions per element (two transcendental, a compare /
// select pair, hazard nops): at 128 elements per lane that is as long as the whole MFMA main loop of the tile (measured with
// tools/micro/gemm_stamps.hip).  Evaluated instead as
//     gelu(x) = x * sigmoid(x * (c0 + c1 x^2 + c2 x^4)),   x^2 clamped at 50,
// with (c0, c1, c2) fitted to the erf form over [-9, 9]: |gelu error| <= 2.6e-5, |derivative error| <= 1.1e-4 absolute
// (tools/micro/fit_gelu.py) -- below the bf16 resolution of the stored activations except next to zero; the backward uses the
// exact derivative of the same expression, so forward and backward stay consistent.  7 (forward) / 13 (backward) plain
// instructions + exp2 + rcp per element.  The fp32 parity path of the model never comes here (torch's erf GELU runs there).
constexpr float kGeluC0 = 1.5950157685537665f, kGeluC1 = 0.07401129205936302f, kGeluC2 = -0.0007030335796160927f;
constexpr float kNegLog2e = -1.4426950408889634f;

__device__ __forceinline__ float gelu_f(float x) {
    const float xx = fminf(x * x, 50.0f);
    float p = fmaf(kNegLog2e * kGeluC2, xx, kNegLog2e * kGeluC1);
    p = fmaf(p, xx, kNegLog2e * kGeluC0);
    const float e = __builtin_amdgcn_exp2f(x * p);           // exp(-u), u = x P(x^2); +inf for very negative x -> s = 0
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

__device__ __forceinline__ float dgelu_f(float x) {
    const float xx = fminf(x * x, 50.0f);
    float p = fmaf(kNegLog2e * kGeluC2, xx, kNegLog2e * kGeluC1);
    p = fmaf(p, xx, kNegLog2e * kGeluC0);
    const float e = __builtin_amdgcn_exp2f(x * p);
    const float s = __builtin_amdgcn_rcpf(1.0f + e);        // sigmoid(u)
    float q = fmaf(5.0f * kGeluC2, xx, 3.0f * kGeluC1);      // du/dx = c0 + 3 c1 x^2 + 5 c2 x^4
    q = fmaf(q, xx, kGeluC0);
    return s * fmaf(x * q, 1.0f - s, 1.0f);                  // s + x s (1 - s) u'
}

// Two elements per instruction for the polynomial parts (v_pk_mul_f32 / v_pk_fma_f32 on register pairs, which the halves of an
// accumulator quad are): the epilogues run with no MFMA in flight, where the packed forms simply halve the issue slots.  Same
// operations in the same order as gelu_f / dgelu_f: identical results.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f gelu2(v2f x) {
    v2f xx = x * x;
    xx = v2f{fminf(xx[0], 50.0f), fminf(xx[1], 50.0f)};
    v2f p = __builtin_elementwise_fma(v2f{kNegLog2e * kGeluC2, kNegLog2e * kGeluC2}, xx, v2f{kNegLog2e * kGeluC1, kNegLog2e * kGeluC1});
    p = __builtin_elementwise_fma(p, xx, v2f{kNegLog2e * kGeluC0, kNegLog2e * kGeluC0});
    const v2f t = x * p;
    const v2f d = v2f{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + v2f{1.0f, 1.0f};
    return x * v2f{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}

__device__ __forceinline__ v2f dgelu2(v2f x) {
    v2f xx = x * x;
    xx = v2f{fminf(xx[0], 50.0f), fminf(xx[1], 50.0f)};
    v2f p = __builtin_elementwise_fma(v2f{kNegLog2e * kGeluC2, kNegLog2e * kGeluC2}, xx, v2f{kNegLog2e * kGeluC1, kNegLog2e * kGeluC1});
    p = __builtin_elementwise_fma(p, xx, v2f{kNegLog2e * kGeluC0, kNegLog2e * kGeluC0});
    const v2f t = x * p;
    const v2f d = v2f{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + v2f{1.0f, 1.0f};
    const v2f s = v2f{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    v2f q = __builtin_elementwise_fma(v2f{5.0f * kGeluC2, 5.0f * kGeluC2}, xx, v2f{3.0f * kGeluC1, 3.0f * kGeluC1});
    q = __builtin_elementwise_fma(q, xx, v2f{kGeluC0, kGeluC0});
    return s * __builtin_elementwise_fma(x * q, v2f{1.0f, 1.0f} - s, v2f{1.0f, 1.0f});
}

// gelu2 and dgelu2 of the same argument, sharing the sigmoid: the same operations in the same order as the two functions above
// (identical results), one exp2 + one rcp per element instead of two each.
__device__ __forceinline__ void gelu_dgelu2(v2f x, v2f& gelu, v2f& dgelu) {
    v2f xx = x * x;
    xx = v2f{fminf(xx[0], 50.0f), fminf(xx[1], 50.0f)};
    v2f p = __builtin_elementwise_fma(v2f{kNegLog2e * kGeluC2, kNegLog2e * kGeluC2}, xx, v2f{kNegLog2e * kGeluC1, kNegLog2e * kGeluC1});
    p = __builtin_elementwise_fma(p, xx, v2f{kNegLog2e * kGeluC0, kNegLog2e * kGeluC0});
    const v2f t = x * p;
    const v2f d = v2f{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + v2f{1.0f, 1.0f};
    const v2f s = v2f{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    gelu = x * s;
    v2f q = __builtin_elementwise_fma(v2f{5.0f * kGeluC2, 5.0f * kGeluC2}, xx, v2f{3.0f * kGeluC1, 3.0f * kGeluC1});
    q = __builtin_elementwise_fma(q, xx, v2f{kGeluC0, kGeluC0});
    dgelu = s * __builtin_elementwise_fma(x * q, v2f{1.0f, 1.0f} - s, v2f{1.0f, 1.0f});
}

// Sum over the 16 lanes of a DPP row (the 16 token lanes of an accumulator quarter) by row rotations: plain VALU.  As four
// __shfl_xor steps it is four ds_bpermute round trips with a full lgkmcnt wait each -- 64 of them per wave in the column-sum
// epilogue of the GELU' GEMM.
__device__ __forceinline__ float row_sum16(float v) {
#define GRIT_ROW_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, false))
    v += GRIT_ROW_ROR(v, 8); v += GRIT_ROW_ROR(v, 4); v += GRIT_ROW_ROR(v, 2); v += GRIT_ROW_ROR(v, 1);
#undef GRIT_ROW_ROR
    return v;
}

template <int BK> __device__ __forceinline__ int chunk_swizzle(int r16) {
    // permutation of the 16-byte chunks of row r16 (row index within its 16-row block) that makes the ds_read_b128 fragment
    // reads conflict-free (BK = 32: 64-byte rows, 4 chunks; BK = 64: 128-byte rows, 8 chunks)
    return BK == 32 ? ((-(r16 >> 2)) & 3) : ((r16 >> 1) & 7);
}

// counted wait on the vector-memory queue (loads, stores and LDS-DMA transfers retire in issue order)
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

}  // namespace
