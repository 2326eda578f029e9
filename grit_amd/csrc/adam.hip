// Adam over one flat parameter range of the bf16-compute / fp32-master layout (grit_amd/amp.py): ONE pass that reads the
// bf16 gradient bucket (as all-reduced by RCCL), updates the fp32 master and both fp32 moments, and writes the bf16
// compute copy the next forward reads.  torch's multi-tensor fused Adam needs the gradients widened to fp32 first and a
// separate fp32 -> bf16 copy afterwards: 40 B of HBM traffic per parameter against 28 B here, and ~40 launches against ~7.
//
// Arithmetic = torch.optim.Adam (amsgrad = False, weight_decay = 0, maximize = False; reference build_optimizers,
// engine/caption_engine.py:18-73 -- the `weight_decay_rate` key of its groups is ignored by torch):
//   m <- m + (g - m) (1 - beta1);  v <- beta2 v + (1 - beta2) g^2;  p <- p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grit_hip.h"

namespace {

template <typename GT> __device__ __forceinline__ void load4(const GT* p, float (&g)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&g)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
}
template <> __device__ __forceinline__ void load4<__hip_bfloat16>(const __hip_bfloat16* p, float (&g)[4]) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    g[0] = __uint_as_float(u.x << 16); g[1] = __uint_as_float(u.x & 0xffff0000u);
    g[2] = __uint_as_float(u.y << 16); g[3] = __uint_as_float(u.y & 0xffff0000u);
}

typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ f4 ldq(const f4* q) {
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

template <typename GT, bool NT>
__global__ __launch_bounds__(256)
void adam_flat(float* __restrict__ p, const GT* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
               __hip_bfloat16* __restrict__ compute, long n4, float step_size, float beta1, float beta2, float eps,
               float inv_bc2_sqrt, float grad_scale, const float* __restrict__ hyper) {
    // hyper != NULL: the two per-step scalars come from device memory {lr / bias_correction1, 1 / sqrt(bias_correction2)} -- a
    // launch captured in a HIP graph is replayed with the learning rate and the step count of the step it is replayed for
    if (hyper) {
        step_size = hyper[0];
        inv_bc2_sqrt = hyper[1];
    }
    // masters and moments are touched once per step: nontemporal loads / stores (they need not displace the weights the next
    // forward is about to read); two quads per thread and trip: eight 16-byte loads in flight
    const long stride = (long)gridDim.x * 256;
    auto update = [&](long i, const float (&g)[4], f4 pp, f4 mm, f4 vv) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = g[k] * grad_scale;
            mm[k] = mm[k] + (gk - mm[k]) * (1.0f - beta1);
            vv[k] = beta2 * vv[k] + (1.0f - beta2) * gk * gk;
            const float denom = sqrtf(vv[k]) * inv_bc2_sqrt + eps;
            pp[k] -= step_size * (mm[k] / denom);
        }
        if constexpr (NT) {
            __builtin_nontemporal_store(pp, reinterpret_cast<f4*>(p + 4 * i));
            __builtin_nontemporal_store(mm, reinterpret_cast<f4*>(m + 4 * i));
            __builtin_nontemporal_store(vv, reinterpret_cast<f4*>(v + 4 * i));
        } else {
            *reinterpret_cast<f4*>(p + 4 * i) = pp;
            *reinterpret_cast<f4*>(m + 4 * i) = mm;
            *reinterpret_cast<f4*>(v + 4 * i) = vv;
        }
        if (compute) {
            union { __hip_bfloat16 h[4]; uint2 u; } pk;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk.h[k] = __float2bfloat16(pp[k]);
            *reinterpret_cast<uint2*>(compute + 4 * i) = pk.u;
        }
    };
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const long j = i + stride;
        float g0[4], g1[4];
        load4<GT>(grad + 4 * i, g0);
        load4<GT>(grad + 4 * j, g1);
        const f4 p0 = ldq<NT>(reinterpret_cast<const f4*>(p + 4 * i));
        const f4 m0 = ldq<NT>(reinterpret_cast<const f4*>(m + 4 * i));
        const f4 v0 = ldq<NT>(reinterpret_cast<const f4*>(v + 4 * i));
        const f4 p1 = ldq<NT>(reinterpret_cast<const f4*>(p + 4 * j));
        const f4 m1 = ldq<NT>(reinterpret_cast<const f4*>(m + 4 * j));
        const f4 v1 = ldq<NT>(reinterpret_cast<const f4*>(v + 4 * j));
        update(i, g0, p0, m0, v0);
        update(j, g1, p1, m1, v1);
    }
    if (i < n4) {
        float g0[4];
        load4<GT>(grad + 4 * i, g0);
        update(i, g0, ldq<NT>(reinterpret_cast<const f4*>(p + 4 * i)),
               ldq<NT>(reinterpret_cast<const f4*>(m + 4 * i)),
               ldq<NT>(reinterpret_cast<const f4*>(v + 4 * i)));
    }
}

}  // namespace

namespace {
int adam_launch(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq, void* compute_bf16, long n,
                float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt, float grad_scale,
                const float* hyper, void* stream);
}

extern "C" int grit_adam_flat(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq,
                              void* compute_bf16, long n, float lr, float beta1, float beta2, float eps, float bias_correction1,
                              float bias_correction2_sqrt, float grad_scale, void* stream) {
    if (bias_correction1 <= 0.f || bias_correction2_sqrt <= 0.f) return GRIT_ERR_BAD_ARG;
    return adam_launch(param, grad, grad_is_bf16, exp_avg, exp_avg_sq, compute_bf16, n, lr, beta1, beta2, eps, bias_correction1,
                       bias_correction2_sqrt, grad_scale, nullptr, stream);
}

extern "C" int grit_adam_flat_dev(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq,
                                  void* compute_bf16, long n, float beta1, float beta2, float eps, float grad_scale,
                                  const float* hyper, void* stream) {
    if (!hyper || ((uintptr_t)hyper % 8)) return GRIT_ERR_BAD_ARG;
    return adam_launch(param, grad, grad_is_bf16, exp_avg, exp_avg_sq, compute_bf16, n, 0.f, beta1, beta2, eps, 1.f, 1.f, grad_scale,
                       hyper, stream);
}

namespace {
int adam_launch(float* param, const void* grad, int grad_is_bf16, float* exp_avg, float* exp_avg_sq, void* compute_bf16, long n,
                float lr, float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt, float grad_scale,
                const float* hyper, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0)
        return GRIT_ERR_BAD_ARG;
    const uintptr_t align = (uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq;
    if (n % 4 != 0 || (align % 16) != 0 || ((uintptr_t)grad % (grad_is_bf16 ? 8 : 16)) != 0 ||
        (compute_bf16 && ((uintptr_t)compute_bf16 % 8) != 0))
        return GRIT_ERR_UNSUPPORTED;
    const long n4 = n / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;  // grid-stride: 32 workgroups per CU
    const float step_size = lr / bias_correction1, inv_bc2_sqrt = 1.0f / bias_correction2_sqrt;
    // GRIT_ADAM_NT=0 (A/B): plain loads / stores of the masters and moments
    static const bool nt = !(getenv("GRIT_ADAM_NT") && atoi(getenv("GRIT_ADAM_NT")) == 0);
#define GRIT_ADAM_LAUNCH(GT_, NT_, GPTR_)                                                                                          \
    hipLaunchKernelGGL((adam_flat<GT_, NT_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, GPTR_, exp_avg,    \
                       exp_avg_sq, (__hip_bfloat16*)compute_bf16, n4, step_size, beta1, beta2, eps, inv_bc2_sqrt, grad_scale, hyper)
    if (grad_is_bf16) {
        if (nt) GRIT_ADAM_LAUNCH(__hip_bfloat16, true, (const __hip_bfloat16*)grad);
        else GRIT_ADAM_LAUNCH(__hip_bfloat16, false, (const __hip_bfloat16*)grad);
    } else {
        if (nt) GRIT_ADAM_LAUNCH(float, true, (const float*)grad);
        else GRIT_ADAM_LAUNCH(float, false, (const float*)grad);
    }
#undef GRIT_ADAM_LAUNCH
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
}  // namespace
