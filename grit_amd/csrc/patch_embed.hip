// Patch embedding of the Swin backbone as ONE pass (gfx950):
//
//     tokens[b, hh * Wp + ww, :] = LayerNorm_C( conv4x4/4(img)[b, :, hh, ww] + bias )        C = 96 / 128 / 192 channels
//
// reference: models/common/swin_model.py:336-365 (PatchEmbed: Conv2d(3, C, kernel 4, stride 4) -> flatten -> LayerNorm), frozen in
// GRIT's training configuration (frozen_stages = 2), so only the forward exists.  The unfused path is four passes over the maps:
// the fp32 -> bf16 cast of the image, the im2col permute copy, the K = 48 GEMM (+ bias) and the LayerNorm (0.3 ms per step at 32 x
// 640 x 640); here the image is read once (157 MB as fp32) and the 210 MB token map written once.
//   * a wave owns 16 horizontally adjacent patches (one MFMA tile of tokens); lane (t = lane & 15, g = lane >> 4) fetches the k-slots
//     8 g .. 8 g + 7 of patch t straight from the image -- k = c * 16 + kh * 4 + kw is the conv weight's own (c, kh, kw) order, so a
//     slot group is two 16-byte rows (kh = 2 (g & 1), + 1) of channel g >> 1: per (c, kh) the 16 patches read 256 contiguous bytes;
//   * v_mfma_f32_16x16x32_bf16 with the WEIGHT rows as the A operand (K padded 48 -> 64 with zeros): a lane ends up with channels
//     16 j + 4 g + r (j < C / 16, r < 4) of token t -- four consecutive channels per accumulator quad, 8-byte stores;
//   * the conv output is rounded to bf16 (what the unfused GEMM stores), LayerNorm statistics over the rounded values in fp32:
//     a token's C channels live in the four lanes t, t + 16, t + 32, t + 48 -> two cross-lane adds per statistic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_round(float v) { return (float)(__bf16)v; }

template <typename IMG> __device__ __forceinline__ void load4(const IMG* p, float (&f)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&f)[4]) {
    const v4f v = *reinterpret_cast<const v4f*>(p);
    f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
template <> __device__ __forceinline__ void load4<__bf16>(const __bf16* p, float (&f)[4]) {
    const v4bf v = *reinterpret_cast<const v4bf*>(p);
    f[0] = (float)v[0]; f[1] = (float)v[1]; f[2] = (float)v[2]; f[3] = (float)v[3];
}

// NT = C / 16 channel tiles.  img [B, 3, H, W] (H, W multiples of 4, W / 4 a multiple of 16), w [C, 48] bf16, out [B, H/4 * W/4, C]
template <typename IMG, int NT>
__global__ __launch_bounds__(256)
void patch_embed_ln(const IMG* __restrict__ img, const __bf16* __restrict__ w, const __bf16* __restrict__ bias,
                    const __bf16* __restrict__ gamma, const __bf16* __restrict__ beta, float eps, int B, int H, int W,
                    __bf16* __restrict__ out, int ntiles) {
    constexpr int C = 16 * NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = lane & 15, g = lane >> 4;
    // weight fragments (A operand): rows = channels 16 j + t, k-slots 8 g .. (+ 32 for the second half; k >= 48 is zero padding)
    v8bf wf[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        wf[j][0] = *reinterpret_cast<const v8bf*>(w + (size_t)(16 * j + t) * 48 + 8 * g);
        v8bf z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
        wf[j][1] = g < 2 ? *reinterpret_cast<const v8bf*>(w + (size_t)(16 * j + t) * 48 + 32 + 8 * g) : z;
    }
    // per-lane channel constants: channels 16 j + 4 g + r
    v4bf bq[NT], gq[NT], eq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        bq[j] = *reinterpret_cast<const v4bf*>(bias + 16 * j + 4 * g);
        gq[j] = *reinterpret_cast<const v4bf*>(gamma + 16 * j + 4 * g);
        eq[j] = *reinterpret_cast<const v4bf*>(beta + 16 * j + 4 * g);
    }
    const int Wp = W >> 2, Hp = H >> 2, tiles_per_row = Wp >> 4;
    const size_t plane = (size_t)H * W;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const int row = tile / tiles_per_row, tw = tile - row * tiles_per_row;  // row = b * Hp + hh
        const int b = row / Hp, hh = row - b * Hp;
        const int ww = 16 * tw + t;
        // patch fragments (B operand): k-slots 8 g ..: channel g >> 1, rows kh = 2 (g & 1) and + 1, the four kw of each
        const IMG* p0 = img + ((size_t)b * 3 + (g >> 1)) * plane + (size_t)(4 * hh + 2 * (g & 1)) * W + 4 * ww;
        float f0[4], f1[4], f2[4] = {0.f, 0.f, 0.f, 0.f}, f3[4] = {0.f, 0.f, 0.f, 0.f};
        load4<IMG>(p0, f0);
        load4<IMG>(p0 + W, f1);
        if (g < 2) {  // k-slots 32 + 8 g ..: channel 2, rows kh = 2 g and + 1
            const IMG* p2 = img + ((size_t)b * 3 + 2) * plane + (size_t)(4 * hh + 2 * g) * W + 4 * ww;
            load4<IMG>(p2, f2);
            load4<IMG>(p2 + W, f3);
        }
        v8bf pf0, pf1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            pf0[e] = (__bf16)f0[e]; pf0[4 + e] = (__bf16)f1[e];
            pf1[e] = (__bf16)f2[e]; pf1[4 + e] = (__bf16)f3[e];
        }
        v4f acc[NT];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][0], pf0, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][1], pf1, acc[j], 0, 0, 0);
        }
        // acc[j][r] = conv[channel 16 j + 4 g + r][token t]: + bias, rounded to bf16 as the unfused GEMM stores it
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[j][r] = bf16_round(acc[j][r] + (float)bq[j][r]);
                s += acc[j][r];
            }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mu = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = acc[j][r] - mu;
                q = fmaf(d, d, q);
            }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rs = rsqrtf(q * (1.0f / C) + eps);
        __bf16* o = out + ((size_t)row * Wp + ww) * C + 4 * g;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            v4bf y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (__bf16)fmaf((acc[j][r] - mu) * rs, (float)gq[j][r], (float)eq[j][r]);
            __builtin_nontemporal_store(y, reinterpret_cast<v4bf*>(o + 16 * j));
        }
    }
}

}  // namespace

extern "C" int grit_patch_embed_ln_fwd(const void* img, int img_is_bf16, int B, int H, int W, int C, const void* weight,
                                       const void* bias, const void* gamma, const void* beta, float eps, void* out, void* stream) {
    if (!img || !weight || !bias || !gamma || !beta || !out || B <= 0 || H <= 0 || W <= 0) return GRIT_ERR_BAD_ARG;
    if (H % 4 || W % 64 || (C != 96 && C != 128 && C != 192)) return GRIT_ERR_UNSUPPORTED;  // 16 patches per wave tile
    if (((uintptr_t)img | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)out) & 15) return GRIT_ERR_UNSUPPORTED;
    const long long ntiles = (long long)B * (H / 4) * (W / 64);
    if (ntiles > 0x7fffffffLL) return GRIT_ERR_UNSUPPORTED;
    const int blocks = (int)(ntiles < 4 * 2048 ? (ntiles + 3) / 4 : 2048);
#define GRIT_PE_LAUNCH(IMG_, NT_)                                                                                          \
    hipLaunchKernelGGL((patch_embed_ln<IMG_, NT_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const IMG_*)img,    \
                       (const __bf16*)weight, (const __bf16*)bias, (const __bf16*)gamma, (const __bf16*)beta, eps, B, H, W, \
                       (__bf16*)out, (int)ntiles)
    if (img_is_bf16) {
        if (C == 96) GRIT_PE_LAUNCH(__bf16, 6); else if (C == 128) GRIT_PE_LAUNCH(__bf16, 8); else GRIT_PE_LAUNCH(__bf16, 12);
    } else {
        if (C == 96) GRIT_PE_LAUNCH(float, 6); else if (C == 128) GRIT_PE_LAUNCH(float, 8); else GRIT_PE_LAUNCH(float, 12);
    }
#undef GRIT_PE_LAUNCH
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
