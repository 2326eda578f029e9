// Column sums of a row-major [M, N] matrix (bf16 or f32) into f32 [N]: the bias gradient of a Linear layer,
// db = sum_rows(dY).  torch's generic reduce kernel runs this at ~2 TB/s on the Swin shapes (M = 51 200 .. 204 800,
// N = 256 .. 4096; 215 launches and 5.9 ms per training step, profiles/r01); it is a pure HBM stream.
//
// Layout: a workgroup of 256 threads owns a 512-column strip (64 lanes x 8 columns, one 16-byte load per lane) and a
// slab of rows; its 4 waves walk the slab row-interleaved with 4 independent loads in flight per lane, fold through
// LDS, and write one partial row per (slab, strip) into a workspace that the caller sums over slabs (<= 64 rows).
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

__device__ __forceinline__ void load8(const __hip_bfloat16* p, float (&f)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ void load8(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

template <typename T>
__device__ __forceinline__ void colsum_body(const T* __restrict__ x, size_t ld, int M, int N, int rows_per_slab,
                                            float* __restrict__ partial, unsigned bx, unsigned by, float (*red)[512]);

template <typename T>
__global__ __launch_bounds__(256)
void colsum_kernel(const T* __restrict__ x, int M, int N, int rows_per_slab, float* __restrict__ partial) {
    __shared__ float red[4][512];
    colsum_body<T>(x, (size_t)N, M, N, rows_per_slab, partial, blockIdx.x, blockIdx.y, red);
}

// Column sums of several bf16 matrices in ONE launch (job table by value): the bias gradients of the decoders' short-map Linears
// whose weight gradients go through grit_wgrad_tn_grouped.
struct ColsumGroupArgs {
    grit_colsum_job job[GRIT_COLSUM_GROUP_MAX];
    unsigned first_block[GRIT_COLSUM_GROUP_MAX + 1];
    unsigned strips[GRIT_COLSUM_GROUP_MAX];
    int n_jobs;
};

__global__ __launch_bounds__(256)
void colsum_grouped_kernel(const ColsumGroupArgs a) {
    __shared__ float red[4][512];
    int j = 0;
    while (j + 1 < a.n_jobs && blockIdx.x >= a.first_block[j + 1]) ++j;
    const unsigned local = blockIdx.x - a.first_block[j];
    const unsigned by = local / a.strips[j], bx = local - by * a.strips[j];
    const grit_colsum_job& jb = a.job[j];
    colsum_body<__hip_bfloat16>((const __hip_bfloat16*)jb.x, (size_t)jb.ld, jb.M, jb.N, (jb.M + jb.slabs - 1) / jb.slabs, jb.partial, bx, by, red);
}

template <typename T>
__device__ __forceinline__ void colsum_body(const T* __restrict__ x, size_t ld, int M, int N, int rows_per_slab,
                                            float* __restrict__ partial, unsigned bx, unsigned by, float (*red)[512]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = bx * 512 + lane * 8;
    const int r0 = by * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        int r = r0 + wave;
        for (; r + 28 < r1; r += 32) {  // 8 rows of this wave in flight
            float a[8], b[8], c[8], d[8], e[8], f[8], g[8], h[8];
            load8(x + (size_t)r * ld + col, a);
            load8(x + (size_t)(r + 4) * ld + col, b);
            load8(x + (size_t)(r + 8) * ld + col, c);
            load8(x + (size_t)(r + 12) * ld + col, d);
            load8(x + (size_t)(r + 16) * ld + col, e);
            load8(x + (size_t)(r + 20) * ld + col, f);
            load8(x + (size_t)(r + 24) * ld + col, g);
            load8(x + (size_t)(r + 28) * ld + col, h);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += ((a[i] + b[i]) + (c[i] + d[i])) + ((e[i] + f[i]) + (g[i] + h[i]));
        }
        for (; r + 12 < r1; r += 16) {  // 4 rows of this wave in flight
            float a[8], b[8], c[8], d[8];
            load8(x + (size_t)r * ld + col, a);
            load8(x + (size_t)(r + 4) * ld + col, b);
            load8(x + (size_t)(r + 8) * ld + col, c);
            load8(x + (size_t)(r + 12) * ld + col, d);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += (a[i] + b[i]) + (c[i] + d[i]);
        }
        for (; r < r1; r += 4) {
            float a[8];
            load8(x + (size_t)r * ld + col, a);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += a[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave][lane * 8 + i] = acc[i];
    __syncthreads();
    for (int j = threadIdx.x; j < 512; j += 256) {
        const int c = bx * 512 + j;
        if (c < N) partial[(size_t)by * N + c] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    }
}

// Second stage of every slab-wise reduction in the step (column sums, split-M weight gradients, LayerNorm dgamma/dbeta):
// out[g][i] = cast(sum_s partial[g][s][i]).  One launch replaces torch's reduce + dtype-cast pair (~350 launches a step).
// A workgroup covers CW float4 column groups x (256 / CW) interleaved slab lanes, 4 slabs in flight per thread, LDS fold.
// Wide outputs with a moderate number of slabs (the weight-gradient slices: 16-85 slabs of 0.25-1 M elements): a thread owns one
// float4 column group and walks ALL slabs, eight 16-byte loads in flight -- no LDS fold, no barrier.  (The shape below, 4 loads in
// flight and an LDS fold, ran these sums at ~2 TB/s inside the training step.)
template <typename OT>
__device__ __forceinline__ void slab_sum_wide(const float* __restrict__ partial, long group_stride, int slabs, long n,
                                              OT* __restrict__ out, unsigned bx, unsigned by, const float* __restrict__ extra) {
    const long col = ((long)bx * 256 + threadIdx.x) * 4;
    if (col >= n) return;
    const float* src = partial + (long)by * group_stride + col;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc4 = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 7 < slabs; s += 8) {  // read once: streamed past the caches' retention (nontemporal)
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + (long)(s + u) * n));
#pragma unroll
        for (int u = 0; u < 8; ++u) acc4 += v[u];
    }
    for (; s < slabs; ++s) acc4 += __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + (long)s * n));
    if (extra) acc4 += *reinterpret_cast<const f4*>(extra + (long)by * n + col);  // one more term, added last
    const float4 acc = make_float4(acc4[0], acc4[1], acc4[2], acc4[3]);
    OT* dst = out + (long)by * n + col;
    if constexpr (sizeof(OT) == 4) {
        *reinterpret_cast<float4*>(dst) = acc;
    } else {
        union { __hip_bfloat16 h[4]; uint2 u; } pk;
        pk.h[0] = __float2bfloat16(acc.x); pk.h[1] = __float2bfloat16(acc.y);
        pk.h[2] = __float2bfloat16(acc.z); pk.h[3] = __float2bfloat16(acc.w);
        *reinterpret_cast<uint2*>(dst) = pk.u;
    }
}

constexpr int kWideMark = 31;  // cw_log2 value that selects slab_sum_wide

template <typename OT>
__device__ __forceinline__ void slab_sum_body(const float* __restrict__ partial, long group_stride, int slabs, long n, int cw_log2,
                                              OT* __restrict__ out, unsigned bx, unsigned by, float4* red,
                                              const float* __restrict__ extra = nullptr) {
    if (cw_log2 == kWideMark) {  // workgroup-uniform
        slab_sum_wide<OT>(partial, group_stride, slabs, n, out, bx, by, extra);
        return;
    }
    const int cw = 1 << cw_log2, sg_count = 256 >> cw_log2;
    const int c = threadIdx.x & (cw - 1), sg = threadIdx.x >> cw_log2;
    const long col = ((long)bx * cw + c) * 4;
    const float* src = partial + (long)by * group_stride + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < n) {
        int s = sg;
        for (; s + 3 * sg_count < slabs; s += 4 * sg_count) {
            const float4 a = *reinterpret_cast<const float4*>(src + (long)s * n);
            const float4 b = *reinterpret_cast<const float4*>(src + (long)(s + sg_count) * n);
            const float4 d = *reinterpret_cast<const float4*>(src + (long)(s + 2 * sg_count) * n);
            const float4 e = *reinterpret_cast<const float4*>(src + (long)(s + 3 * sg_count) * n);
            acc.x += (a.x + b.x) + (d.x + e.x); acc.y += (a.y + b.y) + (d.y + e.y);
            acc.z += (a.z + b.z) + (d.z + e.z); acc.w += (a.w + b.w) + (d.w + e.w);
        }
        for (; s < slabs; s += sg_count) {
            const float4 a = *reinterpret_cast<const float4*>(src + (long)s * n);
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (sg == 0 && col < n) {
        for (int j = 1; j < sg_count; ++j) {
            const float4 a = red[(j << cw_log2) + c];
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
        }
        if (extra) {  // one more term, added last
            const float4 e = *reinterpret_cast<const float4*>(extra + (long)by * n + col);
            acc.x += e.x; acc.y += e.y; acc.z += e.z; acc.w += e.w;
        }
        OT* dst = out + (long)by * n + col;
        if constexpr (sizeof(OT) == 4) {
            *reinterpret_cast<float4*>(dst) = acc;
        } else {
            union { __hip_bfloat16 h[4]; uint2 u; } pk;
            pk.h[0] = __float2bfloat16(acc.x); pk.h[1] = __float2bfloat16(acc.y);
            pk.h[2] = __float2bfloat16(acc.z); pk.h[3] = __float2bfloat16(acc.w);
            *reinterpret_cast<uint2*>(dst) = pk.u;
        }
    }
}

template <typename OT>
__global__ __launch_bounds__(256)
void slab_sum_kernel(const float* __restrict__ partial, long group_stride, int slabs, long n, int cw_log2, OT* __restrict__ out) {
    __shared__ float4 red[256];
    slab_sum_body<OT>(partial, group_stride, slabs, n, cw_log2, out, blockIdx.x, blockIdx.y, red);
}

// Several slab sums in ONE launch: a backward node that produced k sets of partials (split-M weight gradients, bias-gradient
// column sums, LayerNorm dgamma / dbeta, the GELU-epilogue column sums) reduces them together instead of with k dependent
// launches at the ~6 us launch floor each.  The job table travels by value in the kernel arguments (no device table to fill).
struct GroupedArgs {
    grit_slab_job job[GRIT_SLAB_GROUP_MAX];
    unsigned first_block[GRIT_SLAB_GROUP_MAX + 1];  // workgroups [first_block[j], first_block[j + 1]) belong to job j
    unsigned blocks_x[GRIT_SLAB_GROUP_MAX];         // column blocks per group of job j
    int cw_log2[GRIT_SLAB_GROUP_MAX];
    int n_jobs;
};

__global__ __launch_bounds__(256)
void slab_sum_grouped_kernel(const GroupedArgs a) {
    __shared__ float4 red[256];
    int j = 0;
    while (j + 1 < a.n_jobs && blockIdx.x >= a.first_block[j + 1]) ++j;
    const unsigned local = blockIdx.x - a.first_block[j];
    const unsigned by = local / a.blocks_x[j], bx = local - by * a.blocks_x[j];
    const grit_slab_job& jb = a.job[j];
    if (jb.out_is_bf16)
        slab_sum_body<__hip_bfloat16>(jb.partial, jb.group_stride, jb.slabs, jb.n, a.cw_log2[j], (__hip_bfloat16*)jb.out, bx, by, red, jb.extra);
    else
        slab_sum_body<float>(jb.partial, jb.group_stride, jb.slabs, jb.n, a.cw_log2[j], (float*)jb.out, bx, by, red, jb.extra);
}

// column width (log2 of float4 groups per workgroup row) for a job: see grit_slab_sum
int pick_cw_log2(long groups4, int groups, int slabs) {
    static const bool wide_ok = !(getenv("GRIT_SLAB_WIDE") && atoi(getenv("GRIT_SLAB_WIDE")) == 0);
    // >= 64 workgroups of 256 column groups.  Round 5: also for 1-3 slabs (the decoders' deferred weight gradients: 2 slices of a
    // 512 x 512 problem) -- the narrow shape below gives such a job 1 024 workgroups of which half the threads load one float4 each
    // (99 + 87 us for two launches that move 160 MB, profiles/r05/decoder_phase_sequence.txt); same summation order either way
    if (wide_ok && groups4 >= 16384 && slabs <= 512) return kWideMark;
    int cw_log2 = 0;
    while (cw_log2 < 6 && (1L << cw_log2) < groups4) ++cw_log2;
    // tall, narrow partials (LayerNorm: 1024 slabs x 512 columns) would leave the chip to a handful of workgroups that
    // each walk hundreds of slabs: trade column width (>= 128 contiguous bytes per slab row) for slab lanes
    while (cw_log2 > 3 && (((groups4 + (1L << cw_log2) - 1) >> cw_log2) * groups) < 256 && slabs >= 8 * (256 >> (cw_log2 - 1)))
        --cw_log2;
    return cw_log2;
}

}  // namespace

extern "C" int grit_slab_sum_grouped(const grit_slab_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GRIT_SLAB_GROUP_MAX) return GRIT_ERR_BAD_ARG;
    GroupedArgs a;
    a.n_jobs = n_jobs;
    unsigned long long total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const grit_slab_job& jb = jobs[j];
        if (!jb.partial || !jb.out || jb.groups <= 0 || jb.slabs <= 0 || jb.n <= 0 || jb.group_stride < 0) return GRIT_ERR_BAD_ARG;
        if (jb.n % 4 != 0 || jb.group_stride % 4 != 0 || ((uintptr_t)jb.partial % 16) != 0 || ((uintptr_t)jb.out % 8) != 0)
            return GRIT_ERR_UNSUPPORTED;
        if (!jb.out_is_bf16 && ((uintptr_t)jb.out % 16) != 0) return GRIT_ERR_UNSUPPORTED;
        if (((uintptr_t)jb.extra % 16) != 0) return GRIT_ERR_UNSUPPORTED;
        const long groups4 = jb.n / 4;
        a.job[j] = jb;
        a.cw_log2[j] = pick_cw_log2(groups4, jb.groups, jb.slabs);
        const long bx = a.cw_log2[j] == kWideMark ? (groups4 + 255) / 256 : (groups4 + (1L << a.cw_log2[j]) - 1) >> a.cw_log2[j];
        a.blocks_x[j] = (unsigned)bx;
        a.first_block[j] = (unsigned)total;
        total += (unsigned long long)bx * (unsigned)jb.groups;
        if (total > 0x7fffffffULL) return GRIT_ERR_UNSUPPORTED;
    }
    a.first_block[n_jobs] = (unsigned)total;
    hipLaunchKernelGGL(slab_sum_grouped_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_slab_sum(const float* partial, int groups, long group_stride, int slabs, long n, void* out,
                             int out_is_bf16, void* stream) {
    if (!partial || !out || groups <= 0 || slabs <= 0 || n <= 0 || group_stride < 0) return GRIT_ERR_BAD_ARG;
    if (n % 4 != 0 || group_stride % 4 != 0 || ((uintptr_t)partial % 16) != 0 || ((uintptr_t)out % 16) != 0 || groups > 65535)
        return GRIT_ERR_UNSUPPORTED;
    const long groups4 = n / 4;
    const int cw_log2 = pick_cw_log2(groups4, groups, slabs);
    const long blocks = cw_log2 == kWideMark ? (groups4 + 255) / 256 : (groups4 + (1L << cw_log2) - 1) >> cw_log2;
    if (blocks > 0x7fffffffL) return GRIT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)blocks, groups), block(256);
    if (out_is_bf16)
        hipLaunchKernelGGL(slab_sum_kernel<__hip_bfloat16>, grid, block, 0, (hipStream_t)stream, partial, group_stride, slabs, n,
                           cw_log2, (__hip_bfloat16*)out);
    else
        hipLaunchKernelGGL(slab_sum_kernel<float>, grid, block, 0, (hipStream_t)stream, partial, group_stride, slabs, n, cw_log2,
                           (float*)out);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_colsum(const void* x, int M, int N, int x_is_bf16, int slabs, float* partial, void* stream) {
    if (!x || !partial || M <= 0 || N <= 0 || slabs <= 0 || slabs > GRIT_COLSUM_MAX_SLABS) return GRIT_ERR_BAD_ARG;
    if (N % 8 != 0 || ((uintptr_t)x % 16) != 0) return GRIT_ERR_UNSUPPORTED;
    const int rows_per_slab = (M + slabs - 1) / slabs;
    const dim3 grid((N + 511) / 512, slabs), block(256);
    if (x_is_bf16)
        hipLaunchKernelGGL(colsum_kernel<__hip_bfloat16>, grid, block, 0, (hipStream_t)stream, (const __hip_bfloat16*)x, M, N,
                           rows_per_slab, partial);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)x, M, N, rows_per_slab,
                           partial);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_colsum_grouped(const grit_colsum_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GRIT_COLSUM_GROUP_MAX) return GRIT_ERR_BAD_ARG;
    ColsumGroupArgs a;
    a.n_jobs = n_jobs;
    unsigned long long total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const grit_colsum_job& jb = jobs[j];
        if (!jb.x || !jb.partial || jb.M <= 0 || jb.N <= 0 || jb.slabs <= 0 || jb.slabs > GRIT_COLSUM_MAX_SLABS || jb.ld < jb.N)
            return GRIT_ERR_BAD_ARG;
        if (jb.N % 8 || jb.ld % 8 || ((uintptr_t)jb.x % 16)) return GRIT_ERR_UNSUPPORTED;
        a.job[j] = jb;
        a.strips[j] = (unsigned)((jb.N + 511) / 512);
        a.first_block[j] = (unsigned)total;
        total += (unsigned long long)a.strips[j] * (unsigned)jb.slabs;
        if (total > 0x7fffffffULL) return GRIT_ERR_UNSUPPORTED;
    }
    a.first_block[n_jobs] = (unsigned)total;
    hipLaunchKernelGGL(colsum_grouped_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
