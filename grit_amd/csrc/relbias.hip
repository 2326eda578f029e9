// Relative-position bias of Swin window attention: bias[h][i][j] = table[index[i][j]][h]
// (reference models/common/swin_model.py:168-171: table[index.view(-1)].view(N, N, nH).permute(2, 0, 1)) and its gradient.
//
// The table is tiny ((2*12-1)^2 = 529 rows x heads) and the index fixed, but torch turns the lookup into a gather +
// permute copy + cast going forward and a sort-based index_put (or, posed as a one-hot GEMM, a 20 736-deep skinny GEMM
// at ~50 us) going backward, 24 times a step.  Forward: one thread per (i, j) position walks the heads, stores are
// coalesced over positions.  Backward: the positions that share a table row are known up front (index sorted once on
// the host: `order`, `offsets`), so one workgroup per table row adds its <= 144 entries of d(bias) per head -- no atomics,
// deterministic.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(__hip_bfloat16 v) { return __bfloat162float(v); }
__device__ __forceinline__ void from_f32(float* p, float v) { *p = v; }
__device__ __forceinline__ void from_f32(__hip_bfloat16* p, float v) { *p = __float2bfloat16(v); }

template <typename T>
__global__ __launch_bounds__(256)
void relbias_fwd(const T* __restrict__ table, const int64_t* __restrict__ index, int n_rows, int nH, int n_pos,
                 float* __restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pos) return;
    const int64_t r = index[p];
    if (r < 0 || r >= n_rows) return;  // malformed index: leave the entry untouched rather than read out of bounds
    const T* row = table + (size_t)r * nH;
    for (int h = 0; h < nH; ++h) out[(size_t)h * n_pos + p] = to_f32(row[h]);
}

// The gathers of many window-attention modules in ONE launch (24 Swin blocks: 24 dependent ~5 us launches otherwise): blockIdx.y = job.
struct RelbiasGroupArgs {
    grit_relbias_job job[GRIT_RELBIAS_GROUP_MAX];
};

__global__ __launch_bounds__(256)
void relbias_fwd_grouped(const RelbiasGroupArgs a) {
    const grit_relbias_job& jb = a.job[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= jb.n_pos) return;
    const int64_t r = jb.index[p];
    if (r < 0 || r >= jb.n_rows) return;
    if (jb.table_is_bf16) {
        const __hip_bfloat16* row = (const __hip_bfloat16*)jb.table + (size_t)r * jb.num_heads;
        for (int h = 0; h < jb.num_heads; ++h) jb.bias[(size_t)h * jb.n_pos + p] = to_f32(row[h]);
    } else {
        const float* row = (const float*)jb.table + (size_t)r * jb.num_heads;
        for (int h = 0; h < jb.num_heads; ++h) jb.bias[(size_t)h * jb.n_pos + p] = row[h];
    }
}

template <typename T>
__global__ __launch_bounds__(256)
void relbias_bwd(const float* __restrict__ dbias, const int32_t* __restrict__ order, const int32_t* __restrict__ offsets,
                 int nH, int n_pos, T* __restrict__ dtable) {
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e0 = offsets[r], e1 = offsets[r + 1];
    for (int h = wave; h < nH; h += 4) {
        float s = 0.f;
        for (int e = e0 + lane; e < e1; e += 64) s += dbias[(size_t)h * n_pos + order[e]];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) from_f32(dtable + (size_t)r * nH + h, s);
    }
}

// The table gradients of many window-attention modules in ONE launch: blockIdx.y = job, blockIdx.x = table row.
struct RelbiasBwdGroupArgs {
    grit_relbias_bwd_job job[GRIT_RELBIAS_GROUP_MAX];
};

__global__ __launch_bounds__(256)
void relbias_bwd_grouped(const RelbiasBwdGroupArgs a) {
    const grit_relbias_bwd_job& jb = a.job[blockIdx.y];
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (r >= jb.n_rows) return;
    const int e0 = jb.offsets[r], e1 = jb.offsets[r + 1];
    for (int h = wave; h < jb.num_heads; h += 4) {
        float s = 0.f;
        for (int e = e0 + lane; e < e1; e += 64) s += jb.dbias[(size_t)h * jb.n_pos + jb.order[e]];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) {
            if (jb.table_is_bf16) from_f32((__hip_bfloat16*)jb.dtable + (size_t)r * jb.num_heads + h, s);
            else from_f32((float*)jb.dtable + (size_t)r * jb.num_heads + h, s);
        }
    }
}

}  // namespace

extern "C" {

int grit_relbias_fwd(const void* table, const int64_t* index, int n_rows, int num_heads, int n_pos, int table_is_bf16,
                     float* bias, void* stream) {
    if (!table || !index || !bias || n_rows <= 0 || num_heads <= 0 || n_pos <= 0) return GRIT_ERR_BAD_ARG;
    const dim3 grid((n_pos + 255) / 256), block(256);
    if (table_is_bf16)
        hipLaunchKernelGGL(relbias_fwd<__hip_bfloat16>, grid, block, 0, (hipStream_t)stream, (const __hip_bfloat16*)table, index,
                           n_rows, num_heads, n_pos, bias);
    else
        hipLaunchKernelGGL(relbias_fwd<float>, grid, block, 0, (hipStream_t)stream, (const float*)table, index, n_rows,
                           num_heads, n_pos, bias);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_relbias_fwd_grouped(const grit_relbias_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GRIT_RELBIAS_GROUP_MAX) return GRIT_ERR_BAD_ARG;
    RelbiasGroupArgs a;
    int max_pos = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const grit_relbias_job& jb = jobs[j];
        if (!jb.table || !jb.index || !jb.bias || jb.n_rows <= 0 || jb.num_heads <= 0 || jb.n_pos <= 0) return GRIT_ERR_BAD_ARG;
        a.job[j] = jb;
        if (jb.n_pos > max_pos) max_pos = jb.n_pos;
    }
    hipLaunchKernelGGL(relbias_fwd_grouped, dim3((max_pos + 255) / 256, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_relbias_bwd_grouped(const grit_relbias_bwd_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GRIT_RELBIAS_GROUP_MAX) return GRIT_ERR_BAD_ARG;
    RelbiasBwdGroupArgs a;
    int max_rows = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const grit_relbias_bwd_job& jb = jobs[j];
        if (!jb.dbias || !jb.order || !jb.offsets || !jb.dtable || jb.n_rows <= 0 || jb.num_heads <= 0 || jb.n_pos <= 0) return GRIT_ERR_BAD_ARG;
        a.job[j] = jb;
        if (jb.n_rows > max_rows) max_rows = jb.n_rows;
    }
    hipLaunchKernelGGL(relbias_bwd_grouped, dim3(max_rows, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

int grit_relbias_bwd(const float* dbias, const int32_t* order, const int32_t* offsets, int n_rows, int num_heads, int n_pos,
                     int table_is_bf16, void* dtable, void* stream) {
    if (!dbias || !order || !offsets || !dtable || n_rows <= 0 || num_heads <= 0 || n_pos <= 0) return GRIT_ERR_BAD_ARG;
    const dim3 grid(n_rows), block(256);
    if (table_is_bf16)
        hipLaunchKernelGGL(relbias_bwd<__hip_bfloat16>, grid, block, 0, (hipStream_t)stream, dbias, order, offsets, num_heads,
                           n_pos, (__hip_bfloat16*)dtable);
    else
        hipLaunchKernelGGL(relbias_bwd<float>, grid, block, 0, (hipStream_t)stream, dbias, order, offsets, num_heads, n_pos,
                           (float*)dtable);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // extern "C"
