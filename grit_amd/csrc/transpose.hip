// Transposed bf16 copies of up to GRIT_TRANSPOSE_GROUP_MAX weight matrices in one launch.
//
// Why it exists: the input gradient of a Swin Mlp's fc2 runs through the fused GELU' GEMM (gemm.hip), whose B operand is K-contiguous,
// i.e. it needs fc2.weight^T ([4C, C]) -- autograd's Linear backward (davidnvq/grit models/common/swin_model.py:31-37 through
// torch.nn.functional.linear) leaves that transpose to the GEMM library.  As `w.t().contiguous()` inside every block's backward it
// is 24 dependent ~10 us launches per training step on the critical path; the weights only change at the optimizer step, so the 24
// copies are made by ONE launch at the start of the forward pass (grit_amd/ops/transposed.py).
//
// 64 x 64 tiles through LDS (pitch 66 elements: the column reads of the transposed store hit distinct banks), 256 threads, 16-byte
// global accesses on both sides; block -> (job, tile) through a prefix table passed by value.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

struct TransposeArgs {
    grit_transpose_job job[GRIT_TRANSPOSE_GROUP_MAX];
    unsigned first_block[GRIT_TRANSPOSE_GROUP_MAX + 1];
    int n_jobs;
};

__global__ __launch_bounds__(256)
void transpose_grouped_kernel(const TransposeArgs a) {
    __shared__ unsigned short tile[64][66];
    int j = 0;
    while (j + 1 < a.n_jobs && blockIdx.x >= a.first_block[j + 1]) ++j;
    const grit_transpose_job jb = a.job[j];
    const unsigned t = blockIdx.x - a.first_block[j];
    const int tiles_c = jb.cols / 64;
    const int r0 = (int)(t / tiles_c) * 64, c0 = (int)(t % tiles_c) * 64;
    const unsigned short* src = reinterpret_cast<const unsigned short*>(jb.src);
    unsigned short* dst = reinterpret_cast<unsigned short*>(jb.dst);
    const int tid = threadIdx.x;
    // load: 64 rows x 8 chunks of 8 elements; thread -> (row = tid / 8 + 32 k, chunk = tid % 8)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = (tid >> 3) + 32 * k, ch = tid & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)(r0 + r) * jb.cols + c0 + ch * 8);
        const unsigned short* e = reinterpret_cast<const unsigned short*>(&v);
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[r][ch * 8 + i] = e[i];
    }
    __syncthreads();
    // store: output row = source column; thread -> (out row = tid / 8 + 32 k, chunk of 8 source rows)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int oc = (tid >> 3) + 32 * k, ch = tid & 7;
        unsigned short e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = tile[ch * 8 + i][oc];
        *reinterpret_cast<uint4*>(dst + (size_t)(c0 + oc) * jb.rows + r0 + ch * 8) = *reinterpret_cast<const uint4*>(e);
    }
}

}  // namespace

extern "C" int grit_transpose_bf16_grouped(const grit_transpose_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GRIT_TRANSPOSE_GROUP_MAX) return GRIT_ERR_BAD_ARG;
    TransposeArgs a;
    a.n_jobs = n_jobs;
    unsigned long long total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const grit_transpose_job& jb = jobs[j];
        if (!jb.src || !jb.dst || jb.rows <= 0 || jb.cols <= 0) return GRIT_ERR_BAD_ARG;
        if (jb.rows % 64 || jb.cols % 64 || ((uintptr_t)jb.src % 16) || ((uintptr_t)jb.dst % 16)) return GRIT_ERR_UNSUPPORTED;
        a.job[j] = jb;
        a.first_block[j] = (unsigned)total;
        total += (unsigned long long)(jb.rows / 64) * (unsigned)(jb.cols / 64);
        if (total > 0x7fffffffULL) return GRIT_ERR_UNSUPPORTED;
    }
    a.first_block[n_jobs] = (unsigned)total;
    hipLaunchKernelGGL(transpose_grouped_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
