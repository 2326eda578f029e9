// Beam-search candidate selection: the `k` best of every row of a [rows, n] float32 matrix, best first.
//
// Replaces what davidnvq/grit models/caption/transformer.py:184-188 (`select`) does with a full descending torch.sort over
// beam * vocabulary = 51 005 candidates per image and step -- and the torch.topk this build used in round 1, whose multi-block
// radix path cannot be replayed from a captured HIP graph on this stack (memory access fault on the second replay once eager
// allocations happen in between, tools/micro/dbg_decode.py topk).  One workgroup per row: every thread keeps the k best of its
// strided share in registers (sorted insertion), the 256 sorted lists are merged by k rounds of a workgroup arg-max over the list
// heads.  Order: value descending; equal values by ascending index (torch leaves the order of ties unspecified); NaN ranks above
// every number, as in torch.  HBM-bound: the row is read once, 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

constexpr int kThreads = 256, kMaxK = 8;

__device__ __forceinline__ bool better(float a, int ia, float b, int ib) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return an && (!bn || ia < ib);
    return a > b || (a == b && ia < ib);
}

__global__ __launch_bounds__(kThreads)
void topk_rows(const float* __restrict__ x, long ld, int n, int k, int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
    __shared__ float sv[kThreads];
    __shared__ int si[kThreads];
    __shared__ int sw[kThreads];
    const float* row = x + (size_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float bv[kMaxK];
    int bi[kMaxK];
#pragma unroll
    for (int j = 0; j < kMaxK; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
    auto push = [&](float v, int i) {
        if (!better(v, i, bv[kMaxK - 1], bi[kMaxK - 1])) return;
        bv[kMaxK - 1] = v; bi[kMaxK - 1] = i;
#pragma unroll
        for (int j = kMaxK - 1; j > 0; --j)
            if (better(bv[j], bi[j], bv[j - 1], bi[j - 1])) {
                const float tv = bv[j]; bv[j] = bv[j - 1]; bv[j - 1] = tv;
                const int ti = bi[j]; bi[j] = bi[j - 1]; bi[j - 1] = ti;
            }
    };
    // 16-byte loads where the row allows it
    const int head = (int)(((16 - ((uintptr_t)row & 15)) & 15) >> 2);
    const int lead = head < n ? head : n;
    if (tid < lead) push(row[tid], tid);
    const int n4 = (n - lead) >> 2;
    const float4* row4 = reinterpret_cast<const float4*>(row + lead);
    for (int i = tid; i < n4; i += kThreads) {
        const float4 v = row4[i];
        const int base = lead + 4 * i;
        push(v.x, base); push(v.y, base + 1); push(v.z, base + 2); push(v.w, base + 3);
    }
    for (int i = lead + 4 * n4 + tid; i < n; i += kThreads) push(row[i], i);
    // merge: k rounds of arg-max over the heads of the per-thread sorted lists
    int headp = 0;
    for (int r = 0; r < k; ++r) {
        float v = -INFINITY; int i = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < kMaxK; ++j)
            if (j == headp) { v = bv[j]; i = bi[j]; }
        sv[tid] = v; si[tid] = i; sw[tid] = tid;
        __syncthreads();
        for (int s = kThreads / 2; s > 0; s >>= 1) {
            if (tid < s && better(sv[tid + s], si[tid + s], sv[tid], si[tid])) {
                sv[tid] = sv[tid + s]; si[tid] = si[tid + s]; sw[tid] = sw[tid + s];
            }
            __syncthreads();
        }
        const int winner = sw[0];
        if (tid == 0) {
            idx_out[(size_t)blockIdx.x * k + r] = si[0];
            val_out[(size_t)blockIdx.x * k + r] = sv[0];
        }
        if (tid == winner) ++headp;
        __syncthreads();
    }
}

}  // namespace

extern "C" int grit_topk_rows_f32(const float* x, long ld, int rows, int n, int k, int64_t* idx_out, float* val_out, void* stream) {
    if (!x || !idx_out || !val_out || rows <= 0 || n <= 0 || k <= 0) return GRIT_ERR_BAD_ARG;
    if (k > kMaxK || k > n) return GRIT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(topk_rows, dim3(rows), dim3(kThreads), 0, (hipStream_t)stream, x, ld, n, k, idx_out, val_out);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
