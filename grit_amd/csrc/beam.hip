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

// ---------------------------------------------------------------------------------------------------------------------------
// One beam-search step after the word log-probabilities (transformer.py:204-254 `iter`): candidate scores, the k best of
// beam x vocabulary per image, and the per-beam state that moves with the survivors -- two launches instead of ~25.
//
//   alive[b][j]   = first_step ? 1 : seq_mask[b][j] * (prev_words[b][j] != eos)                              (:212-214)
//   cand[b][j][v] = alive ? seq_logprob[b][j] + logp[b][j][v] : (v == 0 ? seq_logprob[b][j] : -999)          (:210,215-218)
//   top-k of cand[b] flattened over (j, v), value descending / index ascending                               (:184-188)
//   sel_beam = index / V, sel_word = index % V, new score = value, new alive = alive[sel_beam],              (:221-233)
//   picked   = logp[b][sel_beam][sel_word] * alive[sel_beam]                                                 (:238-240)
//
// The arithmetic is the reference's (one fp32 add per candidate; the masked blend `m*c + f*(1-m)` of :218 returns c or f
// exactly for m in {1, 0}), so scores and order are bit-identical to the composed form.  Stage 1: a workgroup per
// (image, beam row, part of the vocabulary) reduces its slice to k sorted candidates; stage 2: one wave per image merges them.
constexpr int kBeamThreads = 256;

__device__ __forceinline__ void wave_best(float& v, int& i, int& o) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const float ov = __shfl_xor(v, s, 64);
        const int oi = __shfl_xor(i, s, 64), oo = __shfl_xor(o, s, 64);
        if (better(ov, oi, v, i)) { v = ov; i = oi; o = oo; }
    }
}

template <int KL>  // per-thread list length: k <= KL <= kMaxK (5 for the usual beam of 5: 40 % fewer compare-swaps per insertion)
__global__ __launch_bounds__(kBeamThreads)
void beam_partial(const float* __restrict__ logp, long ld, const float* __restrict__ seq_lp, const float* __restrict__ seq_mask,
                  const int64_t* __restrict__ prev_words, int eos, int first_step, int cur, int V, int k, int parts,
                  float* __restrict__ ws_val, int* __restrict__ ws_idx) {
    __shared__ float sv[2][kBeamThreads / 64];
    __shared__ int si[2][kBeamThreads / 64];
    __shared__ int so[2][kBeamThreads / 64];
    const int part = blockIdx.x, j = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row_id = b * cur + j;
    const float base = seq_lp[row_id];
    bool alive = true;
    if (!first_step) alive = (seq_mask[row_id] * (prev_words[row_id] != (int64_t)eos ? 1.f : 0.f)) != 0.f;
    const float* row = logp + (size_t)row_id * ld;
    const int seg = ((V + parts - 1) / parts + 3) & ~3;
    const int v0 = part * seg, v1 = min(V, v0 + seg);
    float bv[KL];
    int bi[KL];
#pragma unroll
    for (int q = 0; q < KL; ++q) { bv[q] = -INFINITY; bi[q] = 0x7fffffff; }
    auto push = [&](float v, int i) {
        if (!better(v, i, bv[KL - 1], bi[KL - 1])) return;
        bv[KL - 1] = v; bi[KL - 1] = i;
#pragma unroll
        for (int q = KL - 1; q > 0; --q)
            if (better(bv[q], bi[q], bv[q - 1], bi[q - 1])) {
                const float tv = bv[q]; bv[q] = bv[q - 1]; bv[q - 1] = tv;
                const int ti = bi[q]; bi[q] = bi[q - 1]; bi[q - 1] = ti;
            }
    };
    const int flat0 = j * V;
    if (alive) {
        const bool vec = (((uintptr_t)(row + v0)) & 15) == 0;
        if (vec) {
            const int n4 = (v1 - v0) >> 2;
            const float4* r4 = reinterpret_cast<const float4*>(row + v0);
            for (int i = tid; i < n4; i += kBeamThreads) {
                const float4 t = r4[i];
                const int e = flat0 + v0 + 4 * i;
                push(base + t.x, e); push(base + t.y, e + 1); push(base + t.z, e + 2); push(base + t.w, e + 3);
            }
            for (int v = v0 + 4 * n4 + tid; v < v1; v += kBeamThreads) push(base + row[v], flat0 + v);
        } else {
            for (int v = v0 + tid; v < v1; v += kBeamThreads) push(base + row[v], flat0 + v);
        }
    } else {
        // a finished beam survives only through vocabulary index 0; the other entries all score -999 (ties by index)
        const int v = v0 + tid;  // the k <= 8 best of the slice are among its first entries
        if (v < v1) push(v == 0 ? base : -999.f, flat0 + v);
    }
    // merge the 256 sorted lists: k rounds of (wave arg-max over the list heads, then the best of the 4 waves)
    int headp = 0;
    const int out = ((b * cur + j) * parts + part) * kMaxK;
    for (int r = 0; r < k; ++r) {
        float v = -INFINITY; int i = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < KL; ++q)
            if (q == headp) { v = bv[q]; i = bi[q]; }
        int o = tid;
        wave_best(v, i, o);
        const int buf = r & 1;
        if (lane == 0) { sv[buf][w] = v; si[buf][w] = i; so[buf][w] = o; }
        __syncthreads();
        float gv = sv[buf][0]; int gi = si[buf][0], go = so[buf][0];
#pragma unroll
        for (int q = 1; q < kBeamThreads / 64; ++q)
            if (better(sv[buf][q], si[buf][q], gv, gi)) { gv = sv[buf][q]; gi = si[buf][q]; go = so[buf][q]; }
        if (tid == 0) { ws_val[out + r] = gv; ws_idx[out + r] = gi; }
        if (tid == go) ++headp;
    }
}

__global__ __launch_bounds__(64)
void beam_merge(const float* __restrict__ logp, long ld, const float* __restrict__ seq_mask, const int64_t* __restrict__ prev_words,
                int eos, int first_step, int cur, int V, int k, int parts, const float* __restrict__ ws_val,
                const int* __restrict__ ws_idx, int64_t* __restrict__ sel_beam, int64_t* __restrict__ sel_word,
                float* __restrict__ new_lp, float* __restrict__ new_mask, float* __restrict__ picked) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int groups = cur * parts, n = groups * k;  // <= 128 candidates, two per lane
    float v[2]; int i[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = lane + 64 * h;
        v[h] = -INFINITY; i[h] = 0x7fffffff;
        if (c < n) {
            const int g = c / k, r = c - g * k;
            v[h] = ws_val[((size_t)b * groups + g) * kMaxK + r];
            i[h] = ws_idx[((size_t)b * groups + g) * kMaxK + r];
        }
    }
    for (int r = 0; r < k; ++r) {
        const int h = better(v[1], i[1], v[0], i[0]) ? 1 : 0;
        float bvv = v[h]; int bii = i[h], o = lane * 2 + h;
        wave_best(bvv, bii, o);
        if (o == lane * 2) { v[0] = -INFINITY; i[0] = 0x7fffffff; }
        if (o == lane * 2 + 1) { v[1] = -INFINITY; i[1] = 0x7fffffff; }
        if (lane == 0) {
            const int jb = bii / V, word = bii - jb * V;
            const int row_id = b * cur + jb;
            float alive = 1.f;
            if (!first_step) alive = seq_mask[row_id] * (prev_words[row_id] != (int64_t)eos ? 1.f : 0.f);
            const size_t at = (size_t)b * k + r;
            sel_beam[at] = jb; sel_word[at] = word;
            new_lp[at] = bvv; new_mask[at] = alive;
            picked[at] = logp[(size_t)row_id * ld + word] * alive;
        }
    }
}

}  // namespace

extern "C" long grit_beam_step_workspace(int B, int cur_beam, int k) {
    if (B <= 0 || cur_beam <= 0 || k <= 0) return 0;
    const int parts = cur_beam == 1 ? 8 : 2;
    return (long)B * cur_beam * parts * kMaxK * 8;
}

extern "C" int grit_beam_step_f32(const float* logp, long ld, const float* seq_logprob, const float* seq_mask,
                                  const int64_t* prev_words, int eos, int first_step, int B, int cur_beam, int V, int k,
                                  void* workspace, long workspace_bytes, int64_t* sel_beam, int64_t* sel_word,
                                  float* new_seq_logprob, float* new_seq_mask, float* picked_logprob, void* stream) {
    if (!logp || !seq_logprob || !workspace || !sel_beam || !sel_word || !new_seq_logprob || !new_seq_mask || !picked_logprob ||
        B <= 0 || cur_beam <= 0 || V <= 0 || k <= 0 || ld < V)
        return GRIT_ERR_BAD_ARG;
    if (!first_step && (!seq_mask || !prev_words)) return GRIT_ERR_BAD_ARG;
    const int parts = cur_beam == 1 ? 8 : 2;
    if (k > kMaxK || (long)k > (long)cur_beam * V || cur_beam * parts * k > 128 || (long)cur_beam * V > 0x7fffffffL || B > 65535)
        return GRIT_ERR_UNSUPPORTED;
    if (workspace_bytes < grit_beam_step_workspace(B, cur_beam, k)) return GRIT_ERR_BAD_ARG;
    float* ws_val = reinterpret_cast<float*>(workspace);
    int* ws_idx = reinterpret_cast<int*>(ws_val + (size_t)B * cur_beam * parts * kMaxK);
    if (k <= 5)
        hipLaunchKernelGGL(beam_partial<5>, dim3(parts, cur_beam, B), dim3(kBeamThreads), 0, (hipStream_t)stream, logp, ld, seq_logprob,
                           seq_mask, prev_words, eos, first_step, cur_beam, V, k, parts, ws_val, ws_idx);
    else
        hipLaunchKernelGGL(beam_partial<kMaxK>, dim3(parts, cur_beam, B), dim3(kBeamThreads), 0, (hipStream_t)stream, logp, ld,
                           seq_logprob, seq_mask, prev_words, eos, first_step, cur_beam, V, k, parts, ws_val, ws_idx);
    hipLaunchKernelGGL(beam_merge, dim3(B), dim3(64), 0, (hipStream_t)stream, logp, ld, seq_mask, prev_words, eos, first_step,
                       cur_beam, V, k, parts, ws_val, ws_idx, sel_beam, sel_word, new_seq_logprob, new_seq_mask, picked_logprob);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_topk_rows_f32(const float* x, long ld, int rows, int n, int k, int64_t* idx_out, float* val_out, void* stream) {
    if (!x || !idx_out || !val_out || rows <= 0 || n <= 0 || k <= 0) return GRIT_ERR_BAD_ARG;
    if (k > kMaxK || k > n) return GRIT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(topk_rows, dim3(rows), dim3(kThreads), 0, (hipStream_t)stream, x, ld, n, k, idx_out, val_out);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
