// Weight / bias gradient of nn.Linear on the SHORT token maps of GRIT (the two decoders and the grid net: M = 640 .. 4 800 rows):
//
//     dW[N, K] = dY[M, N]^T . X[M, K]          db[N] = sum_m dY[m, n]          (bf16 in, fp32 accumulation)
//
// reference: autograd of the Linear layers in models/detection/det_module.py:313-349 (self_attn.out_proj, cross_attn.*, linear1/2),
// models/caption/grid_net.py:9-42, models/common/attention.py:51-88, models/common/pos_embed.py:34-48.
//
// Why an own kernel here while the long maps stay on the library GEMM: the contraction runs over M, the SLOW dimension of both
// operands ("TN" layout), the output is at most a few hundred 64 x 64 tiles and M is a few thousand -- the tuned library picks
// take 24-47 us per call (0.05-0.1 PFLOP/s: 2.5 GFLOP problems) and the bias gradient costs a column-sum launch plus a reduction
// launch on top (~80 + ~75 + ~75 launches per step).  Here ONE launch per Linear produces dW and db, finished (bf16) when the
// map is short enough for one workgroup per output tile to stream it (<= 4 800 rows: every case of the benchmark), as f32
// split-M partials for the caller's grouped slab sum (grit_slab_sum_grouped) otherwise.
//
// Mapping: workgroup = 4 waves = one 64 (n) x 64 (k) tile of dW over one M split; a step takes 32 rows of dY and X (one 16-byte
// global load per thread and operand, through a 13-stage register ring: 104 KB in flight per workgroup), parks them row-major in LDS and reads BOTH
// MFMA operands through ds_read_b64_tr_b16 (the transpose the TN layout needs happens in the LDS read: lane = output row n /
// output column k, k-slots = the 32 token rows); wave (wy, wx) owns the 32 x 32 quadrant = 2 x 2 v_mfma_f32_16x16x32_bf16 tiles.
// The workgroups of k-tile 0 also sum their dY rows per column (the bias-gradient partial).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grit_hip.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

constexpr int kTile = 64, kStep = 32, kPitch = 72;  // bf16 elements per LDS row: 144 B, rows 16-byte aligned, banks spread
constexpr int kMaxSteps = 13;                       // rows per split <= 416: the whole split sits in registers (104 VGPRs)

__device__ __forceinline__ v8bf tr_pair(const __bf16* lo, const __bf16* hi) {
    const v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)lo);
    const v4s b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)hi);
    const v8s r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(v8bf, r);
}

struct WgradLds {
    __bf16 Ys[2][kStep * kPitch];
    __bf16 Xs[2][kStep * kPitch];
    float colred[4][kTile];
};

__device__ __forceinline__
void wgrad_body(const __bf16* __restrict__ dY, long ldy, const __bf16* __restrict__ X, long ldx, int M, int N, int K,
                int rows_per_split, float* __restrict__ dWp, float* __restrict__ dbp, __bf16* __restrict__ dW16,
                __bf16* __restrict__ db16, int block_id, WgradLds& lds) {
    auto& Ys = lds.Ys;
    auto& Xs = lds.Xs;
    auto& colred = lds.colred;

    const int tiles_k = K / kTile, tiles_n = N / kTile;
    int id = block_id;
    const int tk = id % tiles_k; id /= tiles_k;
    const int tn = id % tiles_n;
    const int s = id / tiles_n;
    const int n0 = tn * kTile, k0 = tk * kTile;
    const int m_lo = s * rows_per_split, m_hi = min(M, m_lo + rows_per_split);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wy = wave >> 1, wx = wave & 1;        // quadrant: n rows 32 wy .., k columns 32 wx ..
    const int l15 = lane & 15, lg = lane >> 4, trq = l15 >> 2, trp = l15 & 3;
    const int lrow = tid >> 3, lcol = (tid & 7) * 8;  // this thread's 16-byte piece of a 32 x 64 chunk

    const __bf16* ysrc = dY + (size_t)n0 + lcol;
    const __bf16* xsrc = X + (size_t)k0 + lcol;
    auto fetch = [&](int m, uint4& y, uint4& x) {
        const int r = m + lrow;
        if (r < m_hi) {
            y = *reinterpret_cast<const uint4*>(ysrc + (size_t)r * ldy);
            x = *reinterpret_cast<const uint4*>(xsrc + (size_t)r * ldx);
        } else {
            y = make_uint4(0, 0, 0, 0);
            x = make_uint4(0, 0, 0, 0);
        }
    };

    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool want_db = (dbp != nullptr || db16 != nullptr) && tk == 0;

    // A ring of kMaxSteps register stages (2 x 16 bytes per thread and stage): up to 13 chunks = 416 rows = 104 KB per workgroup
    // are in flight at any time.  A step is ~150 cycles of LDS / MFMA work -- far too little to cover a memory round trip, and
    // a launch has only 16 .. 256 workgroups (one tile each) to hide it with; every kernel also starts on a cold L2 (the L2s of
    // the 8 XCDs are written back and invalidated at kernel boundaries), so the first touch of every line is a MALL / HBM access.
    // (First version: one-step prefetch, 30 us per call in the step -- no better than the library GEMM it replaced.)
    uint4 yq[kMaxSteps], xq[kMaxSteps];
#pragma unroll
    for (int i = 0; i < kMaxSteps; ++i) fetch(m_lo + kStep * i, yq[i], xq[i]);  // rows past m_hi come back as zeros
    const int nsteps = (m_hi - m_lo + kStep - 1) / kStep;
    for (int base = 0; base < nsteps; base += kMaxSteps) {
#pragma unroll
        for (int i = 0; i < kMaxSteps; ++i) {
            if (base + i < nsteps) {  // workgroup-uniform
                const int buf = i & 1;  // kMaxSteps is odd: the parity flips across the ring's wrap-around as well ...
                const int pb = ((base / kMaxSteps) & 1) ^ buf;  // ... once it is folded in
                *reinterpret_cast<uint4*>(&Ys[pb][lrow * kPitch + lcol]) = yq[i];
                *reinterpret_cast<uint4*>(&Xs[pb][lrow * kPitch + lcol]) = xq[i];
                if (want_db) {
                    const uint32_t w[4] = {yq[i].x, yq[i].y, yq[i].z, yq[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cs[2 * e] += __uint_as_float(w[e] << 16);
                        cs[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                    }
                }
                __syncthreads();  // chunk complete; the other buffer's readers finished before the previous barrier
                fetch(m_lo + kStep * (base + i + kMaxSteps), yq[i], xq[i]);  // refill this stage: one full ring ahead
                const __bf16* yb = &Ys[pb][(4 * lg + trq) * kPitch + 32 * wy + 4 * trp];
                const __bf16* xb = &Xs[pb][(4 * lg + trq) * kPitch + 32 * wx + 4 * trp];
                v8bf a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    a[t] = tr_pair(yb + 16 * t, yb + 16 * t + 16 * kPitch);
                    b[t] = tr_pair(xb + 16 * t, xb + 16 * t + 16 * kPitch);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b[j], acc[t][j], 0, 0, 0);
                // two buffers: the stores of the next step go to the other buffer, whose last readers passed this step's barrier
            }
        }
    }

    // acc[i][j][r] = dW[n0 + 32 wy + 16 i + 4 lg + r][k0 + 32 wx + 16 j + l15]
    if (dW16 != nullptr) {  // one split: the finished gradient, rounded once
        __bf16* out = dW16 + (size_t)(n0 + 32 * wy) * K + k0 + 32 * wx + l15;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(16 * i + 4 * lg + r) * K + 16 * j] = (__bf16)acc[i][j][r];
    } else {
        float* out = dWp + ((size_t)s * N + n0 + 32 * wy) * K + k0 + 32 * wx + l15;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(16 * i + 4 * lg + r) * K + 16 * j] = acc[i][j][r];
    }

    if (want_db) {  // fold the 32 row-threads of every 8-column chunk: rows of a wave by shuffles, the 4 waves through LDS
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = cs[i];
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            cs[i] = v;
        }
        __syncthreads();
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) colred[wave][lane * 8 + i] = cs[i];
        }
        __syncthreads();
        if (tid < kTile) {
            const float v = colred[0][tid] + colred[1][tid] + colred[2][tid] + colred[3][tid];
            if (db16 != nullptr) db16[n0 + tid] = (__bf16)v;
            else dbp[(size_t)s * N + n0 + tid] = v;
        }
    }
}

__global__ __launch_bounds__(256)
void wgrad_small(const __bf16* __restrict__ dY, long ldy, const __bf16* __restrict__ X, long ldx, int M, int N, int K, int S,
                 int rows_per_split, float* __restrict__ dWp, float* __restrict__ dbp, __bf16* __restrict__ dW16,
                 __bf16* __restrict__ db16) {
    __shared__ __attribute__((aligned(16))) WgradLds lds;
    wgrad_body(dY, ldy, X, ldx, M, N, K, rows_per_split, dWp, dbp, dW16, db16, (int)blockIdx.x, lds);
}

// Many problems, one launch: the weight / bias gradients of the short-map Linears of a whole decoder, collected while its
// backward pass runs and executed together when their consumer (the gradient-bucket wrapper) asks for them.  One problem alone
// leaves the chip to ~3 workgroups per CU that each crawl through ~13 latency-bound steps (31 us per call inside the step,
// profiles/r03/negative_results.txt); dozens of problems in one grid fill every CU with independent workgroups.
struct WgradGroupArgs {
    grit_wgrad_job job[GRIT_WGRAD_GROUP_MAX];
    unsigned first_block[GRIT_WGRAD_GROUP_MAX + 1];
    int rows_per_split[GRIT_WGRAD_GROUP_MAX];
    int n_jobs;
};

__global__ __launch_bounds__(256)
void wgrad_small_grouped(const WgradGroupArgs a) {
    __shared__ __attribute__((aligned(16))) WgradLds lds;
    int j = 0;
    while (j + 1 < a.n_jobs && blockIdx.x >= a.first_block[j + 1]) ++j;
    const grit_wgrad_job& jb = a.job[j];
    wgrad_body((const __bf16*)jb.dY, jb.ldy, (const __bf16*)jb.X, jb.ldx, jb.M, jb.N, jb.K, a.rows_per_split[j], jb.dW_partial,
               jb.db_partial, (__bf16*)nullptr, (__bf16*)nullptr, (int)(blockIdx.x - a.first_block[j]), lds);
}

}  // namespace

static int rows_per_split(int M, int splits) {
    int rows = (M + splits - 1) / splits;
    return (rows + kStep - 1) / kStep * kStep;
}

extern "C" int grit_wgrad_small_splits(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0 || N % kTile || K % kTile) return 0;
    // One split (the kernel then writes the finished bf16 gradients itself: no partials, no reduction launch) while a workgroup
    // can stream its rows in ~10 us (64 x 64 tiles: 256 bytes per row against ~120 GB/s per CU); longer maps are cut into
    // splits of <= 4 800 rows whose f32 partials the caller reduces.
    static const int policy = getenv("GRIT_WGRAD_SPLIT_POLICY") ? atoi(getenv("GRIT_WGRAD_SPLIT_POLICY")) : 0;
    if (policy == 1) return (M + 4799) / 4800;
    // policy 0: ~768 workgroups (3 per CU hide each other's step latency), each at least two and at most 13 steps of 32 rows
    const int tiles = (N / kTile) * (K / kTile);
    int S = (768 + tiles - 1) / tiles;
    const int max_s = (M + 63) / 64;
    if (S > max_s) S = max_s;
    const int min_s = (M + kMaxSteps * kStep - 1) / (kMaxSteps * kStep);
    if (S < min_s) S = min_s;
    if (S < 1) S = 1;
    const int rows = rows_per_split(M, S);
    return (M + rows - 1) / rows;
}

extern "C" int grit_wgrad_small(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, void* dW_out,
                                void* db_out, void* stream) {
    if (!dY || !X || !dW_out || M <= 0 || N <= 0 || K <= 0 || splits <= 0) return GRIT_ERR_BAD_ARG;
    if (N % kTile || K % kTile || ldy % 8 || ldx % 8 || ldy < N || ldx < K || ((uintptr_t)dY % 16) || ((uintptr_t)X % 16) ||
        ((uintptr_t)dW_out % 16))
        return GRIT_ERR_UNSUPPORTED;
    if (splits != grit_wgrad_small_splits(M, N, K)) return GRIT_ERR_BAD_ARG;
    const long blocks = (long)(N / kTile) * (K / kTile) * splits;
    if (blocks > 0x7fffffffL) return GRIT_ERR_UNSUPPORTED;
    const bool direct = splits == 1;
    hipLaunchKernelGGL(wgrad_small, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dY, ldy, (const __bf16*)X,
                       ldx, M, N, K, splits, rows_per_split(M, splits), direct ? (float*)nullptr : (float*)dW_out,
                       direct ? (float*)nullptr : (float*)db_out, direct ? (__bf16*)dW_out : (__bf16*)nullptr,
                       direct ? (__bf16*)db_out : (__bf16*)nullptr);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_wgrad_group_splits(int M) {
    // inside a grouped launch the other problems fill the chip: a split is simply one pass of the register ring (<= 416 rows)
    return M <= 0 ? 0 : (M + kMaxSteps * kStep - 1) / (kMaxSteps * kStep);
}

extern "C" int grit_wgrad_small_grouped(const grit_wgrad_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GRIT_WGRAD_GROUP_MAX) return GRIT_ERR_BAD_ARG;
    WgradGroupArgs a;
    a.n_jobs = n_jobs;
    unsigned long long total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const grit_wgrad_job& jb = jobs[j];
        if (!jb.dY || !jb.X || !jb.dW_partial || jb.M <= 0 || jb.N <= 0 || jb.K <= 0) return GRIT_ERR_BAD_ARG;
        if (jb.N % kTile || jb.K % kTile || jb.ldy % 8 || jb.ldx % 8 || jb.ldy < jb.N || jb.ldx < jb.K || ((uintptr_t)jb.dY % 16) ||
            ((uintptr_t)jb.X % 16) || ((uintptr_t)jb.dW_partial % 16))
            return GRIT_ERR_UNSUPPORTED;
        if (jb.splits != grit_wgrad_group_splits(jb.M)) return GRIT_ERR_BAD_ARG;
        a.job[j] = jb;
        a.rows_per_split[j] = rows_per_split(jb.M, jb.splits);
        a.first_block[j] = (unsigned)total;
        total += (unsigned long long)(jb.N / kTile) * (jb.K / kTile) * jb.splits;
        if (total > 0x7fffffffULL) return GRIT_ERR_UNSUPPORTED;
    }
    a.first_block[n_jobs] = (unsigned)total;
    hipLaunchKernelGGL(wgrad_small_grouped, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
