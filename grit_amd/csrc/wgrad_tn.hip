// Weight gradient of nn.Linear on the LONG token maps of GRIT's Swin backbone (gfx950):
//
//     dW[N, K] = dY[M, N]^T . X[M, K]        bf16 in (row-major, the contraction index m is the SLOW index of both operands:
//                                             "TN"), fp32 accumulation;  M = 51 200 tokens at stage 2, N / K = 512 .. 2 048
//
// reference: autograd of the Linear layers of every Swin block (models/common/swin_model.py:26-35 Mlp, :147-149 qkv / proj) --
// half of the backbone's backward GEMM flops.  The library path is a batched GEMM over 16 row slices that writes 16 fp32 copies
// of dW plus a reduction kernel that reads them back (3.6 GB per training step).
//
// Structure:
//   * workgroup = 8 waves (2 x 4) on a 256 (n) x 256 (k) tile of dW over ONE slice of the rows; N/256 x K/256 tiles x S slices
//     = one workgroup per CU, all co-resident, each with a main loop of ~100 steps of 32 rows (the K = 512 forward GEMMs have 8);
//   * a step's operands are 32 rows x 512 bytes of dY and of X, row-major as they lie in memory: global_load_lds_dwordx4 into a
//     four-stage LDS ring with a counted vmcnt (gemm.hip's pipeline).  The transpose the TN layout needs happens in the LDS READ:
//     both MFMA operands come through ds_read_b64_tr_b16 (lane = output row n / output column k, k-slots = the 32 token rows);
//   * the 16 rows a transposing read touches are 512 bytes apart -- the same banks.  The 32-byte groups of a row are XOR-ed with
//     (row & 7) on the DMA's SOURCE address and on the read: 8 distinct bank groups, the 2-way remainder is the instruction's own
//     512 bytes;
//   * v_mfma_f32_16x16x32_bf16 with dY^T as the A operand: accumulator rows = n, lanes = 16 consecutive k.
#include <hip/hip_runtime.h>
#include "per_device.h"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../../include/grit_hip.h"
#include "gemm_math.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kT = 256, kBK = 32, kThreads = 512, kStages = 4;
constexpr int kOpBytes = kBK * kT * 2;       // 16 KB: one operand of one step
constexpr int kStageBytes = 2 * kOpBytes;    // 32 KB
constexpr int kLoads = 4;                    // DMA instructions per thread and stage (2 per operand)
constexpr int kColBytes = 4 * kT * 4;        // bias by-product: four copies (one per k-slot group lg) of 256 fp32 column sums
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

struct TnArgs {
    const __bf16* dY; long ldy;
    const __bf16* X; long ldx;
    float* partial;           // [S][N][K] fp32
    int M, N, K, S, rows_per_split, tiles_n, tiles_k;
    float* db_partial;        // [S][N] fp32 column sums of dY per slice (the bias gradient), or nullptr
    int dbg;                  // diagnostics of the four-wave kernel (GRIT_WGRAD_TN_DBG): 1 = every workgroup of a slice LOADS tile (0, 0)
    // Drop path (round 5; four-wave kernel only): per-sample factors of the branch dY is the gradient of.  The rows of a sample with
    // factor 0 are exact zeros in dY -- they add nothing to dW -- so the 64-row steps inside such samples are not loaded at all and the
    // S slices share the LIVE steps equally (a slice is then a run of live steps, not a fixed row range).  nullptr / 0: every row.
    const float* row_scale;
    int rows_per_sample;      // a multiple of 64; M / rows_per_sample <= 64 samples
};

// Position of a slice in the LIVE 64-row steps of a problem whose dropped samples are skipped (tn4_body computes it, wave-uniform).
struct TnRuns {
    int sps;                      // steps per sample; 0 = no skipping (plain consecutive steps)
    int run_left;                 // steps left in the current sample, the one being loaded included
    unsigned long long rest;      // kept samples from the current one on (bit b = sample b)
    int gap_bytes_y, gap_bytes_x; // bytes of one skipped SAMPLE in dY / X
};

typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

// The transposing reads are issued as inline asm.  Through the builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of every one of
// them (an LDS read it cannot prove disjoint from the LDS-DMA writes still in flight), which serialises the global loads of the
// NEXT stages with the compute of this one (measured: 0.50 PFLOP/s).  As asm the compiler knows nothing about the reads, so the
// waits are this file's: `lgkmcnt(0)` tied to the fragment registers before their first use, a counted `vmcnt` per stage.
// addr: byte offset in LDS of rows 0..15 of the block; the second read takes rows 16..31 (+ 8 192 B).
__device__ __forceinline__ v4i tr_pair(unsigned addr) {
    v2i lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(addr));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(hi) : "v"(addr));
    return v4i{lo[0], lo[1], hi[0], hi[1]};
}

// logical = index of the (slice, tile) pair this workgroup takes, in [0, tiles x S)
__device__ __forceinline__ void tn_body(const TnArgs& g, int logical, char* lds) {
    const int tiles = g.tiles_n * g.tiles_k;
    const int split = logical / tiles, tile = logical - split * tiles;
    const int tn = tile / g.tiles_k, tk = tile - tn * g.tiles_k;
    const int n0 = tn * kT, k0 = tk * kT;
    const int m_begin = split * g.rows_per_split, m_end = min(g.M, m_begin + g.rows_per_split);
    const int nsteps = (m_end - m_begin) / kBK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;  // wave tile: rows n [128 wm, +128), columns k [64 wn, +64)
    const int l15 = lane & 15, lg = lane >> 4, trq = l15 >> 2, trp = l15 & 3;

    // ---- DMA sources: round i covers rows 16 i + tid / 32; the 16-byte position p of the LDS row holds logical chunk
    // (((p >> 1) ^ (row & 7)) << 1) | (p & 1)
    const __bf16* ysrc[2];
    const __bf16* xsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 16 * i + (tid >> 5), p = tid & 31;
        const int c = ((((p >> 1) ^ (row & 7)) << 1) | (p & 1));
        ysrc[i] = g.dY + (size_t)(m_begin + row) * g.ldy + n0 + c * 8;
        xsrc[i] = g.X + (size_t)(m_begin + row) * g.ldx + k0 + c * 8;
    }
    const int wave_dst = wave * 1024;
    auto dma_piece = [&](int slot, int step, int i) {  // i = 0, 1: dY rounds; 2, 3: X rounds
        char* base = lds + slot * kStageBytes;
        if (i < 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(ysrc[i & 1] + (size_t)step * kBK * g.ldy), (lptr_t)(base + (i & 1) * 8192 + wave_dst), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i & 1] + (size_t)step * kBK * g.ldx),
                                             (lptr_t)(base + kOpBytes + (i & 1) * 8192 + wave_dst), 16, 0, 0);
    };

    // ---- fragment read offsets: row (half 16 + 4 lg + trq), 32-byte group (block ^ (row & 7)), 8 bytes at trp
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)lds;
    // Bias gradient as a by-product (db_partial != nullptr, workgroups of k-tile 0, waves wn == 0): the A fragments ARE dY^T -- lane
    // (n = l15 of block i, k-slot group lg) holds 8 of a step's 32 rows of column n.  v_dot2_f32_bf16 with ones folds them (4 per
    // fragment), one ds_add_f32 per fragment adds the lane's share to copy lg of the column sums in LDS: ~40 instructions per step in
    // the shadow of the MFMAs, instead of a separate HBM pass over dY (24 column-sum launches of 30 us per training step).
    const bool colsum = g.db_partial != nullptr && tk == 0;  // workgroup-uniform
    const int cw = __builtin_amdgcn_readfirstlane(wn);    // the four waves of a row half hold the SAME A fragments: wave wn sums
                                                          // fragments wn and wn + 4 (the work spreads over all four SIMDs)
    float* colacc = reinterpret_cast<float*>(lds + kStages * kStageBytes);
    float cs[2] = {0.f, 0.f};  // this lane's share of columns 128 wm + 16 (wn + 4 h) + l15, h = 0, 1 (k-slot group lg)
    const int r_lo = 4 * lg + trq, sr = r_lo & 7;
    // (block ^ sr) is not kept per block (12 address registers): with i < 8, j < 4 and sr < 8,
    //   (8 wm + i) ^ sr = 8 wm | (i ^ sr)        (4 wn + j) ^ sr = (4 wn ^ (sr & 4)) | (j ^ (sr & 3))
    // so a fragment address is a per-lane base plus ((i ^ sr) << 5): two VALU per read instead of one, ten registers back
    const unsigned a_base = lds_base + r_lo * 512 + wm * 8 * 32 + trp * 8;
    const unsigned b_base = lds_base + kOpBytes + r_lo * 512 + ((wn * 4) ^ (sr & 4)) * 32 + trp * 8;
    unsigned srv = (unsigned)sr;  // made opaque once per step (below): otherwise the 12 addresses are hoisted out of the loop again
    auto aoff_of = [&](int i) { return a_base + (((unsigned)i ^ srv) << 5); };
    auto boff_of = [&](int j) { return b_base + (((unsigned)j ^ (srv & 3u)) << 5); };

    v4f acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    // every stage issue is unconditional (past the last step it re-reads the last rows into a slot nobody reads any more): the
    // vmcnt bookkeeping below has no branches
    auto issue_stage_piece = [&](int step, int i) { dma_piece(step % kStages, min(step, nsteps - 1), i); };
#pragma unroll
    for (int st = 0; st < kStages - 1; ++st)
#pragma unroll
        for (int i = 0; i < kLoads; ++i) issue_stage_piece(st, i);

    // Fragments are double-buffered in registers: the 24 transposing reads of step t + 1 are spread between the MFMAs of step t
    // (an LDS round trip is ~2 MFMA groups long and both waves of a SIMD run the same schedule: nothing else would cover it).
    // So a step needs its successor's stage landed: exactly ONE younger stage's DMAs (4 per thread) stay counted across the wait.
    v4i fa[8], fb[4];
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // stage 0 landed
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = tr_pair(boff_of(j));
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = tr_pair(aoff_of(i));
    // One step: MFMAs on the fragments (ca, cb) that are complete, reads of the next step's fragments into (na, nb).  The loop is
    // unrolled by two with the two register sets swapping roles: a register COPY of (na, nb) right after the reads were issued
    // would copy whatever the registers held before the data arrived (the compiler cannot know: the reads are asm).
    auto step = [&](int t, v4i (&ca)[8], v4i (&cb)[4], v4i (&na)[8], v4i (&nb)[4]) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // stage t + 1 landed (stage t + 2 may still be in flight)
        __builtin_amdgcn_s_barrier();                      // ... for every wave; and every wave is done reading stage t - 1's slot
        // the fragments of step t are complete: one lgkmcnt wait, tied to the registers so that no MFMA moves above it
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(ca[0]), "+v"(ca[1]), "+v"(ca[2]), "+v"(ca[3]), "+v"(ca[4]), "+v"(ca[5]), "+v"(ca[6]), "+v"(ca[7]),
                       "+v"(cb[0]), "+v"(cb[1]), "+v"(cb[2]), "+v"(cb[3])
                     :: "memory");
        const unsigned sn = (unsigned)(((t + 1) % kStages) * kStageBytes);
        asm volatile("" : "+v"(srv));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, ca[i]), __builtin_bit_cast(v8bf, cb[j]),
                                                                    acc[i][j], 0, 0, 0);
            if (colsum && (i & 3) == cw) {
                const v2bf one = {(__bf16)1.0f, (__bf16)1.0f};
                const int c0 = ca[i].x, c1 = ca[i].y, c2 = ca[i].z, c3 = ca[i].w;
                float sacc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, c0), one, cs[i >> 2], false);
                sacc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, c1), one, sacc, false);
                sacc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, c2), one, sacc, false);
                cs[i >> 2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2bf, c3), one, sacc, false);
            }
            na[i] = tr_pair(aoff_of(i) + sn);
            if (i < 4) nb[i] = tr_pair(boff_of(i) + sn);
            if (i >= 4) issue_stage_piece(t + kStages - 1, i - 4);  // slot (t + 3) % 4 = (t - 1) % 4: read during step t - 2
        }
    };
    v4i ga[8], gb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) ga[i] = v4i{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) gb[j] = v4i{0, 0, 0, 0};
    int t = 0;
    for (; t + 1 < nsteps; t += 2) {
        step(t, fa, fb, ga, gb);
        step(t + 1, ga, gb, fa, fb);
    }
    if (t < nsteps) step(t, fa, fb, ga, gb);
    // Nothing of this workgroup may land in LDS after it is gone -- and the reads the last step issued into the (now unused)
    // fragment registers must have landed before the compiler re-uses those registers: to the compiler an asm read is complete
    // when it is issued, so the wait is tied to all 24 of them (seen: epilogue addresses overwritten by late LDS data).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                   "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]),
                   "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]), "+v"(ga[4]), "+v"(ga[5]), "+v"(ga[6]), "+v"(ga[7]),
                   "+v"(gb[0]), "+v"(gb[1]), "+v"(gb[2]), "+v"(gb[3])
                 :: "memory");

    // ---- epilogue: acc[i][j][r] = dW[n0 + 128 wm + 16 i + 4 lg + r][k0 + 64 wn + 16 j + l15] of this slice
    float* out = g.partial + ((size_t)split * g.N + n0 + 128 * wm + 4 * lg) * g.K + k0 + 64 * wn + l15;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(size_t)(16 * i + r) * g.K + 16 * j] = acc[i][j][r];
    if (g.db_partial != nullptr && tk == 0) {  // workgroup-uniform
        __syncthreads();  // the ring is free: nothing of the main loop reads LDS any more
#pragma unroll
        for (int h = 0; h < 2; ++h) colacc[lg * kT + wm * 128 + 16 * (cw + 4 * h) + l15] = cs[h];
        __syncthreads();
        if (tid < kT)
            g.db_partial[(size_t)split * g.N + n0 + tid] = (colacc[tid] + colacc[kT + tid]) + (colacc[2 * kT + tid] + colacc[3 * kT + tid]);
    }
}

__global__ __launch_bounds__(kThreads, 2)
void wgrad_tn_256(const TnArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    // XCD-aware order: the workgroups of one XCD (equal blockIdx % 8) take a contiguous band of (slice, tile) pairs, so a slice's
    // rows of dY and X are fetched into one L2 and shared by the tiles that run next to each other
    const int nwg = g.tiles_n * g.tiles_k * g.S;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int qd = nwg >> 3, rm = nwg & 7;
    tn_body(g, (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx, lds);
}

// Several problems in ONE launch (job table by value): the deferred weight gradients of the decoders' short-map Linears.  One of
// them alone (M = 4 800: four 256 x 256 tiles, 150 steps) cannot fill the chip without cutting its rows into slices of a few steps;
// thirty of them together are 128+ tiles whose workgroups each run a long loop -- the regime this kernel is good at.
struct TnGroupArgs {
    TnArgs job[GRIT_WGRAD_GROUP_MAX];
    unsigned first_block[GRIT_WGRAD_GROUP_MAX + 1];
    int n_jobs;
};


// XCD-aware order inside a grouped launch: hardware sends workgroup b to XCD b % 8.  The workgroups of job j occupy the block range
// [first, first + nwg); those of one XCD take a CONTIGUOUS band of the job's (slice, tile) pairs in block order, so that the tiles of a
// slice -- which share its rows of dY and X -- run in one XCD and meet in its L2 (the single-job kernels do the same from block 0).
// Without it the 16 tiles of a slice sit in 8 different L2s: PMC in the step, 1 081 MB fetched per grouped launch against 524 MB of
// unique operands (profiles/r04/pmc_in_step.txt).
__device__ __forceinline__ int tn_group_logical(unsigned bid, unsigned first, unsigned nwg) {
    const unsigned x = bid & 7u;
    unsigned start = 0;
    for (unsigned xx = 0; xx < x; ++xx) {
        const unsigned f0 = first + ((xx + 8u - (first & 7u)) & 7u);  // first block of the job on XCD xx
        start += f0 < first + nwg ? (first + nwg - 1u - f0) / 8u + 1u : 0u;
    }
    const unsigned fx = first + ((x + 8u - (first & 7u)) & 7u);
    return (int)(start + ((bid - fx) >> 3));
}

__global__ __launch_bounds__(kThreads, 2)
void wgrad_tn_256_grouped(const TnGroupArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    int j = 0;
    while (j + 1 < a.n_jobs && blockIdx.x >= a.first_block[j + 1]) ++j;
    tn_body(a.job[j], tn_group_logical(blockIdx.x, a.first_block[j], a.first_block[j + 1] - a.first_block[j]), lds);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Four-wave variant (the default wherever a problem's rows are a multiple of 64): ONE wave per SIMD with a 128 (n) x 128 (k) wave
// tile -- 256 accumulators in the AGPR half of the register file -- and gemm_w4.hip's loop shape: a 64-row K step is 128 MFMAs,
// each followed by at most one LDS read / transfer / address update (plus one VALU of the bias by-product), so everything else
// issues in the shadow of the matrix pipe; two barriers per step.  Per MFMA the wave reads half the LDS bytes of the eight-wave
// kernel above (16 fragments feed 64 MFMAs instead of 12 feeding 32), which is what that kernel's loop was bound by (75 % LDS
// occupancy at two waves per SIMD).
//   * LDS: two buffers of 64 KB (dY 64 rows x 512 B, X 64 rows x 512 B), rows swizzled exactly as above, + 4 KB column sums;
//   * transfers: buffer_load_dwordx4 ... lds, a per-lane offset register per operand, scalar offsets per piece and step, the
//     16 transfers of step s + 2 one at a time between MFMAs once every wave has read the buffer's last fragments;
//   * fragments of k half 1 (rows 32..63) are read during the first 32 MFMAs of the step, those of half 0 of the NEXT step during
//     the last 40; the 16 fragment addresses are kept for the buffer in use and flipped (+- 64 KB) in between.
template <class F, int... Is>
__device__ __forceinline__ void tn4_each(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}

template <int OFF> __device__ __forceinline__ v2i tn4_tr(unsigned addr) {
    v2i v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

constexpr int k4Rows = 64, k4Op = k4Rows * 512, k4Buf = 2 * k4Op, k4Threads = 256;
constexpr int k4Lds = 2 * k4Buf + kColBytes;

// CS: 0 no bias by-product; 1 / 2: this wave folds the even / odd dY fragments into column sums (waves wn = 0 / 1 of k-tile 0)
template <int CS>
__device__ __forceinline__ void tn4_run(const TnArgs& g, int split, int n0, int k0, int m_begin, int nsteps, TnRuns runs, char* lds) {
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;  // wave tile: rows n [128 wm, +128), columns k [128 wn, +128)
    const int l15 = lane & 15, lg = lane >> 4, trq = l15 >> 2, trp = l15 & 3;

    // ---- transfers: piece P = wave + 4 p (p = 0..7) of an operand = rows 2 P, 2 P + 1 of the step, one 1 KB instruction;
    // LDS position p16 of a row holds logical 16-byte chunk (((p16 >> 1) ^ (row & 7)) << 1) | (p16 & 1); row & 7 = (2 wave + lane / 32) & 7
    const int prow = 2 * wave + (lane >> 5), p16 = lane & 31;
    const int pchunk = (((p16 >> 1) ^ (prow & 7)) << 1) | (p16 & 1);
    const unsigned voffY = (unsigned)(((long)prow * g.ldy + pchunk * 8) * 2);
    const unsigned voffX = (unsigned)(((long)prow * g.ldx + pchunk * 8) * 2);
    const int pieceY = (int)(8 * g.ldy * 2), pieceX = (int)(8 * g.ldx * 2);      // bytes between the pieces of a wave
    const int stepY = (int)(k4Rows * g.ldy * 2), stepX = (int)(k4Rows * g.ldx * 2);
    const __amdgpu_buffer_rsrc_t rsY =
        __builtin_amdgcn_make_buffer_rsrc((void*)(g.dY + (size_t)m_begin * g.ldy + ((g.dbg & 1) ? 0 : n0)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX =
        __builtin_amdgcn_make_buffer_rsrc((void*)(g.X + (size_t)m_begin * g.ldx + ((g.dbg & 1) ? 0 : k0)), 0, 0x7fffffff, 0x00020000);
    // the transfer side runs two steps ahead; past the last step it re-fetches the last rows (nobody reads them): no branch in the loop
    int lks = 0, soffY = 0, soffX = 0;
    auto dmaY = [&](int buf, int p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lptr_t)(lds + buf * k4Buf + (wave + 4 * p) * 1024), 16, voffY, p * pieceY + soffY, 0, 0);
    };
    auto dmaX = [&](int buf, int p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(lds + buf * k4Buf + k4Op + (wave + 4 * p) * 1024), 16, voffX, p * pieceX + soffX, 0, 0);
    };
    auto advance_load = [&]() {
        if (lks + 1 < nsteps) {
            ++lks;
            // drop path: leaving the last step of a sample, the next step is the first one of the next KEPT sample (scalar arithmetic,
            // branch-free: ~10 SALU operations per 128-MFMA step)
            runs.run_left -= 1;
            const bool cross = runs.sps != 0 && runs.run_left == 0;
            const unsigned long long next = runs.rest & (runs.rest - 1ull);
            const int skipped = cross ? (__builtin_ctzll(next | (1ull << 63)) - __builtin_ctzll(runs.rest | (1ull << 63)) - 1) : 0;
            runs.rest = cross ? next : runs.rest;
            runs.run_left = cross ? runs.sps : runs.run_left;
            soffY += stepY + skipped * runs.gap_bytes_y;
            soffX += stepX + skipped * runs.gap_bytes_x;
        }
    };

    // ---- fragment addresses (buffer 0): row 4 lg + trq of a 16-row group, 32-byte group (block ^ (row & 7)), 8 bytes at trp;
    // k half h at +16 384 h, the second 16 rows of a half at +8 192 (immediates)
    const int r_lo = 4 * lg + trq, sr = r_lo & 7;
    unsigned ya[8], xa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ya[i] = lds0 + r_lo * 512 + wm * 256 + ((i ^ sr) << 5) + trp * 8;
        xa[i] = lds0 + k4Op + r_lo * 512 + wn * 256 + ((i ^ sr) << 5) + trp * 8;
    }

    v4f acc[8][8];
    v4i Y0[8], X0[8], Y1[8], X1[8];                                      // fragments of k half 0 / 1 as the MFMAs take them
    v2i y0l[8], y0h[8], x0l[8], x0h[8], y1l[8], y1h[8], x1l[8], x1h[8];  // ... as the reads deliver them: rows 0..15 / 16..31 of the half
#define GRIT_TN4_JOIN(D, L, H)                                                                  \
    _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) D[e_] = v4i{L[e_][0], L[e_][1], H[e_][0], H[e_][1]}; \
    GRIT_TN4_TIE(D)
#define GRIT_TN4_TIE(f) asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]))
    // CS: cs[f][r] = column sum of dY at n = 128 wm + 16 (2 f + CS - 1) + 4 lg + r (the same value in all 16 lanes of a group)
    v4f cs[4] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
    v4i ones = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    asm volatile("" : "+v"(ones));

    // ---- prologue: steps 0 and 1 in flight, step 0 landed, its first half in registers
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int p = 0; p < 8; ++p) dmaY(q, p);
#pragma unroll
        for (int p = 0; p < 8; ++p) dmaX(q, p);
        advance_load();
    }
    wait_vm<16>();
    __builtin_amdgcn_s_barrier();
    tn4_each([&](auto kc) { constexpr int k = decltype(kc)::value; const unsigned a = xa[k]; x0l[k] = tn4_tr<0>(a); x0h[k] = tn4_tr<8192>(a); },
             std::make_integer_sequence<int, 8>{});
    tn4_each([&](auto kc) { constexpr int k = decltype(kc)::value; const unsigned a = ya[k]; y0l[k] = tn4_tr<0>(a); y0h[k] = tn4_tr<8192>(a); },
             std::make_integer_sequence<int, 8>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GRIT_TN4_TIE(x0l); GRIT_TN4_TIE(x0h); GRIT_TN4_TIE(y0l); GRIT_TN4_TIE(y0h);
    GRIT_TN4_JOIN(X0, x0l, x0h); GRIT_TN4_JOIN(Y0, y0l, y0h);

    // One 64-row K step = 128 MFMAs (index m: k half m >> 6, dY block i = (m >> 3) & 7, X block j = m & 7):
    //   m   0..31   the 32 transposing reads of k half 1 of this buffer (dY blocks, then X blocks)
    //   m  34       they are in registers + barrier: every wave is done with this buffer
    //   m  36..81   every 3rd: the 16 transfers of step s + 2 into this buffer
    //   m  37..82   every 3rd: the 16 fragment addresses move to the other buffer
    //   m  88       step s + 1 has landed (only the 16 transfers just issued stay counted) + barrier
    //   m  89..120  the 32 reads of k half 0 of the other buffer (X blocks first: the next step's first MFMAs need all of them)
    //   CS: m 16..19 and m 84..87: one more MFMA each (the bias by-product, dY fragment x ones)
    auto kstep = [&](auto firstc, int s) {
        constexpr bool first = decltype(firstc)::value;
        const int b = s & 1;
        const unsigned flip = b ? (unsigned)(-k4Buf) : (unsigned)k4Buf;
        auto slot = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int h = m >> 6, i = (m >> 3) & 7, j = m & 7;
            if constexpr (h == 0 && first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[i][j]) : "v"(Y0[i]), "v"(X0[j]));
            else if constexpr (h == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(Y0[i]), "v"(X0[j]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(Y1[i]), "v"(X1[j]));
            if constexpr (m < 16) {
                const unsigned a = ya[m >> 1];
                if constexpr ((m & 1) == 0) y1l[m >> 1] = tn4_tr<16384>(a);
                else y1h[m >> 1] = tn4_tr<16384 + 8192>(a);
            } else if constexpr (m < 32) {
                const unsigned a = xa[(m - 16) >> 1];
                if constexpr ((m & 1) == 0) x1l[(m - 16) >> 1] = tn4_tr<16384>(a);
                else x1h[(m - 16) >> 1] = tn4_tr<16384 + 8192>(a);
            } else if constexpr (m == 34) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                GRIT_TN4_TIE(y1l); GRIT_TN4_TIE(y1h); GRIT_TN4_TIE(x1l); GRIT_TN4_TIE(x1h);
                GRIT_TN4_JOIN(Y1, y1l, y1h); GRIT_TN4_JOIN(X1, x1l, x1h);
                __builtin_amdgcn_s_barrier();
            } else if constexpr (m >= 36 && m <= 81 && (m - 36) % 3 == 0) {
                constexpr int d = (m - 36) / 3;
                if constexpr (d < 8) dmaY(b, d);
                else dmaX(b, d - 8);
                if constexpr (d == 15) advance_load();
            } else if constexpr (m >= 37 && m <= 82 && (m - 37) % 3 == 0) {
                constexpr int t = (m - 37) / 3;
                if constexpr (t < 8) { unsigned v = ya[t]; asm volatile("v_add_u32 %0, %1, %0" : "+v"(v) : "s"(flip)); ya[t] = v; }
                else { unsigned v = xa[t - 8]; asm volatile("v_add_u32 %0, %1, %0" : "+v"(v) : "s"(flip)); xa[t - 8] = v; }
            } else if constexpr (m == 88) {
                wait_vm<16>();
                __builtin_amdgcn_s_barrier();
            } else if constexpr (m >= 89 && m <= 104) {
                const unsigned a = xa[(m - 89) >> 1];
                if constexpr (((m - 89) & 1) == 0) x0l[(m - 89) >> 1] = tn4_tr<0>(a);
                else x0h[(m - 89) >> 1] = tn4_tr<8192>(a);
            } else if constexpr (m >= 105 && m <= 120) {
                const unsigned a = ya[(m - 105) >> 1];
                if constexpr (((m - 105) & 1) == 0) y0l[(m - 105) >> 1] = tn4_tr<0>(a);
                else y0h[(m - 105) >> 1] = tn4_tr<8192>(a);
            }
            if constexpr (CS != 0) {
                // bias by-product: dY fragment 2 f + CS - 1 times a block of ones on the matrix pipe -- D[n][*] += sum over the half
                // step's 32 rows of dY[row][n]: 8 extra MFMAs per K step (+6 %).  (As v_dot2_f32_bf16 chains on the VALU the same
                // sums cost the workgroup 22 % -- 97 -> 120 us at M 51 200, N 2 048, K 512 -- and the workgroups that carry them fell
                // behind their L2 neighbours.)
                if constexpr (m >= 16 && m < 20) {
                    constexpr int f = m - 16, fi = 2 * f + CS - 1;
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(cs[f]) : "v"(Y0[fi]), "v"(ones));
                } else if constexpr (m >= 84 && m < 88) {
                    constexpr int f = m - 84, fi = 2 * f + CS - 1;
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(cs[f]) : "v"(Y1[fi]), "v"(ones));
                }
            }
        };
        tn4_each(slot, std::make_integer_sequence<int, 128>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GRIT_TN4_TIE(x0l); GRIT_TN4_TIE(x0h); GRIT_TN4_TIE(y0l); GRIT_TN4_TIE(y0h);
        GRIT_TN4_JOIN(X0, x0l, x0h); GRIT_TN4_JOIN(Y0, y0l, y0h);
    };
    const unsigned long long t_loop0 = (g.dbg & 16) ? __builtin_readcyclecounter() : 0ull;
    kstep(std::true_type{}, 0);
    for (int s = 1; s < nsteps; ++s) kstep(std::false_type{}, s);
    const unsigned long long t_loop1 = (g.dbg & 16) ? __builtin_readcyclecounter() : 0ull;
    wait_vm<0>();  // the transfers issued past the end must not land in another workgroup's LDS
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last asm MFMAs retire before their accumulators are read

    // ---- epilogue: acc[i][j][r] = dW[n0 + 128 wm + 16 i + 4 lg + r][k0 + 128 wn + 16 j + l15] of this slice
    float* out = g.partial + ((size_t)split * g.N + n0 + 128 * wm + 4 * lg) * g.K + k0 + 128 * wn + l15;
    const bool nt_out = (g.dbg & 8) != 0;  // A/B: the slice partials streamed past L2
    tn4_each([&](auto qc) {
        constexpr int q = decltype(qc)::value, i = q >> 3, j = q & 7;
        asm volatile("" : "+a"(acc[i][j]));  // (keeps the quad in its AGPRs up to here: no wholesale copy + spill at the top)
        const v4f v = acc[i][j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (nt_out) __builtin_nontemporal_store(v[r], &out[(size_t)(16 * i + r) * g.K + 16 * j]);
            else out[(size_t)(16 * i + r) * g.K + 16 * j] = v[r];
        }
    }, std::make_integer_sequence<int, 64>{});
    if ((g.dbg & 16) && blockIdx.x == 0 && tid == 0)  // diagnostic (corrupts one element): counter cycles of the main loop per K step
        g.partial[0] = (float)(t_loop1 - t_loop0) / (float)nsteps;
    if constexpr (CS != 0) {
        // lanes l15 = 0 write the sums into copy 0 of the column-sum block, lanes 1..3 zero copies 1..3 (tn4_body adds the four)
        asm volatile("s_nop 15" ::: "memory");
        float* colacc = reinterpret_cast<float*>(lds + 2 * k4Buf);
        if (l15 < 4) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) colacc[l15 * kT + wm * 128 + 16 * (2 * f + CS - 1) + 4 * lg + r] = l15 == 0 ? cs[f][r] : 0.f;
        }
    }
}

__device__ __forceinline__ void tn4_body(const TnArgs& g, int logical, char* lds) {
    const int tiles = g.tiles_n * g.tiles_k;
    const int split = logical / tiles, tile = logical - split * tiles;
    const int tn = tile / g.tiles_k, tk = tile - tn * g.tiles_k;
    int m_begin = split * g.rows_per_split;
    int nsteps = (min(g.M, m_begin + g.rows_per_split) - m_begin) / k4Rows;
    TnRuns runs = {0, 0, 0ull, 0, 0};
    if (g.row_scale != nullptr) {
        // the live 64-row steps of the problem, shared equally by the S slices (all of this is wave-uniform scalar work, once per
        // workgroup: <= 64 factor loads and a few bit operations)
        const int samples = g.M / g.rows_per_sample, sps = g.rows_per_sample / k4Rows;
        unsigned long long keep = 0ull;
        for (int b = 0; b < samples; ++b) keep |= (unsigned long long)(g.row_scale[b] != 0.f) << b;
        const int live = __builtin_popcountll(keep) * sps;
        if (live >= g.S && live < samples * sps) {  // (nothing dropped: the plain slices; fewer live steps than slices: likewise -- zeros add up to zero)
            const int j0 = (int)((long)split * live / g.S), j1 = (int)((long)(split + 1) * live / g.S);
            int q = j0 / sps;
            const int r = j0 - q * sps;
            unsigned long long rest = keep;
            for (; q > 0; --q) rest &= rest - 1ull;  // drop the kept samples in front of this slice
            const int first = __builtin_ctzll(rest);
            m_begin = first * g.rows_per_sample + r * k4Rows;
            nsteps = j1 - j0;
            runs.sps = sps;
            runs.run_left = sps - r;
            runs.rest = rest;
            runs.gap_bytes_y = (int)((long)g.rows_per_sample * g.ldy * 2);
            runs.gap_bytes_x = (int)((long)g.rows_per_sample * g.ldx * 2);
        }
    }
    const bool colsum = g.db_partial != nullptr && tk == 0;  // workgroup-uniform
    // (GRIT_WGRAD_TN_DBG & 2: EVERY workgroup of a problem with a by-product runs its MFMAs, so that all run at one pace; measured
    // equal to k-tile 0 alone, stand-alone and in the step -- profiles/r04/wgrad_tn_notes.txt)
    const bool cs_code = (g.dbg & 2) ? g.db_partial != nullptr : colsum;
    const int mode = cs_code ? 1 + (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) : 0;  // wave-uniform; same barriers on all paths
    if (mode == 0) tn4_run<0>(g, split, tn * kT, tk * kT, m_begin, nsteps, runs, lds);
    else if (mode == 1) tn4_run<1>(g, split, tn * kT, tk * kT, m_begin, nsteps, runs, lds);
    else tn4_run<2>(g, split, tn * kT, tk * kT, m_begin, nsteps, runs, lds);
    if (colsum) {
        __syncthreads();
        const float* colacc = reinterpret_cast<const float*>(lds + 2 * k4Buf);
        const int tid = threadIdx.x;
        g.db_partial[(size_t)split * g.N + tn * kT + tid] = (colacc[tid] + colacc[kT + tid]) + (colacc[2 * kT + tid] + colacc[3 * kT + tid]);
    }
}

__global__ __launch_bounds__(k4Threads, 1)
void wgrad_tn4_256(const TnArgs g) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nwg = g.tiles_n * g.tiles_k * g.S;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int qd = nwg >> 3, rm = nwg & 7;
    tn4_body(g, (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx, lds);
}

__global__ __launch_bounds__(k4Threads, 1)
void wgrad_tn4_256_grouped(const TnGroupArgs a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    int j = 0;
    while (j + 1 < a.n_jobs && blockIdx.x >= a.first_block[j + 1]) ++j;
    const int logical = (a.job[j].dbg & 4) ? (int)(blockIdx.x - a.first_block[j])  // (A/B: block order, tiles of a slice over all XCDs)
                                           : tn_group_logical(blockIdx.x, a.first_block[j], a.first_block[j + 1] - a.first_block[j]);
    tn4_body(a.job[j], logical, lds);
}

// GRIT_WGRAD_TN_W4=0: the eight-wave kernel everywhere (A/B switch)
bool tn4_enabled() {
    static const bool on = [] { const char* e = getenv("GRIT_WGRAD_TN_W4"); return !(e && e[0] == '0'); }();
    return on;
}
// the four-wave kernel's conditions on one filled job: 64-row steps, 32-bit byte offsets within a slice
int tn4_dbg() {
    static const int v = [] { const char* e = getenv("GRIT_WGRAD_TN_DBG"); return e ? atoi(e) : 0; }();
    return v;
}
bool tn4_fits(const TnArgs& a) {
    return a.M % k4Rows == 0 && a.rows_per_split % k4Rows == 0 && (long)(a.rows_per_split + 8) * a.ldy * 2 < 0x7fffffffL &&
           (long)(a.rows_per_split + 8) * a.ldx * 2 < 0x7fffffffL;
}
// GRIT_WGRAD_ROW_SKIP=0: never skip the rows of dropped samples (A/B switch; the slices are then fixed row ranges as before)
bool tn_row_skip_enabled() {
    static const bool on = [] { const char* e = getenv("GRIT_WGRAD_ROW_SKIP"); return !(e && e[0] == '0'); }();
    return on;
}
// the factors are used only where the four-wave kernel's skipping applies: whole 64-row steps per sample, <= 64 samples, and a slice
// (which may now span the whole map) addressable with 32-bit byte offsets; anywhere else every row is processed (equally correct)
void tn_set_rows(TnArgs& a, const float* row_scale, int rows_per_sample) {
    a.row_scale = nullptr; a.rows_per_sample = 0;
    if (!row_scale || rows_per_sample <= 0 || !tn_row_skip_enabled() || !tn4_enabled()) return;
    if (rows_per_sample % k4Rows || a.M % rows_per_sample || a.M / rows_per_sample > 64 || a.M / rows_per_sample < 2) return;
    if ((long)(a.M + 8) * a.ldy * 2 >= 0x7fffffffL || (long)(a.M + 8) * a.ldx * 2 >= 0x7fffffffL) return;
    a.row_scale = row_scale; a.rows_per_sample = rows_per_sample;
}

bool tn_shape_ok(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && N % kT == 0 && K % kT == 0 && M % kBK == 0; }

// rows a slice is a multiple of: 64 (the four-wave kernel's K step) wherever the row count allows it, else the 32 of the eight-wave one
inline int tn_granule(int M) { return M % 64 == 0 ? 64 : kBK; }

bool tn_fill(TnArgs& a, const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, float* partial,
             float* db_partial = nullptr) {
    a.db_partial = db_partial;
    a.dbg = tn4_dbg();
    a.row_scale = nullptr; a.rows_per_sample = 0;
    if (!dY || !X || !partial || splits <= 0 || !tn_shape_ok(M, N, K)) return false;
    if (ldy % 8 || ldx % 8 || ldy < N || ldx < K || ((uintptr_t)dY % 16) || ((uintptr_t)X % 16) || ((uintptr_t)partial % 16)) return false;
    const int gr = tn_granule(M), steps = M / gr;
    if (splits > steps) return false;
    a.dY = (const __bf16*)dY; a.ldy = ldy; a.X = (const __bf16*)X; a.ldx = ldx; a.partial = partial;
    a.M = M; a.N = N; a.K = K; a.S = splits;
    a.rows_per_split = ((steps + splits - 1) / splits) * gr;
    if ((M + a.rows_per_split - 1) / a.rows_per_split != splits) return false;  // every slice must own at least one step
    a.tiles_n = N / kT; a.tiles_k = K / kT;
    return true;
}

}  // namespace

extern "C" int grit_wgrad_tn_splits(int M, int N, int K) {
    // one workgroup per CU: S slices of the rows such that tiles x S ~ 256, every slice a whole number of 32-row steps
    if (M <= 0 || N <= 0 || K <= 0 || N % kT || K % kT || M % kBK) return 0;
    const int tiles = (N / kT) * (K / kT);
    int S = 256 / tiles;  // never a second round of workgroups
    if (S < 1) S = 1;
    const int gr = tn_granule(M), steps = M / gr;
    if (S > steps) S = steps;
    const int rows = ((steps + S - 1) / S) * gr;
    return (M + rows - 1) / rows;
}

static int wgrad_tn_launch(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, float* partial,
                           float* db_partial, const float* row_scale, int rows_per_sample, void* stream) {
    if (!dY || !X || !partial || M <= 0 || N <= 0 || K <= 0 || splits <= 0) return GRIT_ERR_BAD_ARG;
    if (N % kT || K % kT || M % kBK || ldy % 8 || ldx % 8 || ldy < N || ldx < K || ((uintptr_t)dY % 16) || ((uintptr_t)X % 16) ||
        ((uintptr_t)partial % 16))
        return GRIT_ERR_UNSUPPORTED;
    if (splits != grit_wgrad_tn_splits(M, N, K)) return GRIT_ERR_BAD_ARG;
    TnArgs a;
    a.dY = (const __bf16*)dY; a.ldy = ldy; a.X = (const __bf16*)X; a.ldx = ldx; a.partial = partial;
    a.M = M; a.N = N; a.K = K; a.S = splits;
    a.db_partial = db_partial;
    a.dbg = tn4_dbg();
    const int gr = tn_granule(M), steps = M / gr;
    a.rows_per_split = ((steps + splits - 1) / splits) * gr;
    a.tiles_n = N / kT; a.tiles_k = K / kT;
    a.row_scale = nullptr; a.rows_per_sample = 0;
    if (tn4_enabled() && tn4_fits(a)) {
        tn_set_rows(a, row_scale, rows_per_sample);
        static grit_detail::PerDevice<bool> attr4_set_pd; bool& attr4_set = attr4_set_pd();
        if (!attr4_set) {
            if (hipFuncSetAttribute((const void*)wgrad_tn4_256, hipFuncAttributeMaxDynamicSharedMemorySize, k4Lds) != hipSuccess)
                return GRIT_ERR_LAUNCH;
            attr4_set = true;
        }
        hipLaunchKernelGGL(wgrad_tn4_256, dim3((unsigned)(a.tiles_n * a.tiles_k * splits)), dim3(k4Threads), k4Lds, (hipStream_t)stream, a);
        return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
    }
    static grit_detail::PerDevice<bool> attr_set_pd; bool& attr_set = attr_set_pd();  // idempotent attribute
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wgrad_tn_256, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes + kColBytes) != hipSuccess)
            return GRIT_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_tn_256, dim3((unsigned)(a.tiles_n * a.tiles_k * splits)), dim3(kThreads), kStages * kStageBytes + kColBytes,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

extern "C" int grit_wgrad_tn(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, float* partial,
                             float* db_partial, void* stream) {
    return wgrad_tn_launch(dY, ldy, X, ldx, M, N, K, splits, partial, db_partial, nullptr, 0, stream);
}

extern "C" int grit_wgrad_tn_rows(const void* dY, long ldy, const void* X, long ldx, int M, int N, int K, int splits, float* partial,
                                  float* db_partial, const float* row_scale, int rows_per_sample, void* stream) {
    if (row_scale && rows_per_sample <= 0) return GRIT_ERR_BAD_ARG;
    return wgrad_tn_launch(dY, ldy, X, ldx, M, N, K, splits, partial, db_partial, row_scale, rows_per_sample, stream);
}

extern "C" int grit_wgrad_tn_group_ok(int M, int N, int K) { return tn_shape_ok(M, N, K) ? 1 : 0; }

extern "C" int grit_wgrad_tn_grouped(const grit_wgrad_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > GRIT_WGRAD_GROUP_MAX) return GRIT_ERR_BAD_ARG;
    TnGroupArgs a;
    a.n_jobs = n_jobs;
    unsigned long long total = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const grit_wgrad_job& jb = jobs[j];
        if (!tn_fill(a.job[j], jb.dY, jb.ldy, jb.X, jb.ldx, jb.M, jb.N, jb.K, jb.splits, jb.dW_partial, jb.db_partial))
            return GRIT_ERR_UNSUPPORTED;
        a.first_block[j] = (unsigned)total;
        total += (unsigned long long)a.job[j].tiles_n * a.job[j].tiles_k * jb.splits;
        if (total > 0x7fffffffULL) return GRIT_ERR_UNSUPPORTED;
    }
    a.first_block[n_jobs] = (unsigned)total;
    bool four = tn4_enabled();
    for (int j = 0; j < n_jobs && four; ++j) four = tn4_fits(a.job[j]);
    if (four)
        for (int j = 0; j < n_jobs; ++j) tn_set_rows(a.job[j], jobs[j].row_scale, jobs[j].rows_per_sample);
    if (four) {
        static grit_detail::PerDevice<bool> attr4_set_pd; bool& attr4_set = attr4_set_pd();
        if (!attr4_set) {
            if (hipFuncSetAttribute((const void*)wgrad_tn4_256_grouped, hipFuncAttributeMaxDynamicSharedMemorySize, k4Lds) != hipSuccess)
                return GRIT_ERR_LAUNCH;
            attr4_set = true;
        }
        hipLaunchKernelGGL(wgrad_tn4_256_grouped, dim3((unsigned)total), dim3(k4Threads), k4Lds, (hipStream_t)stream, a);
        return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
    }
    static grit_detail::PerDevice<bool> attr_set_pd; bool& attr_set = attr_set_pd();  // idempotent attribute
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)wgrad_tn_256_grouped, hipFuncAttributeMaxDynamicSharedMemorySize, kStages * kStageBytes + kColBytes) != hipSuccess)
            return GRIT_ERR_LAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_tn_256_grouped, dim3((unsigned)total), dim3(kThreads), kStages * kStageBytes + kColBytes, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
