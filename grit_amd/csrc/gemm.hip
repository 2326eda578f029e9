// bf16 MFMA GEMM with fused epilogues for the long token maps of GRIT's Swin backbone (gfx950).
//
//   C[M, N] = epilogue( A[M, K] . B[N, K]^T )        A, B, C bf16 row-major (both operands K-contiguous), fp32 accumulation
//
// replaces "library GEMM + separate elementwise / reduction passes" around the Mlp of every Swin block
// (davidnvq/grit models/common/swin_model.py:31-37 Mlp.forward, :289-298 block tail):
//   GRIT_GEMM_BIAS        C = acc + bias                                   (any Linear)
//   GRIT_GEMM_BIAS_GELU   aux = acc + bias ; C = gelu(aux)                 (fc1 + exact-erf GELU; aux = pre-activation kept for backward)
//   GRIT_GEMM_DGELU       C = acc * gelu'(aux) ; colsum[slab, n] = sum_rows C   (fc2 input gradient x GELU' + fc1 bias gradient)
// so the [M, 4C] hidden map is written once per pass instead of written, re-read and re-written by GELU / GeluBackward / column-sum kernels.
//
// Structure (MI355X_MICROARCH.md / cdna_hip_programming.md section 5):
//   * workgroup = 4 waves (2 x 2) on a 256 x 128 tile, K step 32, three-stage LDS ring (72 KB) filled by global_load_lds_dwordx4
//     with a counted vmcnt (one stage stays in flight across the barrier) -- two workgroups per CU, so one workgroup's
//     prologue / epilogue (bias, GELU, LDS transpose, 128-byte row stores) runs under the other's MFMA loop;
//   * LDS image is lane-linear per wave-instruction (the DMA's rule); bank conflicts of the ds_read_b128 fragment reads are
//     removed by permuting the 16-byte chunks on the SOURCE address and on the read (tools/micro/lds_bank_sim.py);
//   * operands are fed to v_mfma_f32_16x16x32_bf16 swapped (weight rows as the MFMA A operand, token rows as B), which puts a
//     token on the lane and 4 consecutive output channels in the accumulator registers: the epilogue packs 8-byte pieces,
//     transposes through the (now idle) stage ring and stores whole 128-byte row segments;
//   * XCD-aware tile order: the 8 groups of blockIdx % 8 walk disjoint bands of row panels, so an A panel is fetched from HBM
//     by one L2 only and re-used by the N / 128 column tiles that run next to each other.
#include <hip/hip_runtime.h>
#include "per_device.h"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/grit_hip.h"
#include "gemm_math.h"
#include "gemm_launchers.h"

namespace {

struct GemmArgs {
    const __bf16* A; long lda;
    const __bf16* B; long ldb;
    __bf16* C; long ldc;
    const __bf16* bias;
    __bf16* aux; long ldaux;
    int nt_aux;  // non-temporal 1: store of the saved pre-activation, 2: C of BIAS_GELU, 4: C of DGELU, 8: load of the pre-activation
                 // in DGELU (GRIT_GEMM_NT_AUX, default 15:
                 // whole-tile outputs streamed past L2 leave the operand panels resident -- 63.4 -> 62.0 ms per training step)
    float* colsum;
    int M, N, K, tiles_m, tiles_n;
    const float* row_scale;    // GRIT_GEMM_DGELU, optional: per-sample factors that were applied to the rows of A (drop path);
    int rows_per_sample;       //   a tile whose rows all belong to ONE sample with factor 0 has A = 0: its result is written as zeros
    const unsigned long long* seed;  // GRIT_GEMM_BIAS_RELU_DROP: device seed of the dropout hash (read when drop_p > 0)
    float drop_p;                    // GRIT_GEMM_BIAS_RELU_DROP / GRIT_GEMM_DRELU: dropout probability
#ifdef GRIT_GEMM_STAMPS
    unsigned long long* stamps;  // diagnostic build only (tools/micro/gemm_stamps.hip): [workgroup][wave][16] s_memtime values
#endif
};

#ifdef GRIT_GEMM_STAMPS
#define GRIT_STAMP(slot)                                                                                          \
    if (g.stamps && (threadIdx.x & 63) == 0)                                                                      \
        g.stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime();
#else
#define GRIT_STAMP(slot)
#endif

// keep factor of dropout(relu(.)): the counter hash of glue.hip / layernorm.hip over the element index m * N + n
__device__ __forceinline__ float relu_keep_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
    unsigned long long z = idx + seed * 0x9E3779B97F4A7C15ull;
    unsigned int x = (unsigned int)(z ^ (z >> 32));
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

template <int BM, int BN, int BK, int WM, int WN, int NSTAGE, int EPI>
__global__ __launch_bounds__(WM * WN * 64, 2)
void gemm_nt_bf16(const GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    constexpr int MT = WTM / 16, NTL = WTN / 16, KS = BK / 32;
    constexpr int ROWB = BK * 2, CPR = ROWB / 16;  // bytes / 16-byte chunks per staged row
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int LA = A_BYTES / (NT * 16), LB = B_BYTES / (NT * 16), LOADS = LA + LB;
    static_assert(A_BYTES % (NT * 16) == 0 && B_BYTES % (NT * 16) == 0, "stage must be a whole number of DMA rounds");
    static_assert(WTN == 64, "epilogue stores 128-byte row segments per wave");
    constexpr int EPI_BYTES = WTM * WTN * 2;  // per wave
    static_assert(WM * WN * EPI_BYTES <= NSTAGE * STAGE, "epilogue transpose must fit the stage ring");

    extern __shared__ __attribute__((aligned(1024))) char lds[];

    // XCD-aware tile id: blocks with equal blockIdx % 8 share an L2; give each such group a contiguous band of tiles (bijective)
    const int nwg = g.tiles_m * g.tiles_n;
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int qd = nwg >> 3, rm = nwg & 7;
    int tile = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + idx;
    int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    if (g.row_scale != nullptr && (g.tiles_m & 7) == 0) {
        // With tiles that may be skipped (samples removed by drop path: a run of ~6 consecutive row panels) a contiguous band of row
        // panels per XCD leaves the skipping to the one XCD that owns the sample, and the launch waits for the others.  Deal the row
        // panels round-robin instead -- panel 8 p + xcd -- : every XCD still runs whole panels (the 8 tiles that share an A panel), and
        // a dropped sample thins out all of them.
        const int pl = idx / g.tiles_n;
        tn = idx - pl * g.tiles_n;
        tm = 8 * pl + xcd;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    if constexpr (EPI == GRIT_GEMM_DGELU || EPI == GRIT_GEMM_BIAS_GELU) {
        // Backward (DGELU): rows of a sample that drop path removed from this branch arrive as exact zeros (dbranch = 0 * dx):
        // the product is zero whatever the weights are -- no K loop, no GELU', just the zero tile and zero column sums.
        // Forward (BIAS_GELU): the branch output of such a sample is multiplied by 0 and the tensors saved here meet only zero gradients
        // in the backward: the tile is not computed, activation and saved pre-activation are written as zeros (finite: 0 * x stays 0).
        if (g.row_scale != nullptr) {
            const int s_lo = m0 / g.rows_per_sample, s_hi = (min(m0 + BM, g.M) - 1) / g.rows_per_sample;
            if (s_lo == s_hi && g.row_scale[s_lo] == 0.f) {  // workgroup-uniform
                const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int it = 0; it < WTM / 8; ++it) {
                    const int m = mw + it * 8 + (lane >> 3);
                    if (m < g.M) {
                        __builtin_nontemporal_store(z, reinterpret_cast<u32x4*>(g.C + (size_t)m * g.ldc + nw + (lane & 7) * 8));
                        if constexpr (EPI == GRIT_GEMM_BIAS_GELU) {
                            if (g.aux) __builtin_nontemporal_store(z, reinterpret_cast<u32x4*>(g.aux + (size_t)m * g.ldaux + nw + (lane & 7) * 8));
                        }
                    }
                }
                if constexpr (EPI != GRIT_GEMM_BIAS_GELU)
                if (l15 == 0 && mw < g.M) {
                    float* dst = g.colsum + (size_t)(mw / WTM) * g.N + nw + 4 * lq;
#pragma unroll
                    for (int j = 0; j < NTL; ++j) *reinterpret_cast<v4f*>(dst + 16 * j) = v4f{0.f, 0.f, 0.f, 0.f};
                }
                return;
            }
        }
    }

    // ---- DMA source pointers (one per round) and LDS destinations -------------------------------------------------
    const __bf16* asrc[LA];
    const __bf16* bsrc[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int c = i * NT + tid, row = c / CPR, pc = c % CPR;
        const int lc = pc ^ chunk_swizzle<BK>(row & 15);
        const int m = min(m0 + row, g.M - 1);
        asrc[i] = g.A + (size_t)m * g.lda + lc * 8;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int c = i * NT + tid, row = c / CPR, pc = c % CPR;
        const int lc = pc ^ chunk_swizzle<BK>(row & 15);
        bsrc[i] = g.B + (size_t)(n0 + row) * g.ldb + lc * 8;
    }
    const int wave_dst = wave * 1024;  // byte offset of this wave's 1 KiB piece inside a DMA round

    auto stage = [&](int slot, int kt) {
        char* base = lds + slot * STAGE;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + k0), (lptr_t)(base + i * NT * 16 + wave_dst), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + k0), (lptr_t)(base + A_BYTES + i * NT * 16 + wave_dst), 16, 0, 0);
    };

    // ---- fragment read offsets --------------------------------------------------------------------------------
    int foff[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) foff[s] = l15 * ROWB + (((s * 4 + lq) ^ chunk_swizzle<BK>(l15)) * 16);
    const int a_wave = wm * WTM * ROWB, b_wave = A_BYTES + wn * WTN * ROWB;

    v4f acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    GRIT_STAMP(0)
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s, s);
    GRIT_STAMP(1)

    // One K step of the main loop.  Fragments of a 32-deep k-step: NTL weight-row blocks + MT token-row blocks (one
    // ds_read_b128 each).  All reads of a k-step are issued before its first MFMA; the DMA pieces of the stage that is being
    // prefetched and (BK = 64) the reads of the second k-step are spread between the MFMAs, so the wave's instruction stream
    // stays MFMA-dense and the other wave of the SIMD fills what is left.
    auto read_frags = [&](const char* sb, int ks, v8bf (&w)[NTL], v8bf (&x)[MT]) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) w[j] = *reinterpret_cast<const v8bf*>(sb + b_wave + j * 16 * ROWB + foff[ks]);
#pragma unroll
        for (int i = 0; i < MT; ++i) x[i] = *reinterpret_cast<const v8bf*>(sb + a_wave + i * 16 * ROWB + foff[ks]);
    };
    auto dma_piece = [&](int slot, int kt, int i) {
        char* base = lds + slot * STAGE;
        const int k0 = kt * BK;
        if (i < LA)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i < LA ? i : 0] + k0), (lptr_t)(base + i * NT * 16 + wave_dst), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i >= LA ? i - LA : 0] + k0),
                                             (lptr_t)(base + A_BYTES + (i - LA) * NT * 16 + wave_dst), 16, 0, 0);
    };
    auto k_tile = [&](int t, const bool kStage) {  // kStage is wave-uniform: scalar branches around the DMA pieces only
        const char* sb = lds + (t % NSTAGE) * STAGE;
        const int pslot = (t + NSTAGE - 1) % NSTAGE, pkt = t + NSTAGE - 1;
        v8bf w0[NTL], x0[MT];
        read_frags(sb, 0, w0, x0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KS == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j], x0[i], acc[i][j], 0, 0, 0);
                if (i < LOADS && kStage) dma_piece(pslot, pkt, i);
            }
#pragma unroll
            for (int i = MT; i < LOADS; ++i)  // (small tiles: more pieces than token blocks)
                if (kStage) dma_piece(pslot, pkt, i);
        } else {
            v8bf w1[NTL], x1[MT];
            // k-step 0 MFMAs, with the reads of k-step 1 and the first DMA pieces in between
#pragma unroll
            for (int j = 0; j < NTL; ++j) w1[j] = *reinterpret_cast<const v8bf*>(sb + b_wave + j * 16 * ROWB + foff[KS - 1]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j], x0[i], acc[i][j], 0, 0, 0);
                x1[i] = *reinterpret_cast<const v8bf*>(sb + a_wave + i * 16 * ROWB + foff[KS - 1]);
                if (i < LOADS / 2 && kStage) dma_piece(pslot, pkt, i);
            }
#pragma unroll
            for (int i = MT; i < LOADS / 2; ++i)  // (small tiles: more pieces than token blocks)
                if (kStage) dma_piece(pslot, pkt, i);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j], x1[i], acc[i][j], 0, 0, 0);
                if (i < LOADS - LOADS / 2 && kStage) dma_piece(pslot, pkt, LOADS / 2 + i);
            }
#pragma unroll
            for (int i = MT; i < LOADS - LOADS / 2; ++i)
                if (kStage) dma_piece(pslot, pkt, LOADS / 2 + i);
        }
    };

#ifdef GRIT_GEMM_STAMPS
    unsigned long long ph_wait = 0, ph_barrier = 0, ph_compute = 0, ph_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (int t = 0; t < nk; ++t) {
        // tile t has landed once at most the (NSTAGE - 2) younger tiles' DMAs are still counted
        if (t + NSTAGE - 2 < nk) {
            if constexpr (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if constexpr (LOADS * (NSTAGE - 2) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if constexpr (LOADS * (NSTAGE - 2) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (LOADS * (NSTAGE - 2) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (LOADS * (NSTAGE - 2) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifdef GRIT_GEMM_STAMPS
        const unsigned long long ph_t1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef GRIT_GEMM_STAMPS
        const unsigned long long ph_t2 = __builtin_amdgcn_s_memtime();
#endif
        if (t == 0) { GRIT_STAMP(2) }
        if (t == 1) { GRIT_STAMP(3) }
        if (t == 2) { GRIT_STAMP(4) }
        k_tile(t, t + NSTAGE - 1 < nk);
#ifdef GRIT_GEMM_STAMPS
        {
            const unsigned long long ph_t3 = __builtin_amdgcn_s_memtime();
            if (t > 0) { ph_wait += ph_t1 - ph_t0; ph_barrier += ph_t2 - ph_t1; ph_compute += ph_t3 - ph_t2; }
            ph_t0 = ph_t3;
        }
#endif
    }
    GRIT_STAMP(5)
#ifdef GRIT_GEMM_STAMPS
    if (g.stamps && (threadIdx.x & 63) == 0) {
        unsigned long long* o = g.stamps + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
        o[8] = ph_wait; o[9] = ph_barrier; o[10] = ph_compute;
    }
#endif

    // ---- epilogue ---------------------------------------------------------------------------------------------------
    // acc[i][j][r] = C[m0 + wm*WTM + 16 i + l15][n0 + wn*WTN + 16 j + 4 lq + r]
    __builtin_amdgcn_s_barrier();  // every wave is done with the last stage: the ring becomes the transpose buffer
    asm volatile("" ::: "memory");
    GRIT_STAMP(6)
    char* eb = lds + wave * EPI_BYTES;
    const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;

    v4f bias4[NTL];
    if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU || EPI == GRIT_GEMM_BIAS_RES || EPI == GRIT_GEMM_BIAS_RELU_DROP) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const v4bf b = *reinterpret_cast<const v4bf*>(g.bias + nw + 16 * j + 4 * lq);
            bias4[j] = v4f{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
        }
    }

    // write one packed 8-byte piece per (i, j) into the wave's [WTM][64] bf16 image (16-byte chunks XOR-ed with row & 7)
    auto put = [&](int i, int j, const v4f& v) {
        v4bf p;
        p[0] = (__bf16)v[0]; p[1] = (__bf16)v[1]; p[2] = (__bf16)v[2]; p[3] = (__bf16)v[3];
        const int row = 16 * i + l15;
        const int chunk = (2 * j + (lq >> 1)) ^ (row & 7);
        *reinterpret_cast<v4bf*>(eb + row * 128 + chunk * 16 + (lq & 1) * 8) = p;
    };
    // stream the image out: 8 rows x 128 B per wave-instruction
    const bool full_rows = mw + WTM <= g.M;  // wave-uniform: no per-store row test on interior tiles
    auto flush = [&](__bf16* dst, long ld, bool nt = false) {
        __bf16* base = dst + (size_t)(mw + (lane >> 3)) * ld + nw + (lane & 7) * 8;
        if (full_rows && nt) {  // streaming output: nothing reads it back soon, keep it out of the way of the operands in L2
#pragma unroll
            for (int it = 0; it < WTM / 8; ++it) {
                const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(eb + row * 128 + ((chunk ^ (row & 7)) * 16)),
                                            reinterpret_cast<u32x4*>(base + (size_t)it * 8 * ld));
            }
        } else if (full_rows) {
#pragma unroll
            for (int it = 0; it < WTM / 8; ++it) {
                const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                *reinterpret_cast<uint4*>(base + (size_t)it * 8 * ld) =
                    *reinterpret_cast<const uint4*>(eb + row * 128 + ((chunk ^ (row & 7)) * 16));
            }
        } else {
#pragma unroll
            for (int it = 0; it < WTM / 8; ++it) {
                const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(eb + row * 128 + ((chunk ^ (row & 7)) * 16));
                if (mw + row < g.M) *reinterpret_cast<uint4*>(base + (size_t)it * 8 * ld) = v;
            }
        }
    };

    if constexpr (EPI == GRIT_GEMM_NONE || EPI == GRIT_GEMM_BIAS) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                v4f v = acc[i][j];
                if constexpr (EPI == GRIT_GEMM_BIAS) v += bias4[j];
                put(i, j, v);
            }
        flush(g.C, g.ldc);
    } else if constexpr (EPI == GRIT_GEMM_BIAS_GELU) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] += bias4[j];
        if (g.aux) {  // the pre-activation, kept for the backward pass (not needed under no_grad / in frozen stages)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) put(i, j, acc[i][j]);
            flush(g.aux, g.ldaux, (g.nt_aux & 1) != 0);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const v2f lo = gelu2(v2f{acc[i][j][0], acc[i][j][1]}), hi = gelu2(v2f{acc[i][j][2], acc[i][j][3]});
                put(i, j, v4f{lo[0], lo[1], hi[0], hi[1]});
            }
        flush(g.C, g.ldc, (g.nt_aux & 2) != 0);
    } else if constexpr (EPI == GRIT_GEMM_BIAS_RELU_DROP) {
        // C = dropout(relu(bf16(acc + bias))): the Linear's result rounded to bf16 as an unfused Linear stores it, then exactly what
        // grit_relu_dropout_fwd computes from it (same hash over the element index m * N + n, same product)
        const unsigned long long seed = g.drop_p > 0.f ? *g.seed : 0ull;
        const float inv_keep = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const unsigned long long m = (unsigned long long)(mw + 16 * i + l15);
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const v4f a = acc[i][j] + bias4[j];
                const unsigned long long e0 = m * (unsigned long long)g.N + (unsigned long long)(nw + 16 * j + 4 * lq);
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = (float)(__bf16)a[e];
                    const float k = g.drop_p > 0.f ? relu_keep_scale(seed, e0 + e, g.drop_p, inv_keep) : 1.0f;
                    v[e] = x > 0.f ? x * k : 0.f;
                }
                put(i, j, v);
            }
        }
        flush(g.C, g.ldc);
    } else if constexpr (EPI == GRIT_GEMM_DRELU) {
        // C = aux > 0 ? bf16(acc) / (1 - p) : 0 with aux = the dropout(relu(.)) OUTPUT of the forward pass (positive exactly where the
        // unit was active and kept): the input gradient of the following Linear with the backward of ReLU + dropout in its epilogue --
        // bit for bit grit_relu_dropout_bwd applied to the stored bf16 product
        {
            const int chunk = lane & 7;
#pragma unroll
            for (int it = 0; it < WTM / 8; ++it) {
                const int row = it * 8 + (lane >> 3);
                const int m = min(mw + row, g.M - 1);
                const __bf16* src = g.aux + (size_t)m * g.ldaux + nw + ((chunk ^ (row & 7)) * 8);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(eb + it * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const float inv_keep = g.drop_p > 0.f ? 1.0f / (1.0f - g.drop_p) : 1.0f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = 16 * i + l15;
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int chunk = (2 * j + (lq >> 1)) ^ (row & 7);
                const v4bf y = *reinterpret_cast<const v4bf*>(eb + row * 128 + chunk * 16 + (lq & 1) * 8);
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (float)y[e] > 0.f ? (float)(__bf16)acc[i][j][e] * inv_keep : 0.f;
                put(i, j, v);
            }
        }
        flush(g.C, g.ldc);
    } else if constexpr (EPI == GRIT_GEMM_BIAS_RES) {
        // C = residual + factor[sample of the row] * bf16(acc + bias): the residual tile (g.aux) comes into the wave's transpose image as
        // whole 128-byte row segments, every lane combines its 8-byte pieces in place -- the branch rounded to bf16 as an unfused Linear
        // would store it, the sum in fp32 (multiply and add not fused) rounded once: bit for bit grit_add_layernorm_fwd's x
        {
            const int chunk = lane & 7;
#pragma unroll
            for (int it = 0; it < WTM / 8; ++it) {
                const int row = it * 8 + (lane >> 3);
                const int m = min(mw + row, g.M - 1);
                const __bf16* src = g.aux + (size_t)m * g.ldaux + nw + ((chunk ^ (row & 7)) * 8);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(eb + it * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = 16 * i + l15;
            const float sc = g.row_scale ? g.row_scale[min(mw + row, g.M - 1) / g.rows_per_sample] : 1.f;
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int chunk = (2 * j + (lq >> 1)) ^ (row & 7);
                const v4bf r = *reinterpret_cast<const v4bf*>(eb + row * 128 + chunk * 16 + (lq & 1) * 8);
                const v4f a = acc[i][j] + bias4[j];
                v4f v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __fadd_rn((float)r[e], __fmul_rn((float)(__bf16)a[e], sc));
                put(i, j, v);
            }
        }
        flush(g.C, g.ldc);
    } else {  // GRIT_GEMM_DGELU
        // the pre-activation tile comes in the way the result goes out: whole 128-byte row segments (one DMA piece = 8 rows) into
        // the wave's transpose image, from where every lane picks its 8-byte pieces -- accumulator-shaped global loads (16 rows x
        // 32 bytes per instruction) cost 11k cycles per tile more (tools/micro/gemm_stamps.hip)
        {
            const int chunk = lane & 7;
#pragma unroll
            for (int it = 0; it < WTM / 8; ++it) {
                const int row = it * 8 + (lane >> 3);
                const int m = min(mw + row, g.M - 1);
                // image chunk (lane & 7) of row holds logical chunk (lane & 7) ^ (row & 7): permute on the source address
                const __bf16* src = g.aux + (size_t)m * g.ldaux + nw + ((chunk ^ (row & 7)) * 8);
                if (g.nt_aux & 8)  // read once: do not let the 210 MB of pre-activations displace the operand panels in L2
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(eb + it * 1024), 16, 0, 2);
                else
                    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(eb + it * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        v4f cs[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) cs[j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = 16 * i + l15;
            const bool live = full_rows || mw + row < g.M;
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int chunk = (2 * j + (lq >> 1)) ^ (row & 7);
                const v4bf x = *reinterpret_cast<const v4bf*>(eb + row * 128 + chunk * 16 + (lq & 1) * 8);
                v2f dlo = v2f{(float)x[0], (float)x[1]}, dhi = v2f{(float)x[2], (float)x[3]};
                dlo = dgelu2(dlo); dhi = dgelu2(dhi);
                const v4f v = {acc[i][j][0] * dlo[0], acc[i][j][1] * dlo[1], acc[i][j][2] * dhi[0], acc[i][j][3] * dhi[1]};
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[j][r] += live ? v[r] : 0.f;  // bias gradient from the unrounded products
                put(i, j, v);  // in place: this lane's piece of the image
            }
        }
        flush(g.C, g.ldc, (g.nt_aux & 4) != 0);
        // column sums over this wave's WTM rows: fold the 16 token lanes of each quarter
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cs[j][r] = row_sum16(cs[j][r]);
            }
        // (a wave whose rows all lie past M has no slab in colsum [ceil(M / 128), N]: found in round 3 as 4 KB of zeros written
        // behind the partials of a 37-row problem)
        if (l15 == 0 && mw < g.M) {
            float* dst = g.colsum + (size_t)(mw / WTM) * g.N + nw + 4 * lq;
#pragma unroll
            for (int j = 0; j < NTL; ++j) *reinterpret_cast<v4f*>(dst + 16 * j) = cs[j];
        }
    }
    GRIT_STAMP(7)
}

// =====================================================================================================================
// Persistent "ping-pong" kernel (variant 5): 256 x 256 tiles, K step 32, four-slot LDS ring (128 KB) + a 32 KB epilogue
// image, one 8-wave workgroup per CU that walks its share of the tiles.
//
// What the per-tile kernel above leaves on the table (tools/micro/gemm_stamps.hip, M = 51200, N = 2048, K = 512): of the 33k
// cycles a 256 x 256 tile takes, 16.4k are MFMA; ~3k are the prologue (first DMAs of every tile), ~4k DMA waits + barrier
// skew inside the loop (a two-slot ring gives a DMA one K step to land), ~4k fragment-read latency that both waves of a SIMD
// pay at the same moment, ~6k the epilogue.  Here
//   * the K steps of consecutive tiles form ONE stream: the DMAs of the next tile's first steps are issued during the last
//     steps of the current one, three steps ahead (a DMA has ~3k cycles to land, counted vmcnt, never drained in the loop);
//   * the two waves of every SIMD run half a step apart (waves 4-7 pass one extra barrier up front): while one wave reads its
//     fragments from LDS the other issues its 32 MFMAs, so the matrix pipe does not idle during LDS latency and barrier skew
//     (MI355X_MICROARCH.md "Two waves per SIMD", cdna_hip_programming.md 8-phase template);
//   * the epilogue goes through a private 4 KB LDS image per wave, so the ring keeps filling meanwhile.
//
// Barrier phases (b = barrier index; L(q) = fragment reads of step q, M(q) = its MFMAs + the DMAs of step q + 3):
//   waves 0-3:  L(q) in [2q, 2q+1)    M(q) in [2q+1, 2q+2)        waves 4-7:  L(q) in [2q+1, 2q+2)   M(q) in [2q+2, 2q+3)
//   WAR: the DMAs of step q+3 (issued in L(q), i.e. after barrier 2q) overwrite slot (q-1) % 4, last read in L(q-1) -- finished
//        (lgkmcnt(0)) before barrier 2q by both groups;
//   RAW: step s is first read at barrier 2s; its DMAs were issued in L(s-3); both groups wait vmcnt(8) (the DMAs of steps
//        s+1, s+2 may stay in flight) before barrier 2s: waves 0-3 at the end of M(s-1), waves 4-7 at the end of L(s-1).
template <int EPI>
__global__ __launch_bounds__(512, 2)
void gemm_pp_bf16(const GemmArgs g) {
    constexpr int BM = 256, BN = 256, BK = 32, NSLOT = 4;
    constexpr int ROWB = BK * 2;                       // 64-byte rows, 4 chunks of 16 B
    constexpr int A_BYTES = BM * ROWB, SLOT = (BM + BN) * ROWB;
    constexpr int MT = 8, NTL = 4;                     // wave tile 128 x 64
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* epi = lds + NSLOT * SLOT;                    // 8 x 4 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wu = wave & 3;          // group = row half, wu = 64-column strip
    const int l15 = lane & 15, lq = lane >> 4;

    // tiles of this workgroup: the 8 XCD groups (blockIdx % 8) own contiguous bands of the row-major tile list; inside a band
    // the workgroups take tiles round-robin, so the CUs of an XCD work on neighbouring tiles (shared A panels, B in L2)
    const int ntiles = g.tiles_m * g.tiles_n;
    const int ngroups = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % ngroups, idx = blockIdx.x / ngroups, per_xcd = ((int)gridDim.x - xcd + ngroups - 1) / ngroups;
    const int band_lo = (int)((long long)ntiles * xcd / ngroups), band_hi = (int)((long long)ntiles * (xcd + 1) / ngroups);
    const int my_tiles = band_lo + idx < band_hi ? (band_hi - band_lo - idx + per_xcd - 1) / per_xcd : 0;
    if (my_tiles == 0) return;
    const int KT = g.K / BK;
    const int total = my_tiles * KT;
    auto tile_of = [&](int i) { return band_lo + idx + i * per_xcd; };

    // ---- DMA side: this wave moves pieces {wave, wave + 8} of A and of B (16 rows x 64 B each) per step
    const int prow = lane >> 2;
    const int pchunk = (lane & 3) ^ chunk_swizzle<BK>(prow);
    const __bf16* asrc[2];
    const __bf16* bsrc[2];
    int lti = 0, lks = 0, lslot = 0;  // DMA side of the stream: tile, k step and ring slot of the next step to be fetched
    auto set_load_tile = [&](int i) {
        const int t = tile_of(i), tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 16 * (wave + 8 * j) + prow;
            asrc[j] = g.A + (size_t)min(tm * BM + r, g.M - 1) * g.lda + pchunk * 8;
            bsrc[j] = g.B + (size_t)(tn * BN + r) * g.ldb + pchunk * 8;
        }
    };
    set_load_tile(0);
    auto dma = [&](int piece) {  // piece 0,1: A; 2,3: B -- of the step (lti, lks), into slot lslot
        char* slot = lds + lslot * SLOT;
        const int j = piece & 1;
        if (piece < 2)
            __builtin_amdgcn_global_load_lds((gptr_t)(asrc[j] + lks * BK), (lptr_t)(slot + (wave + 8 * j) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[j] + lks * BK), (lptr_t)(slot + A_BYTES + (wave + 8 * j) * 1024), 16, 0, 0);
    };
    auto advance_load = [&]() {
        lslot = (lslot + 1) & (NSLOT - 1);
        if (++lks == KT) {
            lks = 0;
            if (++lti < my_tiles) set_load_tile(lti);
        }
    };

    // ---- fragment reads
    const int foff = l15 * ROWB + ((lq ^ chunk_swizzle<BK>(l15)) * 16);
    const int a_wave = grp * 128 * ROWB, b_wave = A_BYTES + wu * 64 * ROWB;

    v4f acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    // ---- epilogue of one finished tile (no barrier inside: it only lengthens this wave's phase)
    char* eb = epi + wave * 4096;
    auto epilogue = [&](int ti) {
        const int t = tile_of(ti), tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
        const int mw = tm * BM + grp * 128, nw = tn * BN + wu * 64;
        v4f bias4[NTL];
        if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const v4bf b = *reinterpret_cast<const v4bf*>(g.bias + nw + 16 * j + 4 * lq);
                bias4[j] = v4f{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
        }
        const bool full_rows = mw + 128 <= g.M;
        auto put = [&](int il, int j, const v4f& v) {  // il = m-tile within the 32-row chunk
            v4bf p;
            p[0] = (__bf16)v[0]; p[1] = (__bf16)v[1]; p[2] = (__bf16)v[2]; p[3] = (__bf16)v[3];
            const int row = 16 * il + l15;
            const int chunk = (2 * j + (lq >> 1)) ^ (row & 7);
            *reinterpret_cast<v4bf*>(eb + row * 128 + chunk * 16 + (lq & 1) * 8) = p;
        };
        auto flush = [&](__bf16* dst, long ld, int c) {  // rows 32 c .. 32 c + 31 of the wave tile
            asm volatile("" ::: "memory");
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                const uint4 v = *reinterpret_cast<const uint4*>(eb + row * 128 + ((chunk ^ (row & 7)) * 16));
                const int m = mw + 32 * c + row;
#ifdef GRIT_GEMM_NOSTORE  // diagnostic build: everything but the global stores (only rows that cannot exist are "stored")
                if (m < 0) *reinterpret_cast<uint4*>(dst + (size_t)m * ld + nw + chunk * 8) = v;
#else
                if (full_rows || m < g.M) *reinterpret_cast<uint4*>(dst + (size_t)m * ld + nw + chunk * 8) = v;
#endif
            }
            asm volatile("" ::: "memory");
        };
        v4f cs[NTL];
        if constexpr (EPI == GRIT_GEMM_DGELU) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) cs[j] = v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (EPI == GRIT_GEMM_NONE || EPI == GRIT_GEMM_BIAS) {
#pragma unroll
                for (int il = 0; il < 2; ++il)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        v4f v = acc[2 * c + il][j];
                        if constexpr (EPI == GRIT_GEMM_BIAS) v += bias4[j];
                        put(il, j, v);
                    }
                flush(g.C, g.ldc, c);
            } else if constexpr (EPI == GRIT_GEMM_BIAS_GELU) {
#pragma unroll
                for (int il = 0; il < 2; ++il)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) acc[2 * c + il][j] += bias4[j];
                if (g.aux) {
#pragma unroll
                    for (int il = 0; il < 2; ++il)
#pragma unroll
                        for (int j = 0; j < NTL; ++j) put(il, j, acc[2 * c + il][j]);
                    flush(g.aux, g.ldaux, c);
                }
#pragma unroll
                for (int il = 0; il < 2; ++il)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        v4f v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_f(acc[2 * c + il][j][r]);
                        put(il, j, v);
                    }
                flush(g.C, g.ldc, c);
            } else {  // GRIT_GEMM_DGELU: pre-activation pieces straight from memory in accumulator shape
#pragma unroll
                for (int il = 0; il < 2; ++il) {
                    const int m = mw + 32 * c + 16 * il + l15;
                    const bool live = full_rows || m < g.M;
                    const __bf16* xr = g.aux + (size_t)min(m, g.M - 1) * g.ldaux + nw + 4 * lq;
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        const v4bf x = *reinterpret_cast<const v4bf*>(xr + 16 * j);
                        v4f v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc[2 * c + il][j][r] * dgelu_f((float)x[r]);
                            cs[j][r] += live ? v[r] : 0.f;
                        }
                        put(il, j, v);
                    }
                }
                flush(g.C, g.ldc, c);
            }
        }
        if constexpr (EPI == GRIT_GEMM_DGELU) {
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    cs[j][r] = row_sum16(cs[j][r]);
                }
            if (l15 == 0 && mw < g.M) {
                float* dst = g.colsum + (size_t)(mw / 128) * g.N + nw + 4 * lq;
#pragma unroll
                for (int j = 0; j < NTL; ++j) *reinterpret_cast<v4f*>(dst + 16 * j) = cs[j];
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    };

    // ---- prologue: steps 0, 1, 2 in flight, step 0 landed for everybody
#pragma unroll
    for (int q = 0; q < 3; ++q)
        if (q < total) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) dma(pc);
            advance_load();
        }
    if (total > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: waves 4-7 run one phase behind
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");

    int ks = 0, ti = 0, cslot = 0;  // compute side of the stream
#ifdef GRIT_GEMM_STAMPS
    unsigned long long ph_l = 0, ph_b1 = 0, ph_m = 0, ph_b2 = 0, ph_epi = 0, ph_t = __builtin_amdgcn_s_memtime();
    const unsigned long long ph_start = ph_t;
#define GRIT_PH(acc_var) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_var += now_ - ph_t; ph_t = now_; }
#else
#define GRIT_PH(acc_var)
#endif
    for (int q = 0; q < total; ++q) {
        // ---- L(q): the DMA pieces of step q + 3 (their issue is slow -- ~100 cycles each -- and must not sit between this
        // wave's MFMAs: the partner wave owns the matrix pipe during this phase), then the 12 fragments of this step
        const bool more = q + 3 < total;
        if (more) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) dma(pc);
            advance_load();
        }
        const char* sb = lds + cslot * SLOT;
        v8bf wf[NTL], xf[MT];
#pragma unroll
        for (int j = 0; j < NTL; ++j) wf[j] = *reinterpret_cast<const v8bf*>(sb + b_wave + j * 16 * ROWB + foff);
#pragma unroll
        for (int i = 0; i < MT; ++i) xf[i] = *reinterpret_cast<const v8bf*>(sb + a_wave + i * 16 * ROWB + foff);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grp == 1) {  // step q + 1 (read by waves 0-3 right after the coming barrier) must have landed
            if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        GRIT_PH(ph_l)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GRIT_PH(ph_b1)
        // ---- M(q): 32 MFMAs
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) {  // step q + 1 is read by this group right after the coming barrier
            if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        cslot = (cslot + 1) & (NSLOT - 1);
        GRIT_PH(ph_m)
        if (++ks == KT) {
            epilogue(ti);
            ks = 0;
            ++ti;
            GRIT_PH(ph_epi)
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GRIT_PH(ph_b2)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#ifdef GRIT_GEMM_STAMPS
    if (g.stamps && (threadIdx.x & 63) == 0) {
        unsigned long long* o = g.stamps + ((size_t)blockIdx.x * 8 + wave) * 16;
        o[0] = ph_start; o[7] = __builtin_amdgcn_s_memtime();
        o[8] = ph_l; o[9] = ph_b1; o[10] = ph_m; o[11] = ph_b2; o[12] = ph_epi; o[13] = (unsigned long long)total;
    }
#endif
#undef GRIT_PH
}

int launch_pp(const GemmArgs& a, int epilogue, hipStream_t st) {
    constexpr int LDS = 4 * (256 + 256) * 64 + 8 * 4096;  // 160 KB: the whole CU
    GemmArgs g = a;
    g.tiles_m = (a.M + 255) / 256;
    g.tiles_n = a.N / 256;
    static grit_detail::PerDevice<int> cus_pd; int& cus = cus_pd();
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GRIT_ERR_LAUNCH;
        cus = prop.multiProcessorCount;
    }
    const int ntiles = g.tiles_m * g.tiles_n;
    const dim3 grid(ntiles < cus ? ntiles : cus), block(512);
#define GRIT_GEMM_LAUNCH_PP(E)                                                                                       \
    {                                                                                                                \
        auto kern = gemm_pp_bf16<E>;                                                                                 \
        static grit_detail::PerDevice<bool> attr_done_pd; bool& attr_done = attr_done_pd();                                                                               \
        if (!attr_done) {                                                                                            \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) \
                return GRIT_ERR_LAUNCH;                                                                              \
            attr_done = true;                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, grid, block, LDS, st, g);                                                           \
    }
    switch (epilogue) {
        case GRIT_GEMM_NONE: GRIT_GEMM_LAUNCH_PP(GRIT_GEMM_NONE) break;
        case GRIT_GEMM_BIAS: GRIT_GEMM_LAUNCH_PP(GRIT_GEMM_BIAS) break;
        case GRIT_GEMM_BIAS_GELU: GRIT_GEMM_LAUNCH_PP(GRIT_GEMM_BIAS_GELU) break;
        case GRIT_GEMM_DGELU: GRIT_GEMM_LAUNCH_PP(GRIT_GEMM_DGELU) break;
        default: return GRIT_ERR_BAD_ARG;
    }
#undef GRIT_GEMM_LAUNCH_PP
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

template <int BM, int BN, int BK, int WM, int WN, int NSTAGE>
int launch(const GemmArgs& a, int epilogue, hipStream_t st) {
    constexpr int STAGE = (BM + BN) * BK * 2;
    constexpr int LDS = NSTAGE * STAGE;
    GemmArgs g = a;
    g.tiles_m = (a.M + BM - 1) / BM;
    g.tiles_n = a.N / BN;
    const dim3 grid(g.tiles_m * g.tiles_n), block(WM * WN * 64);
#define GRIT_GEMM_LAUNCH(E)                                                                                          \
    {                                                                                                                \
        auto kern = gemm_nt_bf16<BM, BN, BK, WM, WN, NSTAGE, E>;                                                     \
        static grit_detail::PerDevice<bool> attr_done_pd; bool& attr_done = attr_done_pd(); /* idempotent: racing threads set the same value */                           \
        if (!attr_done) {                                                                                            \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) \
                return GRIT_ERR_LAUNCH;                                                                              \
            attr_done = true;                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, grid, block, LDS, st, g);                                                           \
    }
    switch (epilogue) {
        case GRIT_GEMM_NONE: GRIT_GEMM_LAUNCH(GRIT_GEMM_NONE) break;
        case GRIT_GEMM_BIAS: GRIT_GEMM_LAUNCH(GRIT_GEMM_BIAS) break;
        case GRIT_GEMM_BIAS_GELU: GRIT_GEMM_LAUNCH(GRIT_GEMM_BIAS_GELU) break;
        case GRIT_GEMM_DGELU: GRIT_GEMM_LAUNCH(GRIT_GEMM_DGELU) break;
        case GRIT_GEMM_BIAS_RES:  // (one tile shape carries it: the narrow outputs of the stage-0 map)
            if constexpr (BM == 256 && BN == 128 && BK == 32 && NSTAGE == 3) GRIT_GEMM_LAUNCH(GRIT_GEMM_BIAS_RES)
            else return GRIT_ERR_UNSUPPORTED;
            break;
        case GRIT_GEMM_BIAS_RELU_DROP:  // (the short-map tile carries the two ReLU + dropout epilogues: the decoders' FFNs)
            if constexpr (BM == 64 && BN == 64 && BK == 64) GRIT_GEMM_LAUNCH(GRIT_GEMM_BIAS_RELU_DROP)
            else return GRIT_ERR_UNSUPPORTED;
            break;
        case GRIT_GEMM_DRELU:
            if constexpr (BM == 64 && BN == 64 && BK == 64) GRIT_GEMM_LAUNCH(GRIT_GEMM_DRELU)
            else return GRIT_ERR_UNSUPPORTED;
            break;
        default: return GRIT_ERR_BAD_ARG;
    }
#undef GRIT_GEMM_LAUNCH
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // namespace

// (set by grit_gemm_bf16_nt_rows around its call: the plain entry point keeps its signature)
static thread_local const float* g_row_scale = nullptr;
static thread_local int g_rows_per_sample = 0;

extern "C" int grit_gemm_bf16_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                 int epilogue, const void* bias, void* aux, long ldaux, float* colsum, int variant,
                                 void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return GRIT_ERR_BAD_ARG;
    if ((epilogue == GRIT_GEMM_BIAS || epilogue == GRIT_GEMM_BIAS_GELU) && !bias) return GRIT_ERR_BAD_ARG;
    if (epilogue == GRIT_GEMM_DGELU && (!aux || !colsum)) return GRIT_ERR_BAD_ARG;
    // 16-byte DMA pieces and row stores: leading dimensions in multiples of 8 elements, 16-byte aligned bases
    if ((lda | ldb | ldc | (aux ? ldaux : 0)) & 7) return GRIT_ERR_UNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)aux | (uintptr_t)bias) & 15) return GRIT_ERR_UNSUPPORTED;
    if ((N % 128 && !(variant == 12 && N % 64 == 0)) || K % 32) return GRIT_ERR_UNSUPPORTED;
    GemmArgs a;
    a.A = (const __bf16*)A; a.lda = lda; a.B = (const __bf16*)B; a.ldb = ldb; a.C = (__bf16*)C; a.ldc = ldc;
    a.bias = (const __bf16*)bias; a.aux = (__bf16*)aux; a.ldaux = ldaux; a.colsum = colsum;
    static const int nt_aux = [] { const char* e = getenv("GRIT_GEMM_NT_AUX"); return e ? atoi(e) : 15; }();
    a.nt_aux = nt_aux;
    a.M = M; a.N = N; a.K = K; a.tiles_m = a.tiles_n = 0;
    a.row_scale = g_row_scale; a.rows_per_sample = g_rows_per_sample;
    a.seed = nullptr; a.drop_p = 0.f;
#ifdef GRIT_GEMM_STAMPS
    a.stamps = nullptr;  // diagnostic builds set it through launch<>() directly
#endif
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) variant = (N % 256 == 0 && K % 64 == 0) ? 4 : 1;  // measured: tools/bench_gemm.py
    if (variant >= 10 && variant <= 13 && epilogue == GRIT_GEMM_DGELU) return GRIT_ERR_UNSUPPORTED;  // (column sums are laid out per 128 rows)
    switch (variant) {
        case 1: return launch<256, 128, 32, 2, 2, 3>(a, epilogue, st);
        case 2: return (K % 64) ? GRIT_ERR_UNSUPPORTED : launch<256, 128, 64, 2, 2, 2>(a, epilogue, st);  // needs 96 KB: 1 WG / CU
        case 3: return launch<256, 128, 32, 2, 2, 4>(a, epilogue, st);  // deeper ring, 96 KB
        case 4: return (K % 64 || N % 256) ? GRIT_ERR_UNSUPPORTED : launch<256, 256, 64, 2, 4, 2>(a, epilogue, st);  // 8 waves, 128 KB
        case 5: return (N % 256) ? GRIT_ERR_UNSUPPORTED : launch_pp(a, epilogue, st);  // persistent ping-pong, 160 KB
        // short maps (the decoders' 640 .. 12 800 rows): small tiles, two or three workgroups per CU, so that a 4 800 x 512 output is 300-600
        // workgroups instead of 38 (round 6)
        case 10: return (K % 64) ? GRIT_ERR_UNSUPPORTED : launch<64, 128, 64, 2, 2, 3>(a, epilogue, st);   // 72 KB: two per CU
        case 11: return (K % 64) ? GRIT_ERR_UNSUPPORTED : launch<128, 128, 64, 2, 2, 2>(a, epilogue, st);  // 64 KB: two per CU
        case 12: return (K % 64 || N % 64) ? GRIT_ERR_UNSUPPORTED : launch<64, 64, 64, 4, 1, 3>(a, epilogue, st);  // 48 KB: three per CU
        case 13: return launch<64, 128, 32, 2, 2, 4>(a, epilogue, st);   // 48 KB, K step 32, four slots
        case 7:  // four waves, 128 x 128 wave tiles (gemm_w4.hip)
            return grit_detail::gemm_w4_launch(A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, aux, ldaux, colsum,
                                               epilogue == GRIT_GEMM_BIAS ? 0 : nt_aux, stream);
        case 9:  // the same kernel with the tile height (256 or 224 rows) that fills the CUs better: grit_gemm_w4_tile_rows
            return grit_detail::gemm_w4_launch(A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, aux, ldaux, colsum,
                                               epilogue == GRIT_GEMM_BIAS ? 0 : nt_aux, stream, 0);
        default: return GRIT_ERR_BAD_ARG;
    }
}

extern "C" int grit_gemm_w4_tile_rows(int M, int N) { return grit_detail::gemm_w4_tile_rows(M, N); }

// C = residual + row_scale[sample of the row] * (A B^T + bias): the output projection of a Swin branch with its residual connection
// (persistent four-wave kernel, tile height chosen by shape; outputs that are not a multiple of 256 columns wide -- the stage-0 map's
// 128 -- on the 256 x 128 tiles of the per-tile kernel).
extern "C" int grit_gemm_bf16_nt_res(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                     const void* bias, const void* residual, long ldres, const float* row_scale, int rows_per_sample,
                                     void* stream) {
    if (!A || !B || !C || !bias || !residual || M <= 0 || N <= 0 || K <= 0) return GRIT_ERR_BAD_ARG;
    if (row_scale && rows_per_sample <= 0) return GRIT_ERR_BAD_ARG;
    if ((lda | ldb | ldc | ldres) & 7) return GRIT_ERR_UNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)residual | (uintptr_t)bias) & 15) return GRIT_ERR_UNSUPPORTED;
    if (N % 256 == 0 && K % 64 == 0)
        return grit_detail::gemm_w4_launch(A, lda, B, ldb, C, ldc, M, N, K, GRIT_GEMM_BIAS_RES, bias, const_cast<void*>(residual), ldres,
                                           nullptr, 0, stream, 0, row_scale, rows_per_sample);
    if (N % 128 || K % 32) return GRIT_ERR_UNSUPPORTED;
    GemmArgs a;
    a.A = (const __bf16*)A; a.lda = lda; a.B = (const __bf16*)B; a.ldb = ldb; a.C = (__bf16*)C; a.ldc = ldc;
    a.bias = (const __bf16*)bias; a.aux = (__bf16*)const_cast<void*>(residual); a.ldaux = ldres; a.colsum = nullptr; a.nt_aux = 0;
    a.M = M; a.N = N; a.K = K; a.tiles_m = a.tiles_n = 0;
    a.row_scale = row_scale; a.rows_per_sample = rows_per_sample;
    a.seed = nullptr; a.drop_p = 0.f;
#ifdef GRIT_GEMM_STAMPS
    a.stamps = nullptr;
#endif
    return launch<256, 128, 32, 2, 2, 3>(a, GRIT_GEMM_BIAS_RES, (hipStream_t)stream);
}

// The position-wise FFNs of the decoders (Linear -> ReLU -> dropout -> Linear) on the short-map tiles, ReLU + dropout in the epilogues:
//   GRIT_GEMM_BIAS_RELU_DROP   C = dropout(relu(bf16(A B^T + bias)))            (forward of the first Linear; aux unused)
//   GRIT_GEMM_DRELU            C = aux > 0 ? bf16(A B^T) / (1 - p) : 0           (input gradient of the second Linear; aux = the forward's C)
extern "C" int grit_gemm_bf16_nt_relu(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                      int epilogue, const void* bias, const void* aux, long ldaux, float drop_p, const uint64_t* seed_dev,
                                      void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return GRIT_ERR_BAD_ARG;
    if (epilogue == GRIT_GEMM_BIAS_RELU_DROP ? (!bias || (drop_p > 0.f && !seed_dev)) : (epilogue != GRIT_GEMM_DRELU || !aux)) return GRIT_ERR_BAD_ARG;
    if ((lda | ldb | ldc | (aux ? ldaux : 0)) & 7) return GRIT_ERR_UNSUPPORTED;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)aux | (uintptr_t)bias) & 15) return GRIT_ERR_UNSUPPORTED;
    if (N % 64 || K % 64) return GRIT_ERR_UNSUPPORTED;
    if (epilogue == GRIT_GEMM_BIAS_RELU_DROP && ldc != N) return GRIT_ERR_UNSUPPORTED;  // (the dropout hash runs over the index m * N + n)
    GemmArgs a;
    a.A = (const __bf16*)A; a.lda = lda; a.B = (const __bf16*)B; a.ldb = ldb; a.C = (__bf16*)C; a.ldc = ldc;
    a.bias = (const __bf16*)bias; a.aux = (__bf16*)const_cast<void*>(aux); a.ldaux = ldaux; a.colsum = nullptr; a.nt_aux = 0;
    a.M = M; a.N = N; a.K = K; a.tiles_m = a.tiles_n = 0;
    a.row_scale = nullptr; a.rows_per_sample = 0;
    a.seed = (const unsigned long long*)seed_dev; a.drop_p = drop_p;
#ifdef GRIT_GEMM_STAMPS
    a.stamps = nullptr;
#endif
    return launch<64, 64, 64, 4, 1, 3>(a, epilogue, (hipStream_t)stream);
}

// GRIT_GEMM_DGELU / GRIT_GEMM_BIAS_GELU with the per-sample factors of the rows of A (see GemmArgs::row_scale): eight-wave variants only.
extern "C" int grit_gemm_bf16_nt_rows(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                      int epilogue, const void* bias, void* aux, long ldaux, float* colsum, const float* row_scale,
                                      int rows_per_sample, int variant, void* stream) {
    if (epilogue != GRIT_GEMM_DGELU && epilogue != GRIT_GEMM_BIAS_GELU) return GRIT_ERR_BAD_ARG;
    if (row_scale && (rows_per_sample <= 0 || variant > 4)) return GRIT_ERR_BAD_ARG;
    g_row_scale = row_scale; g_rows_per_sample = rows_per_sample;
    const int st = grit_gemm_bf16_nt(A, lda, B, ldb, C, ldc, M, N, K, epilogue, bias, aux, ldaux, colsum, variant, stream);
    g_row_scale = nullptr; g_rows_per_sample = 0;
    return st;
}
