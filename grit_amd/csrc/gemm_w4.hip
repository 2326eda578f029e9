// Persistent bf16 MFMA GEMM, four waves per workgroup with 128 x 128 wave tiles (gfx950) -- variant 7 of grit_gemm_bf16_nt.
//
//   C[M, N] = epilogue( A[M, K] . B[N, K]^T )      A, B, C bf16 row-major (K-contiguous operands), fp32 accumulation
//
// Why another tiling: the eight-wave kernels of gemm.hip (two waves per SIMD, 128 x 64 wave tiles) spend a third of every K step
// in barriers and LDS latency that the two waves of a SIMD pay together: ~50 % matrix-pipe utilisation inside the loop
// (tools/micro/gemm_ps_bench.hip, profiles/r04/gemm_w4.txt).  The platform library's kernel for these shapes runs its loop at ~85 %
// with ONE wave per SIMD and 128 x 128 per wave (256 accumulators in the AGPR half of the unified file): every byte read from LDS
// feeds twice the MFMAs, the operand fragments of the next 32-deep half step are fetched between the MFMAs of the current one, the
// LDS-DMA transfers of the step after next go out one at a time between MFMAs, and a 64-deep K step needs three barriers.  This
// kernel follows that loop shape with its own data layout, and -- what the library does not do at K = 256 / 512, where a tile has
// only 4-8 K steps -- runs the K steps of all of a workgroup's tiles as ONE stream: the transfers of the next tile's first steps are
// in flight and its first fragments in registers before the epilogue of the current tile starts (per tile at K = 512: 17.8 us as a
// per-tile launch, of which 11 us are the loop).
//   * LDS: two buffers of 64 KB (A 256 rows x 128 B, B 256 rows x 128 B), lane-linear DMA image (a transfer instruction = 8 rows),
//     16-byte chunks permuted by (row >> 1) & 7 on the SOURCE address and on the read: conflict-free ds_read_b128 (gemm_math.h);
//     + 8 KB per wave for the epilogue's transpose (32 rows x 256 B): 160 KB, one workgroup per CU;
//   * transfers: buffer_load_dwordx4 ... lds with ONE per-lane offset register per operand and scalar offsets per piece;
//   * fragment reads, MFMAs and the epilogue's LDS traffic are inline asm: hipcc puts vmcnt(0) in front of LDS accesses it sees
//     behind an LDS-DMA, and left to the register allocator the 256 accumulators wander between the two halves of the register
//     file (942 v_accvgpr moves in the loop of the builtin version).  Every statement of a K step is volatile asm or a
//     side-effecting builtin, so the source order IS the schedule; waits are placed by hand.
#include <hip/hip_runtime.h>
#include "per_device.h"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "../../include/grit_hip.h"
#include "gemm_math.h"
#include "gemm_launchers.h"

namespace {

struct W4Args {
    const __bf16* A; long lda;
    const __bf16* B; long ldb;
    __bf16* C; long ldc;
    const __bf16* bias;
    __bf16* aux; long ldaux;   // BIAS_GELU: pre-activation out (may be NULL); DGELU: pre-activation in; BIAS_RES: the residual map in
    const float* row_scale;    // BIAS_RES: per-sample factors of the branch (drop path) or NULL; rows_per_sample >= the tile height
    int rows_per_sample;
    float* colsum;             // DGELU: [2 tiles_m, N] column sums of the result per 128-row wave block (rows of a shifted last tile
                               // that belong to its neighbour are left out)
    int nt;                    // non-temporal accesses, bits as GRIT_GEMM_NT_AUX
    int stagger;               // 1: workgroups with fewer tiles than the busiest start late, spread over one tile time
    int M, N, K, tiles_m, tiles_n;
};

template <int OFF> __device__ __forceinline__ v8bf lds_read16_off(unsigned addr) {
    v8bf v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

template <class F, int... Is>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}


// MI = 16-row token blocks per wave: 8 (256-row tiles, the kernel as described above) or 7 (224-row tiles, round 6).  Why 7: the products
// with N = C output columns (proj, fc2 forward, fc1 / qkv input gradients of every Swin block) have 200 x N / 256 tiles of 256 rows --
// 400, 800, 200 at the three trainable stages: on 256 CUs the last round is 56 %, 12 %, 78 % full and the kernel runs 2, 4, 1 rounds
// of 8 block rows.  With 224-row tiles the same maps are 458, 915, 232 tiles: 2, 4, 1 rounds of 7 block rows -- 12.5 % less time, no
// hand-over between workgroups (the stream-K form of this kernel lost 6-84 %: tools/micro/r06_stream_k.patch).
template <int EPI, int MI>
__global__ __launch_bounds__(256, 1)
void gemm_w4_bf16(const W4Args g) {
    static_assert(MI == 7 || MI == 8, "wave tiles of 112 or 128 rows");
    constexpr int BM = 32 * MI, BN = 256, BK = 64;
    constexpr int NMFMA = 16 * MI;                   // MFMAs of one 64-deep K step of a wave
    constexpr int NDMA = MI + 8;                     // transfers of one K step issued by a wave (MI of A, 8 of B)
    constexpr int ROWB = BK * 2;                     // 128-byte staged rows
    constexpr int A_BYTES = BM * ROWB, BUF = (BM + BN) * ROWB;  // 32 KB + 32 KB
    constexpr int IMG = 8192;                        // per wave: 32 rows x 256 B of finished bf16 output
    // stores (and compiler-visible loads) one epilogue puts between the transfers of step s + 2 and those of step s + 3
    // (EXACT counts -- the wait behind a tile's first step allows this many younger operations: 4 MI row-segment stores per output
    // map; the GELU' / residual epilogues also load the chunks 1.. of their second operand inside the epilogue, 4 MI - 8 pieces)
    constexpr bool kReadsAux = EPI == GRIT_GEMM_DGELU || EPI == GRIT_GEMM_BIAS_RES;
    constexpr int EPI_OPS = EPI == GRIT_GEMM_BIAS_GELU ? 8 * MI : (kReadsAux ? 8 * MI - 8 : 4 * MI);
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    // tiles of this workgroup: the 8 XCD groups (blockIdx % 8) own contiguous bands of the row-major tile list; inside a band the
    // workgroups take tiles round-robin, so the CUs of an XCD work on neighbouring tiles (shared A panels, B in L2)
    const int ntiles = g.tiles_m * g.tiles_n;
    const int ngroups = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % ngroups, idx = blockIdx.x / ngroups, per_xcd = ((int)gridDim.x - xcd + ngroups - 1) / ngroups;
    const int band_lo = (int)((long long)ntiles * xcd / ngroups), band_hi = (int)((long long)ntiles * (xcd + 1) / ngroups);
    const int my_tiles = band_lo + idx < band_hi ? (band_hi - band_lo - idx + per_xcd - 1) / per_xcd : 0;
    if (my_tiles == 0) return;
    const int KT = g.K / BK;
    auto tile_of = [&](int i) { return band_lo + idx + i * per_xcd; };
    // Equal tiles keep all 256 workgroups in lockstep: every CU stores its tile in the same few microseconds (64 MB per round with
    // the two-output epilogue) and waits for HBM, which then idles during the next main loops.  The tile counts are uneven anyway
    // (1 600 tiles on 256 CUs: a quarter of the workgroups take 7, the others 6 and would idle at the end): the workgroups with
    // FEWER tiles start late instead, a quarter of a tile time apart, so the store phases of the chip interleave.
    if (g.stagger) {
        const int busiest = (band_hi - band_lo + per_xcd - 1) / per_xcd;
        if (my_tiles < busiest) {
            const int quarter = (KT * 2900 + 9000) * MI / 32;  // shader cycles (measured per-tile time of this kernel, K step + epilogue)
            const int naps = ((idx & 3) * quarter) >> 12; // s_sleep 64 = 4 096 cycles
            for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }

    // ---- transfers: piece P = wave + 4 p (p = 0..7) of an operand = rows 8 P .. 8 P + 7, one 1 KB instruction
    const int prow = lane >> 3;                                   // row within the piece
    const int r16 = (8 * (wave & 1) + prow) & 15;                 // its index within the 16-row block (the same for every p)
    const int pchunk = (lane & 7) ^ chunk_swizzle<BK>(r16);       // source chunk that lands in LDS chunk (lane & 7)
    const unsigned voffA = (unsigned)(((long)(8 * wave + prow) * g.lda + pchunk * 8) * 2);
    const unsigned voffB = (unsigned)(((long)(8 * wave + prow) * g.ldb + pchunk * 8) * 2);
    const int strideA = (int)(32 * g.lda * 2), strideB = (int)(32 * g.ldb * 2);  // bytes between the pieces of a wave
    // the transfer side of the stream runs two steps ahead of the MFMAs: tile / k step / descriptors of the next step to fetch.
    // Past the end of the stream it keeps re-fetching the last step (nobody reads it): the loop body has no "is there more" branch
    int lti = 0, lks = 0;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    auto set_load_tile = [&](int i) {
        const int t = tile_of(i), tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
        // (the last row tile is shifted back to end at row M: every row of every tile exists)
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)min(tm * BM, g.M - BM) * g.lda), 0, 0x7fffffff, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + (size_t)(tn * BN) * g.ldb), 0, 0x7fffffff, 0x00020000);
    };
    set_load_tile(0);
    auto dmaA = [&](int buf, int p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(lds + buf * BUF + (wave + 4 * p) * 1024), 16, voffA,
                                                 p * strideA + lks * (BK * 2), 0, 0);
    };
    auto dmaB = [&](int buf, int p) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(lds + buf * BUF + A_BYTES + (wave + 4 * p) * 1024), 16, voffB,
                                                 p * strideB + lks * (BK * 2), 0, 0);
    };
    auto advance_load = [&]() {
        if (++lks == KT) {
            if (lti + 1 < my_tiles) { lks = 0; set_load_tile(++lti); }
            else lks = KT - 1;
        }
    };

    // ---- fragment addresses: token block i / weight block j at +2048 i / j; k half h in the chunk index
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned c = (unsigned)(((4 * h + lq) ^ chunk_swizzle<BK>(l15)) * 16);
        aoff[h] = lds0 + (wm * 16 * MI + l15) * ROWB + c;
        boff[h] = lds0 + A_BYTES + (wn * 128 + l15) * ROWB + c;
    }

    v4f acc[MI][8];
    v8bf x0[MI], w0[8], x1[MI], w1[8];
#define GRIT_TIE8(f) asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]))
#define GRIT_TIE7(f) asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]))
#define GRIT_TIEX(f) do { if constexpr (MI == 8) GRIT_TIE8(f); else GRIT_TIE7(f); } while (0)

    // ---- prologue: steps 0 and 1 in flight, step 0 landed, its first half in registers
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int p = 0; p < MI; ++p) dmaA(q, p);
#pragma unroll
        for (int p = 0; p < 8; ++p) dmaB(q, p);
        advance_load();
    }
    wait_vm<NDMA>();
    __builtin_amdgcn_s_barrier();
    for_each_index([&](auto kc) { constexpr int k = decltype(kc)::value; x0[k] = lds_read16_off<k * 2048>(aoff[0]); },
                   std::make_integer_sequence<int, MI>{});
    for_each_index([&](auto kc) { constexpr int k = decltype(kc)::value; w0[k] = lds_read16_off<k * 2048>(boff[0]); },
                   std::make_integer_sequence<int, 8>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GRIT_TIEX(x0); GRIT_TIE8(w0);

    // One 64-deep K step = 128 MFMAs (index m: k half m >> 6, token block (m >> 3) & 7, weight block m & 7), each followed by AT MOST
    // one other instruction, so that its issue hides in the 16 cycles the MFMA occupies the matrix pipe (bursts of LDS reads or
    // transfers between groups of MFMAs cost the pipe ~1 300 cycles per step: 111 -> 96 us at M 51 200, N 512, K 2 048):
    //   m  0..15   the 16 fragments of k half 1 of this buffer (8 token blocks, then 8 weight blocks)
    //   m 20       token fragments in registers (lgkmcnt(8)) + barrier: the A half of the buffer is free
    //   m 24       weight fragments in registers + barrier: the B half is free
    //   m 26..86   every 4th: the 16 transfers of step s + 2 into this buffer (A pieces, then B pieces)
    //   m 92       step s + 1 has landed (only the 16 transfers of step s + 2, and behind a tile's first step the epilogue's stores,
    //              are still counted) + barrier
    //   m 93..123  every 2nd: the 16 fragments of k half 0 of the other buffer; waited for behind the last MFMA
    // first: the step opens a tile -- its k-half-0 MFMAs start from C = 0 (no accumulator to clear).
    auto kstep = [&](auto firstc, int s) {
        constexpr bool first = decltype(firstc)::value;
        const int b = s & 1;
        const unsigned a1 = aoff[1] + (unsigned)(b * BUF), b1 = boff[1] + (unsigned)(b * BUF);
        const unsigned a0n = aoff[0] + (unsigned)((b ^ 1) * BUF), b0n = boff[0] + (unsigned)((b ^ 1) * BUF);
        auto slot = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr int h = m / (8 * MI), i = (m >> 3) % MI, j = m & 7;
            // slots of the other instructions (MI = 8 as listed above; MI = 7: one token fragment and one A transfer fewer, the tail
            // of the step closed up: the wait for step s + 1 at m 84, the 15 fragment reads of the other buffer in m 85 .. 111)
            constexpr int M_WAIT = MI == 8 ? 92 : 84;
            constexpr int nread = m - (M_WAIT + 1);  // position behind the wait
            // MI = 8: every 2nd slot from 93; MI = 7: every 2nd slot 85 .. 107 (12 reads), then 109, 110, 111
            constexpr int r_next = MI == 8 ? ((nread >= 0 && nread <= 30 && nread % 2 == 0) ? nread / 2 : -1)
                                           : ((nread >= 0 && nread <= 22 && nread % 2 == 0) ? nread / 2
                                              : (nread == 24 ? 12 : (nread == 25 ? 13 : (nread == 26 ? 14 : -1))));
            if constexpr (h == 0 && first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[i][j]) : "v"(w0[j]), "v"(x0[i]));
            else if constexpr (h == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(w0[j]), "v"(x0[i]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(w1[j]), "v"(x1[i]));
            if constexpr (m < MI) x1[m] = lds_read16_off<m * 2048>(a1);
            else if constexpr (m < MI + 8) w1[m - MI] = lds_read16_off<(m - MI) * 2048>(b1);
            else if constexpr (m == 20) {
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                GRIT_TIEX(x1);
                __builtin_amdgcn_s_barrier();
            } else if constexpr (m == 24) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                GRIT_TIE8(w1);
                __builtin_amdgcn_s_barrier();
            } else if constexpr (m >= 26 && m < 26 + 4 * NDMA && (m - 26) % 4 == 0) {
                constexpr int d = (m - 26) / 4;
                if constexpr (d < MI) dmaA(b, d);
                else dmaB(b, d - MI);
                if constexpr (d == NDMA - 1) advance_load();
            } else if constexpr (m == M_WAIT) {
                wait_vm<first ? (NDMA + EPI_OPS > 63 ? 63 : NDMA + EPI_OPS) : NDMA>();
                __builtin_amdgcn_s_barrier();
            } else if constexpr (r_next >= 0) {
                constexpr int r = r_next;
                if constexpr (r < MI) x0[r] = lds_read16_off<r * 2048>(a0n);
                else w0[r - MI] = lds_read16_off<(r - MI) * 2048>(b0n);
            }
        };
        for_each_index(slot, std::make_integer_sequence<int, NMFMA>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        GRIT_TIEX(x0); GRIT_TIE8(w0);
    };

    // ---- epilogue of one tile: four chunks of 32 rows through the wave's image (transposed: whole 256-byte row segments out)
    const unsigned img = lds0 + 2 * BUF + wave * IMG;
    // write side: quad (il, j) of a chunk = 8 bytes at row 16 il + l15, 16-byte chunk (2 j + (lq >> 1)) ^ (row & 15)
    unsigned put_base[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) put_base[j] = img + l15 * 256 + (((2 * j + (lq >> 1)) ^ l15) * 16) + (lq & 1) * 8;
    // read side: piece pc = rows 4 pc .. 4 pc + 3; this lane's 16 bytes: row 4 pc + (lane >> 4), chunk lane & 15.
    // row & 15 = (4 pc + (lane >> 4)) & 15 depends on pc: the XOR is applied per piece (pc & 3 selects one of four offsets)
    unsigned get_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * q + (lane >> 4);
        get_off[q] = img + r * 256 + (((lane & 15) ^ r) * 16);
    }
    const bool nt_aux = (g.nt & 1) != 0, nt_c = (g.nt & (EPI == GRIT_GEMM_DGELU ? 4 : 2)) != 0;
    auto store16 = [&](__bf16* dst, const u32x4& v, bool nt) {
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
        else *reinterpret_cast<u32x4*>(dst) = v;
    };
    // What the epilogue reads from memory is fetched BEFORE the tile's last K step, i.e. ahead of that step's transfers in the
    // in-order memory queue: waiting for it later never waits for the prefetch of the next tile.  (The pre-activation rows of the
    // GELU' epilogue: chunk 0 here, chunk c + 1 ahead of chunk c's stores.)
    uint2 braw[8];
    u32x4 hv[2][8];
    auto h_rows = [&](int ti, auto cc, u32x4 (&dst)[8]) {
        constexpr int c = decltype(cc)::value;
        const int t = tile_of(ti), tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
        const int mw = min(tm * BM, g.M - BM) + wm * 16 * MI, nw = tn * BN + wn * 128;
        constexpr int NP = 4 * (MI - 2 * c >= 2 ? 2 : 1);  // pieces of 4 rows in chunk c (MI = 7: the last chunk is one 16-row block)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
            const __bf16* hp = g.aux + (size_t)(mw + 32 * c + 4 * pc + (lane >> 4)) * g.ldaux + nw + (lane & 15) * 8;
            dst[pc] = (g.nt & 8) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(hp)) : *reinterpret_cast<const u32x4*>(hp);
        }
    };
    auto pre_epilogue = [&](int ti) {
        if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU || EPI == GRIT_GEMM_BIAS_RES) {
            const int t = tile_of(ti), tn = t % g.tiles_n;
            const int nw = tn * BN + wn * 128;
#pragma unroll
            for (int j = 0; j < 8; ++j) braw[j] = *reinterpret_cast<const uint2*>(g.bias + nw + 16 * j + 4 * lq);
        }
        if constexpr (kReadsAux) h_rows(ti, std::integral_constant<int, 0>{}, hv[0]);
    };
    auto epilogue = [&](int ti) {
        const int t = tile_of(ti), tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
        const int m_tile = tm * BM, m0 = min(m_tile, g.M - BM);
        const int mw = m0 + wm * 16 * MI, nw = tn * BN + wn * 128;
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last asm MFMAs retire before their accumulators are read
        // BIAS_RES: the factor of a row's sample.  A tile is at most rows_per_sample rows high (host-checked): two samples at most
        float s_lo = 1.f, s_hi = 1.f;
        int m_split = 0x7fffffff;
        if constexpr (EPI == GRIT_GEMM_BIAS_RES) {
            if (g.row_scale != nullptr) {
                const int b0 = m0 / g.rows_per_sample, nb = (g.M + g.rows_per_sample - 1) / g.rows_per_sample;
                m_split = (b0 + 1) * g.rows_per_sample;
                s_lo = g.row_scale[b0];
                s_hi = g.row_scale[min(b0 + 1, nb - 1)];
            }
        }
        v4f bias4[8];
        if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU || EPI == GRIT_GEMM_BIAS_RES) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                bias4[j] = v4f{__builtin_bit_cast(float, braw[j].x << 16), __builtin_bit_cast(float, braw[j].x & 0xffff0000u),
                               __builtin_bit_cast(float, braw[j].y << 16), __builtin_bit_cast(float, braw[j].y & 0xffff0000u)};
        }
        float cs[8];
        if constexpr (EPI == GRIT_GEMM_DGELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[e] = 0.f;
        }
        const size_t col = (size_t)nw + (lane & 15) * 8;
        for_each_index([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int NB = MI - 2 * c >= 2 ? 2 : 1, NP = 4 * NB;  // 16-row blocks / 4-row pieces of this chunk
            for_each_index([&](auto qc) {
                constexpr int q = decltype(qc)::value, il = q >> 3, j = q & 7;
                // (pins the quad to its AGPRs up to here: without it the allocator copies all 256 accumulators into VGPRs at the
                // top of the epilogue and spills 140 of them)
                asm volatile("" : "+a"(acc[2 * c + il][j]));
                v4f v = acc[2 * c + il][j];
                if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU || EPI == GRIT_GEMM_BIAS_RES) v += bias4[j];
                v4bf p;
                p[0] = (__bf16)v[0]; p[1] = (__bf16)v[1]; p[2] = (__bf16)v[2]; p[3] = (__bf16)v[3];
                const uint2 pk = __builtin_bit_cast(uint2, p);
                const unsigned pb = put_base[j];  // (a variable named only in an asm operand of a nested generic lambda is not captured)
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(pb), "v"(pk), "n"(il * 4096) : "memory");
            }, std::make_integer_sequence<int, 8 * NB>{});
            if constexpr (kReadsAux && c < 3) h_rows(ti, std::integral_constant<int, c + 1>{}, hv[(c + 1) & 1]);
            // the chunk's eight pieces come back from the image together (one wait), then leave one by one
            u32x4 pv[8];
            for_each_index([&](auto pcc) {
                constexpr int pc = decltype(pcc)::value;
                const unsigned go = get_off[pc & 3];
                u32x4 tv;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(tv) : "v"(go), "n"((pc >> 2) * 4096) : "memory");
                pv[pc] = tv;
            }, std::make_integer_sequence<int, NP>{});
            if constexpr (NP == 8)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]),
                             "+v"(pv[7]) :: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]) :: "memory");
            for_each_index([&](auto pcc) {
                constexpr int pc = decltype(pcc)::value;
                const u32x4 v = pv[pc];
                const int m = mw + 32 * c + 4 * pc + (lane >> 4);
                if constexpr (EPI == GRIT_GEMM_BIAS_GELU) {
                    // image = the pre-activation as stored for the backward pass (bf16); the activation is GELU of THAT value,
                    // what an unfused Linear -> GELU pair computes
                    if (g.aux) store16(g.aux + (size_t)m * g.ldaux + col, v, nt_aux);
                    unsigned o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const v2f x = {__builtin_bit_cast(float, v[e] << 16), __builtin_bit_cast(float, v[e] & 0xffff0000u)};
                        const v2f y = gelu2(x);
                        typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
                        v2bf pk;
                        pk[0] = (__bf16)y[0]; pk[1] = (__bf16)y[1];
                        o[e] = __builtin_bit_cast(unsigned, pk);
                    }
                    store16(g.C + (size_t)m * g.ldc + col, u32x4{o[0], o[1], o[2], o[3]}, nt_c);
                } else if constexpr (EPI == GRIT_GEMM_DGELU) {
                    // image = the gradient w.r.t. the activation (bf16, as an unfused GEMM would store it); times GELU'(pre)
                    const u32x4 hx = hv[c & 1][pc];
                    const bool own = m >= m_tile;  // rows a shifted last tile shares with its neighbour are the neighbour's in the sums
                    unsigned o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const v2f d = {__builtin_bit_cast(float, v[e] << 16), __builtin_bit_cast(float, v[e] & 0xffff0000u)};
                        const v2f x = {__builtin_bit_cast(float, hx[e] << 16), __builtin_bit_cast(float, hx[e] & 0xffff0000u)};
                        const v2f y = d * dgelu2(x);
                        typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
                        v2bf pk;
                        pk[0] = (__bf16)y[0]; pk[1] = (__bf16)y[1];
                        o[e] = __builtin_bit_cast(unsigned, pk);
                        cs[2 * e] += own ? y[0] : 0.f;
                        cs[2 * e + 1] += own ? y[1] : 0.f;
                    }
                    store16(g.C + (size_t)m * g.ldc + col, u32x4{o[0], o[1], o[2], o[3]}, nt_c);
                } else if constexpr (EPI == GRIT_GEMM_BIAS_RES) {
                    // image = the branch as an unfused Linear would store it (bf16); x = shortcut + factor * branch in fp32, rounded once --
                    // bit for bit what grit_add_layernorm_fwd computes from the stored branch
                    const u32x4 rx = hv[c & 1][pc];
                    const float sc = m < m_split ? s_lo : s_hi;
                    unsigned o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float b0 = __builtin_bit_cast(float, v[e] << 16), b1 = __builtin_bit_cast(float, v[e] & 0xffff0000u);
                        const float r0 = __builtin_bit_cast(float, rx[e] << 16), r1 = __builtin_bit_cast(float, rx[e] & 0xffff0000u);
                        typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
                        v2bf pk;
                        pk[0] = (__bf16)__fadd_rn(r0, __fmul_rn(b0, sc)); pk[1] = (__bf16)__fadd_rn(r1, __fmul_rn(b1, sc));  // (not fused: the add_layernorm kernel's order)
                        o[e] = __builtin_bit_cast(unsigned, pk);
                    }
                    store16(g.C + (size_t)m * g.ldc + col, u32x4{o[0], o[1], o[2], o[3]}, false);
                } else {
                    store16(g.C + (size_t)m * g.ldc + col, v, nt_c);
                }
            }, std::make_integer_sequence<int, NP>{});
        }, std::make_integer_sequence<int, 4>{});
        if constexpr (EPI == GRIT_GEMM_BIAS_GELU) {
            // (no pre-activation kept: half the stores EPI_OPS counts -- drain, so that the next step's counted wait is exact)
            if (!g.aux) wait_vm<0>();
        }
        if constexpr (EPI == GRIT_GEMM_DGELU) {
            // lanes l, l + 16, l + 32, l + 48 hold the same 8 channels (different rows): fold them, lanes 0..15 write 8 floats each
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                cs[e] += __shfl_xor(cs[e], 16);
                cs[e] += __shfl_xor(cs[e], 32);
            }
            if (lane < 16) {
                float* dst = g.colsum + (size_t)(2 * tm + wm) * g.N + nw + lane * 8;
                *reinterpret_cast<v4f*>(dst) = v4f{cs[0], cs[1], cs[2], cs[3]};
                *reinterpret_cast<v4f*>(dst + 4) = v4f{cs[4], cs[5], cs[6], cs[7]};
            }
        }
    };

    int s = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        if (KT == 1) {
            pre_epilogue(ti);
            kstep(std::true_type{}, s++);
        } else {
            kstep(std::true_type{}, s++);
            for (int t = 1; t < KT - 1; ++t) kstep(std::false_type{}, s++);
            pre_epilogue(ti);
            kstep(std::false_type{}, s++);
        }
        epilogue(ti);
    }
    wait_vm<0>();  // the transfers issued past the end of the stream must not land in another workgroup's LDS
}

}  // namespace

namespace grit_detail {

static int device_cus() {
    static grit_detail::PerDevice<int> cus_pd; int& cus = cus_pd();
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    return cus;
}

// Tile height (256 or 224 rows) that needs fewer block rows per CU: rounds of one tile per CU x 16-row blocks per wave (see the kernel).
int gemm_w4_tile_rows(int M, int N) {
    const int cus = device_cus();
    if (cus <= 0 || M < 256 || N % 256) return 256;
    const long t8 = (long)((M + 255) / 256) * (N / 256), t7 = (long)((M + 223) / 224) * (N / 256);
    const long c8 = ((t8 + cus - 1) / cus) * 8, c7 = ((t7 + cus - 1) / cus) * 7;
    return c7 < c8 ? 224 : 256;
}

int gemm_w4_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, float* colsum, int nt, void* stream, int tile_rows,
                   const float* row_scale, int rows_per_sample) {
    if (N % 256 || K % 64 || M < 256) return GRIT_ERR_UNSUPPORTED;
    if (epilogue < GRIT_GEMM_NONE || epilogue > GRIT_GEMM_BIAS_RES) return GRIT_ERR_UNSUPPORTED;
    if ((long)M * lda * 2 >= 0x7fffffffL || (long)N * ldb * 2 >= 0x7fffffffL) return GRIT_ERR_UNSUPPORTED;  // 32-bit buffer offsets
    if (tile_rows == 0) tile_rows = gemm_w4_tile_rows(M, N);
    if (tile_rows != 224 && tile_rows != 256) return GRIT_ERR_BAD_ARG;
    if (epilogue == GRIT_GEMM_BIAS_RES && (!aux || (row_scale && rows_per_sample < tile_rows))) return GRIT_ERR_UNSUPPORTED;
    constexpr int LDS = 2 * (256 + 256) * 128 + 4 * 8192;  // 160 KB: the whole CU (224-row tiles: 8 KB less, same occupancy)
    W4Args g;
    g.A = (const __bf16*)A; g.lda = lda; g.B = (const __bf16*)B; g.ldb = ldb; g.C = (__bf16*)C; g.ldc = ldc;
    g.bias = (const __bf16*)bias; g.aux = (__bf16*)aux; g.ldaux = ldaux; g.colsum = colsum; g.nt = nt;
    g.row_scale = row_scale; g.rows_per_sample = rows_per_sample;
    static const int stagger = [] { const char* e = getenv("GRIT_GEMM_W4_STAGGER"); return e ? atoi(e) : 1; }();
    g.stagger = stagger;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + tile_rows - 1) / tile_rows;
    g.tiles_n = N / 256;
    const int cus = device_cus();
    if (cus <= 0) return GRIT_ERR_LAUNCH;
    const int ntiles = g.tiles_m * g.tiles_n;
    const dim3 grid(ntiles < cus ? ntiles : cus), block(256);
#define GRIT_W4_LAUNCH_MI(E, MI_)                                                                                    \
    {                                                                                                                \
        auto kern = gemm_w4_bf16<E, MI_>;                                                                            \
        static grit_detail::PerDevice<bool> attr_done_pd; bool& attr_done = attr_done_pd();                          \
        if (!attr_done) {                                                                                            \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) \
                return GRIT_ERR_LAUNCH;                                                                              \
            attr_done = true;                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, grid, block, LDS, (hipStream_t)stream, g);                                          \
    }
#define GRIT_W4_LAUNCH(E) { if (tile_rows == 224) GRIT_W4_LAUNCH_MI(E, 7) else GRIT_W4_LAUNCH_MI(E, 8) }
    switch (epilogue) {
        case GRIT_GEMM_NONE: GRIT_W4_LAUNCH(GRIT_GEMM_NONE) break;
        case GRIT_GEMM_BIAS: GRIT_W4_LAUNCH(GRIT_GEMM_BIAS) break;
        case GRIT_GEMM_BIAS_GELU: GRIT_W4_LAUNCH(GRIT_GEMM_BIAS_GELU) break;
        case GRIT_GEMM_DGELU: GRIT_W4_LAUNCH(GRIT_GEMM_DGELU) break;
        default: GRIT_W4_LAUNCH(GRIT_GEMM_BIAS_RES) break;
    }
#undef GRIT_W4_LAUNCH
#undef GRIT_W4_LAUNCH_MI
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // namespace grit_detail
