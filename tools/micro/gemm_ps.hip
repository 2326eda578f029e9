// NEGATIVE RESULT of round 4 (bit-exact, 154 vs 134 us against the per-tile kernel; profiles/r04/README.md): kept here, outside the
// product library, for the record and for tools/micro/gemm_ps_bench.hip.
// Persistent bf16 MFMA GEMM for the long token maps with a TRICKLED epilogue (gfx950) -- variant 6 of grit_gemm_bf16_nt.
//
//   C[M, N] = epilogue( A[M, K] . B[N, K]^T )      A, B, C bf16 row-major (K-contiguous operands), fp32 accumulation
//
// What the per-tile kernels of gemm.hip leave on the table at the Swin shapes (K = 256 / 512: 8-16 K steps per 256 x 256 tile) is
// everything around the main loop: ~3k cycles of ring fill, ~5-8k of epilogue, and -- the larger part -- the store burst: every CU
// finishes its tile at the same moment, 32 MB of output leave the chip's 256 CUs at once, and because stores and LDS-DMA loads
// retire through ONE in-order counter (vmcnt) the next tile's first wait also waits for the write acknowledgements.  The persistent
// ping-pong kernel (variant 5) already runs the K steps of consecutive tiles as one stream; its epilogue still sits between two
// tiles, once per wave group (7.8k of 32k cycles per tile at K = 512, tools/micro/gemm_stamps.hip).
//
// Here the finished tile leaves through the NEXT tile's main loop:
//   * at the end of a tile a wave adds the bias, rounds its 128 x 64 block to bf16 (accumulator quads -> 8-byte pieces), parks the
//     upper 64 rows in its private 8 KB LDS image (transposed: 128-byte row segments) and keeps the lower 64 rows packed in 32
//     registers; the accumulators are free for the next tile at once;
//   * every K step of the next tile drains 16 / KT of the block: one ds_read_b128 + one (GELU: two) 16-byte row-segment store per
//     piece, issued behind the step's 32 MFMAs; half way the held registers move into the image;
//   * so the output leaves at the rate the tiles are computed (210 MB over the whole launch instead of 32 MB bursts), a wave's
//     oldest outstanding store is always a full K step old when the next counted wait covers it, and the GELU arithmetic of the
//     fused fc1 epilogue runs under the other wave group's MFMAs.
// LDS: three-slot ring of K steps (32 deep: 3 x 32 KB, LDS-DMA two steps ahead) + 8 x 8 KB images = 160 KB, one workgroup per CU,
// 8 waves as two groups half a K step apart (while one group reads fragments the other owns the matrix pipe), exactly as variant 5.
//
// Barrier phases (b = barrier index; L(q) = DMA issue of step q + 2 and fragment reads of step q, M(q) = its 32 MFMAs + the drain):
//   waves 0-3:  L(q) in [2q, 2q+1)    M(q) in [2q+1, 2q+2)        waves 4-7:  L(q) in [2q+1, 2q+2)   M(q) in [2q+2, 2q+3)
//   WAR: the DMAs of step q+2 overwrite slot (q-1) % 3, last read in L(q-1) of both groups: finished (lgkmcnt(0)) before barrier 2q;
//   RAW: step s is first read after barrier 2s; every wave has waited for ITS pieces of step s before it arrives there (waves 0-3 at
//        the end of M(s-1), waves 4-7 at the end of L(s-1)) with vmcnt(4 + stores of the previous drain): the four DMAs of step
//        s + 1 and the stores issued after the DMAs of step s may stay in flight, nothing older.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grit_hip.h"
#include "../../grit_amd/csrc/gemm_math.h"
namespace grit_detail {
int gemm_ps_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, int nt, void* stream, unsigned long long* stamps);
}

namespace {

struct PsArgs {
    const __bf16* A; long lda;
    const __bf16* B; long ldb;
    __bf16* C; long ldc;
    const __bf16* bias;
    __bf16* aux; long ldaux;
    int nt;  // non-temporal stores: 1 aux, 2 C (same bits as GRIT_GEMM_NT_AUX)
    int dbg; // diagnostic builds only (results wrong on purpose): 1 = nothing leaves (no drain, no pack), 2 = no stores (LDS side kept)
    int M, N, K, tiles_m, tiles_n;
#ifdef GRIT_GEMM_STAMPS
    unsigned long long* stamps;  // diagnostic build (tools/micro/gemm_ps_bench.hip): [workgroup][wave][16]
#endif
};


// The operands of step q + 1 have landed once at most `younger` + 4 younger operations are still counted: the 4 DMAs of step
// q + 2 and whatever this wave issued behind the DMAs of step q + 1 (stores of its drains, the bias loads of a tile's last step).
// `younger` is wave-uniform; counts without an immediate of their own wait for more than necessary (safe).
__device__ __forceinline__ void wait_dma(bool more, int younger) {
    if (!more) { wait_vm<0>(); return; }
    switch (younger) {
        case 1: wait_vm<5>(); break;
        case 2: wait_vm<6>(); break;
        case 3: wait_vm<7>(); break;
        case 4: wait_vm<8>(); break;
        case 5: wait_vm<9>(); break;
        case 6: wait_vm<10>(); break;
        case 7: wait_vm<11>(); break;
        case 8: wait_vm<12>(); break;
        case 9: wait_vm<13>(); break;
        case 10: wait_vm<14>(); break;
        case 12: wait_vm<16>(); break;
        default: wait_vm<4>(); break;
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2)
void gemm_ps_bf16(const PsArgs g) {
    constexpr int BM = 256, BN = 256, BK = 32, NSLOT = 3;
    constexpr int ROWB = BK * 2;                       // 64-byte staged rows, 4 chunks of 16 B
    constexpr int A_BYTES = BM * ROWB, SLOT = (BM + BN) * ROWB;
    constexpr int MT = 8, NTL = 4;                     // wave tile 128 x 64
    constexpr int IMG = 8192;                          // per wave: 64 rows x 128 B of finished bf16 output
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the group tests below are scalar branches
    const int grp = wave >> 2, wu = wave & 3;          // group = row half, wu = 64-column strip
    const int l15 = lane & 15, lq = lane >> 4;
    // The image is only ever touched through inline asm: hipcc cannot tell an LDS access of its own from the LDS-DMA transfers in
    // flight (one dynamic LDS block) and puts `s_waitcnt vmcnt(0)` in front of every LDS read or write it sees behind one -- here
    // that would make each drain wait for the two K steps of prefetch and for its own stores.  The waits of the asm accesses are
    // placed by hand (lgkmcnt tied to the registers they fill).
    const unsigned img = (unsigned)(uintptr_t)(lptr_t)lds + NSLOT * SLOT + wave * IMG;

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
    const int rate = (16 + KT - 1) / KT;               // image pieces (8 rows x 128 B) drained per K step: all 16 within one tile
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
            asrc[j] = g.A + (size_t)(min(tm * BM, g.M - BM) + r) * g.lda + pchunk * 8;
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
        lslot = lslot == NSLOT - 1 ? 0 : lslot + 1;
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

    // ---- the finished tile on its way out
    uint2 held[16];          // packed quads of row blocks 4..7 (rows 64..127 of the wave tile), [4 (i - 4) + j]
#pragma unroll
    for (int t = 0; t < 16; ++t) held[t] = uint2{0u, 0u};
    int dr = 16;             // next image piece to drain: 0..7 rows 0..63 of the wave tile, 8..15 rows 64..127; 16 = nothing pending
    bool hpend = false;      // the held half has not moved into the image yet
    int pmw = 0, pnw = 0;    // row / column origin of the wave's block in the tile that is draining
    const bool nt_aux = (g.nt & 1) != 0, nt_c = (g.nt & 2) != 0;

    auto pack4 = [](const v4f& v) {
        v4bf p;
        p[0] = (__bf16)v[0]; p[1] = (__bf16)v[1]; p[2] = (__bf16)v[2]; p[3] = (__bf16)v[3];
        return __builtin_bit_cast(uint2, p);
    };
    // one packed 8-byte piece into the [64][64] bf16 image (16-byte chunks XOR-ed with row & 7); il = row block within the image
    // (row & 7 == l15 & 7 for every row block: the byte offset of (il, j) is put_base[j] + 2048 il)
    unsigned put_base[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) put_base[j] = img + l15 * 128 + (((2 * j + (lq >> 1)) ^ (l15 & 7)) * 16) + (lq & 1) * 8;
    auto put = [&](int il, int j, uint2 p) {
        switch (il) {  // immediate offsets: one address register per column block
            case 0: asm volatile("ds_write_b64 %0, %1" :: "v"(put_base[j]), "v"(p) : "memory"); break;
            case 1: asm volatile("ds_write_b64 %0, %1 offset:2048" :: "v"(put_base[j]), "v"(p) : "memory"); break;
            case 2: asm volatile("ds_write_b64 %0, %1 offset:4096" :: "v"(put_base[j]), "v"(p) : "memory"); break;
            default: asm volatile("ds_write_b64 %0, %1 offset:6144" :: "v"(put_base[j]), "v"(p) : "memory"); break;
        }
    };
    // piece pp (rows 8 pp .. 8 pp + 7 of the image): this lane's 16 bytes are at get_base + 1024 pp
    const unsigned get_base = img + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    auto store16 = [&](__bf16* dst, const u32x4& v, bool nt) {
#ifdef GRIT_GEMM_STAMPS
        if (g.dbg & 2) return;
#endif
        if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
        else *reinterpret_cast<u32x4*>(dst) = v;
    };
    // One piece on its way out.  read: the lane's 16 bytes of image piece `piece` (no wait); emit: epilogue arithmetic + the 16-byte
    // row-segment store(s), returns the number of store instructions.  Between the two sits most of the step's MFMA block, so the
    // LDS latency, the GELU arithmetic and the store issue ride in the issue slots the MFMAs leave free.
    auto piece_read = [&](int piece, u32x4& v) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(get_base + 1024 * (piece & 7)) : "memory");
    };
    auto piece_emit = [&](int piece, const u32x4& v) {
        int issued = 0;
        const int m = pmw + (piece >> 3) * 64 + 8 * (piece & 7) + (lane >> 3);
        const size_t col = (size_t)pnw + (lane & 7) * 8;
        if constexpr (EPI == GRIT_GEMM_BIAS_GELU) {
            // image = the pre-activation as it is stored for the backward pass (bf16); the activation is GELU of THAT value,
            // what an unfused Linear -> GELU pair computes
            if (g.aux) {
                store16(g.aux + (size_t)m * g.ldaux + col, v, nt_aux);
                ++issued;
            }
            const unsigned w[4] = {v[0], v[1], v[2], v[3]};
            unsigned o[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const v2f x = {__builtin_bit_cast(float, w[t] << 16), __builtin_bit_cast(float, w[t] & 0xffff0000u)};
                const v2f y = gelu2(x);
                typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
                v2bf pk;
                pk[0] = (__bf16)y[0]; pk[1] = (__bf16)y[1];
                o[t] = __builtin_bit_cast(unsigned, pk);
            }
            store16(g.C + (size_t)m * g.ldc + col, u32x4{o[0], o[1], o[2], o[3]}, nt_c);
            ++issued;
        } else {
            store16(g.C + (size_t)m * g.ldc + col, v, nt_c);
            ++issued;
        }
        return issued;
    };
    auto move_held = [&]() {  // the held half into the image, once pieces 0..7 have been read
        if (dr == 8 && hpend) {
#pragma unroll
            for (int t = 0; t < 16; ++t) put(t >> 2, t & 3, held[t]);
            hpend = false;
        }
    };
    // serial form (more than two pieces per step, and the last tile's flush): returns the number of store instructions issued
    auto drain = [&](int count) {
        int issued = 0;
        for (int c = 0; c < count && dr < 16; ++c) {
            move_held();
            u32x4 v;
            piece_read(dr, v);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory");
            issued += piece_emit(dr, v);
            ++dr;
        }
        move_held();
        return issued;
    };
    // end of tile `ti`: accumulators (+ bias) -> bf16; rows 0..63 into the image, rows 64..127 held; accumulators cleared
    // (the last row tile is shifted back to end at row M: every row of every tile exists, no store is ever masked -- the count of
    // store instructions a drain reports is exact -- and the rows it shares with its neighbour are written twice with equal values)
    uint2 braw[NTL];         // the tile's bias values of this lane, fetched at the top of its last M phase (under the MFMAs)
#pragma unroll
    for (int j = 0; j < NTL; ++j) braw[j] = uint2{0u, 0u};
    auto fetch_bias = [&](int ti) {  // asm: a load hipcc knows about is waited for with vmcnt(0) (it cannot count the asm waits)
        if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU) {
            const int t = tile_of(ti), tn = t % g.tiles_n;
            const __bf16* src = g.bias + tn * BN + wu * 64 + 4 * lq;
            asm volatile("global_load_dwordx2 %0, %4, off\n\tglobal_load_dwordx2 %1, %4, off offset:32\n\t"
                         "global_load_dwordx2 %2, %4, off offset:64\n\tglobal_load_dwordx2 %3, %4, off offset:96"
                         : "=&v"(braw[0]), "=&v"(braw[1]), "=&v"(braw[2]), "=&v"(braw[3]) : "v"(src) : "memory");
        }
    };
    auto finish_tile = [&](int ti, int stores_behind) {
        const int t = tile_of(ti), tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
        const int mw = min(tm * BM, g.M - BM) + grp * 128, nw = tn * BN + wu * 64;
        v4f bias4[NTL];
        if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU) {
            // the four bias loads are older than this phase's stores: everything but those `stores_behind` stores has retired
            switch (stores_behind) {
                case -1: break;  // fetched two steps ago: retired by the counted wait of the step before this one
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            asm volatile("" : "+v"(braw[0]), "+v"(braw[1]), "+v"(braw[2]), "+v"(braw[3]));
#pragma unroll
            for (int j = 0; j < NTL; ++j)
                bias4[j] = v4f{__builtin_bit_cast(float, braw[j].x << 16), __builtin_bit_cast(float, braw[j].x & 0xffff0000u),
                               __builtin_bit_cast(float, braw[j].y << 16), __builtin_bit_cast(float, braw[j].y & 0xffff0000u)};
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                v4f v = acc[i][j];
                if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU) v += bias4[j];
                const uint2 p = pack4(v);
                if (i < 4) put(i, j, p);
                else held[4 * (i - 4) + j] = p;
                acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
            }
        dr = 0;
        hpend = true;
        pmw = mw;
        pnw = nw;
    };

    // ---- prologue: steps 0, 1 in flight, step 0 landed for everybody
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (q < total) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) dma(pc);
            advance_load();
        }
    if (total > 1) wait_vm<4>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger: waves 4-7 run one phase behind
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");

    int ks = 0, ti = 0, cslot = 0;  // compute side of the stream
    int younger = 0;                // stores this wave issued in its previous M phase (behind the DMAs of the step it then waits for)
#ifdef GRIT_GEMM_STAMPS
    unsigned long long ph_l = 0, ph_b1 = 0, ph_m = 0, ph_b2 = 0, ph_out = 0, ph_t = __builtin_amdgcn_s_memtime();
    const unsigned long long ph_start = ph_t;
#define GRIT_PH(acc_var) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_var += now_ - ph_t; ph_t = now_; }
#else
#define GRIT_PH(acc_var)
#endif
    for (int q = 0; q < total; ++q) {
        // ---- L(q): the DMA pieces of step q + 2, then the 12 fragments of this step
        const bool more = q + 2 < total;
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
        if (grp == 1) wait_dma(more, younger);  // step q + 1 (read by waves 0-3 right after the coming barrier) must have landed
        GRIT_PH(ph_l)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GRIT_PH(ph_b1)
        // ---- M(q): 32 MFMAs, with up to two pieces of the previous tile leaving in between
        // The tile's bias values are fetched two K steps before its end: the counted wait of the step in between already retires
        // them (they are older than the DMAs it waits for), so the pack never waits for memory.  (Tiles of fewer than three steps:
        // fetched in the last step, the pack waits for everything but this phase's stores.)
        const bool bias_step = EPI != GRIT_GEMM_NONE && ks == (KT >= 3 ? KT - 3 : KT - 1);
        if (bias_step) fetch_bias(ti);
        const int room = dr < 8 ? 8 - dr : 16 - dr;  // pieces before the image changes hands
        int np = rate <= 2 ? (rate < room ? rate : room) : 0;
#ifdef GRIT_GEMM_STAMPS
        if (g.dbg & 1) np = 0;
#endif
        u32x4 pv0 = u32x4{0u, 0u, 0u, 0u}, pv1 = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[0], acc[0][j], 0, 0, 0);
        if (np > 0) piece_read(dr, pv0);
        if (np > 1) piece_read(dr + 1, pv1);
#pragma unroll
        for (int i = 1; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        int stores_now = 0;
        if (np > 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pv0), "+v"(pv1) :: "memory");
            stores_now += piece_emit(dr, pv0);
            if (np > 1) stores_now += piece_emit(dr + 1, pv1);
            dr += np;
        }
#pragma unroll
        for (int i = 5; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        if (rate > 2) stores_now += drain(rate);
        move_held();
        // step q + 1 is read by waves 0-3 right after the coming barrier: their DMAs of it are older than the stores of the previous
        // drain, the 4 DMAs of step q + 2, this phase's bias loads and stores -- all of those may stay in flight
        const int ops_now = stores_now + (bias_step ? 4 : 0);  // vector-memory operations of this phase behind the DMAs of step q + 2
        if (grp == 0) wait_dma(more, younger + ops_now);
        cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
        GRIT_PH(ph_m)
        if (++ks == KT) {
#ifdef GRIT_GEMM_STAMPS
            if (!(g.dbg & 1))
#endif
            finish_tile(ti, KT >= 3 ? -1 : stores_now);
            ks = 0;
            ++ti;
        }
        younger = ops_now;
        GRIT_PH(ph_out)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        GRIT_PH(ph_b2)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    drain(16);  // the last tile leaves in one go
#ifdef GRIT_GEMM_STAMPS
    if (g.stamps && (threadIdx.x & 63) == 0) {
        unsigned long long* o = g.stamps + ((size_t)blockIdx.x * 8 + wave) * 16;
        o[0] = ph_start; o[7] = __builtin_amdgcn_s_memtime();
        o[8] = ph_l; o[9] = ph_b1; o[10] = ph_m; o[11] = ph_b2; o[12] = ph_out; o[13] = (unsigned long long)total;
    }
#endif
#undef GRIT_PH
}

}  // namespace

namespace grit_detail {

int gemm_ps_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, int nt, void* stream, unsigned long long* stamps) {
    static const int dbg = [] { const char* e = getenv("GRIT_GEMM_PS_DBG"); return e ? atoi(e) : 0; }();
    if (N % 256 || K % 32 || K < 32 || M < 256) return GRIT_ERR_UNSUPPORTED;
    if (epilogue != GRIT_GEMM_NONE && epilogue != GRIT_GEMM_BIAS && epilogue != GRIT_GEMM_BIAS_GELU) return GRIT_ERR_UNSUPPORTED;
    constexpr int LDS = 3 * (256 + 256) * 64 + 8 * 8192;  // 160 KB: the whole CU
    PsArgs g;
    g.A = (const __bf16*)A; g.lda = lda; g.B = (const __bf16*)B; g.ldb = ldb; g.C = (__bf16*)C; g.ldc = ldc;
    g.bias = (const __bf16*)bias; g.aux = (__bf16*)aux; g.ldaux = ldaux; g.nt = nt;
    g.dbg = dbg;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + 255) / 256;
    g.tiles_n = N / 256;
#ifdef GRIT_GEMM_STAMPS
    g.stamps = stamps;
#else
    (void)stamps;
#endif
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GRIT_ERR_LAUNCH;
        cus = prop.multiProcessorCount;
    }
    const int ntiles = g.tiles_m * g.tiles_n;
    const dim3 grid(ntiles < cus ? ntiles : cus), block(512);
#define GRIT_PS_LAUNCH(E)                                                                                            \
    {                                                                                                                \
        auto kern = gemm_ps_bf16<E>;                                                                                 \
        static bool attr_done = false;                                                                               \
        if (!attr_done) {                                                                                            \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) \
                return GRIT_ERR_LAUNCH;                                                                              \
            attr_done = true;                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, grid, block, LDS, (hipStream_t)stream, g);                                          \
    }
    switch (epilogue) {
        case GRIT_GEMM_NONE: GRIT_PS_LAUNCH(GRIT_GEMM_NONE) break;
        case GRIT_GEMM_BIAS: GRIT_PS_LAUNCH(GRIT_GEMM_BIAS) break;
        default: GRIT_PS_LAUNCH(GRIT_GEMM_BIAS_GELU) break;
    }
#undef GRIT_PS_LAUNCH
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // namespace grit_detail
