// ROUND-6 EXPERIMENT, NOT PART OF THE LIBRARY (moved out of grit_amd/csrc after measuring: profiles/r06/fused_p8_wave_split.txt).  To rebuild it:
// copy this file to grit_amd/csrc/, declare gemm_p8_launch in gemm_launchers.h and add `case 16` to grit_gemm_bf16_nt (git show of the
// commit that introduced this header has both hunks).  Passed every epilogue test as variant 16; 5-60 % slower than the eight-wave kernel:
// the four loader waves carry all 64 LDS-DMA issues of a K step, the four storer waves all 256 row stores of a two-map tile.
//
// Persistent eight-wave bf16 MFMA GEMM whose epilogue stores run under the next tile's K loop (gfx950) -- variant 16 of grit_gemm_bf16_nt.
//
//   C[M, N] = epilogue( A[M, K] . B[N, K]^T )      A, B, C bf16 row-major (K-contiguous operands), fp32 accumulation; epilogues of gemm.hip
//
// Why (round 6, profiles/r06/store_bound.txt, mfma_store_probe.txt): the fused Mlp GEMMs of the Swin blocks (K = 512, two 210 MB hidden maps
// out) take T_MFMA + T_stores -- 75 + 88 us -- in every structure tried: one or two workgroups per CU, paced stores, staggered starts.  The
// chip does not force that: waves that only store (5.6 TB/s) beside waves that only issue MFMAs (1.9 PFLOP/s) finish in the maximum of the
// two times, and with LDS-DMA operand loads in the matrix waves in 0.7 x the sum.  What serialises the real kernels is the wave's in-order
// memory counter: a wave that has issued its tile's row stores cannot learn that a YOUNGER operand transfer has landed before every one of
// those stores has been acknowledged by HBM (s_waitcnt vmcnt counts loads, stores and LDS-DMA transfers together, in issue order), so the
// next tile's K loop waits for the previous tile's stores in every wave.
//
// Here the two jobs belong to different waves of the workgroup:
//   * waves 0-3 (one per SIMD) issue EVERY operand transfer (global_load_lds, 16 pieces of 1 KB per wave and K step) and are the only ones
//     that wait for them (s_waitcnt vmcnt) before the step's barrier; they never store;
//   * waves 4-7 issue EVERY global store of the epilogue -- their own 128 x 64 output block and that of their partner wave - 4, read back
//     from the partner's transposed LDS image -- and never wait on the memory counter inside the K loop: the barrier behind the loaders'
//     wait is what tells them a stage has landed.  Their stores stay in flight while all eight waves run the next tile's K steps.
// Everything else is the eight-wave kernel of gemm.hip: 256 x 256 tiles, wave tiles 128 x 64, K step 64, two 64 KB stages, lane-linear DMA
// images with the chunk permutation on the source address and on the fragment reads, operands fed to v_mfma_f32_16x16x32_bf16 swapped (a
// token on the lane, 4 output channels in the accumulator quad), the epilogue's 8-byte pieces transposed through the (idle) stage ring
// into 128-byte row segments.  One workgroup per CU walks its share of the tiles (XCD bands, as gemm_w4.hip).
#include <hip/hip_runtime.h>
#include "per_device.h"
#include <stdint.h>
#include <stdlib.h>
#include "../../include/grit_hip.h"
#include "gemm_math.h"
#include "gemm_launchers.h"

namespace {

struct P8Args {
    const __bf16* A; long lda;
    const __bf16* B; long ldb;
    __bf16* C; long ldc;
    const __bf16* bias;
    __bf16* aux; long ldaux;
    int nt_aux;        // non-temporal accesses, bits as GRIT_GEMM_NT_AUX
    float* colsum;     // DGELU: [ceil(M / 128), N]
    int M, N, K, tiles_m, tiles_n;
};

template <int EPI>
__global__ __launch_bounds__(512, 2)
void gemm_p8_bf16(const P8Args g) {
    constexpr int BM = 256, BN = 256, BK = 64, WN = 4;
    constexpr int WTM = 128, WTN = 64, MT = 8, NTL = 4;
    constexpr int ROWB = BK * 2, CPR = ROWB / 16;                 // 128-byte staged rows, 8 chunks of 16 B
    constexpr int A_BYTES = BM * ROWB, STAGE = (BM + BN) * ROWB;  // 32 KB + 32 KB
    constexpr int PIECES = 8;                                     // 1 KB pieces of A (and of B) per LOADER wave and stage: 4 waves x 8 x 1 KB
    constexpr int IMG = WTM * WTN * 2;                            // 16 KB per wave: the whole ring as eight transposed output images
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    float* cs_lds = reinterpret_cast<float*>(lds + 2 * STAGE);    // [8 waves][64] column sums on their way to the storer waves

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < 4, storer = !loader;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lq = lane >> 4;

    // tiles of this workgroup: XCD bands of the row-major tile list, round-robin inside a band
    const int ntiles = g.tiles_m * g.tiles_n;
    const int ngroups = gridDim.x < 8 ? (int)gridDim.x : 8;
    const int xcd = blockIdx.x % ngroups, idx = blockIdx.x / ngroups, per_xcd = ((int)gridDim.x - xcd + ngroups - 1) / ngroups;
    const int band_lo = (int)((long long)ntiles * xcd / ngroups), band_hi = (int)((long long)ntiles * (xcd + 1) / ngroups);
    const int my_tiles = band_lo + idx < band_hi ? (band_hi - band_lo - idx + per_xcd - 1) / per_xcd : 0;
    const int nk = g.K / BK;

    // ---- loader side: piece i of an operand stage = rows 32 i + (ltid >> 3), chunk ltid & 7 (ltid = tid: loaders are threads 0 .. 255)
    // (buffer addressing: ONE per-lane byte offset per operand, everything else -- piece, K step -- scalar; rows past M read as zeros)
    const int prow = tid >> 3;                                             // 0 .. 31 (loaders)
    const int pchunk = (tid & 7) ^ chunk_swizzle<BK>(prow & 15);           // source chunk that lands in LDS chunk (tid & 7)
    const unsigned voffA = (unsigned)(((long)prow * g.lda + pchunk * 8) * 2);
    const unsigned voffB = (unsigned)(((long)prow * g.ldb + pchunk * 8) * 2);
    const int strideA = (int)(32 * g.lda * 2), strideB = (int)(32 * g.ldb * 2);  // bytes between the pieces of a wave
    const int wave_dst = wave * 1024;

    // ---- fragment read offsets (as gemm.hip)
    int foff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) foff[s] = l15 * ROWB + (((s * 4 + lq) ^ chunk_swizzle<BK>(l15)) * 16);
    const int a_wave = wm * WTM * ROWB, b_wave = A_BYTES + wn * WTN * ROWB;

    char* eb = lds + wave * IMG;                       // this wave's output image
    char* eb_partner = lds + (wave & 3) * IMG;         // storers: the image of wave - 4 (same columns, the other row half)

    for (int ti = 0; ti < my_tiles; ++ti) {
        const int tile = band_lo + idx + ti * per_xcd;
        const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
        const int m0 = tm * BM, n0 = tn * BN;

        // descriptors of this tile's operand panels (A clipped at row M: out-of-range rows arrive as zeros)
        const long a_rows = g.M - m0 < BM ? g.M - m0 : BM;
        __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (size_t)m0 * g.lda), 0, (int)(a_rows * g.lda * 2), 0x00020000);
        __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + (size_t)n0 * g.ldb), 0, (int)((long)BN * g.ldb * 2), 0x00020000);
        auto stage_piece = [&](int slot, int kt, int i) {  // loaders: piece i < 8 of A, 8 <= i < 16 of B
            char* base = lds + slot * STAGE;
            if (i < PIECES) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lptr_t)(base + i * 4096 + wave_dst), 16, voffA, i * strideA + kt * (BK * 2), 0, 0);
            } else {
                const int j = i - PIECES;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lptr_t)(base + A_BYTES + j * 4096 + wave_dst), 16, voffB, j * strideB + kt * (BK * 2), 0, 0);
            }
        };

        v4f acc[MT][NTL];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

        if (loader) {
#pragma unroll
            for (int i = 0; i < 2 * PIECES; ++i) stage_piece(0, 0, i);
        }
        for (int t = 0; t < nk; ++t) {
            if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage t has landed (the only transfers in flight; loaders never store)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* sb = lds + (t & 1) * STAGE;
            const bool more = loader && t + 1 < nk;  // wave-uniform
            const int pslot = (t + 1) & 1, pkt = t + 1;
            v8bf w0[NTL], x0[MT], w1[NTL], x1[MT];
#pragma unroll
            for (int j = 0; j < NTL; ++j) w0[j] = *reinterpret_cast<const v8bf*>(sb + b_wave + j * 16 * ROWB + foff[0]);
#pragma unroll
            for (int i = 0; i < MT; ++i) x0[i] = *reinterpret_cast<const v8bf*>(sb + a_wave + i * 16 * ROWB + foff[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NTL; ++j) w1[j] = *reinterpret_cast<const v8bf*>(sb + b_wave + j * 16 * ROWB + foff[1]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j], x0[i], acc[i][j], 0, 0, 0);
                x1[i] = *reinterpret_cast<const v8bf*>(sb + a_wave + i * 16 * ROWB + foff[1]);
                if (more) stage_piece(pslot, pkt, i);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j], x1[i], acc[i][j], 0, 0, 0);
                if (more) stage_piece(pslot, pkt, PIECES + i);
            }
        }

        // ---- epilogue.  acc[i][j][r] = C[m0 + wm*128 + 16 i + l15][n0 + wn*64 + 16 j + 4 lq + r]
        __builtin_amdgcn_s_barrier();  // every wave is done with the last stage: the ring becomes eight output images
        asm volatile("" ::: "memory");
        const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;

        v4f bias4[NTL];
        if constexpr (EPI == GRIT_GEMM_BIAS || EPI == GRIT_GEMM_BIAS_GELU) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const v4bf b = *reinterpret_cast<const v4bf*>(g.bias + nw + 16 * j + 4 * lq);
                bias4[j] = v4f{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            }
        }
        auto put = [&](int i, int j, const v4f& v) {
            v4bf p;
            p[0] = (__bf16)v[0]; p[1] = (__bf16)v[1]; p[2] = (__bf16)v[2]; p[3] = (__bf16)v[3];
            const int row = 16 * i + l15;
            const int chunk = (2 * j + (lq >> 1)) ^ (row & 7);
            *reinterpret_cast<v4bf*>(eb + row * 128 + chunk * 16 + (lq & 1) * 8) = p;
        };
        // images written -> visible to the storer waves
        auto publish = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        // storers: one image (rows mrow .. mrow + 127 of the tile's 64-column strip) out as 128-byte row segments, 8 rows per instruction
        auto flush_img = [&](const char* img, int mrow, __bf16* dst, long ld, bool nt) {
            const bool full = mrow + WTM <= g.M;
            __bf16* base = dst + (size_t)(mrow + (lane >> 3)) * ld + nw + (lane & 7) * 8;
#pragma unroll
            for (int it = 0; it < WTM / 8; ++it) {
                const int row = it * 8 + (lane >> 3), chunk = lane & 7;
                const u32x4 v = *reinterpret_cast<const u32x4*>(img + row * 128 + ((chunk ^ (row & 7)) * 16));
                if (full || mrow + row < g.M) {
                    if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(base + (size_t)it * 8 * ld));
                    else *reinterpret_cast<u32x4*>(base + (size_t)it * 8 * ld) = v;
                }
            }
        };
        auto flush_pair = [&](__bf16* dst, long ld, bool nt) {  // storers: own block (rows m0 + 128 ..) and the partner's (rows m0 ..)
            flush_img(eb, mw, dst, ld, nt);
            flush_img(eb_partner, mw - WTM, dst, ld, nt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the images are in registers / on their way: they may be overwritten
        };
        // images free again (the storers have READ them; their stores stay in flight)
        auto release = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
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
            publish();
            if (storer) flush_pair(g.C, g.ldc, false);
            release();
        } else if constexpr (EPI == GRIT_GEMM_BIAS_GELU) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[i][j] += bias4[j];
            if (g.aux) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) put(i, j, acc[i][j]);
                publish();
                if (storer) flush_pair(g.aux, g.ldaux, (g.nt_aux & 1) != 0);
                release();
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    const v2f lo = gelu2(v2f{acc[i][j][0], acc[i][j][1]}), hi = gelu2(v2f{acc[i][j][2], acc[i][j][3]});
                    put(i, j, v4f{lo[0], lo[1], hi[0], hi[1]});
                }
            publish();
            if (storer) flush_pair(g.C, g.ldc, (g.nt_aux & 2) != 0);
            release();
        } else {  // GRIT_GEMM_DGELU
            // the pre-activation blocks come in as whole 128-byte row segments by LDS-DMA, issued by the LOADERS for both images of a pair
            if (loader) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    char* img = lds + (wave + 4 * half) * IMG;
                    const int mrow = m0 + half * WTM;
                    const int chunk = lane & 7;
#pragma unroll
                    for (int it = 0; it < WTM / 8; ++it) {
                        const int row = it * 8 + (lane >> 3);
                        const int m = min(mrow + row, g.M - 1);
                        const __bf16* src = g.aux + (size_t)m * g.ldaux + nw + ((chunk ^ (row & 7)) * 8);
                        if (g.nt_aux & 8) __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(img + it * 1024), 16, 0, 2);
                        else __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(img + it * 1024), 16, 0, 0);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            v4f cs[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) cs[j] = v4f{0.f, 0.f, 0.f, 0.f};
            const bool full_rows = mw + WTM <= g.M;
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
                    for (int r = 0; r < 4; ++r) cs[j][r] += live ? v[r] : 0.f;
                    put(i, j, v);
                }
            }
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[j][r] = row_sum16(cs[j][r]);
            if (l15 == 0) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) *reinterpret_cast<v4f*>(cs_lds + wave * 64 + 16 * j + 4 * lq) = cs[j];
            }
            publish();
            if (storer) {
                flush_pair(g.C, g.ldc, (g.nt_aux & 4) != 0);
                // column sums of both blocks of the pair: slab rows 2 tm (partner) and 2 tm + 1 (own), 64 columns each
                if (lane < 32) {
                    const int half = lane >> 4, w = half ? wave : wave - 4, mrow = m0 + half * WTM;
                    if (mrow < g.M) {
                        const v4f v = *reinterpret_cast<const v4f*>(cs_lds + w * 64 + (lane & 15) * 4);
                        *reinterpret_cast<v4f*>(g.colsum + (size_t)(mrow / WTM) * g.N + nw + (lane & 15) * 4) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            release();
        }
    }
}

}  // namespace

namespace grit_detail {

int gemm_p8_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K, int epilogue,
                   const void* bias, void* aux, long ldaux, float* colsum, int nt, void* stream) {
    if (N % 256 || K % 64 || M <= 0) return GRIT_ERR_UNSUPPORTED;
    if (epilogue < GRIT_GEMM_NONE || epilogue > GRIT_GEMM_DGELU) return GRIT_ERR_UNSUPPORTED;
    constexpr int LDS = 2 * (256 + 256) * 128 + 8 * 64 * 4;  // 130 KB: one workgroup per CU
    P8Args g;
    g.A = (const __bf16*)A; g.lda = lda; g.B = (const __bf16*)B; g.ldb = ldb; g.C = (__bf16*)C; g.ldc = ldc;
    g.bias = (const __bf16*)bias; g.aux = (__bf16*)aux; g.ldaux = ldaux; g.colsum = colsum; g.nt_aux = nt;
    g.M = M; g.N = N; g.K = K;
    g.tiles_m = (M + 255) / 256;
    g.tiles_n = N / 256;
    static grit_detail::PerDevice<int> cus_pd; int& cus = cus_pd();
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GRIT_ERR_LAUNCH;
        cus = prop.multiProcessorCount;
    }
    const int ntiles = g.tiles_m * g.tiles_n;
    const dim3 grid(ntiles < cus ? ntiles : cus), block(512);
#define GRIT_P8_LAUNCH(E)                                                                                            \
    {                                                                                                                \
        auto kern = gemm_p8_bf16<E>;                                                                                 \
        static grit_detail::PerDevice<bool> attr_done_pd; bool& attr_done = attr_done_pd();                          \
        if (!attr_done) {                                                                                            \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) \
                return GRIT_ERR_LAUNCH;                                                                              \
            attr_done = true;                                                                                        \
        }                                                                                                            \
        hipLaunchKernelGGL(kern, grid, block, LDS, (hipStream_t)stream, g);                                          \
    }
    switch (epilogue) {
        case GRIT_GEMM_NONE: GRIT_P8_LAUNCH(GRIT_GEMM_NONE) break;
        case GRIT_GEMM_BIAS: GRIT_P8_LAUNCH(GRIT_GEMM_BIAS) break;
        case GRIT_GEMM_BIAS_GELU: GRIT_P8_LAUNCH(GRIT_GEMM_BIAS_GELU) break;
        default: GRIT_P8_LAUNCH(GRIT_GEMM_DGELU) break;
    }
#undef GRIT_P8_LAUNCH
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}

}  // namespace grit_detail
