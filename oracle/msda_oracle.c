/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle for multi-scale deformable attention: a plain-C, single-threaded restatement of the
 * algorithm of davidnvq/grit's CUDA op, used only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / the timed CPU baseline.  Nothing under grit_amd/
 * may call it.
 *
 * Reference followed (paths into /root/reference):
 *   bilinear read, zero-padded corners ........ models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-84
 *   forward accumulation + point skip rule .... ms_deform_im2col_cuda.cuh:237-299 (test at :288)
 *   backward corner rule, grad of weights ..... ms_deform_im2col_cuda.cuh:87-159
 *   backward reduction over channels .......... ms_deform_im2col_cuda.cuh:406-510 (sum over c of the
 *                                               per-channel partials = the smem tree's result)
 *   host-side layout / zero-init of grads ..... models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153
 *
 * Pinned (tests/test_msda_oracle.py) against golden vectors produced by the reference's own
 * pure-PyTorch statement ms_deform_attn_core_pytorch (models/ops/functions/ms_deform_attn_func.py:41-61)
 * on the reference test's shapes and seed (models/ops/test.py:21-36, D in {30,32,64,71}) and on a
 * GRIT-shaped case with out-of-range / exactly-on-the-border points (tests/golden/make_golden.py).
 *
 * Loop order is the reference's (b, q, m, c outermost; l, p innermost; corners 1..4), so float32
 * results reproduce the reference kernel's own rounding for the forward pass.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_MSDA(T, SUF, FLOOR)                                                                        \
    static T bilinear_##SUF(const T* data, int H, int W, int M, int D, T h, T w, int m, int c) {         \
        const int h_low = (int)FLOOR(h), w_low = (int)FLOOR(w);                                          \
        const int h_high = h_low + 1, w_high = w_low + 1;                                                \
        const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;                                \
        const long ws = (long)M * D, hs = (long)W * ws, base = (long)m * D + c;                          \
        T v1 = 0, v2 = 0, v3 = 0, v4 = 0;                                                                \
        if (h_low >= 0 && w_low >= 0) v1 = data[h_low * hs + w_low * ws + base];                         \
        if (h_low >= 0 && w_high <= W - 1) v2 = data[h_low * hs + w_high * ws + base];                   \
        if (h_high <= H - 1 && w_low >= 0) v3 = data[h_high * hs + w_low * ws + base];                   \
        if (h_high <= H - 1 && w_high <= W - 1) v4 = data[h_high * hs + w_high * ws + base];             \
        const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                                  \
        return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                                                  \
    }                                                                                                    \
                                                                                                         \
    void msda_oracle_fwd_##SUF(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,  \
                               const T* aw, int B, int S, int M, int D, int L, int Lq, int P, T* out) {  \
        for (int b = 0; b < B; ++b)                                                                      \
            for (int q = 0; q < Lq; ++q)                                                                 \
                for (int m = 0; m < M; ++m) {                                                            \
                    const long row = ((long)b * Lq + q) * M + m;                                         \
                    for (int c = 0; c < D; ++c) {                                                        \
                        T col = 0;                                                                       \
                        for (int l = 0; l < L; ++l) {                                                    \
                            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                \
                            const T* vl = value + ((long)b * S + lsi[l]) * M * D;                        \
                            for (int p = 0; p < P; ++p) {                                                \
                                const long i = row * L * P + (long)l * P + p;                            \
                                const T h_im = loc[2 * i + 1] * H - (T)0.5;                              \
                                const T w_im = loc[2 * i] * W - (T)0.5;                                  \
                                if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)                      \
                                    col += bilinear_##SUF(vl, H, W, M, D, h_im, w_im, m, c) * aw[i];     \
                            }                                                                            \
                        }                                                                                \
                        out[row * D + c] = col;                                                          \
                    }                                                                                    \
                }                                                                                        \
    }                                                                                                    \
                                                                                                         \
    /* grad_value is accumulated into (caller zeroes it, like the reference); grad_loc / grad_aw are */  \
    /* overwritten.                                                                                  */  \
    void msda_oracle_bwd_##SUF(const T* value, const int64_t* shapes, const int64_t* lsi, const T* loc,  \
                               const T* aw, const T* go, int B, int S, int M, int D, int L, int Lq,      \
                               int P, T* gv, T* gl, T* ga) {                                             \
        memset(gl, 0, sizeof(T) * (size_t)B * Lq * M * L * P * 2);                                       \
        memset(ga, 0, sizeof(T) * (size_t)B * Lq * M * L * P);                                           \
        for (int b = 0; b < B; ++b)                                                                      \
            for (int q = 0; q < Lq; ++q)                                                                 \
                for (int m = 0; m < M; ++m) {                                                            \
                    const long row = ((long)b * Lq + q) * M + m;                                         \
                    for (int l = 0; l < L; ++l) {                                                        \
                        const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                    \
                        const long lvl = ((long)b * S + lsi[l]) * M * D;                                 \
                        const T* vl = value + lvl;                                                       \
                        T* gvl = gv + lvl;                                                               \
                        for (int p = 0; p < P; ++p) {                                                    \
                            const long i = row * L * P + (long)l * P + p;                                \
                            const T h = loc[2 * i + 1] * H - (T)0.5, w = loc[2 * i] * W - (T)0.5;        \
                            if (!(h > -1 && w > -1 && h < H && w < W)) continue;                         \
                            const int h_low = (int)FLOOR(h), w_low = (int)FLOOR(w);                      \
                            const int h_high = h_low + 1, w_high = w_low + 1;                            \
                            const T lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;            \
                            const long ws = (long)M * D, hs = (long)W * ws;                              \
                            const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;              \
                            T sum_a = 0, sum_x = 0, sum_y = 0;                                           \
                            for (int c = 0; c < D; ++c) {                                                \
                                const long base = (long)m * D + c;                                       \
                                const T top = go[row * D + c];                                           \
                                const T tgv = top * aw[i];                                               \
                                T ghw = 0, gww = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                      \
                                if (h_low >= 0 && w_low >= 0) {                                          \
                                    const long o = h_low * hs + w_low * ws + base;                       \
                                    v1 = vl[o]; ghw -= hw * v1; gww -= hh * v1; gvl[o] += w1 * tgv;      \
                                }                                                                        \
                                if (h_low >= 0 && w_high <= W - 1) {                                     \
                                    const long o = h_low * hs + w_high * ws + base;                      \
                                    v2 = vl[o]; ghw -= lw * v2; gww += hh * v2; gvl[o] += w2 * tgv;      \
                                }                                                                        \
                                if (h_high <= H - 1 && w_low >= 0) {                                     \
                                    const long o = h_high * hs + w_low * ws + base;                      \
                                    v3 = vl[o]; ghw += hw * v3; gww -= lh * v3; gvl[o] += w3 * tgv;      \
                                }                                                                        \
                                if (h_high <= H - 1 && w_high <= W - 1) {                                \
                                    const long o = h_high * hs + w_high * ws + base;                     \
                                    v4 = vl[o]; ghw += lw * v4; gww += lh * v4; gvl[o] += w4 * tgv;      \
                                }                                                                        \
                                const T val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);                   \
                                sum_a += top * val;                                                      \
                                sum_x += W * gww * tgv;                                                  \
                                sum_y += H * ghw * tgv;                                                  \
                            }                                                                            \
                            ga[i] = sum_a; gl[2 * i] = sum_x; gl[2 * i + 1] = sum_y;                     \
                        }                                                                                \
                    }                                                                                    \
                }                                                                                        \
    }

DEFINE_MSDA(float, f32, floorf)
DEFINE_MSDA(double, f64, floor)
