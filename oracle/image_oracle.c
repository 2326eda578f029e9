/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle for the image side of the batch contract:
 *
 *   PIL image (RGB, uint8)  --resize(BICUBIC)-->  uint8  --ToTensor-->  f32 / 255  --Normalize-->  (x - mean) / std
 *   --nested_tensor_from_tensor_list-->  zero-padded [B,3,H,W] f32 + mask [B,H,W] (True on padding)
 *
 * Reference call sites (paths into /root/reference):
 *   x.resize((neww, newh), resample=Image.BICUBIC) ....... datasets/caption/transforms/utils.py:11-16 (MaxWHResize),
 *                                                          :26-45 (MinMaxResize)
 *   Compose([resize, ToTensor(), normalize()]) ........... datasets/caption/transforms/__init__.py:6-32
 *   padding + mask ....................................... engine/utils.py:278-295
 *
 * The resampling arithmetic itself is third-party: Pillow (unpinned in the reference's requirements.txt; 12.2.0 in
 * this image), src/libImaging/Resample.c.  Restated here from its published algorithm: per axis a table of
 * (first tap, tap count, normalised taps) computed in double precision with the bicubic convolution kernel
 * (a = -0.5, support 2, stretched by the scale when shrinking), taps rounded to 22-bit fixed point, horizontal pass
 * then vertical pass with a uint8 intermediate, each output = clamp((2^21 + sum tap*pixel) >> 22).
 *
 * Pinned (tests/test_image_oracle.py) against Pillow itself on seeded images (up- and down-scaling, both axes,
 * identity) and against committed vectors tests/golden/image_g11.npz produced by tests/golden/make_golden.py from
 * the reference's own transform classes.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION_BITS 22

static double cubic(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

/* taps for one axis; returns ksize.  bounds = out_size pairs (first, count); taps = out_size * ksize ints. */
static int axis_table(int in_size, int out_size, int** bounds_out, int** taps_out) {
    double scale = (double)((float)in_size - 0.0f) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    int* bounds = (int*)malloc(sizeof(int) * 2 * out_size);
    int* taps = (int*)malloc(sizeof(int) * (size_t)out_size * ksize);
    double* k = (double*)malloc(sizeof(double) * ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale, ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            const double w = cubic((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0;
        for (x = 0; x < ksize; ++x)
            taps[(size_t)xx * ksize + x] =
                k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    free(k);
    *bounds_out = bounds;
    *taps_out = taps;
    return ksize;
}

static uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
}

/* src [h, w, 3] uint8 -> dst [oh, ow, 3] uint8 (Pillow Image.resize(..., BICUBIC) on an RGB image) */
void oracle_resize_bicubic_rgb(const uint8_t* src, int h, int w, uint8_t* dst, int oh, int ow) {
    int *xb, *xt, *yb, *yt;
    const int kx = axis_table(w, ow, &xb, &xt), ky = axis_table(h, oh, &yb, &yt);
    uint8_t* tmp = (uint8_t*)malloc((size_t)h * ow * 3);
    for (int y = 0; y < h; ++y)
        for (int xx = 0; xx < ow; ++xx)
            for (int c = 0; c < 3; ++c) {
                int ss = 1 << (PRECISION_BITS - 1);
                for (int x = 0; x < xb[2 * xx + 1]; ++x)
                    ss += src[((size_t)y * w + x + xb[2 * xx]) * 3 + c] * xt[(size_t)xx * kx + x];
                tmp[((size_t)y * ow + xx) * 3 + c] = clip8(ss);
            }
    for (int yy = 0; yy < oh; ++yy)
        for (int xx = 0; xx < ow; ++xx)
            for (int c = 0; c < 3; ++c) {
                int ss = 1 << (PRECISION_BITS - 1);
                for (int y = 0; y < yb[2 * yy + 1]; ++y)
                    ss += tmp[((size_t)(y + yb[2 * yy]) * ow + xx) * 3 + c] * yt[(size_t)yy * ky + y];
                dst[((size_t)yy * ow + xx) * 3 + c] = clip8(ss);
            }
    free(tmp); free(xb); free(xt); free(yb); free(yt);
}

/* ToTensor + Normalize + placement into a zero-padded batch slot: img [h, w, 3] uint8 -> out [3, H, W] f32 (slot of
 * the batch tensor, H >= h, W >= w), mask [H, W] (1 on padding). */
void oracle_to_padded_slot(const uint8_t* img, int h, int w, const float* mean, const float* std, float* out,
                           uint8_t* mask, int H, int W) {
    memset(out, 0, sizeof(float) * 3 * (size_t)H * W);
    memset(mask, 1, (size_t)H * W);
    for (int c = 0; c < 3; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const float v = (float)img[((size_t)y * w + x) * 3 + c] / 255.0f;
                out[((size_t)c * H + y) * W + x] = (v - mean[c]) / std[c];
            }
    for (int y = 0; y < h; ++y) memset(mask + (size_t)y * W, 0, (size_t)w);
}
