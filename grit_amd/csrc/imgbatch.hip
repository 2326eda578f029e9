// Image side of the batch contract on the device: decoded RGB uint8 images of different sizes ->
//   Pillow-exact bicubic resize -> ToTensor (/255) -> Normalize ((x - mean) / std) -> zero-padded [B,3,H,W] f32 + mask.
// Replaces the per-image host chain of the reference (datasets/caption/transforms/utils.py:4-45,
// transforms/__init__.py:6-32, engine/utils.py:278-295) by two launches per batch.
//
// Arithmetic is Pillow's (src/libImaging/Resample.c, 8 bits per channel): per axis a table of (first tap, tap count,
// taps in 22-bit fixed point) built in double precision on the host (grit_resample_taps_bicubic below), a horizontal
// pass into a uint8 intermediate, a vertical pass, each output = clamp((2^21 + sum tap * pixel) >> 22).  All integer,
// so the result is bit-identical to Image.resize(..., BICUBIC); the float stage is a 3 x 256 lookup table computed by
// the caller with the framework's own float ops.
//
// Both kernels are byte streams: pass 1 reads 3*h*w and writes 3*h*ow bytes per image, pass 2 reads 3*h*ow and writes
// 13*H*W (three f32 planes + the mask).  Tap tables are a few KB per image and stay in L2.  The intermediate keeps
// dword-aligned rows (pitch = 3*ow rounded up to 4) so that pass 2 reads it four bytes at a time.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/grit_hip.h"

namespace {

constexpr int kPrecisionBits = 22;

struct ImageDesc {  // one row of the int64 descriptor table, see grit_hip.h
    int64_t src_off, src_h, src_w, dst_h, dst_w, kx, ky, xb_off, xt_off, yb_off, yt_off, tmp_off;
};
static_assert(sizeof(ImageDesc) == GRIT_IMAGE_DESC_FIELDS * sizeof(int64_t), "descriptor layout");

constexpr int kRows = 4;  // source rows per thread in the horizontal pass

__host__ __device__ __forceinline__ int tmp_pitch(int dst_w) { return (3 * dst_w + 3) & ~3; }  // bytes, dword-aligned rows

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= kPrecisionBits;
    return (uint8_t)min(max(v, 0), 255);
}

// pass 1, generic: one thread per output byte (pixel x, channel c) of one source row; any tap count
__global__ __launch_bounds__(256)
void resample_rows_any(const uint8_t* __restrict__ src, const ImageDesc* __restrict__ desc,
                       const int32_t* __restrict__ tables, uint8_t* __restrict__ tmp) {
    const ImageDesc d = desc[blockIdx.z];
    const int y = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;  // byte within the output row
    const int ow = (int)d.dst_w, w = (int)d.src_w;
    if (y >= d.src_h || j >= ow * 3) return;
    const int xx = j / 3, c = j - xx * 3;
    const int2 b = *reinterpret_cast<const int2*>(tables + d.xb_off + 2 * xx);
    const int32_t* __restrict__ k = tables + d.xt_off + (int64_t)xx * d.kx;
    const uint8_t* __restrict__ p = src + d.src_off + ((int64_t)y * w + b.x) * 3 + c;
    int ss = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < b.y; ++x) ss += (int)p[3 * x] * k[x];
    tmp[d.tmp_off + (int64_t)y * tmp_pitch(ow) + j] = clip8(ss);
}

// pass 1, up to K taps (K = 7 covers every scale <= 1.25 incl. all upscaling, K = 9 scales <= 2, K = 13 scales <= 3): one thread per output
// pixel and kRows consecutive source rows.  The taps (shared by the rows) sit in registers; the 3*K source bytes of a
// pixel-row are fetched as aligned dwords and shifted into place (v_alignbyte) instead of 3*K byte loads.  Bytes read
// beyond the tap count meet zero taps; the caller pads the blob so that they stay readable.
template <int K>
__global__ __launch_bounds__(256)
void resample_rows_k(const uint8_t* __restrict__ src, const ImageDesc* __restrict__ desc,
                     const int32_t* __restrict__ tables, uint8_t* __restrict__ tmp) {
    constexpr int kWords = (3 * K + 3) / 4;  // realigned dwords holding the 3*K bytes
    __shared__ uint32_t packed[4][kRows][48];  // per wave: the 64 pixels x 3 bytes of each row, re-read as 48 dwords
    const ImageDesc d = desc[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x_wave = blockIdx.x * 256 + wave * 64;  // first output pixel of this wave
    const int y0 = blockIdx.y * kRows;
    const int ow = (int)d.dst_w, w = (int)d.src_w, h = (int)d.src_h, kx = (int)d.kx;
    if (y0 >= h || x_wave >= ow) return;  // wave-uniform
    const int xx = min(x_wave + lane, ow - 1);  // lanes past the row end repeat the last pixel; their bytes are not stored
    const int first = tables[d.xb_off + 2 * xx];
    const int32_t* __restrict__ k = tables + d.xt_off + (int64_t)xx * kx;
    int tap[K];
#pragma unroll
    for (int t = 0; t < K; ++t) tap[t] = t < kx ? k[t] : 0;  // table rows are zero past the tap count
    const int pitch = tmp_pitch(ow);
    uint8_t* __restrict__ staged = reinterpret_cast<uint8_t*>(&packed[wave][0][0]);
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
        const int y = min(y0 + r, h - 1);
        const int64_t a = d.src_off + ((int64_t)y * w + first) * 3;
        const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(src + (a & ~(int64_t)3));
        const uint32_t shift = (uint32_t)(a & 3);
        uint32_t raw[kWords + 1], word[kWords];
#pragma unroll
        for (int i = 0; i <= kWords; ++i) raw[i] = q[i];
#pragma unroll
        for (int i = 0; i < kWords; ++i) word[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], shift);
        int s[3] = {1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1), 1 << (kPrecisionBits - 1)};
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int j = 3 * t + c;
                s[c] += __mul24((int)((word[j >> 2] >> (8 * (j & 3))) & 0xffu), tap[t]);
            }
        uint8_t* __restrict__ o = staged + r * 192 + 3 * lane;
        o[0] = clip8(s[0]); o[1] = clip8(s[1]); o[2] = clip8(s[2]);
    }
    // the wave's own LDS rows: no barrier needed beyond the wave's program order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int byte0 = 3 * x_wave + 4 * lane;  // byte of the output row this lane's dword starts at (x_wave*3 % 4 == 0)
    if (lane < 48 && byte0 < pitch) {
#pragma unroll
        for (int r = 0; r < kRows; ++r)
            if (y0 + r < h)
                *reinterpret_cast<uint32_t*>(tmp + d.tmp_off + (int64_t)(y0 + r) * pitch + byte0) = packed[wave][r][lane];
    }
}

// pass 2: one thread per 4 consecutive pixels (12 bytes = 3 aligned dwords of the intermediate row; the taps of a row are
// wave-uniform): vertical taps, lookup, one float4 store per colour plane, 4 mask bytes, zero padding.
__global__ __launch_bounds__(256)
void resample_cols_normalize(const uint8_t* __restrict__ tmp, const ImageDesc* __restrict__ desc,
                             const int32_t* __restrict__ tables, const float* __restrict__ lut, int out_h, int out_w,
                             float* __restrict__ out, uint8_t* __restrict__ mask) {
    __shared__ float table[3 * 256];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) table[i] = lut[i];
    __syncthreads();
    const ImageDesc d = desc[blockIdx.z];
    // Workgroups go to the 8 XCDs round-robin in launch order and every XCD has its own L2.  Neighbouring output rows
    // share most of their source rows, so each XCD gets one contiguous band of rows (gridDim.y is a multiple of 8 when
    // the grid is one workgroup wide) instead of every eighth row -- otherwise each L2 fetches the whole intermediate.
    int yy = blockIdx.y;
    if (gridDim.x == 1) yy = (yy & 7) * (gridDim.y >> 3) + (yy >> 3);
    const int px0 = 4 * (blockIdx.x * 256 + threadIdx.x);
    if (yy >= out_h || px0 >= out_w) return;
    const int ow = (int)d.dst_w, pitch = tmp_pitch(ow);
    int s[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = 1 << (kPrecisionBits - 1);
    const bool inside = yy < d.dst_h && px0 < ow;
    if (inside) {
        const int first = tables[d.yb_off + 2 * yy], count = tables[d.yb_off + 2 * yy + 1];
        const int32_t* __restrict__ k = tables + d.yt_off + (int64_t)yy * d.ky;
        const uint8_t* __restrict__ p = tmp + d.tmp_off + (int64_t)first * pitch + 3 * px0;
        // the last pixels of a row may reach past the pitch: read only the dwords that belong to the row
        const int words = min(3, (pitch - 3 * px0) >> 2);
        for (int y = 0; y < count; ++y) {
            const uint32_t* __restrict__ q = reinterpret_cast<const uint32_t*>(p);
            uint32_t v[3];
            v[0] = q[0];
            v[1] = words > 1 ? q[1] : 0u;
            v[2] = words > 2 ? q[2] : 0u;
            const int t = k[y];
#pragma unroll
            for (int i = 0; i < 12; ++i) s[i] += __mul24((int)((v[i >> 2] >> (8 * (i & 3))) & 0xffu), t);
            p += pitch;
        }
    }
    const int64_t plane = (int64_t)out_h * out_w;
    float* __restrict__ o = out + (int64_t)blockIdx.z * 3 * plane + (int64_t)yy * out_w + px0;
    uint8_t* __restrict__ m = mask + (int64_t)blockIdx.z * plane + (int64_t)yy * out_w + px0;
    float f[3][4];
    uint32_t pad = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool real = inside && px0 + i < ow;
        pad |= real ? 0u : 1u << (8 * i);
#pragma unroll
        for (int c = 0; c < 3; ++c) f[c][i] = real ? table[c * 256 + clip8(s[3 * i + c])] : 0.f;
    }
    if ((out_w & 3) == 0) {  // rows of the planes are 16-byte aligned
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(o + c * plane) = make_float4(f[c][0], f[c][1], f[c][2], f[c][3]);
        *reinterpret_cast<uint32_t*>(m) = pad;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (px0 + i < out_w) {
#pragma unroll
                for (int c = 0; c < 3; ++c) o[c * plane + i] = f[c][i];
                m[i] = (uint8_t)(pad >> (8 * i));
            }
    }
}

}  // namespace

#pragma clang fp contract(off)  // host tables must round exactly like Pillow's: no fused multiply-add
static double cubic_weight(double x) {  // bicubic convolution kernel, a = -0.5
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

extern "C" int grit_resample_taps_bicubic(int in_size, int out_size, int32_t* bounds, int32_t* taps, long taps_capacity) {
    if (in_size <= 0 || out_size <= 0) return -GRIT_ERR_BAD_ARG;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale, inv = 1.0 / filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (!bounds && !taps) return ksize;
    if (!bounds || !taps || taps_capacity < (long)out_size * ksize) return -GRIT_ERR_BAD_ARG;
    double stack_k[64];
    double* k = ksize <= 64 ? stack_k : new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int first = (int)(center - support + 0.5);
        if (first < 0) first = 0;
        int last = (int)(center + support + 0.5);
        if (last > in_size) last = in_size;
        const int count = last - first;
        double total = 0.0;
        for (int x = 0; x < count; ++x) {
            k[x] = cubic_weight((x + first - center + 0.5) * inv);
            total += k[x];
        }
        int32_t* t = taps + (long)xx * ksize;
        for (int x = 0; x < ksize; ++x) {
            double v = 0.0;
            if (x < count) v = total != 0.0 ? k[x] / total : k[x];
            t[x] = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
        }
        bounds[2 * xx] = first;
        bounds[2 * xx + 1] = count;
    }
    if (k != stack_k) delete[] k;
    return ksize;
}

extern "C" int grit_image_batch_fwd(const uint8_t* src, const int64_t* desc, const int32_t* tables, uint8_t* tmp,
                                    const float* lut, int batch, int max_src_h, int max_dst_w, int max_kx, int out_h,
                                    int out_w, float* out, uint8_t* mask, void* stream) {
    if (!src || !desc || !tables || !tmp || !lut || !out || !mask) return GRIT_ERR_BAD_ARG;
    if (batch <= 0 || max_src_h <= 0 || max_dst_w <= 0 || max_kx <= 0 || out_h <= 0 || out_w <= 0 || max_dst_w > out_w)
        return GRIT_ERR_BAD_ARG;
    if (((uintptr_t)src | (uintptr_t)tmp) & 3) return GRIT_ERR_BAD_ARG;
    if (max_src_h > 65535 || out_h > 65535 || batch > 65535) return GRIT_ERR_UNSUPPORTED;
    const ImageDesc* d = reinterpret_cast<const ImageDesc*>(desc);
    const hipStream_t s = (hipStream_t)stream;
    const dim3 pixels((max_dst_w + 255) / 256, (max_src_h + kRows - 1) / kRows, batch);
    if (max_kx <= 7)
        hipLaunchKernelGGL(resample_rows_k<7>, pixels, dim3(256), 0, s, src, d, tables, tmp);
    else if (max_kx <= 9)
        hipLaunchKernelGGL(resample_rows_k<9>, pixels, dim3(256), 0, s, src, d, tables, tmp);
    else if (max_kx <= 13)
        hipLaunchKernelGGL(resample_rows_k<13>, pixels, dim3(256), 0, s, src, d, tables, tmp);
    else
        hipLaunchKernelGGL(resample_rows_any, dim3((max_dst_w * 3 + 255) / 256, max_src_h, batch), dim3(256), 0, s,
                           src, d, tables, tmp);
    const int wide = (out_w + 1023) / 1024;
    hipLaunchKernelGGL(resample_cols_normalize, dim3(wide, wide == 1 ? (out_h + 7) / 8 * 8 : out_h, batch), dim3(256), 0, s,
                       tmp, d, tables, lut, out_h, out_w, out, mask);
    return hipGetLastError() == hipSuccess ? GRIT_OK : GRIT_ERR_LAUNCH;
}
