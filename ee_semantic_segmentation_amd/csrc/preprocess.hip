// Input pipeline on the device (reference get_seg_datasets.py:49-86): the torchvision chain
//   image : Resize (PIL bilinear, antialiased) -> CenterCrop -> ToTensor -> Normalize(mean, std)
//   target: Resize (PIL forces NEAREST on palette images) -> CenterCrop -> ToTensor*255 -> long -> 255 -> void
// on decoded uint8 pixels.  Bit-exact with Pillow: the resampling coefficients / source indices are computed on
// the host exactly as Pillow's Resample.c / Geometry.c do (doubles, 22-bit fixed-point weights, incremental
// nearest coordinates) and the kernels only apply the integer tables, with Pillow's 8-bit intermediate image
// between the horizontal and the vertical pass.
#include <math.h>

#include "eeseg_common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;     // Pillow Resample.c

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: src [H][W][C] u8 -> tmp [H][Wr][C] u8
__global__ __launch_bounds__(256) void resample_h_u8_kernel(const uint8_t* __restrict__ src, int H, int W, int C, int Wr,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk,
                                                            int ksize, uint8_t* __restrict__ tmp) {
    const long long total = (long long)H * Wr * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long t = i / C;
        const int xx = (int)(t % Wr);
        const int y = (int)(t / Wr);
        const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
        int ss = 1 << (PRECISION_BITS - 1);
        const uint8_t* row = src + ((long long)y * W + x0) * C + c;
        for (int k = 0; k < n; ++k) ss += (int)row[(long long)k * C] * kk[xx * ksize + k];
        tmp[i] = (uint8_t)clip8(ss);
    }
}

// vertical pass + centre crop + ToTensor (/255) + Normalize: tmp [H][Wr][C] u8 -> out [C][Dh][Dw] f32
__global__ __launch_bounds__(256) void resample_v_crop_norm_kernel(const uint8_t* __restrict__ tmp, int Wr, int C,
                                                                   const int* __restrict__ bounds,
                                                                   const int* __restrict__ kk, int ksize, int crop_top,
                                                                   int crop_left, int Dh, int Dw,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ stdv, float* __restrict__ out) {
    const long long total = (long long)C * Dh * Dw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Dw);
        const long long t = i / Dw;
        const int y = (int)(t % Dh);
        const int c = (int)(t / Dh);
        const int yy = y + crop_top, xx = x + crop_left;
        const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
        int ss = 1 << (PRECISION_BITS - 1);
        const uint8_t* col = tmp + ((long long)y0 * Wr + xx) * C + c;
        for (int k = 0; k < n; ++k) ss += (int)col[(long long)k * Wr * C] * kk[yy * ksize + k];
        const float v = (float)clip8(ss) / 255.0f;           // ToTensor
        out[i] = (v - mean[c]) / stdv[c];                     // Normalize: sub then div, fp32
    }
}

// target: nearest resize (host index tables) + centre crop + label look-up table -> int64
__global__ __launch_bounds__(256) void label_resize_crop_lut_kernel(const uint8_t* __restrict__ src, int W,
                                                                    const int* __restrict__ yidx,
                                                                    const int* __restrict__ xidx, int crop_top,
                                                                    int crop_left, int Dh, int Dw,
                                                                    const int64_t* __restrict__ lut,
                                                                    int64_t* __restrict__ out) {
    const long long total = (long long)Dh * Dw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Dw), y = (int)(i / Dw);
        const int sy = yidx[y + crop_top], sx = xidx[x + crop_left];
        out[i] = lut[src[(long long)sy * W + sx]];
    }
}

int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

// Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter.
// Returns ksize (taps per output sample); fills bounds [out][2] = (first source index, tap count) and
// kk [out][ksize] when they are given (call once with NULLs to size them).
extern "C" int eeseg_pil_bilinear_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize_capacity) {
    EESEG_CHECK(in_size > 0 && out_size > 0, EESEG_ERR_ARG, "pil_bilinear_coeffs: bad size");
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (!bounds || !kk) return ksize;
    EESEG_CHECK(ksize <= 64, EESEG_ERR_TOO_LARGE, "pil_bilinear_coeffs: down-scaling factor above 31 is not supported");
    EESEG_CHECK(ksize_capacity >= ksize, EESEG_ERR_ARG, "pil_bilinear_coeffs: kk holds %d taps, %d needed", ksize_capacity, ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double w[64];
        double ww = 0.0;
        for (int x = 0; x < ksize; ++x) w[x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double t = (x + xmin - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            w[x] = t < 1.0 ? 1.0 - t : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) w[x] /= ww;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
        for (int x = 0; x < ksize_capacity; ++x) {
            const double v = x < ksize ? w[x] : 0.0;
            kk[xx * ksize_capacity + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
    }
    return ksize;
}

// Pillow Geometry.c ImagingScaleAffine, NEAREST: source index of every output sample; the coordinate is
// ACCUMULATED in double (xo += scale), which is what makes it differ from floor((x + 0.5) * scale).
extern "C" int eeseg_pil_nearest_index(int in_size, int out_size, int32_t* idx) {
    EESEG_CHECK(in_size > 0 && out_size > 0 && idx, EESEG_ERR_ARG, "pil_nearest_index: bad argument");
    const double a0 = (double)in_size / (double)out_size;
    double xo = a0 * 0.5;
    for (int x = 0; x < out_size; ++x) {
        int xin = (int)floor(xo);
        if (xin < 0) xin = 0;
        if (xin > in_size - 1) xin = in_size - 1;
        idx[x] = xin;
        xo += a0;
    }
    return EESEG_OK;
}

extern "C" int eeseg_preprocess_image_u8(const uint8_t* src, int H, int W, int C, int Hr, int Wr, const int32_t* hbounds,
                                         const int32_t* hkk, int hksize, const int32_t* vbounds, const int32_t* vkk,
                                         int vksize, int crop_top, int crop_left, int Dh, int Dw, const float* mean,
                                         const float* stdv, uint8_t* tmp, float* out, void* stream) {
    EESEG_CHECK(src && hbounds && hkk && vbounds && vkk && mean && stdv && tmp && out, EESEG_ERR_ARG,
                "preprocess_image: null pointer");
    EESEG_CHECK(H > 0 && W > 0 && C > 0 && C <= 4 && Hr > 0 && Wr > 0 && hksize > 0 && vksize > 0, EESEG_ERR_ARG,
                "preprocess_image: bad shape");
    EESEG_CHECK(crop_top >= 0 && crop_left >= 0 && Dh > 0 && Dw > 0 && crop_top + Dh <= Hr && crop_left + Dw <= Wr,
                EESEG_ERR_ARG, "preprocess_image: crop window [%d+%d, %d+%d] outside the resized image %dx%d", crop_top, Dh,
                crop_left, Dw, Hr, Wr);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(resample_h_u8_kernel, dim3(grid_for((long long)H * Wr * C)), dim3(256), 0, st, src, H, W, C, Wr, hbounds,
                       hkk, hksize, tmp);
    hipLaunchKernelGGL(resample_v_crop_norm_kernel, dim3(grid_for((long long)C * Dh * Dw)), dim3(256), 0, st, tmp, Wr, C,
                       vbounds, vkk, vksize, crop_top, crop_left, Dh, Dw, mean, stdv, out);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_preprocess_label_u8(const uint8_t* src, int H, int W, const int32_t* yidx, const int32_t* xidx, int Hr,
                                         int Wr, int crop_top, int crop_left, int Dh, int Dw, const int64_t* lut,
                                         int64_t* out, void* stream) {
    EESEG_CHECK(src && yidx && xidx && lut && out, EESEG_ERR_ARG, "preprocess_label: null pointer");
    EESEG_CHECK(H > 0 && W > 0 && crop_top >= 0 && crop_left >= 0 && Dh > 0 && Dw > 0 && crop_top + Dh <= Hr &&
                    crop_left + Dw <= Wr, EESEG_ERR_ARG, "preprocess_label: crop window outside the resized image");
    hipLaunchKernelGGL(label_resize_crop_lut_kernel, dim3(grid_for((long long)Dh * Dw)), dim3(256), 0, (hipStream_t)stream, src,
                       W, yidx, xidx, crop_top, crop_left, Dh, Dw, lut, out);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
