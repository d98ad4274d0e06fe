// Input pipeline on the GPU: the transform of the reference's ImageFolderDataModule (dataloader.py:32-36),
// transforms.Resize((p, p)) + transforms.ToTensor(), applied to decoded 8-bit RGB images resident in device
// memory.
//
// torchvision's Resize on a PIL image is Pillow's Image.resize with the bilinear filter, an antialiasing
// two-pass resampler in fixed point (Pillow 12.2, src/libImaging/Resample.c): per output coordinate a window
// [xmin, xmin + n) of source pixels and weights w(x) = tri((x + xmin - centre + 0.5) / max(scale, 1)),
// normalised in double precision and rounded to 22 fractional bits; a pass computes
// clip8((2^21 + sum pixel * weight) >> 22) and the horizontal pass's 8-bit result feeds the vertical pass.
// vamd_resize_coeffs reproduces those tables on the host (double arithmetic in Pillow's operation order),
// the kernels the two passes; ToTensor's division by 255 (fp32) is fused into the last pass, which writes
// the channels-first fp32 image the data loader would have produced.  Results are bit-identical to the
// Pillow + torchvision path, so a run fed by this pipeline sees the same batches.
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)

namespace vamd {

constexpr int kImgPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ int img_clip8(int v) {
    v >>= kImgPrecisionBits;                              // arithmetic shift, as Pillow's table lookup indexes
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one thread per (row, output column): the three channels of a pixel share the window
__global__ void __launch_bounds__(256)
img_resize_h_kernel(const uint8_t* __restrict__ src, int H, int W, int ps, const int32_t* __restrict__ bounds,
                    const int32_t* __restrict__ coef, int ksize, uint8_t* __restrict__ dst, int Wo) {
    const int64_t total = (int64_t)H * Wo;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(idx % Wo);
        const int64_t y = idx / Wo;
        const int xmin = bounds[2 * xo], n = bounds[2 * xo + 1];
        const int32_t* k = coef + (int64_t)xo * ksize;
        const uint8_t* p = src + (y * W + xmin) * ps;          // ps: bytes per source pixel (3 packed, 4 with a pad byte)
        int s0 = 1 << (kImgPrecisionBits - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            const int w = k[x];
            s0 += (int)p[ps * x] * w;
            s1 += (int)p[ps * x + 1] * w;
            s2 += (int)p[ps * x + 2] * w;
        }
        uint8_t* o = dst + idx * 3;
        o[0] = (uint8_t)img_clip8(s0);
        o[1] = (uint8_t)img_clip8(s1);
        o[2] = (uint8_t)img_clip8(s2);
    }
}

// one thread per (output row, column, channel); bounds == NULL: no vertical resampling (copy + convert)
__global__ void __launch_bounds__(256)
img_resize_v_kernel(const uint8_t* __restrict__ src, int H, int W, int ps, const int32_t* __restrict__ bounds,
                    const int32_t* __restrict__ coef, int ksize, int Ho, float* __restrict__ dst_chw,
                    uint8_t* __restrict__ dst_u8) {
    const int64_t total = (int64_t)Ho * W * 3;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % 3);
        const int64_t pix = idx / 3;
        const int x = (int)(pix % W);
        const int yo = (int)(pix / W);
        int v;
        if (bounds) {
            const int ymin = bounds[2 * yo], n = bounds[2 * yo + 1];
            const int32_t* k = coef + (int64_t)yo * ksize;
            int s = 1 << (kImgPrecisionBits - 1);
            for (int y = 0; y < n; ++y) s += (int)src[((int64_t)(ymin + y) * W + x) * ps + c] * k[y];
            v = img_clip8(s);
        } else {
            v = src[pix * ps + c];
        }
        if (dst_u8) dst_u8[idx] = (uint8_t)v;
        if (dst_chw) dst_chw[((int64_t)c * Ho + yo) * W + x] = (float)v / 255.0f;     // ToTensor: uint8 -> float, div(255)
    }
}

static double img_triangle(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

}  // namespace vamd

using namespace vamd;

extern "C" int vamd_resize_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    double filterscale = (double)(float)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    return (int)ceil(support) * 2 + 1;
}

extern "C" int vamd_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coef, int ksize) {
    VAMD_REQUIRE(in_size > 0 && out_size > 0 && bounds && coef, "vamd_resize_coeffs: bad argument");
    VAMD_REQUIRE(ksize == vamd_resize_ksize(in_size, out_size), "vamd_resize_coeffs: ksize %d, expected %d", ksize,
                 vamd_resize_ksize(in_size, out_size));
    const float in0 = 0.f, in1 = (float)in_size;          // the box is the whole image, held as floats
    double filterscale, scale;
    filterscale = scale = (double)(in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 1.0 * filterscale;
    double* k = (double*)malloc(sizeof(double) * (size_t)ksize);
    if (!k) return fail(VAMD_EINVAL, "vamd_resize_coeffs: out of host memory");
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        const double ss = 1.0 / filterscale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int x;
        for (x = 0; x < xmax; ++x) {
            const double w = img_triangle((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        for (; x < ksize; ++x) k[x] = 0.0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
        for (x = 0; x < ksize; ++x) {
            const double v = k[x];
            coef[(int64_t)xx * ksize + x] = v < 0 ? (int32_t)(-0.5 + v * (1 << kImgPrecisionBits))
                                                  : (int32_t)(0.5 + v * (1 << kImgPrecisionBits));
        }
    }
    free(k);
    return 0;
}

extern "C" int vamd_image_resize_u8(const uint8_t* src, int H, int W, int pixel_bytes, const int32_t* hbounds, const int32_t* hcoef, int hk,
                                    const int32_t* vbounds, const int32_t* vcoef, int vk, uint8_t* tmp, int Ho, int Wo,
                                    float* dst_chw, uint8_t* dst_u8, void* stream) {
    VAMD_REQUIRE(src && H > 0 && W > 0 && Ho > 0 && Wo > 0 && (dst_chw || dst_u8), "vamd_image_resize_u8: bad argument");
    VAMD_REQUIRE(pixel_bytes == 3 || pixel_bytes == 4, "vamd_image_resize_u8: %d bytes per source pixel (3 or 4)", pixel_bytes);
    VAMD_REQUIRE((hbounds != nullptr) == (W != Wo) && (vbounds != nullptr) == (H != Ho),
                 "vamd_image_resize_u8: coefficient tables must be given exactly for the axes that change (%dx%d -> %dx%d)",
                 H, W, Ho, Wo);
    VAMD_REQUIRE(!hbounds || (hcoef && hk > 0 && tmp), "vamd_image_resize_u8: horizontal pass needs its table and a temporary");
    VAMD_REQUIRE(!vbounds || (vcoef && vk > 0), "vamd_image_resize_u8: vertical pass needs its table");
    hipStream_t st = as_stream(stream);
    const uint8_t* mid = src;
    if (hbounds) {
        hipLaunchKernelGGL(img_resize_h_kernel, dim3(grid_for((int64_t)H * Wo, 256, 65536)), dim3(256), 0, st, src, H, W, pixel_bytes,
                           hbounds, hcoef, hk, tmp, Wo);
        VAMD_LAUNCH_OK("img_resize_h");
        mid = tmp;
        pixel_bytes = 3;
    }
    hipLaunchKernelGGL(img_resize_v_kernel, dim3(grid_for((int64_t)Ho * Wo * 3, 256, 65536)), dim3(256), 0, st, mid, H, Wo,
                       pixel_bytes, vbounds, vcoef, vk, Ho, dst_chw, dst_u8);
    VAMD_LAUNCH_OK("img_resize_v");
    return 0;
}
