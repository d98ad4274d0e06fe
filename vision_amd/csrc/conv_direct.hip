// Direct (non-MFMA) convolution kernels: any channel count, fp32 or bf16 storage, fp32 accumulate.
// They serve the thin layers of the UNet (3 input or output channels, where an MFMA tile would be
// >90 % padding), every shape in fp32 mode, and shapes the MFMA path does not cover.
#include "common.h"
#include "conv_shapes.h"

namespace vamd {

template <typename T, int VEC>
__device__ __forceinline__ float dot_chan(const T* __restrict__ a, const T* __restrict__ b, int n) {
    float acc = 0.f;
    if (VEC > 1) {
        for (int i = 0; i < n; i += VEC) {
            const Vec16<T> va = *reinterpret_cast<const Vec16<T>*>(a + i);
            const Vec16<T> vb = *reinterpret_cast<const Vec16<T>*>(b + i);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc = fmaf(to_f(va.v[j]), to_f(vb.v[j]), acc);
        }
    } else {
        for (int i = 0; i < n; ++i) acc = fmaf(to_f(a[i]), to_f(b[i]), acc);
    }
    return acc;
}

// thread per (output pixel, cout), cout fastest
template <typename T, int VEC>
__global__ void conv_fwd_direct_kernel(ConvGeom g, const T* __restrict__ x, const T* __restrict__ w,
                                       const float* __restrict__ bias, const float* __restrict__ temb,
                                       const T* __restrict__ res, const T* __restrict__ res2,
                                       T* __restrict__ y) {
    const int64_t total = (int64_t)g.M * g.Cout;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = idx / g.Cout;
        const int co = (int)(idx - m * g.Cout);
        const int b = (int)(m / ((int64_t)g.Ho * g.Wo));
        const int rem = (int)(m - (int64_t)b * g.Ho * g.Wo);
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        float acc = 0.f;
        for (int r = 0; r < g.k; ++r) {
            const int u = ho * g.stride + r - g.pad;
            if (u < 0 || u >= g.He) continue;
            const int hi = g.up ? (u >> 1) : u;
            for (int s = 0; s < g.k; ++s) {
                const int v = wo * g.stride + s - g.pad;
                if (v < 0 || v >= g.We) continue;
                const int wi = g.up ? (v >> 1) : v;
                const T* xp = x + (((int64_t)b * g.H + hi) * g.W + wi) * g.Cin;
                const T* wp = w + (((int64_t)co * g.k + r) * g.k + s) * g.Cin;
                acc += dot_chan<T, VEC>(xp, wp, g.Cin);
            }
        }
        if (bias) acc += bias[co];
        if (temb) acc += temb[(int64_t)b * g.Cout + co];
        if (res) acc += to_f(res[idx]);
        if (res2) acc += to_f(res2[idx]);
        y[idx] = from_f<T>(acc);
    }
}

// thread per (input pixel, cin), cin fastest; gathers over the output pixels that used this input.
// w_t is [Cin][k][k][Cout] with both kernel axes flipped.
template <typename T, int VEC>
__global__ void conv_dgrad_direct_kernel(ConvGeom g, const T* __restrict__ dy,
                                         const T* __restrict__ wt, const T* __restrict__ add,
                                         T* __restrict__ dx) {
    const int64_t total = (int64_t)g.B * g.H * g.W * g.Cin;
    const int nsub = g.up ? 2 : 1;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / g.Cin;
        const int ci = (int)(idx - p * g.Cin);
        const int b = (int)(p / ((int64_t)g.H * g.W));
        const int rem = (int)(p - (int64_t)b * g.H * g.W);
        const int hi = rem / g.W, wi = rem - hi * g.W;
        float acc = 0.f;
        for (int du = 0; du < nsub; ++du)
            for (int dv = 0; dv < nsub; ++dv) {
                const int U = g.up ? 2 * hi + du : hi;
                const int V = g.up ? 2 * wi + dv : wi;
                for (int rf = 0; rf < g.k; ++rf) {
                    const int nu = U + g.pad - (g.k - 1 - rf);
                    if (nu < 0 || (nu % g.stride) != 0) continue;
                    const int ho = nu / g.stride;
                    if (ho >= g.Ho) continue;
                    for (int sf = 0; sf < g.k; ++sf) {
                        const int nv = V + g.pad - (g.k - 1 - sf);
                        if (nv < 0 || (nv % g.stride) != 0) continue;
                        const int wo = nv / g.stride;
                        if (wo >= g.Wo) continue;
                        const T* dp = dy + (((int64_t)b * g.Ho + ho) * g.Wo + wo) * g.Cout;
                        const T* wp = wt + (((int64_t)ci * g.k + rf) * g.k + sf) * g.Cout;
                        acc += dot_chan<T, VEC>(dp, wp, g.Cout);
                    }
                }
            }
        if (add) acc += to_f(add[idx]);
        dx[idx] = from_f<T>(acc);
    }
}

// grid (weight elements / 256, pixel chunks): thread owns one dw element, loops over a pixel chunk,
// one fp32 atomic per thread at the end.  dw must be zero on entry.
template <typename T>
__global__ void conv_wgrad_direct_kernel(ConvGeom g, const T* __restrict__ dy,
                                         const T* __restrict__ x, float* __restrict__ dw,
                                         int64_t rows_per_chunk) {
    const int64_t nW = (int64_t)g.Cout * g.k * g.k * g.Cin;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= nW) return;
    const int ci = (int)(idx % g.Cin);
    int64_t rest = idx / g.Cin;
    const int s = (int)(rest % g.k); rest /= g.k;
    const int r = (int)(rest % g.k);
    const int co = (int)(rest / g.k);
    const int64_t m0 = (int64_t)blockIdx.y * rows_per_chunk;
    int64_t m1 = m0 + rows_per_chunk;
    if (m1 > g.M) m1 = g.M;
    float acc = 0.f;
    const int HoWo = g.Ho * g.Wo;
    for (int64_t m = m0; m < m1; ++m) {
        const int b = (int)(m / HoWo);
        const int rem = (int)(m - (int64_t)b * HoWo);
        const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
        const int u = ho * g.stride + r - g.pad, v = wo * g.stride + s - g.pad;
        if (u < 0 || u >= g.He || v < 0 || v >= g.We) continue;
        const int hi = g.up ? (u >> 1) : u, wi = g.up ? (v >> 1) : v;
        acc = fmaf(to_f(dy[m * g.Cout + co]),
                   to_f(x[(((int64_t)b * g.H + hi) * g.W + wi) * g.Cin + ci]), acc);
    }
    atomicAdd(dw + idx, acc);
}

// C[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][n][k]; thread per output element; ld* = row strides.
template <typename T, typename TO, int VEC>
__global__ void bmm_nt_direct_kernel(const T* __restrict__ A, const T* __restrict__ Bm,
                                     TO* __restrict__ C, int batch, int M, int N, int K, int lda, int ldb,
                                     int ldc, float alpha) {
    const int64_t total = (int64_t)batch * M * N;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const int64_t bm = idx / N;
        const int64_t b = bm / M;
        const T* ap = A + bm * lda;
        const T* bp = Bm + (b * N + n) * (int64_t)ldb;
        C[bm * ldc + n] = from_f<TO>(alpha * dot_chan<T, VEC>(ap, bp, K));
    }
}

// C[b][m][n] = alpha * sum_k A[b][k][m] * Bm[b][k][n]; thread per output element.
template <typename T>
__global__ void bmm_tn_direct_kernel(const T* __restrict__ A, const T* __restrict__ Bm, T* __restrict__ C,
                                     int batch, int M, int N, int K, int lda, int ldb, int ldc, float alpha) {
    const int64_t total = (int64_t)batch * M * N;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % N);
        const int64_t bm = idx / N;
        const int m = (int)(bm % M);
        const int64_t b = bm / M;
        const T* ap = A + b * (int64_t)K * lda + m;
        const T* bp = Bm + b * (int64_t)K * ldb + n;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(to_f(ap[(int64_t)k * lda]), to_f(bp[(int64_t)k * ldb]), acc);
        C[bm * ldc + n] = from_f<T>(alpha * acc);
    }
}

// y[b][h][w][c] = sum of the 2x2 block of x (gradient of nearest 2x upsampling)
template <typename T>
__global__ void sumpool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    const int64_t total = (int64_t)B * H * W * C;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        int64_t p = idx / C;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int64_t b = p / H;
        const int64_t row = (int64_t)2 * W * C;
        const T* s = x + ((b * 2 * H + 2 * h) * 2 * W + 2 * w) * (int64_t)C + c;
        y[idx] = from_f<T>(to_f(s[0]) + to_f(s[C]) + to_f(s[row]) + to_f(s[row + C]));
    }
}

int conv_fwd_direct(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                    const float* temb, const void* res, const void* res2, void* y, hipStream_t st) {
    const int64_t total = (int64_t)g.M * g.Cout;
    VAMD_DISPATCH(dtype, "conv_fwd_direct", {
        constexpr int VEC = Vec16<T>::N;
        if (g.Cin % VEC == 0)
            hipLaunchKernelGGL((conv_fwd_direct_kernel<T, VEC>), dim3(grid_for(total, 256, 65536)), dim3(256), 0, st,
                               g, (const T*)x, (const T*)w, bias, temb, (const T*)res, (const T*)res2, (T*)y);
        else
            hipLaunchKernelGGL((conv_fwd_direct_kernel<T, 1>), dim3(grid_for(total, 256, 65536)), dim3(256), 0, st,
                               g, (const T*)x, (const T*)w, bias, temb, (const T*)res, (const T*)res2, (T*)y);
    });
    VAMD_LAUNCH_OK("conv_fwd_direct");
    return 0;
}

int conv_dgrad_direct(const ConvGeom& g, int dtype, const void* dy, const void* wt, const void* add,
                      void* dx, hipStream_t st) {
    const int64_t total = (int64_t)g.B * g.H * g.W * g.Cin;
    VAMD_DISPATCH(dtype, "conv_dgrad_direct", {
        constexpr int VEC = Vec16<T>::N;
        if (g.Cout % VEC == 0)
            hipLaunchKernelGGL((conv_dgrad_direct_kernel<T, VEC>), dim3(grid_for(total, 256, 65536)), dim3(256), 0, st,
                               g, (const T*)dy, (const T*)wt, (const T*)add, (T*)dx);
        else
            hipLaunchKernelGGL((conv_dgrad_direct_kernel<T, 1>), dim3(grid_for(total, 256, 65536)), dim3(256), 0, st,
                               g, (const T*)dy, (const T*)wt, (const T*)add, (T*)dx);
    });
    VAMD_LAUNCH_OK("conv_dgrad_direct");
    return 0;
}

int conv_wgrad_direct(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw,
                      hipStream_t st) {
    const int64_t nW = (int64_t)g.Cout * g.k * g.k * g.Cin;
    hipError_t e = hipMemsetAsync(dw, 0, nW * sizeof(float), st);
    if (e != hipSuccess) return fail((int)e, "conv_wgrad_direct: memset failed: %s", hipGetErrorString(e));
    int64_t nchunk = (g.M + 63) / 64;
    if (nchunk > 512) nchunk = 512;
    const int64_t rows_per = (g.M + nchunk - 1) / nchunk;
    dim3 grid((unsigned)((nW + 255) / 256), (unsigned)nchunk);
    VAMD_DISPATCH(dtype, "conv_wgrad_direct", {
        hipLaunchKernelGGL((conv_wgrad_direct_kernel<T>), grid, dim3(256), 0, st, g, (const T*)dy,
                           (const T*)x, dw, rows_per);
    });
    VAMD_LAUNCH_OK("conv_wgrad_direct");
    return 0;
}

}  // namespace vamd

using namespace vamd;

extern "C" {

int vamd_bmm_nt(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda, int ldb,
                int ldc, float alpha, int dtype, int out_f32, int algo, void* stream) {
    VAMD_REQUIRE(A && Bm && C && batch > 0 && batch <= 65535 && M > 0 && N > 0 && K > 0, "vamd_bmm_nt: bad argument");
    VAMD_REQUIRE(lda >= K && ldb >= K && ldc >= N, "vamd_bmm_nt: row strides smaller than the rows");
    VAMD_REQUIRE((uintptr_t)A % 16 == 0 && (uintptr_t)Bm % 16 == 0 && (uintptr_t)C % 16 == 0,
                 "vamd_bmm_nt: operands must be 16-byte aligned");
    if (algo != 1 && bmm_nt_mfma_supported(M, N, K, lda, ldb, ldc, dtype, out_f32))
        return bmm_nt_mfma(A, Bm, C, batch, M, N, K, lda, ldb, ldc, alpha, dtype, out_f32, as_stream(stream));
    if (algo == 2) return fail(VAMD_EUNSUPPORTED, "vamd_bmm_nt: MFMA path does not cover M=%d N=%d K=%d ld=%d,%d,%d", M, N, K, lda, ldb, ldc);
    const int64_t total = (int64_t)batch * M * N;
    VAMD_DISPATCH(dtype, "vamd_bmm_nt", {
        constexpr int VEC = Vec16<T>::N;
        const dim3 grid(grid_for(total, 256, 65536));
        if (K % VEC == 0 && lda % VEC == 0 && ldb % VEC == 0) {
            if (out_f32)
                hipLaunchKernelGGL((bmm_nt_direct_kernel<T, float, VEC>), grid, dim3(256), 0, as_stream(stream),
                                   (const T*)A, (const T*)Bm, (float*)C, batch, M, N, K, lda, ldb, ldc, alpha);
            else
                hipLaunchKernelGGL((bmm_nt_direct_kernel<T, T, VEC>), grid, dim3(256), 0, as_stream(stream),
                                   (const T*)A, (const T*)Bm, (T*)C, batch, M, N, K, lda, ldb, ldc, alpha);
        } else {
            if (out_f32)
                hipLaunchKernelGGL((bmm_nt_direct_kernel<T, float, 1>), grid, dim3(256), 0, as_stream(stream),
                                   (const T*)A, (const T*)Bm, (float*)C, batch, M, N, K, lda, ldb, ldc, alpha);
            else
                hipLaunchKernelGGL((bmm_nt_direct_kernel<T, T, 1>), grid, dim3(256), 0, as_stream(stream),
                                   (const T*)A, (const T*)Bm, (T*)C, batch, M, N, K, lda, ldb, ldc, alpha);
        }
    });
    VAMD_LAUNCH_OK("vamd_bmm_nt");
    return 0;
}

int vamd_bmm_tn(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda, int ldb,
                int ldc, float alpha, int dtype, int algo, void* stream) {
    VAMD_REQUIRE(A && Bm && C && batch > 0 && batch <= 65535 && M > 0 && N > 0 && K > 0, "vamd_bmm_tn: bad argument");
    VAMD_REQUIRE(lda >= M && ldb >= N && ldc >= N, "vamd_bmm_tn: row strides smaller than the rows");
    VAMD_REQUIRE((uintptr_t)A % 16 == 0 && (uintptr_t)Bm % 16 == 0, "vamd_bmm_tn: operands must be 16-byte aligned");
    if (algo != 1 && bmm_tn_mfma_supported(M, N, K, lda, ldb, ldc, dtype))
        return bmm_tn_mfma(A, Bm, C, batch, M, N, K, lda, ldb, ldc, alpha, dtype, as_stream(stream));
    if (algo == 2) return fail(VAMD_EUNSUPPORTED, "vamd_bmm_tn: MFMA path does not cover M=%d N=%d K=%d ld=%d,%d,%d", M, N, K, lda, ldb, ldc);
    const int64_t total = (int64_t)batch * M * N;
    VAMD_DISPATCH(dtype, "vamd_bmm_tn", {
        hipLaunchKernelGGL((bmm_tn_direct_kernel<T>), dim3(grid_for(total, 256, 65536)), dim3(256), 0,
                           as_stream(stream), (const T*)A, (const T*)Bm, (T*)C, batch, M, N, K, lda, ldb, ldc, alpha);
    });
    VAMD_LAUNCH_OK("vamd_bmm_tn");
    return 0;
}

int vamd_sumpool2x2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    VAMD_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0, "vamd_sumpool2x2: bad argument");
    const int64_t total = (int64_t)B * H * W * C;
    VAMD_DISPATCH(dtype, "vamd_sumpool2x2", {
        hipLaunchKernelGGL((sumpool2x2_kernel<T>), dim3(grid_for(total, 256)), dim3(256), 0,
                           as_stream(stream), (const T*)x, (T*)y, B, H, W, C);
    });
    VAMD_LAUNCH_OK("vamd_sumpool2x2");
    return 0;
}

}  // extern "C"
