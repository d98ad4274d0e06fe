// Thin-channel convolutions: the layers of the UNet that touch the 3-channel image (first block,
// last block, End conv) at full resolution.  An MFMA tile would be >90 % padding there; these are
// bandwidth-bound direct kernels (stride 1, 3x3 pad 1 or 1x1) with fp32 accumulation:
//   thin_in   few input channels (<= 4)  -> any Cout : thread = pixel x 16 output channels
//   thin_out  any Cin (multiple of 8/4)  -> few output channels (<= 4) : thread = pixel
//   thin_wgrad  weight gradient when one side has <= 4 channels: lanes span the wide side, each
//               thread keeps the k*k*few partial sums in registers over a strip of pixels; block
//               partials go to a slab that a second kernel sums in a fixed order (deterministic).
// The input gradient of such a layer is the same computation with the flipped, transposed filter.
#include "conv_shapes.h"

namespace vamd {

constexpr int kThinMaxSmall = 4;
constexpr int kCoTile = 16;

// ---------------------------------------------------------------------------------- thin_in
template <typename T, int K>
__global__ void __launch_bounds__(256)
thin_in_kernel(ConvGeom g, const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
               const float* __restrict__ temb, const T* __restrict__ res, const T* __restrict__ res2,
               T* __restrict__ y) {
    // weights of this block's 16 output channels as floats in LDS, [tap][ci (padded to 4)][co]:
    // every lane reads the same address (broadcast), 4 output channels per ds_read_b128
    __shared__ __attribute__((aligned(16))) float ws[K * K * kThinMaxSmall * kCoTile];
    const int co0 = blockIdx.y * kCoTile;
    constexpr int taps = K * K;
    for (int i = threadIdx.x; i < taps * kThinMaxSmall * kCoTile; i += 256) {
        const int co = i % kCoTile, c = (i / kCoTile) % kThinMaxSmall, t = i / (kCoTile * kThinMaxSmall);
        ws[i] = (co0 + co < g.Cout && c < g.Cin) ? to_f(w[((int64_t)(co0 + co) * taps + t) * g.Cin + c]) : 0.f;
    }
    __syncthreads();
    const int HW = g.H * g.W;
    for (int64_t m = blockIdx.x * (int64_t)256 + threadIdx.x; m < g.M; m += (int64_t)gridDim.x * 256) {
        const int b = (int)(m / HW);
        const int rem = (int)(m - (int64_t)b * HW);
        const int ho = rem / g.W, wo = rem - ho * g.W;
        float xin[K * K * kThinMaxSmall];
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const int hi = ho + r - g.pad, wi = wo + s - g.pad;
                const bool ok = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                const T* xp = x + (((int64_t)b * g.H + hi) * g.W + wi) * g.Cin;
#pragma unroll
                for (int c = 0; c < kThinMaxSmall; ++c)
                    xin[(r * K + s) * kThinMaxSmall + c] = (ok && c < g.Cin) ? to_f(xp[c]) : 0.f;
            }
        float acc[kCoTile];
#pragma unroll
        for (int j = 0; j < kCoTile; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < taps; ++t)
#pragma unroll
            for (int c = 0; c < kThinMaxSmall; ++c) {
                if (c < g.Cin) {                      // uniform; padded channels hold zeros anyway
                    const float xv = xin[t * kThinMaxSmall + c];
                    const float4* wv = reinterpret_cast<const float4*>(ws + (t * kThinMaxSmall + c) * kCoTile);
#pragma unroll
                    for (int j4 = 0; j4 < kCoTile / 4; ++j4) {
                        const float4 q = wv[j4];
                        acc[j4 * 4 + 0] = fmaf(xv, q.x, acc[j4 * 4 + 0]);
                        acc[j4 * 4 + 1] = fmaf(xv, q.y, acc[j4 * 4 + 1]);
                        acc[j4 * 4 + 2] = fmaf(xv, q.z, acc[j4 * 4 + 2]);
                        acc[j4 * 4 + 3] = fmaf(xv, q.w, acc[j4 * 4 + 3]);
                    }
                }
            }
        const int64_t o = m * g.Cout + co0;
#pragma unroll
        for (int j = 0; j < kCoTile; ++j) {
            if (co0 + j < g.Cout) {
                float v = acc[j];
                if (bias) v += bias[co0 + j];
                if (temb) v += temb[(int64_t)b * g.Cout + co0 + j];
                if (res) v += to_f(res[o + j]);
                if (res2) v += to_f(res2[o + j]);
                acc[j] = v;
            }
        }
        if (co0 + kCoTile <= g.Cout && (g.Cout * sizeof(T)) % 16 == 0) {
            constexpr int VEC = Vec16<T>::N;
#pragma unroll
            for (int j = 0; j < kCoTile; j += VEC) {
                Vec16<T> ov;
#pragma unroll
                for (int e = 0; e < VEC; ++e) ov.v[e] = from_f<T>(acc[j + e]);
                *reinterpret_cast<Vec16<T>*>(y + o + j) = ov;
            }
        } else {
#pragma unroll
            for (int j = 0; j < kCoTile; ++j)
                if (co0 + j < g.Cout) y[o + j] = from_f<T>(acc[j]);
        }
    }
}

// --------------------------------------------------------------------------------- thin_out
template <typename T, int K>
__global__ void __launch_bounds__(256)
thin_out_kernel(ConvGeom g, const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                const float* __restrict__ temb, const T* __restrict__ res, const T* __restrict__ res2,
                T* __restrict__ y) {
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float wsm[];      // [tap][ci][4]
    const int taps = K * K;
    for (int i = threadIdx.x; i < taps * g.Cin * kThinMaxSmall; i += 256) {
        const int co = i & 3, rest = i >> 2;                         // rest = tap*Cin + ci
        wsm[i] = co < g.Cout ? to_f(w[(int64_t)co * taps * g.Cin + rest]) : 0.f;
    }
    __syncthreads();
    const int HW = g.H * g.W;
    for (int64_t m = blockIdx.x * (int64_t)256 + threadIdx.x; m < g.M; m += (int64_t)gridDim.x * 256) {
        const int b = (int)(m / HW);
        const int rem = (int)(m - (int64_t)b * HW);
        const int ho = rem / g.W, wo = rem - ho * g.W;
        float acc[kThinMaxSmall] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const int hi = ho + r - g.pad, wi = wo + s - g.pad;
                if ((unsigned)hi >= (unsigned)g.H || (unsigned)wi >= (unsigned)g.W) continue;
                const T* xp = x + (((int64_t)b * g.H + hi) * g.W + wi) * g.Cin;
                const float4* wp = reinterpret_cast<const float4*>(wsm) + (r * K + s) * g.Cin;
                for (int c = 0; c < g.Cin; c += VEC) {
                    const Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(xp + c);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float xf = to_f(xv.v[e]);
                        const float4 wv = wp[c + e];
                        acc[0] = fmaf(xf, wv.x, acc[0]);
                        acc[1] = fmaf(xf, wv.y, acc[1]);
                        acc[2] = fmaf(xf, wv.z, acc[2]);
                        acc[3] = fmaf(xf, wv.w, acc[3]);
                    }
                }
            }
#pragma unroll
        for (int co = 0; co < kThinMaxSmall; ++co) {
            if (co < g.Cout) {
                float v = acc[co];
                if (bias) v += bias[co];
                if (temb) v += temb[(int64_t)b * g.Cout + co];
                if (res) v += to_f(res[m * g.Cout + co]);
                if (res2) v += to_f(res2[m * g.Cout + co]);
                y[m * g.Cout + co] = from_f<T>(v);
            }
        }
    }
}

// ------------------------------------------------------------------------------- thin_wgrad
// WIDE_IS_OUT: the wide side is dY (Cout wide, Cin <= 4); else the wide side is X (Cin wide, Cout <= 4).
// Block = PL pixel lanes x WL wide-channel lanes (WL = 64, or 4 when both sides are thin).
template <typename T, int K, bool WIDE_IS_OUT, int WL>
__global__ void __launch_bounds__(256)
thin_wgrad_kernel(ConvGeom g, const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ slab,
                  int64_t pix_per_block) {
    constexpr int PL = 256 / WL;
    constexpr int NACC = K * K * kThinMaxSmall;
    __shared__ float red[PL][NACC][WL + 1];
    const int wl = threadIdx.x % WL, pl = threadIdx.x / WL;
    const int wide = WIDE_IS_OUT ? g.Cout : g.Cin;
    const int small = WIDE_IS_OUT ? g.Cin : g.Cout;
    const int HW = g.H * g.W;
    float* out = slab + (int64_t)blockIdx.x * ((wide + WL - 1) / WL) * NACC * WL;
    for (int w0 = 0; w0 < wide; w0 += WL) {
        const int wc = w0 + wl;
        const bool wok = wc < wide;
        float acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
        const int64_t m0 = (int64_t)blockIdx.x * pix_per_block;
        int64_t m1 = m0 + pix_per_block;
        if (m1 > g.M) m1 = g.M;
        for (int64_t m = m0 + pl; m < m1; m += PL) {
            const int b = (int)(m / HW);
            const int rem = (int)(m - (int64_t)b * HW);
            const int ho = rem / g.W, wo = rem - ho * g.W;
            if (WIDE_IS_OUT) {
                const float d = wok ? to_f(dy[m * g.Cout + wc]) : 0.f;
#pragma unroll
                for (int r = 0; r < K; ++r)
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        const int hi = ho + r - g.pad, wi = wo + s - g.pad;
                        const bool ok = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                        const T* xp = x + (((int64_t)b * g.H + hi) * g.W + wi) * g.Cin;
#pragma unroll
                        for (int c = 0; c < kThinMaxSmall; ++c) {
                            const float xv = (ok && c < small) ? to_f(xp[c]) : 0.f;
                            acc[(r * K + s) * kThinMaxSmall + c] = fmaf(d, xv, acc[(r * K + s) * kThinMaxSmall + c]);
                        }
                    }
            } else {
                float d[kThinMaxSmall];
#pragma unroll
                for (int c = 0; c < kThinMaxSmall; ++c) d[c] = c < small ? to_f(dy[m * g.Cout + c]) : 0.f;
#pragma unroll
                for (int r = 0; r < K; ++r)
#pragma unroll
                    for (int s = 0; s < K; ++s) {
                        const int hi = ho + r - g.pad, wi = wo + s - g.pad;
                        const bool ok = wok && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                        const float xv = ok ? to_f(x[(((int64_t)b * g.H + hi) * g.W + wi) * g.Cin + wc]) : 0.f;
#pragma unroll
                        for (int c = 0; c < kThinMaxSmall; ++c)
                            acc[(r * K + s) * kThinMaxSmall + c] = fmaf(d[c], xv, acc[(r * K + s) * kThinMaxSmall + c]);
                    }
            }
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) red[pl][i][wl] = acc[i];
        __syncthreads();
        for (int i = threadIdx.x; i < NACC * WL; i += 256) {
            const int a = i / WL, l = i - a * WL;
            float sum = 0.f;
#pragma unroll
            for (int p = 0; p < PL; ++p) sum += red[p][a][l];
            out[((int64_t)(w0 / WL) * NACC + a) * WL + l] = sum;
        }
        __syncthreads();
    }
}

// sums the block slabs in a fixed order and scatters to dw[co][tap][ci]
template <int K, bool WIDE_IS_OUT, int WL>
__global__ void __launch_bounds__(256)
thin_wgrad_reduce_kernel(ConvGeom g, const float* __restrict__ slab, int nblocks, float* __restrict__ dw) {
    constexpr int NACC = K * K * kThinMaxSmall;
    __shared__ float red[4][64];
    const int wide = WIDE_IS_OUT ? g.Cout : g.Cin;
    const int small = WIDE_IS_OUT ? g.Cin : g.Cout;
    const int ntile = (wide + WL - 1) / WL;
    const int per_block = ntile * NACC * WL;
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e;                      // consecutive threads -> consecutive entries
    float sum = 0.f;
    if (idx < per_block)
        for (int b = q; b < nblocks; b += 4) sum += slab[(int64_t)b * per_block + idx];
    red[q][e] = sum;
    __syncthreads();
    if (q != 0 || idx >= per_block) return;
    sum = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    const int l = idx % WL, a = (idx / WL) % NACC, tile = idx / (WL * NACC);
    const int wc = tile * WL + l, c = a % kThinMaxSmall, tap = a / kThinMaxSmall;
    if (wc >= wide || c >= small) return;
    const int co = WIDE_IS_OUT ? wc : c, ci = WIDE_IS_OUT ? c : wc;
    dw[((int64_t)co * K * K + tap) * g.Cin + ci] = sum;
}

// ------------------------------------------------------------------------------------ host
bool conv_thin_supported(const ConvGeom& g, int dtype) {
    const int vec = dtype == VAMD_BF16 ? 8 : 4;
    if (g.stride != 1 || g.up) return false;
    if (g.Cin <= kThinMaxSmall) return true;                       // thin_in handles any Cout
    return g.Cout <= kThinMaxSmall && g.Cin % vec == 0;             // thin_out
}

int conv_fwd_thin(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                  const float* temb, const void* res, const void* res2, void* y, hipStream_t st) {
    const int blocks = grid_for(g.M, 256, 16384);
    VAMD_DISPATCH(dtype, "conv_fwd_thin", {
        if (g.Cin <= kThinMaxSmall) {
            dim3 grid(blocks, (g.Cout + kCoTile - 1) / kCoTile);
            if (g.k == 3)
                hipLaunchKernelGGL((thin_in_kernel<T, 3>), grid, dim3(256), 0, st, g, (const T*)x, (const T*)w, bias, temb,
                                   (const T*)res, (const T*)res2, (T*)y);
            else
                hipLaunchKernelGGL((thin_in_kernel<T, 1>), grid, dim3(256), 0, st, g, (const T*)x, (const T*)w, bias, temb,
                                   (const T*)res, (const T*)res2, (T*)y);
        } else {
            const size_t lds = (size_t)g.k * g.k * g.Cin * kThinMaxSmall * sizeof(float);
            if (lds > 64 * 1024) return fail(VAMD_EUNSUPPORTED, "conv_fwd_thin: Cin=%d too wide for the thin-output kernel", g.Cin);
            if (g.k == 3)
                hipLaunchKernelGGL((thin_out_kernel<T, 3>), dim3(blocks), dim3(256), lds, st, g, (const T*)x, (const T*)w, bias,
                                   temb, (const T*)res, (const T*)res2, (T*)y);
            else
                hipLaunchKernelGGL((thin_out_kernel<T, 1>), dim3(blocks), dim3(256), lds, st, g, (const T*)x, (const T*)w, bias,
                                   temb, (const T*)res, (const T*)res2, (T*)y);
        }
    });
    VAMD_LAUNCH_OK("conv_fwd_thin");
    return 0;
}

int conv_dgrad_thin(const ConvGeom& g, int dtype, const void* dy, const void* wt, const void* add, void* dx,
                    hipStream_t st) {
    // stride-1 input gradient = the same kind of convolution over dY with the flipped, transposed
    // filter wt [Cin][k][k][Cout]: swap the channel roles
    ConvGeom t = g;
    t.Cin = g.Cout;
    t.Cout = g.Cin;
    t.pad = g.k - 1 - g.pad;
    return conv_fwd_thin(t, dtype, dy, wt, nullptr, nullptr, add, nullptr, dx, st);
}

static int thin_wgrad_blocks(const ConvGeom& g) {
    int64_t n = (g.M + 2047) / 2048;
    if (n > 1024) n = 1024;
    if (n < 1) n = 1;
    return (int)n;
}

bool conv_wgrad_thin_supported(const ConvGeom& g, int dtype) {
    return g.stride == 1 && !g.up && (g.Cin <= kThinMaxSmall || g.Cout <= kThinMaxSmall);
}

size_t conv_wgrad_thin_workspace(const ConvGeom& g, int dtype) {
    const bool wide_out = g.Cin <= kThinMaxSmall;
    const int wide = wide_out ? g.Cout : g.Cin;
    const int WL = wide <= 4 ? 4 : 64;
    const int nacc = g.k * g.k * kThinMaxSmall;
    return (size_t)thin_wgrad_blocks(g) * ((wide + WL - 1) / WL) * nacc * WL * sizeof(float);
}

template <typename T, int K, bool WO, int WL>
static void launch_thin_wgrad(const ConvGeom& g, const void* dy, const void* x, float* dw, float* slab, hipStream_t st) {
    const int nb = thin_wgrad_blocks(g);
    const int64_t per = (g.M + nb - 1) / nb;
    hipLaunchKernelGGL((thin_wgrad_kernel<T, K, WO, WL>), dim3(nb), dim3(256), 0, st, g, (const T*)dy, (const T*)x, slab, per);
    const int wide = WO ? g.Cout : g.Cin;
    const int entries = ((wide + WL - 1) / WL) * K * K * kThinMaxSmall * WL;
    hipLaunchKernelGGL((thin_wgrad_reduce_kernel<K, WO, WL>), dim3((entries + 63) / 64), dim3(256), 0, st, g, slab, nb, dw);
}

int conv_wgrad_thin(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw, void* workspace,
                    size_t workspace_bytes, hipStream_t st) {
    if (!workspace || workspace_bytes < conv_wgrad_thin_workspace(g, dtype))
        return fail(VAMD_EINVAL, "conv_wgrad_thin: needs %zu workspace bytes", conv_wgrad_thin_workspace(g, dtype));
    float* slab = reinterpret_cast<float*>(workspace);
    const bool wide_out = g.Cin <= kThinMaxSmall;
    const int wide = wide_out ? g.Cout : g.Cin;
    VAMD_DISPATCH(dtype, "conv_wgrad_thin", {
        if (g.k == 3) {
            if (wide <= 4) launch_thin_wgrad<T, 3, true, 4>(g, dy, x, dw, slab, st);
            else if (wide_out) launch_thin_wgrad<T, 3, true, 64>(g, dy, x, dw, slab, st);
            else launch_thin_wgrad<T, 3, false, 64>(g, dy, x, dw, slab, st);
        } else {
            if (wide <= 4) launch_thin_wgrad<T, 1, true, 4>(g, dy, x, dw, slab, st);
            else if (wide_out) launch_thin_wgrad<T, 1, true, 64>(g, dy, x, dw, slab, st);
            else launch_thin_wgrad<T, 1, false, 64>(g, dy, x, dw, slab, st);
        }
    });
    VAMD_LAUNCH_OK("conv_wgrad_thin");
    return 0;
}

}  // namespace vamd
