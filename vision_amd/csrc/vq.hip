// Vector-quantisation codebook lookup of reference models/vqvaeunet.py:115-146 (VQBlock.forward) and its
// gradients, fp32.
//
// The lookup is an argmin over K codes of
//
//     dist[m][k] = (sum_d z[m][d]^2 + sum_d e[k][d]^2) - 2 * sum_d z[m][d] * e[k][d]          (vqvaeunet.py:121-123)
//
// and its result is an INDEX, so the order of the fp32 operations is part of the contract: every sum runs
// over d = 0, 1, ... D-1 with one fused multiply-add per term, starting from +0; the two squared norms are added
// first, then twice the dot product is subtracted; the smallest distance wins and, among equal distances, the
// smallest code index (torch.argmin's first minimum); a NaN distance counts as smaller than every number.  A
// thread owns whole (row, code) sums, so tiling changes nothing about that order, and the CPU restatement used
// by the tests reproduces it with fmaf().
//
// Kernels (all HBM/L2-resident data: 2 MB of codebook and < 1 MB of latents at the butterfly configuration):
//   vq_dist_kernel      32 rows x 128 codes per workgroup, 4 x 4 register tile per thread, operands staged
//                       through LDS in slices of 32 features; the winner of each row is merged across the
//                       code chunks with one 64-bit atomic minimum on (ordered distance bits << 32 | code).
//   vq_finalize_kernel  one wave per row: index out, straight-through output z + (q - z)  (vqvaeunet.py:144),
//                       squared-error partial sums (one per workgroup, fixed order).
//   vq_loss_kernel      sums the partials in a fixed order (fp64) -> mse, vq_loss = mse * beta + mse
//                       (vqvaeunet.py:138-141; the two mse terms have equal values in the forward pass).
//   vq_bwd_latents_kernel / vq_bwd_codebook_kernel   gradients of the commitment term and the straight-through
//                       path w.r.t. the latents, and of the embedding term w.r.t. the codebook (rows of one code
//                       are added in a fixed order by one workgroup, eight interleaved partial sums: bit-reproducible).
#include "common.h"

#pragma clang fp contract(off)

namespace vamd {

constexpr int kVqRows = 32, kVqCodes = 128, kVqSlice = 32;

// Latent element (m, d): rows are (image, pixel) pairs; channels-last rows are contiguous, channels-first
// ("nchw", the layout VQBlock.forward receives) has the pixel index fastest.
struct VqLayout {
    int64_t sB, sP, sD;
    int HW;
    __device__ __forceinline__ int64_t at(int64_t m, int d) const { return (m / HW) * sB + (m % HW) * sP + (int64_t)d * sD; }
};

static VqLayout vq_layout(int D, int HW, int nchw) {
    VqLayout l;
    l.HW = HW;
    if (nchw) { l.sB = (int64_t)D * HW; l.sP = 1; l.sD = HW; }
    else      { l.sB = (int64_t)D * HW; l.sP = D; l.sD = 1; }
    return l;
}

// monotone map of a distance to an unsigned key: smaller distance <=> smaller key; -0 == +0; NaN first
__device__ __forceinline__ uint32_t vq_order_bits(float d) {
    if (d != d) return 0u;
    d = d + 0.0f;
    const uint32_t u = __float_as_uint(d);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(256)
vq_dist_kernel(const float* __restrict__ z, const float* __restrict__ E, int64_t M, int D, int K, VqLayout lay,
               unsigned long long* __restrict__ best) {
    __shared__ __align__(16) float zs[kVqSlice][kVqRows];
    __shared__ __align__(16) float es[kVqSlice][kVqCodes + 4];
    const int tid = threadIdx.x, cg = tid & 31, rg = tid >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * kVqRows;
    const int c0 = blockIdx.y * kVqCodes;
    float dot[4][4], zz[4], ee[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        zz[i] = 0.f;
        ee[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) dot[i][j] = 0.f;
    }
    for (int d0 = 0; d0 < D; d0 += kVqSlice) {
#pragma unroll
        for (int i = 0; i < kVqRows * kVqSlice / 256; ++i) {
            const int e = tid + 256 * i, r = e / kVqSlice, dd = e % kVqSlice;
            const int64_t m = m0 + r;
            zs[dd][r] = (m < M && d0 + dd < D) ? z[lay.at(m, d0 + dd)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < kVqCodes * kVqSlice / 256; ++i) {
            const int e = tid + 256 * i, c = e / kVqSlice, dd = e % kVqSlice;
            es[dd][c] = (c0 + c < K && d0 + dd < D) ? E[(int64_t)(c0 + c) * D + d0 + dd] : 0.f;
        }
        __syncthreads();
        // zero padding past D adds fma(0, 0, acc) = acc: the sums keep their values
#pragma unroll 8
        for (int dd = 0; dd < kVqSlice; ++dd) {
            const float4 a4 = *reinterpret_cast<const float4*>(&zs[dd][rg * 4]);
            const float4 b4 = *reinterpret_cast<const float4*>(&es[dd][cg * 4]);
            const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                zz[i] = __builtin_fmaf(a[i], a[i], zz[i]);
                ee[i] = __builtin_fmaf(b[i], b[i], ee[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) dot[i][j] = __builtin_fmaf(a[i], b[j], dot[i][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned long long key = ~0ull;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int code = c0 + cg * 4 + j;
            const float t = zz[i] + ee[j];
            const float dist = t - 2.0f * dot[i][j];
            const unsigned long long k = ((unsigned long long)vq_order_bits(dist) << 32) | (uint32_t)code;
            if (code < K && k < key) key = k;
        }
        // the 32 lanes that share this row (one half of the wave)
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const uint32_t hi = __shfl_xor((uint32_t)(key >> 32), o, 64), lo = __shfl_xor((uint32_t)key, o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            if (other < key) key = other;
        }
        const int64_t m = m0 + rg * 4 + i;
        if (cg == 0 && m < M) atomicMin(&best[m], key);
    }
}

__global__ void __launch_bounds__(256)
vq_finalize_kernel(const float* __restrict__ z, const float* __restrict__ E, const unsigned long long* __restrict__ best,
                   int64_t M, int D, VqLayout lay, int64_t* __restrict__ indices, float* __restrict__ out,
                   float* __restrict__ parts) {
    __shared__ float wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + wv;
    float acc = 0.f;
    if (m < M) {
        const int64_t idx = (int64_t)(uint32_t)best[m];
        if (lane == 0) indices[m] = idx;
        for (int d = lane; d < D; d += 64) {
            const int64_t a = lay.at(m, d);
            const float zv = z[a], diff = E[idx * D + d] - zv;
            out[a] = zv + diff;
            acc = __builtin_fmaf(diff, diff, acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) wsum[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

__global__ void __launch_bounds__(256)
vq_loss_kernel(const float* __restrict__ parts, int nparts, double count, float beta, float* __restrict__ loss) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) a += (double)parts[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = (float)(red[0] / count);
        loss[0] = mse * beta + mse;
        loss[1] = mse;
    }
}

// dz = g_out + g_loss * beta * 2 (z - q) / (M D): the straight-through output passes its gradient to the latents
// unchanged, the commitment term pulls them to their codes; the embedding term sees detached latents.
__global__ void __launch_bounds__(256)
vq_bwd_latents_kernel(const float* __restrict__ z, const float* __restrict__ E, const int64_t* __restrict__ indices,
                      const float* __restrict__ g_out, const float* __restrict__ g_loss, int64_t M, int D, VqLayout lay,
                      float scale, float* __restrict__ dz) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + wv;
    if (m >= M) return;
    const float coef = g_loss ? g_loss[0] * scale : 0.f;
    const int64_t idx = indices[m];
    for (int d = lane; d < D; d += 64) {
        const int64_t a = lay.at(m, d);
        const float g = coef * (z[a] - E[idx * D + d]);
        dz[a] = g_out ? g_out[a] + g : g;
    }
}

// dE[k] = g_loss * 2 / (M D) * sum over the rows that chose code k of (e[k] - z[m]), in a fixed order
__global__ void __launch_bounds__(256)
vq_bwd_codebook_kernel(const float* __restrict__ z, const float* __restrict__ E, const int64_t* __restrict__ indices,
                       const float* __restrict__ g_loss, int64_t M, int D, VqLayout lay, float scale,
                       float* __restrict__ dE) {
    __shared__ int rows[256];
    __shared__ int nrows;
    const int k = blockIdx.x, tid = threadIdx.x;
    constexpr int DMAX = 4;                                 // features per thread: D <= 1024
    // a popular code may own most of the rows (all of them while the codebook is still near zero): eight rows are in
    // flight per thread, row r of a chunk adds into accumulator r % 8, and the eight are summed pairwise at the end --
    // a fixed order, so the result is reproducible run to run
    constexpr int RU = 8;
    float acc[RU][DMAX], ek[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) {
#pragma unroll
        for (int u = 0; u < RU; ++u) acc[u][j] = 0.f;
        ek[j] = tid + 256 * j < D ? E[(int64_t)k * D + tid + 256 * j] : 0.f;
    }
    // most codes are chosen by no row (M rows, K codes, M <= K in the shipped configuration): one pass of independent
    // loads by all threads decides that, and such a workgroup is done after writing zeros
    {
        int any = 0;
        for (int64_t m = tid; m < M; m += 256) any |= indices[m] == k ? 1 : 0;
        if (!__syncthreads_or(any)) {
#pragma unroll
            for (int j = 0; j < DMAX; ++j)
                if (tid + 256 * j < D) dE[(int64_t)k * D + tid + 256 * j] = 0.f;
            return;
        }
    }
    for (int64_t m0 = 0; m0 < M; m0 += 256) {
        // rows of this chunk that chose code k, compacted in ascending order by the first wave
        if (tid == 0) nrows = 0;
        __syncthreads();
        if (tid < 64) {
            int base = 0;
            for (int q = 0; q < 4; ++q) {
                const int64_t m = m0 + q * 64 + tid;
                const bool hit = m < M && indices[m] == k;
                const unsigned long long mask = __ballot(hit);
                if (hit) rows[base + __popcll(mask & ((1ull << tid) - 1ull))] = q * 64 + tid;
                base += __popcll(mask);
            }
            if (tid == 0) nrows = base;
        }
        __syncthreads();
        const int n = nrows;
        for (int r = 0; r < n; r += RU) {
            float zv[RU][DMAX];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int64_t m = m0 + rows[r + u < n ? r + u : r];
#pragma unroll
                for (int j = 0; j < DMAX; ++j) zv[u][j] = tid + 256 * j < D ? z[lay.at(m, tid + 256 * j)] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < RU; ++u)
                if (r + u < n) {
#pragma unroll
                    for (int j = 0; j < DMAX; ++j) acc[u][j] += ek[j] - zv[u][j];
                }
        }
        __syncthreads();
    }
    const float coef = g_loss ? g_loss[0] * scale : 0.f;
#pragma unroll
    for (int j = 0; j < DMAX; ++j)
        if (tid + 256 * j < D)
            dE[(int64_t)k * D + tid + 256 * j] =
                coef * (((acc[0][j] + acc[1][j]) + (acc[2][j] + acc[3][j])) + ((acc[4][j] + acc[5][j]) + (acc[6][j] + acc[7][j])));
}

static size_t vq_align(size_t n) { return (n + 255) & ~(size_t)255; }

}  // namespace vamd

using namespace vamd;

extern "C" size_t vamd_vq_workspace(int64_t M) {
    if (M <= 0) return 0;
    return vq_align((size_t)M * 8) + vq_align((size_t)((M + 3) / 4) * 4);
}

extern "C" int vamd_vq_forward(const float* z, const float* codebook, int64_t M, int D, int K, int HW, int nchw, float beta,
                               int64_t* indices, float* out, float* loss, void* workspace, size_t workspace_bytes,
                               void* stream) {
    VAMD_REQUIRE(z && codebook && indices && out && loss, "vq_forward: null argument");
    VAMD_REQUIRE(M > 0 && D > 0 && K > 0 && HW > 0 && M % HW == 0, "vq_forward: bad sizes M=%lld D=%d K=%d HW=%d (M must be images * HW)",
                 (long long)M, D, K, HW);
    VAMD_REQUIRE(M <= 0x7FFFFFFFll / 4, "vq_forward: %lld rows exceed the row-tile grid", (long long)M);
    VAMD_REQUIRE(workspace && workspace_bytes >= vamd_vq_workspace(M), "vq_forward: workspace of %zu bytes, need %zu",
                 workspace_bytes, vamd_vq_workspace(M));
    hipStream_t st = as_stream(stream);
    unsigned long long* best = (unsigned long long*)workspace;
    float* parts = (float*)((char*)workspace + vq_align((size_t)M * 8));
    const VqLayout lay = vq_layout(D, HW, nchw);
    hipError_t e = hipMemsetAsync(best, 0xFF, (size_t)M * 8, st);
    if (e != hipSuccess) return fail((int)e, "vq_forward: hipMemsetAsync failed: %s", hipGetErrorString(e));
    const dim3 grid((unsigned)((M + kVqRows - 1) / kVqRows), (unsigned)((K + kVqCodes - 1) / kVqCodes));
    hipLaunchKernelGGL(vq_dist_kernel, grid, dim3(256), 0, st, z, codebook, M, D, K, lay, best);
    VAMD_LAUNCH_OK("vq_dist");
    const int nparts = (int)((M + 3) / 4);
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(nparts), dim3(256), 0, st, z, codebook, best, M, D, lay, indices, out, parts);
    VAMD_LAUNCH_OK("vq_finalize");
    hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(256), 0, st, parts, nparts, (double)M * (double)D, beta, loss);
    VAMD_LAUNCH_OK("vq_loss");
    return 0;
}

extern "C" int vamd_vq_backward(const float* z, const float* codebook, const int64_t* indices, const float* g_out,
                                const float* g_loss, int64_t M, int D, int K, int HW, int nchw, float beta, float* dz,
                                float* dcodebook, void* stream) {
    VAMD_REQUIRE(z && codebook && indices, "vq_backward: null argument");
    VAMD_REQUIRE(M > 0 && D > 0 && K > 0 && HW > 0 && M % HW == 0, "vq_backward: bad sizes M=%lld D=%d K=%d HW=%d",
                 (long long)M, D, K, HW);
    VAMD_REQUIRE(M <= 0x7FFFFFFFll / 4 && D <= 1024, "vq_backward: M=%lld D=%d out of range (D <= 1024)", (long long)M, D);
    hipStream_t st = as_stream(stream);
    const VqLayout lay = vq_layout(D, HW, nchw);
    const float two_over_n = (float)(2.0 / ((double)M * (double)D));
    if (dz) {
        hipLaunchKernelGGL(vq_bwd_latents_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, z, codebook, indices, g_out,
                           g_loss, M, D, lay, beta * two_over_n, dz);
        VAMD_LAUNCH_OK("vq_bwd_latents");
    }
    if (dcodebook) {
        hipLaunchKernelGGL(vq_bwd_codebook_kernel, dim3(K), dim3(256), 0, st, z, codebook, indices, g_loss, M, D, lay,
                           two_over_n, dcodebook);
        VAMD_LAUNCH_OK("vq_bwd_codebook");
    }
    return 0;
}
