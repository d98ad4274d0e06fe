// BatchNorm / GroupNorm statistics, fused scale-shift-LeakyReLU(-dropout) apply, and their backward.
// All tensors are viewed as [G][R][C] (channel innermost): BatchNorm has G = 1, R = B*H*W;
// GroupNorm and per-image sums have G = B, R = H*W.  HBM-bound: every kernel reads its
// inputs once with 16-byte accesses and reduces in registers -> LDS -> one partial row per block.
#include "common.h"

#include <mutex>

namespace vamd {

constexpr int kStatsBlock = 256;
constexpr int kStatsRowsPerChunk = 256;
constexpr int kStatsMaxChunks = 1024;

// chunks per group: at least kStatsRowsPerChunk rows each, and about 2048 workgroups over all groups
static inline int stats_chunks(int64_t R, int G) {
    int64_t n = (R + kStatsRowsPerChunk - 1) / kStatsRowsPerChunk;
    int64_t cap = 2048 / (G > 0 ? G : 1);
    if (cap < 1) cap = 1;
    if (cap > kStatsMaxChunks) cap = kStatsMaxChunks;
    if (n > cap) n = cap;
    if (n < 1) n = 1;
    return (int)n;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave per channel: sums component 0 and 1 of partial[part][C][2] over `nparts` parts
__device__ __forceinline__ void reduce_parts(const float* __restrict__ partial, int nparts, int C, int c,
                                             double& s1, double& s2) {
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x & 63; k < nparts; k += 64) {
        a += (double)partial[((int64_t)k * C + c) * 2 + 0];
        b += (double)partial[((int64_t)k * C + c) * 2 + 1];
    }
    s1 = wave_sum_d(a);
    s2 = wave_sum_d(b);
}

struct NormBwdArgs {
    const void* dy;
    const void* mask;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* rstd;
    float slope;
    float mask_scale;     // dropout: mask holds 0/1 (or any weights), multiplied by this fp32 factor (1/(1-p))
};

// MODE 0: accumulates (x, x*x).  MODE 1: accumulates (gz, gz*xhat) for the backward pass.
// MODE 2: BatchNorm backward with the rows grouped per image: group g is image g, the coefficients are
// the batch's (one set), and a third sum, xhat, is kept: with it the per-image channel sums of dx
// follow from the partial sums alone (bn_bwd_finalize_img_kernel), without another pass over dx.
// Thread layout: tc = C/VEC lanes across channels, tr = 256/tc lanes across rows.
template <typename T, int VEC, int MODE>
__global__ void __launch_bounds__(kStatsBlock)
chan_stats_kernel(const T* __restrict__ x, NormBwdArgs a, float* __restrict__ partial, int64_t R,
                  int C, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [tr][C][2 or 3]
    const int g = blockIdx.y, chunk = blockIdx.x;
    const int cvec = C / VEC;
    const int tc = cvec;                       // host guarantees cvec <= 256
    const int tr = kStatsBlock / tc;
    const int rl = threadIdx.x / tc, cl = threadIdx.x - rl * tc;
    const int64_t rows_per = (R + nchunk - 1) / nchunk;
    const int64_t r0 = (int64_t)chunk * rows_per;
    int64_t r1 = r0 + rows_per;
    if (r1 > R) r1 = R;
    constexpr int NS = MODE == 2 ? 3 : 2;      // sums per channel
    float s1[VEC], s2[VEC], s3[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = s3[j] = 0.f;
    if (rl < tr) {
        const int c0 = cl * VEC;
        float sc[VEC], sh[VEC], mu[VEC], rs[VEC];
        if (MODE != 0) {
            const int64_t cg = MODE == 2 ? 0 : (int64_t)g * C;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                sc[j] = a.scale[cg + c0 + j];
                sh[j] = a.shift[cg + c0 + j];
                mu[j] = a.mean[cg + c0 + j];
                rs[j] = a.rstd[cg + c0 + j];
            }
        }
        const T* dy = (const T*)a.dy;
        const T* mk = (const T*)a.mask;
        auto load = [&](int64_t r, T (&xv)[VEC], T (&dv)[VEC], T (&mv)[VEC]) {
            const int64_t off = ((int64_t)g * R + r) * C + c0;
            if (VEC > 1) {
                *reinterpret_cast<Vec16<T>*>(xv) = *reinterpret_cast<const Vec16<T>*>(x + off);
                if (MODE != 0) {
                    *reinterpret_cast<Vec16<T>*>(dv) = *reinterpret_cast<const Vec16<T>*>(dy + off);
                    if (mk) *reinterpret_cast<Vec16<T>*>(mv) = *reinterpret_cast<const Vec16<T>*>(mk + off);
                }
            } else {
                xv[0] = x[off];
                if (MODE != 0) {
                    dv[0] = dy[off];
                    if (mk) mv[0] = mk[off];
                }
            }
        };
        auto accumulate = [&](const T (&xv)[VEC], const T (&dv)[VEC], const T (&mv)[VEC]) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xf = to_f(xv[j]);
                if (MODE == 0) {
                    s1[j] += xf;
                    s2[j] = fmaf(xf, xf, s2[j]);
                } else {
                    const float z = fmaf(xf, sc[j], sh[j]);
                    float gz = to_f(dv[j]);
                    if (mk) gz *= to_f(mv[j]) * a.mask_scale;
                    if (!(z > 0.f)) gz *= a.slope;
                    const float xh = (xf - mu[j]) * rs[j];
                    s1[j] += gz;
                    s2[j] = fmaf(gz, xh, s2[j]);
                    if (MODE == 2) s3[j] += xh;
                }
            }
        };
        int64_t r = r0 + rl;
        for (; r + 3 * tr < r1; r += 4 * tr) {                 // four rows' loads in flight per lane
            alignas(16) T x0[VEC], x1[VEC], x2[VEC], x3[VEC];
            alignas(16) T d0[VEC], d1[VEC], d2[VEC], d3[VEC];
            alignas(16) T m0[VEC], m1[VEC], m2[VEC], m3[VEC];
            load(r, x0, d0, m0);
            load(r + tr, x1, d1, m1);
            load(r + 2 * tr, x2, d2, m2);
            load(r + 3 * tr, x3, d3, m3);
            accumulate(x0, d0, m0);
            accumulate(x1, d1, m1);
            accumulate(x2, d2, m2);
            accumulate(x3, d3, m3);
        }
        for (; r < r1; r += tr) {
            alignas(16) T xv[VEC]; alignas(16) T dv[VEC]; alignas(16) T mv[VEC];
            load(r, xv, dv, mv);
            accumulate(xv, dv, mv);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            red[((int64_t)rl * C + c0 + j) * NS + 0] = s1[j];
            red[((int64_t)rl * C + c0 + j) * NS + 1] = s2[j];
            if (MODE == 2) red[((int64_t)rl * C + c0 + j) * NS + 2] = s3[j];
        }
    }
    __syncthreads();
    float* out = partial + ((int64_t)g * nchunk + chunk) * C * NS;
    for (int i = threadIdx.x; i < C * NS; i += kStatsBlock) {
        float acc = 0.f;
        for (int r = 0; r < tr; ++r) acc += red[(int64_t)r * C * NS + i];
        out[i] = acc;
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ partial, int nchunk, int C,
                                   double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean,
                                   float* running_var, int64_t* nbt, float momentum, float eps,
                                   int training, float* scale, float* shift, float* mean,
                                   float* rstd) {
    const int c = blockIdx.x;                      // one wave per channel
    float mu, var;
    if (training) {
        double s1, s2;
        reduce_parts(partial, nchunk, C, c, s1, s2);
        if (threadIdx.x != 0) return;
        const double m = s1 / count;
        double v = s2 / count - m * m;
        if (v < 0.0) v = 0.0;
        mu = (float)m;
        var = (float)v;
        if (running_mean) {
            const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
        if (c == 0 && nbt) nbt[0] += 1;
    } else {
        if (threadIdx.x != 0) return;
        mu = running_mean[c];
        var = running_var[c];
    }
    const float r = 1.0f / sqrtf(var + eps);
    const float sc = gamma[c] * r;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
    mean[c] = mu;
    rstd[c] = r;
}

// thread per (b, group)
__global__ void gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int B, int C,
                                   int groups, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float* scale,
                                   float* shift, float* mean, float* rstd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * groups) return;
    const int b = idx / groups, gr = idx - b * groups;
    const int cpg = C / groups;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunk; ++k)
        for (int j = 0; j < cpg; ++j) {
            const int c = gr * cpg + j;
            s1 += (double)partial[(((int64_t)b * nchunk + k) * C + c) * 2 + 0];
            s2 += (double)partial[(((int64_t)b * nchunk + k) * C + c) * 2 + 1];
        }
    const double m = s1 / count;
    double v = s2 / count - m * m;
    if (v < 0.0) v = 0.0;
    const float mu = (float)m;
    const float r = 1.0f / sqrtf((float)v + eps);
    for (int j = 0; j < cpg; ++j) {
        const int c = gr * cpg + j;
        const float sc = gamma[c] * r;
        scale[(int64_t)b * C + c] = sc;
        shift[(int64_t)b * C + c] = beta[c] - mu * sc;
        mean[(int64_t)b * C + c] = mu;
        rstd[(int64_t)b * C + c] = r;
    }
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nchunk, int C,
                                       double count, const float* __restrict__ scale,
                                       float* dgamma, float* dbeta, float* cA, float* cB,
                                       float* cC, int fixed_stats) {
    const int c = blockIdx.x;                      // one wave per channel
    double s1, s2;
    reduce_parts(partial, nchunk, C, c, s1, s2);
    if (threadIdx.x != 0) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    const double sc = (double)scale[c];
    cA[c] = (float)sc;
    // fixed_stats: mean and rstd are constants (eval-mode running statistics), so dx = scale * gz only
    cB[c] = fixed_stats ? 0.f : (float)(-sc * s1 / count);
    cC[c] = fixed_stats ? 0.f : (float)(-sc * s2 / count);
}

// BatchNorm backward from per-image partial sums [B][nchunk][C][3] = (gz, gz*xhat, xhat): the usual
// coefficients and parameter gradients from the totals, and, because dx = cA*gz + cC*xhat + cB is
// affine in the summed quantities, the per-image channel sums of dx (the gradient of a per-image,
// per-channel additive term of the normalised tensor's producer) and their total.  One workgroup of four
// waves per channel.  An image's chunks are shared by L = min(64, largest power of two <= nchunk) lanes of
// one wave and 256 / L images are in flight at a time; every sum has a fixed order of additions.
__global__ void __launch_bounds__(256)
bn_bwd_finalize_img_kernel(const float* __restrict__ partial, int nchunk, int B, int C,
                           double hw, const float* __restrict__ scale, float* dgamma,
                           float* dbeta, float* cA, float* cB, float* cC,
                           float* __restrict__ dsum, float* __restrict__ dsum_total,
                           int fixed_stats) {
    __shared__ double red[4][3];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // totals over every (image, chunk) pair, pairs dealt to the threads
    double t1 = 0.0, t2 = 0.0, t3 = 0.0;
    const int pairs = B * nchunk;
    for (int i = tid; i < pairs; i += 256) {
        const float* q = partial + ((int64_t)i * C + c) * 3;
        t1 += (double)q[0];
        t2 += (double)q[1];
        t3 += (double)q[2];
    }
    t1 = wave_sum_d(t1);
    t2 = wave_sum_d(t2);
    t3 = wave_sum_d(t3);
    if (lane == 0) { red[wave][0] = t1; red[wave][1] = t2; red[wave][2] = t3; }
    __syncthreads();
    const double s1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    const double s2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const double s3 = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    const double count = hw * B;
    const double sc = (double)scale[c];
    const double a = sc, bq = fixed_stats ? 0.0 : -sc * s1 / count, cq = fixed_stats ? 0.0 : -sc * s2 / count;
    int L = 1;
    while (L < 64 && 2 * L <= nchunk) L *= 2;
    const int per_pass = 256 / L, sub = tid / L, l = tid - sub * L;
    for (int b0 = 0; b0 < B; b0 += per_pass) {
        const int b = b0 + sub;
        double g1 = 0.0, g3 = 0.0;
        if (b < B) {
            for (int k = l; k < nchunk; k += L) {
                const float* q = partial + (((int64_t)b * nchunk + k) * C + c) * 3;
                g1 += (double)q[0];
                g3 += (double)q[2];
            }
        }
        for (int o = L >> 1; o > 0; o >>= 1) {           // the L lanes of an image sit in one wave
            g1 += __shfl_xor(g1, o, 64);
            g3 += __shfl_xor(g3, o, 64);
        }
        if (b < B && l == 0) dsum[(int64_t)b * C + c] = (float)(a * g1 + cq * g3 + hw * bq);
    }
    if (tid != 0) return;
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    cA[c] = (float)a;
    cB[c] = (float)bq;
    cC[c] = (float)cq;
    // the sum over the images of a * g1 + cq * g3 + hw * bq, from the totals
    if (dsum_total) dsum_total[c] = (float)(a * s1 + cq * s3 + count * bq);
}

// thread per (b, group): coefficients
__global__ void gn_bwd_coef_kernel(const float* __restrict__ partial, int nchunk, int B, int C,
                                   int groups, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ rstd, float* cA, float* cB, float* cC) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * groups) return;
    const int b = idx / groups, gr = idx - b * groups;
    const int cpg = C / groups;
    double m1 = 0.0, m2 = 0.0;
    for (int j = 0; j < cpg; ++j) {
        const int c = gr * cpg + j;
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < nchunk; ++k) {
            s1 += (double)partial[(((int64_t)b * nchunk + k) * C + c) * 2 + 0];
            s2 += (double)partial[(((int64_t)b * nchunk + k) * C + c) * 2 + 1];
        }
        m1 += (double)gamma[c] * s1;
        m2 += (double)gamma[c] * s2;
    }
    m1 /= count;
    m2 /= count;
    for (int j = 0; j < cpg; ++j) {
        const int c = gr * cpg + j;
        const double r = (double)rstd[(int64_t)b * C + c];
        cA[(int64_t)b * C + c] = (float)(r * (double)gamma[c]);
        cB[(int64_t)b * C + c] = (float)(-r * m1);
        cC[(int64_t)b * C + c] = (float)(-r * m2);
    }
}

// one block of 4 waves per channel: per_g0[g][c] = sum over chunks of component 0 (wave w takes
// g = w, w+4, ...), total0/1[c] = sums over g too, combined in a fixed order
__global__ void __launch_bounds__(256)
sum_over_g_kernel(const float* __restrict__ partial, int nchunk, int G, int C,
                  float* per_g0, float* total0, float* total1) {
    __shared__ double red[4][2];
    const int c = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double t0 = 0.0, t1 = 0.0;
    for (int g = wave; g < G; g += 4) {
        double s0, s1;
        reduce_parts(partial + (int64_t)g * nchunk * C * 2, nchunk, C, c, s0, s1);
        if (per_g0 && lane == 0) per_g0[(int64_t)g * C + c] = (float)s0;
        t0 += s0;
        t1 += s1;
    }
    if (lane == 0) { red[wave][0] = t0; red[wave][1] = t1; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (total0) total0[c] = (float)((red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
    if (total1) total1[c] = (float)((red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
}

// out[q][w] = sum of in[p][w] over p = q, q + nout, ...; threads run along w (coalesced rows)
__global__ void __launch_bounds__(256)
fold_partials_kernel(const float* __restrict__ in, int nparts, int width, int nout, float* __restrict__ out) {
    const int q = blockIdx.x;
    const int w = blockIdx.y * 256 + threadIdx.x;
    if (w >= width) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int p = q;
    for (; p + 3 * nout < nparts; p += 4 * nout) {
        a0 += in[(int64_t)p * width + w];
        a1 += in[(int64_t)(p + nout) * width + w];
        a2 += in[(int64_t)(p + 2 * nout) * width + w];
        a3 += in[(int64_t)(p + 3 * nout) * width + w];
    }
    for (; p < nparts; p += nout) a0 += in[(int64_t)p * width + w];
    out[(int64_t)q * width + w] = (a0 + a1) + (a2 + a3);
}

// HOIST: one group and a grid stride that is a multiple of C: a lane's channels never change
template <typename T, int VEC, bool HOIST>
__global__ void scale_shift_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                       const float* __restrict__ shift, const T* __restrict__ mask,
                                       float slope, float mask_scale, T* __restrict__ y, int64_t RC, int C,
                                       int64_t nvec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    float sc[VEC], sh[VEC];
    auto coef = [&](int64_t gc) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            sc[j] = scale[gc + j];
            sh[j] = shift[gc + j];
        }
    };
    if (HOIST) coef((i0 * VEC) % C);
    for (int64_t i = i0; i < nvec; i += stride) {
        const int64_t e = i * VEC;
        if (!HOIST) coef((e / RC) * C + (e % C));
        alignas(16) T xv[VEC]; alignas(16) T mv[VEC]; alignas(16) T yv[VEC];
        if (VEC > 1) {
            *reinterpret_cast<Vec16<T>*>(xv) = *reinterpret_cast<const Vec16<T>*>(x + e);
            if (mask) *reinterpret_cast<Vec16<T>*>(mv) = *reinterpret_cast<const Vec16<T>*>(mask + e);
        } else {
            xv[0] = x[e];
            if (mask) mv[0] = mask[e];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float z = leaky(fmaf(to_f(xv[j]), sc[j], sh[j]), slope);
            if (mask) z *= to_f(mv[j]) * mask_scale;
            yv[j] = from_f<T>(z);
        }
        if (VEC > 1) *reinterpret_cast<Vec16<T>*>(y + e) = *reinterpret_cast<Vec16<T>*>(yv);
        else y[e] = yv[0];
    }
}

struct NormApplyArgs {
    const void* dy; const void* x; const void* mask;
    const float* scale; const float* shift; const float* mean; const float* rstd;
    const float* cA; const float* cB; const float* cC;
    const void* add; const void* add2;
    float slope;
    float mask_scale;
};

// HOIST: one group (BatchNorm) and a grid stride that is a multiple of C, so a lane keeps its VEC
// channels for the whole loop and the seven per-channel coefficients are read once.
template <typename T, int VEC, bool HOIST>
__global__ void norm_bwd_apply_kernel(NormApplyArgs a, T* __restrict__ dx, int64_t RC, int C,
                                      int64_t nvec) {
    const T* dy = (const T*)a.dy; const T* x = (const T*)a.x; const T* mask = (const T*)a.mask;
    const T* add = (const T*)a.add; const T* add2 = (const T*)a.add2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    float sc[VEC], sh[VEC], mu[VEC], rs[VEC], cA[VEC], cB[VEC], cC[VEC];
    auto coef = [&](int64_t gc) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            sc[j] = a.scale[gc + j];
            sh[j] = a.shift[gc + j];
            mu[j] = a.mean[gc + j];
            rs[j] = a.rstd[gc + j];
            cA[j] = a.cA[gc + j];
            cB[j] = a.cB[gc + j];
            cC[j] = a.cC[gc + j];
        }
    };
    if (HOIST) coef((i0 * VEC) % C);
    for (int64_t i = i0; i < nvec; i += stride) {
        const int64_t e = i * VEC;
        if (!HOIST) coef((e / RC) * C + (e % C));
        alignas(16) T xv[VEC]; alignas(16) T dv[VEC]; alignas(16) T mv[VEC]; alignas(16) T av[VEC]; alignas(16) T a2v[VEC]; alignas(16) T ov[VEC];
        if (VEC > 1) {
            *reinterpret_cast<Vec16<T>*>(xv) = *reinterpret_cast<const Vec16<T>*>(x + e);
            *reinterpret_cast<Vec16<T>*>(dv) = *reinterpret_cast<const Vec16<T>*>(dy + e);
            if (mask) *reinterpret_cast<Vec16<T>*>(mv) = *reinterpret_cast<const Vec16<T>*>(mask + e);
            if (add) *reinterpret_cast<Vec16<T>*>(av) = *reinterpret_cast<const Vec16<T>*>(add + e);
            if (add2) *reinterpret_cast<Vec16<T>*>(a2v) = *reinterpret_cast<const Vec16<T>*>(add2 + e);
        } else {
            xv[0] = x[e]; dv[0] = dy[e];
            if (mask) mv[0] = mask[e];
            if (add) av[0] = add[e];
            if (add2) a2v[0] = add2[e];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xf = to_f(xv[j]);
            const float z = fmaf(xf, sc[j], sh[j]);
            float gz = to_f(dv[j]);
            if (mask) gz *= to_f(mv[j]) * a.mask_scale;
            if (!(z > 0.f)) gz *= a.slope;
            const float xh = (xf - mu[j]) * rs[j];
            float r = fmaf(cA[j], gz, fmaf(cC[j], xh, cB[j]));
            if (add) r += to_f(av[j]);
            if (add2) r += to_f(a2v[j]);
            ov[j] = from_f<T>(r);
        }
        if (VEC > 1) *reinterpret_cast<Vec16<T>*>(dx + e) = *reinterpret_cast<Vec16<T>*>(ov);
        else dx[e] = ov[0];
    }
}

template <typename T, int MODE>
static int launch_stats(const char* name, const T* x, NormBwdArgs a, float* partial, int G,
                        int64_t R, int C, hipStream_t st) {
    constexpr int VEC = Vec16<T>::N;
    const int nchunk = stats_chunks(R, G);
    dim3 grid(nchunk, G);
    const bool vec = (C % VEC == 0) && (C / VEC <= kStatsBlock);
    if (vec) {
        const int tr = kStatsBlock / (C / VEC);
        const size_t lds = (size_t)tr * C * (MODE == 2 ? 3 : 2) * sizeof(float);
        hipLaunchKernelGGL((chan_stats_kernel<T, VEC, MODE>), grid, dim3(kStatsBlock), lds, st, x, a,
                           partial, R, C, nchunk);
    } else {
        if (C > kStatsBlock) return fail(VAMD_EUNSUPPORTED, "%s: C=%d not a multiple of %d and > 256", name, C, VEC);
        const int tr = kStatsBlock / C;
        const size_t lds = (size_t)tr * C * (MODE == 2 ? 3 : 2) * sizeof(float);
        hipLaunchKernelGGL((chan_stats_kernel<T, 1, MODE>), grid, dim3(kStatsBlock), lds, st, x, a,
                           partial, R, C, nchunk);
    }
    return 0;
}

// ---- allocation-free GroupNorm statistics ------------------------------------------------------
// Partial sums of vamd_gn_coef_noalloc live in this array instead of a caller-supplied workspace.
// Every GPU the library runs on holds its own copy at its own address until the process ends.
constexpr size_t kGnScratchFloats = VAMD_GN_NOALLOC_SCRATCH_BYTES / sizeof(float);
static __device__ __align__(256) float g_gn_scratch[kGnScratchFloats];

constexpr int kMaxDevices = 64;
static float* g_gn_scratch_ptr[kMaxDevices];          // device address of g_gn_scratch, per device ordinal
static std::mutex g_gn_scratch_mu;

// The array's name is only a host-side handle: the device address comes from hipGetSymbolAddress, once
// per device, with that device current.
static int gn_scratch_for_current_device(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "vamd_gn_coef_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(VAMD_EUNSUPPORTED, "vamd_gn_coef_noalloc: device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_gn_scratch_mu);
    if (!g_gn_scratch_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_gn_scratch));
        if (e != hipSuccess || !p)
            return fail(e != hipSuccess ? (int)e : VAMD_EUNSUPPORTED, "vamd_gn_coef_noalloc: hipGetSymbolAddress: %s",
                        hipGetErrorString(e));
        g_gn_scratch_ptr[dev] = reinterpret_cast<float*>(p);
    }
    *out = g_gn_scratch_ptr[dev];
    return 0;
}

}  // namespace vamd

using namespace vamd;

extern "C" {

int vamd_stats_chunks(int64_t R, int G) { return stats_chunks(R, G); }

int vamd_chan_stats(const void* x, float* partial, int G, int64_t R, int C, int dtype, void* stream) {
    VAMD_REQUIRE(x && partial && G > 0 && G <= 65535 && R > 0 && C > 0, "vamd_chan_stats: bad argument");
    VAMD_REQUIRE((uintptr_t)x % 16 == 0, "vamd_chan_stats: x must be 16-byte aligned");
    NormBwdArgs a{};
    VAMD_DISPATCH(dtype, "vamd_chan_stats", {
        int rc = launch_stats<T, 0>("vamd_chan_stats", (const T*)x, a, partial, G, R, C, as_stream(stream));
        if (rc) return rc;
    });
    VAMD_LAUNCH_OK("vamd_chan_stats");
    return 0;
}

int vamd_norm_bwd_stats(const void* dy, const void* x, const void* mask, const float* scale,
                        const float* shift, const float* mean, const float* rstd, float slope,
                        float mask_scale, float* partial, int G, int64_t R, int C, int dtype, void* stream) {
    VAMD_REQUIRE(dy && x && scale && shift && mean && rstd && partial, "vamd_norm_bwd_stats: null pointer");
    VAMD_REQUIRE(G > 0 && G <= 65535 && R > 0 && C > 0, "vamd_norm_bwd_stats: bad shape");
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)mask % 16 == 0,
                 "vamd_norm_bwd_stats: tensors must be 16-byte aligned");
    NormBwdArgs a{dy, mask, scale, shift, mean, rstd, slope, mask_scale};
    VAMD_DISPATCH(dtype, "vamd_norm_bwd_stats", {
        int rc = launch_stats<T, 1>("vamd_norm_bwd_stats", (const T*)x, a, partial, G, R, C, as_stream(stream));
        if (rc) return rc;
    });
    VAMD_LAUNCH_OK("vamd_norm_bwd_stats");
    return 0;
}

int vamd_bn_bwd_stats_img(const void* dy, const void* x, const void* mask, const float* scale,
                          const float* shift, const float* mean, const float* rstd, float slope,
                          float mask_scale, float* partial, int B, int64_t HW, int C, int dtype, void* stream) {
    VAMD_REQUIRE(dy && x && scale && shift && mean && rstd && partial, "vamd_bn_bwd_stats_img: null pointer");
    VAMD_REQUIRE(B > 0 && B <= 65535 && HW > 0 && C > 0, "vamd_bn_bwd_stats_img: bad shape");
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)mask % 16 == 0,
                 "vamd_bn_bwd_stats_img: tensors must be 16-byte aligned");
    NormBwdArgs a{dy, mask, scale, shift, mean, rstd, slope, mask_scale};
    VAMD_DISPATCH(dtype, "vamd_bn_bwd_stats_img", {
        int rc = launch_stats<T, 2>("vamd_bn_bwd_stats_img", (const T*)x, a, partial, B, HW, C, as_stream(stream));
        if (rc) return rc;
    });
    VAMD_LAUNCH_OK("vamd_bn_bwd_stats_img");
    return 0;
}

int vamd_bn_bwd_finalize_img(const float* partial, int nchunk, int B, int C, int64_t HW, const float* scale,
                             float* dgamma, float* dbeta, float* cA, float* cB, float* cC, float* dsum,
                             float* dsum_total, int fixed_stats, void* stream) {
    VAMD_REQUIRE(partial && scale && dgamma && dbeta && cA && cB && cC && dsum && nchunk > 0 && B > 0 && C > 0 && HW > 0,
                 "vamd_bn_bwd_finalize_img: bad argument");
    hipLaunchKernelGGL(bn_bwd_finalize_img_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial, nchunk, B, C,
                       (double)HW, scale, dgamma, dbeta, cA, cB, cC, dsum, dsum_total, fixed_stats);
    VAMD_LAUNCH_OK("vamd_bn_bwd_finalize_img");
    return 0;
}

int vamd_bn_finalize(const float* partial, int nchunk, int C, int64_t count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float momentum, float eps, int training,
                     float* scale, float* shift, float* mean, float* rstd, void* stream) {
    VAMD_REQUIRE(gamma && beta && scale && shift && mean && rstd && C > 0, "vamd_bn_finalize: null pointer");
    VAMD_REQUIRE(training ? (partial && nchunk > 0 && count > 0) : (running_mean && running_var),
                 "vamd_bn_finalize: training needs partial sums, eval needs running statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(64), 0, as_stream(stream),
                       partial, nchunk, C, (double)count, gamma, beta, running_mean, running_var,
                       num_batches_tracked, momentum, eps, training, scale, shift, mean, rstd);
    VAMD_LAUNCH_OK("vamd_bn_finalize");
    return 0;
}

int vamd_gn_finalize(const float* partial, int nchunk, int B, int C, int groups, int64_t HW,
                     const float* gamma, const float* beta, float eps, float* scale, float* shift,
                     float* mean, float* rstd, void* stream) {
    VAMD_REQUIRE(partial && gamma && beta && scale && shift && mean && rstd, "vamd_gn_finalize: null pointer");
    VAMD_REQUIRE(B > 0 && C > 0 && groups > 0 && C % groups == 0 && HW > 0 && nchunk > 0,
                 "vamd_gn_finalize: bad shape (C=%d groups=%d)", C, groups);
    const int total = B * groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 63) / 64), dim3(64), 0, as_stream(stream),
                       partial, nchunk, B, C, groups, (double)HW * (C / groups), gamma, beta, eps,
                       scale, shift, mean, rstd);
    VAMD_LAUNCH_OK("vamd_gn_finalize");
    return 0;
}

int vamd_gn_coef_noalloc(const void* x, int B, int C, int groups, int64_t HW, const float* gamma,
                         const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd,
                         int dtype, void* stream) {
    VAMD_REQUIRE(x && gamma && beta && scale && shift && mean && rstd, "vamd_gn_coef_noalloc: null pointer");
    VAMD_REQUIRE(B > 0 && B <= 65535 && C > 0 && groups > 0 && C % groups == 0 && HW > 0,
                 "vamd_gn_coef_noalloc: bad shape (B=%d C=%d groups=%d)", B, C, groups);
    VAMD_REQUIRE((uintptr_t)x % 16 == 0, "vamd_gn_coef_noalloc: x must be 16-byte aligned");
    const int nchunk = stats_chunks(HW, B);
    const size_t need = (size_t)B * nchunk * C * 2;
    if (need > kGnScratchFloats)
        return fail(VAMD_EINVAL, "vamd_gn_coef_noalloc: B=%d HW=%lld C=%d needs %zu bytes of partial sums, the built-in "
                    "buffer holds %zu (VAMD_GN_NOALLOC_SCRATCH_BYTES); use vamd_chan_stats + vamd_gn_finalize with a "
                    "workspace of your own", B, (long long)HW, C, need * sizeof(float), (size_t)VAMD_GN_NOALLOC_SCRATCH_BYTES);
    float* partial = nullptr;
    int rc = gn_scratch_for_current_device(&partial);
    if (rc) return rc;
    // every partial row that the finalise kernel reads is written by the statistics kernel of the same call,
    // so nothing has to be cleared between calls
    NormBwdArgs a{};
    VAMD_DISPATCH(dtype, "vamd_gn_coef_noalloc", {
        rc = launch_stats<T, 0>("vamd_gn_coef_noalloc", (const T*)x, a, partial, B, HW, C, as_stream(stream));
        if (rc) return rc;
    });
    const int total = B * groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 63) / 64), dim3(64), 0, as_stream(stream),
                       partial, nchunk, B, C, groups, (double)HW * (C / groups), gamma, beta, eps,
                       scale, shift, mean, rstd);
    VAMD_LAUNCH_OK("vamd_gn_coef_noalloc");
    return 0;
}

int vamd_scale_shift_act(const void* x, const float* scale, const float* shift, const void* mask,
                         float slope, float mask_scale, void* y, int G, int64_t R, int C, int dtype, void* stream) {
    VAMD_REQUIRE(x && scale && shift && y && G > 0 && R > 0 && C > 0, "vamd_scale_shift_act: bad argument");
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)mask % 16 == 0,
                 "vamd_scale_shift_act: tensors must be 16-byte aligned");
    const int64_t n = (int64_t)G * R * C;
    VAMD_DISPATCH(dtype, "vamd_scale_shift_act", {
        constexpr int VEC = Vec16<T>::N;
        if (C % VEC == 0) {
            if (G == 1 && (256 * VEC) % C == 0)
                hipLaunchKernelGGL((scale_shift_act_kernel<T, VEC, true>), dim3(grid_for(n / VEC, 256)), dim3(256), 0,
                                   as_stream(stream), (const T*)x, scale, shift, (const T*)mask, slope, mask_scale,
                                   (T*)y, R * C, C, n / VEC);
            else
                hipLaunchKernelGGL((scale_shift_act_kernel<T, VEC, false>), dim3(grid_for(n / VEC, 256)), dim3(256), 0,
                                   as_stream(stream), (const T*)x, scale, shift, (const T*)mask, slope, mask_scale,
                                   (T*)y, R * C, C, n / VEC);
        } else {
            hipLaunchKernelGGL((scale_shift_act_kernel<T, 1, false>), dim3(grid_for(n, 256)), dim3(256), 0,
                               as_stream(stream), (const T*)x, scale, shift, (const T*)mask, slope, mask_scale,
                               (T*)y, R * C, C, n);
        }
    });
    VAMD_LAUNCH_OK("vamd_scale_shift_act");
    return 0;
}

int vamd_bn_bwd_finalize(const float* partial, int nchunk, int C, int64_t count, const float* scale,
                         float* dgamma, float* dbeta, float* cA, float* cB, float* cC, int fixed_stats,
                         void* stream) {
    VAMD_REQUIRE(partial && scale && dgamma && dbeta && cA && cB && cC && nchunk > 0 && C > 0 && count > 0,
                 "vamd_bn_bwd_finalize: bad argument");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, as_stream(stream),
                       partial, nchunk, C, (double)count, scale, dgamma, dbeta, cA, cB, cC, fixed_stats);
    VAMD_LAUNCH_OK("vamd_bn_bwd_finalize");
    return 0;
}

int vamd_gn_bwd_finalize(const float* partial, int nchunk, int B, int C, int groups, int64_t HW,
                         const float* gamma, const float* rstd, float* dgamma, float* dbeta,
                         float* cA, float* cB, float* cC, void* stream) {
    VAMD_REQUIRE(partial && gamma && rstd && dgamma && dbeta && cA && cB && cC, "vamd_gn_bwd_finalize: null pointer");
    VAMD_REQUIRE(B > 0 && C > 0 && groups > 0 && C % groups == 0 && HW > 0 && nchunk > 0,
                 "vamd_gn_bwd_finalize: bad shape");
    const int total = B * groups;
    hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3((total + 63) / 64), dim3(64), 0, as_stream(stream),
                       partial, nchunk, B, C, groups, (double)HW * (C / groups), gamma, rstd, cA, cB, cC);
    hipLaunchKernelGGL(sum_over_g_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial,
                       nchunk, B, C, (float*)nullptr, dbeta, dgamma);
    VAMD_LAUNCH_OK("vamd_gn_bwd_finalize");
    return 0;
}

int vamd_norm_bwd_apply(const void* dy, const void* x, const void* mask, const float* scale,
                        const float* shift, const float* mean, const float* rstd, float slope,
                        float mask_scale, const float* cA, const float* cB, const float* cC, const void* add,
                        const void* add2, void* dx, int G, int64_t R, int C, int dtype,
                        void* stream) {
    VAMD_REQUIRE(dy && x && scale && shift && mean && rstd && cA && cB && cC && dx,
                 "vamd_norm_bwd_apply: null pointer");
    VAMD_REQUIRE(G > 0 && R > 0 && C > 0, "vamd_norm_bwd_apply: bad shape");
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0 &&
                     (uintptr_t)mask % 16 == 0 && (uintptr_t)add % 16 == 0 && (uintptr_t)add2 % 16 == 0,
                 "vamd_norm_bwd_apply: tensors must be 16-byte aligned");
    NormApplyArgs a{dy, x, mask, scale, shift, mean, rstd, cA, cB, cC, add, add2, slope, mask_scale};
    const int64_t n = (int64_t)G * R * C;
    VAMD_DISPATCH(dtype, "vamd_norm_bwd_apply", {
        constexpr int VEC = Vec16<T>::N;
        if (C % VEC == 0) {
            // one group and whole channel rows per block: a lane keeps its channels across the grid stride
            if (G == 1 && (256 * VEC) % C == 0)
                hipLaunchKernelGGL((norm_bwd_apply_kernel<T, VEC, true>), dim3(grid_for(n / VEC, 256)), dim3(256), 0,
                                   as_stream(stream), a, (T*)dx, R * C, C, n / VEC);
            else
                hipLaunchKernelGGL((norm_bwd_apply_kernel<T, VEC, false>), dim3(grid_for(n / VEC, 256)), dim3(256), 0,
                                   as_stream(stream), a, (T*)dx, R * C, C, n / VEC);
        } else {
            hipLaunchKernelGGL((norm_bwd_apply_kernel<T, 1, false>), dim3(grid_for(n, 256)), dim3(256), 0,
                               as_stream(stream), a, (T*)dx, R * C, C, n);
        }
    });
    VAMD_LAUNCH_OK("vamd_norm_bwd_apply");
    return 0;
}

int vamd_fold_partials(const float* in, int nparts, int width, int nout, float* out, void* stream) {
    VAMD_REQUIRE(in && out && nparts > 0 && width > 0 && nout > 0 && nout <= nparts, "vamd_fold_partials: bad argument");
    hipLaunchKernelGGL(fold_partials_kernel, dim3(nout, (width + 255) / 256), dim3(256), 0, as_stream(stream), in,
                       nparts, width, nout, out);
    VAMD_LAUNCH_OK("vamd_fold_partials");
    return 0;
}

int vamd_sum_finalize(const float* partial, int nchunk, int G, int C, float* per_g, float* total,
                      void* stream) {
    VAMD_REQUIRE(partial && nchunk > 0 && G > 0 && C > 0 && (per_g || total), "vamd_sum_finalize: bad argument");
    hipLaunchKernelGGL(sum_over_g_kernel, dim3(C), dim3(256), 0, as_stream(stream), partial,
                       nchunk, G, C, per_g, total, (float*)nullptr);
    VAMD_LAUNCH_OK("vamd_sum_finalize");
    return 0;
}

}  // extern "C"
