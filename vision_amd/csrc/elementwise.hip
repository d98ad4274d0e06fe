// HBM-bound elementwise / small kernels: diffusion schedule + q_sample, L1 loss, sampler step,
// layout converters, transpose, row softmax, time-embedding path, AdamW, weight repacking.
#include "common.h"

#include <mutex>
#include <vector>

namespace vamd {

static thread_local char g_err[512] = "";

// ---- launch timing ----------------------------------------------------------------------------
struct TimedSlot { hipEvent_t e0, e1; int kind; double flops, executed; };
static std::mutex g_timed_mu;
static std::vector<TimedSlot> g_timed;       // events are created once and reused
static size_t g_timed_used = 0;
static bool g_timed_on = false;

int timed_launch_begin(int kind, double flops, double executed, hipStream_t st) {
    if (!g_timed_on) return -1;
    std::lock_guard<std::mutex> lk(g_timed_mu);
    if (g_timed_used == g_timed.size()) {
        TimedSlot s{};
        if (hipEventCreate(&s.e0) != hipSuccess || hipEventCreate(&s.e1) != hipSuccess) return -1;
        g_timed.push_back(s);
    }
    TimedSlot& s = g_timed[g_timed_used];
    s.kind = kind;
    s.flops = flops;
    s.executed = executed;
    if (hipEventRecord(s.e0, st) != hipSuccess) return -1;
    return (int)g_timed_used++;
}

void timed_launch_end(int slot, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    (void)hipEventRecord(g_timed[slot].e1, st);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// acos(0.95f), acos(0.02f) evaluated in float like torch.acos on a float32 scalar tensor.
__device__ __forceinline__ void schedule_of(float t, float& n, float& s) {
    const float start = acosf(0.95f);
    const float end = acosf(0.02f);
    const float ang = start + t * (end - start);
    s = cosf(ang);
    n = sinf(ang);
}

__global__ void schedule_kernel(const float* __restrict__ t, float* n, float* s, float* n2, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float nn, ss;
    schedule_of(t[b], nn, ss);
    n[b] = nn;
    s[b] = ss;
    n2[b] = nn * nn;
}

// one thread per pixel: NCHW fp32 reads are coalesced per channel plane, NHWC writes are
// C contiguous elements per thread.
template <typename T>
__global__ void qsample_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                               const float* __restrict__ t, T* __restrict__ xt, float* n_out,
                               float* s_out, float* n2_out, int B, int C, int64_t HW, int CW) {
    // a work item = CW channels of one pixel: all of them for thin inputs (images), one for many-channel inputs
    // (latents), where consecutive lanes then walk the channels of the channels-last tensor
    const int CS = (C + CW - 1) / CW;
    const int64_t total = (int64_t)B * HW * CS;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = w / CS;
        const int c0 = (int)(w - p * CS) * CW, c1 = min(C, c0 + CW);
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        float n, s;
        schedule_of(t[b], n, s);
        if (hw == 0 && c0 == 0) {
            n_out[b] = n;
            s_out[b] = s;
            n2_out[b] = n * n;
        }
        for (int c = c0; c < c1; ++c) {
            const int64_t src = ((int64_t)b * C + c) * HW + hw;
            const float v = s * x0[src] + n * eps[src];
            xt[p * C + c] = from_f<T>(v);
        }
    }
}

template <typename T>
__global__ void l1_fwd_kernel(const T* __restrict__ pred, const float* __restrict__ eps,
                              float* __restrict__ partial, int B, int C, int64_t HW, int CW) {
    const int CS = (C + CW - 1) / CW;                 // work items of CW channels, as in qsample_kernel
    const int64_t total = (int64_t)B * HW * CS;
    float acc = 0.f;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = w / CS;
        const int c0 = (int)(w - p * CS) * CW, c1 = min(C, c0 + CW);
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        for (int c = c0; c < c1; ++c)
            acc += fabsf(to_f(pred[p * C + c]) - eps[((int64_t)b * C + c) * HW + hw]);
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void l1_finalize_kernel(const float* __restrict__ partial, int nchunk, double inv_numel,
                                   float* loss) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nchunk; i += 64) acc += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) loss[0] = (float)(acc * inv_numel);
}

template <typename T>
__global__ void l1_bwd_kernel(const T* __restrict__ pred, const float* __restrict__ eps,
                              const float* __restrict__ gout, T* __restrict__ dpred, int B, int C,
                              int64_t HW, float inv_numel, int CW) {
    const int CS = (C + CW - 1) / CW;
    const int64_t total = (int64_t)B * HW * CS;
    const float g = gout[0] * inv_numel;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = w / CS;
        const int c0 = (int)(w - p * CS) * CW, c1 = min(C, c0 + CW);
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        for (int c = c0; c < c1; ++c) {
            const float d = to_f(pred[p * C + c]) - eps[((int64_t)b * C + c) * HW + hw];
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);   // torch sign(): sign(0) = 0
            dpred[p * C + c] = from_f<T>(g * sg);
        }
    }
}

// Head and reconstruction loss of the VQ-VAE-UNet (reference models/vqvaeunet.py:206-210 nn.Sigmoid() after the last
// channel-wise Linear, :236-240 F.mse_loss(x_rec, x)): h NHWC in T -> x_rec = sigmoid(h) as NCHW fp32 (what the model
// returns) and the partial sums of (x_rec - x)^2; the backward forms g * 2 (x_rec - x) / numel * x_rec (1 - x_rec).
template <typename T>
__global__ void sigmoid_mse_fwd_kernel(const T* __restrict__ h, const float* __restrict__ target, float* __restrict__ xrec,
                                       float* __restrict__ partial, int B, int C, int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    float acc = 0.f;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        for (int c = 0; c < C; ++c) {
            const float y = 1.0f / (1.0f + expf(-to_f(h[p * C + c])));
            const int64_t o = ((int64_t)b * C + c) * HW + hw;
            xrec[o] = y;
            if (target) {
                const float d = y - target[o];
                acc = fmaf(d, d, acc);
            }
        }
    }
    if (!partial) return;
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
__global__ void sigmoid_mse_bwd_kernel(const float* __restrict__ xrec, const float* __restrict__ target,
                                       const float* __restrict__ g_loss, const float* __restrict__ g_xrec, T* __restrict__ dh,
                                       int B, int C, int64_t HW, float two_over_numel) {
    const int64_t total = (int64_t)B * HW;
    const float g = g_loss ? g_loss[0] * two_over_numel : 0.f;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        for (int c = 0; c < C; ++c) {
            const int64_t o = ((int64_t)b * C + c) * HW + hw;
            const float y = xrec[o];
            float dy = target ? g * (y - target[o]) : 0.f;
            if (g_xrec) dy += g_xrec[o];
            dh[p * C + c] = from_f<T>(dy * y * (1.0f - y));
        }
    }
}

template <typename T>
__global__ void denoise_step_kernel(float* __restrict__ cur, const T* __restrict__ eps_hat,
                                    const float* __restrict__ n, const float* __restrict__ s,
                                    const float* __restrict__ n_next,
                                    const float* __restrict__ s_next, T* __restrict__ next_in,
                                    float* __restrict__ pred_images, int B, int C, int64_t HW, int CW) {
    const int CS = (C + CW - 1) / CW;
    const int64_t total = (int64_t)B * HW * CS;
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = w / CS;
        const int c0 = (int)(w - p * CS) * CW, c1 = min(C, c0 + CW);
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        const float nb = n[b], sb = s[b], nn = n_next[b], sn = s_next[b];
        for (int c = c0; c < c1; ++c) {
            const float e = to_f(eps_hat[p * C + c]);
            const float x0h = (cur[p * C + c] - nb * e) / sb;
            const float nx = sn * x0h + nn * e;
            cur[p * C + c] = nx;
            next_in[p * C + c] = from_f<T>(nx);
            if (pred_images) pred_images[((int64_t)b * C + c) * HW + hw] = x0h;
        }
    }
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y,
                                    float* __restrict__ y32, int B, int C, int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        for (int c = 0; c < C; ++c) {
            const float v = x[((int64_t)b * C + c) * HW + hw];
            y[p * C + c] = from_f<T>(v);
            if (y32) y32[p * C + c] = v;
        }
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C,
                                    int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total;
         p += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(p / HW);
        const int64_t hw = p - (int64_t)b * HW;
        for (int c = 0; c < C; ++c) y[((int64_t)b * C + c) * HW + hw] = to_f(x[p * C + c]);
    }
}

// out = a (+ b), every operand in its own dtype, the sum in fp32: the dtype changes and two-term merges around the
// codebook lookup of the first stage (fp32 latents between compute-dtype halves of the network).
template <typename TA, typename TB, typename TO>
__global__ void add_cast_kernel(const TA* __restrict__ a, const TB* __restrict__ b, TO* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = to_f(a[i]);
        if (b) v += to_f(b[i]);
        out[i] = from_f<TO>(v);
    }
}

template <typename TA, typename TB>
static void add_cast_out(const void* a, const void* b, void* out, int out_dtype, int64_t n, hipStream_t st) {
    const dim3 grid(grid_for(n, 256));
    if (out_dtype == VAMD_BF16)
        hipLaunchKernelGGL((add_cast_kernel<TA, TB, bf16_t>), grid, dim3(256), 0, st, (const TA*)a, (const TB*)b, (bf16_t*)out, n);
    else
        hipLaunchKernelGGL((add_cast_kernel<TA, TB, float>), grid, dim3(256), 0, st, (const TA*)a, (const TB*)b, (float*)out, n);
}

// Layout + dtype conversion of many-channel tensors: x [batch][R][Cc] -> y [batch][Cc][R] (and an optional fp32 copy in
// y's layout), 32 x 32 tiles through LDS (+1 pad) so that both sides move whole rows of a tile.
template <typename TIn, typename TOut>
__global__ void convert_transpose_kernel(const TIn* __restrict__ x, TOut* __restrict__ y, float* __restrict__ y32, int R,
                                         int Cc) {
    __shared__ float tile[32][33];
    const int64_t base = (int64_t)blockIdx.z * R * Cc;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        if (r < R && c < Cc) tile[i][tx] = to_f(x[base + (int64_t)r * Cc + c]);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < Cc) {
            const float v = tile[tx][i];
            y[base + (int64_t)c * R + r] = from_f<TOut>(v);
            if (y32) y32[base + (int64_t)c * R + r] = v;
        }
    }
}

// [batch][R][C] (rows ldx elements apart) -> [batch][C][R] (rows ldy apart), 32x32 tiles through LDS (+1 pad).
template <typename T>
__global__ void transpose_kernel(const T* __restrict__ x, T* __restrict__ y, int R, int C, int ldx, int ldy) {
    __shared__ float tile[32][33];
    const int64_t base = (int64_t)blockIdx.z * R * ldx;
    const int64_t obase = (int64_t)blockIdx.z * C * ldy;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        if (r < R && c < C) tile[i][tx] = to_f(x[base + (int64_t)r * ldx + c]);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (r < R && c < C) y[obase + (int64_t)c * ldy + r] = from_f<T>(tile[tx][i]);
    }
}

// one wave per row; x fp32 -> p in T
template <typename T>
__global__ void softmax_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int64_t rows,
                                    int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < rows; r += nwave) {
        const float* xr = x + r * ld;
        float m = -INFINITY;
        for (int c = lane; c < cols; c += 64) m = fmaxf(m, xr[c]);
        m = wave_max(m);
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) sum += expf(xr[c] - m);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int c = lane; c < cols; c += 64) y[r * ld + c] = from_f<T>(expf(xr[c] - m) * inv);
    }
}

template <typename T>
__global__ void softmax_rows_bwd_kernel(const T* __restrict__ p, const float* __restrict__ dp,
                                        T* __restrict__ ds, int64_t rows, int cols, int ld, float alpha) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave; r < rows; r += nwave) {
        const T* pr = p + r * ld;
        const float* dr = dp + r * ld;
        float dot = 0.f;
        for (int c = lane; c < cols; c += 64) dot += to_f(pr[c]) * dr[c];
        dot = wave_sum(dot);
        for (int c = lane; c < cols; c += 64)
            ds[r * ld + c] = from_f<T>(alpha * to_f(pr[c]) * (dr[c] - dot));
    }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y,
                           int64_t nvec, int64_t n) {
    using V = Vec16<T>;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += stride) {
        V va = reinterpret_cast<const V*>(a)[i], vb = reinterpret_cast<const V*>(b)[i], vo;
#pragma unroll
        for (int j = 0; j < V::N; ++j) vo.v[j] = from_f<T>(to_f(va.v[j]) + to_f(vb.v[j]));
        reinterpret_cast<V*>(y)[i] = vo;
    }
    for (int64_t i = nvec * V::N + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = from_f<T>(to_f(a[i]) + to_f(b[i]));
}

__global__ void timeembed_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                 float* __restrict__ emb, int B, int dim) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float a = w[j] * t[b];
    emb[b * dim + j] = sinf(a);
    emb[b * dim + half + j] = cosf(a);
}

// wave per (b, o): lanes split the input features, coalesced reads of both rows
__global__ void __launch_bounds__(256)
linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                  const float* __restrict__ bias, float* __restrict__ y, int B,
                  int I, int O, int act_in, float slope) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= B * O) return;                                 // wave-uniform
    const int b = idx / O, o = idx - b * O;
    const float* xr = x + (int64_t)b * I;
    const float* wr = W + (int64_t)o * I;
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) {
        float v = xr[i];
        if (act_in) v = leaky(v, slope);
        acc = fmaf(v, wr[i], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) y[idx] = acc + (bias ? bias[o] : 0.f);
}

// wave per (b, output of some item): as linear_fwd_kernel, the item found by a scan of the (few) entries
__global__ void __launch_bounds__(256)
linear_fwd_multi_kernel(const float* __restrict__ x, const vamd_linear_item* __restrict__ items, int n,
                        int total_O, int B, int I, int act_in, float slope) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= B * total_O) return;                           // wave-uniform
    const int b = idx / total_O, og = idx - b * total_O;
    int k = 0;
    while (k + 1 < n && items[k + 1].first <= og) ++k;
    const vamd_linear_item it = items[k];
    const int o = og - it.first;
    const float* xr = x + (int64_t)b * I;
    const float* wr = it.W + (int64_t)o * I;
    float acc = 0.f;
    for (int i = lane; i < I; i += 64) {
        float v = xr[i];
        if (act_in) v = leaky(v, slope);
        acc = fmaf(v, wr[i], acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) it.y[(int64_t)b * it.O + o] = acc + (it.bias ? it.bias[o] : 0.f);
}

// block per (b, 16 inputs): 16 slices of the output features are reduced through LDS.
// dx[b][i] = act'(x) * sum_o dy[b][o] W[o][i]
__global__ void __launch_bounds__(256)
linear_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ W,
                     const float* __restrict__ dy, float* __restrict__ dx, int B,
                     int I, int O, int act_in, float slope, int accumulate) {
    __shared__ float red[16][17];
    const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int chunks = (I + 15) / 16;
    const int b = blockIdx.x / chunks, i = (blockIdx.x - b * chunks) * 16 + il;
    float acc = 0.f;
    if (i < I) {
#pragma unroll 4
        for (int o = sl; o < O; o += 16) acc = fmaf(dy[(int64_t)b * O + o], W[(int64_t)o * I + i], acc);
    }
    red[sl][il] = acc;
    __syncthreads();
    if (sl != 0 || i >= I) return;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][il];
    const int64_t idx = (int64_t)b * I + i;
    if (act_in && !(x[idx] > 0.f)) v *= slope;
    dx[idx] = accumulate ? dx[idx] + v : v;
}

// dW[o][i] = sum_b dy[b][o] act(x[b][i]); the i == 0 column also writes dbias[o].  A block owns 16
// outputs o x 64 inputs i and walks the batch in slabs of 64 rows staged in LDS (the sum runs over b
// in ascending order, as a scalar loop would).
__global__ void __launch_bounds__(256)
linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                     float* __restrict__ dW, float* __restrict__ dbias, int B,
                     int I, int O, int act_in, float slope) {
    __shared__ float xs[64][64];
    __shared__ float dys[64][16];
    const int itiles = (I + 63) / 64;
    const int ot = blockIdx.x / itiles, it = blockIdx.x - ot * itiles;
    const int o0 = ot * 16, i0 = it * 64;
    const int il = threadIdx.x & 63, ol = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, accb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int nb = B - b0 < 64 ? B - b0 : 64;
        for (int idx = threadIdx.x; idx < nb * 64; idx += 256) {
            const int b = idx >> 6, i = idx & 63;
            float v = i0 + i < I ? x[(int64_t)(b0 + b) * I + i0 + i] : 0.f;
            if (act_in) v = leaky(v, slope);
            xs[b][i] = v;
        }
        for (int idx = threadIdx.x; idx < nb * 16; idx += 256) {
            const int b = idx >> 4, o = idx & 15;
            dys[b][o] = o0 + o < O ? dy[(int64_t)(b0 + b) * O + o0 + o] : 0.f;
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            const float v = xs[b][il];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g = dys[b][ol * 4 + q];
                acc[q] = fmaf(g, v, acc[q]);
                accb[q] += g;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = o0 + ol * 4 + q, i = i0 + il;
        if (o < O && i < I) {
            dW[(int64_t)o * I + i] = acc[q];
            if (i == 0 && dbias) dbias[o] = accb[q];
        }
    }
}

// The gradients of several linear layers over the SAME input in two launches (every ResNetBlock's linear_temb in the
// backward pass): dx[b][i] = act'(x) * sum over items and their outputs of dy[b][o] W[o][i], and per item dW, dbias as
// linear_bwd_dw_kernel computes them.  Items are found by a scan of the (few) entries.
__global__ void __launch_bounds__(256)
linear_bwd_dx_multi_kernel(const float* __restrict__ x, const vamd_linear_bwd_item* __restrict__ items, int n,
                           float* __restrict__ dx, int B, int I, int act_in, float slope, int accumulate) {
    __shared__ float red[16][17];
    const int il = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int chunks = (I + 15) / 16;
    const int b = blockIdx.x / chunks, i = (blockIdx.x - b * chunks) * 16 + il;
    float acc = 0.f;
    if (i < I) {
        for (int k = 0; k < n; ++k) {
            const vamd_linear_bwd_item it = items[k];
            const float* dyr = it.dy + (int64_t)b * it.O;
#pragma unroll 4
            for (int o = sl; o < it.O; o += 16) acc = fmaf(dyr[o], it.W[(int64_t)o * I + i], acc);
        }
    }
    red[sl][il] = acc;
    __syncthreads();
    if (sl != 0 || i >= I) return;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += red[q][il];
    const int64_t idx = (int64_t)b * I + i;
    if (act_in && !(x[idx] > 0.f)) v *= slope;
    dx[idx] = accumulate ? dx[idx] + v : v;
}

__global__ void __launch_bounds__(256)
linear_bwd_dw_multi_kernel(const float* __restrict__ x, const vamd_linear_bwd_item* __restrict__ items, int n, int B, int I,
                           int act_in, float slope) {
    __shared__ float xs[64][64];
    __shared__ float dys[64][16];
    const int itiles = (I + 63) / 64;
    int ot = blockIdx.x / itiles;
    const int it_ = blockIdx.x - ot * itiles;
    int k = 0;
    while (k + 1 < n && ot >= (items[k].O + 15) / 16) { ot -= (items[k].O + 15) / 16; ++k; }
    const vamd_linear_bwd_item item = items[k];
    const int O = item.O;
    const int o0 = ot * 16, i0 = it_ * 64;
    if (o0 >= O) return;                                       // uniform: past the last item's tiles
    const int il = threadIdx.x & 63, ol = threadIdx.x >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, accb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int nb = B - b0 < 64 ? B - b0 : 64;
        for (int idx = threadIdx.x; idx < nb * 64; idx += 256) {
            const int b = idx >> 6, i = idx & 63;
            float v = i0 + i < I ? x[(int64_t)(b0 + b) * I + i0 + i] : 0.f;
            if (act_in) v = leaky(v, slope);
            xs[b][i] = v;
        }
        for (int idx = threadIdx.x; idx < nb * 16; idx += 256) {
            const int b = idx >> 4, o = idx & 15;
            dys[b][o] = o0 + o < O ? item.dy[(int64_t)(b0 + b) * O + o0 + o] : 0.f;
        }
        __syncthreads();
        for (int b = 0; b < nb; ++b) {
            const float v = xs[b][il];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float g = dys[b][ol * 4 + q];
                acc[q] = fmaf(g, v, acc[q]);
                accb[q] += g;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = o0 + ol * 4 + q, i = i0 + il;
        if (o < O && i < I) {
            item.dW[(int64_t)o * I + i] = acc[q];
            if (i == 0 && item.dbias) item.dbias[o] = accb[q];
        }
    }
}

// L2 false: torch.optim.AdamW (decoupled decay p *= 1 - lr wd); true: torch.optim.Adam with weight_decay (g += wd p)
template <bool L2>
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                             float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                             const float* __restrict__ lr_dev, float beta1, float beta2, float eps,
                             float wd, float bc1, float rsqrt_bc2, float gscale, const int64_t* __restrict__ step_dev) {
    if (lr_dev) lr = lr_dev[0];
    if (step_dev) {                                      // the step count lives on the device (a captured launch is replayed)
        const float st = (float)step_dev[0];
        bc1 = 1.f - powf(beta1, st);
        rsqrt_bc2 = rsqrtf(1.f - powf(beta2, st));
    }
    const float step_size = lr / bc1;
    const float decay = L2 ? 1.f : 1.f - lr * wd;
    const float l2 = L2 ? wd : 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t nvec = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; float* G = &gg.x; float* M = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = L2 ? fmaf(l2, P[j], G[j] * gscale) : G[j] * gscale;
            M[j] = beta1 * M[j] + (1.f - beta1) * gr;
            V[j] = beta2 * V[j] + (1.f - beta2) * gr * gr;
            const float denom = sqrtf(V[j]) * rsqrt_bc2 + eps;
            P[j] = P[j] * decay - step_size * (M[j] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int64_t i = (nvec << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gr = L2 ? fmaf(l2, p[i], g[i] * gscale) : g[i] * gscale;
        const float mi = beta1 * m[i] + (1.f - beta1) * gr;
        const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] * decay - step_size * (mi / (sqrtf(vi) * rsqrt_bc2 + eps));
    }
}

// w [O][I][k][k] fp32 -> wf [O][k][k][I], wd [I][k][k][O] with kernel axes flipped
template <typename T>
__global__ void repack_weight_kernel(const float* __restrict__ w, T* __restrict__ wf,
                                     T* __restrict__ wd, int O, int I, int k) {
    const int64_t total = (int64_t)O * I * k * k;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        // idx enumerates the destination wf layout (o, r, s, i) so writes are coalesced
        const int i = (int)(idx % I);
        int64_t rest = idx / I;
        const int s = (int)(rest % k); rest /= k;
        const int r = (int)(rest % k);
        const int o = (int)(rest / k);
        const float v = w[(((int64_t)o * I + i) * k + r) * k + s];
        wf[idx] = from_f<T>(v);
        if (wd) wd[(((int64_t)i * k + (k - 1 - r)) * k + (k - 1 - s)) * O + o] = from_f<T>(v);
    }
}

// Every convolution of a model in one launch: block -> item by a search over the items' first
// blocks, then one 32 (Cout) x 32 (Cin) x k*k tile through LDS, so that the fp32 source is read in
// contiguous runs and both packed copies are written in 64-byte runs.
template <typename T>
__global__ void __launch_bounds__(256)
repack_weights_kernel(const vamd_repack_item* __restrict__ items, int n) {
    constexpr int TO = 32, TI = 32;
    __shared__ float tile[TO][TI * 9 + 1];
    int lo = 0, hi = n - 1;                                  // last item whose first_block <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const vamd_repack_item it = items[lo];
    const int O = it.Cout, I = it.Cin, k = it.ksize, kk = k * k;
    const int LD = it.ld_dgrad > 0 ? it.ld_dgrad : O;         // elements between the rows [i][tap] of w_dgrad
    const float* __restrict__ w = it.w;
    T* __restrict__ wf = (T*)it.w_fwd;
    T* __restrict__ wd = (T*)it.w_dgrad;
    const int local = (int)((long long)blockIdx.x - it.first_block);
    if (local == 0 && it.bias_out)                           // the bias rides along (a fused projection's concatenated bias)
        for (int n = threadIdx.x; n < O; n += 256) it.bias_out[n] = it.bias[n];
    const int i_tiles = (I + TI - 1) / TI;
    const int o0 = (local / i_tiles) * TO, i0 = (local % i_tiles) * TI;
    const int run = TI * kk;                                 // source floats per output channel in this tile
    for (int idx = threadIdx.x; idx < TO * run; idx += 256) {
        const int ol = idx / run, rem = idx - ol * run;
        const int o = o0 + ol, i = i0 + rem / kk;
        tile[ol][rem] = (o < O && i < I) ? w[((int64_t)o * I + i0) * kk + rem] : 0.f;
    }
    __syncthreads();
    // two consecutive elements per thread and store where the channel counts are even (all but the 3-channel edges)
    const bool pair_i = (I & 1) == 0, pair_o = (O & 1) == 0;
    if (pair_i) {
        for (int idx = threadIdx.x; idx < TO * run / 2; idx += 256) {  // wf[o][tap][i], i in pairs
            const int il = (idx % (TI / 2)) * 2, tap = (idx / (TI / 2)) % kk, ol = idx / (run / 2);
            const int o = o0 + ol, i = i0 + il;
            if (o < O && i < I) {
                T two[2] = {from_f<T>(tile[ol][il * kk + tap]), from_f<T>(tile[ol][(il + 1) * kk + tap])};
                T* dst = wf + ((int64_t)o * kk + tap) * I + i;
                if (sizeof(T) == 2) *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(two);
                else *reinterpret_cast<uint64_t*>(dst) = *reinterpret_cast<const uint64_t*>(two);
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < TO * run; idx += 256) {      // wf[o][tap][i]
            const int il = idx % TI, tap = (idx / TI) % kk, ol = idx / run;
            const int o = o0 + ol, i = i0 + il;
            if (o < O && i < I) wf[((int64_t)o * kk + tap) * I + i] = from_f<T>(tile[ol][il * kk + tap]);
        }
    }
    // The two combined filters of an upsampling convolution (nearest 2x -> 3x3), from the same tile: the 4x4 stride-2
    // filter of its input gradient, w_up4_dgrad [I][4][4][O] (along each axis tap t collects the conv taps {2}, {1,2},
    // {0,1}, {0}), and the four 2x2 class filters of its forward, w_up4_fwd [ph*2+pw][O][a][b][I] (ph = 0: a = 0 -> {0},
    // a = 1 -> {1,2}; ph = 1: a = 0 -> {0,1}, a = 1 -> {2}); sums formed in fp32 and rounded once.
    if (it.w_up4_dgrad && kk == 9) {
        T* __restrict__ g4 = (T*)it.w_up4_dgrad;
        for (int idx = threadIdx.x; idx < TI * 16 * TO; idx += 256) {  // [i][t][u][o], o fastest
            const int ol = idx % TO, u = (idx / TO) & 3, t = (idx / (TO * 4)) & 3, il = idx / (TO * 16);
            const int o = o0 + ol, i = i0 + il;
            if (o >= O || i >= I) continue;
            const int r_lo = t == 0 ? 2 : (t == 1 ? 1 : 0), r_hi = t <= 1 ? 2 : (t == 2 ? 1 : 0);
            const int s_lo = u == 0 ? 2 : (u == 1 ? 1 : 0), s_hi = u <= 1 ? 2 : (u == 2 ? 1 : 0);
            float sum = 0.f;
            for (int r = r_lo; r <= r_hi; ++r)
                for (int q = s_lo; q <= s_hi; ++q) sum += tile[ol][il * 9 + r * 3 + q];
            g4[(((int64_t)i * 4 + t) * 4 + u) * O + o] = from_f<T>(sum);
        }
    }
    if (it.w_up4_fwd && kk == 9) {
        T* __restrict__ f4 = (T*)it.w_up4_fwd;
        for (int idx = threadIdx.x; idx < 4 * TO * 4 * TI; idx += 256) {  // [cls][o][a][b][i], i fastest
            const int il = idx % TI, bq = (idx / TI) & 1, aq = (idx / (TI * 2)) & 1, ol = (idx / (TI * 4)) % TO, cls = idx / (TI * 4 * TO);
            const int o = o0 + ol, i = i0 + il;
            if (o >= O || i >= I) continue;
            const int ph = cls >> 1, pw = cls & 1;
            const int r_lo = ph == 0 ? (aq == 0 ? 0 : 1) : (aq == 0 ? 0 : 2), r_hi = ph == 0 ? (aq == 0 ? 0 : 2) : (aq == 0 ? 1 : 2);
            const int s_lo = pw == 0 ? (bq == 0 ? 0 : 1) : (bq == 0 ? 0 : 2), s_hi = pw == 0 ? (bq == 0 ? 0 : 2) : (bq == 0 ? 1 : 2);
            float sum = 0.f;
            for (int r = r_lo; r <= r_hi; ++r)
                for (int q = s_lo; q <= s_hi; ++q) sum += tile[ol][il * 9 + r * 3 + q];
            f4[((((int64_t)cls * O + o) * 2 + aq) * 2 + bq) * I + i] = from_f<T>(sum);
        }
    }
    if (!wd) return;
    if (pair_o) {
        for (int idx = threadIdx.x; idx < TO * run / 2; idx += 256) {  // wd[i][flipped tap][o], o in pairs
            const int ol = (idx % (TO / 2)) * 2, tap = (idx / (TO / 2)) % kk, il = idx / (TO / 2 * kk);
            const int o = o0 + ol, i = i0 + il;
            if (o < O && i < I) {
                T two[2] = {from_f<T>(tile[ol][il * kk + tap]), from_f<T>(tile[ol + 1][il * kk + tap])};
                T* dst = wd + ((int64_t)i * kk + (kk - 1 - tap)) * LD + o;
                if (sizeof(T) == 2) *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(two);
                else *reinterpret_cast<uint64_t*>(dst) = *reinterpret_cast<const uint64_t*>(two);
            }
        }
    } else {
        for (int idx = threadIdx.x; idx < TO * run; idx += 256) {      // wd[i][flipped tap][o]
            const int ol = idx % TO, tap = (idx / TO) % kk, il = idx / (TO * kk);
            const int o = o0 + ol, i = i0 + il;
            if (o < O && i < I) wd[((int64_t)i * kk + (kk - 1 - tap)) * LD + o] = from_f<T>(tile[ol][il * kk + tap]);
        }
    }
}

// Combined filter of the input gradient of (nearest 2x upsample -> 3x3 conv): along each axis the
// four dY taps t = 0..3 collect the conv taps {2}, {1,2}, {0,1}, {0}.  w [O][I][3][3] fp32 ->
// w4 [I][4][4][O] in T (sums formed in fp32, rounded once).
template <typename T>
__global__ void repack_weight_up4_kernel(const float* __restrict__ w, T* __restrict__ w4, int O, int I) {
    const int64_t total = (int64_t)I * 16 * O;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % O);
        int64_t rest = idx / O;
        const int u = (int)(rest % 4); rest /= 4;
        const int t = (int)(rest % 4);
        const int i = (int)(rest / 4);
        // taps of S(t): t=0 -> {2}, 1 -> {1,2}, 2 -> {0,1}, 3 -> {0}
        const int r_lo = t == 0 ? 2 : (t == 1 ? 1 : 0), r_hi = t <= 1 ? 2 : (t == 2 ? 1 : 0);
        const int s_lo = u == 0 ? 2 : (u == 1 ? 1 : 0), s_hi = u <= 1 ? 2 : (u == 2 ? 1 : 0);
        const float* src = w + ((int64_t)o * I + i) * 9;
        float sum = 0.f;
        for (int r = r_lo; r <= r_hi; ++r)
            for (int q = s_lo; q <= s_hi; ++q) sum += src[r * 3 + q];
        w4[idx] = from_f<T>(sum);
    }
}

// Class filters of the forward of (nearest 2x upsample -> 3x3 conv): class (ph, pw), tap (a, b) sums
// the conv taps that read the same stored pixel (ph = 0: {0}, {1,2}; ph = 1: {0,1}, {2}).
// w [O][I][3][3] fp32 -> w4 [ph*2+pw][O][a][b][I] in T.
template <typename T>
__global__ void repack_weight_up4_fwd_kernel(const float* __restrict__ w, T* __restrict__ w4, int O, int I) {
    const int64_t total = (int64_t)4 * O * 4 * I;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % I);
        int64_t rest = idx / I;
        const int bq = (int)(rest % 2); rest /= 2;
        const int aq = (int)(rest % 2); rest /= 2;
        const int o = (int)(rest % O);
        const int cls = (int)(rest / O);
        const int ph = cls >> 1, pw = cls & 1;
        // ph = 0: a = 0 -> {0}, a = 1 -> {1,2};  ph = 1: a = 0 -> {0,1}, a = 1 -> {2}
        const int r_lo = ph == 0 ? (aq == 0 ? 0 : 1) : (aq == 0 ? 0 : 2), r_hi = ph == 0 ? (aq == 0 ? 0 : 2) : (aq == 0 ? 1 : 2);
        const int s_lo = pw == 0 ? (bq == 0 ? 0 : 1) : (bq == 0 ? 0 : 2), s_hi = pw == 0 ? (bq == 0 ? 0 : 2) : (bq == 0 ? 1 : 2);
        const float* src = w + ((int64_t)o * I + i) * 9;
        float sum = 0.f;
        for (int r = r_lo; r <= r_hi; ++r)
            for (int q = s_lo; q <= s_hi; ++q) sum += src[r * 3 + q];
        w4[idx] = from_f<T>(sum);
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwk, float* __restrict__ dw, int O,
                                    int I, int k) {
    const int64_t total = (int64_t)O * I * k * k;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % I);
        int64_t rest = idx / I;
        const int s = (int)(rest % k); rest /= k;
        const int r = (int)(rest % k);
        const int o = (int)(rest / k);
        dw[(((int64_t)o * I + i) * k + r) * k + s] = dwk[idx];
    }
}

// 2x2 pooling and its transpose on channels-last tensors (the resampling blocks built without a convolution,
// reference models/ddpm.py:21 nn.AvgPool2d((2, 2)) and :33 nn.Upsample(scale_factor=2, mode='nearest')):
//   pool:   y[b][h][w][c] = scale * (sum of x[b][2h..2h+1][2w..2w+1][c]) + add      (H/2 x W/2, floor)
//   unpool: y[b][h][w][c] = scale * x[b][h/2][w/2][c] + add  for h < 2*Hs, w < 2*Ws, else add alone (or 0)
// AvgPool forward = pool(0.25), its gradient = unpool(0.25); nearest upsampling = unpool(1), its gradient = pool(1).
template <typename T, int VEC>
__global__ void pool2x2_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ y, int B, int H, int W,
                               int C, float scale) {
    const int Ho = H / 2, Wo = W / 2, CV = C / VEC;
    const int64_t total = (int64_t)B * Ho * Wo * CV;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % CV);
        int64_t p = idx / CV;
        const int w = (int)(p % Wo); p /= Wo;
        const int h = (int)(p % Ho);
        const int64_t b = p / Ho;
        const T* s = x + ((b * H + 2 * h) * W + 2 * w) * (int64_t)C + cv * VEC;
        const int64_t row = (int64_t)W * C;
        T v00[VEC], v01[VEC], v10[VEC], v11[VEC], av[VEC], ov[VEC];
        if (VEC > 1) {
            *reinterpret_cast<Vec16<T>*>(v00) = *reinterpret_cast<const Vec16<T>*>(s);
            *reinterpret_cast<Vec16<T>*>(v01) = *reinterpret_cast<const Vec16<T>*>(s + C);
            *reinterpret_cast<Vec16<T>*>(v10) = *reinterpret_cast<const Vec16<T>*>(s + row);
            *reinterpret_cast<Vec16<T>*>(v11) = *reinterpret_cast<const Vec16<T>*>(s + row + C);
            if (add) *reinterpret_cast<Vec16<T>*>(av) = *reinterpret_cast<const Vec16<T>*>(add + idx * VEC);
        } else {
            v00[0] = s[0]; v01[0] = s[C]; v10[0] = s[row]; v11[0] = s[row + C];
            if (add) av[0] = add[idx];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float f = ((to_f(v00[j]) + to_f(v01[j])) + to_f(v10[j]) + to_f(v11[j])) * scale;
            if (add) f += to_f(av[j]);
            ov[j] = from_f<T>(f);
        }
        if (VEC > 1) *reinterpret_cast<Vec16<T>*>(y + idx * VEC) = *reinterpret_cast<Vec16<T>*>(ov);
        else y[idx] = ov[0];
    }
}

template <typename T, int VEC>
__global__ void unpool2x2_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ y, int B, int Hs, int Ws,
                                 int C, int H, int W, float scale) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % CV);
        int64_t p = idx / CV;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int64_t b = p / H;
        const bool in = (h >> 1) < Hs && (w >> 1) < Ws;
        T xv[VEC], av[VEC], ov[VEC];
        if (VEC > 1) {
            if (in) *reinterpret_cast<Vec16<T>*>(xv) =
                *reinterpret_cast<const Vec16<T>*>(x + ((b * Hs + (h >> 1)) * Ws + (w >> 1)) * (int64_t)C + cv * VEC);
            if (add) *reinterpret_cast<Vec16<T>*>(av) = *reinterpret_cast<const Vec16<T>*>(add + idx * VEC);
        } else {
            if (in) xv[0] = x[((b * Hs + (h >> 1)) * Ws + (w >> 1)) * (int64_t)C + cv];
            if (add) av[0] = add[idx];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float f = in ? to_f(xv[j]) * scale : 0.f;
            if (add) f += to_f(av[j]);
            ov[j] = from_f<T>(f);
        }
        if (VEC > 1) *reinterpret_cast<Vec16<T>*>(y + idx * VEC) = *reinterpret_cast<Vec16<T>*>(ov);
        else y[idx] = ov[0];
    }
}

}  // namespace vamd

using namespace vamd;

extern "C" {

const char* vamd_last_error(void) { return g_err; }
int vamd_version(void) { return 1; }

// `waiter` runs nothing more until everything `signaller` has been given so far is done (an event recorded on one, waited
// for on the other).  Events come from a per-device ring and are reused: a wait refers to the record made just before it.
int vamd_stream_wait(void* waiter, void* signaller) {
    constexpr int kRing = 256, kDevs = 64;
    static hipEvent_t ring[kDevs][kRing];
    static int next[kDevs];
    static std::mutex mu;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kDevs) return fail((int)e, "vamd_stream_wait: hipGetDevice failed or device %d out of range", dev);
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(mu);
        int& n = next[dev];
        if (!ring[dev][n]) {
            e = hipEventCreateWithFlags(&ring[dev][n], hipEventDisableTiming);
            if (e != hipSuccess) return fail((int)e, "vamd_stream_wait: hipEventCreateWithFlags failed: %s", hipGetErrorString(e));
        }
        ev = ring[dev][n];
        n = (n + 1) % kRing;
    }
    e = hipEventRecord(ev, as_stream(signaller));
    if (e != hipSuccess) return fail((int)e, "vamd_stream_wait: hipEventRecord failed: %s", hipGetErrorString(e));
    e = hipStreamWaitEvent(as_stream(waiter), ev, 0);
    if (e != hipSuccess) return fail((int)e, "vamd_stream_wait: hipStreamWaitEvent failed: %s", hipGetErrorString(e));
    return 0;
}

int vamd_launch_timing(int enable) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    g_timed_on = enable != 0;
    if (g_timed_on) g_timed_used = 0;
    return 0;
}

long vamd_launch_timing_read(int kind, double* total_ms, double* total_flops, double* executed_flops) {
    std::lock_guard<std::mutex> lk(g_timed_mu);
    double ms = 0.0, fl = 0.0, ex = 0.0;
    long n = 0;
    for (size_t i = 0; i < g_timed_used; ++i) {
        const TimedSlot& s = g_timed[i];
        if (s.kind != kind) continue;
        hipError_t e = hipEventSynchronize(s.e1);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, s.e0, s.e1);
        if (e != hipSuccess) return -(long)fail((int)e, "vamd_launch_timing_read: %s", hipGetErrorString(e));
        ms += t;
        fl += s.flops;
        ex += s.executed;
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (executed_flops) *executed_flops = ex;
    return n;
}

int vamd_schedule(const float* t, float* n, float* s, float* n2, int B, void* stream) {
    VAMD_REQUIRE(t && n && s && n2 && B > 0, "vamd_schedule: null pointer or B <= 0");
    hipLaunchKernelGGL(schedule_kernel, dim3((B + 63) / 64), dim3(64), 0, as_stream(stream), t, n, s, n2, B);
    VAMD_LAUNCH_OK("vamd_schedule");
    return 0;
}

// Channels per work item of the kernels at the model boundary: a pixel's whole channel vector for thin tensors (images:
// the NCHW side is then read plane by plane with consecutive lanes on consecutive pixels), single channels otherwise.
static int chan_width(int C) { return C <= 8 ? C : 1; }
static int64_t chan_slices(int C) { return (C + chan_width(C) - 1) / chan_width(C); }
constexpr int kTiledConvertMinC = 16;

int vamd_qsample(const float* x0, const float* eps, const float* t, void* xt, float* n, float* s,
                 float* n2, int B, int C, int H, int W, int dtype, void* stream) {
    VAMD_REQUIRE(x0 && eps && t && xt && n && s && n2, "vamd_qsample: null pointer");
    VAMD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "vamd_qsample: bad shape %d %d %d %d", B, C, H, W);
    const int64_t HW = (int64_t)H * W;
    VAMD_DISPATCH(dtype, "vamd_qsample", {
        hipLaunchKernelGGL(qsample_kernel<T>, dim3(grid_for(B * HW * chan_slices(C), 256)), dim3(256), 0,
                           as_stream(stream), x0, eps, t, (T*)xt, n, s, n2, B, C, HW, chan_width(C));
    });
    VAMD_LAUNCH_OK("vamd_qsample");
    return 0;
}

// at least one workgroup per CU: a many-channel tensor (a latent: 256 channels on 7 x 7 pixels) has few pixels and
// many work items per pixel, and 13 workgroups walked its 0.8 M elements in 139 us
int vamd_l1_chunks(int64_t npix) {
    const int n = grid_for(npix, 256, 1024);
    return n < 256 ? 256 : n;
}

int vamd_l1_loss_fwd(const void* pred, const float* eps, float* partial, float* loss, int B, int C,
                     int H, int W, int dtype, void* stream) {
    VAMD_REQUIRE(pred && eps && partial && loss, "vamd_l1_loss_fwd: null pointer");
    VAMD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "vamd_l1_loss_fwd: bad shape");
    const int64_t HW = (int64_t)H * W;
    const int nchunk = vamd_l1_chunks(B * HW);
    VAMD_DISPATCH(dtype, "vamd_l1_loss_fwd", {
        hipLaunchKernelGGL(l1_fwd_kernel<T>, dim3(nchunk), dim3(256), 0, as_stream(stream),
                           (const T*)pred, eps, partial, B, C, HW, chan_width(C));
    });
    hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), partial, nchunk,
                       1.0 / ((double)B * C * HW), loss);
    VAMD_LAUNCH_OK("vamd_l1_loss_fwd");
    return 0;
}

int vamd_l1_loss_bwd(const void* pred, const float* eps, const float* gout, void* dpred, int B,
                     int C, int H, int W, int dtype, void* stream) {
    VAMD_REQUIRE(pred && eps && gout && dpred, "vamd_l1_loss_bwd: null pointer");
    VAMD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "vamd_l1_loss_bwd: bad shape");
    const int64_t HW = (int64_t)H * W;
    VAMD_DISPATCH(dtype, "vamd_l1_loss_bwd", {
        hipLaunchKernelGGL(l1_bwd_kernel<T>, dim3(grid_for(B * HW * chan_slices(C), 256)), dim3(256), 0,
                           as_stream(stream), (const T*)pred, eps, gout, (T*)dpred, B, C, HW,
                           (float)(1.0 / ((double)B * C * HW)), chan_width(C));
    });
    VAMD_LAUNCH_OK("vamd_l1_loss_bwd");
    return 0;
}

int vamd_sigmoid_mse_fwd(const void* h, const float* target, float* xrec, float* partial, float* loss, int B, int C, int H,
                         int W, int dtype, void* stream) {
    VAMD_REQUIRE(h && xrec && B > 0 && C > 0 && H > 0 && W > 0, "vamd_sigmoid_mse_fwd: bad argument");
    VAMD_REQUIRE(!target == !loss && !target == !partial, "vamd_sigmoid_mse_fwd: target, partial and loss go together");
    const int64_t HW = (int64_t)H * W;
    const int nchunk = vamd_l1_chunks(B * HW);
    VAMD_DISPATCH(dtype, "vamd_sigmoid_mse_fwd", {
        hipLaunchKernelGGL(sigmoid_mse_fwd_kernel<T>, dim3(nchunk), dim3(256), 0, as_stream(stream), (const T*)h, target, xrec,
                           partial, B, C, HW);
    });
    if (target)
        hipLaunchKernelGGL(l1_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), partial, nchunk,
                           1.0 / ((double)B * C * HW), loss);
    VAMD_LAUNCH_OK("vamd_sigmoid_mse_fwd");
    return 0;
}

int vamd_sigmoid_mse_bwd(const float* xrec, const float* target, const float* g_loss, const float* g_xrec, void* dh, int B,
                         int C, int H, int W, int dtype, void* stream) {
    VAMD_REQUIRE(xrec && dh && B > 0 && C > 0 && H > 0 && W > 0, "vamd_sigmoid_mse_bwd: bad argument");
    VAMD_REQUIRE(!target == !g_loss, "vamd_sigmoid_mse_bwd: target and g_loss go together");
    const int64_t HW = (int64_t)H * W;
    VAMD_DISPATCH(dtype, "vamd_sigmoid_mse_bwd", {
        hipLaunchKernelGGL(sigmoid_mse_bwd_kernel<T>, dim3(grid_for(B * HW, 256)), dim3(256), 0, as_stream(stream), xrec, target,
                           g_loss, g_xrec, (T*)dh, B, C, HW, (float)(2.0 / ((double)B * C * HW)));
    });
    VAMD_LAUNCH_OK("vamd_sigmoid_mse_bwd");
    return 0;
}

int vamd_denoise_step(float* cur, const void* eps_hat, const float* n, const float* s,
                      const float* n_next, const float* s_next, void* next_in, float* pred_images,
                      int B, int C, int H, int W, int dtype, void* stream) {
    VAMD_REQUIRE(cur && eps_hat && n && s && n_next && s_next && next_in, "vamd_denoise_step: null pointer");
    VAMD_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "vamd_denoise_step: bad shape");
    const int64_t HW = (int64_t)H * W;
    VAMD_DISPATCH(dtype, "vamd_denoise_step", {
        hipLaunchKernelGGL(denoise_step_kernel<T>, dim3(grid_for(B * HW * chan_slices(C), 256)), dim3(256), 0,
                           as_stream(stream), cur, (const T*)eps_hat, n, s, n_next, s_next,
                           (T*)next_in, pred_images, B, C, HW, chan_width(C));
    });
    VAMD_LAUNCH_OK("vamd_denoise_step");
    return 0;
}

int vamd_nchw_to_nhwc(const float* x, void* y, float* y32, int B, int C, int H, int W, int dtype,
                      void* stream) {
    VAMD_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "vamd_nchw_to_nhwc: bad argument");
    const int64_t HW = (int64_t)H * W;
    if (C >= kTiledConvertMinC && B <= 65535 && (C + 31) / 32 <= 65535) {       // x [B][C][HW] -> y [B][HW][C]
        const dim3 grid((unsigned)((HW + 31) / 32), (C + 31) / 32, B);
        VAMD_DISPATCH(dtype, "vamd_nchw_to_nhwc", {
            hipLaunchKernelGGL((convert_transpose_kernel<float, T>), grid, dim3(256), 0, as_stream(stream), x, (T*)y, y32, C,
                               (int)HW);
        });
        VAMD_LAUNCH_OK("vamd_nchw_to_nhwc");
        return 0;
    }
    VAMD_DISPATCH(dtype, "vamd_nchw_to_nhwc", {
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid_for(B * HW, 256)), dim3(256), 0,
                           as_stream(stream), x, (T*)y, y32, B, C, HW);
    });
    VAMD_LAUNCH_OK("vamd_nchw_to_nhwc");
    return 0;
}

int vamd_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int dtype, void* stream) {
    VAMD_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "vamd_nhwc_to_nchw: bad argument");
    const int64_t HW = (int64_t)H * W;
    if (C >= kTiledConvertMinC && B <= 65535 && (HW + 31) / 32 <= 65535) {      // x [B][HW][C] -> y [B][C][HW]
        const dim3 grid((C + 31) / 32, (unsigned)((HW + 31) / 32), B);
        VAMD_DISPATCH(dtype, "vamd_nhwc_to_nchw", {
            hipLaunchKernelGGL((convert_transpose_kernel<T, float>), grid, dim3(256), 0, as_stream(stream), (const T*)x, y,
                               (float*)nullptr, (int)HW, C);
        });
        VAMD_LAUNCH_OK("vamd_nhwc_to_nchw");
        return 0;
    }
    VAMD_DISPATCH(dtype, "vamd_nhwc_to_nchw", {
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(grid_for(B * HW, 256)), dim3(256), 0,
                           as_stream(stream), (const T*)x, y, B, C, HW);
    });
    VAMD_LAUNCH_OK("vamd_nhwc_to_nchw");
    return 0;
}

int vamd_add_cast(const void* a, int a_dtype, const void* b, int b_dtype, void* out, int out_dtype, int64_t n, void* stream) {
    VAMD_REQUIRE(a && out && n > 0, "vamd_add_cast: bad argument");
    const auto ok = [](int d) { return d == VAMD_F32 || d == VAMD_BF16; };
    VAMD_REQUIRE(ok(a_dtype) && ok(out_dtype) && (!b || ok(b_dtype)), "vamd_add_cast: dtype");
    hipStream_t st = as_stream(stream);
    const bool bb = b && b_dtype == VAMD_BF16;
    if (a_dtype == VAMD_BF16) {
        if (bb) add_cast_out<bf16_t, bf16_t>(a, b, out, out_dtype, n, st);
        else add_cast_out<bf16_t, float>(a, b, out, out_dtype, n, st);
    } else {
        if (bb) add_cast_out<float, bf16_t>(a, b, out, out_dtype, n, st);
        else add_cast_out<float, float>(a, b, out, out_dtype, n, st);
    }
    VAMD_LAUNCH_OK("vamd_add_cast");
    return 0;
}

int vamd_transpose_ld(const void* x, void* y, int batch, int R, int C, int ldx, int ldy, int dtype, void* stream) {
    VAMD_REQUIRE(x && y && batch > 0 && R > 0 && C > 0 && ldx >= C && ldy >= R, "vamd_transpose: bad argument");
    VAMD_REQUIRE(batch <= 65535 && (R + 31) / 32 <= 65535, "vamd_transpose: grid too large");
    dim3 grid((C + 31) / 32, (R + 31) / 32, batch);
    VAMD_DISPATCH(dtype, "vamd_transpose", {
        hipLaunchKernelGGL(transpose_kernel<T>, grid, dim3(256), 0, as_stream(stream), (const T*)x,
                           (T*)y, R, C, ldx, ldy);
    });
    VAMD_LAUNCH_OK("vamd_transpose");
    return 0;
}

int vamd_transpose(const void* x, void* y, int batch, int R, int C, int ldy, int dtype, void* stream) {
    return vamd_transpose_ld(x, y, batch, R, C, C, ldy, dtype, stream);
}

int vamd_softmax_rows(const float* x, void* y, int64_t rows, int cols, int ld, int dtype, void* stream) {
    VAMD_REQUIRE(x && y && rows > 0 && cols > 0 && ld >= cols, "vamd_softmax_rows: bad argument");
    VAMD_DISPATCH(dtype, "vamd_softmax_rows", {
        hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3(grid_for(rows, 4)), dim3(256), 0,
                           as_stream(stream), x, (T*)y, rows, cols, ld);
    });
    VAMD_LAUNCH_OK("vamd_softmax_rows");
    return 0;
}

int vamd_softmax_rows_bwd(const void* p, const float* dp, void* ds, int64_t rows, int cols, int ld,
                          float alpha, int dtype, void* stream) {
    VAMD_REQUIRE(p && dp && ds && rows > 0 && cols > 0 && ld >= cols, "vamd_softmax_rows_bwd: bad argument");
    VAMD_DISPATCH(dtype, "vamd_softmax_rows_bwd", {
        hipLaunchKernelGGL(softmax_rows_bwd_kernel<T>, dim3(grid_for(rows, 4)), dim3(256), 0,
                           as_stream(stream), (const T*)p, dp, (T*)ds, rows, cols, ld, alpha);
    });
    VAMD_LAUNCH_OK("vamd_softmax_rows_bwd");
    return 0;
}

int vamd_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
    VAMD_REQUIRE(a && b && y && n > 0, "vamd_add: bad argument");
    VAMD_REQUIRE(((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)y % 16 == 0),
                 "vamd_add: pointers must be 16-byte aligned");
    VAMD_DISPATCH(dtype, "vamd_add", {
        const int64_t nvec = n / Vec16<T>::N;
        hipLaunchKernelGGL(add_kernel<T>, dim3(grid_for(nvec > 0 ? nvec : 1, 256)), dim3(256), 0,
                           as_stream(stream), (const T*)a, (const T*)b, (T*)y, nvec, n);
    });
    VAMD_LAUNCH_OK("vamd_add");
    return 0;
}

int vamd_pool2x2(const void* x, const void* add, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream) {
    VAMD_REQUIRE(x && y && B > 0 && H >= 2 && W >= 2 && C > 0, "vamd_pool2x2: bad argument");
    const int64_t total = (int64_t)B * (H / 2) * (W / 2) * C;
    VAMD_DISPATCH(dtype, "vamd_pool2x2", {
        constexpr int V = Vec16<T>::N;
        const bool vec = C % V == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)add % 16 == 0;
        if (vec)
            hipLaunchKernelGGL((pool2x2_kernel<T, V>), dim3(grid_for(total / V, 256, 65536)), dim3(256), 0, as_stream(stream),
                               (const T*)x, (const T*)add, (T*)y, B, H, W, C, scale);
        else
            hipLaunchKernelGGL((pool2x2_kernel<T, 1>), dim3(grid_for(total, 256, 65536)), dim3(256), 0, as_stream(stream),
                               (const T*)x, (const T*)add, (T*)y, B, H, W, C, scale);
    });
    VAMD_LAUNCH_OK("vamd_pool2x2");
    return 0;
}

int vamd_unpool2x2(const void* x, const void* add, void* y, int B, int Hs, int Ws, int C, int H, int W, float scale, int dtype,
                   void* stream) {
    VAMD_REQUIRE(x && y && B > 0 && Hs > 0 && Ws > 0 && C > 0, "vamd_unpool2x2: bad argument");
    VAMD_REQUIRE(H / 2 == Hs && W / 2 == Ws, "vamd_unpool2x2: output %d x %d is not twice %d x %d (plus at most one)", H, W, Hs, Ws);
    const int64_t total = (int64_t)B * H * W * C;
    VAMD_DISPATCH(dtype, "vamd_unpool2x2", {
        constexpr int V = Vec16<T>::N;
        const bool vec = C % V == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)add % 16 == 0;
        if (vec)
            hipLaunchKernelGGL((unpool2x2_kernel<T, V>), dim3(grid_for(total / V, 256, 65536)), dim3(256), 0, as_stream(stream),
                               (const T*)x, (const T*)add, (T*)y, B, Hs, Ws, C, H, W, scale);
        else
            hipLaunchKernelGGL((unpool2x2_kernel<T, 1>), dim3(grid_for(total, 256, 65536)), dim3(256), 0, as_stream(stream),
                               (const T*)x, (const T*)add, (T*)y, B, Hs, Ws, C, H, W, scale);
    });
    VAMD_LAUNCH_OK("vamd_unpool2x2");
    return 0;
}

int vamd_timeembed(const float* t, const float* angular_speeds, float* emb, int B, int dim,
                   void* stream) {
    VAMD_REQUIRE(t && angular_speeds && emb && B > 0 && dim > 0 && dim % 2 == 0,
                 "vamd_timeembed: bad argument (dim must be even)");
    const int total = B * (dim / 2);
    hipLaunchKernelGGL(timeembed_kernel, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream),
                       t, angular_speeds, emb, B, dim);
    VAMD_LAUNCH_OK("vamd_timeembed");
    return 0;
}

int vamd_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int I,
                    int O, int act_in, float slope, void* stream) {
    VAMD_REQUIRE(x && W && y && B > 0 && I > 0 && O > 0, "vamd_linear_fwd: bad argument");
    const int total = B * O;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((total + 3) / 4), dim3(256), 0, as_stream(stream),
                       x, W, bias, y, B, I, O, act_in, slope);
    VAMD_LAUNCH_OK("vamd_linear_fwd");
    return 0;
}

int vamd_linear_fwd_multi(const float* x, const vamd_linear_item* items, int n, int total_O, int B, int I,
                          int act_in, float slope, void* stream) {
    VAMD_REQUIRE(x && items && n > 0 && n <= 64 && total_O > 0 && B > 0 && I > 0, "vamd_linear_fwd_multi: bad argument");
    const int64_t total = (int64_t)B * total_O;
    VAMD_REQUIRE(total < (1ll << 30), "vamd_linear_fwd_multi: too many outputs");
    hipLaunchKernelGGL(linear_fwd_multi_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, as_stream(stream),
                       x, items, n, total_O, B, I, act_in, slope);
    VAMD_LAUNCH_OK("vamd_linear_fwd_multi");
    return 0;
}

int vamd_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW,
                    float* dbias, int B, int I, int O, int act_in, float slope, int accumulate_dx,
                    void* stream) {
    VAMD_REQUIRE(x && W && dy && dW && B > 0 && I > 0 && O > 0, "vamd_linear_bwd: bad argument");
    if (dx) {
        const int total = B * ((I + 15) / 16);
        hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(total), dim3(256), 0,
                           as_stream(stream), x, W, dy, dx, B, I, O, act_in, slope, accumulate_dx);
    }
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(((O + 15) / 16) * ((I + 63) / 64)), dim3(256), 0,
                       as_stream(stream), x, dy, dW, dbias, B, I, O, act_in, slope);
    VAMD_LAUNCH_OK("vamd_linear_bwd");
    return 0;
}

int vamd_linear_bwd_multi(const float* x, const vamd_linear_bwd_item* items, int n, int total_tiles, int B, int I, int act_in,
                          float slope, float* dx, int accumulate_dx, void* stream) {
    VAMD_REQUIRE(x && items && n > 0 && n <= 64 && total_tiles > 0 && B > 0 && I > 0, "vamd_linear_bwd_multi: bad argument");
    if (dx)
        hipLaunchKernelGGL(linear_bwd_dx_multi_kernel, dim3(B * ((I + 15) / 16)), dim3(256), 0, as_stream(stream), x, items, n, dx,
                           B, I, act_in, slope, accumulate_dx);
    hipLaunchKernelGGL(linear_bwd_dw_multi_kernel, dim3(total_tiles * ((I + 63) / 64)), dim3(256), 0, as_stream(stream), x, items,
                       n, B, I, act_in, slope);
    VAMD_LAUNCH_OK("vamd_linear_bwd_multi");
    return 0;
}

static int adam_launch(const char* what, bool l2, float* p, const float* g, float* m, float* v, int64_t n, float lr,
                       const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int step,
                       float grad_scale, void* stream, const int64_t* step_dev = nullptr) {
    VAMD_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_dev), "%s: bad argument", what);
    VAMD_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                     ((uintptr_t)v % 16 == 0), "%s: arenas must be 16-byte aligned", what);
    const double bc1 = 1.0 - pow((double)beta1, step >= 1 ? step : 1);
    const double bc2 = 1.0 - pow((double)beta2, step >= 1 ? step : 1);
    const dim3 grid(grid_for(n / 4 + 1, 256, 2048));
    if (l2)
        hipLaunchKernelGGL(adamw_kernel<true>, grid, dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, lr_dev, beta1, beta2,
                           eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, step_dev);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, grid, dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, lr_dev, beta1, beta2,
                           eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, step_dev);
    VAMD_LAUNCH_OK(what);
    return 0;
}

int vamd_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr,
               const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
               int step, float grad_scale, void* stream) {
    return adam_launch("vamd_adamw", false, p, g, m, v, n, lr, lr_dev, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}

int vamd_adam_l2(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                 const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                 int step, float grad_scale, void* stream) {
    return adam_launch("vamd_adam_l2", true, p, g, m, v, n, lr, lr_dev, beta1, beta2, eps, weight_decay, step, grad_scale, stream);
}

int vamd_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, const int64_t* step_dev,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, int l2_decay, void* stream) {
    VAMD_REQUIRE(lr_dev && step_dev, "vamd_adam_dev: null device scalar");
    return adam_launch("vamd_adam_dev", l2_decay != 0, p, g, m, v, n, 0.f, lr_dev, beta1, beta2, eps, weight_decay, 0, grad_scale,
                       stream, step_dev);
}

int vamd_repack_weight(const float* w, void* wf, void* wd, int Cout, int Cin, int ksize, int dtype,
                       void* stream) {
    VAMD_REQUIRE(w && wf && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3),
                 "vamd_repack_weight: bad argument");
    const int64_t total = (int64_t)Cout * Cin * ksize * ksize;
    VAMD_DISPATCH(dtype, "vamd_repack_weight", {
        hipLaunchKernelGGL(repack_weight_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0,
                           as_stream(stream), w, (T*)wf, (T*)wd, Cout, Cin, ksize);
    });
    VAMD_LAUNCH_OK("vamd_repack_weight");
    return 0;
}

int vamd_repack_weights(const vamd_repack_item* items_dev, int n, long long total_blocks, int dtype, void* stream) {
    VAMD_REQUIRE(items_dev && n > 0 && total_blocks > 0 && total_blocks < (1ll << 31),
                 "vamd_repack_weights: bad argument");
    VAMD_DISPATCH(dtype, "vamd_repack_weights", {
        hipLaunchKernelGGL(repack_weights_kernel<T>, dim3((unsigned)total_blocks), dim3(256), 0, as_stream(stream),
                           items_dev, n);
    });
    VAMD_LAUNCH_OK("vamd_repack_weights");
    return 0;
}

int vamd_repack_weight_up4(const float* w, void* w4, int Cout, int Cin, int dtype, void* stream) {
    VAMD_REQUIRE(w && w4 && Cout > 0 && Cin > 0, "vamd_repack_weight_up4: bad argument");
    const int64_t total = (int64_t)Cin * 16 * Cout;
    VAMD_DISPATCH(dtype, "vamd_repack_weight_up4", {
        hipLaunchKernelGGL(repack_weight_up4_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), w,
                           (T*)w4, Cout, Cin);
    });
    VAMD_LAUNCH_OK("vamd_repack_weight_up4");
    return 0;
}

int vamd_repack_weight_up4_fwd(const float* w, void* w4, int Cout, int Cin, int dtype, void* stream) {
    VAMD_REQUIRE(w && w4 && Cout > 0 && Cin > 0, "vamd_repack_weight_up4_fwd: bad argument");
    const int64_t total = (int64_t)16 * Cout * Cin;
    VAMD_DISPATCH(dtype, "vamd_repack_weight_up4_fwd", {
        hipLaunchKernelGGL(repack_weight_up4_fwd_kernel<T>, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), w,
                           (T*)w4, Cout, Cin);
    });
    VAMD_LAUNCH_OK("vamd_repack_weight_up4_fwd");
    return 0;
}

int vamd_unpack_wgrad(const float* dwk, float* dw, int Cout, int Cin, int ksize, void* stream) {
    VAMD_REQUIRE(dwk && dw && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3),
                 "vamd_unpack_wgrad: bad argument");
    const int64_t total = (int64_t)Cout * Cin * ksize * ksize;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                       as_stream(stream), dwk, dw, Cout, Cin, ksize);
    VAMD_LAUNCH_OK("vamd_unpack_wgrad");
    return 0;
}

}  // extern "C"
