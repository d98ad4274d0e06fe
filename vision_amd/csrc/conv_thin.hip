// Thin-channel convolutions: the layers of the UNet that touch the 3-channel image (first block,
// last block, End conv) at full resolution.  An MFMA tile would be >90 % padding there; these are
// HBM-bound direct kernels (stride 1, 3x3 pad 1 or 1x1) with fp32 accumulation, built around one
// rule: every global access of the WIDE tensor is a 16-byte piece per lane with consecutive lanes
// on consecutive addresses.
//   thin_in    few input channels (<= 4) -> any Cout.  Thread = pixel; weights broadcast from LDS;
//              the block's 256 x Cout outputs are staged in LDS and leave as one contiguous span.
//   thin_out   wide Cin -> few output channels (<= 4).  8 lanes share a pixel (16 bytes of channels
//              each), 4 pixels per thread, cross-lane shuffle reduction.
//   thin_wgrad weight gradient with one thin side: same 8-lanes-per-pixel mapping over the wide
//              tensor, one filter row of taps per workgroup (72-96 accumulators per thread), block
//              partials to a slab that a second kernel sums in a fixed order (deterministic).
// The input gradient of such a layer is the same computation with the flipped, transposed filter.
#include "conv_shapes.h"

namespace vamd {

constexpr int kSm = 4;          // channels on the thin side are padded to 4 in registers / LDS
constexpr int kCoTile = 16;

struct Pix {
    int b, h, w;
};
__device__ __forceinline__ Pix decode(int64_t m, int H, int W) {
    const int HW = H * W;
    Pix p;
    p.b = (int)(m / HW);
    const int rem = (int)(m - (int64_t)p.b * HW);
    p.h = rem / W;
    p.w = rem - p.h * W;
    return p;
}

// ---------------------------------------------------------------------------------- thin_in
constexpr int kCoChunk = 64;      // output channels accumulated in registers at a time

template <typename T, int K>
__global__ void __launch_bounds__(256)
thin_in_kernel(ConvGeom g, const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
               const float* __restrict__ temb, const T* __restrict__ res, const T* __restrict__ res2,
               T* __restrict__ y) {
    constexpr int taps = K * K;
    constexpr int VEC = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int nchunk = (g.Cout + kCoChunk - 1) / kCoChunk;
    // LDS: weights as floats [chunk][tap][ci(4)][co(64)] (every lane reads the same address:
    //      broadcast, 4 channels per ds_read_b128), then the output stage [256][Cout] in T
    float* ws = reinterpret_cast<float*>(smem_raw);
    unsigned char* stage = smem_raw + (size_t)nchunk * taps * kSm * kCoChunk * sizeof(float);
    // stage rows are padded by 16 bytes so that the 16-byte row writes of 8 consecutive lanes fall
    // on distinct banks; rows that are not a whole number of 16-byte pieces are kept dense
    const bool vec_rows = (g.Cout % VEC) == 0;
    const int row_bytes = g.Cout * (int)sizeof(T) + (vec_rows ? 16 : 0);
    for (int i = threadIdx.x; i < nchunk * taps * kSm * kCoChunk; i += 256) {
        const int co = i % kCoChunk, c = (i / kCoChunk) % kSm, t = (i / (kCoChunk * kSm)) % taps, ch = i / (kCoChunk * kSm * taps);
        const int oc = ch * kCoChunk + co;
        ws[i] = (oc < g.Cout && c < g.Cin) ? to_f(w[((int64_t)oc * taps + t) * g.Cin + c]) : 0.f;
    }
    __syncthreads();
    const int64_t m0 = (int64_t)blockIdx.x * 256;
    const int64_t m = m0 + threadIdx.x;
    if (m < g.M) {
        const Pix p = decode(m, g.H, g.W);
        for (int ch = 0; ch < nchunk; ++ch) {
            float acc[kCoChunk];
#pragma unroll
            for (int j = 0; j < kCoChunk; ++j) acc[j] = 0.f;
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const int hi = p.h + r - g.pad, wi = p.w + s - g.pad;
                    const bool ok = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                    const T* xp = x + (((int64_t)p.b * g.H + hi) * g.W + wi) * g.Cin;
#pragma unroll
                    for (int c = 0; c < kSm; ++c) {
                        if (c < g.Cin) {                  // wave-uniform
                            const float xv = ok ? to_f(xp[c]) : 0.f;
                            const float4* wv = reinterpret_cast<const float4*>(
                                ws + (((size_t)ch * taps + r * K + s) * kSm + c) * kCoChunk);
#pragma unroll
                            for (int j4 = 0; j4 < kCoChunk / 4; ++j4) {
                                if (ch * kCoChunk + j4 * 4 >= g.Cout) break;      // wave-uniform: few-channel outputs
                                const float4 q = wv[j4];
                                acc[j4 * 4 + 0] = fmaf(xv, q.x, acc[j4 * 4 + 0]);
                                acc[j4 * 4 + 1] = fmaf(xv, q.y, acc[j4 * 4 + 1]);
                                acc[j4 * 4 + 2] = fmaf(xv, q.z, acc[j4 * 4 + 2]);
                                acc[j4 * 4 + 3] = fmaf(xv, q.w, acc[j4 * 4 + 3]);
                            }
                        }
                    }
                }
            const int co0 = ch * kCoChunk;
            T* row = reinterpret_cast<T*>(stage + (size_t)threadIdx.x * row_bytes);
#pragma unroll
            for (int j = 0; j < kCoChunk; j += VEC) {
                if (co0 + j < g.Cout) {                   // wave-uniform
                    float v[VEC];
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int oc = co0 + j + e;
                        v[e] = acc[j + e];
                        if (oc < g.Cout) {
                            if (bias) v[e] += bias[oc];
                            if (temb) v[e] += temb[(int64_t)p.b * g.Cout + oc];
                            if (res) v[e] += to_f(res[m * g.Cout + oc]);
                            if (res2) v[e] += to_f(res2[m * g.Cout + oc]);
                        }
                    }
                    if (vec_rows) {
                        Vec16<T> ov;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) ov.v[e] = from_f<T>(v[e]);
                        *reinterpret_cast<Vec16<T>*>(row + co0 + j) = ov;
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e)
                            if (co0 + j + e < g.Cout) row[co0 + j + e] = from_f<T>(v[e]);
                    }
                }
            }
        }
    }
    __syncthreads();
    // the block's outputs are one contiguous span of global memory: [m0 .. m0+256) x Cout
    int64_t npix = g.M - m0;
    if (npix > 256) npix = 256;
    T* dst = y + m0 * g.Cout;
    if (vec_rows) {                                   // m0 * Cout * sizeof(T) is a multiple of 16 here
        const int ppr = g.Cout / VEC;                 // 16-byte pieces per row
        const int nvec = (int)npix * ppr;
        for (int i = threadIdx.x; i < nvec; i += 256) {
            const int pix = i / ppr, piece = i - pix * ppr;
            reinterpret_cast<Vec16<T>*>(dst)[i] =
                *reinterpret_cast<const Vec16<T>*>(stage + (size_t)pix * row_bytes + piece * 16);
        }
    } else {
        const T* src = reinterpret_cast<const T*>(stage);
        const int nelem = (int)npix * g.Cout;
        for (int i = threadIdx.x; i < nelem; i += 256) dst[i] = src[i];
    }
}

// --------------------------------------------------------------------------------- thin_out
// 8 lanes per pixel (lane c8 owns channel pieces c8, c8+8, ...), 4 pixels per thread, 32 pixel
// slots per block: a block covers 128 consecutive pixels.
template <typename T, int K>
__global__ void __launch_bounds__(256)
thin_out_kernel(ConvGeom g, const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                const float* __restrict__ temb, const T* __restrict__ res, const T* __restrict__ res2,
                T* __restrict__ y) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int taps = K * K;
    constexpr int PPT = 4;
    extern __shared__ __attribute__((aligned(16))) float wsm[];      // [tap][ci][4]
    for (int i = threadIdx.x; i < taps * g.Cin * kSm; i += 256) {
        const int co = i & 3, rest = i >> 2;                         // rest = tap*Cin + ci
        wsm[i] = co < g.Cout ? to_f(w[(int64_t)co * taps * g.Cin + rest]) : 0.f;
    }
    __syncthreads();
    const int c8 = threadIdx.x & 7, slot = threadIdx.x >> 3;         // 32 slots
    const int npiece = g.Cin / VEC;
    const int64_t m0 = (int64_t)blockIdx.x * (32 * PPT);
    Pix p[PPT];
    bool pv[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int64_t m = m0 + i * 32 + slot;
        pv[i] = m < g.M;
        p[i] = decode(pv[i] ? m : 0, g.H, g.W);
    }
    float acc[PPT][kSm];
#pragma unroll
    for (int i = 0; i < PPT; ++i)
#pragma unroll
        for (int c = 0; c < kSm; ++c) acc[i][c] = 0.f;
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int s = 0; s < K; ++s) {
            for (int pc = c8; pc < npiece; pc += 8) {
                Vec16<T> xv[PPT];
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
                    const int hi = p[i].h + r - g.pad, wi = p[i].w + s - g.pad;
                    const bool ok = pv[i] && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                    if (ok) xv[i] = *reinterpret_cast<const Vec16<T>*>(x + (((int64_t)p[i].b * g.H + hi) * g.W + wi) * g.Cin + pc * VEC);
                    else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) xv[i].v[e] = from_f<T>(0.f);
                    }
                }
                const float4* wp = reinterpret_cast<const float4*>(wsm) + (r * K + s) * g.Cin + pc * VEC;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float4 wv = wp[e];
#pragma unroll
                    for (int i = 0; i < PPT; ++i) {
                        const float xf = to_f(xv[i].v[e]);
                        acc[i][0] = fmaf(xf, wv.x, acc[i][0]);
                        acc[i][1] = fmaf(xf, wv.y, acc[i][1]);
                        acc[i][2] = fmaf(xf, wv.z, acc[i][2]);
                        acc[i][3] = fmaf(xf, wv.w, acc[i][3]);
                    }
                }
            }
        }
#pragma unroll
    for (int i = 0; i < PPT; ++i)
#pragma unroll
        for (int c = 0; c < kSm; ++c) {
            float v = acc[i][c];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            acc[i][c] = v;
        }
    if (c8 == 0) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            if (!pv[i]) continue;
            const int64_t m = m0 + i * 32 + slot;
#pragma unroll
            for (int co = 0; co < kSm; ++co) {
                if (co < g.Cout) {
                    float v = acc[i][co];
                    if (bias) v += bias[co];
                    if (temb) v += temb[(int64_t)p[i].b * g.Cout + co];
                    if (res) v += to_f(res[m * g.Cout + co]);
                    if (res2) v += to_f(res2[m * g.Cout + co]);
                    y[m * g.Cout + co] = from_f<T>(v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------- thin_wgrad
// Wide side W (dY when WIDE_IS_OUT, else X) has `wide` channels (multiple of 16 bytes), the other
// side has <= 4.  blockIdx.y selects the filter row r; a thread owns one 16-byte piece of wide
// channels (lane & 7, + 8 per round) and accumulates acc[s][e][c] = sum over its pixels of
// wide[e] * thin[c] for the K taps (r, s).  Slab layout per (block x, r): [round][s][e][c][c8].
template <typename T, int K, bool WIDE_IS_OUT>
__global__ void __launch_bounds__(256)
thin_wgrad_kernel(ConvGeom g, const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ slab,
                  int64_t pix_per_block) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NACC = K * VEC * kSm;
    __shared__ float red[4][NACC][8 + 1];
    const int c8 = threadIdx.x & 7, slot = threadIdx.x >> 3;         // 32 pixel slots, 8 per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.y;
    const int wide = WIDE_IS_OUT ? g.Cout : g.Cin;
    const int small = WIDE_IS_OUT ? g.Cin : g.Cout;
    const int nround = (wide / VEC + 7) / 8;
    const int64_t m0 = (int64_t)blockIdx.x * pix_per_block;
    int64_t m1 = m0 + pix_per_block;
    if (m1 > g.M) m1 = g.M;
    float* out = slab + ((int64_t)blockIdx.x * K + r) * nround * NACC * 8;
    for (int round = 0; round < nround; ++round) {
        const int pc = round * 8 + c8;
        const bool pok = pc * VEC < wide;
        float acc[K][VEC][kSm];
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int c = 0; c < kSm; ++c) acc[s][e][c] = 0.f;
#pragma unroll 2
        for (int64_t m = m0 + slot; m < m1; m += 32) {
            const Pix p = decode(m, g.H, g.W);
            const int hi = p.h + r - g.pad;
            const bool hok = (unsigned)hi < (unsigned)g.H;
            if (WIDE_IS_OUT) {
                Vec16<T> dv;
                if (pok) dv = *reinterpret_cast<const Vec16<T>*>(dy + m * g.Cout + pc * VEC);
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const int wi = p.w + s - g.pad;
                    const bool ok = pok && hok && (unsigned)wi < (unsigned)g.W;
                    const T* xp = x + (((int64_t)p.b * g.H + hi) * g.W + wi) * g.Cin;
                    float xs[kSm];
#pragma unroll
                    for (int c = 0; c < kSm; ++c) xs[c] = (ok && c < small) ? to_f(xp[c]) : 0.f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float d = pok ? to_f(dv.v[e]) : 0.f;
#pragma unroll
                        for (int c = 0; c < kSm; ++c) acc[s][e][c] = fmaf(d, xs[c], acc[s][e][c]);
                    }
                }
            } else {
                float ds[kSm];
#pragma unroll
                for (int c = 0; c < kSm; ++c) ds[c] = c < small ? to_f(dy[m * g.Cout + c]) : 0.f;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const int wi = p.w + s - g.pad;
                    const bool ok = pok && hok && (unsigned)wi < (unsigned)g.W;
                    if (ok) {
                        const Vec16<T> xv = *reinterpret_cast<const Vec16<T>*>(
                            x + (((int64_t)p.b * g.H + hi) * g.W + wi) * g.Cin + pc * VEC);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float xf = to_f(xv.v[e]);
#pragma unroll
                            for (int c = 0; c < kSm; ++c) acc[s][e][c] = fmaf(ds[c], xf, acc[s][e][c]);
                        }
                    }
                }
            }
        }
        // sum over the 8 pixel slots of the wave (lanes differing in bits 3..5), then over the 4 waves
#pragma unroll
        for (int s = 0; s < K; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e)
#pragma unroll
                for (int c = 0; c < kSm; ++c) {
                    float v = acc[s][e][c];
                    v += __shfl_xor(v, 8, 64);
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (lane < 8) red[wave][(s * VEC + e) * kSm + c][c8] = v;
                }
        __syncthreads();
        for (int i = threadIdx.x; i < NACC * 8; i += 256) {
            const int a = i >> 3, l = i & 7;
            out[((int64_t)round * NACC + a) * 8 + l] = (red[0][a][l] + red[1][a][l]) + (red[2][a][l] + red[3][a][l]);
        }
        __syncthreads();
    }
}

// sums the block slabs in a fixed order and scatters to dw[co][tap][ci]
template <typename T, int K, bool WIDE_IS_OUT>
__global__ void __launch_bounds__(256)
thin_wgrad_reduce_kernel(ConvGeom g, const float* __restrict__ slab, int nblocks, float* __restrict__ dw) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int NACC = K * VEC * kSm;
    __shared__ float red[4][64];
    const int wide = WIDE_IS_OUT ? g.Cout : g.Cin;
    const int small = WIDE_IS_OUT ? g.Cin : g.Cout;
    const int nround = (wide / VEC + 7) / 8;
    const int per_r = nround * NACC * 8;                       // entries per (block, filter row)
    const int total = K * per_r;
    const int e64 = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + e64;                     // consecutive threads -> consecutive entries
    float sum = 0.f;
    if (idx < total)
        for (int b = q; b < nblocks; b += 4) sum += slab[(int64_t)b * total + idx];
    red[q][e64] = sum;
    __syncthreads();
    if (q != 0 || idx >= total) return;
    sum = (red[0][e64] + red[1][e64]) + (red[2][e64] + red[3][e64]);
    const int r = idx / per_r;
    int rest = idx - r * per_r;
    const int l = rest & 7;
    rest >>= 3;
    const int a = rest % NACC, round = rest / NACC;
    const int c = a % kSm, e = (a / kSm) % VEC, s = a / (kSm * VEC);
    const int wc = (round * 8 + l) * VEC + e;
    if (wc >= wide || c >= small) return;
    const int co = WIDE_IS_OUT ? wc : c, ci = WIDE_IS_OUT ? c : wc;
    dw[((int64_t)co * K * K + r * K + s) * g.Cin + ci] = sum;
}

// both sides thin (3 -> 3): thread = (pixel lane, wide channel), all taps in registers
template <typename T, int K>
__global__ void __launch_bounds__(256)
tiny_wgrad_kernel(ConvGeom g, const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ slab,
                  int64_t pix_per_block) {
    constexpr int NACC = K * K * kSm;
    __shared__ float red[64][NACC][kSm + 1];
    const int wl = threadIdx.x & 3, pl = threadIdx.x >> 2;      // wl: output channel, 64 pixel lanes
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
    const int64_t m0 = (int64_t)blockIdx.x * pix_per_block;
    int64_t m1 = m0 + pix_per_block;
    if (m1 > g.M) m1 = g.M;
    for (int64_t m = m0 + pl; m < m1; m += 64) {
        const Pix p = decode(m, g.H, g.W);
        const float d = wl < g.Cout ? to_f(dy[m * g.Cout + wl]) : 0.f;
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int s = 0; s < K; ++s) {
                const int hi = p.h + r - g.pad, wi = p.w + s - g.pad;
                const bool ok = (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
                const T* xp = x + (((int64_t)p.b * g.H + hi) * g.W + wi) * g.Cin;
#pragma unroll
                for (int c = 0; c < kSm; ++c) {
                    const float xv = (ok && c < g.Cin) ? to_f(xp[c]) : 0.f;
                    acc[(r * K + s) * kSm + c] = fmaf(d, xv, acc[(r * K + s) * kSm + c]);
                }
            }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) red[pl][i][wl] = acc[i];
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * NACC * kSm;
    for (int i = threadIdx.x; i < NACC * kSm; i += 256) {
        const int a = i >> 2, l = i & 3;
        float sum = 0.f;
        for (int pq = 0; pq < 64; ++pq) sum += red[pq][a][l];
        out[i] = sum;
    }
}

template <int K>
__global__ void tiny_wgrad_reduce_kernel(ConvGeom g, const float* __restrict__ slab, int nblocks,
                                         float* __restrict__ dw) {
    constexpr int NACC = K * K * kSm;
    const int idx = blockIdx.x;                                   // one wave per entry
    double sum = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) sum += (double)slab[(int64_t)b * NACC * kSm + idx];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (threadIdx.x != 0) return;
    const int co = idx & 3, a = idx >> 2, c = a % kSm, tap = a / kSm;
    if (co < g.Cout && c < g.Cin) dw[((int64_t)co * K * K + tap) * g.Cin + c] = (float)sum;
}


// ------------------------------------------------------------------------------------ tiny33
// Both sides are the 3-channel image (3 -> 3, the second conv of the last block and the End conv):
// 6 bytes per pixel, so a thread takes 8 consecutive pixels of one image row (48 bytes in bf16 =
// three 16-byte pieces, W % 8 == 0) plus the pixel left and right of them for a 3x3 filter, and
// keeps the whole 3 x K x K x 3 filter in registers.  Grid-stride over the 8-pixel groups.
template <typename T, int K>
__device__ __forceinline__ void tiny33_load_row(const T* __restrict__ base, int64_t pix0, int w0, int W, bool ok,
                                                float (&v)[8 + K - 1][3]) {
    constexpr int VEC = Vec16<T>::N, NV = 24 / VEC, HALO = K / 2;
#pragma unroll
    for (int i = 0; i < 8 + K - 1; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[i][c] = 0.f;
    if (!ok) return;
    const T* p = base + pix0 * 3;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const Vec16<T> q = *reinterpret_cast<const Vec16<T>*>(p + i * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[HALO + (i * VEC + e) / 3][(i * VEC + e) % 3] = to_f(q.v[e]);
    }
    if (K == 3) {
        if (w0 > 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[0][c] = to_f(p[c - 3]);
        }
        if (w0 + 8 < W) {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[8 + K - 2][c] = to_f(p[24 + c]);
        }
    }
}

template <typename T, int K>
__global__ void __launch_bounds__(256)
tiny33_fwd_kernel(ConvGeom g, const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                  const float* __restrict__ temb, const T* __restrict__ res, const T* __restrict__ res2,
                  T* __restrict__ y) {
    constexpr int VEC = Vec16<T>::N, NV = 24 / VEC, NW = 3 * K * K * 3;
    __shared__ float ws[NW + 3];
    for (int i = threadIdx.x; i < NW + 3; i += 256) ws[i] = i < NW ? to_f(w[i]) : (bias ? bias[i - NW] : 0.f);
    __syncthreads();
    float bo[3];                       // the filter stays in LDS: every lane reads the same word (broadcast)
#pragma unroll
    for (int o = 0; o < 3; ++o) bo[o] = ws[NW + o];
    const int64_t groups = g.M / 8;
    const int wg = g.W / 8;
    for (int64_t grp = blockIdx.x * 256ll + threadIdx.x; grp < groups; grp += (int64_t)gridDim.x * 256) {
        const int w0 = (int)(grp % wg) * 8;
        const int64_t rowi = grp / wg;                     // b * H + h
        const int h = (int)(rowi % g.H);
        float acc[8][3];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int o = 0; o < 3; ++o) acc[p][o] = bo[o];
#pragma unroll 1                       // one filter row at a time: 30 inputs + 27 filter words live, no spills
        for (int r = 0; r < K; ++r) {
            const int hi = h + r - K / 2;
            float v[8 + K - 1][3];
            tiny33_load_row<T, K>(x, (rowi + r - K / 2) * g.W + w0, w0, g.W, (unsigned)hi < (unsigned)g.H, v);
#pragma unroll
            for (int sx = 0; sx < K; ++sx)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        const float wv = ws[((o * K + r) * K + sx) * 3 + c];
#pragma unroll
                        for (int p = 0; p < 8; ++p) acc[p][o] = fmaf(v[p + sx][c], wv, acc[p][o]);
                    }
        }
        const int64_t e0 = grp * 24;
        if (temb) {
            const int64_t b = rowi / g.H;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float tv = temb[b * 3 + o];
#pragma unroll
                for (int p = 0; p < 8; ++p) acc[p][o] += tv;
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            Vec16<T> out;
            Vec16<T> r1, r2;
            if (res) r1 = *reinterpret_cast<const Vec16<T>*>(res + e0 + i * VEC);
            if (res2) r2 = *reinterpret_cast<const Vec16<T>*>(res2 + e0 + i * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float a = acc[(i * VEC + e) / 3][(i * VEC + e) % 3];
                if (res) a += to_f(r1.v[e]);
                if (res2) a += to_f(r2.v[e]);
                out.v[e] = from_f<T>(a);
            }
            *reinterpret_cast<Vec16<T>*>(y + e0 + i * VEC) = out;
        }
    }
}

// weight gradient of the same layer: dw[o][r][s][c] = sum_p dy[p][o] * x[p + (r, s)][c]; every thread
// keeps all 9 * K * K sums, a block writes one partial row to the slab.
template <typename T, int K>
__global__ void __launch_bounds__(256)
tiny33_wgrad_kernel(ConvGeom g, const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ slab) {
    constexpr int VEC = Vec16<T>::N, NV = 24 / VEC, NW = 3 * K * K * 3;
    __shared__ float red[4][NW];
    float acc[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) acc[i] = 0.f;
    const int64_t groups = g.M / 8;
    const int wg = g.W / 8;
    for (int64_t grp = blockIdx.x * 256ll + threadIdx.x; grp < groups; grp += (int64_t)gridDim.x * 256) {
        const int w0 = (int)(grp % wg) * 8;
        const int64_t rowi = grp / wg;
        const int h = (int)(rowi % g.H);
        float d[8][3];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const Vec16<T> q = *reinterpret_cast<const Vec16<T>*>(dy + grp * 24 + i * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) d[(i * VEC + e) / 3][(i * VEC + e) % 3] = to_f(q.v[e]);
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int hi = h + r - K / 2;
            float v[8 + K - 1][3];
            tiny33_load_row<T, K>(x, (rowi + r - K / 2) * g.W + w0, w0, g.W, (unsigned)hi < (unsigned)g.H, v);
#pragma unroll
            for (int sx = 0; sx < K; ++sx)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int o = 0; o < 3; ++o) {
                        float a = acc[((o * K + r) * K + sx) * 3 + c];
#pragma unroll
                        for (int p = 0; p < 8; ++p) a = fmaf(d[p][o], v[p + sx][c], a);
                        acc[((o * K + r) * K + sx) * 3 + c] = a;
                    }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const float sum = wave_sum(acc[i]);
        if (lane == 0) red[wv][i] = sum;
    }
    __syncthreads();
    if (threadIdx.x < NW)
        slab[(int64_t)blockIdx.x * NW + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

template <int NW>
__global__ void tiny33_wgrad_reduce_kernel(const float* __restrict__ slab, int nblocks, float* __restrict__ dw) {
    double sum = 0.0;                                             // one wave per filter entry
    for (int b = threadIdx.x; b < nblocks; b += 64) sum += (double)slab[(int64_t)b * NW + blockIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (threadIdx.x == 0) dw[blockIdx.x] = (float)sum;
}

static bool tiny33_ok(const ConvGeom& g) {
    return g.Cin == 3 && g.Cout == 3 && g.W % 8 == 0 && g.stride == 1 && !g.up && (g.k == 1 || g.k == 3) &&
           g.pad == g.k / 2 && g.Ho == g.H && g.Wo == g.W;
}
static int tiny33_blocks(const ConvGeom& g) {
    int64_t n = (g.M / 8 + 255) / 256;
    if (n > 1024) n = 1024;
    return (int)(n < 1 ? 1 : n);
}

// ------------------------------------------------------------------------------------ host
static size_t thin_in_lds(const ConvGeom& g, int dtype) {
    const int nchunk = (g.Cout + kCoChunk - 1) / kCoChunk;
    const int esz = dtype == VAMD_BF16 ? 2 : 4;
    return (size_t)nchunk * g.k * g.k * kSm * kCoChunk * sizeof(float) + (size_t)256 * (g.Cout * esz + 16);
}

template <typename Kern> static int raise_lds_limit(Kern kernel, size_t bytes, const char* who) {
    if (bytes <= 64 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%zu) failed: %s", who, bytes, hipGetErrorString(e));
    return 0;
}

bool conv_thin_supported(const ConvGeom& g, int dtype) {
    const int vec = dtype == VAMD_BF16 ? 8 : 4;
    if (g.stride != 1 || g.up) return false;
    if (g.Cin <= kSm) return thin_in_lds(g, dtype) <= 150 * 1024;
    return g.Cout <= kSm && g.Cin % vec == 0 && (size_t)g.k * g.k * g.Cin * kSm * sizeof(float) <= 64 * 1024;
}

int conv_fwd_thin(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                  const float* temb, const void* res, const void* res2, void* y, hipStream_t st) {
    if (dtype == VAMD_BF16) {
        if (g.Cin <= kSm && thin_in_mfma_ok(g)) return thin_in_mfma(g, x, w, bias, temb, res, res2, y, st);
        if (g.Cin > kSm && thin_out_mfma_ok(g)) return thin_out_mfma(g, x, w, bias, temb, res, res2, y, st);
    }
    if (tiny33_ok(g)) {
        VAMD_DISPATCH(dtype, "conv_fwd_thin", {
            const dim3 grid((unsigned)((g.M / 8 + 255) / 256 > 4096 ? 4096 : (g.M / 8 + 255) / 256));
            if (g.k == 3)
                hipLaunchKernelGGL((tiny33_fwd_kernel<T, 3>), grid, dim3(256), 0, st, g, (const T*)x, (const T*)w, bias, temb,
                                   (const T*)res, (const T*)res2, (T*)y);
            else
                hipLaunchKernelGGL((tiny33_fwd_kernel<T, 1>), grid, dim3(256), 0, st, g, (const T*)x, (const T*)w, bias, temb,
                                   (const T*)res, (const T*)res2, (T*)y);
        });
        VAMD_LAUNCH_OK("tiny33_fwd");
        return 0;
    }
    VAMD_DISPATCH(dtype, "conv_fwd_thin", {
        if (g.Cin <= kSm) {
            const dim3 grid((unsigned)((g.M + 255) / 256));
            const size_t lds = thin_in_lds(g, dtype);
            int rc = g.k == 3 ? raise_lds_limit(thin_in_kernel<T, 3>, lds, "thin_in") : raise_lds_limit(thin_in_kernel<T, 1>, lds, "thin_in");
            if (rc) return rc;
            if (g.k == 3)
                hipLaunchKernelGGL((thin_in_kernel<T, 3>), grid, dim3(256), lds, st, g, (const T*)x, (const T*)w, bias, temb,
                                   (const T*)res, (const T*)res2, (T*)y);
            else
                hipLaunchKernelGGL((thin_in_kernel<T, 1>), grid, dim3(256), lds, st, g, (const T*)x, (const T*)w, bias, temb,
                                   (const T*)res, (const T*)res2, (T*)y);
        } else {
            const dim3 grid((unsigned)((g.M + 127) / 128));
            const size_t lds = (size_t)g.k * g.k * g.Cin * kSm * sizeof(float);
            if (g.k == 3)
                hipLaunchKernelGGL((thin_out_kernel<T, 3>), grid, dim3(256), lds, st, g, (const T*)x, (const T*)w, bias,
                                   temb, (const T*)res, (const T*)res2, (T*)y);
            else
                hipLaunchKernelGGL((thin_out_kernel<T, 1>), grid, dim3(256), lds, st, g, (const T*)x, (const T*)w, bias,
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
    int64_t n = (g.M + 1023) / 1024;
    if (n > 1024) n = 1024;
    if (n < 1) n = 1;
    return (int)n;
}

static bool both_thin(const ConvGeom& g) { return g.Cin <= kSm && g.Cout <= kSm; }

bool conv_wgrad_thin_supported(const ConvGeom& g, int dtype) {
    const int vec = dtype == VAMD_BF16 ? 8 : 4;
    if (g.stride != 1 || g.up) return false;
    if (both_thin(g)) return true;
    if (g.Cin <= kSm) return g.Cout % vec == 0;
    return g.Cout <= kSm && g.Cin % vec == 0;
}

size_t conv_wgrad_thin_workspace(const ConvGeom& g, int dtype) {
    if (dtype == VAMD_BF16 && thin_wgrad_mfma_ok(g)) return thin_wgrad_mfma_workspace(g);
    const int nb = thin_wgrad_blocks(g);
    if (both_thin(g)) return (size_t)nb * g.k * g.k * kSm * kSm * sizeof(float);
    const int vec = dtype == VAMD_BF16 ? 8 : 4;
    const int wide = g.Cin <= kSm ? g.Cout : g.Cin;
    const int nround = (wide / vec + 7) / 8;
    return (size_t)nb * g.k * nround * (g.k * vec * kSm) * 8 * sizeof(float);
}

template <typename T, int K, bool WO>
static void launch_thin_wgrad(const ConvGeom& g, const void* dy, const void* x, float* dw, float* slab, hipStream_t st) {
    constexpr int VEC = Vec16<T>::N;
    const int nb = thin_wgrad_blocks(g);
    const int64_t per = (g.M + nb - 1) / nb;
    hipLaunchKernelGGL((thin_wgrad_kernel<T, K, WO>), dim3(nb, K), dim3(256), 0, st, g, (const T*)dy, (const T*)x, slab, per);
    const int wide = WO ? g.Cout : g.Cin;
    const int total = K * ((wide / VEC + 7) / 8) * (K * VEC * kSm) * 8;
    hipLaunchKernelGGL((thin_wgrad_reduce_kernel<T, K, WO>), dim3((total + 63) / 64), dim3(256), 0, st, g, slab, nb, dw);
}

template <typename T, int K>
static void launch_tiny_wgrad(const ConvGeom& g, const void* dy, const void* x, float* dw, float* slab, hipStream_t st) {
    const int nb = thin_wgrad_blocks(g);
    const int64_t per = (g.M + nb - 1) / nb;
    hipLaunchKernelGGL((tiny_wgrad_kernel<T, K>), dim3(nb), dim3(256), 0, st, g, (const T*)dy, (const T*)x, slab, per);
    hipLaunchKernelGGL((tiny_wgrad_reduce_kernel<K>), dim3(K * K * kSm * kSm), dim3(64), 0, st, g, slab, nb, dw);
}

int conv_wgrad_thin(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw, void* workspace,
                    size_t workspace_bytes, hipStream_t st) {
    if (!workspace || workspace_bytes < conv_wgrad_thin_workspace(g, dtype))
        return fail(VAMD_EINVAL, "conv_wgrad_thin: needs %zu workspace bytes", conv_wgrad_thin_workspace(g, dtype));
    if (dtype == VAMD_BF16 && thin_wgrad_mfma_ok(g)) return thin_wgrad_mfma(g, dy, x, dw, workspace, st);
    float* slab = reinterpret_cast<float*>(workspace);
    if (tiny33_ok(g)) {
        const int nb = tiny33_blocks(g);
        VAMD_DISPATCH(dtype, "conv_wgrad_thin", {
            if (g.k == 3) {
                hipLaunchKernelGGL((tiny33_wgrad_kernel<T, 3>), dim3(nb), dim3(256), 0, st, g, (const T*)dy, (const T*)x, slab);
                hipLaunchKernelGGL((tiny33_wgrad_reduce_kernel<81>), dim3(81), dim3(64), 0, st, slab, nb, dw);
            } else {
                hipLaunchKernelGGL((tiny33_wgrad_kernel<T, 1>), dim3(nb), dim3(256), 0, st, g, (const T*)dy, (const T*)x, slab);
                hipLaunchKernelGGL((tiny33_wgrad_reduce_kernel<9>), dim3(9), dim3(64), 0, st, slab, nb, dw);
            }
        });
        VAMD_LAUNCH_OK("tiny33_wgrad");
        return 0;
    }
    const bool wide_out = g.Cin <= kSm;
    VAMD_DISPATCH(dtype, "conv_wgrad_thin", {
        if (g.k == 3) {
            if (both_thin(g)) launch_tiny_wgrad<T, 3>(g, dy, x, dw, slab, st);
            else if (wide_out) launch_thin_wgrad<T, 3, true>(g, dy, x, dw, slab, st);
            else launch_thin_wgrad<T, 3, false>(g, dy, x, dw, slab, st);
        } else {
            if (both_thin(g)) launch_tiny_wgrad<T, 1>(g, dy, x, dw, slab, st);
            else if (wide_out) launch_thin_wgrad<T, 1, true>(g, dy, x, dw, slab, st);
            else launch_thin_wgrad<T, 1, false>(g, dy, x, dw, slab, st);
        }
    });
    VAMD_LAUNCH_OK("conv_wgrad_thin");
    return 0;
}

}  // namespace vamd
