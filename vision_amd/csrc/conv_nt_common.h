// Device-side pieces shared by the implicit-GEMM convolution kernels: how a GEMM row and a filter tap
// address the stored input (Gather), the fused epilogue, direct-to-LDS loads, tile order.
#pragma once
#include <stdlib.h>
#include <type_traits>

#include "conv_shapes.h"

#ifndef VAMD_ABLATE
#define VAMD_ABLATE 0
#endif

namespace vamd {


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;


#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

struct FastDiv {
    uint32_t magic, shift;
};
static FastDiv make_fastdiv(uint32_t d) {       // exact for numerators < 2^26
    uint32_t l = 0;
    while ((1u << l) < d) ++l;
    FastDiv f;
    f.shift = 26 + l;
    f.magic = (uint32_t)(((1ull << f.shift) + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t m, FastDiv d) {
    return (uint32_t)(((uint64_t)m * d.magic) >> d.shift);
}

// How GEMM row m (an output pixel) and a filter tap address the stored input tensor.
struct Gather {
    int Hin, Win, Cin;     // stored input extents and channels
    int ldx;               // elements between consecutive stored pixels (Cin unless rows are padded)
    int He, We;            // valid range of the tap coordinate u = ho*stride + r - pad
    int Ho, Wo;            // extents that m enumerates
    int k, stride, pad;    // pad: rows; padw: columns (equal except for the classes of an upsampling forward)
    int padw;
    int shift;             // 1: stored row = u >> 1 (nearest 2x upsample folded into the gather)
    int M, Ktot;           // rows, and (tap rows)*(tap columns)*Cin
    FastDiv div_howo, div_wo;
    // Sub-filter views (stride-2 input gradient, one launch per output parity class):
    int kr, ks;            // tap rows and columns (k, k for a square filter); Ktot = kr * ks * Cin
    int wr0, ws0, wstep, wk;   // tap (tr, ts) reads filter tap (wr0 + wstep*tr, ws0 + wstep*ts) of a wk x wk filter
    int oH, oW, ostep, oph, opw;   // ostep = 2: row (b, ho, wo) is written to pixel (b, 2*ho + oph, 2*wo + opw) of an oH x oW image    // Second source (gather mode 2): after the kr*ks taps the reduction runs on over Cin2 channels of the
    // pixel itself in x2 [M][Cin2] against w2 [N][Cin2] - a 1x1 convolution of another tensor summed into
    // the same accumulators (the shortcut of a ResNetBlock inside its conv2).  Ktot includes Cin2.
    const void* x2;
    const void* w2;
    int Cin2;
};

struct Epilogue {
    const float* bias;     // [N]
    const float* temb;     // [B][N]
    const void* res;       // [M][N] in T
    const void* res2;
    float alpha;
    int out_f32;           // store fp32 instead of T
    float* stats;          // [row tiles][N][2]: per row tile, column sums of the stored y and y*y (or NULL)
    unsigned char* scratch;   // caller's scratch buffer for partial sums of split tiles (or NULL)
    size_t scratch_bytes;
    // Eval-mode BatchNorm + LeakyReLU of the consumer folded into this launch (act_scale != NULL): the stored value
    // becomes round(leaky(fma(q, act_scale[n], act_shift[n]))) with q the rounded convolution output -- bit for bit
    // what a separate normalise pass over the stored q would write -- and q itself goes to `raw` if that is given.
    const float* act_scale;
    const float* act_shift;
    float act_slope;
    void* raw;
    // BatchNorm + LeakyReLU of the INPUT applied by the kernel itself (pre_scale != NULL; resident-filter kernel only):
    // x is the tensor before the normalisation and every staged element becomes round(leaky(fma(x, pre_scale[c],
    // pre_shift[c]))) in LDS -- bit for bit what a separate normalise pass would have stored -- while padding stays zero.
    const float* pre_scale;
    const float* pre_shift;
    float pre_slope;
    // Set by the launch code: the transposing epilogue's whole-line stores are non-temporal.  That keeps a large output from
    // displacing the kernel's own input in L2, but a small one (which the next kernel would have found in the caches) is
    // then read back from memory.
    int nt_store;
};

// Up to four convolutions of one layer in one launch (the output-parity classes of a stride-2 input
// gradient): the grid is cut into per-class block ranges, a workgroup takes the gather, the tile
// count, the filter offset and the statistics offset of its class and then runs as usual.
struct ClassSet {
    int ncls;                  // <= 1: a plain launch, the kernel's own arguments apply
    int first_block[5];        // class c owns blocks [first_block[c], first_block[c+1])
    int m_tiles[4];
    int64_t w_off[4];          // elements added to the filter pointer
    int64_t stats_off[4];      // floats added to the statistics pointer
    int64_t y_off[4];          // output elements added to every output offset (partial products of a split reduction)
    Gather g[4];
};

// Branch-free pointer select: keeps the staging code of a K-step in one basic block, so that the
// scheduler may place it between MFMAs.
template <typename T> __device__ __forceinline__ const T* pick_ptr(bool ok, const T* a, const T* b) {
    const uint64_t m = 0ull - (uint64_t)ok;
    return reinterpret_cast<const T*>(((uint64_t)a & m) | ((uint64_t)b & ~m));
}

// Direct-to-LDS 16-byte load through a buffer resource: the base pointer lives in scalar registers,
// a lane supplies a 32-bit byte offset, and an offset at or beyond the resource's size (kNoLoad)
// makes the hardware write zeros - padding taps and rows past the tensor cost one select, no
// 64-bit pointer arithmetic.  Tensors are < 2 GiB (checked on the host).
constexpr uint32_t kNoLoad = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFF0, 0x00020000);
}
__device__ __forceinline__ void load16_lds(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, bool ok, unsigned char* lds) {
#if VAMD_ABLATE == 9        // every load hits the same 1 KiB: the address arithmetic stays, the memory traffic goes
    byte_off = (byte_off & 0x3f0u);
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LPTR(lds), 16, (int)(ok ? byte_off : kNoLoad), 0, 0, 0);
}

// BatchNorm + LeakyReLU of eight bf16 values held as four 32-bit words, in place: round(leaky(fma(x, sc, sh))) per element,
// bit for bit what scale_shift_act_kernel stores for slopes in [0, 1] (max(z, slope * z) == z > 0 ? z : slope * z there).
// Packed fp32 multiply-adds and a two-operand conversion: 32 vector instructions per 16-byte piece.  keep = 0 clears the
// piece (a padding pixel, whose zeros must survive the affine map), keep = ~0 leaves it.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__device__ __forceinline__ u32x4_t norm_act_piece(u32x4_t v, const f32x2_t (&sc)[4], const f32x2_t (&sh)[4], float slope,
                                                   uint32_t keep) {
    const f32x2_t sl = {slope, slope};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x2_t x = {__builtin_bit_cast(float, v[e] << 16), __builtin_bit_cast(float, v[e] & 0xffff0000u)};
        const f32x2_t z = __builtin_elementwise_fma(x, sc[e], sh[e]);
        const f32x2_t zs = z * sl;
        const f32x2_t m = {fmaxf(z[0], zs[0]), fmaxf(z[1], zs[1])};
        v[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(m, bf16x2_t)) & keep;
    }
    return v;
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<bf16_t> { static constexpr int PIECE = 8; typedef bf16x8_t Frag; };   // elements per 16 bytes
template <> struct ElemTraits<float> { static constexpr int PIECE = 4; typedef f32x4_t Frag; };

constexpr int kBM = 128;
constexpr int kRowBytes = 128;                   // reduction bytes per K-step

__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous id range
    const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// Epilogue of the persistent kernels for the four pixel tiles a lane holds (rows m_first + 16*i, channels
// nb .. nb + 4*NT - 1): y = alpha*acc + temb[image] + res + res2, rounded to T, stored, and (STATS) added
// to the lane's per-channel sums.  Every load of a term is issued before the first use of any of them
// and before any store: written row by row, the loads of a row could not move above the previous
// row's store (the pointers may alias as far as the compiler knows), so a tile paid one memory
// latency per 16-byte piece instead of one per term.  The sums land in the accumulators, which are
// re-initialised for the next tile anyway; the order of the additions is temb, res, res2.
// Core: the four rows are given by the element offset of the lane's first channel (offs), the image
// each belongs to (bimg, for the time embedding; ascending) and a validity flag.
// BUF: the outputs (not the fp32 and fused-normalise forms) leave through buffer stores with 32-bit offsets from yout
// (outputs below 2 GiB), a row past the tensor or a piece past the last channel as a store at an offset the resource's range
// check drops: every lane issues exactly 4 * NP stores, so a caller can count them in a later s_waitcnt vmcnt.
template <typename T, int NT, bool STATS, int C0 = 0, int CW = 4 * NT, bool ACT = false, bool BUF = false>
__device__ __forceinline__ void epilogue_rows_at(f32x4_t (&acc)[NT][4], float* s1, float* s2,
                                                 const float* temb, const void* res, const void* res2, float alpha,
                                                 int out_f32, void* yout, int N, int ldc, const int64_t (&offs)[4],
                                                 const uint32_t (&bimg)[4], const bool (&valid)[4], int nb,
                                                 const float* asc = nullptr, const float* ash = nullptr,
                                                 float aslope = 0.f, void* raw = nullptr, uint32_t xp_base = 0,
                                                 uint32_t xp_stride = 0, int nt_store = 0) {
    static_assert(C0 % 8 == 0 && CW % 8 == 0 && C0 + CW <= 4 * NT, "whole 16-byte pieces of the lane's channels");
    constexpr int CH = 16 / (int)sizeof(T);               // channels per 16-byte piece of T: 8 (bf16) or 4
    constexpr int NP = CW / CH;                           // pieces per row and lane
    constexpr int J0 = C0 / 4, NJ = CW / 4;               // accumulator (channel tile) range
    if (out_f32) {                                     // fp32 outputs of a bf16 product (attention scores)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!valid[i]) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = nb + 4 * j;
                // a trailing partial group is written whole: its extra columns are exact zeros
                // (their filter rows were never loaded) and ldc leaves room for them
                if (n < N && n + 4 <= ldc)
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(yout) + offs[i] + 4 * j) =
                        make_float4(acc[J0 + j][i][0] * alpha, acc[J0 + j][i][1] * alpha, acc[J0 + j][i][2] * alpha,
                                    acc[J0 + j][i][3] * alpha);
            }
        }
        return;
    }
    bool piece_ok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) piece_ok[p] = nb + p * CH < N && nb + p * CH + CH <= ldc;
    const T* r1 = reinterpret_cast<const T*>(res ? res : res2);
    const T* r2 = reinterpret_cast<const T*>(res ? res2 : nullptr);
    // ---- issue: time embedding of the first row's image, first residual
    float tv[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) tv[e] = 0.f;
    if (temb && valid[0]) {                            // rows ascend: an invalid first row means no valid row
        const float* tp = temb + (int64_t)bimg[0] * N + nb;
#pragma unroll
        for (int e = 0; e < CW; e += 4) {
            if (nb + e + 4 <= N) {
                const float4 t4 = *reinterpret_cast<const float4*>(tp + e);
                tv[e] = t4.x; tv[e + 1] = t4.y; tv[e + 2] = t4.z; tv[e + 3] = t4.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) tv[e + r] = nb + e + r < N ? tp[e + r] : 0.f;
            }
        }
    }
    Vec16<T> rv[4][NP];
    if (r1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (valid[i] && piece_ok[p]) rv[i][p] = *reinterpret_cast<const Vec16<T>*>(r1 + offs[i] + p * CH);
    }
    // ---- use
    if (alpha != 1.f) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[J0 + j][i] *= alpha;
    }
    if (temb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (bimg[i] == bimg[0]) {
#pragma unroll
                for (int e = 0; e < CW; ++e) acc[J0 + (e >> 2)][i][e & 3] += tv[e];
            } else if (valid[i]) {                        // the tile crosses into the next image: rare
                const float* tp = temb + (int64_t)bimg[i] * N + nb;
#pragma unroll
                for (int e = 0; e < CW; ++e)
                    if (nb + e < N) acc[J0 + (e >> 2)][i][e & 3] += tp[e];
            }
        }
    }
    if (r1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (valid[i] && piece_ok[p]) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) acc[J0 + ((p * CH + e) >> 2)][i][e & 3] += to_f(rv[i][p].v[e]);
                }
    }
    if (r2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (valid[i] && piece_ok[p]) rv[i][p] = *reinterpret_cast<const Vec16<T>*>(r2 + offs[i] + p * CH);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                if (valid[i] && piece_ok[p]) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) acc[J0 + ((p * CH + e) >> 2)][i][e & 3] += to_f(rv[i][p].v[e]);
                }
    }
    if constexpr (ACT) {                                  // piece by piece, so that only one piece's coefficients are live
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (!piece_ok[p]) continue;
            float sc[CH], sh[CH];
#pragma unroll
            for (int e = 0; e < CH; e += 4) {
                const float4 a4 = *reinterpret_cast<const float4*>(asc + nb + p * CH + e);
                const float4 b4 = *reinterpret_cast<const float4*>(ash + nb + p * CH + e);
                sc[e] = a4.x; sc[e + 1] = a4.y; sc[e + 2] = a4.z; sc[e + 3] = a4.w;
                sh[e] = b4.x; sh[e + 1] = b4.y; sh[e + 2] = b4.z; sh[e + 3] = b4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!valid[i]) continue;
                Vec16<T> ov;
#pragma unroll
                for (int e = 0; e < CH; ++e) ov.v[e] = from_f<T>(acc[J0 + ((p * CH + e) >> 2)][i][e & 3]);
                if (raw) *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(raw) + offs[i] + p * CH) = ov;
#pragma unroll
                for (int e = 0; e < CH; ++e) ov.v[e] = from_f<T>(leaky(fmaf(to_f(ov.v[e]), sc[e], sh[e]), aslope));
                *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(yout) + offs[i] + p * CH) = ov;
            }
        }
        return;
    }
    if constexpr (BUF) {
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(yout);
        // bf16, 16 channels per lane: the four lanes 16 g + pixel hold the 64-channel run of one pixel (128 bytes, a cache
        // line where rows are whole lines).  Written from where the MFMA left them, a store instruction touches 16 lines,
        // 64 scattered bytes of each, and the output displaces the kernel's input from L2 on its way to memory.  Instead a
        // row group's pieces pass through 2 KB of LDS owned by the wave (xp_base: two 1 KB regions xp_stride apart) and come
        // back transposed: lane L stores 16-byte chunk L & 7 of pixel 8 h + (L >> 3), so that every store writes eight whole
        // lines, and the stores are non-temporal.  LDS row = pixel, 128 bytes, chunk index XOR ((pixel >> 1) & 7): writes
        // and reads are conflict-free.  By inline asm: behind an LDS access the compiler knows of it waits for every pending
        // direct-to-LDS load.  LDS operations of one wave execute in order, so only the read results are waited for.
        // 128-channel layer at 112 x 112 x 64: 334 -> 310 us with these addresses, of which 294 without any store.
        constexpr bool XP = sizeof(T) == 2 && NP == 2;
        if constexpr (XP) {
            const int ln = (int)(threadIdx.x & 63);
            const int pix = ln & 15, grp = ln >> 4;
            const uint32_t wrow = xp_base + (uint32_t)(pix >> 3) * xp_stride + (uint32_t)(pix & 7) * 128u;
            const uint32_t wa0 = wrow + (uint32_t)(((2 * grp) ^ ((pix >> 1) & 7)) << 4);
            const uint32_t wa1 = wrow + (uint32_t)(((2 * grp + 1) ^ ((pix >> 1) & 7)) << 4);
            const uint32_t ra0 = xp_base + (uint32_t)(ln >> 3) * 128u + (uint32_t)(((ln & 7) ^ ((ln >> 4) & 7)) << 4);
            const uint32_t ra1 = xp_base + xp_stride + (uint32_t)(ln >> 3) * 128u + (uint32_t)(((ln & 7) ^ ((4 + (ln >> 4)) & 7)) << 4);
            const int run0 = nb - grp * 16;                               // first channel of the wave's run
            const int cb = run0 + (ln & 7) * 8;
            const bool chunk_ok = cb < N && cb + 8 <= ldc;
            const bool lines = nt_store && (ldc & 63) == 0 && (run0 & 63) == 0 && run0 + 64 <= N;   // whole, aligned lines only
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Vec16<T> ov[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) ov[p].v[e] = from_f<T>(acc[J0 + ((p * CH + e) >> 2)][i][e & 3]);
                    if constexpr (STATS) {
                        if (valid[i] && piece_ok[p]) {
#pragma unroll
                            for (int e = 0; e < CH; ++e) {
                                const float q = to_f(ov[p].v[e]);
                                s1[p * CH + e] += q;
                                s2[p * CH + e] = fmaf(q, q, s2[p * CH + e]);
                            }
                        }
                    }
                }
                const u32x4_t w0 = __builtin_bit_cast(u32x4_t, ov[0]), w1 = __builtin_bit_cast(u32x4_t, ov[1]);
                // byte offset of the run of this lane's pixel, or an offset the range check drops; the storing lanes fetch it
                // from a lane that holds their pixel (the output may be scattered, so it is not derived from the lane number)
                const uint32_t rb = valid[i] ? (uint32_t)(offs[i] - grp * 16) * 2u : kNoLoad;
                asm volatile("ds_write_b128 %0, %1" ::"v"(wa0), "v"(w0) : "memory");
                asm volatile("ds_write_b128 %0, %1" ::"v"(wa1), "v"(w1) : "memory");
                u32x4_t o0, o1;
                asm volatile("ds_read_b128 %0, %1" : "=v"(o0) : "v"(ra0) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(o1) : "v"(ra1) : "memory");
                const uint32_t prb0 = (uint32_t)__builtin_amdgcn_ds_bpermute((ln >> 3) * 4, (int)rb);
                const uint32_t prb1 = (uint32_t)__builtin_amdgcn_ds_bpermute((8 + (ln >> 3)) * 4, (int)rb);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0), "+v"(o1)::"memory");
#if VAMD_ABLATE == 12       // timing only: every store is dropped by the range check
                const uint32_t off0 = alpha == -7.25f ? prb0 : kNoLoad, off1 = alpha == -7.25f ? prb1 : kNoLoad;
#else
                const uint32_t off0 = chunk_ok ? prb0 + (uint32_t)(ln & 7) * 16u : kNoLoad;        // kNoLoad + 112 is still out of range
                const uint32_t off1 = chunk_ok ? prb1 + (uint32_t)(ln & 7) * 16u : kNoLoad;
#endif
                if (lines) {
                    __builtin_amdgcn_raw_buffer_store_b128(o0, ry, (int)off0, 0, 2);
                    __builtin_amdgcn_raw_buffer_store_b128(o1, ry, (int)off1, 0, 2);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(o0, ry, (int)off0, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(o1, ry, (int)off1, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);         // row by row: scheduled as one block, the rows' values are all live at once
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#if VAMD_ABLATE == 12       // timing only: every store is dropped by the range check
                const bool ok = alpha == -7.25f;
#else
                const bool ok = valid[i] && piece_ok[p];
#endif
                Vec16<T> ov;
#pragma unroll
                for (int e = 0; e < CH; ++e) ov.v[e] = from_f<T>(acc[J0 + ((p * CH + e) >> 2)][i][e & 3]);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, ov), ry,
                                                       (int)(ok ? (uint32_t)(offs[i] + p * CH) * (uint32_t)sizeof(T) : kNoLoad), 0, 0);
                if constexpr (STATS) {
#pragma unroll
                    for (int e = 0; e < CH; ++e) {
                        const float q = ok ? to_f(ov.v[e]) : 0.f;
                        s1[p * CH + e] += q;
                        s2[p * CH + e] = fmaf(q, q, s2[p * CH + e]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);             // row by row: scheduled as one block, the rows' values are all live at once
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (!valid[i]) continue;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (!piece_ok[p]) continue;
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < CH; ++e) ov.v[e] = from_f<T>(acc[J0 + ((p * CH + e) >> 2)][i][e & 3]);
#if VAMD_ABLATE == 12       // timing only: the epilogue computes its values and stores none of them
            if (alpha == -7.25f)
#endif
            *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(yout) + offs[i] + p * CH) = ov;
            if constexpr (STATS) {                        // statistics of the rounded values a later pass would read
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const float q = to_f(ov.v[e]);
                    s1[p * CH + e] += q;
                    s2[p * CH + e] = fmaf(q, q, s2[p * CH + e]);
                }
            }
        }
    }
}

// The same for rows m_first + 16*i of the GEMM (with the output scatter of a parity class, if any).
template <typename T, int NT, bool STATS, int C0 = 0, int CW = 4 * NT, bool ACT = false, bool BUF = false>
__device__ __forceinline__ void epilogue_rows(f32x4_t (&acc)[NT][4], float* s1, float* s2, const Gather& gp,
                                              const float* temb, const void* res, const void* res2, float alpha,
                                              int out_f32, void* yout, int N, int ldc, int64_t zoff,
                                              uint32_t m_first, int nb, const float* asc = nullptr,
                                              const float* ash = nullptr, float aslope = 0.f, void* raw = nullptr,
                                              int dchan = C0, uint32_t xp_base = 0, uint32_t xp_stride = 0, int nt_store = 0) {
    nb += dchan;                                       // first channel of accumulator tile C0 / 4 (C0 where a lane's channels are one run)
    int64_t offs[4];
    uint32_t bimg[4];
    bool valid[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t m = m_first + 16 * i;
        valid[i] = m < (uint32_t)gp.M;
        bimg[i] = 0;
        int64_t orow = m;
        if (gp.ostep || temb) {
            bimg[i] = fdiv(m, gp.div_howo);
            if (gp.ostep) {
                const uint32_t rem = m - bimg[i] * (uint32_t)(gp.Ho * gp.Wo);
                const uint32_t ho = fdiv(rem, gp.div_wo);
                const uint32_t wo = rem - ho * gp.Wo;
                orow = ((int64_t)bimg[i] * gp.oH + ho * gp.ostep + gp.oph) * gp.oW + wo * gp.ostep + gp.opw;
            }
        }
        offs[i] = zoff + orow * ldc + nb;
    }
    epilogue_rows_at<T, NT, STATS, C0, CW, ACT, BUF>(acc, s1, s2, temb, res, res2, alpha, out_f32, yout, N, ldc, offs, bimg, valid, nb,
                                                     asc, ash, aslope, raw, xp_base, xp_stride, nt_store);
}

// Weight gradient with the gradient tile resident in registers (conv_wgrad_halo.hip): 3 x 3 / stride 1 layers whose
// channel counts are small multiples of 64, bf16.  One partial product [Cout][9][Cin] (+ [Cout] bias sums) per slice.
struct ConvGeom;
bool wgrad_halo_ok(const ConvGeom& g, int dtype);
int wgrad_halo_slices(const ConvGeom& g);
int launch_wgrad_halo(const ConvGeom& g, const void* dy, const void* x, float* parts, float* bias_parts, hipStream_t st,
                      const float* pre_scale = nullptr, const float* pre_shift = nullptr, float pre_slope = 0.f);

}  // namespace vamd
