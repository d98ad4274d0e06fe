// Implicit-GEMM convolution on the gfx950 matrix cores (MFMA), NHWC activations.
//
//   forward / input-gradient ("NT"):  Y[m][n]  = sum_{tap,c} Xg[m][tap][c] * W[n][tap][c]
//   weight-gradient          ("TN"):  dW[n][f] = sum_m dY[m][n] * Xg[m][f],  f = (tap, c)
//
// m runs over output pixels (b, ho, wo), n over output channels; Xg is the input gathered through
// the filter tap (with zero padding, stride 2, nearest-2x upsampling or zero-insertion).  Nothing
// is materialised: every workgroup streams 16-byte pieces of the gathered operand straight from
// global memory into LDS with `global_load_lds_dwordx4` (per-lane source address, linear LDS
// destination), out-of-image pieces are fetched from a zero page.  Tiles are 128 x BN x (128 bytes
// of reduction); 4 waves of 64 lanes, each wave owns a 64 x BN/2 block of 16x16 MFMA tiles with
// fp32 accumulators.  LDS rows are XOR-swizzled (on the source side, as the LDS image of a
// direct-to-LDS load is lane-linear) so that fragment reads are bank-conflict free.
//
// Element types: bf16 (v_mfma_f32_16x16x32_bf16) and fp32 (v_mfma_f32_16x16x4_f32, exact fp32).
#include <stdlib.h>
#include <atomic>
#include <type_traits>

#include "conv_shapes.h"

#ifndef VAMD_ABLATE
#define VAMD_ABLATE 0        // 1..4, 10, 11 / 5..7 / 8 / 12: timing-only experiments on the wide / the 128-row / the weight-gradient kernel / every register epilogue (results are wrong)
#endif
#if VAMD_ABLATE != 0 && !defined(VAMD_TIMING_ONLY_BUILD)
#error "VAMD_ABLATE variants compute wrong results: build them with `build_ext.py --variant NAME -DVAMD_ABLATE=n` (a separately named library), never into libvision_amd.so"
#endif

#include "conv_nt_common.h"

namespace vamd {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

template <typename T, int BN, bool UP>
__global__ void __launch_bounds__(256, 2)
conv_nt_mfma_kernel(Gather gp, const T* __restrict__ x, const T* __restrict__ w, int N, int ldw, int ldc,
                    Epilogue ep, void* __restrict__ yout, int m_tiles, int n_tiles, int64_t strideA,
                    int64_t strideB, int64_t strideC) {
    constexpr int PIECE = ElemTraits<T>::PIECE;          // elements per 16-byte piece
    constexpr int KSTEP = kRowBytes / sizeof(T);          // reduction elements per K-step
    constexpr int A_BYTES = kBM * kRowBytes;
    constexpr int B_BYTES = BN * kRowBytes;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NT = BN / 32;                           // 16-wide n tiles per wave
    constexpr int LDC = BN + 4;                           // fp32 epilogue tile stride

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int mt = bid / n_tiles, nt = bid - mt * n_tiles;
    const int m0 = mt * kBM, n0 = nt * BN;
    x += (int64_t)blockIdx.z * strideA;
    w += (int64_t)blockIdx.z * strideB;

    // ---- staging state: this thread copies piece `slot` of rows srow + 32*i.
    // Per row: the element offset of its window origin and one validity bit per filter tap, so a
    // K-step costs an add, a bit test and a select per row (the tensors are < 2 GiB: 32-bit offsets).
    const int srow = tid >> 3, slot = tid & 7;
    const int chunk = slot ^ (srow & 7);                 // logical piece held by this LDS slot
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    int a_base[4], a_h0[4], a_w0[4];
    uint32_t a_mask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t m = m0 + srow + 32 * i;
        const bool row_ok = m < (uint32_t)gp.M;
        const uint32_t b = fdiv(m, gp.div_howo);
        const uint32_t rem = m - b * (uint32_t)(gp.Ho * gp.Wo);
        const uint32_t ho = fdiv(rem, gp.div_wo);
        const int h0 = (int)ho * gp.stride - gp.pad;
        const int w0 = (int)(rem - ho * gp.Wo) * gp.stride - gp.pad;
        a_h0[i] = h0;
        a_w0[i] = w0;
        a_base[i] = UP ? (int)b * gp.Hin * gp.Win * gp.ldx : (((int)b * gp.Hin + h0) * gp.Win + w0) * gp.ldx;
        uint32_t mask = 0, bit = 1;
        for (int r = 0; r < gp.kr; ++r) {
            const bool rok = row_ok && (unsigned)(h0 + r) < (unsigned)gp.He;
            for (int q = 0; q < gp.ks; ++q, bit <<= 1)
                if (rok && (unsigned)(w0 + q) < (unsigned)gp.We) mask |= bit;
        }
        a_mask[i] = mask;
    }
    const T* b_ptr[BN / 32];
    bool b_ok[BN / 32];
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
        const int n = n0 + srow + 32 * i;
        b_ok[i] = n < N;
        b_ptr[i] = w + (int64_t)(b_ok[i] ? n : 0) * ldw;
    }
    int f = chunk * PIECE;                               // flattened (tap, c) index of my piece
    int c = f, tr = 0, ts = 0, tap = 0;
    auto advance = [&]() {                               // branch-free: at most two taps per K-step
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int adv = c >= gp.Cin ? 1 : 0;
            c -= adv ? gp.Cin : 0;
            tap += adv;
            ts += adv;
            const int wrap = ts == gp.ks ? 1 : 0;
            ts = wrap ? 0 : ts;
            tr += wrap;
        }
    };
    advance();

    const T* zero = reinterpret_cast<const T*>(g_zero_page);
    // One K-step of staging is issued in two halves (activation rows, filter rows) so that each can
    // sit between the fragment reads and the MFMAs of the K-step being computed.
    auto stage_a = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE + wave_s * 1024;
        const bool f_ok = f < gp.Ktot;
        const int tapoff = (tr * gp.Win + ts) * gp.ldx + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = f_ok & (bool)((a_mask[i] >> (tap & 31)) & 1u);
            int off;
            if constexpr (UP) off = a_base[i] + (((a_h0[i] + tr) >> 1) * gp.Win + ((a_w0[i] + ts) >> 1)) * gp.ldx + c;
            else off = a_base[i] + tapoff;
            const T* src = pick_ptr(ok, x + (uint32_t)off, zero);
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sa + i * 4096), 16, 0, 0);
        }
    };
    auto stage_b = [&](int buf) {
        unsigned char* sb = smem + buf * STAGE + A_BYTES + wave_s * 1024;
        const bool f_ok = f < gp.Ktot;
        const int wf = ((gp.wr0 + gp.wstep * tr) * gp.wk + gp.ws0 + gp.wstep * ts) * gp.Cin + c;   // == f for a plain filter
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const T* src = pick_ptr(f_ok & b_ok[i], b_ptr[i] + (uint32_t)wf, zero);
            __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sb + i * 4096), 16, 0, 0);
        }
        f += KSTEP;
        c += KSTEP;
        advance();
    };

    f32x4_t acc[4][NT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    using Frag = typename ElemTraits<T>::Frag;           // 16 bytes of one row: 8 bf16 / 4 fp32
    const int frow = lane & 15, fgrp = lane >> 4, fsw = lane & 7;
    auto read_frags = [&](int buf, int ks, Frag (&a)[4], Frag (&b)[NT]) {
        const unsigned char* sa = smem + buf * STAGE + (wm * 64 + frow) * kRowBytes;
        const unsigned char* sb = smem + buf * STAGE + A_BYTES + (wn * (BN / 2) + frow) * kRowBytes;
        const int po = (((ks * 4 + fgrp) ^ fsw) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const Frag*>(sa + i * 16 * kRowBytes + po);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const Frag*>(sb + j * 16 * kRowBytes + po);
    };
    auto mma = [&](const Frag (&a)[4], const Frag (&b)[NT]) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    };

    const int nk = (gp.Ktot + KSTEP - 1) / KSTEP;
    stage_a(0);
    stage_b(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // Steady state, one basic block per K-step: fragment reads of K-step t, then the address
    // arithmetic and direct-to-LDS loads of K-step t+1 (which cover the LDS latency), then the MFMAs.
    // LDS reads and LDS-DMA loads keep their program order; everything else floats between MFMAs.
    for (int t = 0; t + 1 < nk; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        Frag a[4], b[NT];
        read_frags(cur, 0, a, b);
        stage_a(nxt);
        mma(a, b);
        read_frags(cur, 1, a, b);
        stage_b(nxt);
        mma(a, b);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    {
        const int cur = (nk - 1) & 1;
        Frag a[4], b[NT];
        read_frags(cur, 0, a, b);
        mma(a, b);
        read_frags(cur, 1, a, b);
        mma(a, b);
    }
    __syncthreads();

    // ---- epilogue: accumulators -> fp32 LDS tile -> coalesced 16-byte rows with the fused adds
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(wm * 64 + i * 16 + fgrp * 4 + r) * LDC + wn * (BN / 2) + j * 16 + frow] = acc[i][j][r];
    __syncthreads();

    constexpr int G = BN / PIECE;                         // 16-byte output groups per row
    const int grp = tid % G;
    const int n = n0 + grp * PIECE;
    // a trailing partial group (N not a multiple of PIECE) is written whole: its extra columns are
    // exact zeros (their weight rows were never loaded) and ldc leaves room for them
    float s1[PIECE], s2[PIECE];                           // column sums of what this thread stores
#pragma unroll
    for (int e = 0; e < PIECE; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (n < N && n + PIECE <= ldc) {
        float bv[PIECE];
#pragma unroll
        for (int e = 0; e < PIECE; ++e) bv[e] = (ep.bias && n + e < N) ? ep.bias[n + e] : 0.f;
        const T* res = reinterpret_cast<const T*>(ep.res);
        const T* res2 = reinterpret_cast<const T*>(ep.res2);
        for (int row = tid / G; row < kBM; row += 256 / G) {
            const uint32_t m = m0 + row;
            if (m >= (uint32_t)gp.M) break;
            float v[PIECE];
#pragma unroll
            for (int e = 0; e < PIECE; e += 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(tile + row * LDC + grp * PIECE + e);
                v[e] = t4.x; v[e + 1] = t4.y; v[e + 2] = t4.z; v[e + 3] = t4.w;
            }
#pragma unroll
            for (int e = 0; e < PIECE; ++e) v[e] = v[e] * ep.alpha + bv[e];
            int64_t orow = m;
            if (gp.ostep) {
                const uint32_t b = fdiv(m, gp.div_howo);
                const uint32_t rem = m - b * (uint32_t)(gp.Ho * gp.Wo);
                const uint32_t ho = fdiv(rem, gp.div_wo);
                const uint32_t wo = rem - ho * gp.Wo;
                orow = ((int64_t)b * gp.oH + ho * gp.ostep + gp.oph) * gp.oW + wo * gp.ostep + gp.opw;
            }
            const int64_t off = (int64_t)blockIdx.z * strideC + orow * ldc + n;
            if (ep.temb) {
                const uint32_t b = fdiv(m, gp.div_howo);
#pragma unroll
                for (int e = 0; e < PIECE; ++e) if (n + e < N) v[e] += ep.temb[(int64_t)b * N + n + e];
            }
            if (res) {
                const Vec16<T> rv = *reinterpret_cast<const Vec16<T>*>(res + off);
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[e] += to_f(rv.v[e]);
            }
            if (res2) {
                const Vec16<T> rv = *reinterpret_cast<const Vec16<T>*>(res2 + off);
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[e] += to_f(rv.v[e]);
            }
            if (ep.out_f32) {
                float* yo = reinterpret_cast<float*>(yout) + off;
#pragma unroll
                for (int e = 0; e < PIECE; e += 4)
                    *reinterpret_cast<float4*>(yo + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
            } else {
                Vec16<T> ov;
#pragma unroll
                for (int e = 0; e < PIECE; ++e) ov.v[e] = from_f<T>(v[e]);
                *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(yout) + off) = ov;
                if (ep.stats) {                           // statistics of the rounded values a later pass would read
#pragma unroll
                    for (int e = 0; e < PIECE; ++e) {
                        const float r = to_f(ov.v[e]);
                        s1[e] += r;
                        s2[e] = fmaf(r, r, s2[e]);
                    }
                }
            }
        }
    }
    if (ep.stats) {                                       // uniform: fold the 256/G row groups through LDS
        __syncthreads();
        float* red = tile;                                // [256/G][BN][2]
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            red[((tid / G) * BN + grp * PIECE + e) * 2 + 0] = s1[e];
            red[((tid / G) * BN + grp * PIECE + e) * 2 + 1] = s2[e];
        }
        __syncthreads();
        for (int idx = tid; idx < BN * 2; idx += 256) {
            const int col = idx >> 1;
            float sum = 0.f;
#pragma unroll 4
            for (int rr = 0; rr < 256 / G; ++rr) sum += red[rr * BN * 2 + idx];
            if (n0 + col < N) ep.stats[((int64_t)mt * N + n0 + col) * 2 + (idx & 1)] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Persistent variant of the forward / input-gradient kernel.  A workgroup walks a strided list of
// output tiles that share one column tile, so that
//   * the fixed cost of a tile (address set-up, first-stage load latency, output pass) overlaps the
//     MFMAs of its neighbours: the first K-step of the next tile is staged during the last K-step
//     of the current one and is in flight while the outputs are written;
//   * the outputs leave straight from the accumulators: the MFMA operands are swapped (rows =
//     output channels, columns = pixels) and the filter rows are permuted inside the LDS tile, so
//     every lane ends up with 4*NT consecutive channels of one pixel (no LDS pass, no barrier);
//   * fused column statistics are kept in registers across the tiles and reduced once.
template <typename T, int BN, int GM, int EPI>               // GM: 0 plain gather, 1 upsampling gather, 2 second source; EPI: 0 plain, 1 statistics, 2 fused normalise
__global__ void __launch_bounds__(256, BN == 64 ? 3 : 2)
conv_ntp_mfma_kernel(Gather gp, const T* __restrict__ x, const T* __restrict__ w, int N, int ldw, int ldc,
                     Epilogue ep, void* __restrict__ yout, int m_tiles, int n_tiles, int row_tiles,
                     int64_t strideA, int64_t strideB, int64_t strideC, ClassSet cs) {
    constexpr bool UP = GM == 1, TWO = GM == 2;
    constexpr int PIECE = ElemTraits<T>::PIECE;          // elements per 16-byte piece
    constexpr int KSTEP = kRowBytes / sizeof(T);          // reduction elements per K-step
    constexpr int A_BYTES = kBM * kRowBytes;
    constexpr int B_BYTES = BN * kRowBytes;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr bool STATS = EPI == 1, ACT = EPI == 2;
    constexpr int NT = BN / 32;                           // 16-wide channel tiles per wave
    constexpr int CPL = 4 * NT;                           // consecutive output channels per lane
    using Frag = typename ElemTraits<T>::Frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    int blk = blockIdx.x, nblk = gridDim.x;
    int64_t yoff = 0;
    if (cs.ncls > 1) {                                     // this workgroup's class (uniform)
        const int c = (blk >= cs.first_block[1]) + (blk >= cs.first_block[2]) + (blk >= cs.first_block[3]);
        gp = cs.g[c];
        m_tiles = cs.m_tiles[c];
        row_tiles = m_tiles;
        w += cs.w_off[c];
        if (ep.stats) ep.stats += cs.stats_off[c];
        yoff = cs.y_off[c];
        nblk = cs.first_block[c + 1] - cs.first_block[c];
        blk -= cs.first_block[c];
    }
    const int bid = xcd_remap(blk, nblk);
    const int nt = bid % n_tiles;                          // this workgroup's column tile, fixed
    const int rstep = nblk / n_tiles;                      // row tiles advance by this
    int rt = bid / n_tiles;                                // row tile = (batch z, m tile)
    const int n0 = nt * BN;

    // ---- staging: this thread copies piece `slot` of rows srow + 32*i
    const int srow = tid >> 3, slot = tid & 7;
    const int chunk = slot ^ (srow & 7);                 // logical piece held by this LDS slot
    // filter rows: LDS row R of the wave half h = R / (BN/2), q = R % (BN/2) = 16*jc + i holds
    // channel h*(BN/2) + (i>>2)*CPL + 4*jc + (i&3)
    int b_off[BN / 32], b_off2[TWO ? BN / 32 : 1];
    bool b_ok[BN / 32];
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
        const int R = srow + 32 * i;
        const int h = R / (BN / 2), q = R - h * (BN / 2);
        const int n = n0 + h * (BN / 2) + ((q & 15) >> 2) * CPL + 4 * (q >> 4) + (q & 3);
        b_ok[i] = n < N;
        b_off[i] = (b_ok[i] ? n : 0) * ldw;
        if constexpr (TWO) b_off2[i] = (b_ok[i] ? n : 0) * gp.Cin2;
    }
    int a_base[4], a_h0[4], a_w0[4];
    uint32_t a_mask[4];
    int f, c, tr, ts, tap;
    int a_m0 = 0;                                        // first pixel row of this thread in the tile (second source)
    __amdgpu_buffer_rsrc_t rx = make_rsrc(x), rw = make_rsrc(w);
    const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(TWO ? gp.x2 : x), rw2 = make_rsrc(TWO ? gp.w2 : w);
    const int ntaps = gp.kr * gp.ks;
    auto advance = [&]() {                               // branch-free: at most two taps per K-step
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int adv = c >= gp.Cin ? 1 : 0;
            c -= adv ? gp.Cin : 0;
            tap += adv;
            ts += adv;
            const int wrap = ts == gp.ks ? 1 : 0;
            ts = wrap ? 0 : ts;
            tr += wrap;
        }
    };
    auto setup = [&](int row_tile) {                      // staging state of K-step 0 of a tile
        const int z = row_tile < m_tiles ? 0 : row_tile / m_tiles, mt = row_tile - z * m_tiles;   // one batch entry: no division
        rx = make_rsrc(x + (int64_t)z * strideA);
        rw = make_rsrc(w + (int64_t)z * strideB);
        a_m0 = mt * kBM + srow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t m = mt * kBM + srow + 32 * i;
            const bool row_ok = m < (uint32_t)gp.M;
            const uint32_t b = fdiv(m, gp.div_howo);
            const uint32_t rem = m - b * (uint32_t)(gp.Ho * gp.Wo);
            const uint32_t ho = fdiv(rem, gp.div_wo);
            const int h0 = (int)ho * gp.stride - gp.pad;
            const int w0 = (int)(rem - ho * gp.Wo) * gp.stride - gp.padw;
            a_h0[i] = h0;
            a_w0[i] = w0;
            a_base[i] = UP ? (int)b * gp.Hin * gp.Win * gp.ldx : (((int)b * gp.Hin + h0) * gp.Win + w0) * gp.ldx;
            // tap (r, q) is bit r*ks + q: column validity once, replicated for every valid row
            // (filters are at most 4 x 4; no loops, no branches)
            const uint32_t cm = ((unsigned)w0 < (unsigned)gp.We ? 1u : 0u) |
                                (((gp.ks > 1) & ((unsigned)(w0 + 1) < (unsigned)gp.We)) ? 2u : 0u) |
                                (((gp.ks > 2) & ((unsigned)(w0 + 2) < (unsigned)gp.We)) ? 4u : 0u) |
                                (((gp.ks > 3) & ((unsigned)(w0 + 3) < (unsigned)gp.We)) ? 8u : 0u);
            const uint32_t m0 = (row_ok & ((unsigned)h0 < (unsigned)gp.He)) ? cm : 0u;
            const uint32_t m1 = ((gp.kr > 1) & row_ok & ((unsigned)(h0 + 1) < (unsigned)gp.He)) ? (cm << gp.ks) : 0u;
            const uint32_t m2 = ((gp.kr > 2) & row_ok & ((unsigned)(h0 + 2) < (unsigned)gp.He)) ? (cm << (2 * gp.ks)) : 0u;
            const uint32_t m3 = ((gp.kr > 3) & row_ok & ((unsigned)(h0 + 3) < (unsigned)gp.He)) ? (cm << (3 * gp.ks)) : 0u;
            a_mask[i] = m0 | m1 | m2 | m3;
        }
        f = chunk * PIECE;
        c = f; tr = 0; ts = 0; tap = 0;
        advance();
    };

    auto stage_a = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE + wave_s * 1024;
        const bool f_ok = f < gp.Ktot;
        if constexpr (TWO) {
            if (__builtin_amdgcn_readfirstlane(tap) >= ntaps) {       // uniform: channel counts are multiples of a K-step
                const int c2 = (tap - ntaps) * gp.Cin + c;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = a_m0 + 32 * i;
                    load16_lds(rx2, (uint32_t)(m * gp.Cin2 + c2) * (uint32_t)sizeof(T), f_ok & (m < gp.M), sa + i * 4096);
                }
                return;
            }
        }
        const int tapoff = (tr * gp.Win + ts) * gp.ldx + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = f_ok & (bool)((a_mask[i] >> (tap & 31)) & 1u);
            int off;
            if constexpr (UP) off = a_base[i] + (((a_h0[i] + tr) >> 1) * gp.Win + ((a_w0[i] + ts) >> 1)) * gp.ldx + c;
            else off = a_base[i] + tapoff;
            load16_lds(rx, (uint32_t)off * (uint32_t)sizeof(T), ok, sa + i * 4096);
        }
    };
    auto stage_b = [&](int buf) {
        unsigned char* sb = smem + buf * STAGE + A_BYTES + wave_s * 1024;
        const bool f_ok = f < gp.Ktot;
        const int wf = ((gp.wr0 + gp.wstep * tr) * gp.wk + gp.ws0 + gp.wstep * ts) * gp.Cin + c;   // == f for a plain filter
        bool second = false;
        if constexpr (TWO) second = __builtin_amdgcn_readfirstlane(tap) >= ntaps;
        if (second) {
            const int c2 = (tap - ntaps) * gp.Cin + c;
#pragma unroll
            for (int i = 0; i < BN / 32; ++i)
                load16_lds(rw2, (uint32_t)(b_off2[TWO ? i : 0] + c2) * (uint32_t)sizeof(T), f_ok & b_ok[i], sb + i * 4096);
        } else {
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                load16_lds(rw, (uint32_t)(b_off[i] + wf) * (uint32_t)sizeof(T), f_ok & b_ok[i], sb + i * 4096);
            }
        }
        f += KSTEP;
        c += KSTEP;
        advance();
    };

    f32x4_t acc[NT][4];                                   // [channel tile][pixel tile]

    const int frow = lane & 15, fgrp = lane >> 4, fsw = lane & 7;
    auto read_frags = [&](int buf, int ks, Frag (&a)[4], Frag (&b)[NT]) {
        const unsigned char* sa = smem + buf * STAGE + (wm * 64 + frow) * kRowBytes;
        const unsigned char* sb = smem + buf * STAGE + A_BYTES + (wn * (BN / 2) + frow) * kRowBytes;
        const int po = (((ks * 4 + fgrp) ^ fsw) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const Frag*>(sa + i * 16 * kRowBytes + po);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const Frag*>(sb + j * 16 * kRowBytes + po);
    };
    auto mma = [&](const Frag (&a)[4], const Frag (&b)[NT]) {   // D rows = channels, columns = pixels
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][e], a[i][e], acc[j][i], 0, 0, 0);
        }
    };
    // one K-step: fragment reads of the staged buffer, the next stage's loads between them and the MFMAs
    auto kstep = [&](int buf, auto do_stage) {
        Frag a[4], b[NT];
        constexpr bool stage_now = decltype(do_stage)::value;
        // measured: the narrow tile (3 workgroups per CU) runs faster with all loads of the next
        // K-step issued before the first MFMA half, the wide one with the filter rows issued later
        read_frags(buf, 0, a, b);
#if VAMD_ABLATE != 7
        if constexpr (stage_now) stage_a(buf ^ 1);
        if constexpr (stage_now && BN == 64) stage_b(buf ^ 1);
#endif
        mma(a, b);
        read_frags(buf, 1, a, b);
#if VAMD_ABLATE != 7
        if constexpr (stage_now && BN != 64) stage_b(buf ^ 1);
#endif
        mma(a, b);
    };

    // ---- outputs: lane (p = lane & 15, grp = lane >> 4) holds, per pixel tile, channels nb .. nb+CPL-1
    const int nb = n0 + wn * (BN / 2) + fgrp * CPL;
    float s1[STATS ? CPL : 1], s2[STATS ? CPL : 1];
#pragma unroll
    for (int e = 0; e < (STATS ? CPL : 1); ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // accumulators start from the bias (alpha is 1 whenever a bias is given)
    auto init_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ep.bias) {
                const int n = nb + 4 * j;
                if (n + 4 <= N) {
                    const float4 t4 = *reinterpret_cast<const float4*>(ep.bias + n);
                    bv = f32x4_t{t4.x, t4.y, t4.z, t4.w};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = n + r < N ? ep.bias[n + r] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = bv;
        }
    };
    auto write_tile = [&](int row_tile) {
        const int z = row_tile < m_tiles ? 0 : row_tile / m_tiles, mt = row_tile - z * m_tiles;   // one batch entry: no division
        epilogue_rows<T, NT, STATS, 0, 4 * NT, ACT>(acc, s1, s2, gp, ep.temb, ep.res, ep.res2, ep.alpha, ep.out_f32, yout, N, ldc,
                                                    (int64_t)z * strideC + yoff, (uint32_t)(mt * kBM + wm * 64 + frow), nb,
                                                    ep.act_scale, ep.act_shift, ep.act_slope, ep.raw);
    };

    const int nk = (gp.Ktot + KSTEP - 1) / KSTEP;
    int cur = rt;
    init_acc();
    setup(rt);
    stage_a(0);
    stage_b(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (;;) {
        for (int k = 0; k + 1 < nk; ++k) {
            kstep(buf, std::true_type{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
        // last K-step of `cur`: stage the first K-step of the next tile instead, if there is one
        rt += rstep;
        const bool more = rt < row_tiles;
        if (more) {
#if VAMD_ABLATE != 5
            setup(rt);
#else
            f = chunk * PIECE; c = f; tr = 0; ts = 0; tap = 0; advance();
#endif
            kstep(buf, std::true_type{});
        } else {
            kstep(buf, std::false_type{});
        }
#if VAMD_ABLATE != 6
        write_tile(cur);                                  // the staged loads stay in flight meanwhile
#else
        if (acc[0][0][0] == 123.456f) write_tile(cur);
#endif
        if (!more) break;
        init_acc();
        cur = rt;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }

    if constexpr (STATS) {
        // per-lane sums -> the 16 pixel lanes of each group -> the two pixel waves, through LDS
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
            }
        }
        __syncthreads();                                  // every wave is done with the staged tiles
        float* red = reinterpret_cast<float*>(smem);      // [wm][BN channels of the tile][2]
        if (frow == 0) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                const int col = wn * (BN / 2) + fgrp * CPL + e;
                red[(wm * BN + col) * 2 + 0] = s1[e];
                red[(wm * BN + col) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        const int part = bid / n_tiles;
        for (int idx = tid; idx < BN * 2; idx += 256) {
            const int col = idx >> 1;
            if (n0 + col < N) ep.stats[((int64_t)part * N + n0 + col) * 2 + (idx & 1)] = red[idx] + red[BN * 2 + idx];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Wide persistent variant for large layers: 256 pixel rows x BN channels per tile, 8 waves (each
// still a 64 x BN/2 sub-tile), ONE workgroup per CU and a ring of three K-step stages.  Against the
// 128-row kernel a tile moves 25 % fewer bytes from L2 per MFMA, and the loads of a K-step are
// issued two K-steps before they are read: the wait before a K-step's barrier covers the next
// stage only, the one after it stays in flight across the barrier.  The K-steps of consecutive
// tiles form one stream, so the ring never drains between tiles.
template <typename T, int BN, int GM, int EPI>               // GM: 0 plain gather, 1 upsampling gather, 2 second source; EPI: 0 plain, 1 statistics, 2 fused normalise
__global__ void __launch_bounds__(512, 1)
conv_ntw_mfma_kernel(Gather gp, const T* __restrict__ x, const T* __restrict__ w, int N, int ldw, int ldc,
                     Epilogue ep, void* __restrict__ yout, int m_tiles, int n_tiles, int row_tiles,
                     int64_t strideA, int64_t strideB, int64_t strideC, ClassSet cs) {
    constexpr bool UP = GM == 1, TWO = GM == 2;
    constexpr int PIECE = ElemTraits<T>::PIECE;          // elements per 16-byte piece
    constexpr int KSTEP = kRowBytes / sizeof(T);          // reduction elements per K-step
    constexpr int BM = 256;
    constexpr int NPA = BM / 64, NPB = BN / 64;           // 16-byte pieces per thread and K-step: pixels, filter rows
    constexpr int NLOAD = NPA + NPB;
    constexpr int A_BYTES = BM * kRowBytes;
    constexpr int B_BYTES = BN * kRowBytes;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr bool STATS = EPI == 1, ACT = EPI == 2;
    constexpr int NT = BN / 32;                           // 16-wide channel tiles per wave
    constexpr int CPL = 4 * NT;                           // consecutive output channels per lane
    using Frag = typename ElemTraits<T>::Frag;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    int blk = blockIdx.x, nblk = gridDim.x;
    int64_t yoff = 0;
    if (cs.ncls > 1) {                                     // this workgroup's class (uniform)
        const int c = (blk >= cs.first_block[1]) + (blk >= cs.first_block[2]) + (blk >= cs.first_block[3]);
        gp = cs.g[c];
        m_tiles = cs.m_tiles[c];
        row_tiles = m_tiles;
        w += cs.w_off[c];
        if (ep.stats) ep.stats += cs.stats_off[c];
        yoff = cs.y_off[c];
        nblk = cs.first_block[c + 1] - cs.first_block[c];
        blk -= cs.first_block[c];
    }
    const int bid = xcd_remap(blk, nblk);
    const int nt = bid % n_tiles;                          // this workgroup's column tile, fixed
    const int rstep = nblk / n_tiles;                      // row tiles advance by this
    int rt = bid / n_tiles;                                // row tile = (batch z, m tile)
    const int n0 = nt * BN;

    // ---- staging: this thread copies piece `slot` of rows srow + 64*i
    const int srow = tid >> 3, slot = tid & 7;
    const int chunk = slot ^ (srow & 7);                 // logical piece held by this LDS slot
    // filter rows: LDS row R of the wave half h = R / (BN/2), q = R % (BN/2) = 16*jc + i holds channel
    // h*(BN/2) + chan_of(jc, i>>2) + (i&3): accumulator tile jc of lane group g = i>>2 starts at channel
    // 64 (jc>>2) + 16 g + 4 (jc&3) of the wave's half, so that a lane's tiles jc = 4t .. 4t+3 are 16 consecutive
    // channels and the four lane groups together one 64-channel run (the 128 bytes of a pixel that a store writes whole)
    auto chan_of = [](int jc, int g) { return 64 * (jc >> 2) + 16 * g + 4 * (jc & 3); };
    int b_off[NPB], b_off2[TWO ? NPB : 1];
    bool b_ok[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int R = srow + 64 * i;
        const int h = R / (BN / 2), q = R - h * (BN / 2);
        const int n = n0 + h * (BN / 2) + chan_of(q >> 4, (q & 15) >> 2) + (q & 3);
        b_ok[i] = n < N;
        b_off[i] = (b_ok[i] ? n : 0) * ldw;
        if constexpr (TWO) b_off2[i] = (b_ok[i] ? n : 0) * gp.Cin2;
    }
    int a_base[NPA], a_h0[NPA], a_w0[NPA];
    uint32_t a_mask[NPA];
    int f, c, tr, ts, tap;
    int a_m0 = 0;                                        // first pixel row of this thread in the tile (second source)
    __amdgpu_buffer_rsrc_t rx = make_rsrc(x), rw = make_rsrc(w);
    const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(TWO ? gp.x2 : x), rw2 = make_rsrc(TWO ? gp.w2 : w);
    const int ntaps = gp.kr * gp.ks;
    auto advance = [&]() {                               // branch-free: at most two taps per K-step
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int adv = c >= gp.Cin ? 1 : 0;
            c -= adv ? gp.Cin : 0;
            tap += adv;
            ts += adv;
            const int wrap = ts == gp.ks ? 1 : 0;
            ts = wrap ? 0 : ts;
            tr += wrap;
        }
    };
    auto setup = [&](int row_tile) {                      // staging state of K-step 0 of a tile
        const int z = row_tile < m_tiles ? 0 : row_tile / m_tiles, mt = row_tile - z * m_tiles;   // one batch entry: no division
        rx = make_rsrc(x + (int64_t)z * strideA);
        rw = make_rsrc(w + (int64_t)z * strideB);
        a_m0 = mt * BM + srow;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const uint32_t m = mt * BM + srow + 64 * i;
            const bool row_ok = m < (uint32_t)gp.M;
            const uint32_t b = fdiv(m, gp.div_howo);
            const uint32_t rem = m - b * (uint32_t)(gp.Ho * gp.Wo);
            const uint32_t ho = fdiv(rem, gp.div_wo);
            const int h0 = (int)ho * gp.stride - gp.pad;
            const int w0 = (int)(rem - ho * gp.Wo) * gp.stride - gp.padw;
            a_h0[i] = h0;
            a_w0[i] = w0;
            a_base[i] = UP ? (int)b * gp.Hin * gp.Win * gp.ldx : (((int)b * gp.Hin + h0) * gp.Win + w0) * gp.ldx;
            // tap (r, q) is bit r*ks + q: column validity once, replicated for every valid row
            // (filters are at most 4 x 4; no loops, no branches)
            const uint32_t cm = ((unsigned)w0 < (unsigned)gp.We ? 1u : 0u) |
                                (((gp.ks > 1) & ((unsigned)(w0 + 1) < (unsigned)gp.We)) ? 2u : 0u) |
                                (((gp.ks > 2) & ((unsigned)(w0 + 2) < (unsigned)gp.We)) ? 4u : 0u) |
                                (((gp.ks > 3) & ((unsigned)(w0 + 3) < (unsigned)gp.We)) ? 8u : 0u);
            const uint32_t m0 = (row_ok & ((unsigned)h0 < (unsigned)gp.He)) ? cm : 0u;
            const uint32_t m1 = ((gp.kr > 1) & row_ok & ((unsigned)(h0 + 1) < (unsigned)gp.He)) ? (cm << gp.ks) : 0u;
            const uint32_t m2 = ((gp.kr > 2) & row_ok & ((unsigned)(h0 + 2) < (unsigned)gp.He)) ? (cm << (2 * gp.ks)) : 0u;
            const uint32_t m3 = ((gp.kr > 3) & row_ok & ((unsigned)(h0 + 3) < (unsigned)gp.He)) ? (cm << (3 * gp.ks)) : 0u;
            a_mask[i] = m0 | m1 | m2 | m3;
        }
        f = chunk * PIECE;
        c = f; tr = 0; ts = 0; tap = 0;
        advance();
    };

    auto stage_a = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE + wave_s * 1024;
        const bool f_ok = f < gp.Ktot;
        if constexpr (TWO) {
            if (__builtin_amdgcn_readfirstlane(tap) >= ntaps) {       // uniform: channel counts are multiples of a K-step
                const int c2 = (tap - ntaps) * gp.Cin + c;
#pragma unroll
                for (int i = 0; i < NPA; ++i) {
                    const int m = a_m0 + 64 * i;
                    load16_lds(rx2, (uint32_t)(m * gp.Cin2 + c2) * (uint32_t)sizeof(T), f_ok & (m < gp.M), sa + i * 8192);
                }
                return;
            }
        }
        const int tapoff = (tr * gp.Win + ts) * gp.ldx + c;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const bool ok = f_ok & (bool)((a_mask[i] >> (tap & 31)) & 1u);
            int off;
            if constexpr (UP) off = a_base[i] + (((a_h0[i] + tr) >> 1) * gp.Win + ((a_w0[i] + ts) >> 1)) * gp.ldx + c;
            else off = a_base[i] + tapoff;
            load16_lds(rx, (uint32_t)off * (uint32_t)sizeof(T), ok, sa + i * 8192);
        }
    };
    auto stage_b = [&](int buf) {
        unsigned char* sb = smem + buf * STAGE + A_BYTES + wave_s * 1024;
        const bool f_ok = f < gp.Ktot;
        const int wf = ((gp.wr0 + gp.wstep * tr) * gp.wk + gp.ws0 + gp.wstep * ts) * gp.Cin + c;   // == f for a plain filter
        bool second = false;
        if constexpr (TWO) second = __builtin_amdgcn_readfirstlane(tap) >= ntaps;
        if (second) {
            const int c2 = (tap - ntaps) * gp.Cin + c;
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                load16_lds(rw2, (uint32_t)(b_off2[TWO ? i : 0] + c2) * (uint32_t)sizeof(T), f_ok & b_ok[i], sb + i * 8192);
        } else
#if VAMD_ABLATE == 11       // timing only: filter rows loaded in the first K-step of a tile alone
        if (f < KSTEP)
#endif
        {
#pragma unroll
            for (int i = 0; i < NPB; ++i) {
                load16_lds(rw, (uint32_t)(b_off[i] + wf) * (uint32_t)sizeof(T), f_ok & b_ok[i], sb + i * 8192);
            }
        }
        f += KSTEP;
        c += KSTEP;
        advance();
    };

    f32x4_t acc[NT][4];                                   // [channel tile][pixel tile]

    const int frow = lane & 15, fgrp = lane >> 4, fsw = lane & 7;
    auto read_frags = [&](int buf, int ks, Frag (&a)[4], Frag (&b)[NT]) {
        const unsigned char* sa = smem + buf * STAGE + (wm * 64 + frow) * kRowBytes;
        const unsigned char* sb = smem + buf * STAGE + A_BYTES + (wn * (BN / 2) + frow) * kRowBytes;
        const int po = (((ks * 4 + fgrp) ^ fsw) << 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const Frag*>(sa + i * 16 * kRowBytes + po);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const Frag*>(sb + j * 16 * kRowBytes + po);
    };
    auto mma = [&](const Frag (&a)[4], const Frag (&b)[NT]) {   // D rows = channels, columns = pixels
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][e], a[i][e], acc[j][i], 0, 0, 0);
        }
    };
    // ---- outputs: lane (p = lane & 15, grp = lane >> 4) holds, per pixel tile, channels nb .. nb+15 in accumulator
    // tiles 0 .. 3 (and nb+64 .. nb+79 in tiles 4 .. 7 of the 256-channel tile)
    const int nb = n0 + wn * (BN / 2) + fgrp * 16;
    // Fused statistics: per-lane sums in registers, except for the 256-channel tile, whose 128
    // accumulators leave no room for 64 more: there every (pixel-wave row, channel) owns a slot
    // pair in LDS behind the stages, updated once per tile by the one lane that holds the
    // 16-lane sum (a fixed order of additions, no atomics).
    constexpr bool STATS_LDS = STATS && BN == 256;
    constexpr bool STATS_REG = STATS && !STATS_LDS;
    float* sred = reinterpret_cast<float*>(smem + (BN == 256 ? 2 : 3) * STAGE);      // [4][BN][2]
    if constexpr (STATS_LDS) {
        for (int idx = tid; idx < 4 * BN * 2; idx += 512) sred[idx] = 0.f;            // a barrier follows before any tile is written
    }
    float s1[STATS_REG ? CPL : 1], s2[STATS_REG ? CPL : 1];
#pragma unroll
    for (int e = 0; e < (STATS_REG ? CPL : 1); ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // accumulators start from the bias (alpha is 1 whenever a bias is given), kept in LDS: a load from global memory here
    // would be younger than the stores of the tile just written, and waiting for it would wait for them
    float* bias_lds = sred + (STATS_LDS ? 4 * BN * 2 : 0);                            // [BN]
    for (int idx = tid; idx < BN; idx += 512) bias_lds[idx] = (ep.bias && n0 + idx < N) ? ep.bias[n0 + idx] : 0.f;
    __syncthreads();
    // (read by inline asm: behind a read the compiler knows to be an LDS read it waits for every pending direct-to-LDS load,
    // and with them for the stores)
    const uint32_t bias_addr = (uint32_t)(uintptr_t)LPTR(bias_lds + (nb - n0));
    auto init_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4_t bv;
            asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(bv) : "v"(bias_addr), "n"(4 * (64 * (j >> 2) + 4 * (j & 3))));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = bv;
        }
    };
    // The tile's stores go out as buffer stores, a fixed number per lane (NSTORE), so that the first wait of the next tile
    // can leave exactly them in flight instead of standing still for their round trip to L2; not for fp32 outputs of a
    // bf16 product and the fused normalise (their store counts differ: those keep the full wait).
#ifdef VAMD_NOBUF
    constexpr bool BUF = false;
#else
    constexpr bool BUF = !ACT;
#endif
    // LDS the transposing epilogue passes a row group through, two 1 KB regions per wave: with three stages, this wave's
    // first two pixel slices of the stage just consumed (every wave is past the K-step's barrier, and only this wave's own
    // direct-to-LDS loads write there again); with two stages, 2 KB of its own behind the bias
    const uint32_t lds0 = (uint32_t)(uintptr_t)LPTR(smem);
    uint32_t xp_base = BN == 256 ? (uint32_t)(uintptr_t)LPTR(bias_lds + BN) + (uint32_t)wave * 2048u : 0u;
    constexpr uint32_t XP_STRIDE = BN == 256 ? 1024u : 8192u;
    constexpr int NSTORE = 4 * (4 * NT * (int)sizeof(T) / 16);
    const bool relax = BUF && !ep.out_f32;
    auto write_tile = [&](int row_tile) {
        const int z = row_tile < m_tiles ? 0 : row_tile / m_tiles, mt = row_tile - z * m_tiles;   // one batch entry: no division
        epilogue_rows<T, NT, STATS_REG, 0, 4 * NT, ACT, BUF>(acc, s1, s2, gp, ep.temb, ep.res, ep.res2, ep.alpha, ep.out_f32, yout, N, ldc,
                                                        (int64_t)z * strideC + yoff, (uint32_t)(mt * BM + wm * 64 + frow), nb,
                                                        ep.act_scale, ep.act_shift, ep.act_slope, ep.raw, 0, xp_base, XP_STRIDE, ep.nt_store);
    };

    // the 256-channel tile: the lane's 32 channels in two halves, so that only 16 sums and one half's
    // loads are live next to the 128 accumulators
    auto write_half = [&](int row_tile, auto half) {
        constexpr int H = decltype(half)::value;
        const int z = row_tile < m_tiles ? 0 : row_tile / m_tiles, mt = row_tile - z * m_tiles;   // one batch entry: no division
        float t1[16], t2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
        epilogue_rows<T, NT, STATS_LDS, (NT == 8 ? 16 * H : 0), (NT == 8 ? 16 : 4 * NT), ACT, BUF>(
            acc, t1, t2, gp, ep.temb, ep.res, ep.res2, ep.alpha, ep.out_f32, yout, N, ldc, (int64_t)z * strideC + yoff,
            (uint32_t)(mt * BM + wm * 64 + frow), nb, ep.act_scale, ep.act_shift, ep.act_slope, ep.raw, (NT == 8 ? 64 * H : 0),
            xp_base, XP_STRIDE, ep.nt_store);
        if constexpr (STATS_LDS) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    t1[e] += __shfl_xor(t1[e], o, 64);
                    t2[e] += __shfl_xor(t2[e], o, 64);
                }
            }
            if (frow == 0) {
                float* slot = sred + ((wm * BN) + (nb - n0) + 64 * H) * 2;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    slot[2 * e] += t1[e];
                    slot[2 * e + 1] += t2[e];
                }
            }
        }
    };
    auto write_tile_pieces = [&](int row_tile) {
        write_half(row_tile, std::integral_constant<int, 0>{});
        write_half(row_tile, std::integral_constant<int, 1>{});
    };

    const int nk = (gp.Ktot + KSTEP - 1) / KSTEP;
    // staging runs two K-steps ahead of the MFMAs, through the same (tile, K-step) sequence
    int s_k = 0;
    bool s_more = true;
    auto staged = [&]() {                                  // a stage was issued: move the staging state on
        if (++s_k == nk) {
            s_k = 0;
            rt += rstep;
            s_more = rt < row_tiles;
            if (s_more) setup(rt);
        }
    };
    int cur = rt;
    init_acc();
    setup(rt);
    stage_a(0);
    stage_b(0);
    staged();
    if constexpr (BN == 256) {
        // 256 x 256 tile (wave tile 64 x 128, 128 accumulator registers): two stages, the next K-step's
        // loads issued at the start of this one, fragments read per half K-step, one barrier at the end
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int buf = 0;
        for (;;) {
            for (int k = 0; k < nk; ++k) {
                const bool issue = s_more;
                Frag fa[4], fb[NT];
                read_frags(buf, 0, fa, fb);
                if (issue) stage_a(buf ^ 1);
                mma(fa, fb);
                read_frags(buf, 1, fa, fb);
                if (issue) {
                    stage_b(buf ^ 1);
                    staged();
                }
                mma(fa, fb);
                if (k == nk - 1) {
                    write_tile_pieces(cur);
                    init_acc();
                    // the next stage was issued before the tile's stores: they alone may stay in flight (for a K-step)
                    if (relax) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NSTORE) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                buf ^= 1;
            }
            cur += rstep;
            if (cur >= row_tiles) break;
        }
    } else {
    if (s_more) {
        stage_a(1);
        stage_b(1);
        staged();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    // Fragments are double-buffered in registers: the reads of one half K-step are in flight while
    // the MFMAs of the previous half run, so the eight lock-stepped waves never wait on LDS with
    // the matrix pipe idle.  One barrier per K-step, in its middle: before it a wave has waited for
    // its own loads of the next stage, after it the first half of that stage may be read.
    Frag a0[4], b0[NT], a1[4], b1[NT];
    read_frags(0, 0, a0, b0);
    int buf = 0;
    bool stores_behind = false;                            // the previous K-step ended with a tile's NSTORE buffer stores
    for (;;) {
        for (int k = 0; k < nk; ++k) {
            const int sbuf = buf >= 1 ? buf - 1 : 2;      // (buf + 2) % 3: read during the previous K-step
            const int nbuf = buf == 2 ? 0 : buf + 1;
            const bool issue = s_more;
            const bool last = k == nk - 1;
            read_frags(buf, 1, a1, b1);
#if VAMD_ABLATE == 10       // timing only: pixel loads for the first tap alone, as if a halo tile served the other eight
            if (issue && tap == 0) stage_a(sbuf);
#elif VAMD_ABLATE != 4
            if (issue) stage_a(sbuf);
#endif
            mma(a0, b0);
            // the stage read next was issued during the previous K-step: only this K-step's
            // pixel loads may stay in flight; every LDS read issued so far has long returned
#if VAMD_ABLATE == 1 || VAMD_ABLATE == 3
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            // (after a tile's epilogue: issued behind the stage waited for are the tile's stores and this K-step's pixel
            // loads; the stores get until the next K-step's wait, whose stage was issued behind them)
            if (issue && stores_behind) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPA + NSTORE) : "memory");
            else if (issue) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            stores_behind = false;
#endif
#if VAMD_ABLATE != 2 && VAMD_ABLATE != 3
            __builtin_amdgcn_s_barrier();
#endif
            if (!last || cur + rstep < row_tiles) read_frags(nbuf, 0, a0, b0);
            if (issue) {
#if VAMD_ABLATE != 4
                stage_b(sbuf);
#endif
                staged();
            }
            mma(a1, b1);
            if (last) {
                xp_base = lds0 + (uint32_t)buf * STAGE + (uint32_t)wave_s * 1024u;
                write_tile(cur);
                init_acc();
                stores_behind = relax;
            }
            buf = nbuf;
        }
        cur += rstep;
        if (cur >= row_tiles) break;
    }

    }
    if constexpr (STATS_LDS) {
        __syncthreads();                                  // every wave's slot updates are visible
        const int part = bid / n_tiles;
        for (int idx = tid; idx < BN * 2; idx += 512) {
            const int col = idx >> 1;
            if (n0 + col < N)
                ep.stats[((int64_t)part * N + n0 + col) * 2 + (idx & 1)] =
                    (sred[idx] + sred[BN * 2 + idx]) + (sred[2 * BN * 2 + idx] + sred[3 * BN * 2 + idx]);
        }
    }
    if constexpr (STATS_REG) {
        // per-lane sums -> the 16 pixel lanes of each group -> the four pixel waves, through LDS
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
            }
        }
        __syncthreads();                                  // every wave is done with the staged tiles
        float* red = reinterpret_cast<float*>(smem);      // [wm][BN channels of the tile][2]
        if (frow == 0) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                const int col = wn * (BN / 2) + fgrp * CPL + e;
                red[(wm * BN + col) * 2 + 0] = s1[e];
                red[(wm * BN + col) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        const int part = bid / n_tiles;
        for (int idx = tid; idx < BN * 2; idx += 512) {
            const int col = idx >> 1;
            if (n0 + col < N)
                ep.stats[((int64_t)part * N + n0 + col) * 2 + (idx & 1)] =
                    (red[idx] + red[BN * 2 + idx]) + (red[2 * BN * 2 + idx] + red[3 * BN * 2 + idx]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// 64 input channels, 3x3, stride 1, pad 1 (bf16): the layers at full resolution, whose implicit GEMM
// is only 576 deep (and 64 or 128 columns wide), so that the generic kernels move 1.7 KB from L2 into LDS per
// output pixel (every input pixel nine times, the filter once per 128-pixel tile).  Here the whole
// filter (9 taps x 64 x 64 = 72 KB) stays in LDS for the life of the workgroup and a tile's input is
// fetched ONCE as a halo patch: (PH+2) x (PW+2) pixels x 128 B for PH x PW = 256 output pixels
// (8 x 32 or 16 x 16), 170 B per output pixel.  A-fragments of tap (r, s) are read from the patch at
// pixel offset r*(PW+2) + s; rows are 128 B with the 16-byte chunk index XORed with (row & 7), which
// keeps ds_read_b128 conflict-free at every tap offset.  The next tile's patch is loaded into the
// second buffer while this tile's 144 MFMAs per wave run: one barrier per tile, none in the K-loop.
// One workgroup of 8 waves per CU (wave tile 64 pixels x 32 channels), persistent over the tiles.
struct HaloArgs {
    int B, H, W;
    int pw_log2;            // patch width 32 (8 rows) or 16 (16 rows)
    int tiles_w, tiles_h, ntiles;
    int N, n_tiles;         // output channels (a multiple of 64) and column tiles of 64: a workgroup keeps one column tile's filter
    // Parity classes of an upsampling forward (ncls = 4, ostep = 2): class (ph, pw) is a KR x KS = 2 x 2 convolution over the
    // stored input whose taps start at patch offset (ph, pw); pixel (i, j) is written to (2i + ph, 2j + pw) of a 2H x 2W
    // image; its filter starts w_cls elements after the previous class's.  A workgroup keeps one class's filter.
    int ncls, ostep;
    int64_t w_cls;
};
constexpr int kHaloRows = 344;                             // 10 x 34 = 340 (or 18 x 18 = 324) pixels in whole 8-row pieces
constexpr int kHaloBuf = kHaloRows * 128;
constexpr int halo_lds_bytes(int ntaps) { return ntaps * 64 * 128 + 2 * kHaloBuf + 256; }   // resident filter + two patches + the column tile's bias

template <int EPI, int KR, int KS, bool PRE = false>        // EPI: 0 plain, 1 statistics, 2 fused normalise; PRE: normalise the input
__global__ void __launch_bounds__(512, 1)
conv_c64_halo_kernel(HaloArgs ha, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, Epilogue ep,
                     void* __restrict__ yout) {
    using T = bf16_t;
    using Frag = bf16x8_t;
    constexpr bool STATS = EPI == 1, ACT = EPI == 2;
    constexpr int NT = 2, CPL = 8, NTAPS = KR * KS;
    const int N = ha.N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wt = smem;
    unsigned char* halo = smem + NTAPS * 8192;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int PW = 1 << ha.pw_log2, PH = 256 >> ha.pw_log2, HWd = PW + 2;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int combos = ha.n_tiles * ha.ncls;
    const int n0 = (bid % ha.n_tiles) * 64;                // this workgroup's column tile and class, fixed
    const int cls = (bid % combos) / ha.n_tiles, ph = cls >> 1, pw = cls & 1;
    const int npix = (PH + 2) * HWd;

    // ---- resident filter: tap tile t = 64 rows x 128 B in the row order of the generic kernels' filter
    // tiles (each lane ends up with 8 consecutive channels); thread copies piece `slot` of row srow
    {
        const int srow = tid >> 3, slot = tid & 7;         // 64 rows x 8 pieces
        const int h = srow >> 5, q = srow & 31;
        const int n = n0 + h * 32 + ((q & 15) >> 2) * CPL + 4 * (q >> 4) + (q & 3);
        const int chunk = slot ^ (srow & 7);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(w + cls * ha.w_cls);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
            load16_lds(rw, (uint32_t)((n * NTAPS + t) * 64 + chunk * 8) * 2u, true, wt + t * 8192 + wave_s * 1024);
    }

    // the column tile's bias, for the accumulators' start values
    float* bias_lds = reinterpret_cast<float*>(halo + 2 * kHaloBuf);
    if (tid < 64) bias_lds[tid] = ep.bias ? ep.bias[n0 + tid] : 0.f;
    __syncthreads();

    // ---- halo staging: wave `wave` copies pieces wave, wave + 8, ... (8 patch pixels x 128 B each)
    constexpr int NPC = (kHaloRows / 8 + 7) / 8;            // 6 pieces per wave
    int h_rc[NPC];                                        // (patch row << 8 | patch column) of this lane's pixel, -1: none
    const int h_chunk = ((lane & 7) ^ ((lane >> 3) & 7)) * 8;   // the same for every piece: (hp & 7) == (lane >> 3) & 7
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
        const int hp = (wave + 8 * j) * 8 + (lane >> 3);
        const int hr = hp / HWd, hc = hp - hr * HWd;
        h_rc[j] = hp < npix ? (hr << 8) | hc : -1;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x);
    uint32_t staged_ok = 0;                               // PRE: bit j = piece j of the patch staged last holds a pixel of the image
    auto stage = [&](int tile, int buf) {
        const int tx = tile % ha.tiles_w;
        const int ty = (tile / ha.tiles_w) % ha.tiles_h;
        const int b = tile / (ha.tiles_w * ha.tiles_h);
        const int h0 = ty * PH - 1, w0 = tx * PW - 1;
        staged_ok = 0;
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            if ((wave_s + 8 * j) * 8 >= kHaloRows) continue;               // uniform: no such piece
            const int hi = h0 + (h_rc[j] >> 8), wi = w0 + (h_rc[j] & 255);
            const bool ok = h_rc[j] >= 0 && (unsigned)hi < (unsigned)ha.H && (unsigned)wi < (unsigned)ha.W;
            const int off = ((b * ha.H + hi) * ha.W + wi) * 64 + h_chunk;
            load16_lds(rx, (uint32_t)off * 2u, ok, halo + buf * kHaloBuf + (wave_s + 8 * j) * 1024);
            if constexpr (PRE) staged_ok |= ok ? 1u << j : 0u;
        }
    };
    // PRE: a lane normalises the 16-byte slots it loaded itself (its own loads have landed once its vmcnt is 0: no
    // barrier in between).  Slot (lane & 7) of row (lane >> 3) of every piece holds the same eight channels, so their
    // sixteen coefficients stay in registers; slots of padding pixels keep the zeros the hardware wrote.
    f32x2_t psc[4], psh[4];
    if constexpr (PRE) {
        const int c0 = h_chunk;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            psc[e] = *reinterpret_cast<const f32x2_t*>(ep.pre_scale + c0 + 2 * e);
            psh[e] = *reinterpret_cast<const f32x2_t*>(ep.pre_shift + c0 + 2 * e);
        }
    }
    auto normalise = [&](int buf, uint32_t okbits) {
        // every wave holds pieces 0 .. NPC-2; they go through together (independent chains of arithmetic), the last piece
        // exists for the first waves only
        unsigned char* base = halo + buf * kHaloBuf + wave_s * 1024 + lane * 16;
        u32x4_t v[NPC - 1];
#pragma unroll
        for (int j = 0; j < NPC - 1; ++j) v[j] = *reinterpret_cast<const u32x4_t*>(base + j * 8192);
#pragma unroll
        for (int j = 0; j < NPC - 1; ++j) v[j] = norm_act_piece(v[j], psc, psh, ep.pre_slope, 0u - ((okbits >> j) & 1u));
#pragma unroll
        for (int j = 0; j < NPC - 1; ++j) *reinterpret_cast<u32x4_t*>(base + j * 8192) = v[j];
        if ((wave_s + 8 * (NPC - 1)) * 8 < kHaloRows) {                   // uniform
            u32x4_t* slot = reinterpret_cast<u32x4_t*>(base + (NPC - 1) * 8192);
            *slot = norm_act_piece(*slot, psc, psh, ep.pre_slope, 0u - ((okbits >> (NPC - 1)) & 1u));
        }
    };

    // ---- fragments
    const int frow = lane & 15, fgrp = lane >> 4;
    int hp0[4];                                           // patch pixel of this lane in pixel tile i, as a halo row of tap (0, 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wm * 64 + i * 16 + frow;
        hp0[i] = (q >> ha.pw_log2) * HWd + (q & (PW - 1));
    }
    const unsigned char* sb = wt + (wn * 32 + frow) * 128;
    const int bsw = frow & 7;
    f32x4_t acc[NT][4];
    const int nb = n0 + wn * 32 + fgrp * CPL;
    float s1[STATS ? CPL : 1], s2[STATS ? CPL : 1];
#pragma unroll
    for (int e = 0; e < (STATS ? CPL : 1); ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // Accumulators start from the bias (as in the other kernels: the same order of additions, bit for bit), read from
    // LDS: an accumulator that starts from a value loaded from memory makes the first MFMA of every tile wait for that
    // load, and since the compiler cannot count past the direct-to-LDS loads issued just before, it waits with vmcnt(0)
    // -- the NEXT patch's loads would have to land before this tile's MFMAs may start.
    auto init_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float4 t4 = *reinterpret_cast<const float4*>(bias_lds + nb - n0 + 4 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = f32x4_t{t4.x, t4.y, t4.z, t4.w};
        }
    };

    const int nblk = gridDim.x / combos;                 // tiles advance by this
    int tile = bid / combos;
    int buf = 0;
    if (tile < ha.ntiles) stage(tile, 0);
    init_acc();
    // Waits on vmcnt are written as the builtin, which the compiler's own wait insertion sees (vmcnt(0), other counters
    // untouched): behind an asm wait it takes earlier loads for pending and orders the direct-to-LDS loads of a patch behind
    // them one by one.  A patch is waited for BEFORE the epilogue of the tile in front of it (its loads are a whole tap
    // loop old by then), not at the top of its own tile: there the wait would also cover that epilogue's stores, and
    // every tile would stand still for a store's round trip to L2.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (; tile < ha.ntiles; tile += nblk) {
        if constexpr (PRE) {                              // own slots: this wave's loads have landed, no barrier needed yet
            normalise(buf, staged_ok);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                     // this tile's patch (and the filter) is in LDS; every wave is past the previous tile
        if (tile + nblk < ha.ntiles) stage(tile + nblk, buf ^ 1);
        // ---- this tile's rows, and its epilogue operands: loaded now, used after the 144 MFMAs
        uint32_t offs[4];                                 // element offsets: outputs are below 2 GiB (checked on the host)
        bool valid[4];
        int img;
        {
            const int tx = tile % ha.tiles_w;
            const int ty = (tile / ha.tiles_w) % ha.tiles_h;
            img = tile / (ha.tiles_w * ha.tiles_h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = wm * 64 + i * 16 + frow;
                const int h = ty * PH + (q >> ha.pw_log2), wq = tx * PW + (q & (PW - 1));
                valid[i] = h < ha.H && wq < ha.W;
                offs[i] = ha.ostep ? (uint32_t)(((img * 2 * ha.H + 2 * h + ph) * 2 * ha.W + 2 * wq + pw) * N + nb)
                                   : (uint32_t)(((img * ha.H + h) * ha.W + wq) * N + nb);
            }
        }
        float tv[CPL];
#pragma unroll
        for (int e = 0; e < CPL; ++e) tv[e] = 0.f;
        if (ep.temb) {
            const float4 t0 = *reinterpret_cast<const float4*>(ep.temb + (int64_t)img * N + nb);
            const float4 t1 = *reinterpret_cast<const float4*>(ep.temb + (int64_t)img * N + nb + 4);
            tv[0] = t0.x; tv[1] = t0.y; tv[2] = t0.z; tv[3] = t0.w; tv[4] = t1.x; tv[5] = t1.y; tv[6] = t1.z; tv[7] = t1.w;
        }
        const T* r1 = reinterpret_cast<const T*>(ep.res ? ep.res : ep.res2);
        const T* r2 = reinterpret_cast<const T*>(ep.res ? ep.res2 : nullptr);
        Vec16<T> rv[4], rw[4];                             // both terms before the epilogue: a load issued between its stores would
        if (r1) {                                         // have to wait for the stores before it
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (valid[i]) rv[i] = *reinterpret_cast<const Vec16<T>*>(r1 + offs[i]);
        }
        if (r2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (valid[i]) rw[i] = *reinterpret_cast<const Vec16<T>*>(r2 + offs[i]);
        }

        const unsigned char* sa = halo + buf * kHaloBuf;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int toff = (ph + t / KS) * HWd + pw + (t % KS);
            // The fragment addresses of all nine taps are the same for every tile; hoisted out of the tile loop they would
            // take 72 registers, which this kernel does not have (the epilogue's operands would be spilled): hp0 passes through
            // an empty asm once per tap, so the addresses are formed where they are used (three vector instructions a read).
            int arow[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int h = hp0[i];
                asm volatile("" : "+v"(h));
                arow[i] = h + toff;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag a[4], b[NT];
                const int c = ks * 4 + fgrp;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const Frag*>(sa + arow[i] * 128 + ((c ^ (arow[i] & 7)) << 4));
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    b[j] = *reinterpret_cast<const Frag*>(sb + t * 8192 + j * 16 * 128 + ((c ^ bsw) << 4));
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[j][i], 0, 0, 0);
            }
        }
        // ---- epilogue of this tile: y = acc + temb + res + res2 (in this order), rounded, stored, summed
        __builtin_amdgcn_s_waitcnt(0x0F70);               // the next patch and this tile's terms have landed; no store is pending yet
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!valid[i]) continue;
            float v[CPL];
#pragma unroll
            for (int e = 0; e < CPL; ++e) v[e] = acc[e >> 2][i][e & 3] + tv[e];
            if (r1) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) v[e] += to_f(rv[i].v[e]);
            }
            if (r2) {
#pragma unroll
                for (int e = 0; e < CPL; ++e) v[e] += to_f(rw[i].v[e]);
            }
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < CPL; ++e) ov.v[e] = from_f<T>(v[e]);
            if constexpr (ACT) {                          // the consumer's eval-mode normalise + LeakyReLU of the rounded value
                if (ep.raw) *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(ep.raw) + offs[i]) = ov;
                const float4 a0 = *reinterpret_cast<const float4*>(ep.act_scale + nb), a1 = *reinterpret_cast<const float4*>(ep.act_scale + nb + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(ep.act_shift + nb), b1 = *reinterpret_cast<const float4*>(ep.act_shift + nb + 4);
                const float asc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float ash[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < CPL; ++e) ov.v[e] = from_f<T>(leaky(fmaf(to_f(ov.v[e]), asc[e], ash[e]), ep.act_slope));
            }
            *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(yout) + offs[i]) = ov;
            if constexpr (STATS) {                        // statistics of the rounded values a later pass would read
#pragma unroll
                for (int e = 0; e < CPL; ++e) {
                    const float q = to_f(ov.v[e]);
                    s1[e] += q;
                    s2[e] = fmaf(q, q, s2[e]);
                }
            }
        }
        init_acc();
        buf ^= 1;
    }
    if constexpr (STATS) {
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
            }
        }
        __syncthreads();                                  // every wave is done with the patches
        float* red = reinterpret_cast<float*>(halo);      // [wm][64 channels of the column tile][2]
        if (frow == 0) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                red[(wm * 64 + nb - n0 + e) * 2 + 0] = s1[e];
                red[(wm * 64 + nb - n0 + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < 64 * 2; idx += 512)
            ep.stats[((int64_t)(bid / ha.n_tiles) * N + n0) * 2 + idx] =
                (red[idx] + red[64 * 2 + idx]) + (red[2 * 64 * 2 + idx] + red[3 * 64 * 2 + idx]);
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient: dW[n][f] += sum over a slice of pixels m of dY[m][n] * Xg[m][f].
// Both operands are "reduction-major" in memory (channel innermost), so their MFMA fragments are
// read with the transposing LDS read ds_read_b64_tr_b16 (bf16) / plain ds_read_b32 (fp32).
// LDS stage: dY tile [KP pixels][AN n] and Xg tile [KP pixels][BF f].  Two shapes:
//   AN = 128, BF = 128: waves 2 x 2, each 64 n x 64 f                       (Cout >= 128, attention)
//   AN =  64, BF = 192: waves 1 x 4, each 64 n x 48 f  (9*Cin is a multiple of 192 for Cin % 64 == 0)
// Grid: (n tiles * f tiles, split-K slices); partial tiles are combined with fp32 atomics.
// ds_read_b64_tr_b16 of the 4 rows x 16 columns block this lane's 16-lane group addresses; the
// result is valid only after a later s_waitcnt lgkmcnt (the compiler does not know this is an LDS read)
template <int OFF> __device__ __forceinline__ s16x4_t lds_read_tr16(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "immediate offset of a DS instruction");
    s16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <int ROWB> __device__ __forceinline__ int tn_swizzle(int row) {
    // XOR applied to the 16-byte piece index of a row so that the 8 rows x 32 bytes that 32 lanes
    // fetch with one transposing read fall on 8 different bank groups
    if constexpr (ROWB % 256 == 0) return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
    else return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;      // ROWB = odd multiple of 128
}

template <typename T, int AN, int BF, int NW = 4, bool PIPE = false>   // NW waves: 4 (two workgroups per CU) or 8 (one)
__global__ void __launch_bounds__(64 * NW, NW == 4 ? 2 : 1)
conv_tn_mfma_kernel(Gather gp, const T* __restrict__ dy, const T* __restrict__ x, int N, int lda, int ldc,
                    float* __restrict__ dw, int n_tiles, int f_tiles, int steps_per_slice,
                    int64_t strideA, int64_t strideB, int64_t strideC, int out_mode, void* yout,
                    float alpha, float* __restrict__ dbias, int bias_mode) {
    constexpr int PIECE = ElemTraits<T>::PIECE;
    constexpr bool BF16 = sizeof(T) == 2;
    constexpr int KP = BF16 ? 64 : 32;                    // pixels per K-step
    constexpr int ROWB_A = AN * sizeof(T), ROWB_B = BF * sizeof(T);
    constexpr int PPR_A = ROWB_A / 16, PPR_B = ROWB_B / 16;
    constexpr int TILE_A = KP * ROWB_A, TILE_B = KP * ROWB_B;
    constexpr int STAGE = TILE_A + TILE_B;
    constexpr int NT = 64 * NW;                           // threads
    constexpr int NPASS_A = KP * PPR_A / NT, NPASS_B = KP * PPR_B / NT;
    constexpr int WM = AN / 64, WN = NW / WM;             // waves along n / along f
    constexpr int NJ = BF / (16 * WN);                    // 16-wide f tiles per wave
    static_assert(KP * PPR_A % NT == 0 && KP * PPR_B % NT == 0 && BF % (16 * WN) == 0 && NW % WM == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // workgroups that share an XCD (and its L2) take consecutive (slice, tile) pairs: the tiles of
    // one pixel slice read the same dY / X rows
    const int n_ft = n_tiles * f_tiles;
    const int rid = xcd_remap(blockIdx.x, gridDim.x);
    const int slice = rid / n_ft, bid = rid - slice * n_ft;
    const int nt = bid / f_tiles, ft = bid - nt * f_tiles;
    const int n0 = nt * AN, f0 = ft * BF;
    dy += (int64_t)blockIdx.z * strideA;
    x += (int64_t)blockIdx.z * strideB;

    // staging: in pass i this thread copies LDS piece i*NT + tid (the image of a direct-to-LDS
    // load is lane-linear); the swizzle is applied to the piece it fetches.  The pixel a pass reads
    // advances by KP per K-step, so its (image, row, column) and its dY offset are carried along
    // with adds and single-wrap compares; nothing is divided inside the loop.
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int step0 = slice * steps_per_slice;
    const int total_steps = (gp.M + KP - 1) / KP;
    int step1 = step0 + steps_per_slice;
    if (step1 > total_steps) step1 = total_steps;
    const int kq_w = KP / gp.Wo, kr_w = KP - kq_w * gp.Wo;        // KP pixels = kq_w rows + kr_w columns
    const int kq_h = kq_w / gp.Ho, kr_h = kq_w - kq_h * gp.Ho;    // kq_w rows = kq_h images + kr_h rows

    int a_row[NPASS_A], a_off[NPASS_A];                   // a_row >= 2^30: nothing to fetch
#pragma unroll
    for (int i = 0; i < NPASS_A; ++i) {
        const int pi = i * NT + tid;
        const int row = pi / PPR_A, slot = pi - row * PPR_A;
        const int piece = BF16 ? (slot ^ tn_swizzle<ROWB_A>(row)) : slot;
        const int an = n0 + piece * PIECE;
        a_row[i] = (an + PIECE <= lda && an < N) ? row : (1 << 30);
        a_off[i] = (step0 * KP + row) * lda + an;
    }
    int b_row[NPASS_B], b_tr[NPASS_B], b_ts[NPASS_B], b_c[NPASS_B], b_img[NPASS_B], b_ho[NPASS_B], b_wo[NPASS_B];
#pragma unroll
    for (int i = 0; i < NPASS_B; ++i) {
        const int pi = i * NT + tid;
        const int row = pi / PPR_B, slot = pi - row * PPR_B;
        const int piece = BF16 ? (slot ^ tn_swizzle<ROWB_B>(row)) : slot;
        const int bf = f0 + piece * PIECE;                // flattened column -> (tap, c)
        const int tap = bf / gp.Cin;
        b_row[i] = row;
        b_c[i] = bf - tap * gp.Cin;
        b_tr[i] = bf < gp.Ktot ? tap / gp.k : (1 << 20);  // out of range: the bounds test rejects it
        b_ts[i] = tap - (tap / gp.k) * gp.k;
        const uint32_t m = (uint32_t)step0 * KP + row;
        const uint32_t b = fdiv(m, gp.div_howo);
        const uint32_t rem = m - b * (uint32_t)(gp.Ho * gp.Wo);
        const uint32_t ho = fdiv(rem, gp.div_wo);
        b_img[i] = (int)b;
        b_ho[i] = (int)ho;
        b_wo[i] = (int)(rem - ho * gp.Wo);
    }
    int mleft = gp.M - step0 * KP;                         // rows of the current K-step below this are real pixels
    const int sshift = gp.stride - 1;
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(dy), rxg = make_rsrc(x);

    auto stage_a = [&](int buf) {
        unsigned char* sa = smem + buf * STAGE + wave_s * 1024;
#pragma unroll
        for (int i = 0; i < NPASS_A; ++i) {
            load16_lds(rdy, (uint32_t)a_off[i] * (uint32_t)sizeof(T), a_row[i] < mleft, sa + i * (NT * 16));
            a_off[i] += KP * lda;
        }
    };
    auto stage_b = [&](int buf) {
        unsigned char* sb = smem + buf * STAGE + TILE_A + wave_s * 1024;
#pragma unroll
        for (int i = 0; i < NPASS_B; ++i) {
            const int u = (b_ho[i] << sshift) - gp.pad + b_tr[i];       // stride is 1 or 2
            const int v = (b_wo[i] << sshift) - gp.pad + b_ts[i];
            const bool ok = (b_row[i] < mleft) & ((unsigned)u < (unsigned)gp.He) & ((unsigned)v < (unsigned)gp.We);
            const uint32_t pix = __umul24(__umul24((uint32_t)b_img[i], (uint32_t)gp.Hin) + (uint32_t)(u >> gp.shift), (uint32_t)gp.Win) +
                                 (uint32_t)(v >> gp.shift);
            load16_lds(rxg, (pix * (uint32_t)gp.ldx + (uint32_t)b_c[i]) * (uint32_t)sizeof(T), ok, sb + i * (NT * 16));
            // advance this pass by KP pixels
            int wo = b_wo[i] + kr_w;
            const int cw = wo >= gp.Wo ? 1 : 0;
            b_wo[i] = wo - (cw ? gp.Wo : 0);
            int ho = b_ho[i] + kr_h + cw;
            const int ch = ho >= gp.Ho ? 1 : 0;
            b_ho[i] = ho - (ch ? gp.Ho : 0);
            b_img[i] += kq_h + ch;
        }
        mleft -= KP;
    };

    f32x4_t acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // bias gradient: the workgroups of the first filter-column tile also multiply dY^T by a column
    // of ones (every column of the product is the per-channel pixel sum)
    // bias_mode 2: pixel sums of the gathered operand's columns instead (ones^T x B), by the waves of
    // the first dY-column tile: used when the roles of the operands are swapped
    const bool do_bias = dbias != nullptr && bias_mode == 1 && ft == 0 && wn == 0;      // wave-uniform
    const bool do_bsum = dbias != nullptr && bias_mode == 2 && nt == 0 && wm == 0;
    f32x4_t accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // bf16: one 16x16x32 MFMA k-step covers 32 pixels; lane l supplies row k0 + ((l&15)>>2),
    // columns c0 + 4*(l&3) to the transposing read.  The reads are issued as inline assembly: the
    // compiler orders its own ds_read_b64_tr_b16 after every outstanding direct-to-LDS load
    // (s_waitcnt vmcnt(0)), which would serialise the prefetch of the next K-step behind them.
    // Each read's address is a per-lane constant plus immediates for (k-step half, row group).
    uint32_t a_addr[4], b_addr[NJ];
    if constexpr (BF16) {
        const int q = (lane & 15) >> 2, p = lane & 3, g = lane >> 4;
        const int row0 = g * 8 + q;
        const uint32_t lds0 = (uint32_t)(uintptr_t)LPTR(smem);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = wm * 64 + i * 16 + p * 4;
            a_addr[i] = lds0 + row0 * ROWB_A + ((col >> 3) ^ tn_swizzle<ROWB_A>(row0)) * 16 + (col & 7) * 2;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = wn * (16 * NJ) + j * 16 + p * 4;
            b_addr[j] = lds0 + TILE_A + row0 * ROWB_B + ((col >> 3) ^ tn_swizzle<ROWB_B>(row0)) * 16 + (col & 7) * 2;
        }
    }
    // raw[..][h]: rows +0..3 (h = 0) and +4..7 (h = 1) of a fragment; valid only after frag_wait
    auto read_frags = [&](int buf, auto ks_c, s16x4_t (&ra)[4][2], s16x4_t (&rb)[NJ][2]) {
        constexpr int ks = decltype(ks_c)::value;
        const uint32_t boff = buf * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i][0] = lds_read_tr16<ks * 32 * ROWB_A>(a_addr[i] + boff);
            ra[i][1] = lds_read_tr16<ks * 32 * ROWB_A + 4 * ROWB_A>(a_addr[i] + boff);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            rb[j][0] = lds_read_tr16<ks * 32 * ROWB_B>(b_addr[j] + boff);
            rb[j][1] = lds_read_tr16<ks * 32 * ROWB_B + 4 * ROWB_B>(b_addr[j] + boff);
        }
    };
    auto frag_wait = [&](s16x4_t (&ra)[4][2], s16x4_t (&rb)[NJ][2], bf16x8_t (&a)[4], bf16x8_t (&b)[NJ]) {
        // every later use of the raw registers depends on this wait
        if constexpr (NJ == 4) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                           "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1]));
        } else {
            static_assert(NJ == 3 || NJ == 4, "fragment wait is written for 3 or 4 column tiles");
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                           "+v"(rb[2][0]), "+v"(rb[2][1]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(ra[i][0], ra[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            b[j] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(rb[j][0], rb[j][1], 0, 1, 2, 3, 4, 5, 6, 7));
    };
    // as frag_wait, but lets the 8 + 2 * NJ reads issued after these stay in flight (the counter holds at most 15)
    auto frag_wait_first = [&](s16x4_t (&ra)[4][2], s16x4_t (&rb)[NJ][2], bf16x8_t (&a)[4], bf16x8_t (&b)[NJ]) {
        if constexpr (NJ == 4) {
            asm volatile("s_waitcnt lgkmcnt(15)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                           "+v"(rb[2][0]), "+v"(rb[2][1]), "+v"(rb[3][0]), "+v"(rb[3][1]));
        } else {
            asm volatile("s_waitcnt lgkmcnt(14)"
                         : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]),
                           "+v"(ra[3][0]), "+v"(ra[3][1]), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]),
                           "+v"(rb[2][0]), "+v"(rb[2][1]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a[i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(ra[i][0], ra[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            b[j] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(rb[j][0], rb[j][1], 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](const bf16x8_t (&a)[4], const bf16x8_t (&b)[NJ]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (do_bias) {
            bf16x8_t ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
        }
        if (do_bsum) {
            bf16x8_t ones;
#pragma unroll
            for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b[j], accb[j], 0, 0, 0);
        }
    };
    // fp32: 16x16x4, lane l holds A[i = l&15][k = l>>4]; tile rows are pixels, so element (k, i)
    // sits at row k, column i: plain 4-byte reads, 16 consecutive lanes contiguous.
    auto compute_f32 = [&](int buf) {
        const unsigned char* sa = smem + buf * STAGE;
        const unsigned char* sb = sa + TILE_A;
        const int fr = lane & 15, g = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KP / 4; ++ks) {
            const int row = ks * 4 + g;
            float a[4], b[NJ];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *reinterpret_cast<const float*>(sa + row * ROWB_A + (wm * 64 + i * 16 + fr) * 4);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b[j] = *reinterpret_cast<const float*>(sb + row * ROWB_B + (wn * (16 * NJ) + j * 16 + fr) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], 1.0f, accb[i], 0, 0, 0);
            }
            if (do_bsum) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) accb[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, b[j], accb[j], 0, 0, 0);
            }
        }
    };
    auto compute = [&](int buf) {
        if constexpr (BF16) {
            s16x4_t ra[4][2], rb[NJ][2];
            bf16x8_t a[4], b[NJ];
            read_frags(buf, std::integral_constant<int, 0>{}, ra, rb);
            frag_wait(ra, rb, a, b);
            mma(a, b);
            read_frags(buf, std::integral_constant<int, 1>{}, ra, rb);
            frag_wait(ra, rb, a, b);
            mma(a, b);
        } else {
            compute_f32(buf);
        }
    };

    if (step0 < step1) {
        stage_a(0);
        stage_b(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // steady state: fragment reads of K-step t, address arithmetic + direct-to-LDS loads of
        // K-step t+1 between them and the MFMAs (LDS reads and LDS-DMA loads keep program order)
        for (int t = step0; t + 1 < step1; ++t) {
            const int cur = (t - step0) & 1, nxt = cur ^ 1;
            if constexpr (BF16) {
                s16x4_t ra[4][2], rb[NJ][2];
                bf16x8_t a[4], b[NJ];
                // measured: both halves of the next stage issued before the first MFMA half beat
                // issuing the second half later
                read_frags(cur, std::integral_constant<int, 0>{}, ra, rb);
#if VAMD_ABLATE != 8
                stage_a(nxt);
                stage_b(nxt);
#endif
                if constexpr (PIPE) {
                    // the second half's fragment reads are in flight during the first half's MFMAs: LDS reads return in
                    // order, so once at most as many are outstanding as the second half issued, the first half has landed
                    s16x4_t ra2[4][2], rb2[NJ][2];
                    read_frags(cur, std::integral_constant<int, 1>{}, ra2, rb2);
                    frag_wait_first(ra, rb, a, b);
                    mma(a, b);
                    frag_wait(ra2, rb2, a, b);
                    mma(a, b);
                } else {
                    frag_wait(ra, rb, a, b);
                    mma(a, b);
                    read_frags(cur, std::integral_constant<int, 1>{}, ra, rb);
                    frag_wait(ra, rb, a, b);
                    mma(a, b);
                }
            } else {
                stage_a(nxt);
                stage_b(nxt);
                compute_f32(cur);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        compute((step1 - 1 - step0) & 1);
    }

    // D layout: column (second operand index) = lane & 15, row (first operand index) = 4*(lane>>4)+r
    const int fr = lane & 15, fq = lane >> 4;
    // Split-K without atomics: every pixel slice stores its own partial product [slice][N][ldc]
    // (and partial bias sums [slice][N]); wgrad_reduce_kernel adds the slices in a fixed order.
    if (do_bias && fr == 0) {
        float* pb = dbias + (int64_t)slice * N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nrow = n0 + wm * 64 + i * 16 + fq * 4 + r;
                if (nrow < N) pb[nrow] = accb[i][r];
            }
    }
    if (do_bsum && fq == 0) {                             // every row of ones^T x B is the column sum
        float* pb = dbias + (int64_t)slice * gp.Ktot;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int fcol = f0 + wn * (16 * NJ) + j * 16 + fr;
            if (fcol < gp.Ktot) pb[fcol] = accb[j][0];
        }
    }
    if (out_mode == 0) {
        float* part = dw + (int64_t)slice * N * ldc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int fcol = f0 + wn * (16 * NJ) + j * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nrow = n0 + wm * 64 + i * 16 + fq * 4 + r;
                    if (nrow < N && fcol < gp.Ktot) part[(int64_t)nrow * ldc + fcol] = acc[i][j][r];
                }
            }
    } else {
        // plain batched TN product (attention): C[z][n][f] = alpha * acc, stored in T, no split-K
        T* out = reinterpret_cast<T*>(yout) + (int64_t)blockIdx.z * strideC;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int fcol = f0 + wn * (16 * NJ) + j * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nrow = n0 + wm * 64 + i * 16 + fq * 4 + r;
                    if (nrow < N && fcol < gp.Ktot)
                        out[(int64_t)nrow * ldc + fcol] = from_f<T>(alpha * acc[i][j][r]);
                }
            }
    }
}

// ------------------------------------------------------------------------------------ host side
static Gather make_gather(int Hin, int Win, int Cin, int He, int We, int Ho, int Wo, int k, int stride,
                          int pad, int shift, int64_t M) {
    Gather g;
    g.Hin = Hin; g.Win = Win; g.Cin = Cin; g.ldx = Cin; g.He = He; g.We = We; g.Ho = Ho; g.Wo = Wo;
    g.k = k; g.stride = stride; g.pad = pad; g.padw = pad; g.shift = shift;
    g.M = (int)M; g.Ktot = k * k * Cin;
    g.div_howo = make_fastdiv((uint32_t)(Ho * Wo));
    g.div_wo = make_fastdiv((uint32_t)Wo);
    g.kr = k; g.ks = k; g.wr0 = 0; g.ws0 = 0; g.wstep = 1; g.wk = k;
    g.oH = 0; g.oW = 0; g.ostep = 0; g.oph = 0; g.opw = 0;
    g.x2 = nullptr; g.w2 = nullptr; g.Cin2 = 0;
    return g;
}

template <typename K> static int set_lds_limit(K kernel, int bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%d) failed: %s", who, bytes, hipGetErrorString(e));
    return 0;
}

static bool nt_shape_ok(int Cin, int N, int64_t M, int dtype) {
    const int piece = dtype == VAMD_BF16 ? 8 : 4;
    const int kstep = dtype == VAMD_BF16 ? 64 : 32;
    // a piece must not straddle two taps, and the tap state advances at most twice per K-step
    return Cin % piece == 0 && 2 * Cin >= kstep && N % piece == 0 && M < (1ll << 26) && M > 0;
}
static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// Grid of the persistent kernel: as many workgroups as fit the chip at once (2 or 3 per CU), a whole
// number of column-tile groups, and row tiles spread evenly over them.
static int ntp_row_groups(int row_tiles, int n_tiles, int per_cu) {
    // VAMD_NT_CUS: compute units the persistent kernels spread over (experiments: what a tile costs on a partly idle chip)
    static const int cus = [] { const char* e = getenv("VAMD_NT_CUS"); return e && atoi(e) > 0 ? atoi(e) : 256; }();
    int rmax = cus * per_cu / n_tiles;
    if (rmax < 1) rmax = 1;
    const int rounds = (row_tiles + rmax - 1) / rmax;
    return (row_tiles + rounds - 1) / rounds;             // workgroups per column tile; each walks `rounds` row tiles
}

static constexpr int ntp_per_cu(int BN) { return BN == 64 ? 3 : 2; }   // resident workgroups per CU (registers, LDS)

static bool nt_classic() {
    static const bool v = [] { const char* e = getenv("VAMD_NT_CLASSIC"); return e && e[0] == '1'; }();
    return v;
}

// The 256-row kernel serves wide-channel layers from VAMD_NT_WIDE_MIN tiles on; VAMD_NT_WIDE=0
// switches it off (side-by-side measurements).
static bool nt_wide(int BN, int64_t M, int batch, int n_tiles) {
    static const int force = [] { const char* e = getenv("VAMD_NT_WIDE"); return e ? atoi(e) : -1; }();
    // 128-row tiles for layers with few 256-row tiles win in isolation (4 x 64 x 64 x 256 channels: 37.9 -> 29.8 us) but
    // lose inside the training step and the sampler (flowers 8.04 -> 8.14 / 8.18 ms with thresholds of 60 / 110 tiles)
    static const int min_tiles = [] { const char* e = getenv("VAMD_NT_WIDE_MIN"); return e ? atoi(e) : 1; }();
    if (BN != 128 || nt_classic() || force == 0) return false;
    return ((M + 255) / 256) * batch * n_tiles >= min_tiles;
}

// Non-temporal stores for outputs from VAMD_NT_STORE_MB megabytes on (see Epilogue::nt_store).
static int nt_store_for(int64_t out_bytes) {
    static const int mb = [] { const char* e = getenv("VAMD_NT_STORE_MB"); return e ? atoi(e) : 40; }();
    return out_bytes >= (int64_t)mb << 20;
}

// The launch-timing records count ALGORITHMIC FLOPs: what the reference's operator does (2*M*N*K of the
// convolution as written in the model).  The combined-filter forms of the upsampling conv execute
// 16/36 of that; their host functions announce the operator's count here for the next launch.
static thread_local double t_nominal_flops = 0.0;
static double take_flops(double executed) {
    const double f = t_nominal_flops > 0.0 ? t_nominal_flops : executed;
    t_nominal_flops = 0.0;
    return f;
}

// Block ranges of a class set.  A tile costs its K-steps plus a fixed part (set-up, epilogue) worth
// about `ovh` reduction elements; every class gets the fewest workgroups per column tile that finish
// it within a common time budget, the smallest budget for which all of them fit on the chip at once.
static int plan_class_blocks(ClassSet& cs, int N, int n_tiles, int bm, int slots_per_column, double* flops) {
    static const int ovh = [] { const char* e = getenv("VAMD_CLASS_OVH"); return e ? atoi(e) : 256; }();
    int64_t cost[4], hi = 0;
    for (int c = 0; c < cs.ncls; ++c) {
        cs.m_tiles[c] = (cs.g[c].M + bm - 1) / bm;
        cost[c] = cs.g[c].Ktot + ovh;
        if (cost[c] * cs.m_tiles[c] > hi) hi = cost[c] * cs.m_tiles[c];
        if (flops) *flops += 2.0 * cs.g[c].M * N * cs.g[c].Ktot;
    }
    auto groups_for = [&](int c, int64_t budget) {
        int64_t per = budget / cost[c];
        if (per < 1) per = 1;
        return (int)((cs.m_tiles[c] + per - 1) / per);
    };
    int64_t lo = 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2;
        int need = 0;
        for (int c = 0; c < cs.ncls; ++c) need += groups_for(c, mid);
        if (need <= slots_per_column) hi = mid; else lo = mid + 1;
    }
    cs.first_block[0] = 0;
    for (int c = 0; c < 4; ++c)
        cs.first_block[c + 1] = cs.first_block[c] + (c < cs.ncls ? groups_for(c, lo) * n_tiles : 0);
    return cs.first_block[4];
}

template <typename T, int BN, int UP>
static int launch_nt(const Gather& gp, const T* x, const T* w, int N, int ldw, int ldc, const Epilogue& ep,
                     void* y, int batch, int64_t sA, int64_t sB, int64_t sC, hipStream_t st,
                     const ClassSet* classes = nullptr) {
    constexpr int LDS_PIPE = 2 * (kBM * kRowBytes + BN * kRowBytes);
    ClassSet cs;
    cs.ncls = 0;
    auto plan_classes = [&](int bm, int slots_per_column, double* flops) {
        cs = *classes;
        return plan_class_blocks(cs, N, (N + BN - 1) / BN, bm, slots_per_column, flops);
    };
    constexpr int LDS_EPI = kBM * (BN + 4) * 4;
    constexpr int LDS = LDS_PIPE > LDS_EPI ? LDS_PIPE : LDS_EPI;
    const int m_tiles = (gp.M + kBM - 1) / kBM, n_tiles = (N + BN - 1) / BN;
    if (nt_classic()) {
        static bool configured = false;
        if (!configured) {
            int rc = set_lds_limit(conv_nt_mfma_kernel<T, BN, UP == 1>, LDS, "conv_nt_mfma");
            if (rc) return rc;
            configured = true;
        }
        dim3 grid(m_tiles * n_tiles, 1, batch);
        hipLaunchKernelGGL((conv_nt_mfma_kernel<T, BN, UP == 1>), grid, dim3(256), LDS, st, gp, x, w, N, ldw, ldc, ep, y,
                           m_tiles, n_tiles, sA, sB, sC);
        VAMD_LAUNCH_OK("conv_nt_mfma");
        return 0;
    }
    if constexpr (BN == 128) {
        if (nt_wide(BN, gp.M, batch, n_tiles)) {
            constexpr int LDS_W = 3 * (256 * kRowBytes + BN * kRowBytes) + BN * (int)sizeof(float);   // stages + the column tile's bias
            static bool configured_w = false;
            if (!configured_w) {
                int rc = set_lds_limit(conv_ntw_mfma_kernel<T, BN, UP, false>, LDS_W, "conv_ntw_mfma");
                if (!rc) rc = set_lds_limit(conv_ntw_mfma_kernel<T, BN, UP, true>, LDS_W, "conv_ntw_mfma");
                if constexpr (UP == 0) {
                    if (!rc) rc = set_lds_limit(conv_ntw_mfma_kernel<T, BN, UP, 2>, LDS_W, "conv_ntw_mfma");
                }
                if (rc) return rc;
                configured_w = true;
            }
            const int mw_tiles = (gp.M + 255) / 256;
            const int row_tiles = mw_tiles * batch;
            int blocks = ntp_row_groups(row_tiles, n_tiles, 1) * n_tiles;
            double flops = classes ? 0.0 : 2.0 * gp.M * N * gp.Ktot * batch;
            if (classes) {
                blocks = plan_classes(256, 256 / n_tiles > 0 ? 256 / n_tiles : 1, &flops);
                for (int c = 0; c < cs.ncls; ++c) cs.stats_off[c] = (int64_t)(cs.first_block[c] / n_tiles) * N * 2;
            }
            Epilogue epw = ep;
            epw.nt_store = nt_store_for((int64_t)gp.M * batch * (classes ? cs.ncls : 1) * ldc * (int64_t)sizeof(T));
            const Epilogue& ep = epw;
            TimedLaunch timed(TIMED_NT, take_flops(flops), flops, st);
            if (ep.act_scale) {
                if constexpr (UP == 0)
                    hipLaunchKernelGGL((conv_ntw_mfma_kernel<T, BN, UP, 2>), dim3(blocks), dim3(512), LDS_W, st, gp,
                                       x, w, N, ldw, ldc, ep, y, mw_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
                else
                    return fail(VAMD_EUNSUPPORTED, "conv_ntw_mfma: fused normalise with this gather");
            } else if (ep.stats)
                hipLaunchKernelGGL((conv_ntw_mfma_kernel<T, BN, UP, true>), dim3(blocks), dim3(512), LDS_W, st, gp,
                                   x, w, N, ldw, ldc, ep, y, mw_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
            else
                hipLaunchKernelGGL((conv_ntw_mfma_kernel<T, BN, UP, false>), dim3(blocks), dim3(512), LDS_W, st, gp,
                                   x, w, N, ldw, ldc, ep, y, mw_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
            VAMD_LAUNCH_OK("conv_ntw_mfma");
            return 0;
        }
    }
    static bool configured_p = false;
    if (!configured_p) {
        int rc = set_lds_limit(conv_ntp_mfma_kernel<T, BN, UP, false>, LDS_PIPE, "conv_ntp_mfma");
        if (!rc) rc = set_lds_limit(conv_ntp_mfma_kernel<T, BN, UP, true>, LDS_PIPE, "conv_ntp_mfma");
        if constexpr (UP == 0) {
            if (!rc) rc = set_lds_limit(conv_ntp_mfma_kernel<T, BN, UP, 2>, LDS_PIPE, "conv_ntp_mfma");
        }
        if (rc) return rc;
        configured_p = true;
    }
    const int row_tiles = m_tiles * batch;
    int blocks = ntp_row_groups(row_tiles, n_tiles, ntp_per_cu(BN)) * n_tiles;
    double flops = classes ? 0.0 : 2.0 * gp.M * N * gp.Ktot * batch;
    if (classes) {
        const int per_column = 256 * ntp_per_cu(BN) / n_tiles;
        blocks = plan_classes(kBM, per_column > 0 ? per_column : 1, &flops);
        for (int c = 0; c < cs.ncls; ++c) cs.stats_off[c] = (int64_t)(cs.first_block[c] / n_tiles) * N * 2;
    }
    TimedLaunch timed(TIMED_NT, take_flops(flops), flops, st);
    if (ep.act_scale) {
        if constexpr (UP == 0)
            hipLaunchKernelGGL((conv_ntp_mfma_kernel<T, BN, UP, 2>), dim3(blocks), dim3(256), LDS_PIPE, st, gp, x,
                               w, N, ldw, ldc, ep, y, m_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
        else
            return fail(VAMD_EUNSUPPORTED, "conv_ntp_mfma: fused normalise with this gather");
    } else if (ep.stats)
        hipLaunchKernelGGL((conv_ntp_mfma_kernel<T, BN, UP, true>), dim3(blocks), dim3(256), LDS_PIPE, st, gp, x,
                           w, N, ldw, ldc, ep, y, m_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
    else
        hipLaunchKernelGGL((conv_ntp_mfma_kernel<T, BN, UP, false>), dim3(blocks), dim3(256), LDS_PIPE, st, gp, x,
                           w, N, ldw, ldc, ep, y, m_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
    VAMD_LAUNCH_OK("conv_ntp_mfma");
    return 0;
}

// 256 x 256 tiles for bf16 layers with a multiple of 256 output channels; VAMD_NT_WIDE256=0 switches it off.
// Measured inside the training step: it pays for long reductions (3x3 over >= 256 channels; with the
// upsampling gather only from 512 channels on) and >= 160 tiles; VAMD_NT_WIDE256_MIN=1 forces it
// wherever the shape allows (tests).
static bool nt_wide256(int N, int64_t M, int batch, size_t elem, int Ktot, bool up) {
    static const int on = [] { const char* e = getenv("VAMD_NT_WIDE256"); return e ? atoi(e) : 1; }();
    static const int min_tiles = [] { const char* e = getenv("VAMD_NT_WIDE256_MIN"); return e ? atoi(e) : 160; }();
    if (!on || elem != 2 || N % 256 != 0 || nt_classic()) return false;
    if (min_tiles > 1 && Ktot < (up ? 4608 : 2304)) return false;
    return ((M + 255) / 256) * batch * (N / 256) >= min_tiles;
}

template <typename T, int UP>
static int launch_ntw256(const Gather& gp, const T* x, const T* w, int N, int ldw, int ldc, const Epilogue& ep_in,
                         void* y, int batch, int64_t sA, int64_t sB, int64_t sC, hipStream_t st) {
    Epilogue ep = ep_in;
    ep.nt_store = nt_store_for((int64_t)gp.M * batch * ldc * (int64_t)sizeof(T));
    constexpr int LDS_W = 2 * (256 * kRowBytes + 256 * kRowBytes) + 4 * 256 * 2 * (int)sizeof(float) + 256 * (int)sizeof(float) + 8 * 2048;   // stages + statistics slots + bias + the epilogue's transposing regions
    static bool configured = false;
    if (!configured) {
        int rc = set_lds_limit(conv_ntw_mfma_kernel<T, 256, UP, false>, LDS_W, "conv_ntw_mfma");
        if (!rc) rc = set_lds_limit(conv_ntw_mfma_kernel<T, 256, UP, true>, LDS_W, "conv_ntw_mfma");
        if constexpr (UP == 0) {
            if (!rc) rc = set_lds_limit(conv_ntw_mfma_kernel<T, 256, UP, 2>, LDS_W, "conv_ntw_mfma");
        }
        if (rc) return rc;
        configured = true;
    }
    const int n_tiles = N / 256, mw_tiles = (gp.M + 255) / 256;
    const int row_tiles = mw_tiles * batch;
    const int groups = ntp_row_groups(row_tiles, n_tiles, 1);
    TimedLaunch timed(TIMED_NT, take_flops(2.0 * gp.M * N * gp.Ktot * batch), 2.0 * gp.M * N * gp.Ktot * batch, st);
    ClassSet cs;
    cs.ncls = 0;
    if (ep.act_scale) {
        if constexpr (UP == 0)
            hipLaunchKernelGGL((conv_ntw_mfma_kernel<T, 256, UP, 2>), dim3(groups * n_tiles), dim3(512), LDS_W, st, gp, x, w, N,
                               ldw, ldc, ep, y, mw_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
        else
            return fail(VAMD_EUNSUPPORTED, "conv_ntw_mfma: fused normalise with this gather");
    } else if (ep.stats)
        hipLaunchKernelGGL((conv_ntw_mfma_kernel<T, 256, UP, true>), dim3(groups * n_tiles), dim3(512), LDS_W, st, gp, x, w, N,
                           ldw, ldc, ep, y, mw_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
    else
        hipLaunchKernelGGL((conv_ntw_mfma_kernel<T, 256, UP, false>), dim3(groups * n_tiles), dim3(512), LDS_W, st, gp, x, w, N,
                           ldw, ldc, ep, y, mw_tiles, n_tiles, row_tiles, sA, sB, sC, cs);
    VAMD_LAUNCH_OK("conv_ntw_mfma");
    return 0;
}

// Resident-filter halo kernel: plain 3x3 / stride 1 / pad 1 over 64 -> 64 channels in bf16 with enough
// 256-pixel patches to fill the chip (VAMD_C64_HALO=0 switches it off, VAMD_C64_HALO_MIN=1 forces it
// for every such shape).  The patch is 8 x 32 or 16 x 16 pixels, whichever covers the image with fewer.
static bool c64_halo_shape(int H, int W, int64_t M, int* pw_log2, int* tiles_w, int* tiles_h, int* ntiles) {
    static const int on = [] { const char* e = getenv("VAMD_C64_HALO"); return e ? atoi(e) : 1; }();
    static const int min_tiles = [] { const char* e = getenv("VAMD_C64_HALO_MIN"); return e ? atoi(e) : 512; }();
    if (!on || nt_classic() || H <= 0 || W <= 0 || M % ((int64_t)H * W) != 0 || M * 128 >= 0x7FFFFFF0ll) return false;
    const int B = (int)(M / ((int64_t)H * W));
    const int n32 = ((H + 7) / 8) * ((W + 31) / 32), n16 = ((H + 15) / 16) * ((W + 15) / 16);
    const bool wide = n32 <= n16;
    *pw_log2 = wide ? 5 : 4;
    *tiles_w = wide ? (W + 31) / 32 : (W + 15) / 16;
    *tiles_h = wide ? (H + 7) / 8 : (H + 15) / 16;
    *ntiles = B * (wide ? n32 : n16);
    return *ntiles >= min_tiles;
}
static bool c64_halo_gather(const Gather& gp, int N, int ldw, int ldc, const Epilogue& ep, int batch) {
    return gp.Cin == 64 && gp.ldx == 64 && (N == 64 || N == 128) && ldw == 576 && ldc == N && batch == 1 && gp.kr == 3 && gp.ks == 3 &&
           gp.k == 3 && gp.stride == 1 && gp.pad == 1 && gp.padw == 1 && !gp.shift && !gp.ostep && gp.wr0 == 0 && gp.ws0 == 0 &&
           gp.wstep == 1 && gp.wk == 3 && gp.Cin2 == 0 && gp.Hin == gp.He && gp.Win == gp.We && gp.Ho == gp.Hin && gp.Wo == gp.Win &&
           !ep.out_f32 && ep.alpha == 1.0f && gp.Ktot == 576;
}

static int c64_halo_blocks(int ntiles, int combos) {       // combos: column tiles x classes, each with its own block range
    const int per_combo = 256 / combos;
    return (ntiles < per_combo ? ntiles : per_combo) * combos;
}

template <int KR, int KS>
static int launch_c64_halo(const HaloArgs& ha, const bf16_t* x, const bf16_t* w, const Epilogue& ep, void* y, double flops,
                           hipStream_t st) {
    constexpr int LDS = halo_lds_bytes(KR * KS);
    static bool configured = false;
    if (!configured) {
        int rc = set_lds_limit(conv_c64_halo_kernel<false, KR, KS>, LDS, "conv_c64_halo");
        if (!rc) rc = set_lds_limit(conv_c64_halo_kernel<true, KR, KS>, LDS, "conv_c64_halo");
        if (!rc) rc = set_lds_limit(conv_c64_halo_kernel<2, KR, KS>, LDS, "conv_c64_halo");
        if constexpr (KR == 3) {
            if (!rc) rc = set_lds_limit(conv_c64_halo_kernel<0, KR, KS, true>, LDS, "conv_c64_halo");
            if (!rc) rc = set_lds_limit(conv_c64_halo_kernel<1, KR, KS, true>, LDS, "conv_c64_halo");
        }
        if (rc) return rc;
        configured = true;
    }
    const int blocks = c64_halo_blocks(ha.ntiles, ha.n_tiles * ha.ncls);
    TimedLaunch timed(TIMED_NT, take_flops(flops), flops, st);
    if (ep.pre_scale) {
        if constexpr (KR == 3) {
            if (ep.act_scale) return fail(VAMD_EUNSUPPORTED, "conv_c64_halo: input normalisation and fused output normalisation together");
            if (ep.stats) hipLaunchKernelGGL((conv_c64_halo_kernel<1, KR, KS, true>), dim3(blocks), dim3(512), LDS, st, ha, x, w, ep, y);
            else hipLaunchKernelGGL((conv_c64_halo_kernel<0, KR, KS, true>), dim3(blocks), dim3(512), LDS, st, ha, x, w, ep, y);
        } else {
            return fail(VAMD_EUNSUPPORTED, "conv_c64_halo: input normalisation on a class filter");
        }
    } else if (ep.act_scale) hipLaunchKernelGGL((conv_c64_halo_kernel<2, KR, KS>), dim3(blocks), dim3(512), LDS, st, ha, x, w, ep, y);
    else if (ep.stats) hipLaunchKernelGGL((conv_c64_halo_kernel<true, KR, KS>), dim3(blocks), dim3(512), LDS, st, ha, x, w, ep, y);
    else hipLaunchKernelGGL((conv_c64_halo_kernel<false, KR, KS>), dim3(blocks), dim3(512), LDS, st, ha, x, w, ep, y);
    VAMD_LAUNCH_OK("conv_c64_halo");
    return 0;
}

// ------------------------------------------------------------------------------------------
// Split reduction for layers with fewer tiles than CUs and a long reduction (512 channels at 14 x 14 with 16 images:
// 100 tiles of 72 K-steps each, one per workgroup, each step a full load-to-MFMA latency).  The filter rows become
// up to four "classes" of one launch (ClassSet: same output rows, a one-row sub-filter view each), every class
// stores its fp32 partial product into the caller's scratch buffer, and splitk_reduce_kernel adds the partials in
// class order, applies the epilogue terms in the kernels' order (bias, time embedding, residual, skip), rounds
// once and sums the following BatchNorm's statistics per block of rows.
static std::atomic<int> g_splitk_mode{1};

static int nt_split_rows(const Gather& gp, int N, int ldc, const Epilogue& ep, int batch, size_t elem) {
    // tiles of the kernel that would run unsplit.  Measured: 16 x 14 x 14 x 512 channels (52 tiles of 72 K-steps)
    // 69.9 -> 43.7 us, 4 x 16 x 16 x 256 (8 tiles of 36 K-steps) 28.6 -> 19.3 us; from 128 tiles on the split loses
    // (4 x 64 x 64, 128 -> 256 channels: 29.0 -> 34.9 us).  In the step: flowers 8.22 -> 8.03 ms, its 16-image sampler
    // 7630 -> 8000 image-steps/s, ffhq 16.49 -> 16.28 ms; thresholds of 104 / 136 / 200 tiles measure alike.
    static const int max_tiles = [] { const char* e = getenv("VAMD_SPLITK_MAX_TILES"); return e ? atoi(e) : 104; }();
    if (!g_splitk_mode.load() || nt_classic() || batch != 1 || gp.shift || gp.ostep || gp.stride != 1 || gp.Cin2 > 0) return 0;
    if (ep.out_f32 || ep.alpha != 1.0f || ldc != N || !ep.scratch) return 0;
    if (gp.kr != gp.k || gp.ks != gp.k || gp.k < 3 || gp.k > 4 || gp.wstep != 1) return 0;
    const int piece = 16 / (int)elem;
    if (N % piece != 0 || N / piece > 256 || gp.ldx != gp.Cin) return 0;
    if ((int64_t)gp.ks * gp.Cin < 8 * (elem == 2 ? 64 : 32)) return 0;               // at least 8 K-steps per class
    if (ep.scratch_bytes < 4096 + (size_t)gp.k * (size_t)gp.M * N * 4) return 0;
    const int BN = N <= 64 ? 64 : 128, n_tiles = (N + BN - 1) / BN;
    const int rows = nt_wide(BN, gp.M, 1, n_tiles) ? 256 : kBM;
    const int64_t tiles = ((int64_t)gp.M + rows - 1) / rows * n_tiles;
    return tiles <= max_tiles ? gp.k : 0;
}

static int splitk_rows_per_block(int64_t M) { return M <= 8192 ? 16 : (M <= 16384 ? 32 : 64); }
static int splitk_stats_parts(int64_t M) { const int r = splitk_rows_per_block(M); return (int)((M + r - 1) / r); }

template <typename T, bool STATS>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ part, int nsplit, int64_t MN, int M, int N, FastDiv div_howo, Epilogue ep,
                     T* __restrict__ y, int rows_per_block) {
    constexpr int PIECE = 16 / (int)sizeof(T);
    __shared__ float red[256 * PIECE * 2];               // [row lane][N][2] (row lanes * N = 256 * PIECE at most)
    const int tc = N / PIECE, tr = 256 / tc;
    const int rl = threadIdx.x / tc, cl = threadIdx.x - rl * tc, n = cl * PIECE;
    float s1[PIECE], s2[PIECE], bv[PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) { s1[e] = 0.f; s2[e] = 0.f; bv[e] = 0.f; }
    if (rl < tr) {
        if (ep.bias) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) bv[e] = ep.bias[n + e];
        }
        const int r0 = blockIdx.x * rows_per_block;
        const int r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
        const T* res = reinterpret_cast<const T*>(ep.res);
        const T* res2 = reinterpret_cast<const T*>(ep.res2);
        for (int row = r0 + rl; row < r1; row += tr) {
            const int64_t off = (int64_t)row * N + n;
            float v[PIECE];
#pragma unroll
            for (int e = 0; e < PIECE; ++e) v[e] = 0.f;
            for (int s = 0; s < nsplit; ++s) {
#pragma unroll
                for (int e = 0; e < PIECE; e += 4) {
                    const float4 p4 = *reinterpret_cast<const float4*>(part + (int64_t)s * MN + off + e);
                    v[e] += p4.x; v[e + 1] += p4.y; v[e + 2] += p4.z; v[e + 3] += p4.w;
                }
            }
#pragma unroll
            for (int e = 0; e < PIECE; ++e) v[e] += bv[e];
            if (ep.temb) {
                const float* tp = ep.temb + (int64_t)fdiv((uint32_t)row, div_howo) * N + n;
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[e] += tp[e];
            }
            if (res) {
                const Vec16<T> rv = *reinterpret_cast<const Vec16<T>*>(res + off);
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[e] += to_f(rv.v[e]);
            }
            if (res2) {
                const Vec16<T> rv = *reinterpret_cast<const Vec16<T>*>(res2 + off);
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[e] += to_f(rv.v[e]);
            }
            Vec16<T> ov;
#pragma unroll
            for (int e = 0; e < PIECE; ++e) ov.v[e] = from_f<T>(v[e]);
            if (ep.act_scale) {                          // the consumer's eval-mode normalise + LeakyReLU of the rounded value
                if (ep.raw) *reinterpret_cast<Vec16<T>*>(reinterpret_cast<T*>(ep.raw) + off) = ov;
#pragma unroll
                for (int e = 0; e < PIECE; ++e)
                    ov.v[e] = from_f<T>(leaky(fmaf(to_f(ov.v[e]), ep.act_scale[n + e], ep.act_shift[n + e]), ep.act_slope));
            }
            *reinterpret_cast<Vec16<T>*>(y + off) = ov;
            if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < PIECE; ++e) {
                    const float q = to_f(ov.v[e]);
                    s1[e] += q;
                    s2[e] = fmaf(q, q, s2[e]);
                }
            }
        }
    }
    if constexpr (STATS) {
        if (rl < tr) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                red[(rl * N + n + e) * 2 + 0] = s1[e];
                red[(rl * N + n + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < N * 2; idx += 256) {
            float a = 0.f;
            for (int r = 0; r < tr; ++r) a += red[r * N * 2 + idx];
            ep.stats[(int64_t)blockIdx.x * N * 2 + idx] = a;
        }
    }
}

template <typename T>
static int run_nt_split(const Gather& gp, const T* x, const T* w, int N, int ldw, const Epilogue& ep, void* y, int nsplit,
                        hipStream_t st) {
    ClassSet set;
    set.ncls = 0;
    const int64_t MN = (int64_t)gp.M * N;
    for (int r = 0; r < nsplit; ++r) {
        Gather g = gp;                                    // filter row r alone: taps (r, 0 .. ks-1)
        g.kr = 1;
        g.wr0 = gp.wr0 + r;
        g.pad = gp.pad - r;                               // tap row 0 of the class reads input row ho + r - pad
        g.Ktot = g.ks * g.Cin;
        set.g[set.ncls] = g;
        set.w_off[set.ncls] = 0;
        set.stats_off[set.ncls] = 0;
        set.y_off[set.ncls] = (int64_t)r * MN;
        ++set.ncls;
    }
    float* part = reinterpret_cast<float*>(ep.scratch + 4096);
    Epilogue pe{nullptr, nullptr, nullptr, nullptr, 1.0f, 1, nullptr, nullptr, 0};
    int rc = N <= 64 ? launch_nt<T, 64, false>(set.g[0], x, w, N, ldw, N, pe, part, 1, 0, 0, 0, st, &set)
                     : launch_nt<T, 128, false>(set.g[0], x, w, N, ldw, N, pe, part, 1, 0, 0, 0, st, &set);
    if (rc) return rc;
    const int rpb = splitk_rows_per_block(gp.M), blocks = splitk_stats_parts(gp.M);
    if (ep.stats)
        hipLaunchKernelGGL((splitk_reduce_kernel<T, true>), dim3(blocks), dim3(256), 0, st, part, nsplit, MN, gp.M, N,
                           gp.div_howo, ep, (T*)y, rpb);
    else
        hipLaunchKernelGGL((splitk_reduce_kernel<T, false>), dim3(blocks), dim3(256), 0, st, part, nsplit, MN, gp.M, N,
                           gp.div_howo, ep, (T*)y, rpb);
    VAMD_LAUNCH_OK("splitk_reduce");
    return 0;
}

template <typename T>
static int run_nt(const Gather& gp, const void* x, const void* w, int N, int ldw, int ldc, const Epilogue& ep,
                  void* y, int batch, int64_t sA, int64_t sB, int64_t sC, hipStream_t st,
                  const ClassSet* classes = nullptr) {
    if (!classes) {
        const int nsplit = nt_split_rows(gp, N, ldc, ep, batch, sizeof(T));
        if (nsplit > 0) {
            if (ep.pre_scale) return fail(VAMD_EUNSUPPORTED, "conv: no kernel normalises the input of this shape (split reduction)");
            return run_nt_split<T>(gp, (const T*)x, (const T*)w, N, ldw, ep, y, nsplit, st);
        }
    }
    if constexpr (sizeof(T) == 2) {
        int pwl, tw, th, nt;
        if (!classes && c64_halo_gather(gp, N, ldw, ldc, ep, batch) && c64_halo_shape(gp.Hin, gp.Win, gp.M, &pwl, &tw, &th, &nt))
        {
            HaloArgs ha{(int)(gp.M / ((int64_t)gp.Hin * gp.Win)), gp.Hin, gp.Win, pwl, tw, th, nt, N, N / 64, 1, 0, 0};
            return launch_c64_halo<3, 3>(ha, (const bf16_t*)x, (const bf16_t*)w, ep, y, 2.0 * gp.M * N * 576, st);
        }
    }
    if (ep.pre_scale) return fail(VAMD_EUNSUPPORTED, "conv: only the resident-filter kernel normalises its input (ask vamd_conv2d_prenorm_supported)");
    if (classes) {                                        // class sets: stride-1 gathers without upsampling
        if (N <= 64) return launch_nt<T, 64, false>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st, classes);
        return launch_nt<T, 128, false>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st, classes);
    }
    if (gp.Cin2 > 0) {                                    // second source: stride-1 gather without upsampling
        // tile choice by the main reduction alone, so that the statistics-parts query of the plain conv holds
        if constexpr (sizeof(T) == 2) {
            if (nt_wide256(N, gp.M, batch, sizeof(T), gp.Ktot - gp.Cin2, false))
                return launch_ntw256<T, 2>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
        }
        if (N <= 64) return launch_nt<T, 64, 2>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
        return launch_nt<T, 128, 2>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
    }
    if constexpr (sizeof(T) == 2) {
        // the 256-channel tile's epilogue writes rows in place and in T only
        if (!gp.ostep && !ep.out_f32 && nt_wide256(N, gp.M, batch, sizeof(T), gp.Ktot, gp.shift != 0)) {
            if (gp.shift) return launch_ntw256<T, true>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
            return launch_ntw256<T, false>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
        }
    }
    if (gp.shift) {
        if (N <= 64) return launch_nt<T, 64, true>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
        return launch_nt<T, 128, true>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
    }
    if (N <= 64) return launch_nt<T, 64, false>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
    return launch_nt<T, 128, false>(gp, (const T*)x, (const T*)w, N, ldw, ldc, ep, y, batch, sA, sB, sC, st);
}

bool conv_fwd_mfma_supported(const ConvGeom& g, int dtype) {
    return nt_shape_ok(g.Cin, g.Cout, g.M, dtype);
}

int conv_fwd_mfma_stats_parts(const ConvGeom& g, int dtype) {
    const int m_tiles = (int)((g.M + kBM - 1) / kBM);
    if (nt_classic()) return m_tiles;
    {                                                     // as run_nt splits the reduction
        const Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
        const Epilogue ep{nullptr, nullptr, nullptr, nullptr, 1.0f, 0, nullptr, (unsigned char*)g.scratch, g.scratch_bytes};
        if (nt_split_rows(gp, g.Cout, g.Cout, ep, 1, dtype == VAMD_BF16 ? 2 : 4) > 0) return splitk_stats_parts(g.M);
    }
    if (dtype == VAMD_BF16 && g.Cin == 64 && (g.Cout == 64 || g.Cout == 128) && g.k == 3 && g.stride == 1 && g.pad == 1 && !g.up) {
        int pwl, tw, th, nt;                              // as run_nt picks the resident-filter kernel
        if (c64_halo_shape(g.H, g.W, g.M, &pwl, &tw, &th, &nt)) return c64_halo_blocks(nt, g.Cout / 64) / (g.Cout / 64);
    }
    if (dtype == VAMD_BF16 && nt_wide256(g.Cout, g.M, 1, 2, g.k * g.k * g.Cin, g.up != 0)) return ntp_row_groups((int)((g.M + 255) / 256), g.Cout / 256, 1);
    const int BN = g.Cout <= 64 ? 64 : 128;               // as run_nt picks
    const int n_tiles = (g.Cout + BN - 1) / BN;
    if (nt_wide(BN, g.M, 1, n_tiles)) return ntp_row_groups((int)((g.M + 255) / 256), n_tiles, 1);
    return ntp_row_groups(m_tiles, n_tiles, ntp_per_cu(BN));
}

int conv_fwd_mfma(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                  const float* temb, const void* res, const void* res2, void* y, float* stats, hipStream_t st) {
    Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
    Epilogue ep{bias, temb, res, res2, 1.0f, 0, stats, (unsigned char*)g.scratch, g.scratch_bytes};
    if (dtype == VAMD_BF16) return run_nt<bf16_t>(gp, x, w, g.Cout, gp.Ktot, g.Cout, ep, y, 1, 0, 0, 0, st);
    return run_nt<float>(gp, x, w, g.Cout, gp.Ktot, g.Cout, ep, y, 1, 0, 0, 0, st);
}

// The forward (and weight gradient) of a convolution whose INPUT is BatchNorm + LeakyReLU of a stored tensor, with the
// normalisation applied inside the kernels (no normalised tensor in memory): shapes that the resident-filter forward
// kernel AND the resident-gradient weight-gradient kernel both take.
bool conv_prenorm_mfma_supported(const ConvGeom& g, int dtype) {
    if (dtype != VAMD_BF16 || nt_classic() || g.up) return false;
    const Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
    const Epilogue ep{nullptr, nullptr, nullptr, nullptr, 1.0f, 0, nullptr, (unsigned char*)g.scratch, g.scratch_bytes};
    int pwl, tw, th, nt;
    if (!c64_halo_gather(gp, g.Cout, gp.Ktot, g.Cout, ep, 1) || !c64_halo_shape(g.H, g.W, g.M, &pwl, &tw, &th, &nt)) return false;
    if (nt_split_rows(gp, g.Cout, g.Cout, ep, 1, 2) > 0) return false;
    return wgrad_halo_ok(g, dtype);
}

int conv_fwd_prenorm_mfma(const ConvGeom& g, int dtype, const void* x, const float* pre_scale, const float* pre_shift,
                          float pre_slope, const void* w, const float* bias, const float* temb, const void* res,
                          const void* res2, void* y, float* stats, hipStream_t st) {
    Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
    Epilogue ep{bias, temb, res, res2, 1.0f, 0, stats, (unsigned char*)g.scratch, g.scratch_bytes, nullptr, nullptr, 0.f, nullptr,
                pre_scale, pre_shift, pre_slope};
    return run_nt<bf16_t>(gp, x, w, g.Cout, gp.Ktot, g.Cout, ep, y, 1, 0, 0, 0, st);
}

// The forward with the consumer's eval-mode BatchNorm + LeakyReLU folded into the epilogue.  Kernels that implement
// it: the persistent 4-wave kernels on plain gathers, the resident-filter kernel and the split reduction.
bool conv_fwd_act_mfma_supported(const ConvGeom& g, int dtype) {
    return !nt_classic() && !g.up && nt_shape_ok(g.Cin, g.Cout, g.M, dtype) && g.Cout % (dtype == VAMD_BF16 ? 8 : 4) == 0;
}

int conv_fwd_act_mfma(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias, const float* temb,
                      const void* res, const void* res2, const float* scale, const float* shift, float slope, void* y,
                      void* raw, hipStream_t st) {
    Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
    Epilogue ep{bias, temb, res, res2, 1.0f, 0, nullptr, (unsigned char*)g.scratch, g.scratch_bytes, scale, shift, slope, raw};
    if (dtype == VAMD_BF16) return run_nt<bf16_t>(gp, x, w, g.Cout, gp.Ktot, g.Cout, ep, y, 1, 0, 0, 0, st);
    return run_nt<float>(gp, x, w, g.Cout, gp.Ktot, g.Cout, ep, y, 1, 0, 0, 0, st);
}

// conv(x; w) + conv1x1(x2; w2) in one launch: the 1x1 shortcut of a ResNetBlock rides as Cin2 extra
// reduction steps of its conv2 (no shortcut tensor is written or read back).  Stride 1, no
// upsampling, channel counts that are multiples of a K-step so that a K-step never straddles sources.
bool conv_fwd_short_mfma_supported(const ConvGeom& g, int dtype, int Cin2) {
    const int kstep = dtype == VAMD_BF16 ? 64 : 32;
    if (nt_classic() || g.up || g.stride != 1 || Cin2 <= 0 || Cin2 % kstep != 0 || g.Cin % kstep != 0) return false;
    if ((int64_t)g.M * Cin2 >= (1ll << 30)) return false;      // 32-bit byte offsets into x2
    if (dtype == VAMD_BF16 && g.Cin == 64 && (g.Cout == 64 || g.Cout == 128) && g.k == 3 && g.pad == 1) {
        int pwl, tw, th, nt;                              // the resident-filter kernel serves the plain conv: keep the two launches
        if (c64_halo_shape(g.H, g.W, g.M, &pwl, &tw, &th, &nt)) return false;
    }
    {                                                     // likewise where the plain conv's reduction is split
        const Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
        const Epilogue ep{nullptr, nullptr, nullptr, nullptr, 1.0f, 0, nullptr, (unsigned char*)g.scratch, g.scratch_bytes};
        if (nt_split_rows(gp, g.Cout, g.Cout, ep, 1, dtype == VAMD_BF16 ? 2 : 4) > 0) return false;
    }
    return nt_shape_ok(g.Cin, g.Cout, g.M, dtype);
}

int conv_fwd_short_mfma(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                        const float* temb, const void* x2, const void* w2, int Cin2, const void* res2, void* y,
                        float* stats, hipStream_t st) {
    Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
    const int ldw = gp.Ktot;
    gp.x2 = x2; gp.w2 = w2; gp.Cin2 = Cin2;
    gp.Ktot += Cin2;
    Epilogue ep{bias, temb, nullptr, res2, 1.0f, 0, stats};
    if (dtype == VAMD_BF16) return run_nt<bf16_t>(gp, x, w, g.Cout, ldw, g.Cout, ep, y, 1, 0, 0, 0, st);
    return run_nt<float>(gp, x, w, g.Cout, ldw, g.Cout, ep, y, 1, 0, 0, 0, st);
}

// Forward of (nearest 2x upsample -> 3x3 conv, pad 1) without touching an upsampled pixel twice: output
// pixel (2i+ph, 2j+pw) reads the 2x2 stored neighbourhood rows i-1+ph .. i+ph, columns j-1+pw .. j+pw
// through the class filter Weff[ph][pw] (conv taps that fall on the same stored pixel summed:
// ph = 0: {0}, {1,2};  ph = 1: {0,1}, {2}).  Four 2x2 convolutions with an output scatter of step 2
// in one launch: 16 taps per stored pixel instead of 36.  `w4` is [4 classes][Cout][2][2][Cin]
// (vamd_repack_weight_up4_fwd).
static void fwd_up4_classes(const ConvGeom& g, ClassSet& set) {
    set.ncls = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            Gather gp = make_gather(g.H, g.W, g.Cin, g.H, g.W, g.H, g.W, 2, 1, 1 - ph, 0, (int64_t)g.B * g.H * g.W);
            gp.padw = 1 - pw;
            gp.oH = g.Ho; gp.oW = g.Wo; gp.ostep = 2; gp.oph = ph; gp.opw = pw;
            set.g[set.ncls] = gp;
            set.w_off[set.ncls] = (int64_t)set.ncls * g.Cout * 4 * g.Cin;
            set.stats_off[set.ncls] = 0;
            set.y_off[set.ncls] = 0;
            ++set.ncls;
        }
}

bool conv_fwd_up4_mfma_supported(const ConvGeom& g, int dtype) {
    static const int on = [] { const char* e = getenv("VAMD_FWD_UP4"); return e ? atoi(e) : 1; }();
    if (!on || !g.up || g.k != 3 || g.stride != 1 || g.pad != 1 || nt_classic()) return false;
    return nt_shape_ok(g.Cin, g.Cout, (int64_t)g.B * g.H * g.W, dtype) && 4 * g.Cin >= (dtype == VAMD_BF16 ? 64 : 32);
}

// the four classes on the resident-filter kernel (64 stored channels, bf16): one class filter (32 KB) per workgroup
static bool fwd_up4_halo(const ConvGeom& g, int dtype, int* pwl, int* tw, int* th, int* nt) {
    static const int on = [] { const char* e = getenv("VAMD_UP4_HALO"); return e ? atoi(e) : 1; }();
    if (!on || dtype != VAMD_BF16 || g.Cin != 64 || (g.Cout != 64 && g.Cout != 128)) return false;
    if ((int64_t)g.M * g.Cout * 2 >= 0x7FFFFFF0ll) return false;
    return c64_halo_shape(g.H, g.W, (int64_t)g.B * g.H * g.W, pwl, tw, th, nt);
}

int conv_fwd_up4_mfma_stats_parts(const ConvGeom& g, int dtype) {
    {
        int pwl, tw, th, nt;
        if (fwd_up4_halo(g, dtype, &pwl, &tw, &th, &nt)) return c64_halo_blocks(nt, (g.Cout / 64) * 4) / (g.Cout / 64);
    }
    ClassSet set;
    fwd_up4_classes(g, set);
    const int BN = g.Cout <= 64 ? 64 : 128;
    const int n_tiles = (g.Cout + BN - 1) / BN;
    const bool wide = nt_wide(BN, set.g[0].M, 1, n_tiles);
    const int per_column = wide ? 256 / n_tiles : 256 * ntp_per_cu(BN) / n_tiles;
    return plan_class_blocks(set, g.Cout, n_tiles, wide ? 256 : kBM, per_column > 0 ? per_column : 1, nullptr) / n_tiles;
}

int conv_fwd_up4_mfma(const ConvGeom& g, int dtype, const void* x, const void* w4, const float* bias,
                      const float* temb, const void* res, const void* res2, void* y, float* stats, hipStream_t st) {
    Epilogue ep{bias, temb, res, res2, 1.0f, 0, stats};
    {
        int pwl, tw, th, nt;
        if (fwd_up4_halo(g, dtype, &pwl, &tw, &th, &nt)) {
            HaloArgs ha{g.B, g.H, g.W, pwl, tw, th, nt, g.Cout, g.Cout / 64, 4, 2, (int64_t)g.Cout * 4 * g.Cin};
            return launch_c64_halo<2, 2>(ha, (const bf16_t*)x, (const bf16_t*)w4, ep, y, 2.0 * g.M * g.Cout * 9.0 * g.Cin, st);
        }
    }
    ClassSet set;
    fwd_up4_classes(g, set);
    t_nominal_flops = 2.0 * g.M * g.Cout * 9.0 * g.Cin;
    const int ldw = 4 * g.Cin;
    if (dtype == VAMD_BF16) return run_nt<bf16_t>(set.g[0], x, w4, g.Cout, ldw, g.Cout, ep, y, 1, 0, 0, 0, st, &set);
    return run_nt<float>(set.g[0], x, w4, g.Cout, ldw, g.Cout, ep, y, 1, 0, 0, 0, st, &set);
}

bool conv_dgrad_mfma_supported(const ConvGeom& g, int dtype) {
    if (g.stride == 2 && g.k != 3) return false;       // a strided 1x1 leaves whole parity classes without taps
    const int64_t rows = (int64_t)g.B * g.He * g.We;
    return nt_shape_ok(g.Cout, g.Cin, rows, dtype);
}

size_t conv_dgrad_mfma_workspace(const ConvGeom& g, int dtype) {
    if (!g.up) return 0;
    return (size_t)g.B * g.He * g.We * g.Cin * (dtype == VAMD_BF16 ? 2 : 4);
}

template <typename T>
__global__ void sumpool2x2_add_kernel(const T* __restrict__ x, const T* __restrict__ add, T* __restrict__ y,
                                      int B, int H, int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    const int CV = C / VEC;
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(idx % CV);
        int64_t p = idx / CV;
        const int wq = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int64_t b = p / H;
        const int64_t row = (int64_t)2 * W * C;
        const T* s = x + ((b * 2 * H + 2 * h) * 2 * W + 2 * wq) * (int64_t)C + cv * VEC;
        const Vec16<T> v0 = *reinterpret_cast<const Vec16<T>*>(s);
        const Vec16<T> v1 = *reinterpret_cast<const Vec16<T>*>(s + C);
        const Vec16<T> v2 = *reinterpret_cast<const Vec16<T>*>(s + row);
        const Vec16<T> v3 = *reinterpret_cast<const Vec16<T>*>(s + row + C);
        Vec16<T> o, a;
        if (add) a = *reinterpret_cast<const Vec16<T>*>(add + idx * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float r = to_f(v0.v[e]) + to_f(v1.v[e]) + to_f(v2.v[e]) + to_f(v3.v[e]);
            if (add) r += to_f(a.v[e]);
            o.v[e] = from_f<T>(r);
        }
        *reinterpret_cast<Vec16<T>*>(y + idx * VEC) = o;
    }
}

int conv_dgrad_mfma(const ConvGeom& g, int dtype, const void* dy, const void* wt, const void* add,
                    void* dx, void* workspace, size_t workspace_bytes, hipStream_t st) {
    // The input gradient is itself a convolution over dY with the flipped, transposed filter:
    //   stride 1: plain 3x3 pad 1 (or 1x1) over dY at the output resolution;
    //   upsample: stride-1 result at the upsampled resolution, then a 2x2 sum;
    //   stride 2: input pixel (h, w) only meets the filter taps r = h + pad (mod 2), s likewise, so
    //             each of the 4 parity classes of (h, w) is a small dense convolution of its own
    //             (1, 2, 2 and 4 of the 9 taps for a 3x3 pad-1 filter) over dY: no multiplications
    //             by inserted zeros.
    if (g.stride == 2) {
        const int padT = g.k - 1 - g.pad;
        static const int merged = [] { const char* e = getenv("VAMD_DGRAD_CLASSES"); return e ? atoi(e) : 1; }();
        ClassSet set;
        set.ncls = 0;
        const int ldw = g.k * g.k * g.Cout;
        Epilogue ep{nullptr, nullptr, add, nullptr, 1.0f, 0, nullptr};
        for (int ph = 0; ph < 2; ++ph)
            for (int pw = 0; pw < 2; ++pw) {
                const int Hc = (g.H - ph + 1) / 2, Wc = (g.W - pw + 1) / 2;     // pixels of this class
                if (Hc <= 0 || Wc <= 0) continue;
                // flipped-filter tap r reads dY row (h + r - padT) / 2, which must be an integer
                const int r0 = (ph + padT) & 1, s0 = (pw + padT) & 1;
                const int kr = (g.k - r0 + 1) / 2, kc = (g.k - s0 + 1) / 2;
                if (kr <= 0 || kc <= 0) {
                    return fail(VAMD_EUNSUPPORTED, "conv_dgrad_mfma: stride-2 class without taps (k=%d)", g.k);
                }
                // dY row for class row ho and tap index tr: ho + tr + (ph + r0 - padT) / 2 (exact division)
                const int offr = (ph + r0 - padT) / 2, offs = (pw + s0 - padT) / 2;
                Gather gp = make_gather(g.Ho, g.Wo, g.Cout, g.Ho, g.Wo, Hc, Wc, 1, 1, -offr, 0, (int64_t)g.B * Hc * Wc);
                if (offr != offs)
                    return fail(VAMD_EUNSUPPORTED, "conv_dgrad_mfma: unsupported stride-2 padding");
                gp.kr = kr;
                gp.ks = kc;
                gp.Ktot = kr * kc * g.Cout;
                gp.wr0 = r0; gp.ws0 = s0; gp.wstep = 2; gp.wk = g.k;
                gp.oH = g.H; gp.oW = g.W; gp.ostep = 2; gp.oph = ph; gp.opw = pw;
                if (merged && !nt_classic()) {            // one launch for all classes
                    set.g[set.ncls] = gp;
                    set.w_off[set.ncls] = 0;
                    set.stats_off[set.ncls] = 0;
                    set.y_off[set.ncls] = 0;
                    ++set.ncls;
                    continue;
                }
                int rc = dtype == VAMD_BF16 ? run_nt<bf16_t>(gp, dy, wt, g.Cin, ldw, g.Cin, ep, dx, 1, 0, 0, 0, st)
                                            : run_nt<float>(gp, dy, wt, g.Cin, ldw, g.Cin, ep, dx, 1, 0, 0, 0, st);
                if (rc) return rc;
            }
        if (set.ncls == 1) {
            return dtype == VAMD_BF16 ? run_nt<bf16_t>(set.g[0], dy, wt, g.Cin, ldw, g.Cin, ep, dx, 1, 0, 0, 0, st)
                                      : run_nt<float>(set.g[0], dy, wt, g.Cin, ldw, g.Cin, ep, dx, 1, 0, 0, 0, st);
        }
        if (set.ncls > 1) {
            return dtype == VAMD_BF16 ? run_nt<bf16_t>(set.g[0], dy, wt, g.Cin, ldw, g.Cin, ep, dx, 1, 0, 0, 0, st, &set)
                                      : run_nt<float>(set.g[0], dy, wt, g.Cin, ldw, g.Cin, ep, dx, 1, 0, 0, 0, st, &set);
        }
        return 0;
    }
    const int rowsH = g.up ? g.He : g.H, rowsW = g.up ? g.We : g.W;   // extents the GEMM rows enumerate
    const int64_t rows = (int64_t)g.B * rowsH * rowsW;
    const int padT = g.k - 1 - g.pad;
    Gather gp = make_gather(g.Ho, g.Wo, g.Cout, g.Ho, g.Wo, rowsH, rowsW, g.k, 1, padT, 0, rows);
    void* out = dx;
    const void* res = add;
    if (g.up) {
        if (!workspace || workspace_bytes < conv_dgrad_mfma_workspace(g, dtype))
            return fail(VAMD_EINVAL, "conv_dgrad_mfma: upsample gradient needs %zu workspace bytes", conv_dgrad_mfma_workspace(g, dtype));
        out = workspace;
        res = nullptr;
    }
    Epilogue ep{nullptr, nullptr, res, nullptr, 1.0f, 0, nullptr, (unsigned char*)g.scratch, g.scratch_bytes};
    int rc = dtype == VAMD_BF16 ? run_nt<bf16_t>(gp, dy, wt, g.Cin, gp.Ktot, g.Cin, ep, out, 1, 0, 0, 0, st)
                                : run_nt<float>(gp, dy, wt, g.Cin, gp.Ktot, g.Cin, ep, out, 1, 0, 0, 0, st);
    if (rc) return rc;
    if (g.up) {
        const int64_t total = (int64_t)g.B * g.H * g.W * g.Cin;
        VAMD_DISPATCH(dtype, "conv_dgrad_mfma", {
            hipLaunchKernelGGL((sumpool2x2_add_kernel<T>), dim3(grid_for(total / Vec16<T>::N, 256)), dim3(256), 0, st,
                               (const T*)workspace, (const T*)add, (T*)dx, g.B, g.H, g.W, g.Cin);
        });
        VAMD_LAUNCH_OK("sumpool2x2_add");
    }
    return 0;
}

// Input gradient of (nearest 2x upsample -> 3x3 conv, pad 1) as ONE stride-2 convolution over dY:
// input pixel i receives dY rows 2i-1 .. 2i+2 through the filter taps {2}, {1,2}, {0,1}, {0}
// (the conv taps that read the upsampled copies of row i), so
//   dx[b, i, j, :] = sum_{t, u < 4} G[t][u] . dY[b, 2i-1+t, 2j-1+u, :],  G[t][u] = sum of w over S(t) x S(u).
// 16 taps per input pixel instead of 9 taps per upsampled pixel (36 per input pixel) and no pooling pass.
// `w4` is [Cin][4][4][Cout] in the compute dtype (vamd_repack_weight_up4 builds it).
bool conv_dgrad_up4_mfma_supported(const ConvGeom& g, int dtype) {
    if (!g.up || g.k != 3 || g.stride != 1 || g.pad != 1 || nt_classic()) return false;
    return nt_shape_ok(g.Cout, g.Cin, (int64_t)g.B * g.H * g.W, dtype);
}

int conv_dgrad_up4_mfma(const ConvGeom& g, int dtype, const void* dy, const void* w4, const void* add, void* dx,
                        hipStream_t st) {
    Gather gp = make_gather(g.Ho, g.Wo, g.Cout, g.Ho, g.Wo, g.H, g.W, 4, 2, 1, 0, (int64_t)g.B * g.H * g.W);
    t_nominal_flops = 2.0 * g.M * g.Cout * 9.0 * g.Cin;
    Epilogue ep{nullptr, nullptr, add, nullptr, 1.0f, 0, nullptr, (unsigned char*)g.scratch, g.scratch_bytes};
    if (dtype == VAMD_BF16) return run_nt<bf16_t>(gp, dy, w4, g.Cin, gp.Ktot, g.Cin, ep, dx, 1, 0, 0, 0, st);
    return run_nt<float>(gp, dy, w4, g.Cin, gp.Ktot, g.Cin, ep, dx, 1, 0, 0, 0, st);
}

static bool tn_shape_ok(int Cin, int N, int64_t M, int dtype) {
    const int piece = dtype == VAMD_BF16 ? 8 : 4;
    return Cin % piece == 0 && N % piece == 0 && M < (1ll << 26) && M > 0;
}

bool conv_wgrad_mfma_supported(const ConvGeom& g, int dtype) {
    return tn_shape_ok(g.Cin, g.Cout, g.M, dtype);
}

// Tile shape the weight-gradient launcher picks.  Narrow outputs (Cout <= 64) take a 64 x 192 tile, which wastes no
// MFMA rows and divides 9 * Cin; everything else 128 x 128 on four waves, two workgroups per CU.  Opt-in
// (VAMD_TN_WIDE=1 / 2): 128 x 256 or 256 x 128 on eight waves, one workgroup per CU -- 25 % fewer bytes from L2 into LDS
// per FLOP, yet measured 5-8 % SLOWER per layer (512 -> 512 at 28 x 28 x 64: 347 vs 320 us; 256 -> 256 at 56 x 56: 304 vs
// 288 us; step 19.80 vs 19.72 ms): two independent four-wave workgroups drift apart and hide each other's barriers
// and fragment-read latency, one eight-wave workgroup cannot.
struct TnShape { int AN, BF, NW; };
static int tn_wide_mode() {
    static const int v = [] { const char* e = getenv("VAMD_TN_WIDE"); return e ? atoi(e) : 0; }();
    return v;
}
static bool tn_narrow(int N, int Ktot) { return N <= 64 && Ktot >= 192; }
static TnShape tn_shape(int N, int Ktot, int dtype, int out_mode) {
    if (out_mode == 0 && tn_narrow(N, Ktot)) return {64, 192, 4};
    if (out_mode == 0 && dtype == VAMD_BF16 && tn_wide_mode()) {
        if (N >= 128 && Ktot % 256 == 0 && Ktot >= 512) return {128, 256, 8};
        if (N % 256 == 0 && Ktot % 128 == 0 && Ktot >= 512 && tn_wide_mode() >= 2) return {256, 128, 8};
    }
    return {128, 128, 4};
}
static int tn_shape_tiles(const TnShape& sh, int N, int Ktot) { return ((N + sh.AN - 1) / sh.AN) * ((Ktot + sh.BF - 1) / sh.BF); }
// Pixel slices of the weight gradient: two workgroups (of four waves) per CU in all, or one of eight -- each slice runs long
// and the partial buffer stays small; at least 8 K-steps per slice, every slice non-empty.
static int tn_slice_cap(const TnShape& sh) {
    static const int v = [] { const char* e = getenv("VAMD_TN_BLOCKS"); return e ? atoi(e) : 0; }();
    static const int v8 = [] { const char* e = getenv("VAMD_TN_WIDE_BLOCKS"); return e ? atoi(e) : 0; }();
    // 512 for every filter on the four-wave shapes: in the step it equals a cap of 1024 for many-tile filters at 64 images
    // (19.49 / 19.51 ms) and is ahead at 16 (8.06 -> 7.96 ms: fewer slices to write and add); 256 and 768+ lose 4-15 %
    if (sh.NW == 8) return v8 > 0 ? v8 : 256;
    return v > 0 ? v : 512;
}
static void tn_split(const TnShape& sh, int tiles, int total_steps, int* slices_out, int* steps_per_out) {
    int slices = tn_slice_cap(sh) / tiles;
    const int max_slices = (total_steps + 7) / 8;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    const int steps_per = (total_steps + slices - 1) / slices;
    *slices_out = (total_steps + steps_per - 1) / steps_per;
    *steps_per_out = steps_per;
}
static int tn_wgrad_slices(const ConvGeom& g, int dtype) {
    if (wgrad_halo_ok(g, dtype)) return wgrad_halo_slices(g);
    const int Ktot = g.k * g.k * g.Cin, KP = dtype == VAMD_BF16 ? 64 : 32;
    const TnShape sh = tn_shape(g.Cout, Ktot, dtype, 0);
    int slices, steps_per;
    tn_split(sh, tn_shape_tiles(sh, g.Cout, Ktot), (int)((g.M + KP - 1) / KP), &slices, &steps_per);
    return slices;
}

// split_rows > 0 (1x1 filters only): output rows [j * split_rows, (j + 1) * split_rows) go to dw[j] / dbias[j] instead of
// one tensor -- three projections computed as one convolution, each with its own parameter
struct WgradSplit {
    int split_rows;
    float* dw[2];          // destinations of the second and third row range (the first is the kernel's dw)
    float* dbias[2];
};

// Sums the slices' partial products and writes the gradient either as [Cout][k][k][Cin] or in
// torch's [Cout][Cin][k][k].  A block covers 256/SL groups of 4 consecutive filter columns with SL
// slice lanes each (SL = 32 when there are many slices and few outputs); the blocks after those
// sum the bias partials.  The order of the additions depends only on the shape.
template <int SL>
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ parts, const float* __restrict__ bias_parts, int slices, int N, int Ktot,
                    int Cin, int k, int oihw, float* __restrict__ dw, float* __restrict__ dbias, WgradSplit sp) {
    constexpr int OG = 256 / SL;
    __shared__ float4 red[SL][OG];
    const int tx = threadIdx.x % OG, ty = threadIdx.x / OG;
    const int main_blocks = (int)(((int64_t)N * Ktot / 4 + OG - 1) / OG);
    if ((int)blockIdx.x >= main_blocks) {                 // bias: 8 channels x 32 slice lanes per block
        float* redf = reinterpret_cast<float*>(red);
        const int n = ((int)blockIdx.x - main_blocks) * 8 + (threadIdx.x & 7), lane_s = threadIdx.x >> 3;
        float sum = 0.f;
        if (n < N)
            for (int sl = lane_s; sl < slices; sl += 32) sum += bias_parts[(int64_t)sl * N + n];
        redf[threadIdx.x] = sum;
        __syncthreads();
        if (lane_s == 0 && n < N) {
            float tot = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) tot += redf[i * 8 + (threadIdx.x & 7)];
            if (sp.split_rows > 0 && n >= sp.split_rows) {
                const int j = n / sp.split_rows;
                sp.dbias[j - 1][n - j * sp.split_rows] = tot;
            } else {
                dbias[n] = tot;
            }
        }
        return;
    }
    const int64_t total4 = (int64_t)N * Ktot / 4;
    const int64_t q = (int64_t)blockIdx.x * OG + tx;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (q < total4) {
        const float4* src = reinterpret_cast<const float4*>(parts) + q;
        int sl = ty;
        for (; sl + 3 * SL < slices; sl += 4 * SL) {            // four loads in flight
            const float4 v0 = src[(int64_t)sl * total4], v1 = src[(int64_t)(sl + SL) * total4];
            const float4 v2 = src[(int64_t)(sl + 2 * SL) * total4], v3 = src[(int64_t)(sl + 3 * SL) * total4];
            s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
            s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
            s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
            s3.x += v3.x; s3.y += v3.y; s3.z += v3.z; s3.w += v3.w;
        }
        for (; sl < slices; sl += SL) {
            const float4 v = src[(int64_t)sl * total4];
            s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
        }
    }
    red[ty][tx] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                              (s0.w + s1.w) + (s2.w + s3.w));
    __syncthreads();
    if (ty != 0 || q >= total4) return;
    float4 r4 = red[0][tx];
#pragma unroll
    for (int i = 1; i < SL; ++i) {
        const float4 v = red[i][tx];
        r4.x += v.x; r4.y += v.y; r4.z += v.z; r4.w += v.w;
    }
    if (!oihw || k == 1) {
        if (sp.split_rows > 0) {                          // Ktot % 4 == 0: a group of four stays inside one row
            const int64_t per = (int64_t)sp.split_rows * Ktot / 4;
            const int j = (int)(q / per);
            if (j > 0) {
                reinterpret_cast<float4*>(sp.dw[j - 1])[q - j * per] = r4;
                return;
            }
        }
        reinterpret_cast<float4*>(dw)[q] = r4;
        return;
    }
    const int64_t e = q * 4;                            // (n, tap, c .. c+3): Cin % 4 == 0 keeps one tap
    const int n = (int)(e / Ktot), f = (int)(e - (int64_t)n * Ktot);
    const int tap = f / Cin, c = f - tap * Cin;
    float* o = dw + ((int64_t)n * Cin + c) * (k * k) + tap;
    o[0] = r4.x; o[k * k] = r4.y; o[2 * k * k] = r4.z; o[3 * k * k] = r4.w;
}

static int tn_pipe_mode() {
    static const int v = [] { const char* e = getenv("VAMD_TN_PIPE"); return e ? atoi(e) : 1; }();
    return v;
}

template <typename T, int AN, int BF, int NW>
static int launch_tn_shape(const Gather& gp, const T* dy, const T* x, int N, int lda, int ldc, float* dw, int batch,
                           int64_t sA, int64_t sB, int64_t sC, int out_mode, void* yout, float alpha, float* dbias,
                           int bias_mode, hipStream_t st) {
    constexpr int KP = sizeof(T) == 2 ? 64 : 32;
    constexpr int LDS = 2 * KP * (AN + BF) * (int)sizeof(T);
    static bool configured = false;
    if (!configured) {
        int rc = set_lds_limit(conv_tn_mfma_kernel<T, AN, BF, NW, false>, LDS, "conv_tn_mfma");
        if (!rc && sizeof(T) == 2) rc = set_lds_limit(conv_tn_mfma_kernel<T, AN, BF, NW, true>, LDS, "conv_tn_mfma");
        if (rc) return rc;
        configured = true;
    }
    const TnShape sh{AN, BF, NW};
    const int n_tiles = (N + AN - 1) / AN, f_tiles = (gp.Ktot + BF - 1) / BF;
    const int total_steps = (gp.M + KP - 1) / KP;
    int slices = 1, steps_per = total_steps;
    if (out_mode == 0) tn_split(sh, n_tiles * f_tiles, total_steps, &slices, &steps_per);
    dim3 grid(n_tiles * f_tiles * slices, 1, batch);
    TimedLaunch timed(TIMED_TN, take_flops(2.0 * gp.M * N * gp.Ktot * batch), 2.0 * gp.M * N * gp.Ktot * batch, st);
    if (sizeof(T) == 2 && tn_pipe_mode())
        hipLaunchKernelGGL((conv_tn_mfma_kernel<T, AN, BF, NW, true>), grid, dim3(64 * NW), LDS, st, gp, dy, x, N, lda, ldc, dw,
                           n_tiles, f_tiles, steps_per, sA, sB, sC, out_mode, yout, alpha, dbias, bias_mode);
    else
        hipLaunchKernelGGL((conv_tn_mfma_kernel<T, AN, BF, NW, false>), grid, dim3(64 * NW), LDS, st, gp, dy, x, N, lda, ldc, dw,
                           n_tiles, f_tiles, steps_per, sA, sB, sC, out_mode, yout, alpha, dbias, bias_mode);
    VAMD_LAUNCH_OK("conv_tn_mfma");
    return 0;
}

template <typename T>
static int launch_tn(const Gather& gp, const T* dy, const T* x, int N, int lda, int ldc, float* dw, int batch,
                     int64_t sA, int64_t sB, int64_t sC, int out_mode, void* yout, float alpha, float* dbias,
                     int bias_mode, hipStream_t st) {
    const TnShape sh = tn_shape(N, gp.Ktot, sizeof(T) == 2 ? VAMD_BF16 : VAMD_F32, out_mode);
    if (sh.AN == 64)
        return launch_tn_shape<T, 64, 192, 4>(gp, dy, x, N, lda, ldc, dw, batch, sA, sB, sC, out_mode, yout, alpha, dbias, bias_mode, st);
    if constexpr (sizeof(T) == 2) {
        if (sh.NW == 8 && sh.BF == 256)
            return launch_tn_shape<T, 128, 256, 8>(gp, dy, x, N, lda, ldc, dw, batch, sA, sB, sC, out_mode, yout, alpha, dbias, bias_mode, st);
        if (sh.NW == 8)
            return launch_tn_shape<T, 256, 128, 8>(gp, dy, x, N, lda, ldc, dw, batch, sA, sB, sC, out_mode, yout, alpha, dbias, bias_mode, st);
    }
    return launch_tn_shape<T, 128, 128, 4>(gp, dy, x, N, lda, ldc, dw, batch, sA, sB, sC, out_mode, yout, alpha, dbias, bias_mode, st);
}

size_t conv_wgrad_mfma_workspace(const ConvGeom& g, int dtype) {
    return (size_t)tn_wgrad_slices(g, dtype) * ((size_t)g.Cout * g.k * g.k * g.Cin + g.Cout) * sizeof(float);
}

int conv_wgrad_mfma(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw, float* dbias,
                    int oihw, void* workspace, size_t workspace_bytes, hipStream_t st, const float* pre_scale,
                    const float* pre_shift, float pre_slope, float* const* dw3, float* const* dbias3) {
    WgradSplit sp{0, {nullptr, nullptr}, {nullptr, nullptr}};
    if (dw3) {
        if (g.k != 1 || g.Cout % 3 != 0) return fail(VAMD_EINVAL, "conv_wgrad_mfma: three-way split needs a 1x1 filter and Cout % 3 == 0");
        sp.split_rows = g.Cout / 3;
        dw = dw3[0]; sp.dw[0] = dw3[1]; sp.dw[1] = dw3[2];
        if (dbias3) { dbias = dbias3[0]; sp.dbias[0] = dbias3[1]; sp.dbias[1] = dbias3[2]; }
    }
    Gather gp = make_gather(g.H, g.W, g.Cin, g.He, g.We, g.Ho, g.Wo, g.k, g.stride, g.pad, g.up, g.M);
    if (!workspace || workspace_bytes < conv_wgrad_mfma_workspace(g, dtype))
        return fail(VAMD_EINVAL, "conv_wgrad_mfma: needs %zu workspace bytes", conv_wgrad_mfma_workspace(g, dtype));
    if (gp.Ktot % 4) return fail(VAMD_EUNSUPPORTED, "conv_wgrad_mfma: k*k*Cin must be a multiple of 4");
    const int slices = tn_wgrad_slices(g, dtype);
    float* parts = reinterpret_cast<float*>(workspace);
    float* bias_parts = dbias ? parts + (size_t)slices * g.Cout * gp.Ktot : nullptr;
    int rc;
    if (wgrad_halo_ok(g, dtype))
        rc = launch_wgrad_halo(g, dy, x, parts, bias_parts, st, pre_scale, pre_shift, pre_slope);
    else if (pre_scale)
        return fail(VAMD_EUNSUPPORTED, "conv_wgrad_mfma: only the resident-gradient kernel normalises its input (ask vamd_conv2d_prenorm_supported)");
    else if (dtype == VAMD_BF16)
        rc = launch_tn<bf16_t>(gp, (const bf16_t*)dy, (const bf16_t*)x, g.Cout, g.Cout, gp.Ktot, parts, 1, 0, 0, 0, 0, nullptr, 1.f, bias_parts, 1, st);
    else
        rc = launch_tn<float>(gp, (const float*)dy, (const float*)x, g.Cout, g.Cout, gp.Ktot, parts, 1, 0, 0, 0, 0, nullptr, 1.f, bias_parts, 1, st);
    if (rc) return rc;
    const int64_t total4 = (int64_t)g.Cout * gp.Ktot / 4;
    const unsigned bias_blocks = dbias ? (unsigned)((g.Cout + 7) / 8) : 0u;
    if (slices >= 32)
        hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3((unsigned)((total4 + 7) / 8) + bias_blocks), dim3(256), 0, st, parts,
                           bias_parts, slices, g.Cout, gp.Ktot, g.Cin, g.k, oihw, dw, dbias, sp);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((total4 + 63) / 64) + bias_blocks), dim3(256), 0, st, parts,
                           bias_parts, slices, g.Cout, gp.Ktot, g.Cin, g.k, oihw, dw, dbias, sp);
    VAMD_LAUNCH_OK("wgrad_reduce");
    return 0;
}

// Weight gradient of (nearest 2x upsample -> 3x3 conv): the gradient of the combined 4x4 stride-2
// filter G of the input-gradient form,  dG[c][t][u][o] = sum over input pixels of x[b,i,j,c] *
// dY[b, 2i-1+t, 2j-1+u, o]  (the TN kernel with x as the plain operand and dY gathered 4x4 / stride 2:
// 16 taps per input pixel instead of 36), folded back by  dW[o][c][r][s] = sum of dG over the taps
// t, u whose sets {2}, {1,2}, {0,1}, {0} contain r, s.  The bias gradient is the pixel sum of the
// gathered dY columns of the four central taps, which cover every dY pixel exactly once.
static int up4_slices(const ConvGeom& g, int dtype) {
    const int F = 16 * g.Cout, KP = dtype == VAMD_BF16 ? 64 : 32;
    const TnShape sh = tn_shape(g.Cin, F, dtype, 0);
    int slices, steps_per;
    tn_split(sh, tn_shape_tiles(sh, g.Cin, F), (int)(((int64_t)g.B * g.H * g.W + KP - 1) / KP), &slices, &steps_per);
    return slices;
}

bool conv_wgrad_up4_mfma_supported(const ConvGeom& g, int dtype) {
    static const int on = [] { const char* e = getenv("VAMD_WGRAD_UP4"); return e ? atoi(e) : 1; }();
    if (!on || !g.up || g.k != 3 || g.stride != 1 || g.pad != 1 || g.Cout % 4) return false;
    return tn_shape_ok(g.Cout, g.Cin, (int64_t)g.B * g.H * g.W, dtype);
}

size_t conv_wgrad_up4_mfma_workspace(const ConvGeom& g, int dtype) {
    const size_t dg = (size_t)g.Cin * 16 * g.Cout;
    return ((size_t)up4_slices(g, dtype) * (dg + 16 * g.Cout) + dg) * sizeof(float);
}

// dG [Cin][4][4][Cout] (already summed over the slices) -> dW, and the bias partials -> dbias
__global__ void __launch_bounds__(256)
wgrad_up4_fold_kernel(const float* __restrict__ dg, const float* __restrict__ bias_parts, int slices, int Cin, int Cout,
                      int oihw, float* __restrict__ dw, float* __restrict__ dbias) {
    const int64_t total = (int64_t)Cin * Cout;
    const int64_t idx = blockIdx.x * 256ll + threadIdx.x;
    if (idx < total) {
        const int o = (int)(idx % Cout), c = (int)(idx / Cout);     // consecutive lanes: consecutive dG columns
        float g4[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) g4[t][u] = dg[(((int64_t)c * 4 + t) * 4 + u) * Cout + o];
        // conv tap r collects dY taps: r = 0 <- {2, 3}, r = 1 <- {1, 2}, r = 2 <- {0, 1}
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int t0 = 2 - r, u0 = 2 - q;
                const float v = (g4[t0][u0] + g4[t0][u0 + 1]) + (g4[t0 + 1][u0] + g4[t0 + 1][u0 + 1]);
                if (oihw) dw[(((int64_t)o * Cin + c) * 3 + r) * 3 + q] = v;
                else dw[(((int64_t)o * 3 + r) * 3 + q) * Cin + c] = v;
            }
    }
    if (dbias && blockIdx.x == 0) {
        for (int o = threadIdx.x; o < Cout; o += 256) {
            float sum = 0.f;
            for (int sl = 0; sl < slices; ++sl) {
                const float* bp = bias_parts + (int64_t)sl * 16 * Cout;
                sum += (bp[(1 * 4 + 1) * Cout + o] + bp[(1 * 4 + 2) * Cout + o]) + (bp[(2 * 4 + 1) * Cout + o] + bp[(2 * 4 + 2) * Cout + o]);
            }
            dbias[o] = sum;
        }
    }
}

int conv_wgrad_up4_mfma(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw, float* dbias,
                        int oihw, void* workspace, size_t workspace_bytes, hipStream_t st) {
    if (!workspace || workspace_bytes < conv_wgrad_up4_mfma_workspace(g, dtype))
        return fail(VAMD_EINVAL, "conv_wgrad_up4_mfma: needs %zu workspace bytes", conv_wgrad_up4_mfma_workspace(g, dtype));
    const int64_t M = (int64_t)g.B * g.H * g.W;
    Gather gp = make_gather(g.Ho, g.Wo, g.Cout, g.Ho, g.Wo, g.H, g.W, 4, 2, 1, 0, M);
    const int F = gp.Ktot;                                 // 16 * Cout
    const int slices = up4_slices(g, dtype);
    const size_t dgn = (size_t)g.Cin * F;
    float* parts = reinterpret_cast<float*>(workspace);
    float* bias_parts = parts + (size_t)slices * dgn;      // [slices][16 * Cout]
    float* dg = bias_parts + (size_t)slices * F;
    t_nominal_flops = 2.0 * g.M * g.Cout * 9.0 * g.Cin;
    int rc;
    if (dtype == VAMD_BF16)
        rc = launch_tn<bf16_t>(gp, (const bf16_t*)x, (const bf16_t*)dy, g.Cin, g.Cin, F, parts, 1, 0, 0, 0, 0, nullptr, 1.f, dbias ? bias_parts : nullptr, 2, st);
    else
        rc = launch_tn<float>(gp, (const float*)x, (const float*)dy, g.Cin, g.Cin, F, parts, 1, 0, 0, 0, 0, nullptr, 1.f, dbias ? bias_parts : nullptr, 2, st);
    if (rc) return rc;
    const int64_t total4 = (int64_t)dgn / 4;
    if (slices >= 32)
        hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3((unsigned)((total4 + 7) / 8)), dim3(256), 0, st, parts, nullptr, slices,
                           g.Cin, F, g.Cout, 1, 0, dg, nullptr, WgradSplit{0, {nullptr, nullptr}, {nullptr, nullptr}});
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, st, parts, nullptr, slices,
                           g.Cin, F, g.Cout, 1, 0, dg, nullptr, WgradSplit{0, {nullptr, nullptr}, {nullptr, nullptr}});
    VAMD_LAUNCH_OK("wgrad_reduce");
    hipLaunchKernelGGL(wgrad_up4_fold_kernel, dim3((unsigned)(((int64_t)g.Cin * g.Cout + 255) / 256)), dim3(256), 0, st, dg,
                       bias_parts, slices, g.Cin, g.Cout, oihw, dw, dbias);
    VAMD_LAUNCH_OK("wgrad_up4_fold");
    return 0;
}

// Batched products for the attention core on the same kernels.  Rows may be padded: lda/ldb/ldc
// are the element strides between rows (multiples of 16 bytes); batch z starts at z*rows*ld.
//   NT: C[z][m][n] = alpha * sum_k A[z][m][k] B[z][n][k]
static bool ld_ok(int ld, int dtype) { return ld % (dtype == VAMD_BF16 ? 8 : 4) == 0; }

bool bmm_nt_mfma_supported(int M, int N, int K, int lda, int ldb, int ldc, int dtype, int out_f32) {
    const int piece = dtype == VAMD_BF16 ? 8 : 4;
    const int kstep = dtype == VAMD_BF16 ? 64 : 32;
    const int Kc = round_up(K, piece);
    return ld_ok(lda, dtype) && ld_ok(ldb, dtype) && ldc % (out_f32 ? 4 : piece) == 0 && Kc <= lda && Kc <= ldb &&
           round_up(N, piece) <= ldc && 2 * Kc >= kstep && M > 0 && M < (1 << 26);
}

int bmm_nt_mfma(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda, int ldb,
                int ldc, float alpha, int dtype, int out_f32, hipStream_t st) {
    // columns K..round_up(K) of both operands must be finite padding (zeros) owned by the caller
    const int Kc = round_up(K, dtype == VAMD_BF16 ? 8 : 4);
    Gather gp = make_gather(1, M, Kc, 1, M, 1, M, 1, 1, 0, 0, M);
    gp.ldx = lda;
    Epilogue ep{nullptr, nullptr, nullptr, nullptr, alpha, out_f32, nullptr};
    if ((int64_t)batch * M * ldc * (out_f32 || dtype != VAMD_BF16 ? 4 : 2) >= (1ll << 31))    // 32-bit byte offsets of the kernels' stores
        return fail(VAMD_EUNSUPPORTED, "bmm_nt: output of %d x %d x %d elements exceeds the 2 GiB buffer-addressing limit", batch, M, ldc);
    if (dtype == VAMD_BF16)
        return run_nt<bf16_t>(gp, A, Bm, N, ldb, ldc, ep, C, batch, (int64_t)M * lda, (int64_t)N * ldb, (int64_t)M * ldc, st);
    return run_nt<float>(gp, A, Bm, N, ldb, ldc, ep, C, batch, (int64_t)M * lda, (int64_t)N * ldb, (int64_t)M * ldc, st);
}

//   TN: C[z][m][n] = alpha * sum_k A[z][k][m] B[z][k][n]
bool bmm_tn_mfma_supported(int M, int N, int K, int lda, int ldb, int ldc, int dtype) {
    const int piece = dtype == VAMD_BF16 ? 8 : 4;
    return ld_ok(lda, dtype) && ld_ok(ldb, dtype) && round_up(M, piece) <= lda && round_up(N, piece) <= ldb &&
           N <= ldc && K > 0 && K < (1 << 26);
}

int bmm_tn_mfma(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda, int ldb,
                int ldc, float alpha, int dtype, hipStream_t st) {
    // the reduction axis K plays the pixel axis; A's columns (M) play dY's channels, B's columns (N)
    // play the gathered input's.  Padding columns up to the next 16-byte piece must be finite.
    const int piece = dtype == VAMD_BF16 ? 8 : 4;
    Gather gp = make_gather(1, K, round_up(N, piece), 1, K, 1, K, 1, 1, 0, 0, K);
    gp.ldx = ldb;
    gp.Ktot = N;      // valid output columns
    if (dtype == VAMD_BF16)
        return launch_tn<bf16_t>(gp, (const bf16_t*)A, (const bf16_t*)Bm, M, lda, ldc, nullptr, batch, (int64_t)K * lda,
                                 (int64_t)K * ldb, (int64_t)M * ldc, 1, C, alpha, nullptr, 0, st);
    return launch_tn<float>(gp, (const float*)A, (const float*)Bm, M, lda, ldc, nullptr, batch, (int64_t)K * lda,
                            (int64_t)K * ldb, (int64_t)M * ldc, 1, C, alpha, nullptr, 0, st);
}

}  // namespace vamd

namespace vamd {
int conv_splitk_mode(int mode) {
    const int old = g_splitk_mode.load();
    if (mode == 0 || mode == 1) g_splitk_mode.store(mode);
    return old;
}
}  // namespace vamd
