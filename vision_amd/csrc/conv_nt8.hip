// Implicit-GEMM forward / input-gradient kernel with two wave groups running half a phase apart
// ("ping-pong"), bf16, for the wide layers.
//
// One workgroup of 8 waves per CU, persistent over row tiles of one column tile.  A wave owns 128 pixels x
// 64 channels of the block tile (BM x BN = 256 x 256 as 2 x 4 waves, or 512 x 128 as 4 x 2): 128 fp32
// accumulator registers, 12 fragment reads per 32 MFMAs.  The reduction advances in K-tiles of 32 elements
// (64-byte LDS rows); a K-tile is two phases of 16 MFMAs per wave (pixel rows 0..63, then 64..127 of the
// wave tile).  A phase is
//
//     fragment reads of this phase | direct-to-LDS loads of the K-tile two ahead | barrier |
//     16 MFMAs | barrier
//
// and waves 4..7 (the second wave of each SIMD) run one barrier behind waves 0..3: while one wave of a SIMD
// issues its MFMAs, its partner reads fragments, computes gather addresses and issues loads, so the matrix
// pipe of every SIMD is fed by one wave at a time and never waits on the other's LDS traffic.
//
// LDS: a ring of four K-tile stages of (BM + BN) x 64 bytes (128 KB / 160 KB).  Loads are issued two K-tiles
// ahead of their use; a wave waits for its own loads of K-tile t+1 (counted vmcnt, the loads of K-tile t+2
// stay in flight) before the first barrier of the second phase of K-tile t; K-tile t+1 is read two
// barriers later by the wave's own group and three barriers later by the other group.  A stage is refilled
// two K-tiles after it was last read.  The K-tiles of consecutive row tiles form one stream: the ring
// does not drain between tiles.
//
// Work is cut by whole tiles: workgroup j of a column tile takes row tiles j, j + groups, ... so that the workgroups
// running at one time work on neighbouring tiles (their input halos and the filter meet in L2).  (A split of the
// K-tiles of a tile between workgroups, with partial sums exchanged through flags in global memory, was measured to
// cost what it saves and made the result depend on every workgroup of the grid being resident at once; it is gone.)
//
// 64-byte rows are read conflict-free by ds_read_b128 with the 16-byte chunk index XORed with
// f(row) = (4 - (row >> 2 & 3)) & 3; the direct-to-LDS image is lane-linear, so the permutation is applied
// to the source address of the load.
#include "conv_nt_common.h"

#ifndef VAMD_NT8_ABLATE
#define VAMD_NT8_ABLATE 0        // timing-only experiments (results are wrong): 1 no epilogue, 2 no set-up of the next tile's rows
#endif
#if VAMD_NT8_ABLATE != 0 && !defined(VAMD_TIMING_ONLY_BUILD)
#error "VAMD_NT8_ABLATE variants compute wrong results: build them with build_ext.py --variant NAME -DVAMD_NT8_ABLATE=n -DVAMD_TIMING_ONLY_BUILD"
#endif

namespace vamd {

constexpr int kNt8Stages = 4;
constexpr int kNt8RowB = 64;                  // bytes of reduction per LDS row (32 bf16)
constexpr int kNt8BK = 32;

__device__ __forceinline__ int nt8_swz(int row) { return (4 - ((row >> 2) & 3)) & 3; }

// Fragment reads as inline asm: the compiler orders an LDS read it knows of behind every direct-to-LDS load
// in flight (s_waitcnt vmcnt(0) in front of the read), which would drain the prefetch ring every K-tile.  The
// result is valid only after nt8_frag_wait; the ring's own waits and barriers order reads behind the loads.
template <int OFF> __device__ __forceinline__ bf16x8_t nt8_lds_read(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "immediate offset of a DS instruction");
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ void nt8_frag_wait(bf16x8_t (&a)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
}
__device__ __forceinline__ void nt8_frag_wait(bf16x8_t (&a)[4], bf16x8_t (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}

template <int BM, int BN, bool STATS>
__global__ void __launch_bounds__(512)
conv_nt8_kernel(Gather gp, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, int N, int ldw, int ldc,
                Epilogue ep, void* __restrict__ yout, int m_tiles, int n_tiles) {
    using T = bf16_t;
    using Frag = bf16x8_t;
    constexpr int NPA = BM / 128, NPB = BN / 128;          // 16-byte pieces per thread and K-tile
    constexpr int NLOAD = NPA + NPB;
    constexpr int A_BYTES = BM * kNt8RowB;
    constexpr int STAGE = (BM + BN) * kNt8RowB;
    constexpr int WGN = BN / 64;                           // waves along the channels
    constexpr int NT = 4, CPL = 16;                        // 16-channel tiles per wave, consecutive channels per lane
    static_assert((BM / 128) * WGN == 8, "eight waves of 128 x 64");
    static_assert(NPB >= 1, "at least 128 channels");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wn = wave_s % WGN, wm = wave_s / WGN;
    const int grp = wave_s >> 2;                           // waves w and w + 4 share a SIMD
    const int nblk = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int nt = bid % n_tiles;                          // this workgroup's column tile, fixed
    const int groups = nblk / n_tiles;                     // workgroups of this column tile
    const int wj = bid / n_tiles;                          // this workgroup's position among them
    const int n0 = nt * BN;
    const int nk = gp.Ktot / kNt8BK;
    // this workgroup's units (a unit = one K-tile of one row tile): row tiles wj, wj + groups, ...
    const int t_step = groups;
    const int n_units = wj < m_tiles ? (m_tiles - wj + groups - 1) / groups * nk : 0;
    const int t_first = wj;

    // ---- staging: this thread copies physical chunk `pch` of LDS rows lrow + 128*i
    const int lrow = tid >> 2, pch = tid & 3;
    const int kc = ((pch ^ nt8_swz(lrow)) & 3) * 8;      // logical chunk held by this LDS slot, in elements
    int b_off[NPB];
    bool b_ok[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
        const int R = lrow + 128 * i;                      // filter row of the tile as laid out for the waves
        const int h = R >> 6, q = R & 63;
        // MFMA row 4*g + r of channel tile j (LDS row q = 16*j + 4*g + r of the wave's 64) holds channel
        // (j >> 1)*32 + g*8 + (j & 1)*4 + r: a lane (g = lane >> 4) ends up with two runs of 8 consecutive channels,
        // and the four lanes of a pixel write 64 contiguous bytes per store instruction (whole 32-byte sectors)
        const int n = n0 + h * 64 + (q >> 5) * 32 + ((q & 15) >> 2) * 8 + ((q >> 4) & 1) * 4 + (q & 3);
        b_ok[i] = n < N;
        b_off[i] = (b_ok[i] ? n : 0) * ldw + kc;
    }
    int a_base[NPA];
    uint32_t a_mask[NPA];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x), rw = make_rsrc(w);
    // uniform reduction state of the K-tile being staged: channel offset, tap row / column / index
    int s_c = 0, s_tr = 0, s_ts = 0, s_tap = 0, s_k = 0;
    int s_tile = t_first, s_left = n_units;                // tile being staged, units left to stage
    bool s_more = s_left > 0;

    auto setup = [&](int row_tile) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const uint32_t m = row_tile * BM + lrow + 128 * i;
            const bool row_ok = m < (uint32_t)gp.M;
            const uint32_t b = fdiv(m, gp.div_howo);
            const uint32_t rem = m - b * (uint32_t)(gp.Ho * gp.Wo);
            const uint32_t ho = fdiv(rem, gp.div_wo);
            const int h0 = (int)ho * gp.stride - gp.pad;
            const int w0 = (int)(rem - ho * gp.Wo) * gp.stride - gp.padw;
            a_base[i] = (((int)b * gp.Hin + h0) * gp.Win + w0) * gp.ldx + kc;
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
        s_k = 0; s_c = 0; s_tap = 0; s_tr = 0; s_ts = 0;
    };
    // loads of the staged K-tile: half 0 = the first NPA/2 pixel pieces and the filter pieces, half 1 = the rest
    auto stage_half = [&](int stage, int half) {
        unsigned char* sa = smem + stage * STAGE + wave_s * 1024;
        const int tapoff = (s_tr * gp.Win + s_ts) * gp.ldx + s_c;
        const uint32_t bit = 1u << s_tap;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            if ((i < (NPA + 1) / 2) != (half == 0)) continue;
            load16_lds(rx, (uint32_t)(a_base[i] + tapoff) * 2u, (a_mask[i] & bit) != 0, sa + i * 8192);
        }
        if (half == 0) {
            unsigned char* sb = sa + A_BYTES;
            const int wf = ((gp.wr0 + gp.wstep * s_tr) * gp.wk + gp.ws0 + gp.wstep * s_ts) * gp.Cin + s_c;
#pragma unroll
            for (int i = 0; i < NPB; ++i) load16_lds(rw, (uint32_t)(b_off[i] + wf) * 2u, b_ok[i], sb + i * 8192);
        }
    };
    auto staged = [&]() {                                  // the staged K-tile is complete: move the stream on
        s_c += kNt8BK;
        if (s_c >= gp.Cin) {
            s_c = 0;
            ++s_tap;
            if (++s_ts == gp.ks) { s_ts = 0; ++s_tr; }
        }
        s_more = --s_left > 0;
        if (++s_k == nk) {
            s_tile += t_step;
#if VAMD_NT8_ABLATE == 2
            s_k = 0; s_c = 0; s_tr = 0; s_ts = 0; s_tap = 0;
            if (s_more && ldc == -1) setup(s_tile);
#else
            if (s_more) setup(s_tile);
#endif
        }
    };

    // ---- fragments
    const int frow = lane & 15, fgrp = lane >> 4;
    const int po = ((fgrp ^ nt8_swz(frow)) & 3) << 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LPTR(smem);
    const uint32_t fa_base = lds0 + (wm * 128 + frow) * kNt8RowB + po;
    const uint32_t fb_base = lds0 + A_BYTES + (wn * 64 + frow) * kNt8RowB + po;
    f32x4_t acc0[NT][4], acc1[NT][4];                     // [channel tile][pixel tile], pixel rows 0..63 / 64..127
    const int nb = n0 + wn * 64 + fgrp * 8;               // first channel of the lane's first run; the second starts at nb + 32
    float s1[STATS ? CPL : 1], s2[STATS ? CPL : 1];
#pragma unroll
    for (int e = 0; e < (STATS ? CPL : 1); ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    auto init_acc = [&]() {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (ep.bias) {
                const int n = nb + (j >> 1) * 32 + (j & 1) * 4;
                if (n + 4 <= N) {
                    const float4 t4 = *reinterpret_cast<const float4*>(ep.bias + n);
                    bv = f32x4_t{t4.x, t4.y, t4.z, t4.w};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[r] = n + r < N ? ep.bias[n + r] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc0[j][i] = bv; acc1[j][i] = bv; }
        }
    };

    if (!s_more) return;                                   // an empty share (never launched so)
    int cur = t_first, c_left = n_units;                   // tile being computed, units left
    init_acc();
    setup(s_tile);
    // prologue: two K-tiles in flight, the first one landed
    stage_half(0, 0); stage_half(0, 1); staged();
    if (s_more) {
        stage_half(1, 0); stage_half(1, 1); staged();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();            // the stagger: waves 4..7 run one barrier behind

    int st_r = 0, st_w = 2;                                // stage read / stage written
    Frag fa[4], fb[NT];
    while (c_left > 0) {
        // A wait the compiler can see (vmcnt(0), other counters untouched): without it, it takes loads of the
        // segment-end code for pending and drains vmcnt inside the K-tile loop, every K-tile.  (Stores count in
        // vmcnt in order with the loads, so the first counted wait of the loop would wait for the epilogue's
        // stores anyway; the ring is two K-tiles deep again one K-tile later.)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        for (int k = 0; k < nk; ++k) {
            const uint32_t sa = fa_base + st_r * STAGE;
            const uint32_t sb = fb_base + st_r * STAGE;
            // ---------------- phase 0: pixel rows 0..63
            fb[0] = nt8_lds_read<0>(sb);
            fb[1] = nt8_lds_read<16 * kNt8RowB>(sb);
            fb[2] = nt8_lds_read<32 * kNt8RowB>(sb);
            fb[3] = nt8_lds_read<48 * kNt8RowB>(sb);
            fa[0] = nt8_lds_read<0>(sa);
            fa[1] = nt8_lds_read<16 * kNt8RowB>(sa);
            fa[2] = nt8_lds_read<32 * kNt8RowB>(sa);
            fa[3] = nt8_lds_read<48 * kNt8RowB>(sa);
            const bool issue = s_more;
            if (issue) stage_half(st_w, 0);
            __builtin_amdgcn_s_barrier();
            nt8_frag_wait(fa, fb);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc0[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc0[j][i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 1: pixel rows 64..127 (filter fragments kept)
            fa[0] = nt8_lds_read<64 * kNt8RowB>(sa);
            fa[1] = nt8_lds_read<80 * kNt8RowB>(sa);
            fa[2] = nt8_lds_read<96 * kNt8RowB>(sa);
            fa[3] = nt8_lds_read<112 * kNt8RowB>(sa);
            if (issue) {
                stage_half(st_w, 1);
                staged();
            }
            if (issue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");   // the next K-tile has landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            nt8_frag_wait(fa);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc1[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc1[j][i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            st_r = (st_r + 1) & 3;
            st_w = (st_w + 1) & 3;
        }
        c_left -= nk;
        // ---- end of the tile (no barrier: the other group keeps its own pace)
#if VAMD_NT8_ABLATE == 1
        if (ldc == -1)                                     // never true: the epilogue stays compiled, and is skipped
#endif
        {
            // the lane's 16 channels in two halves: half the residual pieces and embedding terms live at a time
            const uint32_t m_first = (uint32_t)(cur * BM + wm * 128 + frow);
            // (the second run's accumulators are tiles 2 and 3 = template offset 8; its channels start at nb + 32)
            epilogue_rows<T, NT, STATS, 0, 8>(acc0, s1, s2, gp, ep.temb, ep.res, ep.res2, ep.alpha, 0, yout, N, ldc, 0, m_first, nb);
            epilogue_rows<T, NT, STATS, 8, 8>(acc0, s1 + (STATS ? 8 : 0), s2 + (STATS ? 8 : 0), gp, ep.temb, ep.res, ep.res2,
                                              ep.alpha, 0, yout, N, ldc, 0, m_first, nb + 24);
            epilogue_rows<T, NT, STATS, 0, 8>(acc1, s1, s2, gp, ep.temb, ep.res, ep.res2, ep.alpha, 0, yout, N, ldc, 0, m_first + 64, nb);
            epilogue_rows<T, NT, STATS, 8, 8>(acc1, s1 + (STATS ? 8 : 0), s2 + (STATS ? 8 : 0), gp, ep.temb, ep.res, ep.res2,
                                              ep.alpha, 0, yout, N, ldc, 0, m_first + 64, nb + 24);
        }
        cur += t_step;
        if (c_left > 0) init_acc();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();            // both groups have passed the same number of barriers

    if constexpr (STATS) {
        // per-lane sums -> the 16 pixel lanes of each group -> the pixel waves of this column group, through LDS
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
            }
        }
        __syncthreads();                                  // every wave is done with the staged tiles
        constexpr int WGM = BM / 128;
        float* red = reinterpret_cast<float*>(smem);      // [wm][BN channels of the tile][2]
        if (frow == 0) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) {
                const int col = wn * 64 + (e >> 3) * 32 + fgrp * 8 + (e & 7);
                red[(wm * BN + col) * 2 + 0] = s1[e];
                red[(wm * BN + col) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        const int part = bid / n_tiles;
        for (int idx = tid; idx < BN * 2; idx += 512) {
            const int col = idx >> 1;
            if (n0 + col < N) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < WGM; ++q) v += red[q * BN * 2 + idx];
                ep.stats[((int64_t)part * N + n0 + col) * 2 + (idx & 1)] = v;
            }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------
static int nt8_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// tile of the shape, or 0: 512 x 128 for 128 output channels per column tile, 256 x 256 for multiples of 256
static int nt8_tile(int N) {
    if (N % 256 == 0) return 256;
    if (N % 128 == 0) return 128;
    return 0;
}

struct Nt8Plan {
    int bm, bn, m_tiles, n_tiles, groups;                  // groups: workgroups per column tile
};

static Nt8Plan nt8_plan(const Gather& gp, int N) {
    Nt8Plan p;
    p.bn = nt8_tile(N);
    p.bm = p.bn == 256 ? 256 : 512;
    p.m_tiles = (gp.M + p.bm - 1) / p.bm;
    p.n_tiles = N / p.bn;
    // as many workgroups per column as fit the chip, row tiles spread evenly
    int rmax = 256 / p.n_tiles;
    if (rmax < 1) rmax = 1;
    const int rounds = (p.m_tiles + rmax - 1) / rmax;
    p.groups = (p.m_tiles + rounds - 1) / rounds;
    return p;
}

bool nt8_applicable(const Gather& gp, int N, int ldw, int ldc, const Epilogue& ep, int batch) {
    static const int on = nt8_env("VAMD_NT8", 1);
    static const int min_tiles = nt8_env("VAMD_NT8_MIN", 200);
    if (!on || batch != 1 || gp.shift || gp.ostep || gp.Cin2 > 0 || ep.out_f32 || ep.alpha != 1.0f) return false;
    if (gp.Cin % kNt8BK != 0 || gp.ldx != gp.Cin || ldc != N) return false;
    if (gp.kr * gp.ks > 16) return false;
    // 32-bit byte offsets through the buffer resources
    if ((int64_t)gp.Hin * gp.Win * gp.Cin * 2 * ((int64_t)gp.M / ((int64_t)gp.Ho * gp.Wo) + 1) >= 0x7FFFFFF0ll) return false;
    if ((int64_t)N * ldw * 2 >= 0x7FFFFFF0ll) return false;
    if (!nt8_tile(N)) return false;
    // Measured on MI355X against the 4-wave kernel (64 images): +4..7 % with 128-channel column tiles, +2 % on
    // 128 -> 256 channels at 56 x 56, -2..-5 % on 256- and 512-channel 3 x 3 layers (deeper reductions, fewer
    // tiles per CU), so the 256-wide tile serves short reductions only.
    // With epilogue terms the older kernels win (measured, 128 -> 128 channels at 112 x 112 x 64 images: residual
    // 401 vs 368 us, residual + skip 485 vs 389 us, accumulating input gradient 389 vs 361 us): all eight waves
    // here reach the epilogue of their 128 x 64 wave tiles together and wait for the terms' loads with no MFMA
    // work left to cover them.
    static const int with_terms = nt8_env("VAMD_NT8_TERMS", 0);
    if (ep.act_scale) return false;                          // the fused normalise lives in the 4-wave kernels
    if (!with_terms && (ep.temb || ep.res || ep.res2 || ep.stats)) return false;
    static const int wide_k = nt8_env("VAMD_NT8_WIDE_K", 1152);
    if (nt8_tile(N) == 256 && gp.Ktot > wide_k) return false;
    const Nt8Plan p = nt8_plan(gp, N);
    return (int64_t)p.m_tiles * p.n_tiles >= min_tiles;
}

int nt8_stats_parts(const Gather& gp, int N, const Epilogue& ep) { (void)ep; return nt8_plan(gp, N).groups; }

template <int BM, int BN>
static int launch_nt8_tile(const Gather& gp, const Nt8Plan& p, const bf16_t* x, const bf16_t* w, int N, int ldw, int ldc,
                           const Epilogue& ep, void* y, double flops, double executed, hipStream_t st) {
    constexpr int LDS = kNt8Stages * (BM + BN) * kNt8RowB;
    static bool configured = false;
    if (!configured) {
        for (const void* k : {reinterpret_cast<const void*>(conv_nt8_kernel<BM, BN, false>),
                              reinterpret_cast<const void*>(conv_nt8_kernel<BM, BN, true>)}) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return fail((int)e, "conv_nt8: hipFuncSetAttribute(%d) failed: %s", LDS, hipGetErrorString(e));
        }
        configured = true;
    }
    TimedLaunch timed(TIMED_NT, flops, executed, st);
    if (ep.stats)
        hipLaunchKernelGGL((conv_nt8_kernel<BM, BN, true>), dim3(p.groups * p.n_tiles), dim3(512), LDS, st, gp, x, w, N, ldw,
                           ldc, ep, y, p.m_tiles, p.n_tiles);
    else
        hipLaunchKernelGGL((conv_nt8_kernel<BM, BN, false>), dim3(p.groups * p.n_tiles), dim3(512), LDS, st, gp, x, w, N, ldw,
                           ldc, ep, y, p.m_tiles, p.n_tiles);
    VAMD_LAUNCH_OK("conv_nt8");
    return 0;
}

int launch_nt8(const Gather& gp, const void* x, const void* w, int N, int ldw, int ldc, const Epilogue& ep, void* y,
               double flops, double executed, hipStream_t st) {
    const Nt8Plan p = nt8_plan(gp, N);
    if (p.bn == 256)
        return launch_nt8_tile<256, 256>(gp, p, (const bf16_t*)x, (const bf16_t*)w, N, ldw, ldc, ep, y, flops, executed, st);
    return launch_nt8_tile<512, 128>(gp, p, (const bf16_t*)x, (const bf16_t*)w, N, ldw, ldc, ep, y, flops, executed, st);
}

}  // namespace vamd
