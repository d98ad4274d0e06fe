// Thin-channel convolutions on the matrix cores (bf16).
//
// The layers that touch the 3-channel image at full resolution move a 64-channel tensor through
// HBM once and do almost no arithmetic per byte, yet on the vector ALUs they are compute-bound
// (64 x 27 FMAs per pixel).  Mapping them onto v_mfma_f32_16x16x32_bf16 with the thin side padded
// inside the MFMA tile makes the arithmetic free and leaves the wide tensor's HBM traffic:
//
//   thin_in    Cin <= 4 -> Cout (K*K*Cin <= 32): one MFMA k-step holds the whole filter.  The
//              input window of a 16x16 pixel tile sits zero-padded in LDS; a lane builds its
//              patch fragment with 8 two-byte LDS reads.  Output channels are permuted inside the
//              MFMA rows so that every lane ends up with a contiguous run of channels.
//   thin_out   wide Cin -> Cout <= 16: a wave walks down a 16-pixel-wide strip; every input row
//              is loaded once (16 bytes per lane, straight into the B fragment) and feeds the K
//              output rows it belongs to, whose accumulators rotate through registers.
//   thin_wgrad weight gradient with one 64-channel side: the contraction runs over pixels, so
//              the wide operand is staged in LDS (global_load_lds, XOR-swizzled rows) and read
//              with the transposing ds_read_b64_tr_b16; block partial sums go to a slab that a
//              second kernel adds up in a fixed order.
#include "conv_shapes.h"

namespace vamd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

static __device__ __attribute__((aligned(256))) unsigned char t_zero_page[256];

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#define TGPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define TLPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ bf16x8_t zero8() {
    bf16x8_t z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    return z;
}

// ---------------------------------------------------------------------------------- thin_in
// Block = kThinInTiles consecutive 16x16 pixel tiles, 4 waves x 4 tile rows each.  Cout = 16 * NTN.
// The filter fragments and the per-channel epilogue constants stay in registers across the tiles.
constexpr int kThinInTiles = 4;

__host__ __device__ constexpr int thin_in_halo_bytes(int HT, int Cin) { return (HT * HT * Cin * 2 + 15) / 16 * 16; }

template <int K, int NTN>
__global__ void __launch_bounds__(256)
thin_in_mfma_kernel(ConvGeom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                    const float* __restrict__ bias, const float* __restrict__ temb,
                    const bf16_t* __restrict__ res, const bf16_t* __restrict__ res2, bf16_t* __restrict__ y,
                    int tiles_w, int tiles_h, int ntiles, int nt_store) {
    constexpr int HT = 16 + K - 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem_raw);          // [HT][HT][Cin], zero outside the image
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Cin = g.Cin, Ktot = K * K * Cin, Cout = 16 * NTN;
    // 64 output channels: a pixel is one 128-byte line, of which a lane holds 32 bytes.  A tile row's 16 pixels pass through
    // 2 KB of LDS owned by the wave (row = pixel, 16-byte chunk index XOR ((pixel >> 1) & 7): conflict-free both ways) and
    // lane L then stores chunk L & 7 of pixel 8 h + (L >> 3): whole lines per store instead of sixteen quarter lines, and
    // non-temporal for a large output (224 x 224 x 64 images: 126 -> 89 us, 1024 x 1024 x 4: 170 -> 107 us)
    constexpr bool XP = NTN == 4;
    const uint32_t xp_base = (uint32_t)(uintptr_t)TLPTR(smem_raw + thin_in_halo_bytes(HT, g.Cin)) + (uint32_t)wave * 2048u;

    // A = filter, rows permuted: MFMA jt, row i  <->  output channel (i>>2)*(4*NTN) + 4*jt + (i&3),
    // so that the 4*NTN accumulators of a lane are consecutive channels
    const int p = lane & 15, grp = lane >> 4;
    bf16x8_t afrag[NTN];
    int off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * grp + j;
        if (k < Ktot) {
            const int tap = k / Cin, c = k - tap * Cin;
            const int r = tap / K, s = tap - r * K;
            off[j] = (r * HT + s) * Cin + c;
        } else {
            off[j] = 0;                       // any finite element: its filter entry is zero
        }
#pragma unroll
        for (int jt = 0; jt < NTN; ++jt) {
            const int co = (p >> 2) * (4 * NTN) + 4 * jt + (p & 3);
            afrag[jt][j] = k < Ktot ? w[(int64_t)co * Ktot + k] : (bf16_t)0.f;
        }
    }
    const int c0 = grp * 4 * NTN;             // this lane's output channels: c0 .. c0 + 4*NTN - 1
    float bch[4 * NTN];
#pragma unroll
    for (int e = 0; e < 4 * NTN; ++e) bch[e] = bias ? bias[c0 + e] : 0.f;

    const int total = HT * HT * Cin;
    const int t_end = min((int)(blockIdx.x + 1) * kThinInTiles, ntiles);
    // halo elements of this thread (the same window positions for every tile) and their prefetch registers:
    // the next tile's window is in flight while this tile is computed and stored
    constexpr int NPRE = (HT * HT * 4 + 255) / 256;               // Cin <= 4
    int pyx[NPRE], pc[NPRE];
    bf16_t pre[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const int e = tid + 256 * q;
        const int c = e % Cin, pp = e / Cin;
        pyx[q] = e < total ? ((pp / HT) << 16) | (pp % HT) : -1;
        pc[q] = c;
    }
    auto fetch = [&](int tile) {
        const int tx = tile % tiles_w;
        const int ty = (tile / tiles_w) % tiles_h;
        const int b = tile / (tiles_w * tiles_h);
#pragma unroll
        for (int q = 0; q < NPRE; ++q) {
            const int hi = ty * 16 + (pyx[q] >> 16) - g.pad, wi = tx * 16 + (pyx[q] & 0xffff) - g.pad;
            bf16_t v = (bf16_t)0.f;
            if (pyx[q] >= 0 && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W)
                v = x[(((int64_t)b * g.H + hi) * g.W + wi) * Cin + pc[q]];
            pre[q] = v;
        }
    };
    int b_prev = -1;
    f32x4_t init[NTN];
    if ((int)blockIdx.x * kThinInTiles < t_end) fetch(blockIdx.x * kThinInTiles);
#pragma unroll 1
    for (int tile = blockIdx.x * kThinInTiles; tile < t_end; ++tile) {
        const int tx = tile % tiles_w;
        const int ty = (tile / tiles_w) % tiles_h;
        const int b = tile / (tiles_w * tiles_h);
        const int h0 = ty * 16, w0 = tx * 16;
        if (b != b_prev) {                    // accumulators start from bias + time embedding of the image
#pragma unroll
            for (int jt = 0; jt < NTN; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    init[jt][r] = bch[4 * jt + r] + (temb ? temb[(int64_t)b * Cout + c0 + 4 * jt + r] : 0.f);
            b_prev = b;
        }
        __syncthreads();                      // the previous tile's halo has been read by every wave
#pragma unroll
        for (int q = 0; q < NPRE; ++q)
            if (pyx[q] >= 0) halo[tid + 256 * q] = pre[q];
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);

#pragma unroll 1
        for (int rr = 0; rr < 4; ++rr) {
            const int tyy = wave * 4 + rr;
            const int h = h0 + tyy;
            if (h >= g.H) break;                  // wave-uniform
            const int base = (tyy * HT + p) * Cin;
            bf16x8_t bfrag;
#pragma unroll
            for (int j = 0; j < 8; ++j) bfrag[j] = halo[base + off[j]];
            f32x4_t acc[NTN];
#pragma unroll
            for (int jt = 0; jt < NTN; ++jt)
                acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[jt], bfrag, init[jt], 0, 0, 0);
            const int wq = w0 + p;
            if (wq < g.W) {
                const int64_t m = ((int64_t)b * g.H + h) * g.W + wq;
                // every residual load before the first store (a store in between would order them)
                Vec16<bf16_t> rv[NTN / 2], rv2[NTN / 2];
                if (res) {
#pragma unroll
                    for (int ch = 0; ch < NTN / 2; ++ch) rv[ch] = *reinterpret_cast<const Vec16<bf16_t>*>(res + m * Cout + c0 + ch * 8);
                }
                if (res2) {
#pragma unroll
                    for (int ch = 0; ch < NTN / 2; ++ch) rv2[ch] = *reinterpret_cast<const Vec16<bf16_t>*>(res2 + m * Cout + c0 + ch * 8);
                }
#pragma unroll
                for (int ch = 0; ch < NTN / 2; ++ch) {            // chunks of 8 channels = 16 bytes
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[ch * 2 + (e >> 2)][e & 3];
                    const int cc = c0 + ch * 8;
                    if (res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += to_f(rv[ch].v[e]);
                    }
                    if (res2) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += to_f(rv2[ch].v[e]);
                    }
                    Vec16<bf16_t> ov;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ov.v[e] = from_f<bf16_t>(v[e]);
                    if constexpr (XP) {
                        const uint32_t wa = xp_base + (uint32_t)p * 128u + (uint32_t)(((2 * grp + ch) ^ ((p >> 1) & 7)) << 4);
                        asm volatile("ds_write_b128 %0, %1" ::"v"(wa), "v"(__builtin_bit_cast(u32x4_t, ov)) : "memory");
                    } else {
                        *reinterpret_cast<Vec16<bf16_t>*>(y + m * Cout + cc) = ov;
                    }
                }
            }
            if constexpr (XP) {                               // every lane: pixels past the image edge are skipped below
                const int64_t m0 = ((int64_t)b * g.H + h) * g.W + w0;
                u32x4_t o0, o1;
                const uint32_t ra0 = xp_base + (uint32_t)(lane >> 3) * 128u + (uint32_t)(((lane & 7) ^ ((lane >> 4) & 7)) << 4);
                const uint32_t ra1 = xp_base + (uint32_t)(8 + (lane >> 3)) * 128u + (uint32_t)(((lane & 7) ^ ((4 + (lane >> 4)) & 7)) << 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(o0) : "v"(ra0) : "memory");
                asm volatile("ds_read_b128 %0, %1" : "=v"(o1) : "v"(ra1) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0), "+v"(o1)::"memory");
                const int px = lane >> 3;
                u32x4_t* d0 = reinterpret_cast<u32x4_t*>(y + (m0 + px) * Cout + (lane & 7) * 8);
                u32x4_t* d1 = reinterpret_cast<u32x4_t*>(y + (m0 + px + 8) * Cout + (lane & 7) * 8);
                if (nt_store) {
                    if (w0 + px < g.W) __builtin_nontemporal_store(o0, d0);
                    if (w0 + px + 8 < g.W) __builtin_nontemporal_store(o1, d1);
                } else {
                    if (w0 + px < g.W) *d0 = o0;
                    if (w0 + px + 8 < g.W) *d1 = o1;
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------- thin_out
// Wave = one strip of 16 pixels x R rows of one image.  Cin = 32 * NH, Cout <= 16.
template <int K, int NH>
__global__ void __launch_bounds__(256, 3)
thin_out_mfma_kernel(ConvGeom g, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                     const float* __restrict__ bias, const float* __restrict__ temb,
                     const bf16_t* __restrict__ res, const bf16_t* __restrict__ res2, bf16_t* __restrict__ y,
                     int strips_w, int strips_h, int R, int ntasks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int task = blockIdx.x * 4 + wave;
    if (task >= ntasks) return;                               // no barriers in this kernel
    const int sx = task % strips_w;
    task /= strips_w;
    const int sy = task % strips_h;
    const int b = task / strips_h;
    const int w0 = sx * 16, h0 = sy * R;
    const int h1 = h0 + R < g.H ? h0 + R : g.H;
    const int p = lane & 15, grp = lane >> 4;
    const int Cin = 32 * NH;

    // A = filter: row i = output channel, k = 8 channels of one tap
    bf16x8_t wf[K * K][NH];
#pragma unroll
    for (int t = 0; t < K * K; ++t)
#pragma unroll
        for (int hf = 0; hf < NH; ++hf)
            wf[t][hf] = p < g.Cout ? *reinterpret_cast<const bf16x8_t*>(w + ((int64_t)p * K * K + t) * Cin + hf * 32 + grp * 8)
                                   : zero8();

    // B = input: column = pixel w0 + p (shifted by the tap), k = the same 8 channels.  A row is loaded
    // once (the strip's 16 pixels, plus the pixel left of it in lane p = 0 and right of it in lane
    // p = 15); the fragments of the outer taps are the centre fragment moved one lane within the
    // 16-lane row (DPP row_shr / row_shl, the edge lane keeps the halo pixel).  K = 3 means pad = 1.
    constexpr int NR = K == 3 ? 2 : 1;                        // raw pieces per half: centre (, halo)
    auto load_row = [&](int hi, bf16x8_t (&f)[NR][NH]) {
        const bool hok = (unsigned)hi < (unsigned)g.H;
        const bf16_t* row = x + ((int64_t)b * g.H + hi) * g.W * Cin + grp * 8;
        const int wi = w0 + p;
        const bool ok = hok && wi < g.W;
#pragma unroll
        for (int hf = 0; hf < NH; ++hf)
            f[0][hf] = ok ? *reinterpret_cast<const bf16x8_t*>(row + (int64_t)wi * Cin + hf * 32) : zero8();
        if constexpr (K == 3) {
            const int wh = p == 0 ? w0 - 1 : w0 + 16;
            const bool okh = hok && (p == 0 || p == 15) && (unsigned)wh < (unsigned)g.W;
#pragma unroll
            for (int hf = 0; hf < NH; ++hf)
                f[1][hf] = okh ? *reinterpret_cast<const bf16x8_t*>(row + (int64_t)wh * Cin + hf * 32) : zero8();
        }
    };
    auto shifted = [&](const bf16x8_t& c, const bf16x8_t& h, auto left) -> bf16x8_t {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        const i32x4_t ci = __builtin_bit_cast(i32x4_t, c), hi4 = __builtin_bit_cast(i32x4_t, h);
        i32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (decltype(left)::value) o[e] = __builtin_amdgcn_update_dpp(hi4[e], ci[e], 0x111, 0xf, 0xf, false);   // row_shr:1
            else o[e] = __builtin_amdgcn_update_dpp(hi4[e], ci[e], 0x101, 0xf, 0xf, false);                                   // row_shl:1
        }
        return __builtin_bit_cast(bf16x8_t, o);
    };
    auto expand = [&](const bf16x8_t (&raw)[NR][NH], bf16x8_t (&f)[K][NH]) {
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            if constexpr (K == 3) {
                f[0][hf] = shifted(raw[0][hf], raw[1][hf], std::true_type{});    // pixel p - 1
                f[1][hf] = raw[0][hf];
                f[2][hf] = shifted(raw[0][hf], raw[1][hf], std::false_type{});   // pixel p + 1
            } else {
                f[0][hf] = raw[0][hf];
            }
        }
    };

    f32x4_t acc[K];
#pragma unroll
    for (int i = 0; i < K; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int hi_first = h0 - g.pad, hi_last = h1 - 1 + (K - 1 - g.pad);
    bf16x8_t cur[K][NH], raw[NR][NH];
    // per-lane epilogue constants (D row = output channel 4*grp + r): loads inside the row loop would
    // be younger than the next row's prefetch, and waiting for them would wait for the prefetch too
    float bt[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 4 * grp + r;
        bt[r] = 0.f;
        if (co < g.Cout) {
            if (bias) bt[r] += bias[co];
            if (temb) bt[r] += temb[(int64_t)b * g.Cout + co];
        }
    }
    // one input row is in flight while one is consumed (a second one in flight spills registers: 2.3x slower)
    auto row_step = [&](int hi) {
        expand(raw, cur);
        const int hout = hi + g.pad - (K - 1);                 // the output row completed by input row hi
        const int wq = w0 + p;
        const bool store = hout >= h0 && hout < h1 && wq < g.W;
        const int64_t m = ((int64_t)b * g.H + hout) * g.W + wq;
        // residual operands of that row first, then the next input row: loads return in order
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (store && (res || res2)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * grp + r;
                if (co < g.Cout) {
                    if (res) rv[r] += to_f(res[m * g.Cout + co]);
                    if (res2) rv[r] += to_f(res2[m * g.Cout + co]);
                }
            }
        }
        if (hi < hi_last) load_row(hi + 1, raw);
        // input row hi feeds output row hi + pad - r through filter row r; slot i = K-1-r holds
        // output row hi + pad - (K-1) + i
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int s = 0; s < K; ++s)
#pragma unroll
                for (int hf = 0; hf < NH; ++hf)
                    acc[K - 1 - r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[r * K + s][hf], cur[s][hf], acc[K - 1 - r], 0, 0, 0);
        if (store) {                                           // slot 0 is complete
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * grp + r;                    // D row = output channel
                if (co < g.Cout) y[m * g.Cout + co] = from_f<bf16_t>(acc[0][r] + bt[r] + rv[r]);
            }
        }
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) acc[i] = acc[i + 1];
        acc[K - 1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    load_row(hi_first, raw);
#pragma unroll 1
    for (int hi = hi_first; hi <= hi_last; ++hi) row_step(hi);
}

// ------------------------------------------------------------------------------- thin_wgrad
// Rows of 64 bf16 channels (128 bytes) in LDS, read with ds_read_b64_tr_b16: 32 lanes read 8 rows
// x 32 bytes.  Swizzling the 32-byte unit of a row by (row bit 1, row bit 3) spreads any 8 rows
// {a..a+3, a+8..a+11} over the 8 bank groups.
__device__ __forceinline__ int sw128(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

__device__ __forceinline__ bf16x4_t tr_read(const unsigned char* base, int row, int col) {
    const int piece = (col >> 3) ^ (sw128(row) << 1);
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(base + row * 128 + piece * 16 + (col & 7) * 2));
    return __builtin_bit_cast(bf16x4_t, v);
}

// dW[n][f] (n < 64 output channels, f = (tap, c) < 32) from dY [pixels][64] and X [pixels][Cin<=4].
// Block walks 16x16 pixel tiles; wave w contracts pixels 64w..64w+63 of each tile for all channels.
template <int K>
__global__ void __launch_bounds__(256)
thin_wgrad_wideout_kernel(ConvGeom g, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                          float* __restrict__ slab, int tiles_w, int tiles_h, int ntiles) {
    constexpr int HT = 16 + K - 1;
    constexpr int DYT = 256 * 128;                            // dY tile bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* dyt = smem_raw;
    bf16_t* halo = reinterpret_cast<bf16_t*>(smem_raw + DYT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Cin = g.Cin, Ktot = K * K * Cin;
    const int j16 = lane & 15, grp = lane >> 4;

    int off[2];
    bool fok[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int f = jt * 16 + j16;
        fok[jt] = f < Ktot;
        const int ff = fok[jt] ? f : 0;
        const int tap = ff / Cin, c = ff - tap * Cin;
        const int r = tap / K, s = tap - r * K;
        off[jt] = (r * HT + s) * Cin + c;
    }
    f32x4_t acc[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[nt][jt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int srow = tid >> 3, slot = tid & 7;
    const int piece = slot ^ (sw128(srow) << 1);              // rows srow + 32*i share the swizzle

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int q = t;
        const int tx = q % tiles_w;
        q /= tiles_w;
        const int ty = q % tiles_h;
        const int b = q / tiles_h;
        const int h0 = ty * 16, w0 = tx * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = i * 32 + srow;                      // tile pixel
            const int h = h0 + (k >> 4), wq = w0 + (k & 15);
            const bool ok = h < g.H && wq < g.W;
            const bf16_t* src = ok ? dy + (((int64_t)b * g.H + h) * g.W + wq) * 64 + piece * 8
                                   : reinterpret_cast<const bf16_t*>(t_zero_page);
            __builtin_amdgcn_global_load_lds(TGPTR(src), TLPTR(dyt + (i * 256 + wave * 64) * 16), 16, 0, 0);
        }
        const int total = HT * HT * Cin;
        for (int e = tid; e < total; e += 256) {
            const int c = e % Cin, pp = e / Cin;
            const int xx = pp % HT, yy = pp / HT;
            const int hi = h0 + yy - g.pad, wi = w0 + xx - g.pad;
            bf16_t v = (bf16_t)0.f;
            if ((unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W)
                v = x[(((int64_t)b * g.H + hi) * g.W + wi) * Cin + c];
            halo[e] = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int k0 = wave * 64 + ks * 32;
            // B = patches: column f, k = pixels k0 + 8*grp .. +7 (one tile row, 8 consecutive x)
            const int hb = (((k0 >> 4) + (grp >> 1)) * HT + 8 * (grp & 1)) * Cin;
            bf16x8_t bfr[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const bf16_t v = halo[hb + off[jt] + jj * Cin];
                    bfr[jt][jj] = fok[jt] ? v : (bf16_t)0.f;
                }
            // A = dY transposed: row n, k = the same pixels
            const int rq = k0 + 8 * grp + (j16 >> 2);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = nt * 16 + 4 * (lane & 3);
                const bf16x4_t lo = tr_read(dyt, rq, col), hi4 = tr_read(dyt, rq + 4, col);
                bf16x8_t a;
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi4[e]; }
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    acc[nt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[jt], acc[nt][jt], 0, 0, 0);
            }
        }
        __syncthreads();                                      // tile consumed before it is overwritten
    }

    // sum the 4 waves' partial products in LDS, then one slab row per block: [n][Ktot]
    float* red = reinterpret_cast<float*>(smem_raw);          // [wave][8][64][4]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
            *reinterpret_cast<f32x4_t*>(red + ((wave * 8 + nt * 2 + jt) * 64 + lane) * 4) = acc[nt][jt];
    __syncthreads();
    float* out = slab + (int64_t)blockIdx.x * 64 * Ktot;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = i * 256 + tid;                          // (idx8, lane', r)
        const float v = (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]);
        const int r = e & 3, l = (e >> 2) & 63, idx = e >> 8;
        const int n = (idx >> 1) * 16 + 4 * (l >> 4) + r, f = (idx & 1) * 16 + (l & 15);
        if (f < Ktot) out[n * Ktot + f] = v;
    }
}

// dW[n][tap][c] (n <= 4 output channels, c < 64) from dY [pixels][Cout<=4] and X [pixels][64].
// Wave w owns input channels 16w..16w+15 and all taps; the X window of a tile sits in LDS.
template <int K>
__global__ void __launch_bounds__(256)
thin_wgrad_widein_kernel(ConvGeom g, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                         float* __restrict__ slab, int tiles_w, int tiles_h, int ntiles) {
    constexpr int HT = 16 + K - 1;
    constexpr int NPASS = (HT * HT * 8 + 255) / 256;
    constexpr int XT = NPASS * 256 * 16;                      // X window bytes (whole passes)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* xt = smem_raw;
    bf16_t* dyt = reinterpret_cast<bf16_t*>(smem_raw + XT);   // [256 pixels][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j16 = lane & 15, grp = lane >> 4;

    f32x4_t acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int q = t;
        const int tx = q % tiles_w;
        q /= tiles_w;
        const int ty = q % tiles_h;
        const int b = q / tiles_h;
        const int h0 = ty * 16, w0 = tx * 16;
#pragma unroll
        for (int i = 0; i < NPASS; ++i) {
            const int pi = i * 256 + tid;
            const int hp = pi >> 3, slot = pi & 7;            // window pixel, physical piece
            const int piece = slot ^ (sw128(hp) << 1);
            const int yy = hp / HT, xx = hp - yy * HT;
            const int hi = h0 + yy - g.pad, wi = w0 + xx - g.pad;
            const bool ok = hp < HT * HT && (unsigned)hi < (unsigned)g.H && (unsigned)wi < (unsigned)g.W;
            const bf16_t* src = ok ? x + (((int64_t)b * g.H + hi) * g.W + wi) * 64 + piece * 8
                                   : reinterpret_cast<const bf16_t*>(t_zero_page);
            __builtin_amdgcn_global_load_lds(TGPTR(src), TLPTR(xt + (i * 256 + wave * 64) * 16), 16, 0, 0);
        }
        {
            const int h = h0 + (tid >> 4), wq = w0 + (tid & 15);
            const bool ok = h < g.H && wq < g.W;
            const bf16_t* src = dy + (((int64_t)b * g.H + h) * g.W + wq) * g.Cout;
#pragma unroll
            for (int n = 0; n < 4; ++n) dyt[tid * 4 + n] = (ok && n < g.Cout) ? src[n] : (bf16_t)0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll 1
        for (int ks = 0; ks < 8; ++ks) {
            const int k0 = ks * 32;
            // B = dY: column n (lanes with n >= 4 carry zeros), k = pixels k0 + 8*grp .. +7
            bf16x8_t bfr;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const bf16_t v = dyt[(k0 + 8 * grp + jj) * 4 + (j16 & 3)];
                bfr[jj] = j16 < 4 ? v : (bf16_t)0.f;
            }
            const int ky = (k0 >> 4) + (grp >> 1), kx = 8 * (grp & 1) + (j16 >> 2);
            const int col = wave * 16 + 4 * (lane & 3);
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const int hp = (ky + r) * HT + kx + s;     // window row of pixel (ky, kx) under tap (r, s)
                    const bf16x4_t lo = tr_read(xt, hp, col), hi4 = tr_read(xt, hp + 4, col);
                    bf16x8_t a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi4[e]; }
                    acc[r * K + s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr, acc[r * K + s], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    // D row = channel 16w + 4*grp + r, column = n; slab row per block laid out like dW [n][tap][c]
    if (j16 < g.Cout) {
        float* out = slab + (int64_t)blockIdx.x * g.Cout * K * K * 64;
#pragma unroll
        for (int t = 0; t < K * K; ++t)
            *reinterpret_cast<f32x4_t*>(out + ((int64_t)j16 * K * K + t) * 64 + wave * 16 + 4 * grp) = acc[t];
    }
}

// dw[i] = sum over blocks of slab[block][i], blocks added in a fixed order
__global__ void __launch_bounds__(256)
slab_sum_kernel(const float* __restrict__ slab, int nblocks, int total, float* __restrict__ dw) {
    __shared__ float red[8][32];
    const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    float sum = 0.f;
    if (idx < total)
        for (int b = q; b < nblocks; b += 8) sum += slab[(int64_t)b * total + idx];
    red[q][e] = sum;
    __syncthreads();
    if (q == 0 && idx < total)
        dw[idx] = ((red[0][e] + red[1][e]) + (red[2][e] + red[3][e])) + ((red[4][e] + red[5][e]) + (red[6][e] + red[7][e]));
}

// ------------------------------------------------------------------------------------ host
template <typename Kern> static int lds_limit(Kern kernel, size_t bytes, const char* who) {
    if (bytes <= 64 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail((int)e, "%s: hipFuncSetAttribute(%zu) failed: %s", who, bytes, hipGetErrorString(e));
    return 0;
}

bool thin_in_mfma_ok(const ConvGeom& g) {
    return g.stride == 1 && !g.up && (g.k == 1 || g.k == 3) && g.Cin <= 4 && g.k * g.k * g.Cin <= 32 &&
           (g.Cout == 32 || g.Cout == 64 || g.Cout == 128);
}

template <int K, int NTN>
static void launch_thin_in(const ConvGeom& g, const void* x, const void* w, const float* bias, const float* temb,
                           const void* res, const void* res2, void* y, hipStream_t st) {
    const int tw = (g.W + 15) / 16, th = (g.H + 15) / 16;
    constexpr int HT = 16 + K - 1;
    const size_t lds = (size_t)thin_in_halo_bytes(HT, g.Cin) + (NTN == 4 ? 4 * 2048 : 0);   // + the waves' transposing regions
    const int ntiles = g.B * th * tw;
    // non-temporal stores from 40 MB of output on: a smaller one the next kernel finds in the caches
    const int nt_store = (int64_t)g.B * g.H * g.W * g.Cout * 2 >= (40ll << 20);
    hipLaunchKernelGGL((thin_in_mfma_kernel<K, NTN>), dim3((unsigned)((ntiles + kThinInTiles - 1) / kThinInTiles)), dim3(256),
                       lds, st, g, (const bf16_t*)x, (const bf16_t*)w, bias, temb, (const bf16_t*)res,
                       (const bf16_t*)res2, (bf16_t*)y, tw, th, ntiles, nt_store);
}

int thin_in_mfma(const ConvGeom& g, const void* x, const void* w, const float* bias, const float* temb,
                 const void* res, const void* res2, void* y, hipStream_t st) {
    const int ntn = g.Cout / 16;
    if (g.k == 3) {
        if (ntn == 2) launch_thin_in<3, 2>(g, x, w, bias, temb, res, res2, y, st);
        else if (ntn == 4) launch_thin_in<3, 4>(g, x, w, bias, temb, res, res2, y, st);
        else launch_thin_in<3, 8>(g, x, w, bias, temb, res, res2, y, st);
    } else {
        if (ntn == 2) launch_thin_in<1, 2>(g, x, w, bias, temb, res, res2, y, st);
        else if (ntn == 4) launch_thin_in<1, 4>(g, x, w, bias, temb, res, res2, y, st);
        else launch_thin_in<1, 8>(g, x, w, bias, temb, res, res2, y, st);
    }
    VAMD_LAUNCH_OK("thin_in_mfma");
    return 0;
}

bool thin_out_mfma_ok(const ConvGeom& g) {
    if (g.stride != 1 || g.up || g.Cout > 16) return false;
    if (g.k == 3) return g.pad == 1 && (g.Cin == 32 || g.Cin == 64);
    if (g.k == 1) return g.Cin == 32 || g.Cin == 64 || g.Cin == 128;
    return false;
}

template <int K, int NH>
static void launch_thin_out(const ConvGeom& g, const void* x, const void* w, const float* bias, const float* temb,
                            const void* res, const void* res2, void* y, hipStream_t st) {
    const int R = 32;
    const int sw = (g.W + 15) / 16, sh = (g.H + R - 1) / R;
    const int ntasks = g.B * sw * sh;
    hipLaunchKernelGGL((thin_out_mfma_kernel<K, NH>), dim3((unsigned)((ntasks + 3) / 4)), dim3(256), 0, st, g,
                       (const bf16_t*)x, (const bf16_t*)w, bias, temb, (const bf16_t*)res, (const bf16_t*)res2,
                       (bf16_t*)y, sw, sh, R, ntasks);
}

int thin_out_mfma(const ConvGeom& g, const void* x, const void* w, const float* bias, const float* temb,
                  const void* res, const void* res2, void* y, hipStream_t st) {
    const int nh = g.Cin / 32;
    if (g.k == 3) {
        if (nh == 1) launch_thin_out<3, 1>(g, x, w, bias, temb, res, res2, y, st);
        else launch_thin_out<3, 2>(g, x, w, bias, temb, res, res2, y, st);
    } else {
        if (nh == 1) launch_thin_out<1, 1>(g, x, w, bias, temb, res, res2, y, st);
        else if (nh == 2) launch_thin_out<1, 2>(g, x, w, bias, temb, res, res2, y, st);
        else launch_thin_out<1, 4>(g, x, w, bias, temb, res, res2, y, st);
    }
    VAMD_LAUNCH_OK("thin_out_mfma");
    return 0;
}

bool thin_wgrad_mfma_ok(const ConvGeom& g) {
    if (g.stride != 1 || g.up || !(g.k == 1 || g.k == 3)) return false;
    if (g.Cin <= 4 && g.Cout == 64) return g.k * g.k * g.Cin <= 32;
    return g.Cin == 64 && g.Cout <= 4;
}

static int thin_wgrad_mfma_blocks(const ConvGeom& g) {
    const int ntiles = g.B * ((g.H + 15) / 16) * ((g.W + 15) / 16);
    return ntiles < 512 ? ntiles : 512;
}

size_t thin_wgrad_mfma_workspace(const ConvGeom& g) {
    return (size_t)thin_wgrad_mfma_blocks(g) * g.Cout * g.k * g.k * g.Cin * sizeof(float);
}

int thin_wgrad_mfma(const ConvGeom& g, const void* dy, const void* x, float* dw, void* workspace, hipStream_t st) {
    const int tw = (g.W + 15) / 16, th = (g.H + 15) / 16;
    const int ntiles = g.B * th * tw;
    const int nb = thin_wgrad_mfma_blocks(g);
    float* slab = reinterpret_cast<float*>(workspace);
    const int total = g.Cout * g.k * g.k * g.Cin;
    const int HT = 16 + g.k - 1;
    if (g.Cout == 64) {
        const size_t lds = 256 * 128 + (size_t)HT * HT * g.Cin * 2;
        if (g.k == 3)
            hipLaunchKernelGGL((thin_wgrad_wideout_kernel<3>), dim3(nb), dim3(256), lds, st, g, (const bf16_t*)dy,
                               (const bf16_t*)x, slab, tw, th, ntiles);
        else
            hipLaunchKernelGGL((thin_wgrad_wideout_kernel<1>), dim3(nb), dim3(256), lds, st, g, (const bf16_t*)dy,
                               (const bf16_t*)x, slab, tw, th, ntiles);
    } else {
        const int npass = (HT * HT * 8 + 255) / 256;
        const size_t lds = (size_t)npass * 256 * 16 + 256 * 4 * 2;
        if (g.k == 3)
            hipLaunchKernelGGL((thin_wgrad_widein_kernel<3>), dim3(nb), dim3(256), lds, st, g, (const bf16_t*)dy,
                               (const bf16_t*)x, slab, tw, th, ntiles);
        else
            hipLaunchKernelGGL((thin_wgrad_widein_kernel<1>), dim3(nb), dim3(256), lds, st, g, (const bf16_t*)dy,
                               (const bf16_t*)x, slab, tw, th, ntiles);
    }
    VAMD_LAUNCH_OK("thin_wgrad_mfma");
    hipLaunchKernelGGL(slab_sum_kernel, dim3((total + 31) / 32), dim3(256), 0, st, slab, nb, total, dw);
    VAMD_LAUNCH_OK("slab_sum");
    return 0;
}

}  // namespace vamd
