// Weight gradient of the 3x3 / stride 1 / pad 1 layers with 64-channel tiles (bf16), gradient resident in registers.
//
//   dW[o][r][s][c] = sum over output pixels p of dY[p][o] * X[p + (r - 1, s - 1)][c]
//
// The generic weight-gradient kernel re-reads a pixel tile of X once per filter tap (nine times) and a tile of dY
// once per 192 filter columns; for 64-channel layers at full resolution that is ~1.4 KB from L2 into LDS per
// output pixel for 74 KFLOP.  Here a workgroup (8 waves, one per CU, persistent over 256-pixel patches) keeps the
// whole 64 x 9 x 64 gradient tile in its accumulators and streams, per patch, the dY patch (256 pixels x 128 B) and
// the X halo patch ((PH + 2) x (PW + 2) pixels x 128 B) through LDS ONCE: 298 B per output pixel.  Both are
// double-buffered direct-to-LDS loads; one barrier per patch.
//
// Work split inside the workgroup: the reduction runs over pixels in K-groups of 32 (one 16x16x32 MFMA step);
// waves 0-3 take K-groups 0-3 of a patch, waves 4-7 K-groups 4-7, and wave (w & 3) owns input-channel tile
// c = 16 (w & 3) .. +15 for all four output-channel tiles and all nine taps: 36 accumulator tiles (144 registers).
// Per K-group a wave reads the four dY fragments once and one X fragment per tap (26 transposing reads per 36
// MFMAs).  At the end waves 4-7 hand their accumulators to waves 0-3 through LDS, which store ONE partial product
// per workgroup; the slices are added in a fixed order by wgrad_reduce_kernel (no atomics, bit-reproducible).
//
// Both operands are pixel-major in memory (channels innermost) and reduction-major for the MFMA, so fragments come
// from ds_read_b64_tr_b16.  A half-wave's transposing read touches 8 LDS rows of 128 B, pixels {a .. a+3, a+8 ..
// a+11} of one patch row, 32 B each: conflict-free when the 16-byte chunk index is XORed with a function of the
// pixel's COLUMN in the patch, 2 * (bit 1 | bit 3 << 1).  Keyed on the column (not on the LDS row), the XOR term
// of a lane is the same for every K-group and tap row, so every read address is a per-lane constant (one per
// column tap and 4-pixel half) plus an immediate offset; the K-loop has no address arithmetic.
#include <utility>

#include "conv_nt_common.h"

namespace vamd {

struct WgHaloArgs {
    int B, H, W;
    int tiles_w, tiles_h, ntiles;
    int Cin, Cout;                  // multiples of 64
    int n_tiles, c_tiles;           // Cout / 64, Cin / 64: a workgroup owns one (output tile, input tile) pair
    int ldc;                        // 9 * Cin: row length of a partial product [Cout][9][Cin]
    // PRE: X is BatchNorm + LeakyReLU of the stored tensor, applied to the halo patch in LDS:
    // round(leaky(fma(x, pre_scale[c], pre_shift[c]))), what a separate normalise pass would have stored
    const float* pre_scale;
    const float* pre_shift;
    float pre_slope;
};

namespace {

template <int OFF> __device__ __forceinline__ s16x4_t wgh_read_tr(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "immediate offset of a DS instruction");
    s16x4_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

template <class F, int... I> __device__ __forceinline__ void wgh_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void wgh_for(F&& f) {
    wgh_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ int wgh_swz(int col) { return (((col >> 1) & 1) | (((col >> 3) & 1) << 1)) << 1; }

constexpr int wgh_halo_rows(int pwl) { return (((256 >> pwl) + 2) * ((1 << pwl) + 2) + 7) / 8 * 8; }
constexpr int kWghDyBytes = 256 * 128;
constexpr int wgh_buf_bytes(int pwl) { return wgh_halo_rows(pwl) * 128 + kWghDyBytes; }
constexpr int kWghRedBytes = 4 * 37 * 64 * 16;           // 36 accumulator tiles + the bias sums of waves 4-7, 16 B per lane
constexpr int kWghCoefBytes = 2 * 64 * 4;                // PRE: scale and shift of the workgroup's 64 input channels
constexpr int wgh_lds_bytes(int pwl) {
    return (2 * wgh_buf_bytes(pwl) > kWghRedBytes ? 2 * wgh_buf_bytes(pwl) : kWghRedBytes) + kWghCoefBytes;
}

template <int PWL, bool PRE>
__global__ void __launch_bounds__(512, 1)
wgrad_halo_kernel(WgHaloArgs a, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ parts,
                  float* __restrict__ bias_parts) {
    constexpr int PW = 1 << PWL, PH = 256 >> PWL, HWd = PW + 2, NPIX = (PH + 2) * HWd;
    constexpr int HROWS = wgh_halo_rows(PWL), XB = HROWS * 128, BUF = wgh_buf_bytes(PWL);
    constexpr int NPCX = (HROWS / 8 + 7) / 8;             // halo pieces (8 pixels x 128 B) per wave
    constexpr int KGR = PWL == 5 ? 1 : 2;                 // patch rows per K-group of 32 pixels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wn = wave_s & 3, grp = wave_s >> 2;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int combos = a.n_tiles * a.c_tiles;
    const int combo = bid % combos, nt = combo % a.n_tiles, ct = combo / a.n_tiles;
    const int wj = bid / combos, nblk = gridDim.x / combos;

    // ---- staging.  The LDS image of a direct-to-LDS load is lane-linear: lane l of piece pc fills 16-byte slot
    // (l & 7) of LDS row pc * 8 + (l >> 3); the swizzle is applied to the channel chunk it fetches.
    // Per piece one packed word: bits 0-7 patch column, 8-15 patch row, 16-30 channel offset, bit 31: no such pixel.
    int xs[NPCX];
#pragma unroll
    for (int j = 0; j < NPCX; ++j) {
        const int L = (wave + 8 * j) * 8 + (lane >> 3);
        const int R = L / HWd, hc = L - R * HWd;
        xs[j] = (L < NPIX ? 0 : (int)0x80000000) | ((ct * 64 + (((lane & 7) ^ wgh_swz(hc)) << 3)) << 16) | (R << 8) | hc;
    }
    int ds[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pp = (wave + 8 * j) * 8 + (lane >> 3);
        const int pr = pp >> PWL, pc = pp & (PW - 1);
        ds[j] = ((nt * 64 + (((lane & 7) ^ wgh_swz(pc)) << 3)) << 16) | (pr << 8) | pc;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x), rdy = make_rsrc(dy);
    uint32_t staged_ok = 0;                               // PRE: bit j = halo piece j of the patch staged last holds a pixel of the image
    auto stage = [&](int tile, int buf) {
        const int tx = tile % a.tiles_w;
        const int ty = (tile / a.tiles_w) % a.tiles_h;
        const int b = tile / (a.tiles_w * a.tiles_h);
        const int h0 = ty * PH, w0 = tx * PW;
        unsigned char* base = smem + buf * BUF + wave_s * 1024;
        staged_ok = 0;
#pragma unroll
        for (int j = 0; j < NPCX; ++j) {
            if ((wave_s + 8 * j) * 8 >= HROWS) continue;                     // uniform: no such piece
            const int hi = h0 - 1 + ((xs[j] >> 8) & 255), wi = w0 - 1 + (xs[j] & 255);
            const bool ok = xs[j] >= 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const int off = ((b * a.H + hi) * a.W + wi) * a.Cin + ((xs[j] >> 16) & 0x7fff);
            load16_lds(rx, (uint32_t)off * 2u, ok, base + j * 8192);
            if constexpr (PRE) staged_ok |= ok ? 1u << j : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hi = h0 + ((ds[j] >> 8) & 255), wi = w0 + (ds[j] & 255);
            const bool ok = hi < a.H && wi < a.W;                             // pixels past the image contribute zeros
            const int off = ((b * a.H + hi) * a.W + wi) * a.Cout + (ds[j] >> 16);
            load16_lds(rdy, (uint32_t)off * 2u, ok, base + XB + j * 8192);
        }
    };

    // PRE: a lane normalises the 16-byte slots it loaded itself, before the patch's barrier (its own loads have landed
    // once its vmcnt is 0); the slot's eight channels depend on the pixel's column, so the coefficients come from a
    // table in LDS behind the two buffers.  Slots of padding pixels keep the zeros the hardware wrote.
    float* coef = reinterpret_cast<float*>(smem + wgh_lds_bytes(PWL) - kWghCoefBytes);     // [2][64]
    if constexpr (PRE) {
        if (tid < 128) coef[tid] = (tid < 64 ? a.pre_scale : a.pre_shift)[ct * 64 + (tid & 63)];
        __syncthreads();
    }
    auto normalise = [&](int buf, uint32_t okbits) {
        // two pieces at a time (eight independent chains of arithmetic; registers do not allow more next to the 148
        // accumulators): every wave holds pieces 0 .. NPCX-2, the last one exists for the first waves only
        static_assert(NPCX == 6, "pieces in pairs; the sixth exists for the first waves only");
        unsigned char* base = smem + buf * BUF + wave_s * 1024 + lane * 16;
        auto one = [&](int j, u32x4_t& v, f32x2_t (&sc)[4], f32x2_t (&sh)[4]) {
            const int c0 = ((xs[j] >> 16) & 0x7fff) - ct * 64;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sc[e] = *reinterpret_cast<const f32x2_t*>(coef + c0 + 2 * e);
                sh[e] = *reinterpret_cast<const f32x2_t*>(coef + 64 + c0 + 2 * e);
            }
            v = *reinterpret_cast<const u32x4_t*>(base + j * 8192);
        };
#pragma unroll
        for (int j = 0; j < 6; j += 2) {
            const bool second = j < 4 || (wave_s + 8 * 5) * 8 < HROWS;          // uniform
            u32x4_t v0, v1 = u32x4_t{0u, 0u, 0u, 0u};
            f32x2_t sc0[4], sh0[4], sc1[4], sh1[4];
            one(j, v0, sc0, sh0);
            if (second) one(j + 1, v1, sc1, sh1);
            v0 = norm_act_piece(v0, sc0, sh0, a.pre_slope, 0u - ((okbits >> j) & 1u));
            *reinterpret_cast<u32x4_t*>(base + j * 8192) = v0;
            if (second) {
                v1 = norm_act_piece(v1, sc1, sh1, a.pre_slope, 0u - ((okbits >> (j + 1)) & 1u));
                *reinterpret_cast<u32x4_t*>(base + (j + 1) * 8192) = v1;
            }
        }
    };

    // ---- fragment addresses: lane (g, q, p) reads pixel kk = 8g + q (+ 4) of a K-group, channels 4p .. 4p + 3 of a tile
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int kk = 8 * g + q;
    const int prow_l = PWL == 5 ? 0 : kk >> 4, pcol = kk & (PW - 1);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LPTR(smem);
    uint32_t bA[4], bB[3][2];                             // of the buffer in use: they move by +-BUF from patch to patch
#pragma unroll
    for (int i = 0; i < 4; ++i)
        bA[i] = lds0 + XB + grp * 4 * 4096 + kk * 128 + (((2 * i + (p >> 1)) ^ wgh_swz(pcol)) << 4) + (p & 1) * 8;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hc = pcol + s + 4 * h;
            bB[s][h] = lds0 + (grp * 4 * KGR + prow_l) * (HWd * 128) + hc * 128 + (((2 * wn + (p >> 1)) ^ wgh_swz(hc)) << 4) + (p & 1) * 8;
        }

    f32x4_t acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t accb = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = bias_parts != nullptr && ct == 0;                    // uniform
    bf16x8_t ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

    int tile = wj, buf = 0;
    if (tile < a.ntiles) stage(tile, 0);
    for (; tile < a.ntiles; tile += nblk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PRE) {                              // own slots: this wave's loads have landed, no barrier needed yet
            normalise(buf, staged_ok);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                     // this patch is in LDS; every wave is done with the other buffer
        if (tile + nblk < a.ntiles) stage(tile + nblk, buf ^ 1);
        s16x4_t ra[4][2], rb[2][2];                        // landing registers of the transposing reads
        auto issueA = [&](auto kg_c) {
            constexpr int kg = decltype(kg_c)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i][0] = wgh_read_tr<kg * 4096>(bA[i]);
                ra[i][1] = wgh_read_tr<kg * 4096 + 512>(bA[i]);
            }
        };
        auto issueB = [&](auto u_c) {
            constexpr int u = decltype(u_c)::value, kg = u / 9, t = u % 9, r = t / 3, s = t % 3;
            constexpr int imm = (kg * KGR + r) * HWd * 128;
            rb[u & 1][0] = wgh_read_tr<imm>(bB[s][0]);
            rb[u & 1][1] = wgh_read_tr<imm>(bB[s][1]);
        };
        issueA(std::integral_constant<int, 0>{});
        issueB(std::integral_constant<int, 0>{});
        bf16x8_t fa[4];
        wgh_for<36>([&](auto u_c) {
            constexpr int u = decltype(u_c)::value, kg = u / 9, t = u % 9;
            if constexpr (u + 1 < 36) issueB(std::integral_constant<int, u + 1>{});
            if constexpr (t == 4 && kg + 1 < 4) issueA(std::integral_constant<int, kg + 1>{});
            // LDS reads return in order: the fragment of this step has landed once at most the reads issued after it are
            // outstanding (the next step's two; around the middle of a K-group also the next K-group's eight dY reads)
            constexpr int later = (u + 1 < 36 ? 2 : 0) + (((t == 4 || t == 5) && kg + 1 < 4) ? 8 : 0);
            if constexpr (t == 0) {
                asm volatile("s_waitcnt lgkmcnt(%10)"
                             : "+v"(rb[u & 1][0]), "+v"(rb[u & 1][1]), "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]),
                               "+v"(ra[1][1]), "+v"(ra[2][0]), "+v"(ra[2][1]), "+v"(ra[3][0]), "+v"(ra[3][1])
                             : "n"(later));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    fa[i] = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(ra[i][0], ra[i][1], 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(rb[u & 1][0]), "+v"(rb[u & 1][1]) : "n"(later));
                __builtin_amdgcn_sched_barrier(0);
            }
            const bf16x8_t fb = __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(rb[u & 1][0], rb[u & 1][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][t], 0, 0, 0);
            if constexpr (t == 0) {
                // bias gradient: per-channel pixel sums of dY as a product with a fragment of ones; wave wn takes output tile wn
                if (do_bias) {
                    if (wn == 0) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], ones, accb, 0, 0, 0);
                    else if (wn == 1) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], ones, accb, 0, 0, 0);
                    else if (wn == 2) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], ones, accb, 0, 0, 0);
                    else accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[3], ones, accb, 0, 0, 0);
                }
            }
        });
        const uint32_t step = buf ? (uint32_t)-BUF : (uint32_t)BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) bA[i] += step;
#pragma unroll
        for (int s = 0; s < 3; ++s) { bB[s][0] += step; bB[s][1] += step; }
        buf ^= 1;
    }

    // ---- waves 4-7 hand their sums to waves 0-3 through LDS; one partial product per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
                *reinterpret_cast<f32x4_t*>(red + (((wn * 37 + i * 9 + t) * 64 + lane) << 2)) = acc[i][t];
        *reinterpret_cast<f32x4_t*>(red + (((wn * 37 + 36) * 64 + lane) << 2)) = accb;
    }
    __syncthreads();
    if (grp == 1) return;
    float* part = parts + (int64_t)wj * a.Cout * a.ldc;
    const int col = ct * 64 + wn * 16 + (lane & 15);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4_t o = *reinterpret_cast<const f32x4_t*>(red + (((wn * 37 + i * 9 + t) * 64 + lane) << 2));
#pragma unroll
            for (int r = 0; r < 4; ++r)
                part[(int64_t)(nt * 64 + i * 16 + g * 4 + r) * a.ldc + t * a.Cin + col] = acc[i][t][r] + o[r];
        }
    if (do_bias && (lane & 15) == 0) {
        const f32x4_t o = *reinterpret_cast<const f32x4_t*>(red + (((wn * 37 + 36) * 64 + lane) << 2));
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_parts[(int64_t)wj * a.Cout + nt * 64 + wn * 16 + g * 4 + r] = accb[r] + o[r];
    }
}

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// patch shape of the image: 8 x 32 or 16 x 16 pixels, whichever covers it with fewer patches
void patch_grid(const ConvGeom& g, int* pwl, int* tiles_w, int* tiles_h) {
    const int n32 = ((g.H + 7) / 8) * ((g.W + 31) / 32), n16 = ((g.H + 15) / 16) * ((g.W + 15) / 16);
    const bool wide = n32 <= n16;
    *pwl = wide ? 5 : 4;
    *tiles_w = wide ? (g.W + 31) / 32 : (g.W + 15) / 16;
    *tiles_h = wide ? (g.H + 7) / 8 : (g.H + 15) / 16;
}

}  // namespace

bool wgrad_halo_ok(const ConvGeom& g, int dtype) {
    static const int on = env_int("VAMD_WGRAD_HALO", 1);
    static const int min_units = env_int("VAMD_WGRAD_HALO_MIN", 512);
    // (output tile, input tile) pairs of 64 x 64 channels a layer may have; each pair has 256 / pairs workgroups
    static const int max_pairs = env_int("VAMD_WGRAD_HALO_PAIRS", 64);
    if (!on || dtype != VAMD_BF16 || g.k != 3 || g.stride != 1 || g.pad != 1 || g.up) return false;
    const int pairs = (g.Cin / 64) * (g.Cout / 64);
    if (g.Cin % 64 || g.Cout % 64 || pairs > max_pairs || pairs > 64) return false;
    if (g.Ho != g.H || g.Wo != g.W) return false;
    const int64_t widest = g.Cin > g.Cout ? g.Cin : g.Cout;
    if ((int64_t)g.B * g.H * g.W * widest * 2 >= 0x7FFFFFF0ll) return false;      // 32-bit byte offsets of the buffer loads
    int pwl, tw, th;
    patch_grid(g, &pwl, &tw, &th);
    // enough (patch, pair) units to keep every CU busy for a few patches
    return (int64_t)g.B * tw * th * pairs >= min_units;
}

int wgrad_halo_slices(const ConvGeom& g) {
    int pwl, tw, th;
    patch_grid(g, &pwl, &tw, &th);
    const int64_t ntiles = (int64_t)g.B * tw * th;
    const int per = 256 / ((g.Cin / 64) * (g.Cout / 64));
    return (int)(ntiles < per ? ntiles : per);
}

int launch_wgrad_halo(const ConvGeom& g, const void* dy, const void* x, float* parts, float* bias_parts, hipStream_t st,
                      const float* pre_scale, const float* pre_shift, float pre_slope) {
    WgHaloArgs a;
    a.pre_scale = pre_scale; a.pre_shift = pre_shift; a.pre_slope = pre_slope;
    int pwl;
    patch_grid(g, &pwl, &a.tiles_w, &a.tiles_h);
    a.B = g.B; a.H = g.H; a.W = g.W;
    a.ntiles = g.B * a.tiles_w * a.tiles_h;
    a.Cin = g.Cin; a.Cout = g.Cout;
    a.n_tiles = g.Cout / 64; a.c_tiles = g.Cin / 64;
    a.ldc = 9 * g.Cin;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipSuccess;
        for (const void* k : {reinterpret_cast<const void*>(wgrad_halo_kernel<5, false>), reinterpret_cast<const void*>(wgrad_halo_kernel<5, true>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, wgh_lds_bytes(5));
        for (const void* k : {reinterpret_cast<const void*>(wgrad_halo_kernel<4, false>), reinterpret_cast<const void*>(wgrad_halo_kernel<4, true>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, wgh_lds_bytes(4));
        if (e != hipSuccess) return fail((int)e, "wgrad_halo: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        configured = true;
    }
    const int blocks = wgrad_halo_slices(g) * a.n_tiles * a.c_tiles;
    const double flops = 2.0 * g.M * g.Cout * 9.0 * g.Cin;
    TimedLaunch timed(TIMED_TN, flops, flops, st);
    if (pwl == 5 && pre_scale)
        hipLaunchKernelGGL((wgrad_halo_kernel<5, true>), dim3(blocks), dim3(512), wgh_lds_bytes(5), st, a, (const bf16_t*)dy,
                           (const bf16_t*)x, parts, bias_parts);
    else if (pwl == 5)
        hipLaunchKernelGGL((wgrad_halo_kernel<5, false>), dim3(blocks), dim3(512), wgh_lds_bytes(5), st, a, (const bf16_t*)dy,
                           (const bf16_t*)x, parts, bias_parts);
    else if (pre_scale)
        hipLaunchKernelGGL((wgrad_halo_kernel<4, true>), dim3(blocks), dim3(512), wgh_lds_bytes(4), st, a, (const bf16_t*)dy,
                           (const bf16_t*)x, parts, bias_parts);
    else
        hipLaunchKernelGGL((wgrad_halo_kernel<4, false>), dim3(blocks), dim3(512), wgh_lds_bytes(4), st, a, (const bf16_t*)dy,
                           (const bf16_t*)x, parts, bias_parts);
    VAMD_LAUNCH_OK("wgrad_halo");
    return 0;
}

}  // namespace vamd
