// C-ABI entry points of the convolution family: argument checks, then direct or MFMA kernels.
#include "conv_shapes.h"

namespace vamd {

int conv_geom(const vamd_conv_desc* d, ConvGeom* g, const char* who) {
    if (!d) return fail(VAMD_EINVAL, "%s: null descriptor", who);
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0)
        return fail(VAMD_EINVAL, "%s: non-positive extent B=%d H=%d W=%d Cin=%d Cout=%d", who, d->B,
                    d->H, d->W, d->Cin, d->Cout);
    if (!((d->ksize == 3 && d->pad == 1) || (d->ksize == 1 && d->pad == 0)))
        return fail(VAMD_EUNSUPPORTED, "%s: only 3x3 pad 1 and 1x1 pad 0 (got k=%d pad=%d)", who, d->ksize, d->pad);
    if (d->stride != 1 && d->stride != 2)
        return fail(VAMD_EUNSUPPORTED, "%s: stride must be 1 or 2 (got %d)", who, d->stride);
    if (d->upsample && (d->stride != 1))
        return fail(VAMD_EUNSUPPORTED, "%s: upsample needs stride 1", who);
    if (d->dtype != VAMD_F32 && d->dtype != VAMD_BF16) return fail(VAMD_EINVAL, "%s: bad dtype %d", who, d->dtype);
    g->B = d->B; g->H = d->H; g->W = d->W; g->Cin = d->Cin; g->Cout = d->Cout;
    g->k = d->ksize; g->stride = d->stride; g->pad = d->pad; g->up = d->upsample ? 1 : 0;
    g->He = g->up ? 2 * d->H : d->H;
    g->We = g->up ? 2 * d->W : d->W;
    g->Ho = (g->He + 2 * g->pad - g->k) / g->stride + 1;
    g->Wo = (g->We + 2 * g->pad - g->k) / g->stride + 1;
    g->M = (int64_t)g->B * g->Ho * g->Wo;
    g->scratch = d->scratch;
    g->scratch_bytes = d->scratch ? (size_t)d->scratch_bytes : 0;
    if (d->scratch && ((uintptr_t)d->scratch % 256 != 0))
        return fail(VAMD_EINVAL, "%s: scratch must be 256-byte aligned", who);
    const int64_t esz = d->dtype == VAMD_BF16 ? 2 : 4;
    const int64_t in_bytes = (int64_t)g->B * g->H * g->W * g->Cin * esz;
    const int64_t out_bytes = g->M * g->Cout * esz;
    if (in_bytes >= (1ll << 31) || out_bytes >= (1ll << 31))
        return fail(VAMD_EUNSUPPORTED, "%s: tensor of %lld bytes exceeds the 2 GiB buffer-addressing limit",
                    who, (long long)(in_bytes > out_bytes ? in_bytes : out_bytes));
    return 0;
}

}  // namespace vamd

using namespace vamd;

namespace {
// out[c] = sum over rows of t[row][c] for C <= 256.  A block's 256 threads cover RPB = 256 / C whole
// rows at a time (thread = (row lane, column), consecutive threads on consecutive addresses), the
// row lanes are folded through LDS, one atomic per block and column.
template <typename T>
__global__ void __launch_bounds__(256)
column_sums_kernel(const T* __restrict__ t, int64_t rows, int C, int ld, float* __restrict__ out) {
    __shared__ float red[256];
    const int rpb = 256 / C;
    const int rl = threadIdx.x / C, c = threadIdx.x - rl * C;
    float acc = 0.f;
    if (rl < rpb) {
        const int64_t step = (int64_t)gridDim.x * rpb;
        int64_t r = (int64_t)blockIdx.x * rpb + rl;
        for (; r + 3 * step < rows; r += 4 * step)
            acc += (to_f(t[r * ld + c]) + to_f(t[(r + step) * ld + c])) + (to_f(t[(r + 2 * step) * ld + c]) + to_f(t[(r + 3 * step) * ld + c]));
        for (; r < rows; r += step) acc += to_f(t[r * ld + c]);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < C) {
        float sum = 0.f;
        for (int q = 0; q < rpb; ++q) sum += red[q * C + threadIdx.x];
        atomicAdd(out + threadIdx.x, sum);
    }
}

int column_sums(int dtype, const void* t, int64_t rows, int C, float* out, hipStream_t st) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)C * sizeof(float), st);
    if (e != hipSuccess) return fail((int)e, "column_sums: memset failed: %s", hipGetErrorString(e));
    VAMD_DISPATCH(dtype, "column_sums", {
        for (int c0 = 0; c0 < C; c0 += 256) {          // 256 columns per launch
            const int cw = C - c0 < 256 ? C - c0 : 256;
            hipLaunchKernelGGL((column_sums_kernel<T>), dim3(grid_for(rows, 64, 2048)), dim3(256), 0, st,
                               (const T*)t + c0, rows, cw, C, out + c0);
        }
    });
    VAMD_LAUNCH_OK("column_sums");
    return 0;
}

// algo: 0 auto (thin -> MFMA -> direct), 1 direct, 2 MFMA, 3 thin
enum { kAuto = 0, kDirect = 1, kMfma = 2, kThin = 3 };

ConvGeom transposed(const ConvGeom& g) {      // geometry of the stride-1 input gradient seen as a convolution
    ConvGeom t = g;
    t.Cin = g.Cout;
    t.Cout = g.Cin;
    return t;
}

int pick(int algo, bool thin_ok, bool mfma_ok, const char* who, const ConvGeom& g) {
    if (algo == kDirect) return kDirect;
    if (algo == kThin) return thin_ok ? kThin : fail(VAMD_EUNSUPPORTED, "%s: thin kernels do not cover Cin=%d Cout=%d stride=%d", who, g.Cin, g.Cout, g.stride);
    if (algo == kMfma) return mfma_ok ? kMfma : fail(VAMD_EUNSUPPORTED, "%s: MFMA path does not cover Cin=%d Cout=%d", who, g.Cin, g.Cout);
    if (thin_ok) return kThin;
    return mfma_ok ? kMfma : kDirect;
}
}  // namespace

extern "C" {

static int conv2d_fwd_impl(const char* who, const vamd_conv_desc* d, const void* x, const void* w, const float* bias,
                           const float* temb, const void* residual, const void* residual2, void* y, float* stats,
                           void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g, who);
    if (rc) return rc;
    VAMD_REQUIRE(x && w && y, "%s: null tensor", who);
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)y % 16 == 0,
                 "%s: tensors must be 16-byte aligned", who);
    const int a = pick(d->algo, conv_thin_supported(g, d->dtype), conv_fwd_mfma_supported(g, d->dtype), who, g);
    if (a < 0) return a;
    if (stats && a != kMfma)
        return fail(VAMD_EUNSUPPORTED, "%s: fused statistics need the MFMA kernel (vamd_conv2d_stats_parts says 0)", who);
    if (a == kThin) return conv_fwd_thin(g, d->dtype, x, w, bias, temb, residual, residual2, y, as_stream(stream));
    if (a == kMfma) return conv_fwd_mfma(g, d->dtype, x, w, bias, temb, residual, residual2, y, stats, as_stream(stream));
    return conv_fwd_direct(g, d->dtype, x, w, bias, temb, residual, residual2, y, as_stream(stream));
}

int vamd_conv2d_fwd(const vamd_conv_desc* d, const void* x, const void* w, const float* bias,
                    const float* temb, const void* residual, const void* residual2, void* y,
                    void* stream) {
    return conv2d_fwd_impl("vamd_conv2d_fwd", d, x, w, bias, temb, residual, residual2, y, nullptr, stream);
}

int vamd_conv2d_fwd_stats(const vamd_conv_desc* d, const void* x, const void* w, const float* bias,
                          const float* temb, const void* residual, const void* residual2, void* y,
                          float* stats, void* stream) {
    VAMD_REQUIRE(stats, "vamd_conv2d_fwd_stats: null statistics buffer");
    return conv2d_fwd_impl("vamd_conv2d_fwd_stats", d, x, w, bias, temb, residual, residual2, y, stats, stream);
}

int vamd_conv2d_fwd_act_supported(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_fwd_act_supported")) return 0;
    const int a = pick(d->algo, conv_thin_supported(g, d->dtype), conv_fwd_mfma_supported(g, d->dtype), "", g);
    return a == kMfma && conv_fwd_act_mfma_supported(g, d->dtype) ? 1 : 0;
}

int vamd_conv2d_fwd_act(const vamd_conv_desc* d, const void* x, const void* w, const float* bias, const float* temb,
                        const void* residual, const void* residual2, const float* scale, const float* shift, float slope,
                        void* y, void* y_raw, void* stream) {
    const char* who = "vamd_conv2d_fwd_act";
    ConvGeom g;
    int rc = conv_geom(d, &g, who);
    if (rc) return rc;
    VAMD_REQUIRE(x && w && y && scale && shift, "%s: null tensor", who);
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)y_raw % 16 == 0 &&
                 (uintptr_t)scale % 16 == 0 && (uintptr_t)shift % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    if (!vamd_conv2d_fwd_act_supported(d))
        return fail(VAMD_EUNSUPPORTED, "%s: no kernel with a fused normalise for Cin=%d Cout=%d k=%d upsample=%d (ask "
                    "vamd_conv2d_fwd_act_supported first)", who, g.Cin, g.Cout, g.k, g.up);
    return conv_fwd_act_mfma(g, d->dtype, x, w, bias, temb, residual, residual2, scale, shift, slope, y, y_raw,
                             as_stream(stream));
}

int vamd_conv2d_prenorm_supported(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_prenorm_supported")) return 0;
    if (pick(d->algo, conv_thin_supported(g, d->dtype), conv_fwd_mfma_supported(g, d->dtype), "", g) != kMfma) return 0;
    if (pick(d->algo, conv_wgrad_thin_supported(g, d->dtype), conv_wgrad_mfma_supported(g, d->dtype), "", g) != kMfma) return 0;
    if (conv_wgrad_up4_mfma_supported(g, d->dtype)) return 0;
    return conv_prenorm_mfma_supported(g, d->dtype) ? 1 : 0;
}

int vamd_conv2d_fwd_prenorm(const vamd_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                            float in_slope, const void* w, const float* bias, const float* temb, const void* residual,
                            const void* residual2, void* y, float* stats, void* stream) {
    const char* who = "vamd_conv2d_fwd_prenorm";
    ConvGeom g;
    int rc = conv_geom(d, &g, who);
    if (rc) return rc;
    VAMD_REQUIRE(x && w && y && in_scale && in_shift, "%s: null tensor", who);
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)in_scale % 16 == 0 &&
                 (uintptr_t)in_shift % 16 == 0 && (uintptr_t)residual % 16 == 0 && (uintptr_t)residual2 % 16 == 0,
                 "%s: tensors must be 16-byte aligned", who);
    VAMD_REQUIRE(in_slope >= 0.f && in_slope <= 1.f, "%s: slope outside [0, 1]", who);
    if (!vamd_conv2d_prenorm_supported(d))
        return fail(VAMD_EUNSUPPORTED, "%s: no kernel normalises the input for Cin=%d Cout=%d k=%d stride=%d H=%d W=%d B=%d (ask "
                    "vamd_conv2d_prenorm_supported first)", who, g.Cin, g.Cout, g.k, g.stride, g.H, g.W, g.B);
    return conv_fwd_prenorm_mfma(g, d->dtype, x, in_scale, in_shift, in_slope, w, bias, temb, residual, residual2, y, stats,
                                 as_stream(stream));
}

int vamd_conv2d_wgrad_prenorm_oihw(const vamd_conv_desc* d, const void* dy, const void* x, const float* in_scale,
                                   const float* in_shift, float in_slope, float* dw, float* dbias, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    const char* who = "vamd_conv2d_wgrad_prenorm_oihw";
    ConvGeom g;
    int rc = conv_geom(d, &g, who);
    if (rc) return rc;
    VAMD_REQUIRE(dy && x && dw && in_scale && in_shift, "%s: null tensor", who);
    VAMD_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dw % 16 == 0 && (uintptr_t)workspace % 16 == 0 &&
                 (uintptr_t)in_scale % 16 == 0 && (uintptr_t)in_shift % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    VAMD_REQUIRE(in_slope >= 0.f && in_slope <= 1.f, "%s: slope outside [0, 1]", who);
    if (!vamd_conv2d_prenorm_supported(d))
        return fail(VAMD_EUNSUPPORTED, "%s: no kernel normalises the input for Cin=%d Cout=%d k=%d stride=%d H=%d W=%d B=%d (ask "
                    "vamd_conv2d_prenorm_supported first)", who, g.Cin, g.Cout, g.k, g.stride, g.H, g.W, g.B);
    return conv_wgrad_mfma(g, d->dtype, dy, x, dw, dbias, 1, workspace, workspace_bytes, as_stream(stream), in_scale, in_shift,
                           in_slope);
}

int vamd_conv2d_stats_parts(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_stats_parts")) return 0;
    const int a = pick(d->algo, conv_thin_supported(g, d->dtype), conv_fwd_mfma_supported(g, d->dtype), "", g);
    return a == kMfma ? conv_fwd_mfma_stats_parts(g, d->dtype) : 0;
}

int vamd_conv_splitk(int mode) { return conv_splitk_mode(mode); }

int vamd_conv2d_uses_mfma(const vamd_conv_desc* d, int pass) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_uses_mfma")) return 0;
    int a;
    if (pass == 0) a = pick(d->algo, conv_thin_supported(g, d->dtype), conv_fwd_mfma_supported(g, d->dtype), "", g);
    else if (pass == 1) a = pick(d->algo, conv_thin_supported(transposed(g), d->dtype), conv_dgrad_mfma_supported(g, d->dtype), "", g);
    else a = pick(d->algo, conv_wgrad_thin_supported(g, d->dtype), conv_wgrad_mfma_supported(g, d->dtype), "", g);
    return a == kMfma ? 1 : 0;
}

size_t vamd_conv2d_dgrad_workspace(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_dgrad_workspace")) return 0;
    const int a = pick(d->algo, conv_thin_supported(transposed(g), d->dtype), conv_dgrad_mfma_supported(g, d->dtype), "", g);
    return a == kMfma ? conv_dgrad_mfma_workspace(g, d->dtype) : 0;
}

int vamd_conv2d_dgrad(const vamd_conv_desc* d, const void* dy, const void* w_t, const void* add,
                      void* dx, void* workspace, size_t workspace_bytes, void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g, "vamd_conv2d_dgrad");
    if (rc) return rc;
    VAMD_REQUIRE(dy && w_t && dx, "vamd_conv2d_dgrad: null tensor");
    VAMD_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)w_t % 16 == 0 && (uintptr_t)dx % 16 == 0,
                 "vamd_conv2d_dgrad: tensors must be 16-byte aligned");
    const int a = pick(d->algo, conv_thin_supported(transposed(g), d->dtype), conv_dgrad_mfma_supported(g, d->dtype), "vamd_conv2d_dgrad", g);
    if (a < 0) return a;
    if (a == kThin) return conv_dgrad_thin(g, d->dtype, dy, w_t, add, dx, as_stream(stream));
    if (a == kMfma) return conv_dgrad_mfma(g, d->dtype, dy, w_t, add, dx, workspace, workspace_bytes, as_stream(stream));
    return conv_dgrad_direct(g, d->dtype, dy, w_t, add, dx, as_stream(stream));
}

int vamd_conv2d_fwd_up4_supported(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_fwd_up4_supported")) return 0;
    return (d->algo == kAuto || d->algo == kMfma) && conv_fwd_up4_mfma_supported(g, d->dtype) ? 1 : 0;
}

int vamd_conv2d_fwd_up4_stats_parts(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_fwd_up4_stats_parts")) return 0;
    return conv_fwd_up4_mfma_supported(g, d->dtype) ? conv_fwd_up4_mfma_stats_parts(g, d->dtype) : 0;
}

int vamd_conv2d_fwd_up4(const vamd_conv_desc* d, const void* x, const void* w_up4, const float* bias,
                        const float* temb, const void* residual, const void* residual2, void* y, float* stats,
                        void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g, "vamd_conv2d_fwd_up4");
    if (rc) return rc;
    VAMD_REQUIRE(x && w_up4 && y, "vamd_conv2d_fwd_up4: null tensor");
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_up4 % 16 == 0 && (uintptr_t)y % 16 == 0 &&
                     (uintptr_t)residual % 16 == 0 && (uintptr_t)residual2 % 16 == 0,
                 "vamd_conv2d_fwd_up4: tensors must be 16-byte aligned");
    if (!conv_fwd_up4_mfma_supported(g, d->dtype))
        return fail(VAMD_EUNSUPPORTED, "vamd_conv2d_fwd_up4: needs an upsampling 3x3 descriptor the MFMA path covers (Cin=%d Cout=%d up=%d)",
                    g.Cin, g.Cout, g.up);
    return conv_fwd_up4_mfma(g, d->dtype, x, w_up4, bias, temb, residual, residual2, y, stats, as_stream(stream));
}

int vamd_conv2d_fwd_short_supported(const vamd_conv_desc* d, int Cin2) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_fwd_short_supported")) return 0;
    return (d->algo == kAuto || d->algo == kMfma) && conv_fwd_short_mfma_supported(g, d->dtype, Cin2) ? 1 : 0;
}

int vamd_conv2d_fwd_short(const vamd_conv_desc* d, const void* x, const void* w, const float* bias, const float* temb,
                          const void* x2, const void* w2, int Cin2, const void* residual2, void* y, float* stats,
                          void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g, "vamd_conv2d_fwd_short");
    if (rc) return rc;
    VAMD_REQUIRE(x && w && x2 && w2 && y, "vamd_conv2d_fwd_short: null tensor");
    VAMD_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)x2 % 16 == 0 &&
                     (uintptr_t)w2 % 16 == 0 && (uintptr_t)residual2 % 16 == 0,
                 "vamd_conv2d_fwd_short: tensors must be 16-byte aligned");
    if (!(d->algo == kAuto || d->algo == kMfma) || !conv_fwd_short_mfma_supported(g, d->dtype, Cin2))
        return fail(VAMD_EUNSUPPORTED, "vamd_conv2d_fwd_short: needs a stride-1 descriptor the MFMA path covers and channel counts "
                    "that are multiples of %d (Cin=%d Cout=%d Cin2=%d)", d->dtype == VAMD_BF16 ? 64 : 32, g.Cin, g.Cout, Cin2);
    return conv_fwd_short_mfma(g, d->dtype, x, w, bias, temb, x2, w2, Cin2, residual2, y, stats, as_stream(stream));
}

int vamd_conv2d_dgrad_up4_supported(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_dgrad_up4_supported")) return 0;
    return (d->algo == kAuto || d->algo == kMfma) && conv_dgrad_up4_mfma_supported(g, d->dtype) ? 1 : 0;
}

int vamd_conv2d_dgrad_up4(const vamd_conv_desc* d, const void* dy, const void* w4, const void* add, void* dx,
                          void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g, "vamd_conv2d_dgrad_up4");
    if (rc) return rc;
    VAMD_REQUIRE(dy && w4 && dx, "vamd_conv2d_dgrad_up4: null tensor");
    VAMD_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)w4 % 16 == 0 && (uintptr_t)dx % 16 == 0,
                 "vamd_conv2d_dgrad_up4: tensors must be 16-byte aligned");
    if (!conv_dgrad_up4_mfma_supported(g, d->dtype))
        return fail(VAMD_EUNSUPPORTED, "vamd_conv2d_dgrad_up4: needs an upsampling 3x3 descriptor the MFMA path covers (Cin=%d Cout=%d up=%d)",
                    g.Cin, g.Cout, g.up);
    return conv_dgrad_up4_mfma(g, d->dtype, dy, w4, add, dx, as_stream(stream));
}

// workspace layout of the thin / direct kernels when the caller wants torch's layout: their own
// scratch first, then a [Cout][k][k][Cin] fp32 staging copy of the gradient
static size_t staging_offset(const ConvGeom& g, int dtype, int a) {
    const size_t own = a == kThin ? conv_wgrad_thin_workspace(g, dtype) : 0;
    return (own + 255) / 256 * 256;
}

size_t vamd_conv2d_wgrad_workspace(const vamd_conv_desc* d) {
    ConvGeom g;
    if (conv_geom(d, &g, "vamd_conv2d_wgrad_workspace")) return 0;
    const int a = pick(d->algo, conv_wgrad_thin_supported(g, d->dtype), conv_wgrad_mfma_supported(g, d->dtype), "", g);
    if (a == kMfma)
        return conv_wgrad_up4_mfma_supported(g, d->dtype) ? conv_wgrad_up4_mfma_workspace(g, d->dtype)
                                                          : conv_wgrad_mfma_workspace(g, d->dtype);
    return staging_offset(g, d->dtype, a) + (size_t)g.Cout * g.k * g.k * g.Cin * sizeof(float);
}

static int conv2d_wgrad_impl(const char* who, const vamd_conv_desc* d, const void* dy, const void* x, float* dw,
                             float* dbias, int oihw, void* workspace, size_t workspace_bytes, void* stream) {
    ConvGeom g;
    int rc = conv_geom(d, &g, who);
    if (rc) return rc;
    VAMD_REQUIRE(dy && x && dw, "%s: null tensor", who);
    VAMD_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dw % 16 == 0 && (uintptr_t)workspace % 16 == 0,
                 "%s: tensors must be 16-byte aligned", who);
    const int a = pick(d->algo, conv_wgrad_thin_supported(g, d->dtype), conv_wgrad_mfma_supported(g, d->dtype), who, g);
    if (a < 0) return a;
    if (a == kMfma) {
        if (conv_wgrad_up4_mfma_supported(g, d->dtype))
            return conv_wgrad_up4_mfma(g, d->dtype, dy, x, dw, dbias, oihw, workspace, workspace_bytes, as_stream(stream));
        return conv_wgrad_mfma(g, d->dtype, dy, x, dw, dbias, oihw, workspace, workspace_bytes, as_stream(stream));
    }
    if (dbias) {                     // the other kernels do not produce it: one extra pass over dy
        rc = column_sums(d->dtype, dy, g.M, g.Cout, dbias, as_stream(stream));
        if (rc) return rc;
    }
    float* out = dw;
    const size_t off = staging_offset(g, d->dtype, a);
    if (oihw && g.k > 1) {
        VAMD_REQUIRE(workspace && workspace_bytes >= off + (size_t)g.Cout * g.k * g.k * g.Cin * sizeof(float),
                     "%s: workspace too small", who);
        out = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + off);
    }
    if (a == kThin) rc = conv_wgrad_thin(g, d->dtype, dy, x, out, workspace, workspace_bytes < off ? workspace_bytes : off, as_stream(stream));
    else rc = conv_wgrad_direct(g, d->dtype, dy, x, out, as_stream(stream));
    if (rc || out == dw) return rc;
    return vamd_unpack_wgrad(out, dw, g.Cout, g.Cin, g.k, stream);
}

int vamd_conv2d_wgrad_oihw_split3(const vamd_conv_desc* d, const void* dy, const void* x, float* const dw[3],
                                  float* const dbias[3], void* workspace, size_t workspace_bytes, void* stream) {
    const char* who = "vamd_conv2d_wgrad_oihw_split3";
    ConvGeom g;
    int rc = conv_geom(d, &g, who);
    if (rc) return rc;
    VAMD_REQUIRE(dy && x && dw && dw[0] && dw[1] && dw[2], "%s: null tensor", who);
    VAMD_REQUIRE(!dbias || (dbias[0] && dbias[1] && dbias[2]), "%s: null bias gradient", who);
    VAMD_REQUIRE((uintptr_t)dy % 16 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dw[0] % 16 == 0 && (uintptr_t)dw[1] % 16 == 0 &&
                 (uintptr_t)dw[2] % 16 == 0 && (uintptr_t)workspace % 16 == 0, "%s: tensors must be 16-byte aligned", who);
    VAMD_REQUIRE(g.k == 1 && g.Cout % 3 == 0 && g.Cin % 4 == 0, "%s: needs a 1x1 filter, Cout a multiple of 3 and Cin of 4", who);
    const int a = pick(d->algo, conv_wgrad_thin_supported(g, d->dtype), conv_wgrad_mfma_supported(g, d->dtype), who, g);
    if (a < 0) return a;
    if (a != kMfma) return fail(VAMD_EUNSUPPORTED, "%s: needs the MFMA weight-gradient kernel (Cin=%d Cout=%d)", who, g.Cin, g.Cout);
    return conv_wgrad_mfma(g, d->dtype, dy, x, nullptr, nullptr, 1, workspace, workspace_bytes, as_stream(stream), nullptr, nullptr,
                           0.f, dw, dbias);
}

int vamd_conv2d_wgrad(const vamd_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                      void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_wgrad_impl("vamd_conv2d_wgrad", d, dy, x, dw, dbias, 0, workspace, workspace_bytes, stream);
}

int vamd_conv2d_wgrad_oihw(const vamd_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                           void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_wgrad_impl("vamd_conv2d_wgrad_oihw", d, dy, x, dw, dbias, 1, workspace, workspace_bytes, stream);
}

}  // extern "C"
