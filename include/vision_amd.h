/*
 * vision_amd.h -- C ABI of libvision_amd.so: hand-written gfx950 (MI355X) HIP kernels for the
 * DDPM UNet training step and sampler of HansaemIanOh/vision `models/ddpm.py`.
 *
 * The reference is pure Python on torch; it has no FFI of its own.  Each entry point below
 * therefore cites the reference *call site* (file:line) whose torch operator(s) it replaces.
 * A maintainer binds these with ctypes; `vision_amd/_lib.py` is that binding.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller.  The library never allocates,
 *     frees or synchronises; every call is asynchronous on `stream` (a hipStream_t passed as
 *     void*, NULL = the default stream).
 *   - Activations are NHWC: [B][H][W][C], element type `dtype` (VAMD_F32 or VAMD_BF16).
 *     Statistics, biases, gradients of parameters and optimiser state are always fp32.
 *   - Convolution weights for the kernels are [Cout][kh][kw][Cin] in `dtype`
 *     (vamd_repack_weight converts from torch's fp32 [Cout][Cin][kh][kw]).
 *   - Return value: 0 ok; <0 invalid argument (text in vamd_last_error()); >0 a hipError_t.
 */
#ifndef VISION_AMD_H
#define VISION_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VAMD_F32 = 0, VAMD_BF16 = 1 };
enum { VAMD_OK = 0, VAMD_EINVAL = -1, VAMD_EUNSUPPORTED = -2 };

/* Thread-local text of the last error returned on this thread ("" if none). */
const char* vamd_last_error(void);
/* Library / ABI version, and whether the MFMA path is compiled in. */
int vamd_version(void);

/* Stream plumbing, no counterpart in the reference (torch's autograd engine orders its streams itself): `waiter`
 * executes nothing more until all work queued on `signaller` before this call has finished.  The engine forks the weight
 * gradients of backward onto a second stream with it and joins them at the end. */
int vamd_stream_wait(void* waiter, void* signaller);

/* Benchmark aid, no counterpart in the reference.  vamd_launch_timing(1) makes the MFMA
 * convolution launchers record a HIP event pair around every kernel launch (on the stream of the
 * launch) and forgets earlier records; vamd_launch_timing(0) stops recording.
 * vamd_launch_timing_read waits for the recorded launches of `kind` (0: forward / input-gradient
 * kernel, 1: weight-gradient kernel) to finish - the one call of this library that blocks - and
 * returns their number, summed duration in ms, summed algorithmic FLOPs (2*M*N*K of the operator each)
 * and, in executed_flops (may be NULL), what the launches really multiplied: less than the operator's
 * count for the combined-filter forms of the upsampling convolution. */
int vamd_launch_timing(int enable);
long vamd_launch_timing_read(int kind, double* total_ms, double* total_flops, double* executed_flops);

/* ------------------------------------------------------------------ convolution
 * One descriptor for nn.Conv2d 3x3 pad 1 stride 1|2, 1x1 pad 0, and "nearest 2x upsample then
 * 3x3 conv" (upsample = 1).  H, W are the stored input extents (before upsampling). */
typedef struct vamd_conv_desc {
    int32_t B, H, W, Cin, Cout;
    int32_t ksize;     /* 1 or 3 */
    int32_t stride;    /* 1 or 2 */
    int32_t pad;       /* 0 or 1 */
    int32_t upsample;  /* 1: input is read through a nearest-neighbour 2x upsample */
    int32_t dtype;     /* VAMD_F32 | VAMD_BF16 (activations and packed weights) */
    int32_t algo;      /* 0 auto (thin -> MFMA -> direct), 1 direct, 2 MFMA, 3 thin-channel kernels */
    /* Optional, may be NULL / 0: device memory of the caller's that forward / input-gradient kernels may use
     * to even out their work over the compute units (partial sums of tiles whose reduction is split between
     * workgroups).  VAMD_CONV_SCRATCH_BYTES is enough for every shape.  The caller zero-fills it ONCE after
     * allocating it and does not touch it afterwards; launches that share one scratch buffer must be ordered
     * (one stream).  Results with and without it differ in the summation order of the split tiles only. */
    void* scratch;
    uint64_t scratch_bytes;
} vamd_conv_desc;
#define VAMD_CONV_SCRATCH_BYTES (4096u + 256u * 256u * 256u * 4u)

/* y = conv(x, w) + bias[c] + temb[b][c] + residual + residual2; each addend may be NULL.
 * residual2 carries the UNet skip tensor that the reference adds in place to the input of the
 * next up-path module (`h += skip[index]`, models/ddpm.py:231), folded into the producer.
 * Replaces nn.Conv2d at models/ddpm.py:19,35,52,59,68,95-98,156-157,203, the broadcast add of
 * the projected time embedding `h += temb` (:78), the residual `x + h` (:87, :123) and
 * nn.Upsample (:33) when desc.upsample. */
int vamd_conv2d_fwd(const vamd_conv_desc* d, const void* x, const void* w, const float* bias,
                    const float* temb, const void* residual, const void* residual2, void* y,
                    void* stream);

/* Forward convolution with the CONSUMER's eval-mode BatchNorm + LeakyReLU folded into the epilogue: the sampler's
 * `self.block2(...)` / next block's `self.block1(...)` prologue (reference models/ddpm.py:49-60 in eval mode, where the
 * normalisation uses running statistics and is a per-channel affine map known before the producer runs).
 *   q     = round_to_dtype(conv(x) + bias + temb + residual + residual2)        (what vamd_conv2d_fwd stores)
 *   y     = round_to_dtype(leaky_relu(fma(q, scale[c], shift[c]), slope))       (what vamd_scale_shift_act makes of q)
 *   y_raw = q, if y_raw != NULL (the pre-normalisation tensor, needed when it also feeds a shortcut or a skip)
 * scale / shift: [Cout] fp32, 16-byte aligned.  Bit-identical to the two separate calls.  Kernels exist for the shapes
 * for which vamd_conv2d_fwd_act_supported returns 1 (MFMA path, no upsampling gather). */
int vamd_conv2d_fwd_act_supported(const vamd_conv_desc* d);
int vamd_conv2d_fwd_act(const vamd_conv_desc* d, const void* x, const void* w, const float* bias, const float* temb,
                        const void* residual, const void* residual2, const float* scale, const float* shift, float slope,
                        void* y, void* y_raw, void* stream);
/* vamd_conv2d_fwd that also emits what a BatchNorm2d over its output needs in training mode
 * (reference models/ddpm.py:52-53: block2 = BatchNorm2d(out) -> act -> Dropout -> Conv2d runs
 * on the output of block1's convolution; :47-48 the same for the next block's block1):
 * stats[part][Cout][2] (fp32) = per row tile, the column sums of y and y*y over the values as
 * stored (rounded to the compute dtype).  vamd_conv2d_stats_parts(d) gives the number of parts,
 * 0 if the kernel serving this descriptor cannot produce them (use vamd_chan_stats then).
 * vamd_fold_partials + vamd_bn_finalize turn the parts into scale/shift. */
int vamd_conv2d_fwd_stats(const vamd_conv_desc* d, const void* x, const void* w, const float* bias,
                          const float* temb, const void* residual, const void* residual2, void* y,
                          float* stats, void* stream);
int vamd_conv2d_stats_parts(const vamd_conv_desc* d);

/* dx = conv_transpose(dy, w) (+ add).  `w_t` is the weight packed for the input gradient,
 * [Cin][kh][kw][Cout] with both kernel axes flipped (vamd_repack_weight makes it).
 * `add` (same shape as dx, may be NULL) is summed in: the second gradient stream of a residual.
 * Autograd mirror of the call sites listed for vamd_conv2d_fwd. */
int vamd_conv2d_dgrad(const vamd_conv_desc* d, const void* dy, const void* w_t, const void* add,
                      void* dx, void* workspace, size_t workspace_bytes, void* stream);
/* conv(x; w) + conv1x1(x2; w2) + bias (+ temb + residual2) in one launch: ResNetBlock's
 * `self.block2(h) + self.layers(x)` (models/ddpm.py:83-87) when the shortcut is a 1x1 convolution.
 * x2 is [B][H][W][Cin2] (the block's input), w2 the packed [Cout][Cin2] shortcut filter, `bias` the
 * SUM of both convolutions' biases; the shortcut runs as Cin2 extra reduction steps of the 3x3, so
 * its output tensor is never written or read.  Stride 1, no upsampling, Cin and Cin2 multiples of 64
 * (bf16) / 32 (fp32): vamd_conv2d_fwd_short_supported tells.  `stats` as in vamd_conv2d_fwd_stats,
 * sized by vamd_conv2d_stats_parts of the same descriptor (may be NULL). */
int vamd_conv2d_fwd_short_supported(const vamd_conv_desc* d, int Cin2);
int vamd_conv2d_fwd_short(const vamd_conv_desc* d, const void* x, const void* w, const float* bias,
                          const float* temb, const void* x2, const void* w2, int Cin2,
                          const void* residual2, void* y, float* stats, void* stream);

/* Forward of an upsampling descriptor (nearest 2x -> 3x3 conv, models/ddpm.py:29-41) as four 2x2
 * convolutions over the stored input, one per output parity class, in one launch: 16 taps per stored
 * pixel instead of 36.  `w_up4` is [4][Cout][2][2][Cin] from vamd_repack_weight_up4_fwd; epilogue
 * operands as in vamd_conv2d_fwd; `stats` (may be NULL) as in vamd_conv2d_fwd_stats with
 * vamd_conv2d_fwd_up4_stats_parts(d) leading rows. */
int vamd_conv2d_fwd_up4_supported(const vamd_conv_desc* d);
int vamd_conv2d_fwd_up4_stats_parts(const vamd_conv_desc* d);
int vamd_conv2d_fwd_up4(const vamd_conv_desc* d, const void* x, const void* w_up4, const float* bias,
                        const float* temb, const void* residual, const void* residual2, void* y, float* stats,
                        void* stream);
int vamd_repack_weight_up4_fwd(const float* w_oihw, void* w_up4, int Cout, int Cin, int dtype, void* stream);
/* Input gradient of an upsampling descriptor (nearest 2x -> 3x3 conv, models/ddpm.py:29-41) as one
 * stride-2 convolution over dy with the combined 4x4 filter `w_up4` ([Cin][4][4][Cout], from
 * vamd_repack_weight_up4): 16 taps per input pixel instead of 36, no pooling pass, no workspace.
 * vamd_conv2d_dgrad_up4_supported tells whether the MFMA path serves the descriptor this way;
 * vamd_conv2d_dgrad remains valid for every descriptor. */
int vamd_conv2d_dgrad_up4_supported(const vamd_conv_desc* d);
int vamd_conv2d_dgrad_up4(const vamd_conv_desc* d, const void* dy, const void* w_up4, const void* add,
                          void* dx, void* stream);
/* fp32 [Cout][Cin][3][3] -> the combined filter above, in `dtype`. */
int vamd_repack_weight_up4(const float* w_oihw, void* w_up4, int Cout, int Cin, int dtype, void* stream);
/* Scratch bytes vamd_conv2d_dgrad needs for this descriptor (0 for most; the gradient through the
 * fused upsample is formed at the upsampled resolution first). */
size_t vamd_conv2d_dgrad_workspace(const vamd_conv_desc* d);

/* dw[Cout][kh][kw][Cin] (fp32) = sum over pixels of dy (x) x.  Overwrites dw.  If `dbias` is not
 * NULL, dbias[Cout] (fp32) = sum over pixels of dy is written too: the bias gradient comes out of
 * the same pass over dy (the MFMA kernel multiplies the staged dy tile by a column of ones).
 * `workspace` (16-byte aligned) must hold vamd_conv2d_wgrad_workspace(d) bytes: the pixel range is
 * split over workgroups, every split stores its partial product there, and a second kernel adds
 * them in a fixed order (no atomics: the result is deterministic). */
int vamd_conv2d_wgrad(const vamd_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                      void* workspace, size_t workspace_bytes, void* stream);
/* The same with dw in torch's layout [Cout][Cin][kh][kw]: writes the parameter's .grad directly
 * (autograd of nn.Conv2d, models/ddpm.py:52-64), no vamd_unpack_wgrad pass. */
int vamd_conv2d_wgrad_oihw(const vamd_conv_desc* d, const void* dy, const void* x, float* dw, float* dbias,
                           void* workspace, size_t workspace_bytes, void* stream);
size_t vamd_conv2d_wgrad_workspace(const vamd_conv_desc* d);
/* vamd_conv2d_wgrad_oihw for a 1x1 convolution that is THREE convolutions side by side (Cout = 3 * C3, output channels
 * [j*C3, (j+1)*C3) belong to convolution j: the q_proj / k_proj / v_proj of AttnBlock, reference models/ddpm.py:95-97,
 * applied to the same normalised input in one launch): gradient rows and bias sums of convolution j are written to
 * dw[j] ([C3][Cin] fp32) and dbias[j] ([C3], or dbias == NULL), i.e. straight into three separate parameters' .grad. */
int vamd_conv2d_wgrad_oihw_split3(const vamd_conv_desc* d, const void* dy, const void* x, float* const dw[3],
                                  float* const dbias[3], void* workspace, size_t workspace_bytes, void* stream);

/* Convolution whose INPUT is BatchNorm2d + LeakyReLU of a stored tensor x (training or eval mode), with that
 * normalisation applied INSIDE the kernels: reference models/ddpm.py:49-60, `self.block1 = Sequential(BatchNorm2d, act,
 * Conv2d)` / `self.block2 = Sequential(BatchNorm2d, act, Dropout(0), Conv2d)` as one operator.  The kernels stage
 * x's halo patch in LDS once and replace every staged element by
 *     round_to_dtype(leaky_relu(fma(x, in_scale[c], in_shift[c]), in_slope))
 * (what vamd_scale_shift_act would have stored; padding stays zero), so the normalised tensor is never written or read:
 * the results are bit-identical to vamd_scale_shift_act followed by vamd_conv2d_fwd(_stats) / vamd_conv2d_wgrad_oihw.
 * in_scale / in_shift: [Cin] fp32, 16-byte aligned (vamd_bn_finalize's scale and shift); 0 <= in_slope <= 1.
 * vamd_conv2d_prenorm_supported: 1 when BOTH the forward and the weight gradient of the descriptor have such a kernel
 * (bf16, 3x3 stride 1, 64 input channels, 64 or 128 output channels, enough 256-pixel patches to fill the chip), so
 * that a caller may drop the normalised tensor altogether; the input gradient does not need it.
 * `stats` of the forward: NULL, or the buffer of vamd_conv2d_fwd_stats (vamd_conv2d_stats_parts rows). */
int vamd_conv2d_prenorm_supported(const vamd_conv_desc* d);
int vamd_conv2d_fwd_prenorm(const vamd_conv_desc* d, const void* x, const float* in_scale, const float* in_shift,
                            float in_slope, const void* w, const float* bias, const float* temb, const void* residual,
                            const void* residual2, void* y, float* stats, void* stream);
int vamd_conv2d_wgrad_prenorm_oihw(const vamd_conv_desc* d, const void* dy, const void* x, const float* in_scale,
                                   const float* in_shift, float in_slope, float* dw, float* dbias, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* Tuning control, process-wide: with a scratch buffer in the descriptor, 3x3 / 4x4 layers that have fewer tiles than the
 * chip has CUs and a reduction of at least 8 K-steps per filter row run one workgroup set per filter row and add the
 * fp32 partial products in a second, small kernel (same epilogue terms, one rounding; results differ from the unsplit
 * launch by fp32 summation order only).  mode 1 (default): as described; 0: never split; other values: query only.
 * Returns the previous mode.  Whether a layer splits depends on its batch size, so callers that compare batch sizes bit
 * for bit switch it off. */
int vamd_conv_splitk(int mode);
/* 1 if the MFMA implicit-GEMM kernels serve this descriptor for pass 0 (forward), 1 (input
 * gradient) or 2 (weight gradient), 0 if the direct kernels do. */
int vamd_conv2d_uses_mfma(const vamd_conv_desc* d, int pass);

/* fp32 [Cout][Cin][k][k] (torch layout) -> w_fwd [Cout][k][k][Cin] and, if non-NULL,
 * w_dgrad [Cin][k][k][Cout] flipped, both in `dtype`. */
int vamd_repack_weight(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin,
                       int ksize, int dtype, void* stream);
/* The same for every convolution of a model in ONE launch (the weight refresh after an optimiser
 * step, models/ddpm.py:339-352 -> next forward).  `items` is an array of n entries in DEVICE
 * memory; entry i covers blocks [first_block, first_block + ceil(Cout/32) * ceil(Cin/32)), entries
 * in ascending first_block order starting at 0; total_blocks = the sum of the counts. */
typedef struct vamd_repack_item {
    const float* w;        /* fp32 [Cout][Cin][k][k] */
    void* w_fwd;           /* [Cout][k][k][Cin] in dtype */
    void* w_dgrad;         /* [Cin][k][k][Cout] flipped, or NULL */
    long long first_block;
    int Cout, Cin, ksize;
    int ld_dgrad;          /* elements between consecutive [Cin][k][k] rows of w_dgrad; 0 = Cout.  Larger when the item's
                            * output channels are a column range of a wider filter (q / k / v projections side by side) */
    const float* bias;     /* with bias_out != NULL: bias_out[0 .. Cout) = bias[0 .. Cout) in the same launch */
    float* bias_out;
    void* w_up4_dgrad;     /* 3x3 filters of an upsampling convolution: what vamd_repack_weight_up4 writes, or NULL */
    void* w_up4_fwd;       /* ... and what vamd_repack_weight_up4_fwd writes, or NULL */
} vamd_repack_item;
int vamd_repack_weights(const vamd_repack_item* items, int n, long long total_blocks, int dtype, void* stream);
/* fp32 [Cout][k][k][Cin] -> fp32 [Cout][Cin][k][k] (gradient back to torch layout). */
int vamd_unpack_wgrad(const float* dw_krsc, float* dw_oihw, int Cout, int Cin, int ksize,
                      void* stream);

/* ------------------------------------------------------- normalisation + activation
 * Per-(g, c) sum and sum of squares of x viewed as [G][R][C] (BatchNorm: G = 1, R = B*H*W;
 * GroupNorm / per-image sums: G = B, R = H*W).  `partial` holds G * nchunk * C * 2 floats,
 * nchunk = vamd_stats_chunks(R, G).  Deterministic (no atomics).
 * Replaces the statistics half of nn.BatchNorm2d (models/ddpm.py:50,56) / nn.GroupNorm (:93). */
int vamd_stats_chunks(int64_t R, int G);
int vamd_chan_stats(const void* x, float* partial, int G, int64_t R, int C, int dtype,
                    void* stream);
/* As vamd_chan_stats, for the backward pass: sums of gz and gz*xhat where
 * z = x*scale+shift, gz = dy * mask * mask_scale * (z > 0 ? 1 : slope), xhat = (x - mean) * rstd.
 * scale/shift/mean/rstd are [G][C] fp32; mask may be NULL; slope = 1 disables the activation.
 * mask_scale is the fp32 factor of nn.Dropout, 1/(1-p), so that the mask itself can hold exact
 * 0/1 values in the compute dtype (1/0.9 is not a bf16 number). */
int vamd_norm_bwd_stats(const void* dy, const void* x, const void* mask, const float* scale,
                        const float* shift, const float* mean, const float* rstd, float slope,
                        float mask_scale, float* partial, int G, int64_t R, int C, int dtype, void* stream);

/* BatchNorm finalise (training = 1: batch statistics from `partial`, running statistics updated
 * with `momentum`, unbiased variance, num_batches_tracked += 1; training = 0: running statistics).
 * Outputs (each [C]): scale = gamma*rstd, shift = beta - mean*scale, mean, rstd. */
int vamd_bn_finalize(const float* partial, int nchunk, int C, int64_t count, const float* gamma,
                     const float* beta, float* running_mean, float* running_var,
                     int64_t* num_batches_tracked, float momentum, float eps, int training,
                     float* scale, float* shift, float* mean, float* rstd, void* stream);
/* GroupNorm finalise: groups of C/groups channels per image; outputs are [B][C]. */
int vamd_gn_finalize(const float* partial, int nchunk, int B, int C, int groups, int64_t HW,
                     const float* gamma, const float* beta, float eps, float* scale, float* shift,
                     float* mean, float* rstd, void* stream);
/* Allocation-free GroupNorm coefficients: vamd_chan_stats + vamd_gn_finalize in one call, with the partial
 * sums kept in a buffer built into the library instead of a workspace the caller allocates.  It has no
 * counterpart in the reference; it is for callers that cannot allocate device memory: code under HIP
 * graph capture, tight sampling loops, harnesses that forbid hipMalloc.  x is [B][HW][C] in `dtype`;
 * outputs are [B][C] fp32, bit-identical to the two-call form.  The built-in buffer holds
 * VAMD_GN_NOALLOC_SCRATCH_BYTES per GPU (kept until the process ends); a shape whose partial sums
 * (B * vamd_stats_chunks(HW, B) * C * 2 floats) do not fit is rejected with VAMD_EINVAL.  Calls on one GPU
 * share that GPU's buffer: calls that overlap in time on one GPU (different streams, threads) race.
 * Under HIP graph capture: the first call on a device looks the buffer's address up (hipGetSymbolAddress, under a
 * mutex; it may load the code object and synchronise), which is not a capturable operation -- make ONE call on the
 * device before the capture begins.  A captured graph holds the address of the one per-GPU buffer, so replays of
 * graphs that contain this call must not overlap on a GPU either. */
#define VAMD_GN_NOALLOC_SCRATCH_BYTES (8u << 20)
int vamd_gn_coef_noalloc(const void* x, int B, int C, int groups, int64_t HW, const float* gamma,
                         const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd,
                         int dtype, void* stream);
/* y = leaky_relu(x*scale[g][c] + shift[g][c], slope) * mask * mask_scale.  slope = 1: no activation;
 * mask (dropout keep mask, same shape as x) may be NULL.
 * Replaces BatchNorm2d/GroupNorm apply + nn.LeakyReLU (models/ddpm.py:51,57,144) + nn.Dropout (:58). */
int vamd_scale_shift_act(const void* x, const float* scale, const float* shift, const void* mask,
                         float slope, float mask_scale, void* y, int G, int64_t R, int C, int dtype, void* stream);
/* Backward finalise for BatchNorm: dgamma, dbeta and the three per-channel coefficients of
 * dx = cA*gz + cB + cC*xhat.  fixed_stats = 1: the forward normalised with the running statistics
 * (module in eval mode), mean and rstd are constants and cB = cC = 0. */
int vamd_bn_bwd_finalize(const float* partial, int nchunk, int C, int64_t count,
                         const float* scale, float* dgamma, float* dbeta, float* cA, float* cB,
                         float* cC, int fixed_stats, void* stream);
/* BatchNorm backward statistics with the rows grouped per image, and a finalise that also returns
 * dsum[b][c] = sum over the pixels of image b of dx[b][.][c] (and dsum_total[c], may be NULL, its sum
 * over b) without reading dx: `partial` holds B * nchunk * C * 3 floats (sums of gz, gz*xhat and
 * xhat per image), nchunk = vamd_stats_chunks(HW, B); scale/shift/mean/rstd are the batch's [C].
 * dsum is the gradient of the projected time embedding that ResNetBlock adds per image and channel
 * to the tensor this BatchNorm normalises (models/ddpm.py:79-82), dsum_total that of conv1's bias. */
int vamd_bn_bwd_stats_img(const void* dy, const void* x, const void* mask, const float* scale,
                          const float* shift, const float* mean, const float* rstd, float slope,
                          float mask_scale, float* partial, int B, int64_t HW, int C, int dtype, void* stream);
int vamd_bn_bwd_finalize_img(const float* partial, int nchunk, int B, int C, int64_t HW, const float* scale,
                             float* dgamma, float* dbeta, float* cA, float* cB, float* cC, float* dsum,
                             float* dsum_total, int fixed_stats, void* stream);
/* Backward finalise for GroupNorm (coefficients are [B][C]). */
int vamd_gn_bwd_finalize(const float* partial, int nchunk, int B, int C, int groups, int64_t HW,
                         const float* gamma, const float* rstd, float* dgamma, float* dbeta,
                         float* cA, float* cB, float* cC, void* stream);
/* dx = cA*gz + cB + cC*xhat (+ add + add2), gz and xhat as in vamd_norm_bwd_stats.  add / add2
 * merge the other gradient streams reaching x (residual shortcut, UNet skip). */
int vamd_norm_bwd_apply(const void* dy, const void* x, const void* mask, const float* scale,
                        const float* shift, const float* mean, const float* rstd, float slope,
                        float mask_scale, const float* cA, const float* cB, const float* cC, const void* add,
                        const void* add2, void* dx, int G, int64_t R, int C, int dtype,
                        void* stream);
/* Sums partial [G][nchunk][C][2] to per_g[G][C] (component 0) and total[C] = sum over g.
 * Used for the gradients of conv bias and of the projected time embedding. */
int vamd_sum_finalize(const float* partial, int nchunk, int G, int C, float* per_g, float* total,
                      void* stream);

/* out[q][width] = sum over parts p = q, q + nout, q + 2*nout, ... of in[p][width], in that order
 * (deterministic).  Shrinks the per-row-tile statistics of vamd_conv2d_fwd_stats (tens of
 * thousands of parts at 224x224) to the few hundred the finalize kernels read. */
int vamd_fold_partials(const float* in, int nparts, int width, int nout, float* out, void* stream);

/* ------------------------------------------------------------------- attention core
 * Batched products with padded rows: lda/ldb/ldc are element strides between rows, batch z of an
 * operand with R rows starts at z*R*ld.  Padding columns that the MFMA path reads (up to the next
 * 16 bytes of a row) must hold zeros.
 * NT: C[z][m][n] = alpha * sum_k A[z][m][k] * Bm[z][n][k]; C is fp32 if out_f32 else `dtype`.
 * TN: C[z][m][n] = alpha * sum_k A[z][k][m] * Bm[z][k][n].
 * Replace the two einsums at models/ddpm.py:117,120 and their gradients. */
int vamd_bmm_nt(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda,
                int ldb, int ldc, float alpha, int dtype, int out_f32, int algo, void* stream);
int vamd_bmm_tn(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda,
                int ldb, int ldc, float alpha, int dtype, int algo, void* stream);
/* y[b][h][w][c] = sum of the 2x2 block of x [B][2H][2W][C]: gradient of nn.Upsample (models/ddpm.py:33). */
int vamd_sumpool2x2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
/* Resampling blocks built without a convolution (with_conv: false): nn.AvgPool2d((2, 2)) (reference
 * models/ddpm.py:21) and bare nn.Upsample(scale_factor=2, mode='nearest') (models/ddpm.py:33), channels-last.
 * pool:   x [B][H][W][C] -> y [B][H/2][W/2][C] = scale * (2x2 block sum) + add   (add in y's shape, or NULL)
 * unpool: x [B][Hs][Ws][C] -> y [B][H][W][C], H/2 == Hs, W/2 == Ws: y = scale * x[h/2][w/2] + add; a trailing odd row or
 *         column of y (dropped by the pooling this transposes) receives add alone.
 * AvgPool2d forward = pool(0.25) and its input gradient = unpool(0.25); Upsample forward = unpool(1), gradient = pool(1). */
int vamd_pool2x2(const void* x, const void* add, void* y, int B, int H, int W, int C, float scale, int dtype, void* stream);
int vamd_unpool2x2(const void* x, const void* add, void* y, int B, int Hs, int Ws, int C, int H, int W, float scale, int dtype,
                   void* stream);
/* x [batch][R][C] -> y [batch][C][ldy] (ldy >= R; padding columns are left untouched). */
int vamd_transpose(const void* x, void* y, int batch, int R, int C, int ldy, int dtype, void* stream);
/* The same for an x whose rows are ldx >= C elements apart (batch z starts at z*R*ldx): a column range of a wider
 * tensor, such as k or v inside the output [B][S][3C] of the fused q / k / v projection. */
int vamd_transpose_ld(const void* x, void* y, int batch, int R, int C, int ldx, int ldy, int dtype, void* stream);
/* The attention core of AttnBlock.forward (reference models/ddpm.py:113-120) in one kernel, bf16: o = softmax(q k^T *
 * scale) v per image; q, k: [B][S][C]; vT: the values transposed, [B][C][ldv] (ldv >= S, a multiple of 8; columns past
 * S are not read); o: [B][S][C].  S <= 1024, C a multiple of 32 up to 512 (vamd_attention_fwd_supported).  The S x S
 * scores and probabilities stay on chip; a backward pass needs the probabilities and therefore the separate
 * vamd_bmm_nt / vamd_softmax_rows calls. */
int vamd_attention_fwd_supported(int S, int C, int dtype);
int vamd_attention_fwd(const void* q, const void* k, const void* vT, void* o, int B, int S, int C, int ldv, float scale,
                       int dtype, void* stream);
/* Row softmax over the first `cols` entries of rows of stride ld (F.softmax, models/ddpm.py:118)
 * and its gradient ds = alpha * p * (dp - sum(dp * p)).  x and dp are fp32, p and ds are `dtype`. */
int vamd_softmax_rows(const float* x, void* y, int64_t rows, int cols, int ld, int dtype,
                      void* stream);
int vamd_softmax_rows_bwd(const void* p, const float* dp, void* ds, int64_t rows, int cols, int ld,
                          float alpha, int dtype, void* stream);
/* y = a + b elementwise (the in-place skip add `h += skip[index]`, models/ddpm.py:231). */
int vamd_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);

/* ----------------------------------------------------------- time-embedding path (fp32)
 * emb[b] = [sin(w_j*t_b), cos(w_j*t_b)], w_j = 2*pi*exp(linspace(0, ln 1000, dim/2)) passed in
 * `angular_speeds` (dim/2 floats, tabulated once by the host exactly as models/ddpm.py:126-134). */
int vamd_timeembed(const float* t, const float* angular_speeds, float* emb, int B, int dim,
                   void* stream);
/* y[b][o] = sum_i act(x[b][i]) * W[o][i] + bias[o]; act = LeakyReLU(slope) if act_in else id.
 * The 1x1 convs on [B,C,1,1] maps at models/ddpm.py:156-157,215-217 and ResNetBlock.linear_temb (:47,77). */
int vamd_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int I,
                    int O, int act_in, float slope, void* stream);
/* Several linear layers over the SAME input in one launch: every ResNetBlock's linear_temb
 * (models/ddpm.py:47,77) projects the one time embedding of the step.  `items` is an array of n
 * entries (n <= 64) in DEVICE memory; entry k writes y[b][o] for o < O, and `first` is the sum of
 * the O of the entries before it (total_O = the sum over all entries). */
typedef struct vamd_linear_item {
    const float* W;        /* [O][I] */
    const float* bias;     /* [O] or NULL */
    float* y;              /* [B][O] */
    int O, first;
} vamd_linear_item;
int vamd_linear_fwd_multi(const float* x, const vamd_linear_item* items, int n, int total_O, int B, int I,
                          int act_in, float slope, void* stream);
/* The backward of vamd_linear_fwd_multi in two launches: dx[b][i] = act'(x) * sum over the entries and their outputs of
 * dy[b][o] W[o][i] (dx may be NULL; accumulate_dx: += instead of =), and every entry's dW [O][I] and dbias [O] (may be
 * NULL).  `items` lives in DEVICE memory, n <= 64; total_tiles = the sum over the entries of ceil(O / 16). */
typedef struct vamd_linear_bwd_item {
    const float* W;        /* [O][I] */
    const float* dy;       /* [B][O] */
    float* dW;             /* [O][I] */
    float* dbias;          /* [O] or NULL */
    int O, pad_;
} vamd_linear_bwd_item;
int vamd_linear_bwd_multi(const float* x, const vamd_linear_bwd_item* items, int n, int total_tiles, int B, int I, int act_in,
                          float slope, float* dx, int accumulate_dx, void* stream);
/* dx (may be NULL), dW, dbias of vamd_linear_fwd.  accumulate_dx: dx += instead of =. */
int vamd_linear_bwd(const float* x, const float* W, const float* dy, float* dx, float* dW,
                    float* dbias, int B, int I, int O, int act_in, float slope,
                    int accumulate_dx, void* stream);

/* ------------------------------------------------------------ diffusion elementwise
 * Cosine schedule + forward noising, fused with the NCHW fp32 -> NHWC `dtype` conversion:
 * angle = acos(.95) + t*(acos(.02)-acos(.95)); n = sin, s = cos; x_t = s*x0 + n*eps.
 * Writes n[B], s[B], n2[B] = n*n.  models/ddpm.py:242-255, :264-266, :258. */
int vamd_qsample(const float* x0_nchw, const float* eps_nchw, const float* t, void* xt_nhwc,
                 float* n, float* s, float* n2, int B, int C, int H, int W, int dtype,
                 void* stream);
/* schedule only (sampler): n, s, n2 for t[B]. */
int vamd_schedule(const float* t, float* n, float* s, float* n2, int B, void* stream);
/* loss = mean |pred - eps| (F.l1_loss, models/ddpm.py:268): pred NHWC `dtype`, eps NCHW fp32.
 * `partial` needs vamd_l1_chunks(n) floats; loss is one float. */
int vamd_l1_chunks(int64_t n);
int vamd_l1_loss_fwd(const void* pred_nhwc, const float* eps_nchw, float* partial, float* loss,
                     int B, int C, int H, int W, int dtype, void* stream);
/* dpred = gout[0] * sign(pred - eps) / numel. */
int vamd_l1_loss_bwd(const void* pred_nhwc, const float* eps_nchw, const float* gout,
                     void* dpred_nhwc, int B, int C, int H, int W, int dtype, void* stream);
/* One step of the deterministic sampler, models/ddpm.py:259 and :285-287:
 * x0_hat = (cur - n*eps_hat)/s ; cur <- s_next*x0_hat + n_next*eps_hat.
 * cur is fp32 NHWC (updated in place), next_in is the `dtype` copy fed to the network,
 * pred_images_nchw (may be NULL) receives x0_hat in NCHW fp32. */
int vamd_denoise_step(float* cur_nhwc, const void* eps_hat_nhwc, const float* n, const float* s,
                      const float* n_next, const float* s_next, void* next_in_nhwc,
                      float* pred_images_nchw, int B, int C, int H, int W, int dtype,
                      void* stream);
/* Layout / dtype converters at the module boundary (forward(x, t) takes and returns NCHW fp32).
 * f32_copy_nhwc (may be NULL) additionally receives an fp32 NHWC copy. */
int vamd_nchw_to_nhwc(const float* x_nchw, void* y_nhwc, float* f32_copy_nhwc, int B, int C,
                      int H, int W, int dtype, void* stream);
int vamd_nhwc_to_nchw(const void* x_nhwc, float* y_nchw, int B, int C, int H, int W, int dtype,
                      void* stream);

/* ------------------------------------------------------------------------ optimiser
 * torch.optim.AdamW (models/ddpm.py:340-345) over one flat fp32 arena of n elements.
 * lr is read from the device (lr_dev[0]) when lr_dev != NULL, else `lr`.  step is 1-based.
 * grad_scale multiplies the gradient first (1/world_size for a summed all-reduce). */
int vamd_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr,
               const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
               int step, float grad_scale, void* stream);
/* torch.optim.Adam with weight_decay (models/vqvaeunet.py:263-269): the decay enters the gradient, g + weight_decay * p,
 * before the moment updates.  Arguments as vamd_adamw. */
int vamd_adam_l2(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                 const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                 int step, float grad_scale, void* stream);
/* The same update with BOTH scalars that change from step to step read from device memory: lr_dev[0] and the 1-based
 * step count step_dev[0] (the bias corrections 1 - beta^step are formed in the kernel).  For a training step captured
 * into a HIP graph once and replayed: the launch's arguments never change, the caller increments step_dev on the device
 * (inside the graph) and rewrites lr_dev when a scheduler moves the rate.  l2_decay != 0 selects vamd_adam_l2's rule. */
int vamd_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* lr_dev, const int64_t* step_dev,
                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, int l2_decay, void* stream);

/* out[i] = a[i] (+ b[i] when b != NULL), i < n; every operand has its own dtype, the sum is taken in fp32.  The dtype
 * changes and two-term merges around the first stage's codebook lookup: the latents and their gradient are fp32 between
 * two halves of a network that runs in the compute dtype (reference models/vqvaeunet.py:210 and `h += skip[0]`, :230). */
int vamd_add_cast(const void* a, int a_dtype, const void* b, int b_dtype, void* out, int out_dtype, int64_t n, void* stream);

/* Head and reconstruction loss of the VQ-VAE-UNet first stage (reference models/vqvaeunet.py:206-210 nn.Sigmoid() after
 * the last channel-wise Linear, :236-240 rec_loss = F.mse_loss(x_rec, x)).
 * fwd: h [B][H][W][C] in `dtype` -> xrec [B][C][H][W] fp32 = sigmoid(h); with target (the input images, NCHW fp32) also
 *      loss[0] = mean((xrec - target)^2) through `partial` (vamd_l1_chunks(B*H*W) floats, fixed summation order).
 * bwd: dh = (g_loss[0] * 2 (xrec - target) / numel + g_xrec) * xrec * (1 - xrec); g_loss (one device float, with
 *      target) and g_xrec (NCHW fp32) may each be NULL. */
int vamd_sigmoid_mse_fwd(const void* h, const float* target, float* xrec, float* partial, float* loss, int B, int C, int H,
                         int W, int dtype, void* stream);
int vamd_sigmoid_mse_bwd(const float* xrec, const float* target, const float* g_loss, const float* g_xrec, void* dh, int B,
                         int C, int H, int W, int dtype, void* stream);

/* ---- latent diffusion first stage: vector-quantisation codebook lookup ------------------------------- */
/* Replaces VQBlock.forward (reference models/vqvaeunet.py:115-146; the same block in models/vqvae.py:13-55).
 * z: M = images * HW latent rows of D features, fp32, channels-first when nchw != 0 ([B, D, H, W], what
 * VQBlock.forward receives) or channels-last rows; codebook: [K, D] (nn.Embedding weight, vqvaeunet.py:112).
 * indices[M]: argmin_k of (|z|^2 + |e_k|^2) - 2 z.e_k evaluated in fp32 with every sum taken over d = 0..D-1 by
 * fused multiply-adds (vqvaeunet.py:121-126); equal distances resolve to the smallest index, a NaN distance wins.
 * out (layout of z): z + (e[index] - z), the straight-through value of vqvaeunet.py:144.
 * loss[0] = mse * beta + mse, loss[1] = mse = mean((e[index] - z)^2)  (vqvaeunet.py:138-141).
 * workspace: vamd_vq_workspace(M) bytes of device memory. */
size_t vamd_vq_workspace(int64_t M);
int vamd_vq_forward(const float* z, const float* codebook, int64_t M, int D, int K, int HW, int nchw, float beta,
                    int64_t* indices, float* out, float* loss, void* workspace, size_t workspace_bytes, void* stream);
/* Gradients of the two outputs of vamd_vq_forward as autograd defines them for vqvaeunet.py:138-144:
 * dz = g_out + g_loss * beta * 2 (z - e[index]) / (M D); dcodebook[k] = g_loss * 2 / (M D) * sum over the rows with
 * index k of (e[k] - z).  g_out (layout of z) and g_loss (one device float) may be NULL (= zero); dz or dcodebook
 * may be NULL (not wanted).  D <= 1024.  Rows of one code are added in a fixed order
 * (eight interleaved partial sums per 256-row chunk, combined pairwise): bit-reproducible. */
int vamd_vq_backward(const float* z, const float* codebook, const int64_t* indices, const float* g_out,
                     const float* g_loss, int64_t M, int D, int K, int HW, int nchw, float beta, float* dz,
                     float* dcodebook, void* stream);

/* ---- input pipeline: Resize((p, p)) + ToTensor of the data module on the GPU --------------------------- */
/* Replaces the per-sample transform of ImageFolderDataModule (reference dataloader.py:32-36): torchvision's
 * Resize on a PIL image (Pillow's two-pass antialiased bilinear resampler on 8-bit data, Pillow 12.2
 * src/libImaging/Resample.c) followed by ToTensor (uint8 HWC -> fp32 CHW / 255).  Bit-identical to that path.
 * vamd_resize_ksize / vamd_resize_coeffs (host only, no GPU): the window [bounds[2i], bounds[2i] + bounds[2i+1]) and
 * the ksize 22-bit fixed-point weights of output coordinate i for an axis resized from in_size to out_size.
 * vamd_image_resize_u8: src [H][W][pixel_bytes] uint8 in device memory (pixel_bytes 3: packed RGB; 4: RGB + one pad
 * byte, Pillow's own storage of an RGB image) -> dst_chw [3][Ho][Wo] fp32 in [0, 1] and / or
 * dst_u8 [Ho][Wo][3]; the tables of an axis are passed (device memory) exactly when its size changes; tmp
 * [H][Wo][3] uint8 holds the horizontal pass (needed when W != Wo). */
int vamd_resize_ksize(int in_size, int out_size);
int vamd_resize_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* coef, int ksize);
int vamd_image_resize_u8(const uint8_t* src, int H, int W, int pixel_bytes, const int32_t* hbounds, const int32_t* hcoef, int hk,
                         const int32_t* vbounds, const int32_t* vcoef, int vk, uint8_t* tmp, int Ho, int Wo,
                         float* dst_chw, uint8_t* dst_u8, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VISION_AMD_H */
