// Convolution geometry shared by the direct and MFMA kernels.
#pragma once
#include "common.h"

namespace vamd {

struct ConvGeom {
    int B, H, W, Cin, Cout;   // H, W: stored input extents
    int k, stride, pad, up;
    int He, We;               // effective input extents (after the optional nearest 2x upsample)
    int Ho, Wo;               // output extents
    int64_t M;                // B * Ho * Wo output pixels
    void* scratch;            // the descriptor's optional scratch buffer
    size_t scratch_bytes;
};

// Validates a vamd_conv_desc and fills `g`.  Returns 0 or a negative status (message set).
int conv_geom(const vamd_conv_desc* d, ConvGeom* g, const char* who);

// 1: layers with fewer tiles than CUs and a long reduction split it between workgroups (needs the descriptor's scratch
// buffer); 0: never.  Returns the previous setting; any other value only queries.
int conv_splitk_mode(int mode);

// direct kernels (conv_direct.hip)
int conv_fwd_direct(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                    const float* temb, const void* res, const void* res2, void* y, hipStream_t st);
int conv_dgrad_direct(const ConvGeom& g, int dtype, const void* dy, const void* wt, const void* add,
                      void* dx, hipStream_t st);
int conv_wgrad_direct(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw,
                      hipStream_t st);

// thin-channel kernels (conv_thin.hip): layers with <= 4 input or output channels, stride 1
bool conv_thin_supported(const ConvGeom& g, int dtype);
int conv_fwd_thin(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                  const float* temb, const void* res, const void* res2, void* y, hipStream_t st);
int conv_dgrad_thin(const ConvGeom& g, int dtype, const void* dy, const void* wt, const void* add, void* dx,
                    hipStream_t st);
bool conv_wgrad_thin_supported(const ConvGeom& g, int dtype);
size_t conv_wgrad_thin_workspace(const ConvGeom& g, int dtype);
int conv_wgrad_thin(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw, void* workspace,
                    size_t workspace_bytes, hipStream_t st);

// bf16 thin-channel kernels on the matrix cores (conv_thin_mfma.hip); conv_thin.hip dispatches to them
bool thin_in_mfma_ok(const ConvGeom& g);
int thin_in_mfma(const ConvGeom& g, const void* x, const void* w, const float* bias, const float* temb,
                 const void* res, const void* res2, void* y, hipStream_t st);
bool thin_out_mfma_ok(const ConvGeom& g);
int thin_out_mfma(const ConvGeom& g, const void* x, const void* w, const float* bias, const float* temb,
                  const void* res, const void* res2, void* y, hipStream_t st);
bool thin_wgrad_mfma_ok(const ConvGeom& g);
size_t thin_wgrad_mfma_workspace(const ConvGeom& g);
int thin_wgrad_mfma(const ConvGeom& g, const void* dy, const void* x, float* dw, void* workspace, hipStream_t st);

// MFMA implicit-GEMM kernels (conv_mfma.hip).  *_supported says whether the shape is covered.
bool conv_fwd_mfma_supported(const ConvGeom& g, int dtype);
bool conv_fwd_act_mfma_supported(const ConvGeom& g, int dtype);
int conv_fwd_act_mfma(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias, const float* temb,
                      const void* res, const void* res2, const float* scale, const float* shift, float slope, void* y,
                      void* raw, hipStream_t st);
int conv_fwd_mfma(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                  const float* temb, const void* res, const void* res2, void* y, float* stats, hipStream_t st);
int conv_fwd_mfma_stats_parts(const ConvGeom& g, int dtype);      // row tiles = leading extent of `stats`
bool conv_dgrad_mfma_supported(const ConvGeom& g, int dtype);
size_t conv_dgrad_mfma_workspace(const ConvGeom& g, int dtype);
int conv_dgrad_mfma(const ConvGeom& g, int dtype, const void* dy, const void* wt, const void* add,
                    void* dx, void* workspace, size_t workspace_bytes, hipStream_t st);
bool conv_fwd_short_mfma_supported(const ConvGeom& g, int dtype, int Cin2);
int conv_fwd_short_mfma(const ConvGeom& g, int dtype, const void* x, const void* w, const float* bias,
                        const float* temb, const void* x2, const void* w2, int Cin2, const void* res2, void* y,
                        float* stats, hipStream_t st);
bool conv_fwd_up4_mfma_supported(const ConvGeom& g, int dtype);
int conv_fwd_up4_mfma_stats_parts(const ConvGeom& g, int dtype);
int conv_fwd_up4_mfma(const ConvGeom& g, int dtype, const void* x, const void* w4, const float* bias,
                      const float* temb, const void* res, const void* res2, void* y, float* stats, hipStream_t st);
bool conv_dgrad_up4_mfma_supported(const ConvGeom& g, int dtype);
int conv_dgrad_up4_mfma(const ConvGeom& g, int dtype, const void* dy, const void* w4, const void* add, void* dx,
                        hipStream_t st);
bool conv_wgrad_up4_mfma_supported(const ConvGeom& g, int dtype);
size_t conv_wgrad_up4_mfma_workspace(const ConvGeom& g, int dtype);
int conv_wgrad_up4_mfma(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw, float* dbias,
                        int oihw, void* workspace, size_t workspace_bytes, hipStream_t st);
bool conv_wgrad_mfma_supported(const ConvGeom& g, int dtype);
size_t conv_wgrad_mfma_workspace(const ConvGeom& g, int dtype);
int conv_wgrad_mfma(const ConvGeom& g, int dtype, const void* dy, const void* x, float* dw, float* dbias,
                    int oihw, void* workspace, size_t workspace_bytes, hipStream_t st, const float* pre_scale = nullptr,
                    const float* pre_shift = nullptr, float pre_slope = 0.f, float* const* dw3 = nullptr,
                    float* const* dbias3 = nullptr);
// input = BatchNorm + LeakyReLU of a stored tensor, applied inside the forward and weight-gradient kernels
bool conv_prenorm_mfma_supported(const ConvGeom& g, int dtype);
int conv_fwd_prenorm_mfma(const ConvGeom& g, int dtype, const void* x, const float* pre_scale, const float* pre_shift,
                          float pre_slope, const void* w, const float* bias, const float* temb, const void* res,
                          const void* res2, void* y, float* stats, hipStream_t st);

// batched products for the attention core on the MFMA kernels (rows may be padded: ld = row stride)
bool bmm_nt_mfma_supported(int M, int N, int K, int lda, int ldb, int ldc, int dtype, int out_f32);
int bmm_nt_mfma(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda, int ldb,
                int ldc, float alpha, int dtype, int out_f32, hipStream_t st);
bool bmm_tn_mfma_supported(int M, int N, int K, int lda, int ldb, int ldc, int dtype);
int bmm_tn_mfma(const void* A, const void* Bm, void* C, int batch, int M, int N, int K, int lda, int ldb,
                int ldc, float alpha, int dtype, hipStream_t st);

}  // namespace vamd
