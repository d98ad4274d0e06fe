// Hardware-semantics probes used by tests/probe_gpu.py (diagnostics only, not on the product path).
#include "common.h"

namespace vamd {
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// LDS image: 16 rows x 64 shorts, element (r, c) = r*256 + c.  Lane l reads with
// ds_read_b64_tr_b16 at row (l&15)>>2, column 16*(l>>4) + 4*(l&3): out[l*4+e] = element e it got.
__global__ void probe_tr16_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[16 * 64];
    for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (short)((i / 64) * 256 + (i % 64));
    __syncthreads();
    const int l = threadIdx.x;
    const int row = (l & 15) >> 2, col = 16 * (l >> 4) + 4 * (l & 3);
    s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + row * 64 + col));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}

// D = A*B with A[i][k] = (i == k) (identity on the first 16 k) and B[k][j] = k*16 + j for k < 16:
// out[l*4+r] shows which (row, col) of D lane l register r holds.
__global__ void probe_mfma_kernel(float* out) {
    const int l = threadIdx.x;
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * (l >> 4) + e;           // assumed operand map: lane holds k = 8*(l>>4)+e
        a[e] = (__bf16)((k == (l & 15)) ? 1.0f : 0.0f);
        b[e] = (__bf16)((k < 16) ? (float)(k * 16 + (l & 15)) : 0.0f);
    }
    f32x4_t c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
}  // namespace vamd

extern "C" {
int vamd_probe_tr16(short* out, void* stream) {
    hipLaunchKernelGGL(vamd::probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
int vamd_probe_mfma(float* out, void* stream) {
    hipLaunchKernelGGL(vamd::probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
}
