// Attention core of reference models/ddpm.py:113-120 (AttnBlock.forward) in ONE kernel, bf16:
//
//     w = softmax(q k^T * C^-0.5) ; h = w v          (einsum 'bhwc,bHWc->bhwHW', softmax over HW, 'bhwHW,bHWc->bhwc')
//
// One workgroup (4 waves) per 16 queries of one image.  S = H*W <= 1024 keys, C <= 512 channels (one head).
//   1. scores: each wave takes key blocks of 16 (wave, wave + 4, ...): the 16 x C query fragments live in registers,
//      key fragments come straight from global memory in MFMA operand layout (a key row's 8 consecutive channels per
//      lane: every key element is used once per workgroup, so LDS staging would only add a copy); the 16 x 16 score
//      blocks, scaled, go to an fp32 LDS row buffer [16][S].
//   2. softmax: one wave per 4 rows with the arithmetic of the separate row-softmax kernel (lane-strided max and
//      sum, wave butterfly, exp(x - m) * (1 / sum)), probabilities rounded to bf16 into LDS [16][S].
//   3. P V: each wave takes channel tiles of 16 (wave, wave + 4, ...), four at a time; the value operand is the transposed value
//      tensor vT [B][C][ldv] (keys contiguous: again the MFMA operand layout), the probabilities come from LDS;
//      operands are swapped so that a lane ends up with 4 consecutive channels of one query (8-byte stores).
// Neither the S x S scores nor the probabilities touch HBM.  The backward pass needs the probabilities, so the
// engine uses this kernel for forwards that no backward follows (the sampler, validation).
#include "common.h"

namespace vamd {

typedef __attribute__((ext_vector_type(8))) __bf16 att_bf16x8;
typedef __attribute__((ext_vector_type(4))) float att_f32x4;

constexpr int kAttQ = 16;              // queries per workgroup
constexpr int kAttMaxFrags = 16;       // C / 32 <= 16

static int att_sk(int S) { return (S + 31) / 32 * 32; }
static size_t att_lds_bytes(int S) { return (size_t)kAttQ * (att_sk(S) + 4) * 4 + (size_t)kAttQ * (att_sk(S) + 8) * 2; }

__global__ void __launch_bounds__(256)
attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ vT,
                bf16_t* __restrict__ o, int S, int C, int ldv, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Sk = (S + 31) / 32 * 32;
    const int SLD = Sk + 4, PLD = Sk + 8;
    float* sc = reinterpret_cast<float*>(smem);                               // [16][SLD]
    bf16_t* pr = reinterpret_cast<bf16_t*>(smem + (size_t)kAttQ * SLD * 4);   // [16][PLD]
    const int b = blockIdx.y, q0 = blockIdx.x * kAttQ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int nfrag = C / 32;
    const att_bf16x8 zero8 = {};

    // ---- 1. scores
    att_bf16x8 qf[kAttMaxFrags];
    {
        const bool ok = q0 + r < S;
        const bf16_t* qp = q + ((int64_t)b * S + (ok ? q0 + r : 0)) * C + kg * 8;
#pragma unroll
        for (int i = 0; i < kAttMaxFrags; ++i)
            qf[i] = (ok && i < nfrag) ? *reinterpret_cast<const att_bf16x8*>(qp + i * 32) : zero8;
    }
    // four key blocks per pass and wave: their loads are in flight together, four independent accumulator chains
    const int nkb = (S + 15) / 16;
    for (int kb0 = wave; kb0 < nkb; kb0 += 16) {
        const bf16_t* kp[4];
        bool ok[4];
        att_f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int key = (kb0 + 4 * t) * 16 + r;
            ok[t] = kb0 + 4 * t < nkb && key < S;
            kp[t] = k + ((int64_t)b * S + (ok[t] ? key : 0)) * C + kg * 8;
            acc[t] = att_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < kAttMaxFrags; ++i) {
            if (i < nfrag) {
                att_bf16x8 kf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) kf[t] = ok[t] ? *reinterpret_cast<const att_bf16x8*>(kp[t] + i * 32) : zero8;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i], kf[t], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (kb0 + 4 * t >= nkb) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) sc[(kg * 4 + j) * SLD + (kb0 + 4 * t) * 16 + r] = acc[t][j] * scale;    // D[query][key]
        }
    }
    __syncthreads();

    // ---- 2. softmax of rows wave*4 .. wave*4+3 (rows past S hold zeros from zero query fragments: harmless)
    for (int row = wave * 4; row < wave * 4 + 4; ++row) {
        const float* xr = sc + row * SLD;
        float m = -INFINITY;
        for (int c = lane; c < S; c += 64) m = fmaxf(m, xr[c]);
        m = wave_max(m);
        float sum = 0.f;
        for (int c = lane; c < S; c += 64) sum += expf(xr[c] - m);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int c = lane; c < Sk; c += 64) pr[row * PLD + c] = c < S ? (bf16_t)(expf(xr[c] - m) * inv) : (bf16_t)0.f;
    }
    __syncthreads();

    // ---- 3. h = P V, operands swapped: D[channel][query]
    // four channel tiles per pass: one probability fragment feeds four independent accumulator chains
    const int ntile = C / 16;
    for (int ct0 = wave; ct0 < ntile; ct0 += 16) {
        const bf16_t* vp[4];
        bool live[4];
        att_f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            live[t] = ct0 + 4 * t < ntile;
            vp[t] = vT + ((int64_t)b * C + (live[t] ? ct0 + 4 * t : 0) * 16 + r) * ldv + kg * 8;
            acc[t] = att_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        auto load_v = [&](int k0, att_bf16x8 (&vf)[4]) {
            const int key = k0 + kg * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                vf[t] = zero8;
                if (live[t]) {
                    if (key + 8 <= S) {
                        vf[t] = *reinterpret_cast<const att_bf16x8*>(vp[t] + k0);
                    } else if (key < S) {                 // the last, partial group of keys: element by element
#pragma unroll
                        for (int e = 0; e < 8; ++e) vf[t][e] = key + e < S ? vp[t][k0 + e] : (bf16_t)0.f;
                    }
                }
            }
        };
        att_bf16x8 cur[4], nxt[4];
        load_v(0, cur);
        for (int k0 = 0; k0 < Sk; k0 += 32) {             // the next step's value fragments are in flight during this step's MFMAs
            if (k0 + 32 < Sk) load_v(k0 + 32, nxt);
            const att_bf16x8 pf = *reinterpret_cast<const att_bf16x8*>(pr + r * PLD + k0 + kg * 8);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur[t], pf, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
        }
        if (q0 + r < S) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!live[t]) continue;
                bf16_t out[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) out[j] = (bf16_t)acc[t][j];
                *reinterpret_cast<uint2*>(o + ((int64_t)b * S + q0 + r) * C + (ct0 + 4 * t) * 16 + kg * 4) =
                    *reinterpret_cast<const uint2*>(out);
            }
        }
    }
}

}  // namespace vamd

using namespace vamd;

extern "C" int vamd_attention_fwd_supported(int S, int C, int dtype) {
    return dtype == VAMD_BF16 && S >= 1 && S <= 1024 && C >= 32 && C <= 32 * kAttMaxFrags && C % 32 == 0 ? 1 : 0;
}

extern "C" int vamd_attention_fwd(const void* q, const void* k, const void* vT, void* o, int B, int S, int C, int ldv,
                                  float scale, int dtype, void* stream) {
    VAMD_REQUIRE(q && k && vT && o && B > 0 && B <= 65535, "vamd_attention_fwd: bad argument");
    VAMD_REQUIRE(vamd_attention_fwd_supported(S, C, dtype), "vamd_attention_fwd: S=%d C=%d dtype=%d not covered (bf16, S <= 1024, "
                 "C a multiple of 32 up to 512)", S, C, dtype);
    VAMD_REQUIRE(ldv >= S && ldv % 8 == 0, "vamd_attention_fwd: rows of the transposed values must be 16-byte aligned (ldv=%d)", ldv);
    VAMD_REQUIRE((uintptr_t)q % 16 == 0 && (uintptr_t)k % 16 == 0 && (uintptr_t)vT % 16 == 0 && (uintptr_t)o % 16 == 0,
                 "vamd_attention_fwd: tensors must be 16-byte aligned");
    const size_t lds = att_lds_bytes(S);
    static size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)att_lds_bytes(1024));
        if (e != hipSuccess) return fail((int)e, "vamd_attention_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        configured = att_lds_bytes(1024);
    }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((S + kAttQ - 1) / kAttQ, B), dim3(256), lds, as_stream(stream), (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)vT, (bf16_t*)o, S, C, ldv, scale);
    VAMD_LAUNCH_OK("attn_fwd");
    return 0;
}
