/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the codebook lookup of reference
 * models/vqvaeunet.py:115-146 (VQBlock.forward).  Never linked into or called by the product library.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared oracle/vq_oracle.c -o oracle/_build/libvq_oracle.so -lm
 *
 * z: [M, D] flat latent rows (vqvaeunet.py:116-118 after the permute), e: [K, D] (vqvaeunet.py:112).
 * The distance of vqvaeunet.py:121-123 is evaluated in fp32 in one stated order -- each sum over d ascending with
 * one fused multiply-add per term, (|z|^2 + |e|^2) first, then minus twice the dot product -- and the first
 * minimum is taken (torch.argmin, vqvaeunet.py:126).  torch's own CPU matmul sums in a blocked order, so on
 * inputs whose best two distances differ by less than fp32 rounding the two may pick different codes; on
 * inputs that are exact in fp32 (small dyadic values) or separated by more than the rounding they agree.
 */
#include <math.h>
#include <stdint.h>

void vq_oracle_forward(const float* z, const float* e, int64_t M, int D, int K, float beta,
                       int64_t* indices, float* out, float* dist_best, float* loss2) {
    double sq = 0.0;
    for (int64_t m = 0; m < M; ++m) {
        const float* zr = z + m * D;
        float zz = 0.f;
        for (int d = 0; d < D; ++d) zz = fmaf(zr[d], zr[d], zz);
        int64_t best = 0;
        float best_d = 0.f;
        int have = 0, best_nan = 0;
        for (int k = 0; k < K; ++k) {
            const float* er = e + (int64_t)k * D;
            float ee = 0.f, dot = 0.f;
            for (int d = 0; d < D; ++d) {
                ee = fmaf(er[d], er[d], ee);
                dot = fmaf(zr[d], er[d], dot);
            }
            const float t = zz + ee;
            const float dist = t - 2.0f * dot;
            if (best_nan) continue;                       /* the first NaN stays the answer */
            if (dist != dist) { best = k; best_d = dist; best_nan = 1; have = 1; continue; }
            if (!have || dist < best_d) { best = k; best_d = dist; have = 1; }
        }
        indices[m] = best;
        if (dist_best) dist_best[m] = best_d;
        const float* q = e + best * D;
        for (int d = 0; d < D; ++d) {
            const float diff = q[d] - zr[d];
            out[m * D + d] = zr[d] + diff;                /* vqvaeunet.py:144 */
            sq += (double)diff * (double)diff;
        }
    }
    const float mse = (float)(sq / ((double)M * (double)D));
    loss2[0] = mse * beta + mse;                          /* vqvaeunet.py:138-141 */
    loss2[1] = mse;
}
