/*
 * rnnt_oracle.c — CPU restatement of the jakepoz/rnnt joint + transducer-loss hot path.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker / the CPU
 * number reported beside the GPU one.  Nothing under rnnt_amd/ imports or links it.
 *
 * What it restates (reference paths relative to /root/reference):
 *   rnnt/joint.py:32-39   broadcast add -> tanh -> Linear(H->V)          (in-repo Python)
 *   rnnt/model.py:35-41   torchaudio.functional.rnnt_loss(..., blank=-1, clamp=-1,
 *                         reduction="mean")                               (THIRD PARTY)
 *   rnnt/train.py:133-134 loss.backward(): autograd of the two above
 *
 * Third-party provenance: the loss arithmetic lives in `torchaudio`, which is NOT under
 * /root/reference, is not installed in this image and is unpinned by the reference
 * (requirements.txt lists neither torch nor torchaudio).  The recurrences and the fused
 * log-softmax gradient are restated from the published algorithm (Graves 2012,
 * "Sequence Transduction with Recurrent Neural Networks") in the form SURVEY.md §8c
 * records.  The reference's own tests hold NO golden vectors for this path,
 * so the LOSS half is pinned by (a) brute-force alignment enumeration
 * (oracle/brute_force.py), (b) torch autograd through an independent log-space alpha
 * recursion (oracle/torch_check.py) and (c) the PUBLISHED known-answer vectors of the
 * third-party algorithm — the warp-transducer unit-test cases torchaudio's own tests reuse
 * to pin rnnt_loss (tests/golden/published_transducer_kat.json: costs and gradients,
 * reproduced to 3e-7; third-party published data, transcribed, not reference-held).
 * torchaudio itself cannot be run here: "parity unpinned" against torchaudio itself stands.
 * The JOINT half IS pinned: the .npz files under tests/golden hold logits and autograd gradients produced
 * by importing the reference's own rnnt.joint.JointNetwork (tests/golden/make_golden.py).
 *
 * Two instantiations: _f64 (ground truth) and _f32 (the reference's arithmetic type; also
 * the cpu_baseline "port").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUFFIX f64
#include "rnnt_oracle_body.inc"
#undef REAL
#undef SUFFIX

#define REAL float
#define SUFFIX f32
#include "rnnt_oracle_body.inc"
#undef REAL
#undef SUFFIX

/* cpu_baseline leg of bench.py ONLY: the fp32 loss above runs one utterance per
 * thread with a double-precision exp per logit, which leaves most host cores idle on the bounded sample
 * (B <= 8 utterances).  Same arithmetic order per (t,u) row, same formulas, but the
 * two O(T*U1*V) loops — log-softmax denominators and the gradient rows — are spread over (b,t,u) rows of the
 * whole batch with OpenMP and use single-precision expf/logf, as an fp32 CPU implementation would; the
 * O(T*U1) alpha/beta recurrences stay serial per utterance (one utterance per thread).  The _f64 / _f32
 * instantiations above are untouched: they remain the ground truth the tests check against, and
 * tests/test_oracle.py checks this variant against them. */
void rnnt_oracle_loss_par_f32(const float *logits, const int32_t *targets, const int32_t *logit_lens,
                              const int32_t *target_lens, int B, int T, int U1, int V, int blank,
                              float *costs, float *grad)
{
    if (blank < 0) blank += V;
    const int Umax = U1 - 1;
    const long n = (long)T * U1, rows = (long)B * n;
    float *denom = (float *)malloc(sizeof(float) * (size_t)rows);
    float *alpha = (float *)malloc(sizeof(float) * (size_t)rows);
    float *beta = (float *)malloc(sizeof(float) * (size_t)rows);
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const int b = (int)(r / n), t = (int)((r % n) / U1), u = (int)(r % U1);
        alpha[r] = -INFINITY; beta[r] = -INFINITY; denom[r] = 0;
        if (t >= logit_lens[b] || u > target_lens[b]) continue;
        const float *x = logits + r * V;
        float m = -INFINITY;
        for (int v = 0; v < V; ++v) if (x[v] > m) m = x[v];
        float sum = 0;
        for (int v = 0; v < V; ++v) sum += expf(x[v] - m);
        denom[r] = m + logf(sum);
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const int Tb = logit_lens[b], Ub = target_lens[b];
        const int32_t *y = targets + (long)b * Umax;
        const float *lg = logits + (long)b * n * V;
        float *al = alpha + (long)b * n, *be = beta + (long)b * n, *de = denom + (long)b * n;
#define PLG(t, u, v) lg[(((long)(t)) * U1 + (u)) * V + (v)]
#define PIX(t, u) ((long)(t) * U1 + (u))
#define PLPB(t, u) (PLG(t, u, blank) - de[PIX(t, u)])
#define PLPE(t, u) (PLG(t, u, y[u]) - de[PIX(t, u)])
        for (int t = 0; t < Tb; ++t)
            for (int u = 0; u <= Ub; ++u) {
                if (t == 0 && u == 0) { al[0] = 0; continue; }
                float a = -INFINITY, e = -INFINITY;
                if (t > 0) a = al[PIX(t - 1, u)] + PLPB(t - 1, u);
                if (u > 0) e = al[PIX(t, u - 1)] + PLPE(t, u - 1);
                al[PIX(t, u)] = logaddexp_f32(a, e);
            }
        for (int t = Tb - 1; t >= 0; --t)
            for (int u = Ub; u >= 0; --u) {
                if (t == Tb - 1 && u == Ub) { be[PIX(t, u)] = PLPB(t, u); continue; }
                float a = -INFINITY, e = -INFINITY;
                if (t < Tb - 1) a = be[PIX(t + 1, u)] + PLPB(t, u);
                if (u < Ub) e = be[PIX(t, u + 1)] + PLPE(t, u);
                be[PIX(t, u)] = logaddexp_f32(a, e);
            }
        costs[b] = -be[0];
    }
    if (grad) {
#pragma omp parallel for schedule(static)
        for (long r = 0; r < rows; ++r) {
            const int b = (int)(r / n), t = (int)((r % n) / U1), u = (int)(r % U1);
            float *g = grad + r * V;
            const int Tb = logit_lens[b], Ub = target_lens[b];
            if (t >= Tb || u > Ub) { memset(g, 0, sizeof(float) * (size_t)V); continue; }
            const float *x = logits + r * V;
            const float *be = beta + (long)b * n;
            const float c = alpha[r] + costs[b] - denom[r];
            const float bt = be[PIX(t, u)];
            for (int v = 0; v < V; ++v) g[v] = expf(x[v] + c + bt);
            {   /* the two corrected entries of the row */
                const float gb = x[blank] + c;
                if (t == Tb - 1 && u == Ub) g[blank] -= expf(gb);
                else if (t < Tb - 1) g[blank] -= expf(gb + be[PIX(t + 1, u)]);
                if (u < Ub) {
                    const int yy = targets[(long)b * Umax + u];
                    g[yy] -= expf(x[yy] + c + be[PIX(t, u + 1)]);
                }
            }
        }
    }
#undef PLG
#undef PIX
#undef PLPB
#undef PLPE
    free(denom); free(alpha); free(beta);
}

int rnnt_oracle_version(void) { return 2; }
