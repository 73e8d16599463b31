// predictor.hip — ConvPredictor forward / backward for gfx950 and the
// joint's input projections (§8f rank 1), behind the C ABI.
//
// Reference rnnt/predictor.py:189-229 (with rnnt/causalconv.py:9-32):
//   x0 = embedding(ids)                       [B,U1,E]
//   x1 = LayerNorm_E(x0)
//   g1 = dropout(gelu(causal_conv_k3(x1)))    (N,C,L) convs: here rows = (b,u), channels contiguous
//   g2 = dropout(gelu(causal_conv_k5(g1)))
//   y  = LayerNorm_O(linear(g2))              [B,U1,O]
// A causal convolution over (b,u) rows with channels contiguous is a GEMM per tap with a row shift
// (smallgemm.hpp): no permute to (N,C,L) and back (predictor.py:216,225) is ever materialised.
// Dropout masks come from the caller (torch's generator) as keep bytes; NULL = eval mode.
// Every intermediate the backward needs is kept in a caller-owned `saved` buffer.
#include "../../include/rnnt_engine.h"
#include "kernels.hpp"
#include "smallgemm.hpp"

namespace {

inline size_t al(size_t x) { return (x + 63) & ~(size_t)63; }  // in floats

struct PredLayout {
    size_t wp1, wp2, x1, st1, y1, g1, y2, g2, z, st2;      // forward (kept for backward)
    size_t dz, t, dg, dyp, dwp, slabs, cs;                 // backward scratch
    size_t total;
};
PredLayout pred_layout(int B, int U1, int E, int O)
{
    const size_t M = (size_t)B * U1;
    PredLayout L;
    size_t o = 0;
    L.wp1 = o; o += al(3 * (size_t)E * E);
    L.wp2 = o; o += al(5 * (size_t)E * E);
    L.x1 = o;  o += al(M * E);
    L.st1 = o; o += al(2 * M);
    L.y1 = o;  o += al(M * E);
    L.g1 = o;  o += al(M * E);
    L.y2 = o;  o += al(M * E);
    L.g2 = o;  o += al(M * E);
    L.z = o;   o += al(M * O);
    L.st2 = o; o += al(2 * M);
    const size_t W = E > O ? E : O;
    L.dz = o;  o += al(M * W);
    L.t = o;   o += al(M * W);
    L.dg = o;  o += al(M * E);
    L.dyp = o; o += al(M * E);
    L.dwp = o; o += al(5 * (size_t)E * E);
    L.slabs = o; o += al((size_t)sgemm_tn_splits((int)M) * (5 * (size_t)E * E > (size_t)O * E ? 5 * (size_t)E * E : (size_t)O * E));
    L.cs = o;  o += al(colsum_scratch_floats((int)M, (int)W));
    L.total = o;
    return L;
}

// ---- LayerNorm over the last dimension, one wave per row; EMBED: the row is embedding[ids[row]]
template <bool EMBED>
__global__ __launch_bounds__(256) void k_ln_fwd(const float *__restrict__ X, const int64_t *__restrict__ ids, int S,
                                                const float *__restrict__ gamma, const float *__restrict__ beta,
                                                float eps, int M, int N, float *__restrict__ Y,
                                                float *__restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    long src = row;
    if (EMBED) { long id = ids[row]; src = id < 0 ? 0 : (id >= S ? S - 1 : id); }
    const float *x = X + src * N;
    float s = 0.f;
    for (int n = lane * 4; n < N; n += 256) { const f32x4 v = *(const f32x4 *)(x + n); s += (v[0] + v[1]) + (v[2] + v[3]); }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k, 64);
    const float mean = s / N;
    float q = 0.f;
    for (int n = lane * 4; n < N; n += 256) {
        const f32x4 v = *(const f32x4 *)(x + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) q += (v[e] - mean) * (v[e] - mean);
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) q += __shfl_xor(q, k, 64);
    const float rstd = rsqrtf(q / N + eps);
    for (int n = lane * 4; n < N; n += 256) {
        const f32x4 v = *(const f32x4 *)(x + n), g = *(const f32x4 *)(gamma + n), b = *(const f32x4 *)(beta + n);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * g[e] + b[e];
        *(f32x4 *)(Y + (long)row * N + n) = o;
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; T = dy * xhat (for dgamma)
template <bool EMBED>
__global__ __launch_bounds__(256) void k_ln_bwd(const float *__restrict__ X, const int64_t *__restrict__ ids, int S,
                                                const float *__restrict__ gamma, const float *__restrict__ stats,
                                                const float *__restrict__ dY, int M, int N,
                                                float *__restrict__ dX, float *__restrict__ T)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    long src = row;
    if (EMBED) { long id = ids[row]; src = id < 0 ? 0 : (id >= S ? S - 1 : id); }
    const float *x = X + src * N, *dy = dY + (long)row * N;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float s1 = 0.f, s2 = 0.f;
    for (int n = lane * 4; n < N; n += 256) {
        const f32x4 v = *(const f32x4 *)(x + n), g = *(const f32x4 *)(gamma + n), d = *(const f32x4 *)(dy + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (v[e] - mean) * rstd, gg = d[e] * g[e];
            s1 += gg;
            s2 += gg * xh;
        }
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) { s1 += __shfl_xor(s1, k, 64); s2 += __shfl_xor(s2, k, 64); }
    const float c1 = s1 / N, c2 = s2 / N;
    for (int n = lane * 4; n < N; n += 256) {
        const f32x4 v = *(const f32x4 *)(x + n), g = *(const f32x4 *)(gamma + n), d = *(const f32x4 *)(dy + n);
        f32x4 o, t;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (v[e] - mean) * rstd;
            o[e] = rstd * (d[e] * g[e] - c1 - xh * c2);
            t[e] = d[e] * xh;
        }
        *(f32x4 *)(dX + (long)row * N + n) = o;
        *(f32x4 *)(T + (long)row * N + n) = t;
    }
}

// dYpre = dG * keep * scale * gelu'(Ypre)
__global__ __launch_bounds__(256) void k_gelu_bwd(const float *__restrict__ dG, const float *__restrict__ Ypre,
                                                  const unsigned char *__restrict__ mask, float scale, long n,
                                                  float *__restrict__ out)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float x = Ypre[idx];
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    float d = dG[idx] * (cdf + x * pdf);
    if (mask) d = mask[idx] ? d * scale : 0.f;
    out[idx] = d;
}

// dEmb[s,:] = sum over the rows m (ascending) with ids[m] == s of dX0[m,:]: deterministic, no atomics
// One workgroup per symbol: pass 1 lists the rows that carry it, in ascending order (256 rows per
// step, ballot + prefix counts); pass 2 adds them up in that order.
__global__ __launch_bounds__(256) void k_embed_bwd(const int64_t *__restrict__ ids, const float *__restrict__ dX0,
                                                   int M, int E, int S, float *__restrict__ dEmb)
{
    constexpr int CAP = 2048;  // rows listed per round
    __shared__ int s_list[CAP];
    __shared__ int s_cnt[5];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // E <= 2048 in two column sweeps of 1024
    const int sweeps = (E + 1023) / 1024;
    for (int base = 0; base < M;) {
        int n = 0;  // rows listed so far this round (block-uniform)
        int m0 = base;
        for (; m0 < M && n + 256 <= CAP; m0 += 256) {
            const int m = m0 + tid;
            bool hit = false;
            if (m < M) {
                long id = ids[m];
                id = id < 0 ? 0 : (id >= S ? S - 1 : id);
                hit = id == s;
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) s_cnt[wave] = __popcll(bal);
            __syncthreads();
            int off = n;
            for (int w = 0; w < wave; ++w) off += s_cnt[w];
            if (hit) s_list[off + __popcll(bal & ((1ull << lane) - 1ull))] = m;
            n += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            __syncthreads();
        }
        base = m0;
        for (int k = 0; k < n; ++k) {
            const float *row = dX0 + (long)s_list[k] * E;
            for (int sw = 0; sw < sweeps && sw < 2; ++sw) {
                const int e0 = sw * 1024 + tid * 4;
                if (e0 < E) acc[sw] += *(const f32x4 *)(row + e0);
            }
        }
        __syncthreads();
    }
    for (int sw = 0; sw < sweeps && sw < 2; ++sw) {
        const int e0 = sw * 1024 + tid * 4;
        if (e0 < E) *(f32x4 *)(dEmb + (long)s * E + e0) = acc[sw];
    }
}

int check_pred_dims(int B, int U1, int S, int E, int O)
{
    if (B <= 0 || U1 <= 0 || S <= 0 || E <= 0 || O <= 0)
        return engine_fail(RNNT_ERR_INVALID_ARG, "non-positive dimension B=%d U1=%d S=%d E=%d O=%d", B, U1, S, E, O);
    if (E % 4 != 0 || O % 4 != 0 || E > 2048)
        return engine_fail(RNNT_ERR_UNSUPPORTED, "ConvPredictor kernels need E %% 4 == 0, E <= 2048 and O %% 4 == 0 (E=%d O=%d)", E, O);
    if ((long)B * U1 > 0x3fffffffL) return engine_fail(RNNT_ERR_UNSUPPORTED, "B*U1 too large");
    return RNNT_OK;
}

int status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return engine_fail(RNNT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return RNNT_OK;
}

SgArgs sg(const float *A, long lda, const float *B, long ldb, float *C, long ldc, int M, int N, int K, int taps, int seg)
{
    SgArgs a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.bias = nullptr; a.Cpre = nullptr; a.mask = nullptr; a.mask_scale = 1.f;
    a.M = M; a.N = N; a.K = K; a.taps = taps; a.seg = seg; a.act = 0; a.ksplit = 1;
    return a;
}

}  // namespace

extern "C" {

int rnnt_engine_conv_predictor_saved_bytes(int B, int U1, int S, int E, int O, size_t *out)
{
    if (!out) return engine_fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (int rc = check_pred_dims(B, U1, S, E, O)) return rc;
    *out = pred_layout(B, U1, E, O).total * 4;
    return RNNT_OK;
}

int rnnt_engine_conv_predictor_fwd(const int64_t *ids, int B, int U1, int S, int E, int O,
                                   const rnnt_conv_predictor_params *p, const uint8_t *keep1,
                                   const uint8_t *keep2, float dropout_p, float ln_in_eps, float ln_out_eps, float *out,
                                   void *saved, size_t saved_bytes, void *stream)
{
    if (int rc = check_pred_dims(B, U1, S, E, O)) return rc;
    if (!ids || !p || !out || !saved) return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const void *ptrs[] = {p->embedding, p->ln_in_w, p->ln_in_b, p->conv1_w, p->conv1_b, p->conv2_w, p->conv2_b,
                          p->linear_w, p->linear_b, p->ln_out_w, p->ln_out_b, out, saved};
    for (const void *q : ptrs)
        if (!q || ((uintptr_t)q & 15)) return engine_fail(RNNT_ERR_INVALID_ARG, "null or not 16-byte aligned parameter pointer");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return engine_fail(RNNT_ERR_INVALID_ARG, "dropout_p outside [0,1)");
    const PredLayout L = pred_layout(B, U1, E, O);
    if (saved_bytes < L.total * 4) return engine_fail(RNNT_ERR_WORKSPACE, "saved buffer %zu < required %zu bytes", saved_bytes, L.total * 4);
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)saved;
    const int M = B * U1;
    const float scale = 1.f / (1.f - dropout_p);

    launch_pack_conv_w(p->conv1_w, ws + L.wp1, E, E, 3, st);
    launch_pack_conv_w(p->conv2_w, ws + L.wp2, E, E, 5, st);
    // x1 = LN(embedding[ids])                                   predictor.py:214-215
    hipLaunchKernelGGL(k_ln_fwd<true>, dim3((M + 3) / 4), dim3(256), 0, st, p->embedding, ids, S, p->ln_in_w,
                       p->ln_in_b, ln_in_eps, M, E, ws + L.x1, ws + L.st1);
    // g1 = dropout(gelu(conv1(x1)))                              predictor.py:217-220
    SgArgs c1 = sg(ws + L.x1, E, ws + L.wp1, E, ws + L.g1, E, M, E, E, 3, U1);
    c1.bias = p->conv1_b; c1.Cpre = ws + L.y1; c1.act = 1; c1.mask = keep1; c1.mask_scale = scale;
    launch_sgemm_nt(c1, st);
    // g2 = dropout(gelu(conv2(g1)))                              predictor.py:222-224
    SgArgs c2 = sg(ws + L.g1, E, ws + L.wp2, E, ws + L.g2, E, M, E, E, 5, U1);
    c2.bias = p->conv2_b; c2.Cpre = ws + L.y2; c2.act = 1; c2.mask = keep2; c2.mask_scale = scale;
    launch_sgemm_nt(c2, st);
    // z = linear(g2); y = LN(z)                                  predictor.py:228-229
    SgArgs l = sg(ws + L.g2, E, p->linear_w, E, ws + L.z, O, M, O, E, 1, M);
    l.bias = p->linear_b;
    launch_sgemm_nt(l, st);
    hipLaunchKernelGGL(k_ln_fwd<false>, dim3((M + 3) / 4), dim3(256), 0, st, ws + L.z, nullptr, 0, p->ln_out_w,
                       p->ln_out_b, ln_out_eps, M, O, out, ws + L.st2);
    return status("rnnt_engine_conv_predictor_fwd");
}

int rnnt_engine_conv_predictor_bwd(const int64_t *ids, int B, int U1, int S, int E, int O,
                                   const rnnt_conv_predictor_params *p, const uint8_t *keep1,
                                   const uint8_t *keep2, float dropout_p, const float *grad_out,
                                   const rnnt_conv_predictor_params *g, void *saved, size_t saved_bytes,
                                   void *stream)
{
    if (int rc = check_pred_dims(B, U1, S, E, O)) return rc;
    if (!ids || !p || !g || !grad_out || !saved) return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const void *ptrs[] = {g->embedding, g->ln_in_w, g->ln_in_b, g->conv1_w, g->conv1_b, g->conv2_w, g->conv2_b,
                          g->linear_w, g->linear_b, g->ln_out_w, g->ln_out_b, p->embedding, p->ln_in_w,
                          p->linear_w, p->ln_out_w, grad_out, saved};
    for (const void *q : ptrs)
        if (!q || ((uintptr_t)q & 15)) return engine_fail(RNNT_ERR_INVALID_ARG, "null or not 16-byte aligned pointer");
    const PredLayout L = pred_layout(B, U1, E, O);
    if (saved_bytes < L.total * 4) return engine_fail(RNNT_ERR_WORKSPACE, "saved buffer %zu < required %zu bytes", saved_bytes, L.total * 4);
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)saved;
    const int M = B * U1;
    const float scale = 1.f / (1.f - dropout_p);
    float *gp[11] = {(float *)g->embedding, (float *)g->ln_in_w, (float *)g->ln_in_b, (float *)g->conv1_w,
                     (float *)g->conv1_b, (float *)g->conv2_w, (float *)g->conv2_b, (float *)g->linear_w,
                     (float *)g->linear_b, (float *)g->ln_out_w, (float *)g->ln_out_b};
    float *cs = ws + L.cs;

    // output LayerNorm: dz, d gamma = colsum(dy * zhat), d beta = colsum(dy)
    hipLaunchKernelGGL(k_ln_bwd<false>, dim3((M + 3) / 4), dim3(256), 0, st, ws + L.z, nullptr, 0, p->ln_out_w,
                       ws + L.st2, grad_out, M, O, ws + L.dz, ws + L.t);
    launch_colsum_pair(ws + L.t, grad_out, O, M, O, gp[9], gp[10], cs, st);
    // linear: dW = dz^T g2, db = colsum(dz), dg2 = dz W
    // weight gradient: contraction split over KS workgroups; the slabs, the conv weights' layout and the bias gradient's
    // second stage are finished by one launch (everything in fixed order)
    auto wgrad = [&](SgArgs a, const float *dy, int out_c, int in_c, float *dW, float *db) {
        const int KS = sgemm_tn_splits_for(M, a.N, a.K, a.taps);
        a.ksplit = KS; a.C = ws + L.slabs;
        launch_sgemm_tn(a, st);
        launch_colsum_stage1(dy, out_c, M, out_c, cs, st);
        launch_wgrad_finish(ws + L.slabs, KS, dW, out_c, in_c, a.taps, cs, colsum_slabs(M), out_c, db, st);
    };
    wgrad(sg(ws + L.dz, O, ws + L.g2, E, nullptr, E, M, O, E, 1, M), ws + L.dz, O, E, gp[7], gp[8]);
    launch_sgemm_nn(sg(ws + L.dz, O, p->linear_w, E, ws + L.dg, E, M, E, O, 1, M), st);
    // conv2: through dropout + gelu, then dW2 / db2 / dg1
    hipLaunchKernelGGL(k_gelu_bwd, dim3((unsigned)(((long)M * E + 255) / 256)), dim3(256), 0, st, ws + L.dg,
                       ws + L.y2, keep2, scale, (long)M * E, ws + L.dyp);
    wgrad(sg(ws + L.dyp, E, ws + L.g1, E, nullptr, E, M, E, E, 5, U1), ws + L.dyp, E, E, gp[5], gp[6]);
    launch_sgemm_nn(sg(ws + L.dyp, E, ws + L.wp2, E, ws + L.dg, E, M, E, E, 5, U1), st);
    // conv1
    hipLaunchKernelGGL(k_gelu_bwd, dim3((unsigned)(((long)M * E + 255) / 256)), dim3(256), 0, st, ws + L.dg,
                       ws + L.y1, keep1, scale, (long)M * E, ws + L.dyp);
    wgrad(sg(ws + L.dyp, E, ws + L.x1, E, nullptr, E, M, E, E, 3, U1), ws + L.dyp, E, E, gp[3], gp[4]);
    launch_sgemm_nn(sg(ws + L.dyp, E, ws + L.wp1, E, ws + L.dg, E, M, E, E, 3, U1), st);
    // input LayerNorm (its input is the embedding row) and the embedding table
    hipLaunchKernelGGL(k_ln_bwd<true>, dim3((M + 3) / 4), dim3(256), 0, st, p->embedding, ids, S, p->ln_in_w,
                       ws + L.st1, ws + L.dg, M, E, ws + L.dz, ws + L.t);
    launch_colsum_pair(ws + L.t, ws + L.dg, E, M, E, gp[1], gp[2], cs, st);
    hipLaunchKernelGGL(k_embed_bwd, dim3(S), dim3(256), 0, st, ids, ws + L.dz, M, E, S, gp[0]);
    return status("rnnt_engine_conv_predictor_bwd");
}

// ---- Linear layer y = x W^T + b and its backward (input projections audio_ln / text_ln of the
// joint, reference rnnt/joint.py:8-12,26-30).  x [M,K] rows ldx apart (k contiguous).
int rnnt_engine_linear_fwd(const float *x, int64_t ldx, const float *W, const float *bias, int M, int K, int N,
                           float *y, void *stream)
{
    if (M <= 0 || K <= 0 || N <= 0) return engine_fail(RNNT_ERR_INVALID_ARG, "non-positive dimension");
    if (K % 4 != 0 || N % 4 != 0 || ldx % 4 != 0) return engine_fail(RNNT_ERR_UNSUPPORTED, "linear kernels need K, N and the row stride to be multiples of 4 (K=%d N=%d)", K, N);
    if (!x || !W || !y || ((uintptr_t)x & 15) || ((uintptr_t)W & 15) || ((uintptr_t)y & 15))
        return engine_fail(RNNT_ERR_INVALID_ARG, "null or not 16-byte aligned pointer");
    SgArgs a = sg(x, ldx, W, K, y, N, M, N, K, 1, M);
    a.bias = bias;
    launch_sgemm_nt(a, (hipStream_t)stream);
    return status("rnnt_engine_linear_fwd");
}

int rnnt_engine_linear_bwd_workspace_bytes(int M, int K, int N, size_t *out)
{
    if (!out) return engine_fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (M <= 0 || K <= 0 || N <= 0) return engine_fail(RNNT_ERR_INVALID_ARG, "non-positive dimension");
    *out = (colsum_scratch_floats(M, N) + 64 + (size_t)sgemm_tn_splits(M) * N * K) * 4 + 256;
    return RNNT_OK;
}

int rnnt_engine_linear_bwd(const float *x, int64_t ldx, const float *W, const float *dy, int M, int K, int N,
                           float *dx /* [M,K] contiguous or NULL */, float *dW, float *db /* or NULL */,
                           void *workspace, size_t ws_bytes, void *stream)
{
    size_t need;
    if (int rc = rnnt_engine_linear_bwd_workspace_bytes(M, K, N, &need)) return rc;
    if (K % 4 != 0 || N % 4 != 0 || ldx % 4 != 0) return engine_fail(RNNT_ERR_UNSUPPORTED, "linear kernels need K, N and the row stride to be multiples of 4 (K=%d N=%d)", K, N);
    if (!x || !W || !dy || !dW || !workspace) return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const void *ptrs[] = {x, W, dy, dW, workspace};
    for (const void *q : ptrs)
        if ((uintptr_t)q & 15) return engine_fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned");
    if ((dx && ((uintptr_t)dx & 15)) || (db && ((uintptr_t)db & 15))) return engine_fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned");
    if (ws_bytes < need) return engine_fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    {
        float *slabs = (float *)workspace + ((colsum_scratch_floats(M, N) + 63) & ~(size_t)63);
        SgArgs a = sg(dy, N, x, ldx, slabs, K, M, N, K, 1, M);
        a.ksplit = sgemm_tn_splits_for(M, N, K, 1);
        launch_sgemm_tn(a, st);
        if (db) launch_colsum_stage1(dy, N, M, N, (float *)workspace, st);
        launch_wgrad_finish(slabs, a.ksplit, dW, N, K, 1, (const float *)workspace, colsum_slabs(M), N, db, st);
    }
    if (dx) launch_sgemm_nn(sg(dy, N, W, K, dx, K, M, K, N, 1, M), st);
    return status("rnnt_engine_linear_bwd");
}

// ---- the same Linear layer on the f16x2 matrix pipes (x2.hip, round 5): the fp32 class of error at ~3x the fp32-MFMA rate for the
// projections' real sizes (thousands of rows); K % 128 == 0 and N % 128 == 0.
int rnnt_engine_linear_x2_workspace_bytes(int M, int K, int N, int backward, size_t *out)
{
    if (!out) return engine_fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (M <= 0 || K <= 0 || N <= 0) return engine_fail(RNNT_ERR_INVALID_ARG, "non-positive dimension");
    if (!x2_linear_ok(M, K, N)) return engine_fail(RNNT_ERR_UNSUPPORTED, "the f16x2 linear kernels need K %% 128 == 0 and N %% 128 == 0 (K=%d N=%d)", K, N);
    *out = x2_linear_ws_bytes(M, K, N, backward != 0);
    return RNNT_OK;
}

int rnnt_engine_linear_x2_fwd(const float *x, int64_t ldx, const float *W, const float *bias, int M, int K, int N, float *y,
                              void *workspace, size_t ws_bytes, void *stream)
{
    size_t need;
    if (int rc = rnnt_engine_linear_x2_workspace_bytes(M, K, N, 0, &need)) return rc;
    if (ldx % 4 != 0 || ldx < K) return engine_fail(RNNT_ERR_UNSUPPORTED, "the row stride must be a multiple of 4 and >= K");
    if (!x || !W || !y || !workspace) return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const void *ptrs[] = {x, W, y, workspace, bias};
    for (const void *q : ptrs)
        if ((uintptr_t)q & 15) return engine_fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned");
    if (ws_bytes < need) return engine_fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    launch_linear_x2_fwd(x, ldx, W, bias, M, K, N, y, workspace, (hipStream_t)stream);
    return status("rnnt_engine_linear_x2_fwd");
}

int rnnt_engine_linear_x2_bwd(const float *x, int64_t ldx, const float *W, const float *dy, int M, int K, int N,
                              float *dx /* [M,K] contiguous or NULL */, float *dW, float *db /* or NULL */,
                              void *workspace, size_t ws_bytes, void *stream)
{
    size_t need;
    if (int rc = rnnt_engine_linear_x2_workspace_bytes(M, K, N, 1, &need)) return rc;
    if (ldx % 4 != 0 || ldx < K) return engine_fail(RNNT_ERR_UNSUPPORTED, "the row stride must be a multiple of 4 and >= K");
    if (!x || !W || !dy || !dW || !workspace) return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const void *ptrs[] = {x, W, dy, dW, workspace, dx, db};
    for (const void *q : ptrs)
        if ((uintptr_t)q & 15) return engine_fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned");
    if (ws_bytes < need) return engine_fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    launch_linear_x2_bwd(x, ldx, W, dy, M, K, N, dx, dW, db, workspace, (hipStream_t)stream);
    return status("rnnt_engine_linear_x2_bwd");
}

}  // extern "C"
