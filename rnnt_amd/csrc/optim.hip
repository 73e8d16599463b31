// optim.hip — gradient-norm clip + AdamW step over a LIST of tensors, for gfx950
// (the step right after the joint+loss path).
//
// Replaces, for the parameters handed over,
//     total_norm = torch.nn.utils.clip_grad_norm_(params, clip)     reference rnnt/train.py:136
//     optimizer.step()   with torch.optim.AdamW(lr, betas, eps, weight_decay)
//                                           rnnt/train.py:164, config/*.yaml training.optimizer
// Arithmetic restated from torch's documented single-tensor AdamW (decoupled weight decay, no
// amsgrad, no maximize) and clip_grad_norm_ (2-norm of all gradients, coefficient
// max_norm / (total_norm + 1e-6) clamped to 1):
//     p <- p * (1 - lr*wd);  m <- m + (1-b1)*(g - m);  v <- b2*v + (1-b2)*g*g
//     p <- p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//
// HBM-bound elementwise work: 28 B per element for the step (read p,g,m,v; write p,m,v), 4 B for
// the norm.  Multi-tensor: ONE launch covers up to MT_MAX tensors — their pointers travel in the
// kernel arguments, a workgroup finds its (tensor, chunk) from the chunk prefix sums — instead of
// one launch per parameter (the reference's models have ~150 parameter tensors).  The clip
// coefficient is read from device memory, so clip + step need no host synchronisation; the norm is
// reduced through per-workgroup partials in a fixed order (bitwise reproducible).
#include "../../include/rnnt_engine.h"
#include "common.hpp"
#include "kernels.hpp"

#define MT_MAX 40          // tensors per launch (kernel arguments stay under 4 KiB)
#define MT_CHUNK 4096      // elements per workgroup
#define MT_THREADS 256

struct MtNormArgs {
    const float *g[MT_MAX];
    long n[MT_MAX];
    int chunk0[MT_MAX + 1];  // chunk prefix sums: tensor i owns chunks [chunk0[i], chunk0[i+1])
    int count;
    float *partials;         // one float per chunk of this launch
};

struct MtAdamArgs {
    float *p[MT_MAX];
    const float *g[MT_MAX];
    float *m[MT_MAX];
    float *v[MT_MAX];
    long n[MT_MAX];
    int chunk0[MT_MAX + 1];
    int count;
    float beta2, eps, max_norm;
    // python-scalar arithmetic of torch's AdamW, done in double on the host: 1-b1, 1-b2, 1-lr*wd, lr/(1-b1^t), sqrt(1-b2^t)
    float omb1, omb2, decay, step_size, bc2_sqrt;
    const float *total_norm;  // device scalar or NULL
    int write_grads;          // store the clipped gradients back (clip_grad_norm_ is in place)
    // capturable form (rnnt_engine_adamw_step_dev): {decay, step_size, bc2_sqrt} prepared on the device by
    // k_adam_prepare from a device step counter and a device learning rate; NULL: the by-value fields above
    const float *dev_hyper;
};

template <class A>
__device__ __forceinline__ int mt_find(const A &a, int blk)
{
    int t = 0;
    while (t + 1 < a.count && a.chunk0[t + 1] <= blk) ++t;  // scalar loop over <= MT_MAX entries
    return t;
}

__global__ __launch_bounds__(MT_THREADS) void k_mt_sumsq(MtNormArgs a)
{
    __shared__ float s_w[MT_THREADS / 64];
    const int t = mt_find(a, blockIdx.x);
    const long base = (long)(blockIdx.x - a.chunk0[t]) * MT_CHUNK;
    const float *g = a.g[t] + base;
    const long left = a.n[t] - base;
    const int cnt = (int)(left < MT_CHUNK ? left : MT_CHUNK);
    float s = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const int n4 = cnt >> 2;
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += MT_THREADS) {
            const f32x4 x = ((const f32x4 *)g)[i];
            s += (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
        }
        for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += MT_THREADS) s += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < cnt; i += MT_THREADS) s += g[i] * g[i];
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// total_norm = sqrt(sum of the partials), fixed summation order
__global__ __launch_bounds__(MT_THREADS) void k_mt_norm_final(const float *__restrict__ partials, int n,
                                                              float *__restrict__ out)
{
    __shared__ double s_w[MT_THREADS / 64];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += MT_THREADS) s += (double)partials[i];
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (float)sqrt((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

// the scalars of one update (kernel arguments or, capturable form, read from the device)
struct AdamScalars { float decay, step_size, bc2_sqrt, omb1, beta2, omb2, eps; };
__device__ __forceinline__ void adamw_one(float &p, float &g, float &m, float &v, const AdamScalars &a, float coef)
{
    g *= coef;
    p *= a.decay;
    m += a.omb1 * (g - m);                  // exp_avg.lerp_(grad, 1 - beta1)
    v = a.beta2 * v + a.omb2 * (g * g);     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= a.step_size * (m / denom);         // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// step counter += 1, then the three scalars of this update in double, as the host path computes them
__global__ void k_adam_prepare(long long *__restrict__ step, const float *__restrict__ lr, double beta1, double beta2,
                               double weight_decay, float *__restrict__ hyper)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long t = step[0] + 1;
    step[0] = t;
    const double l = (double)lr[0];
    hyper[0] = (float)(1.0 - l * weight_decay);
    hyper[1] = (float)(l / (1.0 - pow(beta1, (double)t)));
    hyper[2] = (float)sqrt(1.0 - pow(beta2, (double)t));
}

__global__ __launch_bounds__(MT_THREADS) void k_mt_adamw(const MtAdamArgs a)
{
    // (the argument block is only READ: writing the device-side scalars into it, as round 2 did, made hipcc copy all
    // 1.8 KB of it into every thread's scratch — 470 KB of private-memory traffic per workgroup before the first load)
    AdamScalars sc = {a.decay, a.step_size, a.bc2_sqrt, a.omb1, a.beta2, a.omb2, a.eps};
    if (a.dev_hyper) {  // workgroup-uniform
        sc.decay = a.dev_hyper[0];
        sc.step_size = a.dev_hyper[1];
        sc.bc2_sqrt = a.dev_hyper[2];
    }
    const int t = mt_find(a, blockIdx.x);
    const long base = (long)(blockIdx.x - a.chunk0[t]) * MT_CHUNK;
    float *p = a.p[t] + base, *m = a.m[t] + base, *v = a.v[t] + base;
    const float *g = a.g[t] + base;
    const long left = a.n[t] - base;
    const int cnt = (int)(left < MT_CHUNK ? left : MT_CHUNK);
    float coef = 1.f;
    if (a.total_norm && a.max_norm > 0.f) {
        coef = a.max_norm / (a.total_norm[0] + 1e-6f);
        coef = coef > 1.f ? 1.f : coef;
    }
    const bool wg = a.write_grads && a.total_norm && a.max_norm > 0.f;
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    int done = 0;
    if (vec) {
        const int n4 = cnt >> 2;
        auto one = [&](int i, f32x4 P, f32x4 G, f32x4 M, f32x4 V) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float pq = P[q], gq = G[q], mq = M[q], vq = V[q];
                adamw_one(pq, gq, mq, vq, sc, coef);
                P[q] = pq; G[q] = gq; M[q] = mq; V[q] = vq;
            }
            ((f32x4 *)p)[i] = P; ((f32x4 *)m)[i] = M; ((f32x4 *)v)[i] = V;
            if (wg) ((f32x4 *)const_cast<float *>(g))[i] = G;
        };
        // four quads per thread and trip, all 16 loads requested before the first is used (one quad per trip left the
        // kernel waiting out a memory round trip per 64 bytes and thread: 1.5 TB/s)
        int i = threadIdx.x;
        for (; i + 3 * MT_THREADS < n4; i += 4 * MT_THREADS) {
            f32x4 P[4], G[4], M[4], V[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ik = i + k * MT_THREADS;
                P[k] = ((f32x4 *)p)[ik]; G[k] = ((const f32x4 *)g)[ik]; M[k] = ((f32x4 *)m)[ik]; V[k] = ((f32x4 *)v)[ik];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) one(i + k * MT_THREADS, P[k], G[k], M[k], V[k]);
        }
        for (; i < n4; i += MT_THREADS) one(i, ((f32x4 *)p)[i], ((const f32x4 *)g)[i], ((f32x4 *)m)[i], ((f32x4 *)v)[i]);
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < cnt; i += MT_THREADS) {
        float P = p[i], G = g[i], M = m[i], V = v[i];
        adamw_one(P, G, M, V, sc, coef);
        p[i] = P; m[i] = M; v[i] = V;
        if (wg) const_cast<float *>(g)[i] = G;
    }
}

namespace {
long chunks_of(long n) { return (n + MT_CHUNK - 1) / MT_CHUNK; }
}  // namespace

extern "C" {

int rnnt_engine_grad_norm_workspace_bytes(int n_tensors, const int64_t *numels, size_t *out)
{
    if (!out || (n_tensors > 0 && !numels) || n_tensors < 0) return engine_fail(RNNT_ERR_INVALID_ARG, "bad argument");
    long c = 0;
    for (int i = 0; i < n_tensors; ++i) {
        if (numels[i] < 0) return engine_fail(RNNT_ERR_INVALID_ARG, "negative element count");
        c += chunks_of(numels[i]);
    }
    *out = (size_t)(c + 1) * 4 + 256;
    return RNNT_OK;
}

int rnnt_engine_grad_norm(int n_tensors, const void *const *grads, const int64_t *numels,
                          float *total_norm, void *workspace, size_t ws_bytes, void *stream)
{
    size_t need;
    if (int rc = rnnt_engine_grad_norm_workspace_bytes(n_tensors, numels, &need)) return rc;
    if (!total_norm || !workspace || (n_tensors > 0 && !grads)) return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    if (ws_bytes < need) return engine_fail(RNNT_ERR_WORKSPACE, "gradient-norm workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float *partials = (float *)workspace;
    int total_chunks = 0;
    for (int i0 = 0; i0 < n_tensors;) {
        MtNormArgs a;
        a.count = 0;
        a.chunk0[0] = 0;
        a.partials = partials + total_chunks;
        while (i0 < n_tensors && a.count < MT_MAX) {
            if (numels[i0] > 0) {
                if (!grads[i0] || ((uintptr_t)grads[i0] & 3)) return engine_fail(RNNT_ERR_INVALID_ARG, "null / misaligned gradient pointer");
                a.g[a.count] = (const float *)grads[i0];
                a.n[a.count] = numels[i0];
                a.chunk0[a.count + 1] = a.chunk0[a.count] + (int)chunks_of(numels[i0]);
                ++a.count;
            }
            ++i0;
        }
        if (a.count == 0) break;
        hipLaunchKernelGGL(k_mt_sumsq, dim3(a.chunk0[a.count]), dim3(MT_THREADS), 0, st, a);
        total_chunks += a.chunk0[a.count];
    }
    hipLaunchKernelGGL(k_mt_norm_final, dim3(1), dim3(MT_THREADS), 0, st, partials, total_chunks, total_norm);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return engine_fail(RNNT_ERR_LAUNCH, "%s", hipGetErrorString(e));
    return RNNT_OK;
}

namespace {
int adamw_launch(MtAdamArgs &a, int n_tensors, void *const *params, const void *const *grads, void *const *exp_avg,
                 void *const *exp_avg_sq, const int64_t *numels, hipStream_t st);
}

int rnnt_engine_adamw_step(int n_tensors, void *const *params, const void *const *grads,
                           void *const *exp_avg, void *const *exp_avg_sq, const int64_t *numels,
                           double lr, double beta1, double beta2, double eps, double weight_decay,
                           int64_t step, const float *total_norm, float max_norm, int write_clipped_grads,
                           void *stream)
{
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numels)))
        return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    if (step < 1) return engine_fail(RNNT_ERR_INVALID_ARG, "step must be >= 1 (the count AFTER this update)");
    if (!(lr >= 0.) || !(eps >= 0.) || !(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(weight_decay >= 0.))
        return engine_fail(RNNT_ERR_INVALID_ARG, "invalid AdamW hyper-parameter");
    MtAdamArgs a;
    // scalar arithmetic in double on the host, as torch does with python floats (1 - 0.9999 in
    // fp32 would be off by 1.6e-4 relative)
    a.beta2 = (float)beta2; a.eps = (float)eps;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.decay = (float)(1.0 - lr * weight_decay);
    a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    a.max_norm = max_norm; a.total_norm = total_norm; a.write_grads = write_clipped_grads;
    a.dev_hyper = nullptr;
    return adamw_launch(a, n_tensors, params, grads, exp_avg, exp_avg_sq, numels, (hipStream_t)stream);
}

int rnnt_engine_adamw_step_dev(int n_tensors, void *const *params, const void *const *grads,
                               void *const *exp_avg, void *const *exp_avg_sq, const int64_t *numels,
                               const float *lr_dev, double beta1, double beta2, double eps, double weight_decay,
                               int64_t *step_dev, float *hyper_dev, const float *total_norm, float max_norm,
                               int write_clipped_grads, void *stream)
{
    if (n_tensors < 0 || (n_tensors > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numels)))
        return engine_fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    if (!lr_dev || !step_dev || !hyper_dev) return engine_fail(RNNT_ERR_INVALID_ARG, "null device scalar (lr / step / hyper)");
    if (!(eps >= 0.) || !(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(weight_decay >= 0.))
        return engine_fail(RNNT_ERR_INVALID_ARG, "invalid AdamW hyper-parameter");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, st, (long long *)step_dev, lr_dev, beta1, beta2, weight_decay, hyper_dev);
    MtAdamArgs a;
    a.beta2 = (float)beta2; a.eps = (float)eps;
    a.omb1 = (float)(1.0 - beta1); a.omb2 = (float)(1.0 - beta2);
    a.decay = 1.f; a.step_size = 0.f; a.bc2_sqrt = 1.f;  // replaced on the device
    a.max_norm = max_norm; a.total_norm = total_norm; a.write_grads = write_clipped_grads;
    a.dev_hyper = hyper_dev;
    return adamw_launch(a, n_tensors, params, grads, exp_avg, exp_avg_sq, numels, st);
}

}  // extern "C"

namespace {
int adamw_launch(MtAdamArgs &a, int n_tensors, void *const *params, const void *const *grads, void *const *exp_avg,
                 void *const *exp_avg_sq, const int64_t *numels, hipStream_t st)
{
    for (int i0 = 0; i0 < n_tensors;) {
        a.count = 0;
        a.chunk0[0] = 0;
        while (i0 < n_tensors && a.count < MT_MAX) {
            if (numels[i0] < 0) return engine_fail(RNNT_ERR_INVALID_ARG, "negative element count");
            if (numels[i0] > 0) {
                const void *ptrs[4] = {params[i0], grads[i0], exp_avg[i0], exp_avg_sq[i0]};
                for (const void *q : ptrs)
                    if (!q || ((uintptr_t)q & 3)) return engine_fail(RNNT_ERR_INVALID_ARG, "null / misaligned tensor pointer");
                a.p[a.count] = (float *)params[i0]; a.g[a.count] = (const float *)grads[i0];
                a.m[a.count] = (float *)exp_avg[i0]; a.v[a.count] = (float *)exp_avg_sq[i0];
                a.n[a.count] = numels[i0];
                a.chunk0[a.count + 1] = a.chunk0[a.count] + (int)chunks_of(numels[i0]);
                ++a.count;
            }
            ++i0;
        }
        if (a.count == 0) break;
        hipLaunchKernelGGL(k_mt_adamw, dim3(a.chunk0[a.count]), dim3(MT_THREADS), 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return engine_fail(RNNT_ERR_LAUNCH, "%s", hipGetErrorString(e));
    return RNNT_OK;
}
}  // namespace
