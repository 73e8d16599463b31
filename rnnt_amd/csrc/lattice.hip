// lattice.hip — transducer-loss lattice kernels for gfx950 (HBM/latency-bound part).
//
// Replaces the arithmetic behind torchaudio.functional.rnnt_loss as called at reference
// rnnt/model.py:35-41 (see SURVEY.md §8c for the recurrences):
//   k_logsoftmax_gather  logits -> (denom, lp_blank, lp_emit) in skewed layout
//                        (only for the standalone loss entry; the fused path gets these
//                         from the joint-forward GEMM epilogue)
//   k_lattice<DIR>       alpha / beta anti-diagonal wavefront sweep, one workgroup per
//                        (utterance, direction), previous diagonal staged in LDS
//   k_coef               per-cell gradient coefficients (CellCoef) from alpha/beta
//   k_grad_logits        d cost / d logits, elementwise (standalone loss entry only)
//
// All per-cell work arrays use the skewed layout of common.hpp (anti-diagonal contiguous),
// so every sweep load/store is a coalesced row access.  alpha/beta are kept in fp64: the
// values reach ~1e4 in magnitude at T=1000 and the gradient needs alpha+beta+cost, a
// cancellation fp32 cannot hold to 1e-4; the per-step log-add-exp correction term is
// evaluated in fp32 (it is < ln 2 in magnitude).
#include "common.hpp"
#include "kernels.hpp"

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------
// log-softmax denominators + the two log-probs each lattice cell needs.
// One wave per (b,t,u) row; V % 4 == 0.
__global__ __launch_bounds__(256) void k_logsoftmax_gather(
    const float *__restrict__ logits, const int32_t *__restrict__ targets,
    const int32_t *__restrict__ logit_lens, const int32_t *__restrict__ target_lens,
    float *__restrict__ denom_s, float *__restrict__ lpb_s, float *__restrict__ lpe_s, int B,
    int T, int U1, int V, int D, int blank)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nrows = (long)B * T * U1;
    if (row >= nrows) return;
    const int u = (int)(row % U1);
    const long bt = row / U1;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    if (t >= len_t(logit_lens, b, T) || u > len_u(target_lens, b, U1)) return;
    const float *x = logits + row * V;
    float m = RNNT_NEG_INF;
    for (int v = lane * 4; v < V; v += 256) {
        f32x4 q = *(const f32x4 *)(x + v);
        m = fmaxf(m, fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3])));
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k, 64));
    float s = 0.f;
    for (int v = lane * 4; v < V; v += 256) {
        f32x4 q = *(const f32x4 *)(x + v);
        s += __expf(q[0] - m) + __expf(q[1] - m) + __expf(q[2] - m) + __expf(q[3] - m);
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k, 64);
    if (lane == 0) {
        const float den = m + logf(s);
        const long si = skew_index(b, t, u, D, U1);
        denom_s[si] = den;
        lpb_s[si] = x[blank] - den;
        lpe_s[si] = (u < len_u(target_lens, b, U1)) ? x[targets[(long)b * (U1 - 1) + u]] - den : 0.f;
    }
}

// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double logaddexp_mixed(double a, double e)
{
    const double m = fmax(a, e);
    if (m == (double)RNNT_NEG_INF) return m;
    const float dl = (float)(fmin(a, e) - m);
    // hardware exp2/log2 (v_exp_f32 / v_log_f32): the correction is in (0, ln 2], absolute error
    // ~1e-7 per step, far below the 1e-4 budget after T+U steps, and it shortens the serial chain
    return m + (double)__logf(1.0f + __expf(dl));
}

// One workgroup per (utterance, direction); thread u owns lattice column u.  The previous
// anti-diagonal is exchanged through a double-buffered LDS line (one barrier per step);
// the lp values of the next four diagonals are prefetched into registers because they do
// not depend on the recurrence.  DIR 0: alpha, forward over d = 0..nd-1.  DIR 1: beta,
// backward; also emits costs[b] = -beta[0,0].
template <int DIR>
__device__ __forceinline__ void lattice_sweep(
    const float *__restrict__ lpb_s, const float *__restrict__ lpe_s,
    double *__restrict__ out_s, const int32_t *__restrict__ logit_lens,
    const int32_t *__restrict__ target_lens, float *__restrict__ costs, int U1, int D)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int NT = blockDim.x;
    const int b = blockIdx.x;
    const int u = threadIdx.x;
    const int T = D - U1 + 1;
    const int Tb = len_t(logit_lens, b, T);
    const int Ub = len_u(target_lens, b, U1);
    const int nd = Tb + Ub;  // valid anti-diagonals 0 .. nd-1
    double *buf[2] = {sm, sm + NT + 2};
    const double NINF = (double)RNNT_NEG_INF;
    buf[0][u + 1] = NINF;
    buf[1][u + 1] = NINF;
    if (u == 0) {
        buf[0][0] = NINF; buf[0][NT + 1] = NINF;
        buf[1][0] = NINF; buf[1][NT + 1] = NINF;
    }
    __syncthreads();
    const long base = (long)b * D * U1;
    const float *lpb = lpb_s + base;
    const float *lpe = lpe_s + base;
    double *out = out_s + base;
    const bool col_ok = (u <= Ub);

    // The step is a chain of dependent instructions on one wave per SIMD, so its length is what
    // a diagonal costs: no divergent branches (selects only), unconditional clamped loads, and
    // the lattice boundary handled by the data — cells outside the lattice hold -inf in the LDS
    // line, so "no such neighbour" needs no test.
    // lp pair step k consumes (k counts steps: diagonal d = k for alpha, d = nd-1-k for beta);
    // 0 when the cell or the transition does not exist (the arrays are only written for lattice
    // cells: whatever else they hold must not reach the arithmetic).
    bool bad = false;
    double last = NINF;
    auto fetch = [&](int k, float &lb, float &le) {
        const int kk = k < nd ? k : nd - 1;
        const int d = DIR == 0 ? kk : nd - 1 - kk;
        const int t = d - u;
        const bool cell = (k < nd) && col_ok && t >= 0 && t < Tb;
        const int uc = u < U1 ? u : U1 - 1;  // the block is padded to whole waves: stay inside the row
        if (DIR == 0) {
            const int dr = d > 0 ? d - 1 : 0;
            const float vb = lpb[(long)dr * U1 + uc];
            const float ve = lpe[(long)dr * U1 + (uc > 0 ? uc - 1 : 0)];
            lb = (cell && t > 0) ? vb : 0.f;
            le = (cell && u > 0) ? ve : 0.f;
        } else {
            const float vb = lpb[(long)d * U1 + uc];
            const float ve = lpe[(long)d * U1 + uc];
            bad = bad || (cell && vb != vb);  // a NaN lp_blank inside the lattice -> a NaN cost (see lattice_chain)
            lb = cell ? vb : 0.f;
            le = (cell && u < Ub) ? ve : 0.f;
        }
    };
    // -inf-safe log(exp(a) + exp(e)): the correction term is evaluated in fp32 (it is < ln 2)
    auto lae = [&](double a, double e) {
        const double m = fmax(a, e);
        const float dl = (m == NINF) ? 0.f : (float)(fmin(a, e) - m);
        // hardware exp2/log2 directly: the argument of the log is in (1, 2], no range fix-ups
        // (logf/__logf expand to ~15 instructions of them) in this serial chain
        return m + (double)(__builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(dl * RNNT_LOG2E)) * 0.6931471805599453f);
    };
    auto step = [&](int k, float lb, float le) {
        const double *prev = buf[(k & 1) ^ 1];
        double *cur = buf[k & 1];
        const int d = DIR == 0 ? k : nd - 1 - k;
        const int t = d - u;
        const bool valid = (k < nd) && col_ok && t >= 0 && t < Tb;
        // alpha: prev[u+1] = alpha[t-1,u], prev[u] = alpha[t,u-1]
        // beta:  prev[u+1] = beta[t+1,u],  prev[u+2] = beta[t,u+1]
        const double a = prev[u + 1] + (double)lb;
        const double e = prev[DIR == 0 ? u : u + 2] + (double)le;
        double val = lae(a, e);
        // the first cell of each sweep has no predecessor: alpha[0,0] = 0, beta[Tb-1,Ub] = lp_blank
        if (k == 0) val = DIR == 0 ? 0.0 : (double)lb;  // k is uniform; only one thread is valid
        val = valid ? val : NINF;
        cur[u + 1] = val;
        if (k < nd && u < U1) out[(long)d * U1 + u] = val;
        if (DIR == 1 && k == nd - 1) last = val;  // (thread u == 0 holds beta[0,0]; the cost is written after the sweep)
        // LDS-only barrier: __syncthreads() would also wait for vmcnt(0), i.e. for this step's
        // store and for the lp prefetch of four steps ahead
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    float lb0, le0, lb1, le1, lb2, le2, lb3, le3;
    fetch(0, lb0, le0); fetch(1, lb1, le1); fetch(2, lb2, le2); fetch(3, lb3, le3);
    for (int k = 0; k < nd; k += 4) {  // nd is workgroup-uniform: every thread runs the
        step(k, lb0, le0);             // same number of barriers
        fetch(k + 4, lb0, le0);
        step(k + 1, lb1, le1);
        fetch(k + 5, lb1, le1);
        step(k + 2, lb2, le2);
        fetch(k + 6, lb2, le2);
        step(k + 3, lb3, le3);
        fetch(k + 7, lb3, le3);
    }
    if (DIR == 1) {
        const int any_bad = __syncthreads_or(bad ? 1 : 0);
        if (u == 0) costs[b] = any_bad ? __builtin_nanf("") : (float)(-last);
    }
}

// ---------------------------------------------------------------------------------------
// Wave-chain sweep (used whenever its mailboxes fit in LDS): thread u owns column u as above, but
// the waves of the workgroup are NOT joined by a barrier per step.  Inside a wave the neighbour
// column of the previous diagonal moves by a DPP wave shift (no LDS in the dependent chain); the
// one value per step that crosses a wave boundary goes through an LDS mailbox [boundary][step]
// (value, then tag = step+1; written and read in program order, so a matching tag guarantees the
// value) that the consuming wave reads one step ahead of its use.  The dependency is one-way
// (alpha: wave w needs wave w-1; beta: w needs w+1), a producer never waits, so the waves settle
// one step behind each other like a systolic array and a step costs its own ~45 instructions
// instead of LDS write -> s_barrier -> LDS read across four waves (cfg2: 0.36 -> see DESIGN).
// Every mailbox slot is written exactly once (no ring, no back-pressure); the consumer's spin is
// bounded, so every wave reaches its exit whatever happens.
template <int CTRL>
__device__ __forceinline__ double wave_shift(double v, double fill)
{
    const u32x2_t x = __builtin_bit_cast(u32x2_t, v), f = __builtin_bit_cast(u32x2_t, fill);
    u32x2_t r;
    r[0] = (unsigned)__builtin_amdgcn_update_dpp((int)f[0], (int)x[0], CTRL, 0xf, 0xf, false);
    r[1] = (unsigned)__builtin_amdgcn_update_dpp((int)f[1], (int)x[1], CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, r);
}

#define LAT_PF 6
template <int DIR>
__device__ __forceinline__ void lattice_chain(
    const float *__restrict__ lpb_s, const float *__restrict__ lpe_s, double *__restrict__ out_s,
    const int32_t *__restrict__ logit_lens, const int32_t *__restrict__ target_lens,
    float *__restrict__ costs, int U1, int D, unsigned *__restrict__ err)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int NW = blockDim.x >> 6;
    const int u = threadIdx.x, l = u & 63;
    const int w = __builtin_amdgcn_readfirstlane(u >> 6);
    // explicit LDS address space: volatile accesses through a generic pointer become flat_* ops
    typedef __attribute__((address_space(3))) volatile double lds_f64;
    typedef __attribute__((address_space(3))) volatile int lds_i32;
    lds_f64 *mval = (lds_f64 *)sm;                        // [NW-1][D]
    lds_i32 *mtag = (lds_i32 *)(sm + (size_t)(NW - 1) * D);  // [NW-1][D]
    for (int i = u; i < (NW - 1) * D; i += blockDim.x) mtag[i] = 0;
    __syncthreads();  // the only barrier of the sweep

    const int b = blockIdx.x;
    const int T = D - U1 + 1;
    const int Tb = len_t(logit_lens, b, T), Ub = len_u(target_lens, b, U1);
    const int nd = Tb + Ub;  // valid anti-diagonals 0 .. nd-1
    const long base = (long)b * D * U1;
    const float *lpb = lpb_s + base, *lpe = lpe_s + base;
    double *out = out_s + base;
    const double NINF = (double)RNNT_NEG_INF;
    const int ukey = u <= Ub ? u : (1 << 30);  // a key no diagonal matches outside the lattice
    const bool st_ok = u < U1;
    const int uc = st_ok ? u : U1 - 1;
    // alpha: wave w publishes its lane 63 into boundary w and consumes boundary w-1;
    // beta:  wave w publishes its lane 0 into boundary w-1 and consumes boundary w
    const bool has_prod = DIR == 0 ? w < NW - 1 : w > 0;
    const bool has_cons = DIR == 0 ? w > 0 : w < NW - 1;
    const int pb = (DIR == 0 ? w : w - 1) * D, cb = (DIR == 0 ? w - 1 : w) * D;
    const bool plane = l == (DIR == 0 ? 63 : 0);

    // lp values of the transitions INTO the cells of step k (diagonal d = k / nd-1-k): alpha from
    // the cells of diagonal d-1 (columns u and u-1), beta the cell's own.  Clamped to <= 0 (a
    // log-prob; whatever an unwritten slot holds, NaN included, becomes harmless: it is only ever
    // added to -inf or dropped by the validity select).
    auto fetch = [&](int k, float &lb, float &le) {
        const int kk = k < nd ? k : nd - 1;  // k >= 1
        if (DIR == 0) {
            const long r = (long)(kk - 1) * U1;
            lb = lpb[r + uc];
            le = lpe[r + (uc > 0 ? uc - 1 : 0)];
        } else {
            const long r = (long)(nd - 1 - kk) * U1;
            lb = lpb[r + uc];
            le = lpe[r + uc];
        }
    };
    auto publish = [&](int k, double v) {
        if (has_prod && plane) {
            mval[pb + k] = v;
            mtag[pb + k] = k + 1;
        }
    };

    // A NaN log-prob INSIDE the lattice (non-finite enc / pred / W / bias: a diverged run) must come out as a NaN cost, as it does from the
    // reference's loss — but the clamp below canonicalises NaNs away (it has to: slots outside the lattice hold anything).  The beta sweep
    // reads lp_blank of every lattice cell exactly once, and a cell whose logits hold a NaN has a NaN lp_blank (its denominator is NaN): one
    // unordered compare per step under the validity mask, off the dependent chain, joined once at the end of the sweep.
    bool bad = false;
    // ---- step 0: alpha[0,0] = 0; beta[Tb-1,Ub] = lp_blank there
    double prev;
    {
        const int d = DIR == 0 ? 0 : nd - 1;
        const double first = DIR == 0 ? 0.0 : (double)lpb[(long)d * U1 + Ub];
        if (DIR == 1) bad = (u == Ub) && (first != first);
        prev = (u == (DIR == 0 ? 0 : Ub)) ? first : NINF;
        if (st_ok) out[(long)d * U1 + u] = prev;
        publish(0, prev);
    }
    // mailbox of step k-1, requested one step early
    int tg = 0;
    double bv = NINF;
    if (has_cons) { tg = mtag[cb]; bv = mval[cb]; }
    auto step = [&](int k, float lb, float le) {
        const int d = DIR == 0 ? k : nd - 1 - k;
        if (has_cons && tg != k) {  // wave-uniform; rare once the waves have settled a step apart
            for (int spin = 0; tg != k && spin < (1 << 22); ++spin) { tg = mtag[cb + k - 1]; bv = mval[cb + k - 1]; }
            // bounded spin exhausted (the producer wave never published step k-1: cannot happen
            // while the workgroup is resident, but a silent stale value would be a wrong loss):
            // raise the error word; k_lattice_status then turns every cost of the call into NaN
            if (tg != k && l == 0) atomicOr(err, 1u);
        }
        // neighbour column of the previous diagonal: u-1 (alpha) / u+1 (beta); the lane at the
        // wave's edge keeps the mailbox value (-inf at the lattice's edge)
        const double nb = DIR == 0 ? wave_shift<0x138>(prev, bv) : wave_shift<0x130>(prev, bv);
        if (has_cons) { tg = mtag[cb + k]; bv = mval[cb + k]; }  // next step's, early (slot k < D always exists)
        // fminf, not a bare v_min_f32: in IEEE mode the instruction turns a SIGNALLING NaN into a
        // quiet NaN result, and slots of the lp arrays that no kernel wrote may hold any bit pattern
        // (found by tools/fuzz_lattice.py); fminf canonicalises first (v_max x,x) and then returns 0
        const bool lb_nan = lb != lb;  // (DIR == 1: lp_blank of THIS step's cell; used under `valid` below)
        lb = fminf(lb, 0.f);
        le = fminf(le, 0.f);
        const double a = prev + (double)lb;
        const double e = nb + (double)le;
        // log(exp(a) + exp(e)) = max + log(1 + exp(-|a - e|)); the correction is < ln 2 and
        // evaluated in fp32 with the hardware exp2/log2.  A cell of the lattice has at least one
        // finite predecessor; everything else is overwritten by the select.
        const float dl = -fabsf((float)(a - e));
        const double v = fmax(a, e) +
            (double)(__builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(dl * RNNT_LOG2E)) * 0.6931471805599453f);
        const bool valid = (unsigned)(d - ukey) < (unsigned)Tb;
        if (DIR == 1) bad = bad || (valid && lb_nan);
        prev = valid ? v : NINF;
        publish(k, prev);
        if (st_ok) out[(long)d * U1 + u] = prev;
    };
    // full blocks of LAT_PF steps with the lp values prefetched one block ahead, then the tail
    float lbr[LAT_PF], ler[LAT_PF];
    int k0 = 1;
    if (nd - 1 >= LAT_PF) {
#pragma unroll
        for (int q = 0; q < LAT_PF; ++q) fetch(1 + q, lbr[q], ler[q]);
        for (; k0 + LAT_PF <= nd; k0 += LAT_PF) {
#pragma unroll
            for (int q = 0; q < LAT_PF; ++q) {
                step(k0 + q, lbr[q], ler[q]);
                fetch(k0 + q + LAT_PF, lbr[q], ler[q]);  // clamped to the last diagonal inside
            }
        }
    }
    for (; k0 < nd; ++k0) {
        float lb, le;
        fetch(k0, lb, le);
        step(k0, lb, le);
    }
    if (DIR == 1) {  // (block-uniform branch: every wave of the beta sweep reaches this barrier — all spins are bounded)
        const int any_bad = __syncthreads_or(bad ? 1 : 0);
        if (u == 0) costs[b] = any_bad ? __builtin_nanf("") : (float)(-prev);  // -beta[0,0]
    }
}

__global__ __launch_bounds__(1024) void k_lattice_chain(
    const float *__restrict__ lpb_s, const float *__restrict__ lpe_s,
    double *__restrict__ alpha_s, double *__restrict__ beta_s,
    const int32_t *__restrict__ logit_lens, const int32_t *__restrict__ target_lens,
    float *__restrict__ costs, int U1, int D, unsigned *__restrict__ err)
{
    if (blockIdx.y == 0)
        lattice_chain<0>(lpb_s, lpe_s, alpha_s, logit_lens, target_lens, costs, U1, D, err);
    else
        lattice_chain<1>(lpb_s, lpe_s, beta_s, logit_lens, target_lens, costs, U1, D, err);
}

// Failure report of the chained sweep: if any wave ran out of its bounded spin, the costs of the
// whole call become NaN (a loud failure the training loop sees in its loss) instead of numbers
// computed from a stale mailbox value.
__global__ void k_lattice_status(const unsigned *__restrict__ err, float *__restrict__ costs, int B)
{
    if (*err == 0u) return;
    for (int b = threadIdx.x; b < B; b += blockDim.x) costs[b] = __builtin_nanf("");
}

// ---------------------------------------------------------------------------------------
// Per-cell gradient coefficients.  One thread per skewed slot (b,d,u); reads are coalesced
// rows of the skewed arrays, the 16-byte CellCoef is written at the cell's natural
// (b,t,u) index where the GEMM kernels gather it.
__global__ __launch_bounds__(256) void k_coef(
    const double *__restrict__ alpha_s, const double *__restrict__ beta_s,
    const float *__restrict__ denom_s, const float *__restrict__ lpb_s,
    const float *__restrict__ lpe_s, const int32_t *__restrict__ targets,
    const int32_t *__restrict__ logit_lens, const int32_t *__restrict__ target_lens,
    CellCoef *__restrict__ coef, int B, int T, int U1, int D, float scale)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)D * U1;
    if (idx >= per * B) return;
    const int b = (int)(idx / per);
    const long rem = idx - (long)b * per;
    const int d = (int)(rem / U1);
    const int u = (int)(rem - (long)d * U1);
    const int t = d - u;
    if (t < 0 || t >= T) return;
    const int Tb = len_t(logit_lens, b, T), Ub = len_u(target_lens, b, U1);
    CellCoef c;
    c.c1 = RNNT_NEG_INF; c.sb = 0.f; c.se = 0.f; c.y = -1;
    if (t < Tb && u <= Ub) {
        const double a = alpha_s[idx];
        const double cost = -beta_s[(long)b * per];
        const double ac = a + cost;
        c.c1 = (float)((ac + beta_s[idx] - (double)denom_s[idx] + (double)logf(scale)) *
                       1.4426950408889634);
        if (t < Tb - 1) c.sb = scale * expf((float)(ac + beta_s[idx + U1] + (double)lpb_s[idx]));
        else if (u == Ub) c.sb = scale * expf((float)(ac + (double)lpb_s[idx]));
        if (u < Ub) {
            c.se = scale * expf((float)(ac + beta_s[idx + U1 + 1] + (double)lpe_s[idx]));
            c.y = targets[(long)b * (U1 - 1) + u];
        }
    }
    coef[((long)b * T + t) * U1 + u] = c;
}

// ---------------------------------------------------------------------------------------
// d cost / d logits, elementwise from CellCoef (standalone loss entry).  One wave per row.
__global__ __launch_bounds__(256) void k_grad_logits(const float *__restrict__ logits,
                                                     const CellCoef *__restrict__ coef,
                                                     float *__restrict__ grad, long nrows, int V,
                                                     int blank, float clamp)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const CellCoef c = coef[row];
    const float *x = logits + row * V;
    float *g = grad + row * V;
    const bool live = c.c1 != RNNT_NEG_INF;
    for (int v = lane * 4; v < V; v += 256) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            const f32x4 q = *(const f32x4 *)(x + v);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float e = __builtin_amdgcn_exp2f(fmaf(q[s], RNNT_LOG2E, c.c1));
                if (v + s == blank) e -= c.sb;
                if (v + s == c.y) e -= c.se;
                if (clamp > 0.f) e = fminf(fmaxf(e, -clamp), clamp);
                o[s] = e;
            }
        }
        *(f32x4 *)(g + v) = o;
    }
}

// ---------------------------------------------------------------------------------------
void launch_logsoftmax_gather(const float *logits, const int32_t *targets,
                              const int32_t *logit_lens, const int32_t *target_lens,
                              float *denom_s, float *lpb_s, float *lpe_s, int B, int T, int U1,
                              int V, int D, int blank, hipStream_t st)
{
    const long nrows = (long)B * T * U1;
    hipLaunchKernelGGL(k_logsoftmax_gather, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st,
                       logits, targets, logit_lens, target_lens, denom_s, lpb_s, lpe_s, B, T, U1,
                       V, D, blank);
}

// grid (B, 2): blockIdx.y selects the direction, so the 2B independent sweeps of a batch
// run concurrently on 2B compute units in ONE launch.
__global__ __launch_bounds__(1024) void k_lattice(
    const float *__restrict__ lpb_s, const float *__restrict__ lpe_s,
    double *__restrict__ alpha_s, double *__restrict__ beta_s,
    const int32_t *__restrict__ logit_lens, const int32_t *__restrict__ target_lens,
    float *__restrict__ costs, int U1, int D)
{
    if (blockIdx.y == 0)
        lattice_sweep<0>(lpb_s, lpe_s, alpha_s, logit_lens, target_lens, costs, U1, D);
    else
        lattice_sweep<1>(lpb_s, lpe_s, beta_s, logit_lens, target_lens, costs, U1, D);
}

void launch_lattice(const float *lpb_s, const float *lpe_s, double *alpha_s, double *beta_s,
                    const int32_t *logit_lens, const int32_t *target_lens, float *costs, int B,
                    int U1, int D, unsigned *err, hipStream_t st)
{
    const int NW = (U1 + 63) / 64;
    const size_t mbox = (size_t)(NW - 1) * D * 12;  // chain mailboxes: 8 B value + 4 B tag per boundary and step
    if (mbox <= 64 * 1024) {
        if (NW > 1) launch_fill32(err, 0u, 4, st);  // one wave: no mailbox, no spin
        hipLaunchKernelGGL(k_lattice_chain, dim3(B, 2), dim3(64 * NW), (mbox + 15) / 16 * 16, st, lpb_s, lpe_s, alpha_s,
                           beta_s, logit_lens, target_lens, costs, U1, D, err);
        if (NW > 1) hipLaunchKernelGGL(k_lattice_status, dim3(1), dim3(64), 0, st, err, costs, B);
        return;
    }
    const int NT = ((U1 + 63) / 64) * 64;
    const size_t lds = 2 * (size_t)(NT + 2) * sizeof(double);
    hipLaunchKernelGGL(k_lattice, dim3(B, 2), dim3(NT), lds, st, lpb_s, lpe_s, alpha_s, beta_s,
                       logit_lens, target_lens, costs, U1, D);
}

void launch_coef(const double *alpha_s, const double *beta_s, const float *denom_s,
                 const float *lpb_s, const float *lpe_s, const int32_t *targets,
                 const int32_t *logit_lens, const int32_t *target_lens, CellCoef *coef, int B,
                 int T, int U1, int D, float scale, hipStream_t st)
{
    const long n = (long)B * D * U1;
    hipLaunchKernelGGL(k_coef, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, alpha_s,
                       beta_s, denom_s, lpb_s, lpe_s, targets, logit_lens, target_lens, coef, B,
                       T, U1, D, scale);
}

void launch_grad_logits(const float *logits, const CellCoef *coef, float *grad, long nrows,
                        int V, int blank, float clamp, hipStream_t st)
{
    hipLaunchKernelGGL(k_grad_logits, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, st, logits,
                       coef, grad, nrows, V, blank, clamp);
}
