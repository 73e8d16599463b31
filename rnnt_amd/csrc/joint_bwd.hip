// joint_bwd.hip — backward GEMMs of the fused joint + transducer loss for gfx950.
//
// Replaces what loss.backward() (reference rnnt/train.py:134) runs for the path
// rnnt/model.py:32-41: the loss gradient w.r.t. logits (torchaudio),
// joint_ln's backward and the tanh / broadcast-add backward of rnnt/joint.py:32-39.
//
// The gradient w.r.t. logits, G, is never stored.  Both GEMMs regenerate it on the fly
// from the materialised logits and the 16-byte per-cell CellCoef:
//      G[c,v] = exp2(logit[c,v]*log2e + c1[c]) - (v==blank)*sb[c] - (v==y[c])*se[c]
//
//  k_dhidden  dHidden[c,:] = G[c,:] @ W          (M = cells, K = V, N = H)
//             then dPre = dHidden * (1 - tanh^2), reduced in the epilogue over the 16 u of
//             the tile (-> dEnc partial) and over its 8 t (-> dPred partial); deterministic
//             partial slabs, summed by k_reduce_*.
//  k_dw       dW[v,h] = sum_c G[c,v] * hidden[c,h]   (M = V, N = H, K = cells, split-K)
//             hidden is recomputed from enc/pred (tanh in registers); db is the in-lane row
//             sum of the same G fragments.
//
// Like the forward kernel these feed v_mfma_f32_32x32x2_f32 straight from registers
// (one VGPR per operand, 64 matrix-pipe cycles per instruction): no LDS staging in the
// main loops.  A lane's 16-byte load supplies either 4 k-steps (k contiguous in memory:
// logits for k_dhidden) or 4 interleaved tiles (m/n contiguous in memory: W rows, logits
// for k_dw, enc/pred), so every global access is a full 16 B per lane.
#include "common.hpp"
#include "kernels.hpp"

#define DH_BT 8
#define DH_BU 16

struct DhFrag {
    f32x4 x[2];  // logits slices for the two M tiles
    f32x4 w[4];  // W rows k0+4*half+s, columns n0+4j..4j+3
};

// One 8-wide k chunk (k = vocabulary index): 32 MFMAs.  Same single-buffer rolling refill
// as the forward kernel: the logits slices of chunk c+1 are requested once G has been
// generated from chunk c's, and W row s of chunk c+1 right after the 8 MFMAs that read row
// s of chunk c.  Steady-state loads are unconditional (exact vmcnt counting); the last chunk
// is peeled.  `kill` zeroes G for lanes whose k range lies beyond V (V % 8 == 4 tail).
template <bool LAST>
__device__ __forceinline__ void dh_chunk(DhFrag &f, const CellCoef (&cf)[2], int vb, int blank,
                                         bool kill, const float *x0n, const float *x1n,
                                         const float *wn, int H, f32x16 (&acc)[2][4])
{
    float g[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int dy = cf[mt].y - vb;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float e = __builtin_amdgcn_exp2f(fmaf(f.x[mt][s], RNNT_LOG2E, cf[mt].c1));
            if (dy == s) e -= cf[mt].se;
            if (vb + s == blank) e -= cf[mt].sb;
            g[mt][s] = (LAST && kill) ? 0.f : e;
        }
    }
    if (!LAST) {
        f.x[0] = *(const f32x4 *)x0n;
        f.x[1] = *(const f32x4 *)x1n;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[mt][q] =
                    __builtin_amdgcn_mfma_f32_32x32x2f32(g[mt][s], f.w[s][q], acc[mt][q], 0, 0, 0);
        if (!LAST) f.w[s] = *(const f32x4 *)(wn + (long)s * H);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// grid (n_ublk, n_ttile, B * n_hblk); 8 waves = 2(M) x 4(N); wave tile 64 cells x 128 cols.
__global__ __launch_bounds__(512, 2) void k_dhidden(JointBwdArgs a)
{
    __shared__ float s_red[4][64][33];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int i = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int ub = blockIdx.x, tt = blockIdx.y;
    const int b = blockIdx.z % a.B, hb = blockIdx.z / a.B;
    const int Tb = a.logit_lens[b];
    const int t0 = tt * DH_BT, u0 = ub * DH_BU;
    if (t0 >= Tb) return;  // workgroup-uniform
    const int ncol0 = hb * 512 + wn * 128;
    const int col = ncol0 + 4 * i;
    const bool colok = col < H;
    const bool wave_on = ncol0 < H;

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;

    if (wave_on) {
        CellCoef cf[2];
        const float *lptr[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = wm * 64 + mt * 32 + i;
            const int t = t0 + (row >> 4), u = u0 + (row & 15);
            const bool ok = t < Tb && u < U1;
            const long cell = ((long)b * T + (ok ? t : t0)) * U1 + (ok ? u : u0);
            cf[mt] = a.coef[cell];
            if (!ok) { cf[mt].c1 = RNNT_NEG_INF; cf[mt].sb = 0.f; cf[mt].se = 0.f; cf[mt].y = -1; }
            lptr[mt] = a.logits + cell * V + 4 * half;
        }
        // columns beyond H (last column block only) read column 0 instead: their products
        // land in accumulator columns that are never stored
        const float *wptr = a.W + (long)(4 * half) * H + (colok ? col : 0);
        const int VK = (V + 7) / 8;
        // V % 8 == 4: in the last chunk lanes 32-63 would read k >= V; step them back 4
        const bool kill = ((V & 7) != 0) && half == 1;
        const int back = kill ? 4 : 0;
        DhFrag f;
        {
            const int b0 = (VK == 1) ? back : 0;
            f.x[0] = *(const f32x4 *)(lptr[0] - b0);
            f.x[1] = *(const f32x4 *)(lptr[1] - b0);
#pragma unroll
            for (int s = 0; s < 4; ++s) f.w[s] = *(const f32x4 *)(wptr + (long)(s - b0) * H);
        }
        for (int c8 = 0; c8 + 2 < VK; ++c8) {
            const int kn = 8 * (c8 + 1);
            dh_chunk<false>(f, cf, 8 * c8 + 4 * half, a.blank, false, lptr[0] + kn, lptr[1] + kn,
                            wptr + (long)kn * H, H, acc);
        }
        if (VK >= 2) {
            const int c8 = VK - 2, kn = 8 * (c8 + 1) - back;
            dh_chunk<false>(f, cf, 8 * c8 + 4 * half, a.blank, false, lptr[0] + kn, lptr[1] + kn,
                            wptr + (long)kn * H, H, acc);
        }
        dh_chunk<true>(f, cf, 8 * (VK - 1) + 4 * half, a.blank, kill, nullptr, nullptr, nullptr, H,
                       acc);
    }

    // ---- epilogue: dPre = dHidden * (1 - tanh^2); reduce over u (dEnc) and over t (dPred)
    float psum[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) psum[k][q] = 0.f;

    const long BTH = (long)a.B * T * H, BUH = (long)a.B * U1 * H;
    if (wave_on) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                const int tl = wm * 4 + mt * 2 + rh;
                const int t = t0 + tl;
                const bool tok = t < Tb;
                f32x4 e4 = {0.f, 0.f, 0.f, 0.f};
                if (tok && colok)
                    e4 = *(const f32x4 *)(a.enc + (long)b * a.enc_sb + (long)t * a.enc_st + col);
                float esum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r7 = 0; r7 < 8; ++r7) {
                    const int ul = 8 * (r7 >> 2) + 4 * half + (r7 & 3);
                    const int u = u0 + ul;
                    const bool ok = tok && u < U1 && colok;
                    f32x4 p4 = {0.f, 0.f, 0.f, 0.f};
                    if (ok) p4 = *(const f32x4 *)(a.pred + ((long)b * U1 + u) * H + col);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float h = fast_tanh(e4[q] + p4[q]);
                        const float d = ok ? acc[mt][q][rh * 8 + r7] * (1.f - h * h) : 0.f;
                        esum[q] += d;
                        psum[r7][q] += d;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) esum[q] += __shfl_xor(esum[q], 32, 64);
                if (half == 0 && tok && colok) {
                    f32x4 o = {esum[0], esum[1], esum[2], esum[3]};
                    *(f32x4 *)(a.slab_enc + (long)ub * BTH + ((long)b * T + t) * H + col) = o;
                }
            }
    }
    if (wm == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_red[wn][lane][k * 4 + q] = psum[k][q];
    }
    __syncthreads();
    if (wm == 0 && wave_on && colok) {
#pragma unroll
        for (int r7 = 0; r7 < 8; ++r7) {
            const int u = u0 + 8 * (r7 >> 2) + 4 * half + (r7 & 3);
            if (u < U1) {
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = psum[r7][q] + s_red[wn][lane][r7 * 4 + q];
                *(f32x4 *)(a.slab_pred + (long)tt * BUH + ((long)b * U1 + u) * H + col) = o;
            }
        }
    }
}

// out[b,t,:] = sum_ub slab_enc[ub][b,t,:]  (0 for t >= T_b)
__global__ __launch_bounds__(256) void k_reduce_enc(const float *__restrict__ slab,
                                                    const int32_t *__restrict__ logit_lens,
                                                    float *__restrict__ out, int B, int T, int H,
                                                    int n_ublk)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
    const int H4 = H / 4;
    const long n = (long)B * T * H4;
    if (idx >= n) return;
    const long bt = idx / H4;
    const int t = (int)(bt % T), b = (int)(bt / T);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (t < logit_lens[b])
        for (int k = 0; k < n_ublk; ++k) s += ((const f32x4 *)slab)[(long)k * n + idx];
    ((f32x4 *)out)[idx] = s;
}

// out[b,u,:] = sum_{tt < ceil(T_b/8)} slab_pred[tt][b,u,:]
__global__ __launch_bounds__(256) void k_reduce_pred(const float *__restrict__ slab,
                                                     const int32_t *__restrict__ logit_lens,
                                                     float *__restrict__ out, int B, int U1, int H)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H4 = H / 4;
    const long n = (long)B * U1 * H4;
    if (idx >= n) return;
    const int b = (int)(idx / ((long)U1 * H4));
    const int ntt = (logit_lens[b] + DH_BT - 1) / DH_BT;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ntt; ++k) s += ((const f32x4 *)slab)[(long)k * n + idx];
    ((f32x4 *)out)[idx] = s;
}

void launch_dhidden(const JointBwdArgs &a, hipStream_t st)
{
    const int n_hblk = (a.H + 511) / 512;
    dim3 grid(a.n_ublk, a.n_ttile, a.B * n_hblk);
    hipLaunchKernelGGL(k_dhidden, grid, dim3(512), 0, st, a);
    const long n4e = (long)a.B * a.T * (a.H / 4);
    hipLaunchKernelGGL(k_reduce_enc, dim3((unsigned)((n4e + 255) / 256)), dim3(256), 0, st,
                       a.slab_enc, a.logit_lens, a.grad_enc, a.B, a.T, a.H, a.n_ublk);
    const long n4p = (long)a.B * a.U1 * (a.H / 4);
    hipLaunchKernelGGL(k_reduce_pred, dim3((unsigned)((n4p + 255) / 256)), dim3(256), 0, st,
                       a.slab_pred, a.logit_lens, a.grad_pred, a.B, a.U1, a.H);
}

// ---------------------------------------------------------------------------------------
// dW split-K GEMM.  grid (n_vblk, n_hblk, n_split); 4 waves = 2(M) x 2(N); wave tile
// 128 (v) x 128 (h) = 16 accumulator tiles (256 VGPRs, one wave per SIMD).
struct DwRaw {
    f32x4 x;  // logits[cell][v0+4i .. +3]
    f32x4 p;  // pred[b,u][h0+4j .. +3]
    f32x4 e;  // enc[b,t][h0+4j .. +3]
    CellCoef c;
    bool live;  // false for padding k-steps and for the u == U1 half of an odd row
};

// Walks the (b,t) rows of this split, two lattice cells (u, u+1) per k-step, purely with
// wave-uniform integer selects so that the software-pipelined loop body stays ONE basic
// block with unconditional loads (exact vmcnt counting).
struct DwCursor {
    long bt, step;
    int b, t, us;
};

struct DwLane {   // per-lane constants of the issue path
    int vsafe, hsafe;  // first column of this lane's logits / hidden slice (clamped to 0)
    int hV, hH, h1;    // half ? V : 0, half ? H : 0, half
};

__device__ __forceinline__ void dw_issue(DwRaw &r, DwCursor &c, const JointBwdArgs &a,
                                         const DwLane &ln, long nstep, long bt_last, int SPB)
{
    const int U1 = a.U1;
    // lanes 0-31 take cell u = 2*us, lanes 32-63 cell u+1; on the odd tail of a row the upper
    // half re-reads cell u (valid memory) and is masked through `live`.  All row/step
    // arithmetic is wave-uniform (SALU); the per-lane part is one select + adds.
    const bool tail = (2 * c.us + 1) >= U1;
    const long cell0 = c.bt * U1 + 2 * c.us;
    const long prow0 = (long)c.b * U1 + 2 * c.us;
    r.live = (c.step < nstep) && !(tail && ln.h1);
    r.x = *(const f32x4 *)(a.logits + cell0 * a.V + ((tail ? 0 : ln.hV) + ln.vsafe));
    r.p = *(const f32x4 *)(a.pred + prow0 * a.H + ((tail ? 0 : ln.hH) + ln.hsafe));
    r.e = *(const f32x4 *)(a.enc + (long)c.b * a.enc_sb + (long)c.t * a.enc_st + ln.hsafe);
    r.c = a.coef[cell0 + (tail ? 0 : ln.h1)];
    // advance (stays on the last row once the split is exhausted: addresses remain valid)
    c.step += 1;
    const bool wrap = (c.us + 1) == SPB;
    const bool adv = wrap && (c.bt < bt_last);
    c.us = wrap ? 0 : c.us + 1;
    const bool nb = adv && (c.t + 1 == a.T);
    c.bt += adv ? 1 : 0;
    c.t = adv ? (nb ? 0 : c.t + 1) : c.t;
    c.b += nb ? 1 : 0;
}

// One k-step of the software pipeline, written as four sub-blocks so that the VALU work
// that generates the NEXT step's operands (G = A operand, 4 interleaved v tiles; hidden =
// B operand, 4 interleaved h tiles) issues in the shadow of the CURRENT step's 16 MFMAs:
// sub-block q = {generate operand q of step s+1, 4 MFMAs of row q of step s}.  Cells
// outside the lattice (k_coef marks them with c1 = -inf) contribute exactly 0.
__device__ __forceinline__ void dw_step(const DwRaw &rn, float (&gN)[4], float (&hN)[4],
                                        const float (&gC)[4], const float (&hC)[4], int vbase,
                                        int blank, f32x16 (&acc)[4][4], float (&dbacc)[4])
{
    const bool ok = rn.live && (rn.c.c1 != RNNT_NEG_INF);
    const float c1 = ok ? rn.c.c1 : RNNT_NEG_INF;
    const float sb = ok ? rn.c.sb : 0.f, se = ok ? rn.c.se : 0.f;
    const int dy = rn.c.y - vbase, db = blank - vbase;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = ok ? rn.x[q] : 0.f;  // rows t >= T_b were never written by the forward
        float e = __builtin_amdgcn_exp2f(fmaf(x, RNNT_LOG2E, c1));
        if (dy == q) e -= se;
        if (db == q) e -= sb;
        gN[q] = e;
        dbacc[q] += e;
        hN[q] = fast_tanh(rn.e[q] + rn.p[q]);
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
            acc[q][qn] = __builtin_amdgcn_mfma_f32_32x32x2f32(gC[q], hC[qn], acc[q][qn], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // 5 VALU in its shadow
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void dw_gen_first(const DwRaw &r, int vbase, int blank, float (&g)[4],
                                             float (&hd)[4], float (&dbacc)[4])
{
    const bool ok = r.live && (r.c.c1 != RNNT_NEG_INF);
    const float c1 = ok ? r.c.c1 : RNNT_NEG_INF;
    const float sb = ok ? r.c.sb : 0.f, se = ok ? r.c.se : 0.f;
    const int dy = r.c.y - vbase, db = blank - vbase;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = ok ? r.x[q] : 0.f;
        float e = __builtin_amdgcn_exp2f(fmaf(x, RNNT_LOG2E, c1));
        if (dy == q) e -= se;
        if (db == q) e -= sb;
        g[q] = e;
        dbacc[q] += e;
        hd[q] = fast_tanh(r.e[q] + r.p[q]);
    }
}

__global__ __launch_bounds__(256, 1) void k_dw(JointBwdArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int v0 = blockIdx.x * 256 + wm * 128;
    const int h0 = blockIdx.y * 256 + wn * 128;
    const int split = blockIdx.z;
    const int vbase = v0 + 4 * i, hbase = h0 + 4 * i;
    const bool vok = vbase < V, hok = hbase < H;
    const long nbt = (long)a.B * T;
    const long bt_lo = nbt * split / a.n_split, bt_hi = nbt * (split + 1) / a.n_split;

    f32x16 acc[4][4];
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};

    if (v0 < V && h0 < H && bt_hi > bt_lo) {  // wave-uniform
        const int SPB = (U1 + 1) / 2;
        const long nstep = (bt_hi - bt_lo) * SPB;
        // rows / columns beyond V / H (edge tiles) read row/column 0: their results sit in
        // accumulator rows / columns that are never stored
        DwLane ln;
        ln.vsafe = vok ? vbase : 0; ln.hsafe = hok ? hbase : 0;
        ln.hV = half ? V : 0; ln.hH = half ? H : 0; ln.h1 = half;
        DwCursor cur;
        cur.bt = bt_lo; cur.step = 0; cur.us = 0;
        cur.b = (int)(bt_lo / T); cur.t = (int)(bt_lo - (long)cur.b * T);
        const long bt_last = bt_hi - 1;
        DwRaw r0, r1, r2, r3;
        float gA[4], hA[4], gB[4], hB[4];
        // 4 raw k-steps in flight (~4 x 1024 matrix-pipe cycles of lead for the HBM-streamed
        // logits); operands of step s+1 are generated while the 16 MFMAs of step s issue.
        dw_issue(r0, cur, a, ln, nstep, bt_last, SPB);
        dw_issue(r1, cur, a, ln, nstep, bt_last, SPB);
        dw_issue(r2, cur, a, ln, nstep, bt_last, SPB);
        dw_issue(r3, cur, a, ln, nstep, bt_last, SPB);
        dw_gen_first(r0, vbase, a.blank, gA, hA, dbacc);
        dw_issue(r0, cur, a, ln, nstep, bt_last, SPB);
        __builtin_amdgcn_sched_barrier(0);
        for (long s = 0; s < nstep; s += 4) {
            // at the top: (gA,hA) = operands of step s; r1..r3,r0 hold raw steps s+1..s+4
            dw_step(r1, gB, hB, gA, hA, vbase, a.blank, acc, dbacc);
            dw_issue(r1, cur, a, ln, nstep, bt_last, SPB);
            __builtin_amdgcn_sched_barrier(0);
            dw_step(r2, gA, hA, gB, hB, vbase, a.blank, acc, dbacc);
            dw_issue(r2, cur, a, ln, nstep, bt_last, SPB);
            __builtin_amdgcn_sched_barrier(0);
            dw_step(r3, gB, hB, gA, hA, vbase, a.blank, acc, dbacc);
            dw_issue(r3, cur, a, ln, nstep, bt_last, SPB);
            __builtin_amdgcn_sched_barrier(0);
            dw_step(r0, gA, hA, gB, hB, vbase, a.blank, acc, dbacc);
            dw_issue(r0, cur, a, ln, nstep, bt_last, SPB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: partial slab [split][V,H]; bias partial [split][V]
    float *sw = a.slab_w + (long)split * V * H;
    if (hok) {
#pragma unroll
        for (int qm = 0; qm < 4; ++qm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int v = v0 + 4 * row + qm;
                if (v < V) {
                    f32x4 o = {acc[qm][0][r], acc[qm][1][r], acc[qm][2][r], acc[qm][3][r]};
                    *(f32x4 *)(sw + (long)v * H + hbase) = o;
                }
            }
    }
    if (blockIdx.y == 0 && wn == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dbacc[q] += __shfl_xor(dbacc[q], 32, 64);
        if (half == 0 && vok) {
            f32x4 o = {dbacc[0], dbacc[1], dbacc[2], dbacc[3]};
            *(f32x4 *)(a.slab_b + (long)split * V + vbase) = o;
        }
    }
}

// out[i] = sum_s slab[s][i]   (float4 granularity)
__global__ __launch_bounds__(256) void k_reduce_slabs(const float *__restrict__ slab,
                                                      float *__restrict__ out, long n4, int ns)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ns; ++k) s += ((const f32x4 *)slab)[(long)k * n4 + idx];
    ((f32x4 *)out)[idx] = s;
}

void launch_dw(const JointBwdArgs &a, hipStream_t st)
{
    dim3 grid((a.V + 255) / 256, (a.H + 255) / 256, a.n_split);
    hipLaunchKernelGGL(k_dw, grid, dim3(256), 0, st, a);
    const long n4w = (long)a.V * a.H / 4;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n4w + 255) / 256)), dim3(256), 0, st,
                       a.slab_w, a.grad_W, n4w, a.n_split);
    const long n4b = a.V / 4;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n4b + 255) / 256)), dim3(256), 0, st,
                       a.slab_b, a.grad_bias, n4b, a.n_split);
}
