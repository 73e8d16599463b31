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

__device__ __forceinline__ void dh_load(DhFrag &f, const float *const (&lptr)[2],
                                        const float *wptr, int k0, int H, int vlim, bool colok)
{
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const bool vok = k0 < vlim;  // vlim = V - 4*half
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) f.x[mt] = vok ? *(const f32x4 *)(lptr[mt] + k0) : z;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        f.w[s] = (vok && colok) ? *(const f32x4 *)(wptr + (long)(k0 + s) * H) : z;
}

__device__ __forceinline__ void dh_compute(const DhFrag &f, const CellCoef (&cf)[2], int k0,
                                           int half, int blank, f32x16 (&acc)[2][4])
{
    float g[2][4];
    const int vb = k0 + 4 * half;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int dy = cf[mt].y - vb;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float e = __builtin_amdgcn_exp2f(fmaf(f.x[mt][s], RNNT_LOG2E, cf[mt].c1));
            if (dy == s) e -= cf[mt].se;
            if (vb + s == blank) e -= cf[mt].sb;
            g[mt][s] = e;
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[mt][q] =
                    __builtin_amdgcn_mfma_f32_32x32x2f32(g[mt][s], f.w[s][q], acc[mt][q], 0, 0, 0);
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
        const float *wptr = a.W + (long)(4 * half) * H + col;
        const int vlim = V - 4 * half;
        const int VK = (V + 7) / 8;
        DhFrag f0, f1;
        dh_load(f0, lptr, wptr, 0, H, vlim, colok);
        for (int c8 = 0; c8 < VK; c8 += 2) {
            const bool has1 = (c8 + 1) < VK;
            if (has1) dh_load(f1, lptr, wptr, 8 * (c8 + 1), H, vlim, colok);
            dh_compute(f0, cf, 8 * c8, half, a.blank, acc);
            if (has1) {
                if (c8 + 2 < VK) dh_load(f0, lptr, wptr, 8 * (c8 + 2), H, vlim, colok);
                dh_compute(f1, cf, 8 * (c8 + 1), half, a.blank, acc);
            }
        }
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
struct DwFrag {
    f32x4 x;   // logits[cell][v0+4i .. +3]
    f32x4 p;   // pred[b,u][h0+4j .. +3]
    CellCoef c;
};

__device__ __forceinline__ void dw_load(DwFrag &f, const float *lrow, const float *prow,
                                        const CellCoef *crow, int u, int U1, int V, int H,
                                        bool vok, bool hok)
{
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const bool uok = u < U1;
    const int uc = uok ? u : U1 - 1;
    f.x = vok ? *(const f32x4 *)(lrow + (long)uc * V) : z;
    f.p = hok ? *(const f32x4 *)(prow + (long)uc * H) : z;
    f.c = crow[uc];
    if (!uok) { f.c.c1 = RNNT_NEG_INF; f.c.sb = 0.f; f.c.se = 0.f; f.c.y = -1; }
}

__device__ __forceinline__ void dw_compute(const DwFrag &f, const f32x4 &e4, int vbase, int blank,
                                           f32x16 (&acc)[4][4], float (&dbacc)[4])
{
    float g[4], hd[4];
    const int dy = f.c.y - vbase, db = blank - vbase;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float e = __builtin_amdgcn_exp2f(fmaf(f.x[q], RNNT_LOG2E, f.c.c1));
        if (dy == q) e -= f.c.se;
        if (db == q) e -= f.c.sb;
        g[q] = e;
        dbacc[q] += e;
        hd[q] = fast_tanh(e4[q] + f.p[q]);
    }
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
            acc[qm][qn] =
                __builtin_amdgcn_mfma_f32_32x32x2f32(g[qm], hd[qn], acc[qm][qn], 0, 0, 0);
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

    if (v0 < V && h0 < H) {  // wave-uniform
        for (long bt = bt_lo; bt < bt_hi; ++bt) {
            const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
            if (t >= a.logit_lens[b]) continue;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 e4 =
                hok ? *(const f32x4 *)(a.enc + (long)b * a.enc_sb + (long)t * a.enc_st + hbase) : z;
            const float *lrow = a.logits + bt * U1 * V + (vok ? vbase : 0);
            const float *prow = a.pred + (long)b * U1 * H + (hok ? hbase : 0);
            const CellCoef *crow = a.coef + bt * U1;
            DwFrag f0, f1;
            dw_load(f0, lrow, prow, crow, half, U1, V, H, vok, hok);
            for (int u = 0; u < U1; u += 4) {
                const bool has1 = (u + 2) < U1;
                if (has1) dw_load(f1, lrow, prow, crow, u + 2 + half, U1, V, H, vok, hok);
                dw_compute(f0, e4, vbase, a.blank, acc, dbacc);
                if (has1) {
                    if (u + 4 < U1) dw_load(f0, lrow, prow, crow, u + 4 + half, U1, V, H, vok, hok);
                    dw_compute(f1, e4, vbase, a.blank, acc, dbacc);
                }
            }
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
