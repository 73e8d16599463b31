// joint_fwd.hip — joint-network forward GEMM for gfx950 with the log-softmax fused in.
//
// Replaces reference rnnt/joint.py:32-39 (broadcast add -> tanh -> Linear H->V) and the
// log-softmax / gather half of the loss called at rnnt/model.py:35-41:
//   logits[b,t,u,:] = tanh(enc[b,t,:] + pred[b,u,:]) @ W^T + bias          (materialised)
//   denom = logsumexp_v logits ; lp_blank = logits[blank]-denom ; lp_emit = logits[y_u]-denom
//
// Design (MI355X-first, fp32 exact):
//  * GEMM M = lattice cells (128 per workgroup, linear cell index inside one utterance),
//    K = H, N = V.  v_mfma_f32_32x32x2_f32 takes ONE f32 VGPR per operand and occupies the
//    matrix pipe for 64 cycles, so operands go straight from L2/HBM to registers — no LDS
//    staging, no barriers in the main loop; LDS only carries the per-row softmax state.
//  * The A operand (hidden) is never stored: each lane loads 16 B of enc and pred for its
//    row and applies tanh in registers.  A lane's 4 consecutive k feed 4 MFMAs because the
//    k-order inside an 8-wide chunk is a free permutation: lanes 0-31 take k0..k0+3, lanes
//    32-63 take k0+4..k0+7 for both A and B.
//  * W is re-packed once per call (pack_w_fwd) so that a B fragment is one lane-linear
//    1 KiB load; columns are interleaved by 4 (tile q of a 128-column group holds columns
//    4j+q) so that a lane's accumulators for q=0..3 are 4 consecutive logits -> 16-byte
//    coalesced stores (512 contiguous bytes per half-wave).
//  * 8 waves = 4(M) x 2(N); each wave owns a 32 x 256 tile (8 accumulator tiles, 128 VGPRs)
//    and the workgroup walks V in passes of 512 columns keeping a running (max, sum) per
//    row, so the row's log-sum-exp is complete when the last pass ends.
#include "common.hpp"
#include "kernels.hpp"

#define FWD_ROWS 128

size_t wpack_floats(int H, int V)
{
    const size_t HK = (size_t)(H + 7) / 8, NG = (size_t)(V + 127) / 128;
    return HK * NG * 1024;
}

// wpack float4 index ((c8*NG + ng)*4 + q)*64 + lane holds
//   W[ng*128 + 4*(lane&31) + q][8*c8 + 4*(lane>>5) + 0..3]   (zero outside [V,H]).
__global__ __launch_bounds__(256) void k_pack_w_fwd(const float *__restrict__ W,
                                                    f32x4 *__restrict__ wpack, int H, int V,
                                                    int HK, int NG)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)HK * NG * 256;
    if (idx >= n) return;
    const int lane = (int)(idx & 63);
    const int q = (int)((idx >> 6) & 3);
    const long r = idx >> 8;
    const int ng = (int)(r % NG);
    const int c8 = (int)(r / NG);
    const int v = ng * 128 + 4 * (lane & 31) + q;
    const int h = 8 * c8 + 4 * (lane >> 5);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (v < V) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            if (h + s < H) o[s] = W[(long)v * H + h + s];
    }
    wpack[idx] = o;
}

void launch_pack_w_fwd(const float *W, float *wpack, int H, int V, hipStream_t st)
{
    const int HK = (H + 7) / 8, NG = (V + 127) / 128;
    const long n = (long)HK * NG * 256;
    hipLaunchKernelGGL(k_pack_w_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W,
                       (f32x4 *)wpack, H, V, HK, NG);
}

__global__ __launch_bounds__(256) void k_copy_enc(const float *__restrict__ enc, long sb, long st_,
                                                  long sh, float *__restrict__ dst, int B, int T,
                                                  int H)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)B * T * H;
    if (idx >= n) return;
    const int h = (int)(idx % H);
    const long bt = idx / H;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    dst[idx] = enc[(long)b * sb + (long)t * st_ + (long)h * sh];
}

void launch_copy_enc(const float *enc, long sb, long st_, long sh, float *dst, int B, int T, int H,
                     hipStream_t st)
{
    const long n = (long)B * T * H;
    hipLaunchKernelGGL(k_copy_enc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, enc, sb,
                       st_, sh, dst, B, T, H);
}

struct FwdFrag {
    f32x4 e, p;
    f32x4 w[8];
};

__device__ __forceinline__ void fwd_load_a(FwdFrag &f, const float *erow, const float *prow,
                                           int c8, bool koob, int hlim)
{
    const int k = 8 * c8;
    const bool ok = !koob || (k < hlim);  // hlim = H - 4*half
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f.e = ok ? *(const f32x4 *)(erow + k) : z;
    f.p = ok ? *(const f32x4 *)(prow + k) : z;
}

// One 8-wide k chunk: 32 MFMAs.  Operands are single-buffered with a rolling refill: the
// enc/pred slices of chunk c8+1 are requested as soon as tanh has consumed chunk c8's, and
// B fragment q of chunk c8+1 is requested right after the 4 MFMAs that read fragment q of
// chunk c8 — each request then has >= 28 MFMAs (~1800 cycles) of matrix work in front of
// its first use, without a second register set.
template <bool G1>
__device__ __forceinline__ void fwd_chunk(FwdFrag &f, const float *erow, const float *prow,
                                          const f32x4 *wp, int c8, int HK, long wstride,
                                          bool koob, int hlim, f32x16 (&acc)[8])
{
    float a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = fast_tanh(f.e[s] + f.p[s]);
    const bool more = (c8 + 1) < HK;
    if (more) fwd_load_a(f, erow, prow, c8 + 1, koob, hlim);
    const f32x4 *wn = wp + (long)(c8 + 1) * wstride;
#pragma unroll
    for (int q = 0; q < (G1 ? 8 : 4); ++q) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], f.w[q][s], acc[q], 0, 0, 0);
        if (more) f.w[q] = wn[q * 64];
    }
}

template <bool WITH_LOSS>
__global__ __launch_bounds__(512, 2) void k_joint_fwd(JointFwdArgs a)
{
    __shared__ float s_m[2][FWD_ROWS], s_s[2][FWD_ROWS], s_blank[FWD_ROWS], s_emit[FWD_ROWS];
    __shared__ int s_y[FWD_ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    const int m0 = blockIdx.x * FWD_ROWS;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int Tb = WITH_LOSS ? a.logit_lens[b] : T;
    const int ncell = Tb * U1;
    if (m0 >= ncell) return;
    const int Ub = WITH_LOSS ? a.target_lens[b] : U1 - 1;

    if (WITH_LOSS) {
        if (tid < FWD_ROWS) {
            const int c = m0 + tid;
            int y = -1;
            if (c < ncell) {
                const int t = c / U1, u = c - t * U1;
                if (u < Ub) y = a.targets[(long)b * (U1 - 1) + u];
            }
            s_y[tid] = y;
            s_m[0][tid] = RNNT_NEG_INF; s_m[1][tid] = RNNT_NEG_INF;
            s_s[0][tid] = 0.f; s_s[1][tid] = 0.f;
            s_blank[tid] = 0.f; s_emit[tid] = 0.f;
        }
        __syncthreads();
    }

    // this lane's A row: cell -> (t,u) -> 16-byte slices of enc / pred
    const int crow = min(m0 + wm * 32 + i, ncell - 1);
    const int trow = crow / U1, urow = crow - trow * U1;
    const float *erow = a.enc + (long)b * a.enc_sb + (long)trow * a.enc_st + 4 * half;
    const float *prow = a.pred + ((long)b * U1 + urow) * H + 4 * half;
    const int HK = (H + 7) / 8, NG = (V + 127) / 128;
    const long wstride = (long)NG * 256;  // float4 per 8-wide k chunk
    const bool koob = (H & 7) != 0;
    const int hlim = H - 4 * half;
    const int npass = (NG + 3) / 4;

    for (int pass = 0; pass < npass; ++pass) {
        const int ng0 = pass * 4 + wn * 2;
        if (ng0 >= NG) continue;  // wave-uniform: no columns for this wave in this pass
        const bool g1 = (ng0 + 1) < NG;
        const int col0[2] = {ng0 * 128 + 4 * i, (ng0 + 1) * 128 + 4 * i};
        const bool cok[2] = {col0[0] < V, g1 && col0[1] < V};

        f32x16 acc[8];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if (cok[g]) bq = *(const f32x4 *)(a.bias + col0[g]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g * 4 + q][r] = bq[q];
        }

        const f32x4 *wp = (const f32x4 *)a.wpack + (long)ng0 * 256 + lane;
        FwdFrag f;
        fwd_load_a(f, erow, prow, 0, koob, hlim);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f.w[q] = (q < 4 || g1) ? wp[q * 64] : z;
        }
        if (g1) {
            for (int c8 = 0; c8 < HK; ++c8)
                fwd_chunk<true>(f, erow, prow, wp, c8, HK, wstride, koob, hlim, acc);
        } else {
            for (int c8 = 0; c8 < HK; ++c8)
                fwd_chunk<false>(f, erow, prow, wp, c8, HK, wstride, koob, hlim, acc);
        }

        // ---- epilogue: store logits, fold this pass into the running row log-sum-exp
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rowl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int c = m0 + rowl;
            const bool rv = c < ncell;
            float *lrow = a.logits + ((long)b * T * U1 + c) * V;
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (rv && cok[g]) {
                    f32x4 o = {acc[g * 4 + 0][r], acc[g * 4 + 1][r], acc[g * 4 + 2][r],
                               acc[g * 4 + 3][r]};
                    *(f32x4 *)(lrow + col0[g]) = o;
                }
            if (WITH_LOSS) {
                float mloc = RNNT_NEG_INF;
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (cok[g]) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) mloc = fmaxf(mloc, acc[g * 4 + q][r]);
                    }
                const float mrow = half_max(mloc);
                float sloc = 0.f;
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (cok[g]) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) sloc += __expf(acc[g * 4 + q][r] - mrow);
                    }
                const float srow = half_sum(sloc);
                const int y = s_y[rowl];
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (cok[g]) {
                        const int dy = y - col0[g], db = a.blank - col0[g];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (dy == q) s_emit[rowl] = acc[g * 4 + q][r];
                            if (db == q) s_blank[rowl] = acc[g * 4 + q][r];
                        }
                    }
                if (i == 0) {
                    const float mo = s_m[wn][rowl], so = s_s[wn][rowl];
                    const float mn = fmaxf(mo, mrow);
                    s_s[wn][rowl] = so * __expf(mo - mn) + srow * __expf(mrow - mn);
                    s_m[wn][rowl] = mn;
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the 16 row bodies from interleaving
        }
    }

    if (WITH_LOSS) {
        __syncthreads();
        if (tid < FWD_ROWS) {
            const int c = m0 + tid;
            if (c < ncell) {
                const int t = c / U1, u = c - t * U1;
                const float ma = s_m[0][tid], mb = s_m[1][tid];
                const float m = fmaxf(ma, mb);
                const float s = s_s[0][tid] * __expf(ma - m) + s_s[1][tid] * __expf(mb - m);
                const float den = m + logf(s);
                const long si = skew_index(b, t, u, a.D, U1);
                a.denom_s[si] = den;
                a.lpb_s[si] = s_blank[tid] - den;
                a.lpe_s[si] = (s_y[tid] >= 0) ? s_emit[tid] - den : 0.f;
            }
        }
    }
}

void launch_joint_fwd(const JointFwdArgs &a, hipStream_t st)
{
    const int tiles = (int)(((long)a.T * a.U1 + FWD_ROWS - 1) / FWD_ROWS);
    dim3 grid(tiles, a.B), block(512);
    if (a.denom_s)
        hipLaunchKernelGGL(k_joint_fwd<true>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(k_joint_fwd<false>, grid, block, 0, st, a);
}
