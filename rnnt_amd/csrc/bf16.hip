// bf16.hip — the RNNT_DTYPE_BF16 route of the fused pipeline (BASELINE config 3).
//
// Same path as the fp32 route (reference rnnt/joint.py:32-39 + torchaudio rnnt_loss called at
// rnnt/model.py:35-41 + their autograd), but the operands of the three GEMMs are rounded to
// bf16 (round-to-nearest-even) and multiplied by v_mfma_f32_32x32x16_bf16 with fp32
// accumulation — 16x the fp32 matrix rate, so this route is HBM-bound:
//   hidden = bf16(tanh(enc+pred))        k_make_hidden_bf16      writes  2H  B/cell
//   logits = hidden . bf16(W)^T + b      k_joint_fwd_bf16        reads 2H, writes 4V B/cell
//   softmax statistics, lattice, coef    fp32 / fp64, shared with the fp32 route (lattice.hip)
//   G = bf16(exp2(logit*log2e+c1) - ..)  k_dhidden_bf16          reads 4V+2H, writes 2V B/cell
//   dHidden = G . bf16(W), x(1-h^2), sums                        (same kernel)
//   dW = G^T . hidden, db = colsum(G)    k_dw_bf16               reads 2V+2H B/cell
// Logits stay fp32 (the loss needs them); G overwrites the first half of its logits row.
//
// MFMA operand maps (cdna_hip_programming.md §3): lane l = (r = l&31, h = l>>5) holds
// A[row r][k = 8h+j] and B[k = 8h+j][col r], j = 0..7; C/D: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Two conventions used throughout:
//  * K permutation: a 32-wide k chunk is consumed by 2 MFMAs; the lane's 16 consecutive k
//    (32 contiguous bytes of bf16, or 64 of fp32 logits) feed MFMA s=0 with its first 8 and
//    MFMA s=1 with its last 8: MFMA s, slot (h,j)  <->  k = 32c + 16h + 8s + j.  Both operands
//    use the same map, so the dot product is unchanged and every global access is >= 32 B/lane.
//  * Column interleave by 4: accumulator tile 4g+q holds columns 128g + 4*(l&31) + q, so a
//    lane's 4 tiles of a group are 4 adjacent columns -> 16-byte epilogue accesses.
// W is re-packed once per call (2 x 1 MB at H=512,V=1024) into exactly the order the B
// fragments are consumed, so staging a chunk is a linear copy L2 -> VGPR -> LDS.
#include "kernels.hpp"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// publish this wave's LDS writes / retire its LDS reads, then join the workgroup.  Not
// __syncthreads(): that also waits vmcnt(0) and would drain the global prefetch rings.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------
// producers
// ---------------------------------------------------------------------------------------
// hidden[c,:] = bf16(tanh(enc[b,t,:] + pred[b,u,:])), zero rows for c >= cells (row padding)
__global__ __launch_bounds__(256) void k_make_hidden_bf16(const float *__restrict__ enc, long sb,
                                                          long st_, const float *__restrict__ pred,
                                                          u32x2 *__restrict__ hid, int B, int T,
                                                          int U1, int H, long rows)
{
    const int H4 = H / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H4) return;
    const long c = idx / H4;
    const int h = (int)(idx - c * H4) * 4;
    u32x2 o = {0u, 0u};
    if (c < (long)B * T * U1) {
        const int u = (int)(c % U1);
        const long bt = c / U1;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const f32x4 e = *(const f32x4 *)(enc + (long)b * sb + (long)t * st_ + h);
        const f32x4 p = *(const f32x4 *)(pred + ((long)b * U1 + u) * H + h);
        o[0] = pack_bf16(fast_tanh(e[0] + p[0]), fast_tanh(e[1] + p[1]));
        o[1] = pack_bf16(fast_tanh(e[2] + p[2]), fast_tanh(e[3] + p[3]));
    }
    hid[idx] = o;
}

// forward B operand, fragment order: [pass][c][s][tile(8)][lane] x 8 bf16,
// element j = W[v = 256*pass + 128*(tile>>2) + 4*(lane&31) + (tile&3)][h = 32c + 16*(lane>>5) + 8s + j]
__global__ __launch_bounds__(256) void k_pack_w_fwd_bf16(const float *__restrict__ W,
                                                         u32x4 *__restrict__ out, int H, int V,
                                                         int KC, long n)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63), tile = (int)(idx >> 6) & 7, s = (int)(idx >> 9) & 1;
    const long cc = idx >> 10;
    const int c = (int)(cc % KC), pass = (int)(cc / KC);
    const int v = 256 * pass + 128 * (tile >> 2) + 4 * (lane & 31) + (tile & 3);
    const int h0 = 32 * c + 16 * (lane >> 5) + 8 * s;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (v < V) {
        const float *w = W + (long)v * H + h0;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf16(w[2 * j], w[2 * j + 1]);
    }
    out[idx] = o;
}

// dHidden B operand, fragment order: [c][s][tile(16)][lane] x 8 bf16,
// element j = W[v = 32c + 16*(lane>>5) + 8s + j][h = 128*(tile>>2) + 4*(lane&31) + (tile&3)]
__global__ __launch_bounds__(256) void k_pack_w_dh_bf16(const float *__restrict__ W,
                                                        u32x4 *__restrict__ out, int H, int V,
                                                        long n)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63), tile = (int)(idx >> 6) & 15, s = (int)(idx >> 10) & 1;
    const int c = (int)(idx >> 11);
    const int h = 128 * (tile >> 2) + 4 * (lane & 31) + (tile & 3);
    const int v0 = 32 * c + 16 * (lane >> 5) + 8 * s;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (h < H) {
        const float *w = W + (long)v0 * H + h;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf16(w[(long)(2 * j) * H], w[(long)(2 * j + 1) * H]);
    }
    out[idx] = o;
}

size_t bf16_wpack_fwd_bytes(int H, int V) { return (size_t)((V + 255) / 256) * (H / 32) * 2 * 8 * 64 * 16; }
size_t bf16_wpack_dh_bytes(int V) { return (size_t)(V / 32) * 2 * 16 * 64 * 16; }

void launch_bf16_producers(const Bf16Args &a, hipStream_t st)
{
    const long nh = a.rows_alloc * (a.H / 4);
    hipLaunchKernelGGL(k_make_hidden_bf16, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, st,
                       a.enc, a.enc_sb, a.enc_st, a.pred, (u32x2 *)a.hidden, a.B, a.T, a.U1, a.H,
                       a.rows_alloc);
    const long nf = (long)(bf16_wpack_fwd_bytes(a.H, a.V) / 16);
    hipLaunchKernelGGL(k_pack_w_fwd_bf16, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, a.W,
                       (u32x4 *)a.wpack_fwd, a.H, a.V, a.H / 32, nf);
    const long nd = (long)(bf16_wpack_dh_bytes(a.V) / 16);
    hipLaunchKernelGGL(k_pack_w_dh_bf16, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, a.W,
                       (u32x4 *)a.wpack_dh, a.H, a.V, nd);
}

// ---------------------------------------------------------------------------------------
// shared pieces of the two row-tile GEMMs: 4 waves, wave w owns rows 32w..32w+31 of the
// 128-row tile and ALL NT column tiles of the pass; the B chunk (32 k x 32*NT columns) is
// shared through LDS: each wave copies its quarter L2 -> VGPR (two chunks ahead) -> LDS.
// ---------------------------------------------------------------------------------------
template <int NT>
struct BStage {
    u32x4 r[NT / 2];
    __device__ __forceinline__ void load(const u32x4 *wp, long chunk, int wave, int lane)
    {
        const u32x4 *p = wp + chunk * (2 * NT * 64) + (wave * (NT / 2)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) r[i] = p[i * 64];
    }
    __device__ __forceinline__ void store(u32x4 *slot, int wave, int lane) const
    {
        u32x4 *p = slot + (wave * (NT / 2)) * 64 + lane;
#pragma unroll
        for (int i = 0; i < NT / 2; ++i) p[i * 64] = r[i];
    }
};

template <int NT>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NT], u32x4 a0, u32x4 a1, const u32x4 *slot,
                                          int lane)
{
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
            acc[tl] = mfma_bf16(s == 0 ? a0 : a1, slot[(s * NT + tl) * 64 + lane], acc[tl]);
}

// ---------------------------------------------------------------------------------------
// k_joint_fwd_bf16: logits[c, :] = hidden[c, :] . W^T + bias   (fp32 out)
// grid = rows_alloc/128 workgroups of 256 threads, 2 per CU (128 accumulator registers);
// pass = 256 columns (8 tiles), chunks run linearly over (pass, c): the staging pipeline
// never drains at a pass boundary.  A fragments: 32 B per lane per chunk straight from the
// row-major hidden (ring of 4 chunks); re-read from L1/L2 on every pass.  Requires H % 128 == 0.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_joint_fwd_bf16(Bf16Args a)
{
    constexpr int NT = 8;
    __shared__ __attribute__((aligned(16))) u32x4 s_b[2][2 * NT * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, half = lane >> 5;
    const int H = a.H, V = a.V, KC = H / 32;
    const int npass = (V + 255) / 256;
    const long NC = (long)npass * KC;
    const long row0 = (long)blockIdx.x * 128 + wave * 32;
    const u32x4 *ap = (const u32x4 *)(a.hidden + (row0 + j) * H) + 2 * half;  // chunk c: ap[4c], ap[4c+1]
    const u32x4 *wp = (const u32x4 *)a.wpack_fwd;

    f32x16 acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;

    u32x4 ar[4][2];
    BStage<NT> bx, by;  // even / odd chunks
#pragma unroll
    for (int q = 0; q < 3; ++q) { ar[q][0] = ap[4 * (q % KC)]; ar[q][1] = ap[4 * (q % KC) + 1]; }
    bx.load(wp, 0, wave, lane);
    by.load(wp, NC > 1 ? 1 : 0, wave, lane);
    bx.store(s_b[0], wave, lane);
    bx.load(wp, NC > 2 ? 2 : NC - 1, wave, lane);
    lds_barrier();

    int c = 0, pass = 0;
    for (long cc = 0; cc < NC; cc += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // stage chunk cc+q+1 into the other slot, fetch chunk cc+q+3
            const long nxt = cc + q + 3 < NC ? cc + q + 3 : NC - 1;
            if (q & 1) { bx.store(s_b[0], wave, lane); bx.load(wp, nxt, wave, lane); }
            else       { by.store(s_b[1], wave, lane); by.load(wp, nxt, wave, lane); }
            // A fragments of chunk cc+q+3 (same rows, k wraps into the next pass)
            {
                int c3 = c + q + 3; if (c3 >= KC) c3 -= KC;
                ar[(q + 3) & 3][0] = ap[4 * c3];
                ar[(q + 3) & 3][1] = ap[4 * c3 + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk<NT>(acc, ar[q][0], ar[q][1], s_b[q & 1], lane);
            if (q == 3 && c + 4 == KC) {  // pass complete (KC % 4 == 0): bias, store, restart
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int col0 = 256 * pass + 128 * g + 4 * j;
                    if (col0 < V) {
                        const f32x4 b4 = *(const f32x4 *)(a.bias + col0);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long orow = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            f32x4 o = {acc[4 * g][r] + b4[0], acc[4 * g + 1][r] + b4[1],
                                       acc[4 * g + 2][r] + b4[2], acc[4 * g + 3][r] + b4[3]};
                            *(f32x4 *)(a.logits + orow * V + col0) = o;
                        }
                    }
                }
#pragma unroll
                for (int tl = 0; tl < NT; ++tl)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;
            }
            lds_barrier();
        }
        c += 4;
        if (c == KC) { c = 0; ++pass; }
    }
}

void launch_joint_fwd_bf16(const Bf16Args &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_joint_fwd_bf16, dim3((unsigned)(a.rows_alloc / 128)), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------
// k_dhidden_bf16: G from logits (one thread per cell x 16 vocabulary entries per chunk), G
// stored as bf16 over the first half of its logits row, dHidden = G . W accumulated over all
// H <= 512 columns (16 tiles, 256 accumulator registers, one workgroup per CU), epilogue as
// the fp32 kernel: x (1 - hidden^2), sum over u -> dEnc slab, sum over t -> dPred slab.
// Tile = 8 t x 16 u cells; wave w owns t-rows 2w, 2w+1.  grid (n_ublk, ceil(T/8), B).
// Requires V % 128 == 0, H % 128 == 0, H <= 512.
// ---------------------------------------------------------------------------------------
#define BG_BT 8
#define BG_BU 16
__global__ __launch_bounds__(256, 1) void k_dhidden_bf16(Bf16Args a)
{
    constexpr int NT = 16;
    __shared__ __attribute__((aligned(16))) u32x4 s_b[2][2 * NT * 64];  // 64 KiB; reused by the epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int ub = blockIdx.x, tt = blockIdx.y, b = blockIdx.z;
    const int Tb = a.logit_lens[b];
    const int t0 = tt * BG_BT, u0 = ub * BG_BU;
    const int VC = V / 32;

    // this lane's producer row: cell (pt, pu) or none
    const int pt = t0 + 2 * wave + (j >> 4), pu = u0 + (j & 15);
    const bool pexists = pt < T && pu < U1;
    const long zrow = (long)a.B * T * U1;  // first zero padding row
    const long pcell = pexists ? ((long)b * T + pt) * U1 + pu : zrow;
    float *lrow = a.logits + pcell * V;
    u32x4 *grow = (u32x4 *)lrow + 2 * half;  // chunk c: grow[4c], grow[4c+1]  (32 B of bf16)

    if (t0 >= Tb) {  // workgroup-uniform: no products, but k_dw_bf16 must find zeros here
        if (pexists) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int c = 0; c < VC; ++c) { grow[4 * c] = z; grow[4 * c + 1] = z; }
        }
        return;
    }

    CellCoef cf = a.coef[pexists ? pcell : 0];
    const bool live = pexists && pt < Tb && cf.c1 != RNNT_NEG_INF;
    if (!live) { cf.c1 = RNNT_NEG_INF; cf.sb = 0.f; cf.se = 0.f; cf.y = -1; }
    // rows outside the lattice read the zero padding row (finite) with c1 = -inf -> G = 0
    const f32x4 *xsrc = (const f32x4 *)(live ? lrow : a.logits + zrow * V) + 4 * half;  // chunk c: xsrc[8c .. 8c+3]
    const int blank = a.blank;
    const u32x4 *wp = (const u32x4 *)a.wpack_dh;

    f32x16 acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;

    f32x4 xr[4][4];  // logits ring, 4 chunks ahead (slot = chunk & 3)
    BStage<NT> bx, by;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[q][i] = xsrc[8 * (q < VC ? q : VC - 1) + i];
    bx.load(wp, 0, wave, lane);
    by.load(wp, 1, wave, lane);
    bx.store(s_b[0], wave, lane);
    bx.load(wp, 2, wave, lane);
    lds_barrier();

    for (int c0 = 0; c0 < VC; c0 += 4) {  // VC % 4 == 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q;
            const int nxt = c + 3 < VC ? c + 3 : VC - 1;
            if (q & 1) { bx.store(s_b[0], wave, lane); bx.load(wp, nxt, wave, lane); }
            else       { by.store(s_b[1], wave, lane); by.load(wp, nxt, wave, lane); }
            // ---- G of chunk c for this lane's 16 vocabulary entries v = 32c + 16*half + e
            float g[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    g[4 * i + e] = __builtin_amdgcn_exp2f(fmaf(xr[q][i][e], RNNT_LOG2E, cf.c1));
            const int vb = 32 * c + 16 * half;
            const unsigned dy = (unsigned)(cf.y - vb);
            if (__any(dy < 16u)) {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (dy == (unsigned)e) g[e] -= cf.se;
            }
            if ((unsigned)(blank - 32 * c) < 32u) {  // wave-uniform
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (vb + e == blank) g[e] -= cf.sb;
            }
            {
                const int c4 = c + 4 < VC ? c + 4 : VC - 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[q][i] = xsrc[8 * c4 + i];
            }
            u32x4 a0, a1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = pack_bf16(g[2 * e], g[2 * e + 1]);
                a1[e] = pack_bf16(g[8 + 2 * e], g[8 + 2 * e + 1]);
            }
            if (pexists) { grow[4 * c] = a0; grow[4 * c + 1] = a1; }
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk<NT>(acc, a0, a1, s_b[q & 1], lane);
            lds_barrier();
        }
    }

    // ---- epilogue.  C layout: accumulator register r of tile 4g+q holds row
    // (r&3) + 8*(r>>2) + 4*half of the wave's 32 rows = (t-row r>>3, u (r&3)+8*((r>>2)&1)+4*half),
    // column 128g + 4j + q.
    float *s_red = (float *)s_b;  // [4 waves][64 lanes][33]
    const long BTH = (long)a.B * T * H, BUH = (long)a.B * U1 * H;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col0 = 128 * g + 4 * j;
        const bool colok = col0 < H;
        float psum[8][4];
#pragma unroll
        for (int r7 = 0; r7 < 8; ++r7)
#pragma unroll
            for (int q = 0; q < 4; ++q) psum[r7][q] = 0.f;
#pragma unroll
        for (int tl_ = 0; tl_ < 2; ++tl_) {
            const int t = t0 + 2 * wave + tl_;
            const bool tok = t < Tb;
            float esum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r7 = 0; r7 < 8; ++r7) {
                const int u = u0 + (r7 & 3) + 8 * (r7 >> 2) + 4 * half;
                const bool ok = tok && u < U1 && colok;
                u32x2 h2 = {0u, 0u};
                if (ok) h2 = *(const u32x2 *)(a.hidden + (((long)b * T + t) * U1 + u) * H + col0);
                const float hv[4] = {bf16_lo(h2[0]), bf16_hi(h2[0]), bf16_lo(h2[1]), bf16_hi(h2[1])};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float av = acc[4 * g + q][tl_ * 8 + r7];
                    const float d = ok ? av * (1.f - hv[q] * hv[q]) : 0.f;
                    esum[q] += d;
                    psum[r7][q] += d;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) esum[q] += __shfl_xor(esum[q], 32, 64);
            if (half == 0 && tok && colok) {
                f32x4 o = {esum[0], esum[1], esum[2], esum[3]};
                *(f32x4 *)(a.slab_enc + (long)ub * BTH + ((long)b * T + t) * H + col0) = o;
            }
        }
        __syncthreads();  // previous g's readers are done with s_red
#pragma unroll
        for (int r7 = 0; r7 < 8; ++r7)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_red[(wave * 64 + lane) * 33 + r7 * 4 + q] = psum[r7][q];
        __syncthreads();
        // thread (wave, lane) sums rows r7 = 2*wave, 2*wave+1 of source lane `lane` over the 4 waves
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r7 = 2 * wave + k;
            const int u = u0 + (r7 & 3) + 8 * (r7 >> 2) + 4 * half;
            if (u < U1 && colok) {
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    o[q] = (s_red[(0 * 64 + lane) * 33 + r7 * 4 + q] + s_red[(1 * 64 + lane) * 33 + r7 * 4 + q]) +
                           (s_red[(2 * 64 + lane) * 33 + r7 * 4 + q] + s_red[(3 * 64 + lane) * 33 + r7 * 4 + q]);
                *(f32x4 *)(a.slab_pred + (long)tt * BUH + ((long)b * U1 + u) * H + col0) = o;
            }
        }
    }
}

void launch_dhidden_bf16(const Bf16Args &a, hipStream_t st)
{
    const long cells = (long)a.B * a.T * a.U1;
    // zero padding rows: G of rows k_dw_bf16 walks past the last cell, and the "dead row" source
    (void)hipMemsetAsync(a.logits + cells * a.V, 0, (size_t)(a.rows_alloc - cells) * a.V * 4, st);
    dim3 grid(a.n_ublk, (a.T + BG_BT - 1) / BG_BT, a.B);
    hipLaunchKernelGGL(k_dhidden_bf16, grid, dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------
// k_dw_bf16: dW[v,h] = sum_c G[c,v] hidden[c,h] (split-K slabs), db[v] = sum_c G[c,v].
// 4 waves = 2 (M) x 2 (N), workgroup tile 256 v x 256 h, wave 128 x 128 = 16 tiles.  Both
// operands are row-major with K (the cell) as the ROW, the MFMA wants 8 consecutive k per
// lane: a lane loads, for its 8 cells, 4 adjacent columns (8 B) and transposes the 8 x 4
// block in registers with 16 v_perm_b32 — the 4 columns are its 4 interleaved tiles.
// The bias gradient rides the matrix pipe: one extra B fragment of ones (column 0).
// ---------------------------------------------------------------------------------------
#define BW_RING 4  // k-steps (16 cells each) of raw operands in flight per wave

__device__ __forceinline__ void transpose8x4(const u32x2 (&d)[8], u32x4 (&f)[4])
{
    // d[i] = cell i: {col0,col1}, {col2,col3};  f[q] = column q: cells {0,1},{2,3},{4,5},{6,7}
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f[0][p] = __builtin_amdgcn_perm(d[2 * p + 1][0], d[2 * p][0], 0x05040100u);
        f[1][p] = __builtin_amdgcn_perm(d[2 * p + 1][0], d[2 * p][0], 0x07060302u);
        f[2][p] = __builtin_amdgcn_perm(d[2 * p + 1][1], d[2 * p][1], 0x05040100u);
        f[3][p] = __builtin_amdgcn_perm(d[2 * p + 1][1], d[2 * p][1], 0x07060302u);
    }
}

__global__ __launch_bounds__(256, 1) void k_dw_bf16(Bf16Args a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    const int H = a.H, V = a.V;
    const int n_vblk = (V + 255) / 256, n_hblk = (H + 255) / 256;
    const int tiles = n_vblk * n_hblk;
    const int total = tiles * a.n_split;
    int id = blockIdx.x;  // XCD-aware remap: the tiles of one split share an XCD's L2
    {
        const int q8 = total / 8, r8 = total % 8, x = id % 8;
        id = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + id / 8;
    }
    const int tile = id % tiles, split = id / tiles;
    const int vb = tile / n_hblk, hb = tile % n_hblk;
    const int v0 = vb * 256 + wm * 128, h0 = hb * 256 + wn * 128;
    const int vbase = v0 + 4 * i, hbase = h0 + 4 * i;
    const bool vok = vbase < V, hok = hbase < H;
    const long nchunk = a.rows_pad / 16;
    const long k_lo = nchunk * split / a.n_split, k_hi = nchunk * (split + 1) / a.n_split;
    const long nstep = k_hi - k_lo;  // k-steps of 16 cells

    f32x16 acc[4][4];
    f32x16 accb[2];
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accb[0][r] = 0.f; accb[1][r] = 0.f; }
    const bool do_b = hb == 0;  // workgroup-uniform: these workgroups also produce db
    const unsigned one2 = i == 0 ? 0x3f803f80u : 0u;
    const u32x4 ones = {one2, one2, one2, one2};

    if (nstep > 0) {
        // G rows are 4V bytes apart (bf16 in the first half of the fp32 logits row)
        const char *gp = (const char *)a.logits + (k_lo * 16 + 8 * half) * (long)V * 4 + (vok ? vbase : V - 4) * 2L;
        const char *hp = (const char *)a.hidden + (k_lo * 16 + 8 * half) * (long)H * 2 + (hok ? hbase : H - 4) * 2L;
        const long grow = 4L * V, hrow = 2L * H;
        u32x2 ra[BW_RING][8], rb[BW_RING][8];
#pragma unroll
        for (int s_ = 0; s_ < BW_RING; ++s_) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ra[s_][e] = *(const u32x2 *)(gp + e * grow);
                rb[s_][e] = *(const u32x2 *)(hp + e * hrow);
            }
            gp += 16 * grow;
            hp += 16 * hrow;
        }
        // the buffers carry >= 16*BW_RING zero rows past rows_pad, so the ring may overrun
        for (long st = 0; st < nstep; st += BW_RING) {
#pragma unroll
            for (int s_ = 0; s_ < BW_RING; ++s_) {
                if (st + s_ < nstep) {  // workgroup-uniform
                    u32x4 fa[4], fb[4];
                    transpose8x4(ra[s_], fa);
                    transpose8x4(rb[s_], fb);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        ra[s_][e] = *(const u32x2 *)(gp + e * grow);
                        rb[s_][e] = *(const u32x2 *)(hp + e * hrow);
                    }
                    gp += 16 * grow;
                    hp += 16 * hrow;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
                        for (int qn = 0; qn < 4; ++qn)
                            acc[qm][qn] = mfma_bf16(fa[qm], fb[qn], acc[qm][qn]);
                    if (do_b) {
                        accb[0] = mfma_bf16(wn == 0 ? fa[0] : fa[2], ones, accb[0]);
                        accb[1] = mfma_bf16(wn == 0 ? fa[1] : fa[3], ones, accb[1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

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
    if (do_b && i == 0) {  // column 0 of the ones product: lanes 0 and 32
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int qm = 2 * wn + k;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int v = v0 + 4 * row + qm;
                if (v < V) a.slab_b[(long)split * V + v] = accb[k][r];
            }
        }
    }
}

void launch_dw_bf16(const Bf16Args &a, hipStream_t st)
{
    const int tiles = ((a.V + 255) / 256) * ((a.H + 255) / 256);
    hipLaunchKernelGGL(k_dw_bf16, dim3(tiles * a.n_split), dim3(256), 0, st, a);
}
