// bf16.hip — the RNNT_DTYPE_BF16 route of the fused pipeline (BASELINE config 3).
//
// Same path as the fp32 route (reference rnnt/joint.py:32-39 + torchaudio rnnt_loss called at
// rnnt/model.py:35-41 + their autograd), but the operands of the three GEMMs are rounded to
// bf16 (round-to-nearest-even) and multiplied by v_mfma_f32_32x32x16_bf16 with fp32
// accumulation — 16x the fp32 matrix rate, so this route is HBM-bound:
//   hidden = bf16(tanh(enc+pred))        forward-tile prologue   writes  2H  B/cell
//   logits = f16(hidden . bf16(W)^T + b) k_joint_fwd_bf16        reads 2H, writes 2V B/cell
//   softmax statistics, lattice, coef    fp32 / fp64, shared with the fp32 route (lattice.hip)
//   G = bf16(exp2(logit*log2e+c1) - ..)  k_dhidden_bf16          reads 2V+2H, writes 2V B/cell
//   dHidden = G . bf16(W), x(1-h^2), sums                        (same kernel)
//   dW = G^T . hidden, db = colsum(G)    k_dw_bf16               reads 2V+2H B/cell
// The logits are STORED in fp16 (round-to-nearest-even; half the bytes of the two kernels that
// stream them) and everything downstream — softmax statistics, the blank/label log-probs, G —
// is computed from the stored values, so the route is exactly "fp16 logits": the accumulators
// are fp32, |logit| stays far below the fp16 range (hidden is in [-1,1]), and the rounding
// (2^-11 relative) is below the bf16 rounding of G it feeds.  G (bf16) overwrites its logits
// row in place, byte for byte.
//
// MFMA operand maps (cdna_hip_programming.md §3): lane l = (r = l&31, h = l>>5) holds
// A[row r][k = 8h+j] and B[k = 8h+j][col r], j = 0..7; C/D: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Two conventions used throughout:
//  * K permutation: a 32-wide k chunk is consumed by 2 MFMAs; the lane's 16 consecutive k
//    (32 contiguous bytes of bf16 or fp16) feed MFMA s=0 with its first 8 and
//    MFMA s=1 with its last 8: MFMA s, slot (h,j)  <->  k = 32c + 16h + 8s + j.  Both operands
//    use the same map, so the dot product is unchanged and every global access is >= 32 B/lane.
//  * Column interleave by 4: accumulator tile 4g+q holds columns 128g + 4*(l&31) + q, so a
//    lane's 4 tiles of a group are 4 adjacent columns -> 16-byte epilogue accesses.
// W is re-packed once per call (2 x 1 MB at H=512,V=1024) into exactly the order the B
// fragments are consumed, so staging a chunk is a linear copy L2 -> VGPR -> LDS.
#include "kernels.hpp"

// Diagnostic build only (-DBF_CLOCK, with an engine.o built -DRNNT_STAMPS so that Bf16Args::debug is set; tools/bf16_whatif.sh): workgroup
// (0,0,0) stamps the core clock counter and the 100 MHz reference at its start and end into debug[SLOT .. SLOT+3] (a buffer nothing
// else reads): the clock the launch ran at = d(s_memtime) / d(s_memrealtime) x 100 MHz (tools/exp_bf16_clock.py).
#ifdef BF_CLOCK
#define BF_CLOCK_STAMP(SLOT)                                                                                               \
    do {                                                                                                                   \
        if (a.debug && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {                        \
            unsigned long long t_, r_;                                                                                     \
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");      \
            a.debug[SLOT] = t_; a.debug[(SLOT) + 1] = r_;                                                                  \
        }                                                                                                                  \
    } while (0)
#else
#define BF_CLOCK_STAMP(SLOT) do {} while (0)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) i16x4 *lds_i16x4_ptr;
typedef __attribute__((address_space(3))) void *lds_vptr;

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned, v);
}
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ unsigned pack_f16(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));  // v_cvt_pk_f16_f32 (RNE)
}
__device__ __forceinline__ unsigned pk_max_f16(unsigned x, unsigned y)  // max of two pairs of halves
{
    unsigned d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
    return d;
}
__device__ __forceinline__ unsigned max_halves_f16(unsigned x)  // low half of the result = max(lo, hi) of x
{
    unsigned d;
    asm("v_max_f16_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "=v"(d) : "v"(x));
    return d;
}
__device__ __forceinline__ float f16_lo(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
__device__ __forceinline__ float f16_hi(unsigned u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
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
// hidden[c,:] = bf16(tanh(enc[b,t,:] + pred[b,u,:])) for the 128 cells of one forward tile, produced
// by the tile's own workgroup (the separate 6.6 GB producer pass of round 1, 1.6 ms at cfg2, is
// gone: its loads, tanh and stores now run beside the co-resident workgroup's main loop).  A thread
// owns 8 columns (16 B of bf16) of every (256 / (H/8))-th row; (b,t,u) is carried along, no division
// inside the loop.  EVERY cell of the tile gets a finite row, also past its utterance's length: the
// backward GEMMs multiply those rows by exact zeros.  Rows past the last cell are zeroed by the host.
__device__ __forceinline__ void make_hidden_tile_bf16(const Bf16Args &a, long c_first, int tid)
{
    const int H = a.H, H8 = H / 8, U1 = a.U1, T = a.T;
    const long cells = (long)a.B * T * U1;
    const int rstep = 256 / H8 > 0 ? 256 / H8 : 1;      // rows per sweep (H <= 2048)
    const int r0 = tid / H8, h = (tid - r0 * H8) * 8;
    if (r0 >= rstep) return;                             // H8 not a divisor of 256: idle threads
    long c = c_first + r0;
    int u = (int)(c % U1);
    long bt = c / U1;
    int t = (int)(bt % T), b = (int)(bt / T);
    u32x4 *hid = (u32x4 *)a.hidden;
    for (int r = r0; r < 128 && c < cells; r += rstep, c += rstep) {
        const float *ep = a.enc + (long)b * a.enc_sb + (long)t * a.enc_st + h;
        const float *pp = a.pred + ((long)b * U1 + u) * H + h;
        const f32x4 e0 = *(const f32x4 *)ep, e1 = *(const f32x4 *)(ep + 4);
        const f32x4 p0 = *(const f32x4 *)pp, p1 = *(const f32x4 *)(pp + 4);
        u32x4 o;
        const f32x4 t0 = fast_tanh_sum4(e0, p0), t1 = fast_tanh_sum4(e1, p1);
        o[0] = pack_bf16(t0[0], t0[1]);
        o[1] = pack_bf16(t0[2], t0[3]);
        o[2] = pack_bf16(t1[0], t1[1]);
        o[3] = pack_bf16(t1[2], t1[3]);
        hid[c * H8 + h / 8] = o;
        u += rstep;
        while (u >= U1) { u -= U1; if (++t == T) { t = 0; ++b; } }
    }
}

// forward B operand, fragment order: [pass][c][s][tile(8)][lane] x 8 bf16,
// element j = W[v = 256*pass + 8*(lane&31) + tile][h = 32c + 16*(lane>>5) + 8s + j]
// (columns interleaved by 8 here: a lane's 8 accumulator tiles are 8 adjacent logits = one
// 16-byte fp16 store)
__global__ __launch_bounds__(256) void k_pack_w_fwd_bf16(const float *__restrict__ W,
                                                         u32x4 *__restrict__ out, int H, int V,
                                                         int KC, long n)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63), tile = (int)(idx >> 6) & 7, s = (int)(idx >> 9) & 1;
    const long cc = idx >> 10;
    const int c = (int)(cc % KC), pass = (int)(cc / KC);
    const int v = 256 * pass + 8 * (lane & 31) + tile;
    const int h0 = 32 * c + 16 * (lane >> 5) + 8 * s;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (v < V) {
        const float *w = W + (long)v * H + h0;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf16(w[2 * j], w[2 * j + 1]);
    }
    out[idx] = o;
}

// dHidden B operand, fragment order: [hp][c][s][tile(16)][lane] x 8 bf16 (hp = 512-column pass),
// element j = W[v = 32c + 16*(lane>>5) + 8s + j][h = 512hp + 128*(tile>>2) + 4*(lane&31) + (tile&3)]
__global__ __launch_bounds__(256) void k_pack_w_dh_bf16(const float *__restrict__ W,
                                                        u32x4 *__restrict__ out, int H, int V,
                                                        long n)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63), tile = (int)(idx >> 6) & 15, s = (int)(idx >> 10) & 1;
    const int VC = V / 32;
    const int c = (int)((idx >> 11) % VC), hp = (int)((idx >> 11) / VC);
    const int h = 512 * hp + 128 * (tile >> 2) + 4 * (lane & 31) + (tile & 3);
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
size_t bf16_wpack_dh_bytes(int H, int V) { return (size_t)((H + 511) / 512) * (V / 32) * 2 * 16 * 64 * 16; }

static void launch_pack_w_fwd_bf16(const Bf16Args &a, hipStream_t st);  // the layout the chosen forward kernel reads
void launch_bf16_producers(const Bf16Args &a, hipStream_t st)
{
    // hidden is produced by the forward kernel's tiles; only the zero padding rows past the last cell
    // (the dW DMA ring walks them) are written here
    const long cells = (long)a.B * a.T * a.U1;
    launch_fill32(a.hidden + cells * a.H, 0u, (size_t)(a.rows_alloc - cells) * a.H * 2, st);
    launch_pack_w_fwd_bf16(a, st);
    const long nd = (long)(bf16_wpack_dh_bytes(a.H, a.V) / 16);
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
    __device__ __forceinline__ void load(const u32x4 *wp, long chunk, int wave, int lane, bool on = true)
    {
        if (!on) return;  // experiment switch
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

// 2*NT MFMAs of one chunk; B fragments LDS -> VGPR two reads ahead of their MFMA (explicit
// software pipeline: left alone, hipcc serialises ds_read -> s_waitcnt -> v_mfma).
template <int NT, int DEPTH>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NT], u32x4 a0, u32x4 a1, const u32x4 *slot,
                                          int lane)
{
    const u32x4 *p = slot + lane;
    u32x4 b[DEPTH + 1];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) b[i] = p[i * 64];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
        if (i + DEPTH < 2 * NT) b[(i + DEPTH) % (DEPTH + 1)] = p[(i + DEPTH) * 64];
        acc[i % NT] = mfma_bf16(i < NT ? a0 : a1, b[i % (DEPTH + 1)], acc[i % NT]);
    }
    // pin the interleave: DEPTH reads up front, then (1 MFMA, 1 read) pairs
    __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
#pragma unroll
    for (int i = 0; i < 2 * NT - DEPTH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, DEPTH, 0);
}

// ---------------------------------------------------------------------------------------
// k_joint_fwd_bf16: logits[c, :] = f16(hidden[c, :] . W^T + bias)
// grid = rows_alloc/128 workgroups of 256 threads, 2 per CU (128 accumulator registers);
// pass = 256 columns (8 tiles), chunks run linearly over (pass, c): the staging pipeline
// never drains at a pass boundary.  A fragments: 32 B per lane per chunk straight from the
// row-major hidden (ring of 4 chunks); re-read from L1/L2 on every pass.  Requires H % 128 == 0.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_joint_fwd_bf16(Bf16Args a)
{
    constexpr int NT = 8;
    __shared__ __attribute__((aligned(16))) u32x4 s_b[2][2 * NT * 64];
    __shared__ float s_den[128];
    // per-lane running (max, sum exp) of each of its 16 row-slots, parked in LDS between the
    // pass epilogues (32 registers the main loop needs for its fragment pipeline)
    __shared__ float s_stat[2][16][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, half = lane >> 5;
    const int H = a.H, V = a.V, KC = H / 32;
    const int npass = (V + 255) / 256;
    const long NC = (long)npass * KC;
    {   // tiles with no live cell: the row padding past the last cell, and tiles that lie entirely
        // in the dead time steps (t >= T_b) of one utterance — their logits are never read
        // (k_dhidden_bf16 zero-fills the G rows of dead tiles itself)
        const long cells = (long)a.B * a.T * a.U1, per = (long)a.T * a.U1;
        const long c_first = (long)blockIdx.x * 128, c_last = c_first + 127;
        if (c_first >= cells) return;
        make_hidden_tile_bf16(a, c_first, tid);  // also for dead tiles: hidden must be finite everywhere
        const long b_first = c_first / per;
        if (c_last < cells && c_last / per == b_first &&
            (c_first - b_first * per) / a.U1 >= len_t(a.logit_lens, b_first, a.T)) return;
    }
    __syncthreads();  // the tile's hidden rows are stored (vmcnt(0)) and every wave is past them
    const long row0 = (long)blockIdx.x * 128 + wave * 32;
    const u32x4 *ap = (const u32x4 *)(a.hidden + (row0 + j) * H) + 2 * half;  // chunk c: ap[4c], ap[4c+1]
    const u32x4 *wp = (const u32x4 *)a.wpack_fwd;

    // the accumulators start from the bias of the pass's columns (this lane's 8 adjacent ones): the pass
    // epilogue then has no bias add — VALU instructions share the issue port with the MFMA stream of the
    // CU's other workgroup, every one of them counts
    f32x16 acc[NT];
    {
        const int c0 = 8 * j < V ? 8 * j : 0;  // pass 0 (V = 128: the upper half of the pass has no columns)
        const f32x4 b0 = *(const f32x4 *)(a.bias + c0), b1 = *(const f32x4 *)(a.bias + c0 + 4);
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tl][r] = tl < 4 ? b0[tl] : b1[tl - 4];
    }

#pragma unroll
    for (int r = 0; r < 16; ++r) { s_stat[0][r][tid] = RNNT_NEG_INF; s_stat[1][r][tid] = 0.f; }

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
            if (q & 1) { bx.store(s_b[0], wave, lane); bx.load(wp, nxt, wave, lane, !RNNT_XP(a.flags, 2048)); }
            else       { by.store(s_b[1], wave, lane); by.load(wp, nxt, wave, lane, !RNNT_XP(a.flags, 2048)); }
            // A fragments of chunk cc+q+3 (same rows, k wraps into the next pass)
            if (!RNNT_XP(a.flags, 4096)) {
                int c3 = c + q + 3; if (c3 >= KC) c3 -= KC;
                ar[(q + 3) & 3][0] = ap[4 * c3];
                ar[(q + 3) & 3][1] = ap[4 * c3 + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!RNNT_XP(a.flags, 1024)) mma_chunk<NT, 5>(acc, ar[q][0], ar[q][1], s_b[q & 1], lane);
            if (q == 3 && c + 4 == KC) {  // pass complete (KC % 4 == 0): bias, store, softmax statistics
                const int col0 = 256 * pass + 8 * j;  // this lane's 8 adjacent columns
                const bool cok = col0 < V;            // V % 128 == 0: the last pass may be half empty
                // bias of the NEXT pass's columns, requested before the row loop, used after it
                const int coln = col0 + 256 < V ? col0 + 256 : 0;
                const f32x4 b0 = *(const f32x4 *)(a.bias + coln);
                const f32x4 b1 = *(const f32x4 *)(a.bias + coln + 4);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long orow = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    // fp16 logits: round, store, and take the statistics from the rounded values
                    const u32x4 qv = {pack_f16(acc[0][r], acc[1][r]), pack_f16(acc[2][r], acc[3][r]),
                                      pack_f16(acc[4][r], acc[5][r]), pack_f16(acc[6][r], acc[7][r])};
                    // streaming stores: the logits are not re-read by this kernel's CUs and should
                    // not evict the A rows the next pass re-reads from L2
                    if (cok && !RNNT_XP(a.flags, 256)) __builtin_nontemporal_store(qv, (u32x4 *)(a.logits + orow * V + col0));
                    const f32x4 o0 = {f16_lo(qv[0]), f16_hi(qv[0]), f16_lo(qv[1]), f16_hi(qv[1])};
                    const f32x4 o1 = {f16_lo(qv[2]), f16_hi(qv[2]), f16_lo(qv[3]), f16_hi(qv[3])};
                    if (RNNT_XP(a.flags, 512) || !cok) continue;  // columns past V (zero weights) are no logits
                    // running (max, sum exp) of this lane's columns of row-slot r.  The max of the 8 rounded
                    // values is taken on the packed halves (3 v_pk_max_f16 + 1 + one conversion instead of 8
                    // conversions + 7 max; the exp arguments below read the halves through v_fma_mix_f32)
                    // (spelled as asm: __builtin_elementwise_max on the half pairs was reduced by hipcc 7.2 to
                    // the max of the first pair only)
                    const float lmax = f16_lo(max_halves_f16(pk_max_f16(pk_max_f16(qv[0], qv[1]), pk_max_f16(qv[2], qv[3]))));
                    const float m_old = s_stat[0][r][tid];
                    float mn;  // fmaxf would first canonicalise the LDS value (one more instruction); -inf / finite only
                    asm("v_max_f32 %0, %1, %2" : "=v"(mn) : "v"(m_old), "v"(lmax));
                    const float nm2 = -mn * RNNT_LOG2E;
                    float e = s_stat[1][r][tid] * __builtin_amdgcn_exp2f(fmaf(m_old, RNNT_LOG2E, nm2));
                    e += (__builtin_amdgcn_exp2f(fmaf(o0[0], RNNT_LOG2E, nm2)) +
                          __builtin_amdgcn_exp2f(fmaf(o0[1], RNNT_LOG2E, nm2))) +
                         (__builtin_amdgcn_exp2f(fmaf(o0[2], RNNT_LOG2E, nm2)) +
                          __builtin_amdgcn_exp2f(fmaf(o0[3], RNNT_LOG2E, nm2)));
                    e += (__builtin_amdgcn_exp2f(fmaf(o1[0], RNNT_LOG2E, nm2)) +
                              __builtin_amdgcn_exp2f(fmaf(o1[1], RNNT_LOG2E, nm2))) +
                             (__builtin_amdgcn_exp2f(fmaf(o1[2], RNNT_LOG2E, nm2)) +
                              __builtin_amdgcn_exp2f(fmaf(o1[3], RNNT_LOG2E, nm2)));
                    s_stat[1][r][tid] = e;
                    s_stat[0][r][tid] = mn;
                }
#pragma unroll
                for (int tl = 0; tl < NT; ++tl)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tl][r] = tl < 4 ? b0[tl] : b1[tl - 4];
            }
            lds_barrier();
        }
        c += 4;
        if (c == KC) { c = 0; ++pass; }
    }

    // ---- log-softmax denominators and the two log-probs each lattice cell needs.
    // Row-slot r of half h is row (r&3) + 8*(r>>2) + 4h of the wave's 32; its 32 lanes combine.
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        // on the DPP crossbar (as the fp32 forward): the ds_bpermute butterflies were 160 LDS round trips
        const float m_l = s_stat[0][r][tid];
        const float M = half_max_dpp(m_l, half);
        const float S = half_sum_dpp(s_stat[1][r][tid] * __builtin_amdgcn_exp2f((m_l - M) * RNNT_LOG2E), half);
        if (j == 31) s_den[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = M + __logf(S);  // lanes 31 / 63 hold the sums
    }
    // lane L: row L&31; lanes 0-31 fetch logit[blank], lanes 32-63 logit[label].  Both were
    // stored by THIS wave: wait for the stores, then read through L2 (agent-scope loads bypass
    // the CU's vector L1).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long cell = row0 + j;
    if (cell < (long)a.B * a.T * a.U1) {
        const int U1 = a.U1, T = a.T;
        const int u = (int)(cell % U1);
        const long bt = cell / U1;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int Ub = len_u(a.target_lens, b, a.U1);
        if (t < len_t(a.logit_lens, b, a.T) && u <= Ub) {
            const float den = s_den[wave * 32 + j];
            const unsigned short *lrow = a.logits + cell * V;
            const long si = skew_index(b, t, u, a.D, U1);
            auto stored = [&](int v) {  // fp16 logit v of this row, read through L2
                const unsigned w = __hip_atomic_load((const unsigned *)(lrow + (v & ~1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return (v & 1) ? f16_hi(w) : f16_lo(w);
            };
            if (half == 0) {
                const float lb = stored(a.blank);
                a.denom_s[si] = den;
                a.lpb_s[si] = lb - den;
            } else {
                float le = 0.f;
                if (u < Ub) {
                    const int y = a.targets[(long)b * (U1 - 1) + u];
                    le = stored(y) - den;
                }
                a.lpe_s[si] = le;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// k_joint_fwd_bf16_ra<KC>: the forward for H = 64*KC with the tile's hidden rows held in REGISTERS (round 3).
// The kernel above re-fetches a tile's hidden rows on every column pass (V/256 times; counted: 2.9x the
// algorithmic traffic) and stages W through VGPRs.  Here a wave produces the hidden rows of its 32 cells straight
// into MFMA fragment registers (KC x 4 fragments of 16 B: H/4 registers per lane), stores them once for
// k_dw_bf16, and keeps them for all column passes: no hidden read at all, no HBM round trip between production
// and the products, and the only memory stream of the main loop is W by LDS-DMA.
//  * workgroup = 4 waves x 32 cells = 128 consecutive cells; two workgroups per CU up to H = 512 (128 fragment
//    registers + 64 accumulators per wave), one at H = 1024;
//  * the product is TRANSPOSED: logits^T tile = W tile (A operand, 32 vocabulary rows, from LDS) x hidden^T (B
//    operand, the wave's 32 cells, from registers), so a lane owns ONE cell (column l&31) and 16 vocabulary entries
//    per accumulator tile: the log-softmax statistics are lane-local (a running max and sum per lane, the two
//    halves of the wave combined once per tile) — no cross-lane reduction, no LDS exchange in the pass epilogue;
//  * k permutation: chunk c (64 deep), MFMA s (0..3), lane (r, h), element e  <->  k = 64c + 32h + 8s + e: a lane
//    owns 32 consecutive k of its cell per chunk;
//  * pass = 128 logits columns = 4 accumulator tiles; W row rho of tile t = column 128*pass + 32*(rho>>3) +
//    16*((rho>>2)&1) + 4t + (rho&3), so accumulator register 4g+i of tile t in lane (j, half) is column
//    128*pass + 32g + 16*half + 4t + i of cell j: per g a lane holds 16 adjacent logits (32 bytes) of its row and the
//    wave 64 contiguous bytes of each of its 32 rows.  Stored per lane these are 64 cache lines per instruction
//    (measured: the stores then clog the memory pipe, +1.9 ms); they go through 2 KiB of wave-private LDS instead
//    and leave row-major, 16 rows x 64 contiguous bytes per instruction;
//  * W chunk (pass, c) = 16 fragments [s][tile][lane] x 16 B = 16 KiB, 4-slot LDS ring filled by raw-buffer LDS-DMA
//    (4 per wave and chunk).  The barrier at the top of chunk n publishes chunk n+1, so the first fragment group
//    of a chunk is requested before the barrier that precedes it;
//  * bias: the accumulators start at zero (first MFMA of a pass takes C = 0) and the pass epilogue adds the
//    pass's 128 bias values from a double-buffered LDS table (broadcast reads).
// Requires V % 128 == 0.  Same rounding points as the kernel above (bf16 operands, fp32 accumulation, fp16 logits,
// statistics from the stored values); the k order inside an accumulator differs.
// ---------------------------------------------------------------------------------------
// Diagnostic builds only (-DFR_EXP=bits, tools/build_bf16_variants.sh): parts of k_joint_fwd_bf16_ra compiled out —
// 1 no MFMAs, 2 no statistics, 4 no logits stores, 8 no hidden stores, 16 no W DMA, 32 no tanh,
// 64 one workgroup per CU (100 KiB of LDS requested), 128 every W fragment feeds two MFMAs and half of W's DMA pieces are issued
#ifndef FR_EXP
#define FR_EXP 0
#endif
#define FR_OFF(bit) ((FR_EXP) & (bit))
#ifdef RNNT_STAMPS
// Diagnostic build only (-DRNNT_STAMPS): s_memtime stamps of workgroup FRS_BLOCK, every wave: debug[wave*128 + slot]
#ifndef FRS_BLOCK
#define FRS_BLOCK 20000
#endif
#define FRSTAMP(slot)                                                                       \
    do {                                                                                    \
        if (a.debug && blockIdx.x == FRS_BLOCK && lane == 0) {                              \
            unsigned long long t_;                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
            a.debug[wave * 128 + (slot)] = t_;                                              \
        }                                                                                   \
    } while (0)
#else
#define FRSTAMP(slot) do {} while (0)
#endif
#define FR_SLOT 16384
template <int N> struct BInt { static constexpr int value = N; };

// W in the order k_joint_fwd_bf16_ra consumes it: [pass][c][s][tile(4)][lane] x 8 bf16, element e =
// W[v = 128*pass + 32*(rho>>3) + 16*((rho>>2)&1) + 4*tile + (rho&3)][h = 64c + 32*(lane>>5) + 8s + e], rho = lane&31
__global__ __launch_bounds__(256) void k_pack_w_fwd_bf16_ra(const float *__restrict__ W, u32x4 *__restrict__ out, int H, long n)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int lane = (int)(idx & 63), tile = (int)(idx >> 6) & 3, s = (int)(idx >> 8) & 3;
    const long cc = idx >> 10;
    const int KC = H / 64;
    const int c = (int)(cc % KC), pass = (int)(cc / KC);
    const int rho = lane & 31;
    const int v = 128 * pass + 32 * (rho >> 3) + 16 * ((rho >> 2) & 1) + 4 * tile + (rho & 3);
    const float *w = W + (long)v * H + 64 * c + 32 * (lane >> 5) + 8 * s;
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16(w[2 * j], w[2 * j + 1]);
    out[idx] = o;
}

template <int KC>
__global__ __launch_bounds__(256, (KC <= 8 ? 2 : 1)) void k_joint_fwd_bf16_ra(Bf16Args a)
{
    // [0, 64 KiB): W ring, 4 slots (slot 3 doubles as the production's transposition space);  then s_bias[2][128];
    // then the logits staging space, 2 KiB per wave
    extern __shared__ __attribute__((aligned(1024))) char s_fr[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, half = lane >> 5;
    constexpr int H = 64 * KC;
    const int V = a.V, U1 = a.U1, T = a.T;
    const int npass = V / 128;
    const int NC = npass * KC;
    const long cells = (long)a.B * T * U1, per = (long)T * U1;
    const long c_first = (long)blockIdx.x * 128;
    if (c_first >= cells) return;
    BF_CLOCK_STAMP(300);
    // a tile entirely in the dead time steps (t >= T_b) of one utterance: hidden only (finite rows for k_dw_bf16);
    // its logits are never read (k_dhidden_bf16 zero-fills the G rows of dead tiles itself)
    bool dead;
    {
        const long c_last = c_first + 127, b_first = c_first / per;
        dead = c_last < cells && c_last / per == b_first && (c_first - b_first * per) / U1 >= len_t(a.logit_lens, (int)b_first, T);
    }
    const long row0 = c_first + wave * 32;
    FRSTAMP(0);
    const int lds0 = (int)(size_t)(lds_vptr)s_fr;
    const int bias_w = lds0 + 4 * FR_SLOT + 16 * j;  // table write: lanes (j, .) hold columns 4j .. 4j+3 of a pass
    {   // pass 0's bias (every wave writes the same 512 bytes)
        const f32x4 bv = *(const f32x4 *)(a.bias + 4 * j);
        asm volatile("ds_write_b128 %0, %1" :: "v"(bias_w), "v"(bv) : "memory");
    }

    // ---- W ring: fragment 4*wave + q of chunk n -> slot n & 3
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(a.wpack_fwd, 0, (int)((long)V * H * 2), 0x00020000);
    const int wvo = wave * 4096 + lane * 16;
    auto wdma = [&](int n, int q) {
        if (FR_OFF(16)) return;
        const int src = n < NC ? n : NC - 1;  // past the end: the last chunk again (landed, never read)
        // (round 5: the four pieces of a chunk on ONE LDS base (M0) and ONE scalar offset — the instruction's 12-bit immediate offset
        // advances the memory and the LDS address alike; an M0 write in front of every LDS-DMA cost the f16x2 forward 0.6 ms of 21.7)
        lds_vptr dst = (lds_vptr)(s_fr + (n & 3) * FR_SLOT + wave * 4096);
        if (q == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, wvo, src * FR_SLOT, 0, 0);
        if (q == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, wvo, src * FR_SLOT, 1024, 0);
        if (q == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, wvo, src * FR_SLOT, 2048, 0);
        if (q == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, wvo, src * FR_SLOT, 3072, 0);
    };
    if (!dead) {
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) wdma(n, q);
    }
    FRSTAMP(1);
    asm volatile("" ::: "memory");  // the production's loads stay behind the DMAs: their data implies the DMAs landed

    // ---- hidden = bf16(tanh(enc + pred)) of this lane's cell, in fragment order, kept in registers.
    // Loads and stores are row-major (16 lanes cover 64 consecutive k of a row: whole 128-byte lines; a lane that read
    // its own fragment slots instead touched 64 lines per instruction); the transposition into fragment order goes
    // through 4 KiB of LDS per wave (ring slot 3, idle until the main loop): lane (sub = lane>>4, kq = lane&15)
    // handles rows 4i + sub (i = 0..7), k = 64c + 4kq .. +3 of chunk c; 16-byte positions inside a row's 128 bytes are
    // XORed with (row>>1)&7 (conflict-free ds_read_b128 of the fragments: row j, bytes 64*half + 16s).
    u32x4 A[KC][4];
    {
        const int sub = lane >> 4, kq = lane & 15;
        const long rowc = row0 < cells ? row0 : cells - 1;  // rows past the lattice: the last cell again (same bits, same place)
        const int rmax = (int)(cells - 1 - rowc < 31 ? cells - 1 - rowc : 31);
        const int u0 = (int)(rowc % U1);
        const long bt0 = rowc / U1;
        const int t0 = (int)(bt0 % T), b0 = (int)(bt0 / T);
        unsigned eo[8], po[8];  // element offsets of this lane's 8 rows
        const float *ebase = a.enc + (long)b0 * a.enc_sb + 4 * kq;
        const float *pbase = a.pred + (long)b0 * U1 * H + 4 * kq;
        unsigned short *hbase = a.hidden + rowc * H + 4 * kq;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int r = 4 * i + sub;
            r = r < rmax ? r : rmax;
            int u = u0 + r, t = t0, db = 0;
            while (u >= U1) { u -= U1; if (++t == T) { t = 0; ++db; } }
            eo[i] = (unsigned)(db * a.enc_sb + (long)t * a.enc_st);
            po[i] = (unsigned)((db * U1 + u) * H);
        }
        const int tbase = lds0 + 3 * FR_SLOT + wave * 4096;
        const int trd = tbase + j * 128;  // fragment s: + ((4half + s) ^ ((j>>1)&7)) * 16
        // operands of a half batch st (chunk st>>1, rows i = 4(st&1) .. +3): 4 + 4 vectors; three buffers, loads issued
        // two half batches ahead of their use (the L2 round trip is longer than one half batch of arithmetic)
        f32x4 lb[3][8];
        auto ld = [&](f32x4 (&b)[8], int st) {
            const int c = st >> 1, i0 = 4 * (st & 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                b[k] = *(const f32x4 *)(ebase + eo[i0 + k] + 64 * c);
                b[4 + k] = *(const f32x4 *)(pbase + po[i0 + k] + 64 * c);
            }
        };
        ld(lb[0], 0);
        if (2 * KC > 1) ld(lb[1], 1);
#pragma unroll
        for (int st = 0; st < 2 * KC; ++st) {
            if (st + 2 < 2 * KC) ld(lb[(st + 2) % 3], st + 2);
            const f32x4(&b)[8] = lb[st % 3];
            const int c = st >> 1, i0 = 4 * (st & 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k;
                const f32x4 tv = FR_OFF(32) ? b[k] + b[4 + k] : fast_tanh_sum4(b[k], b[4 + k]);
                const u32x2 o = {pack_bf16(tv[0], tv[1]), pack_bf16(tv[2], tv[3])};
                const int row = 4 * i + sub;
                const int r = row < rmax ? row : rmax;
                if (!FR_OFF(8)) *(u32x2 *)(hbase + (unsigned)(r * H) + 64 * c) = o;
                const int wa = tbase + row * 128 + (((kq >> 1) ^ ((row >> 1) & 7)) << 4) + 8 * (kq & 1);
                asm volatile("ds_write_b64 %0, %1" :: "v"(wa), "v"(o) : "memory");
            }
            if (st & 1) {
                const int x = (j >> 1) & 7;
                asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(A[c][0]), "=&v"(A[c][1]), "=&v"(A[c][2]), "=&v"(A[c][3])
                             : "v"(trd + (((4 * half + 0) ^ x) << 4)), "v"(trd + (((4 * half + 1) ^ x) << 4)),
                               "v"(trd + (((4 * half + 2) ^ x) << 4)), "v"(trd + (((4 * half + 3) ^ x) << 4))
                             : "memory");
            }
        }
    }
    FRSTAMP(2);
    if (dead) return;

    const int rb = lds0 + lane * 16;  // fragment (s, tile) of slot k: rb + k*FR_SLOT + (4s + tile)*1024
    float st_m = RNNT_NEG_INF, st_s = 0.f;  // running (max, sum exp) of this lane's cell over its 64 columns of every pass

    f32x16 acc[4];
    u32x4 g0[4], g1[4];
    auto reads = [&](u32x4 (&g)[4], int base, auto s_c) {
        constexpr int S = decltype(s_c)::value;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g[0]) : "v"(base), "n"((4 * S + 0) * 1024));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g[1]) : "v"(base), "n"((4 * S + 1) * 1024));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g[2]) : "v"(base), "n"((4 * S + 2) * 1024));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(g[3]) : "v"(base), "n"((4 * S + 3) * 1024));
    };
    auto landed = [&](u32x4 (&g)[4], bool wait) {
        if (wait) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) :: "memory");
        else asm volatile("" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]));
    };
    // 4 MFMAs: W fragments g (A operand: 32 vocabulary rows each) x this wave's cells (B operand, registers)
    auto mma4 = [&](auto first_c, const u32x4 &hf, const u32x4 (&g)[4], int n, int q) {
        constexpr bool FIRST = decltype(first_c)::value != 0;  // first k-step of a pass: C = 0
        if (!FR_OFF(1)) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[0] = mfma_bf16(g[0], hf, FIRST ? z : acc[0]);
            acc[1] = mfma_bf16(g[1], hf, FIRST ? z : acc[1]);
        }
        wdma(n + 3, q);
        if (!FR_OFF(1)) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[2] = mfma_bf16(g[2], hf, FIRST ? z : acc[2]);
            acc[3] = mfma_bf16(g[3], hf, FIRST ? z : acc[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    if (FR_OFF(1)) {
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;
    }

    // chunks 0..2 have landed (the production consumed loads issued after their DMAs; stores may still fly)
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * KC < 63 ? 8 * KC : 63) : "memory");
    lds_barrier();
    FRSTAMP(3);
    reads(g0, rb, BInt<0>{});

    // logits staging: lane (j, half) writes its 32 bytes of a g group to row j (64 bytes per row), 16-byte chunk
    // 2*half + k at position chunk ^ ((j>>2)&3); lane l then reads chunk l&3 of rows (l>>2), 16 + (l>>2) and stores it
    const int stg = lds0 + 4 * FR_SLOT + 1024 + wave * 2048;
    for (int pass = 0; pass < npass; ++pass) {
        const int n0 = pass * KC;
        FRSTAMP(4 + 3 * pass);
        auto chunk = [&](auto c_c) {
            constexpr int c = decltype(c_c)::value;
            const int n = n0 + c;
            // this wave's share of chunk n+1 has landed.  vmcnt retires in order; younger than those DMAs are the 4
            // of chunk n+2 and, in the first two chunks of a pass, the 8 logits stores + the bias load of the epilogue
            // (the first pass's chunks 1 and 2 landed before the loop: no wait behind the hidden stores)
            if (FR_OFF(128)) { if (c < 2) { if (pass != 0) asm volatile(RNNT_VMCNT(10) ::: "memory"); } else asm volatile(RNNT_VMCNT(2) ::: "memory"); }
            else if (c < 2) { if (pass != 0) asm volatile(RNNT_VMCNT(12) ::: "memory"); }
            else asm volatile(RNNT_VMCNT(4) ::: "memory");
            if (pass == 3) FRSTAMP(64 + 3 * c);
            lds_barrier();  // publishes chunk n+1; every wave is past its reads of chunk n-1 (slot of chunk n+3)
            if (pass == 3) FRSTAMP(65 + 3 * c);
            landed(g0, false);
            const int sb = rb + ((n & 3) << 14), sbn = rb + (((n + 1) & 3) << 14);
            if (FR_OFF(128)) {  // what-if (wrong results): every W fragment feeds TWO MFMAs and half of W's bytes move — the price list of a 256-cell tile
                mma4(BInt<(c == 0)>{}, A[c][0], g0, n, 0);
                reads(g1, sb, BInt<2>{});
                mma4(BInt<0>{}, A[c][1], g0, n, 5);
                landed(g1, true);
                mma4(BInt<0>{}, A[c][2], g1, n, 2);
                reads(g0, sbn, BInt<0>{});
                mma4(BInt<0>{}, A[c][3], g1, n, 5);
            } else {
            reads(g1, sb, BInt<1>{});
            mma4(BInt<(c == 0)>{}, A[c][0], g0, n, 0);
            landed(g1, true);
            reads(g0, sb, BInt<2>{});
            mma4(BInt<0>{}, A[c][1], g1, n, 1);
            landed(g0, true);
            reads(g1, sb, BInt<3>{});
            mma4(BInt<0>{}, A[c][2], g0, n, 2);
            landed(g1, true);
            reads(g0, sbn, BInt<0>{});  // first group of chunk n+1 (published above; past the end: an unused landed slot)
            mma4(BInt<0>{}, A[c][3], g1, n, 3);
            }
            if (pass == 3) FRSTAMP(66 + 3 * c);
        };
        if constexpr (KC >= 1) chunk(BInt<0>{});
        if constexpr (KC >= 2) chunk(BInt<1>{});
        if constexpr (KC >= 3) chunk(BInt<2>{});
        if constexpr (KC >= 4) chunk(BInt<3>{});
        if constexpr (KC >= 5) chunk(BInt<4>{});
        if constexpr (KC >= 6) chunk(BInt<5>{});
        if constexpr (KC >= 7) chunk(BInt<6>{});
        if constexpr (KC >= 8) chunk(BInt<7>{});
        if constexpr (KC >= 9) chunk(BInt<8>{});
        if constexpr (KC >= 10) chunk(BInt<9>{});
        if constexpr (KC >= 11) chunk(BInt<10>{});
        if constexpr (KC >= 12) chunk(BInt<11>{});
        if constexpr (KC >= 13) chunk(BInt<12>{});
        if constexpr (KC >= 14) chunk(BInt<13>{});
        if constexpr (KC >= 15) chunk(BInt<14>{});
        if constexpr (KC >= 16) chunk(BInt<15>{});

        // ---- pass end: + bias, fp16 logits out (through the wave's staging space), statistics from the rounded values
        FRSTAMP(5 + 3 * pass);
        {
            // the next pass's bias: requested here, written to the other half of the table at the end of this epilogue
            const f32x4 bn = *(const f32x4 *)(a.bias + 128 * (pass + 1 < npass ? pass + 1 : pass) + 4 * j);
            char *rowp = (char *)(a.logits + row0 * V + 128 * pass);  // wave-uniform
            // per-lane addresses from a fresh lane id (not carried through the main loop: registers)
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const int jj = ln & 31, hh = ln >> 5;
            const int stw0 = stg + jj * 64 + (((2 * hh) ^ ((jj >> 2) & 3)) << 4), stw1 = stg + jj * 64 + (((2 * hh + 1) ^ ((jj >> 2) & 3)) << 4);
            const int strd = stg + (ln >> 2) * 64 + (((ln & 3) ^ ((ln >> 4) & 3)) << 4);  // rows l>>2 and 16 + (l>>2): same XOR term
            const unsigned lane_off = (unsigned)(((ln >> 2) * V + 8 * (ln & 3)) * 2);
            const int bias_r = lds0 + 4 * FR_SLOT + (pass & 1) * 512 + 64 * hh;  // columns 32g + 16*half + 4t + i
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 bq[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bq[t]) : "v"(bias_r), "n"(128 * g + 16 * t));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) :: "memory");
                u32x4 q0, q1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float x0, x1, x2, x3;  // (volatile: the reads stay here, one group at a time)
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[t][4 * g + 0]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[t][4 * g + 1]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x2) : "a"(acc[t][4 * g + 2]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x3) : "a"(acc[t][4 * g + 3]));
                    const unsigned lo = pack_f16(x0 + bq[t][0], x1 + bq[t][1]), hi = pack_f16(x2 + bq[t][2], x3 + bq[t][3]);
                    if (t < 2) { q0[2 * t] = lo; q0[2 * t + 1] = hi; } else { q1[2 * (t - 2)] = lo; q1[2 * (t - 2) + 1] = hi; }
                }
                if (!FR_OFF(4)) asm volatile("ds_write_b128 %0, %2\n\tds_write_b128 %1, %3" :: "v"(stw0), "v"(stw1), "v"(q0), "v"(q1) : "memory");
                if (!FR_OFF(2)) {
                    const unsigned m4 = pk_max_f16(pk_max_f16(pk_max_f16(q0[0], q0[1]), pk_max_f16(q0[2], q0[3])),
                                                   pk_max_f16(pk_max_f16(q1[0], q1[1]), pk_max_f16(q1[2], q1[3])));
                    const float lmax = f16_lo(max_halves_f16(m4));
                    float mn;
                    asm("v_max_f32 %0, %1, %2" : "=v"(mn) : "v"(st_m), "v"(lmax));
                    const float nm2 = -mn * RNNT_LOG2E;
                    float e = st_s * __builtin_amdgcn_exp2f(fmaf(st_m, RNNT_LOG2E, nm2));
                    float e2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        e += __builtin_amdgcn_exp2f(fmaf(f16_lo(q0[k]), RNNT_LOG2E, nm2)) + __builtin_amdgcn_exp2f(fmaf(f16_hi(q0[k]), RNNT_LOG2E, nm2));
                        e2 += __builtin_amdgcn_exp2f(fmaf(f16_lo(q1[k]), RNNT_LOG2E, nm2)) + __builtin_amdgcn_exp2f(fmaf(f16_hi(q1[k]), RNNT_LOG2E, nm2));
                    }
                    st_s = e + e2;
                    st_m = mn;
                }
                if (!FR_OFF(4)) {  // the staged group leaves row-major: rows l>>2 and 16 + (l>>2), 16 bytes per lane
                    u32x4 o0, o1;
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(o0), "=&v"(o1) : "v"(strd) : "memory");
                    *(u32x4 *)(rowp + lane_off + 64 * g) = o0;
                    *(u32x4 *)(rowp + (long)V * 32 + lane_off + 64 * g) = o1;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(bias_w ^ ((pass & 1) ? 0 : 512)), "v"(bn), "n"(0) : "memory");
        }
            FRSTAMP(6 + 3 * pass);
    }
    // ---- the over-issued DMAs (chunks NC .. NC+2) must land before this workgroup's LDS is released; all stores done
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed(g0, true);
    FRSTAMP(60);
    BF_CLOCK_STAMP(302);

    // ---- log-softmax denominator of cell j: the two column halves (lanes j and j+32), then the two log-probs
    float den;
    {
        const float m_o = __shfl_xor(st_m, 32, 64), s_o = __shfl_xor(st_s, 32, 64);
        const float M = fmaxf(st_m, m_o);
        const float S = st_s * __builtin_amdgcn_exp2f((st_m - M) * RNNT_LOG2E) + s_o * __builtin_amdgcn_exp2f((m_o - M) * RNNT_LOG2E);
        den = M + __logf(S);
    }
    const long cell = row0 + j;
    if (cell < cells) {
        const int u = (int)(cell % U1);
        const long bt = cell / U1;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const int Ub = len_u(a.target_lens, b, a.U1);
        if (t < len_t(a.logit_lens, b, a.T) && u <= Ub) {
            const unsigned short *lrow = a.logits + cell * V;
            const long si = skew_index(b, t, u, a.D, U1);
            auto stored = [&](int v) {  // fp16 logit v of this row, stored by THIS wave, read through L2
                const unsigned w = __hip_atomic_load((const unsigned *)(lrow + (v & ~1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return (v & 1) ? f16_hi(w) : f16_lo(w);
            };
            if (half == 0) {
                const float lb = stored(a.blank);
                a.denom_s[si] = den;
                a.lpb_s[si] = lb - den;
            } else {
                float le = 0.f;
                if (u < Ub) {
                    const int y = a.targets[(long)b * (U1 - 1) + u];
                    le = stored(y) - den;
                }
                a.lpe_s[si] = le;
            }
        }
    }
}

// H values the register-resident forward is instantiated for (else: the generic kernel above)
static int bf16_fwd_ra_kc(int H) { return (H == 128 || H == 256 || H == 512 || H == 1024) ? H / 64 : 0; }

template <int KC>
static void launch_fwd_ra(const Bf16Args &a, hipStream_t st)
{
    static bool attr_set[16] = {false};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    const int lds = FR_OFF(64) ? 100 * 1024 : 4 * FR_SLOT + 2 * 512 + 4 * 2048;
    if (dev < 0 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)k_joint_fwd_bf16_ra<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (dev >= 0) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k_joint_fwd_bf16_ra<KC>, dim3((unsigned)(a.rows_alloc / 128)), dim3(256), lds, st, a);
}

static void launch_pack_w_fwd_bf16(const Bf16Args &a, hipStream_t st)
{
    if (bf16_fwd_ra_kc(a.H)) {
        const long nf = (long)a.V * a.H / 8;
        hipLaunchKernelGGL(k_pack_w_fwd_bf16_ra, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, a.W, (u32x4 *)a.wpack_fwd, a.H, nf);
    } else {
        const long nf = (long)(bf16_wpack_fwd_bytes(a.H, a.V) / 16);
        hipLaunchKernelGGL(k_pack_w_fwd_bf16, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, a.W, (u32x4 *)a.wpack_fwd, a.H, a.V,
                           a.H / 32, nf);
    }
}

void launch_joint_fwd_bf16(const Bf16Args &a, hipStream_t st)
{
    switch (bf16_fwd_ra_kc(a.H)) {
    case 2: launch_fwd_ra<2>(a, st); break;
    case 4: launch_fwd_ra<4>(a, st); break;
    case 8: launch_fwd_ra<8>(a, st); break;
    case 16: launch_fwd_ra<16>(a, st); break;
    default: hipLaunchKernelGGL(k_joint_fwd_bf16, dim3((unsigned)(a.rows_alloc / 128)), dim3(256), 0, st, a);
    }
}

// ---------------------------------------------------------------------------------------
// k_dhidden_bf16: G from the fp16 logits, stored as bf16 over its logits row (in place), and
// dHidden = G . W over 512 columns of H per launch; epilogue as the fp32 kernel: x (1 - hidden^2),
// sum over u -> dEnc slab, sum over t -> dPred slab.  Tile = 8 t x 16 u cells.
// 8 waves (two per SIMD, 128 accumulator registers each):
//  * production: wave w turns t-row w of the tile into G: lane (u = l&15, quarter = l>>4) owns
//    8 consecutive vocabulary entries per 32-wide chunk (16 B of fp16 logits in, 16 B of bf16
//    out, same bytes), one chunk ahead of the MFMAs, and drops its
//    16 B straight into the MFMA A-fragment image of its M-tile in LDS (double-buffered);
//  * consumption: wave (wm = w&1, wn = w>>1) multiplies M-tiles 2wm, 2wm+1 (t-rows 4wm..4wm+3) by
//    the 128-column group wn (4 tiles): every B fragment it reads feeds two MFMAs.  The loop is
//    bound by LDS bandwidth (with everything but the exchange switched off it still took 8.4 of
//    12 ms): a 32-row x 256-column wave tile needs 18 fragment reads per 16 MFMAs, this one 12.
//    A fragments from the exchange, B fragments from the staged W chunk (each wave copies
//    4 KiB of it L2 -> VGPR -> LDS two chunks ahead).
// One barrier per chunk publishes both.  grid (n_ublk, ceil(T/8), B), 512 threads.
// Requires V % 128 == 0, H % 128 == 0.  One launch covers 512 columns of H: FIRST = true (columns
// 0-511) is the kernel described above; H > 512 (the reference's joint is 1024 wide) adds one
// FIRST = false launch per further 512 columns (`hp`), which finds G in place of the logits, copies
// it into the exchange instead of producing it, and stores nothing but its slabs.
// ---------------------------------------------------------------------------------------
// Diagnostic builds only (-DDH_EXP=bits, tools/build_bf16_variants.sh): parts of k_dhidden_bf16 compiled out — 1 no MFMAs,
// 2 no G arithmetic (exponentials), 4 no G stores, 8 no logits loads, 16 no W staging (loads + LDS writes), 32 no epilogue
#ifndef DH_EXP
#define DH_EXP 0
#endif
#define DH_OFF(bit) ((DH_EXP) & (bit))
#define BG_BT 8
#define BG_BU 16
template <bool FIRST>
__global__ __launch_bounds__(512, 1) void k_dhidden_bf16(Bf16Args a, const int hp)
{
    // [0, 64 KiB): W ring, 2 slots x [s(2)][tile(16)][lane(64)] x 16 B;  [64, 80 KiB): G exchange,
    // 2 slots x [M-tile(4)][s(2)][lane(64)] x 16 B.  The epilogue reuses all of it.
    __shared__ __attribute__((aligned(16))) u32x4 s_mem[2 * 2048 + 2 * 512];
    u32x4 *s_b = s_mem, *s_g = s_mem + 2 * 2048;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int j = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int ub = blockIdx.x, tt = blockIdx.y, b = blockIdx.z;
    const int Tb = len_t(a.logit_lens, b, a.T);
    const int t0 = tt * BG_BT, u0 = ub * BG_BU;
    const int VC = V / 32;

    // ---- producer role: cell (pt, pu), vocabulary quarter qd of every chunk
    const int r16 = lane & 15, qd = lane >> 4;
    const int pt = t0 + wave, pu = u0 + r16;
    const bool pexists = pt < T && pu < U1;
    const long zrow = (long)a.B * T * U1;  // first zero padding row
    const long pcell = pexists ? ((long)b * T + pt) * U1 + pu : zrow;
    unsigned short *lrow = a.logits + pcell * V;
    u32x4 *grow = (u32x4 *)lrow + qd;  // chunk c: grow[4c]  (16 B of bf16 at byte 64c + 16qd)

    // workgroup-uniform: no products past the utterance's length or in a u block past U_b (no
    // lattice cell; the reductions skip its slabs), but k_dw_bf16 must find zeros in these rows
    if (t0 >= Tb || u0 > len_u(a.target_lens, b, a.U1)) {
        if (FIRST && pexists) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int c = 0; c < VC; ++c) grow[4 * c] = z;
        }
        return;
    }

    CellCoef cf = a.coef[pexists ? pcell : 0];
    const bool live = pexists && pt < Tb && cf.c1 != RNNT_NEG_INF;
    if (!live) { cf.c1 = RNNT_NEG_INF; cf.sb = 0.f; cf.se = 0.f; cf.y = -1; }
    // rows outside the lattice read the zero padding row (finite) with c1 = -inf -> G = 0
    // FIRST: fp16 logits of the live rows; later passes: the bf16 G every existing row now holds
    const u32x4 *xsrc = (const u32x4 *)((FIRST ? live : pexists) ? lrow : a.logits + zrow * V) + qd;  // chunk c: xsrc[4c]
    const int blank = a.blank;
    // fragment image: MFMA s, lane (r, h) holds k = 16h + 8s + 0..7 of the chunk = quarter 2h + s
    // -> this lane's 16 B go to [M-tile wave>>1][s = qd&1][lane (qd>>1)*32 + 16*(wave&1) + r16]
    const int gdst = (wave >> 1) * 128 + (qd & 1) * 64 + (qd >> 1) * 32 + 16 * (wave & 1) + r16;
    const u32x4 *wp = (const u32x4 *)a.wpack_dh + (long)hp * VC * 2048 + (wave * 4) * 64 + lane;  // this wave's 4 of the chunk's 32 pieces

    f32x16 acc[8];
#pragma unroll
    for (int tl = 0; tl < 8; ++tl)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tl][r] = 0.f;

    if (FIRST) BF_CLOCK_STAMP(304);
    auto produce = [&](const u32x4 &x, int c, int slot) {
        if constexpr (!FIRST) { s_g[slot * 512 + gdst] = x; return; }
        if (DH_OFF(2)) { s_g[slot * 512 + gdst] = x; return; }
        float g[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[2 * e] = __builtin_amdgcn_exp2f(fmaf(f16_lo(x[e]), RNNT_LOG2E, cf.c1));
            g[2 * e + 1] = __builtin_amdgcn_exp2f(fmaf(f16_hi(x[e]), RNNT_LOG2E, cf.c1));
        }
        const int vb = 32 * c + 8 * qd;
        const unsigned dy = (unsigned)(cf.y - vb);
        if (__any(dy < 8u)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (dy == (unsigned)e) ? g[e] - cf.se : g[e];
        }
        if ((unsigned)(blank - 32 * c) < 32u) {  // wave-uniform
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (vb + e == blank) ? g[e] - cf.sb : g[e];
        }
        const u32x4 o = {pack_bf16(g[0], g[1]), pack_bf16(g[2], g[3]), pack_bf16(g[4], g[5]), pack_bf16(g[6], g[7])};
        s_g[slot * 512 + gdst] = o;
    };
    // G goes to memory in PAIRS of chunks, read back from this wave's own part of the exchange in
    // row-major order: lane (row = 8i + (l>>3), piece = l&7) stores 16 B, so a store instruction
    // writes 8 whole 128-byte lines (the producer layout — 4 lanes per row — wrote 16 half lines
    // per instruction and was bound by store issue).  Pair (c-1, c), c odd: both chunks sit in
    // the two exchange slots between the production of c and that of c+1.
    const int sp = lane & 7, sr = lane >> 3;  // piece: chunk parity sp>>2, quarter sp&3
    const int gsrc = (sp >> 2) * 512 + (wave >> 1) * 128 + (sp & 1) * 64 + ((sp >> 1) & 1) * 32 + 16 * (wave & 1) + sr;
    u32x4 *gst[2];
    bool gst_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int su = u0 + 8 * i + sr;
        gst_ok[i] = pt < T && su < U1;
        gst[i] = (u32x4 *)(a.logits + (gst_ok[i] ? ((long)b * T + pt) * U1 + su : zrow) * V) + sp;
    }
    auto wload = [&](u32x4 (&w)[4], int c) {
        if (DH_OFF(16)) return;
        const u32x4 *p = wp + (long)(c < VC ? c : VC - 1) * 2048;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = p[i * 64];
    };
    auto wstore = [&](const u32x4 (&w)[4], int slot) {
        if (DH_OFF(16)) return;
        u32x4 *p = s_b + slot * 2048 + (wave * 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i * 64] = w[i];
    };
    auto xload = [&](u32x4 &x, int c) { if (!DH_OFF(8)) x = xsrc[4 * (c < VC ? c : VC - 1)]; };

    u32x4 xr[4] = {};  // logits ring (slot = chunk & 3), 4 chunks ahead of production
    u32x4 wx[4] = {}, wy[4] = {};  // staged W: even / odd chunks
    xload(xr[0], 0); xload(xr[1], 1); xload(xr[2], 2); xload(xr[3], 3);
    wload(wx, 0);
    wload(wy, 1);
    wstore(wx, 0);
    wload(wx, 2);
    produce(xr[0], 0, 0);
    xload(xr[0], 4);
    lds_barrier();

    for (int c0 = 0; c0 < VC; c0 += 4) {  // VC % 4 == 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q;
            if (q & 1) { wstore(wx, 0); wload(wx, c + 3); }
            else       { wstore(wy, 1); wload(wy, c + 3); }
            // G of chunk c+1 (logits requested 4 chunks ago) into the other exchange slot
            produce(xr[(q + 1) & 3], c + 1, (q + 1) & 1);
            xload(xr[(q + 1) & 3], c + 5);
            __builtin_amdgcn_sched_barrier(0);
            {
                u32x4 af[2][2];  // [M-tile][k-step]
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int s_ = 0; s_ < 2; ++s_) af[mt][s_] = s_g[(q & 1) * 512 + (2 * wm + mt) * 128 + s_ * 64 + lane];
                // this wave's 4 column tiles of the staged chunk: tiles 4wn .. 4wn+3 of each k-step
                const u32x4 *pb = s_b + (q & 1) * 2048 + (4 * wn) * 64 + lane;
                constexpr int DEPTH = 3;
                u32x4 bf[DEPTH + 1];
#pragma unroll
                for (int n = 0; n < DEPTH; ++n) bf[n] = pb[((n >> 2) * 16 + (n & 3)) * 64];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    if (n + DEPTH < 8) bf[(n + DEPTH) % (DEPTH + 1)] = pb[(((n + DEPTH) >> 2) * 16 + ((n + DEPTH) & 3)) * 64];
                    if (!DH_OFF(1)) {
                        acc[n & 3] = mfma_bf16(af[0][n >> 2], bf[n % (DEPTH + 1)], acc[n & 3]);
                        acc[4 + (n & 3)] = mfma_bf16(af[1][n >> 2], bf[n % (DEPTH + 1)], acc[4 + (n & 3)]);
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4 + DEPTH + 2, 0);
#pragma unroll
                for (int n = 0; n < 8 - DEPTH; ++n) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * DEPTH, 0);
            }
            if (FIRST && !(q & 1) && !RNNT_XP(a.flags, 256) && !DH_OFF(4)) {  // chunk c+1 is odd: the pair (c, c+1) is complete
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 gb0 = s_g[gsrc], gb1 = s_g[gsrc + 8];
                if (gst_ok[0]) gst[0][4 * c] = gb0;
                if (gst_ok[1]) gst[1][4 * c] = gb1;
            }
            lds_barrier();
        }
    }

    if (FIRST) BF_CLOCK_STAMP(306);
    // ---- epilogue.  Accumulator register r of tile 4mt+q (mt = 0,1): row (r&3) + 8(r>>2) + 4*half
    // of M-tile 2wm+mt = (t-row 2(2wm+mt) + (r>>3), u (r&3) + 8((r>>2)&1) + 4*half), column 128wn + 4j + q.
    if (RNNT_XP(a.flags, 8192) || DH_OFF(32)) {  // (the accumulators stay "used": without this the MFMAs are dead code too)
#pragma unroll
        for (int tl = 0; tl < 8; ++tl) asm volatile("" :: "a"(acc[tl]));
        return;
    }
    float *s_red = (float *)s_mem;  // [8 waves][64 lanes][33]
    const long BTH = (long)a.B * T * H, BUH = (long)a.B * U1 * H;
    const int col0 = 512 * hp + 128 * wn + 4 * j;
    const bool colok = col0 < H;
    // All 32 hidden fragments of the epilogue requested up front through a raw buffer over the
    // tile's rows (scalar row offset + per-lane offset, no predicates: one memory round trip
    // instead of four; rows outside the lattice have G = 0 and therefore an exactly zero
    // accumulator, whatever finite — or, out of range, zero — hidden value they meet).
    u32x2 hq[2][2][8];
    {
        const long cell0 = ((long)b * T + t0) * U1 + u0;
        const long rows_left = a.rows_alloc - cell0;
        const long span = (long)(BG_BT - 1) * U1 + BG_BU < rows_left ? (long)(BG_BT - 1) * U1 + BG_BU : rows_left;
        const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.hidden + cell0 * H), 0, (int)(span * H * 2), 0x00020000);
        const unsigned voff = colok ? (unsigned)(((4 * half) * H + col0) * 2) : 0xfffffff0u;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int tl_ = 0; tl_ < 2; ++tl_)
#pragma unroll
                for (int r7 = 0; r7 < 8; ++r7) {
                    const unsigned soff = (unsigned)(((2 * (2 * wm + mt) + tl_) * U1 + (r7 & 3) + 8 * (r7 >> 2)) * H) * 2u;
                    hq[mt][tl_][r7] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(hrsrc, voff, soff, 0));
                }
    }
    float psum[8][4];
#pragma unroll
    for (int r7 = 0; r7 < 8; ++r7)
#pragma unroll
        for (int q = 0; q < 4; ++q) psum[r7][q] = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int tl_ = 0; tl_ < 2; ++tl_) {
            const int t = t0 + 2 * (2 * wm + mt) + tl_;
            const bool tok = t < Tb;
            float esum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r7 = 0; r7 < 8; ++r7) {
                const u32x2 h2 = hq[mt][tl_][r7];
                const float hv[4] = {bf16_lo(h2[0]), bf16_hi(h2[0]), bf16_lo(h2[1]), bf16_hi(h2[1])};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float av = acc[4 * mt + q][tl_ * 8 + r7];
                    const float d = av * (1.f - hv[q] * hv[q]);
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
    __syncthreads();  // main loop done with the LDS being reused
#pragma unroll
    for (int r7 = 0; r7 < 8; ++r7)
#pragma unroll
        for (int q = 0; q < 4; ++q) s_red[(wave * 64 + lane) * 33 + r7 * 4 + q] = psum[r7][q];
    __syncthreads();
    // thread (wm, lane) of column group wn sums rows r7 = 4wm .. 4wm+3 of source lane `lane` over
    // the 2 waves (t-row quadruples) that share wn
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r7 = 4 * wm + k;
        const int u = u0 + (r7 & 3) + 8 * (r7 >> 2) + 4 * half;
        if (u < U1 && colok) {
            f32x4 o;
            const float *sr = s_red + ((2 * wn) * 64 + lane) * 33 + r7 * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = sr[q] + sr[64 * 33 + q];
            *(f32x4 *)(a.slab_pred + (long)tt * BUH + ((long)b * U1 + u) * H + col0) = o;
        }
    }
}

void launch_dhidden_bf16(const Bf16Args &a, hipStream_t st)
{
    const long cells = (long)a.B * a.T * a.U1;
    // zero padding rows: G of rows k_dw_bf16 walks past the last cell, and the "dead row" source
    launch_fill32(a.logits + cells * a.V, 0u, (size_t)(a.rows_alloc - cells) * a.V * 2, st);
    dim3 grid(a.n_ublk, (a.T + BG_BT - 1) / BG_BT, a.B);
    hipLaunchKernelGGL(k_dhidden_bf16<true>, grid, dim3(512), 0, st, a, 0);
    for (int hp = 1; hp * 512 < a.H; ++hp) hipLaunchKernelGGL(k_dhidden_bf16<false>, grid, dim3(512), 0, st, a, hp);
}

// ---------------------------------------------------------------------------------------
// k_dw_bf16: dW[v,h] = sum_c G[c,v] hidden[c,h] (split-K slabs), db[v] = sum_c G[c,v].
// 4 waves = 2 (M) x 2 (N), workgroup tile 256 v x 256 h, wave 128 x 128 = 16 tiles (256
// accumulator registers).  Both operands are row-major with K (the cell) as the ROW while the
// MFMA wants 8 consecutive k per lane, so they go through LDS and come back transposed:
//  * HBM -> LDS by LDS-DMA (no VGPRs), 4-stage ring of 32 cells x (256 v + 256 h) = 32 KiB,
//    three stages in flight (96 KiB per CU); wave w fills operand tile w ([32 cells][128 cols]);
//  * LDS -> VGPR by ds_read_b64_tr_b16 (hardware 4x16 transpose read, cdna_hip_programming.md
//    T10): two reads give a lane its 8 consecutive cells of one column = the MFMA fragment;
//  * tile image (b) of T10: 256-byte rows, 16-byte chunk ch of row r at 16*(ch ^ swz(r)),
//    swz(r) = ((r&3)<<2) | ((r>>2)&3).  The DMA writes LDS linearly, so the swizzle is applied
//    on the SOURCE side: LDS chunk position p of row r is fetched from global chunk p ^ swz(r).
// The only vector-memory instructions in the loop are the DMAs, so one counted
// s_waitcnt vmcnt(16) + one barrier per stage publishes a stage.
// The bias gradient rides the matrix pipe: an extra B fragment of ones (column 0).
// ---------------------------------------------------------------------------------------
#define BW_ROWS 32   // cells per stage (2 MFMA k-steps)
#define BW_NST 4     // ring stages
// Diagnostic builds only (-DBW_EXP=bits, VAR=BW_EXP tools/build_bf16_variants.sh): parts of k_dw_bf16 compiled out (results wrong by
// construction) — 1 no MFMAs, 2 no db (v_dot2c), 4 no transposed fragment reads, 8 no DMA bytes (requested past the buffer's range: the
// instruction and its vmcnt stay), 16 no DMA instructions, 32 no barrier, 64 no lockstep with the split's other tiles
#ifndef BW_EXP
#define BW_EXP 0
#endif
#define BW_OFF(bit) ((BW_EXP) & (bit))

__global__ __launch_bounds__(256, 1) void k_dw_bf16(Bf16Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char s_ring[];  // BW_NST x 4 tiles x 8 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5;
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
    // this split's share of the LIVE 32-cell stages (k_dw_table): a ragged batch costs its lengths
    const long *tab = a.dw_tab;
    const int B = a.B;
    const long nlive = tab[2 * B + 1];
    const long g_lo = nlive * split / a.n_split, g_hi = nlive * (split + 1) / a.n_split;
    BF_CLOCK_STAMP(308);

    f32x16 acc[4][4];
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;
    // db: the waves with wn == 0 of the hb == 0 workgroups add up their G fragments (v_dot2 with
    // a pair of ones: 16 VALU per k-step); a 17th accumulator tile would not fit the 256 AGPRs
    const bool do_b = hb == 0 && wn == 0;  // wave-uniform
    float dbl[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned one_pair_u = 0x3f803f80u;  // (1.0bf16, 1.0bf16)

    if (g_hi > g_lo) {
        // ---- DMA source of this wave's operand tile: wave 0/1 -> G column halves, 2/3 -> hidden
        const bool is_g = wave < 2;
        int col0 = (is_g ? vb : hb) * 256 + 128 * (wave & 1);
        if (col0 >= (is_g ? V : H)) col0 = 0;  // tile beyond the matrix: never stored, read something valid
        const long rstride = is_g ? 2L * V : 2L * H;  // bytes between cells
        const char *src0 = (is_g ? (const char *)a.logits : (const char *)a.hidden) + 2L * col0;
        const char *src = src0;  // + the first row of the range being walked
        // DMA i (0..7) of a stage: rows 4i .. 4i+3; lane L: row 4i + (L>>4), LDS chunk position L&15
        // <- global chunk (L&15) ^ swz(row), swz = ((L>>4)<<2) | (i&3)
        int soff[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            soff[i] = BW_OFF(8) ? 0x7ffffff0 : (int)((4 * i + (lane >> 4)) * rstride) + 16 * ((lane & 15) ^ (((lane >> 4) << 2) | (i & 3))) - 1024 * (i & 3);
        // (round 5: pieces 4g .. 4g+3 share one LDS base (M0) and carry the immediate offset 1024 (i & 3), which advances the memory address
        // too — taken back out of the per-lane offset above: rstride >= 256 bytes, so 4 i rows >= 1024 (i & 3) bytes)
        // (raw-buffer form: the 32 rows of a stage as a buffer with a wave-uniform base — scalar arithmetic only; the
        // per-lane part of an address is the 32-bit soff.  A global_load_lds with a 64-bit per-lane address costs the
        // issuing wave more: measured on the bf16x3 forward)
        auto stage_rsrc = [&](long st) {
            return __builtin_amdgcn_make_buffer_rsrc((void *)(src + st * (BW_ROWS * rstride)), 0, (int)(BW_ROWS * rstride), 0x00020000);
        };
        auto dma_stage = [&](long st, int slot) {
            if (BW_OFF(16)) return;
            const __amdgpu_buffer_rsrc_t r = stage_rsrc(st);
            char *dst = s_ring + slot * 32768 + wave * 8192;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                lds_vptr d4 = (lds_vptr)(dst + 4096 * (i >> 2));
                if ((i & 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d4, 16, soff[i], 0, 0, 0);
                if ((i & 3) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d4, 16, soff[i], 0, 1024, 0);
                if ((i & 3) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d4, 16, soff[i], 0, 2048, 0);
                if ((i & 3) == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d4, 16, soff[i], 0, 3072, 0);
            }
        };
        // ---- transposed fragment reads.  Fragment of 32-column tile m, k-step ks: lane
        // (g = lane>>4, q = (lane&15)>>2, p = lane&3) reads rows 16ks + 8(g>>1) + 4sec + q at
        // chunk 4m + 2(g&1) + (p>>1), +8(p&1) bytes, sec = 0,1 (cells 0-3 / 4-7 of its 8).
        // The reads are inline asm: hipcc guards every LDS read it can see that follows an
        // LDS-DMA with s_waitcnt vmcnt(0), which would drain the ring each stage.  Their
        // results are only used after frag_wait(), which names them as in/out operands.
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, hh = g >> 1;
        int foff[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int sec = 0; sec < 2; ++sec) {
                const int row = 8 * hh + 4 * sec + q;  // + 16ks
                const int ch = 4 * m + 2 * (g & 1) + (p >> 1);
                const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
                foff[m][sec] = 256 * row + 16 * (ch ^ swz) + 8 * (p & 1);
            }
        const int lds0 = (int)(size_t)(lds_vptr)s_ring;  // LDS byte address of the ring
        const int a_tile = lds0 + wm * 8192, b_tile = lds0 + 16384 + wn * 8192;
        auto dma_piece = [&](long st, int slot, int i) {  // (i: a constant once the caller's loop is unrolled)
            lds_vptr d4 = (lds_vptr)(s_ring + slot * 32768 + wave * 8192 + 4096 * (i >> 2));
            if ((i & 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(stage_rsrc(st), d4, 16, soff[i], 0, 0, 0);
            if ((i & 3) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(stage_rsrc(st), d4, 16, soff[i], 0, 1024, 0);
            if ((i & 3) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(stage_rsrc(st), d4, 16, soff[i], 0, 2048, 0);
            if ((i & 3) == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(stage_rsrc(st), d4, 16, soff[i], 0, 3072, 0);
        };
        struct Frags { u32x2 al[4], ah[4], bl[4], bh[4]; };
        auto reads = [&](Frags &f, int slot, int ks) {  // 16 transposed reads, NOT waited for
            if (RNNT_XP(a.flags, 4096) || BW_OFF(4)) return;  // experiment switch
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int a0 = a_tile + slot * 32768 + 4096 * ks, b0 = b_tile + slot * 32768 + 4096 * ks;
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.al[m]) : "v"(a0 + foff[m][0]));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.ah[m]) : "v"(a0 + foff[m][1]));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.bl[m]) : "v"(b0 + foff[m][0]));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.bh[m]) : "v"(b0 + foff[m][1]));
            }
        };
        auto landed = [&](Frags &f, bool wait) {  // every later use of f depends on this point
            if (wait)
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(f.al[0]), "+v"(f.al[1]), "+v"(f.al[2]), "+v"(f.al[3]), "+v"(f.ah[0]),
                               "+v"(f.ah[1]), "+v"(f.ah[2]), "+v"(f.ah[3])
                             :: "memory");
            else
                asm volatile("" : "+v"(f.al[0]), "+v"(f.al[1]), "+v"(f.al[2]), "+v"(f.al[3]), "+v"(f.ah[0]),
                                  "+v"(f.ah[1]), "+v"(f.ah[2]), "+v"(f.ah[3]));
            asm volatile("" : "+v"(f.bl[0]), "+v"(f.bl[1]), "+v"(f.bl[2]), "+v"(f.bl[3]), "+v"(f.bh[0]),
                              "+v"(f.bh[1]), "+v"(f.bh[2]), "+v"(f.bh[3]));
        };
        // 16 MFMAs of one k-step with 4 DMA pieces of stage `dst` threaded through them (a DMA
        // costs ~60 issue cycles: issued in one block they stall the matrix pipe)
        auto mma_step = [&](const Frags &f, long dst, int dslot, int piece0) {
            u32x4 fa[4], fb[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                fa[m] = u32x4{f.al[m][0], f.al[m][1], f.ah[m][0], f.ah[m][1]};
                fb[m] = u32x4{f.bl[m][0], f.bl[m][1], f.bh[m][0], f.bh[m][1]};
            }
#pragma unroll
            for (int qm = 0; qm < 4; ++qm) {
                if (!RNNT_XP(a.flags, 1024) && !BW_OFF(1)) {
#pragma unroll
                    for (int qn = 0; qn < 4; ++qn) acc[qm][qn] = mfma_bf16(fa[qm], fb[qn], acc[qm][qn]);
                }
                if (!RNNT_XP(a.flags, 8192) && !BW_OFF(16)) dma_piece(dst, dslot, piece0 + qm);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (do_b && !RNNT_XP(a.flags, 2048) && !BW_OFF(2)) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        // (the fdot2_f32_bf16 builtin on a vector element was folded to element 0
                        // by hipcc 7.2: spelled out instead)
                        asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(dbl[m]) : "v"(one_pair_u), "v"(fa[m][e]));
            }
        };

        // Soft lockstep of the tiles of a split.  They walk the same stages and share every operand byte through their
        // XCD's L2 (G between the h blocks, hidden between the v blocks); nothing else holds them together, and at this
        // kernel's pace a tile that falls ~10 stages behind finds its lines evicted, becomes its own HBM stream and
        // slows everybody further — counted: 1.85x the algorithmic traffic.  Every 4th stage a tile publishes its stage
        // count and looks at its right-hand neighbour's in the ring of the split's tiles (one coherent scalar load,
        // issued a stage before its value is used); a tile more than DW_LAG stages ahead of that neighbour naps.
        // Bounded: after DW_NAPS naps without the neighbour moving (a partner that is not resident) the tile stops
        // looking, so every wave reaches the end whatever the others do.
        constexpr int DW_LAG = 3, DW_NAPS = 256;
        int *prog = a.dw_prog ? a.dw_prog + split * 16 : nullptr;
        bool sync_on = prog != nullptr && tiles > 1 && tiles <= 16 && !BW_OFF(64);
        const int *nb = prog ? prog + (tile + 1 < tiles ? tile + 1 : 0) : nullptr;  // the neighbour's word
        int nb_at = 0x7fffffff;  // the neighbour's stage count as of the last look
        int done = 0;  // stages behind this workgroup, over all ranges
        int ub = 0;
        while (ub + 1 < B && tab[B + 1 + ub + 1] <= g_lo) ++ub;  // utterance holding live stage g_lo
        for (long gq = g_lo; gq < g_hi; ++ub) {  // workgroup-uniform: one pipeline run per live range
        const long cum0 = tab[B + 1 + ub], cum1 = ub + 1 < B ? tab[B + 1 + ub + 1] : nlive;
        const long ge = cum1 < g_hi ? cum1 : g_hi;
        if (ge <= gq) continue;
        const long nstage = ge - gq;
        src = src0 + ((tab[ub] + (gq - cum0)) * BW_ROWS) * rstride;
        gq = ge;
        // Pipeline.  B_s = barrier publishing stage s (every wave has landed its share and has
        // finished reading stage s-1).  After B_s: reads of stage s, DMA of stage s+3 into the
        // slot of stage s-1.  Fragment reads run one k-step ahead of their MFMAs.
        dma_stage(0, 0);
        dma_stage(1, 1);
        dma_stage(2, 2);
        asm volatile(RNNT_VMCNT(16) ::: "memory");
        lds_barrier();  // B_0
        Frags X, Y;
        reads(X, 0, 0);
        landed(X, true);
        for (long st = 0; st < nstage; ++st) {
            const int slot = (int)(st & 3);
            const int mine = done + (int)st;
            const bool look = sync_on && (mine & 3) == 0;  // wave-uniform
            if (look) {
                // the value requested at the previous look has long landed (lgkmcnt(0) of the barriers since)
                int naps = 0;
                while (nb_at + DW_LAG + 4 < mine) {  // (+4: the value is one look old)
                    if (++naps > DW_NAPS) { sync_on = false; break; }
                    __builtin_amdgcn_s_sleep(4);
                    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(nb_at) : "s"(nb) : "memory");
                }
                if (tid == 0) __hip_atomic_store(prog + tile, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(nb_at) : "s"(nb) : "memory");  // used at the next look
            }
            reads(Y, slot, 1);
            mma_step(X, st + 3, (slot + 3) & 3, 0);
            // stage st+1: younger in flight = stage st+2 (8 DMAs) + the 4 pieces just issued
            // (+ wave 0's progress store now and then: one more outstanding operation only makes the wait stricter)
            asm volatile(RNNT_VMCNT(12) ::: "memory");
            if (BW_OFF(32)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
            lds_barrier();  // B_{st+1}; its lgkmcnt(0) also covers Y and the neighbour's progress word
            landed(Y, false);
            reads(X, (slot + 1) & 3, 0);  // past the last stage: reads a landed, unused slot
            mma_step(Y, st + 3, (slot + 3) & 3, 4);
            landed(X, true);  // X is loop-carried: landed before the back-edge, copies are safe
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the over-issued DMAs before the
        lds_barrier();                                     // ring is refilled / the kernel exits
        done += (int)nstage;
        }
        if (prog && tid == 0) __hip_atomic_store(prog + tile, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // done: nobody waits for this tile
    }

    // ---- epilogue: partial slab [split][V,H]; bias partial [split][V].  Accumulator register r
    // of tile (qm,qn): v = v0 + 32qm + (r&3) + 8(r>>2) + 4half, h = h0 + 32qn + (lane&31).
    BF_CLOCK_STAMP(310);
    const int v0 = vb * 256 + wm * 128, h0 = hb * 256 + wn * 128;
    float *sw = a.slab_w + (long)split * V * H;
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + 32 * qm + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (v < V) {
#pragma unroll
                for (int qn = 0; qn < 4; ++qn) {
                    const int h = h0 + 32 * qn + (lane & 31);
                    if (h < H) sw[(long)v * H + h] = acc[qm][qn][r];
                }
            }
        }
    if (do_b) {  // lane (v = l&31, half) summed the cells 8*half .. 8*half+7 of every k-step
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float t = dbl[m] + __shfl_xor(dbl[m], 32, 64);
            const int v = v0 + 32 * m + (lane & 31);
            if (half == 0 && v < V) a.slab_b[(long)split * V + v] = t;
        }
    }
}

void launch_dw_bf16(const Bf16Args &a, hipStream_t st)
{
    launch_dw_table(a.logit_lens, a.B, a.T, a.U1, BW_ROWS, a.dw_tab, st);
    const int tiles = ((a.V + 255) / 256) * ((a.H + 255) / 256);
    // > 64 KiB of dynamic LDS needs the opt-in, once per DEVICE (a function attribute belongs to the
    // device's code object); a read-mostly fact, like device_cus()
    static bool attr_set[16] = {false};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;  // unknown: set it every time
    if (dev < 0 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)k_dw_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, BW_NST * 32768);
        if (dev >= 0) attr_set[dev] = true;
    }
    if (a.dw_prog) launch_fill32(a.dw_prog, 0u, (size_t)a.n_split * 64, st);
    hipLaunchKernelGGL(k_dw_bf16, dim3(tiles * a.n_split), dim3(256), BW_NST * 32768, st, a);
}
