// x2.hip — the RNNT_DTYPE_F32_F16X2 route: the bf16x3 route's design (x3.hip) at HALF its matrix work.
//
// Why (round 4, tools/exp_x3_clock.py): the three bf16x3 GEMM kernels are not issue-bound but POWER-bound — under their
// six-product MFMA stream plus its HBM traffic the chip holds 1.7-2.1 GHz instead of 2.4 (k_dw_x3: 1.77 GHz at 0.82
// MFMA-busy), so every cycle saved by a tighter schedule comes back as a lower clock.  What is left is less work per
// product.  fp16 carries 11 significant bits per piece against bf16's 8, so TWO pieces hold 22 bits and THREE products
//       x = hi + mid,  hi = f16(s x), mid = f16(s x - hi)   (RNE each; s = a power of two that lifts the operand's largest
//                                                            magnitude to ~2^14: fp16's 5-bit exponent needs it)
//       a.b  ~  ah.bh + ah.bm + am.bh                         (dropped: am.bm and the residuals, <= 3 x 2^-22 |a.b|)
// give the fp32 class of error (measured beside the fp32-MFMA route in tests/test_x2_gpu.py) on half the MFMAs and
// two thirds of the operand bytes of bf16x3: 3 products instead of 6, 2 planes instead of 3 (4 bytes per element — G's
// planes exactly fill the logits they replace, no third plane beside them).
// Scales: hidden = tanh(.) in [-1, 1] -> 2^14 (constant); G: |G| <= grad_scale -> 2^k from grad_scale (host, X3Args::
// g_scale); W: 2^k from max |W|, found on the device every call (k_x2_wscale -> X3Args::scales = {s_W, 1 / s_W}).  Powers of
// two: scaling and unscaling are exact.  Values beyond fp16's range after scaling (only possible for non-finite or
// garbage operands) are clamped to +-65504 where they are split.
// Same boundary, same workspace objects, same stage structure and variants as the bf16x3 route (engine.hip runs both through
// one code path); logits, log-softmax, lattice, coefficients and all reductions are the fp32 / fp64 code of the fp32 route.
//
// MFMA operand maps (v_mfma_f32_32x32x16_f16, as the bf16 form): lane l = (r = l&31, h = l>>5) holds A[row r][k = 8h+j] and
// B[k = 8h+j][col r], j = 0..7; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include "kernels.hpp"

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) void *lds_vptr;
typedef float f2 __attribute__((ext_vector_type(2)));

// Diagnostic build only (-DRNNT_STAMPS): workgroup 0 stamps the core clock counter and the 100 MHz reference at its start and
// end into debug[SLOT..] (a buffer nothing else reads): the clock this launch ran at (tools/exp_x3_clock.py).
#ifdef RNNT_STAMPS
#define X2_CLOCK_STAMP(SLOT)                                                                                               \
    do {                                                                                                                   \
        if (a.debug && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {                        \
            unsigned long long t_, r_;                                                                                     \
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");      \
            a.debug[SLOT] = t_; a.debug[(SLOT) + 1] = r_;                                                                  \
        }                                                                                                                  \
    } while (0)
#else
#define X2_CLOCK_STAMP(SLOT) do {} while (0)
#endif

#ifndef X2_REREAD
#define X2_REREAD 0  // 1: passes after a tile's first load the stored hidden planes back instead of producing them again — built, parity-green, measured SLOWER (cfg2 forward 21.23 vs 20.98 ms, same box: profiles/r06_fwd_whatif.txt): the planes are unique bytes per lane from L2, E and P are L1-resident rows
#endif
// compile-time experiment switches (-DX2_NT=bits): 1 non-temporal logits loads in k_dhidden_x2, 2 non-temporal operand DMAs in k_dw_x2<4>
#ifndef X2_NT
#define X2_NT 0
#endif
// -DX2_EXP=bits (what-if builds of k_joint_fwd_x2, WRONG results; tools/build_x2_variants.sh): 2 W DMAs requested past the buffer's range (instructions stay, no bytes move); 4 hidden stores aimed at the
// first tile's rows (cache-resident); 8 logits stores aimed at the first tile's rows; 16 logits stores with the default cache policy
// instead of non-temporal; 32 no production arithmetic; 64 no softmax statistics; 128 no MFMAs; 256 no logits stores at all; 512 no operand
// loads; 1024 no W DMA instructions; 2048 no fragment reads from LDS; 4096 no barrier in the k-step
// (round 5, on the round-4 schedule: 1 = half of W's bytes, 8192 / 16384 = a second set of operand loads / a second production per k-step;
// 1 | 8192 | 16384 priced the memory / VALU mix of a 256-cell x 256-column tile: 22.0 ms against 22.0 — profiles/r05_fwd_whatif.txt)
#ifndef X2_EXP
#define X2_EXP 0
#endif
#ifndef XF2_IMM
#define XF2_IMM 1  // 1: a k-step's W DMAs (forward, dHidden) in groups of four on one M0 / scalar offset, told apart by the instruction's 12-bit
#endif             // immediate offset, which advances the memory AND the LDS address: 12 scalar instructions fewer per k-step (forward 21.7 -> 21.05 ms)
#define X2_SH 16384.0f          // scale of the hidden operand (|tanh| <= 1)
#define X2_INV_SH (1.0f / 16384.0f)
#define X2_F16_MAX 65504.0f

__device__ __forceinline__ unsigned x2_pack(float lo, float hi)
{
    f16x2 v = {(_Float16)lo, (_Float16)hi};  // v_cvt_pk_f16_f32: RNE
    return __builtin_bit_cast(unsigned, v);
}
// two (scaled) fp32 values -> their (hi, mid) fp16 pieces, packed pairwise (element 0 in the low half): 4 instructions
struct X2Pieces { unsigned h, m; };
__device__ __forceinline__ X2Pieces x2_split2(float a, float b)
{
    X2Pieces p;
    p.h = x2_pack(a, b);
    float ra, rb;  // x - hi, exact (the residual has <= 14 significant bits): v_fma_mix reads the fp16 half in place
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p.h), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p.h), "v"(b));
    p.m = x2_pack(ra, rb);
    return p;
}
__device__ __forceinline__ float x2_clamp(float x) { return __builtin_amdgcn_fmed3f(x, -X2_F16_MAX, X2_F16_MAX); }
// four consecutive (scaled) fp32 values -> 2 packed dwords per plane at dword positions d, d+1 of the plane vectors
#define X2_SPLIT4(x4, ph, pm, d)                                      \
    do {                                                              \
        const X2Pieces p0_ = x2_split2((x4)[0], (x4)[1]);             \
        const X2Pieces p1_ = x2_split2((x4)[2], (x4)[3]);             \
        (ph)[d] = p0_.h; (pm)[d] = p0_.m;                             \
        (ph)[(d) + 1] = p1_.h; (pm)[(d) + 1] = p1_.m;                 \
    } while (0)
__device__ __forceinline__ f32x16 x2_mfma(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void x2_lds_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> struct X2Int { static constexpr int value = N; };
#define X2_FLAG_LINEAR 0x40000000  // X3Args::flags: launch_joint_fwd_x2 runs the plain-GEMM form (k_joint_fwd_x2<2>: the joint's input projections)

// ---------------------------------------------------------------------------------------
// s_W = 2^(14 - ceil(log2 max|W|)) (1 for an all-zero W): one workgroup, V*H/4 float4 reads.  A W with a NaN or an infinity in it (a diverged
// run) keeps s_W = 1 and gets a NaN as 1 / s_W: the packs' clamp (v_med3) would otherwise turn the entry into a finite fp16 and the loss would stay
// finite, where every other route — and the reference — report NaN.  Every logit and every dHidden entry is multiplied by 1 / s_W.
// ---------------------------------------------------------------------------------------
#define X2_F32_MAX 3.4028234663852886e38f
__global__ __launch_bounds__(1024) void k_x2_wscale(const float *__restrict__ W, long n4, float *__restrict__ scales)
{
    __shared__ float s_m[16];
    float m = 0.f;
    bool bad = false;
    long i = threadIdx.x;
    for (; i + 7 * 1024 < n4; i += 8 * 1024) {  // eight loads in flight per thread (one at a time: a round trip per 16 KiB, 58 us for 2 MB)
        f32x4 w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = ((const f32x4 *)W)[i + 1024 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float mw = fmaxf(fmaxf(fabsf(w[k][0]), fabsf(w[k][1])), fmaxf(fabsf(w[k][2]), fabsf(w[k][3])));
            bad |= (w[k][0] != w[k][0]) | (w[k][1] != w[k][1]) | (w[k][2] != w[k][2]) | (w[k][3] != w[k][3]);  // (infinities show in the maximum)
            m = fmaxf(m, mw);
        }
    }
    for (; i < n4; i += 1024) {
        const f32x4 w = ((const f32x4 *)W)[i];
        bad |= (w[0] != w[0]) | (w[1] != w[1]) | (w[2] != w[2]) | (w[3] != w[3]);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) m = fmaxf(m, s_m[k]);
        float s = 1.0f;
        const unsigned bits = __float_as_uint(m);
        const int e = (int)(bits >> 23) & 0xff;  // m in [2^(e-127), 2^(e-126))
        if (e > 0 && e < 255) {
            int k = 14 - (e - 126);               // m * 2^k in [2^13, 2^14)
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            s = __uint_as_float((unsigned)(127 + k) << 23);
        }
        scales[0] = s;
        scales[1] = (any_bad || e == 255) ? __uint_as_float(0x7fc00000u) : 1.0f / s;
    }
}

// ---------------------------------------------------------------------------------------
// Plain producers (the unfused variants RNNT_VARIANT_X3_FP32_*: every fused kernel is checked against the same pipeline
// with one stage swapped; also the plain statement of the data layout).
// ---------------------------------------------------------------------------------------
// hidden planes [2][rows_alloc][H] fp16 of 2^14 tanh(enc + pred), every cell.  One thread = 8 columns of one row.
__global__ __launch_bounds__(256) void k_x2_make_hidden(X3Args a)
{
    const int H8 = a.H / 8;
    const long cells = (long)a.B * a.T * a.U1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= cells * H8) return;
    const long c = idx / H8;
    const int h = (int)(idx - c * H8) * 8;
    const int u = (int)(c % a.U1);
    const long bt = c / a.U1;
    const int t = (int)(bt % a.T), b = (int)(bt / a.T);
    const float *ep = a.enc + (long)b * a.enc_sb + (long)t * a.enc_st + h;
    const float *pp = a.pred + ((long)b * a.U1 + u) * a.H + h;
    f32x4 t0 = fast_tanh_sum4(*(const f32x4 *)ep, *(const f32x4 *)pp);
    f32x4 t1 = fast_tanh_sum4(*(const f32x4 *)(ep + 4), *(const f32x4 *)(pp + 4));
#pragma unroll
    for (int k = 0; k < 4; ++k) { t0[k] = x2_clamp(t0[k] * X2_SH); t1[k] = x2_clamp(t1[k] * X2_SH); }  // (hidden = tanh(.) is finite or NaN; v_med3 maps a NaN to -65504: non-finite enc / pred are caught upstream by k_x2_make_ep's flag and run the exact form)
    u32x4 ph, pm;
    X2_SPLIT4(t0, ph, pm, 0);
    X2_SPLIT4(t1, ph, pm, 2);
    u32x4 *o = (u32x4 *)(a.hidden + c * a.H + h);
    const long ps = a.plane_stride / 8;  // u32x4 units
    o[0] = ph; o[ps] = pm;
}

// fp32 G (what the fp32 route's kernels leave in place of the logits) -> g_scale G as two planes, hi | mid over the same 128
// bytes of every 32-wide chunk.  One thread = one chunk of one row (it reads the whole 128 bytes before it overwrites them).
__global__ __launch_bounds__(256) void k_x2_split_g(X3Args a, long rows)
{
    const int VC = a.V / 32;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * VC) return;
    const long r = idx / VC;
    const int c = (int)(idx - r * VC);
    f32x4 *p = (f32x4 *)(a.logits + r * a.V + 32 * c);
    f32x4 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x[i] = p[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[i][k] = x2_clamp(x[i][k] * a.g_scale);
    }
    u32x4 ph[4], pm[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) X2_SPLIT4(x[i], ph[i >> 1], pm[i >> 1], 2 * (i & 1));
    u32x4 *o = (u32x4 *)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = ph[i]; o[4 + i] = pm[i]; }
}

void launch_x2_make_hidden(const X3Args &a, hipStream_t st)
{
    const long n = (long)a.B * a.T * a.U1 * (a.H / 8);
    hipLaunchKernelGGL(k_x2_make_hidden, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}
void launch_x2_split_g(const X3Args &a, hipStream_t st)
{
    const long rows = (long)a.B * a.T * a.U1;
    const long n = rows * (a.V / 32);
    hipLaunchKernelGGL(k_x2_split_g, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, rows);
}
// zero rows past the last cell, both planes (the dW ring walks up to 96 of them; the "dead row" source of dHidden tiles).
// `what` 1: hidden (before the forward), 2: G (after the forward, whose last tile writes logits there)
void launch_x2_zero_padding(const X3Args &a, int what, hipStream_t st)
{
    const long cells = (long)a.B * a.T * a.U1;
    const size_t pad = (size_t)(a.rows_alloc - cells);
    if (what & 1)
        for (int p = 0; p < 2; ++p) launch_fill32(a.hidden + p * a.plane_stride + cells * a.H, 0u, pad * a.H * 2, st);
    if (what & 2) launch_fill32(a.logits + cells * a.V, 0u, pad * a.V * 4, st);
}

// ---------------------------------------------------------------------------------------
// k_dw_x2: dW[v,h] = sum_c G[c,v] hidden[c,h] (split-K slabs), db[v] = sum_c G[c,v] — k_dw_x3's design on two planes:
// 4 waves = 2 (M) x 2 (N), workgroup tile 256 v x 256 h, wave 128 x 128 = 16 accumulator tiles (256 registers).  Both
// operands row-major with K (the cell) as the ROW, two fp16 planes each:
//  * HBM -> LDS by LDS-DMA, ring of XW2_NST (4) stages of 16 cells x (256 v + 256 h) x 2 planes = 32 KiB; wave w fills operand tile w
//    (0,1: the 128-column halves of the G tile, 2,3: of the hidden tile), 8 DMAs of 1 KiB (4 rows x 256 B) per stage;
//  * LDS -> VGPR by ds_read_b64_tr_b16 (4x16 transpose read): two reads give a lane its 8 consecutive cells of one column;
//  * tile image: 256-byte rows, 16-byte chunk ch of row r at 16*(ch ^ swz(r)), swz(r) = ((r&3)<<2) | ((r>>2)&3), applied
//    on the DMA's SOURCE side (the DMA writes LDS linearly).
// One k-step = 16 cells = 3 products x 16 tiles = 48 MFMAs (1536 matrix-pipe cycles) against 32 KiB staged.  Per k-step:
// counted vmcnt + one barrier publish stage ks, the 8 DMAs of stage ks+2 go into the slot of ks-1 threaded through the
// MFMAs (3 + 3 + 2), fragment reads run one product ahead of their MFMAs.  Products ah.bh, am.bh, ah.bm.
// The accumulators hold g_scale x 2^14 x dW: the epilogue multiplies by X3Args::dw_rescale (a power of two).
// ---------------------------------------------------------------------------------------
#define XW2_ROWS 16
#ifndef XW2_NST
#define XW2_NST 4   // ring stages: the DMAs of stage ks + NST - 1 are issued during k-step ks
#endif
#define XW2_PLANE 4096            // one operand tile of one plane: 16 rows x 256 B
#define XW2_STAGE (2 * XW2_PLANE)  // one stage of one operand tile
#define XW2_TILE (XW2_NST * XW2_STAGE)  // ring of one operand tile: [stage][plane][16 x 256 B] = 24 KiB
#define XW2_GRAN 32  // granule of the live-row table (shared with the bf16 routes: 2 k-steps)

struct X2Frag { u32x2 lo[4], hi[4]; };  // 4 tiles: cells 0-3 / 4-7 of a lane's 8
#define X2_LANDED(f, N)                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(" #N ")"                                                                     \
                 : "+v"(f.lo[0]), "+v"(f.lo[1]), "+v"(f.lo[2]), "+v"(f.lo[3]), "+v"(f.hi[0]), "+v"(f.hi[1]),     \
                   "+v"(f.hi[2]), "+v"(f.hi[3])                                                                  \
                 :: "memory")

// NW = waves per workgroup.  4: one wave per SIMD, wave tile 128 v x 128 h.  8: TWO waves per SIMD (256 registers each), wave
// tile 128 v x 64 h = 8 accumulator tiles: what one wave waits for — the barrier, a counted vmcnt, the ~80 cycles each of its
// LDS-DMA issues blocks it — the SIMD's other wave fills with MFMAs.  MEASURED EQUAL (15.5 ms both, cfg2): the kernel is bound by
// the bytes it stages — 32 KiB per k-step and CU at ~12.6 B/clk/CU (6.8 TB/s over the chip out of L2), the same rate the forward's
// and dHidden's W streams reach — not by issue stalls.  4 is the default; 8 stays behind RNNT_VARIANT_X2_DW_8W.
// PARTH (H % 256 == 128: the last h block is half empty — H = 640): the waves of the dead 128-column half run no product MFMAs and the
// dead hidden tile's DMAs are requested past their buffer's range (no bytes moved; the instruction and its vmcnt stay).
template <int NW, bool PARTH = false>
__global__ __launch_bounds__(64 * NW, 1) void k_dw_x2(X3Args a)
{
    constexpr int WN = NW / 2;      // waves along h
    constexpr int QN = 8 / WN;      // 32-column h tiles per wave (4 or 2)
    constexpr int ND = 32 / NW;     // DMA pieces per wave and stage (8 or 4)
    extern __shared__ __attribute__((aligned(1024))) char s_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
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
    const long *tab = a.dw_tab;
    const int B = a.B;
    const long nlive = tab[2 * B + 1];
    const long g_lo = nlive * split / a.n_split, g_hi = nlive * (split + 1) / a.n_split;
    X2_CLOCK_STAMP(128 + 104);

    f32x16 acc[4][QN];
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int qn = 0; qn < QN; ++qn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;
    // db rides the matrix pipe as in k_dw_x3: one more MFMA per plane and k-step against a column SELECTOR of ones, for one
    // (one h block: two) of the wave's M tiles, into a 17th accumulator tile held in VGPRs
    // (8 waves: the four wn of h block 0 take the four M tiles of their wm half)
    const bool do_b = NW == 8 ? hb == 0 : hb < 2;  // workgroup-uniform
    const int bsel0 = NW == 8 ? wn : (n_hblk >= 2 ? (hb & 1) * 2 + wn : wn);
    const bool two_b = NW == 4 && n_hblk < 2;  // one h block, 4 waves: each wave takes two tiles, bsel0 and bsel0 + 2
    f32x16 dacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
    const unsigned sel0 = (lane & 31) == 0 ? 0x3c003c00u : 0u, sel1 = (lane & 31) == 1 ? 0x3c003c00u : 0u;  // fp16 ones

    if (g_hi > g_lo) {
        // ---- DMA source of this wave's operand tile (8 waves: two waves per operand tile, one plane each)
        const int otile = NW == 8 ? wave >> 1 : wave;
        const bool is_g = otile < 2;
        int col0 = (is_g ? vb : hb) * 256 + 128 * (otile & 1);
        const bool dead_tile = PARTH && !is_g && col0 >= H;  // (wave-uniform)
        if (col0 >= (is_g ? V : H)) col0 = 0;  // tile beyond the matrix: never stored, read something valid
        const char *pbase[2];  // plane p of this wave's operand: base pointer (wave-uniform), row stride in bytes
        long rstride;
        if (is_g) {
            pbase[0] = (const char *)a.logits + 4L * col0;        // hi: first 64 bytes of each 128-byte chunk
            pbase[1] = (const char *)a.logits + 4L * col0 + 64;   // mid: last 64
            rstride = 4L * V;
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) pbase[p] = (const char *)(a.hidden + p * a.plane_stride) + 2L * col0;
            rstride = 2L * H;
        }
        // DMA i (0..3) of a plane's 16 rows: rows 4i .. 4i+3; lane L: row 4i + (L>>4), LDS chunk position L&15
        // <- global chunk jg = (L&15) ^ swz(row), swz = ((L>>4)<<2) | (i&3)
        int soff[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int jg = (lane & 15) ^ (((lane >> 4) << 2) | (i & 3));
            // interleaved planes (G): chunk jg of the plane is 16 bytes at 128*(jg>>2) + 16*(jg&3)
            const int cb = is_g ? 128 * (jg >> 2) + 16 * (jg & 3) : 16 * jg;
            soff[i] = dead_tile ? 0x7ffffff0 : (int)((4 * i + (lane >> 4)) * rstride) + cb;
#if XF2_IMM  // the four pieces of a plane share one LDS base (M0): piece i carries the immediate offset 1024 i, which advances the memory
            // address too — taken back out of its per-lane offset (rstride >= 256 bytes: 4 i rows >= 1024 i bytes, never negative)
            if (NW == 4 && !dead_tile) soff[i] -= 1024 * i;
#endif
        }
        const bool live_n = !PARTH || hb * 256 + wn * 32 * QN < H;  // this wave's h columns exist (wave-uniform)
        long row_first = 0;  // first cell of the range being walked
        // ---- transposed fragment reads.  Fragment of 32-column tile m: lane (g = lane>>4, q = (lane&15)>>2,
        // p = lane&3) reads rows 8(g>>1) + 4sec + q at chunk 4m + 2(g&1) + (p>>1), +8(p&1) bytes, sec = 0,1.
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, hh = g >> 1;
        const int lds0 = (int)(size_t)(lds_vptr)s_ring;
        int abase[4][2], bbase[QN][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int sec = 0; sec < 2; ++sec) {
                const int row = 8 * hh + 4 * sec + q;
                const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
                const int ch = 4 * m + 2 * (g & 1) + (pp >> 1);
                abase[m][sec] = lds0 + wm * XW2_TILE + 256 * row + 16 * (ch ^ swz) + 8 * (pp & 1);
                if (m < QN) {  // this wave's h tiles: columns 32 QN wn + 32 m of the 256 = operand tile 2 + (col >> 7), tile (col & 127) / 32
                    const int col = 32 * QN * wn + 32 * m;
                    const int chb = 4 * ((col & 127) >> 5) + 2 * (g & 1) + (pp >> 1);
                    bbase[m < QN ? m : 0][sec] = lds0 + (2 + (col >> 7)) * XW2_TILE + 256 * row + 16 * (chb ^ swz) + 8 * (pp & 1);
                }
            }
        int sbase[2][2];  // db: fragment bases of the M tile(s) this wave sums (bsel0, and bsel0 + 2 with one h block)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int sec = 0; sec < 2; ++sec) {
                const int m = (bsel0 + 2 * k) & 3;
                const int row = 8 * hh + 4 * sec + q;
                const int ch = 4 * m + 2 * (g & 1) + (pp >> 1);
                const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
                sbase[k][sec] = lds0 + wm * XW2_TILE + 256 * row + 16 * (ch ^ swz) + 8 * (pp & 1);
            }

        // one k-step on ring stage ST (compile-time: every LDS offset is an immediate)
        auto kstep = [&](auto st_c, long ks, f32x16 &dacc) {
            constexpr int ST = decltype(st_c)::value, DST = (ST + XW2_NST - 1) % XW2_NST;
            // the 16 rows of stage ks+2 as two raw buffers (wave-uniform base; the per-lane part is the 32-bit soff)
            __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
                rs[p] = __builtin_amdgcn_make_buffer_rsrc((void *)(pbase[p] + (row_first + (ks + XW2_NST - 1) * XW2_ROWS) * rstride), 0,
                                                          (int)(XW2_ROWS * rstride), 0x00020000);
            auto dma_piece = [&](auto n_c) {  // piece n of this wave's share of stage ks+NST-1 -> ring stage DST: plane n>>2 (8 waves: wave & 1), rows 4(n&3)..
                constexpr int n = decltype(n_c)::value, i = n & 3;
                if (NW == 8) {
                    if (wave & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[1], (lds_vptr)(s_ring + otile * XW2_TILE + DST * XW2_STAGE + XW2_PLANE + 1024 * i), 16, soff[i], 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[0], (lds_vptr)(s_ring + otile * XW2_TILE + DST * XW2_STAGE + 1024 * i), 16, soff[i], 0, 0, 0);
                } else {
                    constexpr int p = (n >> 2) & 1;
#if XF2_IMM
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[p], (lds_vptr)(s_ring + otile * XW2_TILE + DST * XW2_STAGE + p * XW2_PLANE),
                                                             16, soff[i], 0, 1024 * i, (X2_NT & 2) ? 2 : 0);
#else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[p], (lds_vptr)(s_ring + otile * XW2_TILE + DST * XW2_STAGE + p * XW2_PLANE + 1024 * i),
                                                             16, soff[i], 0, 0, (X2_NT & 2) ? 2 : 0);  // (experiment 2: aux = 2, non-temporal)
#endif
                }
            };
            // 8 transposed reads of plane P of the A / B operand (inline asm: hipcc guards every LDS read it can see behind
            // an LDS-DMA with vmcnt(0)); results are used only after X2_LANDED named them
            auto reads = [&](X2Frag &f, const auto &base, auto p_c, auto nt_c) {
                constexpr int off = ST * XW2_STAGE + decltype(p_c)::value * XW2_PLANE;
#pragma unroll
                for (int m = 0; m < decltype(nt_c)::value; ++m) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo[m]) : "v"(base[m][0]), "n"(off));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi[m]) : "v"(base[m][1]), "n"(off));
                }
            };
            // 16 MFMAs of one product with DMA pieces N0 .. N0+CNT-1 threaded through them
            auto product = [&](const X2Frag &fa_, const X2Frag &fb_, auto n0_c, auto cnt_c) {
                constexpr int N0 = decltype(n0_c)::value, CNT = decltype(cnt_c)::value;
                u32x4 fa[4], fb[QN];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    fa[m] = u32x4{fa_.lo[m][0], fa_.lo[m][1], fa_.hi[m][0], fa_.hi[m][1]};
                    if (m < QN) fb[m < QN ? m : 0] = u32x4{fb_.lo[m][0], fb_.lo[m][1], fb_.hi[m][0], fb_.hi[m][1]};
                }
#pragma unroll
                for (int qm = 0; qm < 4; ++qm) {
                    if (live_n) {
#pragma unroll
                        for (int qn = 0; qn < QN; ++qn) acc[qm][qn] = x2_mfma(fa[qm], fb[qn], acc[qm][qn]);
                    }
                    if (qm == 0 && CNT >= 1) dma_piece(X2Int<N0>{});
                    if (qm == 1 && CNT >= 2) dma_piece(X2Int<N0 + (CNT >= 2 ? 1 : 0)>{});
                    if (qm == 3 && CNT == 3) dma_piece(X2Int<N0 + (CNT == 3 ? 2 : 0)>{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            auto bias_read = [&](u32x2 &lo, u32x2 &hi, auto p_c, int k) {
                constexpr int off = ST * XW2_STAGE + decltype(p_c)::value * XW2_PLANE;
                const int b0 = sbase[k][0], b1 = sbase[k][1];  // (locals: asm operands cannot name a capture of the enclosing generic lambda)
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(b0), "n"(off));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(b1), "n"(off));
            };
            auto bias_mfma = [&](u32x2 &lo, u32x2 &hi, f32x16 &dacc, int k) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) :: "memory");
                const u32x4 fa = {lo[0], lo[1], hi[0], hi[1]};
                const unsigned sv = k ? sel1 : sel0;
                const u32x4 sel = {sv, sv, sv, sv};
                // accumulator in VGPRs, spelled as asm (left to hipcc the 17th tile is shuttled through the full AGPR file)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(dacc) : "v"(fa), "v"(sel));
            };
            // stage ks landed (the 8 (NST - 2) younger pieces of the stages after it may still fly); every wave is past its reads of
            // stage ks-1, whose ring stage the DMAs below refill
            if (ND * (XW2_NST - 2) == 16) asm volatile(RNNT_VMCNT(16) ::: "memory");
            else if (ND * (XW2_NST - 2) == 8) asm volatile(RNNT_VMCNT(8) ::: "memory");
            else asm volatile(RNNT_VMCNT(4) ::: "memory");
            x2_lds_barrier();
            X2Frag Ah, Bh, Am, Bm;
            u32x2 dl[2], dh[2];
            reads(Ah, abase, X2Int<0>{}, X2Int<4>{});
            reads(Bh, bbase, X2Int<0>{}, X2Int<QN>{});
            reads(Am, abase, X2Int<1>{}, X2Int<4>{});
            X2_LANDED(Ah, 8);
            X2_LANDED(Bh, 8);
            // DMA pieces of the k-step: 3 + 3 + 2 (8 waves: 2 + 1 + 1)
            product(Ah, Bh, X2Int<0>{}, X2Int<(NW == 8 ? 2 : 3)>{});
            reads(Bm, bbase, X2Int<1>{}, X2Int<QN>{});
            if (QN == 4) X2_LANDED(Am, 8); else X2_LANDED(Am, 4);
            product(Am, Bh, X2Int<(NW == 8 ? 2 : 3)>{}, X2Int<(NW == 8 ? 1 : 3)>{});
            X2_LANDED(Bm, 0);
            if (do_b) { bias_read(dl[0], dh[0], X2Int<0>{}, 0); bias_read(dl[1], dh[1], X2Int<1>{}, 0); }
            product(Ah, Bm, X2Int<(NW == 8 ? 3 : 6)>{}, X2Int<(NW == 8 ? 1 : 2)>{});
            if (do_b) {
                bias_mfma(dl[0], dh[0], dacc, 0); bias_mfma(dl[1], dh[1], dacc, 0);
                if (two_b) {  // one h block: the wave's second tile (column 1 of the selector product), H <= 256 only
                    bias_read(dl[0], dh[0], X2Int<0>{}, 1); bias_read(dl[1], dh[1], X2Int<1>{}, 1);
                    bias_mfma(dl[0], dh[0], dacc, 1); bias_mfma(dl[1], dh[1], dacc, 1);
                }
            }
        };
        auto dma_stage = [&](long ks, int st) {  // pipeline prologue: this wave's pieces of stage ks
#pragma unroll
            for (int n = 0; n < ND; ++n) {
                const int p = NW == 8 ? (wave & 1) : n >> 2, i = n & 3;
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(pbase[p] + (row_first + ks * XW2_ROWS) * rstride), 0, (int)(XW2_ROWS * rstride), 0x00020000);
#if XF2_IMM
                if (NW == 4) {
                    if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(s_ring + otile * XW2_TILE + st * XW2_STAGE + p * XW2_PLANE), 16, soff[i], 0, 0, 0);
                    if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(s_ring + otile * XW2_TILE + st * XW2_STAGE + p * XW2_PLANE), 16, soff[i], 0, 1024, 0);
                    if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(s_ring + otile * XW2_TILE + st * XW2_STAGE + p * XW2_PLANE), 16, soff[i], 0, 2048, 0);
                    if (i == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(s_ring + otile * XW2_TILE + st * XW2_STAGE + p * XW2_PLANE), 16, soff[i], 0, 3072, 0);
                } else
#endif
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(s_ring + otile * XW2_TILE + st * XW2_STAGE + p * XW2_PLANE + 1024 * i),
                                                         16, soff[i], 0, 0, 0);
            }
        };

        // Soft lockstep of the tiles of a split (k_dw_bf16's, round 3).  They walk the same k-steps and share every operand byte
        // through their XCD's L2 (G between the h blocks, hidden between the v blocks); nothing else holds them together, and at
        // this kernel's pace a tile that falls behind finds its lines evicted and becomes its own HBM stream — counted without
        // it: 83 GB fetched for 39.5 GB of operands (profiles/r04_f16x2_hbm_traffic_pmc.txt, first pass).  Every NST k-steps a
        // tile publishes its k-step count and looks at its right-hand neighbour's in the ring of the split's tiles (one coherent
        // scalar load, issued a look before its value is used); a tile more than DW_LAG k-steps ahead of that neighbour naps.
        // Bounded: after DW_NAPS naps without the neighbour catching up (a partner that is not resident) the tile stops looking,
        // so every wave reaches the end whatever the others do.
        constexpr int DW_LAG = 6, DW_NAPS = 256;
        int *prog = a.dw_prog ? a.dw_prog + split * 16 : nullptr;
        bool sync_on = prog != nullptr && tiles > 1 && tiles <= 16;
        const int *nb = prog ? prog + (tile + 1 < tiles ? tile + 1 : 0) : nullptr;  // the neighbour's word
        int nb_at = 0x7fffffff;  // the neighbour's k-step count as of the last look
        int done = 0;            // k-steps behind this workgroup, over all ranges
        auto lockstep = [&](int mine) {  // (wave-uniform)
            // the value requested at the previous look has long landed; the wait names nb_at so that no copy of the register made before
            // the data arrived can be what the comparison reads (round-4 advice: a stale value only mis-paces, but makes timing and traffic
            // depend on the compiler's register copies)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nb_at) :: "memory");
            int naps = 0;
            while (sync_on && nb_at + DW_LAG + XW2_NST < mine) {  // (+NST: the value is one look old)
                if (++naps > DW_NAPS) { sync_on = false; break; }
                __builtin_amdgcn_s_sleep(4);
                asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(nb_at) : "s"(nb) : "memory");
            }
            if (sync_on) {
                if (tid == 0) __hip_atomic_store(prog + tile, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (one more outstanding store only makes the counted waits stricter)
                asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(nb_at) : "s"(nb) : "memory");  // used at the next look
            }
        };
        int ub = 0;
        while (ub + 1 < B && tab[B + 1 + ub + 1] <= g_lo) ++ub;
        for (long gq = g_lo; gq < g_hi; ++ub) {  // workgroup-uniform: one pipeline run per live range
            const long cum0 = tab[B + 1 + ub], cum1 = ub + 1 < B ? tab[B + 1 + ub + 1] : nlive;
            const long ge = cum1 < g_hi ? cum1 : g_hi;
            if (ge <= gq) continue;
            const long nks = 2 * (ge - gq);  // 16-cell k-steps of this range
            row_first = (tab[ub] + (gq - cum0)) * XW2_GRAN;
            gq = ge;
            dma_stage(0, 0);
            dma_stage(1, 1);
            if (XW2_NST == 4) dma_stage(2, 2);
            for (long ks = 0;;) {  // the ring stage of a k-step is ks % NST: unrolled by NST
                if (ks >= nks) break;
                lockstep(done + (int)ks);
                kstep(X2Int<0>{}, ks, dacc); ++ks;
                if (ks >= nks) break;
                kstep(X2Int<1>{}, ks, dacc); ++ks;
                if (ks >= nks) break;
                kstep(X2Int<2>{}, ks, dacc); ++ks;
                if (XW2_NST == 4) {
                    if (ks >= nks) break;
                    kstep(X2Int<3 % XW2_NST>{}, ks, dacc); ++ks;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the over-issued DMAs before the ring
            x2_lds_barrier();                                  // is refilled / the kernel exits
        }
    }

    // ---- epilogue: partial slab [split][V,H]; bias partial [split][V].  Accumulator register r of tile
    // (qm,qn): v = v0 + 32qm + (r&3) + 8(r>>2) + 4half, h = h0 + 32qn + (lane&31).
    X2_CLOCK_STAMP(128 + 106);
    const int v0 = vb * 256 + wm * 128, h0 = hb * 256 + wn * 32 * QN;
    float *sw = a.slab_w + (long)split * V * H;
    const float rw = a.dw_rescale, rb = a.db_rescale;
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + 32 * qm + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (v < V) {
#pragma unroll
                for (int qn = 0; qn < QN; ++qn) {
                    const int h = h0 + 32 * qn + (lane & 31);
                    if (h < H) sw[(long)v * H + h] = acc[qm][qn][r] * rw;
                }
            }
        }
    if (do_b && (lane & 31) < (two_b ? 2 : 1)) {  // column k of the selector products: lanes k / 32+k store
        const int m = bsel0 + 2 * (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (v < V) a.slab_b[(long)split * V + v] = dacc[r] * rb;
        }
    }
}


// ---------------------------------------------------------------------------------------
// k_dw_x2m (round 6): dW for H % 256 == 128 with at least two whole 256-column h blocks and an even number of v blocks (config 4:
// H = 640, V = 1024).  k_dw_x2<4, true> covers the odd last 128 columns with one half-empty 256 x 256 tile per v block: two of its four
// waves run no MFMAs, the workgroup naps in the split's lockstep, and the split count is sized for 3 tiles per v block although only 2.5
// tiles' worth of products exist (config 4: 12 tiles x 21 splits, dW 13 % over its time per cell at config 2).  Here the odd block of TWO v
// blocks is one TALL tile — 512 v x 128 h, four waves stacked along v, wave tile 128 x 128 as everywhere (16 accumulator tiles, 48 MFMAs per
// k-step: every wave of every workgroup works) — beside the ordinary 256 x 256 tiles of the whole h blocks: (V/256) (H/256) + V/512 tiles
// per split (config 4: 10 tiles x 25 splits).  A tall tile stages 4 G operand tiles (one per wave: the wave's own A rows) and 1 hidden
// operand tile (its 8 DMA pieces per stage shared out two per wave): 40 KiB per k-step instead of 32, a 5-tile ring = 160 KiB of LDS.
// The kernel's two tile kinds are two complete copies of the loop behind ONE workgroup-uniform branch (a branch inside the k loop would split
// the hand-placed MFMA / DMA / read schedule into scheduling regions); everything else — ring stage layout, source-side swizzle, transposed
// reads, the three products, db on the matrix pipe (whole h blocks 0 and 1 only), live-row table, soft lockstep, slab epilogue — is
// k_dw_x2<4>'s, statement for statement.
// ---------------------------------------------------------------------------------------
#ifndef X2_DW_MIXED
#define X2_DW_MIXED 1  // 0 (diagnostic builds): k_dw_x2<4, true> everywhere, round 5's form — the A/B partner of k_dw_x2m
#endif
int x2_dw_mixed_ok(int H, int V) { return X2_DW_MIXED && H % 256 == 128 && H >= 640 && V % 512 == 0; }
int x2_dw_tiles(int H, int V) { return x2_dw_mixed_ok(H, V) ? (V / 256) * (H / 256) + V / 512 : ((V + 255) / 256) * ((H + 255) / 256); }

__global__ __launch_bounds__(256, 1) void k_dw_x2m(X3Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char s_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int H = a.H, V = a.V;
    const int n_vblk = V / 256, n_fh = H / 256;  // whole h blocks; the odd block = columns 256 n_fh .. +127
    const int n_full = n_vblk * n_fh, tiles = n_full + n_vblk / 2;
    const int total = tiles * a.n_split;
    int id = blockIdx.x;  // XCD-aware remap: the tiles of one split share an XCD's L2
    {
        const int q8 = total / 8, r8 = total % 8, x = id % 8;
        id = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + id / 8;
    }
    const int tile = id % tiles, split = id / tiles;
    const long *tab = a.dw_tab;
    const int B = a.B;
    const long nlive = tab[2 * B + 1];
    const long g_lo = nlive * split / a.n_split, g_hi = nlive * (split + 1) / a.n_split;
    const unsigned sel0 = (lane & 31) == 0 ? 0x3c003c00u : 0u;  // fp16 ones in column 0 of the selector fragment

    auto body = [&](auto tall_c) {
        constexpr bool TALL = decltype(tall_c)::value != 0;
        // tile geometry.  Whole-block tile: (vb, hb), waves 2 (v) x 2 (h).  Tall tile: v blocks 2 vt, 2 vt + 1, the odd h block, waves 4 (v) x 1
        const int vb = TALL ? 0 : tile / n_fh, hb = TALL ? n_fh : tile % n_fh, vt = TALL ? tile - n_full : 0;
        const int wm = TALL ? wave : wave >> 1, wn = TALL ? 0 : wave & 1;
        const int v0 = TALL ? vt * 512 + wave * 128 : vb * 256 + wm * 128;
        const int h0 = TALL ? n_fh * 256 : hb * 256 + wn * 128;

        f32x16 acc[4][4];
#pragma unroll
        for (int qm = 0; qm < 4; ++qm)
#pragma unroll
            for (int qn = 0; qn < 4; ++qn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;
        // db as in k_dw_x2<4>: the waves of the whole-block tiles of h blocks 0 and 1 each take one of their wm half's four M tiles
        const bool do_b = !TALL && hb < 2;  // workgroup-uniform
        const int bsel0 = (hb & 1) * 2 + wn;
        f32x16 dacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[r] = 0.f;

        if (g_hi > g_lo) {
            // ---- DMA sources.  Own operand tile (8 pieces per stage): whole-block tile: waves 0, 1 the G tile's 128-column halves, waves 2, 3
            // the hidden tile's; tall tile: every wave the 128 G columns of its own A rows.  Tall tile, in addition: pieces 2 (wave & 1),
            // 2 (wave & 1) + 1 of plane wave >> 1 of the ONE hidden operand tile all four waves multiply by.
            const bool is_g = TALL || wave < 2;
            const int col0 = TALL ? v0 : (is_g ? vb : hb) * 256 + 128 * (wave & 1);
            const char *pbase[2];
            long rstride;
            if (is_g) {
                pbase[0] = (const char *)a.logits + 4L * col0;        // hi: first 64 bytes of each 128-byte chunk
                pbase[1] = (const char *)a.logits + 4L * col0 + 64;   // mid: last 64
                rstride = 4L * V;
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p) pbase[p] = (const char *)(a.hidden + p * a.plane_stride) + 2L * col0;
                rstride = 2L * H;
            }
            int soff[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int jg = (lane & 15) ^ (((lane >> 4) << 2) | (i & 3));
                const int cb = is_g ? 128 * (jg >> 2) + 16 * (jg & 3) : 16 * jg;
                soff[i] = (int)((4 * i + (lane >> 4)) * rstride) + cb - 1024 * i;  // (the piece's immediate offset 1024 i advances the memory address too)
            }
            const int xp = wave >> 1, xi0 = 2 * (wave & 1);  // tall: this wave's two pieces of the hidden tile — plane xp, pieces xi0, xi0 + 1
            const char *xbase = (const char *)(a.hidden + xp * a.plane_stride) + 2L * h0;
            const long xstride = 2L * H;
            int xoff[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = xi0 + k;
                const int jg = (lane & 15) ^ (((lane >> 4) << 2) | (i & 3));
                xoff[k] = (int)((4 * i + (lane >> 4)) * xstride) + 16 * jg;
            }
            long row_first = 0;
            // ---- transposed fragment reads (k_dw_x2's): A from operand tile wm (tall: wave), B from the hidden tile(s)
            const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, hh = g >> 1;
            const int lds0 = (int)(size_t)(lds_vptr)s_ring;
            const int a_tile = TALL ? wave : wm;            // operand tile holding this wave's A rows
            const int own_tile = TALL ? wave : wave;        // operand tile this wave stages
            int abase[4][2], bbase[4][2], sbase[2];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int sec = 0; sec < 2; ++sec) {
                    const int row = 8 * hh + 4 * sec + q;
                    const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
                    const int ch = 4 * m + 2 * (g & 1) + (pp >> 1);
                    abase[m][sec] = lds0 + a_tile * XW2_TILE + 256 * row + 16 * (ch ^ swz) + 8 * (pp & 1);
                    bbase[m][sec] = lds0 + (TALL ? 4 : 2 + wn) * XW2_TILE + 256 * row + 16 * (ch ^ swz) + 8 * (pp & 1);
                    if (m == (bsel0 & 3)) sbase[sec] = abase[m][sec];
                }

            auto kstep = [&](auto st_c, long ks, f32x16 &dacc) {
                constexpr int ST = decltype(st_c)::value, DST = (ST + XW2_NST - 1) % XW2_NST;
                __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    rs[p] = __builtin_amdgcn_make_buffer_rsrc((void *)(pbase[p] + (row_first + (ks + XW2_NST - 1) * XW2_ROWS) * rstride), 0,
                                                              (int)(XW2_ROWS * rstride), 0x00020000);
                const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(xbase + (row_first + (ks + XW2_NST - 1) * XW2_ROWS) * xstride), 0, (int)(XW2_ROWS * xstride), 0x00020000);
                auto dma_piece = [&](auto n_c) {  // piece n of this wave's share of stage ks+NST-1 -> ring stage DST
                    constexpr int n = decltype(n_c)::value;
                    if constexpr (n < 8) {
                        constexpr int i = n & 3, p = (n >> 2) & 1;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[p], (lds_vptr)(s_ring + own_tile * XW2_TILE + DST * XW2_STAGE + p * XW2_PLANE),
                                                                 16, soff[i], 0, 1024 * i, 0);
                    } else {  // (tall) the hidden tile: operand tile 4
                        constexpr int k = n - 8;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_vptr)(s_ring + 4 * XW2_TILE + DST * XW2_STAGE + xp * XW2_PLANE + 1024 * (xi0 + k)),
                                                                 16, xoff[k], 0, 0, 0);
                    }
                };
                auto reads = [&](X2Frag &f, const auto &base, auto p_c) {
                    constexpr int off = ST * XW2_STAGE + decltype(p_c)::value * XW2_PLANE;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo[m]) : "v"(base[m][0]), "n"(off));
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi[m]) : "v"(base[m][1]), "n"(off));
                    }
                };
                // 16 MFMAs of one product with DMA pieces N0 .. N0+CNT-1 threaded through them (CNT <= 4: one per row of wave tiles)
                auto product = [&](const X2Frag &fa_, const X2Frag &fb_, auto n0_c, auto cnt_c) {
                    constexpr int N0 = decltype(n0_c)::value, CNT = decltype(cnt_c)::value;
                    u32x4 fa[4], fb[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        fa[m] = u32x4{fa_.lo[m][0], fa_.lo[m][1], fa_.hi[m][0], fa_.hi[m][1]};
                        fb[m] = u32x4{fb_.lo[m][0], fb_.lo[m][1], fb_.hi[m][0], fb_.hi[m][1]};
                    }
#pragma unroll
                    for (int qm = 0; qm < 4; ++qm) {
#pragma unroll
                        for (int qn = 0; qn < 4; ++qn) acc[qm][qn] = x2_mfma(fa[qm], fb[qn], acc[qm][qn]);
                        if (qm == 0 && CNT >= 1) dma_piece(X2Int<N0>{});
                        if (qm == 1 && CNT >= 2) dma_piece(X2Int<N0 + (CNT >= 2 ? 1 : 0)>{});
                        if (qm == 2 && CNT >= 4) dma_piece(X2Int<N0 + (CNT >= 4 ? 2 : 0)>{});
                        if (qm == 3 && CNT >= 3) dma_piece(X2Int<N0 + (CNT >= 3 ? CNT - 1 : 0)>{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                auto bias_read = [&](u32x2 &lo, u32x2 &hi, auto p_c) {
                    constexpr int off = ST * XW2_STAGE + decltype(p_c)::value * XW2_PLANE;
                    const int b0 = sbase[0], b1 = sbase[1];
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(b0), "n"(off));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(b1), "n"(off));
                };
                auto bias_mfma = [&](u32x2 &lo, u32x2 &hi, f32x16 &dacc) {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi) :: "memory");
                    const u32x4 fa = {lo[0], lo[1], hi[0], hi[1]};
                    const u32x4 sel = {sel0, sel0, sel0, sel0};
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(dacc) : "v"(fa), "v"(sel));
                };
                // stage ks landed (the ND (NST - 2) younger pieces of the stages after it may still fly); every wave is past its reads of
                // stage ks-1, whose ring stage the DMAs below refill
                static_assert(XW2_NST == 4, "counted waits below: two stages in flight behind the one waited for");
                if (TALL) asm volatile(RNNT_VMCNT(20) ::: "memory");
                else asm volatile(RNNT_VMCNT(16) ::: "memory");
                x2_lds_barrier();
                X2Frag Ah, Bh, Am, Bm;
                u32x2 dl[2], dh[2];
                reads(Ah, abase, X2Int<0>{});
                reads(Bh, bbase, X2Int<0>{});
                reads(Am, abase, X2Int<1>{});
                X2_LANDED(Ah, 8);
                X2_LANDED(Bh, 8);
                product(Ah, Bh, X2Int<0>{}, X2Int<(TALL ? 4 : 3)>{});
                reads(Bm, bbase, X2Int<1>{});
                X2_LANDED(Am, 8);
                product(Am, Bh, X2Int<(TALL ? 4 : 3)>{}, X2Int<3>{});
                X2_LANDED(Bm, 0);
                if (do_b) { bias_read(dl[0], dh[0], X2Int<0>{}); bias_read(dl[1], dh[1], X2Int<1>{}); }
                product(Ah, Bm, X2Int<(TALL ? 7 : 6)>{}, X2Int<(TALL ? 3 : 2)>{});
                if (do_b) { bias_mfma(dl[0], dh[0], dacc); bias_mfma(dl[1], dh[1], dacc); }
            };
            auto dma_stage = [&](long ks, int st) {  // pipeline prologue: this wave's pieces of stage ks
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const int p = n >> 2, i = n & 3;
                    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(pbase[p] + (row_first + ks * XW2_ROWS) * rstride), 0, (int)(XW2_ROWS * rstride), 0x00020000);
                    lds_vptr d = (lds_vptr)(s_ring + own_tile * XW2_TILE + st * XW2_STAGE + p * XW2_PLANE);
                    if (i == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d, 16, soff[i], 0, 0, 0);
                    if (i == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d, 16, soff[i], 0, 1024, 0);
                    if (i == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d, 16, soff[i], 0, 2048, 0);
                    if (i == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d, 16, soff[i], 0, 3072, 0);
                }
                if (TALL) {
                    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(xbase + (row_first + ks * XW2_ROWS) * xstride), 0, (int)(XW2_ROWS * xstride), 0x00020000);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(s_ring + 4 * XW2_TILE + st * XW2_STAGE + xp * XW2_PLANE + 1024 * (xi0 + k)),
                                                                 16, xoff[k], 0, 0, 0);
                }
            };

            // soft lockstep of the tiles of a split (k_dw_x2's; tall and whole-block tiles walk the same k-steps and share G through L2)
            constexpr int DW_LAG = 6, DW_NAPS = 256;
            int *prog = a.dw_prog ? a.dw_prog + split * 16 : nullptr;
            bool sync_on = prog != nullptr && tiles > 1 && tiles <= 16;
            const int *nb = prog ? prog + (tile + 1 < tiles ? tile + 1 : 0) : nullptr;
            int nb_at = 0x7fffffff;
            int done = 0;
            auto lockstep = [&](int mine) {  // (wave-uniform)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(nb_at) :: "memory");
                int naps = 0;
                while (sync_on && nb_at + DW_LAG + XW2_NST < mine) {
                    if (++naps > DW_NAPS) { sync_on = false; break; }
                    __builtin_amdgcn_s_sleep(4);
                    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(nb_at) : "s"(nb) : "memory");
                }
                if (sync_on) {
                    if (tid == 0) __hip_atomic_store(prog + tile, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_load_dword %0, %1, 0x0 glc" : "=s"(nb_at) : "s"(nb) : "memory");
                }
            };
            int ub = 0;
            while (ub + 1 < B && tab[B + 1 + ub + 1] <= g_lo) ++ub;
            for (long gq = g_lo; gq < g_hi; ++ub) {  // workgroup-uniform: one pipeline run per live range
                const long cum0 = tab[B + 1 + ub], cum1 = ub + 1 < B ? tab[B + 1 + ub + 1] : nlive;
                const long ge = cum1 < g_hi ? cum1 : g_hi;
                if (ge <= gq) continue;
                const long nks = 2 * (ge - gq);  // 16-cell k-steps of this range
                row_first = (tab[ub] + (gq - cum0)) * XW2_GRAN;
                gq = ge;
                dma_stage(0, 0);
                dma_stage(1, 1);
                dma_stage(2, 2);
                for (long ks = 0;;) {  // the ring stage of a k-step is ks % 4: unrolled by 4
                    if (ks >= nks) break;
                    lockstep(done + (int)ks);
                    kstep(X2Int<0>{}, ks, dacc); ++ks;
                    if (ks >= nks) break;
                    kstep(X2Int<1>{}, ks, dacc); ++ks;
                    if (ks >= nks) break;
                    kstep(X2Int<2>{}, ks, dacc); ++ks;
                    if (ks >= nks) break;
                    kstep(X2Int<3>{}, ks, dacc); ++ks;
                }
                done += (int)nks;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the over-issued DMAs before the ring
                x2_lds_barrier();                                  // is refilled / the kernel exits
            }
        }

        // ---- epilogue: partial slab [split][V,H]; bias partial [split][V] (k_dw_x2's)
        float *sw = a.slab_w + (long)split * V * H;
        const float rw = a.dw_rescale, rb = a.db_rescale;
#pragma unroll
        for (int qm = 0; qm < 4; ++qm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = v0 + 32 * qm + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
                for (int qn = 0; qn < 4; ++qn) sw[(long)v * H + h0 + 32 * qn + (lane & 31)] = acc[qm][qn][r] * rw;  // (V % 512 == 0, H % 128 == 0: every column exists)
            }
        if (do_b && (lane & 31) == 0) {  // column 0 of the selector products: lanes 0 / 32 store
#pragma unroll
            for (int r = 0; r < 16; ++r) a.slab_b[(long)split * V + v0 + 32 * bsel0 + (r & 3) + 8 * (r >> 2) + 4 * half] = dacc[r] * rb;
        }
    };
    if (tile >= n_full) body(X2Int<1>{});  // (workgroup-uniform)
    else body(X2Int<0>{});
}

#ifdef RNNT_LAB
#include "lab/x2_lab_dw.inc"  // k_dw_x2p (RNNT_VARIANT_X2_DW_P16): measured equal to k_dw_x2<4>, kept as lab equipment
#endif

void launch_dw_x2(const X3Args &a, hipStream_t st, bool build_table, bool zero_prog)
{
    if (build_table) launch_dw_table(a.logit_lens, a.B, a.T, a.U1, XW2_GRAN, a.dw_tab, st);
    if (a.dw_prog && zero_prog) launch_fill32(a.dw_prog, 0u, (size_t)a.n_split * 64, st);
    const int tiles = x2_dw_tiles(a.H, a.V);
    static bool attr_set[16] = {false};  // > 64 KiB of dynamic LDS: opt-in once per device (read-mostly fact)
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    if (dev < 0 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)k_dw_x2<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * XW2_TILE);
#ifdef RNNT_LAB
        (void)hipFuncSetAttribute((const void *)k_dw_x2<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * XW2_TILE);
        (void)hipFuncSetAttribute((const void *)k_dw_x2p, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * XW2_TILE);
#endif
        (void)hipFuncSetAttribute((const void *)k_dw_x2<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * XW2_TILE);
        (void)hipFuncSetAttribute((const void *)k_dw_x2m, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * XW2_TILE);
        if (dev >= 0) attr_set[dev] = true;
    }
#ifdef RNNT_LAB  // (the product build refuses these variants at the C boundary: engine.hip)
    if (a.flags & RNNT_VARIANT_X2_DW_P16) { hipLaunchKernelGGL(k_dw_x2p, dim3(tiles * a.n_split), dim3(256), 4 * XW2_TILE, st, a); return; }
    if (a.flags & RNNT_VARIANT_X2_DW_8W) { hipLaunchKernelGGL(k_dw_x2<8>, dim3(tiles * a.n_split), dim3(512), 4 * XW2_TILE, st, a); return; }
#endif
    // H % 256 == 128: whole-block tiles + tall tiles over the odd block (k_dw_x2m) where the shape allows; else the half-empty last h block
    if (x2_dw_mixed_ok(a.H, a.V)) hipLaunchKernelGGL(k_dw_x2m, dim3(tiles * a.n_split), dim3(256), 5 * XW2_TILE, st, a);
    else if (a.H % 256 != 0) hipLaunchKernelGGL((k_dw_x2<4, true>), dim3(((a.V + 255) / 256) * ((a.H + 255) / 256) * a.n_split), dim3(256), 4 * XW2_TILE, st, a);
    else hipLaunchKernelGGL(k_dw_x2<4>, dim3(tiles * a.n_split), dim3(256), 4 * XW2_TILE, st, a);
}

#define XG2_WAIT8(b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) :: "memory")
#define XG2_WAIT8_BUT(b, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) :: "memory")

// ---------------------------------------------------------------------------------------
// W (scaled by s_W) for the dHidden product, fragment order, two planes:
//   [hp (512-column pass)][c (16-deep k-step = 16 vocabulary rows)][plane][tile(16)][lane] x 8 fp16,
//   element j = piece_plane(s_W W[v = 16c + 8*(lane>>5) + j][h = 512hp + 128*(tile>>2) + 4*(lane&31) + (tile&3)])
// (columns interleaved by 4: a lane's 4 tiles of a 128-column group are 4 adjacent columns -> 16-byte epilogue accesses).
// One k-step = 2 x 16 KiB, staged by one linear LDS-DMA copy.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_x2_pack_w_dh(const float *__restrict__ W, const float *__restrict__ scales, u32x4 *__restrict__ out, int H, int V, long n)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (hp, c, tile, lane): one thread writes both planes
    if (idx >= n) return;
    const int lane = (int)(idx & 63), tile = (int)(idx >> 6) & 15;
    const int VC = V / 16;
    const int c = (int)((idx >> 10) % VC), hp = (int)((idx >> 10) / VC);
    const int h = 512 * hp + 128 * (tile >> 2) + 4 * (lane & 31) + (tile & 3);
    const int v0 = 16 * c + 8 * (lane >> 5);
    const float sw = scales[0];
    u32x4 ph = {0u, 0u, 0u, 0u}, pm = ph;
    if (h < H) {
        const float *w = W + (long)v0 * H + h;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const X2Pieces q = x2_split2(x2_clamp(w[(long)(2 * j) * H] * sw), x2_clamp(w[(long)(2 * j + 1) * H] * sw));
            ph[j] = q.h; pm[j] = q.m;
        }
    }
    u32x4 *o = out + ((long)hp * VC + c) * 2048 + tile * 64 + lane;
    o[0] = ph; o[1024] = pm;
}
size_t x2_wpack_dh_bytes(int H, int V) { return (size_t)((H + 511) / 512) * (V / 16) * 2 * 16 * 64 * 16; }

// ---------------------------------------------------------------------------------------
// k_dhidden_x2: k_dhidden_x3 on two planes.  G from the fp32 logits (exp2 + the two occupancy corrections, as the fp32 route),
// scaled by g_scale, split into its two fp16 planes, stored in place (hi | mid over each 32-wide chunk of the logits row: the
// planes fill exactly the bytes of the logits they replace) and multiplied: dHidden = G . W over 512 columns of H per launch;
// epilogue as the fp32 kernel: x (1 - hidden^2), sum over u -> dEnc slab, sum over t -> dPred slab.  Tile = 8 t x 16 u cells.
// 4 waves = 2 (M) x 2 (N), wave tile 64 cells x 256 columns = 16 accumulator tiles (256 registers).
//  * production: wave w turns M tile w (32 cells) into G: lane (cell i = l&31, half) owns the 8 vocabulary entries 16c +
//    8*half .. +7 of its cell per k-step — its slot of the MFMA A fragment — and drops 16 bytes per plane into the LDS exchange
//    [2 slots][M tile][plane][lane]; logits requested 4 k-steps ahead (register ring);
//  * G leaves in WHOLE 128-byte lines (32 x hi | 32 x mid of one cell and chunk = the fragments of two k-steps x two halves x
//    two planes), read back by the producing wave from its part of the exchange every second k-step: 4 store instructions;
//  * W: one linear LDS-DMA copy of 32 KiB per k-step into a 3-slot ring, two k-steps ahead (8 DMAs per wave);
//  * per k-step ONE barrier publishes W slot c and exchange slot c; then 3 products x 16 MFMAs:
//      block 0  ah.bh   + the 8 fragment reads of W's mid plane + production slices 0-7 of G(c+1) (exp2, corrections, split)
//      block 1  am.bh   + exchange write of G(c+1), W DMAs 0-6 of k-step c+2, (even c) the pair's line reads
//      block 2  ah.bm   + W DMA 7, (even c) the 4 line stores, the 2 raw logits loads of k-step c+5
//    (memory operations in ONE fixed order per k-step — DMAs, then stores, then loads — so that every vmcnt is a count).
// FIRST = false (H > 512: columns 512.., one launch per further 512): G's planes are read back from memory into the exchange
// instead of being produced; nothing is stored but the slabs.
// The accumulators hold g_scale s_W dHidden: the epilogue's sums are multiplied by 1 / (g_scale s_W) (a power of two).
// grid (n_ublk, ceil(T/8), B).  Requires V % 128 == 0, H % 128 == 0.
// ---------------------------------------------------------------------------------------
#define XG2_BT 8
#define XG2_BU 16
#define XG2_WSLOT 32768   // one k-step of W: 2 planes x 16 tiles x 1 KiB
#define XG2_XSLOT 8192    // one k-step of G fragments: 4 M tiles x 2 planes x 1 KiB
#define XG2_NW 3          // W ring slots: the DMAs of k-step c+2 are issued during k-step c (as in k_joint_fwd_x2)
// -DXG2_EXP=bits (what-if builds of k_dhidden_x2, WRONG results; tools/build_x2_variants.sh with X2_FLAGS=-DXG2_EXP=..): 1 no MFMAs, 2 W's DMAs
// requested past the pack's range (instructions stay, no bytes move), 4 no production arithmetic (the raw bits go to the exchange), 8 the line
// stores aimed past the buffer's range (dropped), 16 the raw logits loads aimed at the zero padding row (cache-resident), 32 no fragment reads,
// 64 no barrier
#ifndef XG2_EXP
#define XG2_EXP 0
#endif
// PART: the pass covers fewer than 512 columns (H = 640: the second pass has 128; H < 512): the dead 128-column groups run no MFMAs
// and their W pieces are requested past the pack's range (an out-of-range LDS-DMA moves no bytes and writes zeros: the instruction
// stays, so every vmcnt stays a count) — a pass then costs what its live columns cost plus the G stream.
template <bool FIRST, bool PART>
__global__ __launch_bounds__(256, 1) void k_dhidden_x2(X3Args a, const int hp)
{
    // [0, 96 KiB): W ring, 3 slots;  [96, 112 KiB): G exchange, 2 slots.  The epilogue reuses the W ring.
    extern __shared__ __attribute__((aligned(1024))) char s_dh[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int ub = blockIdx.x, tt = blockIdx.y, b = blockIdx.z;
    int Tb, Ub;
    len_tu_uniform(a.logit_lens, a.target_lens, b, a.T, a.U1, Tb, Ub);
    const int t0 = tt * XG2_BT, u0 = ub * XG2_BU;
    const int VC = V / 16;
    const int ngrp = PART ? (H - 512 * hp) / 128 : 4;  // live 128-column groups of this pass (H % 128 == 0)
    if (FIRST) X2_CLOCK_STAMP(128 + 108);  // (one tile of ~0.1 ms: 1e-4 resolution at the 100 MHz reference)

    // ---- producer role: M tile `wave`, row i = cell (pt, pu)
    const int prow = wave * 32 + i;
    const int pt = t0 + (prow >> 4), pu = u0 + (prow & 15);
    const bool pexists = pt < T && pu < U1;
    const long zrow = (long)a.B * T * U1;  // first zero padding row
    const long pcell = pexists ? ((long)b * T + pt) * U1 + pu : zrow;

    // workgroup-uniform: no products past the utterance's length or in a u block past U_b (no lattice cell; the reductions
    // skip its slabs), but k_dw_x2 must find zeros in these rows (both planes = the whole logits row)
    if (t0 >= Tb || u0 > Ub) {
        if (FIRST && pexists) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            u32x4 *g = (u32x4 *)(a.logits + pcell * V) + half;
            for (int c = 0; c < VC; ++c) {  // this lane's 16 B of each plane per k-step (layout below)
                g[8 * (c >> 1) + 2 * (c & 1)] = z;
                g[8 * (c >> 1) + 4 + 2 * (c & 1)] = z;
            }
        }
        return;
    }

    CellCoef cf = a.coef[pexists ? pcell : 0];
    const bool live = pexists && pt < Tb && cf.c1 != RNNT_NEG_INF;
    if (!live) { cf.c1 = RNNT_NEG_INF; cf.sb = 0.f; cf.se = 0.f; cf.y = -1; }
    const float gs = a.g_scale;
    // The producer's memory, per k-step c (16 vocabulary entries, this lane: 8 of them, v = 16c + 8half + j):
    //   logits (fp32): 32 bytes at row + 64c + 32half               (f32x4 index 4c + 2half, +1)
    //   G hi: 16 bytes at row + 128(c>>1) + 32(c&1) + 16half        (u32x4 index 8(c>>1) + 2(c&1) + half);  G mid: + 64 bytes
    // rows outside the lattice read the zero padding row (finite) with c1 = -inf -> G = 0; FIRST = false: every existing row
    // holds its G planes already
    const float *xsrc = a.logits + ((XG2_EXP & 16) ? zrow : ((FIRST ? live : pexists) ? pcell : zrow)) * V;
    const int blank = a.blank;
    // whole-line stores (k_dhidden_x3, round 4): lane L -> row 8n + (L >> 3) of the M tile (n = 0..3: four store instructions,
    // 8 whole lines each), piece L & 7 = (plane, k-step parity, half), read back from this wave's part of the exchange (both
    // slots hold the pair between the exchange write of the odd k-step and the next even one's).  Raw-buffer stores over the
    // tile's rows: rows outside the lattice get an offset past the range (dropped): every wave issues the same instructions.
    const int lds0 = (int)(size_t)(lds_vptr)s_dh;
    const int lpiece = lane & 7, lrow = lane >> 3;
    // LDS: [slot = parity][M tile wave][plane][lane slot = row + 32 half]
    const int xl = lds0 + XG2_NW * XG2_WSLOT + ((lpiece >> 1) & 1) * XG2_XSLOT + wave * 2048 + (lpiece >> 2) * 1024 + 16 * (lrow + 32 * (lpiece & 1));
    const long tile_cell0 = ((long)b * T + t0) * U1 + u0;  // the tile's first cell (exists: t0 < Tb <= T, u0 <= Ub < U1)
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.logits + tile_cell0 * V), 0, (int)((((long)(XG2_BT - 1) * U1 + XG2_BU) * V) * 4), 0x00020000);
    int lvo[4];  // byte offset of this lane's piece of line n in the buffer: row (t0 + 2 wave + (n >> 1), u0 + 8 (n & 1) + lrow)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int lt = 2 * wave + (n >> 1), lu = 8 * (n & 1) + lrow;
        const bool ex = t0 + lt < T && u0 + lu < U1;
        lvo[n] = (ex && !(XG2_EXP & 8)) ? (int)(((long)lt * U1 + lu) * V * 4) + 64 * (lpiece >> 2) + 16 * (lpiece & 3) : 0x7ffffff0;
    }

    f32x16 acc[2][8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;

    const int xw = lds0 + XG2_NW * XG2_WSLOT + wave * 2048 + 16 * lane;       // exchange write: [slot][M tile wave][plane][lane]
    const int xa = lds0 + XG2_NW * XG2_WSLOT + (2 * wm) * 2048 + 16 * lane;   // exchange read: M tiles 2wm, 2wm+1
    const int wb = lds0 + (8 * wn) * 1024 + 16 * lane;                   // W read: tiles 8wn .. 8wn+7 of each plane
    // W DMA: wave w copies pieces 8w .. 8w+7 of the k-step's 32 (piece = 1 KiB = one (plane, tile)); raw-buffer form: scalar
    // base (this launch's 512-column pass) and offsets, one constant per-lane offset
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((char *)a.wpack_dh + (long)hp * VC * XG2_WSLOT, 0, VC * XG2_WSLOT, 0x00020000);
    const int wvo = lane * 16;

    struct Raw { f32x4 x0, x1; };  // FIRST: 8 fp32 logits; else: hi | mid planes (as x0, x1 bits)
    auto xload = [&](Raw &r, int c, int part = 3) {  // part: 1 first half, 2 second half, 3 both
        const int cc = c < VC ? c : VC - 1;
        if (FIRST) {
            const f32x4 *p = (const f32x4 *)xsrc + 4 * cc + 2 * half;
            if (X2_NT & 1) {  // experiment: the logits are read once, by this CU only -> non-temporal loads
                if (part & 1) r.x0 = __builtin_nontemporal_load(p);
                if (part & 2) r.x1 = __builtin_nontemporal_load(p + 1);
            } else {
                if (part & 1) r.x0 = p[0];
                if (part & 2) r.x1 = p[1];
            }
        } else {
            const u32x4 *p = (const u32x4 *)xsrc + 8 * (cc >> 1) + 2 * (cc & 1) + half;
            if (part & 1) r.x0 = __builtin_bit_cast(f32x4, p[0]);
            if (part & 2) r.x1 = __builtin_bit_cast(f32x4, p[4]);
        }
    };
    // G of k-step c from the raw values -> exchange slot (c & 1).  The work is cut into slices (0..8) that the main loop
    // threads through the gaps of its first MFMA blocks.
    struct Prod { f32x4 g0, g1; u32x4 ph, pm; };
    auto produce_slice = [&](Prod &P, const Raw &r, int c, int sl) {
        if (!FIRST || (XG2_EXP & 4)) {
            if (sl == 0) { P.ph = __builtin_bit_cast(u32x4, r.x0); P.pm = __builtin_bit_cast(u32x4, r.x1); }
        } else if (sl < 4) {
            P.g0[sl] = __builtin_amdgcn_exp2f(fmaf(r.x0[sl], RNNT_LOG2E, cf.c1));
            P.g1[sl] = __builtin_amdgcn_exp2f(fmaf(r.x1[sl], RNNT_LOG2E, cf.c1));
        } else if (sl == 4) {
            const int vb = 16 * c + 8 * half;
            const unsigned dy = (unsigned)(cf.y - vb);
            if (__any(dy < 8u)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    P.g0[e] = (dy == (unsigned)e) ? P.g0[e] - cf.se : P.g0[e];
                    P.g1[e] = (dy == (unsigned)(e + 4)) ? P.g1[e] - cf.se : P.g1[e];
                }
            }
        } else if (sl == 5) {
            const int vb = 16 * c + 8 * half;
            if ((unsigned)(blank - 16 * c) < 16u) {  // wave-uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    P.g0[e] = (vb + e == blank) ? P.g0[e] - cf.sb : P.g0[e];
                    P.g1[e] = (vb + e + 4 == blank) ? P.g1[e] - cf.sb : P.g1[e];
                }
            }
        } else if (sl == 6) {
            const f32x4 s4 = P.g0 * gs;  // (|G| <= grad_scale: no clamp needed; exact: a power of two)
            X2_SPLIT4(s4, P.ph, P.pm, 0);
        } else if (sl == 7) {
            const f32x4 s4 = P.g1 * gs;
            X2_SPLIT4(s4, P.ph, P.pm, 2);
        }
        if (sl == 8) {
            const int dst = xw + (c & 1) * XG2_XSLOT;
            asm volatile("ds_write_b128 %0, %1" :: "v"(dst), "v"(P.ph) : "memory");
            asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(dst), "v"(P.pm) : "memory");
        }
    };
    auto produce = [&](const Raw &r, int c) {  // all slices at once (pipeline prologue)
        Prod P;
#pragma unroll
        for (int sl = 0; sl < 9; ++sl) produce_slice(P, r, c, sl);
    };
    auto wdma = [&](int c, int slot, int n) {  // piece n (0..7) of this wave's share of W k-step c -> ring slot `slot`
        const int cc = c < VC ? c : VC - 1;
        const int vo = (!(XG2_EXP & 2) && (!PART || (((wave * 8 + n) & 15) >> 2) < ngrp)) ? wvo : 0x7ffffff0;  // (piece = plane (pc >> 4), tile pc & 15 = group (pc & 15) >> 2)
#if XF2_IMM  // pieces 4g .. 4g+3 on one LDS base (M0) and one scalar offset, told apart by the immediate offset (as the forward's)
        const int g4 = n & 4;
        lds_vptr dst = (lds_vptr)(s_dh + slot * XG2_WSLOT + (wave * 8 + g4) * 1024);
        const int so = (cc * 32 + wave * 8 + g4) * 1024;
        if ((n & 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, vo, so, 0, 0);
        if ((n & 3) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, vo, so, 1024, 0);
        if ((n & 3) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, vo, so, 2048, 0);
        if ((n & 3) == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, vo, so, 3072, 0);
#else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_vptr)(s_dh + slot * XG2_WSLOT + (wave * 8 + n) * 1024), 16, vo,
                                                 (cc * 32 + wave * 8 + n) * 1024, 0, 0);
#endif
    };

    Raw xr[4];  // raw ring (slot = k-step & 3), 4 k-steps ahead of production
    xload(xr[0], 0); xload(xr[1], 1); xload(xr[2], 2); xload(xr[3], 3);
#pragma unroll
    for (int n = 0; n < 8; ++n) wdma(0, 0, n);
#pragma unroll
    for (int n = 0; n < 8; ++n) wdma(1, 1, n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the in-place stores of the first pair overwrite logits of k-steps 0, 1
    produce(xr[0], 0);
    xload(xr[0], 4);

    int wsl = 0;  // W ring slot of k-step c (c % 3)
    for (int c0 = 0; c0 < VC; c0 += 4) {  // VC % 4 == 0 (V % 128 == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            // W k-step c landed (this wave's share; requested during k-step c-2), G fragments of step c written: publish both; every
            // wave is past its reads of step c-1 (W slot of c+2, exchange slot of c+1).  vmcnt retires in order: behind the DMAs of
            // k-step c-2 come its (even) 4 line stores and 2 raw loads, then k-step c-1's 8 DMAs, (even) 4 line stores and 2 raw
            // loads: 16 operations whatever c's parity.  In-place safety: a pair's line overwrites the logits of its own two
            // k-steps, both loaded and consumed (production) before the store is issued.
            if (FIRST) asm volatile(RNNT_VMCNT(16) ::: "memory");
            else asm volatile(RNNT_VMCNT(12) ::: "memory");
            if (!(XG2_EXP & 64)) x2_lds_barrier();
            const int ws = wb + wsl * XG2_WSLOT, xs = xa + (j & 1) * XG2_XSLOT;
            const int wsn = wsl == 0 ? 2 : wsl - 1;  // (c + 2) % 3
            u32x4 af[2][2], bf[8], bn[8];
            // fragment reads: A (4) and the hi plane of W (8)
            if (!(XG2_EXP & 32)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[mt][p]) : "v"(xs), "n"(mt * 2048 + p * 1024));
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[q]) : "v"(ws), "n"(q * 1024));
            }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]),
                           "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]), "+v"(bf[4]), "+v"(bf[5]), "+v"(bf[6]), "+v"(bf[7])
                         :: "memory");
            Prod P;
            // (round 5: G(c+1) is produced UNCONDITIONALLY — at the last k-step that is a k-step past the end, made from the clamped raw loads
            // and written into an exchange slot nobody reads any more: fourteen uniform branches per k-step cut the MFMA stream into as
            // many scheduling regions before)
            constexpr bool prod_on = true;
            const Raw &rawn = xr[(j + 1) & 3];
            u32x4 ln[4];  // the pair's lines, 16 bytes per lane each (even k-steps)
            auto line_read = [&](u32x4 &v, auto n_c) {  // rows 8n .. 8n+7 of the M tile: lane slot + 8n
                const int xl_ = xl;  // (a local: asm operands cannot name a capture of the enclosing generic lambda)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(xl_), "n"(128 * decltype(n_c)::value));
            };
            auto line_store = [&](u32x4 &v, int n, int ce) {  // line n of the pair (ce, ce + 1): chunk ce >> 1 of the rows
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) :: "memory");
                __builtin_amdgcn_raw_buffer_store_b128(v, grs, lvo[n], 128 * (ce >> 1), 0);
            };
            // one product block: 16 MFMAs = (2 M tiles) x (8 column tiles) for A plane PA against the B plane held in `bcur`;
            // BLK names the work threaded through it (the kernel comment's table)
            auto block = [&](auto pa_c, const u32x4 (&bcur)[8], auto blk_c) {
                constexpr int PA = decltype(pa_c)::value, BLK = decltype(blk_c)::value;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (!(XG2_EXP & 1) && (!PART || 2 * wn + (q >> 2) < ngrp)) {  // (wave-uniform)
                        acc[0][q] = x2_mfma(af[0][PA], bcur[q], acc[0][q]);
                        acc[1][q] = x2_mfma(af[1][PA], bcur[q], acc[1][q]);
                    }
                    if (BLK == 0) {
                        if (!(XG2_EXP & 32)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bn[q]) : "v"(ws), "n"(16384 + q * 1024));
                        if (prod_on) produce_slice(P, rawn, c + 1, q);
                    }
                    if (BLK == 1) {
                        if (q == 0 && prod_on) produce_slice(P, rawn, c + 1, 8);
                        if (q >= 1) wdma(c + 2, wsn, q - 1);
                        if (FIRST && q == 2 && !(j & 1) && prod_on) { line_read(ln[0], X2Int<0>{}); line_read(ln[1], X2Int<1>{}); }
                        if (FIRST && q == 4 && !(j & 1) && prod_on) { line_read(ln[2], X2Int<2>{}); line_read(ln[3], X2Int<3>{}); }
                    }
                    if (BLK == 2) {
                        if (q == 0) wdma(c + 2, wsn, 7);
                        if (FIRST && q >= 2 && q <= 5 && !(j & 1)) {
                            if (prod_on) line_store(ln[q - 2 < 0 ? 0 : (q - 2 > 3 ? 3 : q - 2)], q - 2, c);
                            else { u32x4 z = {0u, 0u, 0u, 0u}; __builtin_amdgcn_raw_buffer_store_b128(z, grs, 0x7ffffff0, 0, 0); }  // (never: VC is even; keeps the count)
                        }
                        if (q == 6) xload(xr[(j + 1) & 3], c + 5, 1);
                        if (q == 7) xload(xr[(j + 1) & 3], c + 5, 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            block(X2Int<0>{}, bf, X2Int<0>{});   // ah.bh
            block(X2Int<1>{}, bf, X2Int<1>{});   // am.bh
            XG2_WAIT8(bn);
            block(X2Int<0>{}, bn, X2Int<2>{});   // ah.bm
            wsl = wsl == 2 ? 0 : wsl + 1;
        }
    }
    // the epilogue's pred rows are requested BEFORE the drain below (they ride out the G stores' acknowledgements with it)
    const int colg[2] = {512 * hp + 256 * wn + 4 * i, 512 * hp + 256 * wn + 128 + 4 * i};
    const bool colok[2] = {colg[0] < H, colg[1] < H};
    f32x4 pr[8][2];  // pred rows of this lane's 8 u slots, its 2 x 4 columns (zero where u >= U1 or the column >= H)
#pragma unroll
    for (int r7 = 0; r7 < 8; ++r7) {
        const int u = u0 + 8 * (r7 >> 2) + (r7 & 3) + 4 * half;
#pragma unroll
        for (int g = 0; g < 2; ++g)
            pr[r7][g] = (u < U1 && colok[g]) ? *(const f32x4 *)(a.pred + ((long)b * U1 + u) * H + colg[g]) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float unscale = 4.0f * a.db_rescale * a.scales[1];  // 4 (the tanh' form below) / (g_scale s_W)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-issued ring loads / DMAs
    __syncthreads();

    // ---- epilogue (k_dhidden_x3's).  Accumulator register rr = 8rh + r7 of M tile (2wm + mt), column tile q: row (rr&3) +
    // 8(rr>>2) + 4half of its 32 = t row 2(2wm+mt) + rh, u slot 8(r7>>2) + (r7&3) + 4half; column 512hp + 256wn + 128(q>>2) +
    // 4i + (q&3).  The tanh' factor 1 - hidden^2 is recomputed from enc and pred: 4 w / (1 + w)^2, w = exp(-2|x|).
    float (*s_red)[64][65] = (float (*)[64][65])s_dh;  // [wn][lane][8 u slots x 8 columns]
    const long BTH = (long)a.B * T * H, BUH = (long)a.B * U1 * H;
    auto dfac_q = [](f2 x) {  // (1 - tanh^2 x) / 4 of two values
        const f2 a = x * (2.0f * RNNT_LOG2E);
        const f2 w = {__builtin_amdgcn_exp2f(-__builtin_fabsf(a[0])), __builtin_amdgcn_exp2f(-__builtin_fabsf(a[1]))};
        const f2 e1 = w + 1.0f;
        const f2 r = {__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1])};
        return (w * r) * r;
    };
    f2 psum2[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) psum2[k][q] = f2{0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            const int tl = 2 * (2 * wm + mt) + rh;  // t row inside the tile
            const int t = t0 + tl;
            f2 esum2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) esum2[q] = f2{0.f, 0.f};
            f32x4 er[2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
                er[g] = (t < T && colok[g]) ? *(const f32x4 *)(a.enc + (long)b * a.enc_sb + (long)t * a.enc_st + colg[g]) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r7 = 0; r7 < 8; ++r7)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {  // columns 2qq, 2qq+1 of the group
                        const f2 x = {er[g][2 * qq] + pr[r7][g][2 * qq], er[g][2 * qq + 1] + pr[r7][g][2 * qq + 1]};
                        const f2 av = {acc[mt][g * 4 + 2 * qq][rh * 8 + r7], acc[mt][g * 4 + 2 * qq + 1][rh * 8 + r7]};
                        const f2 d = av * dfac_q(x);
                        esum2[g * 2 + qq] += d;
                        psum2[r7][g * 2 + qq] += d;
                    }
            float esum[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) esum[q] = unscale * esum2[q >> 1][q & 1];
#pragma unroll
            for (int q = 0; q < 8; ++q) esum[q] += __shfl_xor(esum[q], 32, 64);
            if (half == 0 && t < Tb) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (colok[g]) {
                        const f32x4 o = {esum[g * 4], esum[g * 4 + 1], esum[g * 4 + 2], esum[g * 4 + 3]};
                        *(f32x4 *)(a.slab_enc + (long)ub * BTH + ((long)b * T + t) * H + colg[g]) = o;
                    }
            }
        }
    float psum[8][8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) psum[k][q] = unscale * psum2[k][q >> 1][q & 1];
    if (wm == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) s_red[wn][lane][k * 8 + q] = psum[k][q];
    }
    __syncthreads();
    if (wm == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = u0 + 8 * (k >> 2) + (k & 3) + 4 * half;
            if (u < U1) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (colok[g]) {
                        f32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = psum[k][g * 4 + q] + s_red[wn][lane][k * 8 + g * 4 + q];
                        *(f32x4 *)(a.slab_pred + (long)tt * BUH + ((long)b * U1 + u) * H + colg[g]) = o;
                    }
            }
        }
    }
    if (FIRST) X2_CLOCK_STAMP(128 + 110);
}

// ---------------------------------------------------------------------------------------
// k_dhidden_x2r (round 5): the LAST 128 columns of dHidden when H % 512 == 128 (config 4: H = 640).  k_dhidden_x2<false, true> ran them as
// a second full-cost pass — G re-read into the exchange, barrier, 12 fragment reads and a 32-piece W stage per k-step for a quarter of
// the MFMAs: ~2 600 cycles per k-step whatever rides in it (round 4).  Here a wave owns a whole 8 t x 16 u block of cells (128 rows = four
// M tiles) x the 128 columns (four N tiles): 16 accumulator tiles, 48 MFMAs per k-step like every other kernel of the route, and
//  * its A fragments come STRAIGHT from G's planes in memory — lane (i, half) of M tile m loads the 16 bytes of hi and of mid of row i,
//    k = 8 half .. +7, which IS the MFMA fragment: no exchange, no producer role — into a 4-deep register ring, three k-steps ahead;
//  * the 8 KiB of W per k-step (2 planes x 4 tiles of the pass's pack) go through a 4-slot LDS ring, 2 DMA pieces per wave, shared by the
//    workgroup's four waves = four consecutive t blocks of one u block; one barrier per k-step publishes a slot;
//  * the epilogue is k_dhidden_x2's for one wave: x (1 - hidden^2) recomputed from enc and pred, sum over u -> dEnc slab, sum over the
//    block's 8 t -> dPred slab (no cross-wave reduction: the block's t rows all sit in this wave).
// ---------------------------------------------------------------------------------------
#define XR2_WSLOT 8192
#define XR2_NW 4
__global__ __launch_bounds__(256, 1) void k_dhidden_x2r(X3Args a, const int hp)
{
    extern __shared__ __attribute__((aligned(1024))) char s_dr[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int ub = blockIdx.x, tq = blockIdx.y, b = blockIdx.z;
    int Tb, Ub;
    len_tu_uniform(a.logit_lens, a.target_lens, b, a.T, a.U1, Tb, Ub);
    const int u0 = ub * XG2_BU;
    const int tt = 4 * tq + wave, t0 = tt * XG2_BT;  // this wave's t block
    const int VC = V / 16;
    if (4 * tq * XG2_BT >= Tb || u0 > Ub) return;    // workgroup-uniform: no lattice cell in any of the four blocks
    const bool wlive = t0 < Tb;                        // wave-uniform: this wave's block has lattice cells (else: zeros in, nothing out)
    const long zrow = (long)a.B * T * U1;              // first zero padding row
    // row i of M tile m = cell (t0 + 2m + (i >> 4), u0 + (i & 15)); rows outside the lattice read the zero padding row.  G's planes of
    // k-step c: hi = 16 bytes at u32x4 index 8 (c >> 1) + 2 (c & 1) + half of the row, mid 64 bytes behind (k_dhidden_x2's layout)
    const u32x4 *arow[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int t = t0 + 2 * m + (i >> 4), u = u0 + (i & 15);
        const bool ex = wlive && t < T && u < U1;
        arow[m] = (const u32x4 *)(a.logits + (ex ? ((long)b * T + t) * U1 + u : zrow) * V) + half;
    }
    struct AFrag { u32x4 h[4], m[4]; };
    auto aload = [&](AFrag &f, int c) {
        const int cc = c < VC ? c : VC - 1;
        const int o = 8 * (cc >> 1) + 2 * (cc & 1);
#pragma unroll
        for (int m = 0; m < 4; ++m) { f.h[m] = arow[m][o]; f.m[m] = arow[m][o + 4]; }
    };
    // W: piece (plane p, tile tl) of k-step c at ((c 2 + p) 16 + tl) KiB of the pass's pack; wave w stages plane w >> 1, tiles 2 (w & 1), +1
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((char *)a.wpack_dh + (long)hp * VC * XG2_WSLOT, 0, VC * XG2_WSLOT, 0x00020000);
    const int wvo = lane * 16;
    const int wp = wave >> 1, wt2 = 2 * (wave & 1);
    auto wdma = [&](int c, int slot) {  // both pieces on one M0 / scalar offset (the immediate offset advances memory and LDS address)
        const int cc = c < VC ? c : VC - 1;
        lds_vptr dst = (lds_vptr)(s_dr + slot * XR2_WSLOT + (wp * 4 + wt2) * 1024);
        const int so = ((cc * 2 + wp) * 16 + wt2) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, wvo, so, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, dst, 16, wvo, so, 1024, 0);
    };
    const int lds0 = (int)(size_t)(lds_vptr)s_dr;
    const int wb = lds0 + 16 * lane;  // W read: tile n of plane p of slot s at wb + s * XR2_WSLOT + (p * 4 + n) * 1024

    f32x16 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    AFrag ar[4];  // ring: k-step c in ar[c & 3], loaded three k-steps ahead
    aload(ar[0], 0); aload(ar[1], 1); aload(ar[2], 2);
    wdma(0, 0); wdma(1, 1);
    int wsl = 0;  // W ring slot of k-step c (c & 3)
    for (int c0 = 0; c0 < VC; c0 += 4) {  // VC % 4 == 0 (V % 128 == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            // this wave's pieces of W(c) landed: issued at the end of k-step c-2; younger: k-step c-1's 8 A loads and 2 DMAs
            asm volatile(RNNT_VMCNT(10) ::: "memory");
            x2_lds_barrier();  // publishes W(c); every wave is past its reads of W(c-1): slot (c + 3) & 3 ... (c + 2) & 3 are free
            const int ws = wb + wsl * XR2_WSLOT;
            u32x4 bh[4], bm[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bh[n]) : "v"(ws), "n"(n * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]) :: "memory");
            AFrag &ac = ar[j], &an = ar[(j + 3) & 3];
            const int cn = c + 3;
            const int o = 8 * ((cn < VC ? cn : VC - 1) >> 1) + 2 * ((cn < VC ? cn : VC - 1) & 1);
            // block 0: ah.bh + the mid plane's reads + the A loads of k-step c+3 (one per two MFMAs)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = x2_mfma(ac.h[m], bh[n], acc[m][n]);
                    if (m == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bm[n]) : "v"(ws), "n"(4096 + n * 1024));
                    if (n == 1) an.h[m] = arow[m][o];
                    if (n == 3) an.m[m] = arow[m][o + 4];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // block 1: am.bh
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = x2_mfma(ac.m[m], bh[n], acc[m][n]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bm[0]), "+v"(bm[1]), "+v"(bm[2]), "+v"(bm[3]) :: "memory");
            // block 2: ah.bm + the 2 DMA pieces of W(c+2) (the youngest memory operations of the k-step)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[m][n] = x2_mfma(ac.h[m], bm[n], acc[m][n]);
                    if (m == 3 && n == 0) wdma(c + 2, (wsl + 2) & 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            wsl = (wsl + 1) & 3;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-issued loads / DMAs: none may land in an LDS the next workgroup owns
    if (!wlive) return;

    // ---- epilogue (k_dhidden_x2's, one wave = one block).  Accumulator register rr = 8 rh + r7 of M tile m, column tile n: t row
    // 2m + rh, u slot 8 (r7 >> 2) + (r7 & 3) + 4 half; column 512 hp + 4i + n.
    const int colg = 512 * hp + 4 * i;
    const bool colok = colg < H;
    f32x4 pr[8];
#pragma unroll
    for (int r7 = 0; r7 < 8; ++r7) {
        const int u = u0 + 8 * (r7 >> 2) + (r7 & 3) + 4 * half;
        pr[r7] = (u < U1 && colok) ? *(const f32x4 *)(a.pred + ((long)b * U1 + u) * H + colg) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float unscale = 4.0f * a.db_rescale * a.scales[1];  // 4 (the tanh' form below) / (g_scale s_W)
    const long BTH = (long)a.B * T * H, BUH = (long)a.B * U1 * H;
    auto dfac_q = [](f2 x) {  // (1 - tanh^2 x) / 4 of two values
        const f2 av = x * (2.0f * RNNT_LOG2E);
        const f2 w = {__builtin_amdgcn_exp2f(-__builtin_fabsf(av[0])), __builtin_amdgcn_exp2f(-__builtin_fabsf(av[1]))};
        const f2 e1 = w + 1.0f;
        const f2 r = {__builtin_amdgcn_rcpf(e1[0]), __builtin_amdgcn_rcpf(e1[1])};
        return (w * r) * r;
    };
    f2 psum2[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k) { psum2[k][0] = f2{0.f, 0.f}; psum2[k][1] = f2{0.f, 0.f}; }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            const int t = t0 + 2 * m + rh;
            const f32x4 er = (t < T && colok) ? *(const f32x4 *)(a.enc + (long)b * a.enc_sb + (long)t * a.enc_st + colg) : f32x4{0.f, 0.f, 0.f, 0.f};
            f2 esum2[2] = {f2{0.f, 0.f}, f2{0.f, 0.f}};
#pragma unroll
            for (int r7 = 0; r7 < 8; ++r7)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const f2 x = {er[2 * qq] + pr[r7][2 * qq], er[2 * qq + 1] + pr[r7][2 * qq + 1]};
                    const f2 av = {acc[m][2 * qq][rh * 8 + r7], acc[m][2 * qq + 1][rh * 8 + r7]};
                    const f2 d = av * dfac_q(x);
                    esum2[qq] += d;
                    psum2[r7][qq] += d;
                }
            f32x4 es = {unscale * esum2[0][0], unscale * esum2[0][1], unscale * esum2[1][0], unscale * esum2[1][1]};
#pragma unroll
            for (int q = 0; q < 4; ++q) es[q] += __shfl_xor(es[q], 32, 64);
            if (half == 0 && t < Tb && colok) *(f32x4 *)(a.slab_enc + (long)ub * BTH + ((long)b * T + t) * H + colg) = es;
        }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int u = u0 + 8 * (k >> 2) + (k & 3) + 4 * half;
        if (u < U1 && colok) {
            const f32x4 o = {unscale * psum2[k][0][0], unscale * psum2[k][0][1], unscale * psum2[k][1][0], unscale * psum2[k][1][1]};
            *(f32x4 *)(a.slab_pred + (long)tt * BUH + ((long)b * U1 + u) * H + colg) = o;
        }
    }
}

bool x2_dhidden_ok(int U1, int H, int V)
{
    // raw buffers over one tile's logits rows (32-bit byte offsets), W pack addressing
    return (long)((XG2_BT - 1) * (long)U1 + XG2_BU) * V * 4 < 0x7fffffffL && V % 128 == 0 && H % 128 == 0;
}

void launch_dhidden_x2(const X3Args &a, hipStream_t st)
{
    static bool attr_set[16] = {false};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    const int lds = XG2_NW * XG2_WSLOT + 2 * XG2_XSLOT;
    if (dev < 0 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)k_dhidden_x2<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)k_dhidden_x2<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)k_dhidden_x2<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)k_dhidden_x2<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (dev >= 0) attr_set[dev] = true;
    }
    dim3 grid(a.n_ublk16, (a.T + XG2_BT - 1) / XG2_BT, a.B);
    if (a.H < 512) hipLaunchKernelGGL((k_dhidden_x2<true, true>), grid, dim3(256), lds, st, a, 0);
    else hipLaunchKernelGGL((k_dhidden_x2<true, false>), grid, dim3(256), lds, st, a, 0);
    for (int hp = 1; hp * 512 < a.H; ++hp) {
        if (a.H - 512 * hp == 128 && !(X2_NT & 4)) {  // the last 128 columns: a wave per 8 t x 16 u block, A fragments straight from G's planes
            dim3 gr(a.n_ublk16, ((a.T + XG2_BT - 1) / XG2_BT + 3) / 4, a.B);
            hipLaunchKernelGGL(k_dhidden_x2r, gr, dim3(256), XR2_NW * XR2_WSLOT, st, a, hp);
        }
        else if (a.H - 512 * hp < 512) hipLaunchKernelGGL((k_dhidden_x2<false, true>), grid, dim3(256), lds, st, a, hp);
        else hipLaunchKernelGGL((k_dhidden_x2<false, false>), grid, dim3(256), lds, st, a, hp);
    }
}

// ---------------------------------------------------------------------------------------
// k_x2_make_ep (round 5): the forward produces hidden = tanh(enc + pred) once per column pass for every cell, H values per cell and
// pass — its largest VALU cost (2.3 ms of 22 at config 2 when compiled out).  tanh(e + p) = 1 - 2 / (1 + exp(2e) exp(2p)): with
//      E[b,t,h] = exp(2 enc[b,t,h])      P[b,u,h] = exp(2 pred[b,u,h])
// computed ONCE per call here (B (T + U1) H exponentials, in double, rounded once to fp32), a hidden value costs the forward one fma,
// one reciprocal and one fma instead of add, multiply, exp2, add, reciprocal, fma.  Layout: k-step major,
//      Et[b][kc][t][16]   Pt[b][kc][u][16]      (kc = h / 16)
// so that the 32 rows of an MFMA tile (consecutive u) read one contiguous 2 KiB per k-step instead of 32 half cache lines.
// The factored form is exact only while neither factor has to be clamped: |enc|, |pred| <= 43 (2 x 43 x log2 e = 124 < 126; the
// product then over- / underflows to inf / 0 exactly where tanh saturates to +-1).  Beyond that — or with non-finite inputs — this
// kernel raises `ep_flag` and the forward runs its exact form (k_joint_fwd_x2<false>: tanh of the sum, from enc and pred) instead.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_x2_make_ep(X3Args a)
{
    const int H = a.H, H4 = H / 4, KC = H / 16;
    const long n_enc = (long)a.B * a.T * H4, n_all = n_enc + (long)a.B * a.U1 * H4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    bool big = false;
    if (idx < n_all) {
        const bool is_enc = idx < n_enc;
        const long j = is_enc ? idx : idx - n_enc;
        const long row = j / H4;            // (b, t) or (b, u)
        const int h = (int)(j - row * H4) * 4;
        const int R = is_enc ? a.T : a.U1;  // rows per utterance
        const int b = (int)(row / R), r = (int)(row - (long)b * R);
        const float *src = is_enc ? a.enc + (long)b * a.enc_sb + (long)r * a.enc_st + h : a.pred + row * H + h;
        const f32x4 x = *(const f32x4 *)src;
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            big = big || !(fabsf(x[k]) <= 43.0f);  // (a NaN compares false: flagged too)
            double arg = (double)x[k] * 2.8853900817779268;  // 2 log2 e
            arg = arg > 126.0 ? 126.0 : (arg < -126.0 ? -126.0 : arg);
            y[k] = (float)exp2(arg);
        }
        float *dst = (is_enc ? a.ep_enc : a.ep_pred) + (((long)b * KC + (h >> 4)) * R + r) * 16 + (h & 15);
        *(f32x4 *)dst = y;
    }
    if (__builtin_amdgcn_ballot_w64(big) != 0 && (threadIdx.x & 63) == 0) atomicOr(a.ep_flag, 1u);
}
void launch_x2_make_ep(const X3Args &a, hipStream_t st)
{
    launch_fill32(a.ep_flag, 0u, 4, st);
    const long n = ((long)a.B * a.T + (long)a.B * a.U1) * (a.H / 4);
    hipLaunchKernelGGL(k_x2_make_ep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}
size_t x2_ep_bytes(int B, int T, int U1, int H) { return ((size_t)B * T + (size_t)B * U1) * H * 4; }

// ---------------------------------------------------------------------------------------
// W (scaled by s_W) for the forward product, fragment order, two planes:
//   [pass (512 logits columns)][c (16-deep k-step)][plane][tile(16)][lane] x 8 fp16,
//   element j = piece_plane(s_W W[v = 512pass + 128*(tile>>2) + 4*(lane&31) + (tile&3)][h = 16c + 8*(lane>>5) + j])
// (columns interleaved by 4, as in the dHidden pack: a lane's 4 tiles of a 128-column group are 4 adjacent logits).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_x2_pack_w_fwd(const float *__restrict__ W, const float *__restrict__ scales, u32x4 *__restrict__ out, int H, int V, long n)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (pass, c, tile, lane): one thread writes both planes
    if (idx >= n) return;
    const int lane = (int)(idx & 63), tile = (int)(idx >> 6) & 15;
    const int KC = H / 16;
    const int c = (int)((idx >> 10) % KC), pass = (int)((idx >> 10) / KC);
    const int v = 512 * pass + 128 * (tile >> 2) + 4 * (lane & 31) + (tile & 3);
    const int h0 = 16 * c + 8 * (lane >> 5);
    const float sw = scales[0];
    u32x4 ph = {0u, 0u, 0u, 0u}, pm = ph;
    if (v < V) {
        const float *w = W + (long)v * H + h0;
        f32x4 w0 = *(const f32x4 *)w, w1 = *(const f32x4 *)(w + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { w0[k] = x2_clamp(w0[k] * sw); w1[k] = x2_clamp(w1[k] * sw); }
        X2_SPLIT4(w0, ph, pm, 0);
        X2_SPLIT4(w1, ph, pm, 2);
    }
    u32x4 *o = out + ((long)pass * KC + c) * 2048 + tile * 64 + lane;
    o[0] = ph; o[1024] = pm;
}
size_t x2_wpack_fwd_bytes(int H, int V) { return (size_t)((V + 511) / 512) * (H / 16) * 2 * 16 * 64 * 16; }

// ---------------------------------------------------------------------------------------
// k_joint_fwd_x2: k_joint_fwd_x3 on two planes.  hidden = 2^14 tanh(enc + pred) split into its two fp16 planes (produced per
// k-step in fragment order by the lane that owns the slot), logits = hidden . W^T + bias (fp32, stored), log-softmax statistics
// and the two log-probs per lattice cell (as the fp32 route's forward).  Tile = 128 consecutive cells; 4 waves = 2 (M) x 2 (N),
// wave tile 64 cells x 256 columns = 16 accumulator tiles (256 registers); a pass = 512 logits columns, passes run back to
// back over one linear k-step sequence.  Per k-step 3 products x 16 MFMAs and ONE barrier, in the MIDDLE of the k-step (round 5):
//      block 0  ah.bh   + the 8 fragment reads of W(cs)'s mid plane + A(cs+1): the rcp / fma (tanh) pieces
//      block 1  am.bh   + the operand loads of k-step cs+2 + A(cs+1): the split pieces, then (5th tile) the two ring writes
//      --- this wave's share of W(cs+1) landed (one counted vmcnt), ONE barrier: A(cs+1), W(cs+1) published; W(cs)'s slot free ---
//      block 2  ah.bm   + the 12 fragment reads of k-step cs+1 (other register set) + the 8 DMAs of W(cs+3) + (first pass) 2 hidden stores
// (memory operations unconditional and in one fixed order per k-step — loads, DMAs, stores: every vmcnt is a count; every one of them
// is issued inside an MFMA's shadow, one per MFMA pair: four loads in a row in front of a block cost ~500 cycles of stall).
// W ring of THREE slots, filled THREE k-steps ahead (round 4: two ahead, barrier + 12 fragment reads in front of every k-step).
// The accumulators hold 2^14 s_W (logits - bias): the pass end multiplies by 2^-14 / s_W and adds the bias (one fma; the bias
// cannot ride in the accumulators' initial value here: the padding columns' -1e30 would overflow under the scale).
// Persistent workgroups, one per CU (115 KiB of LDS), tiles from one atomic counter.  Requires H % 128 == 0, V % 128 == 0.
// ---------------------------------------------------------------------------------------
#define XF2_WSLOT 32768
#define XF2_ASLOT 8192
#define XF2_NW 3   // W ring slots: the DMAs of k-step cs+3 are issued during k-step cs, behind its mid-step barrier
#ifdef RNNT_STAMPS
#ifndef X2S_WAVE
#define X2S_WAVE 0
#endif
// Diagnostic build only (-DRNNT_STAMPS): s_memtime stamps of workgroup 0, wave X2S_WAVE (0), k-steps 8..23 of its first tile:
// debug[(step-8)*8 + slot] (tools/exp_x3_stamps.py)
#define X2STAMP(slot)                                                                                       \
    do {                                                                                                    \
        if (a.debug && blockIdx.x == 0 && wave == X2S_WAVE && lane == 0 && it == 1 && cs >= 8 && cs < 24) {  \
            unsigned long long t_;                                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                       \
            a.debug[(cs - 8) * 8 + (slot)] = t_;                                                            \
        }                                                                                                   \
    } while (0)
#else
#define X2STAMP(slot) do {} while (0)
#endif
// MODE 1 (EP): hidden from the factored exponentials of k_x2_make_ep (the shipped form); MODE 0: the exact form — tanh of the sum, from
// enc and pred — for inputs outside the factored form's range.  Both are launched; `ep_flag` says which one runs (the other exits at once).
// MODE 2 (LIN): the same pipeline as a plain GEMM  Y[M,N] = X[M,K] W[N,K]^T (+ bias)  on the f16x2 pipes — the joint's input projections
// audio_ln / text_ln (reference rnnt/joint.py:8-12,26-30) and their data gradient: the A operand is a row of X scaled by a
// power of two and split (no tanh, no second operand, nothing stored beside Y), rows = cells with T = U1 = 1 (B = M, enc = X, enc_sb = its
// row stride, H = K, V = N, logits = Y, scales = {s_W, 1/s_W, s_X, 1/s_X}), no statistics, no lattice outputs; rows >= M are not stored.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_joint_fwd_x2(X3Args a, const int ntiles)
{
    constexpr bool EP = MODE == 1, LIN = MODE == 2;
    if (!LIN && (*a.ep_flag != 0) == EP) return;  // (wave-uniform: the whole grid of the form that is not selected leaves here)
    // [0, 96 KiB): W ring, 3 slots;  [96, 112 KiB): A ring, 2 slots;  then: s_den[128], s_part[2][128][2], s_next[2]
    extern __shared__ __attribute__((aligned(1024))) char s_fw[];
    float *s_den = (float *)(s_fw + XF2_NW * XF2_WSLOT + 2 * XF2_ASLOT);
    float *s_part = s_den + 128;  // [wn][row][max, sum]
    int *s_next = (int *)(s_part + 512);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    const int H = a.H, V = a.V, KC = H / 16, U1 = a.U1, T = a.T;
    const int npass = (V + 511) / 512;
    const int NS = LIN ? KC : npass * KC;  // k-steps of a tile (LIN: a tile is ONE column pass of a row block — the passes of a plain GEMM share nothing)
    const long cells = (long)a.B * T * U1;
    const float unscale = (LIN ? a.scales[3] : X2_INV_SH) * a.scales[1];
    const float sx = LIN ? a.scales[2] : 1.0f;

    const int lds0 = (int)(size_t)(lds_vptr)s_fw;
    const int xa = lds0 + XF2_NW * XF2_WSLOT + (2 * wm) * 2048 + 16 * lane;  // A read: M tiles 2wm, 2wm+1: [slot][M tile][plane][lane]
    const int xw = lds0 + XF2_NW * XF2_WSLOT + wave * 2048 + 16 * lane;       // A write: M tile `wave` (this lane's own fragment slot)
    const int wb = lds0 + (8 * wn) * 1024 + 16 * lane;                  // W read: tiles 8wn .. 8wn+7 of each plane
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(a.wpack_fwd, 0, npass * KC * XF2_WSLOT, 0x00020000);
    const int wvo = lane * 16;

    X2_CLOCK_STAMP(128 + 100);
    if (tid == 0) s_next[0] = (int)atomicAdd(a.counter, 1u);
    __syncthreads();
    int tile = s_next[0];
    for (int it = 1; tile < ntiles; ++it) {
        if (tid == 0) s_next[it & 1] = (int)atomicAdd(a.counter, 1u);
        const long row0 = (long)(LIN ? tile / npass : tile) * 128;
        const int pass0 = LIN ? tile % npass : 0, cs0 = pass0 * KC;  // LIN: this tile's column pass and the pack index of its first k-step
        // running (max, sum exp) of every row over the columns seen so far, per column half (wn): s_part[wn][row],
        // kept by the lanes 31 / 63 that end up with a row slot's wave-level statistics
        for (int k = tid; k < 256; k += 256) { s_part[2 * k] = RNNT_NEG_INF; s_part[2 * k + 1] = 0.f; }
        __syncthreads();  // s_next, s_part visible; every wave is past the previous tile's LDS reads
        const int next = s_next[it & 1];
        // A tile entirely in the time steps past one utterance's length: its logits are never read (k_dhidden_x2 zero-fills the
        // G rows of dead tiles itself), but its hidden rows must be finite (k_dw_x2 multiplies them by zeros): such a tile runs
        // the production of its first pass WITHOUT the MFMAs.
        bool dead = false;
        if (!LIN) {
            const long per = (long)T * U1, c_last = row0 + 127;
            const long b_first = row0 / per;
            dead = c_last < cells && c_last / per == b_first && (row0 - b_first * per) / U1 >= len_t(a.logit_lens, (int)b_first, T);
        }

        // ---- hidden = 2^14 tanh(enc + pred), produced per k-step IN FRAGMENT ORDER by the lane that owns the slot: lane
        // (i, half) of wave w holds row 32w + i, k = 16c + 8*half .. +7 of the MFMA A operand — 8 values from 2 x 32 bytes of
        // enc and pred, split two ways, 16 bytes per plane into the LDS ring of k-step c+1 and, in the first pass, to the
        // hidden planes in memory for k_dw_x2.  Later passes produce it again (the forward never re-reads hidden from memory).
        const long prow = row0 + 32 * wave + i;
        const long pc_ = prow < cells ? prow : cells - 1;  // rows past the lattice (last tile): any valid cell, never stored
        const int pu = (int)(pc_ % U1);
        const long pbt = pc_ / U1;
        const int pt = (int)(pbt % T), pb = (int)(pbt / T);
        // operand rows of this lane's cell.  EP: E = exp(2 enc), P = exp(2 pred), k-step major ([b][kc][row][16]: consecutive rows of a
        // k-step are contiguous); else enc and pred themselves, row major
        const float *ep = EP ? a.ep_enc + ((long)pb * KC * T + pt) * 16 + 8 * half : a.enc + (long)pb * a.enc_sb + (long)pt * a.enc_st + 8 * half;
        const float *pp = LIN ? ep : EP ? a.ep_pred + ((long)pb * KC * U1 + pu) * 16 + 8 * half : a.pred + ((long)pb * U1 + pu) * H + 8 * half;
        const long ek = EP ? (long)T * 16 : 16, pk = EP ? (long)U1 * 16 : 16;  // floats from one k-step's operands to the next's
        // (rows past the lattice produce — and store, unconditionally — the last cell's row again: the same bits to the
        // same place; hipcc counts vmcnt exactly only through unconditional memory operations)
        u32x4 *hdst = (u32x4 *)a.hidden + ((X2_EXP & 4) ? (long)(32 * wave + i) : pc_) * (H / 8) + half;  // + 2c: this lane's 16 bytes of k-step c; planes `ps` apart
        const long ps = a.plane_stride / 8;
        struct Opd { f32x4 e0, e1, p0, p1; };
        struct Prod { f2 w[4]; float ra, rb; u32x4 ph, pm; };
        // operands of k index kcs (inside a pass): plain loads, hipcc keeps their vmcnt.  (Its count for a loop-carried register is
        // the MINIMUM over every path into the loop — vmcnt(3) / vmcnt(0) in front of block 0 here, which also waits for the W
        // DMAs issued behind these loads a k-step ago.  Spelling the loads as asm with a counted wait was tried and is WRONG: the
        // loaded registers are loop-carried, and the copies hipcc places on the loop's back edge read them before the data has
        // landed — intermittently different results at full size, tools/dbg_x2_loss.py.)
        // PL (round 6): the operand of k index kcs is the PAIR OF PLANES this lane stored for it in the tile's first pass (its own 2 x 16 bytes,
        // hdst[2 kcs] and the same slot of the second plane): two loads instead of four, and nothing to compute — see run_pass.
        auto op_load1 = [&](Opd &o, int kcs, int k, auto planes_c) {  // one of the four (PL: two) 16-byte operand loads of k index kcs
            if (decltype(planes_c)::value) {
                if (k == 0) o.e0 = __builtin_bit_cast(f32x4, hdst[2 * kcs]);
                else if (k == 1) o.e1 = __builtin_bit_cast(f32x4, hdst[2 * kcs + ps]);
                return;
            }
            if (X2_EXP & 512) { const float c = (float)kcs * 0.01f; o.e0 = o.e1 = o.p0 = o.p1 = f32x4{c, -c, 0.5f * c, 0.25f}; return; }
            const float *e = ep + ek * kcs, *q = pp + pk * kcs;
            if (k == 0) o.e0 = *(const f32x4 *)e;
            else if (k == 1) o.e1 = *(const f32x4 *)(e + 4);
            else if (LIN) return;  // (one operand: two loads per k-step)
            else if (k == 2) o.p0 = *(const f32x4 *)q;
            else o.p1 = *(const f32x4 *)(q + 4);
        };
        auto op_load = [&](Opd &o, int kcs) {
#pragma unroll
            for (int k = 0; k < 4; ++k) op_load1(o, kcs, k, X2Int<0>{});
        };
        // pieces 0-7: 2^14 tanh of the 4 pairs (fast_tanh2's arithmetic: exp2 half, reciprocal half); 8-15: the 2-way split
        // of each pair (hi + residuals, then mid); 16: the two ring writes
        auto prod_piece = [&](Prod &P, const Opd &o, auto off_c, int k, auto planes_c) {  // off_c: byte offset of the target A slot in the ring
            if (decltype(planes_c)::value && k < 16) {  // the operand IS the two planes: nothing to produce
                if (k == 15) { P.ph = __builtin_bit_cast(u32x4, o.e0); P.pm = __builtin_bit_cast(u32x4, o.e1); }
            } else if ((X2_EXP & 32) && k < 16) {
                if (k == 0) { P.ph = __builtin_bit_cast(u32x4, o.e0 + o.e1); P.pm = __builtin_bit_cast(u32x4, o.p0 + o.p1); }
            } else if (k < 8) {
                const int j = k >> 1;
                const f32x4 &e = j < 2 ? o.e0 : o.e1, &pv = j < 2 ? o.p0 : o.p1;
                const int q = 2 * (j & 1);
                if (LIN) {  // s_X x, clamped into fp16's range (non-finite / garbage operands only)
                    if (!(k & 1)) P.w[j] = f2{x2_clamp(e[q] * sx), x2_clamp(e[q + 1] * sx)};
                } else if (EP) {  // 2^14 (1 - 2 / (1 + E P)): E P overflows to inf / underflows to 0 exactly where tanh is +-1
                    if (!(k & 1)) {
                        P.w[j] = f2{__builtin_amdgcn_rcpf(fmaf(e[q], pv[q], 1.0f)), __builtin_amdgcn_rcpf(fmaf(e[q + 1], pv[q + 1], 1.0f))};
                    } else {
                        P.w[j] = f2{fmaf(-2.0f * X2_SH, P.w[j][0], X2_SH), fmaf(-2.0f * X2_SH, P.w[j][1], X2_SH)};
                    }
                } else if (!(k & 1)) {
                    const f2 x = {e[q] + pv[q], e[q + 1] + pv[q + 1]};
                    const f2 av = x * (2.0f * RNNT_LOG2E);
                    P.w[j] = f2{__builtin_amdgcn_exp2f(av[0]), __builtin_amdgcn_exp2f(av[1])};
                } else {
                    const f2 ex = P.w[j] + 1.0f;
                    const f2 rr = {__builtin_amdgcn_rcpf(ex[0]), __builtin_amdgcn_rcpf(ex[1])};
                    P.w[j] = X2_SH - (2.0f * X2_SH) * rr;  // 2^14 (1 - 2 / (1 + e^2x)): the same rounding as the unscaled form
                }
            } else if (k < 16) {
                const int j = (k - 8) >> 1;
                if (!(k & 1)) {
                    const unsigned hh = x2_pack(P.w[j][0], P.w[j][1]);
                    P.ph[j] = hh;
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(P.ra) : "v"(hh), "v"(P.w[j][0]));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(P.rb) : "v"(hh), "v"(P.w[j][1]));
                } else {
                    P.pm[j] = x2_pack(P.ra, P.rb);
                }
            } else {
                const int dst = xw;  // (a local: asm operands cannot name a capture of the enclosing generic lambda)
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(dst), "v"(P.ph), "n"(decltype(off_c)::value) : "memory");
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(dst), "v"(P.pm), "n"(decltype(off_c)::value + 1024) : "memory");
            }
        };
        auto hid_store = [&](const Prod &P, int kcs) { hdst[2 * kcs] = P.ph; hdst[2 * kcs + ps] = P.pm; };
        // piece n (0..7) of this wave's share of W k-step cs -> ring slot `slot`
        auto wdma = [&](int cs, int slot, int n) {  // raw-buffer form: scalar base and offsets, one constant per-lane offset register
            if (X2_EXP & 1024) return;
#if XF2_IMM
            // pieces n = 4g .. 4g+3 share one LDS base (M0) and one scalar offset: the instruction's 12-bit immediate offset advances the
            // memory address AND the LDS address (LDS_ADDR = M0 + inst_offset + lane x 16) — the pack and the ring slot are both linear in n
            const int g4 = n & 4;
            if ((n & 3) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_vptr)(s_fw + slot * XF2_WSLOT + (wave * 8 + g4) * 1024), 16, wvo, (X2_EXP & 2) ? 0x7ff00000 : (cs * 32 + wave * 8 + g4) * 1024, 0, 0);
            if ((n & 3) == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_vptr)(s_fw + slot * XF2_WSLOT + (wave * 8 + g4) * 1024), 16, wvo, (X2_EXP & 2) ? 0x7ff00000 : (cs * 32 + wave * 8 + g4) * 1024, 1024, 0);
            if ((n & 3) == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_vptr)(s_fw + slot * XF2_WSLOT + (wave * 8 + g4) * 1024), 16, wvo, (X2_EXP & 2) ? 0x7ff00000 : (cs * 32 + wave * 8 + g4) * 1024, 2048, 0);
            if ((n & 3) == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_vptr)(s_fw + slot * XF2_WSLOT + (wave * 8 + g4) * 1024), 16, wvo, (X2_EXP & 2) ? 0x7ff00000 : (cs * 32 + wave * 8 + g4) * 1024, 3072, 0);
#else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_vptr)(s_fw + slot * XF2_WSLOT + (wave * 8 + n) * 1024), 16, wvo,
                                                     (X2_EXP & 2) ? 0x7ff00000 : (cs * 32 + wave * 8 + n) * 1024, 0, 0);
#endif
        };

        if (dead) {  // hidden rows only (finite values for k_dw_x2), no products
            for (int kc = 0; kc < KC; ++kc) {
                Opd o; Prod P;
                op_load(o, kc);
#pragma unroll
                for (int pc = 0; pc < 16; ++pc) prod_piece(P, o, X2Int<0>{}, pc, X2Int<0>{});
                hid_store(P, kc);
            }
            tile = next;
            continue;
        }

        f32x16 acc[2][8];
        auto acc_init = [&]() {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;
        };
        // pipeline prologue: W of k-steps 0, 1 and 2 by DMA; A of k-step 0 produced, stored, written to ring slot 0; operands of
        // k-step 1 requested; then everything landed + one barrier, and the fragments of k-step 0 (A slot 0, W slot 0: hi plane) read
        Opd oset[2];  // operands of k-step cs+1 live in oset[(cs+1) & 1] during k-step cs (KC is even: the k loop is unrolled by 2)
        // MFMA fragments of the CURRENT k-step, read during the previous one (two sets alternating by k-step parity: compile-time)
        struct Frag { u32x4 af[2][2], bf[8]; };
        Frag fr[2];
        // read n (0..11) of a k-step's fragments: 0-3 the A fragments (slot at byte offset xs_c of the A ring), 4-11 W's hi plane (slot `wslot`)
        auto frag_read1 = [&](Frag &f, auto xs_c, const int wslot, const int n) {  // (n: a constant once the caller's loop is unrolled)
            const int xs = xa, ws = wb + wslot * XF2_WSLOT;
            if (X2_EXP & 2048) {
                if (n >= 4) f.bf[n - 4] = u32x4{(unsigned)xs, (unsigned)ws, 0x3c003c00u, (unsigned)n};
                else f.af[n >> 1][n & 1] = u32x4{(unsigned)xs, (unsigned)ws, 0x3c003c00u, (unsigned)n};
                return;
            }
            if (n < 4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.af[(n >> 1) & 1][n & 1]) : "v"(xs), "n"(decltype(xs_c)::value + ((n >> 1) & 1) * 2048 + (n & 1) * 1024));
            else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.bf[n >= 4 ? n - 4 : 0]) : "v"(ws), "n"((n >= 4 ? n - 4 : 0) * 1024));
        };
        auto frag_read = [&](Frag &f, auto xs_c, const int wslot) {
#pragma unroll
            for (int n = 0; n < 12; ++n) frag_read1(f, xs_c, wslot, n);
        };
        // the fragments have landed: every register of the set is named, so that nothing hipcc places behind this point (the loop's
        // back-edge copies of loop-carried registers included) reads one before its data is there
        auto frag_landed = [&](Frag &f) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(f.af[0][0]), "+v"(f.af[0][1]), "+v"(f.af[1][0]), "+v"(f.af[1][1]),
                           "+v"(f.bf[0]), "+v"(f.bf[1]), "+v"(f.bf[2]), "+v"(f.bf[3]), "+v"(f.bf[4]), "+v"(f.bf[5]), "+v"(f.bf[6]), "+v"(f.bf[7])
                         :: "memory");
        };
        {
#pragma unroll
            for (int n = 0; n < 8; ++n) wdma(cs0, 0, n);
#pragma unroll
            for (int n = 0; n < 8; ++n) wdma(cs0 + (NS > 1 ? 1 : 0), 1, n);
#pragma unroll
            for (int n = 0; n < 8; ++n) wdma(cs0 + (NS > 2 ? 2 : NS - 1), 2, n);
            Opd o; Prod P;
            op_load(oset[1], KC > 1 ? 1 : 0);
            op_load(o, 0);
#pragma unroll
            for (int pc = 0; pc < 17; ++pc) prod_piece(P, o, X2Int<0>{}, pc, X2Int<0>{});
            if (!LIN) hid_store(P, 0);  // (the plain GEMM stores nothing beside Y: X3Args::hidden is null there)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // W of k-steps 0-2 (this wave's share), operands, the stores
            x2_lds_barrier();                                 // ... of every wave; A slot 0 written
            frag_read(fr[0], X2Int<0>{}, 0);
            frag_landed(fr[0]);
        }

        int cs = 0, wsl = 0;  // k-step (linear over the passes) and its W ring slot (cs % 3)
        // one pass; STORE: the first — the produced planes also go to memory.  Two straight-line instantiations, the first pass
        // outside the loop over the others (a branch between two k loops, like a conditional accumulator re-initialisation
        // inside one, makes hipcc carry the 256 accumulator registers through VGPR phis and spill)
        auto run_pass = [&](auto store_c, const int pass) {
          acc_init();
          constexpr bool STORE = decltype(store_c)::value != 0;
          // One k-step (round 5: the barrier sits in the MIDDLE of the k-step and the fragments of a k-step are read during the
          // one before — the round-4 form paid barrier + 12 fragment reads, ~370 of 2 860 cycles, in front of every k-step's MFMAs):
          //   block 0  ah.bh  + the 8 reads of W(cs)'s mid plane + A(cs+1): the tanh pieces
          //            the operand loads of k-step cs+2
          //   block 1  am.bh  + A(cs+1): the split pieces, then its two ring writes
          //   --- this wave's share of W(cs+1) landed (counted vmcnt), its ring writes done, ONE barrier: A(cs+1) and W(cs+1) are
          //       published, and every wave is past its last read of W(cs): that ring slot is free ---
          //   block 2  ah.bm  + the 12 fragment reads of k-step cs+1 (into the other register set) + the 8 DMAs of W(cs+3) into
          //            the slot W(cs) just left (three slots, filled THREE k-steps ahead) + (first pass) the 2 hidden stores
          // PAR = the k-step's parity = its A ring slot and fragment register set (KC is even: cs and kc have the same parity).
          // OCP: the operands of k-step cs+1 (requested during the previous k-step) are its stored planes; ONP: this k-step requests the
          // planes of k-step cs+2 (round 6, below)
          auto kstep = [&](auto par_c, const int kc, auto ocp_c, auto onp_c) {
            constexpr int par = decltype(par_c)::value;
            constexpr bool OCP = decltype(ocp_c)::value != 0, ONP = decltype(onp_c)::value != 0;
            constexpr int XN = (1 - par) * XF2_ASLOT;
            Frag &fc = fr[par], &fn = fr[1 - par];
            const int ws = wb + wsl * XF2_WSLOT;  // (the W slot is a run-time third: one v_add per k-step)
            // the k-step whose W is requested now (past the end: the last one again, never read), the next k-step of the pass and
            // the one after (operand loads)
            const int csn = cs0 + (cs + 3 < NS ? cs + 3 : NS - 1), kcn = kc + 1 < KC ? kc + 1 : 0, kcnn = kcn + 1 < KC ? kcn + 1 : 0;
            const int wsn = wsl == 2 ? 0 : wsl + 1;  // (cs + 1) % 3: the slot whose fragments are read in block 2
            const Opd &ocur = oset[(par + 1) & 1];  // operands of k-step cs+1 (requested during the previous k-step)
            Opd &onext = oset[par & 1];       // refilled with those of k-step cs+2
            Prod P;
            u32x4 bn[8];
            X2STAMP(0);
            auto block = [&](auto pa_c, const u32x4 (&bcur)[8], auto blk_c) {
                constexpr int PA = decltype(pa_c)::value, BLK = decltype(blk_c)::value;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (!(X2_EXP & 128)) {
                        acc[0][q] = x2_mfma(fc.af[0][PA], bcur[q], acc[0][q]);
                        acc[1][q] = x2_mfma(fc.af[1][PA], bcur[q], acc[1][q]);
                    }
                    if (BLK == 0) {
                        if (!(X2_EXP & 2048)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bn[q]) : "v"(ws), "n"(16384 + q * 1024));
                        prod_piece(P, ocur, X2Int<XN>{}, q, X2Int<OCP>{});
                    }
                    if (BLK == 1) {
                        if (q < ((LIN || ONP) ? 2 : 4)) op_load1(onext, kcnn, q, X2Int<ONP>{});  // operands of k-step cs+2 (needed a whole k-step from now): in FRONT of the k-step's DMAs
                        if (q < 4) { prod_piece(P, ocur, X2Int<XN>{}, 8 + 2 * q, X2Int<OCP>{}); prod_piece(P, ocur, X2Int<XN>{}, 9 + 2 * q, X2Int<OCP>{}); }
                        if (q == 4) prod_piece(P, ocur, X2Int<XN>{}, 16, X2Int<OCP>{});  // the ring writes: done well before the barrier's lgkmcnt(0)
                    }
                    if (BLK == 2) {  // nothing of the k-step is issued outside an MFMA's shadow: the 12 fragment reads of k-step cs+1, the 8 DMAs, the 2 stores
                        if (q < 4) { frag_read1(fn, X2Int<XN>{}, wsn, 2 * q); frag_read1(fn, X2Int<XN>{}, wsn, 2 * q + 1); }
                        else frag_read1(fn, X2Int<XN>{}, wsn, 4 + q);
                        wdma(csn, wsl, q);
                        if (STORE && q == 6) hdst[2 * kcn] = P.ph;
                        if (STORE && q == 7) hdst[2 * kcn + ps] = P.pm;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            block(X2Int<0>{}, fc.bf, X2Int<0>{});   // ah.bh
            X2STAMP(1);
            block(X2Int<1>{}, fc.bf, X2Int<1>{});   // am.bh
            X2STAMP(2);
            // this wave's share of W(cs+1) landed: its DMAs were issued in block 2 of k-step cs-2.  vmcnt retires in order; behind
            // them came (first pass) 2 hidden stores, k-step cs-1's 4 operand loads, 8 DMAs and (first pass) 2 stores, and this
            // k-step's 4 operand loads.  The first two k-steps of a pass: their W(cs+1) was waited for before the previous pass's
            // logits stores (pass end below) / with the tile prologue.
            // (round 6: a k-step whose operands are stored planes issues 2 loads, not 4: previous k-step's loads = OCP ? 2 : 4, this one's
            // = ONP ? 2 : 4)
            if (kc >= 2) {
                constexpr int NV = LIN ? 12 : (STORE ? 4 : 0) + (OCP ? 2 : 4) + 8 + (ONP ? 2 : 4);
                static_assert(NV == 12 || NV == 16 || NV == 18 || NV == 20, "counted vmcnt of the forward's k-step");
                if (NV == 12) asm volatile(RNNT_VMCNT(12) ::: "memory");  // (LIN: 2 operand loads per k-step: 2 + 8 + 2; planes: the same)
                else if (NV == 16) asm volatile(RNNT_VMCNT(16) ::: "memory");
                else if (NV == 18) asm volatile(RNNT_VMCNT(18) ::: "memory");
                else asm volatile(RNNT_VMCNT(20) ::: "memory");
            }
            X2STAMP(3);
            if (!(X2_EXP & 4096)) x2_lds_barrier();  // (lgkmcnt(0): bn and the ring writes) publishes A(cs+1), W(cs+1); frees W(cs)'s slot
            X2STAMP(4);
            asm volatile("" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]), "+v"(bn[4]), "+v"(bn[5]), "+v"(bn[6]), "+v"(bn[7]));
            block(X2Int<0>{}, bn, X2Int<2>{});   // ah.bm  (the hidden stores: the youngest memory operations of the k-step; the pass's last k-step re-stores k-step 0)
            frag_landed(fn);
            X2STAMP(5);
            (void)kcn;
            ++cs;
            wsl = wsn;
          };
          // Round 6: only a tile's FIRST pass produces the hidden values.  It stores them as two fp16 planes for k_dw_x2 anyway, and every
          // lane stores exactly the 2 x 16 bytes it later needs as its A-ring slot — so the passes after the first load those back (the
          // lane's own stores, a whole pass old: landed; no other workgroup touches the rows) instead of loading E and P and running the
          // fma / rcp / fma / split arithmetic again: 2 operand loads per k-step instead of 4 and no production VALU in V/512 - 1 of the
          // V/512 passes (config 2: one of two; config 5: 31 of 32).  Operands are requested two k-steps ahead, so the first pass's last two
          // k-steps already request planes (of the next pass's k-steps 0 and 1), and its last one copies where it used to produce.
          if constexpr (LIN || !X2_REREAD) {
              for (int kc0 = 0; kc0 < KC; kc0 += 2) { kstep(X2Int<0>{}, kc0, X2Int<0>{}, X2Int<0>{}); kstep(X2Int<1>{}, kc0 + 1, X2Int<0>{}, X2Int<0>{}); }
          } else if constexpr (STORE) {
              for (int kc0 = 0; kc0 < KC - 2; kc0 += 2) { kstep(X2Int<0>{}, kc0, X2Int<0>{}, X2Int<0>{}); kstep(X2Int<1>{}, kc0 + 1, X2Int<0>{}, X2Int<0>{}); }
              kstep(X2Int<0>{}, KC - 2, X2Int<0>{}, X2Int<1>{});
              kstep(X2Int<1>{}, KC - 1, X2Int<1>{}, X2Int<1>{});
          } else {
              for (int kc0 = 0; kc0 < KC; kc0 += 2) { kstep(X2Int<0>{}, kc0, X2Int<1>{}, X2Int<1>{}); kstep(X2Int<1>{}, kc0 + 1, X2Int<1>{}, X2Int<1>{}); }
          }
          // pass complete: unscale, add the bias, store the logits, update the statistics.  V % 128 == 0: a lane's two 4-column
          // groups exist or not for the whole wave.  The row loop is ONE basic block per case; the store address is a scalar
          // row pointer + one 32-bit per-lane offset.
          // (W of the next pass's first two k-steps — requested during the last two above — lands before the logits stores are
          // queued behind it: the k-steps' counted waits cannot see past 63 younger operations)
          if (STORE) asm volatile(RNNT_VMCNT(2) ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          {
            const int cw = 512 * pass + 256 * wn;
            const unsigned lane_off = (unsigned)(((4 * half) * V + 4 * i) * 4);
            char *tile_base = (char *)(a.logits + ((X2_EXP & 8) ? 0L : row0) * V + cw);
            const bool has_b = !LIN || a.bias != nullptr;
            const f32x4 b0 = has_b && cw + 4 * i < V ? *(const f32x4 *)(a.bias + cw + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 b1 = has_b && cw + 128 + 4 * i < V ? *(const f32x4 *)(a.bias + cw + 128 + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
            auto epilogue = [&](auto both_c) {
                constexpr bool BOTH = decltype(both_c)::value != 0;
                // one row slot: unscale + bias, store, (max, sum exp) over this wave's 128 / 256 columns of the pass (8 values per
                // lane, then the 32 lanes of the half on the DPP crossbar), running statistics of the row (lanes 31 / 63)
                auto slot = [&](int mt, int r) {
                    // accumulator reads spelled as (volatile) asm: they stay here — left to hipcc, all 256 v_accvgpr_read are
                    // hoisted in front of the first store and spilled
                    f32x4 o0, o1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float x0, x1;
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x0) : "a"(acc[mt][q][r]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x1) : "a"(acc[mt][4 + q][r]));
                        o0[q] = fmaf(x0, unscale, b0[q]); o1[q] = fmaf(x1, unscale, b1[q]);
                    }
                    char *rowp = tile_base + (long)(32 * (2 * wm + mt) + (r & 3) + 8 * (r >> 2)) * V * 4;  // wave-uniform
                    if (LIN) {  // Y has no padding rows: rows past M are not stored (the k loop's memory operations stay unconditional)
                        if (row0 + 32 * (2 * wm + mt) + (r & 3) + 8 * (r >> 2) + 4 * half < cells) {
                            *(f32x4 *)(rowp + lane_off) = o0;
                            if (BOTH) *(f32x4 *)(rowp + lane_off + 512) = o1;
                        }
                        return;
                    }
                    if (X2_EXP & 256) {
                        asm volatile("" :: "v"(o0), "v"(o1));
                    } else if (X2_EXP & 16) {
                        *(f32x4 *)(rowp + lane_off) = o0;
                        if (BOTH) *(f32x4 *)(rowp + lane_off + 512) = o1;
                    } else {
                        __builtin_nontemporal_store(o0, (f32x4 *)(rowp + lane_off));
                        if (BOTH) __builtin_nontemporal_store(o1, (f32x4 *)(rowp + lane_off + 512));
                    }
                    if (X2_EXP & 64) return;
                    float m8 = fmaxf(fmaxf(o0[0], o0[1]), fmaxf(o0[2], o0[3]));
                    if (BOTH) m8 = fmaxf(m8, fmaxf(fmaxf(o1[0], o1[1]), fmaxf(o1[2], o1[3])));
                    const float M = half_max_dpp(m8, half);
                    const float nm2 = -M * RNNT_LOG2E;
                    float e = (__builtin_amdgcn_exp2f(fmaf(o0[0], RNNT_LOG2E, nm2)) + __builtin_amdgcn_exp2f(fmaf(o0[1], RNNT_LOG2E, nm2))) +
                              (__builtin_amdgcn_exp2f(fmaf(o0[2], RNNT_LOG2E, nm2)) + __builtin_amdgcn_exp2f(fmaf(o0[3], RNNT_LOG2E, nm2)));
                    if (BOTH)
                        e += (__builtin_amdgcn_exp2f(fmaf(o1[0], RNNT_LOG2E, nm2)) + __builtin_amdgcn_exp2f(fmaf(o1[1], RNNT_LOG2E, nm2))) +
                             (__builtin_amdgcn_exp2f(fmaf(o1[2], RNNT_LOG2E, nm2)) + __builtin_amdgcn_exp2f(fmaf(o1[3], RNNT_LOG2E, nm2)));
                    const float S_ = half_sum_dpp(e, half);  // lanes 31 / 63 hold the sums
                    if (i == 31) {
                        float *sp = s_part + (wn * 128 + 32 * (2 * wm + mt) + (r & 3) + 8 * (r >> 2) + 4 * half) * 2;
                        const float m_o = sp[0], s_o = sp[1];
                        const float mn = fmaxf(m_o, M);
                        sp[0] = mn;
                        sp[1] = s_o * __builtin_amdgcn_exp2f((m_o - mn) * RNNT_LOG2E) + S_ * __builtin_amdgcn_exp2f((M - mn) * RNNT_LOG2E);
                    }
                };
                // TWO row slots at a time: with one wave per SIMD a slot is a chain of dependent latencies (accumulator reads, the
                // two DPP reductions, exp2, the LDS update); the second slot's chain fills the first one's gaps
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        slot(mt, r);
                        slot(mt, r + 1);
                        __builtin_amdgcn_sched_barrier(0);  // (a pair at a time: see the accumulator reads)
                    }
            };
            if (cw + 128 < V) epilogue(X2Int<1>{});
            else if (cw < V) epilogue(X2Int<0>{});
          }
        };
        run_pass(X2Int<(LIN ? 0 : 1)>{}, pass0);
        if (!LIN)
            for (int pass = 1; pass < npass; ++pass) run_pass(X2Int<0>{}, pass);

        // ---- log-softmax denominators: the two column halves (wn) of every row
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every logits / hidden store of the workgroup has left its wave; s_part complete
        if (LIN) { tile = next; continue; }
        if (tid < 128) {
            const float m0 = s_part[tid * 2], s0 = s_part[tid * 2 + 1];
            const float m1 = s_part[(128 + tid) * 2], s1 = s_part[(128 + tid) * 2 + 1];
            const float M = fmaxf(m0, m1);
            const float S_ = s0 * __builtin_amdgcn_exp2f((m0 - M) * RNNT_LOG2E) + s1 * __builtin_amdgcn_exp2f((m1 - M) * RNNT_LOG2E);
            s_den[tid] = M + __logf(S_);
        }
        __syncthreads();
        // thread = (row = tid & 127, which = tid >> 7): logit[blank] / logit[label] of the row, read through L2
        // (agent-scope loads bypass the CU's vector L1; the stores above are complete: vmcnt(0) + barrier)
        {
            const int row = tid & 127, which = tid >> 7;
            const long cell = row0 + row;
            if (cell < cells) {
                const int u = (int)(cell % U1);
                const long bt = cell / U1;
                const int t = (int)(bt % T), b = (int)(bt / T);
                const int Ub = len_u(a.target_lens, b, U1);
                if (t < len_t(a.logit_lens, b, T) && u <= Ub) {
                    const float den = s_den[row];
                    const float *lrow = a.logits + cell * V;
                    const long si = skew_index(b, t, u, a.D, U1);
                    if (which == 0) {
                        const float lb = __hip_atomic_load(lrow + a.blank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        a.denom_s[si] = den;
                        a.lpb_s[si] = lb - den;
                    } else {
                        float le = 0.f;
                        if (u < Ub) {
                            const int y = a.targets[(long)b * (U1 - 1) + u];
                            le = __hip_atomic_load(lrow + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - den;
                        }
                        a.lpe_s[si] = le;
                    }
                }
            }
        }
        tile = next;
    }
    X2_CLOCK_STAMP(128 + 102);
}

bool x2_fwd_ok(int U1, int H, int V) { return H % 128 == 0 && V % 128 == 0 && (long)128 * H * 2 < 0x7fffffffL; }

void launch_joint_fwd_x2(const X3Args &a, hipStream_t st)
{
    static bool attr_set[16] = {false};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    const int lds = XF2_NW * XF2_WSLOT + 2 * XF2_ASLOT + 128 * 4 + 2 * 128 * 2 * 4 + 16;
    if (dev < 0 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)k_joint_fwd_x2<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)k_joint_fwd_x2<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)k_joint_fwd_x2<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (dev >= 0) attr_set[dev] = true;
    }
    const long cells = (long)a.B * a.T * a.U1;
    const bool lin = (a.flags & X2_FLAG_LINEAR) != 0;
    const int ntiles = (int)((cells + 127) / 128) * (lin ? (a.V + 511) / 512 : 1);  // (plain GEMM: a tile = one column pass of a row block)
    if (!lin) launch_fill32(a.counter, 0u, 4, st);  // tile counter of the persistent workgroups (plain GEMM: zeroed with its scales, k_x2_lin_scales)
    const int nwg = ntiles < a.n_cu ? ntiles : a.n_cu;  // one workgroup per CU
    // both forms; k_x2_make_ep's flag (device memory: no host round trip) selects the one that runs, the other's workgroups exit at once
    if (a.flags & X2_FLAG_LINEAR) { hipLaunchKernelGGL(k_joint_fwd_x2<2>, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles); return; }
    hipLaunchKernelGGL(k_joint_fwd_x2<1>, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
    hipLaunchKernelGGL(k_joint_fwd_x2<0>, dim3((unsigned)nwg), dim3(256), lds, st, a, ntiles);
}

#ifdef RNNT_LAB
#include "lab/x2_lab_fwd.inc"  // k_joint_fwd_x2d (RNNT_VARIANT_X2_FWD_2WG): measured equal to k_joint_fwd_x2, kept as lab equipment
#endif

void launch_x2_pack_w(const X3Args &a, float *scales, hipStream_t st)
{
    hipLaunchKernelGGL(k_x2_wscale, dim3(1), dim3(1024), 0, st, a.W, (long)a.V * a.H / 4, scales);
    const long nd = (long)((a.H + 511) / 512) * (a.V / 16) * 16 * 64;
    hipLaunchKernelGGL(k_x2_pack_w_dh, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, a.W, (const float *)scales, (u32x4 *)a.wpack_dh, a.H, a.V, nd);
    const long nf = (long)((a.V + 511) / 512) * (a.H / 16) * 16 * 64;
    hipLaunchKernelGGL(k_x2_pack_w_fwd, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, a.W, (const float *)scales, (u32x4 *)a.wpack_fwd, a.H, a.V, nf);
}


// ---------------------------------------------------------------------------------------
// The joint's input projections on the f16x2 pipes (round 5; reference rnnt/joint.py:8-12,26-30: audio_ln / text_ln are
// nn.Linear layers applied to the encoder / predictor outputs before the joint).  A Linear layer and its backward are three GEMMs:
//      y  = x W^T + b        (M x K) (N x K)^T   -> k_joint_fwd_x2<2>: the joint forward's pipeline as a plain GEMM (rows of x split on the fly)
//      dx = dy W             (M x N) (K x N)^T   -> the same kernel on W^T (a 64 x 64-tiled transposing copy of the weight)
//      dW = dy^T x, db = column sums of dy       -> k_dw_x2<4>, unchanged: dy as the "G" operand (two fp16 planes interleaved per
//                                                   32-column chunk), x as the "hidden" operand (two separate planes), K = the M rows
// Operand scales (powers of two) come from the data's largest magnitude, found on the device every call (k_x2_absmax + k_x2_lin_scales);
// the packs / planes are scaled when they are made and the outputs unscaled where they are written.  K % 128 == 0, N % 128 == 0.
// Workspace (caller-owned, rnnt_engine_linear_x2_workspace_bytes): 512 B of scale / counter / table words, then the forward's W pack
// (forward) or W^T, its pack, dy's and x's planes and the dW split-K slabs (backward).
// ---------------------------------------------------------------------------------------
// largest magnitudes of up to three row-major matrices in ONE launch: grid (256, n), workgroup (b, t) leaves the maximum of its share of
// tensor t in partial[t][b] — no atomics, nothing to zero first (round 5: an atomicMax per wave of 1 024 workgroups on one word took 48 us for
// a 4 MB tensor — the 4 096 same-address atomics, not the bytes — five times per Linear forward + backward: 0.25 of its 0.62 ms at 6 432 rows)
struct X2AbsArgs { const float *x[3]; long ld[3], rows[3]; int cols4[3]; };
__global__ __launch_bounds__(256) void k_x2_absmax(X2AbsArgs a, float *__restrict__ partial)
{
    __shared__ float s_m[4];
    const int t = blockIdx.y;
    const float *x = a.x[t];
    const long ld = a.ld[t], n = a.rows[t] * a.cols4[t];
    const int cols4 = a.cols4[t];
    float m = 0.f;
    bool bad = false;  // a NaN or an infinity among the entries (fmaxf drops NaNs; the split's clamp would turn either into a finite fp16)
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
        const long r = idx / cols4;
        const int c = (int)(idx - r * cols4);
        const f32x4 w = *(const f32x4 *)(x + r * ld + 4 * c);
        const float mw = fmaxf(fmaxf(fabsf(w[0]), fabsf(w[1])), fmaxf(fabsf(w[2]), fabsf(w[3])));
        bad = bad || !(fabsf(w[0]) <= X2_F32_MAX) || !(fabsf(w[1]) <= X2_F32_MAX) || !(fabsf(w[2]) <= X2_F32_MAX) || !(fabsf(w[3]) <= X2_F32_MAX);
        m = fmaxf(m, mw);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int any_bad = __syncthreads_or(bad ? 1 : 0);
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    // a non-finite operand leaves a NaN as its workgroup's "maximum": k_x2_lin_scales turns it into a NaN reciprocal scale, and every output
    // the operand feeds is multiplied by that — non-finite inputs give non-finite outputs, as torch.nn.functional.linear's do (round-5 advice)
    if (threadIdx.x == 0) partial[t * 256 + blockIdx.x] = any_bad ? __uint_as_float(0x7fc00000u) : fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
}
// one workgroup of 256: scales[2i] = 2^(14 - ceil(log2 max_i)), scales[2i + 1] = its reciprocal (1 for an all-zero operand; NaN for an operand with a NaN or an infinity in it),
// i = 0 .. n-1, from the partial maxima; the call's counter words zeroed (the forward's tile counter; dW's progress words) and dW's table written
// (k_dw_table's format, B = 1, every granule live) — what four more launches did before
__global__ __launch_bounds__(256) void k_x2_lin_scales(const float *__restrict__ partial, float *__restrict__ scales, int n, unsigned *__restrict__ zero0,
                                                       int nzero0, unsigned *__restrict__ zero1, int nzero1, long *__restrict__ tab, long ngran)
{
    __shared__ float s_m[4];
    for (int i = 0; i < n; ++i) {
        float m = partial[i * 256 + threadIdx.x];
        const int nonfinite = __syncthreads_or(m != m ? 1 : 0);  // (k_x2_absmax: NaN = the operand holds a NaN or an infinity)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0 && nonfinite) {  // the planes / packs are finite garbage (scale 1, clamped); the outputs' unscale factor is NaN
            scales[2 * i] = 1.0f;
            scales[2 * i + 1] = __uint_as_float(0x7fc00000u);
        } else if (threadIdx.x == 0) {
            const unsigned bits = __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])));
            float s = 1.0f;
            const int e = (int)(bits >> 23) & 0xff;
            if (e > 0 && e < 255) {
                int k = 14 - (e - 126);
                k = k > 100 ? 100 : (k < -100 ? -100 : k);
                s = __uint_as_float((unsigned)(127 + k) << 23);
            }
            scales[2 * i] = s;
            scales[2 * i + 1] = 1.0f / s;
        }
    }
    for (int j = threadIdx.x; j < nzero0; j += 256) zero0[j] = 0u;
    for (int j = threadIdx.x; j < nzero1; j += 256) zero1[j] = 0u;
    if (tab && threadIdx.x == 0) { tab[0] = 0; tab[1] = ngran; tab[2] = 0; tab[3] = ngran; }
}
// rows of a row-major fp32 matrix -> s x as two fp16 planes.  INTER: the planes interleaved per 32-column chunk, [32 x hi | 32 x mid] over the
// chunk's 128 bytes (the G operand's layout: one thread = one chunk); else two separate planes `plane_stride` elements apart (the hidden
// operand's layout: one thread = 8 columns).
template <bool INTER>
__global__ __launch_bounds__(256) void k_x2_split_rows(const float *__restrict__ x, long ld, long rows, long rows_out, int cols,
                                                       const float *__restrict__ scale, void *__restrict__ dst, long plane_stride)
{
    // rows [rows, rows_out): zeros (the padding rows the dW kernel's last k-steps read; three fill launches before)
    const float s = scale[0];
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (INTER) {
        const int VC = cols / 32;
        if (idx >= rows_out * VC) return;
        const long r = idx / VC;
        const int c = (int)(idx - r * VC);
        const f32x4 *p = (const f32x4 *)(x + (r < rows ? r : 0) * ld + 32 * c);
        u32x4 ph[4], pm[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 v = r < rows ? p[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = x2_clamp(v[k] * s);
            X2_SPLIT4(v, ph[i >> 1], pm[i >> 1], 2 * (i & 1));
        }
        u32x4 *o = (u32x4 *)((float *)dst + r * cols + 32 * c);
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[i] = ph[i]; o[4 + i] = pm[i]; }
    } else {
        const int H8 = cols / 8;
        if (idx >= rows_out * H8) return;
        const long r = idx / H8;
        const int h = (int)(idx - r * H8) * 8;
        const float *xr = x + (r < rows ? r : 0) * ld + h;
        f32x4 t0 = *(const f32x4 *)xr, t1 = *(const f32x4 *)(xr + 4);
        if (r >= rows) t0 = t1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) { t0[k] = x2_clamp(t0[k] * s); t1[k] = x2_clamp(t1[k] * s); }
        u32x4 ph, pm;
        X2_SPLIT4(t0, ph, pm, 0);
        X2_SPLIT4(t1, ph, pm, 2);
        u32x4 *o = (u32x4 *)((unsigned short *)dst + r * cols + h);
        o[0] = ph; o[plane_stride / 8] = pm;
    }
}
// out[i] = scale_a scale_b sum_s slab[s][i]  (fixed order: bitwise reproducible)
__global__ __launch_bounds__(256) void k_x2_reduce_scaled(const float *__restrict__ slab, float *__restrict__ out, long n4, long stride4, int nsplit,
                                                          const float *__restrict__ sa, const float *__restrict__ sb)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 acc = ((const f32x4 *)slab)[i];
    for (int s = 1; s < nsplit; ++s) acc += ((const f32x4 *)slab)[i + s * stride4];
    const float r = sa[0] * (sb ? sb[0] : 1.0f);
    ((f32x4 *)out)[i] = acc * r;
}

namespace {
struct LinWs { size_t wt, pack, pa, pb, slab_w, slab_b, prog, total; long rows_pad, rows_alloc; int n_split; };
LinWs lin_layout(int M, int K, int N, bool bwd)
{
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    LinWs L{};
    size_t o = 4096;  // +0 scales[8], +128 tile counter, +192 the dW table (4 longs), +1024 the partial maxima [3][256]
    if (!bwd) { L.pack = o; o += al(x2_wpack_fwd_bytes(K, N)); L.total = o; return L; }
    L.rows_pad = ((long)M + 1 + 31) / 32 * 32;
    L.rows_alloc = (L.rows_pad + 96 + 127) / 128 * 128;
    const long tiles = x2_dw_tiles(K, N);  // (the dW kernel's H = K, V = N)
    long ns = 256 / tiles;
    if (ns < 1) ns = 1;
    if (ns > L.rows_pad / 32) ns = L.rows_pad / 32;
    L.n_split = (int)ns;
    L.wt = o; o += al((size_t)N * K * 4);
    L.pack = o; o += al(x2_wpack_fwd_bytes(N, K));       // pack of W^T as the [K x N] "weight" of dx = dy (W^T)^T
    L.pa = o; o += al((size_t)L.rows_alloc * N * 4);     // dy: interleaved planes
    L.pb = o; o += al((size_t)L.rows_alloc * K * 4);     // x: two planes
    L.slab_w = o; o += al((size_t)L.n_split * N * K * 4);
    L.slab_b = o; o += al((size_t)L.n_split * N * 4);
    L.prog = o; o += al((size_t)L.n_split * 64);
    L.total = o;
    return L;
}
int lin_cus()
{
    static thread_local int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!cus[dev]) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus[dev] = p.multiProcessorCount;
        if (cus[dev] <= 0) cus[dev] = 256;
    }
    return cus[dev];
}
// the operand scales of a call: one launch for the maxima of its n tensors, one for the scales (+ the counter words and dW's table)
struct LinT { const float *x; long ld, rows; int cols; };
void lin_scales(const LinT *t, int n, char *w, unsigned *zero1, int nzero1, long *tab, long ngran, hipStream_t st)
{
    X2AbsArgs a{};
    for (int i = 0; i < n; ++i) { a.x[i] = t[i].x; a.ld[i] = t[i].ld; a.rows[i] = t[i].rows; a.cols4[i] = t[i].cols / 4; }
    float *partial = (float *)(w + 1024);
    hipLaunchKernelGGL(k_x2_absmax, dim3(256, n), dim3(256), 0, st, a, partial);
    hipLaunchKernelGGL(k_x2_lin_scales, dim3(1), dim3(256), 0, st, partial, (float *)w, n, (unsigned *)(w + 128), 16, zero1, nzero1, tab, ngran);
}
// y[M,N] = x[M,K] wmat[N,K]^T (+ bias) through k_joint_fwd_x2<2>; scales = {s_W, 1/s_W, s_X, 1/s_X} on the device, the pack made here
void lin_gemm_nt(const float *x, long ldx, const float *wmat, const float *bias, int M, int K, int N, float *y, const float *scales, void *pack,
                 unsigned *counter, hipStream_t st)
{
    const long nf = (long)((N + 511) / 512) * (K / 16) * 16 * 64;
    hipLaunchKernelGGL(k_x2_pack_w_fwd, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, wmat, scales, (u32x4 *)pack, K, N, nf);
    X3Args a{};
    a.enc = x; a.enc_sb = ldx; a.enc_st = 0; a.pred = nullptr; a.W = wmat; a.bias = bias;
    a.B = M; a.T = 1; a.U1 = 1; a.H = K; a.V = N; a.logits = y; a.wpack_fwd = pack; a.scales = scales;
    a.counter = counter; a.n_cu = lin_cus(); a.flags = X2_FLAG_LINEAR;  // (the counter was zeroed by k_x2_lin_scales)
    launch_joint_fwd_x2(a, st);
}
}  // namespace

size_t x2_linear_ws_bytes(int M, int K, int N, bool bwd) { return lin_layout(M, K, N, bwd).total; }
bool x2_linear_ok(int M, int K, int N) { return M > 0 && K % 128 == 0 && N % 128 == 0 && x2_fwd_ok(1, K, N) && x2_fwd_ok(1, N, K); }

void launch_linear_x2_fwd(const float *x, long ldx, const float *W, const float *bias, int M, int K, int N, float *y, void *ws, hipStream_t st)
{
    const LinWs L = lin_layout(M, K, N, false);
    char *w = (char *)ws;
    float *scales = (float *)w;
    const LinT t[2] = {{W, K, N, K}, {x, ldx, M, K}};
    lin_scales(t, 2, w, nullptr, 0, nullptr, 0, st);
    lin_gemm_nt(x, ldx, W, bias, M, K, N, y, scales, w + L.pack, (unsigned *)(w + 128), st);
}

void launch_linear_x2_bwd(const float *x, long ldx, const float *W, const float *dy, int M, int K, int N, float *dx, float *dW, float *db, void *ws,
                          hipStream_t st)
{
    const LinWs L = lin_layout(M, K, N, true);
    char *w = (char *)ws;
    float *scales = (float *)w;             // {s_W, 1/s_W, s_dy, 1/s_dy, s_x, 1/s_x}
    long *tab = (long *)(w + 192);
    const LinT t[3] = {{W, K, N, K}, {dy, N, M, N}, {x, ldx, M, K}};
    lin_scales(t, 3, w, (unsigned *)(w + L.prog), L.n_split * 16, tab, L.rows_pad / XW2_GRAN, st);
    if (dx) {  // dx[M,K] = dy[M,N] (W^T)[K,N]^T
        float *wt = (float *)(w + L.wt);
        launch_copy_enc(W, 0, 1, K, wt, 1, K, N, st);  // wt[k][n] = W[n][k]
        lin_gemm_nt(dy, N, wt, nullptr, M, N, K, dx, scales, w + L.pack, (unsigned *)(w + 128), st);
    }
    // dW[N,K] = dy^T x, db = column sums of dy: k_dw_x2 on dy's interleaved planes (its G operand) and x's planes (its hidden operand);
    // the padding rows [M, rows_alloc) are written as zeros by the split kernels
    float *pa = (float *)(w + L.pa);
    unsigned short *pb = (unsigned short *)(w + L.pb);
    {
        const long na = L.rows_alloc * (N / 32), nb = L.rows_alloc * (K / 8);
        hipLaunchKernelGGL(k_x2_split_rows<true>, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, dy, (long)N, (long)M, L.rows_alloc, N, scales + 2, (void *)pa, 0L);
        hipLaunchKernelGGL(k_x2_split_rows<false>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, x, ldx, (long)M, L.rows_alloc, K, scales + 4, (void *)pb,
                           L.rows_alloc * (long)K);
    }
    X3Args a{};
    a.logits = pa; a.hidden = pb; a.plane_stride = L.rows_alloc * (long)K; a.rows_pad = L.rows_pad; a.rows_alloc = L.rows_alloc;
    a.B = 1; a.T = 1; a.U1 = 1; a.H = K; a.V = N; a.n_split = L.n_split; a.dw_tab = tab; a.dw_prog = (int *)(w + L.prog);
    a.slab_w = (float *)(w + L.slab_w); a.slab_b = (float *)(w + L.slab_b); a.dw_rescale = 1.0f; a.db_rescale = 1.0f; a.n_cu = lin_cus();
    launch_dw_x2(a, st, false, false);  // (no table kernel, no progress-word fill: both done by k_x2_lin_scales)
    const long n4w = (long)N * K / 4, n4b = N / 4;
    hipLaunchKernelGGL(k_x2_reduce_scaled, dim3((unsigned)((n4w + 255) / 256)), dim3(256), 0, st, a.slab_w, dW, n4w, n4w, L.n_split,
                       (const float *)(scales + 3), (const float *)(scales + 5));
    if (db)
        hipLaunchKernelGGL(k_x2_reduce_scaled, dim3((unsigned)((n4b + 255) / 256)), dim3(256), 0, st, a.slab_b, db, n4b, n4b, L.n_split,
                           (const float *)(scales + 3), (const float *)nullptr);
}
