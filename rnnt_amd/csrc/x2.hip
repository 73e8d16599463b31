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

// ---------------------------------------------------------------------------------------
// s_W = 2^(14 - ceil(log2 max|W|)) (1 for an all-zero or non-finite W): one workgroup, V*H/4 float4 reads.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_x2_wscale(const float *__restrict__ W, long n4, float *__restrict__ scales)
{
    __shared__ float s_m[16];
    float m = 0.f;
    for (long i = threadIdx.x; i < n4; i += 1024) {
        const f32x4 w = ((const f32x4 *)W)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(w[0]), fabsf(w[1]))), fmaxf(fabsf(w[2]), fabsf(w[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
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
        scales[1] = 1.0f / s;
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
    for (int k = 0; k < 4; ++k) { t0[k] = x2_clamp(t0[k] * X2_SH); t1[k] = x2_clamp(t1[k] * X2_SH); }  // (NaN operands stay NaN, as on every route)
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
//  * HBM -> LDS by LDS-DMA, ring of 3 stages of 16 cells x (256 v + 256 h) x 2 planes = 32 KiB; wave w fills operand tile w
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
#define XW2_NST 3
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

__global__ __launch_bounds__(256, 1) void k_dw_x2(X3Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char s_ring[];
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
    const long *tab = a.dw_tab;
    const int B = a.B;
    const long nlive = tab[2 * B + 1];
    const long g_lo = nlive * split / a.n_split, g_hi = nlive * (split + 1) / a.n_split;

    f32x16 acc[4][4];
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;
    // db rides the matrix pipe as in k_dw_x3: one more MFMA per plane and k-step against a column SELECTOR of ones, for one
    // (one h block: two) of the wave's M tiles, into a 17th accumulator tile held in VGPRs
    const bool do_b = hb < 2;  // workgroup-uniform
    const int bsel0 = n_hblk >= 2 ? (hb & 1) * 2 + wn : wn;
    f32x16 dacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
    const unsigned sel0 = (lane & 31) == 0 ? 0x3c003c00u : 0u, sel1 = (lane & 31) == 1 ? 0x3c003c00u : 0u;  // fp16 ones

    if (g_hi > g_lo) {
        // ---- DMA source of this wave's operand tile
        const bool is_g = wave < 2;
        int col0 = (is_g ? vb : hb) * 256 + 128 * (wave & 1);
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
            soff[i] = (int)((4 * i + (lane >> 4)) * rstride) + cb;
        }
        long row_first = 0;  // first cell of the range being walked
        // ---- transposed fragment reads.  Fragment of 32-column tile m: lane (g = lane>>4, q = (lane&15)>>2,
        // p = lane&3) reads rows 8(g>>1) + 4sec + q at chunk 4m + 2(g&1) + (p>>1), +8(p&1) bytes, sec = 0,1.
        const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3, hh = g >> 1;
        const int lds0 = (int)(size_t)(lds_vptr)s_ring;
        int abase[4][2], bbase[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int sec = 0; sec < 2; ++sec) {
                const int row = 8 * hh + 4 * sec + q;
                const int ch = 4 * m + 2 * (g & 1) + (pp >> 1);
                const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
                const int fo = 256 * row + 16 * (ch ^ swz) + 8 * (pp & 1);
                abase[m][sec] = lds0 + wm * XW2_TILE + fo;
                bbase[m][sec] = lds0 + (2 + wn) * XW2_TILE + fo;
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
            constexpr int ST = decltype(st_c)::value, DST = (ST + 2) % 3;
            // the 16 rows of stage ks+2 as two raw buffers (wave-uniform base; the per-lane part is the 32-bit soff)
            __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
                rs[p] = __builtin_amdgcn_make_buffer_rsrc((void *)(pbase[p] + (row_first + (ks + 2) * XW2_ROWS) * rstride), 0,
                                                          (int)(XW2_ROWS * rstride), 0x00020000);
            auto dma_piece = [&](auto n_c) {  // piece n of stage ks+2 -> ring stage DST: plane n>>2, rows 4(n&3)..
                constexpr int n = decltype(n_c)::value, p = n >> 2, i = n & 3;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[p], (lds_vptr)(s_ring + wave * XW2_TILE + DST * XW2_STAGE + p * XW2_PLANE + 1024 * i),
                                                         16, soff[i], 0, 0, 0);
            };
            // 8 transposed reads of plane P of the A / B operand (inline asm: hipcc guards every LDS read it can see behind
            // an LDS-DMA with vmcnt(0)); results are used only after X2_LANDED named them
            auto reads = [&](X2Frag &f, const int (&base)[4][2], auto p_c) {
                constexpr int off = ST * XW2_STAGE + decltype(p_c)::value * XW2_PLANE;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo[m]) : "v"(base[m][0]), "n"(off));
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi[m]) : "v"(base[m][1]), "n"(off));
                }
            };
            // 16 MFMAs of one product with DMA pieces N0 .. N0+CNT-1 threaded through them
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
                    if (qm == 0) dma_piece(X2Int<N0>{});
                    if (qm == 1) dma_piece(X2Int<N0 + 1>{});
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
            // stage ks landed (the 8 younger pieces of ks+1 may still fly); every wave is past its reads of stage ks-1,
            // whose ring stage the DMAs below refill
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            x2_lds_barrier();
            X2Frag Ah, Bh, Am, Bm;
            u32x2 dl[2], dh[2];
            reads(Ah, abase, X2Int<0>{});
            reads(Bh, bbase, X2Int<0>{});
            reads(Am, abase, X2Int<1>{});
            X2_LANDED(Ah, 8);
            X2_LANDED(Bh, 8);
            product(Ah, Bh, X2Int<0>{}, X2Int<3>{});
            reads(Bm, bbase, X2Int<1>{});
            X2_LANDED(Am, 8);
            product(Am, Bh, X2Int<3>{}, X2Int<3>{});
            X2_LANDED(Bm, 0);
            if (do_b) { bias_read(dl[0], dh[0], X2Int<0>{}, 0); bias_read(dl[1], dh[1], X2Int<1>{}, 0); }
            product(Ah, Bm, X2Int<6>{}, X2Int<2>{});
            if (do_b) {
                bias_mfma(dl[0], dh[0], dacc, 0); bias_mfma(dl[1], dh[1], dacc, 0);
                if (n_hblk < 2) {  // one h block: the wave's second tile (column 1 of the selector product), H <= 256 only
                    bias_read(dl[0], dh[0], X2Int<0>{}, 1); bias_read(dl[1], dh[1], X2Int<1>{}, 1);
                    bias_mfma(dl[0], dh[0], dacc, 1); bias_mfma(dl[1], dh[1], dacc, 1);
                }
            }
        };
        auto dma_stage = [&](long ks, int st) {  // pipeline prologue: all 8 pieces of stage ks
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const int p = n >> 2, i = n & 3;
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(pbase[p] + (row_first + ks * XW2_ROWS) * rstride), 0, (int)(XW2_ROWS * rstride), 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_vptr)(s_ring + wave * XW2_TILE + st * XW2_STAGE + p * XW2_PLANE + 1024 * i),
                                                         16, soff[i], 0, 0, 0);
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
            for (long ks = 0;;) {  // the ring stage of a k-step is ks % 3: unrolled by 3
                if (ks >= nks) break;
                kstep(X2Int<0>{}, ks, dacc); ++ks;
                if (ks >= nks) break;
                kstep(X2Int<1>{}, ks, dacc); ++ks;
                if (ks >= nks) break;
                kstep(X2Int<2>{}, ks, dacc); ++ks;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the over-issued DMAs before the ring
            x2_lds_barrier();                                  // is refilled / the kernel exits
        }
    }

    // ---- epilogue: partial slab [split][V,H]; bias partial [split][V].  Accumulator register r of tile
    // (qm,qn): v = v0 + 32qm + (r&3) + 8(r>>2) + 4half, h = h0 + 32qn + (lane&31).
    const int v0 = vb * 256 + wm * 128, h0 = hb * 256 + wn * 128;
    float *sw = a.slab_w + (long)split * V * H;
    const float rw = a.dw_rescale, rb = a.db_rescale;
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + 32 * qm + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (v < V) {
#pragma unroll
                for (int qn = 0; qn < 4; ++qn) {
                    const int h = h0 + 32 * qn + (lane & 31);
                    if (h < H) sw[(long)v * H + h] = acc[qm][qn][r] * rw;
                }
            }
        }
    if (do_b && (lane & 31) < (n_hblk >= 2 ? 1 : 2)) {  // column k of the selector products: lanes k / 32+k store
        const int m = bsel0 + 2 * (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = v0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (v < V) a.slab_b[(long)split * V + v] = dacc[r] * rb;
        }
    }
}

void launch_dw_x2(const X3Args &a, hipStream_t st)
{
    launch_dw_table(a.logit_lens, a.B, a.T, a.U1, XW2_GRAN, a.dw_tab, st);
    const int tiles = ((a.V + 255) / 256) * ((a.H + 255) / 256);
    static bool attr_set[16] = {false};  // > 64 KiB of dynamic LDS: opt-in once per device (read-mostly fact)
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
    if (dev < 0 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void *)k_dw_x2, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * XW2_TILE);
        if (dev >= 0) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(k_dw_x2, dim3(tiles * a.n_split), dim3(256), 4 * XW2_TILE, st, a);
}

// (to come: the fused forward and dHidden kernels; until then those stages run on the fp32 route's kernels + the plain producers)
bool x2_fwd_ok(int, int, int) { return false; }
bool x2_dhidden_ok(int, int, int) { return false; }
size_t x2_wpack_fwd_bytes(int H, int V) { return (size_t)((V + 511) / 512) * (H / 16) * 2 * 16 * 64 * 16; }
size_t x2_wpack_dh_bytes(int H, int V) { return (size_t)((H + 511) / 512) * (V / 16) * 2 * 16 * 64 * 16; }
void launch_x2_pack_w(const X3Args &a, float *scales, hipStream_t st)
{
    hipLaunchKernelGGL(k_x2_wscale, dim3(1), dim3(1024), 0, st, a.W, (long)a.V * a.H / 4, scales);
}
void launch_joint_fwd_x2(const X3Args &, hipStream_t) {}
void launch_dhidden_x2(const X3Args &, hipStream_t) {}
