// joint_bwd.hip — backward GEMMs of the fused joint + transducer loss for gfx950.
//
// Replaces what loss.backward() (reference rnnt/train.py:134) runs for the path
// rnnt/model.py:32-41: the loss gradient w.r.t. logits (torchaudio),
// joint_ln's backward and the tanh / broadcast-add backward of rnnt/joint.py:32-39.
//
// The gradient w.r.t. logits, G, is produced ONCE from the materialised logits and the 16-byte
// per-cell CellCoef, in place of the logits (k_dhidden_gen does it inside the dHidden GEMM; k_make_g
// on the fallback route):
//      G[c,v] = exp2(logit[c,v]*log2e + c1[c]) - (v==blank)*sb[c] - (v==y[c])*se[c]
//
//  k_dhidden_gen  dHidden[c, 0:512] = G[c,:] @ W[:, 0:512]   (M = cells, K = V, N = H), G stored in
//             place; dPre = dHidden * (1 - tanh^2) reduced in the epilogue over the tile's u
//             (-> dEnc partial slab) and over its t (-> dPred partial slab); deterministic
//             partial slabs, summed by k_reduce_*.  Tiles 8 t x 16 u or 16 t x 8 u (short targets).
//  k_dhidden  the same product for the columns of H beyond 512 (the reference's joint is 1024 wide)
//             and for shapes k_dhidden_gen does not take: persistent independent waves.
//  k_dw       dW[v,h] = sum_c G[c,v] * hidden[c,h]   (M = V, N = H, K = cells, split-K); hidden is
//             the tensor the forward kernel stored; db is the in-lane row sum of the same G
//             fragments.  K walks the live cells only (contiguous ranges, or a device-built list of
//             live 16-row granules on ragged batches).
//
// Like the forward kernel these feed v_mfma_f32_32x32x2_f32 straight from registers
// (one VGPR per operand, 64 matrix-pipe cycles per instruction): no LDS staging in the
// main loops.  A lane's 16-byte load supplies either 4 k-steps (k contiguous in memory:
// G for the dHidden kernels) or 4 interleaved tiles (m/n contiguous in memory: W rows, G and
// hidden for k_dw), so every global access is a full 16 B per lane.
#include "common.hpp"
#include "kernels.hpp"

typedef __attribute__((address_space(3))) void *lds_void_ptr;

#define DH_BT 8
#define DW_KC 16   // cells per dW split-range granule (row padding unit)
#define DH_BU 16

// dHidden GEMM: dHidden[c,:] = G[c,:] @ W   (K = V), then dPre = dHidden * (1 - hidden^2)
// reduced over the item's 16 u (-> dEnc partial slab) and its 4 t (-> dPred partial slab).
//
// Persistent, fully independent waves.  Measured facts behind this shape: the
// SIMD arbitrates oldest-first, so two co-resident waves do not interleave — the older runs
// at its own pace and the younger fills its bubbles; any barrier / end-of-workgroup join then
// leaves a wave slot idle while its partner finishes alone.  So here NOTHING joins waves:
//   * work item = 64 cells (4 t x 16 u of one utterance) x 128 columns of H: 8 accumulator
//     tiles (128 registers) -> two waves per SIMD fit, and one wave's epilogue (hidden loads,
//     reductions, slab stores) overlaps its SIMD partner's MFMA stream;
//   * each wave pulls items from a global atomic counter until none are left (perfect balance,
//     no workgroup tail); items are ordered column-block fastest, so the waves that share a G
//     tile run at about the same time and meet in L2;
//   * operands go straight HBM/L2 -> VGPR (register-destination loads cost the matrix pipe
//     nothing to issue; an LDS-DMA piece costs ~60 cycles), two 8-wide k chunks ahead, every
//     load unconditional; no LDS, no barrier, no VALU in the main loop.
// The item's reductions are in-wave: dEnc sums a lane's 8 rows + one cross-half shuffle, dPred
// sums over the 4 (M-tile, row-half) groups in-lane.
#define PW_BT 4
struct PwChunk {
    f32x4 x[2];  // G slices (4 consecutive k) of the two M tiles
    f32x4 w[4];  // W rows k0+4*half+s, columns n0+4j..4j+3
};

// `cb0`: first 128-column block this launch covers (0, or 4 when k_dhidden_gen has produced G and
// the first 512 columns: H > 512).
// BU: u width of an item (16: 4 t x 16 u; 8: 8 t x 8 u for short targets, as k_dhidden_gen<8>).
template <int BU>
__global__ __launch_bounds__(512, 2) void k_dhidden(JointBwdArgs a, int cb0)
{
    constexpr int IT = 64 / BU;  // t rows per item
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int n_cb = (H + 127) / 128 - cb0, n_ub = (U1 + BU - 1) / BU, n_tt = (T + IT - 1) / IT;
    const long zero_row = (long)a.B * T * U1;  // first padding row: G == 0
    // V % 8 == 4: in the last chunk lanes 32-63 would start at k >= V; they step back 4
    // (valid addresses) and their G values are zeroed
    const long BTH = (long)a.B * T * H, BUH = (long)a.B * U1 * H;

    // Work queues: one counter per XCD.  A tile's n_cb column-block items all read the same
    // 64 G rows, so they are queued back to back on ONE XCD (tile id mod 8) and the three
    // re-reads hit that XCD's L2 (one global queue spread them over all XCDs: 105 GB fetched
    // for 26 GB of G).  HW_REG_XCC_ID only decides which queue a wave serves: any placement is
    // correct, an uneven one is merely slower.
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const long ntile = (long)a.B * n_tt * n_ub;
    // serve the own XCD's queue first, then help the others (work stealing): every item is
    // processed exactly once whatever the placement, even if some XCD hosts no wave at all
    for (int dq = 0; dq < 8; ++dq) {
      const unsigned q = (xcc + dq) & 7u;
      const long q_items = ((ntile - q + 7) / 8) * n_cb;  // tiles q, q+8, ... x column blocks
      for (;;) {
        long item = 0;
        if (lane == 0) item = (long)atomicAdd(a.counter + q * 16, 1u);
        item = __builtin_amdgcn_readfirstlane((int)item);
        if (item >= q_items) break;
        const int cb = cb0 + (int)(item % n_cb);
        long r_ = (item / n_cb) * 8 + q;  // tile id
        const int ub = (int)(r_ % n_ub); r_ /= n_ub;
        const int tt = (int)(r_ % n_tt);
        const int b = (int)(r_ / n_tt);
        const int Tb = len_t(a.logit_lens, b, a.T);
        const int t0 = tt * IT, u0 = ub * BU;
        if (t0 >= Tb) continue;  // wave-uniform
        // a u block past U_b holds no lattice cell (G == 0 there): its slabs are never read
        // (k_reduce_enc / k_reduce_pred stop at U_b)
        if (u0 > len_u(a.target_lens, b, a.U1)) continue;
        const int col = cb * 128 + 4 * i;
        const bool colok = col < H;

        f32x16 acc[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;

        const float *gptr[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int row = mt * 32 + i;
            const int t = t0 + row / BU, u = u0 + row % BU;
            const long cell = (t < T && u < U1) ? ((long)b * T + t) * U1 + u : zero_row;
            gptr[mt] = a.logits + cell * V + 4 * half;
        }
        // columns beyond H (last column block) read column 0: they feed unstored accumulators
        const float *wptr = a.W + (long)(4 * half) * H + (colok ? col : 0);
        // Main loop: 32 MFMAs per 8-wide k chunk, operands two chunks ahead in two register sets
        // (copy, refill, then compute: refilling a set right behind its own MFMAs, or rotating
        // three sets, measured slower / spilled).  The k order of a dot product is free: every
        // item starts at its own chunk (`rot`) so the waves do not walk W and the 4 KiB-strided
        // G rows in step.  V % 8 == 4: in the last chunk lanes 32-63 step back 4 (valid
        // addresses) and their G values are zeroed.
        const int VK = (V + 7) / 8, last = VK - 1;
        const bool kill = ((V & 7) != 0) && half == 1;
        const int back = kill ? 4 : 0;
        const int rot = (int)((((item / n_cb) * 8 + q) * 37) % VK);  // per TILE: its column-block
        // items then read the same G chunk at about the same time and meet in the XCD's L2
        auto chunk_of = [&](int c8) { int cc = c8 + rot; cc -= cc >= VK ? VK : 0; cc -= cc >= VK ? VK : 0; return cc; };
        auto load = [&](PwChunk &c, int c8) {
            const int cc = chunk_of(c8);
            const int k0 = 8 * cc - (cc == last ? back : 0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) c.x[mt] = *(const f32x4 *)(gptr[mt] + k0);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) c.w[s_] = *(const f32x4 *)(wptr + (long)(k0 + s_) * H);
        };
        auto compute = [&](const PwChunk &c, bool zero_hi) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float gv = zero_hi ? 0.f : c.x[mt][s_];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[mt][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, c.w[s_][q], acc[mt][q],
                                                                          0, 0, 0);
                }
        };
        const bool xp_noepi = RNNT_XP(a.flags, 4) != 0;  // experiment switch
        PwChunk r0, r1;
        load(r0, 0);
        load(r1, 1);
        for (int c8 = 0; c8 < VK; c8 += 2) {
            {
                const PwChunk cur = r0;
                load(r0, c8 + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(cur, kill && chunk_of(c8) == last);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (c8 + 1 < VK) {
                const PwChunk cur = r1;
                load(r1, c8 + 3);
                __builtin_amdgcn_sched_barrier(0);
                compute(cur, kill && chunk_of(c8 + 1) == last);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (xp_noepi) {
            if (acc[0][0][0] == 12345.f) a.slab_enc[0] = acc[1][3][7];
            continue;
        }
        // ---- epilogue (in-wave): dPre = dHidden * (1 - hidden^2); sum over u and over t.
        // Register 8 rh + r7 of M tile mt: BU = 16: t row 2 mt + rh, u slot 8 (r7>>2) + (r7&3) + 4 half;
        // BU = 8: t row 4 mt + 2 rh + (r7>>2), u slot (r7&3) + 4 half.
        constexpr int NU = BU / 2, TB = BU == 16 ? 1 : 2;
        float psum[NU][4];
#pragma unroll
        for (int k = 0; k < NU; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) psum[k][q] = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rh = 0; rh < 2; ++rh) {
                float esum[TB][4];
#pragma unroll
                for (int e = 0; e < TB; ++e)
#pragma unroll
                    for (int q = 0; q < 4; ++q) esum[e][q] = 0.f;
#pragma unroll
                for (int r7 = 0; r7 < 8; ++r7) {
                    const int t = t0 + (BU == 16 ? mt * 2 + rh : mt * 4 + 2 * rh + (r7 >> 2));
                    const int u = u0 + (BU == 16 ? 8 * (r7 >> 2) + (r7 & 3) : (r7 & 3)) + 4 * half;
                    const bool ok = t < Tb && u < U1 && colok;
                    f32x4 h4 = {0.f, 0.f, 0.f, 0.f};
                    if (ok) h4 = *(const f32x4 *)(a.hidden + (((long)b * T + t) * U1 + u) * H + col);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float d = ok ? acc[mt][q][rh * 8 + r7] * (1.f - h4[q] * h4[q]) : 0.f;
                        esum[BU == 16 ? 0 : (r7 >> 2)][q] += d;
                        psum[BU == 16 ? r7 : (r7 & 3)][q] += d;
                    }
                }
#pragma unroll
                for (int e = 0; e < TB; ++e) {
                    const int t = t0 + (BU == 16 ? mt * 2 + rh : mt * 4 + 2 * rh + e);
#pragma unroll
                    for (int q = 0; q < 4; ++q) esum[e][q] += __shfl_xor(esum[e][q], 32, 64);
                    if (half == 0 && t < Tb && colok) {
                        f32x4 o = {esum[e][0], esum[e][1], esum[e][2], esum[e][3]};
                        *(f32x4 *)(a.slab_enc + (long)ub * BTH + ((long)b * T + t) * H + col) = o;
                    }
                }
            }
        if (colok) {
#pragma unroll
            for (int k = 0; k < NU; ++k) {
                const int u = u0 + (BU == 16 ? 8 * (k >> 2) + (k & 3) : k) + 4 * half;
                if (u < U1) {
                    f32x4 o = {psum[k][0], psum[k][1], psum[k][2], psum[k][3]};
                    *(f32x4 *)(a.slab_pred + (long)tt * BUH + ((long)b * U1 + u) * H + col) = o;
                }
            }
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// k_dhidden_gen — dHidden GEMM that also PRODUCES G (replaces k_make_g, and k_dhidden for the first
// 512 columns of H — all of them when H <= 512; needs V % 32 == 0).
//
// k_make_g costs ~10 ms of pure HBM traffic (read 26 GB of logits, write 26 GB of G).  Here the
// workgroup that owns a 128-cell tile (8 t x 16 u) reads the tile's logits once, turns them into
// G in registers — exactly one thread per (cell, 4 consecutive k): wave w produces the MFMA
// A-fragment of M-tile w, in fragment order — hands the fragments to the other waves through a
// double-buffered 4 KiB LDS exchange (one s_barrier per 8-wide k chunk = per 64 MFMAs per
// wave) and stores G back over the logits for k_dw.  Each logits element is read and
// overwritten by the same thread, so the in-place update needs no ordering.
// 4 waves (one per SIMD) = 2 (M) x 2 (N), wave tile 64 cells x 256 columns, 256 AGPR
// accumulators; W fragments straight L2 -> VGPR one chunk ahead; logits three chunks ahead.
// Tiles with t0 >= T_b (ragged batches) only zero their G rows.  grid (n_ublk, ceil(T/8), B).
#define DG_BT 8
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#ifdef RNNT_STAMPS
// Diagnostic build only (make EXTRA=-DRNNT_STAMPS): s_memtime stamps of every 16th workgroup.
#define GSTAMP(slot)                                                                          \
    do {                                                                                      \
        const unsigned wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;  \
        if (a.debug && threadIdx.x == 0 && (wg_ & 15) == 0 && wg_ < 16 * 4096) {              \
            unsigned long long t_;                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            a.debug[(wg_ >> 4) * 8 + (slot)] = t_;                                            \
        }                                                                                     \
    } while (0)
#else
#define GSTAMP(slot) do {} while (0)
#endif
// BU = u width of the tile (16: 8 t x 16 u; 8: 16 t x 8 u, for short targets — U1 = 101 fills 7 blocks
// of 16 with 10 % dead rows, 13 blocks of 8 with 3 %); an M tile of 32 rows is (32/BU) t rows of BU u.
// GEN = false (H >= 1024: further whole groups of 512 columns, `col_base` = 512, 1024, ...): the same tile,
// schedule and epilogue on the G the GEN launch left in place of the logits — loaded, handed through the
// exchange and multiplied as is; nothing is produced or stored but the slabs.
template <int BU, bool GEN>
__global__ __launch_bounds__(256, 1) void k_dhidden_gen(JointBwdArgs a, const int col_base)
{
    constexpr int BT = 128 / BU;  // t rows per tile
    __shared__ __attribute__((aligned(16))) float smem[4 * 4 * 256 + 2 * 64 * 65];  // 4-slot G exchange + epilogue
    float(*s_red)[64][65] = (float(*)[64][65])(smem + 4 * 4 * 256);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    const int T = a.T, U1 = a.U1, H = a.H, V = a.V;
    const int ub = blockIdx.x, tt = blockIdx.y, b = blockIdx.z;
    int Tb, Ub;  // both lengths by one pair of scalar loads (common.hpp)
    len_tu_uniform(a.logit_lens, a.target_lens, b, a.T, a.U1, Tb, Ub);
    const int t0 = tt * BT, u0 = ub * BU;
    const int VK = V / 8;

    // ---- this lane's producer row (M-tile `wave`, row i): one lattice cell or none
    const int prow = wave * 32 + i;
    const int pt = t0 + prow / BU, pu = u0 + prow % BU;
    const bool pexists = pt < T && pu < U1;
    const long pcell = pexists ? ((long)b * T + pt) * U1 + pu : (long)a.B * T * U1;  // else zero row
    float *lptr = (float *)a.logits + pcell * V + 4 * half;

    if (!GEN && (t0 >= Tb || u0 > Ub)) return;  // dead tile: no slab is read
    if (t0 >= Tb) {  // workgroup-uniform: nothing to multiply, but k_dw must find zeros here
        // k_dw only walks the live rows, rounded out to 16-cell granules (k_dw_table): up to 15
        // cells past this utterance's live end, and up to 15 cells before the next utterance's
        // first cell (= the last cells of this one).  Tiles that touch neither stay unwritten.
        if ((long)t0 * U1 >= (long)Tb * U1 + DW_KC && (long)(t0 + BT) * U1 <= (long)T * U1 - DW_KC) return;
        if (pexists) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (int c8 = 0; c8 < VK; ++c8) *(f32x4 *)(lptr + 8 * c8) = z;
        }
        return;
    }
    if (u0 > Ub) {  // workgroup-uniform: a u block past U_b holds no lattice cell — its G
        if (pexists) {  // rows are zeros (k_dw walks them), its slabs are never read
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            for (int c8 = 0; c8 < VK; ++c8) *(f32x4 *)(lptr + 8 * c8) = z;
        }
        return;
    }
    GSTAMP(0);

    CellCoef cf;
    cf.c1 = 0.f; cf.sb = 0.f; cf.se = 0.f; cf.y = -1;
    if (GEN) cf = a.coef[pexists ? pcell : 0];
    // GEN = false: the G rows of this tile's cells outside the lattice are exact zeros already
    const bool live = GEN ? (pexists && pt < Tb && cf.c1 != RNNT_NEG_INF) : pexists;
    const int ncol0 = col_base + wn * 256;
    const int colg[2] = {ncol0 + 4 * i, ncol0 + 128 + 4 * i};
    const bool colok[2] = {colg[0] < H, colg[1] < H};
    // columns beyond H read column 0: they feed accumulators that are never stored
    const float *wptr[2] = {a.W + (long)(4 * half) * H + (colok[0] ? colg[0] : 0),
                            a.W + (long)(4 * half) * H + (colok[1] ? colg[1] : 0)};
    const unsigned wrow_bytes = (unsigned)H * 4u;
    // W as a raw buffer (V*H*4 bytes < 4 GiB is checked by the engine)
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)a.W, 0, (int)((unsigned)V * wrow_bytes), 0x00020000);
    // The tile's logits rows as a raw buffer (8 t x 16 u cells span < (7 U1 + 16) rows): the
    // per-lane row offset is a 32-bit VGPR, the chunk offset a scalar, and lanes whose row is
    // outside the lattice point past the end — their loads return 0 (with c1 = -inf: G = 0) and
    // their stores are dropped, so neither needs a branch or an exec mask.
    const long cell0 = ((long)b * T + t0) * U1 + u0;
    const long rows_left = a.rows_pad + 16 - cell0;  // rows of the allocation from cell0 on
    const long span_rows = (long)(BT - 1) * U1 + BU < rows_left ? (long)(BT - 1) * U1 + BU : rows_left;
    const __amdgpu_buffer_rsrc_t lrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.logits + cell0 * V), 0, (int)(span_rows * V * 4), 0x00020000);
    const unsigned lane_row_off = (unsigned)((((prow / BU) * U1 + (prow % BU)) * (long)V + 4 * half) * 4);
    const unsigned xvoff = live ? lane_row_off : 0xfffffff0u;    // loads: rows of the lattice only
    const unsigned svoff = pexists ? lane_row_off : 0xfffffff0u;  // stores: every existing cell
    const unsigned woff[2] = {(unsigned)(((4 * half) * H + (colok[0] ? colg[0] : 0)) * 4),
                              (unsigned)(((4 * half) * H + (colok[1] ? colg[1] : 0)) * 4)};

    f32x16 acc[2][8];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][q][r] = 0.f;

    // G of chunk c8 for this lane's (row, 4 k): exp2(x*log2e + c1) - fixups.  Rows outside the
    // lattice read the zero padding row with c1 = -inf (exp2 -> 0): no per-element select.  The
    // two fixups touch one element of one chunk per row, so they sit behind wave-uniform tests.
    if (!live) { cf.c1 = RNNT_NEG_INF; cf.sb = 0.f; cf.se = 0.f; cf.y = -1; }
    auto gen = [&](const f32x4 &x, int c8) {
        if (!GEN) return x;
        f32x4 g;
        const int vb = 8 * c8 + 4 * half;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) g[s_] = __builtin_amdgcn_exp2f(fmaf(x[s_], RNNT_LOG2E, cf.c1));
        const unsigned dy = (unsigned)(cf.y - vb);
        if (__any(dy < 4u)) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                if (dy == (unsigned)s_) g[s_] -= cf.se;
        }
        if (8 * c8 <= a.blank && a.blank < 8 * c8 + 8) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
                if (vb + s_ == a.blank) g[s_] -= cf.sb;
        }
        return g;
    };
    auto wload = [&](f32x4 (&w)[4][2], int c8) {
        const int cc = c8 < VK ? c8 : VK - 1;  // the last refill re-reads the final chunk (unused)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int g = 0; g < 2; ++g) w[s_][g] = *(const f32x4 *)(wptr[g] + (long)(8 * cc + s_) * H);
    };
    auto xload = [&](int c8) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(lrsrc, xvoff, 32 * (c8 < VK ? c8 : VK - 1), 0));
    };
    auto gstore = [&](const f32x4 &g, int c8) {
        if (GEN) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, g), lrsrc, svoff, 32 * c8, 0);
    };

    f32x4 xr[4];           // raw logits of chunks c+3 .. c+6 (ring, slot = chunk & 3)
    f32x4 wf[2][4][2];     // W fragments of chunks c, c+1 (slot = chunk & 1)
    xr[0] = xload(0); xr[1] = xload(1); xr[2] = xload(2); xr[3] = xload(3);
    wload(wf[0], 0);
    wload(wf[1], 1);
    // G is produced THREE chunks ahead of its MFMAs (4-slot LDS exchange, slot = chunk & 3) and
    // read one chunk ahead, inside the previous chunk's MFMA stream: no LDS latency stands at the
    // top of a chunk, and the workgroup only needs a barrier every SECOND chunk (after the even
    // ones): between the write of chunk k's fragments (during chunk k-3) and their read (during
    // chunk k-1), and between the last read of a slot (chunk k-5) and its rewrite (chunk k-3),
    // one of the two chunk ends in between is a barrier.  Half the barriers = half the time the
    // four waves spend waiting for the slowest of them.
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const f32x4 gk = gen(xr[k], k);
        *(f32x4 *)(smem + k * 1024 + wave * 256 + 4 * lane) = gk;
        gstore(gk, k);
        xr[k] = xload(4 + k);
    }
    __syncthreads();
    GSTAMP(1);
    f32x4 a0 = *(const f32x4 *)(smem + (2 * wm) * 256 + 4 * lane);       // fragments of chunk 0
    f32x4 a1 = *(const f32x4 *)(smem + (2 * wm + 1) * 256 + 4 * lane);
    // One MFMA of the chunk: number m (0..15) of k-step s_.  PIN() keeps what the source puts
    // between two MFMAs there: the matrix pipe runs an MFMA for 64 cycles while issuing it takes
    // a few, so the chunk's other work (producing the next chunk's G, copying / refilling the W
    // ring, LDS and memory traffic) rides in those gaps instead of standing in front of the 64
    // MFMAs (where it cost ~450 of every ~4800 cycles).
#define PIN() __builtin_amdgcn_sched_barrier(0)
    const int blank_chunk = a.blank >> 3;
    for (int c0 = 0; c0 < VK; c0 += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c8 = c0 + j;  // VK % 4 == 0
            f32x4 cur[4][2];
            cur[0][0] = wf[j & 1][0][0];
            cur[0][1] = wf[j & 1][0][1];
            PIN();
            auto mf = [&](int s_, int m) {
                const int mt = m >> 3, g = (m >> 2) & 1, q = m & 3;
                const float gv = mt == 0 ? a0[s_] : a1[s_];
                acc[mt][g * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv, cur[s_][g][q], acc[mt][g * 4 + q], 0, 0, 0);
            };
            // ---- k-step 0, with the production of chunk c8+3 (logits requested 4 chunks ago)
            const bool produce = c8 + 3 < VK;  // workgroup-uniform
            const int cn = c8 + 3, vbn = 8 * cn + 4 * half;
            f32x4 gn;
            {
                const f32x4 &x = xr[(j + 3) & 3];
                mf(0, 0); cur[1][0] = wf[j & 1][1][0]; PIN();
                mf(0, 1); cur[1][1] = wf[j & 1][1][1]; PIN();
                mf(0, 2);
                if (GEN) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gn[e] = fmaf(x[e], RNNT_LOG2E, cf.c1); asm volatile("" : "+v"(gn[e])); }
                } else {
                    gn = x;
                }
                PIN();
                mf(0, 3);
                if (GEN) {
                    gn[0] = __builtin_amdgcn_exp2f(gn[0]); gn[1] = __builtin_amdgcn_exp2f(gn[1]);
                    asm volatile("" : "+v"(gn[0]), "+v"(gn[1]));
                }
                PIN();
                mf(0, 4);
                if (GEN) {
                    gn[2] = __builtin_amdgcn_exp2f(gn[2]); gn[3] = __builtin_amdgcn_exp2f(gn[3]);
                    asm volatile("" : "+v"(gn[2]), "+v"(gn[3]));
                }
                PIN();
                mf(0, 5);
                if (GEN) {
                    const unsigned dy = (unsigned)(cf.y - vbn);
                    if (__any(dy < 4u)) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (dy == (unsigned)e) gn[e] -= cf.se;
                    }
                }
                PIN();
                mf(0, 6);
                if (GEN && cn == blank_chunk) {  // wave-uniform, one chunk in V/8
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (vbn + e == a.blank) gn[e] -= cf.sb;
                }
                PIN();
                mf(0, 7);
                if (produce) {
                    *(f32x4 *)(smem + ((j + 3) & 3) * 1024 + wave * 256 + 4 * lane) = gn;
                    gstore(gn, cn);
                }
                PIN();
                mf(0, 8);
                xr[(j + 3) & 3] = xload(c8 + 7);
                PIN();
#pragma unroll
                for (int m = 9; m < 16; ++m) mf(0, m);
                PIN();
            }
            // ---- k-step 1, with the rest of the W ring copies
            mf(1, 0); cur[2][0] = wf[j & 1][2][0]; PIN();
            mf(1, 1); cur[2][1] = wf[j & 1][2][1]; PIN();
            mf(1, 2); cur[3][0] = wf[j & 1][3][0]; PIN();
            mf(1, 3); cur[3][1] = wf[j & 1][3][1]; PIN();
            mf(1, 4);
            // fragments of chunk c8+1 (written during chunk c8-2; a barrier has passed since)
            const f32x4 an0 = *(const f32x4 *)(smem + ((j + 1) & 3) * 1024 + (2 * wm) * 256 + 4 * lane);
            const f32x4 an1 = *(const f32x4 *)(smem + ((j + 1) & 3) * 1024 + (2 * wm + 1) * 256 + 4 * lane);
            PIN();
#pragma unroll
            for (int m = 5; m < 16; ++m) mf(1, m);
            PIN();
            // ---- k-steps 2, 3; the ring slot is copied out: refill it with chunk c8+2, ONE load
            // per MFMA gap (a gap hides ~56 issue cycles; 8 loads + their address arithmetic in one
            // gap overflowed it)
            {
                // buffer loads: descriptor + uniform row offset in SGPRs, 32-bit per-lane offset —
                // no vector address arithmetic (every instruction between MFMAs costs ~6
                // matrix-pipe cycles)
                const int cc = c8 + 2 < VK ? c8 + 2 : VK - 1;
                const unsigned rowb = (unsigned)(8 * cc) * wrow_bytes;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    mf(2, m);
                    wf[j & 1][m >> 1][m & 1] = __builtin_bit_cast(
                        f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff[m & 1], rowb + (m >> 1) * wrow_bytes, 0));
                    PIN();
                }
            }
#pragma unroll
            for (int m = 8; m < 16; ++m) mf(2, m);
#pragma unroll
            for (int m = 0; m < 16; ++m) mf(3, m);
            PIN();
            // publish chunk c8+1's fragments / free chunk c8's buffer.  Raw barrier: a
            // __syncthreads() would add s_waitcnt vmcnt(0) and drain the logits / W prefetch.
            if (!(j & 1)) {  // after even chunks (chunk 0's fragments were read in the prologue)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            PIN();
            a0 = an0;
            a1 = an1;
        }
    }
#undef PIN
    GSTAMP(2);

    // ---- epilogue: dPre = dHidden * (1 - hidden^2); reduce over u (dEnc) and over t (dPred)
    if (RNNT_XP(a.flags, 8192)) return;  // experiment switch
    // Accumulator register rr = 8 rh + r7 of M tile (2 wm + mt) is row (rr&3) + 8 (rr>>2) + 4 half of
    // its 32: BU = 16: t row rh, u slot 8 (r7>>2) + (r7&3) + 4 half;  BU = 8: t row 2 rh + (r7>>2), u
    // slot (r7&3) + 4 half.  A batch = (mt, rh) = 8 registers = 16 hidden loads.
    constexpr int TPM = 32 / BU;        // t rows per M tile
    constexpr int NU = BU / 2;          // u slots per lane half
    constexpr int TB = BU == 16 ? 1 : 2;  // t rows per batch
    float psum[NU][8];
#pragma unroll
    for (int k = 0; k < NU; ++k)
#pragma unroll
        for (int q = 0; q < 8; ++q) psum[k][q] = 0.f;
    const long BTH = (long)a.B * T * H, BUH = (long)a.B * U1 * H;
    // the hidden rows of the tile end with the utterance: rows t >= T would be the NEXT utterance's
    // cells, which may be unwritten (the forward produces lattice cells only) — out of range, they read 0
    const long rows_utt = (long)(T - t0) * U1 - u0;
    const long hspan = span_rows < rows_utt ? span_rows : rows_utt;
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.hidden + cell0 * H), 0, (int)(hspan * H * 4), 0x00020000);
    const unsigned hvoff[2] = {colok[0] ? (unsigned)(((4 * half) * H + colg[0]) * 4) : 0xfffffff0u,
                               colok[1] ? (unsigned)(((4 * half) * H + colg[1]) * 4) : 0xfffffff0u};
    auto t_of = [&](int mt, int rh, int r7) { return (2 * wm + mt) * TPM + (BU == 16 ? rh : 2 * rh + (r7 >> 2)); };  // t row inside the tile
    auto us_of = [&](int r7) { return BU == 16 ? 8 * (r7 >> 2) + (r7 & 3) : (r7 & 3); };  // u slot without the lane half
    // Four batches of 16 hidden loads, two batches in flight: each batch otherwise waits out a full
    // memory round trip (4 x ~6 000 cycles per tile, stamps).
    auto hload = [&](int k, f32x4 (&hb)[16]) {  // k = mt*2 + rh
        const int mt = k >> 1, rh = k & 1;
#pragma unroll
        for (int r7 = 0; r7 < 8; ++r7) {
            // hidden row of (t, u) through the tile buffer: scalar row offset + per-lane (half,
            // column) offset, no predicate — rows outside the lattice have G = 0, hence an exactly
            // zero accumulator, whatever (finite, or out of range -> 0) hidden value they meet
            const unsigned soff = (unsigned)((t_of(mt, rh, r7) * U1 + us_of(r7)) * H) * 4u;
#pragma unroll
            for (int g = 0; g < 2; ++g)
                hb[r7 * 2 + g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, hvoff[g], soff, 0));
        }
    };
    auto hcomp = [&](int k, const f32x4 (&hb)[16]) {
        const int mt = k >> 1, rh = k & 1;
        float esum[TB][8];
#pragma unroll
        for (int e = 0; e < TB; ++e)
#pragma unroll
            for (int q = 0; q < 8; ++q) esum[e][q] = 0.f;
#pragma unroll
        for (int r7 = 0; r7 < 8; ++r7)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x4 h4 = hb[r7 * 2 + g];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = acc[mt][g * 4 + q][rh * 8 + r7] * (1.f - h4[q] * h4[q]);
                    esum[BU == 16 ? 0 : (r7 >> 2)][g * 4 + q] += d;
                    psum[BU == 16 ? r7 : (r7 & 3)][g * 4 + q] += d;
                }
            }
#pragma unroll
        for (int e = 0; e < TB; ++e) {
#pragma unroll
            for (int q = 0; q < 8; ++q) esum[e][q] += __shfl_xor(esum[e][q], 32, 64);
            const int t = t0 + t_of(mt, rh, 4 * e);
            if (half == 0 && t < Tb) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (colok[g]) {
                        f32x4 o = {esum[e][g * 4], esum[e][g * 4 + 1], esum[e][g * 4 + 2], esum[e][g * 4 + 3]};
                        *(f32x4 *)(a.slab_enc + (long)ub * BTH + ((long)b * T + t) * H + colg[g]) = o;
                    }
            }
        }
    };
    {
        f32x4 hA[16], hB[16];
        hload(0, hA);
        hload(1, hB);
        __builtin_amdgcn_sched_barrier(0);
        hcomp(0, hA);
        __builtin_amdgcn_sched_barrier(0);
        hload(2, hA);
        __builtin_amdgcn_sched_barrier(0);
        hcomp(1, hB);
        __builtin_amdgcn_sched_barrier(0);
        hload(3, hB);
        __builtin_amdgcn_sched_barrier(0);
        hcomp(2, hA);
        hcomp(3, hB);
    }
    if (wm == 1) {
#pragma unroll
        for (int k = 0; k < NU; ++k)
#pragma unroll
            for (int q = 0; q < 8; ++q) s_red[wn][lane][k * 8 + q] = psum[k][q];
    }
    __syncthreads();
    if (wm == 0) {
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int u = u0 + (BU == 16 ? 8 * (k >> 2) + (k & 3) : k) + 4 * half;
            if (u < U1) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    if (colok[g]) {
                        f32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            o[q] = psum[k][g * 4 + q] + s_red[wn][lane][k * 8 + g * 4 + q];
                        *(f32x4 *)(a.slab_pred + (long)tt * BUH + ((long)b * U1 + u) * H + colg[g]) = o;
                    }
            }
        }
    }
    GSTAMP(3);
}

// k_dhidden_gen applies: V in whole 32-wide chunk quadruples, and its raw buffers (num_records is
// 32-bit, per-lane offsets are 32-bit) can span W and one tile's logits / hidden rows.  It covers
// the first 512 columns of H; launch_dhidden sends the rest to k_dhidden.
// u width of k_dhidden_gen's tiles: the one that pads the (T, U1) lattice least
int dhidden_gen_bu(int T, int U1)
{
    const long c16 = (long)((U1 + 15) / 16 * 16) * ((T + 7) / 8 * 8);
    const long c8 = (long)((U1 + 7) / 8 * 8) * ((T + 15) / 16 * 16);
    return c8 < c16 ? 8 : 16;
}
bool dhidden_gen_ok(int H, int V, int U1)
{
    const long span_rows = (long)(16 - 1) * U1 + 16;  // the taller tile form (16 t x 8 u)
    const long wide = V > H ? V : H;
    return (V % 32) == 0 && (long)V * H * 4 < 0xffffffffL && span_rows * wide * 4 < 0x7fffffffL;
}
#define DG_COLS 512  // columns of H one k_dhidden_gen workgroup covers
// column groups of 512 the tile kernel takes: the first (which produces G) and every further WHOLE one
int dhidden_gen_groups(int H) { return H <= DG_COLS ? 1 : H / DG_COLS; }

// out[b,t,:] = sum_ub slab_enc[ub][b,t,:]  (0 for t >= T_b)
__global__ __launch_bounds__(256) void k_reduce_enc(const float *__restrict__ slab,
                                                    const int32_t *__restrict__ logit_lens,
                                                    const int32_t *__restrict__ target_lens,
                                                    float *__restrict__ out, int B, int T, int U1,
                                                    int H, int n_ublk, int bu_lo, int col_split)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
    const int H4 = H / 4;
    const long n = (long)B * T * H4;
    if (idx >= n) return;
    const long bt = idx / H4;
    const int t = (int)(bt % T), b = (int)(bt / T);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (t < len_t(logit_lens, b, T)) {
        // u blocks that start past U_b hold no lattice cell; the fused dHidden kernels skip them
        // without writing their slab
        // slabs written: u blocks of width bu_lo (both dHidden kernels use the same u width)
        const int bu = bu_lo;
        int nub = len_u(target_lens, b, U1) / bu + 1;
        if (nub > n_ublk) nub = n_ublk;
        for (int k = 0; k < nub; ++k) s += ((const f32x4 *)slab)[(long)k * n + idx];
    }
    ((f32x4 *)out)[idx] = s;
}

// out[b,u,:] = sum_{tt < ceil(T_b/bt)} slab_pred[tt][b,u,:]; the t tiles are bt_lo rows high for
// columns < col_split (k_dhidden_gen: 8) and bt_hi for the others (k_dhidden: 4)
__global__ __launch_bounds__(256) void k_reduce_pred(const float *__restrict__ slab,
                                                     const int32_t *__restrict__ logit_lens,
                                                     const int32_t *__restrict__ target_lens,
                                                     float *__restrict__ out, int B, int T, int U1,
                                                     int H, int bt_lo, int bt_hi, int col_split)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int H4 = H / 4;
    const long n = (long)B * U1 * H4;
    if (idx >= n) return;
    const int b = (int)(idx / ((long)U1 * H4));
    const int bt = 4 * (int)(idx % H4) < col_split ? bt_lo : bt_hi;
    const int ntt = (len_t(logit_lens, b, T) + bt - 1) / bt;  // slabs written: t tiles of height bt
    const int u = (int)((idx / H4) % U1);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (u <= len_u(target_lens, b, U1))  // rows past U_b: no lattice cell, and their u block may not have been written
        for (int k = 0; k < ntt; ++k) s += ((const f32x4 *)slab)[(long)k * n + idx];
    ((f32x4 *)out)[idx] = s;
}

void launch_dhidden(const JointBwdArgs &a, hipStream_t st)
{
    int cb0 = 0;  // first 128-column block left to the persistent kernel
    if (a.flags & 16) {  // fused G producer (engine decides: dhidden_gen_ok)
        // zero the padding rows k_dw may touch (k_make_g used to)
        const long cells = (long)a.B * a.T * a.U1;
        launch_fill32((float *)a.logits + cells * a.V, 0u, (size_t)(a.rows_pad + 16 - cells) * a.V * 4, st);
        // H > 512 (the reference's joint is 1024 wide): G now stands in place of the logits; every
        // further WHOLE group of 512 columns runs on the same tile kernel reading G (GEN = false), what
        // is left (H % 512 columns: cfg4's H = 640) on the persistent kernel
        const int n_full = dhidden_gen_groups(a.H);
        if (a.gen_bu == 8) {
            dim3 grid((a.U1 + 7) / 8, (a.T + 15) / 16, a.B);
            hipLaunchKernelGGL((k_dhidden_gen<8, true>), grid, dim3(256), 0, st, a, 0);
            for (int hp = 1; hp < n_full; ++hp) hipLaunchKernelGGL((k_dhidden_gen<8, false>), grid, dim3(256), 0, st, a, hp * DG_COLS);
        } else {
            dim3 grid((a.U1 + 15) / 16, (a.T + DG_BT - 1) / DG_BT, a.B);
            hipLaunchKernelGGL((k_dhidden_gen<16, true>), grid, dim3(256), 0, st, a, 0);
            for (int hp = 1; hp < n_full; ++hp) hipLaunchKernelGGL((k_dhidden_gen<16, false>), grid, dim3(256), 0, st, a, hp * DG_COLS);
        }
        cb0 = n_full * (DG_COLS / 128);
        if (cb0 * 128 >= a.H) return;
    }
    launch_fill32(a.counter, 0u, 8 * 64, st);  // per-XCD work-item counters (64 B apart)
    if (a.gen_bu == 8) hipLaunchKernelGGL(k_dhidden<8>, dim3(a.n_cu), dim3(512), 0, st, a, cb0);
    else hipLaunchKernelGGL(k_dhidden<16>, dim3(a.n_cu), dim3(512), 0, st, a, cb0);
}

void launch_dhidden_reduce(const JointBwdArgs &a, hipStream_t st)
{
    const long n4e = (long)a.B * a.T * (a.H / 4);
    hipLaunchKernelGGL(k_reduce_enc, dim3((unsigned)((n4e + 255) / 256)), dim3(256), 0, st,
                       a.slab_enc, a.logit_lens, a.target_lens, a.grad_enc, a.B, a.T, a.U1, a.H, a.n_ublk, a.gen_bu, a.pred_split_col);
    const long n4p = (long)a.B * a.U1 * (a.H / 4);
    hipLaunchKernelGGL(k_reduce_pred, dim3((unsigned)((n4p + 255) / 256)), dim3(256), 0, st,
                       a.slab_pred, a.logit_lens, a.target_lens, a.grad_pred, a.B, a.T, a.U1, a.H,
                       128 / a.gen_bu, 64 / a.gen_bu, a.pred_split_col);
}

// ---------------------------------------------------------------------------------------
// Elementwise producers for the backward GEMMs (HBM-bound, a few ms at the BASELINE sizes):
//   k_make_hidden  hidden[c,:] = tanh(enc[b,t,:] + pred[b,u,:])          (rnnt/joint.py:32-37)
//   k_make_g       logits[c,:] -> G[c,:] IN PLACE (formula at the top of this file); cells
//                  outside the lattice and the zero-padding rows become exact zeros
// so that both backward GEMMs are pure matrix products whose operands stream HBM -> LDS by
// LDS-DMA with no VALU work in the main loops (operand generation inside the dW loop cost
// ~10 VALU per MFMA and held it at 46 % of the matrix pipe).
__global__ __launch_bounds__(256) void k_make_hidden(const float *__restrict__ enc, long sb,
                                                     long st_, const float *__restrict__ pred,
                                                     float *__restrict__ hid, int B, int T, int U1,
                                                     int H, long rows_pad)
{
    const int H4 = H / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows_pad * H4) return;
    const long c = idx / H4;
    const int h = (int)(idx - c * H4) * 4;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (c < (long)B * T * U1) {
        const int u = (int)(c % U1);
        const long bt = c / U1;
        const int t = (int)(bt % T), b = (int)(bt / T);
        const f32x4 e = *(const f32x4 *)(enc + (long)b * sb + (long)t * st_ + h);
        const f32x4 p = *(const f32x4 *)(pred + ((long)b * U1 + u) * H + h);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fast_tanh(e[q] + p[q]);
    }
    __builtin_nontemporal_store(o, (f32x4 *)(hid + c * H + h));  // 13 GB streamed once: keep it out of the caches
}

__global__ __launch_bounds__(256) void k_make_g(float *__restrict__ logits,
                                                const CellCoef *__restrict__ coef, long rows,
                                                long rows_pad, int V, int blank)
{
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_pad) return;
    float *x = logits + row * V;
    CellCoef c;
    c.c1 = RNNT_NEG_INF; c.sb = 0.f; c.se = 0.f; c.y = -1;
    if (row < rows) c = coef[row];
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
                o[s] = e;
            }
        }
        *(f32x4 *)(x + v) = o;
    }
}

void launch_make_hidden(const JointBwdArgs &a, hipStream_t st)
{
    const long n = a.rows_pad * (a.H / 4);
    hipLaunchKernelGGL(k_make_hidden, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.enc,
                       a.enc_sb, a.enc_st, a.pred, a.hidden, a.B, a.T, a.U1, a.H, a.rows_pad);
}

void launch_make_g(const JointBwdArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_make_g, dim3((unsigned)((a.rows_pad + 3) / 4)), dim3(256), 0, st,
                       (float *)a.logits, a.coef, (long)a.B * a.T * a.U1, a.rows_pad, a.V, a.blank);
}

// ---------------------------------------------------------------------------------------
// dW split-K GEMM:  dW[v,h] = sum_c G[c,v] * hidden[c,h]   (M = V, N = H, K = lattice cells).
//
// Workgroup = 4 waves (one per SIMD) = 2 (M) x 2 (N); workgroup tile 256 (v) x 256 (h); each
// wave owns 128 x 128 = 16 accumulator tiles (256 AGPRs).  Both operands are row-major with the
// K index (cell) as the row: k-step ks multiplies cells 2ks (lanes 0-31) and 2ks+1 (lanes
// 32-63), and a lane's 16-byte load of 4 consecutive v (or h) feeds 4 interleaved MFMA tiles
// (tile q holds rows 4i+q), so the epilogue stores are 16-byte coalesced too.
//
// Operand path (measured, tools/mfma_mix*.hip): register-destination global loads cost the
// matrix pipe nothing to issue, whereas every LDS-DMA piece costs ~60 cycles of issue and an
// LDS ring adds barriers — so the fragments go straight HBM/L2 -> VGPR through a DW_RING-deep
// register ring (8 k-steps = 8192 matrix-pipe cycles of lead, enough for an HBM miss), all
// loads unconditional and compiler-visible (inline-asm ring loads were tried: hipcc's allocator
// copies the still-pending destinations across the loop back-edge); no LDS, no barrier, no
// VALU in the loop except the 4 adds of the bias gradient.  The two waves that share an operand slice
// (same wm or same wn) run in step and meet in L1/L2.
// The grid is 1-D and XCD-aware: the tiles of one split (same cells, different v/h blocks)
// get consecutive remapped ids and therefore share an XCD's L2.
#define DW_RING 8  // k-steps of operands in flight per wave

// Live-row table of the dW GEMM.  K runs over lattice cells, but an utterance shorter than T only
// has G != 0 in its first T_b*U1 cells: the K ranges of the splits are cut from the LIVE granules
// (16 cells here, 32 on the bf16 route) only, so a ragged batch costs what its lengths cost.  tab[b] = first granule of
// utterance b's live range, tab[B+1+b] = live granules before it, tab[2B+1] = their total.  A
// live range is rounded out to whole granules (the extra rows are cells of time step T_b, which
// k_dhidden_gen zero-fills) and never overlaps the previous one.
__global__ void k_dw_table(const int32_t *__restrict__ logit_lens, int B, int T, int U1, int gran, long *__restrict__ tab)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    long prev_end = 0, cum = 0;
    for (int b = 0; b < B; ++b) {
        const long c0 = (long)b * T * U1;
        long s = c0 / gran, e = (c0 + (long)len_t(logit_lens, b, T) * U1 + gran - 1) / gran;
        if (s < prev_end) s = prev_end;
        if (e < s) e = s;
        tab[b] = s;
        tab[B + 1 + b] = cum;
        cum += e - s;
        prev_end = e;
    }
    tab[B] = prev_end;
    tab[2 * B + 1] = cum;
}

// Workgroups per split: 2x2 blocks of 128x128 wave tiles (a 256x256 workgroup tile); when H has
// an odd number of 128-column tiles (cfg4: H = 640) the last column is covered by 4x1 blocks
// instead of a half-empty 256-wide tile (12 -> 10 workgroup tiles per split at cfg4).
void launch_dw_table(const int32_t *logit_lens, int B, int T, int U1, int gran, long *tab, hipStream_t st)
{
    hipLaunchKernelGGL(k_dw_table, dim3(1), dim3(64), 0, st, logit_lens, B, T, U1, gran, tab);
}

// Live-GRANULE list of the fp32 dW GEMM: list[i] = index of the i-th 16-row granule of the [cells]
// buffers that holds at least one lattice cell (t < T_b and u <= U_b), ascending; head[0] = their
// number.  A ragged batch then costs its live CELLS, not its live time steps: the cells with
// u > U_b of every live time step (a quarter of the rows when U_b is uniform in [U/2, U]) drop out
// of the K walk, apart from the few that share a granule with a live cell (k_dhidden_gen / k_make_g
// write exact zeros there).  Built by three small launches (flags + per-block counts, scan of the
// block counts, compaction): the order is ascending, so dW stays bitwise reproducible.
__device__ __forceinline__ bool granule_live(long g, int gran, const int32_t *logit_lens,
                                             const int32_t *target_lens, int B, int T, int U1)
{
    const long cells = (long)B * T * U1;
    long c = g * gran;
    if (c >= cells) return false;
    long bt = c / U1;
    int u = (int)(c - bt * U1);
    int t = (int)(bt % T), b = (int)(bt / T);
    int left = gran;
    while (left > 0 && b < B) {  // the rows (b,t) this granule touches: first from u, then from 0
        if (t < len_t(logit_lens, b, T) && u <= len_u(target_lens, b, U1)) return true;
        left -= U1 - u;
        u = 0;
        if (++t == T) { t = 0; ++b; }
    }
    return false;
}
#define DWL_BLK 1024
__global__ __launch_bounds__(DWL_BLK) void k_dw_list_count(const int32_t *__restrict__ logit_lens,
                                                           const int32_t *__restrict__ target_lens, int B, int T,
                                                           int U1, int gran, long n_gran, int *__restrict__ blk)
{
    const long g = (long)blockIdx.x * DWL_BLK + threadIdx.x;
    const bool live = g < n_gran && granule_live(g, gran, logit_lens, target_lens, B, T, U1);
    const int cnt = __syncthreads_count(live);
    if (threadIdx.x == 0) blk[blockIdx.x] = cnt;
}
// exclusive scan of the block counts in place (one workgroup); head[0] = total, head[1] = 1 when the
// list is shorter than the table's live ranges (some granule of a live time step is dead): list walk
__global__ __launch_bounds__(1024) void k_dw_list_scan(int *__restrict__ blk, int nblk, long *__restrict__ head,
                                                       const long *__restrict__ tab, int B)
{
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? blk[i] : 0;
        int x = v;  // inclusive scan inside the wave, then across the 16 waves
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if ((threadIdx.x & 63) >= d) x += y; }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        int off = s_carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) off += s_w[w];
        if (i < nblk) blk[i] = off + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) { head[0] = s_carry; head[1] = s_carry < tab[2 * B + 1] ? 1 : 0; }
}
__global__ __launch_bounds__(DWL_BLK) void k_dw_list_write(const int32_t *__restrict__ logit_lens,
                                                           const int32_t *__restrict__ target_lens, int B, int T,
                                                           int U1, int gran, long n_gran, const int *__restrict__ blk,
                                                           int *__restrict__ list)
{
    __shared__ int s_w[DWL_BLK / 64];
    const long g = (long)blockIdx.x * DWL_BLK + threadIdx.x;
    const bool live = g < n_gran && granule_live(g, gran, logit_lens, target_lens, B, T, U1);
    const unsigned long long bal = __ballot(live);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = __popcll(bal);
    __syncthreads();
    int off = blk[blockIdx.x];
    for (int w = 0; w < wave; ++w) off += s_w[w];
    if (live) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int)g;
}

// region: [0] head (live count, mode) 64 B | table (2B+2 longs) | block offsets | list
static size_t dwl_tab_bytes(int B) { return (((size_t)2 * B + 2) * 8 + 63) & ~(size_t)63; }
size_t dw_list_bytes(int B, int T, int U1, int gran)
{
    const size_t n_gran = ((size_t)B * T * U1 + gran - 1) / gran;
    const size_t nblk = (n_gran + DWL_BLK - 1) / DWL_BLK;
    return 64 + dwl_tab_bytes(B) + ((nblk * 4 + 63) & ~(size_t)63) + (n_gran + 16) * 4;
}
const int *dw_list_ptr(const void *region, int B, int T, int U1, int gran)
{
    const long n_gran = ((long)B * T * U1 + gran - 1) / gran;
    const size_t nblk = (size_t)((n_gran + DWL_BLK - 1) / DWL_BLK);
    return (const int *)((const char *)region + 64 + dwl_tab_bytes(B) + ((nblk * 4 + 63) & ~(size_t)63));
}
void launch_dw_list(const int32_t *logit_lens, const int32_t *target_lens, int B, int T, int U1, int gran,
                    void *region, hipStream_t st)
{
    const long n_gran = ((long)B * T * U1 + gran - 1) / gran;
    const int nblk = (int)((n_gran + DWL_BLK - 1) / DWL_BLK);
    long *head = (long *)region;
    long *tab = head + 8;
    int *blk = (int *)((char *)region + 64 + dwl_tab_bytes(B));
    int *list = const_cast<int *>(dw_list_ptr(region, B, T, U1, gran));
    launch_dw_table(logit_lens, B, T, U1, gran, tab, st);
    hipLaunchKernelGGL(k_dw_list_count, dim3(nblk), dim3(DWL_BLK), 0, st, logit_lens, target_lens, B, T, U1, gran, n_gran, blk);
    hipLaunchKernelGGL(k_dw_list_scan, dim3(1), dim3(1024), 0, st, blk, nblk, head, tab, B);
    hipLaunchKernelGGL(k_dw_list_write, dim3(nblk), dim3(DWL_BLK), 0, st, logit_lens, target_lens, B, T, U1, gran, n_gran, blk, list);
}

int dw_tiles(int H, int V)
{
    const int nv = (V + 127) / 128, nh = (H + 127) / 128;
    return (nh / 2) * ((nv + 1) / 2) + (nh & 1) * ((nv + 3) / 4);
}

template <bool LIST>
__global__ __launch_bounds__(256, 1) void k_dw(JointBwdArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    const int H = a.H, V = a.V;
    const int nv = (V + 127) / 128, nh = (H + 127) / 128;  // 128-wide wave tiles
    const int n22 = (nh / 2) * ((nv + 1) / 2);                // 2x2 blocks
    const int tiles = n22 + (nh & 1) * ((nv + 3) / 4);        // + 4x1 blocks of an odd last column
    const int total = tiles * a.n_split;
    // XCD-aware remap (bijective for any total): ids that are congruent mod 8 share an XCD
    int id = blockIdx.x;
    {
        const int q8 = total / 8, r8 = total % 8, x = id % 8;
        id = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + id / 8;
    }
    const int tile = id % tiles, split = id / tiles;
    // the waves never join, so a workgroup is just four wave tiles that share operands in L1
    int vt, ht;
    if (tile < n22) { vt = 2 * (tile / (nh / 2)) + wm; ht = 2 * (tile % (nh / 2)) + wn; }
    else            { vt = 4 * (tile - n22) + wave;    ht = nh - 1; }
    const int v0 = vt * 128, h0 = ht * 128;
    const int vbase = v0 + 4 * i, hbase = h0 + 4 * i;
    const bool vok = vbase < V, hok = hbase < H;
    // Two K walks, chosen on the DEVICE by the list builder (head[1]): batches whose live time steps
    // hold no dead granule (all target lengths full, or nearly) take the contiguous-range walk — one
    // running pointer, nothing but pointer increments between the MFMAs; ragged batches take the
    // live-granule list (≈14 scalar instructions per 128 MFMAs more, ≈1 %, for every dead granule
    // skipped).  The kernel is launched once per mode; the launch whose mode is not selected exits.
    if ((((const long *)a.dw_tab)[1] != 0) != LIST) return;
    f32x16 acc[4][4];
#pragma unroll
    for (int qm = 0; qm < 4; ++qm)
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (LIST) {
    // this split's share of the LIVE granules (launch_dw_list): a contiguous piece of the list
    const long nlive = ((const long *)a.dw_tab)[0];
    const int *__restrict__ lst = a.dw_list;
    const long g_lo = nlive * split / a.n_split, g_hi = nlive * (split + 1) / a.n_split;


    if (g_hi > g_lo) {  // workgroup-uniform
        // columns beyond V / H (edge tiles) re-read the last valid 16 bytes: they only feed
        // accumulators that are never stored.
        const float *gbase = a.logits + (long)half * V + (vok ? vbase : V - 4);
        const float *hbase_p = a.hidden + (long)half * H + (hok ? hbase : H - 4);
        const long gstep = 2L * V, hstep = 2L * H;
        const long ggran = (long)DW_KC * V, hgran = (long)DW_KC * H;  // floats per granule
        // Register ring of DW_RING k-steps = ONE granule (16 cells).  The MFMAs read their ring slot
        // directly; the slot is refilled one k-step LATER, from inside the next k-step's MFMA stream
        // (its own MFMAs have all issued by then, so the load does not wait on them), and the
        // bias-gradient adds ride between MFMAs too: nothing stands in front of a k-step's 16 MFMAs.
        // Slot 7 is refilled with the CURRENT granule's last k-step, slots 0-6 with the NEXT live
        // granule's k-steps 0-6 — wherever the list says that granule lies: the stream runs on
        // across dead stretches without a pipeline restart.
        // (Earlier forms: copy the slot out + refill in front of the MFMAs 47.4 ms; refill right
        // behind its own MFMAs 50.7 ms; inline-asm loads: hipcc copies pending destinations.)
        // The loop body is rotated by one k-step — k-steps 1..7 of granule r, then k-step 0 of granule
        // r+1 — so that all eight refills of an iteration read the SAME granule (r+1) through one
        // pointer; with two pointers in the body hipcc copied the ring at the back-edge behind a
        // vmcnt(0), draining it every round.
        const long g0 = lst[g_lo];
        const float *gc = gbase + g0 * ggran, *hc = hbase_p + g0 * hgran;
        f32x4 ra[DW_RING], rb[DW_RING];
#pragma unroll
        for (int s_ = 0; s_ < DW_RING; ++s_) {
            ra[s_] = *(const f32x4 *)(gc + s_ * gstep);
            rb[s_] = *(const f32x4 *)(hc + s_ * hstep);
        }
        // one k-step: 16 MFMAs from slot SL, the refill of slot RF (if any) behind the first row of
        // MFMAs, the bias-gradient adds between the others
#define DW_KSTEP(SL, REFILL)                                                                              \
        do {                                                                                              \
            auto row = [&](int qm) {                                                                      \
                _Pragma("unroll") for (int qn = 0; qn < 4; ++qn)                                          \
                    acc[qm][qn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[SL][qm], rb[SL][qn], acc[qm][qn], 0, 0, 0); \
            };                                                                                            \
            row(0);                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            REFILL;                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            row(1);                                                                                       \
            dbacc[0] += ra[SL][0];                                                                        \
            dbacc[1] += ra[SL][1];                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            row(2);                                                                                       \
            dbacc[2] += ra[SL][2];                                                                        \
            dbacc[3] += ra[SL][3];                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            row(3);                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                            \
        } while (0)
        DW_KSTEP(0, (void)0);
        for (long r = g_lo; r + 1 < g_hi; ++r) {
            const long gn_i = lst[r + 1];  // the next live granule, wherever it lies
            const float *gn = gbase + gn_i * ggran, *hn = hbase_p + gn_i * hgran;
#pragma unroll
            for (int s_ = 1; s_ < DW_RING; ++s_)
                DW_KSTEP(s_, (ra[s_ - 1] = *(const f32x4 *)(gn + (s_ - 1) * gstep),
                              rb[s_ - 1] = *(const f32x4 *)(hn + (s_ - 1) * hstep)));
            DW_KSTEP(0, (ra[DW_RING - 1] = *(const f32x4 *)(gn + (DW_RING - 1) * gstep),
                         rb[DW_RING - 1] = *(const f32x4 *)(hn + (DW_RING - 1) * hstep)));
        }
#pragma unroll
        for (int s_ = 1; s_ < DW_RING; ++s_) DW_KSTEP(s_, (void)0);
#undef DW_KSTEP
    }

    } else {
    // this split's share of the LIVE granules (k_dw_table), walked utterance by utterance
    const long *tab = a.dw_tab + 8;  // the table sits 64 bytes into the region (head words first)
    const int B = a.B;
    const long nlive = tab[2 * B + 1];
    const long g_lo = nlive * split / a.n_split, g_hi = nlive * (split + 1) / a.n_split;


    int ub = 0;
    while (ub + 1 < B && tab[B + 1 + ub + 1] <= g_lo) ++ub;  // utterance holding live granule g_lo
    for (long gq = g_lo; gq < g_hi; ++ub) {  // workgroup-uniform
        const long cum0 = tab[B + 1 + ub], cum1 = ub + 1 < B ? tab[B + 1 + ub + 1] : nlive;
        const long ge = cum1 < g_hi ? cum1 : g_hi;  // end of this utterance's part (live index)
        if (ge <= gq) continue;
        const long k_lo = tab[ub] + (gq - cum0);     // actual first granule
        const long nstep = (ge - gq) * (DW_KC / 2);  // k-steps (cell pairs); multiple of DW_RING
        gq = ge;
        // columns beyond V / H (edge tiles) re-read the last valid 16 bytes: they only feed
        // accumulators that are never stored.  The buffers carry DW_KC extra rows, so the
        // ring may run DW_RING k-steps past the end of a range (loaded, never multiplied).
        const float *gp = a.logits + (k_lo * DW_KC + half) * (long)V + (vok ? vbase : V - 4);
        const float *hp = a.hidden + (k_lo * DW_KC + half) * (long)H + (hok ? hbase : H - 4);
        const long gstep = 2L * V, hstep = 2L * H;
        // Register ring of DW_RING k-steps.  The MFMAs read their ring slot directly; the slot
        // is refilled one k-step LATER, from inside the next k-step's MFMA stream (its own MFMAs
        // have all issued by then, so the load does not wait on them), and the bias-gradient
        // adds ride between MFMAs too: nothing stands in front of a k-step's 16 MFMAs.
        // (Earlier forms: copy the slot out + refill in front of the MFMAs 47.4 ms; refill right
        // behind its own MFMAs 50.7 ms; inline-asm loads: hipcc copies pending destinations.)
        f32x4 ra[DW_RING], rb[DW_RING];
#pragma unroll
        for (int s_ = 0; s_ < DW_RING; ++s_) {
            ra[s_] = *(const f32x4 *)(gp + s_ * gstep);
            rb[s_] = *(const f32x4 *)(hp + s_ * hstep);
        }
        gp += DW_RING * gstep;  // -> k-step DW_RING
        hp += DW_RING * hstep;
        for (long st = 0; st < nstep; st += DW_RING) {
#pragma unroll
            for (int s_ = 0; s_ < DW_RING; ++s_) {
                auto row = [&](int qm) {
#pragma unroll
                    for (int qn = 0; qn < 4; ++qn)
                        acc[qm][qn] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s_][qm], rb[s_][qn], acc[qm][qn], 0, 0, 0);
                };
                row(0);
                __builtin_amdgcn_sched_barrier(0);
                {   // refill the previous slot with k-step st + s_ - 1 + DW_RING (at the very first
                    // step that re-reads k-step DW_RING-1 into the slot that already holds it)
                    const int ps = (s_ + DW_RING - 1) % DW_RING;
                    ra[ps] = *(const f32x4 *)(gp + (s_ - 1) * gstep);
                    rb[ps] = *(const f32x4 *)(hp + (s_ - 1) * hstep);
                }
                __builtin_amdgcn_sched_barrier(0);
                row(1);
                dbacc[0] += ra[s_][0];
                dbacc[1] += ra[s_][1];
                __builtin_amdgcn_sched_barrier(0);
                row(2);
                dbacc[2] += ra[s_][2];
                dbacc[3] += ra[s_][3];
                __builtin_amdgcn_sched_barrier(0);
                row(3);
                __builtin_amdgcn_sched_barrier(0);
            }
            gp += DW_RING * gstep;
            hp += DW_RING * hstep;
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
    if (ht == 0) {
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
    launch_dw_list(a.logit_lens, a.target_lens, a.B, a.T, a.U1, DW_KC, a.dw_tab, st);
    JointBwdArgs b = a;
    b.dw_list = dw_list_ptr(a.dw_tab, a.B, a.T, a.U1, DW_KC);
    hipLaunchKernelGGL(k_dw<false>, dim3(dw_tiles(a.H, a.V) * a.n_split), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_dw<true>, dim3(dw_tiles(a.H, a.V) * a.n_split), dim3(256), 0, st, b);
}

void launch_dw_reduce(const JointBwdArgs &a, hipStream_t st)
{
    const long n4w = (long)a.V * a.H / 4;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n4w + 255) / 256)), dim3(256), 0, st,
                       a.slab_w, a.grad_W, n4w, a.n_split);
    const long n4b = a.V / 4;
    hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n4b + 255) / 256)), dim3(256), 0, st,
                       a.slab_b, a.grad_bias, n4b, a.n_split);
}
