// smallgemm.hip — see smallgemm.hpp.  Used by predictor.hip (ConvPredictor forward / backward,
// reference rnnt/predictor.py:189-229) and the joint's input projections (rnnt/joint.py:26-30).
#include "smallgemm.hpp"

namespace {

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// acc += the other waves' partial tiles, through `red` (NT tiles x 1024 floats): rounds 1..3, wave
// w publishes, wave 0 adds.  Ends with every wave past the last barrier.
template <int NTILES>
__device__ __forceinline__ void reduce_to_wave0(f32x16 (&acc)[NTILES], float *red, int wave, int lane)
{
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < NTILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < NTILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
}

// ---- NT: wave tile 32 rows x 128 columns (4 tiles of 32 columns); grid (ceil(N/128), ceil(M/32))
__global__ __launch_bounds__(256) void k_sgemm_nt(SgArgs a)
{
    __shared__ float red[4 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int M = a.M, N = a.N, K = a.K;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 128;
    const int row = min(m0 + i, M - 1), u = row % a.seg;
    const int KC = (K + 7) / 8, total = a.taps * KC;
    long wrow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wrow[q] = (long)min(n0 + 32 * q + i, N - 1) * a.ldb;

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    struct Ops { f32x4 x, w[4]; };
    auto load = [&](Ops &o, int c) {
        const int tap = c / KC, kc = c - tap * KC;
        const int k = 8 * kc + 4 * half;
        const bool kok = k < K;
        const int kk = kok ? k : K - 4;
        const int shift = tap - (a.taps - 1);
        const bool rok = u + shift >= 0;
        o.x = *(const f32x4 *)(a.A + (long)(rok ? row + shift : row) * a.lda + kk);
        if (!rok || !kok) o.x = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *wt = a.B + (long)tap * N * a.ldb + kk;
#pragma unroll
        for (int q = 0; q < 4; ++q) o.w[q] = *(const f32x4 *)(wt + wrow[q]);
    };
    auto compute = [&](const Ops &o) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.x[s], o.w[q][s], acc[q], 0, 0, 0);
    };
    if (wave < total) {  // operands requested two chunks ahead (one chunk of 16 MFMAs does not cover an L2 round trip)
        Ops cur, nxt, nx2;
        load(cur, wave);
        if (wave + 4 < total) load(nxt, wave + 4);
        for (int c = wave; c < total; c += 4) {
            if (c + 8 < total) load(nx2, c + 8);
            compute(cur);
            cur = nxt;
            nxt = nx2;
        }
    }
    reduce_to_wave0<4>(acc, red, wave, lane);
    if (wave != 0) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + 32 * q + i;
        if (n >= N) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= M) continue;
            float v = acc[q][r] + bv;
            if (a.Cpre) a.Cpre[(long)m * a.ldc + n] = v;
            if (a.act == 1) v = gelu_exact(v);
            if (a.mask) v = a.mask[(long)m * N + n] ? v * a.mask_scale : 0.f;
            a.C[(long)m * a.ldc + n] = v;
        }
    }
}

// ---- NN: wave tile 32 rows x 128 columns (4 interleaved tiles: columns c0 + 4j + q)
__global__ __launch_bounds__(256) void k_sgemm_nn(SgArgs a)
{
    __shared__ float red[4 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int M = a.M, N = a.N, Kc = a.K;
    const int m0 = blockIdx.y * 32, c0 = blockIdx.x * 128;
    const int row = min(m0 + i, M - 1), u = row % a.seg;
    const int KC = (Kc + 7) / 8, total = a.taps * KC;
    const bool cok = c0 + 4 * i < N;
    const int col = cok ? c0 + 4 * i : N - 4;

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    struct Ops { f32x4 y, w[4]; };
    auto load = [&](Ops &o, int c) {
        const int tap = c / KC, kc = c - tap * KC;
        const int k = 8 * kc + 4 * half;
        const bool kok = k < Kc;
        const int kk = kok ? k : Kc - 4;
        const int shift = (a.taps - 1) - tap;
        const bool rok = u + shift < a.seg && row + shift < M;
        o.y = *(const f32x4 *)(a.A + (long)(rok ? row + shift : row) * a.lda + kk);
        if (!rok || !kok) o.y = f32x4{0.f, 0.f, 0.f, 0.f};
        const float *wt = a.B + ((long)tap * Kc + kk) * a.ldb + col;
#pragma unroll
        for (int s = 0; s < 4; ++s) o.w[s] = *(const f32x4 *)(wt + (long)s * a.ldb);
    };
    auto compute = [&](const Ops &o) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.y[s], o.w[s][q], acc[q], 0, 0, 0);
    };
    if (wave < total) {  // operands requested two chunks ahead (one chunk of 16 MFMAs does not cover an L2 round trip)
        Ops cur, nxt, nx2;
        load(cur, wave);
        if (wave + 4 < total) load(nxt, wave + 4);
        for (int c = wave; c < total; c += 4) {
            if (c + 8 < total) load(nx2, c + 8);
            compute(cur);
            cur = nxt;
            nxt = nx2;
        }
    }
    reduce_to_wave0<4>(acc, red, wave, lane);
    if (wave != 0 || !cok) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= M) continue;
        const f32x4 o = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        *(f32x4 *)(a.C + (long)m * a.ldc + col) = o;
    }
}

// ---- TN: workgroup tile 128 x 128, wave (wy, wx) owns the 64 x 64 quarter (2 x 2 interleaved tiles: rows
// n0 + 64wy + 2i + qm, columns k0 + 64wx + 2j + qn) and walks the WHOLE contraction range of the split by itself: no
// barrier, no LDS, 64 accumulator registers, the operands of its next four steps in flight.  (Round 2's form — one wave
// = the whole 128 x 128 tile, the four waves splitting the contraction, a 3-round LDS reduction at the end — was one wave
// per SIMD with one step of operands in flight: it waited out a memory round trip per 16 MFMAs, 0.23 of the fp32 MFMA
// peak at 6 432 rows.)  grid (ceil(K/128), ceil(N/128), taps x splits)
__global__ __launch_bounds__(256) void k_sgemm_tn(SgArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 31, half = lane >> 5;
    const int wy = wave >> 1, wx = wave & 1;
    const int Mc = a.M, N = a.N, K = a.K;
    const int k0 = blockIdx.x * 128 + 64 * wx, n0 = blockIdx.y * 128 + 64 * wy;
    const int ks_n = a.ksplit > 0 ? a.ksplit : 1;
    const int tap = blockIdx.z / ks_n, split = blockIdx.z - tap * ks_n;
    const bool nok = n0 + 2 * i < N, kok = k0 + 2 * i < K;  // N, K % 4 == 0: a pair exists or not as a whole
    const int ncol = nok ? n0 + 2 * i : 0, kcol = kok ? k0 + 2 * i : 0;
    const int shift = tap - (a.taps - 1);
    const int steps_all = (Mc + 1) / 2;
    const int s_lo = (int)((long)steps_all * split / ks_n), s_hi = (int)((long)steps_all * (split + 1) / ks_n);
    const int steps = s_hi - s_lo;
    typedef float f32x2v __attribute__((ext_vector_type(2)));

    f32x16 acc[2][2];
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[qm][qn][r] = 0.f;

    struct Ops { f32x2v y, x; };
    auto load = [&](Ops &o, int ks) {
        const int m = 2 * (s_lo + ks) + half;
        const bool mok = m < Mc;
        const int mm = mok ? m : Mc - 1;
        const bool xok = mok && (mm % a.seg) + shift >= 0;
        o.y = *(const f32x2v *)(a.A + (long)mm * a.lda + ncol);
        o.x = *(const f32x2v *)(a.B + (long)(xok ? mm + shift : mm) * a.ldb + kcol);
        if (!xok) o.x = f32x2v{0.f, 0.f};
        if (!mok) o.y = f32x2v{0.f, 0.f};
    };
    auto compute = [&](const Ops &o) {
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int qn = 0; qn < 2; ++qn)
                acc[qm][qn] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.y[qm], o.x[qn], acc[qm][qn], 0, 0, 0);
    };
    {
        Ops r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) load(r[j], j < steps ? j : 0);
        int ks = 0;
        for (; ks + 4 <= steps; ks += 4) {  // whole groups of four: statically indexed ring
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                compute(r[j]);
                load(r[j], ks + j + 4 < steps ? ks + j + 4 : steps - 1);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (ks + j < steps) compute(r[j]);
    }
    if (!kok) return;
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * half) + qm;
            if (n >= N) continue;
            const f32x2v o = {acc[qm][0][r], acc[qm][1][r]};
            *(f32x2v *)(a.C + (((long)split * a.taps + tap) * N + n) * a.ldc + kcol) = o;
        }
}

__global__ __launch_bounds__(256) void k_pack_conv_w(const float *__restrict__ w, float *__restrict__ wp,
                                                     int out_c, int in_c, int taps, int unpack)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)out_c * in_c * taps;
    if (idx >= n) return;
    // idx walks the torch layout [out][in][tap]
    const int t = (int)(idx % taps);
    const long oi = idx / taps;
    const int ci = (int)(oi % in_c), co = (int)(oi / in_c);
    const long pidx = ((long)t * out_c + co) * in_c + ci;
    if (unpack) const_cast<float *>(w)[idx] = wp[pidx];
    else wp[pidx] = w[idx];
}

// Column sums in two launches, fixed order (bitwise reproducible).  Stage 1: a workgroup sums 256 rows of 64 columns —
// thread (cq = tid & 15, rl = tid >> 4) adds rows rl, rl+16, .. of column quad cq (16-byte loads, a row of the
// workgroup = 256 contiguous bytes), the 16 row lanes are combined through LDS in index order; stage 2 sums the
// ceil(M/256) slabs.  (Round 2's form — one thread per column, 64-row slabs — spent 16 + 12 us on 13 MB: the second
// stage walked 50-100 slabs serially; seven of these pairs are a fifth of the predictor's backward.)
__global__ __launch_bounds__(256) void k_colsum1(const float *__restrict__ Y0, const float *__restrict__ Y1, long ldy, int M, int N,
                                                 float *__restrict__ scratch0)
{
    __shared__ f32x4 red[256];
    const float *Y = blockIdx.z ? Y1 : Y0;  // two tensors of one shape in one launch
    float *scratch = scratch0 + (long)blockIdx.z * gridDim.y * N;
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int n = blockIdx.x * 64 + 4 * cq;
    const int m0 = blockIdx.y * 256;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (n < N) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int m = m0 + rl + 16 * j;
            if (m < M) s += *(const f32x4 *)(Y + (long)m * ldy + n);
        }
    }
    red[tid] = s;
    __syncthreads();
    if (rl == 0 && n < N) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[16 * k + cq];
        *(f32x4 *)(scratch + (long)blockIdx.y * N + n) = s;
    }
}
__global__ __launch_bounds__(256) void k_colsum2(const float *__restrict__ scratch0, int nslab, int N,
                                                 float *__restrict__ out0, float *__restrict__ out1)
{
    const float *scratch = scratch0 + (long)blockIdx.y * nslab * N;
    float *out = blockIdx.y ? out1 : out0;
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n >= N) return;
    f32x4 s = *(const f32x4 *)(scratch + n);
    for (int k = 1; k < nslab; ++k) s += *(const f32x4 *)(scratch + (long)k * N + n);
    *(f32x4 *)(out + n) = s;
}

__global__ __launch_bounds__(256) void k_sum_slabs(const float *__restrict__ slabs, float *__restrict__ out,
                                                   long n4, int ns)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    f32x4 s = ((const f32x4 *)slabs)[idx];
    for (int k = 1; k < ns; ++k) s += ((const f32x4 *)slabs)[(long)k * n4 + idx];
    ((f32x4 *)out)[idx] = s;
}

// The tail of a weight-gradient GEMM in ONE launch: dW = sum of the split-K slabs (conv: written back in torch's
// [out][in][tap] order from the packed [tap][out][in] the GEMM produced) and, in the blocks behind those, the second
// stage of the bias gradient's column sum.  Fixed order throughout.
__global__ __launch_bounds__(256) void k_wgrad_finish(const float *__restrict__ slabs, int ns, float *__restrict__ dW, int out_c,
                                                      int in_c, int taps, int nbw, const float *__restrict__ cs, int nslab, int N,
                                                      float *__restrict__ db)
{
    if ((int)blockIdx.x >= nbw) {  // bias part
        const int n = (((int)blockIdx.x - nbw) * 256 + threadIdx.x) * 4;
        if (n >= N) return;
        f32x4 s = *(const f32x4 *)(cs + n);
        for (int k = 1; k < nslab; ++k) s += *(const f32x4 *)(cs + (long)k * N + n);
        *(f32x4 *)(db + n) = s;
        return;
    }
    const long n4 = (long)taps * out_c * in_c / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // one packed quad: (tap, co, ci .. ci+3)
    if (idx >= n4) return;
    f32x4 s = ((const f32x4 *)slabs)[idx];
    for (int k = 1; k < ns; ++k) s += ((const f32x4 *)slabs)[(long)k * n4 + idx];
    if (taps == 1) { ((f32x4 *)dW)[idx] = s; return; }
    const long e = idx * 4;
    const int ci = (int)(e % in_c);
    const long tc = e / in_c;
    const int co = (int)(tc % out_c), t = (int)(tc / out_c);
    float *o = dW + ((long)co * in_c + ci) * taps + t;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[(long)q * taps] = s[q];
}

}  // namespace

void launch_wgrad_finish(const float *slabs, int ns, float *dW, int out_c, int in_c, int taps, const float *cs, int nslab, int N,
                         float *db, hipStream_t st)
{
    const int nbw = (int)(((long)taps * out_c * in_c / 4 + 255) / 256), nbb = db ? (N / 4 + 255) / 256 : 0;
    hipLaunchKernelGGL(k_wgrad_finish, dim3(nbw + nbb), dim3(256), 0, st, slabs, ns, dW, out_c, in_c, taps, nbw, cs, nslab, N, db);
}

void launch_sgemm_nt(const SgArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_sgemm_nt, dim3((a.N + 127) / 128, (a.M + 31) / 32), dim3(256), 0, st, a);
}
void launch_sgemm_nn(const SgArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_sgemm_nn, dim3((a.N + 127) / 128, (a.M + 31) / 32), dim3(256), 0, st, a);
}
void launch_sgemm_tn(const SgArgs &a, hipStream_t st)
{
    const int ks = a.ksplit > 0 ? a.ksplit : 1;
    hipLaunchKernelGGL(k_sgemm_tn, dim3((a.K + 127) / 128, (a.N + 127) / 128, a.taps * ks), dim3(256), 0, st, a);
}
// one range per ~256 contraction rows, at most 16: a few hundred MFMA k-steps per wave
int sgemm_tn_splits(int Mc) { const int s = (Mc + 255) / 256; return s < 1 ? 1 : (s > 16 ? 16 : s); }
// Splits for ONE weight-gradient GEMM, never more than sgemm_tn_splits(Mc) (which sizes the slab buffer).  Each split
// ends in a 3-round LDS reduction of its 256 x 128 tile and writes a full [taps][N][K] slab, so a split should carry
// ~450 contraction rows — but the grid should also fill the chip once.  Measured on the predictor's three GEMMs
// (tools/bench_predictor.py, fwd+bwd): 3 232 rows 1.171 ms with 13 splits each, 1.152 with 6 / 10 / 13; 6 432 rows
// 1.923 with 16 each, 1.999 with 6 / 10 / 16, 2.062 with 4 / 6 / 8.
int sgemm_tn_splits_for(int Mc, int N, int K, int taps)
{
    const int per = ((N + 127) / 128) * ((K + 127) / 128) * taps;
    const int by_rows = (Mc + 447) / 448, by_fill = (256 + per - 1) / per;
    const int s = by_rows > by_fill ? by_rows : by_fill, cap = sgemm_tn_splits(Mc);
    return s < 1 ? 1 : (s > cap ? cap : s);
}
void launch_sum_slabs(const float *slabs, float *out, long n, int ns, hipStream_t st)
{
    hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, slabs, out, n / 4, ns);
}
void launch_pack_conv_w(const float *w, float *wp, int out_c, int in_c, int taps, hipStream_t st)
{
    const long n = (long)out_c * in_c * taps;
    hipLaunchKernelGGL(k_pack_conv_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, wp, out_c, in_c, taps, 0);
}
void launch_unpack_conv_w(const float *wp, float *w, int out_c, int in_c, int taps, hipStream_t st)
{
    const long n = (long)out_c * in_c * taps;
    hipLaunchKernelGGL(k_pack_conv_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, const_cast<float *>(wp), out_c, in_c, taps, 1);
}
size_t colsum_scratch_floats(int M, int N) { const int s = (M + 63) / 64; return (size_t)(s < 2 ? 2 : s) * N; }  // >= two tensors' 256-row slabs
void launch_colsum(const float *Y, long ldy, int M, int N, float *out, float *scratch, hipStream_t st)
{
    const int nslab = colsum_slabs(M);  // (colsum_scratch_floats still sizes the scratch for 64-row slabs: room for 4 tensors)
    hipLaunchKernelGGL(k_colsum1, dim3((N + 63) / 64, nslab), dim3(256), 0, st, Y, Y, ldy, M, N, scratch);
    hipLaunchKernelGGL(k_colsum2, dim3((N / 4 + 255) / 256), dim3(256), 0, st, scratch, nslab, N, out, out);
}
int colsum_slabs(int M) { return (M + 255) / 256; }
void launch_colsum_stage1(const float *Y, long ldy, int M, int N, float *scratch, hipStream_t st)
{
    hipLaunchKernelGGL(k_colsum1, dim3((N + 63) / 64, colsum_slabs(M)), dim3(256), 0, st, Y, Y, ldy, M, N, scratch);
}
void launch_colsum_pair(const float *Y0, const float *Y1, long ldy, int M, int N, float *out0, float *out1, float *scratch, hipStream_t st)
{
    const int nslab = colsum_slabs(M);
    hipLaunchKernelGGL(k_colsum1, dim3((N + 63) / 64, nslab, 2), dim3(256), 0, st, Y0, Y1, ldy, M, N, scratch);
    hipLaunchKernelGGL(k_colsum2, dim3((N / 4 + 255) / 256, 2), dim3(256), 0, st, scratch, nslab, N, out0, out1);
}
