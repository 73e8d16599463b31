// smallgemm.hpp — fp32 MFMA GEMMs for the SMALL matrices either side of the joint (input
// projections, ConvPredictor): M = B*T or B*(U+1) rows (hundreds to a few thousand), K and N of
// 512-1024.  Three forms, the same three the joint's own GEMMs have:
//   NT  C[m,n]  = sum_tap sum_k X[m+shift(tap), k] * W[tap][n][k]  (+bias, GELU, dropout mask)
//   NN  C[m,k]  = sum_tap sum_n Y[m+shift(tap), n] * W[tap][n][k]
//   TN  C[tap][n][k] = sum_m Y[m,n] * X[m+shift(tap), k]
// `taps` > 1 turns them into the three GEMMs of a causal 1-D convolution over sequences of `seg`
// rows (CausalConv1d, reference rnnt/causalconv.py:9-32: left zero padding, stride 1): tap j of a
// K-tap kernel reads row u - (taps-1) + j, rows before the start of a sequence are zeros; the weight
// is packed [tap][out][in] (k_pack_conv_w) so that one array serves NT (rows n, k contiguous) and NN
// (rows n = contraction, k = output columns contiguous).
// v_mfma_f32_32x32x2_f32, operands straight L2 -> VGPR with the same register idioms as the joint
// kernels (a lane's 16-byte load = 4 k-steps, or 4 interleaved tiles).  These problems are a few
// GFLOP: the design goal is latency — every workgroup splits its contraction over its 4 waves
// (each wave one SIMD) and sums the partial tiles through LDS — not the last percent of the pipe.
#pragma once
#include "common.hpp"

struct SgArgs {
    const float *A; long lda;   // NT: X [M,K]; NN: Y [M,Kc]; TN: Y [Mc,N]
    const float *B; long ldb;   // NT: W [taps][N][K]; NN: W [taps][Kc][N]; TN: X [Mc,K]
    float *C; long ldc;         // NT/NN: [M,N]; TN: [taps][N][K] (ldc = K)
    const float *bias;          // NT: [N] or NULL
    float *Cpre;                // NT: pre-activation copy (NULL: none)
    const unsigned char *mask;  // NT: keep mask [M,N] (NULL: none); kept values x mask_scale
    float mask_scale;
    int M, N, K;                // NT: rows, cols, contraction per tap; NN: rows, cols, contraction
                                // per tap (Kc); TN: contraction rows (Mc), output rows N, cols K
    int taps, seg;              // conv taps (1: plain GEMM), rows per sequence
    int act;                    // NT: 0 none, 1 exact GELU (torch.nn.functional.gelu default)
    int ksplit;                 // TN: the contraction rows are cut into `ksplit` ranges, one workgroup each,
                                // partial results in C[split][taps][N][K] (launch_sum_slabs adds them up)
};

void launch_sgemm_nt(const SgArgs &a, hipStream_t st);
void launch_sgemm_nn(const SgArgs &a, hipStream_t st);  // conv: shift(tap) = (taps-1) - tap (transposed conv)
void launch_sgemm_tn(const SgArgs &a, hipStream_t st);
int sgemm_tn_splits(int Mc);  // upper bound of the ranges a launch_sgemm_tn caller asks for (sizes the slab buffer)
int sgemm_tn_splits_for(int Mc, int N, int K, int taps);  // the ranges to ask for in one GEMM (<= sgemm_tn_splits(Mc))
// out[i] = sum_s slabs[s][i], i < n (n % 4 == 0), fixed order
void launch_sum_slabs(const float *slabs, float *out, long n, int ns, hipStream_t st);
// conv weight [out][in][tap] (torch Conv1d) -> [tap][out][in]; and back (gradient)
void launch_pack_conv_w(const float *w, float *wp, int out_c, int in_c, int taps, hipStream_t st);
void launch_unpack_conv_w(const float *wp, float *w, int out_c, int in_c, int taps, hipStream_t st);
// out[n] = sum_m Y[m,n] (fixed order)
void launch_colsum(const float *Y, long ldy, int M, int N, float *out, float *scratch, hipStream_t st);
size_t colsum_scratch_floats(int M, int N);
int colsum_slabs(int M);  // slabs stage 1 leaves in `scratch` ([slab][N])
// two tensors of one shape (both strides ldy) in the same two launches
void launch_colsum_pair(const float *Y0, const float *Y1, long ldy, int M, int N, float *out0, float *out1, float *scratch, hipStream_t st);
// stage 1 only: the slabs are summed by launch_wgrad_finish
void launch_colsum_stage1(const float *Y, long ldy, int M, int N, float *scratch, hipStream_t st);
// dW = sum of `ns` split-K slabs (taps > 1: packed [tap][out][in] -> torch's [out][in][tap]) and db = sum of the colsum
// slabs in `cs` (db NULL: none), one launch
void launch_wgrad_finish(const float *slabs, int ns, float *dW, int out_c, int in_c, int taps, const float *cs, int nslab, int N,
                         float *db, hipStream_t st);
