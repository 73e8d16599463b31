// common.hpp — shared device helpers for the gfx950 RNN-T engine kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RNNT_NEG_INF (-__builtin_inff())
// Ablation switches (rnnt_engine_set_flags) exist only in diagnostic builds
// (make EXTRA=-DRNNT_ABLATE, used by tools/exp_*.py); the shipped kernels carry none.
#ifdef RNNT_ABLATE
#define RNNT_XP(flags, bit) ((flags) & (bit))
#else
#define RNNT_XP(flags, bit) (0)
#endif
#define RNNT_LOG2E 1.4426950408889634f

// Skewed ("anti-diagonal major") lattice layout used by every per-cell work array the
// sweep touches: cell (b,t,u) lives at [b][d = t+u][u], D = T+U1-1 diagonals per utterance,
// so that one anti-diagonal is contiguous in HBM (coalesced sweep loads/stores).
__device__ __forceinline__ long skew_index(int b, int t, int u, int D, int U1)
{
    return ((long)b * D + (t + u)) * U1 + u;
}

// Utterance lengths as every kernel reads them: clamped into the lattice (1 <= T_b <= T,
// 0 <= U_b <= U1-1), so a bad length from a caller that skipped the host-side checks
// (check_lengths=False, or the C ABI called directly) can never index outside the per-utterance
// arrays or the LDS mailboxes; it yields the loss of the clamped lattice instead of a fault.
__device__ __forceinline__ int len_t(const int32_t *logit_lens, int b, int T)
{
    const int v = logit_lens[b];
    return v < 1 ? 1 : (v > T ? T : v);
}
__device__ __forceinline__ int len_u(const int32_t *target_lens, int b, int U1)
{
    const int v = target_lens[b];
    return v < 0 ? 0 : (v > U1 - 1 ? U1 - 1 : v);
}

// tanh via one v_exp_f32 + one v_rcp_f32: 1 - 2/(exp(2x)+1).  Saturates correctly at
// +-inf; absolute error <~3e-7 (the result feeds a dot product, absolute error matters).
__device__ __forceinline__ float fast_tanh(float x)
{
    float e = __builtin_amdgcn_exp2f(x * (2.0f * RNNT_LOG2E));
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// 32-lane (half-wave) butterfly reductions; xor masks < 32 never cross the wave halves,
// matching the 32x32 MFMA accumulator layout where lanes 0-31 / 32-63 hold different rows.
__device__ __forceinline__ float half_max(float v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ float half_sum(float v)
{
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// The same reductions on the DPP crossbar (VALU rate, no LDS round trip: __shfl_xor lowers to
// ds_bpermute_b32): quad_perm xor 1, xor 2, row_ror 4, 8 give every lane its 16-lane row's
// result; row_bcast15 into rows 1 and 3 completes the 32-lane half in lanes 16-31 / 48-63;
// half_bcast() hands that to every lane of the half.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float half_bcast(float v, int half)
{
    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return half ? hi : lo;
}
__device__ __forceinline__ float half_max_dpp(float v, int half)
{
    v = fmaxf(v, dpp_mov<0xB1, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x4E, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x124, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x128, 0xf>(v, v));
    v = fmaxf(v, dpp_mov<0x142, 0xa>(RNNT_NEG_INF, v));
    return half_bcast(v, half);
}
__device__ __forceinline__ float half_sum_dpp(float v, int half)  // valid in lanes 16-31 / 48-63
{
    v += dpp_mov<0xB1, 0xf>(v, v);
    v += dpp_mov<0x4E, 0xf>(v, v);
    v += dpp_mov<0x124, 0xf>(v, v);
    v += dpp_mov<0x128, 0xf>(v, v);
    v += dpp_mov<0x142, 0xa>(0.f, v);
    return v;
}

// Packed per-cell gradient coefficients written by k_coef, read by the backward GEMMs:
//   G[v] = exp2(logit[v]*log2e + c1) - (v==blank)*sb - (v==y)*se
struct __attribute__((aligned(16))) CellCoef {
    float c1;  // (alpha+beta+cost-denom+log(scale))*log2e, -inf for cells outside the lattice
    float sb;  // blank-transition occupancy * scale
    float se;  // emit-transition occupancy * scale
    int y;     // emit label of this cell (targets[b][u]) or -1
};
