// Variant bits of the diagnostic library (build_variants/lab/librnnt_engine_lab.so, tools/build_lab.sh; -DRNNT_LAB): kernels that were
// built, parity-checked and measured equal to or slower than the shipped ones (DESIGN_APPENDIX.md A3-A5).  Not part of the public C ABI
// (include/rnnt_engine.h): librnnt_engine.so refuses every bit of RNNT_VARIANT_LAB_MASK.
#pragma once
// RNNT_DTYPE_F32_BF16X3: the forward in its two-waves-per-SIMD forms (k_joint_fwd_x3d: each wave owns 32 rows x 256 columns, the A operand
// never leaves its registers) and as k_joint_fwd_x3z (one wave per SIMD, two M tiles, 256 x 256 tiles)
#define RNNT_VARIANT_X3_FWD_2WG 16384    // two 4-wave workgroups per CU, 128-cell tiles
#define RNNT_VARIANT_X3_FWD_8W 65536     // one 8-wave workgroup per CU, 256-cell tiles
#define RNNT_VARIANT_X3_FWD_Z 262144
#define RNNT_VARIANT_X3_DW_P16 131072    // dW on v_mfma_f32_16x16x32_bf16, two of the six products per MFMA (k_dw_x3p)
// RNNT_DTYPE_F32_F16X2
#define RNNT_VARIANT_X2_FWD_2WG 1048576  // the forward as two 4-wave workgroups per CU (k_joint_fwd_x2d: A in registers, 256-column passes)
#define RNNT_VARIANT_X2_DW_P16 2097152   // dW on v_mfma_f32_16x16x32_f16 (k_dw_x2p: a k = 32 MFMA spans two 16-cell ring stages)
#define RNNT_VARIANT_X2_DW_8W 524288     // dW as 8 waves per workgroup (two per SIMD, k_dw_x2<8>) instead of the default 4
static_assert(((RNNT_VARIANT_X3_FWD_2WG | RNNT_VARIANT_X3_FWD_8W | RNNT_VARIANT_X3_FWD_Z | RNNT_VARIANT_X3_DW_P16 | RNNT_VARIANT_X2_FWD_2WG |
                RNNT_VARIANT_X2_DW_P16 | RNNT_VARIANT_X2_DW_8W) & ~RNNT_VARIANT_LAB_MASK) == 0, "lab bits live inside RNNT_VARIANT_LAB_MASK");
