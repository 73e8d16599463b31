// kernels.hpp — host-side launcher declarations shared by the engine's translation units.
#pragma once
#include "common.hpp"
#include "../../include/rnnt_engine.h"  // rnnt_conv_predictor_params (DecLoopArgs)
#include "lab/rnnt_engine_lab.h"       // variant bits of the diagnostic library (refused by the product library)

// ---- engine.hip: sets the thread-local error message, returns `code`
int engine_fail(int code, const char *fmt, ...);
// fills / copies as ordinary kernels on the caller's stream (in place of hipMemsetAsync / hipMemcpyAsync):
// a captured call is then a chain of kernel nodes only — with memset nodes in the chain, replays of a
// whole-pipeline HIP graph ran kernels against stale counters (ROCm 7.2; tests/test_gpu_parity.py::
// test_fused_call_is_hip_graph_capturable).  `bytes` and `p` multiples of 4.
void launch_fill32(void *p, unsigned value, size_t bytes, hipStream_t st);
void launch_copy_bytes(void *dst, const void *src, size_t bytes, hipStream_t st);

// ---- lattice.hip
void launch_logsoftmax_gather(const float *logits, const int32_t *targets,
                              const int32_t *logit_lens, const int32_t *target_lens,
                              float *denom_s, float *lpb_s, float *lpe_s, int B, int T, int U1,
                              int V, int D, int blank, hipStream_t st);
void launch_lattice(const float *lpb_s, const float *lpe_s, double *alpha_s, double *beta_s,
                    const int32_t *logit_lens, const int32_t *target_lens, float *costs, int B,
                    int U1, int D, unsigned *err /* one word of workspace */, hipStream_t st);
void launch_coef(const double *alpha_s, const double *beta_s, const float *denom_s,
                 const float *lpb_s, const float *lpe_s, const int32_t *targets,
                 const int32_t *logit_lens, const int32_t *target_lens, CellCoef *coef, int B,
                 int T, int U1, int D, float scale, hipStream_t st);
void launch_grad_logits(const float *logits, const CellCoef *coef, float *grad, long nrows,
                        int V, int blank, float clamp, hipStream_t st);

// ---- joint_fwd.hip
struct JointFwdArgs {
    const float *enc;   // [B,T,H], h-stride 1
    long enc_sb, enc_st;
    const float *pred;  // [B,U1,H] contiguous
    const float *wpack; // packed W (pack_w_fwd)
    const float *hidden; // [rows,H] tanh(enc+pred) (fused path), else NULL
    int make_hidden;     // 1: the forward kernel itself fills `hidden` for its tile (no k_make_hidden pass)
    const float *bias;  // [V]
    const int32_t *targets, *logit_lens, *target_lens;  // all NULL for the plain joint
    float *logits;      // [B,T,U1,V]
    float *denom_s, *lpb_s, *lpe_s;  // skewed [B,D,U1]; NULL for the plain joint
    int B, T, U1, H, V, D, blank;
    int flags;  // bit0: non-temporal logits stores
    unsigned *counter;  // one zeroable word: tile counter of the persistent forward (NULL: one workgroup per tile)
    int n_cu;           // compute units
    unsigned long long *debug;  // diagnostic stamp buffer (RNNT_STAMPS builds), else NULL
};
size_t wpack_floats(int H, int V);
void launch_pack_w_fwd(const float *W, float *wpack, int H, int V, hipStream_t st);
void launch_joint_fwd(const JointFwdArgs &a, hipStream_t st);
int fwd_occupancy(int with_loss);
void launch_zero_dead_hidden(float *hidden, const int32_t *logit_lens, const int32_t *target_lens, int B, int T,
                             int U1, int H, hipStream_t st);
void launch_copy_enc(const float *enc, long sb, long st_, long sh, float *dst, int B, int T, int H,
                     hipStream_t st);

// ---- joint_bwd.hip
struct JointBwdArgs {
    const float *enc; long enc_sb, enc_st;
    const float *pred;
    const float *W;       // [V,H] natural layout
    const float *logits;  // [rows_pad,V]: logits, overwritten by G (k_make_g)
    float *hidden;        // [rows_pad,H] tanh(enc+pred) (k_make_hidden)
    long rows_pad;        // B*T*U1 rounded up to a multiple of 16 (zero rows)
    const CellCoef *coef; // [B,T,U1]
    const int32_t *logit_lens, *target_lens;
    float *slab_enc;   // [n_ublk][B,T,H]
    float *slab_pred;  // [n_ttile][B,U1,H]
    float *slab_w;     // [n_split][V,H]
    float *slab_b;     // [n_split][V]
    float *grad_enc, *grad_pred, *grad_W, *grad_bias;
    int B, T, U1, H, V, blank;
    int n_ublk, n_ttile, n_split;
    unsigned *counter;  // 8 x 64 zeroed bytes: per-XCD work-item counters of the persistent kernels
    long *dw_tab;       // dW live-granule region (launch_dw_list): [0] live count, block offsets, list
    const int *dw_list; // set by launch_dw: the list inside that region
    int n_cu;           // compute units (grid size of the persistent kernels)
    int flags;          // bit 4 (16): G is produced by k_dhidden_gen; others: experiment switches
    int gen_bu;         // u width of k_dhidden_gen's tiles (16, or 8 for short targets); dEnc slabs of columns < pred_split_col
    int pred_split_col; // dPred slabs: columns < this come in 8-row t tiles (k_dhidden_gen, bf16 route), the rest in 4-row tiles (k_dhidden)
    unsigned long long *debug;  // diagnostic stamp buffer (RNNT_STAMPS builds), else NULL
};
void launch_dhidden(const JointBwdArgs &a, hipStream_t st);
void launch_dw(const JointBwdArgs &a, hipStream_t st);
int dw_tiles(int H, int V);  // workgroup tiles per split of k_dw
void launch_dw_table(const int32_t *logit_lens, int B, int T, int U1, int gran, long *tab, hipStream_t st);
size_t dw_list_bytes(int B, int T, int U1, int gran);  // bytes of the live-granule region of launch_dw
void launch_dhidden_reduce(const JointBwdArgs &a, hipStream_t st);
void launch_dw_reduce(const JointBwdArgs &a, hipStream_t st);
void launch_make_hidden(const JointBwdArgs &a, hipStream_t st);
void launch_make_g(const JointBwdArgs &a, hipStream_t st);
bool dhidden_gen_ok(int H, int V, int U1);
int dhidden_gen_groups(int H);  // 512-column groups on the tile kernel (the rest: persistent k_dhidden)
int dhidden_gen_bu(int T, int U1);  // 16 or 8: the tile form that pads the lattice least  // k_dhidden_gen (G produced inside the dHidden GEMM) applies

// ---- bf16.hip (RNNT_DTYPE_BF16 route: bf16 GEMM operands, fp32 accumulate / logits / loss)
struct Bf16Args {
    const float *enc; long enc_sb, enc_st;
    const float *pred;
    const float *W;      // [V,H] fp32 (re-packed to bf16 fragment order every call)
    const float *bias;
    unsigned short *hidden;  // bf16 [rows_alloc,H] tanh(enc+pred)
    void *wpack_fwd, *wpack_dh;
    unsigned short *logits;  // fp16 [rows_alloc,V]; G (bf16) overwrites each row in place
    const CellCoef *coef;
    const int32_t *targets, *logit_lens, *target_lens;
    float *denom_s, *lpb_s, *lpe_s;  // skewed [B,D,U1] softmax statistics (forward epilogue)
    int D;
    float *slab_enc, *slab_pred, *slab_w, *slab_b;
    long rows_alloc;     // multiple of 128, >= rows_pad + 96; rows >= B*T*U1 are zero
    long rows_pad;       // K extent of the dW GEMM (multiple of 32)
    int B, T, U1, H, V, blank;
    int n_ublk, n_split;
    int flags;  // experiment switches (bits 8..: 256 no stores, 512 no statistics, 1024 no MFMA)
    long *dw_tab;  // 2B+2 longs: live-row table of k_dw_bf16 (k_dw_table, 32-cell granules)
    unsigned long long *debug;  // diagnostic stamp buffer (RNNT_STAMPS builds), else NULL
    int *dw_prog;  // [n_split][16] progress words of k_dw_bf16's tiles (zeroed by its launcher), or NULL
};
size_t bf16_wpack_fwd_bytes(int H, int V);
size_t bf16_wpack_dh_bytes(int H, int V);
void launch_bf16_producers(const Bf16Args &a, hipStream_t st);
void launch_joint_fwd_bf16(const Bf16Args &a, hipStream_t st);
void launch_dhidden_bf16(const Bf16Args &a, hipStream_t st);
void launch_dw_bf16(const Bf16Args &a, hipStream_t st);

// ---- x3.hip (RNNT_DTYPE_F32_BF16X3 route: fp32-accurate products as six bf16 MFMA products of 3-way split operands)
struct X3Args {
    const float *enc; long enc_sb, enc_st;
    const float *pred;
    const float *W;      // [V,H] fp32 (split and re-packed in fragment order every call)
    const float *bias;
    unsigned short *hidden;  // bf16 planes [3][rows_alloc][H] of tanh(enc+pred): hi, mid, lo
    long plane_stride;       // elements between two planes of `hidden` (= rows_alloc * H)
    void *wpack_fwd, *wpack_dh;
    float *logits;           // fp32 [rows_alloc,V]; G's hi | mid planes overwrite each 32-wide chunk in place
    unsigned short *g_lo;    // bf16 [rows_alloc,V]: lo plane of G
    const CellCoef *coef;
    const int32_t *targets, *logit_lens, *target_lens;
    float *denom_s, *lpb_s, *lpe_s;  // skewed [B,D,U1] softmax statistics (forward epilogue)
    int D;
    float *slab_enc, *slab_pred, *slab_w, *slab_b;
    long rows_alloc;     // multiple of 128, >= rows_pad + 96; rows >= B*T*U1 are zero
    long rows_pad;       // K extent of the dW GEMM (multiple of 32)
    int B, T, U1, H, V, blank;
    int n_ublk, n_split;
    int n_ublk16;        // u blocks of k_dhidden_x3 (16 wide)
    int flags;
    long *dw_tab;        // 2B+2 longs: live-row table of k_dw_x3 (k_dw_table, 32-cell granules)
    unsigned *counter;   // zeroable word: tile counter of the persistent forward
    int n_cu;
    unsigned long long *debug;  // diagnostic stamp buffer (RNNT_STAMPS builds), else NULL
    // RNNT_DTYPE_F32_F16X2 (x2.hip; `hidden` then holds two fp16 planes, G two fp16 planes in place of the logits, no g_lo):
    float g_scale;              // power of two with |g_scale G| <= 2^13 (from grad_scale)
    float dw_rescale, db_rescale;  // 1 / (g_scale 2^14), 1 / g_scale
    const float *scales;        // device: {s_W, 1 / s_W} (k_x2_wscale, every call)
    int *dw_prog;               // [n_split][16] progress words of k_dw_x2's tiles (zeroed by its launcher), or NULL
    float *ep_enc, *ep_pred;    // exp(2 enc) [B][H/16][T][16], exp(2 pred) [B][H/16][U1][16] (k_x2_make_ep, every call)
    unsigned *ep_flag;          // device word: != 0 when an input lies outside the factored tanh's range (the exact forward runs)
};
bool x3_fwd_ok(int U1, int H, int V);      // the bf16x3 forward kernel covers this shape (else: the fp32 route's)
bool x3_dhidden_ok(int U1, int H, int V);  // likewise k_dhidden_x3
size_t x3_wpack_fwd_bytes(int H, int V);
size_t x3_wpack_dh_bytes(int H, int V);
void launch_x3_make_hidden(const X3Args &a, hipStream_t st);
void launch_x3_split_g(const X3Args &a, hipStream_t st);
void launch_x3_zero_padding(const X3Args &a, int what, hipStream_t st);
void launch_x3_pack_w(const X3Args &a, hipStream_t st);
void launch_joint_fwd_x3(const X3Args &a, hipStream_t st);   // one 512-register wave per SIMD (default)
bool x3_fwd_d_ok(int U1, int H, int V);
void launch_joint_fwd_x3z(const X3Args &a, hipStream_t st);  // one wave per SIMD, 2 M tiles per wave, 256 x 256 tiles, A in registers (RNNT_VARIANT_X3_FWD_Z)
void launch_joint_fwd_x3d(const X3Args &a, int nw, hipStream_t st);  // two waves per SIMD, A in registers (RNNT_VARIANT_X3_FWD_2WG / _8W)
void launch_dhidden_x3(const X3Args &a, hipStream_t st);
void launch_dw_x3(const X3Args &a, hipStream_t st);   // v_mfma_f32_32x32x16_bf16, six products per k-step (default)
void launch_dw_x3p(const X3Args &a, hipStream_t st);
// RNNT_DTYPE_F32_F16X2 (x2.hip): the bf16x3 route's stages on two fp16 planes and three products
bool x2_fwd_ok(int U1, int H, int V);
bool x2_dhidden_ok(int U1, int H, int V);
size_t x2_wpack_fwd_bytes(int H, int V);
size_t x2_wpack_dh_bytes(int H, int V);
void launch_x2_make_hidden(const X3Args &a, hipStream_t st);
void launch_x2_split_g(const X3Args &a, hipStream_t st);
void launch_x2_zero_padding(const X3Args &a, int what, hipStream_t st);
void launch_x2_pack_w(const X3Args &a, float *scales, hipStream_t st);  // s_W from max |W|, then both packs
void launch_x2_make_ep(const X3Args &a, hipStream_t st);               // exp(2 enc), exp(2 pred) in k-step-major layout + the range flag
size_t x2_ep_bytes(int B, int T, int U1, int H);
void launch_joint_fwd_x2(const X3Args &a, hipStream_t st);   // one 512-register wave per SIMD
bool x2_fwd_d_ok(int U1, int H, int V);
void launch_joint_fwd_x2d(const X3Args &a, hipStream_t st);  // two 4-wave workgroups per CU, A in registers (RNNT_VARIANT_X2_FWD_2WG)
void launch_dhidden_x2(const X3Args &a, hipStream_t st);
// the joint's input projections (audio_ln / text_ln) and their backward on the f16x2 pipes (x2.hip, round 5)
size_t x2_linear_ws_bytes(int M, int K, int N, bool bwd);
bool x2_linear_ok(int M, int K, int N);
void launch_linear_x2_fwd(const float *x, long ldx, const float *W, const float *bias, int M, int K, int N, float *y, void *ws, hipStream_t st);
void launch_linear_x2_bwd(const float *x, long ldx, const float *W, const float *dy, int M, int K, int N, float *dx, float *dW, float *db, void *ws,
                          hipStream_t st);
int x2_dw_tiles(int H, int V);  // workgroup tiles per split of launch_dw_x2 (k_dw_x2 / k_dw_x2m)
void launch_dw_x2(const X3Args &a, hipStream_t st, bool build_table = true, bool zero_prog = true);  // k_dw_x2<4> (k_dw_x2<4, true> when H % 256 == 128); -DRNNT_LAB builds: also k_dw_x2<8> / k_dw_x2p behind RNNT_VARIANT_X2_DW_8W / _P16

// ---- decode.hip
void launch_scan_logits(const float *enc, long enc_st, const float *pred, const float *W, const float *bias,
                        float *logits, int K, int H, int V, hipStream_t st);
void launch_argmax_scan(const float *logits, int K, int V, int blank, int t0, int32_t *out, hipStream_t st);

// device-resident greedy decode (decode.hip)
struct DecLoopArgs {
    const float *frames; long frame_stride; int T;  // [T,H] audio frames (after audio_ln), h-stride 1
    rnnt_conv_predictor_params p; int S, E, O; float ln_in_eps, ln_eps;  // eps of input_layer_norm / output_layer_norm
    const float *text_W, *text_b;                   // joint.text_ln [H,O], [H] or NULL (then O == H)
    const float *W, *bias; int H, V, blank;         // joint_ln
    int max_length, max_per_frame, scan_frames, iterations;
    int init;              // 1: initialise state, rings and weight packs first; 0: continue a decode in progress
    int32_t *host_flag;    // device pointer of a mapped host word set to 1 when the loop ends, or NULL
    int32_t *state, *tokens;
    void *workspace;
    const void *tables = nullptr;  // launch_dec_persist: the model's tables (launch_dec_build_tables), or NULL: built in the workspace
};
size_t dec_loop_workspace_floats(int H, int V, int E, int O, int nframes);
void launch_dec_loop(const DecLoopArgs &a, hipStream_t st);
// the same loop as ONE persistent launch (decode.hip, k_dec_persist): granule buffers, tables and folded matrices live in the workspace
struct DecPersistArgs {
    unsigned long long *g2g, *zg, *sg, *cg;         // hand-off granules: g2 [E] | z or q [H] | z's (mean, M2) per workgroup | candidates [2][16][G]
    const float *Eenc; int T;                        // exp(2 frames) [T,H]
    const float *A0, *A1, *A2, *conv1_b;             // conv1 as tables: tap j's row of symbol s at A_j + 3 E s
    const float *wp2, *conv2_b;                      // conv2 pack [5][E][E]
    const float *Wl, *bl;                            // linear [O,E], [O]
    const float *M, *dvec, *rvec, *cvec;             // with text_ln: M [H,E], d, r, c [H]
    const float *gamma, *beta; float eps;            // output LayerNorm
    const float *W, *bias;                           // joint_ln [V,H], [V]
    int S, E, O, H, V, blank, max_length, max_per_frame, has_text, max_iters;
    int32_t *state, *tokens, *host_flag;
};
int dec_persist_groups(int V);
const char *dec_persist_refusal(int T, int S, int E, int O, int H, int V, int has_text);  // NULL: supported
size_t dec_persist_workspace_floats(int T, int S, int E, int O, int H, int V, int has_text);
size_t dec_persist_lds_bytes(int E);
int launch_dec_persist(const DecLoopArgs &a, hipStream_t st);  // hipSuccess, or why the kernel's LDS limit could not be raised (nothing launched);  // scan_frames / iterations / init of `a` are not used
size_t dec_tables_floats(int S, int E, int O, int H, int has_text);
void launch_dec_build_tables(const rnnt_conv_predictor_params &p, int S, int E, int O, float ln_in_eps, const float *text_W, const float *text_b, int H,
                             float *tables, hipStream_t st);
