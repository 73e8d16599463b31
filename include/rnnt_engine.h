/*
 * rnnt_engine.h — C ABI of librnnt_engine.so, the MI355X (gfx950) RNN-T joint + transducer
 * loss engine.  Plain pointers and sizes only; no torch types cross this boundary.
 *
 * The reference (jakepoz/rnnt) has NO native/FFI layer: its boundary for this path is the
 * Python pair
 *     rnnt.joint.JointNetwork.forward            /root/reference/rnnt/joint.py:25-39
 *     torchaudio.functional.rnnt_loss(...)       /root/reference/rnnt/model.py:35-41
 * followed by loss.backward()                    /root/reference/rnnt/train.py:133-134.
 * Each entry point below names the reference interface it replaces.  The Python host side
 * (rnnt_amd/engine.py) binds these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller owns all
 *     memory (inputs, outputs, workspace); the library never allocates or frees device
 *     memory and keeps no pointer after a call returns;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and returns without synchronising;
 *   - return value 0 = success, negative = error (RNNT_ERR_*); the message is available
 *     from rnnt_engine_last_error() (thread-local);
 *   - lattice layout: logits [B,T,U1,V] row-major contiguous, U1 = max target length + 1;
 *     targets [B,U1-1] int32, blank never appears in targets; lengths int32 [B];
 *   - precondition on the lengths (torchaudio checks the same on the host): 1 <= logit_lens[b] <= T,
 *     0 <= target_lens[b] <= U1-1.  The kernels CLAMP what they read into those ranges, so a
 *     violated precondition gives the loss of the clamped lattice, never an out-of-bounds access;
 *   - the library keeps no mutable state besides the thread-local error string; the device the
 *     pointers live on must be the calling thread's current HIP device.
 */
#ifndef RNNT_ENGINE_H
#define RNNT_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNNT_ENGINE_VERSION 3

#define RNNT_DTYPE_F32 0 /* fp32 in, fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 out */
/* BASELINE config 3 ("bf16"): pointers stay fp32 at this boundary (inputs, parameters, costs,
 * gradients); the operands of the three GEMMs — tanh(enc+pred), W and the logits gradient —
 * are rounded to bf16 (nearest-even) and multiplied by v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation; the logits are kept in fp16 (workspace only; nearest-even, the loss is computed
 * from the stored values); log-softmax, the lattice and every reduction stay fp32/fp64.  Fused
 * entry only (rnnt_engine_joint_loss_fwd_bwd, rnnt_engine_run_stage, the workspace queries);
 * needs H % 128 == 0, V % 128 == 0. */
#define RNNT_DTYPE_BF16 1
/* fp32-ACCURATE arithmetic on the bf16 matrix pipes ("bf16x3"): same boundary, same 1e-4 parity bar as
 * RNNT_DTYPE_F32 — every operand of the three GEMMs is split once into three bf16 pieces (hi + mid + lo
 * = the fp32 value to 2^-25) and each product is the sum of six bf16 MFMA products with fp32 accumulation
 * (rnnt_amd/csrc/x3.hip); logits stay fp32, log-softmax / lattice / reductions are the fp32 route's.
 * Measured error against fp64 is below the fp32 MFMA route's (tools/bf16x3_accuracy.py, tests).  Fused
 * entry only; needs H % 128 == 0, V % 128 == 0 (the Python host side zero-pads other shapes). */
#define RNNT_DTYPE_F32_BF16X3 2
/* fp32-CLASS arithmetic on the fp16 matrix pipes at HALF the matrix work of RNNT_DTYPE_F32_BF16X3 ("f16x2", round 4): the
 * three bf16x3 GEMM kernels turned out power-bound (the chip holds 1.7-2.1 GHz under their MFMA stream), so the remaining
 * lever is fewer products.  Every operand of the three GEMMs is scaled by a power of two into fp16's range and split into
 * TWO fp16 pieces (hi + mid = the value to 2^-22: 11 + 11 significant bits); a product is the sum of THREE fp16 MFMA
 * products (hi.hi + hi.mid + mid.hi, fp32 accumulation) — rnnt_amd/csrc/x2.hip.  Same boundary, same 1e-4 parity bar, same
 * stages, variants and workspace objects as the bf16x3 route (no third plane); measured error against fp64: tests/test_x2_gpu.py.
 * Fused entry only; needs H % 128 == 0, V % 128 == 0. */
#define RNNT_DTYPE_F32_F16X2 3

#define RNNT_OK 0
#define RNNT_ERR_INVALID_ARG (-1) /* null pointer, non-positive dim, bad blank ...        */
#define RNNT_ERR_UNSUPPORTED (-2) /* dims the kernels do not cover (H%4, V%4, U1>1024 ...) */
#define RNNT_ERR_WORKSPACE (-3)   /* workspace too small                                   */
#define RNNT_ERR_LAUNCH (-4)      /* HIP reported an error at launch                       */

/* Library version (RNNT_ENGINE_VERSION it was built with). */
int rnnt_engine_version(void);

/* Diagnostic builds only (make EXTRA=-DRNNT_ABLATE / -DRNNT_STAMPS): process-wide ablation switches
 * of tools/exp_*.py and the device buffer of the in-kernel time stamps.  In the shipped library
 * both are no-ops (set_flags returns 0) — it has no global mutable state; alternative kernel
 * variants are chosen per call through rnnt_engine_run_stages. */
int rnnt_engine_set_flags(int flags);
void rnnt_engine_set_debug(void *buf);

/* Kernel variants a caller may ask for per call (rnnt_engine_run_stages).  Every variant multiplies
 * the same numbers in the same order as the default kernels: results are bit-identical. */
#define RNNT_VARIANT_SEPARATE_G 32          /* G by its own pass (k_make_g) + the persistent dHidden kernel */
#define RNNT_VARIANT_SEPARATE_HIDDEN 64     /* hidden by its own pass instead of the forward prologue       */
#define RNNT_VARIANT_FWD_LDS_RING 128       /* forward main loop: W through an LDS-DMA ring                 */
#define RNNT_VARIANT_FWD_ONE_WG_PER_TILE 256 /* forward: one workgroup per tile instead of persistent ones   */
/* RNNT_DTYPE_F32_BF16X3 only: run one stage on the fp32 route's kernel instead of the bf16x3 one (same data
 * layout downstream, plain splitting kernels in between) — how each bf16x3 kernel is checked in isolation.
 * NOT bit-identical to the default bf16x3 kernels (different summation), same tolerance. */
#define RNNT_VARIANT_X3_FP32_FWD 4096       /* forward GEMM + hidden by the fp32 kernel, then k_x3_make_hidden */
#define RNNT_VARIANT_X3_FP32_DH 8192        /* dHidden + G by the fp32 kernels, then k_x3_split_g (needs _FWD too) */
/* Bits 14 and up name kernels of the DIAGNOSTIC library only (rnnt_amd/csrc/lab/rnnt_engine_lab.h, tools/build_lab.sh):
 * librnnt_engine.so answers them with RNNT_ERR_UNSUPPORTED. */
#define RNNT_VARIANT_LAB_MASK 0x7fffc000

/* Diagnostic queries (0/1: predicted resident forward-kernel workgroups per CU). */
int rnnt_engine_debug_query(int what);

/* Message of the last error on the calling thread ("" if none). */
const char *rnnt_engine_last_error(void);

/* Bytes of device workspace rnnt_engine_joint_loss_fwd_bwd needs for these dims.  The workspace is
 * scratch: it need not be initialised (any bit pattern, NaNs included, is fine) and carries no
 * state from one call to the next. */
int rnnt_engine_workspace_bytes(int B, int T, int U1, int H, int V, int dtype, size_t *out);

/* Bytes of device workspace rnnt_engine_loss_fwd_bwd needs for these dims. */
int rnnt_engine_loss_workspace_bytes(int B, int T, int U1, int V, int dtype, size_t *out);

/* Bytes of device workspace rnnt_engine_joint_fwd needs for these dims. */
int rnnt_engine_joint_fwd_workspace_bytes(int B, int T, int U1, int H, int V, int dtype,
                                          size_t *out);

/*
 * logits[b,t,u,:] = tanh(enc[b,t,:] + pred[b,u,:]) @ W^T + bias
 * Replaces JointNetwork.forward's last three statements, reference rnnt/joint.py:32-39
 * (the optional audio_ln/text_ln projections of joint.py:26-30 stay with the caller).
 *   enc   [B,T,H] with element strides enc_strides[3] (the reference hands a permuted,
 *         non-contiguous view: rnnt/model.py:28); pred [B,U1,H] contiguous;
 *   W     [V,H] (torch.nn.Linear layout, joint.py:18); bias [V]; logits [B,T,U1,V] out.
 */
int rnnt_engine_joint_fwd(const void *enc, const int64_t enc_strides[3], const void *pred,
                          const void *W, const void *bias, int B, int T, int U1, int H, int V,
                          int dtype, void *logits, void *workspace, size_t ws_bytes,
                          void *stream);

/*
 * Transducer loss on given logits: per-utterance costs and d cost_b / d logits.
 * Replaces torchaudio.functional.rnnt_loss as called at reference rnnt/model.py:35-41
 * (fused log-softmax; `blank` already resolved to [0,V); clamp <= 0 means off).
 *   costs [B] out; grad_logits [B,T,U1,V] out or NULL (costs only).
 * Gradients are those of sum_b costs[b] (i.e. unscaled, as torchaudio stores them);
 * the caller applies the reduction / upstream factor.
 */
int rnnt_engine_loss_fwd_bwd(const void *logits, const int32_t *targets,
                             const int32_t *logit_lens, const int32_t *target_lens, int B,
                             int T, int U1, int V, int blank, float clamp, int dtype,
                             float *costs, void *grad_logits, void *workspace, size_t ws_bytes,
                             void *stream);

/*
 * Fused joint + transducer loss, forward AND backward in one call: everything between
 * `self.joint(...)` (reference rnnt/model.py:32) and the gradients loss.backward()
 * (rnnt/train.py:134) delivers to the joint's inputs and parameters.
 *   costs     [B]       per-utterance negative log-likelihood (fp32)
 *   grad_enc  [B,T,H]   contiguous, d(mean_b costs)/d enc     (reduction="mean", model.py:41)
 *   grad_pred [B,U1,H]  d(mean)/d pred
 *   grad_W    [V,H]     d(mean)/d W        grad_bias [V]  d(mean)/d bias
 * `grad_scale` multiplies every gradient (1/B for reduction="mean" on one device, 1/B_global
 * when the batch is sharded over ranks).  The (B,T,U1,V) logits live only in `workspace`.
 */
int rnnt_engine_joint_loss_fwd_bwd(const void *enc, const int64_t enc_strides[3],
                                   const void *pred, const void *W, const void *bias,
                                   const int32_t *targets, const int32_t *logit_lens,
                                   const int32_t *target_lens, int B, int T, int U1, int H,
                                   int V, int blank, float clamp, float grad_scale, int dtype,
                                   float *costs, void *grad_enc, void *grad_pred, void *grad_W,
                                   void *grad_bias, void *workspace, size_t ws_bytes,
                                   void *stream);

/*
 * Forward only: the per-utterance costs of the fused path and nothing else (no coefficient, dHidden
 * or dW kernels, no gradient buffers).  What `RNNTModel.forward` costs under torch.no_grad() —
 * the validation loss of reference rnnt/train.py:170-201 / rnnt/model.py:32-41.  Same workspace size
 * as rnnt_engine_joint_loss_fwd_bwd.
 */
int rnnt_engine_joint_loss_fwd(const void *enc, const int64_t enc_strides[3], const void *pred,
                               const void *W, const void *bias, const int32_t *targets,
                               const int32_t *logit_lens, const int32_t *target_lens, int B, int T,
                               int U1, int H, int V, int blank, int dtype, float *costs,
                               void *workspace, size_t ws_bytes, void *stream);

/*
 * Backward of the UNFUSED joint: given d loss / d logits (any upstream gradient, [B,T,U1,V]
 * contiguous fp32) returns the gradients autograd sends through reference rnnt/joint.py:32-39
 * (joint_ln, tanh, the broadcast add): grad_enc [B,T,H], grad_pred [B,U1,H], grad_W [V,H],
 * grad_bias [V].  Serves callers that keep `joint(...)` and the loss as two calls (a maintainer
 * who swaps only rnnt/joint.py); fp32 only.
 */
int rnnt_engine_joint_bwd_workspace_bytes(int B, int T, int U1, int H, int V, int dtype, size_t *out);
int rnnt_engine_joint_bwd(const void *enc, const int64_t enc_strides[3], const void *pred,
                          const void *W, const void *grad_logits, int B, int T, int U1, int H, int V,
                          int dtype, void *grad_enc, void *grad_pred, void *grad_W, void *grad_bias,
                          void *workspace, size_t ws_bytes, void *stream);

/*
 * Greedy-decode scan (next-step row SURVEY.md 8f-2).  The reference's decode loop, rnnt/model.py:108-125,
 * evaluates joint.single_forward (rnnt/joint.py:44-55) for one audio frame at a time and syncs on
 * argmax(...).item() per frame.  This call evaluates frames t0 .. t0+nframes-1 (nframes <= 128) of
 * one utterance against ONE predictor state and reduces on the device:
 *   out[0] = first frame whose argmax is not `blank` (t0+nframes if every frame says blank)
 *   out[1] = that token (blank if none);   out[2+k] = argmax of frame t0+k (first index on ties)
 *   enc  [T,H] rows enc_stride_t apart, elements enc_stride_h apart (the permuted encoder view
 *        of rnnt/model.py:102 works), already projected by audio_ln when the model has one;
 *   pred [H] predictor output (after text_ln when present); W [V,H]; bias [V]; out int32[2+nframes].
 */
int rnnt_engine_greedy_scan_workspace_bytes(int nframes, int H, int V, size_t *out);
int rnnt_engine_greedy_scan(const void *enc, int64_t enc_stride_t, int64_t enc_stride_h,
                            const void *pred, const void *W, const void *bias, int t0, int nframes,
                            int H, int V, int blank, int32_t *out, void *workspace, size_t ws_bytes,
                            void *stream);

/*
 * Gradient-norm clip + AdamW step over a list of fp32 tensors (next-step row SURVEY.md 8f-4): what
 * follows loss.backward() in the reference's loop,
 *     total_norm = torch.nn.utils.clip_grad_norm_(params, clip)        rnnt/train.py:136
 *     optimizer.step()            torch.optim.AdamW(lr, betas, eps, weight_decay)   rnnt/train.py:164,
 *                                 rnnt/config/basic_sp_convjs_fullcausal.yaml:82-87
 * as multi-tensor kernels (pointer lists are HOST arrays of DEVICE pointers, packed into kernel
 * arguments; up to 40 tensors per launch).
 *   rnnt_engine_grad_norm   total_norm[0] (device) = 2-norm of all gradients, reduced in a fixed
 *                           order; workspace from rnnt_engine_grad_norm_workspace_bytes.
 *   rnnt_engine_adamw_step  decoupled weight decay, no amsgrad: p *= 1 - lr*wd; m += (1-b1)(g-m);
 *                           v = b2 v + (1-b2) g^2; p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps).
 *                           The hyper-parameters are doubles: their scalar arithmetic (1-b1, 1-b2,
 *                           1-lr*wd, the bias corrections) is done in double, as torch does with python
 *                           floats.  `step` = the update count including this one (>= 1).  When `total_norm`
 *                           (device scalar) is non-NULL and max_norm > 0, every gradient is first
 *                           scaled by min(1, max_norm / (total_norm + 1e-6)) — clip_grad_norm_'s
 *                           coefficient, read on the device: no host synchronisation between
 *                           backward and step; write_clipped_grads != 0 also stores the scaled
 *                           gradients (clip_grad_norm_ works in place).
 *   rnnt_engine_adamw_step_dev  the same update with the step count and the learning rate RESIDENT ON THE
 *                           DEVICE: `step_dev` (int64, the count of updates done so far) is incremented
 *                           by the call, `lr_dev` (float) is read when the kernels run, the three derived
 *                           scalars go through `hyper_dev` (float[4], scratch).  Nothing of the update is
 *                           baked into the launch arguments, so the call can be captured into a HIP graph
 *                           and replayed every iteration (an LR scheduler writes lr_dev between replays).
 */
int rnnt_engine_grad_norm_workspace_bytes(int n_tensors, const int64_t *numels, size_t *out);
int rnnt_engine_grad_norm(int n_tensors, const void *const *grads, const int64_t *numels,
                          float *total_norm, void *workspace, size_t ws_bytes, void *stream);
int rnnt_engine_adamw_step(int n_tensors, void *const *params, const void *const *grads,
                           void *const *exp_avg, void *const *exp_avg_sq, const int64_t *numels,
                           double lr, double beta1, double beta2, double eps, double weight_decay,
                           int64_t step, const float *total_norm, float max_norm,
                           int write_clipped_grads, void *stream);
int rnnt_engine_adamw_step_dev(int n_tensors, void *const *params, const void *const *grads,
                               void *const *exp_avg, void *const *exp_avg_sq, const int64_t *numels,
                               const float *lr_dev, double beta1, double beta2, double eps,
                               double weight_decay, int64_t *step_dev, float *hyper_dev,
                               const float *total_norm, float max_norm, int write_clipped_grads,
                               void *stream);

/*
 * ConvPredictor forward / backward (next-step row SURVEY.md 8f-3): reference rnnt/predictor.py:189-229
 * — embedding, LayerNorm, CausalConv1d k=3 + GELU + dropout, CausalConv1d k=5 + GELU + dropout
 * (rnnt/causalconv.py:9-32: left zero padding), Linear, LayerNorm — on rows (b,u) with channels
 * contiguous: the two permutes of predictor.py:216,225 are never materialised, each convolution is
 * one MFMA GEMM per tap.  All pointers fp32 device pointers, 16-byte aligned; E % 4 == 0, O % 4 == 0.
 *   ids [B,U1] int64 (as the reference passes them, rnnt/model.py:20-21); out [B,U1,O];
 *   keep1 / keep2 [B,U1,E] bytes: dropout keep masks drawn by the caller (NULL = eval mode), kept
 *   values are scaled by 1/(1-dropout_p);  ln_in_eps / ln_out_eps: the eps of input_layer_norm / output_layer_norm
 *   (torch.nn.LayerNorm's default: 1e-5 each);
 *   `saved`: caller-owned buffer (rnnt_engine_conv_predictor_saved_bytes) the forward fills and the
 *   backward of the SAME call reads; `g`: where each parameter's gradient is written (same field
 *   order as the parameters; every gradient is overwritten, not accumulated).
 */
typedef struct rnnt_conv_predictor_params {
    const float *embedding;           /* [S,E]                 embedding.weight           */
    const float *ln_in_w, *ln_in_b;   /* [E]                   input_layer_norm.*         */
    const float *conv1_w, *conv1_b;   /* [E,E,3], [E]          conv1.conv.* (Conv1d layout) */
    const float *conv2_w, *conv2_b;   /* [E,E,5], [E]          conv2.conv.*               */
    const float *linear_w, *linear_b; /* [O,E], [O]            linear.*                   */
    const float *ln_out_w, *ln_out_b; /* [O]                   output_layer_norm.*        */
} rnnt_conv_predictor_params;

int rnnt_engine_conv_predictor_saved_bytes(int B, int U1, int S, int E, int O, size_t *out);
int rnnt_engine_conv_predictor_fwd(const int64_t *ids, int B, int U1, int S, int E, int O,
                                   const rnnt_conv_predictor_params *p, const uint8_t *keep1,
                                   const uint8_t *keep2, float dropout_p, float ln_in_eps, float ln_out_eps, float *out,
                                   void *saved, size_t saved_bytes, void *stream);
int rnnt_engine_conv_predictor_bwd(const int64_t *ids, int B, int U1, int S, int E, int O,
                                   const rnnt_conv_predictor_params *p, const uint8_t *keep1,
                                   const uint8_t *keep2, float dropout_p, const float *grad_out,
                                   const rnnt_conv_predictor_params *g, void *saved, size_t saved_bytes,
                                   void *stream);

/*
 * Device-resident greedy decode of ONE utterance (next-step row SURVEY.md 8f-2, second half): the whole loop of
 * rnnt/model.py:108-125 — scan frames from t for the first non-blank argmax, append the token, at most `max_per_frame`
 * tokens per frame, re-run the predictor — with the stateless ConvPredictor of rnnt/predictor.py:189-229 (eval mode:
 * no dropout) evaluated incrementally on the device (its output frame is a function of the last 7 tokens).  The call
 * only ENQUEUES: (init != 0) state initialisation, then `iterations` times a fixed kernel sequence (0 = the upper bound
 * max_length + ceil(T / scan_frames) + 1) whose kernels return at once after the loop has ended.  A caller may enqueue the
 * bound in one call, or a chunk at a time (init = 1 first, then init = 0 with the same buffers) and stop when the loop is
 * over: `host_flag` (NULL, or one int32 of PINNED host memory the caller zeroed) is set to 1 by the device when the loop
 * ends and can be polled without a synchronisation.  The caller synchronises once and reads
 *   state  int32[8]: [0] t, [1] emitted, [2] ntok = decoded tokens, [3] done, [5] iterations that did work;
 *   tokens int32[max_length]: tokens[0] = blank (rnnt/model.py:100), tokens[1 .. ntok] = the decoded ids.
 * frames [T,H] fp32 audio frames, rows frame_stride apart, unit element stride (already projected by audio_ln when the
 * joint has one); text_W [H,O] / text_b [H]: joint.text_ln, or NULL when the joint adds the predictor output directly
 * (then O == H); W [V,H], bias [V]: joint_ln.  E, O <= 1024, E % 4 == 0, O % 4 == 0, H % 8 == 0, V % 4 == 0,
 * 1 <= scan_frames <= 128, max_length >= 2.
 */
int rnnt_engine_greedy_decode_workspace_bytes(int H, int V, int E, int O, int scan_frames, size_t *out);
int rnnt_engine_greedy_decode(const void *frames, int64_t frame_stride, int T, const rnnt_conv_predictor_params *p,
                              int S, int E, int O, float ln_in_eps, float ln_out_eps, const void *text_W, const void *text_b,
                              const void *W, const void *bias, int H, int V, int blank, int max_length,
                              int max_per_frame, int scan_frames, int iterations, int init, int32_t *host_flag,
                              int32_t *state, int32_t *tokens, void *workspace, size_t ws_bytes, void *stream);

/*
 * The same loop (rnnt/model.py:108-125 with the ConvPredictor of rnnt/predictor.py:189-229, eval mode) for one utterance as ONE
 * persistent launch: 16 - 128 workgroups stay resident for the whole utterance and hand scan candidates, the conv2 output and the
 * predictor's output to each other through tagged 8-byte words in the workspace (rnnt_amd/csrc/decode.hip, k_dec_persist), with
 * conv1 replaced by per-token table rows and joint.text_ln folded into the predictor's linear layer (tables and folded matrices
 * come from `tables`, below: the caller's buffer, or rebuilt from the parameters inside this call).  Same arguments, state and tokens
 * as rnnt_engine_greedy_decode without scan_frames / iterations / init (the block is 16 frames, the call is the whole decode);
 * state[5] = iterations, state[6] = workgroups, state[7] != 0: the loop did NOT decode (state and tokens are then not a decode;
 * nothing is raised on the stream) — 1..9: a hand-off never arrived within the bounded spin (~0.4 s: a workgroup was not resident)
 * and the loop gave up; 10 / 11: an audio frame / a text vector holds an entry beyond +-30 or a non-finite one, where the loop's
 * factored tanh(e + p) = 1 - 2 / (1 + exp 2e exp 2p) is not exact.  In every such case decode the utterance with
 * rnnt_engine_greedy_decode, which needs no residency and takes tanh of the sum.  The call only enqueues; the caller synchronises once.
 * RNNT_ERR_UNSUPPORTED (call rnnt_engine_greedy_decode instead): H % 64 != 0, H / E / O > 1024, S > 4096, T + max_length >= 2^20,
 * a device with fewer compute units than workgroups or with less LDS per workgroup than the kernel keeps (100-138 KB).  Token lists
 * equal rnnt_engine_greedy_decode's wherever the argmax is not a rounding-level tie (sums are associated differently).
 */
int rnnt_engine_greedy_decode_persistent_workspace_bytes(int T, int S, int E, int O, int H, int V, int has_text, size_t *out);
/* `tables`: NULL (the tables are rebuilt in the workspace by this call: ~0.13 ms) or a caller-owned buffer filled by
 * rnnt_engine_greedy_decode_build_tables for the SAME parameters (conv2's pack, the conv1 tap tables, the folded text_ln): build once per
 * set of weights, decode any number of utterances — concurrently on several streams too (the tables are only read). */
int rnnt_engine_greedy_decode_tables_bytes(int S, int E, int O, int H, int has_text, size_t *out);
int rnnt_engine_greedy_decode_build_tables(const rnnt_conv_predictor_params *p, int S, int E, int O, float ln_in_eps, const void *text_W,
                                           const void *text_b, int H, void *tables, size_t tables_bytes, void *stream);
int rnnt_engine_greedy_decode_persistent(const void *frames, int64_t frame_stride, int T, const rnnt_conv_predictor_params *p,
                                         int S, int E, int O, float ln_in_eps, float ln_out_eps, const void *text_W, const void *text_b,
                                         const void *W, const void *bias, int H, int V, int blank, int max_length,
                                         int max_per_frame, const void *tables, int32_t *host_flag, int32_t *state, int32_t *tokens,
                                         void *workspace, size_t ws_bytes, void *stream);

/*
 * y = x W^T + b and its backward as MFMA kernels: the joint's optional input projections
 * audio_ln / text_ln (next-step row SURVEY.md 8f-1; reference rnnt/joint.py:8-12,26-30).
 * x [M,K] with rows ldx floats apart, W [N,K] (torch.nn.Linear layout), y / dy [M,N] contiguous.
 * Backward: dW [N,K] = dy^T x, db [N] = column sums of dy (NULL: skipped), dx [M,K] = dy W (NULL:
 * skipped).  K % 4 == 0, N % 4 == 0.
 */
int rnnt_engine_linear_fwd(const float *x, int64_t ldx, const float *W, const float *bias, int M, int K,
                           int N, float *y, void *stream);
int rnnt_engine_linear_bwd_workspace_bytes(int M, int K, int N, size_t *out);
int rnnt_engine_linear_bwd(const float *x, int64_t ldx, const float *W, const float *dy, int M, int K,
                           int N, float *dx, float *dW, float *db, void *workspace, size_t ws_bytes,
                           void *stream);

/*
 * The same Linear layer (reference rnnt/joint.py:8-12,26-30: audio_ln / text_ln) on the f16x2 matrix pipes
 * (RNNT_DTYPE_F32_F16X2's arithmetic: three fp16 MFMA products of power-of-two-scaled, 2-way split operands, fp32
 * accumulation — the fp32 class of error): y = x W^T + b through the joint forward's pipeline as a plain GEMM, dx through
 * the same kernel on W^T, dW / db through the joint's dW kernel with dy and x as its two operands.  Operand scales are
 * found on the device every call.  Needs K % 128 == 0 and N % 128 == 0; x [M,K] rows ldx floats apart; W [N,K], dy [M,N],
 * y [M,N], dx [M,K] contiguous; bias / dx / db may be NULL.  `backward` selects the workspace of _bwd (W^T, operand planes,
 * split-K slabs) or of _fwd (the W pack).  Same ownership / stream / error rules as every other entry point.
 */
int rnnt_engine_linear_x2_workspace_bytes(int M, int K, int N, int backward, size_t *out);
int rnnt_engine_linear_x2_fwd(const float *x, int64_t ldx, const float *W, const float *bias, int M, int K,
                              int N, float *y, void *workspace, size_t ws_bytes, void *stream);
int rnnt_engine_linear_x2_bwd(const float *x, int64_t ldx, const float *W, const float *dy, int M, int K,
                              int N, float *dx, float *dW, float *db, void *workspace, size_t ws_bytes,
                              void *stream);

/*
 * Multi-GPU step of the path (the suggested export of 8b): ONE sum all-reduce, in
 * place, of `count` fp32 values — the flat [dW (V*H) | db (V) | loss] buffer of a batch-sharded
 * step — on the caller's RCCL communicator (`comm` is an ncclComm_t) and stream.  Stands where the
 * reference relies on DDP's gradient all-reduce (rnnt/train.py:28,68: backend "nccl" = RCCL on
 * ROCm).  The engine does not link RCCL: ncclAllReduce is resolved at call time from the RCCL the
 * process already has loaded (PyTorch's), else from librccl.so.1 on the loader path; it creates no
 * communicator and keeps none.  Returns RNNT_ERR_UNSUPPORTED when no RCCL can be found,
 * RNNT_ERR_LAUNCH with RCCL's message when the collective fails.
 */
int rnnt_engine_allreduce(void *buf, size_t count, void *comm, void *stream);

/*
 * Diagnostic view of the last fused call's intermediate buffers inside `workspace`
 * (offsets in bytes; valid for the dims given).  Used by tests and bench.py to time or
 * inspect single stages; not needed by training code.
 */
typedef struct rnnt_engine_ws_layout {
    size_t logits, hidden, denom_s, lpb_s, lpe_s, alpha_s, beta_s, coef, wpack, enc_copy;
    size_t slab_enc, slab_pred, slab_w, slab_b, counters, total, rows_pad;
    int n_ublk, n_ttile, n_split, D;
    size_t g_lo;          /* RNNT_DTYPE_F32_BF16X3: lo plane of G */
    size_t aux, aux_bytes; /* RNNT_DTYPE_F32_BF16X3: fp32 hidden + W pack of the RNNT_VARIANT_X3_FP32_* stages, placed BEHIND
                            * `total` (aux == total): only a call with such a variant needs total + aux_bytes */
    size_t ep;            /* RNNT_DTYPE_F32_F16X2: exp(2 enc) [B][H/16][T][16] then exp(2 pred) [B][H/16][U1][16], fp32 (0: none) */
} rnnt_engine_ws_layout;

int rnnt_engine_workspace_layout(int B, int T, int U1, int H, int V, int dtype,
                                 rnnt_engine_ws_layout *out);

/*
 * Run ONE stage of the fused pipeline on an already laid-out workspace (bench/profiling
 * aid: lets bench.py time the dominant kernels with HIP events on the launch stream).
 * stage: 0 operand producers (hidden = tanh(enc+pred), W repack), 1 joint-forward GEMM kernel,
 * 2 lattice sweep (alpha & beta), 3 gradient coefficients + G in place of logits, 4 dHidden GEMM
 * kernel, 5 dEnc/dPred slab reduction, 6 dW GEMM kernel, 7 dW/db slab reduction.
 * Arguments as for rnnt_engine_joint_loss_fwd_bwd.
 */
int rnnt_engine_run_stage(int stage, const void *enc, const int64_t enc_strides[3],
                          const void *pred, const void *W, const void *bias,
                          const int32_t *targets, const int32_t *logit_lens,
                          const int32_t *target_lens, int B, int T, int U1, int H, int V,
                          int blank, float clamp, float grad_scale, int dtype, float *costs,
                          void *grad_enc, void *grad_pred, void *grad_W, void *grad_bias,
                          void *workspace, size_t ws_bytes, void *stream);

/*
 * Any subset of the stages (bit s of `stage_mask` = stage s above; 255 = the whole pipeline) with
 * the kernel variants of `variant` (RNNT_VARIANT_*), chosen for this call only.  Gradient
 * pointers may be NULL when no backward stage (4..7) is selected.  Test / bench aid.
 */
int rnnt_engine_run_stages(int stage_mask, int variant, const void *enc, const int64_t enc_strides[3],
                           const void *pred, const void *W, const void *bias, const int32_t *targets,
                           const int32_t *logit_lens, const int32_t *target_lens, int B, int T, int U1,
                           int H, int V, int blank, float clamp, float grad_scale, int dtype,
                           float *costs, void *grad_enc, void *grad_pred, void *grad_W,
                           void *grad_bias, void *workspace, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RNNT_ENGINE_H */
