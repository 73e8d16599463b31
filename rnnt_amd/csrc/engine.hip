// engine.hip — C-ABI entry points of librnnt_engine.so (see include/rnnt_engine.h).
//
// Host-side only: argument validation, workspace carving and kernel sequencing on the
// caller's stream.  No allocation, no synchronisation, no state kept between calls.
#include "../../include/rnnt_engine.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <string>

#include "kernels.hpp"

namespace {

thread_local std::string g_err;
// The shipped library keeps NO mutable state besides the thread-local error string: kernel
// variants are selected per call (rnnt_engine_run_stages).  Only diagnostic builds
// (-DRNNT_ABLATE: in-kernel ablation switches; -DRNNT_STAMPS: s_memtime stamps) carry the
// process-wide words rnnt_engine_set_flags / rnnt_engine_set_debug write.
#if defined(RNNT_ABLATE) || defined(RNNT_STAMPS)
int g_flags = 0;
unsigned long long *g_debug = nullptr;
#else
constexpr int g_flags = 0;
constexpr unsigned long long *g_debug = nullptr;
#endif

}  // namespace

// record the calling thread's error message (rnnt_engine_last_error) and hand the code back
int engine_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {
#define fail engine_fail

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

int check_dims(int B, int T, int U1, int H, int V, int dtype, bool need_h)
{
    if (dtype != RNNT_DTYPE_F32 && dtype != RNNT_DTYPE_BF16 && dtype != RNNT_DTYPE_F32_BF16X3 && dtype != RNNT_DTYPE_F32_F16X2)
        return fail(RNNT_ERR_UNSUPPORTED, "dtype %d not supported (RNNT_DTYPE_F32 / RNNT_DTYPE_BF16 / RNNT_DTYPE_F32_BF16X3 / RNNT_DTYPE_F32_F16X2)", dtype);
    const char *dname = dtype == RNNT_DTYPE_BF16 ? "RNNT_DTYPE_BF16" : dtype == RNNT_DTYPE_F32_F16X2 ? "RNNT_DTYPE_F32_F16X2" : "RNNT_DTYPE_F32_BF16X3";
    if (dtype != RNNT_DTYPE_F32 && !need_h)
        return fail(RNNT_ERR_UNSUPPORTED, "%s only applies to the fused joint+loss entry", dname);
    if (dtype != RNNT_DTYPE_F32 && (H <= 0 || H % 128 != 0 || V <= 0 || V % 128 != 0))
        return fail(RNNT_ERR_UNSUPPORTED, "%s needs H %% 128 == 0 and V %% 128 == 0 (H=%d V=%d)", dname, H, V);
    if (B <= 0 || T <= 0 || U1 <= 0 || V <= 0 || (need_h && H <= 0))
        return fail(RNNT_ERR_INVALID_ARG, "non-positive dimension B=%d T=%d U1=%d H=%d V=%d", B, T, U1, H, V);
    if (V % 4 != 0) return fail(RNNT_ERR_UNSUPPORTED, "V=%d must be a multiple of 4 (pad on the host side)", V);
    if (need_h && H % 4 != 0) return fail(RNNT_ERR_UNSUPPORTED, "H=%d must be a multiple of 4 (pad on the host side)", H);
    if (U1 > 1024) return fail(RNNT_ERR_UNSUPPORTED, "U1=%d exceeds 1024 lattice columns", U1);
    if ((long)T * U1 > 0x7fffffffL / 2) return fail(RNNT_ERR_UNSUPPORTED, "T*U1 too large");
    return RNNT_OK;
}

int dw_splits(int B, int T, int H, int V, int dtype)
{
    const long tiles = dtype == RNNT_DTYPE_F32 ? dw_tiles(H, V) : dtype == RNNT_DTYPE_F32_F16X2 ? x2_dw_tiles(H, V) : (long)((V + 255) / 256) * ((H + 255) / 256);
    long s = 256 / tiles;
    if (s < 1) s = 1;
    if (s > (long)B * T) s = (long)B * T;
    return (int)s;
}

// forward row tiles of 128; the dW DMA ring over-reads up to 96 (zero) rows past rows_pad
long bf16_rows_alloc(size_t rows_pad) { return (long)((rows_pad + 96 + 127) / 128 * 128); }

void layout(int B, int T, int U1, int H, int V, int dtype, rnnt_engine_ws_layout *L)
{
    const size_t D = (size_t)T + U1 - 1;
    const size_t cells = (size_t)B * T * U1, skew = (size_t)B * D * U1;
    const bool bf = dtype == RNNT_DTYPE_BF16, x2 = dtype == RNNT_DTYPE_F32_F16X2, x3 = dtype == RNNT_DTYPE_F32_BF16X3 || x2;
    // (x3: the split-operand routes — bf16x3, and f16x2 with two planes instead of three)
    // G / hidden rows are padded with >= 1 zero row up to a multiple of 16 (dW chunk size)
    // (bf16 / bf16x3 routes: multiple of 32 = one granule of their live-row table)
    const size_t rows_pad = (bf || x3) ? (cells + 1 + 31) / 32 * 32 : (cells + 1 + 15) / 16 * 16;
    L->rows_pad = rows_pad;
    L->D = (int)D;
    // dEnc slabs: one per u block; the fp32 route's fused dHidden kernel may use 8-wide blocks (the bf16x3
    // route can run its dHidden stage on that kernel: RNNT_VARIANT_X3_FP32_DH)
    L->n_ublk = bf ? (U1 + 15) / 16 : x3 ? (U1 + 7) / 8 : (U1 + dhidden_gen_bu(T, U1) - 1) / dhidden_gen_bu(T, U1);
    L->n_ttile = (T + 3) / 4;  // dPred slabs: at most one per 4 t rows (the persistent kernel's 16-wide items)
    L->n_split = dw_splits(B, T, H, V, dtype);
    L->g_lo = 0; L->aux = 0; L->aux_bytes = 0;
    size_t o = 0;
    if (bf) {
        const size_t ra = (size_t)bf16_rows_alloc(rows_pad);
        L->logits = o;   o += align_up(ra * V * 2);  // fp16
        L->hidden = o;   o += align_up(ra * H * 2);
    } else if (x3) {
        const size_t ra = (size_t)bf16_rows_alloc(rows_pad);
        L->logits = o;   o += align_up(ra * V * 4);      // fp32 logits; G hi | mid planes in place
        L->hidden = o;   o += align_up((x2 ? 2 : 3) * ra * H * 2);  // three bf16 planes (f16x2: two fp16 planes)
        if (!x2) { L->g_lo = o; o += align_up(ra * V * 2); }      // lo plane of G (f16x2: none — g_lo stays 0)
    } else {
        L->logits = o;   o += align_up((rows_pad + 16) * V * 4);
        L->hidden = o;   o += align_up((rows_pad + 16) * H * 4);
    }
    L->denom_s = o;  o += align_up(skew * 4);
    L->lpb_s = o;    o += align_up(skew * 4);
    L->lpe_s = o;    o += align_up(skew * 4);
    L->alpha_s = o;  o += align_up(skew * 8);
    L->beta_s = o;   o += align_up(skew * 8);
    L->coef = o;     o += align_up(cells * 16);
    L->wpack = o;
    if (bf) o += align_up(bf16_wpack_fwd_bytes(H, V)) + align_up(bf16_wpack_dh_bytes(H, V));
    else if (x2) o += align_up(x2_wpack_fwd_bytes(H, V)) + align_up(x2_wpack_dh_bytes(H, V));
    else if (x3) o += align_up(x3_wpack_fwd_bytes(H, V)) + align_up(x3_wpack_dh_bytes(H, V));
    else o += align_up(wpack_floats(H, V) * 4);
    L->enc_copy = o; o += align_up((size_t)B * T * H * 4);
    L->ep = 0;
    if (x2) { L->ep = o; o += align_up(x2_ep_bytes(B, T, U1, H)); }  // exp(2 enc) | exp(2 pred), k-step major (k_x2_make_ep)
    L->slab_enc = o; o += align_up((size_t)L->n_ublk * B * T * H * 4);
    L->slab_pred = o; o += align_up((size_t)L->n_ttile * B * U1 * H * 4);
    L->slab_w = o;   o += align_up((size_t)L->n_split * V * H * 4);
    L->slab_b = o;   o += align_up((size_t)L->n_split * V * 4);
    {   // + the dW live-row table (bf16 routes) / live-granule list (fp32 route), whichever is larger
        const size_t tab = (2 * (size_t)B + 2) * 8, lst = (bf || x3) ? 0 : dw_list_bytes(B, T, U1, 16);
        L->counters = o; o += 1024 + align_up(tab > lst ? tab : lst);
        if (bf || x2) o += align_up((size_t)L->n_split * 64);  // k_dw_bf16's / k_dw_x2's progress words, behind the table
    }
    L->total = o;
    if (x3) {  // fp32 hidden + fp32 W pack of the stages that can run on the fp32 route's kernels (RNNT_VARIANT_X3_FP32_*):
        // BEHIND `total` — only a call that asks for such a variant needs a workspace of total + aux_bytes
        L->aux = o; L->aux_bytes = align_up((rows_pad + 16) * H * 4) + align_up(wpack_floats(H, V) * 4);
    }
}

}  // namespace

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_fill32(unsigned *__restrict__ p, unsigned value, size_t n_words, int vec)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {  // 16-byte aligned, whole uint4s
        if (i < n_words / 4) ((u32x4 *)p)[i] = u32x4{value, value, value, value};
    } else if (i < n_words) {
        p[i] = value;
    }
}

__global__ __launch_bounds__(256) void k_copy32(unsigned *__restrict__ d, const unsigned *__restrict__ s, size_t n_words, int vec)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        if (i < n_words / 4) ((u32x4 *)d)[i] = ((const u32x4 *)s)[i];
    } else if (i < n_words) {
        d[i] = s[i];
    }
}

void launch_fill32(void *p, unsigned value, size_t bytes, hipStream_t st)
{
    const size_t n = bytes / 4;
    if (!n) return;
    const int vec = ((uintptr_t)p % 16 == 0) && (n % 4 == 0);
    const size_t thr = vec ? n / 4 : n;
    hipLaunchKernelGGL(k_fill32, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, (unsigned *)p, value, n, vec);
}

void launch_copy_bytes(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    const size_t n = bytes / 4;
    if (!n) return;
    const int vec = ((uintptr_t)dst % 16 == 0) && ((uintptr_t)src % 16 == 0) && (n % 4 == 0);
    const size_t thr = vec ? n / 4 : n;
    hipLaunchKernelGGL(k_copy32, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, (unsigned *)dst, (const unsigned *)src, n, vec);
}

namespace {

int launch_status(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RNNT_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return RNNT_OK;
}

// Resolve enc into an h-contiguous, 16-byte friendly view (copying into the workspace when
// the caller's strides do not allow 16-byte row loads, e.g. the (B,C,T)->(B,T,C) permuted
// view of reference rnnt/model.py:28).
void resolve_enc(const void *enc, const int64_t s[3], int B, int T, int H, float *copy_buf,
                 hipStream_t st, const float **out, long *sb, long *st_)
{
    const bool direct = s[2] == 1 && (s[1] % 4) == 0 && (s[0] % 4) == 0 && aligned16(enc);
    if (direct) {
        *out = (const float *)enc; *sb = s[0]; *st_ = s[1];
    } else {
        launch_copy_enc((const float *)enc, s[0], s[1], s[2], copy_buf, B, T, H, st);
        *out = copy_buf; *sb = (long)T * H; *st_ = H;
    }
}

// compute units of the CURRENT device (the caller makes the tensors' device current); read-only
// facts of the hardware, cached per device id
int device_cus()
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

enum { ST_PROD = 1, ST_FWD = 2, ST_LATTICE = 4, ST_COEF = 8, ST_DH = 16, ST_DH_RED = 32, ST_DW = 64,
       ST_DW_RED = 128, ST_ALL = 255 };

int run_fused(int stages, int variant, const void *enc, const int64_t enc_strides[3], const void *pred,
              const void *W, const void *bias, const int32_t *targets, const int32_t *logit_lens,
              const int32_t *target_lens, int B, int T, int U1, int H, int V, int blank,
              float clamp, float grad_scale, int dtype, float *costs, void *grad_enc,
              void *grad_pred, void *grad_W, void *grad_bias, void *workspace, size_t ws_bytes,
              void *stream)
{
    if (int rc = check_dims(B, T, U1, H, V, dtype, true)) return rc;
    if (!enc || !enc_strides || !pred || !W || !bias || !targets || !logit_lens || !target_lens ||
        !costs || !workspace)
        return fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const bool backward = (stages & (ST_DH | ST_DH_RED | ST_DW | ST_DW_RED)) != 0;
    if (backward && (!grad_enc || !grad_pred || !grad_W || !grad_bias))
        return fail(RNNT_ERR_INVALID_ARG, "null gradient pointer");
    if (blank < 0 || blank >= V) return fail(RNNT_ERR_INVALID_ARG, "blank=%d outside [0,%d)", blank, V);
    if (clamp > 0.f) return fail(RNNT_ERR_UNSUPPORTED, "clamp>0 is only supported by rnnt_engine_loss_fwd_bwd");
    if (!(grad_scale > 0.f)) return fail(RNNT_ERR_INVALID_ARG, "grad_scale must be > 0");
    if (!aligned16(pred) || !aligned16(W) || !aligned16(bias) || !aligned16(grad_enc) ||
        !aligned16(grad_pred) || !aligned16(grad_W) || !aligned16(grad_bias) ||
        ((uintptr_t)workspace & 255))
        return fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned (workspace 256)");
    // kernels that were measured equal to (or slower than) the shipped ones live in the diagnostic library only
    // (-DRNNT_LAB, tools/build_lab.sh): the product library refuses their variant bits instead of silently running something else
    constexpr int lab_only = RNNT_VARIANT_LAB_MASK;
#ifndef RNNT_LAB
    if (variant & lab_only)
        return fail(RNNT_ERR_UNSUPPORTED, "variant 0x%x names a kernel of the diagnostic library (build_variants/lab/librnnt_engine_lab.so, "
                    "tools/build_lab.sh): librnnt_engine.so ships the default kernels only", variant & lab_only);
#endif
    const int xflags = g_flags | (variant & (RNNT_VARIANT_SEPARATE_G | RNNT_VARIANT_SEPARATE_HIDDEN |
                                             RNNT_VARIANT_FWD_LDS_RING | RNNT_VARIANT_FWD_ONE_WG_PER_TILE |
                                             RNNT_VARIANT_X3_FP32_FWD | RNNT_VARIANT_X3_FP32_DH |
                                             RNNT_VARIANT_X3_FWD_2WG | RNNT_VARIANT_X3_FWD_8W | RNNT_VARIANT_X3_DW_P16 | RNNT_VARIANT_X3_FWD_Z | RNNT_VARIANT_X2_DW_8W | RNNT_VARIANT_X2_FWD_2WG | RNNT_VARIANT_X2_DW_P16));
    rnnt_engine_ws_layout L;
    layout(B, T, U1, H, V, dtype, &L);
    if (ws_bytes < L.total)
        return fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, L.total);

    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *logits = (float *)(ws + L.logits);
    float *denom_s = (float *)(ws + L.denom_s), *lpb_s = (float *)(ws + L.lpb_s),
          *lpe_s = (float *)(ws + L.lpe_s);
    double *alpha_s = (double *)(ws + L.alpha_s), *beta_s = (double *)(ws + L.beta_s);
    CellCoef *coef = (CellCoef *)(ws + L.coef);
    float *wpack = (float *)(ws + L.wpack);

    const float *encp; long esb, est;
    resolve_enc(enc, enc_strides, B, T, H, (float *)(ws + L.enc_copy), st, &encp, &esb, &est);

    JointBwdArgs g;
    g.enc = encp; g.enc_sb = esb; g.enc_st = est; g.pred = (const float *)pred;
    g.W = (const float *)W; g.logits = logits; g.coef = coef; g.logit_lens = logit_lens; g.target_lens = target_lens;
    g.hidden = (float *)(ws + L.hidden); g.rows_pad = (long)L.rows_pad;
    g.slab_enc = (float *)(ws + L.slab_enc); g.slab_pred = (float *)(ws + L.slab_pred);
    g.slab_w = (float *)(ws + L.slab_w); g.slab_b = (float *)(ws + L.slab_b);
    g.grad_enc = (float *)grad_enc; g.grad_pred = (float *)grad_pred;
    g.grad_W = (float *)grad_W; g.grad_bias = (float *)grad_bias;
    g.B = B; g.T = T; g.U1 = U1; g.H = H; g.V = V; g.blank = blank;
    g.n_ublk = L.n_ublk; g.n_ttile = L.n_ttile; g.n_split = L.n_split;
    g.counter = (unsigned *)(ws + L.counters); g.dw_tab = (long *)(ws + L.counters + 1024); g.n_cu = device_cus(); g.flags = xflags & ~16; g.debug = g_debug; g.pred_split_col = 0;
    g.gen_bu = dtype == RNNT_DTYPE_BF16 ? 16 : dhidden_gen_bu(T, U1);  // u width of the dHidden tiles
    if (dtype == RNNT_DTYPE_F32_BF16X3 || dtype == RNNT_DTYPE_F32_F16X2) {
        // RNNT_DTYPE_F32_F16X2 (x2.hip): the same stages on two fp16 planes and three products; operand scales below
        const bool x2 = dtype == RNNT_DTYPE_F32_F16X2;
        // fp32-accurate route on the bf16 matrix pipes (x3.hip).  Stage by stage the fp32 route's own kernel can
        // stand in (RNNT_VARIANT_X3_FP32_FWD / _DH): same data, one stage swapped — how each x3 kernel is checked.
        X3Args h;
        h.enc = encp; h.enc_sb = esb; h.enc_st = est; h.pred = (const float *)pred;
        h.W = (const float *)W; h.bias = (const float *)bias;
        h.rows_pad = (long)L.rows_pad; h.rows_alloc = bf16_rows_alloc(L.rows_pad);
        h.hidden = (unsigned short *)(ws + L.hidden); h.plane_stride = h.rows_alloc * (long)H;
        h.wpack_fwd = ws + L.wpack; h.wpack_dh = ws + L.wpack + align_up(x2 ? x2_wpack_fwd_bytes(H, V) : x3_wpack_fwd_bytes(H, V));
        {   // f16x2 operand scales: |G| <= grad_scale <= 2^e -> g_scale = 2^(13 - e); hidden: 2^14 (x2.hip)
            int e = 0;
            (void)frexpf(grad_scale, &e);
            const int k = 13 - e < -100 ? -100 : (13 - e > 100 ? 100 : 13 - e);
            h.g_scale = ldexpf(1.0f, k);
            h.dw_rescale = 1.0f / (h.g_scale * 16384.0f); h.db_rescale = 1.0f / h.g_scale;
            h.scales = (const float *)(ws + L.counters + 640);
            h.dw_prog = x2 ? (int *)(ws + L.counters + 1024 + align_up((2 * (size_t)B + 2) * 8)) : nullptr;
            h.ep_enc = (float *)(ws + L.ep); h.ep_pred = h.ep_enc + (size_t)B * T * H;
            h.ep_flag = (unsigned *)(ws + L.counters + 896);
        }
        h.logits = logits; h.g_lo = (unsigned short *)(ws + L.g_lo); h.coef = coef;
        h.targets = targets; h.logit_lens = logit_lens; h.target_lens = target_lens;
        h.denom_s = denom_s; h.lpb_s = lpb_s; h.lpe_s = lpe_s; h.D = L.D;
        h.slab_enc = g.slab_enc; h.slab_pred = g.slab_pred; h.slab_w = g.slab_w; h.slab_b = g.slab_b;
        h.B = B; h.T = T; h.U1 = U1; h.H = H; h.V = V; h.blank = blank;
        h.n_ublk = L.n_ublk; h.n_split = L.n_split; h.flags = xflags; h.n_ublk16 = (U1 + 15) / 16;
        h.dw_tab = (long *)(ws + L.counters + 1024);
        h.counter = (unsigned *)(ws + L.counters + 512); h.n_cu = device_cus(); h.debug = g_debug;
        const bool f32_dh = (xflags & RNNT_VARIANT_X3_FP32_DH) != 0 || !(x2 ? x2_dhidden_ok(U1, H, V) : x3_dhidden_ok(U1, H, V));
        const bool f32_fwd = (xflags & RNNT_VARIANT_X3_FP32_FWD) != 0 || !(x2 ? x2_fwd_ok(U1, H, V) : x3_fwd_ok(U1, H, V)) || f32_dh;  // fp32 dHidden reads fp32 hidden
        if ((f32_fwd || f32_dh) && ws_bytes < L.total + L.aux_bytes)
            return fail(RNNT_ERR_WORKSPACE, "workspace %zu < %zu bytes: a stage on the fp32 route's kernels (RNNT_VARIANT_X3_FP32_*, or a "
                        "shape k_joint_fwd_x3 / k_dhidden_x3 do not cover) needs total + aux_bytes of rnnt_engine_workspace_layout",
                        ws_bytes, L.total + L.aux_bytes);
        float *hid32 = (float *)(ws + L.aux);
        float *wpack32 = (float *)(ws + L.aux + align_up((L.rows_pad + 16) * (size_t)H * 4));
        g.hidden = hid32;
        const bool fuse_g32 = dhidden_gen_ok(H, V, U1);
        if (f32_dh) {
            if (fuse_g32) { g.flags |= 16; g.pred_split_col = 512 * dhidden_gen_groups(H); }
        } else {
            g.gen_bu = 16; g.pred_split_col = H;  // x3 tiles: 8 t x 16 u, every dPred slab 8 t rows high
        }
        if (stages & ST_PROD) {
            if (x2) { launch_x2_zero_padding(h, 1, st); launch_x2_pack_w(h, (float *)(ws + L.counters + 640), st); launch_x2_make_ep(h, st); }
            else { launch_x3_zero_padding(h, 1, st); launch_x3_pack_w(h, st); }
            if (f32_fwd) {
                const size_t cells = (size_t)B * T * U1;
                launch_fill32(hid32 + cells * H, 0u, (L.rows_pad + 16 - cells) * H * 4, st);
                launch_zero_dead_hidden(hid32, logit_lens, target_lens, B, T, U1, H, st);
                launch_pack_w_fwd((const float *)W, wpack32, H, V, st);
            }
        }
        if (stages & ST_FWD) {
            if (f32_fwd) {
                JointFwdArgs f;
                f.enc = encp; f.enc_sb = esb; f.enc_st = est; f.pred = (const float *)pred;
                f.wpack = wpack32; f.hidden = hid32; f.bias = (const float *)bias; f.targets = targets;
                f.logit_lens = logit_lens; f.target_lens = target_lens; f.logits = logits;
                f.denom_s = denom_s; f.lpb_s = lpb_s; f.lpe_s = lpe_s;
                f.B = B; f.T = T; f.U1 = U1; f.H = H; f.V = V; f.D = L.D; f.blank = blank; f.flags = 0; f.debug = nullptr;
                f.make_hidden = 1; f.counter = h.counter; f.n_cu = h.n_cu;
                launch_joint_fwd(f, st);
                if (x2) launch_x2_make_hidden(h, st);
                else launch_x3_make_hidden(h, st);  // the planes the backward reads
            } else if (x2) {
#ifdef RNNT_LAB
                if ((xflags & RNNT_VARIANT_X2_FWD_2WG) && x2_fwd_d_ok(U1, H, V)) launch_joint_fwd_x2d(h, st);
                else
#endif
                launch_joint_fwd_x2(h, st);
            } else {
#ifdef RNNT_LAB  // each variant bit launches exactly its own kernel (round-4 advice: _FWD_Z used to fall through to x3d<4>)
                if ((xflags & RNNT_VARIANT_X3_FWD_Z) && x3_fwd_d_ok(U1, H, V)) launch_joint_fwd_x3z(h, st);
                else if ((xflags & RNNT_VARIANT_X3_FWD_8W) && x3_fwd_d_ok(U1, H, V)) launch_joint_fwd_x3d(h, 8, st);
                else if ((xflags & RNNT_VARIANT_X3_FWD_2WG) && x3_fwd_d_ok(U1, H, V)) launch_joint_fwd_x3d(h, 4, st);
                else
#endif
                launch_joint_fwd_x3(h, st);
            }
        }
        if (stages & ST_LATTICE)
            launch_lattice(lpb_s, lpe_s, alpha_s, beta_s, logit_lens, target_lens, costs, B, U1, L.D,
                           (unsigned *)(ws + L.counters + 768), st);
        if (stages & ST_COEF)
            launch_coef(alpha_s, beta_s, denom_s, lpb_s, lpe_s, targets, logit_lens, target_lens, coef,
                        B, T, U1, L.D, grad_scale, st);
        if (stages & ST_DH) {
            if (x2) launch_x2_zero_padding(h, 2, st);
            else launch_x3_zero_padding(h, 2, st);
            if (f32_dh) {
                if (!fuse_g32) launch_make_g(g, st);
                launch_dhidden(g, st);     // leaves fp32 G in place of the logits
                if (x2) launch_x2_split_g(h, st);
                else launch_x3_split_g(h, st);  // -> hi | mid in place, lo beside
            } else if (x2) {
                launch_dhidden_x2(h, st);
            } else {
                launch_dhidden_x3(h, st);
            }
        }
        if (stages & ST_DH_RED) launch_dhidden_reduce(g, st);
        if (stages & ST_DW) {
            if (x2) launch_dw_x2(h, st);
#ifdef RNNT_LAB
            else if (xflags & RNNT_VARIANT_X3_DW_P16) launch_dw_x3p(h, st);
#endif
            else launch_dw_x3(h, st);
        }
        if (stages & ST_DW_RED) launch_dw_reduce(g, st);
        return launch_status(x2 ? "rnnt_engine fused pipeline (f16x2)" : "rnnt_engine fused pipeline (bf16x3)");
    }
    if (dtype == RNNT_DTYPE_BF16) {
        Bf16Args h;
        h.enc = encp; h.enc_sb = esb; h.enc_st = est; h.pred = (const float *)pred;
        h.W = (const float *)W; h.bias = (const float *)bias;
        h.hidden = (unsigned short *)(ws + L.hidden);
        h.wpack_fwd = ws + L.wpack; h.wpack_dh = ws + L.wpack + align_up(bf16_wpack_fwd_bytes(H, V));
        h.logits = (unsigned short *)logits; h.coef = coef; h.logit_lens = logit_lens;
        h.targets = targets; h.target_lens = target_lens;
        h.denom_s = denom_s; h.lpb_s = lpb_s; h.lpe_s = lpe_s; h.D = L.D;
        h.slab_enc = g.slab_enc; h.slab_pred = g.slab_pred; h.slab_w = g.slab_w; h.slab_b = g.slab_b;
        h.rows_pad = (long)L.rows_pad; h.rows_alloc = bf16_rows_alloc(L.rows_pad);
        h.B = B; h.T = T; h.U1 = U1; h.H = H; h.V = V; h.blank = blank;
        h.n_ublk = L.n_ublk; h.n_split = L.n_split; h.flags = xflags;
        h.dw_tab = (long *)(ws + L.counters + 1024);
        h.dw_prog = (int *)(ws + L.counters + 1024 + align_up((2 * (size_t)B + 2) * 8));
        g.pred_split_col = H;  // reductions: every dPred slab is 8 t-rows high, as in k_dhidden_gen
        h.debug = g_debug;
        if (stages & ST_PROD) launch_bf16_producers(h, st);
        if (stages & ST_FWD) launch_joint_fwd_bf16(h, st);  // softmax statistics in its epilogue
        if (stages & ST_LATTICE)
            launch_lattice(lpb_s, lpe_s, alpha_s, beta_s, logit_lens, target_lens, costs, B, U1, L.D,
                           (unsigned *)(ws + L.counters + 768), st);
        if (stages & ST_COEF)
            launch_coef(alpha_s, beta_s, denom_s, lpb_s, lpe_s, targets, logit_lens, target_lens, coef,
                        B, T, U1, L.D, grad_scale, st);
        if (stages & ST_DH) launch_dhidden_bf16(h, st);
        if (stages & ST_DH_RED) launch_dhidden_reduce(g, st);
        if (stages & ST_DW) launch_dw_bf16(h, st);
        if (stages & ST_DW_RED) launch_dw_reduce(g, st);
        return launch_status("rnnt_engine fused pipeline (bf16)");
    }
    // G inside the dHidden GEMM unless the shape needs the separate pass (or flag 32 forces it)
    const bool fuse_g = dhidden_gen_ok(H, V, U1) && !(xflags & 32);
    if (fuse_g) { g.flags |= 16; g.pred_split_col = 512 * dhidden_gen_groups(H); }  // columns on the tile kernel (8-row dPred slabs)

    // hidden (A operand of all three GEMMs) is produced by the forward kernel for its own tile
    // unless flag 64 asks for the separate k_make_hidden pass
    const bool fuse_hid = !(xflags & 64) && !(xflags & 8);
    if (stages & ST_PROD) {
        if (fuse_hid) {  // only the zero padding rows the dW GEMM walks past the last cell
            const size_t cells = (size_t)B * T * U1;
            launch_fill32(g.hidden + cells * H, 0u, (L.rows_pad + 16 - cells) * H * 4, st);
            // and the dead cells next to lattice cells (the forward writes lattice cells only)
            launch_zero_dead_hidden(g.hidden, logit_lens, target_lens, B, T, U1, H, st);
        } else {
            launch_make_hidden(g, st);
        }
        launch_pack_w_fwd((const float *)W, wpack, H, V, st);
    }
    if (stages & ST_FWD) {
        JointFwdArgs f;
        f.enc = encp; f.enc_sb = esb; f.enc_st = est; f.pred = (const float *)pred;
        f.wpack = wpack; f.hidden = (xflags & 8) ? nullptr : g.hidden; f.bias = (const float *)bias; f.targets = targets;
        f.logit_lens = logit_lens; f.target_lens = target_lens; f.logits = logits;
        f.denom_s = denom_s; f.lpb_s = lpb_s; f.lpe_s = lpe_s;
        f.B = B; f.T = T; f.U1 = U1; f.H = H; f.V = V; f.D = L.D; f.blank = blank; f.flags = xflags; f.debug = g_debug;
        f.make_hidden = fuse_hid ? 1 : 0;
        f.counter = (unsigned *)(ws + L.counters + 512); f.n_cu = device_cus();
        launch_joint_fwd(f, st);
    }
    if (stages & ST_LATTICE)
        launch_lattice(lpb_s, lpe_s, alpha_s, beta_s, logit_lens, target_lens, costs, B, U1, L.D,
                           (unsigned *)(ws + L.counters + 768), st);
    if (stages & ST_COEF)
        launch_coef(alpha_s, beta_s, denom_s, lpb_s, lpe_s, targets, logit_lens, target_lens, coef,
                    B, T, U1, L.D, grad_scale, st);
    {
        if ((stages & ST_COEF) && !fuse_g) launch_make_g(g, st);  // logits -> G in place
        if (stages & ST_DH) launch_dhidden(g, st);
        if (stages & ST_DH_RED) launch_dhidden_reduce(g, st);
        if (stages & ST_DW) launch_dw(g, st);
        if (stages & ST_DW_RED) launch_dw_reduce(g, st);
    }
    return launch_status("rnnt_engine fused pipeline");
}

}  // namespace

extern "C" {

int rnnt_engine_version(void) { return RNNT_ENGINE_VERSION; }

#if defined(RNNT_ABLATE) || defined(RNNT_STAMPS)
int rnnt_engine_set_flags(int flags) { int o = g_flags; g_flags = flags; return o; }
void rnnt_engine_set_debug(void *buf) { g_debug = (unsigned long long *)buf; }
#else  // shipped build: no process-wide state; the symbols stay so diagnostic tools link
int rnnt_engine_set_flags(int) { return 0; }
void rnnt_engine_set_debug(void *) {}
#endif

int rnnt_engine_debug_query(int what) { return fwd_occupancy(what); }

const char *rnnt_engine_last_error(void) { return g_err.c_str(); }

// ---- RCCL all-reduce of the flat gradient buffer.  No link-time dependency: the symbol
// comes from the RCCL already in the process (soname librccl.so.1: PyTorch-ROCm's copy when torch is
// loaded), so the communicator the caller made and the function called belong to the same library.
int rnnt_engine_allreduce(void *buf, size_t count, void *comm, void *stream)
{
    if (!buf || !comm) return fail(RNNT_ERR_INVALID_ARG, "null buffer / communicator");
    if (count == 0) return RNNT_OK;
    typedef int (*allreduce_fn)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef const char *(*errstr_fn)(int);
    void *sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    void *lib = nullptr;
    if (!sym) {
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);  // the copy that is already loaded
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW);
        if (lib) sym = dlsym(lib, "ncclAllReduce");
    }
    if (!sym) return fail(RNNT_ERR_UNSUPPORTED, "RCCL (librccl.so.1: ncclAllReduce) not found in this process");
    const int ncclFloat32_ = 7, ncclSum_ = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t
    const int rc = ((allreduce_fn)sym)(buf, buf, count, ncclFloat32_, ncclSum_, comm, (hipStream_t)stream);
    if (rc != 0) {
        void *es = lib ? dlsym(lib, "ncclGetErrorString") : dlsym(RTLD_DEFAULT, "ncclGetErrorString");
        return fail(RNNT_ERR_LAUNCH, "ncclAllReduce failed (%d): %s", rc, es ? ((errstr_fn)es)(rc) : "?");
    }
    return RNNT_OK;
}

int rnnt_engine_workspace_layout(int B, int T, int U1, int H, int V, int dtype,
                                 rnnt_engine_ws_layout *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null layout pointer");
    if (int rc = check_dims(B, T, U1, H, V, dtype, true)) return rc;
    layout(B, T, U1, H, V, dtype, out);
    return RNNT_OK;
}

int rnnt_engine_workspace_bytes(int B, int T, int U1, int H, int V, int dtype, size_t *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    rnnt_engine_ws_layout L;
    if (int rc = rnnt_engine_workspace_layout(B, T, U1, H, V, dtype, &L)) return rc;
    *out = L.total;
    return RNNT_OK;
}

int rnnt_engine_loss_workspace_bytes(int B, int T, int U1, int V, int dtype, size_t *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (int rc = check_dims(B, T, U1, 4, V, dtype, false)) return rc;
    const size_t D = (size_t)T + U1 - 1, skew = (size_t)B * D * U1, cells = (size_t)B * T * U1;
    *out = 3 * align_up(skew * 4) + 2 * align_up(skew * 8) + align_up(cells * 16) + 256;  // + error word
    return RNNT_OK;
}

int rnnt_engine_joint_fwd_workspace_bytes(int B, int T, int U1, int H, int V, int dtype,
                                          size_t *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (int rc = check_dims(B, T, U1, H, V, dtype, true)) return rc;
    *out = align_up(wpack_floats(H, V) * 4) + align_up((size_t)B * T * H * 4);
    return RNNT_OK;
}

int rnnt_engine_joint_fwd(const void *enc, const int64_t enc_strides[3], const void *pred,
                          const void *W, const void *bias, int B, int T, int U1, int H, int V,
                          int dtype, void *logits, void *workspace, size_t ws_bytes, void *stream)
{
    if (int rc = check_dims(B, T, U1, H, V, dtype, true)) return rc;
    if (!enc || !enc_strides || !pred || !W || !bias || !logits || !workspace)
        return fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    if (!aligned16(pred) || !aligned16(W) || !aligned16(bias) || !aligned16(logits) ||
        ((uintptr_t)workspace & 255))
        return fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned (workspace 256)");
    size_t need;
    rnnt_engine_joint_fwd_workspace_bytes(B, T, U1, H, V, dtype, &need);
    if (ws_bytes < need) return fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *wpack = (float *)ws;
    float *enc_copy = (float *)(ws + align_up(wpack_floats(H, V) * 4));
    const float *encp; long esb, est;
    resolve_enc(enc, enc_strides, B, T, H, enc_copy, st, &encp, &esb, &est);
    launch_pack_w_fwd((const float *)W, wpack, H, V, st);
    JointFwdArgs f;
    memset(&f, 0, sizeof f);
    f.enc = encp; f.enc_sb = esb; f.enc_st = est; f.pred = (const float *)pred; f.wpack = wpack;
    f.bias = (const float *)bias; f.logits = (float *)logits;
    f.B = B; f.T = T; f.U1 = U1; f.H = H; f.V = V; f.D = T + U1 - 1; f.blank = V - 1; f.flags = g_flags; f.debug = g_debug;
    launch_joint_fwd(f, st);
    return launch_status("rnnt_engine_joint_fwd");
}

int rnnt_engine_joint_bwd_workspace_bytes(int B, int T, int U1, int H, int V, int dtype, size_t *out)
{
    if (dtype != RNNT_DTYPE_F32) return fail(RNNT_ERR_UNSUPPORTED, "rnnt_engine_joint_bwd is fp32 only");
    return rnnt_engine_workspace_bytes(B, T, U1, H, V, dtype, out);
}

int rnnt_engine_joint_bwd(const void *enc, const int64_t enc_strides[3], const void *pred,
                          const void *W, const void *grad_logits, int B, int T, int U1, int H, int V,
                          int dtype, void *grad_enc, void *grad_pred, void *grad_W, void *grad_bias,
                          void *workspace, size_t ws_bytes, void *stream)
{
    if (dtype != RNNT_DTYPE_F32) return fail(RNNT_ERR_UNSUPPORTED, "rnnt_engine_joint_bwd is fp32 only");
    if (int rc = check_dims(B, T, U1, H, V, dtype, true)) return rc;
    if (!enc || !enc_strides || !pred || !W || !grad_logits || !grad_enc || !grad_pred || !grad_W ||
        !grad_bias || !workspace)
        return fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    if (!aligned16(pred) || !aligned16(W) || !aligned16(grad_logits) || !aligned16(grad_enc) ||
        !aligned16(grad_pred) || !aligned16(grad_W) || !aligned16(grad_bias) || ((uintptr_t)workspace & 255))
        return fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned (workspace 256)");
    rnnt_engine_ws_layout L;
    layout(B, T, U1, H, V, dtype, &L);
    if (ws_bytes < L.total) return fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, L.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    const float *encp; long esb, est;
    resolve_enc(enc, enc_strides, B, T, H, (float *)(ws + L.enc_copy), st, &encp, &esb, &est);
    // every utterance at full length: the upstream gradient is dense (zero where the caller's loss
    // ignored a cell); the length arrays the kernels read live in the (unused) coefficient region
    int32_t *ll = (int32_t *)(ws + L.coef), *tl = ll + B;
    launch_fill32(ll, (unsigned)T, (size_t)B * 4, st);
    launch_fill32(tl, (unsigned)(U1 - 1), (size_t)B * 4, st);
    // G with the zero padding rows the dW GEMM walks past the last cell
    const size_t cells = (size_t)B * T * U1;
    float *G = (float *)(ws + L.logits);
    launch_copy_bytes(G, grad_logits, cells * V * 4, st);
    launch_fill32(G + cells * V, 0u, (L.rows_pad + 16 - cells) * V * 4, st);
    JointBwdArgs g;
    memset(&g, 0, sizeof g);
    g.enc = encp; g.enc_sb = esb; g.enc_st = est; g.pred = (const float *)pred; g.W = (const float *)W;
    g.logits = G; g.hidden = (float *)(ws + L.hidden); g.rows_pad = (long)L.rows_pad;
    g.logit_lens = ll; g.target_lens = tl;
    g.slab_enc = (float *)(ws + L.slab_enc); g.slab_pred = (float *)(ws + L.slab_pred);
    g.slab_w = (float *)(ws + L.slab_w); g.slab_b = (float *)(ws + L.slab_b);
    g.grad_enc = (float *)grad_enc; g.grad_pred = (float *)grad_pred;
    g.grad_W = (float *)grad_W; g.grad_bias = (float *)grad_bias;
    g.B = B; g.T = T; g.U1 = U1; g.H = H; g.V = V; g.blank = V - 1;
    g.n_ublk = L.n_ublk; g.n_ttile = L.n_ttile; g.n_split = L.n_split;
    g.counter = (unsigned *)(ws + L.counters); g.dw_tab = (long *)(ws + L.counters + 1024);
    g.n_cu = device_cus(); g.flags = 0; g.debug = nullptr; g.gen_bu = dhidden_gen_bu(T, U1);
    launch_make_hidden(g, st);
    launch_dhidden(g, st);
    launch_dhidden_reduce(g, st);
    launch_dw(g, st);
    launch_dw_reduce(g, st);
    return launch_status("rnnt_engine_joint_bwd");
}

int rnnt_engine_greedy_scan_workspace_bytes(int nframes, int H, int V, size_t *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (nframes < 1 || nframes > 128) return fail(RNNT_ERR_INVALID_ARG, "nframes=%d outside [1,128]", nframes);
    if (int rc = check_dims(1, nframes, 1, H, V, RNNT_DTYPE_F32, true)) return rc;
    if (H % 8 != 0) return fail(RNNT_ERR_UNSUPPORTED, "greedy scan needs H %% 8 == 0 (H=%d)", H);
    *out = align_up((size_t)nframes * H * 4) + align_up((size_t)nframes * V * 4);
    return RNNT_OK;
}

int rnnt_engine_greedy_scan(const void *enc, int64_t enc_stride_t, int64_t enc_stride_h, const void *pred,
                            const void *W, const void *bias, int t0, int nframes, int H, int V, int blank,
                            int32_t *out, void *workspace, size_t ws_bytes, void *stream)
{
    size_t need;
    if (int rc = rnnt_engine_greedy_scan_workspace_bytes(nframes, H, V, &need)) return rc;
    if (!enc || !pred || !W || !bias || !out || !workspace) return fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    if (t0 < 0) return fail(RNNT_ERR_INVALID_ARG, "t0=%d is negative", t0);
    if (blank < 0 || blank >= V) return fail(RNNT_ERR_INVALID_ARG, "blank=%d outside [0,%d)", blank, V);
    if (!aligned16(pred) || !aligned16(W) || !aligned16(bias) || ((uintptr_t)workspace & 255))
        return fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned (workspace 256)");
    if (ws_bytes < need) return fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *enc_copy = (float *)ws;
    float *logits = (float *)(ws + align_up((size_t)nframes * H * 4));
    const int64_t strides[3] = {0, enc_stride_t, enc_stride_h};
    const float *encp; long esb, est;
    resolve_enc((const float *)enc + (int64_t)t0 * enc_stride_t, strides, 1, nframes, H, enc_copy, st, &encp, &esb, &est);
    launch_scan_logits(encp, est, (const float *)pred, (const float *)W, (const float *)bias, logits, nframes, H, V, st);
    launch_argmax_scan(logits, nframes, V, blank, t0, out, st);
    return launch_status("rnnt_engine_greedy_scan");
}

int rnnt_engine_greedy_decode_workspace_bytes(int H, int V, int E, int O, int scan_frames, size_t *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (scan_frames < 1 || scan_frames > 128) return fail(RNNT_ERR_INVALID_ARG, "scan_frames=%d outside [1,128]", scan_frames);
    if (int rc = check_dims(1, scan_frames, 1, H, V, RNNT_DTYPE_F32, true)) return rc;
    if (H % 8 != 0) return fail(RNNT_ERR_UNSUPPORTED, "greedy decode needs H %% 8 == 0 (H=%d)", H);
    if (E < 4 || O < 4 || E > 1024 || O > 1024 || E % 4 || O % 4)
        return fail(RNNT_ERR_UNSUPPORTED, "greedy decode needs 4 <= E, O <= 1024, multiples of 4 (E=%d, O=%d)", E, O);
    *out = align_up(dec_loop_workspace_floats(H, V, E, O, scan_frames) * 4);
    return RNNT_OK;
}

// argument checks shared by the two decode entry points; fills `a` (workspace / iteration fields left to the caller)
static int dec_check_args(const void *frames, int64_t frame_stride, int T, const rnnt_conv_predictor_params *p, int S, int E, int O,
                          float ln_in_eps, float ln_eps, const void *text_W, const void *text_b, const void *W, const void *bias, int H, int V, int blank,
                          int max_length, int max_per_frame, int32_t *host_flag, int32_t *state, int32_t *tokens, void *workspace,
                          DecLoopArgs &a)
{
    int32_t *flag_dev = nullptr;
    if (host_flag) {  // the device's address of the caller's pinned word (an ordinary host pointer is refused, never written through)
        void *dp = nullptr;
        if (hipHostGetDevicePointer(&dp, host_flag, 0) != hipSuccess || !dp) {
            (void)hipGetLastError();
            return fail(RNNT_ERR_INVALID_ARG, "host_flag is not mapped pinned host memory (hipHostMalloc / torch pin_memory)");
        }
        flag_dev = (int32_t *)dp;
    }
    if (!frames || !p || !W || !bias || !state || !tokens || !workspace) return fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const void *ptrs[] = {p->embedding, p->ln_in_w, p->ln_in_b, p->conv1_w, p->conv1_b, p->conv2_w, p->conv2_b,
                          p->linear_w, p->linear_b, p->ln_out_w, p->ln_out_b, W, bias, frames};
    for (const void *q : ptrs)
        if (!q || !aligned16(q)) return fail(RNNT_ERR_INVALID_ARG, "null or not 16-byte aligned parameter pointer");
    if ((text_W == nullptr) != (text_b == nullptr) || (text_W && (!aligned16(text_W) || !aligned16(text_b))))
        return fail(RNNT_ERR_INVALID_ARG, "text_W / text_b: both or neither, 16-byte aligned");
    if (!text_W && O != H) return fail(RNNT_ERR_INVALID_ARG, "without text_ln the predictor's output dim (%d) must equal H (%d)", O, H);
    if (T < 1 || S < 1 || max_length < 2 || max_per_frame < 1 || frame_stride < H || frame_stride % 4)
        return fail(RNNT_ERR_INVALID_ARG, "T=%d S=%d max_length=%d max_per_frame=%d frame_stride=%lld", T, S, max_length,
                    max_per_frame, (long long)frame_stride);
    if (blank < 0 || blank >= V) return fail(RNNT_ERR_INVALID_ARG, "blank=%d outside [0,%d)", blank, V);
    if ((uintptr_t)workspace & 255) return fail(RNNT_ERR_INVALID_ARG, "workspace must be 256-byte aligned");
    a.frames = (const float *)frames; a.frame_stride = (long)frame_stride; a.T = T;
    a.p = *p; a.S = S; a.E = E; a.O = O; a.ln_in_eps = ln_in_eps; a.ln_eps = ln_eps;
    a.text_W = (const float *)text_W; a.text_b = (const float *)text_b;
    a.W = (const float *)W; a.bias = (const float *)bias; a.H = H; a.V = V; a.blank = blank;
    a.max_length = max_length; a.max_per_frame = max_per_frame; a.scan_frames = 0; a.iterations = 0; a.init = 1;
    a.host_flag = flag_dev; a.state = state; a.tokens = tokens; a.workspace = workspace;
    return RNNT_OK;
}

int rnnt_engine_greedy_decode(const void *frames, int64_t frame_stride, int T, const rnnt_conv_predictor_params *p,
                              int S, int E, int O, float ln_in_eps, float ln_eps, const void *text_W, const void *text_b,
                              const void *W, const void *bias, int H, int V, int blank, int max_length,
                              int max_per_frame, int scan_frames, int iterations, int init, int32_t *host_flag,
                              int32_t *state, int32_t *tokens, void *workspace, size_t ws_bytes, void *stream)
{
    size_t need;
    if (int rc = rnnt_engine_greedy_decode_workspace_bytes(H, V, E, O, scan_frames, &need)) return rc;
    if (iterations < 0) return fail(RNNT_ERR_INVALID_ARG, "iterations=%d", iterations);
    DecLoopArgs a;
    if (int rc = dec_check_args(frames, frame_stride, T, p, S, E, O, ln_in_eps, ln_eps, text_W, text_b, W, bias, H, V, blank, max_length, max_per_frame,
                                host_flag, state, tokens, workspace, a))
        return rc;
    if (ws_bytes < need) return fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    a.scan_frames = scan_frames;
    a.iterations = iterations ? iterations : max_length + (T + scan_frames - 1) / scan_frames + 1;
    a.init = init;
    launch_dec_loop(a, (hipStream_t)stream);
    return launch_status("rnnt_engine_greedy_decode");
}

static int dec_persist_check(int T, int S, int E, int O, int H, int V, int has_text, int max_length)
{
    if (int rc = check_dims(1, 16, 1, H, V, RNNT_DTYPE_F32, true)) return rc;
    if (const char *why = dec_persist_refusal(T, S, E, O, H, V, has_text))
        return fail(RNNT_ERR_UNSUPPORTED, "persistent greedy decode: %s (T=%d S=%d E=%d O=%d H=%d V=%d)", why, T, S, E, O, H, V);
    if ((long)T + max_length + 2 >= (1L << 20)) return fail(RNNT_ERR_UNSUPPORTED, "persistent greedy decode: T + max_length must stay below 2^20");
    // the kernel keeps 100-138 KB in LDS: a device with less (not a gfx950) is "unsupported", so that callers take the kernel-per-layer loop
    // instead of a failed launch.  Without a device (size queries on a CPU-only host) there is nothing to check against.
    int dev = 0, lds_max = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess) {
        if (dec_persist_lds_bytes(E) > (size_t)lds_max)
            return fail(RNNT_ERR_UNSUPPORTED, "persistent greedy decode needs %zu bytes of LDS per workgroup, the device allows %d",
                        dec_persist_lds_bytes(E), lds_max);
    } else {
        (void)hipGetLastError();
    }
    return RNNT_OK;
}

int rnnt_engine_greedy_decode_persistent_workspace_bytes(int T, int S, int E, int O, int H, int V, int has_text, size_t *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (T < 1 || S < 1) return fail(RNNT_ERR_INVALID_ARG, "T=%d S=%d", T, S);
    if (int rc = dec_persist_check(T, S, E, O, H, V, has_text, 2)) return rc;
    *out = align_up(dec_persist_workspace_floats(T, S, E, O, H, V, has_text) * 4);
    return RNNT_OK;
}

int rnnt_engine_greedy_decode_tables_bytes(int S, int E, int O, int H, int has_text, size_t *out)
{
    if (!out) return fail(RNNT_ERR_INVALID_ARG, "null size pointer");
    if (S < 1) return fail(RNNT_ERR_INVALID_ARG, "S=%d", S);
    if (int rc = dec_persist_check(8, S, E, O, H, 1024, has_text, 2)) return rc;  // (the tables do not depend on V or T)
    *out = align_up(dec_tables_floats(S, E, O, H, has_text) * 4);
    return RNNT_OK;
}

int rnnt_engine_greedy_decode_build_tables(const rnnt_conv_predictor_params *p, int S, int E, int O, float ln_in_eps, const void *text_W,
                                           const void *text_b, int H, void *tables, size_t tables_bytes, void *stream)
{
    size_t need;
    if (int rc = rnnt_engine_greedy_decode_tables_bytes(S, E, O, H, text_W ? 1 : 0, &need)) return rc;
    if (!p || !tables) return fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    const void *ptrs[] = {p->embedding, p->ln_in_w, p->ln_in_b, p->conv1_w, p->conv1_b, p->conv2_w, p->conv2_b,
                          p->linear_w, p->linear_b, p->ln_out_w, p->ln_out_b};
    for (const void *q : ptrs)
        if (!q || !aligned16(q)) return fail(RNNT_ERR_INVALID_ARG, "null or not 16-byte aligned parameter pointer");
    if ((text_W == nullptr) != (text_b == nullptr) || (text_W && (!aligned16(text_W) || !aligned16(text_b))))
        return fail(RNNT_ERR_INVALID_ARG, "text_W / text_b: both or neither, 16-byte aligned");
    if (!text_W && O != H) return fail(RNNT_ERR_INVALID_ARG, "without text_ln the predictor's output dim (%d) must equal H (%d)", O, H);
    if ((uintptr_t)tables & 255) return fail(RNNT_ERR_INVALID_ARG, "tables must be 256-byte aligned");
    if (tables_bytes < need) return fail(RNNT_ERR_WORKSPACE, "tables %zu < required %zu bytes", tables_bytes, need);
    launch_dec_build_tables(*p, S, E, O, ln_in_eps, (const float *)text_W, (const float *)text_b, H, (float *)tables, (hipStream_t)stream);
    return launch_status("rnnt_engine_greedy_decode_build_tables");
}

int rnnt_engine_greedy_decode_persistent(const void *frames, int64_t frame_stride, int T, const rnnt_conv_predictor_params *p,
                                         int S, int E, int O, float ln_in_eps, float ln_eps, const void *text_W, const void *text_b,
                                         const void *W, const void *bias, int H, int V, int blank, int max_length,
                                         int max_per_frame, const void *tables, int32_t *host_flag, int32_t *state, int32_t *tokens,
                                         void *workspace, size_t ws_bytes, void *stream)
{
    if (T < 1 || S < 1 || max_length < 2) return fail(RNNT_ERR_INVALID_ARG, "T=%d S=%d max_length=%d", T, S, max_length);
    if (int rc = dec_persist_check(T, S, E, O, H, V, text_W ? 1 : 0, max_length)) return rc;
    DecLoopArgs a;
    if (int rc = dec_check_args(frames, frame_stride, T, p, S, E, O, ln_in_eps, ln_eps, text_W, text_b, W, bias, H, V, blank, max_length, max_per_frame,
                                host_flag, state, tokens, workspace, a))
        return rc;
    const size_t need = align_up(dec_persist_workspace_floats(T, S, E, O, H, V, text_W ? 1 : 0) * 4);
    if (ws_bytes < need) return fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return fail(RNNT_ERR_LAUNCH, "cannot query the device's compute-unit count");
    if (dec_persist_groups(V) > cus)  // the loop's workgroups wait for each other: all of them must be resident
        return fail(RNNT_ERR_UNSUPPORTED, "persistent greedy decode needs %d compute units, the device has %d", dec_persist_groups(V), cus);
    if (tables && ((uintptr_t)tables & 255)) return fail(RNNT_ERR_INVALID_ARG, "tables must be 256-byte aligned");
    a.tables = tables;
    if (const int e = launch_dec_persist(a, (hipStream_t)stream))
        return fail(RNNT_ERR_LAUNCH, "persistent greedy decode: cannot raise the kernel's dynamic LDS limit to %zu bytes: %s",
                    dec_persist_lds_bytes(E), hipGetErrorString((hipError_t)e));
    return launch_status("rnnt_engine_greedy_decode_persistent");
}

int rnnt_engine_loss_fwd_bwd(const void *logits, const int32_t *targets, const int32_t *logit_lens,
                             const int32_t *target_lens, int B, int T, int U1, int V, int blank,
                             float clamp, int dtype, float *costs, void *grad_logits,
                             void *workspace, size_t ws_bytes, void *stream)
{
    if (int rc = check_dims(B, T, U1, 4, V, dtype, false)) return rc;
    if (!logits || !targets || !logit_lens || !target_lens || !costs || !workspace)
        return fail(RNNT_ERR_INVALID_ARG, "null pointer argument");
    if (blank < 0 || blank >= V) return fail(RNNT_ERR_INVALID_ARG, "blank=%d outside [0,%d)", blank, V);
    if (!aligned16(logits) || (grad_logits && !aligned16(grad_logits)) || ((uintptr_t)workspace & 255))
        return fail(RNNT_ERR_INVALID_ARG, "pointers must be 16-byte aligned (workspace 256)");
    size_t need;
    rnnt_engine_loss_workspace_bytes(B, T, U1, V, dtype, &need);
    if (ws_bytes < need) return fail(RNNT_ERR_WORKSPACE, "workspace %zu < required %zu bytes", ws_bytes, need);
    const int D = T + U1 - 1;
    const size_t skew = (size_t)B * D * U1;
    char *ws = (char *)workspace;
    float *denom_s = (float *)ws; ws += align_up(skew * 4);
    float *lpb_s = (float *)ws;   ws += align_up(skew * 4);
    float *lpe_s = (float *)ws;   ws += align_up(skew * 4);
    double *alpha_s = (double *)ws; ws += align_up(skew * 8);
    double *beta_s = (double *)ws;  ws += align_up(skew * 8);
    CellCoef *coef = (CellCoef *)ws; ws += align_up((size_t)B * T * U1 * 16);
    unsigned *err = (unsigned *)ws;
    hipStream_t st = (hipStream_t)stream;
    launch_logsoftmax_gather((const float *)logits, targets, logit_lens, target_lens, denom_s, lpb_s,
                             lpe_s, B, T, U1, V, D, blank, st);
    launch_lattice(lpb_s, lpe_s, alpha_s, beta_s, logit_lens, target_lens, costs, B, U1, D, err, st);
    if (grad_logits) {
        launch_coef(alpha_s, beta_s, denom_s, lpb_s, lpe_s, targets, logit_lens, target_lens, coef,
                    B, T, U1, D, 1.0f, st);
        launch_grad_logits((const float *)logits, coef, (float *)grad_logits, (long)B * T * U1, V,
                           blank, clamp, st);
    }
    return launch_status("rnnt_engine_loss_fwd_bwd");
}

int rnnt_engine_joint_loss_fwd_bwd(const void *enc, const int64_t enc_strides[3], const void *pred,
                                   const void *W, const void *bias, const int32_t *targets,
                                   const int32_t *logit_lens, const int32_t *target_lens, int B,
                                   int T, int U1, int H, int V, int blank, float clamp,
                                   float grad_scale, int dtype, float *costs, void *grad_enc,
                                   void *grad_pred, void *grad_W, void *grad_bias, void *workspace,
                                   size_t ws_bytes, void *stream)
{
    return run_fused(ST_ALL, 0, enc, enc_strides, pred, W, bias, targets, logit_lens, target_lens, B, T,
                     U1, H, V, blank, clamp, grad_scale, dtype, costs, grad_enc, grad_pred, grad_W,
                     grad_bias, workspace, ws_bytes, stream);
}

int rnnt_engine_joint_loss_fwd(const void *enc, const int64_t enc_strides[3], const void *pred,
                               const void *W, const void *bias, const int32_t *targets,
                               const int32_t *logit_lens, const int32_t *target_lens, int B, int T,
                               int U1, int H, int V, int blank, int dtype, float *costs,
                               void *workspace, size_t ws_bytes, void *stream)
{
    return run_fused(ST_PROD | ST_FWD | ST_LATTICE, 0, enc, enc_strides, pred, W, bias, targets, logit_lens,
                     target_lens, B, T, U1, H, V, blank, -1.0f, 1.0f, dtype, costs, nullptr, nullptr,
                     nullptr, nullptr, workspace, ws_bytes, stream);
}

int rnnt_engine_run_stages(int stage_mask, int variant, const void *enc, const int64_t enc_strides[3],
                           const void *pred, const void *W, const void *bias, const int32_t *targets,
                           const int32_t *logit_lens, const int32_t *target_lens, int B, int T, int U1,
                           int H, int V, int blank, float clamp, float grad_scale, int dtype,
                           float *costs, void *grad_enc, void *grad_pred, void *grad_W,
                           void *grad_bias, void *workspace, size_t ws_bytes, void *stream)
{
    if (stage_mask <= 0 || stage_mask > ST_ALL) return fail(RNNT_ERR_INVALID_ARG, "stage_mask %d outside [1,255]", stage_mask);
    return run_fused(stage_mask, variant, enc, enc_strides, pred, W, bias, targets, logit_lens, target_lens,
                     B, T, U1, H, V, blank, clamp, grad_scale, dtype, costs, grad_enc, grad_pred, grad_W,
                     grad_bias, workspace, ws_bytes, stream);
}

int rnnt_engine_run_stage(int stage, const void *enc, const int64_t enc_strides[3],
                          const void *pred, const void *W, const void *bias,
                          const int32_t *targets, const int32_t *logit_lens,
                          const int32_t *target_lens, int B, int T, int U1, int H, int V, int blank,
                          float clamp, float grad_scale, int dtype, float *costs, void *grad_enc,
                          void *grad_pred, void *grad_W, void *grad_bias, void *workspace,
                          size_t ws_bytes, void *stream)
{
    if (stage < 0 || stage > 7) return fail(RNNT_ERR_INVALID_ARG, "stage %d outside [0,7]", stage);
    return run_fused(1 << stage, 0, enc, enc_strides, pred, W, bias, targets, logit_lens, target_lens,
                     B, T, U1, H, V, blank, clamp, grad_scale, dtype, costs, grad_enc, grad_pred,
                     grad_W, grad_bias, workspace, ws_bytes, stream);
}

}  // extern "C"
