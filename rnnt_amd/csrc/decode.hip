// decode.hip — device side of the greedy-decode scan.
//
// Reference: rnnt/model.py:108-125 calls joint.single_forward (rnnt/joint.py:44-55) for ONE
// (audio frame, predictor state) pair, takes argmax(dim=-1).item() and decides on the host —
// one device->host sync per audio frame although most frames emit blank.  Here the joint is
// evaluated for a block of consecutive frames against the same predictor state (k_scan_logits)
// and k_argmax_scan reduces the [nframes, V] logits to
//   out[0] = first frame whose argmax is not blank (t0 + nframes if none)
//   out[1] = its token (blank if none)
//   out[2 + k] = argmax of frame t0 + k                 (first index on ties, like torch.argmax)
// so the host synchronises once per emitted token or per all-blank block.
#include "kernels.hpp"
#include "smallgemm.hpp"

// logits[k, v] = tanh(enc[t0+k, :] + pred[:]) . W[v, :] + bias[v] for a handful of frames: M is tiny,
// so the parallelism is over the vocabulary — one workgroup per 32 vocabulary rows (x 32 frames),
// its 4 waves split H and meet in LDS.  fp32 MFMA 32x32x2 with the engine's K permutation: a
// lane's float4 of 4 consecutive h feeds 4 MFMAs, for enc/pred and W alike (W in its natural
// [V,H] layout, no re-packing).  H % 8 == 0.
__global__ __launch_bounds__(256) void k_scan_logits(const float *__restrict__ enc, long enc_st,
                                                     const float *__restrict__ pred,
                                                     const float *__restrict__ W,
                                                     const float *__restrict__ bias,
                                                     float *__restrict__ logits, int K, int H, int V)
{
    __shared__ float s_acc[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    const int v0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int frame = min(k0 + i, K - 1), vrow = min(v0 + i, V - 1);
    const float *er = enc + (long)frame * enc_st + 4 * half;
    const float *pr = pred + 4 * half;
    const float *wr = W + (long)vrow * H + 4 * half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c = wave; c < H / 8; c += 4) {
        const f32x4 e4 = *(const f32x4 *)(er + 8 * c), p4 = *(const f32x4 *)(pr + 8 * c);
        const f32x4 w4 = *(const f32x4 *)(wr + 8 * c);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fast_tanh(e4[s] + p4[s]), w4[s], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int v = v0 + i;
        const float bv = v < V ? bias[v] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float x = ((acc[r] + s_acc[0][r][lane]) + (s_acc[1][r][lane] + s_acc[2][r][lane])) + bv;
            if (k < K && v < V) logits[(long)k * V + v] = x;
        }
    }
}

void launch_scan_logits(const float *enc, long enc_st, const float *pred, const float *W, const float *bias,
                        float *logits, int K, int H, int V, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_logits, dim3((V + 31) / 32, (K + 31) / 32), dim3(256), 0, st, enc, enc_st, pred, W,
                       bias, logits, K, H, V);
}

// One workgroup of 16 waves, a wave per frame (4 waves walked 8 frames each one after the other: a load and twelve
// cross-lane steps of latency per frame, 28 us per call of a 32-frame block).
__global__ __launch_bounds__(1024) void k_argmax_scan(const float *__restrict__ logits, int K, int V,
                                                      int blank, int t0, int32_t *__restrict__ out)
{
    __shared__ int s_tok[128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < K; k += 16) {
        const float *x = logits + (long)k * V;
        float best = RNNT_NEG_INF;
        int bi = 0x7fffffff;
        for (int v = lane * 4; v < V; v += 256) {  // V % 4 == 0
            const f32x4 q = *(const f32x4 *)(x + v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (q[e] > best) { best = q[e]; bi = v + e; }  // increasing v: first index wins ties
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = __shfl_xor(best, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) s_tok[k] = bi;
    }
    __syncthreads();
    if (threadIdx.x < K) out[2 + threadIdx.x] = s_tok[threadIdx.x];
    if (threadIdx.x == 0) {
        int hit = K, tok = blank;
        for (int k = 0; k < K; ++k)
            if (s_tok[k] != blank) { hit = k; tok = s_tok[k]; break; }
        out[0] = t0 + hit;
        out[1] = tok;
    }
}

void launch_argmax_scan(const float *logits, int K, int V, int blank, int t0, int32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_argmax_scan, dim3(1), dim3(1024), 0, st, logits, K, V, blank, t0, out);
}

// ---------------------------------------------------------------------------------------
// Device-resident greedy decode (round 4; SURVEY.md §8f rank 2, second half): the whole loop of reference
// rnnt/model.py:108-125 for one utterance and the stateless ConvPredictor (rnnt/predictor.py:189-229) as a fixed
// kernel sequence per ITERATION with every decision taken on the device — the host enqueues an upper bound of
// iterations and synchronises ONCE per utterance (rnnt_engine_greedy_scan alone still needed one sync per token).
//
// state (int32[8]): [0] t  [1] emitted  [2] ntok (tokens after the leading blank)  [3] done  [4] newtok
// tokens (int32[max_length]): tokens[0] = blank (model.py:100), tokens[1 .. ntok] the decoded ids.
//
// The reference re-runs the predictor on the whole history and keeps the LAST frame; the module is causal (left
// zero padding, rnnt/causalconv.py:28-29), so that frame is a function of the last 7 tokens only, computed here
// incrementally: rings of the last LayerNorm'd embeddings (conv1's input) and conv1 outputs (conv2's input),
// zero-initialised = the left padding.  One row per step: the layers are matrix-VECTOR products (k_dec_gemv: a wave
// per output feature over the [tap][out][in] pack of launch_pack_conv_w), LayerNorms are folded into their
// consumers (every workgroup normalises its input vector itself: a few hundred floats).
//
// Iteration = [conv1 <- LN(embedding[token])] [conv2] [linear] [text_ln(LN(.)) if the joint has one] [scan logits of
// `n` frames from t] [argmax + the loop's bookkeeping]; the predictor kernels return at once when the previous
// iteration emitted nothing (an all-blank block), every kernel when the loop is over.
// ---------------------------------------------------------------------------------------
__global__ void k_dec_init(int32_t *state, int32_t *tokens, float *rings, int nring, int blank)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nring) rings[i] = 0.f;
    if (i == 0) {
        state[0] = 0; state[1] = 0; state[2] = 0; state[3] = 0; state[4] = 1; state[5] = 0; state[6] = 0; state[7] = 0;
        tokens[0] = blank;
    }
}

__device__ __forceinline__ float dec_block_sum(float v, float *red)  // 256 threads; red[4]
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float dec_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// y[o] = act(bias[o] + sum_tap sum_i Wp[(tap * N + o) * K + i] * x_tap[i]),  one wave per o, 4 per workgroup.
// MODE 0: x_tap = ring[(p - (TAPS-1) + tap) & mask] (zeros for positions < 0), p = ntok: the ring's newest entry.
// MODE 2: as 0, but the newest tap is LN(embedding[tokens[p]]) computed here (workgroup 0 also stores it in the ring).
// MODE 1: TAPS = 1, x = LN(vec) (gamma, beta, eps): the LayerNorm of the producer folded in.
// MODE 3: NO product: y = LN(vec) (N == K).
// Output: out + (ring_out ? (p & out_mask) * N : 0).  K % 4 == 0, K <= 1024.
template <int MODE>
__global__ __launch_bounds__(256) void k_dec_gemv(const int32_t *__restrict__ state, const int32_t *__restrict__ tokens,
                                                   const float *__restrict__ Wp, const float *__restrict__ bias,
                                                   const float *__restrict__ in, int in_mask, int taps, int K, int N, int act,
                                                   const float *__restrict__ emb_or_gamma, const float *__restrict__ gamma_or_beta,
                                                   const float *__restrict__ beta, float eps, int S,
                                                   float *__restrict__ out, int ring_out, int out_mask, float *__restrict__ ring_store)
{
    __shared__ float xs[5 * 1024];
    __shared__ float red[4];
    if (state[3] || !state[4]) return;  // the loop is over / nothing new to predict from (workgroup-uniform)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = state[2];
    // this wave's weights (one output feature: taps x K floats, <= 20 float4 per lane) are requested FIRST: their round
    // trip through L2 overlaps the staging / normalisation of the input vector (a chain of dependent loads, reductions and
    // barriers: 6.3 us per layer with the weights behind it, tools/bench_decode.py)
    const int o = blockIdx.x * 4 + wave, oc = o < N ? o : N - 1;
    f32x4 wreg[5][4];
    if (MODE != 3) {
#pragma unroll
        for (int tap = 0; tap < 5; ++tap)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * lane + 256 * j;
                wreg[tap][j] = (tap < taps && i < K) ? *(const f32x4 *)(Wp + ((long)tap * N + oc) * K + i) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    if (MODE == 1 || MODE == 3) {  // x = LN(in) * gamma + beta
        float s = 0.f, s2 = 0.f;
        for (int i = tid; i < K; i += 256) { const float v = in[i]; s += v; }
        const float mean = dec_block_sum(s, red) / K;
        for (int i = tid; i < K; i += 256) { const float d = in[i] - mean; s2 += d * d; }
        const float rstd = rsqrtf(dec_block_sum(s2, red) / K + eps);
        for (int i = tid; i < K; i += 256) {
            const float y = (in[i] - mean) * rstd * emb_or_gamma[i] + gamma_or_beta[i];
            if (MODE == 3) { if (blockIdx.x == 0) out[i] = y; }
            else xs[i] = y;
        }
        if (MODE == 3) return;
    } else {
        for (int tap = 0; tap < taps; ++tap) {
            const int q = p - (taps - 1) + tap;
            if (MODE == 2 && tap == taps - 1) continue;
            const float *src = in + (long)(q & in_mask) * K;
            for (int i = tid; i < K; i += 256) xs[tap * K + i] = q >= 0 ? src[i] : 0.f;
        }
        if (MODE == 2) {  // newest tap: LN(embedding[token])  (rnnt/predictor.py:214-215)
            int tok = tokens[p];
            tok = tok < 0 ? 0 : (tok >= S ? S - 1 : tok);
            const float *e = emb_or_gamma + (long)tok * K;
            float s = 0.f, s2 = 0.f;
            for (int i = tid; i < K; i += 256) s += e[i];
            const float mean = dec_block_sum(s, red) / K;
            for (int i = tid; i < K; i += 256) { const float d = e[i] - mean; s2 += d * d; }
            const float rstd = rsqrtf(dec_block_sum(s2, red) / K + eps);
            for (int i = tid; i < K; i += 256) {
                const float y = (e[i] - mean) * rstd * gamma_or_beta[i] + beta[i];
                xs[(taps - 1) * K + i] = y;
                if (blockIdx.x == 0) ring_store[(long)(p & in_mask) * K + i] = y;
            }
        }
    }
    __syncthreads();
    if (o >= N) return;
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 5; ++tap)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * lane + 256 * j;
            if (tap < taps && i < K) {
                const f32x4 wv = wreg[tap][j], xv = *(const f32x4 *)(xs + tap * K + i);
                acc = fmaf(wv[0], xv[0], acc); acc = fmaf(wv[1], xv[1], acc); acc = fmaf(wv[2], xv[2], acc); acc = fmaf(wv[3], xv[3], acc);
            }
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) {
        float v = acc + (bias ? bias[o] : 0.f);
        if (act) v = dec_gelu(v);
        out[(ring_out ? (long)(p & out_mask) * N : 0) + o] = v;
    }
}

// the scan of `K` frames from t = state[0] (frames past T-1 repeat the last one; k_dec_update ignores them): each
// workgroup's 32 frames x 32 vocabulary entries leave as ONE (max, first index) pair per frame — part[frame][v block] —
// not as logits: k_dec_update then looks at V/32 candidates per frame instead of V values
// LN: `pred` is the predictor's un-normalised output z (no text_ln in the joint): every workgroup applies the output
// LayerNorm itself (H <= 1024 floats; saves the iteration a kernel), keeping the result in LDS.
template <bool LN>
__global__ __launch_bounds__(256) void k_dec_scan_logits(const int32_t *__restrict__ state, const float *__restrict__ enc, long enc_st, int T,
                                                         const float *__restrict__ pred, const float *__restrict__ W,
                                                         const float *__restrict__ bias, float2 *__restrict__ part, int K, int H, int V,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta, float eps)
{
    __shared__ float s_acc[3][16][64];
    __shared__ __attribute__((aligned(16))) float s_pred[LN ? 1024 : 4];
    __shared__ float red[4];
    if (state[3]) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5;
    const int v0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int t0 = state[0];
    const int frame = min(t0 + min(k0 + i, K - 1), T - 1), vrow = min(v0 + i, V - 1);
    const float *er = enc + (long)frame * enc_st + 4 * half;
    if (LN) {  // y = LN(z) * gamma + beta   (rnnt/predictor.py:229)
        const int tid = threadIdx.x;
        float s1 = 0.f, s2 = 0.f;
        for (int j = tid; j < H; j += 256) s1 += pred[j];
        const float mean = dec_block_sum(s1, red) / H;
        for (int j = tid; j < H; j += 256) { const float d = pred[j] - mean; s2 += d * d; }
        const float rstd = rsqrtf(dec_block_sum(s2, red) / H + eps);
        for (int j = tid; j < H; j += 256) s_pred[j] = (pred[j] - mean) * rstd * gamma[j] + beta[j];
        __syncthreads();
    }
    const float *pr = (LN ? (const float *)s_pred : pred) + 4 * half;
    const float *wr = W + (long)vrow * H + 4 * half;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A chunk (8 h: 4 MFMAs, ~256 matrix-pipe cycles) does not cover an L2 round trip and a 32-frame block is only V/32
    // workgroups: the wave's chunks go in GROUPS of 8 with all 24 loads of the next group in flight while the current
    // one is multiplied (two chunks ahead measured 21 us per block at H = 1024: one round trip per chunk).
    const int NC = H / 8;
    struct Ops { f32x4 e, p, w; };
    auto load = [&](Ops &o, int c) {
        const int cc = c < NC ? c : NC - 1;
        o.e = *(const f32x4 *)(er + 8 * cc); o.p = *(const f32x4 *)(pr + 8 * cc); o.w = *(const f32x4 *)(wr + 8 * cc);
    };
    auto load8 = [&](Ops (&g)[8], int c0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) load(g[j], c0 + 4 * j);
    };
    auto mul8 = [&](const Ops (&g)[8], int c0) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + 4 * j < NC) {  // wave-uniform
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fast_tanh(g[j].e[s] + g[j].p[s]), g[j].w[s], acc, 0, 0, 0);
            }
    };
    Ops ga[8], gb[8];
    load8(ga, wave);
    for (int c = wave; c < NC; c += 64) {  // two groups (2 x 8 chunks x 4 waves) per trip
        load8(gb, c + 32);
        mul8(ga, c);
        load8(ga, c + 64);
        mul8(gb, c + 32);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int v = v0 + i;
        const float bv = v < V ? bias[v] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float x = ((acc[r] + s_acc[0][r][lane]) + (s_acc[1][r][lane] + s_acc[2][r][lane])) + bv;
            if (v >= V) x = RNNT_NEG_INF;
            const float M = half_max_dpp(x, half);  // every lane of the half: the maximum of the frame's 32 entries
            const unsigned long long hit = __ballot(x == M);  // first lane of the half that holds it = the lowest index
            const unsigned h32 = half ? (unsigned)(hit >> 32) : (unsigned)hit;
            if (i == 0 && k < K) part[(long)k * gridDim.x + blockIdx.x] = float2{M, __int_as_float(v0 + __builtin_ctz(h32))};
        }
    }
}

// argmax of every scanned frame from its V/32 candidates (first index on ties, like torch.argmax: candidates are in
// increasing index order), then the loop's bookkeeping (rnnt/model.py:108-125; the host-side twin: rnnt_amd/model.py):
//   hit = first frame of [t, t + n) whose argmax is not blank, n = min(K, T - t)
//   none: t += n, emitted = 0                                  (an all-blank block)
//   else: t = hit (emitted = 0 if hit > t), append the token, ++emitted; emitted == max_per_frame: ++t, emitted = 0
//   done = t >= T or 1 + ntok >= max_length
__global__ __launch_bounds__(128) void k_dec_update(const float2 *__restrict__ part, int K, int NB, int blank, int T, int max_length,
                                                    int max_per_frame, int32_t *__restrict__ state, int32_t *__restrict__ tokens,
                                                    int32_t *host_flag)
{
    __shared__ int s_tok[128];
    if (state[3]) return;
    if ((int)threadIdx.x < K) {
        const float2 *p = part + (long)threadIdx.x * NB;
        float best = RNNT_NEG_INF;
        int bi = 0;
        for (int j = 0; j < NB; ++j) {
            const float2 c = p[j];
            if (c.x > best) { best = c.x; bi = __float_as_int(c.y); }
        }
        s_tok[threadIdx.x] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = state[0], emitted = state[1], ntok = state[2];
        const int n = min(K, T - t);
        int hit = -1;
        for (int k = 0; k < n; ++k)
            if (s_tok[k] != blank) { hit = k; break; }
        int newtok = 0;
        if (hit < 0) { t += n; emitted = 0; }
        else {
            if (hit > 0) emitted = 0;
            t += hit;
            tokens[++ntok] = s_tok[hit];
            newtok = 1;
            if (++emitted >= max_per_frame) { ++t; emitted = 0; }
        }
        state[0] = t; state[1] = emitted; state[2] = ntok; state[4] = newtok;
        const int done = (t >= T || ntok + 1 >= max_length) ? 1 : 0;
        state[3] = done;
        state[5] += 1;  // iterations that did work (diagnostic)
        // the host's cue to stop enqueueing iterations (mapped pinned memory, polled without a synchronisation)
        if (done && host_flag) __hip_atomic_store(host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

size_t dec_loop_workspace_floats(int H, int V, int E, int O, int nframes)
{
    // x1 ring [4][E], g1 ring [8][E], g2 [E], z [O], y [O], pvec [H], candidates [nframes][V/32] x 2, conv packs [3+5][E][E]
    return (size_t)4 * E + 8 * E + E + 2 * (size_t)O + H + (size_t)nframes * ((V + 31) / 32) * 2 + (size_t)8 * E * E + 64;
}

void launch_dec_loop(const DecLoopArgs &a, hipStream_t st)
{
    float *ws = (float *)a.workspace;
    const int E = a.E, O = a.O, H = a.H, V = a.V, n = a.scan_frames;
    float *x1ring = ws, *g1ring = x1ring + 4 * (size_t)E, *g2 = g1ring + 8 * (size_t)E, *z = g2 + E, *y = z + O, *pvec = y + O;
    const int NB = (V + 31) / 32;
    float2 *part = (float2 *)(pvec + ((H + 15) / 16) * 16);
    float *wp1 = (float *)part + (((size_t)n * NB * 2 + 3) & ~(size_t)3), *wp2 = wp1 + (size_t)3 * E * E;  // 16-byte aligned whatever n * NB
    const int nring = 12 * E;
    if (a.init) {
        hipLaunchKernelGGL(k_dec_init, dim3((nring + 255) / 256), dim3(256), 0, st, a.state, a.tokens, x1ring, nring, a.blank);
        launch_pack_conv_w(a.p.conv1_w, wp1, E, E, 3, st);  // [tap][out][in]
        launch_pack_conv_w(a.p.conv2_w, wp2, E, E, 5, st);
    }
    const float *nul = nullptr;
    for (int it = 0; it < a.iterations; ++it) {
        // g1[p] = gelu(conv1(LN(embedding[token p]), x1[p-1], x1[p-2]))            predictor.py:214-219 (dropout: eval)
        hipLaunchKernelGGL(k_dec_gemv<2>, dim3((E + 3) / 4), dim3(256), 0, st, a.state, a.tokens, wp1, a.p.conv1_b, x1ring, 3, 3, E, E, 1,
                           a.p.embedding, a.p.ln_in_w, a.p.ln_in_b, a.ln_in_eps, a.S, g1ring, 1, 7, x1ring);
        // g2 = gelu(conv2(g1[p-4 .. p]))                                            predictor.py:222-223
        hipLaunchKernelGGL(k_dec_gemv<0>, dim3((E + 3) / 4), dim3(256), 0, st, a.state, a.tokens, wp2, a.p.conv2_b, g1ring, 7, 5, E, E, 1,
                           nul, nul, nul, 0.f, 0, g2, 0, 0, (float *)nullptr);
        // z = linear(g2)                                                            predictor.py:228
        hipLaunchKernelGGL(k_dec_gemv<0>, dim3((O + 3) / 4), dim3(256), 0, st, a.state, a.tokens, a.p.linear_w, a.p.linear_b, g2, 0, 1, E, O, 0,
                           nul, nul, nul, 0.f, 0, z, 0, 0, (float *)nullptr);
        // the joint's text input: text_ln(LN(z)) (rnnt/joint.py:28-30), or LN(z) itself — then applied by the scan's workgroups
        if (a.text_W) {
            hipLaunchKernelGGL(k_dec_gemv<1>, dim3((H + 3) / 4), dim3(256), 0, st, a.state, a.tokens, a.text_W, a.text_b, z, 0, 1, O, H, 0,
                               a.p.ln_out_w, a.p.ln_out_b, nul, a.ln_eps, 0, pvec, 0, 0, (float *)nullptr);
            hipLaunchKernelGGL(k_dec_scan_logits<false>, dim3(NB, (n + 31) / 32), dim3(256), 0, st, a.state, a.frames, a.frame_stride, a.T,
                               pvec, a.W, a.bias, part, n, H, V, nul, nul, 0.f);
        } else {
            hipLaunchKernelGGL(k_dec_scan_logits<true>, dim3(NB, (n + 31) / 32), dim3(256), 0, st, a.state, a.frames, a.frame_stride, a.T,
                               z, a.W, a.bias, part, n, H, V, a.p.ln_out_w, a.p.ln_out_b, a.ln_eps);
        }
        hipLaunchKernelGGL(k_dec_update, dim3(1), dim3(128), 0, st, part, n, NB, a.blank, a.T, a.max_length, a.max_per_frame, a.state, a.tokens, a.host_flag);
    }
}

// ---------------------------------------------------------------------------------------
// Round 5: the same loop as ONE persistent launch (rnnt_engine_greedy_decode_persistent; reference rnnt/model.py:108-125 +
// rnnt/predictor.py:189-229).  The chain above spends ~8 us per dependent kernel (exec ~5 + boundary) on products that take
// a fraction of a microsecond; here G <= 128 workgroups (one per CU, all resident) stay for the whole utterance and hand
// their results to each other through 8-byte {value, tag} granules — the data is the flag: ONE agent-scope (sc1,
// write-through) store per value, consumers sweep with agent-scope loads until every tag is the iteration's
// (tools/grid_barrier_probe.hip: a counter or flag barrier costs 1.4 us at 16 workgroups and 50 ns per workgroup more; a
// sweep of <= 1024 granules by 64 workgroups is one round trip).  Every phase is an all-to-all hand-off, so the sweeps are the
// loop's only synchronisation; a buffer is rewritten only after every workgroup has published the NEXT hand-off (candidates:
// two buffers, the all-blank iterations have no hand-off between two of them).
//
// The chain per emitted token, shortened by algebra (every workgroup keeps the loop's state itself — t, emitted, the last
// three tokens, the ring of the last conv1 outputs — and derives it redundantly from the same candidates):
//   conv1 (3 taps on LN(embedding[token]))  = three TABLE rows  A_j[token] = W1_j LN(emb[token])  (S x E each, built per call by
//                                             three GEMMs): g1[p] = gelu(b1 + A_2[tok_p] + A_1[tok_p-1] + A_0[tok_p-2]) costs
//                                             every workgroup E adds, no hand-off;
//   conv2 (5 taps on g1[p-4..p])            = the four old taps were accumulated while the PREVIOUS token's hand-offs were in
//                                             flight (`pre`), the newest tap is one E x E matrix-vector product  -> hand-off 1 (g2)
//   linear (+ output LayerNorm + text_ln)   = z = W_l g2 + b_l; with joint.text_ln the projection is folded into the same
//                                             phase: q = (W_t diag(gamma) W_l) g2 + (W_t gamma) b_l beside z, whose mean and
//                                             variance travel as per-workgroup (mean, M2) pairs; text = rstd (q - mean r) + c,
//                                             r = rowsum(W_t gamma), c = W_t beta + b_t              -> hand-off 2 (z, or q + stats)
//   scan of 16 frames from t                = hidden = 1 - 2 / (1 + exp(2 enc) exp(2 text)) (exp(2 enc) for all frames by one
//                                             kernel before the loop, exp(2 text) once per token), logits for 16 vocabulary
//                                             entries per workgroup on v_mfma_f32_16x16x4_f32 (4 waves split H), one
//                                             (max, first index) candidate per frame and workgroup      -> hand-off 3 (candidates)
//   argmax over the candidates + the loop's bookkeeping: every workgroup for itself.
// All-blank iterations run the scan and hand-off 3 only.  Arithmetic: fp32 throughout (fp32 MFMA for the logits as in
// k_dec_scan_logits); sums are associated differently from the per-kernel chain (tables, folded projection), i.e. logits
// differ in the last bits — token lists are equal wherever the per-frame loop's argmax is not a rounding-level tie
// (tests/test_gpu_parity.py::test_greedy_decode_persistent_matches_per_frame_loop).
// Every spin is bounded: a workgroup that gives up writes state[7] and leaves, the others follow.
// ---------------------------------------------------------------------------------------
typedef unsigned long long dp_u64;
// Diagnostic build only (make EXTRA=-DDP_STAMPS): s_memtime spent per phase, summed over the iterations, by workgroup 0 — 16 counters
// behind the candidate granules (tools/exp_decode_persist.py prints them)
#ifdef DP_STAMPS
#define DP_T(k)                                                                                  \
    do {                                                                                         \
        dp_u64 t_;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        dpt[k] += t_ - dpt0;                                                                     \
        dpt0 = t_;                                                                               \
    } while (0)
#else
#define DP_T(k)
#endif
#define DP_FRAMES 16
#define DP_TOKS 2048
#define DP_FACTOR_RANGE 30.f      // |x| up to which exp(2 x) is used as a factor of the hidden value (k_dp_exp_frames below)
#define DP_CODE_FRAME_RANGE 10   // state[7]: an audio frame, or (11) a text vector, beyond it — not a decode, the caller takes the other loop
#define DP_CODE_TEXT_RANGE 11
#ifndef DP_SPIN_LIMIT
#define DP_SPIN_LIMIT (1 << 18)  // polls (~1.5 us each: ~0.4 s) before a sweep gives up — several times the longest kernel another stream can hold a compute unit with (a cfg4 training step: 0.2 s); round 5: 2^21 = 3 s of stall before the fallback
#endif
// -DDP_TEST_STALL=n (tools/check_decode_giveup.sh, with a small -DDP_SPIN_LIMIT): workgroup 1 leaves at iteration n without publishing — the
// other workgroups must give up at their next sweep, report it in state[7] and leave: the "every spin is bounded" claim, exercised

__device__ __forceinline__ dp_u64 dp_load(const dp_u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dp_store(dp_u64 *p, unsigned tag, float v)
{
    __hip_atomic_store(p, ((dp_u64)tag << 32) | (dp_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// NT threads (tid < NT): dst[i] = the value of granule i, i < n <= 8 NT, once its tag is `tag`.  false: gave up.
template <int NT>
__device__ __forceinline__ bool dp_sweep(const dp_u64 *gr, int n, unsigned tag, float *dst, int tid)
{
    unsigned pending = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (tid + NT * j < n) pending |= 1u << j;
    int spins = 0;
    while (pending) {
        dp_u64 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((pending >> j) & 1) x[j] = dp_load(gr + tid + NT * j);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (((pending >> j) & 1) && (unsigned)(x[j] >> 32) == tag) {
                dst[tid + NT * j] = __uint_as_float((unsigned)x[j]);
                pending &= ~(1u << j);
            }
        if (pending) {
            if (++spins > DP_SPIN_LIMIT) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return true;
}

// a wave's weight row (K floats, K % 4 == 0, K <= 256 KR): lane l holds floats 4l + 256j .. +3, j < KR
template <int KR>
__device__ __forceinline__ void dp_row_load(f32x4 (&w)[KR], const float *row, int K, int lane, bool valid)
{
#pragma unroll
    for (int j = 0; j < KR; ++j) {
        const int i = 4 * lane + 256 * j;
        w[j] = (valid && i < K) ? *(const f32x4 *)(row + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
template <int KR>
__device__ __forceinline__ float dp_row_mul(const f32x4 (&w)[KR], const float *x, int K, int lane)  // x: LDS, 16-byte aligned; the lane's partial sum
{
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < KR; ++j) {
        const int i = 4 * lane + 256 * j;
        if (i < K) {
            const f32x4 xv = *(const f32x4 *)(x + i);
            acc = fmaf(w[j][0], xv[0], acc); acc = fmaf(w[j][1], xv[1], acc); acc = fmaf(w[j][2], xv[2], acc); acc = fmaf(w[j][3], xv[3], acc);
        }
    }
    return acc;
}
// every lane: the wave's sum — five DPP adds inside the rows and across them (half_sum_dpp + row_bcast31) and one v_readlane,
// ~70 cycles; six dependent ds_bpermute steps (__shfl_xor) are ~700, and a token's chain holds seven such reductions
__device__ __forceinline__ float dp_wave_sum(float v)
{
    v = half_sum_dpp(v, 0);
    v += dpp_mov<0x143, 0xc>(0.f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int N>
__device__ __forceinline__ void dp_wave_sums(float (&v)[N])
{
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = dp_wave_sum(v[k]);
}

// Stores: a hand-off's values are staged in LDS and leave by ONE store instruction of wave 3 (a store per row, each behind the
// compiler's vmcnt(0) for the previous one, cost a write-through round trip per row); every wave multiplies rows and sweeps.
// (Measured: keeping wave 3 out of the products and sweeps so that no sweeping wave has a store outstanding did not shorten the
// sweeps — 1.5 / 2.3 / 3 us with either arrangement — and lengthened the row phases: 4.06 -> 4.50 ms.)
// KF > 0: the workgroup's weight rows stay in REGISTERS for the whole utterance — E <= 256 KF, per wave at most 2 conv2 rows (x 5 taps),
// 4 linear rows and 2 folded text rows (dec_persist_kf picks; the reference's widths: KF = 2).  KF = 0: any size, rows loaded at every
// use (each load a round trip through L2 on the token's critical path).  The workgroup's first vocabulary block (16 x H floats, in MFMA
// fragment order), the LayerNorm / folded-text vectors and conv1's bias stay in LDS either way.
// Frames: lane i16 of the scan holds the frame of [t, t + 16) whose number is i16 mod 16, so when t moves to the hit's frame only
// the lanes whose frame left the window load (the window's 16 x H floats of exp(2 enc) are 64 KB per workgroup: at the L1's 64 B/clk
// a full reload is 1 024 cycles in front of the token's table rows).
template <int KF>
__global__ __launch_bounds__(256) void k_dec_persist(DecPersistArgs a)
{
    constexpr bool RES = KF > 0;
    constexpr int KR = RES ? KF : 4;
    extern __shared__ __attribute__((aligned(16))) float dp_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, g = blockIdx.x;
    const int E = a.E, O = a.O, H = a.H, V = a.V, T = a.T;
    float *s_ring = dp_lds;                  // g1[p & 7][E]: conv2's input ring (zeros = the left padding)
    float *s_x = s_ring + 8 * E;             // hand-off 1 as swept: g2
    float *s_y = s_x + 1024;                 // hand-off 2 as swept: z, or q
    float *s_P = s_y + 1024;                 // exp(2 text)
    float *s_g = s_P + 1024;                 // output LayerNorm's gamma, or r with text_ln
    float *s_b = s_g + 1024;                 // beta, or c
    float *s_c1 = s_b + 1024;                // conv1's bias
    float *s_Wf = s_c1 + 1024;               // the first vocabulary block's W fragments [wave][16 chunks][lane] x 4 floats, zeros past H
    float *s_part = s_Wf + 16 * 1024;        // [4][16][17]
    float *s_pre = s_part + 4 * DP_FRAMES * 17;  // [64] (KF = 0)
    float *s_pub = s_pre + 64;               // [128] values on their way out: rows of hand-off 1 / 2 ([64 + r]: folded text rows)
    float *s_st = s_pub + 128;               // [256]
    dp_u64 *s_cand = (dp_u64 *)(s_st + 256);  // [16] candidate granules on their way out
    int *s_tok = (int *)(s_cand + DP_FRAMES);  // [4][16] the waves' best index per frame
    float *s_wbx = (float *)(s_tok + 4 * DP_FRAMES);  // [4][16] and its logit
    int *s_failp = (int *)(s_wbx + 4 * DP_FRAMES);
    int *s_toks = s_failp + 4;                // [DP_TOKS] the decoded ids, written out at the end
#define s_fail (*s_failp)
    const int rpA = (E + G - 1) / G, rpZ = (O + G - 1) / G, rpQ = (H + G - 1) / G;  // rows per workgroup (<= 64)
    const int HW = H / 4, NCH = HW / 16;                                             // a wave's share of H, in MFMA chunks of 16
    const int i16 = lane & 15, kq = lane >> 4;
    const int fr = tid >> 4, vv = tid & 15;
    for (int j = tid; j < 8 * E; j += 256) s_ring[j] = 0.f;
    for (int j = tid; j < H; j += 256) {
        s_g[j] = a.has_text ? a.rvec[j] : a.gamma[j];
        s_b[j] = a.has_text ? a.cvec[j] : a.beta[j];
    }
    for (int j = tid; j < E; j += 256) s_c1[j] = a.conv1_b[j];
    if (tid < 64) s_pre[tid] = 0.f;
    if (tid == 0) s_fail = 0;
    const bool has_v = g * 16 < V;  // (workgroups past the vocabulary's blocks only sweep)
    {
        const float *wr = a.W + (long)min(g * 16 + i16, V - 1) * H + wave * HW + 4 * kq;
        for (int c = 0; c < 16; ++c)
            *(f32x4 *)(s_Wf + ((wave * 16 + c) * 64 + lane) * 4) = (has_v && c < NCH) ? *(const f32x4 *)(wr + 16 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int j = tid; j < 1024; j += 256) s_P[j] = 0.f;  // (chunks past H multiply zeros of s_Wf: keep their other operand finite)
    const float bias0 = (has_v && g * 16 + vv < V) ? a.bias[g * 16 + vv] : 0.f;
    // resident rows: conv2 rows wave + 4 ra (5 taps), linear rows wave + 4 k (k < 4), folded text rows wave + 4 k (k < 2)
    f32x4 wA[RES ? 2 : 1][RES ? 5 : 1][KR], wB[RES ? 6 : 1][KR];
    float cb2[2] = {0.f, 0.f}, bB[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, prA[2] = {0.f, 0.f};
    if constexpr (RES) {
#pragma unroll
        for (int ra = 0; ra < 2; ++ra) {
            const int r = wave + 4 * ra, o = g * rpA + r;
            const bool ok = r < rpA && o < E;
#pragma unroll
            for (int tap = 0; tap < 5; ++tap) dp_row_load<KR>(wA[ra][tap], a.wp2 + ((long)tap * E + (ok ? o : 0)) * E, E, lane, ok);
            cb2[ra] = ok ? a.conv2_b[o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = wave + 4 * k, o = g * rpZ + r;
            const bool ok = r < rpZ && o < O;
            dp_row_load<KR>(wB[k], a.Wl + (long)(ok ? o : 0) * E, E, lane, ok);
            bB[k] = ok ? a.bl[o] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = wave + 4 * k, h = g * rpQ + r;
            const bool ok = a.has_text && r < rpQ && h < H;
            dp_row_load<KR>(wB[4 + k], a.M + (long)(ok ? h : 0) * E, E, lane, ok);
            bB[4 + k] = ok ? a.dvec[h] : 0.f;
        }
    }
    __syncthreads();

    int t = 0, emitted = 0, ntok = 0, newtok = 1, done = 0;
    int tk0 = a.blank, tk1 = -1, tk2 = -1;  // tokens at positions p, p-1, p-2 (-1: before the start)
    int it = 0, code = 0;
    int fheld = -1;  // the frame whose exp(2 enc) fragments this lane holds
    f32x4 ef[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) ef[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef DP_STAMPS
    dp_u64 dpt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dpt0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dpt0)::"memory");
#endif

    if (__hip_atomic_load(a.state + 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) done = 1;  // k_dp_exp_frames: a frame beyond the factored form's range
    for (it = 1; it <= a.max_iters && !done; ++it) {
        // ---- exp(2 enc) fragments: lane i16 holds the frame of [t, t + 16) that is i16 mod 16 (frames past T-1 repeat the last; the
        // bookkeeping ignores them); only lanes whose frame changed load
        {
            const int fnew = min(t + ((i16 - t) & 15), T - 1);
            if (fnew != fheld) {
                const float *er = a.Eenc + (long)fnew * H + wave * HW + 4 * kq;
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < NCH) ef[c] = *(const f32x4 *)(er + 16 * c);
                fheld = fnew;
            }
        }
        DP_T(0);
#ifdef DP_TEST_STALL
        if (g == 1 && it == DP_TEST_STALL) { code = 9; break; }
#endif
        if (newtok) {
            const int p = ntok;
            // ---- g1[p] from the three conv1 tables
            {
                const int c0 = min(max(tk0, 0), a.S - 1), c1 = min(max(tk1, 0), a.S - 1), c2 = min(max(tk2, 0), a.S - 1);
                for (int e = tid; e < E; e += 256) {
                    const float a2 = a.A2[(long)c0 * 3 * E + e], a1 = tk1 >= 0 ? a.A1[(long)c1 * 3 * E + e] : 0.f, a0 = tk2 >= 0 ? a.A0[(long)c2 * 3 * E + e] : 0.f;
                    s_ring[(p & 7) * E + e] = dec_gelu(((s_c1[e] + a2) + a1) + a0);
                }
            }
            __syncthreads();
            // ---- hand-off 1: g2[o] = gelu(b2 + pre[o] + W2_4[o] . g1[p])        rnnt/predictor.py:222-223
            {
                if constexpr (RES) {
                    float acc[2];
#pragma unroll
                    for (int ra = 0; ra < 2; ++ra) acc[ra] = dp_row_mul<KR>(wA[ra][4], s_ring + (p & 7) * E, E, lane);
                    dp_wave_sums<2>(acc);
                    const float v = lane == 0 ? cb2[0] + prA[0] + acc[0] : cb2[1] + prA[1] + acc[1];
                    const int r = wave + 4 * lane;
                    if (lane < 2 && r < rpA) s_pub[r] = dec_gelu(v);
                } else {
                    for (int r = wave; r < rpA; r += 4) {
                        const int o = g * rpA + r;
                        if (o < E) {  // wave-uniform
                            f32x4 w[KR];
                            dp_row_load<KR>(w, a.wp2 + ((long)4 * E + o) * E, E, lane, true);
                            const float acc = dp_wave_sum(dp_row_mul<KR>(w, s_ring + (p & 7) * E, E, lane));
                            if (lane == 0) s_pub[r] = dec_gelu(a.conv2_b[o] + s_pre[r] + acc);
                        }
                    }
                }
            }
            __syncthreads();
            if (wave == 3 && lane < rpA && g * rpA + lane < E) dp_store(a.g2g + g * rpA + lane, (unsigned)it, s_pub[lane]);
            DP_T(1);
            if (!dp_sweep<256>(a.g2g, E, (unsigned)it, s_x, tid)) s_fail = 1;
            __syncthreads();
            if (s_fail) { code = 1; break; }
            DP_T(2);
            // ---- hand-off 2: z = W_l g2 + b_l  (rnnt/predictor.py:228); with text_ln also q = M g2 + d
            {
                if constexpr (RES) {
                    float acc[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) acc[k] = dp_row_mul<KR>(wB[k], s_x, E, lane);
                    dp_wave_sums<6>(acc);
                    float v = acc[0] + bB[0];
#pragma unroll
                    for (int k = 1; k < 6; ++k) v = lane == k ? acc[k] + bB[k] : v;
                    const bool isq = lane >= 4;
                    const int r = wave + 4 * (isq ? lane - 4 : lane);
                    if (lane < 6 && (isq ? (a.has_text && r < rpQ) : r < rpZ)) s_pub[(isq ? 64 : 0) + r] = v;
                } else {
                    for (int r = wave; r < rpZ; r += 4) {
                        const int o = g * rpZ + r;
                        if (o < O) {
                            f32x4 w[KR];
                            dp_row_load<KR>(w, a.Wl + (long)o * E, E, lane, true);
                            const float acc = dp_wave_sum(dp_row_mul<KR>(w, s_x, E, lane));
                            if (lane == 0) s_pub[r] = acc + a.bl[o];
                        }
                    }
                    if (a.has_text)
                        for (int r = wave; r < rpQ; r += 4) {
                            const int h = g * rpQ + r;
                            if (h < H) {
                                f32x4 w[KR];
                                dp_row_load<KR>(w, a.M + (long)h * E, E, lane, true);
                                const float acc = dp_wave_sum(dp_row_mul<KR>(w, s_x, E, lane));
                                if (lane == 0) s_pub[64 + r] = acc + a.dvec[h];
                            }
                        }
                }
            }
            __syncthreads();
            if (wave == 3) {
                if (a.has_text) {  // this workgroup's z rows leave as (mean, M2), combined by every consumer (Chan et al.); q rows as they are
                    const int n = min(rpZ, O - g * rpZ);  // may be <= 0 for the last workgroups
                    const float zv = lane < n ? s_pub[lane] : 0.f;
                    const float mg = n > 0 ? dp_wave_sum(zv) / n : 0.f;
                    const float d2 = dp_wave_sum(lane < n ? (zv - mg) * (zv - mg) : 0.f);
                    if (lane < 2) dp_store(a.sg + 2 * g + lane, (unsigned)it, lane ? d2 : mg);
                    if (lane < rpQ && g * rpQ + lane < H) dp_store(a.zg + g * rpQ + lane, (unsigned)it, s_pub[64 + lane]);
                } else if (lane < rpZ && g * rpZ + lane < O) dp_store(a.zg + g * rpZ + lane, (unsigned)it, s_pub[lane]);
            }
            DP_T(3);
            // ---- while hand-off 2 is in flight: the four old taps of the NEXT token's conv2 (its newest tap needs the token)
            {
                if constexpr (RES) {
                    float acc[2] = {0.f, 0.f};
#pragma unroll
                    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
                        for (int ra = 0; ra < 2; ++ra) acc[ra] += dp_row_mul<KR>(wA[ra][tap], s_ring + ((p - 3 + tap) & 7) * E, E, lane);
                    dp_wave_sums<2>(acc);
                    prA[0] = acc[0]; prA[1] = acc[1];
                } else {
                    for (int r = wave; r < rpA; r += 4) {
                        const int o = g * rpA + r;
                        if (o < E) {
                            float acc = 0.f;
#pragma unroll
                            for (int tap = 0; tap < 4; ++tap) {
                                f32x4 w[KR];
                                dp_row_load<KR>(w, a.wp2 + ((long)tap * E + o) * E, E, lane, true);
                                acc += dp_row_mul<KR>(w, s_ring + ((p - 3 + tap) & 7) * E, E, lane);
                            }
                            acc = dp_wave_sum(acc);
                            if (lane == 0) s_pre[r] = acc;
                        }
                    }
                }
            }
            DP_T(4);
            // ---- the joint's text input and exp(2 text)
            if (!dp_sweep<256>(a.zg, H, (unsigned)it, s_y, tid)) s_fail = 1;
            if (a.has_text && !dp_sweep<256>(a.sg, 2 * G, (unsigned)it, s_st, tid)) s_fail = 1;
            __syncthreads();
            if (s_fail) { code = 2; break; }
            DP_T(5);
            float mean, rstd;  // every wave for itself, in the same order: the same bits
            if (a.has_text) {  // (mean, M2) of z from the G partial pairs (Chan et al.'s combination)
                float s1 = 0.f;
                for (int j = lane; j < G; j += 64) s1 += s_st[2 * j] * (float)max(0, min(rpZ, O - j * rpZ));
                mean = dp_wave_sum(s1) / O;
                float m2 = 0.f;
                for (int j = lane; j < G; j += 64) {
                    const float d = s_st[2 * j] - mean;
                    m2 += s_st[2 * j + 1] + d * d * (float)max(0, min(rpZ, O - j * rpZ));
                }
                rstd = rsqrtf(dp_wave_sum(m2) / O + a.eps);
                for (int h = tid; h < H; h += 256) {  // text = rstd (q - mean r) + c
                    const float x = rstd * (s_y[h] - mean * s_g[h]) + s_b[h];
                    if (!(fabsf(x) <= DP_FACTOR_RANGE)) s_fail = 2;
                    s_P[h] = __builtin_amdgcn_exp2f(fminf(fmaxf(x, -DP_FACTOR_RANGE), DP_FACTOR_RANGE) * (2.0f * RNNT_LOG2E));
                }
            } else {  // text = LN(z) gamma + beta   (rnnt/predictor.py:229; O == H)
                float s1 = 0.f;
                for (int h = 4 * lane; h < H; h += 256) { const f32x4 v = *(const f32x4 *)(s_y + h); s1 += (v[0] + v[1]) + (v[2] + v[3]); }
                mean = dp_wave_sum(s1) / H;
                float s2 = 0.f;
                for (int h = 4 * lane; h < H; h += 256) {
                    const f32x4 v = *(const f32x4 *)(s_y + h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; s2 = fmaf(d, d, s2); }
                }
                rstd = rsqrtf(dp_wave_sum(s2) / H + a.eps);
                for (int h = tid; h < H; h += 256) {
                    const float x = (s_y[h] - mean) * rstd * s_g[h] + s_b[h];
                    if (!(fabsf(x) <= DP_FACTOR_RANGE)) s_fail = 2;
                    s_P[h] = __builtin_amdgcn_exp2f(fminf(fmaxf(x, -DP_FACTOR_RANGE), DP_FACTOR_RANGE) * (2.0f * RNNT_LOG2E));
                }
            }
            __syncthreads();
            if (s_fail) { code = DP_CODE_TEXT_RANGE; break; }  // every workgroup computes the same text vector: all of them leave here
            DP_T(6);
        }
        // ---- hand-off 3: the scan.  Workgroup g owns vocabulary blocks g, g + G, ... of 16 entries
        {
            float best = RNNT_NEG_INF;
            int bloc = 0;  // index inside the workgroup's blocks: 16 m + entry
            for (int m = 0; (g + G * m) * 16 < V; ++m) {
                const int v0 = (g + G * m) * 16;
                const float *wr = a.W + (long)min(v0 + i16, V - 1) * H + wave * HW + 4 * kq;
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                // all of a half's fragments are requested before the first MFMA (one select of LDS / global per chunk made every
                // chunk a flat load with its own round trip: 4 us per scan); chunks past H multiply zeros
                auto half = [&](const int c0) {
                    f32x4 wv[8], pv[8];
                    if (m == 0) {
#pragma unroll
                        for (int c = 0; c < 8; ++c) wv[c] = *(const f32x4 *)(s_Wf + ((wave * 16 + c0 + c) * 64 + lane) * 4);
                    } else {
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const f32x4 x = *(const f32x4 *)(wr + 16 * min(c0 + c, NCH - 1));
                            wv[c] = c0 + c < NCH ? x : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) pv[c] = *(const f32x4 *)(s_P + ((wave * HW + 16 * (c0 + c) + 4 * kq) & 1023));
                    // a chunk's four hidden values side by side — fma, rcp, fma each four times, so that no instruction waits for the one
                    // before it — and the NEXT chunk's between this chunk's MFMAs.  Spelled as asm: the compiler's scheduler puts every value's
                    // fma -> rcp -> fma -> mfma back into one dependent chain (~100 cycles per value)
                    auto hid4 = [&](f32x4 &h, const f32x4 &e, const f32x4 &p) {
                        asm volatile("v_fma_f32 %0, %4, %8, 1.0\n\tv_fma_f32 %1, %5, %9, 1.0\n\tv_fma_f32 %2, %6, %10, 1.0\n\tv_fma_f32 %3, %7, %11, 1.0\n\t"
                                     "v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
                                     "s_nop 0\n\t"
                                     "v_fma_f32 %0, %0, -2.0, 1.0\n\tv_fma_f32 %1, %1, -2.0, 1.0\n\tv_fma_f32 %2, %2, -2.0, 1.0\n\tv_fma_f32 %3, %3, -2.0, 1.0"
                                     : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3])
                                     : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
                    };
                    f32x4 hid, hnx;
                    hid4(hid, ef[c0], pv[0]);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        __builtin_amdgcn_sched_barrier(0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hid[0], wv[c][0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hid[1], wv[c][1], acc1, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        if (c < 7) hid4(hnx, ef[c0 + c + 1], pv[c + 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(hid[2], wv[c][2], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(hid[3], wv[c][3], acc1, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        hid = hnx;
                    }
                };
                half(0);
                if (NCH > 8) half(8);  // wave-uniform
                DP_T(10);
                if (m > 0) __syncthreads();  // the previous block's partials have been read
#pragma unroll
                for (int r = 0; r < 4; ++r) s_part[(wave * DP_FRAMES + 4 * kq + r) * 17 + i16] = acc0[r] + acc1[r];
                __syncthreads();
                const int v = v0 + vv;
                const float bv = m == 0 ? bias0 : (v < V ? a.bias[v] : 0.f);
                float x = v < V ? ((s_part[(0 * DP_FRAMES + fr) * 17 + vv] + s_part[(1 * DP_FRAMES + fr) * 17 + vv]) +
                                   (s_part[(2 * DP_FRAMES + fr) * 17 + vv] + s_part[(3 * DP_FRAMES + fr) * 17 + vv])) + bv
                                : RNNT_NEG_INF;
                int xi = 16 * m + vv;
#pragma unroll
                for (int mm = 8; mm >= 1; mm >>= 1) {  // the 16 lanes of a row: max, lowest index on ties
                    const float ox = __shfl_xor(x, mm, 64);
                    const int oi = __shfl_xor(xi, mm, 64);
                    if (ox > x || (ox == x && oi < xi)) { x = ox; xi = oi; }
                }
                if (x > best) { best = x; bloc = xi; }  // later blocks hold higher indices: strictly greater only
            }
            // MFMA row fr holds the frame that is fr mod 16: frame t + ((fr - t) & 15)
            if (vv == 0) s_cand[(fr - t) & 15] = ((dp_u64)(((unsigned)it << 12) | (unsigned)bloc) << 32) | (dp_u64)__float_as_uint(best);
            __syncthreads();
            if (wave == 3 && lane < DP_FRAMES)  // the workgroup's 16 candidates: one 128-byte line (frame-major, [frame][g], put 16 workgroups'
                                                // 8-byte pieces into every line: 3 us per sweep)
                __hip_atomic_store(a.cg + ((long)(it & 1) * G + g) * DP_FRAMES + lane, s_cand[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        DP_T(7);
        // ---- the candidates of all workgroups: wave w takes the lines of workgroups 4 q .. 4 q + 3, q = w, w + 4, ... (a load = four whole
        // lines: lane = 16 (g & 3) + frame), keeps per lane the best of its workgroups for its frame, joins the wave's four lanes per frame and
        // leaves 16 (value, index) pairs in LDS; the bookkeeping (k_dec_update's) joins the four waves' pairs — first index on ties, like
        // torch.argmax
        {
            {
                const dp_u64 *cb = a.cg + (long)(it & 1) * G * DP_FRAMES;
                const int NQ = (G + 3) / 4;  // <= 32 groups of four workgroups: <= 8 per wave
                dp_u64 x[8];
                unsigned pending = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (wave + 4 * j < NQ && 4 * (wave + 4 * j) + kq < G) pending |= 1u << j;
                const unsigned mine = pending;
                int spins = 0;
                while (pending) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if ((pending >> j) & 1) x[j] = dp_load(cb + (long)(4 * (wave + 4 * j) + kq) * DP_FRAMES + i16);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (((pending >> j) & 1) && (unsigned)(x[j] >> 44) == (unsigned)it) pending &= ~(1u << j);
                    if (pending) {
                        if (++spins > DP_SPIN_LIMIT) { s_fail = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                DP_T(11);
                float bx = RNNT_NEG_INF;
                int bv = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if ((mine >> j) & 1) {
                        const float cx = __uint_as_float((unsigned)x[j]);
                        const int loc = (int)((x[j] >> 32) & 0xfff);
                        const int cv = ((4 * (wave + 4 * j) + kq) + G * (loc >> 4)) * 16 + (loc & 15);
                        if (cx > bx || (cx == bx && cv < bv)) { bx = cx; bv = cv; }
                    }
#pragma unroll
                for (int mm = 32; mm >= 16; mm >>= 1) {
                    const float ox = __shfl_xor(bx, mm, 64);
                    const int ov = __shfl_xor(bv, mm, 64);
                    if (ox > bx || (ox == bx && ov < bv)) { bx = ox; bv = ov; }
                }
                if (lane < DP_FRAMES) { s_wbx[wave * DP_FRAMES + lane] = bx; s_tok[wave * DP_FRAMES + lane] = bv; }
            }
            __syncthreads();
            if (s_fail) { code = 3; break; }
            DP_T(8);
            const int n = min(DP_FRAMES, T - t);
            // lane k (of every wave alike) joins the four waves' pairs of frame k; a ballot finds the first frame that is not blank (the loop
            // over k with its LDS reads one after the other took 4 kclk per iteration)
            int mytok;
            {
                const int k = lane & 15;
                float bx = s_wbx[k];
                int bv = s_tok[k];
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float ox = s_wbx[w * DP_FRAMES + k];
                    const int ov = s_tok[w * DP_FRAMES + k];
                    if (ox > bx || (ox == bx && ov < bv)) { bx = ox; bv = ov; }
                }
                mytok = bv;
            }
            const unsigned long long nb = __ballot(lane < n && mytok != a.blank);
            const int hit = nb ? (int)__builtin_ctzll(nb) : -1;
            const int tok = hit >= 0 ? __builtin_amdgcn_readlane(mytok, hit) : a.blank;
            if (hit < 0) { t += n; emitted = 0; newtok = 0; }
            else {
                if (hit > 0) emitted = 0;
                t += hit;
                ++ntok;
                if (g == 0 && tid == 0) {
                    if (a.max_length <= DP_TOKS) s_toks[ntok] = tok;
                    else a.tokens[ntok] = tok;
                }
                tk2 = tk1; tk1 = tk0; tk0 = tok;
                newtok = 1;
                if (++emitted >= a.max_per_frame) { ++t; emitted = 0; }
            }
            done = (t >= T || ntok + 1 >= a.max_length) ? 1 : 0;
            __syncthreads();  // s_tok is rewritten by the next iteration
        }
        DP_T(9);
    }
#ifdef DP_STAMPS
    if (g == 0 && tid == 0)
        for (int k = 0; k < 16; ++k) a.cg[2 * DP_FRAMES * 128 + k] = dpt[k];
#endif
    if (g == 0 && a.max_length <= DP_TOKS)
        for (int j = 1 + tid; j <= ntok; j += 256) a.tokens[j] = s_toks[j];
    if (g == 0 && tid == 0) {
        a.state[0] = t; a.state[1] = emitted; a.state[2] = ntok; a.state[3] = done; a.state[4] = newtok; a.state[5] = it - 1; a.state[6] = G;
        if (a.host_flag) __hip_atomic_store(a.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (code && tid == 0) __hip_atomic_store(a.state + 7, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a hand-off never arrived
#undef s_fail
}

// exp(2 x) of every audio frame.  tanh(e + p) = 1 - 2 / (1 + exp(2 e) exp(2 p)) is exact while neither factor leaves fp32's range:
// |x| <= DP_FACTOR_RANGE.  A frame (here) or a text vector (in the loop) beyond it, or non-finite, is NOT clamped into a wrong answer
// (enc = 40, text = -35: the clamped factors multiply to tanh(0), the truth is tanh(5)) — the decode reports it in state[7]
// (DP_CODE_*_RANGE) and leaves, and the caller runs the kernel-per-layer loop, which takes tanh of the SUM (round-5 advice).
__global__ __launch_bounds__(256) void k_dp_exp_frames(const float *__restrict__ frames, long st, int T, int H, float *__restrict__ out,
                                                       int32_t *__restrict__ state)
{
    const long n4 = (long)T * (H / 4);
    bool bad = false;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n4; j += (long)gridDim.x * 256) {
        const long t = j / (H / 4), h = 4 * (j - t * (H / 4));
        const f32x4 x = *(const f32x4 *)(frames + t * st + h);
        f32x4 y;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bad |= !(fabsf(x[e]) <= DP_FACTOR_RANGE);  // (true for NaN as well)
            y[e] = __builtin_amdgcn_exp2f(fminf(fmaxf(x[e], -DP_FACTOR_RANGE), DP_FACTOR_RANGE) * (2.0f * RNNT_LOG2E));
        }
        *(f32x4 *)(out + t * H + h) = y;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) __hip_atomic_store(state + 7, DP_CODE_FRAME_RANGE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// XE[s] = LN(embedding[s]) gamma + beta for every symbol (rnnt/predictor.py:214-215)
__global__ __launch_bounds__(256) void k_dp_ln_rows(const float *__restrict__ X, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                    float eps, int K, float *__restrict__ Y)
{
    __shared__ float red[4];
    const float *x = X + (long)blockIdx.x * K;
    float s = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) s += x[i];
    const float mean = dec_block_sum(s, red) / K;
    for (int i = threadIdx.x; i < K; i += 256) { const float d = x[i] - mean; s2 += d * d; }
    const float rstd = rsqrtf(dec_block_sum(s2, red) / K + eps);
    for (int i = threadIdx.x; i < K; i += 256) Y[(long)blockIdx.x * K + i] = (x[i] - mean) * rstd * gamma[i] + beta[i];
}

// joint.text_ln folded into the predictor's output (rnnt/joint.py:28-30 after rnnt/predictor.py:228-229), a wave per row h:
//   Wg[h] = W_t[h] * gamma;  d[h] = Wg[h] . b_l;  r[h] = sum Wg[h];  c[h] = W_t[h] . beta + b_t[h]
__global__ __launch_bounds__(256) void k_dp_fold_text(const float *__restrict__ Wt, const float *__restrict__ bt, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, const float *__restrict__ bl, int H, int O,
                                                      float *__restrict__ Wg, float *__restrict__ d, float *__restrict__ r, float *__restrict__ c)
{
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (h >= H) return;
    double sd = 0., sr = 0., sc = 0.;
    for (int o = lane; o < O; o += 64) {
        const float w = Wt[(long)h * O + o], wg = w * gamma[o];
        Wg[(long)h * O + o] = wg;
        sd += (double)wg * bl[o]; sr += wg; sc += (double)w * beta[o];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { sd += __shfl_xor(sd, m, 64); sr += __shfl_xor(sr, m, 64); sc += __shfl_xor(sc, m, 64); }
    if (lane == 0) { d[h] = (float)sd; r[h] = (float)sr; c[h] = (float)(sc + bt[h]); }
}

// float4 per lane a resident weight row takes (k_dec_persist<KF>), or 0: rows are loaded at every use
int dec_persist_kf(int E, int O, int H, int G, int has_text)
{
    const int rpA = (E + G - 1) / G, rpZ = (O + G - 1) / G, rpQ = (H + G - 1) / G;
    if (rpA > 8 || rpZ > 16 || (has_text && rpQ > 8) || E > 512) return 0;
    return E <= 256 ? 1 : 2;
}

int dec_persist_groups(int V)
{
    const int vb = (V + 15) / 16;
    return vb < 16 ? 16 : (vb > 128 ? 128 : vb);
}

// 0: the persistent loop takes these sizes; else a short reason
const char *dec_persist_refusal(int T, int S, int E, int O, int H, int V, int has_text)
{
    if (H % 64 || H > 1024) return "H must be a multiple of 64, <= 1024";
    if (E % 4 || E < 4 || E > 1024 || O % 4 || O < 4 || O > 1024) return "E, O must be multiples of 4 in [4, 1024]";
    if (S > 4096) return "more than 4096 symbols (the conv1 tables are S x E x 3)";
    if ((long)(V + 15) / 16 > 128L * 256) return "V too large";
    if (!has_text && O != H) return "without text_ln O must equal H";
    return nullptr;
}

// the MODEL's part of the persistent decode's memory — what depends on the parameters only: conv2's pack, the three conv1 tap tables, with
// joint.text_ln the folded matrix and vectors (+ the intermediates they are built from) — either inside the call's workspace, rebuilt on every
// call, or a buffer of the caller's built once per set of weights (rnnt_engine_greedy_decode_build_tables: 0.13 ms per utterance saved)
struct DecTablesLayout { size_t xe, wp1, wp2, tab, wg, m, vec, total; };
static DecTablesLayout dec_tables_layout(int S, int E, int O, int H, int has_text)
{
    DecTablesLayout L;
    size_t o = 0;
    auto take = [&](size_t floats) { const size_t at = o; o += (floats + 63) & ~(size_t)63; return at; };
    L.xe = take((size_t)S * E);
    L.wp1 = take((size_t)3 * E * E);
    L.wp2 = take((size_t)5 * E * E);
    L.tab = take((size_t)3 * S * E);
    L.wg = take(has_text ? (size_t)H * O : 0);
    L.m = take(has_text ? (size_t)H * E : 0);
    L.vec = take(has_text ? (size_t)3 * H : 0);
    L.total = o;
    return L;
}
size_t dec_tables_floats(int S, int E, int O, int H, int has_text) { return dec_tables_layout(S, E, O, H, has_text).total; }

void launch_dec_build_tables(const rnnt_conv_predictor_params &p, int S, int E, int O, float ln_in_eps, const float *text_W, const float *text_b, int H,
                             float *tb, hipStream_t st)
{
    const int has_text = text_W ? 1 : 0;
    const DecTablesLayout L = dec_tables_layout(S, E, O, H, has_text);
    hipLaunchKernelGGL(k_dp_ln_rows, dim3(S), dim3(256), 0, st, p.embedding, p.ln_in_w, p.ln_in_b, ln_in_eps, E, tb + L.xe);
    launch_pack_conv_w(p.conv1_w, tb + L.wp1, E, E, 3, st);  // [tap][out][in]
    launch_pack_conv_w(p.conv2_w, tb + L.wp2, E, E, 5, st);
    {  // the three tap tables by ONE GEMM: [S, E] x [3E, E]^T (the [tap][out][in] pack read as one 3E x E matrix) -> tab[s][tap][out]
        SgArgs g;
        g.A = tb + L.xe; g.lda = E; g.B = tb + L.wp1; g.ldb = E; g.C = tb + L.tab; g.ldc = 3 * E;
        g.bias = nullptr; g.Cpre = nullptr; g.mask = nullptr; g.mask_scale = 1.f;
        g.M = S; g.N = 3 * E; g.K = E; g.taps = 1; g.seg = S; g.act = 0; g.ksplit = 1;
        launch_sgemm_nt(g, st);
    }
    if (has_text) {
        float *vec = tb + L.vec;
        hipLaunchKernelGGL(k_dp_fold_text, dim3((H + 3) / 4), dim3(256), 0, st, text_W, text_b, p.ln_out_w, p.ln_out_b, p.linear_b, H, O,
                           tb + L.wg, vec, vec + H, vec + 2 * H);
        SgArgs g;  // M = Wg W_l: [H,O] x [O,E]
        g.A = tb + L.wg; g.lda = O; g.B = p.linear_w; g.ldb = E; g.C = tb + L.m; g.ldc = E;
        g.bias = nullptr; g.Cpre = nullptr; g.mask = nullptr; g.mask_scale = 1.f;
        g.M = H; g.N = E; g.K = O; g.taps = 1; g.seg = H; g.act = 0; g.ksplit = 1;
        launch_sgemm_nn(g, st);
    }
}

struct DecPersistLayout { size_t gran, eenc, tables, total; };
static DecPersistLayout dec_persist_layout(int T, int S, int E, int O, int H, int V, int has_text)
{
    DecPersistLayout L;
    size_t o = 0;
    auto take = [&](size_t floats) { const size_t at = o; o += (floats + 63) & ~(size_t)63; return at; };
    L.gran = take(2 * ((size_t)1024 + 1024 + 256 + 2 * DP_FRAMES * 128 + 16));  // g2 | z/q | stats | candidates | (DP_STAMPS: 16 counters), 8 bytes each
    L.eenc = take((size_t)T * H);
    L.tables = take(dec_tables_floats(S, E, O, H, has_text));  // (used when the caller brings no tables of its own)
    L.total = o;
    return L;
}
size_t dec_persist_workspace_floats(int T, int S, int E, int O, int H, int V, int has_text) { return dec_persist_layout(T, S, E, O, H, V, has_text).total; }

// dynamic LDS of k_dec_persist, bytes (100-138 KB: more than the 64 KB default limit, and more than some devices have at all)
size_t dec_persist_lds_bytes(int E)
{
    return ((size_t)8 * E + 6 * 1024 + (size_t)16 * 1024 + 4 * DP_FRAMES * 17 + 64 + 128 + 256 + 2 * DP_FRAMES + 8 * DP_FRAMES + 4 + DP_TOKS) * 4;
}

// returns hipSuccess, or the error of raising the kernel's dynamic-LDS limit (nothing has been launched then)
int launch_dec_persist(const DecLoopArgs &a, hipStream_t st)
{
    const int E = a.E, O = a.O, H = a.H, V = a.V, S = a.S, T = a.T, has_text = a.text_W ? 1 : 0;
    const DecPersistLayout L = dec_persist_layout(T, S, E, O, H, V, has_text);
    const DecTablesLayout TL = dec_tables_layout(S, E, O, H, has_text);
    float *ws = (float *)a.workspace;
    const int G = dec_persist_groups(V);
    // every polled word starts at tag 0 (iterations count from 1): k_dec_init — a kernel, not a memset node (captured calls are kernel chains)
    const int kf = dec_persist_kf(E, O, H, G, has_text);
    const size_t lds = dec_persist_lds_bytes(E);
    const void *kern = kf == 1 ? (const void *)k_dec_persist<1> : kf == 2 ? (const void *)k_dec_persist<2> : (const void *)k_dec_persist<0>;
    if (const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e != hipSuccess) return (int)e;
    // (the limit is raised BEFORE anything is enqueued: a refusal leaves the stream untouched)
    const int ngran = (int)(L.eenc - L.gran);
    hipLaunchKernelGGL(k_dec_init, dim3((ngran + 255) / 256), dim3(256), 0, st, a.state, a.tokens, ws + L.gran, ngran, a.blank);
    hipLaunchKernelGGL(k_dp_exp_frames, dim3(512), dim3(256), 0, st, a.frames, a.frame_stride, T, H, ws + L.eenc, a.state);
    const float *tb = (const float *)a.tables;
    if (!tb) {
        launch_dec_build_tables(a.p, S, E, O, a.ln_in_eps, a.text_W, a.text_b, H, ws + L.tables, st);
        tb = ws + L.tables;
    }
    DecPersistArgs k;
    dp_u64 *gr = (dp_u64 *)(ws + L.gran);
    k.g2g = gr; k.zg = gr + 1024; k.sg = gr + 2048; k.cg = gr + 2048 + 256;
    k.Eenc = ws + L.eenc; k.T = T;
    k.A0 = tb + TL.tab; k.A1 = k.A0 + E; k.A2 = k.A1 + E;  // rows 3E apart: tab[s][tap][E]
    k.conv1_b = a.p.conv1_b; k.wp2 = tb + TL.wp2; k.conv2_b = a.p.conv2_b;
    k.Wl = a.p.linear_w; k.bl = a.p.linear_b;
    k.M = tb + TL.m; k.dvec = tb + TL.vec; k.rvec = tb + TL.vec + H; k.cvec = tb + TL.vec + 2 * (size_t)H;
    k.gamma = a.p.ln_out_w; k.beta = a.p.ln_out_b; k.eps = a.ln_eps;
    k.W = a.W; k.bias = a.bias;
    k.S = S; k.E = E; k.O = O; k.H = H; k.V = V; k.blank = a.blank; k.max_length = a.max_length; k.max_per_frame = a.max_per_frame;
    k.has_text = has_text; k.max_iters = a.max_length + T + 2;
    k.state = a.state; k.tokens = a.tokens; k.host_flag = a.host_flag;
    if (kf == 1) hipLaunchKernelGGL(k_dec_persist<1>, dim3(G), dim3(256), lds, st, k);
    else if (kf == 2) hipLaunchKernelGGL(k_dec_persist<2>, dim3(G), dim3(256), lds, st, k);
    else hipLaunchKernelGGL(k_dec_persist<0>, dim3(G), dim3(256), lds, st, k);
    return (int)hipSuccess;
}
