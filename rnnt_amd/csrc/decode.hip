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
