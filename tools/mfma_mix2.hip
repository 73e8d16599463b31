// Micro-benchmark 3: the forward kernel's software-pipelined chunk, with single ingredients
// removed, 8 waves per CU (2 per SIMD).  Finds which ingredient costs matrix-pipe time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float fast_tanh(float x)
{
    float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// FLAGS: 1 A loads, 2 tanh, 4 B refill, 8 B window tiny (L1), 16 A gather pattern (32 rows)
template <int FLAGS>
__global__ __launch_bounds__(512, 2) void k(const f32x4 *__restrict__ wbuf, const float *__restrict__ abuf,
                                            float *out, int HK, int passes)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wn = wave & 1, wm = wave >> 1;
    const int i = lane & 31, half = lane >> 5;
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const long wstride = 8 * 256;
    const f32x4 *wp = wbuf + (long)(wn * 2) * 256 + lane;
    const int row = (FLAGS & 16) ? (blockIdx.x * 128 + wm * 32 + i) % 201 : 0;
    const float *erow = abuf + (long)(blockIdx.x % 1000) * 512 + 4 * half;
    const float *prow = abuf + 1000 * 512 + (long)row * 512 + 4 * half;
    for (int pass = 0; pass < passes; ++pass) {
        f32x4 w[8];
        float a_cur[4], a_nxt[4];
        f32x4 e = *(const f32x4 *)erow, p = *(const f32x4 *)prow;
        for (int q = 0; q < 8; ++q) w[q] = wp[q * 64];
        for (int s = 0; s < 4; ++s) a_cur[s] = e[s] + p[s];
        int cn = 1;
        for (int c8 = 0; c8 < HK; ++c8) {
            const f32x4 *wn_ = wp + (long)((FLAGS & 8) ? 0 : cn) * wstride;
            cn = cn + 1 >= HK ? 0 : cn + 1;
            const int a2 = 8 * cn;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q < 4) a_nxt[q] = (FLAGS & 2) ? fast_tanh(e[q] + p[q]) : e[q] + p[q];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[s], w[q][s], acc[q], 0, 0, 0);
                if (FLAGS & 4) w[q] = wn_[q * 64];
                if ((FLAGS & 1) && q == 7) {
                    e = *(const f32x4 *)(erow + a2);
                    p = *(const f32x4 *)(prow + a2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int s = 0; s < 4; ++s) a_cur[s] = a_nxt[s];
        }
    }
    float s = 0;
    for (int q = 0; q < 8; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int FLAGS>
void run(const char *name, const f32x4 *wbuf, const float *abuf, float *out)
{
    const int HK = 64, passes = 30, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FLAGS>), dim3(grid), dim3(512), 0, 0, wbuf, abuf, out, HK, passes);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FLAGS>), dim3(grid), dim3(512), 0, 0, wbuf, abuf, out, HK, passes);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 8 * passes * HK * 32 * 4096.0;
    printf("%-52s %.1f TFLOP/s\n", name, flops / ms / 1e9);
}
int main()
{
    f32x4 *wbuf; float *abuf, *out;
    (void)hipMalloc(&wbuf, 64 * 8 * 256 * 16 + 65536); (void)hipMalloc(&abuf, 1300 * 512 * 4);
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMemset(wbuf, 0, 64 * 8 * 256 * 16 + 65536); (void)hipMemset(abuf, 0, 1300 * 512 * 4);
    run<0>("mfma only (pipelined skeleton)", wbuf, abuf, out);
    run<2>("+ tanh", wbuf, abuf, out);
    run<4>("+ B refill (2MB L2 window)", wbuf, abuf, out);
    run<12>("+ B refill (one 8KB window, L1 hits)", wbuf, abuf, out);
    run<1>("+ A loads (coalesced rows)", wbuf, abuf, out);
    run<17>("+ A loads (32-row gather)", wbuf, abuf, out);
    run<6>("+ tanh + B refill", wbuf, abuf, out);
    run<7>("+ tanh + B refill + A loads", wbuf, abuf, out);
    run<23>("+ tanh + B refill + A gather  (= forward kernel)", wbuf, abuf, out);
    run<31>("same, B from L1 window", wbuf, abuf, out);
    return 0;
}
