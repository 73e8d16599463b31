// Micro-benchmark: does the fp32 MFMA rate hold over the duration of the engine's GEMM kernels
// (~50 ms), or does the clock drop under sustained matrix load?  Back-to-back launches of ~10 ms of
// pure v_mfma_f32_32x32x2_f32 (8 accumulators, 4 dependent in a row, 2 waves per SIMD), each
// timed with HIP events; the figure the kernels are priced against is 157.3 TFLOP/s = 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool RANDOM>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters, float a0, float b0, const float *rnd)
{
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    // RANDOM: operands with random mantissas (as in a real GEMM) instead of two constants — the
    // switching activity, hence the power, of the matrix pipe depends on the data
    float av[4], bv[8];
    for (int i = 0; i < 4; ++i) av[i] = RANDOM ? rnd[(threadIdx.x * 4 + i) & 4095] : a0 + threadIdx.x;
    for (int i = 0; i < 8; ++i) bv[i] = RANDOM ? rnd[(threadIdx.x * 8 + i + 1000) & 4095] : b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int d = 0; d < 4; ++d)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[d], bv[q], acc[q], 0, 0, 0);
    }
    float s = 0;
    for (int q = 0; q < 8; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool RANDOM>
void run(float *out, const float *rnd)
{
    const int iters = 6000, n = 30;  // 6000 * 32 MFMAs * 64 cycles * 2 waves/SIMD = 24.6 M cycles ~ 10 ms
    hipEvent_t ev[n + 1];
    for (int i = 0; i <= n; ++i) hipEventCreate(&ev[i]);
    hipLaunchKernelGGL(k<RANDOM>, dim3(256), dim3(512), 0, 0, out, 100, 1.f, 2.f, rnd);
    hipDeviceSynchronize();
    hipEventRecord(ev[0]);
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(k<RANDOM>, dim3(256), dim3(512), 0, 0, out, iters, 1.f, 2.f, rnd);
        hipEventRecord(ev[i + 1]);
    }
    hipEventSynchronize(ev[n]);
    const double flops = 256.0 * 8 * iters * 32 * 4096.0;
    printf("%s operands\n", RANDOM ? "random" : "constant");
    for (int i = 0; i < n; i += 3) {
        float ms;
        hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
        printf("  launch %2d: %6.2f ms  %.1f TFLOP/s  (%.0f MHz equivalent)\n", i, ms, flops / ms / 1e9,
               flops / ms / 1e9 / 157.3 * 2400);
    }
}

int main()
{
    float *out, *rnd, h[4096];
    unsigned x = 12345;
    for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) / 16777216.f - 0.5f; }
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&rnd, sizeof h);
    hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice);
    run<false>(out, rnd);
    run<true>(out, rnd);
    run<false>(out, rnd);
    return 0;
}
