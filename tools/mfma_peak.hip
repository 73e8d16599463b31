// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate under the issue patterns the engine
// kernels use (calibrates the ceiling the GEMM kernels are measured against).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int DEP>  // NACC accumulators; DEP consecutive dependent MFMAs per acc
__global__ __launch_bounds__(512, 2) void k(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q)
#pragma unroll
            for (int d = 0; d < DEP; ++d)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    }
    float s = 0;
    for (int q = 0; q < NACC; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int DEP>
void run(const char *name, int threads, int blocks_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
    int iters = 20000 / (NACC * DEP) * 4;
    int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, DEP>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, DEP>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)grid * threads / 64;
    double flops = waves * iters * NACC * DEP * 4096.0;
    printf("%-40s threads=%d blocks/CU=%d  %.1f TFLOP/s (%.2f ms)\n", name, threads, blocks_per_cu,
           flops / ms / 1e9, ms);
    hipFree(out);
}

int main()
{
    run<4, 1>("4 acc, independent", 256, 1);
    run<8, 1>("8 acc, independent", 256, 1);
    run<8, 4>("8 acc, 4 dependent in a row", 256, 1);
    run<8, 1>("8 acc, independent", 512, 1);
    run<8, 4>("8 acc, 4 dependent in a row", 512, 1);
    run<8, 4>("8 acc, 4 dep, 2 WGs of 4 waves", 256, 2);
    run<1, 8>("1 acc, fully dependent", 256, 1);
    run<1, 8>("1 acc, fully dependent", 512, 1);
    return 0;
}
