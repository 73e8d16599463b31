// Micro-benchmark 2: the forward kernel's chunk (32 MFMAs + 10 x 16-byte loads + 4 tanh) with
// pieces switched off, to find which non-MFMA component costs matrix-pipe time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_tanh(float x)
{
    float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// MODE bit0: tanh VALU, bit1: B loads from memory, bit2: A loads from memory, bit3: sched_barrier pinning
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void k(const f32x4 *__restrict__ wbuf, const f32x4 *__restrict__ abuf,
                                               float *out, int chunks, int wmask)
{
    f32x16 acc[8];
    for (int q = 0; q < 8; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const int lane = threadIdx.x & 63;
    f32x4 w[8], e, p;
    const f32x4 *wp = wbuf + lane + (blockIdx.x & 7) * 512;
    const f32x4 *ap = abuf + threadIdx.x;
    for (int q = 0; q < 8; ++q) w[q] = wp[q * 64];
    e = ap[0]; p = ap[512];
    int off = 0;
    for (int c = 0; c < chunks; ++c) {
        float a[4];
        for (int s = 0; s < 4; ++s) a[s] = (MODE & 1) ? fast_tanh(e[s] + p[s]) : e[s] + p[s];
        off = (off + 8192) & wmask;  // float4 units: 8192 = 128 KB per chunk step
        if (MODE & 4) { e = ap[(off & 0xffff)]; p = ap[(off & 0xffff) + 512]; }
        if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w[q][s], acc[q], 0, 0, 0);
            if (MODE & 2) w[q] = wp[off + q * 64];
            if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int q = 0; q < 8; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int THREADS>
void run(const char *name, int blocks_per_cu, const f32x4 *wbuf, const f32x4 *abuf, float *out, int wmask)
{
    int chunks = 2000;
    int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, wbuf, abuf, out, chunks, wmask);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, wbuf, abuf, out, chunks, wmask);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double waves = (double)grid * THREADS / 64;
    double flops = waves * chunks * 32 * 4096.0;
    printf("%-44s thr=%d blk/CU=%d  %.1f TFLOP/s\n", name, THREADS, blocks_per_cu, flops / ms / 1e9);
}

int main()
{
    f32x4 *wbuf, *abuf; float *out;
    size_t wn = (size_t)1 << 22;  // 4M float4 = 64 MB
    (void)hipMalloc(&wbuf, wn * 16 + (1 << 20)); (void)hipMalloc(&abuf, (1 << 20) * 16);
    (void)hipMalloc(&out, 256 * 8 * 1024 * 4);
    (void)hipMemset(wbuf, 0, wn * 16 + (1 << 20)); (void)hipMemset(abuf, 0, (1 << 20) * 16);
    const int m2MB = (1 << 17) - 1;   // 128K float4 = 2 MB window
    const int m64MB = (1 << 22) - 1;
    run<0, 512>("mfma only", 1, wbuf, abuf, out, m2MB);
    run<1, 512>("mfma + tanh", 1, wbuf, abuf, out, m2MB);
    run<2, 512>("mfma + B loads (2MB window)", 1, wbuf, abuf, out, m2MB);
    run<6, 512>("mfma + A,B loads (2MB)", 1, wbuf, abuf, out, m2MB);
    run<7, 512>("mfma + tanh + A,B loads (2MB)", 1, wbuf, abuf, out, m2MB);
    run<15, 512>("same, sched_barrier pinned", 1, wbuf, abuf, out, m2MB);
    run<15, 256>("same pinned, 2 WGs x 4 waves", 2, wbuf, abuf, out, m2MB);
    run<15, 512>("pinned, 64MB window", 1, wbuf, abuf, out, m64MB);
    run<10, 512>("mfma + B loads pinned (2MB)", 1, wbuf, abuf, out, m2MB);
    run<10, 256>("mfma + B loads pinned, 1 wave/SIMD", 1, wbuf, abuf, out, m2MB);
    return 0;
}
