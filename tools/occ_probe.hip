// Probe: does the dispatcher keep TWO 4-wave workgroups of 256 VGPRs each on a CU?  Each
// workgroup spins for a fixed number of clocks; 512 workgroups take one spin time if two are
// resident per CU, two if they run one after the other.  Variants: scratch enabled or not, LDS size.
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool SCRATCH, int LDS_BYTES>
__global__ __launch_bounds__(256, 2) void k(float *out, long spin, int idx)
{
    __shared__ char lds[LDS_BYTES];
    // force 256 VGPRs: 240 live floats
    float r[240];
#pragma unroll
    for (int i = 0; i < 240; ++i) r[i] = out[(threadIdx.x + i) & 1023];
    float stack[SCRATCH ? 32 : 1];
    if (SCRATCH) {
        for (int i = 0; i < 32; ++i) stack[i] = out[(threadIdx.x * 7 + i) & 1023];
    }
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const long t0 = clock64();
    while (clock64() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < 240; ++i) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = lds[(threadIdx.x + 1) & 255];
#pragma unroll
    for (int i = 0; i < 240; ++i) s += r[i];
    if (SCRATCH) s += stack[idx & 31];  // dynamic index keeps the array in scratch
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool SCRATCH, int LDS_BYTES>
void run(float *out, const char *name)
{
    const long spin = 2000000;  // s_memtime ticks (100 MHz) -> 20 ms
    {   // dispatch rate: 100 000 workgroups that exit at once
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<SCRATCH, LDS_BYTES>), dim3(100000), dim3(256), 0, 0, out, 0L, 3);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SCRATCH, LDS_BYTES>), dim3(100000), dim3(256), 0, 0, out, 0L, 3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s 100000 empty workgroups: %7.3f ms = %.2f us per workgroup per XCD\n", name, ms, ms * 1e3 / (100000 / 8));
    }
    for (int grid : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<SCRATCH, LDS_BYTES>), dim3(grid), dim3(256), 0, 0, out, 1000L, 3);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SCRATCH, LDS_BYTES>), dim3(grid), dim3(256), 0, 0, out, spin, 3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s grid %4d: %7.2f ms\n", name, grid, ms);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 100000L * 256 * sizeof(float));
    hipMemset(out, 0, 100000L * 256 * sizeof(float));
    run<false, 1024>(out, "no scratch, 1 KB LDS");
    run<true, 1024>(out, "scratch, 1 KB LDS");
    run<false, 33792>(out, "no scratch, 33 KB LDS");
    run<true, 33792>(out, "scratch, 33 KB LDS");
    return 0;
}
