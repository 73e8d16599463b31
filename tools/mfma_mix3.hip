// Micro-benchmark 4: the dW kernel's k-step loop (16 MFMAs on 256 AGPR accumulators, 16-byte
// LDS fragment reads, LDS-DMA ring, one barrier per chunk) with ingredients switched off.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_void_ptr;
#define KC 16
#define ROWF 256
#define CHUNKF (2 * KC * ROWF)
// FLAGS: 1 ds_read fragments each k-step, 2 DMA ring, 4 barrier per chunk, 8 dbacc adds
template <int FLAGS>
__global__ __launch_bounds__(256, 1) void k(const float *__restrict__ g, const float *__restrict__ h,
                                            float *out, int nk)
{
    __shared__ __attribute__((aligned(16))) float smem[4 * CHUNKF];
    const int tid = threadIdx.x, lane = tid & 63, wave = (FLAGS & 32) ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6), wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, half = lane >> 5;
    f32x16 acc[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float dbacc[4] = {0, 0, 0, 0};
    for (int x = tid; x < 4 * CHUNKF; x += 256) smem[x] = 0.f;
    __syncthreads();
    const float *gsrc = g + (long)blockIdx.x * nk * KC * 1024 + 4 * lane;
    const float *hsrc = h + (long)blockIdx.x * nk * KC * 512 + 4 * lane;
    const int aoff = half * ROWF + wm * 128 + 4 * i, boff = (KC + half) * ROWF + wn * 128 + 4 * i;
    auto dma2 = [&](int kk_, int slot, int j) {
        const int kk = (FLAGS & 16) ? (kk_ & 3) : kk_;  // 16: re-read a small (L2-resident) window
        float *dst = smem + slot * CHUNKF;
        const int r = wave + 4 * j;
        __builtin_amdgcn_global_load_lds(gsrc + ((long)kk * KC + r) * 1024, (lds_void_ptr)(dst + r * ROWF), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(hsrc + ((long)kk * KC + r) * 512, (lds_void_ptr)(dst + (KC + r) * ROWF), 16, 0, 0);
    };
    if (FLAGS & 2) { for (int j = 0; j < 4; ++j) dma2(0, 0, j); for (int j = 0; j < 4; ++j) dma2(1, 1, j); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 a4 = *(const f32x4 *)(smem + aoff), b4 = *(const f32x4 *)(smem + boff);
    for (int kk = 0; kk < nk; ++kk) {
        const float *sl = smem + (kk & 3) * CHUNKF, *sn = smem + ((kk + 1) & 3) * CHUNKF;
        const bool more2 = kk + 2 < nk;
#pragma unroll
        for (int ks = 0; ks < KC / 2; ++ks) {
            f32x4 an = a4, bn = b4;
            if (FLAGS & 1) {
                if (ks + 1 < KC / 2) { an = *(const f32x4 *)(sl + aoff + (ks + 1) * 2 * ROWF); bn = *(const f32x4 *)(sl + boff + (ks + 1) * 2 * ROWF); }
                else { an = *(const f32x4 *)(sn + aoff); bn = *(const f32x4 *)(sn + boff); }
            }
            __builtin_amdgcn_sched_barrier(0);
            for (int qm = 0; qm < 2; ++qm) {
                if (FLAGS & 8) dbacc[qm] += a4[qm];
                for (int qn = 0; qn < 4; ++qn) acc[qm][qn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qm], b4[qn], acc[qm][qn], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((FLAGS & 2) && ks < 4 && more2) dma2(kk + 2, (kk + 2) & 3, ks);
            __builtin_amdgcn_sched_barrier(0);
            for (int qm = 2; qm < 4; ++qm) {
                if (FLAGS & 8) dbacc[qm] += a4[qm];
                for (int qn = 0; qn < 4; ++qn) acc[qm][qn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[qm], b4[qn], acc[qm][qn], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 3 && (FLAGS & 4)) {
                if (FLAGS & 2) { if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            a4 = an; b4 = bn;
        }
    }
    float s = dbacc[0] + dbacc[1] + dbacc[2] + dbacc[3];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[blockIdx.x * 256 + tid] = s;
}
template <int FLAGS> void run(const char *name, const float *g, const float *h, float *out)
{
    const int nk = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<FLAGS>), dim3(256), dim3(256), 0, 0, g, h, out, nk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FLAGS>), dim3(256), dim3(256), 0, 0, g, h, out, nk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-50s %.1f TFLOP/s\n", name, 256.0 * 4 * nk * 8 * 16 * 4096.0 / ms / 1e9);
}
int main()
{
    float *g, *h, *out;
    const size_t ng = (size_t)256 * 400 * KC * 1024, nh = (size_t)256 * 400 * KC * 512;
    (void)hipMalloc(&g, ng * 4 + 65536); (void)hipMalloc(&h, nh * 4 + 65536); (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMemset(g, 0, ng * 4); (void)hipMemset(h, 0, nh * 4);
    run<0>("mfma only (256 AGPR acc)", g, h, out);
    run<8>("+ dbacc adds", g, h, out);
    run<1>("+ LDS fragment reads", g, h, out);
    run<4>("+ barrier per chunk", g, h, out);
    run<5>("+ LDS reads + barrier", g, h, out);
    run<7>("+ LDS reads + barrier + DMA ring (HBM stream)", g, h, out);
    run<15>("all (= dW kernel loop)", g, h, out);
    run<31>("all, DMA source = small L2-resident window", g, h, out);
    run<23>("LDS reads + barrier + DMA from L2 window", g, h, out);
    run<47>("all, wave id hoisted to an SGPR (readfirstlane once)", g, h, out);
    return 0;
}
