// Micro-benchmark (round 4): what does a memory instruction issued by ONE wave of a SIMD cost the MFMA stream of the
// OTHER wave of that SIMD?  512-thread workgroups, one per CU: waves 0-3 (one per SIMD) run a pure
// v_mfma_f32_32x32x16_bf16 stream of fixed length and time it with s_memtime; waves 4-7 (their SIMD partners) issue, until
// the MFMA waves are done, one memory instruction every `gap` iterations of an s_sleep loop:
//   mode 0: nothing            mode 1: LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB, L2-resident source)
//   mode 2: global_load_dwordx4 to registers (same bytes)        mode 3: ds_read_b128
//   mode 4: global_store_dwordx4 (1 KiB)
//   mode 5: LDS-DMA piece with NO address VGPR: the descriptor's ADD_TID_ENABLE adds lane x stride (16 bytes) itself
// Reported per mode and rate: cycles per MFMA of the MFMA waves, and memory instructions issued per 96 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void *lds_vptr;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *out, const char *src, char *dst, int n_mfma, int sleep, unsigned long long *res, int prio, int agpr)
{
    extern __shared__ __attribute__((aligned(1024))) char s[];
    volatile int *done = (volatile int *)(s + 65536);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *done = 0;
    __syncthreads();
    const bool mfma_wave = wave < 4;
    if (mfma_wave) {
        f32x16 acc[4];
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        u32x4 a = {0x3f803f80u + lane, 0x3f003f80u, 0x3e803f00u, 0x3f803e80u}, b = {0x3f003f00u, 0x3f803f80u + lane, 0x3f803f00u, 0x3e803f80u};
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < n_mfma / 4; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (agpr) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[q]) : "v"(a), "v"(b));  // accumulators in AGPRs
                else acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[q], 0, 0, 0);
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        float sum = 0.f;
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 16; ++r) sum += acc[q][r];
        out[(size_t)blockIdx.x * 512 + tid] = sum;
        if (lane == 0) {
            res[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
            atomicAdd((int *)done, 1);
        }
    } else {
        // the loader: one instruction per iteration until every MFMA wave of the workgroup is done
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 3 << 20, 0x00020000);
        // word 1: stride 16 in [29:16]; word 3: ADD_TID_ENABLE (bit 23), DATA_FORMAT bits left 0 (they extend the stride there)
        const __amdgpu_buffer_rsrc_t rs_tid = __builtin_amdgcn_make_buffer_rsrc((void *)src, 16, 0x7fffffff, 0x00800000);
        const int voff = lane * 16;
        // at equal priority the SIMD issues oldest-first and this wave starves behind its partner's pending MFMA (first
        // version of this tool: 20 loop iterations in 3 ms): priority outranks age
        if (prio) __builtin_amdgcn_s_setprio(3);
        unsigned long long n = 0, tl0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tl0)::"memory");
        u32x4 sink = {0, 0, 0, 0};
        int soff = ((blockIdx.x * 4 + (wave & 3)) * 7919) & 0x3fff;
        // modes 2 and 3 return data into registers asynchronously: the eight destinations of a batch are named by the
        // wait that ends the batch, so the compiler keeps them allocated while the loads are in flight
        u32x4 v[8];
        for (int j = 0; j < 8; ++j) v[j] = sink;
        while (*done < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                soff = (soff + 1) & 0xbff;  // walk 3 MiB in 1 KiB pieces
                if (MODE == 1) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(s + (j + 8 * (int)(n & 3) + 32 * (wave & 1)) * 1024), 16, voff, soff * 1024, 0, 0);
                } else if (MODE == 5) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_tid, (lds_vptr)(s + (j + 8 * (int)(n & 3) + 32 * (wave & 1)) * 1024), 16, 0, soff * 1024, 0, 0);
                } else if (MODE == 2) {
                    const char *p = src + (size_t)soff * 1024 + voff;
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(p) : "memory");
                } else if (MODE == 3) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(v[j]) : "v"(voff + (j + 8 * (int)(n & 3)) * 1024) : "memory");
                } else if (MODE == 4) {
                    *(u32x4 *)(dst + ((size_t)(blockIdx.x * 4 + (wave & 3)) * 64 + j + 8 * (n & 7)) * 1024 + voff) = sink;
                }
                for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
            n += 8;
        }
        sink = v[0] ^ v[1] ^ v[2] ^ v[3] ^ v[4] ^ v[5] ^ v[6] ^ v[7];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (sink[0] == 0x12345678u) out[tid] = 1.f;
        if (lane == 0) res[(blockIdx.x * 4 + (wave & 3)) * 2 + 1] = n;
        if (lane == 0 && blockIdx.x == 0) { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory"); res[2048 + (wave & 3)] = t - tl0; }
    }
}

__global__ void k_check(const char *src, unsigned *out)
{
    extern __shared__ __attribute__((aligned(1024))) char s[];
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 3 << 20, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_tid = __builtin_amdgcn_make_buffer_rsrc((void *)src, 16, 0x7fffffff, 0x00800000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(s), 16, lane * 16, 5 * 1024, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_tid, (lds_vptr)(s + 1024), 16, 0, 5 * 1024, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned bad = 0;
    for (int k = 0; k < 4; ++k) bad += ((unsigned *)s)[lane * 4 + k] != ((unsigned *)(s + 1024))[lane * 4 + k];
    out[lane] = bad;
    out[64 + lane] = ((unsigned *)(s + 1024))[lane * 4];
}
void check_tid(const char *src)
{
    unsigned *o, h[128];
    hipMalloc(&o, 512);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 2048, 0, src, o);
    hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
    unsigned bad = 0;
    for (int i = 0; i < 64; ++i) bad += h[i];
    printf("TID-addressed DMA vs VGPR-addressed DMA: %u mismatching dwords (lane 0 dword %08x, lane 63 dword %08x)\n", bad, h[64], h[127]);
}

template <int MODE>
void run(const char *name, float *out, const char *src, char *dst, unsigned long long *res, int n_mfma)
{
    (void)hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    std::vector<unsigned long long> h(256 * 4 * 2 + 8);
    for (int agpr : {0, 1})
    for (int prio : {0, 1})
    for (int sleep : {16, 4, 0}) {
        if (MODE == 0 && sleep != 0) continue;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536 + 64, 0, out, src, dst, n_mfma, sleep, res, prio, agpr);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 65536 + 64, 0, out, src, dst, n_mfma, sleep, res, prio, agpr);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), res, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, rate;
        for (int i = 0; i < 256 * 4; ++i) { cyc.push_back((double)h[2 * i] / n_mfma); rate.push_back((double)h[2 * i + 1] * 96.0 / n_mfma); }
        if (getenv("RAW")) printf("   loader elapsed %llu %llu\n", h[2048], h[2049]);
        if (getenv("RAW")) printf("   raw: cycles %llu %llu %llu %llu  n %llu %llu %llu %llu\n", h[0], h[2], h[4], h[6], h[1], h[3], h[5], h[7]);
        std::sort(cyc.begin(), cyc.end()); std::sort(rate.begin(), rate.end());
        printf("%-22s acc in %s prio %d sleep %3d: %6.2f cycles/MFMA (median over waves; p90 %6.2f)   %6.2f memory instructions per 96 MFMAs\n", name, agpr ? "AGPRs" : "VGPRs", prio, sleep,
               cyc[cyc.size() / 2], cyc[cyc.size() * 9 / 10], rate[rate.size() / 2]);
    }
}

int main(int argc, char **argv)
{
    const int n_mfma = argc > 1 ? atoi(argv[1]) : 200000;
    float *out; char *src, *dst; unsigned long long *res;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&src, 4 << 20);
    hipMalloc(&dst, (size_t)256 * 4 * 64 * 1024);
    hipMalloc(&res, (256 * 4 * 2 + 8) * 8);
    {
        std::vector<unsigned> hs((4 << 20) / 4);
        for (size_t i = 0; i < hs.size(); ++i) hs[i] = 0x3c003c00u + (unsigned)((i * 2654435761u) >> 20 & 0xff) * 0x10001u;  // small positive bf16 pairs
        hipMemcpy(src, hs.data(), 4 << 20, hipMemcpyHostToDevice);
    }
    run<0>("no memory instructions", out, src, dst, res, n_mfma);
    run<1>("LDS-DMA pieces", out, src, dst, res, n_mfma);
    run<2>("global loads to VGPRs", out, src, dst, res, n_mfma);
    run<3>("ds_read_b128", out, src, dst, res, n_mfma);
    run<4>("global stores", out, src, dst, res, n_mfma);
    run<5>("LDS-DMA, TID-addressed", out, src, dst, res, n_mfma);
    // what the TID-addressed DMA put into LDS must be what the VGPR-addressed one puts there: checked by check_tid below
    check_tid(src);
    return 0;
}
