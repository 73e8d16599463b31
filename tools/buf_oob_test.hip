// Probe of raw-buffer range checking on gfx950: which stores/loads are dropped when the vector
// offset is out of range and a scalar offset is added (build: hipcc --offload-arch=gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *buf, int nrec_bytes, unsigned *out)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, nrec_bytes, 0x00020000);
    const int l = threadIdx.x;
    u32x4 v = {1000u + l, 1000u + l, 1000u + l, 1000u + l};
    // case A: lanes 0-3 valid voffset (16*l), lanes 4-7 voffset 0xfffffff0, soffset 64
    unsigned voff = l < 4 ? 16u * l : 0xfffffff0u;
    if (l < 8) __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 64, 0);
    // case B: lane 8: voffset in range, soffset pushes it past num_records
    if (l == 8) __builtin_amdgcn_raw_buffer_store_b128(v, r, 0, nrec_bytes + 64, 0);
    // case C: lane 9: voffset = nrec_bytes - 16 (last valid), soffset 0
    if (l == 9) __builtin_amdgcn_raw_buffer_store_b128(v, r, nrec_bytes - 16, 0, 0);
    // loads: lane 10 OOB voffset + soffset 64
    u32x4 x = {7u, 7u, 7u, 7u};
    if (l == 10) x = __builtin_amdgcn_raw_buffer_load_b128(r, 0xfffffff0u, 64, 0);
    if (l == 10) out[0] = x[0];
}
int main()
{
    float *buf; unsigned *out;
    const int n = 4096;  // floats allocated
    hipMalloc(&buf, n * 4); hipMalloc(&out, 64);
    hipMemset(buf, 0, n * 4); hipMemset(out, 0xff, 64);
    const int nrec = 1024;  // bytes covered by the descriptor
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, buf, nrec, out);
    unsigned h[4096]; unsigned ho[16];
    hipMemcpy(h, buf, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ho, out, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) if (h[i]) printf("word %d (byte %d) = %u\n", i, i * 4, h[i]);
    printf("oob load returned %u\n", ho[0]);
    return 0;
}
