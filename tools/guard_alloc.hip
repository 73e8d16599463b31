// Diagnostic helper (tools/guard_sweep.py): device memory that ENDS at an unmapped page.  A virtual range
// of the rounded size plus one granule is reserved, only the first part is mapped: an access past the
// end of the buffer faults instead of reading a neighbour.   hipcc -shared -fPIC -o libguard.so guard_alloc.hip
#include <hip/hip_runtime.h>
#include <cstdio>

extern "C" int guard_alloc(size_t bytes, int device, void **base_out, size_t *mapped_out, void **end_aligned_out)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) return 1;
    const size_t mapped = (bytes + gran - 1) / gran * gran;
    void *base = nullptr;
    if (hipMemAddressReserve(&base, mapped + gran, gran, nullptr, 0) != hipSuccess) return 2;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, mapped, &prop, 0) != hipSuccess) return 3;
    if (hipMemMap(base, mapped, 0, h, 0) != hipSuccess) return 4;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(base, mapped, &acc, 1) != hipSuccess) return 5;
    *base_out = base;
    *mapped_out = mapped;
    const size_t b16 = (bytes + 15) & ~(size_t)15;
    *end_aligned_out = (char *)base + mapped - b16;  // the buffer's last 16-byte block is the last mapped one
    return 0;
}
