// Device code of tools/aql_probe.cpp (hipcc --genco): an empty kernel, and one that adds 1 to a word (to see that the chain ran in order).
#include <hip/hip_runtime.h>
extern "C" __global__ void aql_empty(int* p) { if (p && threadIdx.x == 12345) *p = 0; }
extern "C" __global__ void aql_step(unsigned* p, unsigned expect) { if (threadIdx.x == 0 && blockIdx.x == 0) { if (*p == expect) *p = expect + 1; } }
// a kernel that is busy for `ticks` of the 100 MHz real-time counter (8 us = 800) on every workgroup: the "work" of a launch in a chain
extern "C" __global__ void aql_busy(unsigned ticks, int* p)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (p && threadIdx.x == 12345) *p = 0;
}
