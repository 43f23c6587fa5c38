// kernel-argument preload probe: cycles from kernel entry until a kernel argument is usable, arguments read from memory vs
// preloaded into SGPRs (-mllvm -amdgpu-kernarg-preload-count=4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Tail { int a[100]; };
__global__ void k3(unsigned long long* __restrict__ out, int v, const Tail t)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    int x = v;
    asm volatile("s_nop 0" ::"s"(x));                 // the argument must be in a register here
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    int y = t.a[50];                                  // a field beyond the preloaded part
    asm volatile("s_nop 0" ::"s"(y));
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = t2 - t1; }
}
int main()
{
    unsigned long long* d; hipMalloc(&d, 1 << 20);
    Tail t{};
    for (int rep = 0; rep < 3; rep++) {
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k3, dim3(256), dim3(256), 0, 0, d, i, t);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(512);
        hipMemcpy(h.data(), d, 512 * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int w = 0; w < 256; w++) { a += h[2 * w]; b += h[2 * w + 1]; }
        printf("entry -> first argument usable: %.0f cycles; -> a later field of the block: %.0f cycles (mean over 256 workgroups)\n", a / 256, b / 256);
    }
    return 0;
}
