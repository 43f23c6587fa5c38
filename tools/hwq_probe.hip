// Round 5 probe: which HIP streams share a hardware queue?  N streams are created one after the other; for every pair (i, j) a 300 us
// kernel goes to stream i and, right behind it, an empty kernel to stream j: if the empty kernel completes only after the long one, the
// two streams sit on one hardware queue (the runtime keeps GPU_MAX_HW_QUEUES = 4 of them and deals streams onto them).
//   hipcc --offload-arch=gfx950 -O2 tools/hwq_probe.hip -o tools/_bin/hwq_probe ; tools/_bin/hwq_probe [streams = 10]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void busy(unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
__global__ void empty() {}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<hipStream_t> s(n);
    for (int i = 0; i < n; i++) if (hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) != hipSuccess) return 1;
    for (int i = 0; i < n; i++) { hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, s[i]); hipStreamSynchronize(s[i]); }   // first use
    std::vector<int> group(n, -1);
    int ngroups = 0;
    printf("time until an empty kernel on stream j completes behind a 300 us kernel on stream i (us):\n      ");
    for (int j = 0; j < n; j++) printf(" j=%-4d", j);
    printf("\n");
    for (int i = 0; i < n; i++) {
        printf("i=%-3d ", i);
        for (int j = 0; j < n; j++) {
            if (i == j) { printf("   -   "); continue; }
            hipLaunchKernelGGL(busy, dim3(1), dim3(64), 0, s[i], 30000u);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(empty, dim3(1), dim3(64), 0, s[j]);
            hipStreamSynchronize(s[j]);
            const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
            hipStreamSynchronize(s[i]);
            printf(" %6.0f", us);
            if (us > 200.0) {                          // same hardware queue
                if (group[i] < 0 && group[j] < 0) group[i] = group[j] = ngroups++;
                else if (group[i] < 0) group[i] = group[j];
                else if (group[j] < 0) group[j] = group[i];
            }
        }
        printf("\n");
    }
    for (int i = 0; i < n; i++) if (group[i] < 0) group[i] = ngroups++;
    printf("streams in creation order -> hardware queue (labels in order of discovery):");
    for (int i = 0; i < n; i++) printf(" %d", group[i]);
    printf("\n");
    return 0;
}
