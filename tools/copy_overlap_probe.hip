// Does a host <-> device copy overlap a kernel that fills the chip?  (round 6: the sliced host calls overlapped less than their copies are long)
//   hipcc --offload-arch=gfx950 -O3 tools/copy_overlap_probe.hip -o tools/_bin/copy_overlap_probe
// A busy kernel of `wgs` workgroups x 256 threads spinning for `us` microseconds on stream A; an H2D / D2H copy of `mb` MiB on stream B,
// from pinned (hipHostMalloc) and from pageable memory; each alone and both together, wall clock.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void busy_kernel(unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const double us = argc > 2 ? atof(argv[2]) : 1000.0;
    const size_t bytes = (size_t)(argc > 3 ? atof(argv[3]) : 5.0) * (1 << 20);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    void *dev, *pin;
    CK(hipMalloc(&dev, bytes));
    CK(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    std::vector<char> page(bytes, 1);
    memset(pin, 1, bytes);
    auto kernel = [&]() { hipLaunchKernelGGL(busy_kernel, dim3(wgs), dim3(256), 0, a, (unsigned)(us * 100.0)); };
    for (int i = 0; i < 3; i++) { kernel(); CK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, b)); CK(hipMemcpyAsync(dev, page.data(), bytes, hipMemcpyHostToDevice, b)); CK(hipDeviceSynchronize()); }
    struct Case { const char* name; void* host; hipMemcpyKind kind; } cases[] = {{"H2D pinned", pin, hipMemcpyHostToDevice}, {"H2D pageable", page.data(), hipMemcpyHostToDevice},
                                                                                  {"D2H pinned", pin, hipMemcpyDeviceToHost}, {"D2H pageable", page.data(), hipMemcpyDeviceToHost}};
    double t0 = now();
    for (int i = 0; i < 10; i++) { kernel(); CK(hipStreamSynchronize(a)); }
    const double tk = (now() - t0) / 10;
    printf("busy kernel of %d workgroups alone: %.3f ms\n", wgs, tk * 1e3);
    for (const Case& cs : cases) {
        auto copy = [&]() { return cs.kind == hipMemcpyHostToDevice ? hipMemcpyAsync(dev, cs.host, bytes, cs.kind, b) : hipMemcpyAsync(cs.host, dev, bytes, cs.kind, b); };
        t0 = now();
        for (int i = 0; i < 10; i++) { CK(copy()); CK(hipStreamSynchronize(b)); }
        const double tc = (now() - t0) / 10;
        t0 = now();
        for (int i = 0; i < 10; i++) { kernel(); CK(copy()); CK(hipStreamSynchronize(b)); CK(hipStreamSynchronize(a)); }
        const double tb = (now() - t0) / 10;
        t0 = now();
        double call = 0;
        for (int i = 0; i < 10; i++) { kernel(); const double c0 = now(); CK(copy()); call += now() - c0; CK(hipDeviceSynchronize()); }
        printf("%-13s %5.1f MiB alone %.3f ms (%.1f GB/s); beside the kernel: both done after %.3f ms (sum %.3f, max %.3f); the copy call itself returns after %.3f ms\n", cs.name, bytes / 1048576.0,
               tc * 1e3, bytes / tc / 1e9, tb * 1e3, (tk + tc) * 1e3, (tk > tc ? tk : tc) * 1e3, call / 10 * 1e3);
    }
    return 0;
}
