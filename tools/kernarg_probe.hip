// Diagnostic (not part of the product): is a kernel's argument block stable for the whole life of the kernel when several
// host threads launch on their own streams?  Every workgroup re-reads its 1 KiB argument block from memory (scalar cache
// invalidated each round) and counts words that are not what the host passed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/kernarg_probe.hip -o tools/_bin/kernarg_probe -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <vector>
#include <atomic>
struct Big { unsigned v[240]; unsigned seed; int rounds; unsigned* err; };
__global__ __launch_bounds__(64) void probe(const Big b)
{
    typedef const unsigned __attribute__((address_space(4))) cu;
    cu* k = (cu*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned bad = 0;
    for (int r = 0; r < b.rounds; r++) {
        asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        for (int i = 0; i < 240; i++) {
            unsigned x;
            asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) : "s"(k), "s"(4 * i) : "memory");
            if (x != b.seed * 1000u + i) ++bad;
        }
        __builtin_amdgcn_s_sleep(20);
    }
    if (bad && threadIdx.x == 0) atomicAdd(b.err, bad);
}
__global__ void filler(float* p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f; }
int main(int argc, char** argv)
{
    const int nthreads = argc > 1 ? atoi(argv[1]) : 3, launches = argc > 2 ? atoi(argv[2]) : 3000;
    unsigned* derr; hipMalloc(&derr, 4 * 8); hipMemset(derr, 0, 32);
    float* dbuf; hipMalloc(&dbuf, 64 << 20);
    std::vector<std::thread> ts;
    for (int t = 0; t < nthreads; t++)
        ts.emplace_back([=]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            for (int l = 0; l < launches; l++) {
                Big b; b.seed = 1 + t * 100000 + l; b.rounds = 20 + (l % 7) * 10; b.err = derr + t;
                for (int i = 0; i < 240; i++) b.v[i] = b.seed * 1000u + i;
                hipLaunchKernelGGL(probe, dim3(64 + 32 * (l % 5)), dim3(64), 0, s, b);
                if (t == 0 && l % 4 == 0) hipLaunchKernelGGL(filler, dim3(2048), dim3(256), 0, s, dbuf, 16 << 20);
                if (l % 16 == 15) hipStreamSynchronize(s);
            }
            hipStreamSynchronize(s);
        });
    for (auto& t : ts) t.join();
    unsigned h[8]; hipMemcpy(h, derr, 32, hipMemcpyDeviceToHost);
    for (int t = 0; t < nthreads; t++) printf("thread %d: %u wrong argument words seen\n", t, h[t]);
    return 0;
}
