// Diagnostic (not part of the product): does a plain VALU kernel give repeatable results while a dense-MFMA kernel from
// another host thread / stream shares the chip?  No code of the library is involved.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/corun_probe.hip -o tools/_bin/corun_probe -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void victim(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int npix)
{
    __shared__ float xs[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) xs[i] = x[(blockIdx.x * 4096 + i) & 0xfffff];
    const int cg = threadIdx.x & 15, psub = threadIdx.x >> 4;
    f32x4 wv[25];
    for (int t = 0; t < 25; t++) wv[t] = *reinterpret_cast<const f32x4*>(w + t * 64 + 4 * cg);
    __syncthreads();
    for (int pix = psub; pix < npix; pix += 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 25; t++) acc += xs[(pix * 3 + t * 7) & 4095] * wv[t];
        for (int i = 0; i < 4; i++) acc[i] = fmaxf(0.1f * acc[i], acc[i]);
        *reinterpret_cast<f32x4*>(y + ((size_t)blockIdx.x * npix + pix) * 64 + 4 * cg) = acc;
    }
}

__global__ __launch_bounds__(256) void mfma_partner(float* out, int iters)
{
    f32x16 acc[4];
    for (int k = 0; k < 4; k++) for (int i = 0; i < 16; i++) acc[k][i] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    for (int it = 0; it < iters; it++)
        for (int k = 0; k < 4; k++) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    float s = 0.f;
    for (int k = 0; k < 4; k++) for (int i = 0; i < 16; i++) s += acc[k][i];
    if (s == 12345.f) out[0] = s;
}

int main(int argc, char** argv)
{
    const int partners = argc > 1 ? atoi(argv[1]) : 2, reps = argc > 2 ? atoi(argv[2]) : 2000, iters = argc > 3 ? atoi(argv[3]) : 4000;
    const int nwg = 1024, npix = 192;
    float *dx, *dw, *dy, *dp;
    hipMalloc(&dx, 4 << 20); hipMalloc(&dw, 25 * 64 * 4); hipMalloc(&dy, (size_t)nwg * npix * 64 * 4); hipMalloc(&dp, 64);
    std::vector<float> hx(1 << 20), hw(25 * 64);
    srand(3);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 1e-3f;
    hipMemcpy(dx, hx.data(), 4 << 20, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), 25 * 64 * 4, hipMemcpyHostToDevice);
    std::atomic<bool> stop{false};
    std::vector<std::thread> ts;
    for (int t = 0; t < partners; t++)
        ts.emplace_back([&]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            while (!stop) {
                for (int k = 0; k < 8; k++) hipLaunchKernelGGL(mfma_partner, dim3(1024), dim3(256), 0, s, dp, iters);
                hipStreamSynchronize(s);
            }
        });
    hipStream_t sv; hipStreamCreateWithFlags(&sv, hipStreamNonBlocking);
    const size_t yb = (size_t)nwg * npix * 64 * 4;
    std::vector<char> ref(yb), got(yb);
    hipLaunchKernelGGL(victim, dim3(nwg), dim3(256), 0, sv, dx, dw, dy, npix);
    hipStreamSynchronize(sv);
    hipMemcpy(ref.data(), dy, yb, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < reps; r++) {
        hipMemsetAsync(dy, 0, yb, sv);
        hipLaunchKernelGGL(victim, dim3(nwg), dim3(256), 0, sv, dx, dw, dy, npix);
        hipStreamSynchronize(sv);
        hipMemcpy(got.data(), dy, yb, hipMemcpyDeviceToHost);
        if (memcmp(ref.data(), got.data(), yb)) {
            if (++bad <= 3) {
                const float* a = (const float*)ref.data(); const float* b = (const float*)got.data();
                size_t nd = 0, first = 0;
                for (size_t i = 0; i < yb / 4; i++) if (memcmp(a + i, b + i, 4)) { if (!nd) first = i; ++nd; }
                printf("rep %d: %zu floats differ, first index %zu (element %zu of its float4, lane group %zu): want %g got %g\n", r, nd, first, first & 3, (first >> 2) & 15, a[first], b[first]);
            }
        }
    }
    stop = true;
    for (auto& t : ts) t.join();
    printf("victim beside %d MFMA partner thread(s): %d / %d runs differ\n", partners, bad, reps);
    return 0;
}
