// Diagnostic: sustained f32 / f16 MFMA rate and in-kernel clock on this device (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, unsigned long long* clk)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, unsigned long long* clk)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}


typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// f16 32x32x16 (the split-precision kernels' instruction).  DEP = 1 issues, per accumulator, the hi*hi product for all
// accumulators first and then the two cross products back to back on the SAME accumulator (tapgemm_sp_kernel's order).
template <int NACC, int DEP>
__global__ __launch_bounds__(256) void kh32(float* out, int iters, unsigned long long* clk)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0;
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(1.0f + threadIdx.x * 1e-4f); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (DEP) {
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <typename F>
void run(const char* name, F launch, int blocks, int iters, double flop_per_iter_per_wave, float* out, unsigned long long* clk)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0);
        launch(blocks, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * flop_per_iter_per_wave;
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        if (rep == 3) printf("%-28s blocks=%4d: %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", name, blocks, ms, flops / ms / 1e9, (double)h[0] / h[1] * 100.0);
    }
}

int main2(float* out, unsigned long long* clk)
{
    for (int blocks : {256, 512, 1024}) {
        run("16x16x4 f32, 8 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<8>, dim3(b), dim3(256), 0, 0, out, it, clk); }, blocks, 40000, 8 * 2048.0, out, clk);
        run("16x16x4 f32, 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<4>, dim3(b), dim3(256), 0, 0, out, it, clk); }, blocks, 80000, 4 * 2048.0, out, clk);
        run("32x32x2 f32, 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k32<4>, dim3(b), dim3(256), 0, 0, out, it, clk); }, blocks, 40000, 4 * 4096.0, out, clk);
        const double fh = 3 * 2.0 * 32 * 32 * 16;
        run("32x32x16 f16, 4 acc indep", [&](int b, int it) { hipLaunchKernelGGL((kh32<4, 0>), dim3(b), dim3(256), 0, 0, out, it, clk); }, blocks, 20000, 4 * fh, out, clk);
        run("32x32x16 f16, 4 acc sp-order", [&](int b, int it) { hipLaunchKernelGGL((kh32<4, 1>), dim3(b), dim3(256), 0, 0, out, it, clk); }, blocks, 20000, 4 * fh, out, clk);
        run("32x32x16 f16, 2 acc sp-order", [&](int b, int it) { hipLaunchKernelGGL((kh32<2, 1>), dim3(b), dim3(256), 0, 0, out, it, clk); }, blocks, 40000, 2 * fh, out, clk);
        run("32x32x16 f16, 6 acc sp-order", [&](int b, int it) { hipLaunchKernelGGL((kh32<6, 1>), dim3(b), dim3(256), 0, 0, out, it, clk); }, blocks, 20000, 6 * fh, out, clk);
    }
    return 0;
}

int main()
{
    float* out; unsigned long long* clk;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 1024 * 16);
    return main2(out, clk);
}
