// Diagnostic (not part of the product): how many independent VALU instructions (v_fma_f32, optionally with LDS reads) fit into
// the shadow of a wave's OWN back-to-back v_mfma_f32_32x32x16_f16 for free?  One or two waves per SIMD, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/mfma_shadow_probe.hip -o tools/_bin/mfma_shadow_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, bool LDS>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mfma_waves)
{
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = (float)i * 1e-3f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    f32x16 acc[3];
    for (int k2 = 0; k2 < 3; k2++) for (int i = 0; i < 16; i++) acc[k2][i] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    float v[3][NV > 0 ? NV : 1];
    for (int m = 0; m < 3; m++) for (int i = 0; i < (NV > 0 ? NV : 1); i++) v[m][i] = 0.5f + i + m;
    const float w = 1.0001f;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < mfma_waves) {
        t0 = __builtin_amdgcn_s_memtime();
        int idx = threadIdx.x & 1023;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 3; m++) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    float x = w;
                    if (LDS && (j & 3) == 0) { x = sm[idx]; idx = (idx + 64) & 4095; }
                    v[m][j] = __builtin_fmaf(v[m][j], x, 0.25f);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (NV) __builtin_amdgcn_sched_group_barrier(LDS ? 0x102 : 0x002, NV + (LDS ? (NV + 3) / 4 : 0), 0);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    float s = 0.f;
    for (int k2 = 0; k2 < 3; k2++) for (int i = 0; i < 16; i++) s += acc[k2][i];
    for (int m = 0; m < 3; m++) for (int i = 0; i < (NV > 0 ? NV : 1); i++) s += v[m][i];
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, bool LDS>
static void run(int threads, int mfma_waves, float* dout, unsigned long long* dc)
{
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, LDS>), dim3(256), dim3(threads), 0, 0, dout, dc, iters, mfma_waves);
    hipLaunchKernelGGL((k<NV, LDS>), dim3(256), dim3(threads), 0, 0, dout, dc, iters, mfma_waves);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, dc, sizeof h, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; i++) s += (double)h[i];
    printf("  %2d VALU%s per MFMA: %.1f cycles per MFMA\n", NV, LDS ? " (every 4th operand from LDS)" : "", s / 256 / iters / 3);
}

int main()
{
    float* dout; unsigned long long* dc;
    hipMalloc(&dout, 64); hipMalloc(&dc, 256 * 8);
    for (int cfg = 0; cfg < 2; cfg++) {
        const int threads = cfg == 0 ? 256 : 512, mw = 4;
        printf("%d waves per workgroup (%d per SIMD), waves 0-3 issue MFMAs, the others only wait:\n", threads / 64, threads / 256);
        run<0, false>(threads, mw, dout, dc); run<2, false>(threads, mw, dout, dc); run<4, false>(threads, mw, dout, dc);
        run<6, false>(threads, mw, dout, dc); run<8, false>(threads, mw, dout, dc); run<12, false>(threads, mw, dout, dc);
        run<4, true>(threads, mw, dout, dc); run<6, true>(threads, mw, dout, dc); run<8, true>(threads, mw, dout, dc);
    }
    return 0;
}
