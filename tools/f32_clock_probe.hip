// Diagnostic (not part of the product): what costs tapgemm_f32_kernel its clock?  One wave per SIMD, five 32x32 accumulators, 40
// v_mfma_f32_32x32x2_f32 per "chunk" like the 128 x 160 tile -- with, per chunk and switchable: 10 ds_read_b128 weight fragments
// from LDS (L), 2 global 16-byte loads per lane with the FC layers' 4800-byte row stride (G), 2.5 LDS-DMA refills of 1 KiB (D), and
// operands that are random data instead of two constant registers (R).  Prints TFLOP/s and the in-kernel clock
// (s_memtime / s_memrealtime x 100 MHz) per variant: the bench's exact-f32 FC kernel runs its loop at 0.976 of the matrix rate in
// CYCLES while the SMU reports 2.4 GHz and 1160 W of a 1400 W cap, yet the cycles themselves come at 2.07 GHz.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/f32_clock_probe.hip -o tools/_bin/f32_clock_probe && gpurun -- ./tools/_bin/f32_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool L, bool G, bool D, bool R>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ x, const float* __restrict__ w, float* out, int chunks, unsigned long long* clk)
{
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];      // 2 x 640 pieces of "weights" + a DMA landing area
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * 640; i += 256) lds[i] = reinterpret_cast<const f32x4*>(w)[(blockIdx.x * 1280 + i) & 0xffff];
    __syncthreads();
    f32x16 acc[5];
    for (int i = 0; i < 5; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    f32x4 wf[5][2], a[2];
    for (int nt = 0; nt < 5; nt++) { wf[nt][0] = lds[(2 * h) * 160 + nt * 32 + l31]; wf[nt][1] = lds[(2 * h + 1) * 160 + nt * 32 + l31]; }
    const size_t row = (size_t)(blockIdx.x * 128 + wave * 32 + l31) * 1200 + 8 * h;
    a[0] = *reinterpret_cast<const f32x4*>(x + row); a[1] = *reinterpret_cast<const f32x4*>(x + row + 4);
    if (!R) { for (int nt = 0; nt < 5; nt++) { wf[nt][0] = (f32x4){1.f, 1.f, 1.f, 1.f}; wf[nt][1] = wf[nt][0]; } a[0] = (f32x4){1e-3f, 1e-3f, 1e-3f, 1e-3f}; a[1] = a[0]; }
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 0xffffffffu, 0x00020000);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < chunks; c++) {
        f32x4 wn[5][2], an[2];
#pragma unroll
        for (int e = 0; e < 8; e++) {
#pragma unroll
            for (int nt = 0; nt < 5; nt++) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][e >> 2][e & 3], a[e >> 2][e & 3], acc[nt], 0, 0, 0);
                const int sl = e * 5 + nt;
                if (L && sl % 4 == 0 && sl / 4 < 10) { const int k = sl / 4; wn[k >> 1][k & 1] = lds[((c & 1) * 640) + (2 * h + (k & 1)) * 160 + (k >> 1) * 32 + l31]; }
                if (G && (sl == 13 || sl == 17)) an[sl == 17] = *reinterpret_cast<const f32x4*>(x + row + (size_t)((c + 1) % 75) * 16 + 4 * (sl == 17));
                if (D && (sl == 22 || sl == 30 || (sl == 38 && (c & 1))))
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(lds + 1280 + 64 * wave), 16,
                                                             (unsigned)(((blockIdx.x * 40 + c * 3 + sl) & 0xfff) * 1024 + lane * 16), 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (L) for (int nt = 0; nt < 5; nt++) { wf[nt][0] = R ? wn[nt][0] : wf[nt][0]; wf[nt][1] = R ? wn[nt][1] : wf[nt][1]; if (!R) asm volatile("" ::"v"(wn[nt][0]), "v"(wn[nt][1])); }
        if (G) { if (R) { a[0] = an[0]; a[1] = an[1]; } else asm volatile("" ::"v"(an[0]), "v"(an[1])); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 5; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <bool L, bool G, bool D, bool R>
void run(const char* name, const float* x, const float* w, float* out, unsigned long long* clk)
{
    const int chunks = 1500, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)probe<L, G, D, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    float ms = 0;
    const int reps = getenv("PROBE_REPS") ? atoi(getenv("PROBE_REPS")) : 5;      // 1500 repetitions = 2.5 s of back-to-back launches
    for (int rep = 0; rep < reps; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<L, G, D, R>), dim3(blocks), dim3(256), (1280 + 256) * 16, 0, x, w, out, chunks, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; i++) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    const double flops = (double)blocks * 4 * chunks * 40 * 4096.0;
    printf("%-52s %.3f ms  %6.1f TFLOP/s  in-kernel clock %4.0f MHz  cycles per chunk %.0f (MFMA work 2560)\n", name, ms, flops / ms / 1e9, cyc / rt * 100.0,
           cyc / blocks / chunks);
}

int main()
{
    float *x, *w, *out; unsigned long long* clk;
    const size_t nx = (size_t)256 * 128 * 1200, nw = (size_t)1 << 20;
    std::vector<float> hx(nx), hw(nw);
    srand(3);
    for (auto& v : hx) v = (rand() % 20001 - 10000) * 1e-4f;
    for (auto& v : hw) v = (rand() % 20001 - 10000) * 1e-5f;
    hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4 + 8 * 1024 * 1024); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 16);
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
    run<false, false, false, false>("MFMA only, constant operands", x, w, out, clk);
    run<false, false, false, true>("MFMA only, random operands", x, w, out, clk);
    run<true, false, false, true>("+ LDS fragment reads (random operands)", x, w, out, clk);
    run<false, true, false, true>("+ global activation loads (random operands)", x, w, out, clk);
    run<false, false, true, true>("+ LDS-DMA refills (random operands)", x, w, out, clk);
    run<true, true, true, true>("+ all three (random operands)", x, w, out, clk);
    run<true, true, true, false>("+ all three, constant operands", x, w, out, clk);
    return 0;
}
