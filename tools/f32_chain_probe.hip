// Round 5: what one dependent chain of v_mfma_f32_16x16x4_f32 costs per instruction on gfx950, alone and with the per-instruction work
// of tapgemm_f32_small_kernel around it (operand selects on the VALU, LDS fragment reads): cycles by s_memtime, one wave.
//   hipcc --offload-arch=gfx950 -O3 tools/f32_chain_probe.hip -o tools/_bin/f32_chain_probe && gpurun -- ./tools/_bin/f32_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(64) void probe(const float* in, float* out, unsigned long long* cyc, int n)
{
    __shared__ f32x4 lds[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = (f32x4){in[i & 63], 1.f, 2.f, 3.f};
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc32;
    for (int i = 0; i < 16; i++) acc32[i] = 0.f;
    const bool odd = (lane >> 5) != 0;
    f32x4 w0 = lds[lane], w1 = lds[lane + 64], x0 = lds[lane + 128], x1 = lds[lane + 192];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it++) {
        if (MODE == 0) {                              // the bare chain: 4 dependent MFMAs per iteration
#pragma unroll
            for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i], x0[i], acc, 0, 0, 0);
        } else if (MODE == 1) {                       // + lane-dependent element selects (2 per MFMA)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const f32x4 wv = (i >> 1) ? w1 : w0, xv = (i >> 1) ? x1 : x0;
                const float w = odd ? wv[2 * (i & 1) + 1] : wv[2 * (i & 1)];
                const float x = odd ? xv[2 * (i & 1) + 1] : xv[2 * (i & 1)];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, acc, 0, 0, 0);
            }
        } else if (MODE == 2) {                       // + the next chunk's four 16-byte LDS reads
            const f32x4* src = lds + ((it & 3) << 8) + lane;
            const f32x4 nw0 = src[0], nw1 = src[64], nx0 = src[128], nx1 = src[192];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const f32x4 wv = (i >> 1) ? w1 : w0, xv = (i >> 1) ? x1 : x0;
                const float w = odd ? wv[2 * (i & 1) + 1] : wv[2 * (i & 1)];
                const float x = odd ? xv[2 * (i & 1) + 1] : xv[2 * (i & 1)];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, acc, 0, 0, 0);
            }
            w0 = nw0; w1 = nw1; x0 = nx0; x1 = nx1;
        } else if (MODE == 6 || MODE == 7 || MODE == 8) {
            // two chunks per iteration, two register sets (no copies): the kernel's real loop.  6: both reads of the next chunk in front of
            // a chunk's four MFMAs; 7: one read behind MFMA 0, one behind MFMA 2; 8: as 6 with a 32-byte-per-lane layout read as ONE
            // ds_read_b128 + nothing (weights only: the activations stay in registers)
            const f32x4* src = lds + ((it & 1) << 9) + lane;
            f32x4 nw, nx;
            if (MODE == 6) { nw = src[0]; nx = src[64]; __builtin_amdgcn_sched_barrier(0); }
            if (MODE == 8) { nw = src[0]; nx = x1; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[i], x0[i], acc, 0, 0, 0);
                if (MODE == 7 && i == 0) { nw = src[0]; __builtin_amdgcn_sched_barrier(0); }
                if (MODE == 7 && i == 2) { nx = src[64]; __builtin_amdgcn_sched_barrier(0); }
            }
            f32x4 mw, mx;
            if (MODE == 6) { mw = src[128]; mx = src[192]; __builtin_amdgcn_sched_barrier(0); }
            if (MODE == 8) { mw = src[128]; mx = x0; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(nw[i], nx[i], acc, 0, 0, 0);
                if (MODE == 7 && i == 0) { mw = src[128]; __builtin_amdgcn_sched_barrier(0); }
                if (MODE == 7 && i == 2) { mx = src[192]; __builtin_amdgcn_sched_barrier(0); }
            }
            w0 = mw; x0 = mx;
        } else if (MODE == 3) {                       // the 32x32x2 chain: 8 dependent MFMAs per 16 k
#pragma unroll
            for (int i = 0; i < 8; i++) acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[i & 3], x0[i & 3], acc32, 0, 0, 0);
        } else if (MODE == 4) {                       // 4x4x1 (16 blocks): 16 dependent MFMAs per 16 k
#pragma unroll
            for (int i = 0; i < 16; i++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(w0[i & 3], x0[i & 3], acc, 0, 0, 0);
        } else if (MODE == 5) {                       // a v_fma_f32 chain: 16 dependent FMAs per 16 k
#pragma unroll
            for (int i = 0; i < 16; i++) acc[0] = __builtin_fmaf(w0[i & 3], x0[i & 3], acc[0]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[MODE] = t1 - t0;
    out[lane + 64 * MODE] = acc[0] + acc[1] + acc[2] + acc[3] + acc32[0] + acc32[5];
}

int main()
{
    float *in, *out; unsigned long long* cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 128);
    hipMemset(in, 0, 4096);
    const int n = 2000;
    const char* names[9] = {"16x16x4 chain, bare", "16x16x4 chain + operand selects (hoisted by the compiler)", "16x16x4 chain + selects + LDS reads (exposed)", "32x32x2 chain, bare", "4x4x1 chain, bare", "v_fma_f32 chain",
                            "16x16x4 chain + 2 LDS reads per chunk, in front", "16x16x4 chain + 2 LDS reads per chunk, interleaved", "16x16x4 chain + 1 LDS read per chunk"};
    const int per[9] = {4, 4, 4, 8, 16, 16, 8, 8, 8};
    const int chunks[9] = {1, 1, 1, 1, 1, 1, 2, 2, 2};
    for (int rep = 0; rep < 2; rep++) {
        probe<0><<<1, 64>>>(in, out, cyc, n); probe<1><<<1, 64>>>(in, out, cyc, n); probe<2><<<1, 64>>>(in, out, cyc, n);
        probe<3><<<1, 64>>>(in, out, cyc, n); probe<4><<<1, 64>>>(in, out, cyc, n); probe<5><<<1, 64>>>(in, out, cyc, n);
        probe<6><<<1, 64>>>(in, out, cyc, n); probe<7><<<1, 64>>>(in, out, cyc, n); probe<8><<<1, 64>>>(in, out, cyc, n);
        hipDeviceSynchronize();
    }
    unsigned long long h[16];
    hipMemcpy(h, cyc, 72, hipMemcpyDeviceToHost);
    for (int m = 0; m < 9; m++)
        printf("%-62s %7.1f cycles per 16-deep chunk, %6.1f per instruction, %5.1f per k\n", names[m], (double)h[m] / n / chunks[m], (double)h[m] / n / per[m], (double)h[m] / n / 16 / chunks[m]);
    return 0;
}
