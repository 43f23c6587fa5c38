// Diagnostic (not part of the product): v_pk_fma_f32 with op_sel taking the HIGH half of a source pair for the LOW result
// -- does it always read the right value?  Checked on the device against scalar fmaf, alone and beside a dense-MFMA kernel on
// another stream.  Forms: A destination == the broadcast source, high half (what the compiler emitted in conv_cin1_kernel<5>);
// B the same with the low half; E / F pure accumulation (destination == addend), low half, single and chained; G pure
// accumulation, HIGH half.  Result on MI355X: A and G fail beside MFMA streams (tens to hundreds in 5e11), B, E, F never --
// so it is the high-half selection that misreads, not the in-place destination (round 1's reading of A alone).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfma_probe.hip -o build_tmp/pkfma_probe -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <atomic>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void probe(unsigned* err, int rounds, unsigned seed)
{
    unsigned bad_a = 0, bad_b = 0, bad_c = 0, bad_d = 0, bad_e = 0, bad_f = 0, bad_g = 0;
    unsigned s = seed + blockIdx.x * 977u + threadIdx.x * 131u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (int r = 0; r < rounds; r++) {
        f32x2 w = {rnd(), rnd()}, x = {rnd(), rnd()}, c = {rnd(), rnd()};
        // form A (conv_cin1_kernel<5>): both halves take src1's HIGH half
        f32x2 xa = x;
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(xa) : "v"(w), "v"(c));
        const float ea0 = __builtin_fmaf(w[0], x[1], c[0]), ea1 = __builtin_fmaf(w[1], x[1], c[1]);
        // form B: both halves take src1's LOW half
        f32x2 xb = x;
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(xb) : "v"(w), "v"(c));
        const float eb0 = __builtin_fmaf(w[0], x[0], c[0]), eb1 = __builtin_fmaf(w[1], x[0], c[1]);
        // form E (round 2): pure accumulation -- vdst == src2, the multiplier pair `w` per half, the multiplicand the LOW half of
        // a third pair for both halves; no half reads a register the other half writes.  F: five of them back to back on one
        // accumulator (the shape of a first-convolution tap loop)
        f32x2 acc_e = c;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc_e) : "v"(w), "v"(x));
        const float ee0 = __builtin_fmaf(w[0], x[0], c[0]), ee1 = __builtin_fmaf(w[1], x[0], c[1]);
        bad_e += (acc_e[0] != ee0) + (acc_e[1] != ee1);
        f32x2 w2 = {rnd(), rnd()}, x2 = {rnd(), rnd()}, acc_f = c;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %3, %4, %0 op_sel_hi:[1,0,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %3, %2, %0 op_sel_hi:[1,0,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]"
                     : "+v"(acc_f) : "v"(w), "v"(x), "v"(w2), "v"(x2));
        float f0 = c[0], f1 = c[1];
        f0 = __builtin_fmaf(w[0], x[0], f0);  f1 = __builtin_fmaf(w[1], x[0], f1);
        f0 = __builtin_fmaf(w2[0], x2[0], f0); f1 = __builtin_fmaf(w2[1], x2[0], f1);
        f0 = __builtin_fmaf(w[0], x2[0], f0); f1 = __builtin_fmaf(w[1], x2[0], f1);
        f0 = __builtin_fmaf(w2[0], x[0], f0); f1 = __builtin_fmaf(w2[1], x[0], f1);
        f0 = __builtin_fmaf(w[0], x[0], f0);  f1 = __builtin_fmaf(w[1], x[0], f1);
        bad_f += (acc_f[0] != f0) + (acc_f[1] != f1);
        // form G: the same accumulation with the HIGH half of the multiplicand pair for both halves (the odd taps of a pair load)
        f32x2 acc_g = c;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc_g) : "v"(w), "v"(x));
        const float eg0 = __builtin_fmaf(w[0], x[1], c[0]), eg1 = __builtin_fmaf(w[1], x[1], c[1]);
        bad_g += (acc_g[0] != eg0) + (acc_g[1] != eg1);
        bad_a += (xa[0] != ea0) + (xa[1] != ea1);
        bad_b += (xb[0] != eb0) + (xb[1] != eb1);
        // the 64-bit integer VALU forms that remain in the library's code (address arithmetic), destination == source
        unsigned long long q = ((unsigned long long)s << 32) | (s * 2654435761u), q0 = q, add = ((unsigned long long)(s >> 7) << 29) | 0xfffffff0u;
        asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q) : "v"(add));
        bad_c += q != ((q0 << 3) + add);
        unsigned long long m = q0;
        const unsigned ma = s | 0x80000001u, mb = (s >> 3) | 0x40000000u;
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m) : "v"(ma), "v"(mb) : "vcc");
        bad_d += m != (unsigned long long)ma * mb + q0;
    }
    if (bad_a) atomicAdd(err, bad_a);
    if (bad_b) atomicAdd(err + 1, bad_b);
    if (bad_c) atomicAdd(err + 2, bad_c);
    if (bad_d) atomicAdd(err + 3, bad_d);
    if (bad_e) atomicAdd(err + 4, bad_e);
    if (bad_f) atomicAdd(err + 5, bad_f);
    if (bad_g) atomicAdd(err + 6, bad_g);
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
    const int partners = argc > 1 ? atoi(argv[1]) : 2, reps = argc > 2 ? atoi(argv[2]) : 500;
    unsigned* derr; float* dp;
    hipMalloc(&derr, 32); hipMemset(derr, 0, 32); hipMalloc(&dp, 64);
    std::atomic<bool> stop{false};
    std::vector<std::thread> ts;
    for (int t = 0; t < partners; t++)
        ts.emplace_back([&]() {
            hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            while (!stop) {
                for (int k = 0; k < 8; k++) hipLaunchKernelGGL(mfma_partner, dim3(1024), dim3(256), 0, s, dp, 4000);
                hipStreamSynchronize(s);
            }
        });
    hipStream_t sv; hipStreamCreateWithFlags(&sv, hipStreamNonBlocking);
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(probe, dim3(2048), dim3(256), 0, sv, derr, 2000, (unsigned)r * 7919u);
        if (r % 8 == 7) hipStreamSynchronize(sv);
    }
    hipStreamSynchronize(sv);
    stop = true;
    for (auto& t : ts) t.join();
    unsigned h[8]; hipMemcpy(h, derr, 32, hipMemcpyDeviceToHost);
    printf("beside %d MFMA partner thread(s), %d launches x 2048 x 256 threads x 2000 rounds: v_pk_fma_f32 in place, form A (op_sel hi) %u wrong, form B (op_sel lo) %u wrong; "
           "v_lshl_add_u64 in place %u wrong, v_mad_u64_u32 in place %u wrong; v_pk_fma_f32 accumulating (vdst == src2, low-half broadcast) %u wrong, five of them chained %u wrong, with the high half as multiplicand %u wrong\n",
           partners, reps, h[0], h[1], h[2], h[3], h[4], h[5], h[6]);
    return 0;
}
