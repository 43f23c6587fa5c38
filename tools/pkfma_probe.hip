// Diagnostic (not part of the product): v_pk_fma_f32 whose destination pair is also its broadcast source pair, op_sel taking
// the HIGH half for the low result -- does it always read the old value?  Checked on the device against scalar v_fma_f32,
// alone and beside a dense-MFMA kernel on another stream.  Forms: A destination == the broadcast source, high half (what the
// compiler emitted in conv_cin1_kernel<5>); B the same with the low half; E / F pure accumulation (destination == addend),
// low half, single and chained five deep; G pure accumulation, high half; H the v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16
// split of the GEMM epilogues.  Result on MI355X: A fails beside MFMA streams (32 ... 600 in 5e11), everything else
// passes -- PROVIDED the references are scalar instructions: this file is built with packed-fp32 code generation, and
// with __builtin_fmaf references the compiler packed pairs of THEM into form A; forms B, E, F, G then "failed" too (32 ...
// 256 per run), which a first reading took for a wider erratum.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfma_probe.hip -o tools/_bin/pkfma_probe -lpthread
#include <hip/hip_runtime.h>
#include <cstdio>
#include <thread>
#include <atomic>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// the references: ONE scalar v_fma_f32 each, as an asm statement -- left to the compiler (this file is built WITH packed-fp32
// code generation) pairs of them become v_pk_fma_f32 themselves, possibly in the very form under test
__device__ __forceinline__ float sfma(float a, float b, float c)
{
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__global__ __launch_bounds__(256) void probe(unsigned* err, int rounds, unsigned seed)
{
    unsigned bad_a = 0, bad_b = 0, bad_c = 0, bad_d = 0, bad_e = 0, bad_f = 0, bad_g = 0, bad_h = 0;
    unsigned s = seed + blockIdx.x * 977u + threadIdx.x * 131u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 9) % 2001 - 1000) * 1e-3f; };
    for (int r = 0; r < rounds; r++) {
        f32x2 w = {rnd(), rnd()}, x = {rnd(), rnd()}, c = {rnd(), rnd()};
        // form A (conv_cin1_kernel<5>): both halves take src1's HIGH half
        f32x2 xa = x;
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0]" : "+v"(xa) : "v"(w), "v"(c));
        const float ea0 = sfma(w[0], x[1], c[0]), ea1 = sfma(w[1], x[1], c[1]);
        // form B: both halves take src1's LOW half
        f32x2 xb = x;
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[1,0,1]" : "+v"(xb) : "v"(w), "v"(c));
        const float eb0 = sfma(w[0], x[0], c[0]), eb1 = sfma(w[1], x[0], c[1]);
        // form E (round 2): pure accumulation -- vdst == src2, the multiplier pair `w` per half, the multiplicand the LOW half of
        // a third pair for both halves; no half reads a register the other half writes.  F: five of them back to back on one
        // accumulator (the shape of a first-convolution tap loop)
        f32x2 acc_e = c;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc_e) : "v"(w), "v"(x));
        const float ee0 = sfma(w[0], x[0], c[0]), ee1 = sfma(w[1], x[0], c[1]);
        bad_e += (acc_e[0] != ee0) + (acc_e[1] != ee1);
        f32x2 w2 = {rnd(), rnd()}, x2 = {rnd(), rnd()}, acc_f = c;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %3, %4, %0 op_sel_hi:[1,0,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %3, %2, %0 op_sel_hi:[1,0,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]"
                     : "+v"(acc_f) : "v"(w), "v"(x), "v"(w2), "v"(x2));
        float f0 = c[0], f1 = c[1];
        f0 = sfma(w[0], x[0], f0);  f1 = sfma(w[1], x[0], f1);
        f0 = sfma(w2[0], x2[0], f0); f1 = sfma(w2[1], x2[0], f1);
        f0 = sfma(w[0], x2[0], f0); f1 = sfma(w[1], x2[0], f1);
        f0 = sfma(w2[0], x[0], f0); f1 = sfma(w2[1], x[0], f1);
        f0 = sfma(w[0], x[0], f0);  f1 = sfma(w[1], x[0], f1);
        bad_f += (acc_f[0] != f0) + (acc_f[1] != f1);
        // form G: the same accumulation with the HIGH half of the multiplicand pair for both halves (the odd taps of a pair load)
        f32x2 acc_g = c;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc_g) : "v"(w), "v"(x));
        const float eg0 = sfma(w[0], x[1], c[0]), eg1 = sfma(w[1], x[1], c[1]);
        bad_g += (acc_g[0] != eg0) + (acc_g[1] != eg1);
        // form H (round 2): the hi / lo split of the GEMM epilogues (pnn_device_common.h split4): v_cvt_pk_f16_f32 for the pair,
        // then lo = (f16)(v - hi) by v_fma_mixlo_f16 / v_fma_mixhi_f16, whose third operand is the LOW / HIGH f16 half of the
        // packed hi register (op_sel on a VOP3P encoding, like the failing v_pk_fma_f32 form) -- against the compiler's scalar code
        {
            const float v0 = w[0] * 300.f, v1 = x[1] * 300.f;
            unsigned hpk, lpk;
            asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hpk) : "v"(v0), "v"(v1));
            asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                         : "=&v"(lpk) : "v"(v0), "v"(v1), "v"(hpk));
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
            const unsigned eh = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            const unsigned el = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
            bad_h += (hpk != eh) + (lpk != el);
            if ((hpk != eh || lpk != el) && atomicAdd(err + 8, 1u) == 0) {   // the first mismatch, for the report
                err[9] = __builtin_bit_cast(unsigned, v0); err[10] = __builtin_bit_cast(unsigned, v1); err[11] = hpk; err[12] = eh; err[13] = lpk; err[14] = el;
            }
        }
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
    if (bad_h) atomicAdd(err + 7, bad_h);
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
    hipMalloc(&derr, 64); hipMemset(derr, 0, 64); hipMalloc(&dp, 64);
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
    unsigned h[16]; hipMemcpy(h, derr, 64, hipMemcpyDeviceToHost);
    if (h[8]) printf("first split mismatch: v = (%g, %g): hi %08x expected %08x, lo %08x expected %08x\n", __builtin_bit_cast(float, h[9]), __builtin_bit_cast(float, h[10]), h[11], h[12], h[13], h[14]);
    printf("beside %d MFMA partner thread(s), %d launches x 2048 x 256 threads x 2000 rounds: v_pk_fma_f32 in place, form A (op_sel hi) %u wrong, form B (op_sel lo) %u wrong; "
           "v_lshl_add_u64 in place %u wrong, v_mad_u64_u32 in place %u wrong; v_pk_fma_f32 accumulating (vdst == src2, low-half broadcast) %u wrong, five of them chained %u wrong, with the high half as multiplicand %u wrong; v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 split %u wrong\n",
           partners, reps, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    return 0;
}
