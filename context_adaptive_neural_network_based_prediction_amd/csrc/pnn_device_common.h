// Device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace pnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Kernel arguments live in host-visible memory: the first scalar load of each 64-byte line of the argument block is a
// round trip of 1-2 us, and the compiler requests the lines one by one where their fields are first used (measured in
// tapgemm_ring_kernel: 4.5k cycles of "setup").  Touch every line of an N-byte argument block in ONE batch at kernel
// entry; the later field loads then hit the scalar cache.
template <int N>
__device__ __forceinline__ void touch_kernargs()
{
    typedef const int __attribute__((address_space(4))) kint;
    kint* k = (kint*)__builtin_amdgcn_kernarg_segment_ptr();
    int t = 0;
#pragma unroll
    for (int off = 0; off < N; off += 64) t += k[off / 4];
    asm volatile("" ::"s"(t));
}

// pnn/tfutils.py:192, max(0.1 v, v).  fmaxf() would put a canonicalising v_max_f32 v, v in front (the IEEE quieting of a
// signalling NaN the compiler cannot rule out): three VALU instructions per value in epilogues that are VALU-bound; the
// values here come out of v_fma_f32, which never produces a signalling NaN.
__device__ __forceinline__ float leaky(float v)
{
    const float t = 0.1f * v;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(v));
    return r;
}

// TComPrediction.cpp:632: (int) std::round(max(0, min(255, p + mean))), half away from zero.
__device__ __forceinline__ int hm_round(float p, float mean)
{
    float v = p + mean;
    v = fminf(v, 255.f);
    v = fmaxf(v, 0.f);
    return (int)roundf(v);
}

// Range guard of the split-precision path.  An activation v is carried as hi = (f16) v, lo = (f16)(v - hi); |v| >= 65520
// would make hi infinite and lo NaN, and the HM epilogue would turn that NaN into 255 without a trace.  Every kernel that
// writes split activations therefore tracks max |v| of what it converts and raises the context's flag (host-visible
// memory) when the f16 range is left; pnn_abi.cpp then repeats the pass on the exact-f32 kernels (host entry points) or
// reports PNN_E_RANGE (device entry points, pnn_check_range).  Cost: one v_max_f32 per converted value.
constexpr float kF16Max = 65504.f;
__device__ __forceinline__ float amax4(float m, f32x4 v)
{
    return fmaxf(fmaxf(m, fabsf(v[0])), fmaxf(fmaxf(fabsf(v[1]), fabsf(v[2])), fabsf(v[3])));
}
__device__ __forceinline__ void report_range(int* flag, float amax)
{
    if (flag && !(amax < kF16Max)) *flag = 1;
}

// hi = (f16) v, lo = (f16)(v - (float) hi), four values at a time, in 6 VALU instructions instead of 14: both halves of a
// pair are rounded by one v_cvt_pk_f16_f32 (round to nearest even, like the scalar conversion), and v_fma_mix{lo,hi}_f16
// computes v * 1.0 - hi with the f16 operand widened inside the instruction and ONE rounding to f16 at the end -- the
// same bits as the two-step form, because v - (float) hi is exact in f32 (hi is v rounded to 11 significant bits).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4(f32x4 v, f16x4& hi, f16x4& lo)
{
    unsigned h01, h23, l01, l23;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v[0]), "v"(v[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v[2]), "v"(v[3]));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(l01) : "v"(v[0]), "v"(v[1]), "v"(h01));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(l23) : "v"(v[2]), "v"(v[3]), "v"(h23));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(f16x4, (u2){h01, h23});
    lo = __builtin_bit_cast(f16x4, (u2){l01, l23});
}

// 16-byte store of a layer's output tile, written THROUGH the XCD's L2 (sc0 sc1).  A plain store leaves the line dirty in
// L2 until the end-of-kernel write-back, and the next (dependent) launch waits for that: ~20 MB of activations per FC-8
// layer at batch 4096 are ~3 us at the rate the write-back runs.  Written through, the early workgroups' tiles drain
// while the late ones still compute (measured, same box, plain -> through: FC-8 pass 103.8 -> 102.1 us, conv-16 pass
// 400.8 -> 397.4 us; `nt` alone changes nothing: it is a cache hint, not write-through).  -DPNN_PLAIN_STORES: the A/B build.
// NOT for values that come straight out of an MFMA: the compiler pads the MFMA -> vector-memory read hazard only for its
// own instructions, an asm operand gets no wait states (tried on the fused layer's partial sums: garbage).  The same holds
// for split4 / leaky above -- every caller has a compiler-generated v_fma (scale, bias) between the matrix result and them.
__device__ __forceinline__ void store16_through(f32x4* dst, f32x4 v)
{
#ifdef PNN_PLAIN_STORES
    *dst = v;
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#endif
}

// Split activation layout of the split-precision GEMM: element (pixel, channel n) of a [pixels][C] tensor lives at
// f16 index 2*pixel*C + (n/16)*32 + n%16 (hi) and +16 (lo); x = hi + lo with hi = (f16) x, lo = (f16)(x - hi).
__device__ __forceinline__ void store_split4(void* base, size_t pixel_times_c, int n, f32x4 v, float& amax)
{
    amax = amax4(amax, v);
    f16x4 hi, lo;
    split4(v, hi, lo);
    _Float16* dst = reinterpret_cast<_Float16*>(base) + 2 * pixel_times_c + (n >> 4) * 32 + (n & 15);
    *reinterpret_cast<f16x4*>(dst) = hi;
    *reinterpret_cast<f16x4*>(dst + 16) = lo;
}

__device__ __forceinline__ void store_split1(void* base, size_t pixel_times_c, int n, float v, float& amax)
{
    amax = fmaxf(amax, fabsf(v));
    const _Float16 hi = (_Float16)v;
    _Float16* dst = reinterpret_cast<_Float16*>(base) + 2 * pixel_times_c + (n >> 4) * 32 + (n & 15);
    dst[0] = hi;
    dst[16] = (_Float16)(v - (float)hi);
}

}  // namespace pnn
