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

__device__ __forceinline__ float leaky(float v) { return fmaxf(0.1f * v, v); }   // pnn/tfutils.py:192

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

// Split activation layout of the split-precision GEMM: element (pixel, channel n) of a [pixels][C] tensor lives at
// f16 index 2*pixel*C + (n/16)*32 + n%16 (hi) and +16 (lo); x = hi + lo with hi = (f16) x, lo = (f16)(x - hi).
__device__ __forceinline__ void store_split4(void* base, size_t pixel_times_c, int n, f32x4 v, float& amax)
{
    amax = amax4(amax, v);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; i++) { hi[i] = (_Float16)v[i]; lo[i] = (_Float16)(v[i] - (float)hi[i]); }
    _Float16* dst = reinterpret_cast<_Float16*>(base) + 2 * pixel_times_c + (n >> 4) * 32 + (n & 15);
    *reinterpret_cast<h4*>(dst) = hi;
    *reinterpret_cast<h4*>(dst + 16) = lo;
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
