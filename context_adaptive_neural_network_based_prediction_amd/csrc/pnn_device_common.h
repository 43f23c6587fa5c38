// Device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace pnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float leaky(float v) { return fmaxf(0.1f * v, v); }   // pnn/tfutils.py:192

// TComPrediction.cpp:632: (int) std::round(max(0, min(255, p + mean))), half away from zero.
__device__ __forceinline__ int hm_round(float p, float mean)
{
    float v = p + mean;
    v = fminf(v, 255.f);
    v = fmaxf(v, 0.f);
    return (int)roundf(v);
}

}  // namespace pnn
