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

// Completion signal of a host call (option "flag_wait", pnn_abi.cpp): the LAST kernel of the call -- after its results, which it
// writes straight into pinned host memory -- raises a sequence number in pinned host memory, and the host thread spins on that
// word instead of waiting for the runtime's completion signal (which the command processor writes only after the end-of-kernel
// cache write-back, and which hipStreamSynchronize turns into a return a few microseconds later still).  Every thread of every
// workgroup calls signal_done() after its last store; workgroups count themselves on a device-memory counter (agent-scope
// atomics execute memory-side, coherent across the XCDs), the last one resets it and writes the flag.
__device__ __forceinline__ void signal_done(const DoneSignal& d)
{
    if (!d.host_flag) return;                        // launch-uniform
    __threadfence_system();                          // this thread's results are on their way to the host ...
    __syncthreads();                                 // ... and so are the whole workgroup's
    if (threadIdx.x == 0) {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        bool last = nwg == 1;
        if (!last) {
            last = atomicAdd(d.counter, 1u) == nwg - 1;
            if (last) atomicExch(d.counter, 0u);
        }
        if (last) {
            __threadfence_system();
            __hip_atomic_store(d.host_flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The same signal where ONE wave of the workgroup wrote all of the workgroup's results and every workgroup has a flag word of its own
// (DoneSignal::per_wg): called by that wave alone, behind its stores -- one fence, one flag store, nothing to wait for in between.
// (The counter form above costs fc_out's last kernel a fence in every wave, a barrier, an atomic round trip and a second fence:
// 3.8 us from its last MFMA to its exit, profiles/r06_b1_stamps_fc_out.txt.)
__device__ __forceinline__ void signal_done_by_wave(const DoneSignal& d, unsigned wg)
{
    if (!d.host_flag) return;                        // launch-uniform
    __threadfence_system();                          // this wave's results are on their way to the host, in front of the flag
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(d.host_flag + wg, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
// The same guard for values that are split WITHOUT passing store_split4: the raw context a first convolution reads (FirstConv
// below splits its taps in registers).  A finite context element of 1e5 would give hi = inf, lo = -inf and a NaN out of the MFMA
// chain -- which amax4 / fmaxf then DROP, so the output-side guard alone stays silent.  Checked where the plane is staged into
// LDS: one compare per input element (a 16x16-net image: 1280 elements against ~10 k tap reads), NaN and infinity included.
__device__ __forceinline__ bool leaves_f16(float v) { return !(fabsf(v) < kF16Max); }

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
// The same store for a value that IS a matrix-instruction result (the K-segment planes of tapgemm_f32_small_kernel): the wait states
// between the MFMA's register write and the vector-memory read are inside the asm block -- 16 of them, what an 8-pass instruction
// (v_mfma_f32_16x16x4_f32) asks for with room to spare -- instead of resting on where the register allocator put the accumulator
// (in AGPRs the compiler's own v_accvgpr_read sits in between; in VGPRs nothing would: ADVICE r5).
__device__ __forceinline__ void store16_through_mfma(f32x4* dst, f32x4 v)
{
#ifdef PNN_PLAIN_STORES
    *dst = v;
#else
    asm volatile("s_nop 7\n\ts_nop 7\n\tglobal_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#endif
}

// A lane's four consecutive output channels 16 G + 4 q + r of one pixel (group = the pixel's 16-channel group G), in channel order or in
// CHAIN ORDER (tapgemm_f32_small_body, XCH: within a group channel kk sits at 4 g + i, g = (kk >> 3) + 2 (kk & 1), i = (kk & 7) >> 1 --
// the order the 16x16x4 chain's lane groups consume): channels 4q + {0, 2} are neighbours there, and so are 4q + {1, 3}, eight floats on.
__device__ __forceinline__ void store4_chain(float* group, int q, f32x4 v, bool chain)
{
    if (!chain) { *reinterpret_cast<f32x4*>(group + 4 * q) = v; return; }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    float* d = group + 4 * (q >> 1) + 2 * (q & 1);
    *reinterpret_cast<f32x2*>(d) = (f32x2){v[0], v[2]};
    *reinterpret_cast<f32x2*>(d + 8) = (f32x2){v[1], v[3]};
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

// ---------------------------------------------------------------------------------------------------------------------
// First convolution of a branch (Cin = 1, k x k taps, pnn/tfutils.py:75-139) on the split-precision path: a contraction
// over the TAPS on the matrix cores.  out[pixel][co] = sum_t x[pixel, tap t] * W[t][co], t = ky * k + kx, K = k * k taps
// zero-padded to a multiple of 16 (3 x 3: one 16-deep chunk, 5 x 5: two), operands split into f16 hi / lo halves and
// multiplied as w_hi * x_hi, w_hi * x_lo, w_lo * x_hi per chunk -- the order of every split-precision GEMM kernel, so this IS
// "the tap GEMM over the im2col rows of the context", computed without materialising them.  Used by conv_cin1_kernel /
// conv_cin1_pair_kernel (split output) and by convimg_sp_kernel's fused first convolution: one definition, one summation
// order at every batch size.  (Until round 3 the first convolution was 25 dependent v_fma_f32 per output on the VALU:
// 12 k SIMD-cycles per 16x16-net image, as much as the matrix work of the layer it feeds.)
// Weights: the layer's [K][Cout] matrix in the split pack of the GEMM layers, [chunk][hi / lo][k-half][Npad][8 x f16],
// pre-scaled by a power of two that the caller undoes (out_scale).
// ---------------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K>
struct FirstConv {
    static constexpr int NCH = (K * K + 15) / 16;
    f16x8 xhi[NCH], xlo[NCH];
    // Per-lane tap offsets into the zero-padded plane (row pitch PW): lane half h supplies taps 16 c + 8 h + j of chunk c.  Set once
    // per kernel; taps past K * K (the K padding of the last chunk) point at the pixel's own first tap -- their WEIGHTS are zero, so
    // any finite value will do, and the read needs no select.  (Until round 4 every load() recomputed the offsets and masked the
    // values: ~100 of the fused first convolution's ~280 VALU instructions per 32-pixel tile.)
    int off[NCH][8];
    __device__ __forceinline__ void setup(int PW, int h)
    {
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int t0 = 16 * c + j, t1 = t0 + 8;             // the tap of k-half 0 / 1
                const int o0 = t0 < K * K ? (t0 / K) * PW + t0 % K : 0, o1 = t1 < K * K ? (t1 / K) * PW + t1 % K : 0;
                off[c][j] = h ? o1 : o0;
            }
    }
    // This lane's operand: 8 taps (k-half h = lane >> 5) of each chunk for ITS pixel.  xr = the pixel's top-left tap in the
    // zero-padded f32 plane (LDS or global); an idle row of the tile passes any readable pixel (its outputs are not stored).
    __device__ __forceinline__ void load(const float* xr)
    {
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            f32x4 v[2];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j >> 2][j & 3] = xr[off[c][j]];
            f16x4 h0, l0, h1, l1;
            split4(v[0], h0, l0);
            split4(v[1], h1, l1);
            xhi[c] = (f16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            xlo[c] = (f16x8){l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
    }
    // Weight fragments of one 32-channel column tile: lane's channel n = tile * 32 + (lane & 31), k-half h.
    struct W { f32x4 hi[NCH], lo[NCH]; };
    static __device__ __forceinline__ W weights(const f32x4* __restrict__ wsp, int npad, int n, int h)
    {
        W w;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            w.hi[c] = wsp[((c * 2 + 0) * 2 + h) * npad + n];
            w.lo[c] = wsp[((c * 2 + 1) * 2 + h) * npad + n];
        }
        return w;
    }
    // 32 pixels x 32 channels: acc[4 g + e] = channel tile * 32 + 8 g + 4 h + e of pixel (lane & 31), before scale / bias.
    __device__ __forceinline__ f32x16 tile(const W& w) const
    {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const f16x8 whi = __builtin_bit_cast(f16x8, w.hi[c]), wlo = __builtin_bit_cast(f16x8, w.lo[c]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xhi[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, xlo[c], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, xhi[c], acc, 0, 0, 0);
        }
        return acc;
    }
};

__device__ __forceinline__ void store_split1(void* base, size_t pixel_times_c, int n, float v, float& amax)
{
    amax = fmaxf(amax, fabsf(v));
    const _Float16 hi = (_Float16)v;
    _Float16* dst = reinterpret_cast<_Float16*>(base) + 2 * pixel_times_c + (n >> 4) * 32 + (n & 15);
    dst[0] = hi;
    dst[16] = (_Float16)(v - (float)hi);
}

}  // namespace pnn
