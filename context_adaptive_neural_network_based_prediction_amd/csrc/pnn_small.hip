// The non-GEMM kernels of the PNN forward pass (gfx950): the L-shaped context gather, the Cin = 1 first
// convolution of each branch, the channel-wise fully-connected merger, the Cout = 1 last transposed
// convolution (with the HM epilogue fused) and the stand-alone epilogue.  All are bandwidth-side work: the
// designs below stage what is re-read in LDS so that global memory sees each activation once.
//
// Reference semantics: extraction_context.cpp:3-208 (gather), pnn/tfutils.py:75-139 (conv, SAME),
// pnn/tfutils.py:8-73 + components.py:231-237 (merger), pnn/tfutils.py:395-462 (transposed conv, SAME),
// TComPrediction.cpp:621-635 (epilogue).
#include <algorithm>
#include "pnn_kernels.h"
#include <time.h>
#include "pnn_device_common.h"
#include "pnn_small_bodies.h"

namespace pnn {

// ------------------------------------------------------------------------------------------------
// Cin == 1 forward convolution + bias + LeakyReLU (k = 3, stride 1 or k = 5, stride 2).
// One workgroup per image (or band of output rows): the zero-padded input plane sits in LDS (SAME padding becomes plain
// indexing).  f32 output (exact-f32 path): each lane keeps its 4 output channels' k*k weights in registers, Cout/4 lanes
// share a pixel so a wave stores 1 KiB (Cout 64) of contiguous NHWC output per instruction.  Split output (the default
// arithmetic): the contraction over the taps runs on the matrix cores, see FirstConv.  HBM-bound on the output write.
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) Conv1Params CConv1;   // read in place in the kernel-argument segment: scalar loads
// Stages rows [oy0 * s - pad, ... + PH) of block b's zero-padded input plane (PH x PW floats) in LDS; returns whether a value would leave
// the f16 range (split path: leaves_f16, pnn_device_common.h).
__device__ __forceinline__ bool conv_cin1_stage(CConv1& p, const long b, const int oy0, const int PH, const int PW, float* xs)
{
    bool in_bad = false;
    if (!p.X) {
        // the gather fused in: straight from the picture plane through this block's TB descriptor
        const TbDev d = p.tbs[b];
        const int w0 = p.w;
        for (int idx = threadIdx.x; idx < PH * PW; idx += 256) {
            const int r = idx / PW, c = idx - r * PW;
            const int iy = oy0 * p.s + r - p.pad, ix = c - p.pad;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
                bool ok;
                long off;
                if (p.branch == 0) {
                    ok = ix < w0 || ((d.above_mask >> ((ix - w0) / p.unit)) & 1u);
                    off = d.origin + (long)(iy - w0) * d.stride + (ix - w0);
                } else {
                    ok = iy < d.left_units * p.unit;
                    off = d.origin + (long)iy * d.stride + (ix - w0);
                }
                if (ok) v = (p.pel_bytes == 4 ? (float)reinterpret_cast<const int32_t*>(p.plane)[off] : (float)reinterpret_cast<const uint8_t*>(p.plane)[off]) - p.mean;
            }
            in_bad |= leaves_f16(v);
            xs[idx] = v;
        }
    } else {
        const float* xb = p.X + b * p.IH * p.IW;
        for (int idx = threadIdx.x; idx < PH * PW; idx += 256) {
            const int r = idx / PW, c = idx - r * PW;
            const int iy = oy0 * p.s + r - p.pad, ix = c - p.pad;
            const float v = ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) ? xb[iy * p.IW + ix] : 0.f;
            in_bad |= leaves_f16(v);
            xs[idx] = v;
        }
    }
    return in_bad;
}

template <int K>
__device__ __forceinline__ void conv_cin1_body(CConv1& p, const int bx, const int by)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    // blockIdx.y = band of p.band_rows output rows (small batches: one image is spread over several workgroups; a
    // single workgroup per image took 72 us for a 64x192 portion at batch 1)
    const int oy0 = by * p.band_rows;
    const int oy1 = oy0 + p.band_rows < p.OH ? oy0 + p.band_rows : p.OH;
    const int PH = (oy1 - oy0 - 1) * p.s + K, PW = (p.OW - 1) * p.s + K;
    const long b = bx;
    const bool in_bad = conv_cin1_stage(p, b, oy0, PH, PW, xs);
    if (p.split && in_bad && p.range_flag) *p.range_flag = 1;
    const int npix = p.OH * p.OW;
    if (p.split) {
        // Split-precision path: the contraction over the taps on the matrix cores (FirstConv, pnn_device_common.h).  A wave
        // takes 32 consecutive pixels of the band at a time: operand from the staged plane, one MFMA chain per 32-channel
        // column tile, scale / bias / LeakyReLU / split, then through a wave-private LDS tile so that the 32 pixels leave as
        // ONE contiguous run of 16-byte pieces (the accumulator layout owns 8-byte fragments of 32 different pixels).
        __syncthreads();
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
        const int cq = p.Cout >> 2, PITCH = cq + 1, nct = p.Cout >> 5;
        f32x4* stage = reinterpret_cast<f32x4*>(xs + ((PH * PW + 3) & ~3)) + wave * 32 * PITCH;
        const f32x4* wsp = reinterpret_cast<const f32x4*>(p.Wsp);
        typename FirstConv<K>::W w[2];
        f32x4 bv[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
            if (ct < nct) w[ct] = FirstConv<K>::weights(wsp, p.npad, ct * 32 + l31, h);
#pragma unroll
            for (int g = 0; g < 4; g++) bv[ct][g] = *reinterpret_cast<const f32x4*>(p.bias + (ct < nct ? ct * 32 + 8 * g + 4 * h : 0));
        }
        const int band_pix = (oy1 - oy0) * p.OW;
        float amax = 0.f;                            // range guard of the split output (pnn_device_common.h)
        f32x4* yo = reinterpret_cast<f32x4*>(p.Y) + ((size_t)b * npix + (size_t)oy0 * p.OW) * cq;
        FirstConv<K> fc;
        fc.setup(PW, h);
        for (int rt = wave; rt * 32 < band_pix; rt += 4) {
            const int lp = rt * 32 + l31;
            const bool valid = lp < band_pix;
            const int ly = valid ? lp / p.OW : 0, lx = valid ? lp - ly * p.OW : 0;
            fc.load(xs + (ly * p.s) * PW + lx * p.s);      // (an idle row reads pixel 0: its outputs are not stored)
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                if (ct >= nct) break;
                const f32x16 acc = fc.tile(w[ct]);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]} * p.out_scale + bv[ct][g];
                    v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                    if (valid) amax = amax4(amax, v);
                    f16x4 hi, lo;
                    split4(v, hi, lo);
                    const int n = ct * 32 + 8 * g + 4 * h;
                    _Float16* dst = reinterpret_cast<_Float16*>(stage + l31 * PITCH) + (n >> 4) * 32 + (n & 15);
                    *reinterpret_cast<f16x4*>(dst) = hi;
                    *reinterpret_cast<f16x4*>(dst + 16) = lo;
                }
            }
            __builtin_amdgcn_wave_barrier();         // the tile is this wave's own: LDS keeps a wave's accesses in order
            const int rows = band_pix - rt * 32 < 32 ? band_pix - rt * 32 : 32;
            for (int i = lane; i < rows * cq; i += 64) {
                const int row = i / cq, q = i - row * cq;
                yo[(size_t)(rt * 32) * cq + i] = stage[row * PITCH + q];
            }
            __builtin_amdgcn_wave_barrier();
        }
        report_range(p.range_flag, amax);
        return;
    }
    // Exact f32 (round 6): the contraction over the taps on the f32 matrix instruction.  Per output it is the SAME chain as the VALU
    // form this replaces -- acc = bias; acc = fma(x[tap], w[tap], acc) for tap = 0 .. K^2 - 1 in (ky, kx) order: the matrix instruction's
    // addend is the bias, its k-steps are the taps in ascending order (step i: tap 2i from lane half 0, then 2i + 1 from half 1; the odd
    // one out multiplies a zero weight: fma(x, 0, acc) = acc) -- at 256 instead of 64 multiply-adds per cycle and CU (packed-fp32 VALU
    // code is not built here: Makefile).  Same bits (tools/lib_ab_bits.py), so the launcher picks: this form for small passes, where
    // one workgroup's arithmetic is on the call's critical path (single-block calls of the 16x16 / 32x32 / 64x64 nets: -2 us each); at
    // the bench batches the layer is bound by its output write and this form is 0-1 % SLOWER (profiles/r06_cin1_mfma.txt).  A wave
    // takes 32 consecutive pixels of the band at a time and leaves them through a wave-private LDS tile as ONE contiguous run (the
    // accumulator layout owns 16-byte fragments of 32 different pixels).
    __syncthreads();
    if (p.mfma) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
        constexpr int NK = (K * K + 1) / 2;
        const int cq = p.Cout >> 2, PITCH = cq + 1, nct = p.Cout >> 5;
        f32x4* stage = reinterpret_cast<f32x4*>(xs + ((PH * PW + 3) & ~3)) + wave * 32 * PITCH;
        float wv[2][NK];
        int off[NK];
#pragma unroll
        for (int i = 0; i < NK; i++) {
            const int t = 2 * i + h;
            off[i] = t < K * K ? (t / K) * PW + t % K : 0;
#pragma unroll
            for (int ct = 0; ct < 2; ct++) wv[ct][i] = (t < K * K && ct < nct) ? p.W[(size_t)t * p.Cout + ct * 32 + l31] : 0.f;
        }
        f32x16 bacc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + (ct < nct ? ct * 32 + 8 * g + 4 * h : 0));
                bacc[ct][4 * g] = bv[0]; bacc[ct][4 * g + 1] = bv[1]; bacc[ct][4 * g + 2] = bv[2]; bacc[ct][4 * g + 3] = bv[3];
            }
        const int band_pix = (oy1 - oy0) * p.OW;
        f32x4* yo = reinterpret_cast<f32x4*>(p.Y) + ((size_t)b * npix + (size_t)oy0 * p.OW) * cq;
        const bool chain = p.chain != 0;
        for (int rt = wave; rt * 32 < band_pix; rt += 4) {
            const int lp = rt * 32 + l31;
            const bool valid = lp < band_pix;
            const int ly = valid ? lp / p.OW : 0, lx = valid ? lp - ly * p.OW : 0;
            const float* xr = xs + (ly * p.s) * PW + lx * p.s;       // (an idle row reads pixel 0: its outputs are not stored)
            float xv[NK];
#pragma unroll
            for (int i = 0; i < NK; i++) xv[i] = xr[off[i]];
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                if (ct >= nct) break;
                f32x16 acc = bacc[ct];
#pragma unroll
                for (int i = 0; i < NK; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[ct][i], xv[i], acc, 0, 0, 0);
                // lane (pixel l31, half h): register 4g + r = channel 32 ct + 8g + 4h + r
                float* row = reinterpret_cast<float*>(stage + l31 * PITCH);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                    const int c0 = ct * 32 + 8 * g + 4 * h;
                    store4_chain(row + (c0 & ~15), (c0 & 15) >> 2, v, chain);
                }
            }
            __builtin_amdgcn_wave_barrier();             // the tile is this wave's own: LDS keeps a wave's accesses in order
            const int rows = band_pix - rt * 32 < 32 ? band_pix - rt * 32 : 32;
            for (int i = lane; i < rows * cq; i += 64) {
                const int row = i / cq, q = i - row * cq;
                yo[(size_t)(rt * 32) * cq + i] = stage[row * PITCH + q];
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    const int CG = p.Cout >> 2;                       // lanes per pixel (8 or 16)
    const int cg = threadIdx.x % CG, psub = threadIdx.x / CG, ppi = 256 / CG;
    f32x4 w[K * K];
#pragma unroll
    for (int t = 0; t < K * K; t++) w[t] = *reinterpret_cast<const f32x4*>(p.W + (size_t)t * p.Cout + 4 * cg);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + 4 * cg);
    float* yb = p.Y + b * npix * p.Cout;
    for (int pix = oy0 * p.OW + psub; pix < oy1 * p.OW; pix += ppi) {
        const int oy = pix / p.OW, ox = pix - oy * p.OW;
        const float* xr = xs + ((oy - oy0) * p.s) * PW + ox * p.s;
        f32x4 acc = bv;
#pragma unroll
        for (int ky = 0; ky < K; ky++)
#pragma unroll
            for (int kx = 0; kx < K; kx++) acc += xr[ky * PW + kx] * w[ky * K + kx];
        acc[0] = leaky(acc[0]); acc[1] = leaky(acc[1]); acc[2] = leaky(acc[2]); acc[3] = leaky(acc[3]);
        store4_chain(yb + (size_t)pix * p.Cout + ((4 * cg) & ~15), cg & 3, acc, p.chain != 0);
    }
}

template <int K>
__global__ __launch_bounds__(256) void conv_cin1_kernel(const Conv1Params p)
{
    touch_kernargs<sizeof(Conv1Params)>();
    (void)p;
    conv_cin1_body<K>(*(CConv1*)__builtin_amdgcn_kernarg_segment_ptr(), blockIdx.x, blockIdx.y);
}

// Both branches' first convolutions in ONE launch (blockIdx.z = branch): at small batch every launch costs ~4 us whatever it
// does, and the two branches do not depend on each other.
struct Conv1Pair { Conv1Params a, b; };
template <int K>
__global__ __launch_bounds__(256) void conv_cin1_pair_kernel(const Conv1Pair pr)
{
    touch_kernargs<sizeof(Conv1Pair)>();
    (void)pr;
    const auto* k = (const __attribute__((address_space(4))) Conv1Pair*)__builtin_amdgcn_kernarg_segment_ptr();
    CConv1* p = blockIdx.z ? &k->b : &k->a;
    if ((int)blockIdx.y * p->band_rows >= p->OH) return;          // the other branch has more row bands
    conv_cin1_body<K>(*p, blockIdx.x, blockIdx.y);
}

static bool conv_cin1_bands(const Conv1Params& p, Conv1Params* q, int* bands_out, size_t* lds_out)
{
    *q = p;
    int bands = 1;                                    // enough workgroups to fill the chip at small batch
    while ((long)p.B * bands < 512 && bands * 2 <= p.OH) bands *= 2;
    q->band_rows = (p.OH + bands - 1) / bands;
    *bands_out = (p.OH + q->band_rows - 1) / q->band_rows;
    // staged plane (+ the split path's four wave-private output tiles of 32 pixels x (Cout / 4 + 1) 16-byte slots)
    auto lds_of = [&](int rows) {
        return ((size_t)((rows - 1) * p.s + p.k) * ((p.OW - 1) * p.s + p.k) + 3) / 4 * 16 + (size_t)4 * 32 * (p.Cout / 4 + 1) * 16;
    };
    while (lds_of(q->band_rows) > 64 * 1024 && q->band_rows > 1) q->band_rows = (q->band_rows + 1) / 2;
    *bands_out = (p.OH + q->band_rows - 1) / q->band_rows;
    *lds_out = lds_of(q->band_rows);
    q->mfma = !p.split && p.B < 256;                  // f32 output: the matrix-instruction form of the same chain for small passes (conv_cin1_body)
    return *lds_out <= 64 * 1024 && (p.Cout == 32 || p.Cout == 64) && (p.k == 3 || p.k == 5) && (!p.split || p.Wsp);
}

hipError_t launch_conv_cin1(const Conv1Params& p, hipStream_t s)
{
    if (p.B <= 0) return hipSuccess;
    Conv1Params q;
    int bands;
    size_t lds;
    if (!conv_cin1_bands(p, &q, &bands, &lds)) return hipErrorInvalidValue;
    if (p.k == 3) hipLaunchKernelGGL(conv_cin1_kernel<3>, dim3(p.B, bands), dim3(256), lds, s, q);
    else hipLaunchKernelGGL(conv_cin1_kernel<5>, dim3(p.B, bands), dim3(256), lds, s, q);
    return hipGetLastError();
}

hipError_t launch_conv_cin1_pair(const Conv1Params& a, const Conv1Params& b, hipStream_t s)
{
    if (a.B <= 0) return hipSuccess;
    Conv1Pair pr;
    int ba, bb;
    size_t la, lb;
    if (a.B != b.B || a.k != b.k || !conv_cin1_bands(a, &pr.a, &ba, &la) || !conv_cin1_bands(b, &pr.b, &bb, &lb)) return hipErrorInvalidValue;
    const dim3 grid(a.B, ba > bb ? ba : bb, 2);
    const size_t lds = la > lb ? la : lb;
    if (a.k == 3) hipLaunchKernelGGL(conv_cin1_pair_kernel<3>, grid, dim3(256), lds, s, pr);
    else hipLaunchKernelGGL(conv_cin1_pair_kernel<5>, grid, dim3(256), lds, s, pr);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Cout == 1 transposed convolution + bias (last merger layer, linear), optional HM epilogue.
// One workgroup per 16x16 output tile of one image.  The input tile (with halo, zeros outside the image)
// is staged in LDS channel-chunk-major -- [Cin/4][pixel] float4 -- so that the 64 lanes of a wave, which
// read 64 different pixels at the same channel chunk, hit consecutive 16-byte slots.  For stride 2 each
// wave owns one output-parity class (py, px): its tap set is wave-uniform, so the weights are scalar
// loads and there is no divergence.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tconv_cout1_kernel(const TConv1Params p)
{
    touch_kernargs<sizeof(TConv1Params)>();
    extern __shared__ __attribute__((aligned(16))) f32x4 xt[];       // [image in WG][Cin/4][TI*TI]
    tconv_cout1_lds_tile<false>(p, xt, blockIdx.x, blockIdx.y);        // pnn_small_bodies.h
    signal_done(p.done);
}

// The same layer for Cin == 64 (every reference net ends in it: k = 5, s = 2, 64 -> 1) on the fp32 matrix cores.
// Phase 1 is a GEMM: T[input pixel][tap] = sum_c x[pixel][c] * w[tap][c] with v_mfma_f32_32x32x2_f32 (M = 32 pixels per
// wave tile, N = 25 taps padded to 32, K = 64 in 32 steps), T kept in LDS.  Phase 2 is the col2im: every output pixel
// adds the <= 9 entries of T its parity class reaches.  Workgroup = one band of full-width input rows of one block, so
// the pixel rows of a tile are contiguous in memory.  The LDS version above reads every x value from LDS once per tap
// (1.6 KB of ds_read per output, 26 us for the 16x16 net at batch 1024); here x goes global -> registers -> MFMA once.
// The K order inside the MFMA is a fixed permutation of the channels (lane half h, step 4q+i <-> channel 8q+4h+i), the
// same for every batch size.
template <int s, int K>                              // compile-time stride and kernel size: the parity tests of phase 2 fold
__global__ __launch_bounds__(256) void tconv_cout1_mfma_kernel(const TConv1Params p)
{
    touch_kernargs<sizeof(TConv1Params)>();
    extern __shared__ __attribute__((aligned(16))) float T[];        // [pixels of the band, padded to 32][kTcTP]
    tconv_cout1_mfma_band<s, K, false>(p, T, blockIdx.x, blockIdx.y);   // pnn_small_bodies.h
    signal_done(p.done);
}

hipError_t launch_tconv_cout1(const TConv1Params& pin, hipStream_t s)
{
    TConv1Params p = pin;
    if (p.B <= 0) return hipSuccess;
    if (p.Cin == 64 && ((p.s == 2 && p.k == 5) || (p.s == 1 && p.k == 3)) && p.pad <= p.k - 1) {
        // band = as many output rows as keep T <= ~42 KB (320 input pixels): whole image up to 16x16 inputs
        const int OH = p.IH * p.s;
        int toh = OH;
        auto band_px = [&](int t) { return ((t - 1 + p.pad) / p.s + (p.k - 1 - p.pad + p.s - 1) / p.s + 1) * p.IW; };
        while (toh > p.s && band_px(toh) > 320 && toh % 2 == 0) toh /= 2;
        const int npx = std::min(band_px(toh), p.IH * p.IW);
        const size_t lds = (size_t)((npx + 31) / 32) * 32 * kTcTP * sizeof(float);
        if (lds <= 64 * 1024) {
            p.ni = toh;
            const dim3 grid(p.B, (OH + toh - 1) / toh);
            if (p.s == 2) hipLaunchKernelGGL((tconv_cout1_mfma_kernel<2, 5>), grid, dim3(256), lds, s, p);
            else hipLaunchKernelGGL((tconv_cout1_mfma_kernel<1, 3>), grid, dim3(256), lds, s, p);
            return hipGetLastError();
        }
    }
    if ((p.s != 1 && p.s != 2) || p.Cin % 4 || p.IH != p.IW) return hipErrorInvalidValue;
    const int OH = p.IH * p.s;
    const int TO = OH < 16 ? OH : 16;
    if (p.s == 2 && (TO & 1)) return hipErrorInvalidValue;
    const int lo = p.pad - (p.k - 1);
    const int i0 = lo >= 0 ? lo / p.s : -((-lo + p.s - 1) / p.s);
    const int TI = (TO - 1 + p.pad) / p.s - i0 + 1;
    const size_t per_img = (size_t)TI * TI * p.Cin * sizeof(float);
    int ni = p.s == 2 ? 64 / ((TO / 2) * (TO / 2)) : 256 / (TO * TO);
    const size_t wbytes = (size_t)p.k * p.k * p.Cin * sizeof(float);          // the weights share the workgroup's LDS
    while (ni > 1 && ni * per_img + wbytes > 60 * 1024) ni >>= 1;
    if (ni < 1 || per_img + wbytes > 64 * 1024) return hipErrorInvalidValue;
    p.ni = ni;
    const int tiles = ((OH + TO - 1) / TO) * ((OH + TO - 1) / TO);
    dim3 grid((p.B + ni - 1) / ni, tiles);
    hipLaunchKernelGGL(tconv_cout1_kernel, grid, dim3(256), ni * per_img + wbytes, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Channel-wise fully-connected merger + LeakyReLU.  Per channel c: out[b][j][c] = leaky(sum_p v[b][p][c] *
// W[c][p][j] + bias[c][j]) with v = [above 4x12 | left 8x4] (80 values).  Thread = (channel c, 4 outputs j,
// MB blocks); lanes run over c, the innermost NHWC index, so every load and store is coalesced; W is
// pre-arranged as [p][j][c].
// ------------------------------------------------------------------------------------------------
template <int MB, int J>
__global__ __launch_bounds__(256) void merger_kernel(const MergerParams p)
{
    touch_kernargs<sizeof(MergerParams)>();
    constexpr int NJ = 16 / J;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(gid % p.C);
    const long r = gid / p.C;
    const int jq = (int)(r % NJ);
    const long b0 = (r / NJ) * MB;
    if (b0 >= p.B) return;
    float acc[MB][J];
#pragma unroll
    for (int m = 0; m < MB; m++)
#pragma unroll
        for (int j = 0; j < J; j++) acc[m][j] = 0.f;
    long brow[MB];
#pragma unroll
    for (int m = 0; m < MB; m++) brow[m] = (b0 + m < p.B) ? b0 + m : b0;
    // two plain loops (above part, then left part) instead of a per-iteration select between the sources
    auto part = [&](const float* src, int np, int pbase) {
#pragma unroll 4
        for (int pp = 0; pp < np; pp++) {
            float xv[MB];
#pragma unroll
            for (int m = 0; m < MB; m++) xv[m] = src[(brow[m] * np + pp) * p.C + c];
#pragma unroll
            for (int j = 0; j < J; j++) {
                const float wv = p.Wp[((size_t)(pbase + pp) * 16 + J * jq + j) * p.C + c];
#pragma unroll
                for (int m = 0; m < MB; m++) acc[m][j] += xv[m] * wv;
            }
        }
    };
    part(p.A, p.na, 0);
    part(p.L, p.nl, p.na);
    float amax = 0.f;                                // range guard of the split output (pnn_device_common.h)
#pragma unroll
    for (int j = 0; j < J; j++) {
        const float bv = p.bias[(size_t)(J * jq + j) * p.C + c];
#pragma unroll
        for (int m = 0; m < MB; m++)
            if (b0 + m < p.B) {
                const float v = leaky(acc[m][j] + bv);
                if (p.split) store_split1(p.Y, ((size_t)(b0 + m) * 16 + J * jq + j) * p.C, c, v, amax);
                else p.Y[((b0 + m) * 16 + J * jq + j) * p.C + c] = v;
            }
    }
    if (p.split) report_range(p.range_flag, amax);
}

// The batch kernel on the fp32 matrix cores.  Per channel the merger is a [B x 80] x [80 x 16] product; with the channel
// innermost in memory one tile is 16 blocks x 16 CHANNELS: lane (block bi = lane % 16, position slot lane / 16) loads
// the 64 contiguous bytes x[b][p][c0 .. c0 + 15] and w[p][j = lane % 16][c0 .. c0 + 15], and channel i's 16x16x4 MFMA
// takes element i of both -- 8 x 16-byte loads feed 16 MFMAs per 4 positions.
// The layer is bound by memory LATENCY, not by arithmetic or L1 traffic (the scalar kernel above and a one-wave-per-tile
// MFMA version both took 29-31 us for the 16x16 net at batch 1024, 42 MB of input: ten dependent round trips with only
// ~4 MB in flight).  So the 20 position steps of a tile are split over the 4 waves of a workgroup, every wave issues
// ALL its loads (5 steps x 8) before the first MFMA, and the partial sums meet in LDS in a fixed order.
// Workgroups i, i + 8, ... run on one XCD (round-robin dispatch): they get the channel groups of the same blocks, so an
// XCD's L2 fetches whole rows instead of 64-byte pieces of everyone's.  Measured: ~3.6 TB/s + ~5 us fixed per launch
// (8x8 net at batch 4096: 101 MB in 31.7 us).  Two channel groups per 8-wave workgroup (whole 128-byte lines per CU, but
// 96 KB of LDS = one workgroup per CU) was slower: 34.8 / 22.3 / 18.2 us against 31.7 / 19.4 / 12.1 us (8x8 / 16x16 / 32x32 nets).
template <bool SPLIT>
__global__ __launch_bounds__(256) void merger_mfma_kernel(const MergerParams p)
{
    touch_kernargs<sizeof(MergerParams)>();
    __shared__ f32x4 red[3][16][64];
    const int ngc = p.C >> 4;
    const long rest = blockIdx.x >> 3;
    const int c0 = (int)(rest % ngc) * 16;
    const long bg = (rest / ngc) * 8 + (blockIdx.x & 7);
    if (bg * 16 >= p.B) return;
    merger_mfma_tile<SPLIT, false>(p, red, c0, bg * 16);               // pnn_small_bodies.h
}

hipError_t launch_merger(const MergerParams& p, hipStream_t s)
{
    if (p.B <= 0) return hipSuccess;
    if (p.nout != 16) return hipErrorInvalidValue;
    if (p.C % 16 == 0 && p.na % 4 == 0 && p.na + p.nl == 80) {
        const long bgs = ((p.B + 15) / 16 + 7) / 8 * 8;       // block groups, padded to the 8-way XCD interleave
        const dim3 grid((unsigned)(bgs * (p.C / 16))), block(256);
        if (p.split) hipLaunchKernelGGL(merger_mfma_kernel<true>, grid, block, 0, s, p);
        else hipLaunchKernelGGL(merger_mfma_kernel<false>, grid, block, 0, s, p);
        return hipGetLastError();
    }
    const long threads = (long)((p.B + 1) / 2) * 4 * p.C;
    hipLaunchKernelGGL((merger_kernel<2, 4>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// L-shaped context gather: Pel (int32 or uint8) -> float, minus mean, unavailable units -> 0.
// Equivalent to extraction_context.cpp:3-208 for every flag pattern: the above portion is masked per
// unit; the left portion holds the first 4*left_units source rows (the reference advances source and
// destination only on available units, extraction_context.cpp:189-205).
// ------------------------------------------------------------------------------------------------
template <typename Pel>
__global__ __launch_bounds__(256) void gather_kernel(const GatherParams p)
{
    touch_kernargs<sizeof(GatherParams)>();
    const int w = p.w;
    const int na = 3 * w * w, per = 5 * w * w;
    const long total = (long)p.N * per;
    const Pel* plane = reinterpret_cast<const Pel*>(p.plane);
    float unused = 0.f;                              // 8-bit samples minus the mean never leave the f16 range
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long tb = e / per;
        const int r = (int)(e - tb * per);
        const TbDev d = p.tbs[tb];
        if (r < na) {
            const int row = r / (3 * w), col = r - row * 3 * w;
            bool ok = true;
            if (col >= w) ok = (d.above_mask >> ((col - w) / p.unit)) & 1u;
            float v = 0.f;
            if (ok) v = (float)plane[d.origin + (long)(row - w) * d.stride + (col - w)] - p.mean;
            if (p.split) store_split1(p.above, (size_t)tb * per, r, v, unused);
            else p.above[tb * p.pitch_above + r] = v;
        } else {
            const int rl = r - na;
            const int row = rl / w, col = rl - row * w;
            const bool ok = row < d.left_units * p.unit;
            float v = 0.f;
            if (ok) v = (float)plane[d.origin + (long)row * d.stride + (col - w)] - p.mean;
            if (p.split) store_split1(p.above, (size_t)tb * per, r, v, unused);  // FC row: left part follows the above part
            else p.left[tb * p.pitch_left + rl] = v;
        }
    }
}

// FC rows in the split layout, four consecutive elements per thread (they share a context row: w and 3w are multiples
// of 4): one 8-byte store of the hi halves and one of the lo halves instead of eight 2-byte stores, constant divisors.
template <typename Pel, int W>
__global__ __launch_bounds__(256) void gather_split4_kernel(const GatherParams p)
{
    touch_kernargs<sizeof(GatherParams)>();
    constexpr int NA = 3 * W * W, PER = 5 * W * W, Q = PER / 4;
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    if (gid >= (unsigned)p.N * Q) return;
    const unsigned tb = gid / Q;
    const int r = (int)(gid - tb * Q) * 4;
    const TbDev d = p.tbs[tb];
    const Pel* plane = reinterpret_cast<const Pel*>(p.plane) + d.origin;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < NA) {
        const int row = r / (3 * W), col = r - row * (3 * W);
        const bool ok = col < W || ((d.above_mask >> ((col - W) / p.unit)) & 1u);   // a 4-pixel unit never straddles the 4 elements
        if (ok) {
            const Pel* src = plane + (long)(row - W) * d.stride + (col - W);
            v = (f32x4){(float)src[0] - p.mean, (float)src[1] - p.mean, (float)src[2] - p.mean, (float)src[3] - p.mean};
        }
    } else {
        const int rl = r - NA;
        const int row = rl / W, col = rl - row * W;
        if (row < d.left_units * p.unit) {
            const Pel* src = plane + (long)row * d.stride + (col - W);
            v = (f32x4){(float)src[0] - p.mean, (float)src[1] - p.mean, (float)src[2] - p.mean, (float)src[3] - p.mean};
        }
    }
    float unused = 0.f;                              // 8-bit samples minus the mean never leave the f16 range
    store_split4(p.above, (size_t)tb * PER, r, v, unused);
}

// The same for f32 outputs (above / left portions of the convolutional nets, or f32 FC rows): one 16-byte store per thread.
template <typename Pel>
__global__ __launch_bounds__(256) void gather_f32x4_kernel(const GatherParams p)
{
    touch_kernargs<sizeof(GatherParams)>();
    const int w = p.w, na = 3 * w * w, q = 5 * w * w / 4;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)p.N * q) return;
    const long tb = gid / q;
    const int r = (int)(gid - tb * q) * 4;
    const TbDev d = p.tbs[tb];
    const Pel* plane = reinterpret_cast<const Pel*>(p.plane) + d.origin;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < na) {
        const int row = r / (3 * w), col = r - row * (3 * w);
        const bool ok = col < w || ((d.above_mask >> ((col - w) / p.unit)) & 1u);
        if (ok) {
            const Pel* src = plane + (long)(row - w) * d.stride + (col - w);
            v = (f32x4){(float)src[0] - p.mean, (float)src[1] - p.mean, (float)src[2] - p.mean, (float)src[3] - p.mean};
        }
        *reinterpret_cast<f32x4*>(p.above + tb * p.pitch_above + r) = v;
    } else {
        const int rl = r - na;
        const int row = rl / w, col = rl - row * w;
        if (row < d.left_units * p.unit) {
            const Pel* src = plane + (long)row * d.stride + (col - w);
            v = (f32x4){(float)src[0] - p.mean, (float)src[1] - p.mean, (float)src[2] - p.mean, (float)src[3] - p.mean};
        }
        *reinterpret_cast<f32x4*>(p.left + tb * p.pitch_left + rl) = v;
    }
}

template <typename Pel>
static bool launch_gather_split4(const GatherParams& p, hipStream_t s)
{
    if (p.unit != 4) return false;
    const long threads = (long)p.N * (5 * p.w * p.w / 4);
    if (threads >= 0x7fffffffL) return false;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    switch (p.w) {
        case 4: hipLaunchKernelGGL((gather_split4_kernel<Pel, 4>), grid, block, 0, s, p); return true;
        case 8: hipLaunchKernelGGL((gather_split4_kernel<Pel, 8>), grid, block, 0, s, p); return true;
        case 16: hipLaunchKernelGGL((gather_split4_kernel<Pel, 16>), grid, block, 0, s, p); return true;
        default: return false;
    }
}

hipError_t launch_gather(const GatherParams& p, hipStream_t s)
{
    const long total = (long)p.N * 5 * p.w * p.w;
    if (total <= 0) return hipSuccess;
    if (p.split && (p.pel_bytes == 4 ? launch_gather_split4<int32_t>(p, s) : p.pel_bytes == 1 ? launch_gather_split4<uint8_t>(p, s) : false))
        return hipGetLastError();
    if (!p.split && p.unit == 4 && p.pitch_above % 4 == 0 && p.pitch_left % 4 == 0 && ((uintptr_t)p.above & 15) == 0 && ((uintptr_t)p.left & 15) == 0 &&
        (p.pel_bytes == 4 || p.pel_bytes == 1)) {
        const long threads = (long)p.N * (5 * p.w * p.w / 4);
        const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
        if (p.pel_bytes == 4) hipLaunchKernelGGL(gather_f32x4_kernel<int32_t>, grid, block, 0, s, p);
        else hipLaunchKernelGGL(gather_f32x4_kernel<uint8_t>, grid, block, 0, s, p);
        return hipGetLastError();
    }
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (p.pel_bytes == 4) hipLaunchKernelGGL(gather_kernel<int32_t>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else if (p.pel_bytes == 1) hipLaunchKernelGGL(gather_kernel<uint8_t>, dim3((unsigned)blocks), dim3(256), 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void epilogue_kernel(const float* pred, long n, float mean, int32_t* dst)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        dst[i] = hm_round(pred[i], mean);
}

hipError_t launch_epilogue(const float* pred, long n, float mean, int32_t* dst, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(epilogue_kernel, dim3((unsigned)blocks), dim3(256), 0, s, pred, n, mean, dst);
    return hipGetLastError();
}

// (f4) Distortion of N predicted blocks against the original picture, as HM's first intra pass computes it for a
// candidate mode (TEncSearch.cpp:2376-2389 -> TComRdCost::xGetHADs / xGetSAD, TComRdCost.cpp:1753-1824, 1549-1751):
// one thread per 8x8 (4x4 for 4-wide blocks) sub-block, Walsh-Hadamard transform in registers, integer adds only
// (the per-block sum over sub-blocks is an integer atomicAdd: order-independent, bit-exact).
template <int T>
__device__ __forceinline__ void wht_rows_cols(int (&d)[T * T])
{
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {            // rows, then columns
        const int es = pass == 0 ? 1 : T, vs = pass == 0 ? T : 1;
#pragma unroll
        for (int v = 0; v < T; v++)
#pragma unroll
            for (int len = 1; len < T; len <<= 1)
#pragma unroll
                for (int i = 0; i < T; i += len << 1)
#pragma unroll
                    for (int j = i; j < i + len; j++) {
                        const int a = d[v * vs + j * es], b = d[v * vs + (j + len) * es];
                        d[v * vs + j * es] = a + b;
                        d[v * vs + (j + len) * es] = a - b;
                    }
    }
}

template <typename Pel, int T>
__global__ __launch_bounds__(256) void block_cost_kernel(const BlockCostParams p)
{
    const int per = (p.w / T) * (p.w / T);
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)p.N * per) return;
    const int blk = (int)(gid / per), sb = (int)(gid - (long)blk * per);
    const int by = (sb / (p.w / T)) * T, bx = (sb % (p.w / T)) * T;
    const TbDev tb = p.tbs[blk];
    const Pel* org = reinterpret_cast<const Pel*>(p.org_plane) + tb.origin + (long)by * tb.stride + bx;
    const int32_t* cur = p.pred + ((size_t)blk * p.w + by) * p.w + bx;
    int d[T * T];
#pragma unroll
    for (int y = 0; y < T; y++)
#pragma unroll
        for (int x = 0; x < T; x++) d[y * T + x] = (int)org[(long)y * tb.stride + x] - cur[y * p.w + x];
    unsigned s = 0;
    if (p.hadamard) {
        wht_rows_cols<T>(d);
#pragma unroll
        for (int k = 0; k < T * T; k++) s += (unsigned)abs(d[k]);
        s = T == 8 ? (s + 2) >> 2 : (s + 1) >> 1;
    } else {
#pragma unroll
        for (int k = 0; k < T * T; k++) s += (unsigned)abs(d[k]);
    }
    atomicAdd(p.cost + blk, s);
}

hipError_t launch_block_cost(const BlockCostParams& p, hipStream_t s)
{
    if (p.N <= 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(p.cost, 0, (size_t)p.N * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    const int t = p.w >= 8 ? 8 : 4;
    const long threads = (long)p.N * (p.w / t) * (p.w / t);
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    if (p.pel_bytes == 4 && t == 8) hipLaunchKernelGGL((block_cost_kernel<int32_t, 8>), grid, block, 0, s, p);
    else if (p.pel_bytes == 4) hipLaunchKernelGGL((block_cost_kernel<int32_t, 4>), grid, block, 0, s, p);
    else if (p.pel_bytes == 1 && t == 8) hipLaunchKernelGGL((block_cost_kernel<uint8_t, 8>), grid, block, 0, s, p);
    else if (p.pel_bytes == 1) hipLaunchKernelGGL((block_cost_kernel<uint8_t, 4>), grid, block, 0, s, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---- which streams share a hardware queue? (pnn_streams_on_distinct_queues, pnn_abi.cpp) -------------------------------------------------
// The runtime deals its streams onto a few hardware queues (GPU_MAX_HW_QUEUES = 4); two streams on one queue run their kernels in
// submission order, so two width workers of the batching service whose streams share a queue wait for each other's whole calls.
// A `busy_us` kernel goes to stream a and an empty one behind it to stream b: b finishing only after a = one queue.
__global__ void queue_probe_busy_kernel(unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
__global__ void queue_probe_empty_kernel() {}

hipError_t probe_queue_shared(hipStream_t a, hipStream_t b, bool* shared)
{
    constexpr double kBusyUs = 150.0;
    // warm-up first: the streams' first submissions (lazy queue creation, code-object load) must not fall into the timed window -- a slow
    // first launch on `b` would read as "shared" 
    hipLaunchKernelGGL(queue_probe_empty_kernel, dim3(1), dim3(64), 0, a);
    hipLaunchKernelGGL(queue_probe_empty_kernel, dim3(1), dim3(64), 0, b);
    hipError_t w = hipStreamSynchronize(a);
    if (w == hipSuccess) w = hipStreamSynchronize(b);
    if (w != hipSuccess) return w;
    hipLaunchKernelGGL(queue_probe_busy_kernel, dim3(1), dim3(64), 0, a, (unsigned)(kBusyUs * 100.0));
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    hipLaunchKernelGGL(queue_probe_empty_kernel, dim3(1), dim3(64), 0, b);
    hipError_t e = hipStreamSynchronize(b);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const hipError_t e2 = hipStreamSynchronize(a);
    if (e == hipSuccess) e = e2;
    *shared = ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3) > kBusyUs * 0.6;
    return e;
}

}  // namespace pnn
