// Split-precision tap GEMM for SMALL M: one 4-wave workgroup (1 MFMA wave + 3 LDS-DMA loader waves) per 32 x 32 output tile.
//
// Why: HM asks for one block at a time (TComPrediction.cpp:572-579,601-608) and the batching service for a handful, so
// M -- blocks x pixels of the layer -- is 1 ... a few hundred rows.  The big-tile kernels (tapgemm_ring / tapgemm_sp /
// convimg_sp) then run ONE or two workgroups, each walking the whole K of a 64-192-row tile alone: 14-60 us per layer
// (profiles/r02_batch1_*), most rows padding.  The old answer, a split-K f32 kernel, is fast but adds K in another order,
// so a block predicted alone could differ in the last bit from the same block predicted inside a batch -- encoder/decoder
// drift unless "canonical_order" forced the slow path.
//
// This kernel keeps the big kernels' arithmetic EXACTLY -- same instruction (v_mfma_f32_32x32x16_f16), same operand roles
// (A = weights, B = activations), per accumulator the 16-deep K chunks in ascending order and per chunk
// w_hi*a_hi, w_hi*a_lo, w_lo*a_hi -- so its results are bit-identical to theirs (tests: every batch size, every kernel
// family), and gets its speed from parallelism over tiles instead of over K:
//   * grid = (M / 32) x (Cout / 32) x classes workgroups of 4 waves, one 32 x 32 tile each; a 1200 x 1200 FC layer at
//     batch 1 is 38 workgroups on 38 CUs, each streaming only ITS 2 KiB of weights per chunk;
//   * wave 0 only reads fragments from an LDS ring and issues MFMAs; waves 1-3 only issue LDS-DMA (buffer_load ... lds, no
//     VGPRs): a stage is 3 chunks, loader j owns chunk j of every stage -- 4 instructions of 1 KiB (w_hi, w_lo and the two
//     activation pieces), every lane fetching exactly the 16 bytes some lane of wave 0 will feed to the MFMA, so the ring
//     is read back lane-linear, conflict-free -- and retires them with counted vmcnt; one s_barrier per stage joins the
//     roles (the ring kernel's protocol at tile size 32 x 32).  4 stages = 48 KiB in flight per workgroup.
//     (A first version had ONE wave do both jobs: 16 us per 1200-deep layer instead of the ~4 us its 225 MFMAs need -- a
//     wave is blocked ~70 cycles while it issues a 1-KiB vector-memory instruction, 280 of its ~350 cycles per chunk.)
//   * rows past M, taps outside the image and chunks past the end of K are buffer-descriptor range misses: zeros, no traffic.
//
// Extras: AF32 -- the activations are plain f32 rows (an FC net's input as HM hands it over) and are split into f16 pairs
// in registers, the same conversion split_kernel applies (saves that launch); `seg` > 0 -- K-segment mode for the output
// layer of an FC net: blockIdx.z selects `seg` chunks of K and the raw partial sums go to part[z][m][64], exactly the
// per-column-tile partials the ring kernel's fused output layer produces (pnn_gemm_ring.hip, FUSE, BN = 16 * seg), so
// fuse_reduce_kernel finishes both the same way.
#include "pnn_kernels.h"
#include <cstddef>
#include <cstring>
#include <type_traits>
#include "pnn_device_common.h"

namespace pnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kSmallCS = 3;                         // chunks per stage = loader waves
#ifndef PNN_SMALL_LA
#define PNN_SMALL_LA 4
#endif
constexpr int kSmallLA = PNN_SMALL_LA;                         // stages in flight ahead of the one being computed (8 stages / 108 KiB measured no
                                                    // faster: 9.0 vs 8.7 us per 1200-deep layer -- the loop is issue-bound, not latency-bound)
constexpr int kSmallD = kSmallLA + 1;               // ring slots (stages): 5 x 12 KiB

template <int N>
__device__ __forceinline__ void small_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// INL (with AF32): the f32 input rows travel INSIDE the kernel-argument block (a single-block FC call: 320 B / 1280 B).  The
// argument block lives in device memory, written by the host with the launch packet, so the first layer's loaders read the
// input like any device buffer instead of fetching it from pinned host memory across PCIe (measured on the K = 80 / 320 first
// layers: 7.5 / 8.4 us with the PCIe read, profiles/r02_batch1_w{4,8}_timeline.txt).
constexpr int kSmallInlineFloats = 512;
struct SmallArgs { TapGemmParams p; int seg; };
struct SmallArgsInline { TapGemmParams p; int seg; float in[kSmallInlineFloats]; };

// The parameter block is read in place, through the CONSTANT address space (the kernel-argument segment): every field, also
// the runtime-indexed tap table, is then a scalar load and nothing is copied to private memory.
typedef const __attribute__((address_space(4))) TapGemmParams CSmallParams;

template <bool AF32, bool INL>
__device__ __forceinline__ void tapgemm_small_body(CSmallParams& p, const int seg, const int bx, const int by, const int bz)
{
    constexpr int CS = kSmallCS, LA = kSmallLA, D = kSmallD;
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [D stages][CS chunks][4 pieces][64 lanes]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: everything derived from it (chunk, tap) stays in SGPRs
    const int l31 = lane & 31, h = lane >> 5;
    const int cls = seg > 0 ? 0 : bz;
    const int n0 = by * 32;
    const int mblk = bx * 32;
    const int SP = p.SH * p.SW;
    const int cpt = p.Cin >> 4;
    const int t0 = p.tap_begin[cls], t1 = p.tap_begin[cls + 1];
    const int nchunks = (t1 - t0) * cpt;
    int c0 = 0, c1 = nchunks;                        // this workgroup's chunks [c0, c1) of the class
    if (seg > 0) {
        c0 = bz * seg;
        c1 = c0 + seg < nchunks ? c0 + seg : nchunks;
    }
    const int nst = (c1 - c0 + CS - 1) / CS;

    // activation row of this lane: m = mblk + l31 -> (block, i, j)
    const int mg = mblk + l31;
    const bool rowok = mg < p.M;
    const int mc = rowok ? mg : 0;
    const int rb = mc / SP;
    const int rq = mc - rb * SP;
    const int ri = rq / p.SW, rj = rq - ri * p.SW;

    if (wave != 0) {
        // ---- loader wave j: chunk j of every stage ---------------------------------------------------------------------------
        const int j = wave - 1;
        const void* xbase = p.X;
        if (INL) xbase = (const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(SmallArgsInline, in);   // generic pointer to the argument block
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, p.x_bytes, 0x00020000);
        const f32x4* __restrict__ Wg = reinterpret_cast<const f32x4*>(p.Wp) + (size_t)p.chunk_begin[cls] * 4 * p.Npad;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0x7fffffffu, 0x00020000);
        const unsigned bstride = (unsigned)(4 * p.Npad) << 4;               // bytes per packed chunk
        const unsigned wlane = (unsigned)((h * p.Npad + n0 + l31) << 4);    // plane (hi, h); the lo planes are 2 * Npad pieces further
        const unsigned wlo_delta = (unsigned)(2 * p.Npad) << 4;
        constexpr unsigned kOob = 0x80000000u;                               // past every descriptor: the load returns zeros
        int ci = c0 + j;                             // next chunk to issue (class-relative), its tap and position inside the tap
        int it = t0 + ci / cpt, icc = ci - (ci / cpt) * cpt;
        unsigned apix = kOob;                        // byte offset of this lane's input pixel for tap `it`
        auto tap_setup = [&](int t) {
            const int tp = p.tap[t < t1 ? t : t1 - 1];
            const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
            const int iy = ri * p.a + dy, ix = rj * p.a + dx;
            const bool ok = rowok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            apix = ok ? (((unsigned)((rb * p.IH + iy) * p.IW + ix) * (unsigned)p.Cin) << 2) : kOob;
        };
        tap_setup(it);
        auto issue = [&](int slot) {                 // DMA of chunk `ci` into its place of ring slot `slot`, then advance by one stage
            f32x4* dst = ring + (slot * CS + j) * 256;
            const bool live = ci < c1;
            const unsigned wo = live ? wlane + (unsigned)ci * bstride : kOob;
            const unsigned ao = (live && apix != kOob) ? apix + (unsigned)(icc << 6) + (unsigned)(h << (AF32 ? 5 : 4)) : kOob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst), 16, wo, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(dst + 64), 16, live ? wo + wlo_delta : kOob, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(dst + 128), 16, ao, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(dst + 192), 16, ao == kOob ? kOob : ao + (AF32 ? 16u : 32u), 0, 0, 0);
            ci += CS; icc += CS;
            if (icc >= cpt) {
                do { icc -= cpt; ++it; } while (icc >= cpt);
                tap_setup(it);
            }
        };
#pragma unroll
        for (int s = 0; s < LA; s++) issue(s);       // stages past the end of K are range misses (zeros): the counts stay uniform
        small_wait_vm<4 * (LA - 1)>();               // stage 0 has landed
        __builtin_amdgcn_s_barrier();
        int slot = LA;                               // slot of stage s + LA = (s + LA) % D, the one stage s - 1 has left
        for (int s = 0; s + 1 < nst; s++) {
            issue(slot);
            if (++slot == D) slot = 0;
            small_wait_vm<4 * (LA - 1)>();           // stage s + 1 has landed
            __builtin_amdgcn_s_barrier();            // barrier s
        }
        small_wait_vm<0>();                          // trailing (range-miss) DMAs must not outlive the workgroup's LDS
        return;
    }

    // ---- MFMA wave -------------------------------------------------------------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    float amax_in = 0.f;                             // AF32: range guard of the converted inputs
    __builtin_amdgcn_s_barrier();                    // stage 0 is in the ring
    // Fragment reads run one chunk ahead of the MFMAs (three register sets, one per chunk of a stage), and the stage barrier
    // sits in front of the LAST chunk's MFMAs: by then the whole stage is in registers, so its slot may be refilled, and the
    // first chunk of the next stage is requested right behind the barrier, under those MFMAs.  (With reads and MFMAs of a
    // chunk back to back, a stage took ~460 cycles for 288 cycles of matrix work.)
    f32x4 fr[CS][4];
    auto read_chunk = [&](int slot, int k) {
        const f32x4* src = ring + (slot * CS + k) * 256 + lane;
        fr[k][0] = src[0]; fr[k][1] = src[64]; fr[k][2] = src[128]; fr[k][3] = src[192];
    };
    auto mfma_chunk = [&](int k) {
        f16x8 ahi, alo;
        if (AF32) {
            // the same split as split_kernel / store_split4: hi = (f16) x, lo = (f16)(x - hi)
            amax_in = amax4(amax4(amax_in, fr[k][2]), fr[k][3]);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ahi[i] = (_Float16)fr[k][2][i]; alo[i] = (_Float16)(fr[k][2][i] - (float)ahi[i]);
                ahi[4 + i] = (_Float16)fr[k][3][i]; alo[4 + i] = (_Float16)(fr[k][3][i] - (float)ahi[4 + i]);
            }
        } else {
            ahi = __builtin_bit_cast(f16x8, fr[k][2]);
            alo = __builtin_bit_cast(f16x8, fr[k][3]);
        }
        const f16x8 whi = __builtin_bit_cast(f16x8, fr[k][0]), wlo = __builtin_bit_cast(f16x8, fr[k][1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi, acc, 0, 0, 0);
    };
    int slot = 0;
    read_chunk(0, 0);
    // No branch inside a stage (round 5, as in tapgemm_f32_small_kernel): the chunks past c1 (tail of the last stage) hold zeros in both
    // operands -- range misses of the loaders -- and three MFMAs on zeros add exact zeros, so they are multiplied like any other instead
    // of being tested for; the chain wave pays every branch on top of its MFMAs.
    for (int s = 0; s + 1 < nst; s++) {              // every stage but the last
        const int nslot = slot + 1 == D ? 0 : slot + 1;
#pragma unroll
        for (int k = 0; k < CS; k++) {
            if (k + 1 < CS) {
                read_chunk(slot, k + 1);
            } else {
                // the whole stage is in registers.  Through the builtin, not inline asm: the compiler's own wait insertion then knows
                // that chunk CS - 1's registers are ready and does not wait for the NEXT stage's first reads in front of its MFMAs
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt / expcnt untouched (gfx9 encoding)
                __builtin_amdgcn_s_barrier();        // barrier s: stage s + 1 is in the ring, slot of stage s may be refilled
                read_chunk(nslot, 0);
            }
            __builtin_amdgcn_sched_barrier(0);       // keep the requests in front of the MFMAs they hide under
            mfma_chunk(k);
        }
        slot = nslot;
    }
#pragma unroll
    for (int k = 0; k < CS; k++) {                   // the last stage: nothing behind it
        if (k + 1 < CS) read_chunk(slot, k + 1);
        else __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_sched_barrier(0);
        mfma_chunk(k);
    }
    if (AF32) report_range(p.range_flag, amax_in);

    if (seg > 0) {                                   // raw partial sums of this K segment (scale, bias, epilogue: fuse_reduce_kernel)
        if (rowok) {
            float* __restrict__ part = p.part + ((size_t)bz * p.M + mg) * 64;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = n0 + 8 * g + 4 * h;
                if (n < 64) *reinterpret_cast<f32x4*>(part + n) = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            }
        }
        return;
    }

    // ---- epilogue: undo the weight scale, bias (+ LeakyReLU); f32 and/or split-f16 outputs, optional HM epilogue --------
    if (!rowok) return;
    const int py = p.py[cls], px = p.px[cls];
    const int oy = ri * p.os + py, ox = rj * p.os + px;
    const size_t obase = (((size_t)rb * p.OH + oy) * p.OW + ox) * p.Cout;
    const bool act = p.act != 0;
    float amax = 0.f;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int n = n0 + 8 * g + 4 * h;
        if (n < p.Cout) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            f32x4 v = (f32x4){acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]} * p.out_scale + bv;
            if (act) {
                v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
            }
            if (p.Y) *reinterpret_cast<f32x4*>(p.Y + obase + n) = v;
            if (p.Yhi) store_split4(p.Yhi, obase, n, v, amax);
            if (p.Yi) {
                int4 iv = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
                *reinterpret_cast<int4*>(p.Yi + obase + n) = iv;
            }
        }
    }
    if (p.Yhi) report_range(p.range_flag, amax);
}

template <bool AF32>
__global__ __launch_bounds__(256) void tapgemm_small_kernel(const SmallArgs args)
{
    touch_kernargs<sizeof(SmallArgs)>();
    (void)args;                                      // == the kernel-argument segment, read in place
    const auto* k = (const __attribute__((address_space(4))) SmallArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    tapgemm_small_body<AF32, false>(k->p, k->seg, blockIdx.x, blockIdx.y, blockIdx.z);
}
__global__ __launch_bounds__(256) void tapgemm_small_inline_kernel(const SmallArgsInline args)
{
    touch_kernargs<sizeof(TapGemmParams)>();
    (void)args;
    const auto* k = (const __attribute__((address_space(4))) SmallArgsInline*)__builtin_amdgcn_kernarg_segment_ptr();
    tapgemm_small_body<true, true>(k->p, k->seg, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Two independent layers in ONE launch (the same layer of the two branches of a convolutional net): workgroups [0, na) work
// on `a`, the rest on `b`.  At small batch a launch costs ~4 us whatever it does; the branches then run side by side
// instead of one after the other (single-block call of the 16x16 net: 13 -> 9 launches).
struct SmallArgs2 { TapGemmParams a, b; int na; };
__global__ __launch_bounds__(256) void tapgemm_small_pair_kernel(const SmallArgs2 args)
{
    touch_kernargs<sizeof(SmallArgs2)>();
    (void)args;
    const auto* k = (const __attribute__((address_space(4))) SmallArgs2*)__builtin_amdgcn_kernarg_segment_ptr();
    const int na = k->na;
    const bool second = (int)blockIdx.x >= na;
    CSmallParams* p = second ? &k->b : &k->a;
    const int wg = second ? (int)blockIdx.x - na : (int)blockIdx.x;
    const int gx = (p->M + 31) >> 5, gy = (p->Cout + 31) >> 5;
    const int bz = wg / (gx * gy), r = wg - bz * gx * gy;
    tapgemm_small_body<false, false>(*p, 0, r % gx, r / gx, bz);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Output layer of a fully-connected PNN with <= 64 outputs at small M, in ONE launch: K-segment partial sums AND their
// reduction (what tapgemm_small_kernel's K-segment mode + fuse_reduce_kernel do in two launches, ~4 us of launch floor each:
// a single-block FC call was 5 launches, 32.8 us of kernels).  One workgroup of 8 waves per 32-row tile; wave z owns K
// segment z (SEG chunks, the fused ring layer's column tile) for both 32-column tiles: the same MFMA chain from a zero
// accumulator -- chunks ascending, w_hi*a_hi, w_hi*a_lo, w_lo*a_hi -- so the partials are the bits of the two-launch path;
// operands go global -> registers (a segment is 10 chunks: NPF of them in flight, no LDS ring, no barrier in the loop);
// the partials meet in LDS and are added in segment order, scaled, biased and rounded exactly as fuse_reduce_kernel does.
// ---------------------------------------------------------------------------------------------------------------------------
struct FcOutArgs { TapGemmParams p; int seg; int pad; DoneSignal done; };
template <int SEG>
__global__ __launch_bounds__(512) void fc_out_small_kernel(const FcOutArgs args)
{
    touch_kernargs<sizeof(FcOutArgs)>();
    (void)args;
    const auto* k = (const __attribute__((address_space(4))) FcOutArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    CSmallParams& p = k->p;
    __shared__ __attribute__((aligned(16))) float part[8][32][64];
    constexpr int NPF = 5;                           // chunks in flight per wave: 6 x 16 bytes per lane each
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int z = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mblk = blockIdx.x * 32;
    const int nchunks = p.Cin >> 4;
    const int nseg = (nchunks + SEG - 1) / SEG;
    const int nct = (p.Cout + 31) >> 5;
    if (z < nseg) {
        const int c0 = z * SEG, c1 = c0 + SEG < nchunks ? c0 + SEG : nchunks;
        const int m = mblk + l31 < p.M ? mblk + l31 : p.M - 1;        // rows past M: any valid row, never stored
        const f32x4* __restrict__ xa = reinterpret_cast<const f32x4*>(p.X) + (size_t)m * (p.Cin >> 2) + h;   // piece (chunk, hi, half h); lo: + 2
        const f32x4* __restrict__ wg = reinterpret_cast<const f32x4*>(p.Wp) + (size_t)h * p.Npad + l31;      // piece (chunk, hi, half h, column l31)
        const size_t wchunk = (size_t)4 * p.Npad, wlo = (size_t)2 * p.Npad;
        f32x4 fa[NPF][2], fw[NPF][2][2];
        auto load = [&](int slot, int c) {
            const int cc = c < c1 ? c : c1 - 1;
            fa[slot][0] = xa[cc * 4];
            fa[slot][1] = xa[cc * 4 + 2];
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                const int col = ct < nct ? ct * 32 : 0;
                fw[slot][ct][0] = wg[cc * wchunk + col];
                fw[slot][ct][1] = wg[cc * wchunk + wlo + col];
            }
        };
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[ct][i] = 0.f;
#pragma unroll
        for (int i = 0; i < NPF; i++) load(i, c0 + i);
#pragma unroll
        for (int i = 0; i < SEG; i++) {
            const int slot = i % NPF;
            if (c0 + i < c1) {
                const f16x8 ahi = __builtin_bit_cast(f16x8, fa[slot][0]), alo = __builtin_bit_cast(f16x8, fa[slot][1]);
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
                    const f16x8 whi = __builtin_bit_cast(f16x8, fw[slot][ct][0]), wl = __builtin_bit_cast(f16x8, fw[slot][ct][1]);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo, acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ahi, acc[ct], 0, 0, 0);
                }
            }
            if (i + NPF < SEG) load(slot, c0 + i + NPF);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<f32x4*>(&part[z][l31][ct * 32 + 8 * g + 4 * h]) = (f32x4){acc[ct][4 * g], acc[ct][4 * g + 1], acc[ct][4 * g + 2], acc[ct][4 * g + 3]};
    }
    __syncthreads();
    // fuse_reduce_kernel's arithmetic, one thread per 4 consecutive columns of a row
    const int mr = tid >> 4, n = (tid & 15) << 2;
    const int mg = mblk + mr;
    if (mg < p.M && n < p.Cout) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < nseg; t++) sum += *reinterpret_cast<const f32x4*>(&part[t][mr][n]);
        const f32x4 v = sum * p.out_scale + *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.Y) *reinterpret_cast<f32x4*>(p.Y + (size_t)mg * p.Cout + n) = v;
        if (p.Yi) *reinterpret_cast<int4*>(p.Yi + (size_t)mg * p.Cout + n) = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
    }
    const DoneSignal done = {k->done.counter, k->done.host_flag, k->done.seq, 0};
    signal_done(done);
}

// p: the output layer as a one-tap GEMM (X = split activations [M][Cin], Wp = its split pack, bias, out_scale, mean, Y / Yi);
// seg_chunks must be the K-segment length of the fused ring layer (10).  false: the layer does not fit this kernel.
bool fc_out_small_fits(const TapGemmParams& p, int seg_chunks)
{
    const int nchunks = p.Cin >> 4;
    return seg_chunks == 10 && p.ncls == 1 && p.SH * p.SW == 1 && p.Cout <= 64 && p.Cout % 4 == 0 && (nchunks + seg_chunks - 1) / seg_chunks <= 8 && p.M > 0;
}

hipError_t launch_fc_out_small(const TapGemmParams& p, int seg_chunks, hipStream_t s, const DoneSignal& done)
{
    if (!fc_out_small_fits(p, seg_chunks)) return hipErrorInvalidValue;
    const FcOutArgs a{p, seg_chunks, 0, done};
    pnn_launch(fc_out_small_kernel<10>, dim3((unsigned)((p.M + 31) / 32)), dim3(512), 0, s, a);
    return hipGetLastError();
}

size_t tapgemm_small_lds_bytes() { return (size_t)kSmallD * kSmallCS * 4 * 64 * 16; }

// Number of workgroups (one 32 x 32 tile each) the layer needs (what the caller compares with the chip): row tiles x column tiles x classes.
long tapgemm_small_tiles(const TapGemmParams& p) { return (long)((p.M + 31) / 32) * ((p.Cout + 31) / 32) * p.ncls; }

hipError_t launch_tapgemm_small(const TapGemmParams& p, bool a_is_f32, int seg_chunks, hipStream_t s, const float* host_input)
{
    static bool attr_done = false;
    if (!attr_done) {                                // 60 KiB of dynamic LDS: above the 48 KiB a kernel gets without asking
        const void* fns[4] = {reinterpret_cast<const void*>(&tapgemm_small_kernel<false>), reinterpret_cast<const void*>(&tapgemm_small_kernel<true>),
                              reinterpret_cast<const void*>(&tapgemm_small_inline_kernel), reinterpret_cast<const void*>(&tapgemm_small_pair_kernel)};
        for (const void* f : fns) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tapgemm_small_lds_bytes());
            if (e != hipSuccess) return e;
        }
        attr_done = true;
    }
    if (p.M <= 0) return hipSuccess;
    unsigned gz = (unsigned)p.ncls;
    if (seg_chunks > 0) {
        if (p.ncls != 1 || p.Cout > 64 || !p.part) return hipErrorInvalidValue;
        const int nchunks = (p.tap_begin[1] - p.tap_begin[0]) * (p.Cin >> 4);
        gz = (unsigned)((nchunks + seg_chunks - 1) / seg_chunks);
    }
    const dim3 grid((p.M + 31) / 32, (p.Cout + 31) / 32, gz);
    const size_t nin = (size_t)p.M * p.IH * p.IW * p.Cin;
    if (a_is_f32 && host_input && nin <= (size_t)kSmallInlineFloats) {   // the input rides in the argument block
        SmallArgsInline a;
        a.p = p; a.seg = seg_chunks;
        memcpy(a.in, host_input, nin * sizeof(float));
        pnn_launch(tapgemm_small_inline_kernel, grid, dim3(256), tapgemm_small_lds_bytes(), s, a);
        return hipGetLastError();
    }
    const SmallArgs a{p, seg_chunks};
    if (a_is_f32) pnn_launch(tapgemm_small_kernel<true>, grid, dim3(256), tapgemm_small_lds_bytes(), s, a);
    else pnn_launch(tapgemm_small_kernel<false>, grid, dim3(256), tapgemm_small_lds_bytes(), s, a);
    return hipGetLastError();
}

hipError_t launch_tapgemm_small_pair(const TapGemmParams& a, const TapGemmParams& b, hipStream_t s)
{
    if (a.M <= 0 || b.M <= 0) return hipErrorInvalidValue;
    SmallArgs2 args;
    args.a = a; args.b = b;
    args.na = (int)tapgemm_small_tiles(a);
    const hipError_t e = launch_tapgemm_small(TapGemmParams{}, false, 0, s);   // M = 0: only makes sure the LDS attribute is set
    if (e != hipSuccess) return e;
    pnn_launch(tapgemm_small_pair_kernel, dim3((unsigned)(args.na + tapgemm_small_tiles(b))), dim3(256), tapgemm_small_lds_bytes(), s, args);
    return hipGetLastError();
}

}  // namespace pnn
