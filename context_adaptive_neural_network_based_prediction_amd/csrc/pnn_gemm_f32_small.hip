// Exact-f32 tap GEMM for SMALL M in the canonical f32 order: one 4-wave workgroup (1 MFMA wave + 3 LDS-DMA loader waves) per
// 16 x 16 output tile, on v_mfma_f32_16x16x4_f32.
//
// Why (round 5): the reference runs Session::Run in float32 (TComPrediction.cpp:572-579,601-608) one block at a time, and the
// batching service hands over handfuls.  tapgemm_f32_kernel's canonical order -- per output a k-ordered fmaf chain, per 16-deep chunk
// k = 0, 8, 1, 9, ..., 7, 15 (step e of v_mfma_f32_32x32x2_f32 adds k = e from lane half 0, then k = 8 + e from lane half 1) -- is one
// dependent chain of K / 2 matrix instructions of 64 cycles each: 32 cycles per k, 16 us for a 1200-deep FC layer whatever the batch
// (profiles/r04_batch1_latency.txt: FC 8x8 72.5 us per single-block call against 39.5 on the split-f16 mode, conv 16x16 212 against
// 77), which is why HM ran on the split mode.  The f32 matrix instructions are bit for bit a k-ordered fmaf chain (one rounding per
// product, no wider accumulation), so the SAME chain can be issued through the 16x16x4 form: 4 k per instruction at 40 cycles of
// dependent latency = 10 cycles per k, 3.2 x shorter -- IF its four lane groups q are fed the k the canonical order visits next:
//     instruction i (0..3) of a chunk, lane group q:   k = 8 (q & 1) + 2 i + (q >> 1)        i = 0: 0, 8, 1, 9;  i = 1: 2, 10, 3, 11; ...
// The operands reach the MFMA wave already in that order: the weights from a second pack in device memory (pack_kn_chain,
// pnn_model.cpp: [K/16][q][Npad][4], element i of lane group q = the k above), the activations (NHWC) permuted on their way into LDS
// by the LDS-DMA itself (4-byte pieces, the loader waves choose each lane's source) -- the MFMA wave itself issues two 16-byte LDS reads and four MFMAs per chunk and nothing else, because a wave
// that runs ONE dependent chain pays every other instruction on top of the chain's 32 cycles per MFMA (tools/f32_chain_probe.hip,
// profiles/r05_f32_chain_probe.txt).  Same bits as every tile of tapgemm_f32_kernel at every batch size
// (tests/test_gpu_parity.py: test_one_summation_order_at_every_batch_size, test_f32_small_kernel_bit_identical).
//
// Structure = tapgemm_small_kernel's (pnn_gemm_small.hip) at tile 16 x 16: grid = (M / 16) x (Cout / 16) x (classes x K segments);
// wave 0 reads fragments from an LDS ring and issues MFMAs; waves 1-3 only issue LDS-DMA -- loader j owns chunk j of every 3-chunk
// stage (the chunk's weights [4 lane groups][16 columns]: one 1-KiB instruction; its activations [4 lane groups][16 rows][4]: four
// 256-byte instructions), counted vmcnt, one s_barrier per stage.  Rows past M, taps outside the image: buffer-descriptor range misses (zeros, no traffic).  A 1200 x 1200 FC
// layer at batch 1 is 75 workgroups, each streaming its 75 KiB of weights.  K segments of the deep convolution layers
// (GemmLayer::nseg) as in tapgemm_f32_kernel: z = class * nseg + segment, raw sums to plane `segment`, seg_reduce_kernel finishes.
#include "pnn_kernels.h"
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include "pnn_device_common.h"
#include "pnn_small_bodies.h"

namespace pnn {

#ifndef PNN_F32S_CPL
#define PNN_F32S_CPL 1
#endif
#ifndef PNN_F32S_LA
#define PNN_F32S_LA 6
#endif
constexpr int kF32SmallNL = 3;                      // loader waves
constexpr int kF32SmallCPL = PNN_F32S_CPL;          // chunks per loader and stage
constexpr int kF32SmallCS = kF32SmallNL * kF32SmallCPL;   // chunks per stage: one barrier per stage in the MFMA wave
constexpr int kF32SmallLA = PNN_F32S_LA;            // stages in flight ahead of the one being computed
// ring slots (stages).  TWO more than in flight: the loaders refill the slot of stage s - 2 behind the barrier that ends stage s - 1 in
// the MFMA wave, so that wave need not wait for its last fragment reads in front of the barrier (with LA + 1 slots it had to:
// 212 cycles per chunk with nothing else in the way, tools/f32_chain_probe.hip says 146)
constexpr int kF32SmallD = kF32SmallLA + 2;

constexpr int kF32SmallInlineFloats = 512;
struct F32SmallArgs { TapGemmParams p; };
struct F32SmallArgsInline { TapGemmParams p; float in[kF32SmallInlineFloats]; };
typedef const __attribute__((address_space(4))) TapGemmParams CF32SmallParams;

namespace {
template <int N>
__device__ __forceinline__ void f32s_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void f32s_dma16(const __amdgpu_buffer_rsrc_t& r, unsigned voff, f32x4* l)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, 0, 0, 0);
}
}  // namespace

// INL: the f32 input rows of an FC net's first layer travel INSIDE the kernel-argument block (see tapgemm_small_inline_kernel).
// XCH (round 6): the activations are in CHAIN ORDER -- NHWC as ever, but within every 16-channel group channel kk sits at position
// 4 g + i with g = (kk >> 3) + 2 (kk & 1), i = (kk & 7) >> 1, i.e. the four values lane group g multiplies in a chunk are 16 contiguous
// bytes.  A loader then fetches a chunk's activations with ONE 16-byte LDS-DMA instruction (lane = (g, row): piece g of the row's
// group) instead of four 4-byte ones whose lanes pick single floats.  Those four were what a conv layer's chunk cost above the chain:
// 200-250 cycles per chunk against the FC layers' 152 (their rows past M = 1 are range misses), and with one instruction in their place
// (ablation build) a single 16x16 block took 73.7 us instead of 86.9 (profiles/r06_b1_chain_order.txt).  The PRODUCER of such a tensor
// is the epilogue below (p.chain_io bit 1: two 8-byte stores per lane instead of one 16-byte); the pass decides which tensors travel
// that way (both ends on these kernels: pnn_passes.cpp).  Same values in another place: not a bit of any sum changes.
// WT (round 6): the tile is written THROUGH to memory (store16_through) as 1 KiB of its own (tile-major, pnn_small_bodies.h) -- its
// reader is a tail of this same launch, on another XCD whose L2 does not see this one's (small_tail_* below); no K segments.
template <bool INL, int LA, bool XCH = false, bool WT = false>
__device__ __forceinline__ void tapgemm_f32_small_body(CF32SmallParams& p, const int bx, const int by, const int bz, const int gy)
{
    static_assert(!(INL && XCH), "the argument-block input is the caller's raw context");
    constexpr int NL = kF32SmallNL, CPL = kF32SmallCPL, CS = kF32SmallCS, D = LA + 2;
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [D stages][CS chunks][weights 64 pieces | activations 64 pieces]
#ifdef PNN_F32_DIAG                                 // diagnostic library only (make diag): 100 MHz stamps of the MFMA wave -> p.Xlo[workgroup][8]
    const unsigned long long de0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    // Which wave runs the MFMA chain rotates with the tile (role 0 = the chain, roles 1-3 = the loaders): a workgroup's wave i sits on
    // SIMD i, and with several small launches on the chip at once (the batching service's five width workers) two tiles that share a
    // CU would otherwise both run their chains on SIMD 0 -- each at half the rate -- beside three SIMDs that only issue loads.
#ifdef PNN_F32S_NO_ROT
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
    const int wave = __builtin_amdgcn_readfirstlane((int)(((threadIdx.x >> 6) + (unsigned)(bx + by + bz)) & 3u));
#endif
    const int l15 = lane & 15, q = lane >> 4;
    const int nseg = p.nseg > 1 ? p.nseg : 1;
    const int cls = nseg > 1 ? bz / nseg : bz;
    const int seg = bz - cls * nseg;
    const int n0 = by * 16;
    const int SP = p.SH * p.SW;
    const int cpt = p.Cin >> 4;
    const int t0 = p.tap_begin[cls], t1 = p.tap_begin[cls + 1];
    // this workgroup's taps [ts0, ts1) of the class (class-relative): all of them, or segment `seg`'s share -- the class's taps dealt in
    // order, the first (taps % nseg) segments one more (tapgemm_f32_kernel's smask)
    int ts0 = 0, ts1 = t1 - t0;
    if (nseg > 1) {
        const int base = (t1 - t0) / nseg, rem = (t1 - t0) - base * nseg;
        ts0 = seg * base + (seg < rem ? seg : rem);
        ts1 = ts0 + base + (seg < rem ? 1 : 0);
    }
    const int c0 = ts0 * cpt, c1 = ts1 * cpt;        // chunks [c0, c1) of the class
    const int nst = (c1 - c0 + CS - 1) / CS;

    // the activation row this lane works for: m = 16 bx + l15 -> (block, i, j)
    const int mg = bx * 16 + l15;
    const bool rowok = mg < p.M;
    const int mc = rowok ? mg : 0;
    int rb, ri, rj;
    if (SP == 1) { rb = mc; ri = 0; rj = 0; }
    else {
        rb = mc / SP;
        const int rq = mc - rb * SP;
        ri = rq / p.SW; rj = rq - ri * p.SW;
    }

    if (wave != 0) {
        // ---- loader wave j: chunk j of every stage, LDS-DMA only (no registers, no LDS instructions).  Weights: piece (q, column l15)
        // of the chain-ordered pack, one 16-byte instruction.  Activations (NHWC in memory, k = 4 piece + e): the MFMA wave's lane group
        // g wants k = 8 (g & 1) + 2 i + (g >> 1), i = 0..3 -- four 4-byte instructions, one per lane group g: lane l fetches element
        // i = l & 3 of row l >> 2, landing lane-linear at float (g * 16 + row) * 4 + i of the chunk's activation block.
        const int j = wave - 1;
        const void* xbase = p.X;
        if (INL) xbase = (const char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(F32SmallArgsInline, in);
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, p.x_bytes, 0x00020000);
        const f32x4* __restrict__ Wg = reinterpret_cast<const f32x4*>(p.Wp) + (size_t)p.chunk_begin[cls] * 4 * p.Npad;
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0x7fffffffu, 0x00020000);
        const unsigned bstride = (unsigned)(4 * p.Npad) << 4;               // bytes per packed chunk
        const unsigned wlane = (unsigned)((q * p.Npad + n0 + l15) << 4);
        constexpr unsigned kOob = 0x80000000u;
        // the activation row this lane FETCHES for: m = 16 bx + (lane >> 2) (element lane & 3 of it), XCH: 16 bx + (lane & 15) (piece lane >> 4)
        const int mx = bx * 16 + (XCH ? (lane & 15) : (lane >> 2));
        const bool xok = mx < p.M;
        const int mxc = xok ? mx : 0;
        int xb, xi, xj;
        if (SP == 1) { xb = mxc; xi = 0; xj = 0; }
        else {
            xb = mxc / SP;
            const int rq = mxc - xb * SP;
            xi = rq / p.SW; xj = rq - xi * p.SW;
        }
        const unsigned xlane = XCH ? (unsigned)((lane >> 4) << 4) : (unsigned)((lane & 3) << 3);   // element i = lane & 3: k advances by 2 per i; XCH: 16-byte piece lane >> 4
        int ci = c0 + j;                             // this loader's chunks: c0 + j, + NL, + 2 NL, ...
        int it = t0 + ci / cpt, icc = ci - (ci / cpt) * cpt;
        unsigned apix = kOob;
        auto tap_setup = [&](int t) {
            const int tp = p.tap[t < t1 ? t : t1 - 1];
            const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
            const int iy = xi * p.a + dy, ix = xj * p.a + dx;
            const bool ok = xok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            apix = ok ? (((unsigned)((xb * p.IH + iy) * p.IW + ix) * (unsigned)p.Cin) << 2) + xlane : kOob;
        };
        tap_setup(it);
        auto issue1 = [&](int slot, int u) {         // chunk `ci` into place j + NL u of ring slot `slot`, then on to this loader's next chunk
            f32x4* dst = ring + (slot * CS + j + NL * u) * 128;
            const bool live = ci < c1;
            const unsigned wo = live ? wlane + (unsigned)ci * bstride : kOob;
            const unsigned ao = live ? apix : kOob;
            const unsigned so = (unsigned)(icc << 6);
            ci += NL; icc += NL;
#ifdef PNN_F32S_NO_DMA                              // ablation build: the loaders issue nothing (timing only, results are garbage)
            (void)wo; (void)ao; (void)so; (void)dst;
            return;
#endif
            f32s_dma16(wrsrc, wo, dst);
            float* xd = reinterpret_cast<float*>(dst + 64);
            if (XCH) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd), 16, ao, so, 0, 0);
            } else {
                // lane group g: first k = 8 (g & 1) + (g >> 1) -> byte offsets 0, 32, 4, 36 -- in the SCALAR offset: the instruction's
                // immediate offset moves the LDS destination as well as the source
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd), 4, ao, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 64), 4, ao, so + 32u, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 128), 4, ao, so + 4u, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 192), 4, ao, so + 36u, 0, 0);
            }
            if (icc >= cpt) {
                do { icc -= cpt; ++it; } while (icc >= cpt);
                tap_setup(it);
            }
        };
        auto issue = [&](int slot) {                 // this loader's CPL chunks of one stage
#pragma unroll
            for (int u = 0; u < CPL; u++) issue1(slot, u);
        };
        constexpr int PER = (XCH ? 2 : 5) * CPL;     // vector-memory instructions per issue()
        static_assert(PER * LA < 64, "vmcnt is a 6-bit counter");
#pragma unroll
        for (int s = 0; s < LA; s++) issue(s);
        f32s_wait_vm<PER * (LA - 1)>();              // stage 0 has landed
        __builtin_amdgcn_s_barrier();
        int slot = LA;
        for (int s = 0; s + 1 < nst; s++) {
            issue(slot);                             // stage s + LA into the slot stage s - 2 left
            if (++slot == D) slot = 0;
            f32s_wait_vm<PER * (LA - 1)>();          // stage s + 1 has landed
            __builtin_amdgcn_s_barrier();
        }
        f32s_wait_vm<0>();                           // trailing (range-miss) DMAs must not outlive the workgroup's LDS
        return;
    }

    // ---- MFMA wave: per chunk two 16-byte LDS reads (this lane's four weights, its four activations) and four MFMAs -- nothing else:
    // whatever else the wave issues is ADDED to the chain's 32 cycles per instruction (tools/f32_chain_probe.hip; the first version
    // of this kernel read the standard packs and picked its elements with v_cndmask: 370 cycles per chunk for 128 of matrix work)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#ifndef PNN_F32S_NO_PRIO
    __builtin_amdgcn_s_setprio(3);                   // the chain cannot hide a lost issue slot; the loaders beside it can
#endif
    __builtin_amdgcn_s_barrier();                    // stage 0 is in the ring
    f32x4 fw[CS], fx[CS];
    int slot = 0;
    { const f32x4* src = ring + lane; fw[0] = src[0]; fx[0] = src[64]; }
#ifdef PNN_F32_DIAG                                 // diagnostic library only (make diag): cycles and 100 MHz ticks of the MFMA wave's loop
    const unsigned long long dq0 = __builtin_amdgcn_s_memtime(), dr0 = __builtin_amdgcn_s_memrealtime();
#endif
    // The next chunk's two reads sit BETWEEN this chunk's MFMAs (one behind the first, one behind the third): both in front of the
    // four measured 170 cycles per chunk in tools/f32_chain_probe.hip, interleaved 146 (the bare chain: 128).  And NO branch inside a
    // stage: a chunk past the end of K (the tail of the last stage) holds zeros in both operands -- range misses of the loaders -- and
    // fma(0, 0, acc) = acc exactly, so it is multiplied like any other instead of being skipped (with a liveness test per chunk the
    // loop ran at 211 cycles per chunk, without at 165: PNN_F32S_DIAG).
    auto chunk = [&](int k, const f32x4* nsrc, int kn, bool rd) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][0], fx[k][0], acc, 0, 0, 0);
        if (rd) fw[kn] = nsrc[0];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][1], fx[k][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][2], fx[k][2], acc, 0, 0, 0);
        if (rd) fx[kn] = nsrc[64];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][3], fx[k][3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s + 1 < nst; s++) {              // every stage but the last: its last chunk fetches the next stage's first
        const int nslot = slot + 1 == D ? 0 : slot + 1;
#pragma unroll
        for (int k = 0; k + 1 < CS; k++) chunk(k, ring + (slot * CS + k + 1) * 128 + lane, k + 1, true);
        __builtin_amdgcn_s_barrier();                // stage s + 1 is in the ring, the slot of stage s - 1 may be refilled
        chunk(CS - 1, ring + (nslot * CS) * 128 + lane, 0, true);
        slot = nslot;
    }
#pragma unroll
    for (int k = 0; k + 1 < CS; k++) chunk(k, ring + (slot * CS + k + 1) * 128 + lane, k + 1, true);
    chunk(CS - 1, ring, 0, false);
#ifdef PNN_F32_DIAG
    unsigned long long* const dstamp = p.Xlo ? (unsigned long long*)p.Xlo + 8 * ((bz * gy + by) * ((p.M + 15) >> 4) + bx) : nullptr;
    if (dstamp && lane == 0) {
        dstamp[0] = __builtin_amdgcn_s_memtime() - dq0; dstamp[1] = __builtin_amdgcn_s_memrealtime() - dr0; dstamp[2] = (unsigned long long)(c1 - c0); dstamp[3] = dr0;
        dstamp[4] = de0;
    }
    // (the exit stamp: behind the wave's last stores, acknowledged)
#define F32S_DIAG_EXIT() do { if (dstamp && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dstamp[5] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define F32S_DIAG_EXIT() do { } while (0)
#endif
    // ---- epilogue: lane (q, l15) holds row m = l15, channels n0 + 4 q + r ----------------------------------------------------
    const int n = n0 + 4 * q;
    const int py = p.py[cls], px = p.px[cls];
    const int oy = ri * p.os + py, ox = rj * p.os + px;
    const size_t obase = (((size_t)rb * p.OH + oy) * p.OW + ox) * p.Cout;
    if (!INL && nseg > 1 && p.seg_cnt) {
        // K segments added up inside the launch.  This wave's raw sums go to plane `seg` WRITTEN THROUGH to memory (the tile's other
        // segments ran on other XCDs, whose L2s this XCD's does not see); it then counts itself on the tile's counter (memory-side atomic),
        // and the LAST of the tile's nseg waves to arrive reads all planes back PAST the caches, adds them in plane order -- the order of
        // seg_reduce_kernel, whichever wave happens to be last --, applies bias and activation and stores the layer's output.  One launch
        // instead of two per segmented layer of a single-block call (32x32 net: 6, 64x64 net: 9).
        // The planes are TILE-major here ([segment][tile][lane] x 16 bytes, 1 KiB per (segment, tile), the wave's lanes side by side):
        // every 128-byte line of them is written by ONE workgroup and read by one, so no XCD's L2 ever holds a line that another
        // workgroup completes later (in the [pixel][channel] layout two 16-channel tiles share a line).
        const bool ok = rowok && n < p.Cout;
        const int gxs = (p.M + 15) >> 4, tile = (cls * gy + by) * gxs + bx;
        const size_t plane_floats = (size_t)p.ncls * gy * gxs * 256;
        float* const plane0 = p.Y + (size_t)tile * 256 + lane * 4;
        store16_through_mfma(reinterpret_cast<f32x4*>(plane0 + (size_t)seg * plane_floats), acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // acknowledged: the plane's bytes are in memory
        unsigned* const cnt = p.seg_cnt + tile;
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        if (old != (unsigned)nseg - 1u) { F32S_DIAG_EXIT(); return; }
        if (lane == 0) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        if (!ok) { F32S_DIAG_EXIT(); return; }
        // (GemmLayer::nseg <= 8) all planes requested, one wait, then the additions in plane order.  Loads and wait are ONE asm block:
        // the compiler must not touch a destination register (a copy, a select) while its load is in flight.  Planes past nseg: plane 0 again, unused.
        f32x4 pl[8];
        const f32x4* pa[8];
#pragma unroll
        for (int sgm = 0; sgm < 8; sgm++) pa[sgm] = reinterpret_cast<const f32x4*>(plane0 + (size_t)(sgm < nseg ? sgm : 0) * plane_floats);
        asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\tglobal_load_dwordx4 %2, %10, off sc0 sc1\n\t"
                     "global_load_dwordx4 %3, %11, off sc0 sc1\n\tglobal_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                     "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(pl[0]), "=&v"(pl[1]), "=&v"(pl[2]), "=&v"(pl[3]), "=&v"(pl[4]), "=&v"(pl[5]), "=&v"(pl[6]), "=&v"(pl[7])
                     : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pa[4]), "v"(pa[5]), "v"(pa[6]), "v"(pa[7])
                     : "memory");
        f32x4 t = pl[0];
#pragma unroll
        for (int sgm = 1; sgm < 8; sgm++) if (sgm < nseg) t += pl[sgm];
        f32x4 v = t + *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
        store4_chain(p.seg_Y + obase + n0, q, v, (p.chain_io & 2) != 0);
        F32S_DIAG_EXIT();
        return;
    }
    if (!rowok || n >= p.Cout) { F32S_DIAG_EXIT(); return; }
    float* const Yo = (nseg > 1 && p.Y) ? p.Y + (size_t)seg * p.seg_stride : p.Y;
    f32x4 v = acc + *reinterpret_cast<const f32x4*>(p.bias + n);
    if (p.act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
    if (WT) store16_through(reinterpret_cast<f32x4*>(Yo) + ((size_t)(by * ((p.M + 15) >> 4) + bx) * 64 + lane), v);   // tile-major: tile_major_piece, pnn_small_bodies.h
    else if (Yo) store4_chain(Yo + obase + n0, q, v, (p.chain_io & 2) != 0 && !(nseg > 1));   // (raw K-segment planes: as they are; the fold writes the layer's output)
    if (p.Yi) *reinterpret_cast<int4*>(p.Yi + obase + n) = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
    F32S_DIAG_EXIT();
}

// LA (stages of 3 chunks in flight ahead of the chain) is a template parameter: kF32SmallLA = 6 (48 KiB of ring: three workgroups per
// CU) for launches of many tiles, kF32SmallLADeep = 12 (84 KiB: one per CU; the most the 6-bit vmcnt counts) for launches of at most one
// workgroup per CU -- the FC layers' 75, the small conv layers.  The deep ring changes nothing for a call that runs alone, its weights
// coming from L2 (37-39 us per FC call either way); inside a campaign five nets' weights (135 MB) take turns in 32 MB of L2 and arrive
// from the MALL / HBM with a latency that 18 chunks of lookahead (1.1 us at the chain's pace) do not cover and 36 chunks do: a 4x4 / 8x8
// call 55 -> 47 us inside configs[3], a 16x16 call 117 -> 107, the campaign 5.8-6.1 -> 5.5 s.  (Deep rings everywhere: conv 64x64 single
// block 236 -> 259 us -- its 1024-tile layers then run in rounds.)
template <int LA, bool XCH>
__global__ __launch_bounds__(256) void tapgemm_f32_small_kernel(const F32SmallArgs args)
{
    touch_kernargs<sizeof(F32SmallArgs)>();
    (void)args;
    const auto* k = (const __attribute__((address_space(4))) F32SmallArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    tapgemm_f32_small_body<false, LA, XCH>(k->p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}
template <int LA>
__global__ __launch_bounds__(256) void tapgemm_f32_small_inline_kernel(const F32SmallArgsInline args)
{
    touch_kernargs<sizeof(TapGemmParams)>();
    (void)args;
    const auto* k = (const __attribute__((address_space(4))) F32SmallArgsInline*)__builtin_amdgcn_kernarg_segment_ptr();
    tapgemm_f32_small_body<true, LA>(k->p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}

// Two independent layers in ONE launch (the same layer of the two branches of a convolutional net), see tapgemm_small_pair_kernel.
struct F32SmallArgs2 { TapGemmParams a, b; int na; };
template <int LA, bool XCH>
__global__ __launch_bounds__(256) void tapgemm_f32_small_pair_kernel(const F32SmallArgs2 args)
{
    touch_kernargs<sizeof(F32SmallArgs2)>();
    (void)args;
    const auto* k = (const __attribute__((address_space(4))) F32SmallArgs2*)__builtin_amdgcn_kernarg_segment_ptr();
    const int na = k->na;
    const bool second = (int)blockIdx.x >= na;
    CF32SmallParams* p = second ? &k->b : &k->a;
    const int wg = second ? (int)blockIdx.x - na : (int)blockIdx.x;
    const int gx = (p->M + 15) >> 4, gy = (p->Cout + 15) >> 4;
    const int bz = wg / (gx * gy), r = wg - bz * gx * gy;
    tapgemm_f32_small_body<false, LA, XCH>(*p, r % gx, r / gx, bz, gy);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// TAILS (round 6): the small layer BEHIND a GEMM layer of a small conv pass inside that layer's launch.  A single-block call of the
// 16x16 net is nine launches, and the device-side stamps price a kernel boundary at 1.4-2.0 us from the last exit to the next entry plus
// 1.1-1.4 us until the first MFMA (profiles/r06_b1_stamps.txt) -- more than the merger (8 active workgroups) or the last transposed
// convolution (one workgroup per block) take themselves.  So the workgroups of the layer in front write their tiles through to memory,
// count themselves on a counter per tail instance (memory-side atomics), and the LAST one to arrive for an instance runs it: no
// workgroup ever waits for another (nothing to dead-lock behind other contexts' kernels), the instance starts the moment its inputs are
// complete instead of one launch later.  Kind 1 (pair launch of the branches' last layers): the merger's tile (block b, channels 16 by ..)
// needs the 3 + 2 tiles (b, by) of the two branch maps -- 5 arrivals; kind 2 (the last GEMM of the transposed stack): the last layer of
// block b needs all (pixels per block / 16) x (Cout / 16) tiles of it.  The bodies are those of merger_mfma_kernel and
// tconv_cout1_mfma_kernel (pnn_small_bodies.h): the same bits as the separate launches (tests/test_gpu_parity.py).
struct F32SmallTailArgs { TapGemmParams p; SmallTail t; };
struct F32SmallTail2Args { TapGemmParams a, b; int na; SmallTail t; };
typedef const __attribute__((address_space(4))) SmallTail CSmallTail;

// every wave of the workgroup calls this behind tapgemm_f32_small_body; true in ALL its waves of the one workgroup that completes `group`
__device__ __forceinline__ bool small_tail_arrive(unsigned* cnt, const unsigned expected, unsigned* lds_word)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the chain wave's write-through stores are acknowledged (the loaders have none in flight)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == expected - 1u) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        *lds_word = old == expected - 1u ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *lds_word != 0u;
    __syncthreads();                                 // (the word is part of the LDS the tail is about to use)
    return last;
}

template <int LA, bool XCH>
__global__ __launch_bounds__(256) void tapgemm_f32_small_tail_kernel(const F32SmallTailArgs args)
{
    touch_kernargs<sizeof(F32SmallTailArgs)>();
    const auto* k = (const __attribute__((address_space(4))) F32SmallTailArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];
    tapgemm_f32_small_body<false, LA, XCH, true>(k->p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
    // kind 2: the last transposed convolution of the block this tile belongs to
    const int SP = k->p.SH * k->p.SW;
    const long b = ((long)blockIdx.x * 16) / SP;
    if (!small_tail_arrive(k->t.cnt + b, (unsigned)(SP >> 4) * gridDim.y, reinterpret_cast<unsigned*>(ring))) return;
    const TConv1Params& tp = args.t.t;
    if (tp.Cin == 64) tconv_cout1_mfma_band<2, 5, true>(tp, reinterpret_cast<float*>(ring), b, 0);   // (launch-uniform)
    else tconv_cout1_lds_tile<true>(tp, ring, (int)b, 0);
    if (tp.done.host_flag) {                         // one flag word per block (DoneSignal::per_wg)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(tp.done.host_flag + b, tp.done.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int LA, bool XCH>
__global__ __launch_bounds__(256) void tapgemm_f32_small_pair_tail_kernel(const F32SmallTail2Args args)
{
    touch_kernargs<sizeof(F32SmallTail2Args)>();
    const auto* k = (const __attribute__((address_space(4))) F32SmallTail2Args*)__builtin_amdgcn_kernarg_segment_ptr();
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];
    const int na = k->na;
    const bool second = (int)blockIdx.x >= na;
    CF32SmallParams* p = second ? &k->b : &k->a;
    const int wg = second ? (int)blockIdx.x - na : (int)blockIdx.x;
    const int gx = (p->M + 15) >> 4, gy = (p->Cout + 15) >> 4;
    const int bz = wg / (gx * gy), r = wg - bz * gx * gy;
    const int bx = r % gx, by = r / gx;
    tapgemm_f32_small_body<false, LA, XCH, true>(*p, bx, by, bz, gy);
    // kind 1: the merger's tile (block b, channel group by)
    const int SP = p->SH * p->SW;
    const long b = ((long)bx * 16) / SP;
    const MergerParams& mp = args.t.m;
    if (!small_tail_arrive(k->t.cnt + b * gy + by, (unsigned)((mp.na + mp.nl) >> 4), reinterpret_cast<unsigned*>(ring))) return;
    merger_mfma_tile<false, true>(mp, reinterpret_cast<f32x4 (*)[16][64]>(ring), by * 16, b);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// K-segmented ONE-TAP (FC) layers at small M (round 6).  The device-side stamps of a single-block FC 8x8 call
// (profiles/r06_b1_stamps_fc_out.txt) show each 1200-deep hidden layer as 1.3 us from entry to the first MFMA, 4.7 us of ONE dependent chain
// of 75 chunks, 0.4 us until its stores are acknowledged -- on 75 of 256 CUs, with three of a CU's four matrix pipes idle.  The layer's
// canonical order is therefore (GemmLayer::fc_seg_chunks, pnn_model.cpp) the sum of <= 4 K segments of kFcSegChunks chunks, each the
// k-ordered fmaf chain of the kernel above, added up in order -- and here the FOUR chains of an output tile run side by side in ONE
// workgroup: waves 0-3 are the chain waves (one per SIMD, segment = wave), waves 4-7 their loaders (loader j feeds chain j: a ring of its
// own, CPL chunks per stage, LA stages ahead, the LDS-DMA forms of the kernel above), one s_barrier per stage for all eight.  The segment
// sums meet in LDS and wave 0 adds them in order, + bias, activation -- total = ((p0 + p1) + p2) + p3, the bits of tapgemm_f32_kernel's
// seg_seq form at any batch size (a chain sum that starts from +0 is never -0, so 0 + p0 = p0 there).  Short segments (the last of
// 1200 = 3 x 320 + 240) and missing ones (nseg < 4) multiply zeros that their loaders' range misses deliver: fma(0, 0, acc) = acc.
constexpr int kFcSegNS = 4, kFcSegLA = 5, kFcSegCPL = 2;
static_assert(5 * kFcSegCPL * kFcSegLA < 64, "vmcnt is a 6-bit counter");
constexpr size_t kFcSegLds = ((size_t)kFcSegNS * (kFcSegLA + 2) * kFcSegCPL * 128 + kFcSegNS * 64) * 16;

template <bool XCH>
__global__ __launch_bounds__(512) void fcseg_f32_small_kernel(const F32SmallArgs args)
{
    touch_kernargs<sizeof(F32SmallArgs)>();
    (void)args;
    const auto* ka = (const __attribute__((address_space(4))) F32SmallArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    CF32SmallParams& p = ka->p;
    constexpr int NS = kFcSegNS, LA = kFcSegLA, CPL = kFcSegCPL, D = LA + 2;
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [NS chains][D stages][CPL chunks][weights 64 pieces | activations 64 pieces] | [NS][64] segment sums
#ifdef PNN_F32_DIAG
    const unsigned long long de0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, q = lane >> 4;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int n0 = by * 16;
    const int cpt = p.Cin >> 4, segc = (int)p.seg_chunks;
    const int sg = wave & (NS - 1);                   // the segment this wave runs (waves 0-3) or feeds (waves 4-7)
    int c0 = sg * segc, c1 = c0 + segc;               // its chunks [c0, c1) of the tap
    if (c0 > cpt) c0 = cpt;
    if (c1 > cpt) c1 = cpt;
    if (sg >= p.nseg) c1 = c0;
    const int nst = (segc + CPL - 1) / CPL;           // the same number of stages (barriers) for all eight waves
    f32x4* const myring = ring + sg * (D * CPL * 128);
    f32x4* const part = ring + NS * (D * CPL * 128);

    if (wave >= NS) {
        // ---- loader of chain sg: LDS-DMA only.  Weights: piece (q, column l15) of the chain-ordered pack, one 16-byte instruction per
        // chunk; activations (rows of [M][Cin]): lane l fetches element i = l & 3 of row l >> 2 for lane group g, four 4-byte instructions
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7fffffffu, 0x00020000);
        const unsigned bstride = (unsigned)(4 * p.Npad) << 4;               // bytes per packed chunk
        const unsigned wlane = (unsigned)((q * p.Npad + n0 + l15) << 4);
        constexpr unsigned kOob = 0x80000000u;
        const int mx = bx * 16 + (XCH ? (lane & 15) : (lane >> 2));
        const unsigned apix = mx < p.M ? (((unsigned)mx * (unsigned)p.Cin) << 2) + (XCH ? (unsigned)((lane >> 4) << 4) : (unsigned)((lane & 3) << 3)) : kOob;
        int ci = c0;
        auto issue = [&](int slot) {
#pragma unroll
            for (int u = 0; u < CPL; u++) {
                f32x4* dst = myring + (slot * CPL + u) * 128;
                const bool live = ci < c1;
                const unsigned wo = live ? wlane + (unsigned)ci * bstride : kOob;
                const unsigned ao = live ? apix : kOob;
                const unsigned so = (unsigned)(ci << 6);
                ci++;
                f32s_dma16(wrsrc, wo, dst);
                float* xd = reinterpret_cast<float*>(dst + 64);
                if (XCH) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd), 16, ao, so, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd), 4, ao, so, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 64), 4, ao, so + 32u, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 128), 4, ao, so + 4u, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 192), 4, ao, so + 36u, 0, 0);
                }
            }
        };
        constexpr int PER = (XCH ? 2 : 5) * CPL;
#pragma unroll
        for (int st = 0; st < LA; st++) issue(st);
        f32s_wait_vm<PER * (LA - 1)>();              // stage 0 has landed
        __builtin_amdgcn_s_barrier();
        int slot = LA;
        for (int st = 0; st + 1 < nst; st++) {
            issue(slot);                             // stage st + LA into the slot stage st - 2 left
            if (++slot == D) slot = 0;
            f32s_wait_vm<PER * (LA - 1)>();          // stage st + 1 has landed
            __builtin_amdgcn_s_barrier();
        }
        f32s_wait_vm<0>();                           // trailing (range-miss) DMAs must not outlive the workgroup's LDS
        __builtin_amdgcn_s_barrier();                // the segment sums' barrier (below): every wave of the workgroup joins it
        return;
    }

    // ---- chain wave of segment sg: two 16-byte LDS reads and four MFMAs per chunk, nothing else (see tapgemm_f32_small_body) ----
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_barrier();                    // stage 0 is in the rings
    f32x4 fw[CPL], fx[CPL];
    int slot = 0;
    { const f32x4* src = myring + lane; fw[0] = src[0]; fx[0] = src[64]; }
#ifdef PNN_F32_DIAG
    const unsigned long long dq0 = __builtin_amdgcn_s_memtime(), dr0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto chunk = [&](int k, const f32x4* nsrc, int kn, bool rd) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][0], fx[k][0], acc, 0, 0, 0);
        if (rd) fw[kn] = nsrc[0];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][1], fx[k][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][2], fx[k][2], acc, 0, 0, 0);
        if (rd) fx[kn] = nsrc[64];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[k][3], fx[k][3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int st = 0; st + 1 < nst; st++) {
        const int nslot = slot + 1 == D ? 0 : slot + 1;
#pragma unroll
        for (int k = 0; k + 1 < CPL; k++) chunk(k, myring + (slot * CPL + k + 1) * 128 + lane, k + 1, true);
        __builtin_amdgcn_s_barrier();                // stage st + 1 is in the rings, the slot of stage st - 1 may be refilled
        chunk(CPL - 1, myring + (nslot * CPL) * 128 + lane, 0, true);
        slot = nslot;
    }
#pragma unroll
    for (int k = 0; k + 1 < CPL; k++) chunk(k, myring + (slot * CPL + k + 1) * 128 + lane, k + 1, true);
    chunk(CPL - 1, myring, 0, false);
#ifdef PNN_F32_DIAG
    unsigned long long* const dstamp = (p.Xlo && wave == 0) ? (unsigned long long*)p.Xlo + 8 * (by * gridDim.x + bx) : nullptr;
    if (dstamp && lane == 0) {
        dstamp[0] = __builtin_amdgcn_s_memtime() - dq0; dstamp[1] = __builtin_amdgcn_s_memrealtime() - dr0; dstamp[2] = (unsigned long long)(c1 - c0); dstamp[3] = dr0;
        dstamp[4] = de0;
    }
#endif
    // ---- the segment sums meet in LDS; wave 0 adds them in order.  Lane (q, l15): row m = 16 bx + l15, channels n0 + 4 q + r ----
    part[sg * 64 + lane] = acc;
    __builtin_amdgcn_s_barrier();
    if (wave != 0) return;
    f32x4 t = part[lane];
    for (int k = 1; k < p.nseg; k++) t += part[k * 64 + lane];
    const int mg = bx * 16 + l15, n = n0 + 4 * q;
    if (mg < p.M && n < p.Cout) {
        f32x4 v = t + *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
        const size_t o = (size_t)mg * p.Cout + n;
        if (p.Y) store4_chain(p.Y + (size_t)mg * p.Cout + n0, q, v, (p.chain_io & 2) != 0);
        if (p.Yi) *reinterpret_cast<int4*>(p.Yi + o) = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
    }
#ifdef PNN_F32_DIAG
    if (dstamp && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dstamp[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// The same layer with ALL of its operands resident (round 6, second form).  The ring above keeps 10 of a segment's 20 chunks in flight per
// chain and measured 2.7 us for the 20-chunk chains (profiles/r06_fcseg_ab.txt: 324 cycles per chunk against the chain's 152) -- the
// loaders request the second half only as slots come free.  A 1200-deep layer is 75 chunks of 2 KiB per output tile: 76 chunk slots
// (every segment rounded up to whole stages of 4: 20 + 20 + 20 + 16) are 152 KiB, one workgroup per CU.  So: 4 chain waves + EIGHT
// loader waves (two per chain, chunks alternating), every loader issues its 10 chunks -- 50 LDS-DMA instructions, under the 6-bit
// vmcnt -- at once, and the whole layer slice is in flight one memory latency after entry; a stage (4 chunks per chain) is published
// by a barrier once both of its loaders count it landed (vmcnt <= 40, 30, 20, 10, 0).  A chunk past the end of a segment inside a live
// stage (the last segment's 16th) holds zeros (range misses); whole stages past the end are skipped by their chain wave (barriers only).
constexpr int kFcAllCPS = 4, kFcAllStages = kFcSegChunks / kFcAllCPS;            // chunks per stage and chain; stages per full segment
constexpr int kFcAllPerLoader = kFcSegChunks / 2;                                // chunks per loader wave
static_assert(kFcSegChunks % kFcAllCPS == 0 && 5 * kFcAllPerLoader < 64, "whole stages; all of a loader's instructions under one vmcnt");
constexpr int kFcAllMaxSlots = 76;
constexpr size_t kFcAllLds = ((size_t)kFcAllMaxSlots * 128 + kFcSegNS * 64) * 16;

template <bool XCH>
__global__ __launch_bounds__(768) void fcseg_f32_small_all_kernel(const F32SmallArgs args)
{
    touch_kernargs<sizeof(F32SmallArgs)>();
    (void)args;
    const auto* ka = (const __attribute__((address_space(4))) F32SmallArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    CF32SmallParams& p = ka->p;
    constexpr int NS = kFcSegNS, CPS = kFcAllCPS, NST = kFcAllStages, PL = kFcAllPerLoader;
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [76 chunk slots][weights 64 pieces | activations 64 pieces] | [NS][64] segment sums
#ifdef PNN_F32_DIAG
    const unsigned long long de0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int l15 = lane & 15, q = lane >> 4;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int n0 = by * 16;
    const int cpt = p.Cin >> 4, segc = kFcSegChunks;
    // waves 0-3: the chains; waves 4-11: loader (wave - 4) >> 1 ... no: chain sg = (wave - 4) & 3, half = (wave - 4) >> 2
    const int sg = wave < NS ? wave : (wave - NS) & (NS - 1);
    int c0 = sg * segc, c1 = c0 + segc;
    if (c0 > cpt) c0 = cpt;
    if (c1 > cpt) c1 = cpt;
    if (sg >= p.nseg) c1 = c0;
    // this chain's chunk slots: the segments' stage-rounded lengths, one after the other
    int slot0 = 0;
    for (int k = 0; k < sg; k++) {
        int a = k * segc, b = a + segc;
        if (a > cpt) a = cpt;
        if (b > cpt) b = cpt;
        if (k >= p.nseg) b = a;
        slot0 += (b - a + CPS - 1) / CPS * CPS;
    }
    const int nlive = c1 - c0, nst = (nlive + CPS - 1) / CPS;      // this chain's live stages
    f32x4* const mine = ring + slot0 * 128;
    f32x4* const part = ring + kFcAllMaxSlots * 128;

    if (wave >= NS) {
        const int half = (wave - NS) >> 2;                          // this loader's chunks of the segment: half, half + 2, ...
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, 0x7fffffffu, 0x00020000);
        const unsigned bstride = (unsigned)(4 * p.Npad) << 4;
        const unsigned wlane = (unsigned)((q * p.Npad + n0 + l15) << 4);
        constexpr unsigned kOob = 0x80000000u;
        const int mx = bx * 16 + (XCH ? (lane & 15) : (lane >> 2));
        const unsigned apix = mx < p.M ? (((unsigned)mx * (unsigned)p.Cin) << 2) + (XCH ? (unsigned)((lane >> 4) << 4) : (unsigned)((lane & 3) << 3)) : kOob;
        const int nslots = nst * CPS;                               // slots this chain owns (live stages)
#pragma unroll
        for (int i = 0; i < PL; i++) {
            const int j = 2 * i + half;                            // chunk of the segment
            const int ci = c0 + j;
            const bool live = ci < c1;
            // a chunk past the segment inside a live stage: zeros into its slot (range miss).  Past the live stages there is nothing to
            // fill, but the instructions still go out -- every loader issues exactly PL chunks, the vmcnt thresholds below count on it --
            // as range misses into the segment-sum area, which nobody reads or writes before all of them have landed (vmcnt 0, last barrier)
            f32x4* dst = j < nslots ? mine + j * 128 : part;
            const unsigned wo = live ? wlane + (unsigned)ci * bstride : kOob;
            const unsigned ao = live ? apix : kOob;
            const unsigned so = (unsigned)(ci << 6);
            f32s_dma16(wrsrc, wo, dst);
            float* xd = reinterpret_cast<float*>(dst + 64);
            if (XCH) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd), 16, ao, so, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd), 4, ao, so, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 64), 4, ao, so + 32u, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 128), 4, ao, so + 4u, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(xd + 192), 4, ao, so + 36u, 0, 0);
            }
        }
        // stage t is whole when both loaders have seen their chunks 2t and 2t + 1 land (loads complete in issue order)
        constexpr int IPC = XCH ? 2 : 5;             // vector-memory instructions per chunk
        f32s_wait_vm<IPC * (PL - 2)>(); __builtin_amdgcn_s_barrier();
        f32s_wait_vm<IPC * (PL - 4)>(); __builtin_amdgcn_s_barrier();
        f32s_wait_vm<IPC * (PL - 6)>(); __builtin_amdgcn_s_barrier();
        f32s_wait_vm<IPC * (PL - 8)>(); __builtin_amdgcn_s_barrier();
        f32s_wait_vm<0>(); __builtin_amdgcn_s_barrier();
        static_assert(NST == 5 && PL == 10, "the five waits above");
        __builtin_amdgcn_s_barrier();                // the segment sums' barrier
        return;
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_setprio(3);
#ifdef PNN_F32_DIAG
    unsigned long long dq0 = 0, dr0 = 0;
#endif
    // per chunk two 16-byte LDS reads and four MFMAs; the next chunk's reads sit between this chunk's MFMAs (inside a stage); the first
    // chunk of a stage is read behind the stage's barrier
    auto chunk = [&](const f32x4 fw, const f32x4 fx, const f32x4* nsrc, f32x4& nw, f32x4& nx, bool rd) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[0], fx[0], acc, 0, 0, 0);
        if (rd) nw = nsrc[0];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[1], fx[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[2], fx[2], acc, 0, 0, 0);
        if (rd) nx = nsrc[64];
        __builtin_amdgcn_sched_barrier(0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[3], fx[3], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int t = 0; t < NST; t++) {
        __builtin_amdgcn_s_barrier();                // stage t of every chain has landed
        if (t >= nst) continue;                      // past this chain's segment (wave-uniform): the barriers only
#ifdef PNN_F32_DIAG
        if (t == 0) { dq0 = __builtin_amdgcn_s_memtime(); dr0 = __builtin_amdgcn_s_memrealtime(); }
#endif
        const f32x4* src = mine + (t * CPS) * 128 + lane;
        f32x4 w0 = src[0], x0 = src[64], w1, x1;
        chunk(w0, x0, src + 128, w1, x1, true);
        chunk(w1, x1, src + 256, w0, x0, true);
        chunk(w0, x0, src + 384, w1, x1, true);
        chunk(w1, x1, src, w0, x0, false);
        static_assert(CPS == 4, "four chunks per stage, unrolled");
    }
#ifdef PNN_F32_DIAG
    unsigned long long* const dstamp = (p.Xlo && wave == 0) ? (unsigned long long*)p.Xlo + 8 * (by * gridDim.x + bx) : nullptr;
    if (dstamp && lane == 0) {
        dstamp[0] = __builtin_amdgcn_s_memtime() - dq0; dstamp[1] = __builtin_amdgcn_s_memrealtime() - dr0; dstamp[2] = (unsigned long long)(c1 - c0); dstamp[3] = dr0;
        dstamp[4] = de0;
    }
#endif
    part[sg * 64 + lane] = acc;
    __builtin_amdgcn_s_barrier();
    if (wave != 0) return;
    f32x4 tsum = part[lane];
    for (int k = 1; k < p.nseg; k++) tsum += part[k * 64 + lane];
    const int mg = bx * 16 + l15, n = n0 + 4 * q;
    if (mg < p.M && n < p.Cout) {
        f32x4 v = tsum + *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
        const size_t o = (size_t)mg * p.Cout + n;
        if (p.Y) store4_chain(p.Y + (size_t)mg * p.Cout + n0, q, v, (p.chain_io & 2) != 0);
        if (p.Yi) *reinterpret_cast<int4*>(p.Yi + o) = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
    }
#ifdef PNN_F32_DIAG
    if (dstamp && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dstamp[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

// does the resident form hold this layer?  (76 chunk slots: K <= 1216 in segments of kFcSegChunks)
static bool fcseg_all_fits(const TapGemmParams& p)
{
    if ((int)p.seg_chunks != kFcSegChunks || p.nseg > kFcSegNS) return false;
    const int cpt = p.Cin >> 4;
    int slots = 0;
    for (int k = 0; k < p.nseg; k++) {
        const int a = std::min(k * kFcSegChunks, cpt), b = std::min(a + kFcSegChunks, cpt);
        slots += (b - a + kFcAllCPS - 1) / kFcAllCPS * kFcAllCPS;
    }
    return slots <= kFcAllMaxSlots;
}

long fcseg_f32_small_tiles(const TapGemmParams& p) { return (long)((p.M + 15) / 16) * ((p.Cout + 15) / 16); }
bool fcseg_f32_small_fits(const TapGemmParams& p)
{
    return p.ncls == 1 && p.SH * p.SW == 1 && p.tap_begin[1] == 1 && p.nseg > 1 && p.nseg <= kFcSegNS && p.seg_chunks > 0 && p.Cout % 4 == 0 && p.M > 0;
}

hipError_t launch_fcseg_f32_small(const TapGemmParams& p, hipStream_t s)
{
    if (!fcseg_f32_small_fits(p)) return hipErrorInvalidValue;
    static int done[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const int di = dev >= 0 && dev < 16 ? dev : 0;
    if (!__atomic_load_n(&done[di], __ATOMIC_ACQUIRE)) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fcseg_f32_small_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFcSegLds)) != hipSuccess) return e;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fcseg_f32_small_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFcSegLds)) != hipSuccess) return e;
        __atomic_store_n(&done[di], 1, __ATOMIC_RELEASE);
    }
    const F32SmallArgs a{p};
    const bool xch = (p.chain_io & 1) != 0;          // the activations are in chain order (see tapgemm_f32_small_body)
    static const bool no_all = getenv("PNN_FCSEG_RING") != nullptr;   // A/B: the ring form for every layer
    if (fcseg_all_fits(p) && !no_all) {
        static int done_all[16] = {};
        if (!__atomic_load_n(&done_all[di], __ATOMIC_ACQUIRE)) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fcseg_f32_small_all_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFcAllLds)) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fcseg_f32_small_all_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFcAllLds)) != hipSuccess) return e;
            __atomic_store_n(&done_all[di], 1, __ATOMIC_RELEASE);
        }
        if (xch) pnn_launch(fcseg_f32_small_all_kernel<true>, dim3((unsigned)((p.M + 15) / 16), (unsigned)((p.Cout + 15) / 16)), dim3(768), kFcAllLds, s, a);
        else pnn_launch(fcseg_f32_small_all_kernel<false>, dim3((unsigned)((p.M + 15) / 16), (unsigned)((p.Cout + 15) / 16)), dim3(768), kFcAllLds, s, a);
        return hipGetLastError();
    }
    if (xch) pnn_launch(fcseg_f32_small_kernel<true>, dim3((unsigned)((p.M + 15) / 16), (unsigned)((p.Cout + 15) / 16)), dim3(512), kFcSegLds, s, a);
    else pnn_launch(fcseg_f32_small_kernel<false>, dim3((unsigned)((p.M + 15) / 16), (unsigned)((p.Cout + 15) / 16)), dim3(512), kFcSegLds, s, a);
    return hipGetLastError();
}

constexpr int kF32SmallLADeep = 12;
static_assert(5 * kF32SmallCPL * kF32SmallLADeep < 64, "vmcnt is a 6-bit counter");
size_t tapgemm_f32_small_lds_bytes(bool deep) { return (size_t)((deep ? kF32SmallLADeep : kF32SmallLA) + 2) * kF32SmallCS * 128 * 16; }

long tapgemm_f32_small_tiles(const TapGemmParams& p)
{
    return (long)((p.M + 15) / 16) * ((p.Cout + 15) / 16) * p.ncls * (p.nseg > 1 ? p.nseg : 1);
}

static hipError_t f32_small_attrs()
{
    static int done[16] = {};                        // per device: the attribute belongs to the function ON a device (several contexts, several
    int dev = 0;                                     // threads: an acquire / release flag each -- setting an attribute twice is harmless)
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const int di = dev >= 0 && dev < 16 ? dev : 0;
    if (__atomic_load_n(&done[di], __ATOMIC_ACQUIRE)) return hipSuccess;
    const void* fns[10] = {reinterpret_cast<const void*>(&tapgemm_f32_small_kernel<kF32SmallLA, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_kernel<kF32SmallLA, true>),
                           reinterpret_cast<const void*>(&tapgemm_f32_small_inline_kernel<kF32SmallLA>),
                           reinterpret_cast<const void*>(&tapgemm_f32_small_pair_kernel<kF32SmallLA, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_pair_kernel<kF32SmallLA, true>),
                           reinterpret_cast<const void*>(&tapgemm_f32_small_kernel<kF32SmallLADeep, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_kernel<kF32SmallLADeep, true>),
                           reinterpret_cast<const void*>(&tapgemm_f32_small_inline_kernel<kF32SmallLADeep>),
                           reinterpret_cast<const void*>(&tapgemm_f32_small_pair_kernel<kF32SmallLADeep, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_pair_kernel<kF32SmallLADeep, true>)};
    for (int i = 0; i < 10; i++)
        if ((e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)tapgemm_f32_small_lds_bytes(i >= 5))) != hipSuccess) return e;
    __atomic_store_n(&done[di], 1, __ATOMIC_RELEASE);
    return hipSuccess;
}

// the deep ring for launches of at most one workgroup per CU (see the kernels' comment).  mode 0: never; 1: the FC layers only (nothing to
// lose alone: 38 us either way; a conv 16x16 call alone 82 -> 87 us with it); 2: every such launch (the batching service's contexts)
static bool f32_small_deep(long tiles, int mode, bool fc) { return mode > 0 && (mode > 1 || fc) && tiles <= (long)device_info().cus; }

hipError_t launch_tapgemm_f32_small(const TapGemmParams& p, hipStream_t s, const float* host_input, int deep_mode)
{
    hipError_t e = f32_small_attrs();
    if (e != hipSuccess) return e;
    if (p.M <= 0) return hipSuccess;
    const dim3 grid((p.M + 15) / 16, (p.Cout + 15) / 16, (unsigned)(p.ncls * (p.nseg > 1 ? p.nseg : 1)));
    const bool deep = f32_small_deep(tapgemm_f32_small_tiles(p), deep_mode, p.SH * p.SW == 1);
    const size_t nin = (size_t)p.M * p.IH * p.IW * p.Cin;
    if (host_input && p.SH * p.SW == 1 && nin <= (size_t)kF32SmallInlineFloats) {
        F32SmallArgsInline a;
        a.p = p;
        memcpy(a.in, host_input, nin * sizeof(float));
        if (deep) pnn_launch(tapgemm_f32_small_inline_kernel<kF32SmallLADeep>, grid, dim3(256), tapgemm_f32_small_lds_bytes(true), s, a);
        else pnn_launch(tapgemm_f32_small_inline_kernel<kF32SmallLA>, grid, dim3(256), tapgemm_f32_small_lds_bytes(false), s, a);
        return hipGetLastError();
    }
    const F32SmallArgs a{p};
    const bool xch = (p.chain_io & 1) != 0;          // the activations are in chain order (see tapgemm_f32_small_body)
    if (deep && xch) pnn_launch(tapgemm_f32_small_kernel<kF32SmallLADeep, true>, grid, dim3(256), tapgemm_f32_small_lds_bytes(true), s, a);
    else if (deep) pnn_launch(tapgemm_f32_small_kernel<kF32SmallLADeep, false>, grid, dim3(256), tapgemm_f32_small_lds_bytes(true), s, a);
    else if (xch) pnn_launch(tapgemm_f32_small_kernel<kF32SmallLA, true>, grid, dim3(256), tapgemm_f32_small_lds_bytes(false), s, a);
    else pnn_launch(tapgemm_f32_small_kernel<kF32SmallLA, false>, grid, dim3(256), tapgemm_f32_small_lds_bytes(false), s, a);
    return hipGetLastError();
}

// May the layer run with a tail?  Whole 16-row tiles per block, one class, no K segments, channel-order output, Cout in whole tiles.
static bool f32_small_tail_layer_ok(const TapGemmParams& p)
{
    const int SP = p.SH * p.SW;
    return SP >= 16 && SP % 16 == 0 && p.ncls == 1 && p.nseg <= 1 && !(p.chain_io & 2) && p.Cout % 16 == 0 && p.Y && !p.Yi && p.M % SP == 0;
}
static hipError_t f32_small_tail_attrs()
{
    static int done[16] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const int di = dev >= 0 && dev < 16 ? dev : 0;
    if (__atomic_load_n(&done[di], __ATOMIC_ACQUIRE)) return hipSuccess;
    const void* fns[8] = {reinterpret_cast<const void*>(&tapgemm_f32_small_tail_kernel<kF32SmallLA, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_tail_kernel<kF32SmallLA, true>),
                          reinterpret_cast<const void*>(&tapgemm_f32_small_pair_tail_kernel<kF32SmallLA, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_pair_tail_kernel<kF32SmallLA, true>),
                          reinterpret_cast<const void*>(&tapgemm_f32_small_tail_kernel<kF32SmallLADeep, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_tail_kernel<kF32SmallLADeep, true>),
                          reinterpret_cast<const void*>(&tapgemm_f32_small_pair_tail_kernel<kF32SmallLADeep, false>), reinterpret_cast<const void*>(&tapgemm_f32_small_pair_tail_kernel<kF32SmallLADeep, true>)};
    for (int i = 0; i < 8; i++)
        if ((e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)tapgemm_f32_small_lds_bytes(i >= 4))) != hipSuccess) return e;
    __atomic_store_n(&done[di], 1, __ATOMIC_RELEASE);
    return hipSuccess;
}

// The last GEMM of the transposed stack + the net's last layer per block (SmallTail kind 2).  hipErrorInvalidValue: not this shape.
bool f32_small_cout1_tail_ok(const TapGemmParams& p, const TConv1Params& t)
{
    if (!f32_small_tail_layer_ok(p) || p.Cout != t.Cin) return false;
    if (t.IH != p.OH || t.IW != p.OW || p.os != 1 || t.B * p.SH * p.SW != p.M) return false;   // the GEMM's output map is the last layer's input
    if (t.Cin == 64) {                                // tconv_cout1_mfma_kernel's form: one band, the whole input map in T
        if (t.s != 2 || t.k != 5 || t.pad > t.k - 1) return false;
        const int npx = t.IH * t.IW;
        return (size_t)((npx + 31) / 32) * 32 * kTcTP * sizeof(float) <= tapgemm_f32_small_lds_bytes(false);
    }
    // tconv_cout1_kernel's form (the 4x4 net: 32 maps, 3x3, stride 1): one output tile, one image per workgroup
    if ((t.s != 1 && t.s != 2) || t.Cin % 4 || t.IH != t.IW || t.IH * t.s > 16 || (t.s == 2 && ((t.IH * t.s) & 1))) return false;
    const int TO = t.IH * t.s, lo = t.pad - (t.k - 1);
    const int i0 = lo >= 0 ? lo / t.s : -((-lo + t.s - 1) / t.s);
    const int TI = (TO - 1 + t.pad) / t.s - i0 + 1;
    return (size_t)(TI * TI + t.k * t.k) * t.Cin * sizeof(float) <= tapgemm_f32_small_lds_bytes(false);
}
hipError_t launch_tapgemm_f32_small_tail(const TapGemmParams& p, const SmallTail& t, hipStream_t s, int deep_mode)
{
    hipError_t e = f32_small_tail_attrs();
    if (e != hipSuccess) return e;
    if (t.kind != 2 || !t.cnt || !f32_small_cout1_tail_ok(p, t.t) || (t.t.done.host_flag && !t.t.done.per_wg)) return hipErrorInvalidValue;
    F32SmallTailArgs a;
    a.p = p; a.t = t;
    a.t.t.ni = t.t.Cin == 64 ? t.t.IH * t.t.s : 1;    // one band of all output rows / one image per workgroup
    const dim3 grid((p.M + 15) / 16, (p.Cout + 15) / 16, 1);
    const bool deep = f32_small_deep(tapgemm_f32_small_tiles(p), deep_mode, false), xch = (p.chain_io & 1) != 0;
    if (deep && xch) pnn_launch(tapgemm_f32_small_tail_kernel<kF32SmallLADeep, true>, grid, dim3(256), tapgemm_f32_small_lds_bytes(true), s, a);
    else if (deep) pnn_launch(tapgemm_f32_small_tail_kernel<kF32SmallLADeep, false>, grid, dim3(256), tapgemm_f32_small_lds_bytes(true), s, a);
    else if (xch) pnn_launch(tapgemm_f32_small_tail_kernel<kF32SmallLA, true>, grid, dim3(256), tapgemm_f32_small_lds_bytes(false), s, a);
    else pnn_launch(tapgemm_f32_small_tail_kernel<kF32SmallLA, false>, grid, dim3(256), tapgemm_f32_small_lds_bytes(false), s, a);
    return hipGetLastError();
}

// The branches' last layers + the merger per (block, channel group) (SmallTail kind 1).
bool f32_small_merger_tail_ok(const TapGemmParams& a, const TapGemmParams& b, const MergerParams& m)
{
    if (!f32_small_tail_layer_ok(a) || !f32_small_tail_layer_ok(b) || a.Cout != b.Cout || a.Cout != m.C || m.split || m.nout != 16) return false;
    if (a.SH * a.SW != m.na || b.SH * b.SW != m.nl || m.na + m.nl != 80 || a.os != 1 || b.os != 1) return false;
    return a.M == m.B * m.na && b.M == m.B * m.nl && sizeof(f32x4) * 3 * 16 * 64 <= tapgemm_f32_small_lds_bytes(false);
}
hipError_t launch_tapgemm_f32_small_pair_tail(const TapGemmParams& a, const TapGemmParams& b, const SmallTail& t, hipStream_t s, int deep_mode)
{
    hipError_t e = f32_small_tail_attrs();
    if (e != hipSuccess) return e;
    if (t.kind != 1 || !t.cnt || !f32_small_merger_tail_ok(a, b, t.m) || (a.chain_io & 1) != (b.chain_io & 1)) return hipErrorInvalidValue;
    F32SmallTail2Args args;
    args.a = a; args.b = b; args.t = t;
    args.na = (int)tapgemm_f32_small_tiles(a);
    const long total = args.na + tapgemm_f32_small_tiles(b);
    const bool xch = (a.chain_io & 1) != 0, deep = f32_small_deep(total, deep_mode, false);
    if (deep && xch) pnn_launch(tapgemm_f32_small_pair_tail_kernel<kF32SmallLADeep, true>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(true), s, args);
    else if (deep) pnn_launch(tapgemm_f32_small_pair_tail_kernel<kF32SmallLADeep, false>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(true), s, args);
    else if (xch) pnn_launch(tapgemm_f32_small_pair_tail_kernel<kF32SmallLA, true>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(false), s, args);
    else pnn_launch(tapgemm_f32_small_pair_tail_kernel<kF32SmallLA, false>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(false), s, args);
    return hipGetLastError();
}

hipError_t launch_tapgemm_f32_small_pair(const TapGemmParams& a, const TapGemmParams& b, hipStream_t s, int deep_mode)
{
    if (a.M <= 0 || b.M <= 0) return hipErrorInvalidValue;
    const hipError_t e = f32_small_attrs();
    if (e != hipSuccess) return e;
    F32SmallArgs2 args;
    args.a = a; args.b = b;
    args.na = (int)tapgemm_f32_small_tiles(a);
    const long total = args.na + tapgemm_f32_small_tiles(b);
    if ((a.chain_io & 1) != (b.chain_io & 1)) return hipErrorInvalidValue;   // one instantiation per launch: both layers' inputs in the same order
    const bool xch = (a.chain_io & 1) != 0, deep = f32_small_deep(total, deep_mode, false);
    if (deep && xch) pnn_launch(tapgemm_f32_small_pair_kernel<kF32SmallLADeep, true>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(true), s, args);
    else if (deep) pnn_launch(tapgemm_f32_small_pair_kernel<kF32SmallLADeep, false>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(true), s, args);
    else if (xch) pnn_launch(tapgemm_f32_small_pair_kernel<kF32SmallLA, true>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(false), s, args);
    else pnn_launch(tapgemm_f32_small_pair_kernel<kF32SmallLA, false>, dim3((unsigned)total), dim3(256), tapgemm_f32_small_lds_bytes(false), s, args);
    return hipGetLastError();
}

}  // namespace pnn
