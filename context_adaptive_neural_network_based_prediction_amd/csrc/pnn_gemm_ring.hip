// Split-precision tap GEMM with dedicated loader waves and an LDS-DMA ring.
//
// Measured on tapgemm_sp_kernel and on a first ring version of this file (tools/ring_prof.hip): a CU's vector-memory
// path accepts 64 B/clk, a wave that issues a 1-KiB load is BLOCKED until the path takes it (in-order issue), and with
// four waves per CU the ~600 cycles a stage's loads need are therefore added to its ~960 MFMA cycles instead of
// running under them -- however far ahead the loads are requested.  So the two jobs get separate waves:
//
//   workgroup = 512 threads = 8 waves, two per SIMD: waves 0-3 ("MFMA waves", WM x WN, wave tile (32*RT) x (32*NT))
//   only read fragments from LDS and issue MFMAs; waves 4-7 ("loader waves") only issue global_load_lds_dwordx4 into a
//   D-deep ring of stage buffers (no VGPR destination, no ds_write) and wait for them with counted vmcnt.
//   One raw s_barrier per stage joins them: before barrier s the loaders have seen stage s+1 land (D-3 later stages
//   may stay in flight); after it the MFMA waves may read stage s+1 (they prefetch its first fragments under the
//   last chunk of stage s) and the loaders refill buffer (s-1) % D, which nobody reads any more.
//   BM = 32*RT*WM, BN = 32*NT*WN; stage = KC 16-deep chunks: activations [BM rows][PPR = 4*KC 16-byte pieces] (hi|lo
//   halves of each chunk), a row's pieces XOR-swizzled so that the fragment reads (ds_read_b128, 16-lane groups) are
//   conflict-free -- an LDS-DMA writes lane-linear, so the permutation is applied to the per-lane SOURCE address and
//   again by the reader; weights [KC][4 planes][BN] pieces, read lane-contiguous.
//   Out-of-image taps, rows past M and chunks past the end of a one-tap layer's K are fetched from a zero page.
// Same contract, parameter block, weight packing and per-output summation order as tapgemm_sp_kernel: bit-identical.
//
// What bounds it now (tools/ring_prof.hip, -DPNN_RING_DIAG3, FC 1200x1200, tile 128x160): a loader wave needs ~1300
// cycles to ISSUE its 9 LDS-DMA instructions of a stage while the MFMA waves run (~800 with the MFMAs ablated: the DMA
// writes and the fragment reads share the LDS), i.e. 28 B/clk of the CU's 64 B/clk vector-memory path; the MFMA waves
// finish their 960 MFMA cycles in ~1070 and wait ~300-400 at the stage barrier.  Deeper rings do not help (D = 5 / 6
// measured equal to D = 4 on the tiles where they fit): it is issue throughput, not latency.
//
// Later findings: most of that issue cost was the 64-bit flat address per lane -- the loaders now go
// through buffer descriptors (blds16 below) and issue a stage in ~640 cycles; the 128x160 stage takes 1097 cycles.  With
// that, EIGHT loader waves on the small tiles (64x128, 128x64: few registers, three waves per SIMD fit) gain only 2-6 %
// (stage 653 -> 639, 725 -> 677 cycles for 384 of MFMA work): the CU takes ~38 B/clk of LDS-DMA however many waves ask.
//
// Round 2, epilogue (tools/ring_prof.hip stamps, FC 1200x1200 at batch 4096): the workgroup's bias tile is fetched into LDS
// behind the ring by loader wave 0's first instruction (the 20 global bias loads used to sit between the last MFMA and the
// first store: 1.2k cycles -> 40); the hi / lo conversion is v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 (pnn_device_common.h);
// the tile leaves through L2 (store16_through: slower inside the workgroup, 2.8k -> 3.7k cycles of issue, but the next launch
// no longer waits for ~20 MB of dirty lines: kernel-to-kernel 39.1 -> 37.5 us).  The workgroups of a launch end 2.4-3.8 us
// apart although they execute the same cycle count: the spread is by XCD (each holds its own clock under the power limit).
#include "pnn_kernels.h"
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <queue>
#include <type_traits>
#include <vector>
#include "pnn_device_common.h"

namespace pnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* g, f32x4* l)
{
#ifdef PNN_RING_NO_LOAD        // ablation builds of tools/ring_prof.hip only
    asm volatile("" ::"v"(g), "s"(l));
    return;
#endif
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// The same through a buffer descriptor: 32-bit per-lane offset (half the address traffic of a 64-bit flat address), and an
// offset past the descriptor's size delivers zeros -- padding rows, out-of-image taps and the K tail need no zero page.
__device__ __forceinline__ void blds16(const __amdgpu_buffer_rsrc_t& r, unsigned off, f32x4* l)
{
#ifdef PNN_RING_NO_LOAD
    asm volatile("" ::"v"(off), "s"(l));
    return;
#endif
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, off, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most `stages` stages of this wave's LDS-DMA are outstanding; a wave issues HI or HI - 1 instructions per
// stage (`fewer`, wave-uniform)
template <int STAGES, int HI>
__device__ __forceinline__ void wait_stages(bool fewer)
{
    if (fewer) wait_vm<STAGES * (HI - 1)>(); else wait_vm<STAGES * HI>();
}

// One layer's tile on this workgroup: the whole kernel body, callable several times in a row .
// The parameter block is read through the CONSTANT address space (the kernel-argument segment, or a device buffer that
// does not change during the launch): every field, also the runtime-indexed tap tables, is then a scalar load; through a
// generic pointer the fields of a layer chosen at run time end up in vector registers (256 VGPRs + 77 spilled).
typedef const __attribute__((address_space(4))) TapGemmParams CParams;
template <int RT, int NT, int KC, int WM, int D, bool FUSE>
__device__ __forceinline__ void ring_layer(CParams& p, const int tid)
{
    constexpr int WN = 4 / WM;
    constexpr int BM = 32 * RT * WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int E = 4 * BN;                       // 16-byte pieces per staged weight chunk
    constexpr int PPR = KC * 4;                     // pieces per activation row and stage
    constexpr int NLA = BM * PPR / 256;             // activation LDS-DMA instructions per wave and stage
    constexpr int NBI = KC * E / 64;                // weight LDS-DMA instructions per stage, dealt round-robin to the 4 loader waves
    constexpr int NLB = (NBI + 3) / 4;              // ... per wave: NLB for waves < NBX, NLB - 1 for the others (NBX = 0: all NLB)
    constexpr int NBX = NBI % 4;
    constexpr int NI = NLA + NLB;                   // upper bound of a wave's instructions per stage
    constexpr int ASLOTS = BM * PPR, SS = ASLOTS + KC * E;
    constexpr int RPS = 16 / PPR > 0 ? 16 / PPR : 1;   // rows per swizzle step
    static_assert((KC * E) % 64 == 0 && (BM * PPR) % 256 == 0, "stage pieces must split into whole wave instructions");
    static_assert((D - 2) * NI <= 63, "vmcnt range");
    constexpr bool kBiasLds = BN <= 256 && (size_t)(D * SS + 64) * 16 <= 160 * 1024;   // room for the bias tile (64 pieces) behind the ring
    extern __shared__ __attribute__((aligned(16))) f32x4 ring[];   // [D][A pieces | B pieces]

#ifdef PNN_RING_DIAG2           // coarse stamps (tools/ring_prof.hip): entry / loop begin / loop end / exit of wave 0
    const unsigned long long dq0 = __builtin_amdgcn_s_memtime(), dr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = tid & 63;
    const int wave = (tid >> 6) & 3;                 // index within the role (MFMA waves 0-3, loader waves 4-7)
    const bool loader = tid >= 256;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int SP = p.SH * p.SW;
    const int cpt = p.Cin >> 4;
    const int t0 = p.tap_begin[cls], t1 = p.tap_begin[cls + 1];
    // Position-major tiles (p.pm_groups > 0, convolutions at batch): the tile's BM rows are BM blocks at ONE position (pmi,
    // pmj) of the SH x SW grid, first block mblk.  A tap then lies inside the image for every row or for none, and the taps
    // that only meet SAME padding (31 % of a 3x3 layer's on a 4x4 map) are skipped -- no loads, no MFMAs.  Their products
    // are exact zeros, so every output keeps its bits.  tmask = this class's taps that stay (all of them otherwise).
    const int pmg = p.pm_groups;
    int pmi = 0, pmj = 0;
    int mblk = blockIdx.x * BM;
    unsigned tmask = t1 - t0 >= 32 ? 0xffffffffu : (1u << (t1 - t0)) - 1u;
    if (pmg) {
        // launch order: chunks of 8 block groups; within a chunk position rank by position rank, the 8 groups side by side --
        // workgroups i, i + 8, ... run on one XCD, so each XCD walks the positions of ONE group at a time and its L2 keeps that
        // group's input maps while the taps re-read them (all groups interleaved: 8x8 conv net at batch 4096 5 % slower)
        const int gc = blockIdx.x / (SP * 8), r = blockIdx.x - gc * (SP * 8);
        const int c = gc < (pmg >> 3) ? 8 : (pmg & 7);
        const int pr = r / c;
        const int pos = SP <= 64 ? (int)((p.pos_order[pr >> 2] >> ((pr & 3) * 8)) & 0xffu) : pr;
        pmi = pos / p.SW;
        pmj = pos - pmi * p.SW;
        mblk = (gc * 8 + r - pr * c) * BM;
        unsigned m = 0;
        for (int t = t0; t < t1; t++) {
            const int tp = p.tap[t];
            const int iy = pmi * p.a + (tp >> 16), ix = pmj * p.a + (int)(short)(tp & 0xffff);
            if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) m |= 1u << (t - t0);
        }
        tmask = m ? m : 1u;                          // no tap inside: one of them fetches zeros
    }
    const int nchunks = __builtin_popcount(tmask) * cpt;
    const int nstages = (nchunks + KC - 1) / KC;
    const char* __restrict__ Xb = reinterpret_cast<const char*>(p.X);
    const char* __restrict__ Zb = reinterpret_cast<const char*>(p.zero);
    const f32x4* __restrict__ Wg = reinterpret_cast<const f32x4*>(p.Wp) + (size_t)p.chunk_begin[cls] * 4 * p.Npad;

    // ---- epilogue, second half (all 8 waves): split-f16 output tile LDS -> global in 16-byte pieces, whole rows ---------
    // The MFMA waves leave the tile in LDS as [BM rows][BN/16 chunks][hi 16 x f16 | lo 16 x f16] (the global layout of a
    // row segment, row pitch OP bytes); consecutive lanes then store consecutive pieces: full lines instead of the
    // 8-byte fragments a lane of the accumulator layout owns (measured: 21k -> see DESIGN.md cycles per workgroup).
    constexpr int OPP = BN / 4 + 1;                  // out-tile row pitch in 16-byte pieces (+1: spreads the rows over the banks)
    static_assert((size_t)BM * OPP <= (size_t)D * SS, "output tile must fit the ring");
    // FUSE: the NEXT fully-connected layer (at most 64 outputs, no activation) is applied to this workgroup's activated
    // output tile while it sits in LDS: partial[tile_n][m][0..63] = tile[m][BN columns] x W2[those BN rows][64], the
    // column tiles' partials are summed (+ bias, HM epilogue) by fuse_reduce_kernel.  W2's BN/16 packed chunks
    // ([chunk][4 planes][64 columns] pieces) are fetched behind the output tile by the loader waves.
    constexpr int W2OFF = BM * OPP, W2CH = BN / 16, W2PCS = W2CH * 4 * 64;
    static_assert(!FUSE || (WM == 4 && (size_t)W2OFF + W2PCS <= (size_t)D * SS && W2PCS % 256 == 0), "fused layer needs WM = 4 and room behind the tile");
    const bool fuse = FUSE && p.W2p != nullptr;      // FUSE instantiations also run plain layers
    // (Handing the tile to the loader waves one 32-column block at a time, so that the copy-out runs under the conversion,
    // was built and measured: epilogue 8.8k -> 7.9k cycles in tools/ring_prof.hip, but FC-8 and conv-16 passes 1 % SLOWER
    // -- five more workgroup-wide barriers and a third copy of the unrolled group loop in the instruction cache.)
    const bool lin_out = p.ncls == 1 && p.os == 1 && p.OH == p.SH && p.OW == p.SW;   // launch-uniform
    auto copy_out = [&]() {
        if (!p.Yhi) return;
        const int cpy = p.py[cls], cpx = p.px[cls];
        f32x4* __restrict__ yo = reinterpret_cast<f32x4*>(p.Yhi);
        // two pieces per thread in flight (the store is an asm statement: the compiler neither unrolls across it nor moves the
        // LDS reads over it)
        for (int i0 = tid; i0 < BM * (BN / 4); i0 += 2 * 512) {
            f32x4 v[2];
            f32x4* dst[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * 512;
                const int ic = i < BM * (BN / 4) ? i : tid;
                const int row = ic / (BN / 4), q = ic - row * (BN / 4);
                const int mg = mblk + row;          // position-major: the BLOCK
                const int nq = (n0 >> 2) + q;         // piece index within the output pixel's [Cout/4] pieces
                size_t opix = mg;
                bool rowok = mg < p.M;
                if (pmg) {
                    rowok = mg < p.nblk;
                    opix = ((size_t)mg * p.OH + pmi * p.os + cpy) * p.OW + pmj * p.os + cpx;
                } else if (!lin_out && (SP != 1 || p.os != 1)) {   // (a forward convolution's output grid is its row grid: pixel = row, no divisions)
                    const int pbq = mg / SP;
                    const int rq = mg - pbq * SP;
                    const int piq = rq / p.SW, pjq = rq - piq * p.SW;
                    opix = ((size_t)pbq * p.OH + piq * p.os + cpy) * p.OW + pjq * p.os + cpx;
                }
                v[u] = ring[row * OPP + q];
                dst[u] = (i < BM * (BN / 4) && rowok && nq < (p.Cout >> 2)) ? yo + (opix * (p.Cout >> 2) + nq) : nullptr;
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (dst[u]) store16_through(dst[u], v[u]);
        }
    };

    if (loader) {
    // ---- loader side: this lane's pieces of a stage -------------------------------------------------------------
        // activation instruction r of this wave covers ring pieces [64*(wave + 4r), +64): piece L -> row L / PPR, slot
        // L % PPR, which holds source piece (slot ^ swizzle(row)).
        int lb[NLA], li[NLA], lj[NLA], lpiece[NLA];
        bool lv[NLA];
#pragma unroll
        for (int r = 0; r < NLA; r++) {
            const int L = 64 * (wave + 4 * r) + lane;
            const int row = L / PPR, slot = L - row * PPR;
            lpiece[r] = slot ^ ((row / RPS) % PPR);
            const int mg = mblk + row;
            lv[r] = pmg ? mg < p.nblk : mg < p.M;
            const int mc = lv[r] ? mg : 0;
            if (SP == 1) {                            // fully-connected layer (wave-uniform): no divisions on the start-up path
                lb[r] = mc; li[r] = 0; lj[r] = 0;
            } else if (pmg) {
                lb[r] = mc; li[r] = pmi; lj[r] = pmj;
            } else {
                const unsigned b = (unsigned)mc / (unsigned)SP;
                const unsigned q = (unsigned)mc - b * (unsigned)SP;
                lb[r] = (int)b;
                li[r] = (int)(q / (unsigned)p.SW);
                lj[r] = (int)(q - (unsigned)li[r] * (unsigned)p.SW);
            }
        }
        // byte offset of this lane's piece for chunk 0 of the issue-side tap in the activation buffer; bit 31 set = no source
        // (row past M, tap outside the image): beyond the descriptor's size, the DMA writes zeros
        unsigned aoff[NLA];
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Xb, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0xffffffffu, 0x00020000);
        auto tap_setup = [&](int tp) {
            const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
#pragma unroll
            for (int r = 0; r < NLA; r++) {
                const int iy = li[r] * p.a + dy, ix = lj[r] * p.a + dx;
                const bool ok = lv[r] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
                const unsigned pix = (unsigned)((lb[r] * p.IH + iy) * p.IW + ix);
                aoff[r] = ok ? ((pix * (unsigned)p.Cin) << 2) + (unsigned)(lpiece[r] << 4) : 0x80000000u;
            }
        };
        // this workgroup's BN bias values, behind the ring (one instruction of loader wave 0, the oldest of its queue: landed
        // long before the epilogue; columns past Cout read as zeros).  Fetched by the MFMA waves at the end of the tap loop
        // they were 20 global loads whose latency (~1k cycles) sat between the last MFMA and the first store.
        if (kBiasLds && wave == 0) {
            const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (unsigned)p.Cout * 4u, 0x00020000);
            blds16(brsrc, (unsigned)(n0 + 4 * lane) << 2, ring + D * SS);
        }
        unsigned boff[NLB];
#pragma unroll
        for (int r = 0; r < NLB; r++) {
            const int L = 64 * (wave + 4 * r) + lane;
            const int j = L / E, e = L - j * E;
            const int qq = e / BN, nn = e - qq * BN;
            boff[r] = (unsigned)(((j * 4 + qq) * p.Npad + n0 + nn) << 4);
        }
        const unsigned bstride = (unsigned)(KC * 4 * p.Npad) << 4;   // bytes per stage in the packed weights
        // issue-side position: tap `it` (the lowest of tmask first), chunk icc within it, ring stage istage, stage wst of the
        // class's packed weights (== istage unless taps are skipped); rem = the taps after `it`
        unsigned rem = tmask;
        int it = t0 + __builtin_ctz(rem), icc = 0, istage = 0;
        rem &= rem - 1;
        int wst = (it - t0) * (cpt / KC);
        int tp_next = p.tap[rem ? t0 + __builtin_ctz(rem) : it];
        tap_setup(p.tap[it]);
        auto issue = [&]() {                             // fetch stage `istage` into ring buffer istage % D, then advance
            f32x4* dst = ring + (istage % D) * SS;
#pragma unroll
            for (int r = 0; r < NLA; r++) {
                // a one-tap layer's last stage may run past K (the weights there are zero padding): take zeros, not the next row
                const bool past = icc + (lpiece[r] >> 2) >= cpt;
                blds16(xrsrc, (aoff[r] | (past ? 0x80000000u : 0u)) + ((unsigned)icc << 6), dst + 64 * (wave + 4 * r));
            }
#pragma unroll
            for (int r = 0; r < NLB; r++)
                if (wave + 4 * r < NBI) blds16(wrsrc, boff[r] + (unsigned)wst * bstride, dst + ASLOTS + 64 * (wave + 4 * r));   // wave-uniform
            ++istage;
            ++wst;
            icc += KC;
            if (icc >= cpt && rem) {                     // wave-uniform
                icc = 0;
                it = t0 + __builtin_ctz(rem);
                rem &= rem - 1;
                wst = (it - t0) * (cpt / KC);
                tap_setup(tp_next);
                tp_next = p.tap[rem ? t0 + __builtin_ctz(rem) : it];
            }
        };

        // ---- loader pipeline ----
#ifdef PNN_RING_DIAG3
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long dl1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        // stages 0 and 1 only: the MFMA waves start as soon as stage 0 has landed (they need stage 1 at barrier 0); the
        // remaining D-3 stages of lookahead are requested behind the first barrier, off their critical path
        issue();
        if (1 < nstages) issue();
#ifdef PNN_RING_DIAG3
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long dl2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        const bool fewer = NBX != 0 && wave >= NBX;
        if (1 < nstages) wait_stages<1, NI>(fewer); else wait_vm<0>();
#ifdef PNN_RING_DIAG3
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long dl3 = __builtin_amdgcn_s_memtime();
        if (p.Xlo && tid == 256) {
            unsigned long long* e = (unsigned long long*)p.Xlo + (1 << 19) + 8 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
            e[0] = dq0; e[1] = dl1 - dq0; e[2] = dl2 - dl1; e[3] = dl3 - dl2;
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        __builtin_amdgcn_s_barrier();                // stage 0 is visible
#pragma unroll
        for (int s = 2; s < D - 1; s++)
            if (s < nstages) issue();
#ifdef PNN_RING_DIAG3
        unsigned long long dlw = 0, dlb = 0, dli = 0, dlt = __builtin_amdgcn_s_memtime();
#define DL_STAMP(acc_) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - dlt; dlt = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DL_STAMP(acc_) do {} while (0)
#endif
        for (int s = 0; s < nstages; s++) {
            if (s + 1 < nstages) {                   // stage s+1 must have landed; later stages may stay in flight
                if (s + D - 2 < nstages) wait_stages<D - 3, NI>(fewer); else wait_vm<0>();
            }
            DL_STAMP(dlw);
            __builtin_amdgcn_s_barrier();            // barrier s
            DL_STAMP(dlb);
            if (s + D - 1 < nstages) issue();        // into buffer (s-1) % D
            DL_STAMP(dli);
        }
#ifdef PNN_RING_DIAG3
        if (p.Xlo && tid == 256) {
            unsigned long long* e = (unsigned long long*)p.Xlo + (1 << 19) + 8 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
            e[4] = dlw; e[5] = dlb; e[6] = dli;
        }
#endif
        __builtin_amdgcn_s_barrier();                // epilogue barrier A (see below)
        if (fuse) {
            const f32x4* __restrict__ W2 = reinterpret_cast<const f32x4*>(p.W2p);
#pragma unroll
            for (int r = 0; r < W2PCS / 256; r++) {
                const int L = 64 * (wave + 4 * r) + lane;          // piece L = (chunk c, plane q, column n)
                const int c2 = L >> 8, q2 = (L >> 6) & 3, n2 = L & 63;
                const int kc2 = (n0 >> 4) + c2;                     // chunk of the next layer's K = our output columns
                const void* src = kc2 < p.K2chunks ? (const void*)(W2 + ((size_t)kc2 * 4 + q2) * p.Npad2 + n2) : (const void*)(Zb + ((L & 15) << 4));
                glds16(src, ring + W2OFF + 64 * (wave + 4 * r));
            }
            wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();                // epilogue barrier B
        copy_out();
        return;
    }

    // ---- consumer side ------------------------------------------------------------------------------------------------
    f32x16 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[rt][nt][i] = 0.f;
    int arow[RT], aswz[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int row = wm * (32 * RT) + rt * 32 + l31;
        arow[rt] = row * PPR;
        aswz[rt] = (row / RPS) % PPR;
    }
    auto read_frags = [&](const f32x4* buf, int j, f32x4 (&wf)[NT][2], f32x4 (&af)[RT][2]) {   // [..][0] = hi, [..][1] = lo
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            wf[nt][0] = buf[ASLOTS + j * E + (0 + h) * BN + wn * (32 * NT) + nt * 32 + l31];
            wf[nt][1] = buf[ASLOTS + j * E + (2 + h) * BN + wn * (32 * NT) + nt * 32 + l31];
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            af[rt][0] = buf[arow[rt] + ((j * 4 + 0 + h) ^ aswz[rt])];
            af[rt][1] = buf[arow[rt] + ((j * 4 + 2 + h) ^ aswz[rt])];
        }
    };
    auto mfma_chunk = [&](const f32x4 (&wf)[NT][2], const f32x4 (&a)[RT][2]) {
#pragma unroll
        for (int part = 0; part < 2; part++)        // part 0: hi*hi for every tile;  part 1: the two cross terms
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++) {
                    const f16x8 whi = __builtin_bit_cast(f16x8, wf[nt][0]), wlo = __builtin_bit_cast(f16x8, wf[nt][1]);
                    const f16x8 ahi = __builtin_bit_cast(f16x8, a[rt][0]), alo = __builtin_bit_cast(f16x8, a[rt][1]);
#ifdef PNN_RING_NO_MFMA        // ablation builds of tools/ring_prof.hip only: keep the fragment reads alive, skip the matrix work
                    asm volatile("" ::"v"(whi), "v"(wlo), "v"(ahi), "v"(alo));
                    continue;
#endif
                    if (part == 0) {
                        acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi, acc[rt][nt], 0, 0, 0);
                    } else {
                        acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo, acc[rt][nt], 0, 0, 0);
                        acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi, acc[rt][nt], 0, 0, 0);
                    }
                }
    };

#ifdef PNN_RING_DIAG
    // Diagnostic build only (tools/ring_prof.hip): cycle sums of wave 0 per phase, written to p.Xlo.
    unsigned long long dg_t = 0, dg_a = 0, dg_w = 0, dg_i = 0, dg_b = 0;
#define DG_STAMP(acc_)                                                                    \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        unsigned long long now_;                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory");   \
        acc_ += now_ - dg_t; dg_t = now_;                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#else
#define DG_STAMP(acc_) do {} while (0)
#endif
    // Issue order inside a chunk: one MFMA, then two of the NEXT chunk's fragment reads, ... -- two ds_read_b128 fit in the
    // shadow of a 32-cycle MFMA; issued as a burst of 2*(RT+NT) they hold the wave (and its MFMA pipe) for ~16 cycles each.
    auto interleave = [&]() {
#ifndef PNN_RING_DIAG
#pragma unroll
        for (int i = 0; i < RT + NT; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * RT * NT - (RT + NT), 0);
#endif
    };
    // ---- MFMA-wave pipeline -----------------------------------------------------------------------------------------
    static_assert((KC % 2 == 0 || KC == 1) && D >= 3, "fragment register sets alternate per chunk");
    __builtin_amdgcn_s_barrier();                    // stage 0 is visible
    __builtin_amdgcn_sched_barrier(0);
    f32x4 wf0[NT][2], wf1[NT][2], af0[RT][2], af1[RT][2];
    read_frags(ring, 0, wf0, af0);
#ifdef PNN_RING_DIAG
    { unsigned long long d0_ = 0; DG_STAMP(d0_); (void)d0_; }
#endif
#ifdef PNN_RING_DIAG2
    const unsigned long long dq1 = __builtin_amdgcn_s_memtime();
#endif
#ifdef PNN_RING_DIAG3
    unsigned long long dg3 = 0;
#endif
    auto stage_barrier = [&]() {                     // barrier s: stage s+1 is visible, buffer (s-1) % D is released
#ifdef PNN_RING_DIAG3           // time spent waiting at the stage barrier (MFMA wave 0), tools/ring_prof.hip
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long db0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef PNN_RING_DIAG3
        dg3 += __builtin_amdgcn_s_memtime() - db0;
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    if constexpr (KC == 1) {
        // one 16-deep chunk per stage: twice the ring depth in the same LDS.  Measured SLOWER (FC 1200x1200: 2030 cycles per
        // 32 of K against 1450 with KC = 2, D = 4): an activation row is then fetched in 64-byte pieces, half a cache line
        // per request, and the loader waves fall behind.  Kept as two configurations for the record; the fragment sets
        // alternate per STAGE, hence the loop over stage pairs
        for (int s = 0; s < nstages; s += 2) {
            stage_barrier();
            read_frags(ring + ((s + 1) % D) * SS, 0, wf1, af1);      // after the last stage: stale buffer, values dropped
            mfma_chunk(wf0, af0);
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < nstages) {
                stage_barrier();
                read_frags(ring + ((s + 2) % D) * SS, 0, wf0, af0);
                mfma_chunk(wf1, af1);
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int s = 0; s < nstages; s++) {
            const f32x4* buf = ring + (s % D) * SS;
            stage_barrier();
            DG_STAMP(dg_w);
#pragma unroll
            for (int j = 0; j + 1 < KC; j++) {
                if (j & 1) { read_frags(buf, j + 1, wf0, af0); mfma_chunk(wf1, af1); }
                else       { read_frags(buf, j + 1, wf1, af1); mfma_chunk(wf0, af0); }
                interleave();
                __builtin_amdgcn_sched_barrier(0);
            }
            DG_STAMP(dg_a);
            // first fragments of the next stage -- unconditional (after the last stage they come from a stale buffer and are
            // dropped): behind a branch the compiler's merged wait state makes the next MFMA wait for THESE reads too
            read_frags(ring + ((s + 1) % D) * SS, 0, wf0, af0);
            DG_STAMP(dg_i);
            mfma_chunk(wf1, af1);                    // chunk KC-1 (KC even: its fragments are in set 1)
            interleave();
            __builtin_amdgcn_sched_barrier(0);
            DG_STAMP(dg_b);
        }
    }
#ifdef PNN_RING_DIAG
    if (p.Xlo && tid == 0) {
        unsigned long long* d = (unsigned long long*)p.Xlo + 4 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        d[0] = dg_a; d[1] = dg_w; d[2] = dg_i; d[3] = dg_b;
    }
#endif

#ifdef PNN_RING_DIAG2
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long dq2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- epilogue, first half (MFMA waves): scale, bias, LeakyReLU; f32 / HM outputs straight from the accumulator layout,
    // the split-f16 output through LDS (copy_out above) ------------------------------------------------------------
    const int py = p.py[cls], px = p.px[cls];
    // all bias values first (a load placed next to its use cannot be hoisted over the stores in between: the compiler must
    // assume they alias) -- from the tile the loader side left in LDS, or, where the ring fills the LDS, from memory (one L2
    // round trip, ~1k cycles between the last MFMA and the first store)
    f32x4 bvs[NT][4];
    if (!kBiasLds) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = n0 + wn * (32 * NT) + nt * 32 + 8 * g + 4 * h;
                bvs[nt][g] = *reinterpret_cast<const f32x4*>(p.bias + (n < p.Cout ? n : 0));   // columns past Cout: any valid address, never stored
            }
    }
    __builtin_amdgcn_s_barrier();                    // barrier A: every wave is done with the ring
#ifdef PNN_RING_DIAG2
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long de1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // Two copies of the fully unrolled group loop, chosen once: the common case (split-f16 output only) without the f32 / HM
    // branches inside.  The loop runs ONCE per launch from a cold instruction cache, so its time is set by its code
    // footprint: with all three output kinds behind per-group branches it was ~20 KB and 7.9k cycles for 20 groups.
    const bool act = p.act != 0;
    float amax = 0.f;                                // range guard of the split outputs (pnn_device_common.h)
    if (kBiasLds) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) bvs[nt][g] = ring[D * SS + ((wn * (32 * NT) + nt * 32 + 8 * g + 4 * h) >> 2)];
    }
    auto groups = [&](auto direct_tag) {
        constexpr bool kDirect = decltype(direct_tag)::value;        // f32 / HM outputs straight from the accumulator layout
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int lrow = wm * (32 * RT) + rt * 32 + l31;
            const int mg = mblk + lrow;
            const bool rowok = pmg ? mg < p.nblk : mg < p.M;
            size_t obase = 0;
            if (kDirect) {
                const int mc = rowok ? mg : 0;
                int pbq = mc, piq = pmi, pjq = pmj;
                if (!pmg) {
                    pbq = mc / SP;
                    const int rq = mc - pbq * SP;
                    piq = rq / p.SW; pjq = rq - piq * p.SW;
                }
                const int oy = piq * p.os + py, ox = pjq * p.os + px;
                obase = (((size_t)pbq * p.OH + oy) * p.OW + ox) * p.Cout;
            }
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int nl = wn * (32 * NT) + nt * 32 + 8 * g + 4 * h;
                    const int n = n0 + nl;
                    const f32x4 bv = bvs[nt][g];
                    f32x4 v = (f32x4){acc[rt][nt][4 * g], acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]} * p.out_scale + bv;
                    if (act) {
                        v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                    }
#ifdef PNN_RING_NO_STORE       // ablation builds of tools/ring_prof.hip only
                    asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
                    continue;
#endif
                    if (!kDirect || p.Yhi || fuse) {  // same values and rounding as store_split4
                        typedef f16x4 h4;
                        h4 hi, lo;
                        amax = amax4(amax, v);
                        split4(v, hi, lo);
                        _Float16* dst = reinterpret_cast<_Float16*>(ring + lrow * OPP) + (nl >> 4) * 32 + (nl & 15);
                        *reinterpret_cast<h4*>(dst) = hi;
                        *reinterpret_cast<h4*>(dst + 16) = lo;
                    }
                    if (kDirect && rowok && n < p.Cout) {
                        if (p.Y) *reinterpret_cast<f32x4*>(p.Y + obase + n) = v;
                        if (p.Yi) {
                            int4 iv = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean),
                                                hm_round(v[3], p.mean));
                            *reinterpret_cast<int4*>(p.Yi + obase + n) = iv;
                        }
                    }
                }
        }
    };
#ifdef PNN_RING_DIAG4           // experiment: the group loop a second time, from a warm instruction cache (tools/ring_prof.hip)
    unsigned long long dg4[2] = {0, 0};
#pragma nounroll
    for (int rep = 0; rep < 2; rep++) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t4 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        groups(std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        dg4[rep] = __builtin_amdgcn_s_memtime() - t4;
        asm volatile("" : "+s"(rep));
    }
    if (p.Xlo && tid == 0) {
        unsigned long long* e4 = (unsigned long long*)p.Xlo + (1 << 20) + 2 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        e4[0] = dg4[0]; e4[1] = dg4[1];
    }
#else
    if (p.Y || p.Yi) groups(std::true_type{}); else groups(std::false_type{});
#endif
    report_range(p.range_flag, amax);
#ifdef PNN_RING_DIAG2
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long de2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    __builtin_amdgcn_s_barrier();                    // barrier B: the output tile (and the fused layer's weights) are complete
#ifdef PNN_RING_DIAG2
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long de3 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    copy_out();
    if (fuse) {
        f32x16 acc2[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc2[rt][nt][i] = 0.f;
#pragma unroll
        for (int c2 = 0; c2 < W2CH; c2++) {
            f32x4 w2[2][2], a2[RT][2];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                w2[nt][0] = ring[W2OFF + c2 * 256 + (0 + h) * 64 + nt * 32 + l31];
                w2[nt][1] = ring[W2OFF + c2 * 256 + (2 + h) * 64 + nt * 32 + l31];
            }
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int lrow = wm * (32 * RT) + rt * 32 + l31;
                a2[rt][0] = ring[lrow * OPP + c2 * 4 + 0 + h];
                a2[rt][1] = ring[lrow * OPP + c2 * 4 + 2 + h];
            }
#pragma unroll
            for (int part = 0; part < 2; part++)
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        const f16x8 whi = __builtin_bit_cast(f16x8, w2[nt][0]), wlo = __builtin_bit_cast(f16x8, w2[nt][1]);
                        const f16x8 ahi = __builtin_bit_cast(f16x8, a2[rt][0]), alo = __builtin_bit_cast(f16x8, a2[rt][1]);
                        if (part == 0) {
                            acc2[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi, acc2[rt][nt], 0, 0, 0);
                        } else {
                            acc2[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo, acc2[rt][nt], 0, 0, 0);
                            acc2[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi, acc2[rt][nt], 0, 0, 0);
                        }
                    }
        }
        float* __restrict__ part = p.part + ((size_t)blockIdx.y * p.M) * 64;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int mg = mblk + wm * (32 * RT) + rt * 32 + l31;
            if (mg >= p.M) continue;
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *reinterpret_cast<f32x4*>(part + (size_t)mg * 64 + nt * 32 + 8 * g + 4 * h) =
                        (f32x4){acc2[rt][nt][4 * g], acc2[rt][nt][4 * g + 1], acc2[rt][nt][4 * g + 2], acc2[rt][nt][4 * g + 3]};
        }
    }
#ifdef PNN_RING_DIAG2
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long de4 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef PNN_RING_DIAG2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.Xlo && tid == 0) {
        unsigned long long* d = (unsigned long long*)p.Xlo + 4 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        const unsigned long long dq3 = __builtin_amdgcn_s_memtime(), dr3 = __builtin_amdgcn_s_memrealtime();
        d[0] = dq1 - dq0; d[1] = dq2 - dq1; d[2] = dq3 - dq2; d[3] = dr3 - dr0;
        unsigned long long* e = (unsigned long long*)p.Xlo + (1 << 18) + 8 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
#ifdef PNN_RING_DIAG3
        e[5] = dg3; e[6] = dq0; e[7] = dr0;              // e[7]: start on the chip-wide 100 MHz clock (start / end spread over the grid)
#endif
        e[0] = de1 - dq2; e[1] = de2 - de1; e[2] = de3 - de2; e[3] = de4 - de3; e[4] = dq3 - de4;
    }
#endif
}

template <int RT, int NT, int KC, int WM, int D, bool FUSE = false>
__global__ __launch_bounds__(512) void tapgemm_ring_kernel(const TapGemmParams p)
{
    touch_kernargs<sizeof(TapGemmParams)>();
    (void)p;                                          // == the kernel-argument segment, read in place
    ring_layer<RT, NT, KC, WM, D, FUSE>(*(CParams*)__builtin_amdgcn_kernarg_segment_ptr(), threadIdx.x);
}

// (A variant that ran the hidden layers + output layer of an FC net as ONE launch, workgroups handing over between layers
// through counters, was built in round 1, measured 2 % slower than the per-layer launches -- what a launch boundary costs
// is start skew and stragglers, which the handshake waits for too -- and removed in round 2; see DESIGN.md.)

// X(rt, nt, kc, wm, d): tile 32*rt*wm x 32*nt*(4/wm), d-deep ring
#define PNN_RING_CFGS(X) \
    X(1, 4, 2, 4, 4) X(1, 4, 2, 4, 3) X(2, 2, 2, 2, 4) X(2, 2, 2, 2, 3) X(2, 4, 2, 4, 3) X(4, 2, 2, 2, 3) X(2, 3, 2, 2, 3) X(1, 3, 2, 4, 3) \
    X(1, 2, 2, 4, 4) X(1, 2, 2, 4, 3) X(2, 1, 2, 2, 4) X(2, 1, 2, 2, 3) X(1, 2, 2, 2, 4) X(2, 2, 2, 4, 3) X(1, 5, 2, 4, 4) X(1, 5, 2, 4, 3) \
    X(2, 3, 2, 4, 3) X(3, 2, 2, 2, 3) X(1, 3, 2, 4, 4) X(2, 3, 2, 2, 4) \
    X(1, 5, 1, 4, 8) X(1, 2, 1, 4, 8) \
    X(3, 2, 2, 2, 4)

static const TileCfg kCfgsRing[] = {
#define X(rt, nt, kc, wm, d) {rt, nt, kc, 316, wm, d},
    PNN_RING_CFGS(X)
#undef X
};

int tapgemm_ring_num_cfgs() { return (int)(sizeof(kCfgsRing) / sizeof(kCfgsRing[0])); }
TileCfg tapgemm_ring_cfg(int idx) { return kCfgsRing[idx]; }

size_t tapgemm_ring_lds_bytes(const TileCfg& t)
{
    const size_t bm = 32 * (size_t)t.rt * t.wm, bn = 32 * (size_t)t.nt * (4 / t.wm);
    const size_t ring = (size_t)t.d * (bm * 4 * t.kc + (size_t)t.kc * 4 * bn) * 16;
    return ring + 1024 <= 160 * 1024 ? ring + 1024 : ring;                        // + the bias tile (64 pieces) where it fits
}

// Sums the column tiles' partial products of a fused layer in tile order, undoes the weight scale, adds the bias and
// writes float and / or HM-epilogue outputs: Y[m][n] = bias[n] + scale * sum_t part[t][m][n], n < N2 <= 64.
__global__ __launch_bounds__(256) void fuse_reduce_kernel(const float* __restrict__ part, int ntiles, int M, int N2, const float* __restrict__ bias,
                                                          float scale, float mean, float* __restrict__ Y, int32_t* __restrict__ Yi, const DoneSignal done)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;          // one thread per 4 consecutive columns
    const int m = (int)(i >> 4), n = (int)(i & 15) << 2;
    if (m < M && n < N2) {
        // (eight loads in flight, then the additions in tile order: one memory round trip per 8 tiles instead of one per tile)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < ntiles; t0 += 8) {
            f32x4 v8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int t = t0 + j < ntiles ? t0 + j : ntiles - 1;
                v8[j] = *reinterpret_cast<const f32x4*>(part + ((size_t)t * M + m) * 64 + n);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) if (t0 + j < ntiles) acc += v8[j];
        }
        const f32x4 v = acc * scale + *reinterpret_cast<const f32x4*>(bias + n);
        if (Y) *reinterpret_cast<f32x4*>(Y + (size_t)m * N2 + n) = v;
        if (Yi) *reinterpret_cast<int4*>(Yi + (size_t)m * N2 + n) = make_int4(hm_round(v[0], mean), hm_round(v[1], mean), hm_round(v[2], mean), hm_round(v[3], mean));
    }
    signal_done(done);
}

hipError_t launch_fuse_reduce(const float* part, int ntiles, int M, int N2, const float* bias, float scale, float mean, float* Y, int32_t* Yi,
                              hipStream_t s, const DoneSignal& done)
{
    if (M <= 0) return hipSuccess;
    if (N2 % 4 || N2 > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fuse_reduce_kernel, dim3((unsigned)(((long)M * 16 + 255) / 256)), dim3(256), 0, s, part, ntiles, M, N2, bias, scale, mean, Y, Yi, done);
    return hipGetLastError();
}

// Position-major tiles (see ring_layer): when are they the faster launch?  Skipping the padding taps makes the workgroups
// unequal (a corner of a 3x3 layer keeps 4 taps of 9, the interior all 9), and at the batch sizes of the bench a layer is only
// one or two workgroups per CU, so what counts is not the work saved but the LAST workgroup to finish.  Both launches are
// therefore list-scheduled on the chip's workgroup slots with a cost of (kWgFixed + stages) per workgroup -- measured: start
// and epilogue of a 128 x 128 tile cost about as much as ten stages -- and position-major tiles are taken when they do not
// end later (a tie is a win in practice: less traffic and power for the same critical path; conv 16x16 at batch 1024, same
// box: 40.8 -> 39.8, 30.5 -> 28.7, 19.8 -> 18.7, 30.7 -> 27.5 us on the four layers where that holds).  One block group's
// input maps must fit an XCD's L2 (the positions of a group run on one XCD and re-read them tap by tap): 192-row tiles on
// the 8x24x64 maps of the 16x16 net's third layer fetched 190 MB instead of 71 and ran 78 us instead of 61.
// p.pm_groups on entry: -1 = never (option ring_pm = 0), 1 = whenever possible (ring_pm = 2, tests), 0 = by this model.
namespace {
constexpr double kWgFixed = 10.0;

double list_schedule(const std::vector<double>& cost, int slots)
{
    if ((int)cost.size() <= slots) return cost.empty() ? 0.0 : *std::max_element(cost.begin(), cost.end());
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (int i = 0; i < slots; i++) free_at.push(0.0);
    double end = 0.0;
    for (double c : cost) {
        const double t = free_at.top() + c;
        free_at.pop();
        free_at.push(t);
        end = std::max(end, t);
    }
    return end;
}

PmPlan plan_position_major(const TapGemmParams& p, int BM, int BN, int KC, size_t lds, double l2_mb)
{
    PmPlan plan;
    const int SP = p.SH * p.SW, ntaps = p.tap_begin[p.ncls];
    if (p.pm_groups < 0 || SP <= 1 || ntaps <= p.ncls || p.W2p || p.M % SP || (p.Cin / 16) % KC) return plan;
    const long B = p.M / SP;
    if (2 * B < BM) return plan;
    const long groups = (B + BM - 1) / BM;
    if (groups * SP > 0x7fffffffL / BM) return plan;
    std::vector<int> cnt((size_t)SP * p.ncls), total(SP, 0);         // in-image taps per (class, position)
    long inside = 0;
    for (int pos = 0; pos < SP; pos++) {
        const int i = pos / p.SW, j = pos % p.SW;
        for (int cls = 0; cls < p.ncls; cls++) {
            int n = 0;
            for (int t = p.tap_begin[cls]; t < p.tap_begin[cls + 1]; t++) {
                const int iy = i * p.a + (p.tap[t] >> 16), ix = j * p.a + (int)(short)(p.tap[t] & 0xffff);
                n += iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            }
            cnt[(size_t)cls * SP + pos] = n ? n : 1;
            total[pos] += n ? n : 1;
        }
        inside += total[pos];
    }
    if (inside >= (long)SP * ntaps) return plan;                     // nothing to skip
    {
        long live = 0;                               // (cnt / total hold 1 for a class without a live tap: it fetches zeros once)
        for (int pos = 0; pos < SP; pos++)
            for (int t = 0; t < ntaps; t++) {
                const int iy = (pos / p.SW) * p.a + (p.tap[t] >> 16), ix = (pos % p.SW) * p.a + (int)(short)(p.tap[t] & 0xffff);
                live += iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            }
        plan.live_frac = (double)live / ((double)SP * ntaps);
    }
    std::vector<int> order(SP);
    for (int i = 0; i < SP; i++) order[i] = i;
    if (SP <= 64) {                                  // heaviest positions first: the light ones fill the tail of the launch
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return total[a] > total[b]; });
        for (int i = 0; i < SP; i++) plan.order[i >> 2] |= (unsigned)order[i] << ((i & 3) * 8);
    }
    plan.groups = (int)groups;
    if (p.pm_groups == 1) { plan.use = true; return plan; }
    if ((double)BM * p.IH * p.IW * p.Cin * 4.0 > l2_mb * 1048576.0) return plan;   // a block group's input against the 4 MB L2 of an XCD (l2_mb: 4.5 for the split kernels; the f32 kernel, a third of their matrix rate, tolerates maps that spill to the MALL)
    const int slots = 256 * (lds <= 80 * 1024 ? 2 : 1);
    const int spt = p.Cin / 16 / KC, gy = (p.Cout + BN - 1) / BN;
    std::vector<double> bm, pm;
    for (int cls = 0; cls < p.ncls; cls++) {
        const int nt = p.tap_begin[cls + 1] - p.tap_begin[cls];
        bm.insert(bm.end(), (size_t)((p.M + BM - 1) / BM) * gy, kWgFixed + (double)nt * spt);
        for (int y = 0; y < gy; y++)
            for (long g0 = 0; g0 < groups; g0 += 8)                  // the kernel's launch order: chunks of 8 groups
                for (int pr = 0; pr < SP; pr++) pm.insert(pm.end(), (size_t)std::min(8L, groups - g0), kWgFixed + (double)cnt[(size_t)cls * SP + order[pr]] * spt);
    }
    plan.use = list_schedule(pm, slots) <= list_schedule(bm, slots);
    return plan;
}

}  // namespace

// the plan of a launch shape is computed once per thread (a few microseconds of host time otherwise, per launch)
const PmPlan& position_major_plan(const TapGemmParams& p, int BM, int BN, int KC, size_t lds, double l2_mb)
{
    typedef std::array<int, 14> Key;
    thread_local std::map<Key, PmPlan> plans;
    int taps_hash = 0;
    for (int t = 0; t < p.tap_begin[p.ncls]; t++) taps_hash = taps_hash * 31 + p.tap[t];
    const Key key = {p.M, p.SH, p.SW, p.IH, p.IW, p.Cin, p.Cout, p.a, p.ncls, taps_hash, BM, BN, KC * 4 + (p.pm_groups + 1), (int)(lds >> 10) + 1024 * (int)(l2_mb * 8.0)};
    auto it = plans.find(key);
    if (it == plans.end()) {
        if (plans.size() >= 4096) plans.clear();     // a caller that never repeats a batch size: start over rather than grow
        it = plans.emplace(key, plan_position_major(p, BM, BN, KC, lds, l2_mb)).first;
    }
    return it->second;
}

template <int RT, int NT, int KC, int WM, int D, bool FUSE = false>
static hipError_t launch_ring(const TapGemmParams& p, hipStream_t s)
{
    constexpr int BM = 32 * RT * WM, BN = 32 * NT * (4 / WM);
    const size_t lds = tapgemm_ring_lds_bytes(TileCfg{RT, NT, KC, 316, WM, D});
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tapgemm_ring_kernel<RT, NT, KC, WM, D, FUSE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.ncls);
    TapGemmParams q = p;
    q.pm_groups = 0;
    g_last_issued_frac = 1.0;
    if (!FUSE && p.pm_groups >= 0 && p.SH * p.SW > 1) {
        static const double l2_mb = getenv("PNN_RING_PM_L2_MB") ? atof(getenv("PNN_RING_PM_L2_MB")) : 4.5;   // (round 4 re-check at 7 MB, which the f32 kernel takes: see NOTES.md)
        const PmPlan& plan = position_major_plan(p, BM, BN, KC, lds, l2_mb);
        g_last_issued_frac = plan.use ? plan.live_frac : 1.0;
        if (plan.use) {
            q.pm_groups = plan.groups;
            q.nblk = p.M / (p.SH * p.SW);
            for (int i = 0; i < 16; i++) q.pos_order[i] = plan.order[i];
            grid.x = (unsigned)(plan.groups * p.SH * p.SW);
            static const bool debug = getenv("PNN_DEBUG") != nullptr;
            if (debug) fprintf(stderr, "[pnn] ring %dx%d: position-major tiles, %d block groups x %d positions\n", BM, BN, plan.groups, p.SH * p.SW);
        }
    }
    pnn_launch(tapgemm_ring_kernel<RT, NT, KC, WM, D, FUSE>, grid, dim3(512), lds, s, q);
    return hipGetLastError();
}

// Tiles that also exist with the fused next layer (FUSE = true): WM = 4 and room for W2 behind the output tile.
#define PNN_RING_FUSE_CFGS(X) X(1, 5, 2, 4, 4) X(1, 4, 2, 4, 4) X(1, 3, 2, 4, 4) X(1, 2, 2, 4, 4)

bool tapgemm_ring_can_fuse(int idx)
{
    const TileCfg t = tapgemm_ring_cfg(idx);
#define X(r_, n_, k_, w_, d_) if (t.rt == r_ && t.nt == n_ && t.kc == k_ && t.wm == w_ && t.d == d_) return true;
    PNN_RING_FUSE_CFGS(X)
#undef X
    return false;
}

hipError_t launch_tapgemm_ring(const TapGemmParams& p, int idx, hipStream_t s)
{
    if (p.M <= 0) return hipSuccess;
    if (!p.zero) return hipErrorInvalidValue;
    if (p.W2p) {                                      // fused next layer requested
        if (p.ncls != 1 || p.SH * p.SW != 1 || !p.part) return hipErrorInvalidValue;
        const TileCfg t = tapgemm_ring_cfg(idx);
#define X(r_, n_, k_, w_, d_) if (t.rt == r_ && t.nt == n_ && t.kc == k_ && t.wm == w_ && t.d == d_) return launch_ring<r_, n_, k_, w_, d_, true>(p, s);
        PNN_RING_FUSE_CFGS(X)
#undef X
        return hipErrorInvalidValue;
    }
    int i = 0;
#define X(rt, nt, kc, wm, d) if (idx == i++) return launch_ring<rt, nt, kc, wm, d>(p, s);
    PNN_RING_CFGS(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace pnn
