// Exact-f32 tap GEMM on v_mfma_f32_32x32x2_f32 with ONE wave per SIMD and nothing left for that wave to wait for.
//
// The contract (TapGemmParams, pnn_kernels.h):
//   Y[pix(m)][n] = act( sum_{taps, ci} X[b, i*a+dy, j*a+dx, ci] * W[tap][ci][n] + bias[n] )
// i.e. the FC layers, convolutions and transposed convolutions of pnn/components.py:10-261 in the REFERENCE's arithmetic
// (IEEE float32 products, float32 accumulation: pnn/tfutils.py:107-139, components.py:169-176).  Per-output summation order: 16-deep
// chunks in K order; MFMA step e of a chunk adds x[8h + e] * w[8h + e] for the two k-halves h, i.e. a k-ordered fmaf chain 0, 8, 1, 9,
// ..., 7, 15 per chunk.  Every tile gives the same bits: this is the canonical f32 order of the library -- for a layer whose deepest
// class stays under kSegMinDepth.  A DEEPER convolution layer is summed in K segments (GemmLayer::nseg, pnn_model.cpp: whole taps, at
// most kSegDepth deep): grid z = class * nseg + segment, a workgroup walks its segment's taps only and leaves its raw sums in plane
// `segment` of a partial buffer; seg_reduce_kernel adds the planes in order, then bias and LeakyReLU.  That is the layer's order at
// EVERY batch size and tile (the bit-identity tests cover the 32x32 and 64x64 nets, which have such layers).  Why: a chain of K / 2
// dependent 64-cycle MFMAs per output is a latency no tiling hides -- the 6400-deep third layer of the 64x64 net took 185 us at batch
// 64 on 192 workgroups (1536 wave tiles on 1024 SIMDs: two rounds for one and a half) and 171 us at batch 1; in four segments 145
// and 45.  conv 64x64 at batch 64: 46.0 k -> 52.7 k blocks/s, a single-block call 538 -> 347 us; conv 32x32 +2 % / 333 -> 275 us.
// (The round-1 kernels -- tapgemm_kernel on 16x16x4, tapgemm32_kernel, the split-K kernel for small M -- were removed in round 5; the
// small-M form of THIS order is tapgemm_f32_small_kernel, pnn_gemm_f32_small.hip.)
//
// Why another kernel (tools/mfma_peak.hip, tools/f32_sweep.py; profiles/r04_f32_tile_sweep.txt): the 32x32x2 instruction sustains
// 154.5 TFLOP/s (0.98 of the 157.3 peak) with one or two waves per SIMD and 124 with four, the 16x16x4 instruction 124-139 with
// as many waves as fit -- the f32 roof is only within reach of a kernel that runs FEW waves with BIG wave tiles.  tapgemm32_kernel
// has the tiles but not the schedule: one wave per SIMD exposes every wait (fragment reads at the head of a stage, the issue
// time of its ten 1-KiB global loads, ds_write + barrier at its end), and its 128 x 128 / 128 x 160 tiles measured 63-78 TFLOP/s
// on the FC 8x8 layers where the 64 x 64 tiles of the 16x16x4 kernel reach 109.  Here:
//   * workgroup = 4 waves, wave w owns rows [32*RT*w, +32*RT) of the BM = 128*RT row tile and all BN = 32*NT columns (FC 8x8 at
//     batch 4096: 128 x 160 tiles = 32 x 8 workgroups, exactly one per CU, 0.9375 of the columns real);
//   * weights: global -> LDS by LDS-DMA (no staging registers, no ds_write), THREE stage buffers of KC chunks; the one barrier of a
//     stage sits BEFORE its last chunk, so that the first fragments of the next stage are read under that chunk's MFMAs, and stage
//     s+2 is fetched all along stage s into the buffer stage s-1 left (a whole stage to land);
//   * fragments: two register sets, chunk j+1's read under chunk j's MFMAs; activations: global -> registers a stage ahead,
//     two sets alternating by stage parity (the loop body is instantiated for both parities: no copies);
//   * every memory instruction sits alone between two MFMAs (sched_barrier after each): a 1-KiB vector-memory instruction
//     holds its wave for about as long as a 32x32x2 MFMA runs (64 cycles), two in a row stall the matrix pipe.  First version of
//     this file, pairs of loads per k-step: FC 1200 x 1200 at batch 4096 99 us; one per MFMA slot: see DESIGN.md section 4;
//   * FUSE: the net's output layer (<= 64 outputs, no activation: components.py:173-176) is applied to the activated output
//     tile straight from the accumulators -- the D layout of one MFMA (lane = row m, registers = columns 8g + 4h + r) IS the
//     B-operand layout of the next -- and leaves as per-column-tile partial sums [tile][M][64] for fuse_reduce_kernel.  The column
//     tiles are the K segments of the output layer's canonical order; fc_out_f32_kernel reproduces them from stored activations
//     for the batch sizes that do not take this tile.
#include "pnn_kernels.h"
#include "pnn_device_common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace pnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, f32x4* l)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_le()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// SEQ: K segments one after the other inside the workgroup (p.seg_seq; big launches, where more workgroups buy nothing): the
// stages of all segments run as ONE pipeline, and where a segment ends the accumulators are folded into a running total --
// total = (p0 + p1) + ..., the order seg_reduce_kernel adds the planes in -- so both forms give the same bits.
// It is the ONLY form of a segmented one-tap (FC) layer (round 6, GemmLayer::fc_seg_chunks: segments of p.seg_chunks chunks of the
// tap), with the fused output layer too: the fold leaves the layer's sums in `acc`, the epilogues follow.  SEQ = 1: the general form
// (segments of whole taps, a fold wherever a stage ends one: the convolution layers); SEQ = 2: a one-tap layer whose segments are an
// EVEN number of stages -- an outer loop over the segments around the plain kernel's two-stage loop, the fold between two inner loops.
// (Why two forms: with the fold behind a test after every stage the compiler carries the accumulators through a copy per stage -- 80
// v_accvgpr_mov per 160 MFMAs, 3 % of an FC layer at batch 4096, FC 8x8 18.1 M -> 17.3 M blocks/s whatever the number of folds,
// profiles/r06_fcseg_ab.txt; the folds themselves are 0.5 us each per layer.)
// One tile (bx, by, bz) of the launch's gx x gy x gz tiles -- what a workgroup of the plain launch does once and a workgroup of a
// PERSISTENT launch (fewer workgroups than tiles, tapgemm_f32_kernel below) does for one tile after the other.
template <int RT, int NT, int KC, bool FUSE, int SEQ>
__device__ __forceinline__ void f32_tile(const TapGemmParams& p, f32x4* const lds, const int bx, const int by, const int bz, const int gx, const int gy)
{
    static_assert(KC >= 2 && KC % 2 == 0, "fragment sets alternate by chunk parity");
    constexpr int BM = 128 * RT, BN = 32 * NT;
    constexpr int E = 4 * BN;                        // 16-byte pieces per staged weight chunk: [q = 4][BN]
    constexpr int SE = KC * E;                       // pieces per stage
    constexpr int NDMA = SE / 64;                    // LDS-DMA wave-instructions per stage = 2 KC NT, dealt round-robin to the 4 waves
    constexpr int NPW = NDMA / 4;
    constexpr int W2R = BN / 4;                      // FUSE: rows of 64 pieces of the output layer's weight tile
    static_assert(SE % 64 == 0 && NDMA % 4 == 0, "a stage is a whole number of wave instructions, the same number for every wave");
    // memory work of one chunk, one instruction per MFMA slot: 2 NT fragment reads (next chunk), 2 RT activation loads (this chunk
    // of the NEXT stage), this chunk's share of the LDS-DMA of stage s+2
    constexpr int SLOTS = 8 * NT * RT;
    constexpr bool kCoalescedOut = true;
    // lds: [3][SE] weight stages | FUSE: [W2R][64] output-layer tile
    f32x4* const W2s = lds + 3 * SE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // K segments (p.nseg > 1, never with the fused output layer): z = class * nseg + segment
    const int nseg = (!FUSE && !SEQ && p.nseg > 1) ? p.nseg : 1;
    const int cls = nseg > 1 ? bz / nseg : bz;
    const int seg = bz - cls * nseg;
    const int n0 = by * BN;
    const int m0 = bx * BM + wave * (32 * RT);
#ifdef PNN_F32_DIAG             // diagnostic library only (make diag): cycle stamps of wave 0 -> p.Xlo[workgroup][8]
    const unsigned long long dq0 = __builtin_amdgcn_s_memtime(), dr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dq1 = 0, dq2 = 0;
#endif

    const int SP = p.SH * p.SW;
    const int cpt = p.Cin >> 4;                      // 16-deep chunks per tap (a multiple of KC, or one tap)
    const int t0 = p.tap_begin[cls], t1 = p.tap_begin[cls + 1];
    // Position-major tiles (p.pm_groups > 0, convolutions at batch; launch_f32 decides with the ring kernel's planner,
    // pnn_gemm_ring.hip): the tile's BM rows are BM BLOCKS at ONE position (pmi, pmj) of the SH x SW grid, first block mblk.  A tap
    // then lies inside the image for every row or for none, and the taps that only meet SAME padding (17 % of the 16x16 net's
    // multiply-adds) are skipped -- no loads, no MFMAs.  Their products are exact zeros: every output keeps its bits.
    // tmask = this class's taps that stay (all of them otherwise).
    const int pmg = p.pm_groups;
    int pmi = 0, pmj = 0, mblk = 0;
    unsigned tmask = t1 - t0 >= 32 ? 0xffffffffu : (1u << (t1 - t0)) - 1u;
    unsigned smask = 0xffffffffu;                    // this segment's taps: the class's taps dealt in order, the first (taps % nseg) segments one more
    if (nseg > 1) {
        const int base = (t1 - t0) / nseg, rem = (t1 - t0) - base * nseg;
        smask = ((1u << (base + (seg < rem ? 1 : 0))) - 1u) << (seg * base + (seg < rem ? seg : rem));
        tmask &= smask;
    }
    if (pmg) {
        // launch order: chunks of 8 block groups; within a chunk rank by position rank, the 8 groups side by side -- workgroups
        // i, i + 8, ... run on one XCD, so each XCD walks the positions of ONE group at a time and its L2 keeps that group's maps
        const int gc = bx / (SP * 8), r = bx - gc * (SP * 8);
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
        m &= smask;
        tmask = m ? m : (smask & (0u - smask));      // no tap inside: one of them fetches zeros
    }
    auto next_tap = [&](int t) {                     // the next tap that stays after t (class-relative), or t itself at the end
        const unsigned rest = tmask & ~((2u << t) - 1u);
        return rest ? (int)__builtin_ctz(rest) : t;
    };
    int pb[RT], pi[RT], pj[RT];
    bool mv[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int row = wave * (32 * RT) + rt * 32 + l31;
        if (pmg) {
            mv[rt] = mblk + row < p.nblk;
            pb[rt] = mv[rt] ? mblk + row : 0; pi[rt] = pmi; pj[rt] = pmj;
            continue;
        }
        const int mg = m0 + rt * 32 + l31;
        mv[rt] = mg < p.M;
        const int mc = mv[rt] ? mg : 0;
        if (SP == 1) { pb[rt] = mc; pi[rt] = 0; pj[rt] = 0; }       // fully-connected layer (launch-uniform): no divisions
        else {
            const int b = mc / SP;
            const int r = mc - b * SP;
            pb[rt] = b;
            pi[rt] = r / p.SW;
            pj[rt] = r - pi[rt] * p.SW;
        }
    }
    const int nchunks = __builtin_popcount(tmask) * cpt;
    const int nstages = (nchunks + KC - 1) / KC;     // the packed weights are zero-padded to whole stages (kChunkPad)
    // SEQ: stages of segment k = its live taps x (cpt / KC) (a segmented layer has whole stages per tap); fold_at = the stage count at
    // which the running segment ends (segments without a live tap end where they begin and fold nothing)
    const int sq_n = SEQ != 0 ? p.nseg : 1, sq_base = (t1 - t0) / sq_n, sq_rem = (t1 - t0) - sq_base * sq_n;
    auto seg_stages = [&](int k) {
        const unsigned mk = ((1u << (sq_base + (k < sq_rem ? 1 : 0))) - 1u) << (k * sq_base + (k < sq_rem ? k : sq_rem));
        return __builtin_popcount(tmask & mk) * (cpt / KC);
    };
    int sq_k = 0, fold_at = SEQ == 1 ? seg_stages(0) : 0, sq_done = 0;

    // ---- weights: this lane's pieces of a stage (LDS-DMA: lane-linear destination, per-lane source) ------------------------
    const unsigned wbytes = (unsigned)p.chunk_begin[p.ncls] * 4u * (unsigned)p.Npad * 16u;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, wbytes, 0x00020000);
    const unsigned wcls = (unsigned)p.chunk_begin[cls] * 4u * (unsigned)p.Npad * 16u;
    const unsigned wchunk = 4u * (unsigned)p.Npad * 16u;                      // bytes per packed chunk
    // cursor of the next stage to be fetched: tap (class-relative) and chunk within it
    int dt = (int)__builtin_ctz(tmask), dcc = 0;
    auto advance_d = [&]() {
        dcc += KC;
        if (dcc >= cpt) {
            const int n = next_tap(dt);
            if (n != dt) { dt = n; dcc = 0; } else dcc -= KC;       // past the end: the last stage again (harmless, no exec-masked code)
        }
    };
    unsigned bsrc[NPW];
#pragma unroll
    for (int ii = 0; ii < NPW; ii++) {
        const int i = wave + 4 * ii;
        const int e = 64 * i + lane;
        const int j = e / E, ee = e - j * E;
        const int q = ee / BN, nn = ee - q * BN;
        bsrc[ii] = (unsigned)(((j * 4 + q) * p.Npad + n0 + nn) << 4) + wcls;
    }
    auto dma_b = [&](int buf, int ii) {              // instruction ii of this wave, of the stage under the cursor
        dma16(wrsrc, bsrc[ii], (unsigned)(dt * cpt + dcc) * wchunk, lds + buf * SE + 64 * (wave + 4 * ii));
    };

    // ---- activations: buffer-descriptor loads, out-of-image taps / rows past M read zeros, per-tap byte offsets ------------
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
    unsigned aoff[RT];
    auto tap_setup = [&](int tp) {                   // tp = (dy << 16) | (dx & 0xffff)
        const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int iy = pi[rt] * p.a + dy, ix = pj[rt] * p.a + dx;
            const bool ok = mv[rt] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            const unsigned off = ((unsigned)((pb[rt] * p.IH + iy) * p.IW + ix) * (unsigned)p.Cin + (h << 3)) << 2;
            aoff[rt] = ok ? off : 0x80000000u;
        }
    };
    // cursor of the NEXT stage to be loaded
    int lt = (int)__builtin_ctz(tmask), lcc = 0;
    tap_setup(p.tap[t0 + lt]);
    int ltp_next = p.tap[t0 + next_tap(lt)];          // the next tap's word, fetched a whole tap early
    auto load_a_chunk = [&](int j, f32x4 (&dst)[KC][RT][2]) {
        const int cj = lcc + j < cpt ? lcc + j : cpt - 1;          // padding chunk: zero weights, any finite data will do
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            dst[j][rt][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[rt], cj << 6, 0));
            dst[j][rt][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[rt] + 16u, cj << 6, 0));
        }
    };
    auto advance_a = [&]() {                         // to the stage after the one just loaded (wave-uniform)
        lcc += KC;
        if (lcc >= cpt) {
            const int n = next_tap(lt);
            if (n != lt) {
                lcc = 0;
                lt = n;
                tap_setup(ltp_next);
                ltp_next = p.tap[t0 + next_tap(lt)];
            }
        }
    };

    f32x16 acc[RT][NT];
    f32x16 total[SEQ ? RT : 1][SEQ ? NT : 1];        // SEQ: the sum of the finished segments
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                acc[rt][nt][i] = 0.f;
                if (SEQ) total[SEQ ? rt : 0][SEQ ? nt : 0][i] = 0.f;
            }
    // SEQ = 1: a full stage is done; where its segment ends with it, fold (wave-uniform): total += acc, acc = 0.
    // (Round 6, tried and removed: the fold INSIDE the next stage's first k-step -- per accumulator tile, its first MFMA taking a zero
    // addend -- needs a second form of the stage body; with it the register allocator spilled 199 VGPRs and FC 8x8 at batch 4096 lost
    // 10 %, profiles/r06_fcseg_ab.txt.)
    // first: nothing has been folded yet -- the total is 0, and 0 + p = p exactly for a chain sum p (one that starts from +0 is never
    // -0): a copy in place of read, read, add, write
    auto fold_now = [&](bool first = false) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (first) total[SEQ ? rt : 0][SEQ ? nt : 0][i] = acc[rt][nt][i];
                    else total[SEQ ? rt : 0][SEQ ? nt : 0][i] += acc[rt][nt][i];
                    acc[rt][nt][i] = 0.f;
                }
    };
    auto after_stage = [&]() {
        if (SEQ != 1) return;
        ++sq_done;
        if (sq_done != fold_at || sq_k >= sq_n - 1) return;
        fold_now();
        do { ++sq_k; fold_at += seg_stages(sq_k); } while (fold_at == sq_done && sq_k < sq_n - 1);
    };

    auto read_wf = [&](int buf, int j, int nt, f32x4 (&wf)[NT][2]) {
        wf[nt][0] = lds[buf * SE + j * E + (2 * h) * BN + nt * 32 + l31];
        wf[nt][1] = lds[buf * SE + j * E + (2 * h + 1) * BN + nt * 32 + l31];
    };

    // ---- prologue: stages 0 and 1 on their way, the output layer's weight tile behind them -------------------------------
    f32x4 a0[KC][RT][2], a1[KC][RT][2];              // activations of the even / odd stages
    f32x4 wf0[NT][2], wf1[NT][2];                    // weight fragments of the even / odd chunks
#pragma unroll
    for (int ii = 0; ii < NPW; ii++) dma_b(0, ii);
    advance_d();
#pragma unroll
    for (int j = 0; j < KC; j++) load_a_chunk(j, a0);
    advance_a();
#pragma unroll
    for (int ii = 0; ii < NPW; ii++) dma_b(1, ii);
    advance_d();
    if (FUSE) {
        // rows n/4 in [n0/4, n0/4 + BN/4) of the output layer's pack [k/4][Npad2][4 floats]: its first 64 columns, 1 KiB per row;
        // rows past its K read zeros (descriptor bound)
        const __amdgpu_buffer_rsrc_t w2rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2p, 0, (unsigned)p.K2chunks * 4u * (unsigned)p.Npad2 * 16u, 0x00020000);
        for (int r = wave; r < W2R; r += 4) dma16(w2rsrc, (unsigned)(((n0 >> 2) + r) * p.Npad2 + lane) << 4, 0, W2s + r * 64);
    }
    static_assert(!FUSE || W2R % 4 == 0, "every wave fetches the same number of rows of the output layer's tile");
    wait_vm_le<NPW + (FUSE ? W2R / 4 : 0)>();        // stage 0 and its activations; stage 1 (and the tile behind it) is waited for where stage 0 ends
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < NT; nt++) read_wf(0, 0, nt, wf0);

    // ---- one stage: KC chunks of 8 k-steps x NT x RT MFMAs; after each MFMA at most ONE memory instruction -------------------
    // chunk j carries: the fragment reads of chunk j+1 (the last chunk: the next stage's first, from its buffer), the
    // activation loads of chunk j of stage s+1, and its share of the LDS-DMA of stage s+2 into buffer (s+2) % 3.
    auto stage = [&](int buf, f32x4 (&acur)[KC][RT][2], f32x4 (&anxt)[KC][RT][2]) {
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf == 0 ? 2 : buf - 1;      // (s+1) % 3, (s+2) % 3
#pragma unroll
        for (int j = 0; j < KC; j++) {
            constexpr int kLast = KC - 1;
            const int d0 = j * NPW / KC, d1 = (j + 1) * NPW / KC;   // this chunk's LDS-DMA instructions [d0, d1)
            const int nops = 2 * NT + 2 * RT + (d1 - d0);
            if (j == kLast) {
                // every LDS-DMA of stage s+1 has landed (issued during stage s-1; what this stage issued so far -- 2 RT activation
                // loads and the DMA share per chunk, all younger -- may stay in flight), this wave's fragment reads are complete:
                // after the barrier stage s+1 is whole in its buffer
                wait_vm_le<(2 * RT) * (KC - 1) + (KC - 1) * NPW / KC>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            f32x4 (&wc)[NT][2] = (j & 1) ? wf1 : wf0;
            f32x4 (&wn)[NT][2] = (j & 1) ? wf0 : wf1;
#pragma unroll
            for (int e = 0; e < 8; e++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nt][e >> 2][e & 3], acur[j][rt][e >> 2][e & 3], acc[rt][nt], 0, 0, 0);
                        // the memory instruction of this MFMA slot, if any: operation k of the chunk sits in slot k * SLOTS / nops
                        const int sl = (e * NT + nt) * RT + rt;
                        const int k = (sl * nops + SLOTS - 1) / SLOTS;
                        if (k < nops && k * SLOTS / nops == sl) {
                            if (k < 2 * NT) {                       // fragment k of the next chunk
                                const int fnt = k >> 1, fh = k & 1;
                                const int src = (j == kLast ? buf1 * SE : buf * SE + (j + 1) * E) + (2 * h + fh) * BN + fnt * 32 + l31;
                                wn[fnt][fh] = lds[src];
                            } else if (k < 2 * NT + 2 * RT) {       // activations of chunk j of stage s+1
                                const int ar = (k - 2 * NT) >> 1, ah = (k - 2 * NT) & 1;
                                const int cj = lcc + j < cpt ? lcc + j : cpt - 1;   // padding chunk: zero weights, any finite data will do
                                anxt[j][ar][ah] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[ar] + 16u * ah, cj << 6, 0));
                            } else {
                                dma_b(buf2, d0 + (k - 2 * NT - 2 * RT));
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
        advance_a();
        advance_d();
    };
#ifdef PNN_F32_DIAG
    dq1 = __builtin_amdgcn_s_memtime();
#endif
    // the LAST stage: nothing left to fetch, and only its live chunks run -- the zero-padded chunks that fill the stage (K = 1200:
    // chunk 76 of 75) would add exact zeros for 2560 cycles each
    auto tail = [&](int buf, f32x4 (&acur)[KC][RT][2], int nlive) {
#pragma unroll
        for (int j = 0; j < KC; j++) {
            if (j >= nlive) break;
            f32x4 (&wc)[NT][2] = (j & 1) ? wf1 : wf0;
            f32x4 (&wn)[NT][2] = (j & 1) ? wf0 : wf1;
#pragma unroll
            for (int e = 0; e < 8; e++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
#pragma unroll
                    for (int rt = 0; rt < RT; rt++) {
                        acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[nt][e >> 2][e & 3], acur[j][rt][e >> 2][e & 3], acc[rt][nt], 0, 0, 0);
                        const int sl = (e * NT + nt) * RT + rt;
                        constexpr int nops = 2 * NT;
                        const int k = (sl * nops + SLOTS - 1) / SLOTS;
                        if (j + 1 < KC && k < nops && k * SLOTS / nops == sl)
                            wn[k >> 1][k & 1] = lds[buf * SE + (j + 1) * E + (2 * h + (k & 1)) * BN + (k >> 1) * 32 + l31];
                        __builtin_amdgcn_sched_barrier(0);
                    }
        }
    };
    const int nlive = nchunks - (nstages - 1) * KC;
    int s = 0, buf = 0;
    if (SEQ == 1) while (fold_at == 0 && sq_k < sq_n - 1) { ++sq_k; fold_at += seg_stages(sq_k); }   // leading segments without a live tap
    if constexpr (SEQ == 2) {
        // every segment in front of the last LIVE one: an even number of stages (a one-tap layer: seg_chunks / KC; segments of whole taps:
        // live taps x cpt / KC, launch_f32 takes this form only where cpt / KC is even), the plain two-stage loop, then the fold.  The last
        // live segment ends with the tile's last stage (tail) below; segments behind it have no live tap: nothing to add.
        const bool one_tap_seq = t1 - t0 == 1;
        int last_live = 0;
        if (one_tap_seq) last_live = p.nseg - 1;
        else for (int k = 0; k < p.nseg; k++) if (seg_stages(k) > 0) last_live = k;
        for (int k = 0; k < last_live; k++) {
            const int spp = one_tap_seq ? (int)p.seg_chunks / KC : seg_stages(k);
            for (int i = 0; i < spp; i += 2) {
                stage(buf, a0, a1);
                buf = buf == 2 ? 0 : buf + 1;
                stage(buf, a1, a0);
                buf = buf == 2 ? 0 : buf + 1;
            }
            s += spp;
            if (k == 0) fold_now(true); else fold_now();
        }
    }
    for (; s + 2 < nstages; s += 2) {
        stage(buf, a0, a1);
        buf = buf == 2 ? 0 : buf + 1;
        after_stage();
        stage(buf, a1, a0);
        buf = buf == 2 ? 0 : buf + 1;
        after_stage();
    }
    if (s + 1 < nstages) {
        stage(buf, a0, a1);
        buf = buf == 2 ? 0 : buf + 1;
        after_stage();
        tail(buf, a1, nlive);
    } else {
        tail(buf, a0, nlive);
    }
    if (SEQ) {                                       // the last segment's sum joins the others': acc = total + acc
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[rt][nt][i] = total[SEQ ? rt : 0][SEQ ? nt : 0][i] + acc[rt][nt][i];
    }
#ifdef PNN_F32_DIAG
    dq2 = __builtin_amdgcn_s_memtime();
    auto diag_out = [&]() {
        if (tid == 0 && p.Xlo) {
            unsigned long long* d = (unsigned long long*)p.Xlo + 8 * ((bz * gy + by) * gx + bx);
            d[0] = dq1 - dq0; d[1] = dq2 - dq1; d[2] = __builtin_amdgcn_s_memtime() - dq2; d[3] = __builtin_amdgcn_s_memrealtime() - dr0;
            d[4] = dr0;
        }
    };
#else
    auto diag_out = [&]() {};
#endif

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    // (the last stages' refills -- re-fetches of the last stage, never read -- must have landed before this wave may end: an LDS-DMA
    // still in flight would write into LDS that the next workgroup on this CU already owns; issued a chunk or more ago, so no wait in practice)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (FUSE && nstages < 2) __builtin_amdgcn_s_barrier();   // no stage ran its barrier behind the prologue's fetches: the output layer's tile is whole only now
    const int py = p.py[cls], px = p.px[cls];
    f32x4 bvs[NT][4];                                // all bias loads before the first store (see pnn_gemm_sp.hip)
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int n = n0 + nt * 32 + 8 * g + 4 * h;
            bvs[nt][g] = *reinterpret_cast<const f32x4*>(p.bias + (n < p.Cout ? n : 0));
        }
    if (FUSE && p.W2p) {
        // out2[m][o] = sum_n leaky(acc[m][n] + bias[n]) * W2[n][o] over this tile's BN columns n: accumulator register 4g + r of
        // column tile nt is n = n0 + 32 nt + 8g + 4h + r of row m = lane & 31 -- fed as the B operand of k-step (g, r); the A
        // operand (rows o = 32 ot + (lane & 31)) is piece ((32 nt + 8g + 4h) / 4, o) of the output layer's pack, element r.
        // Columns past Cout: zero weights in both layers.
        static_assert(!FUSE || RT == 1, "the fused output layer is written for one row tile per wave");
        constexpr int OT = 2;                        // 64 outputs
        f32x16 acc2[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[ot][i] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            f32x4 w2[4][OT];
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int ot = 0; ot < OT; ot++) w2[g][ot] = W2s[(8 * nt + 2 * g + h) * 64 + ot * 32 + l31];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                f32x4 v = (f32x4){acc[0][nt][4 * g], acc[0][nt][4 * g + 1], acc[0][nt][4 * g + 2], acc[0][nt][4 * g + 3]} + bvs[nt][g];
                if (p.act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int ot = 0; ot < OT; ot++) acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[g][ot][r], v[r], acc2[ot], 0, 0, 0);
            }
        }
        if (mv[0]) {
            float* dst = p.part + ((size_t)by * p.M + (m0 + l31)) * 64;
#pragma unroll
            for (int ot = 0; ot < OT; ot++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *reinterpret_cast<f32x4*>(dst + ot * 32 + 8 * g + 4 * h) =
                        (f32x4){acc2[ot][4 * g], acc2[ot][4 * g + 1], acc2[ot][4 * g + 2], acc2[ot][4 * g + 3]};
        }
        diag_out();
        return;
    }
    // f32 output of an FC layer: through LDS, whole rows.  A lane of the accumulator layout owns 16-byte pieces of 32 different rows
    // -- 32-byte fragments per row and store instruction, and with one workgroup per CU every workgroup of the launch reaches this
    // point at the same moment (20 MB through the write path at once); transposed through a wave-private LDS tile (the ring is dead)
    // consecutive lanes store consecutive pieces of a row.  Same-box A/B: FC 8x8 f32 at batch 4096 0.2312 -> 0.2295 ms, FC 4x4
    // 0.2114 -> 0.2107; convolution layers (several co-resident workgroups, epilogues already staggered) lost 0.7 % and keep the
    // direct stores.  The stores go THROUGH L2 (store16_through, pnn_device_common.h: the next layer's workgroups run on other XCDs, and
    // what is written through early is not left for the end-of-kernel write-back): FC 8x8 0.2277 -> 0.2256 ms, FC 4x4 0.2074 -> 0.2051.
    constexpr int TP = BN / 4 + 1;                   // tile row pitch in 16-byte pieces (+1: conflict-free for both accesses)
    float* const Yo = (nseg > 1 && p.Y) ? p.Y + (size_t)seg * p.seg_stride : p.Y;
    if (kCoalescedOut && p.Y && !p.Yi && SP == 1) {      // launch-uniform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // nobody reads the ring any more, no LDS-DMA in flight
        f32x4* tile = lds + wave * (32 * RT * TP);
#pragma unroll
        for (int rt = 0; rt < RT; rt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v = (f32x4){acc[rt][nt][4 * g], acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]} + bvs[nt][g];
                    if (p.act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
                    tile[(rt * 32 + l31) * TP + nt * 8 + 2 * g + h] = v;
                }
        __builtin_amdgcn_wave_barrier();              // the tile is this wave's own: LDS keeps a wave's accesses in order
        const int cq = p.Cout >> 2, nq0 = n0 >> 2;
        const int rows = p.M - m0 < 32 * RT ? p.M - m0 : 32 * RT;          // rows of this wave that exist (<= 0: none)
        f32x4* __restrict__ yo = reinterpret_cast<f32x4*>(Yo) + (size_t)m0 * cq + nq0;
#pragma unroll 4
        for (int i = lane; i < 32 * RT * (BN / 4); i += 64) {
            const int row = i / (BN / 4), col = i - row * (BN / 4);
            if (row < rows && nq0 + col < cq) store16_through(yo + (size_t)row * cq + col, tile[row * TP + col]);
        }
        diag_out();
        return;
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        if (!mv[rt]) continue;
        const int oy = pi[rt] * p.os + py, ox = pj[rt] * p.os + px;
        const size_t obase = (((size_t)pb[rt] * p.OH + oy) * p.OW + ox) * p.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = n0 + nt * 32 + 8 * g + 4 * h;
                if (n < p.Cout) {
                    f32x4 v = (f32x4){acc[rt][nt][4 * g], acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]} + bvs[nt][g];
                    if (p.act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
                    if (p.Y) *reinterpret_cast<f32x4*>(Yo + obase + n) = v;    // (through L2 like the FC rows: conv 16x16 pass 0.7215 -> 0.7330 ms)
                    if (p.Yi) {
                        int4 iv = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
                        *reinterpret_cast<int4*>(p.Yi + obase + n) = iv;
                    }
                }
            }
    }
    diag_out();
}

// The launch: a 1-D grid over the gx x gy x gz tiles in the order a 3-D grid is dispatched (x fastest; tile i runs on XCD i % 8 either
// way).  Plain launch: one workgroup per tile.  PERSISTENT launch (launch_f32: fewer workgroups than tiles, at most one or two per
// CU): workgroup w takes tiles w, w + G, w + 2 G, ... -- see launch_f32 for when.
template <int RT, int NT, int KC, bool FUSE, int SEQ = 0>
__global__ __launch_bounds__(256) void tapgemm_f32_kernel(const TapGemmParams p)
{
    touch_kernargs<sizeof(TapGemmParams)>();
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    const int gx = p.grid_x, gy = p.grid_y, ntiles = gx * gy * p.grid_z;
    for (int lin = blockIdx.x; lin < ntiles; lin += gridDim.x) {
        if (lin != (int)blockIdx.x) __syncthreads();   // the previous tile's last fragment reads (and its LDS epilogue) before this tile's first LDS-DMA
        const int bx = lin % gx, rest = lin / gx;
        f32_tile<RT, NT, KC, FUSE, SEQ>(p, lds, bx, rest % gy, rest / gy, gx, gy);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Output layer (<= 64 outputs) of an FC net from STORED activations, in the fused kernel's order: K segment t = columns
// [160 t, 160 t + 160) of the last hidden layer, one chain of 80 k-steps per segment (5 column tiles x (g, r), the two k-halves
// h = hidden units 8g + r and 8g + 4 + r of the tile), partial sums [segment][M][64] for fuse_reduce_kernel.  One wave per
// (32 rows, segment).  For the batch sizes whose last hidden layer does not run on the fused 128 x 160 tile.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fc_out_f32_kernel(const TapGemmParams p)
{
    constexpr int NT = 5, OT = 2;
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 32 + l31, seg = blockIdx.y, n0 = seg * 32 * NT;
    const bool mv = m < p.M;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)p.chunk_begin[1] * 4u * (unsigned)p.Npad * 16u, 0x00020000);
    f32x16 acc2[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ot++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc2[ot][i] = 0.f;
    f32x4 x[NT][4], w2[NT][4][OT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int n = n0 + 32 * nt + 8 * g + 4 * h;
            // activations past Cin (the last segment: 1120 + 160 > 1200) and rows past M read zeros
            const unsigned xo = (mv && n < p.Cin) ? ((unsigned)m * (unsigned)p.Cin + (unsigned)n) << 2 : 0x80000000u;
            x[nt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo, 0, 0));
#pragma unroll
            for (int ot = 0; ot < OT; ot++)
                w2[nt][g][ot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)((n >> 2) * p.Npad + ot * 32 + l31) << 4, 0, 0));
        }
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int ot = 0; ot < OT; ot++) acc2[ot] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[nt][g][ot][r], x[nt][g][r], acc2[ot], 0, 0, 0);
    if (mv) {
        float* dst = p.part + ((size_t)seg * p.M + m) * 64;
#pragma unroll
        for (int ot = 0; ot < OT; ot++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                *reinterpret_cast<f32x4*>(dst + ot * 32 + 8 * g + 4 * h) = (f32x4){acc2[ot][4 * g], acc2[ot][4 * g + 1], acc2[ot][4 * g + 2], acc2[ot][4 * g + 3]};
    }
}

hipError_t launch_fc_out_f32(const TapGemmParams& p, hipStream_t s, int* segments)
{
    if (p.M <= 0) return hipSuccess;
    const int segs = (p.Cin + 159) / 160;
    if (segments) *segments = segs;
    pnn_launch(fc_out_f32_kernel, dim3((p.M + 31) / 32, segs), dim3(64), 0, s, p);
    return hipGetLastError();
}

// The same output layer AND its reduction in ONE launch, for small M (round 5): a single-block call is a chain of dependent launches
// that cost 3-4 us each on the host and more on the device once the batching service's five width workers share the chip
// (tools/corun_threads.cpp), so fc_out_f32_kernel + fuse_reduce_kernel become one workgroup of 8 waves per (32-row tile, 32-column tile):
// wave z runs fc_out_f32_kernel's chain for K segment z of that tile (half the chain of that kernel's wave, which owns both column
// tiles), all its operands requested up front (two waves per SIMD: 256 registers per lane; a 16-wave form that owned both column
// tiles had 128, fetched block by block and paid the memory latency five times: 16.6 us), the partial sums meet in LDS and are added
// in segment order, + bias, HM epilogue -- fuse_reduce_kernel's arithmetic.
struct FcOutF32Args { TapGemmParams p; DoneSignal done; };
__global__ __launch_bounds__(512) void fc_out_f32_small_kernel(const FcOutF32Args a)
{
    touch_kernargs<sizeof(FcOutF32Args)>();
#ifdef PNN_F32_DIAG
    const unsigned long long de0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dr0 = 0, dr1 = 0;
#endif
    const TapGemmParams& p = a.p;
    constexpr int NT = 5;
    __shared__ __attribute__((aligned(16))) float part[8][32][32];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int z = __builtin_amdgcn_readfirstlane(tid >> 6), ot = blockIdx.y;
    const int mblk = blockIdx.x * 32;
    const int segs = (p.Cin + 159) / 160;
    if (z < segs) {
        const int m = mblk + l31, n0 = z * 32 * NT;
        const bool mv = m < p.M;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)p.chunk_begin[1] * 4u * (unsigned)p.Npad * 16u, 0x00020000);
        f32x16 acc2;
#pragma unroll
        for (int i = 0; i < 16; i++) acc2[i] = 0.f;
        f32x4 x[NT][4], w2[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = n0 + 32 * nt + 8 * g + 4 * h;
                const unsigned xo = (mv && n < p.Cin) ? ((unsigned)m * (unsigned)p.Cin + (unsigned)n) << 2 : 0x80000000u;
                x[nt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo, 0, 0));
                w2[nt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)((n >> 2) * p.Npad + ot * 32 + l31) << 4, 0, 0));
            }
        __builtin_amdgcn_sched_barrier(0);           // every request before the first MFMA: ONE memory latency, not one per block
#ifdef PNN_F32_DIAG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dr0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[nt][g][r], x[nt][g][r], acc2, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; g++)
            *reinterpret_cast<f32x4*>(&part[z][l31][8 * g + 4 * h]) = (f32x4){acc2[4 * g], acc2[4 * g + 1], acc2[4 * g + 2], acc2[4 * g + 3]};
#ifdef PNN_F32_DIAG
        dr1 = __builtin_amdgcn_s_memrealtime();
#endif
    }
    __syncthreads();
    const int mr = tid >> 3, nl = (tid & 7) << 2, n = ot * 32 + nl;
    const int mg = mblk + mr;
    if (tid < 256 && mg < p.M && n < p.Cout) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < segs; t++) sum += *reinterpret_cast<const f32x4*>(&part[t][mr][nl]);
        const f32x4 v = sum + *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.Y) *reinterpret_cast<f32x4*>(p.Y + (size_t)mg * p.Cout + n) = v;
        if (p.Yi) *reinterpret_cast<int4*>(p.Yi + (size_t)mg * p.Cout + n) = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
    }
#ifdef PNN_F32_DIAG
    const unsigned long long dr2 = __builtin_amdgcn_s_memrealtime();
#endif
    signal_done(a.done);
#ifdef PNN_F32_DIAG
    if (p.Xlo && tid == 0) {
        unsigned long long* d = (unsigned long long*)p.Xlo + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        d[1] = dr1 - dr0; d[3] = dr0; d[4] = de0; d[5] = __builtin_amdgcn_s_memrealtime(); d[6] = dr2;
    }
#endif
}

// Round 6: the same output layer, segments and reduction on v_mfma_f32_16x16x4_f32 -- the last kernel of a single-block FC call was its
// slowest (profiles/r05_batch1_w8_f32_timeline.txt: 10.9 us of 34.9) with its 160-deep segment chains as 80 dependent 64-cycle
// instructions.  The chain of segment z, per (column tile nt, group g) of the fused kernel: hidden units 8g + 0, 4, 1, 5, 2, 6, 3, 7
// (step r of the 32x32x2 instruction adds unit 8g + r from lane half 0, then 8g + 4 + r from lane half 1) -- through the 16x16x4
// form that is two instructions whose lane groups q = 0..3 supply units 8g + 4 (q & 1) + 2 i + (q >> 1), i = 0, 1: 40 dependent
// instructions of 32 cycles per segment instead of 80 of 64, the same fmaf chain bit for bit (the f32 matrix instructions are a
// k-ordered fmaf chain, one rounding per product: pnn_gemm_f32_small.hip).  One workgroup of 8 waves per 16 x 16 output tile (a
// single 8x8 block: four workgroups instead of two), wave z = K segment z; each lane requests its 20 + 20 sixteen-byte operand pieces
// up front (ONE memory latency) and keeps the two elements of each that its lane group multiplies; the partial sums meet in LDS and
// wave 0 adds them in segment order, + bias, HM epilogue -- fuse_reduce_kernel's arithmetic.
__global__ __launch_bounds__(512) void fc_out_f32_chain_kernel(const FcOutF32Args a)
{
    touch_kernargs<sizeof(FcOutF32Args)>();
#ifdef PNN_F32_DIAG
    const unsigned long long de0 = __builtin_amdgcn_s_memrealtime();
#endif
    const TapGemmParams& p = a.p;
    constexpr int NT = 5;
    __shared__ __attribute__((aligned(16))) float part[8][16][16];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int z = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mblk = blockIdx.x * 16, nblk = blockIdx.y * 16;
    const int segs = (p.Cin + 159) / 160;
    const bool odd = (q >> 1) != 0;                  // this lane group multiplies elements 1 and 3 of its pieces (else 0 and 2)
#ifdef PNN_F32_DIAG
    unsigned long long dr0 = 0, dr1 = 0;
#endif
    // wave 0 finishes the tile: its bias piece is requested now, not behind the barrier (one exposed memory latency less)
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (z == 0 && nblk + 4 * q < p.Cout) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nblk + 4 * q);
    if (z < segs) {
        const int m = mblk + l15, n0 = z * 32 * NT;
        const bool mv = m < p.M;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, (unsigned)p.chunk_begin[1] * 4u * (unsigned)p.Npad * 16u, 0x00020000);
        f32x4 x[NT][4], w2[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int n = n0 + 32 * nt + 8 * g + 4 * (q & 1);
                // activations past Cin (the last segment: 1120 + 160 > 1200) and rows past M read zeros, weights past the pack too
                const unsigned xo = (mv && n < p.Cin) ? ((unsigned)m * (unsigned)p.Cin + (unsigned)n) << 2 : 0x80000000u;
                x[nt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo, 0, 0));
                w2[nt][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (unsigned)((n >> 2) * p.Npad + nblk + l15) << 4, 0, 0));
            }
        __builtin_amdgcn_sched_barrier(0);           // every request before the first use: ONE memory latency, not one per piece
        float xs[NT][4][2], ws[NT][4][2];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                xs[nt][g][0] = odd ? x[nt][g][1] : x[nt][g][0]; xs[nt][g][1] = odd ? x[nt][g][3] : x[nt][g][2];
                ws[nt][g][0] = odd ? w2[nt][g][1] : w2[nt][g][0]; ws[nt][g][1] = odd ? w2[nt][g][3] : w2[nt][g][2];
            }
        __builtin_amdgcn_sched_barrier(0);
#ifdef PNN_F32_DIAG
        dr0 = __builtin_amdgcn_s_memrealtime();
#endif
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[nt][g][0], xs[nt][g][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[nt][g][1], xs[nt][g][1], acc, 0, 0, 0);
            }
        // lane (l15, q): row m = l15, outputs nblk + 4 q + r
        *reinterpret_cast<f32x4*>(&part[z][l15][4 * q]) = acc;
#ifdef PNN_F32_DIAG
        dr1 = __builtin_amdgcn_s_memrealtime();
#endif
    }
    __syncthreads();
    const int mg = mblk + l15, n = nblk + 4 * q;
    if (z == 0 && mg < p.M && n < p.Cout) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < segs; t++) sum += *reinterpret_cast<const f32x4*>(&part[t][l15][4 * q]);
        const f32x4 v = sum + bias4;
        if (p.Y) *reinterpret_cast<f32x4*>(p.Y + (size_t)mg * p.Cout + n) = v;
        if (p.Yi) *reinterpret_cast<int4*>(p.Yi + (size_t)mg * p.Cout + n) = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean), hm_round(v[3], p.mean));
    }
#ifdef PNN_F32_DIAG
    const unsigned long long dr2 = __builtin_amdgcn_s_memrealtime();
#endif
    // wave 0 wrote all of the workgroup's results: it alone signals (per-workgroup flag words), or the counter form for a caller that asked for it
    if (a.done.per_wg) { if (z == 0) signal_done_by_wave(a.done, blockIdx.y * gridDim.x + blockIdx.x); }
    else signal_done(a.done);
#ifdef PNN_F32_DIAG
    if (p.Xlo && tid == 0) {
        unsigned long long* d = (unsigned long long*)p.Xlo + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        d[1] = dr1 - dr0; d[3] = dr0; d[4] = de0; d[5] = __builtin_amdgcn_s_memrealtime(); d[6] = dr2;
    }
#endif
}

// p: the output layer as a one-tap GEMM (X = f32 activations [M][Cin], Wp = its f32 pack, bias, mean, Y / Yi).  false: not this kernel's case.
bool fc_out_f32_small_fits(const TapGemmParams& p) { return p.ncls == 1 && p.SH * p.SW == 1 && p.Cout <= 64 && p.Cout % 4 == 0 && (p.Cin + 159) / 160 <= 8 && p.M > 0; }

hipError_t launch_fc_out_f32_small(const TapGemmParams& p, hipStream_t s, const DoneSignal& done, bool round5_form)
{
    if (!fc_out_f32_small_fits(p)) return hipErrorInvalidValue;
    const FcOutF32Args a{p, done};
    if (round5_form) pnn_launch(fc_out_f32_small_kernel, dim3((unsigned)((p.M + 31) / 32), (unsigned)((p.Cout + 31) / 32)), dim3(512), 0, s, a);
    else pnn_launch(fc_out_f32_chain_kernel, dim3((unsigned)((p.M + 15) / 16), (unsigned)((p.Cout + 15) / 16)), dim3(512), 0, s, a);
    return hipGetLastError();
}

// The K segments of a layer, summed in order: Y = act(bias + ((part[0] + part[1]) + ...)), one thread per 4 consecutive channels.
__global__ __launch_bounds__(256) void seg_reduce_kernel(const float* __restrict__ part, int nseg, size_t n4, size_t stride4, int cq, const float* __restrict__ bias,
                                                         int act, float* __restrict__ Y)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(part) + i;
    f32x4 acc = src[0];
    for (int sgm = 1; sgm < nseg; sgm++) acc += src[(size_t)sgm * stride4];
    f32x4 v = acc + reinterpret_cast<const f32x4*>(bias)[i % (size_t)cq];
    if (act) { v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]); }
    reinterpret_cast<f32x4*>(Y)[i] = v;
}

hipError_t launch_seg_reduce(const float* part, int nseg, size_t n, int Cout, const float* bias, int act, float* Y, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    if (nseg < 1 || n % 4 || Cout % 4) return hipErrorInvalidValue;
    const size_t n4 = n / 4;
    hipLaunchKernelGGL(seg_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, nseg, n4, n4, Cout / 4, bias, act, Y);   // (not pnn_launch: never the timed GEMM)
    return hipGetLastError();
}

// {rt, nt, kc, fuse-capable}
#define PNN_F32_CFGS(X) \
    X(1, 5, 4) X(1, 5, 2) X(1, 4, 4) X(1, 4, 2) X(1, 3, 4) X(1, 2, 4) X(1, 2, 2) X(1, 1, 4) X(1, 1, 2) \
    X(2, 4, 2) X(2, 2, 4) X(2, 2, 2) X(2, 1, 4) X(2, 3, 2)

static const TileCfg kCfgsF32[] = {
#define X(rt, nt, kc) {rt, nt, kc, 32},
    PNN_F32_CFGS(X)
#undef X
};

// Registers per lane (arch + acc) of the non-fused instantiations, in the order of PNN_F32_CFGS, as the compiler allocated them
// (llvm-readelf --notes on the code object, .vgpr_count): what bounds the waves per SIMD -- 512 / this, whole waves -- in
// choose_cfg_f32's residency estimate.  Re-read after a change of the kernel.
static const int kRegsF32[] = {360, 360, 296, 296, 224, 176, 152, 116, 92, 492, 288, 252, 208, 360};
static_assert(sizeof(kRegsF32) / sizeof(kRegsF32[0]) == sizeof(kCfgsF32) / sizeof(kCfgsF32[0]), "one register count per tile");
int tapgemm_f32_regs(int idx) { return kRegsF32[idx]; }

int tapgemm_f32_num_cfgs() { return (int)(sizeof(kCfgsF32) / sizeof(kCfgsF32[0])); }
TileCfg tapgemm_f32_cfg(int idx) { return kCfgsF32[idx]; }
size_t tapgemm_f32_lds_bytes(const TileCfg& t, bool fuse, bool row_out)
{
    const size_t ring = (size_t)3 * t.kc * 4 * 32 * t.nt, out_tile = (size_t)128 * t.rt * (8 * t.nt + 1);   // pieces: weight stages | the FC epilogue's row tiles
    return (std::max(ring, (row_out && !fuse) ? out_tile : (size_t)0) + (fuse ? (size_t)8 * t.nt * 64 : 0)) * 16;
}
bool tapgemm_f32_can_fuse(int idx) { return kCfgsF32[idx].rt == 1 && kCfgsF32[idx].nt == 5; }

template <int RT, int NT, int KC>
static hipError_t launch_f32(const TapGemmParams& p0, bool fuse, hipStream_t s)
{
    const bool seq = p0.nseg > 1 && p0.seg_seq;
    if (fuse && p0.nseg > 1 && !seq) return hipErrorInvalidValue;       // planes of partial sums cannot feed the fused output layer
    dim3 grid((p0.M + 128 * RT - 1) / (128 * RT), (p0.Cout + 32 * NT - 1) / (32 * NT), p0.ncls * (!seq && p0.nseg > 1 ? p0.nseg : 1));
    const TileCfg t{RT, NT, KC, 32};
    TapGemmParams p = p0;
    p.pm_groups = 0;
    g_last_issued_frac = 1.0;
    // (not with a last block group that is mostly padding rows: 64 blocks of the 64x64 net in 128-row tiles ran 3 % slower)
    const long nblk = p0.M / (p0.SH * p0.SW), bm = 128 * RT;
    if (!fuse && p0.pm_groups >= 0 && p0.SH * p0.SW > 1 && (p0.pm_groups == 1 || (nblk + bm - 1) / bm * bm * 100 <= nblk * 115)) {
        TapGemmParams q = p0;
        q.W2p = nullptr;                              // (the planner reads the fused-layer field of the ring kernel's launches)
        // a block group's input maps may exceed an XCD's 4 MB L2 here (the split kernels' planner stops at 4.5 MB): at a third of their
        // matrix rate the taps' re-reads that spill to the MALL are covered.  Same-box sweep, conv 16x16 f32 at batch 1024: limit 4.5 MB
        // 0.778 ms, 7 MB (lets the 8x24x64 maps of the two biggest layers in: 6.3 MB per group) 0.742-0.746, 12 MB 0.737-0.749;
        // conv 32x32 at batch 256: 1.089 / 1.077-1.083 / 1.128-1.130 ms
        static const double l2_mb = getenv("PNN_F32_PM_L2_MB") ? atof(getenv("PNN_F32_PM_L2_MB")) : 7.0;
        const PmPlan& plan = position_major_plan(q, 128 * RT, 32 * NT, KC, tapgemm_f32_lds_bytes(t, false, false), l2_mb);
        g_last_issued_frac = plan.use ? plan.live_frac : 1.0;
        if (plan.use) {
            p.pm_groups = plan.groups;
            p.nblk = p0.M / (p0.SH * p0.SW);
            for (int i = 0; i < 16; i++) p.pos_order[i] = plan.order[i];
            grid.x = (unsigned)(plan.groups * p0.SH * p0.SW);
            static const bool debug = getenv("PNN_DEBUG") != nullptr;
            if (debug) fprintf(stderr, "[pnn] f32 %dx%d: position-major tiles, %d block groups x %d positions\n", 128 * RT, 32 * NT, plan.groups, p0.SH * p0.SW);
        }
    }
    p.grid_x = (int)grid.x; p.grid_y = (int)grid.y; p.grid_z = (int)grid.z;
    const long ntiles = (long)grid.x * grid.y * grid.z;
    if (ntiles > 0x7fffffffL) return hipErrorInvalidValue;
    // PERSISTENT workgroups (W > 0): 256 W workgroups, at most W per CU (the LDS request says so), each running its tiles one after
    // the other.  p0.persist: 0 = never, N > 0 = N per CU whenever there are more tiles, -1 = by this rule: TWO per CU for launches of
    // more than two and at most six tiles per CU.  That is where a plain launch has every workgroup resident from the start (or a last
    // round that fills half the chip) -- three or four waves per SIMD (the 32x32x2 instruction gives its full rate to one or two) that
    // go through start-up and epilogue together; two persistent workgroups per CU drift apart after their first tile.  Same box, conv
    // 16x16 f32 at batch 1024 (its big layers: 768 tiles) 0.768 -> 0.748 ms, at batch 1536 (1152 tiles) 1.148 -> 1.079, conv 32x32 at
    // batch 256 / 384 1.151 -> 1.124 / 1.714 -> 1.664, conv 64x64 at 64 / 128 1.268 -> 1.244 / 2.359 -> 2.305; one per CU is slower
    // (0.787: nothing hides a tile's start-up and epilogue), three change nothing; at six tiles per CU both launches are equal (conv
    // 16x16 at batch 2048, conv 32x32 at 512), with more the hardware's own turnover does better (conv 16x16 at batch 4096: 2.74 -> 2.83
    // ms with two persistent workgroups), and up to two per CU there is nothing to gain.
    const DeviceInfo& di = device_info();
    int idx_self = 0;
    for (int i = 0; i < tapgemm_f32_num_cfgs(); i++) if (kCfgsF32[i].rt == RT && kCfgsF32[i].nt == NT && kCfgsF32[i].kc == KC) idx_self = i;
    const int by_regs = std::max(1, 512 / kRegsF32[idx_self]);       // workgroups (one wave per SIMD each) the register file keeps resident
    int W = p0.persist >= 0 ? p0.persist : (ntiles > 2L * di.cus && ntiles <= 6L * di.cus ? 2 : 0);
    W = std::min(W, by_regs);                         // never more persistent workgroups than are resident: the rest would run as a second round
    size_t lds = tapgemm_f32_lds_bytes(t, fuse, p0.SH * p0.SW == 1);
    dim3 g1((unsigned)ntiles);
    if (W > 0 && ntiles > (long)di.cus * W && lds * (size_t)W <= di.lds) {   // (a tile whose ring leaves no room for W workgroups per CU: plain launch)
        g1.x = (unsigned)di.cus * (unsigned)W;
        lds = std::max(lds, (size_t)(di.lds / (W + 1) + 1024) / 16 * 16);
    }
    // the LDS attribute belongs to (function, device): set once per device, and a failure is the launch's error, not an opaque launch failure later
    auto set_attr = [&](const void* fn, int (&done)[16]) -> hipError_t {
        const int dev = di.dev >= 0 && di.dev < 16 ? di.dev : 0;
        if (done[dev]) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)di.lds);
        if (e == hipSuccess) done[dev] = 1;
        return e;
    };
    const bool one_tap = p0.ncls == 1 && p0.tap_begin[1] - p0.tap_begin[0] == 1;
    // whole stages per segment; a one-tap layer: an EVEN number of them, and a last segment that is not empty (SEQ = 2, see f32_tile)
    if (seq && (one_tap ? (p0.seg_chunks == 0 || p0.seg_chunks % (2 * KC) || (long)(p0.nseg - 1) * p0.seg_chunks >= p0.Cin / 16) : ((p0.Cin / 16) % KC) != 0)) return hipErrorInvalidValue;
    if constexpr (RT == 1 && NT == 5) {
        if (fuse && seq) {
            if (!one_tap) return hipErrorInvalidValue;
            static int done[16] = {};
            const hipError_t e = set_attr((const void*)tapgemm_f32_kernel<RT, NT, KC, true, 2>, done);
            if (e != hipSuccess) return e;
            pnn_launch(tapgemm_f32_kernel<RT, NT, KC, true, 2>, g1, dim3(256), lds, s, p);
            return hipGetLastError();
        }
        if (fuse) {
            static int done[16] = {};
            const hipError_t e = set_attr((const void*)tapgemm_f32_kernel<RT, NT, KC, true>, done);
            if (e != hipSuccess) return e;
            pnn_launch(tapgemm_f32_kernel<RT, NT, KC, true>, g1, dim3(256), lds, s, p);
            return hipGetLastError();
        }
    }
    if (fuse) return hipErrorInvalidValue;
    if (seq && one_tap) {
        static int done[16] = {};
        const hipError_t e = set_attr((const void*)tapgemm_f32_kernel<RT, NT, KC, false, 2>, done);
        if (e != hipSuccess) return e;
        pnn_launch(tapgemm_f32_kernel<RT, NT, KC, false, 2>, g1, dim3(256), lds, s, p);
        return hipGetLastError();
    }
    if (seq && ((p0.Cin / 16) / KC) % 2 == 0) {       // segments of whole taps, an even number of stages per tap: the two-stage inner loops (SEQ = 2)
        static int done[16] = {};
        const hipError_t e = set_attr((const void*)tapgemm_f32_kernel<RT, NT, KC, false, 2>, done);
        if (e != hipSuccess) return e;
        pnn_launch(tapgemm_f32_kernel<RT, NT, KC, false, 2>, g1, dim3(256), lds, s, p);
        return hipGetLastError();
    }
    if (seq) {
        static int done[16] = {};
        const hipError_t e = set_attr((const void*)tapgemm_f32_kernel<RT, NT, KC, false, 1>, done);
        if (e != hipSuccess) return e;
        pnn_launch(tapgemm_f32_kernel<RT, NT, KC, false, 1>, g1, dim3(256), lds, s, p);
        return hipGetLastError();
    }
    static int done[16] = {};
    const hipError_t e = set_attr((const void*)tapgemm_f32_kernel<RT, NT, KC, false>, done);
    if (e != hipSuccess) return e;
    pnn_launch(tapgemm_f32_kernel<RT, NT, KC, false>, g1, dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_tapgemm_f32(const TapGemmParams& p, int idx, bool fuse, hipStream_t s)
{
    if (p.M <= 0) return hipSuccess;
    int i = 0;
#define X(rt, nt, kc) if (idx == i++) return launch_f32<rt, nt, kc>(p, fuse, s);
    PNN_F32_CFGS(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace pnn
