// Split-precision tap GEMM: f32-class accuracy on the f16 matrix cores (v_mfma_f32_32x32x16_f16).
//
// Every f32 operand x is carried as two halves, x = hi + lo with hi = (f16) x and lo = (f16)(x - hi), i.e. 22
// mantissa bits; a product a*w is formed as a_hi*w_hi + a_lo*w_hi + a_hi*w_lo (each f16 x f16 product is exact in
// the MFMA's f32 accumulator; only the ~2^-22 lo*lo term is dropped).  Three f16 MFMAs (3 x 32 cycles for a
// 32x32x16 block) replace eight f32 MFMAs (8 x 64 cycles): 5.3x fewer matrix-pipe cycles for the same result
// within float rounding.  Weights are pre-scaled by a per-layer power of two so that their lo halves stay in the
// f16 normal range; the scale is undone exactly in the epilogue.
//
// Same contract as tapgemm_kernel (TapGemmParams): FC layers, convolutions, transposed convolutions of
// pnn/components.py:10-261.  Activations arrive as two f16 planes [pixel][Cin] (X = hi, Xlo = lo), written that way
// by the producing layer's epilogue (Yhi) or by split_kernel for network inputs.
//
//   workgroup = 256 threads = 4 waves arranged WM x WN; a wave owns a (32*RT) x (32*NT) tile of the
//   (32*RT*WM) x (32*NT*WN) workgroup tile.
//   MFMA roles: "A" = weights (i = n), "B" = activations (j = m); lane l = (l&31, h = l>>5) supplies k = 8h + j.
//   Packed weights per 16-deep chunk: [hl = hi/lo][h][Npad][8 x f16] = four planes of Npad x 16 bytes -- the same
//   plane geometry as the f32 kernels' [q = 4][Npad][4 x f32], so staging and LDS addressing are shared.
#include "pnn_kernels.h"
#include <type_traits>
#include "pnn_device_common.h"

namespace pnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int RT, int NT, int KC, int WM>
__global__ __launch_bounds__(256) void tapgemm_sp_kernel(const TapGemmParams p)
{
    touch_kernargs<sizeof(TapGemmParams)>();   // see pnn_device_common.h
    constexpr int WN = 4 / WM;                      // waves along N (WM x WN = 4 waves)
    constexpr int BM = 32 * RT * WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int E = 4 * BN;                       // 16-byte slots per staged weight chunk
    constexpr int NLD = (E + 255) / 256;
    constexpr int PPR = KC * 4;                     // 16-byte pieces per activation row and stage (64 B per chunk)
    constexpr int APITCH = PPR + 1;                 // LDS row pitch in 16-byte slots: +1 keeps b128 reads conflict-free
    constexpr int NLA = BM * PPR / 256;             // activation pieces per thread and stage
    __shared__ f32x4 Bs[2][KC][E];
    __shared__ f32x4 As[2][BM * APITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int mblk = blockIdx.x * BM;
    const int SP = p.SH * p.SW;

    // Rows this THREAD stages (full 64*KC-byte runs per row: PPR consecutive lanes share a row).
    int lb[NLA], li[NLA], lj[NLA];
    bool lv[NLA];
    const int lpiece = tid % PPR;
#pragma unroll
    for (int r = 0; r < NLA; r++) {
        const int mg = mblk + (tid + 256 * r) / PPR;
        lv[r] = mg < p.M;
        const int mc = lv[r] ? mg : 0;
        const int b = mc / SP;
        const int q = mc - b * SP;
        lb[r] = b;
        li[r] = q / p.SW;
        lj[r] = q - li[r] * p.SW;
    }

    const int cpt = p.Cin >> 4;
    const int t0 = p.tap_begin[cls], t1 = p.tap_begin[cls + 1];
    const int nchunks = (t1 - t0) * cpt;
    const int nstages = (nchunks + KC - 1) / KC;
    const f32x4* __restrict__ Wg = reinterpret_cast<const f32x4*>(p.Wp) + (size_t)p.chunk_begin[cls] * 4 * p.Npad;

    f32x16 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[rt][nt][i] = 0.f;

    // Activations: [pixel][Cin/16][hi 16 x f16 | lo 16 x f16] -- 64 B per 16-deep chunk, so a stage is one
    // contiguous 64*KC-byte run per row.  Out-of-image taps / rows past M read zeros (descriptor range check).
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
    unsigned aoff[NLA];
    auto tap_setup = [&](int tp) {
        const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
#pragma unroll
        for (int r = 0; r < NLA; r++) {
            const int iy = li[r] * p.a + dy, ix = lj[r] * p.a + dx;
            const bool ok = lv[r] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            const unsigned off = (((unsigned)((lb[r] * p.IH + iy) * p.IW + ix) * (unsigned)p.Cin) << 2) + (lpiece << 4);
            aoff[r] = ok ? off : 0x80000000u;
        }
    };
    auto load_a = [&](int cc, f32x4 (&dst)[NLA]) {
        // the tail stage of a single-tap layer may run past the last chunk: those pieces multiply zero weights,
        // clamp them onto the last valid chunk so they stay finite.
        int sc = cc;
        if (cc + (lpiece >> 2) >= cpt) sc = cpt - 1 - (lpiece >> 2);
#pragma unroll
        for (int r = 0; r < NLA; r++)
            dst[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[r] + (unsigned)(sc << 6), 0, 0));
    };
    auto store_a = [&](int buf, const f32x4 (&src)[NLA]) {
#pragma unroll
        for (int r = 0; r < NLA; r++) As[buf][((tid + 256 * r) / PPR) * APITCH + lpiece] = src[r];
    };
    unsigned bsrc[NLD];                              // byte offsets into the packed weights (buffer loads: a 32-bit offset per lane
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0xffffffffu, 0x00020000);   // issues faster than a 64-bit flat address, see pnn_gemm_ring.hip)
    int bdst[NLD];
#pragma unroll
    for (int r = 0; r < NLD; r++) {
        int e = tid + 256 * r;
        if (E % 256 != 0) e = e < E ? e : E - 1;
        const int qq = e / BN, nn = e - qq * BN;
        bsrc[r] = (unsigned)((qq * p.Npad + n0 + nn) << 4);
        bdst[r] = e;
    }
    const unsigned bstride = (unsigned)(4 * p.Npad) << 4;   // bytes per packed chunk
    auto load_b = [&](int stage, f32x4 (&dst)[KC][NLD]) {
#pragma unroll
        for (int j = 0; j < KC; j++)
#pragma unroll
            for (int r = 0; r < NLD; r++) dst[j][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, bsrc[r] + (unsigned)(stage * KC + j) * bstride, 0, 0));
    };
    auto store_b = [&](int buf, const f32x4 (&src)[KC][NLD]) {
#pragma unroll
        for (int j = 0; j < KC; j++)
#pragma unroll
            for (int r = 0; r < NLD; r++) Bs[buf][j][bdst[r]] = src[j][r];
    };
    auto read_frags = [&](int buf, int j, f32x4 (&wf)[NT][2], f32x4 (&af)[RT][2]) {   // [..][0] = hi, [..][1] = lo
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            wf[nt][0] = Bs[buf][j][(0 + h) * BN + wn * (32 * NT) + nt * 32 + l31];
            wf[nt][1] = Bs[buf][j][(2 + h) * BN + wn * (32 * NT) + nt * 32 + l31];
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int row = wm * (32 * RT) + rt * 32 + l31;
            af[rt][0] = As[buf][row * APITCH + j * 4 + 0 + h];
            af[rt][1] = As[buf][row * APITCH + j * 4 + 2 + h];
        }
    };
    auto mfma_chunk = [&](const f32x4 (&wf)[NT][2], const f32x4 (&a)[RT][2], int part) {
        // part 0: hi*hi;  part 1: the two cross terms
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const f16x8 whi = __builtin_bit_cast(f16x8, wf[nt][0]), wlo = __builtin_bit_cast(f16x8, wf[nt][1]);
                const f16x8 ahi = __builtin_bit_cast(f16x8, a[rt][0]), alo = __builtin_bit_cast(f16x8, a[rt][1]);
                if (part == 0) {
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi, acc[rt][nt], 0, 0, 0);
                } else {
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo, acc[rt][nt], 0, 0, 0);
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi, acc[rt][nt], 0, 0, 0);
                }
            }
    };

    // Two-stage pipeline, one barrier per KC chunks: both operands travel global -> registers -> LDS in full
    // cache lines; MFMA fragments are read from LDS (double-buffered in registers across chunks).
    // (Fetching two stages ahead through a second register set was measured 30-50 % SLOWER: the extra 32 VGPRs
    // cost a resident workgroup per CU, which hides more latency than the deeper prefetch does.)
#ifdef PNN_SP_DIAG
    // Diagnostic build only (`make diag`, tools/sp_prof.py): per-phase cycle sums of wave 0 of every workgroup, written
    // to the buffer passed in p.Xlo.  Findings on FC 1200x1200 (tile 128x128, 768 MFMA cycles per stage): ~740 cycles
    // to issue the eight 1-KiB loads of a stage (the CU's 64 B/clk vector-memory path, shared by 8 waves), ~870 in the
    // MFMA block, ~460 waiting for the loads + LDS stores, ~140 at the barrier.  Interleaving the loads with the MFMAs
    // only moves the stall (in-order issue), prefetching two stages ahead costs a resident workgroup: both measured.
    unsigned long long dg_t = 0, dg_issue = 0, dg_mfma = 0, dg_store = 0, dg_bar = 0;
#define DG_STAMP(acc)                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long now_;                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory");       \
        acc += now_ - dg_t; dg_t = now_;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#else
#define DG_STAMP(acc) do {} while (0)
#endif
    f32x4 a_stage[NLA], b_stage[KC][NLD];
    int t = t0, cc = 0;
    tap_setup(p.tap[t0]);
    int tp_next = p.tap[t0 + 1 < t1 ? t0 + 1 : t0];
    load_a(0, a_stage);
    load_b(0, b_stage);
    store_a(0, a_stage);
    store_b(0, b_stage);
    __syncthreads();
#ifdef PNN_SP_DIAG
    { unsigned long long d0_ = 0; DG_STAMP(d0_); (void)d0_; }
#endif
    for (int s = 0; s < nstages; s++) {
        const int buf = s & 1;
        f32x4 wf0[NT][2], wf1[NT][2], af0[RT][2], af1[RT][2];
        read_frags(buf, 0, wf0, af0);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = s + 1 < nstages;
        if (more) {
            cc += KC;
            if (cc >= cpt) {
                cc = 0;
                ++t;
                tap_setup(tp_next);
                tp_next = p.tap[t + 1 < t1 ? t + 1 : t];
            }
        }
        load_a(cc, a_stage);
        load_b(more ? s + 1 : s, b_stage);
        __builtin_amdgcn_sched_barrier(0);
        DG_STAMP(dg_issue);
#pragma unroll
        for (int j = 0; j + 1 < KC; j++) {
            if (j & 1) { read_frags(buf, j + 1, wf0, af0); mfma_chunk(wf1, af1, 0); mfma_chunk(wf1, af1, 1); }
            else       { read_frags(buf, j + 1, wf1, af1); mfma_chunk(wf0, af0, 0); mfma_chunk(wf0, af0, 1); }
#ifndef PNN_SP_DIAG
            // issue order: one MFMA, two of the next chunk's fragment reads, ... (two ds_read_b128 fit in the shadow of a
            // 32-cycle MFMA; as a burst they hold the wave and its MFMA pipe ~16 cycles each -- measured in the ring kernel)
#pragma unroll
            for (int i = 0; i < RT + NT; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * RT * NT - (RT + NT), 0);
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if ((KC - 1) & 1) { mfma_chunk(wf1, af1, 0); mfma_chunk(wf1, af1, 1); } else { mfma_chunk(wf0, af0, 0); mfma_chunk(wf0, af0, 1); }
        __builtin_amdgcn_sched_barrier(0);
        DG_STAMP(dg_mfma);
        store_a(buf ^ 1, a_stage);                   // after ALL of the stage's MFMAs: the loads get the whole stage to land
        store_b(buf ^ 1, b_stage);
        DG_STAMP(dg_store);
        __syncthreads();
        DG_STAMP(dg_bar);
    }
#ifdef PNN_SP_DIAG
    if (p.Xlo && tid == 0) {
        unsigned long long* d = (unsigned long long*)p.Xlo + 4 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        d[0] = dg_issue; d[1] = dg_mfma; d[2] = dg_store; d[3] = dg_bar;
    }
#endif

    // Epilogue: undo the weight scale, bias (+ LeakyReLU); f32 and/or split-f16 outputs (and the HM epilogue).
    const int py = p.py[cls], px = p.px[cls];
    // all bias values first: a load placed next to its use cannot be hoisted over the stores in between (the compiler
    // must assume they alias), and every group then pays one L2 round trip (measured in the ring kernel: 600 cycles each)
    f32x4 bvs[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int n = n0 + wn * (32 * NT) + nt * 32 + 8 * g + 4 * h;
            bvs[nt][g] = *reinterpret_cast<const f32x4*>(p.bias + (n < p.Cout ? n : 0));
        }
    // Three copies of the unrolled group loop, chosen once (split-f16 only / f32 only / any mix) -- the loop runs once per launch from a cold instruction cache, its time follows its code footprint
    // (measured in the ring kernel: 7.9k -> 4.1k cycles for 20 groups).
    const bool act = p.act != 0;
    float amax = 0.f;                                // range guard of the split outputs (pnn_device_common.h)
    auto groups = [&](auto kind_tag) {
        constexpr int kKind = decltype(kind_tag)::value;             // 0: split only, 1: f32 only, 2: general
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int mg = mblk + wm * (32 * RT) + rt * 32 + l31;
            if (mg >= p.M) continue;
            const int pbq = mg / SP;
            const int rq = mg - pbq * SP;
            const int piq = rq / p.SW, pjq = rq - piq * p.SW;
            const int oy = piq * p.os + py, ox = pjq * p.os + px;
            const size_t opix = ((size_t)pbq * p.OH + oy) * p.OW + ox;
            const size_t obase = opix * p.Cout;
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int n = n0 + wn * (32 * NT) + nt * 32 + 8 * g + 4 * h;
                    if (n < p.Cout) {
                        const f32x4 bv = bvs[nt][g];
                        f32x4 v = (f32x4){acc[rt][nt][4 * g], acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]} * p.out_scale + bv;
                        if (act) {
                            v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                        }
                        if (kKind == 0) {
                            store_split4(p.Yhi, obase, n, v, amax);   // split output for the next split-precision layer
                        } else if (kKind == 1) {
                            *reinterpret_cast<f32x4*>(p.Y + obase + n) = v;
                        } else {
                            if (p.Y) *reinterpret_cast<f32x4*>(p.Y + obase + n) = v;
                            if (p.Yhi) store_split4(p.Yhi, obase, n, v, amax);
                            if (p.Yi) {
                                int4 iv = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean),
                                                    hm_round(v[3], p.mean));
                                *reinterpret_cast<int4*>(p.Yi + obase + n) = iv;
                            }
                        }
                    }
                }
        }
    };
    if (p.Yhi && !p.Y && !p.Yi) groups(std::integral_constant<int, 0>{});
    else if (p.Y && !p.Yhi && !p.Yi) groups(std::integral_constant<int, 1>{});
    else groups(std::integral_constant<int, 2>{});
    report_range(p.range_flag, amax);
}

// X(rt, nt, kc, wm)
#define PNN_SP_CFGS(X) \
    X(1, 4, 2, 4) X(1, 4, 4, 4) X(1, 2, 2, 4) X(1, 2, 4, 4) X(2, 2, 2, 4) X(2, 4, 2, 4) X(1, 4, 1, 4) X(1, 3, 2, 4) X(1, 5, 2, 4) \
    X(2, 2, 1, 4) X(2, 2, 2, 2) X(2, 1, 2, 2) X(1, 2, 2, 2) X(2, 2, 4, 2) X(2, 3, 2, 2) X(1, 1, 2, 2)

static const TileCfg kCfgsSp[] = {
#define X(rt, nt, kc, wm) {rt, nt, kc, 316, wm},     // mf 316: "3 x f16 32x32x16"
    PNN_SP_CFGS(X)
#undef X
};

int tapgemm_sp_num_cfgs() { return (int)(sizeof(kCfgsSp) / sizeof(kCfgsSp[0])); }
TileCfg tapgemm_sp_cfg(int idx) { return kCfgsSp[idx]; }

template <int RT, int NT, int KC, int WM>
static hipError_t launch_sp(const TapGemmParams& p, hipStream_t s)
{
    constexpr int BM = 32 * RT * WM, BN = 32 * NT * (4 / WM);
    dim3 grid((p.M + BM - 1) / BM, (p.Cout + BN - 1) / BN, p.ncls);
    pnn_launch(tapgemm_sp_kernel<RT, NT, KC, WM>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_tapgemm_sp(const TapGemmParams& p, int idx, hipStream_t s)
{
    if (p.M <= 0) return hipSuccess;
    int i = 0;
#define X(rt, nt, kc, wm) if (idx == i++) return launch_sp<rt, nt, kc, wm>(p, s);
    PNN_SP_CFGS(X)
#undef X
    return hipErrorInvalidValue;
}

// f32 [rows][C] -> split activations [rows][C/16][hi 16 x f16 | lo 16 x f16] for network inputs (C % 16 == 0).
__global__ __launch_bounds__(256) void split_kernel(const float* x, long n, _Float16* out, int* range_flag)
{
    float amax = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        amax = fmaxf(amax, fabsf(v));
        const _Float16 hv = (_Float16)v;
        _Float16* d = out + 2 * (i & ~15L) + (i & 15);
        d[0] = hv;
        d[16] = (_Float16)(v - (float)hv);
    }
    report_range(range_flag, amax);
}

hipError_t launch_split(const float* x, long n, void* hi, void* lo, int* range_flag, hipStream_t s)
{
    (void)lo;
    if (n <= 0) return hipSuccess;
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, (_Float16*)hi, range_flag);
    return hipGetLastError();
}

}  // namespace pnn
