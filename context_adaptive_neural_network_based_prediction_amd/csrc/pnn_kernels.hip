// gfx950 (CDNA4 / MI355X) kernels of the PNN intra-prediction forward pass.
//
// Replaces the TensorFlow-1 CPU kernels behind tensorflow::Session::Run in the reference
// (hm_16_15_substitution/source/Lib/TLibCommon/TComPrediction.cpp:572-579,601-608), i.e. the graph
// built by pnn/components.py:10-261 + pnn/tfutils.py:8-139,395-462, plus the L-shaped context gather
// of hevc/hm_common/c++/source_common/extraction_context.cpp:3-208 and the epilogue of
// TComPrediction.cpp:621-635.
//
// All dense contractions (FC layers, convolutions and transposed convolutions with Cin >= 16) run
// through ONE kernel, tapgemm_kernel: an implicit GEMM over a list of spatial taps on the exact-f32
// matrix cores (v_mfma_f32_16x16x4_f32, 64-lane wavefronts).  Weights stream global -> LDS once per
// workgroup in the pre-packed order the MFMA lane groups consume; activations are gathered straight
// into registers (NHWC keeps a pixel's channels contiguous: 64 B per lane group).
#include "pnn_kernels.h"
#include "pnn_device_common.h"
#include <cstdlib>
#include <algorithm>

namespace pnn {

// ------------------------------------------------------------------------------------------------
// Tap GEMM on f32 MFMA.
//   workgroup = 256 threads = 4 waves; wave w owns rows [16*RT*w, 16*RT*(w+1)) of the BM = 64*RT row
//   tile and all BN = 16*NT columns; accumulators: RT*NT tiles of 16x16 (4 VGPRs each).
//   MFMA operand roles: "A" = weights (i = n), "B" = activations (j = m); lane l = (l&15, q = l>>4)
//   supplies k = 4q + e in step e of a 16-deep chunk, for both operands.  D: lane holds column
//   m = l&15, rows n = 4q + r  ->  one float4 store of 4 consecutive output channels.
//   A pipeline stage is KC chunks (16*KC deep): one barrier per stage, and the global prefetch of
//   stage s+1 has the whole MFMA time of stage s (KC * RT * NT * 128 cycles) to land.
// ------------------------------------------------------------------------------------------------
template <int RT, int NT, int KC>
__global__ __launch_bounds__(256) void tapgemm_kernel(const TapGemmParams p)
{
    touch_kernargs<sizeof(TapGemmParams)>();
    constexpr int BM = 64 * RT;
    constexpr int BN = 16 * NT;
    constexpr int E = 4 * BN;                       // float4 per staged weight chunk
    constexpr int NLD = (E + 255) / 256;
    __shared__ f32x4 Bs[2][KC][E];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int m0 = blockIdx.x * BM + wave * (16 * RT);

    int pb[RT], pi[RT], pj[RT];
    bool mv[RT];
    const int SP = p.SH * p.SW;
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int mg = m0 + rt * 16 + l15;
        mv[rt] = mg < p.M;
        const int mc = mv[rt] ? mg : 0;
        const int b = mc / SP;
        const int r = mc - b * SP;
        pb[rt] = b;
        pi[rt] = r / p.SW;
        pj[rt] = r - pi[rt] * p.SW;
    }

    const int cpt = p.Cin >> 4;                     // 16-deep chunks per tap (a multiple of KC, or one tap)
    const int t0 = p.tap_begin[cls], t1 = p.tap_begin[cls + 1];
    const int nchunks = (t1 - t0) * cpt;
    const int nstages = (nchunks + KC - 1) / KC;    // the packed weights are zero-padded to whole stages
    const f32x4* __restrict__ Wg = reinterpret_cast<const f32x4*>(p.Wp) + (size_t)p.chunk_begin[cls] * 4 * p.Npad;

    f32x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Activation loads go through a buffer descriptor: an out-of-image tap (or a row past M) gets an
    // offset beyond num_records and the hardware returns zeros -- no branch, no select, and therefore no
    // s_waitcnt inside the loop other than the ones the prefetch distance calls for.  The per-lane byte
    // offset is computed once per TAP (aoff); the 16-channel chunk inside the tap rides in the scalar
    // offset, which the range check ignores.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
    unsigned aoff[RT];
    auto tap_setup = [&](int tp) {                  // tp = (dy << 16) | (dx & 0xffff)
        const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int iy = pi[rt] * p.a + dy, ix = pj[rt] * p.a + dx;
            const bool ok = mv[rt] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            const unsigned off = ((unsigned)((pb[rt] * p.IH + iy) * p.IW + ix) * (unsigned)p.Cin + (q << 2)) << 2;
            aoff[rt] = ok ? off : 0x80000000u;
        }
    };
    auto load_a = [&](int cc, f32x4 (&dst)[KC][RT]) {
#pragma unroll
        for (int j = 0; j < KC; j++) {
            const int cj = cc + j < cpt ? cc + j : cpt - 1;   // padding chunk: its weights are zero, any finite data will do
#pragma unroll
            for (int rt = 0; rt < RT; rt++)
                dst[j][rt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[rt], cj << 6, 0));
        }
    };
    unsigned bsrc[NLD];                              // byte offsets into the packed weights (buffer loads: a 32-bit offset per lane
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0xffffffffu, 0x00020000);   // issues faster than a 64-bit flat address, see pnn_gemm_ring.hip)
    int bdst[NLD];
#pragma unroll
    for (int r = 0; r < NLD; r++) {
        // Threads past the end re-load / re-store element E-1: a benign duplicate instead of exec-masked
        // code (whose conditional vmcnt wait the compiler would repeat in front of the next MFMAs).
        int e = tid + 256 * r;
        if (E % 256 != 0) e = e < E ? e : E - 1;
        const int qq = e / BN, nn = e - qq * BN;
        bsrc[r] = (unsigned)((qq * p.Npad + n0 + nn) << 4);
        bdst[r] = e;
    }
    const unsigned bstride = (unsigned)(4 * p.Npad) << 4;   // bytes per packed chunk      // float4 per 16-deep chunk
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
    auto read_bf = [&](int buf, int j, f32x4 (&bf)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) bf[nt] = Bs[buf][j][q * BN + nt * 16 + l15];
    };
    auto mfma_chunk = [&](const f32x4 (&bf)[NT], const f32x4 (&a)[RT], int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; e++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[nt][e], a[rt][e], acc[rt][nt], 0, 0, 0);
    };

    // Two-stage software pipeline.  Order inside an iteration (pinned with sched_barrier, the compiler
    // otherwise sinks the prefetch below the MFMAs):
    //   LDS fragment reads of the stage's first chunk -> issue the global prefetch of stage s+1 (its
    //   address math hides the LDS latency) -> MFMAs of all chunks but half of the last -> write the
    //   staged weights to the other LDS buffer -> the remaining MFMAs (hide the LDS write) -> barrier.
    // The body has no exec-masked code: the last iteration re-fetches the last stage instead of testing.
    f32x4 a_cur[KC][RT], a_nxt[KC][RT], b_stage[KC][NLD];
    int t = t0, cc = 0;
    tap_setup(p.tap[t0]);
    int tp_next = p.tap[t0 + 1 < t1 ? t0 + 1 : t0];  // the NEXT tap's word, fetched a whole tap early
    load_a(0, a_cur);
    load_b(0, b_stage);
    store_b(0, b_stage);
    __syncthreads();
    for (int s = 0; s < nstages; s++) {
        const int buf = s & 1;
        f32x4 bf0[NT], bf1[NT];
        read_bf(buf, 0, bf0);
        __builtin_amdgcn_sched_barrier(0);
        const bool more = s + 1 < nstages;
        if (more) {
            cc += KC;
            if (cc >= cpt) {                        // wave-uniform: the next stage starts the next tap
                cc = 0;
                ++t;
                tap_setup(tp_next);
                tp_next = p.tap[t + 1 < t1 ? t + 1 : t];
            }
        }
        load_a(cc, a_nxt);
        load_b(more ? s + 1 : s, b_stage);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j + 1 < KC; j++) {          // all chunks but the last, fragments double-buffered
            if (j & 1) { read_bf(buf, j + 1, bf0); mfma_chunk(bf1, a_cur[j], 0, 4); }
            else       { read_bf(buf, j + 1, bf1); mfma_chunk(bf0, a_cur[j], 0, 4); }
        }
        if ((KC - 1) & 1) mfma_chunk(bf1, a_cur[KC - 1], 0, 2); else mfma_chunk(bf0, a_cur[KC - 1], 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        store_b(buf ^ 1, b_stage);
        __builtin_amdgcn_sched_barrier(0);
        if ((KC - 1) & 1) mfma_chunk(bf1, a_cur[KC - 1], 2, 4); else mfma_chunk(bf0, a_cur[KC - 1], 2, 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KC; j++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) a_cur[j][rt] = a_nxt[j][rt];
        __syncthreads();
    }

    // Epilogue: bias (+ LeakyReLU), float4 store of channels n .. n+3 of this lane's pixel.
    const int py = p.py[cls], px = p.px[cls];
    f32x4 bvs[NT];                                   // all bias loads before the first store (see pnn_gemm_sp.hip)
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int n = n0 + nt * 16 + (q << 2);
        bvs[nt] = *reinterpret_cast<const f32x4*>(p.bias + (n < p.Cout ? n : 0));
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        if (!mv[rt]) continue;
        const int oy = pi[rt] * p.os + py, ox = pj[rt] * p.os + px;
        const size_t obase = (((size_t)pb[rt] * p.OH + oy) * p.OW + ox) * p.Cout;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int n = n0 + nt * 16 + (q << 2);
            if (n < p.Cout) {
                const f32x4 bv = bvs[nt];
                f32x4 v = acc[rt][nt] + bv;
                if (p.act) {
                    v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                }
                if (p.Y) *reinterpret_cast<f32x4*>(p.Y + obase + n) = v;
                if (p.Yi) {
                    int4 iv = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean),
                                        hm_round(v[3], p.mean));
                    *reinterpret_cast<int4*>(p.Yi + obase + n) = iv;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Tap GEMM for SMALL M (last FC layer, batch-1 calls from inside HM): the four waves of a workgroup
// share one 16-row x 16*NT-column tile and split K between them (wave w takes chunks w, w+NW, ...).
// Nothing is shared before the end, so weights and activations stream straight into registers -- no
// LDS staging, no barrier in the loop -- and one LDS reduction combines the four partial tiles.
// Each wave's walk over its chunks is a chain of dependent L2 / MALL round trips with D chunks in flight, and the layer
// takes (chunks per wave / D) round trips: layers with long K (the 32x32 / 64x64 nets: up to 400 chunks) take the
// NW = 8, D = 8 instance -- twice the waves, twice the depth.  (Splitting K over WORKGROUPS with a last-arriver fix-up
// was tried: the device-scope release / acquire it needs writes back and invalidates the XCD's whole L2, ~25 us per
// launch -- single-block 64x64 call 223 -> 915 us.)
// ------------------------------------------------------------------------------------------------
template <int NT, int NW, int D>
__global__ __launch_bounds__(64 * NW) void tapgemm_splitk_kernel(const TapGemmParams p)
{
    touch_kernargs<sizeof(TapGemmParams)>();
    constexpr int BN = 16 * NT;
    __shared__ f32x4 red[NW][NT][64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q = lane >> 4;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int mg = blockIdx.x * 16 + l15;
    const bool mv = mg < p.M;
    const int SP = p.SH * p.SW;
    const int mc = mv ? mg : 0;
    const int pb = mc / SP;
    const int rr = mc - pb * SP;
    const int pi = rr / p.SW, pj = rr - pi * p.SW;

    const int cpt = p.Cin >> 4;
    const int t0 = p.tap_begin[cls], t1 = p.tap_begin[cls + 1];
    const int nchunks = (t1 - t0) * cpt;
    // weights through a buffer descriptor: uniform base + 32-bit lane offset (a 64-bit flat address per lane costs the wave
    // about twice the issue time, measured on the ring kernel's loaders)
    const __amdgpu_buffer_rsrc_t wrsrc_sk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const f32x4*>(p.Wp) + (size_t)p.chunk_begin[cls] * 4 * p.Npad), 0, 0xffffffffu, 0x00020000);
    const unsigned woff = (unsigned)((q * p.Npad + n0 + l15) << 4);
    const unsigned bstride = (unsigned)(4 * p.Npad) << 4;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    unsigned aoff = 0x80000000u;
    auto tap_setup = [&](int t) {
        const int tp = p.tap[t];
        const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
        const int iy = pi * p.a + dy, ix = pj * p.a + dx;
        const bool ok = mv && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const unsigned off = ((unsigned)((pb * p.IH + iy) * p.IW + ix) * (unsigned)p.Cin + (q << 2)) << 2;
        aoff = ok ? off : 0x80000000u;
    };
    auto load_chunk = [&](int chunk, int cc, f32x4& a, f32x4 (&b)[NT]) {
        a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff, cc << 6, 0));
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
            b[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc_sk, woff + (unsigned)chunk * bstride + (unsigned)(nt << 8), 0, 0));
    };

    if (wave < nchunks) {                           // wave-uniform
        // This wave's chunks are wave, wave+4, ...  A ring of D register sets keeps D chunks in flight: each layer
        // of a small pass is a dependent chain of L2/MALL round trips, so depth -- not bandwidth -- sets its time.
        f32x4 a_r[D], b_r[D][NT];
        int c_ld = wave;                            // load cursor (stays on this wave's last chunk once it gets there)
        int t = t0 + c_ld / cpt, cc = c_ld - (c_ld / cpt) * cpt;
        tap_setup(t);
        auto advance_ld = [&]() {
            if (c_ld + NW < nchunks) {
                c_ld += NW;
                cc += NW;
                if (cc >= cpt) {                    // the next chunk of this wave lies in a later tap
                    while (cc >= cpt) { cc -= cpt; ++t; }
                    tap_setup(t);
                }
            }
        };
#pragma unroll
        for (int u = 0; u < D; u++) {
            load_chunk(c_ld, cc, a_r[u], b_r[u]);
            advance_ld();
        }
        for (int c = wave; c < nchunks; c += NW * D) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                if (c + NW * u < nchunks) {         // wave-uniform
#pragma unroll
                    for (int e = 0; e < 4; e++)
#pragma unroll
                        for (int nt = 0; nt < NT; nt++)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(b_r[u][nt][e], a_r[u][e], acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    load_chunk(c_ld, cc, a_r[u], b_r[u]);
                    advance_ld();
                }
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) red[wave][nt][lane] = acc[nt];
    __syncthreads();

    if (!mv) return;
    const int oy = pi * p.os + p.py[cls], ox = pj * p.os + p.px[cls];
    const size_t obase = (((size_t)pb * p.OH + oy) * p.OW + ox) * p.Cout;
    for (int nt = wave; nt < NT; nt += NW) {
        const int n = n0 + nt * 16 + (q << 2);
        if (n < p.Cout) {
            f32x4 v = (red[0][nt][lane] + red[1][nt][lane]) + (red[2][nt][lane] + red[3][nt][lane]);
            if (NW == 8) v += (red[4][nt][lane] + red[5][nt][lane]) + (red[6][nt][lane] + red[7][nt][lane]);
            v += *reinterpret_cast<const f32x4*>(p.bias + n);
            if (p.act) {
                v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
            }
            if (p.Y) *reinterpret_cast<f32x4*>(p.Y + obase + n) = v;
            if (p.Yi) {
                int4 iv = make_int4(hm_round(v[0], p.mean), hm_round(v[1], p.mean), hm_round(v[2], p.mean),
                                    hm_round(v[3], p.mean));
                *reinterpret_cast<int4*>(p.Yi + obase + n) = iv;
            }
        }
    }
}

// {rt, nt, kc}: rt = 0 marks the split-K kernel (one 16-row tile shared by the four waves).
#define PNN_TG_CFGS(X) \
    X(2, 8, 1) X(2, 5, 1) X(2, 4, 1) X(2, 2, 1) X(2, 1, 1) X(1, 8, 1) X(1, 5, 1) X(1, 4, 1) X(1, 2, 1) X(1, 1, 1) \
    X(2, 8, 2) X(2, 5, 2) X(2, 4, 2) X(2, 2, 2) X(1, 8, 2) X(1, 5, 2) X(1, 4, 2) X(1, 2, 2)                   \
    X(2, 5, 4) X(2, 4, 4) X(1, 5, 4) X(1, 4, 4)
#define PNN_SK_CFGS(X) X(5) X(4) X(2) X(1)

static const TileCfg kCfgs[] = {
#define X(rt, nt, kc) {rt, nt, kc, 16},
    PNN_TG_CFGS(X)
#undef X
#define X(nt) {0, nt, 1, 16},
    PNN_SK_CFGS(X)
#undef X
};

static int n16() { return (int)(sizeof(kCfgs) / sizeof(kCfgs[0])); }
int tapgemm_num_cfgs() { return n16() + tapgemm32_num_cfgs(); }
TileCfg tapgemm_cfg(int idx) { return idx < n16() ? kCfgs[idx] : tapgemm32_cfg(idx - n16()); }

template <int RT, int NT, int KC>
static hipError_t launch_tg(const TapGemmParams& p, hipStream_t s)
{
    dim3 grid((p.M + 64 * RT - 1) / (64 * RT), (p.Cout + 16 * NT - 1) / (16 * NT), p.ncls);
    static const int lds_pad = getenv("PNN_LDS_PAD") ? atoi(getenv("PNN_LDS_PAD")) : 0;   // experiment: cap workgroups per CU
    pnn_launch(tapgemm_kernel<RT, NT, KC>, grid, dim3(256), lds_pad, s, p);
    return hipGetLastError();
}

template <int NT>
static hipError_t launch_sk(const TapGemmParams& p, hipStream_t s)
{
    dim3 grid((p.M + 15) / 16, (p.Cout + 16 * NT - 1) / (16 * NT), p.ncls);
    int chunks = 0;                                   // longest class
    for (int k = 0; k < p.ncls; k++) chunks = std::max(chunks, (p.tap_begin[k + 1] - p.tap_begin[k]) * (p.Cin >> 4));
    static const int force_nw = getenv("PNN_SK_WAVES") ? atoi(getenv("PNN_SK_WAVES")) : 0;
    if (force_nw ? force_nw == 8 : chunks >= 64) pnn_launch(tapgemm_splitk_kernel<NT, 8, (NT <= 2 ? 8 : 4)>, grid, dim3(512), 0, s, p);
    else pnn_launch(tapgemm_splitk_kernel<NT, 4, 4>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_tapgemm(const TapGemmParams& p, int cfg_idx, hipStream_t s)
{
    if (p.M <= 0) return hipSuccess;
    if (cfg_idx >= n16()) return launch_tapgemm32(p, cfg_idx - n16(), s);
    int i = 0;
#define X(rt, nt, kc) if (cfg_idx == i++) return launch_tg<rt, nt, kc>(p, s);
    PNN_TG_CFGS(X)
#undef X
#define X(nt) if (cfg_idx == i++) return launch_sk<nt>(p, s);
    PNN_SK_CFGS(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace pnn
