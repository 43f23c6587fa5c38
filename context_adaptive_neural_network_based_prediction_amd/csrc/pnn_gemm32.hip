// Tap GEMM on v_mfma_f32_32x32x2_f32 (the f32 MFMA shape that sustains the full 64 FLOP/clk/SIMD on
// gfx950; the 16x16x4 shape tops out near 85 % of it in a pure issue loop on this part).
//
// Same contract as tapgemm_kernel in pnn_kernels.hip (see TapGemmParams in pnn_kernels.h):
//   Y[pix(m)][n] = act( sum_{taps, ci} X[b, i*a+dy, j*a+dx, ci] * W[tap][ci][n] + bias[n] )
// i.e. the FC layers, convolutions and transposed convolutions of pnn/components.py:10-261.
//
//   workgroup = 256 threads = 4 waves; wave w owns rows [32*RT*w, 32*RT*(w+1)) of the BM = 128*RT row
//   tile and all BN = 32*NT columns; accumulators: RT*NT tiles of 32x32 (16 VGPRs each).
//   MFMA operand roles: "A" = weights (i = n), "B" = activations (j = m).  Lane l = (l&31, h = l>>5)
//   supplies, in step e (0..7) of a 16-deep chunk, k = 8h + e for both operands -- two float4 per lane
//   and chunk.  In the packed weight layout [chunk][q = 4][Npad][4] that is planes q = 2h and 2h+1.
//   D: lane holds column m = l&31; register 4g + r is row n = 8g + 4h + r  ->  four float4 stores.
#include "pnn_kernels.h"
#include "pnn_device_common.h"

namespace pnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int RT, int NT, int KC>
__global__ __launch_bounds__(256) void tapgemm32_kernel(const TapGemmParams p)
{
    touch_kernargs<sizeof(TapGemmParams)>();
    constexpr int BM = 128 * RT;
    constexpr int BN = 32 * NT;
    constexpr int E = 4 * BN;                       // float4 per staged weight chunk
    constexpr int NLD = (E + 255) / 256;
    __shared__ f32x4 Bs[2][KC][E];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const int m0 = blockIdx.x * BM + wave * (32 * RT);

    int pb[RT], pi[RT], pj[RT];
    bool mv[RT];
    const int SP = p.SH * p.SW;
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int mg = m0 + rt * 32 + l31;
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

    f32x16 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[rt][nt][i] = 0.f;

    // Activations: buffer-descriptor loads, out-of-image taps / rows past M read zeros (offset beyond
    // num_records), per-tap byte offsets, the chunk inside the tap in the scalar offset.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, p.x_bytes, 0x00020000);
    unsigned aoff[RT];
    auto tap_setup = [&](int tp) {                  // tp = (dy << 16) | (dx & 0xffff)
        const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int iy = pi[rt] * p.a + dy, ix = pj[rt] * p.a + dx;
            const bool ok = mv[rt] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            const unsigned off = ((unsigned)((pb[rt] * p.IH + iy) * p.IW + ix) * (unsigned)p.Cin + (h << 3)) << 2;
            aoff[rt] = ok ? off : 0x80000000u;
        }
    };
    auto load_a = [&](int cc, f32x4 (&dst)[KC][RT][2]) {
#pragma unroll
        for (int j = 0; j < KC; j++) {
            const int cj = cc + j < cpt ? cc + j : cpt - 1;   // padding chunk: zero weights, any finite data will do
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                dst[j][rt][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[rt], cj << 6, 0));
                dst[j][rt][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, aoff[rt] + 16u, cj << 6, 0));
            }
        }
    };
    unsigned bsrc[NLD];                              // byte offsets into the packed weights (buffer loads: a 32-bit offset per lane
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0xffffffffu, 0x00020000);   // issues faster than a 64-bit flat address, see pnn_gemm_ring.hip)
    int bdst[NLD];
#pragma unroll
    for (int r = 0; r < NLD; r++) {
        int e = tid + 256 * r;                      // past-the-end threads duplicate element E-1 (no exec-masked code)
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
    auto read_wf = [&](int buf, int j, f32x4 (&wf)[NT][2]) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            wf[nt][0] = Bs[buf][j][(2 * h) * BN + nt * 32 + l31];
            wf[nt][1] = Bs[buf][j][(2 * h + 1) * BN + nt * 32 + l31];
        }
    };
    auto mfma_chunk = [&](const f32x4 (&wf)[NT][2], const f32x4 (&a)[RT][2], int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; e++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int rt = 0; rt < RT; rt++)
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[nt][e >> 2][e & 3], a[rt][e >> 2][e & 3], acc[rt][nt], 0, 0, 0);
    };

    // Two-stage pipeline, one barrier per KC chunks; order pinned with sched_barrier:
    //   fragment reads of the first chunk -> global prefetch of stage s+1 -> MFMAs (all but half of the
    //   last chunk) -> staged weights to the other LDS buffer -> remaining MFMAs -> barrier.
    f32x4 a_cur[KC][RT][2], a_nxt[KC][RT][2], b_stage[KC][NLD];
    int t = t0, cc = 0;
    tap_setup(p.tap[t0]);
    int tp_next = p.tap[t0 + 1 < t1 ? t0 + 1 : t0];
    load_a(0, a_cur);
    load_b(0, b_stage);
    store_b(0, b_stage);
    __syncthreads();
    for (int s = 0; s < nstages; s++) {
        const int buf = s & 1;
        f32x4 wf0[NT][2], wf1[NT][2];
        read_wf(buf, 0, wf0);
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
        for (int j = 0; j + 1 < KC; j++) {
            if (j & 1) { read_wf(buf, j + 1, wf0); mfma_chunk(wf1, a_cur[j], 0, 8); }
            else       { read_wf(buf, j + 1, wf1); mfma_chunk(wf0, a_cur[j], 0, 8); }
        }
        if ((KC - 1) & 1) mfma_chunk(wf1, a_cur[KC - 1], 0, 4); else mfma_chunk(wf0, a_cur[KC - 1], 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        store_b(buf ^ 1, b_stage);
        __builtin_amdgcn_sched_barrier(0);
        if ((KC - 1) & 1) mfma_chunk(wf1, a_cur[KC - 1], 4, 8); else mfma_chunk(wf0, a_cur[KC - 1], 4, 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < KC; j++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) { a_cur[j][rt][0] = a_nxt[j][rt][0]; a_cur[j][rt][1] = a_nxt[j][rt][1]; }
        __syncthreads();
    }

    // Epilogue: bias (+ LeakyReLU); register group g of a tile holds channels 8g + 4h .. +3 of this lane's pixel.
    const int py = p.py[cls], px = p.px[cls];
    f32x4 bvs[NT][4];                                // all bias loads before the first store (see pnn_gemm_sp.hip)
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int n = n0 + nt * 32 + 8 * g + 4 * h;
            bvs[nt][g] = *reinterpret_cast<const f32x4*>(p.bias + (n < p.Cout ? n : 0));
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
                    const f32x4 bv = bvs[nt][g];
                    f32x4 v = (f32x4){acc[rt][nt][4 * g], acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]} + bv;
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

#define PNN_TG32_CFGS(X) \
    X(1, 1, 2) X(1, 1, 4) X(1, 2, 1) X(1, 2, 2) X(1, 2, 4) X(2, 1, 2) X(2, 1, 4) X(2, 2, 1) X(2, 2, 2) \
    X(1, 3, 2) X(1, 4, 1) X(1, 4, 2) X(1, 5, 1) X(1, 5, 2) X(2, 4, 1)

static const TileCfg kCfgs32[] = {
#define X(rt, nt, kc) {rt, nt, kc, 32},
    PNN_TG32_CFGS(X)
#undef X
};

int tapgemm32_num_cfgs() { return (int)(sizeof(kCfgs32) / sizeof(kCfgs32[0])); }
TileCfg tapgemm32_cfg(int idx) { return kCfgs32[idx]; }

template <int RT, int NT, int KC>
static hipError_t launch_tg32(const TapGemmParams& p, hipStream_t s)
{
    dim3 grid((p.M + 128 * RT - 1) / (128 * RT), (p.Cout + 32 * NT - 1) / (32 * NT), p.ncls);
    pnn_launch(tapgemm32_kernel<RT, NT, KC>, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_tapgemm32(const TapGemmParams& p, int idx, hipStream_t s)
{
    int i = 0;
#define X(rt, nt, kc) if (idx == i++) return launch_tg32<rt, nt, kc>(p, s);
    PNN_TG32_CFGS(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace pnn
