// Split-precision tap GEMM with the activation IMAGES resident in LDS (convolutions on small feature maps).
//
// tapgemm_sp_kernel re-reads every input pixel once per tap through the CU's vector-memory path (9x for a 3x3
// layer), and that path -- not the matrix cores -- bounds it.  The feature maps of the PNNs are tiny (4x4 ... 32x96
// pixels), so this kernel stages G whole images [pixels][Cin] (split f16 layout, 4 B per element) in LDS ONCE and
// builds the MFMA operand of every tap from LDS; only the weights still stream global -> registers -> LDS.
// Same contract and parameter block as the other tap-GEMM kernels (TapGemmParams; forward convs of any stride,
// transposed convs incl. the four stride-2 output-parity classes), same per-output summation order as
// tapgemm_sp_kernel, hence bit-identical results.
//
//   workgroup = 256 threads = 4 waves as WM x WN; wave tile (32*RT) x (32*NT); workgroup rows = G images x SH*SW
//   output pixels (rows past G*SH*SW idle), all of them taken from the G staged input images.
//   LDS: weights [3][KC][4 planes][BN] slots (round 4: an LDS-DMA ring, see the K loop) | images [(G*IH*IW + 1)][Cin/4 + 1] slots
//   (16-B slots; the +1 slot of pitch keeps the 64-lane fragment reads conflict-free; the extra pixel is all zeros = SAME
//   padding / idle rows).
// Measured (tools/convimg_prof.hip, 3x3 conv 64 -> 64 on 1024 images of 8x24, tile 192 x 64): per workgroup ~9.5k cycles
// to stage its 52 KB image, ~30k in the tap loop (1650 per 32-deep stage for 576 MFMA cycles, two workgroups per CU), ~10k in
// the epilogue.  Staging by LDS-DMA with an XOR-swizzled layout was tried: same 9.3k cycles -- the 50 MB of activations of
// all resident workgroups arrive as one burst at ~5.8 TB/s, it is memory bandwidth, not the copy loop.
// Round 4 (same tool, same layer): weights by LDS-DMA into three stage buffers, the stage barrier between its two chunks, the next
// stage's first fragments and the refill issued between MFMA groups, and a copy-out without divisions: 57.9 -> 48.3 us, tap loop
// 1620 -> 1346 cycles per stage (two co-resident workgroups: 1152 MFMA cycles), epilogue 10.5k -> 7.6k cycles.
// Round 3: the ring kernel's issue order inside a chunk (one MFMA, then two of the next chunk's fragment reads, pinned with
// sched_group_barrier) was tried here too: conv 16x16 pass 0.3830 -> 0.3849 ms same-box, i.e. nothing -- with two workgroups
// per CU the other workgroup's wave fills the SIMD while this one issues its read burst.
#include "pnn_kernels.h"
#include <type_traits>
#include "pnn_device_common.h"

namespace pnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kCiRing = 3;       // weight stage buffers (round 4: LDS-DMA ring, see the K loop)

// 16 bytes per lane, global -> LDS without passing registers (lane-linear destination).  In a __device__ helper on purpose: called
// straight from a lambda inside the __global__ template, the builtin made hipcc emit the host object WITHOUT its device code
// (no error, no .hip_fatbin section: the library then loads and every launch of this file's kernels fails).
__device__ __forceinline__ void ci_dma16(const __amdgpu_buffer_rsrc_t& r, unsigned voff, unsigned soff, f32x4* l)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

template <int RT, int NT, int KC, int WM>
__global__ __launch_bounds__(256) void convimg_sp_kernel(const TapGemmParams p, const int G)
{
    touch_kernargs<sizeof(TapGemmParams) + 4>();   // see pnn_device_common.h
#ifdef PNN_CI_DIAG              // coarse phase stamps of wave 0 (tools/convimg_prof.hip), written to p.Xlo
    const unsigned long long dq0 = __builtin_amdgcn_s_memtime();
#endif
    constexpr int WN = 4 / WM;
    constexpr int BN = 32 * NT * WN;
    constexpr int E = 4 * BN;
    constexpr int NLD = (E + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];
    f32x4* Bs = lds;                                  // [kCiRing][KC][E]
    f32x4* Ai = lds + kCiRing * KC * E;               // [G*NPIN][PITCH] | zero region [ZP]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int cls = blockIdx.z;
    const int n0 = blockIdx.y * BN;
    const long img0 = (long)blockIdx.x * G;           // first image of this workgroup
    const int SP = p.SH * p.SW;
    const int NPIN = p.IH * p.IW;
    const int PITCH = (p.Cin >> 2) + 1;
    // the zero region behind the images: Cin/4 + 16 slots.  A lane whose tap leaves the image reads zeros from the slot of this
    // region that falls on the banks its in-image slot would have had (slot index mod 16: a 64-lane read covers 16 lanes x
    // 16 bytes = all 64 banks per cycle, and consecutive pixels at pitch Cin/4 + 1 are consecutive mod 16) -- with ONE zero pixel
    // for everyone it collided with whichever neighbour shares that class: 23 % of the LDS cycles of the 16x16 net's image
    // launches were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r03_conv16_pmc_summary.txt)
    const int ZP = (p.Cin >> 2) + 16;
    const int nimg = (int)((long)p.M / SP - img0 < G ? (long)p.M / SP - img0 : G);   // images that exist

    // ---- stage the images and the zero pixel -------------------------------------------------------------------------
    if (p.X0 || p.plane0) {
        // Fused first convolution of the branch: the maps are COMPUTED here from the raw f32 context (a few KB per image)
        // instead of being written by conv_cin1_kernel and read back (50 MB each way at batch 1024 for the 16x16 net).
        // Same arithmetic as conv_cin1_kernel's split path (FirstConv: MFMA chain over the taps, scale, bias, LeakyReLU, split),
        // so the two paths agree bit for bit.  Scratch for the raw tiles = the weight staging area, free until the tap loop starts.
        float* raw = reinterpret_cast<float*>(Bs);
        const int K0 = p.k0, S0 = p.s0, IH0 = p.IH * S0, IW0 = p.IW * S0;
        const int PH = (p.IH - 1) * S0 + K0, PW = (p.IW - 1) * S0 + K0;
        bool in_bad = false;                           // a raw context element outside the f16 range (leaves_f16, pnn_device_common.h)
        if (p.plane0) {
            // the context gather fused in as well: straight from the picture plane through the TB descriptors (same values as
            // gather_f32x4_kernel writes: (float) pel - mean, unavailable units 0), no gather launch, no 5 KB per block round trip
            const TbDev* __restrict__ tbs = reinterpret_cast<const TbDev*>(p.tbs0) + img0;
            const int w0 = p.w0;
            TbDev d = tbs[0];                          // a thread's elements walk the images in order: one descriptor fetch per image,
            int dli = 0;                               // not per element
            for (int idx = tid; idx < nimg * PH * PW; idx += 256) {
                const int li = idx / (PH * PW), r0 = idx - li * PH * PW;
                const int r = r0 / PW, c = r0 - r * PW;
                const int iy = r - p.pad0, ix = c - p.pad0;
                float v = 0.f;
                if ((unsigned)iy < (unsigned)IH0 && (unsigned)ix < (unsigned)IW0) {
                    if (li != dli) { d = tbs[li]; dli = li; }
                    bool ok;
                    long off;
                    if (p.branch0 == 0) {
                        ok = ix < w0 || ((d.above_mask >> ((ix - w0) / p.unit0)) & 1u);
                        off = d.origin + (long)(iy - w0) * d.stride + (ix - w0);
                    } else {
                        ok = iy < d.left_units * p.unit0;
                        off = d.origin + (long)iy * d.stride + (ix - w0);
                    }
                    if (ok) v = (p.pel0 == 4 ? (float)reinterpret_cast<const int32_t*>(p.plane0)[off] : (float)reinterpret_cast<const uint8_t*>(p.plane0)[off]) - p.mean;
                }
                in_bad |= leaves_f16(v);
                raw[idx] = v;
            }
        } else
        for (int idx = tid; idx < nimg * PH * PW; idx += 256) {
            const int li = idx / (PH * PW), r0 = idx - li * PH * PW;
            const int r = r0 / PW, c = r0 - r * PW;
            const int iy = r - p.pad0, ix = c - p.pad0;
            const float v = ((unsigned)iy < (unsigned)IH0 && (unsigned)ix < (unsigned)IW0) ? p.X0[((size_t)(img0 + li) * IH0 + iy) * IW0 + ix] : 0.f;
            in_bad |= leaves_f16(v);
            raw[idx] = v;
        }
        if (in_bad && p.range_flag) *p.range_flag = 1;
        __syncthreads();
        // The contraction over the taps on the matrix cores (FirstConv, pnn_device_common.h): a wave takes 32 staged pixels at a
        // time, builds its operand from the raw tile, runs one MFMA chain per 32-channel column tile and writes scale / bias /
        // LeakyReLU / split straight into the image slots.  (Until round 3: 25 dependent v_fma_f32 per output on the VALU, 12 k
        // SIMD-cycles per image of the 16x16 net -- as much as this layer's own matrix work.)
        float amax0 = 0.f;                             // range guard (pnn_device_common.h)
        auto conv0 = [&](auto k_tag) {
            constexpr int K = decltype(k_tag)::value;
            const f32x4* wsp = reinterpret_cast<const f32x4*>(p.W0sp);
            const int nct = p.Cin >> 5;
            typename FirstConv<K>::W w0[2];             // first convolutions have 32 (stride 1) or 64 (stride 2) output channels
            f32x4 bv0[2][4];
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                if (ct < nct) w0[ct] = FirstConv<K>::weights(wsp, p.Npad0, ct * 32 + l31, h);
#pragma unroll
                for (int g = 0; g < 4; g++) bv0[ct][g] = *reinterpret_cast<const f32x4*>(p.B0 + (ct < nct ? ct * 32 + 8 * g + 4 * h : 0));
            }
            const int total = nimg * NPIN;
            FirstConv<K> fc;
            fc.setup(PW, h);
            for (int rt0 = wave; rt0 * 32 < total; rt0 += 4) {
                const int pix = rt0 * 32 + l31;
                const bool valid = pix < total;
                const int li = valid ? pix / NPIN : 0, q = valid ? pix - li * NPIN : 0;
                const int oy = q / p.IW, ox = q - oy * p.IW;
                fc.load(raw + li * PH * PW + (oy * S0) * PW + ox * S0);   // (an idle row reads image 0, pixel 0: its outputs are not stored)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) {
                    if (ct >= nct) break;
                    const f32x16 a0 = fc.tile(w0[ct]);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        f32x4 v = (f32x4){a0[4 * g], a0[4 * g + 1], a0[4 * g + 2], a0[4 * g + 3]} * p.scale0 + bv0[ct][g];
                        v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                        f16x4 hi, lo;
                        split4(v, hi, lo);
                        if (valid) {
                            amax0 = amax4(amax0, v);
                            const int n = ct * 32 + 8 * g + 4 * h;
                            _Float16* dst = reinterpret_cast<_Float16*>(Ai + pix * PITCH) + (n >> 4) * 32 + (n & 15);
                            *reinterpret_cast<f16x4*>(dst) = hi;
                            *reinterpret_cast<f16x4*>(dst + 16) = lo;
                        }
                    }
                }
            }
        };
        if (K0 == 5) conv0(std::integral_constant<int, 5>{}); else conv0(std::integral_constant<int, 3>{});
        report_range(p.range_flag, amax0);
        for (int idx = tid; idx < ZP; idx += 256) Ai[G * NPIN * PITCH + idx] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();                              // the scratch becomes the weight staging area again
    } else {
        const f32x4* src = reinterpret_cast<const f32x4*>(p.X) + (size_t)img0 * NPIN * (p.Cin >> 2);
        const int c4n = p.Cin >> 2;
        const int total = nimg * NPIN * c4n;
        constexpr int U = 8;                          // loads in flight per thread (a serial copy pays the full
        for (int base = 0; base < total; base += 256 * U) {   // memory latency once per 16 bytes)
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int idx = base + tid + 256 * u;
                v[u] = src[idx < total ? idx : total - 1];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int idx = base + tid + 256 * u;
                const int pix = idx / c4n, c4 = idx - pix * c4n;
                if (idx < total) Ai[pix * PITCH + c4] = v[u];
            }
        }
        for (int idx = tid; idx < ZP; idx += 256) Ai[G * NPIN * PITCH + idx] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // ---- this lane's output pixels ---------------------------------------------------------------------------------
    int pg[RT], pi[RT], pj[RT];
    bool mv[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) {
        const int ml = wm * (32 * RT) + rt * 32 + l31;
        const int g = ml / SP;
        mv[rt] = g < nimg;
        const int r = ml - g * SP;
        pg[rt] = g;
        pi[rt] = r / p.SW;
        pj[rt] = r - pi[rt] * p.SW;
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

    const f32x4* arow[RT];                            // this lane's source pixel for the current tap: its hi piece of chunk 0, k-half h
    auto tap_setup = [&](int tp) {
        const int dy = tp >> 16, dx = (int)(short)(tp & 0xffff);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int iy = pi[rt] * p.a + dy, ix = pj[rt] * p.a + dx;
            const bool ok = mv[rt] && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
            const int nat = (pg[rt] * NPIN + iy * p.IW + ix) * PITCH;          // where the pixel is -- or would be
            // (a pointer with the lane's k-half folded in, set once per tap: a fragment read is then base + a wave-uniform chunk
            // offset -- the index form cost three VALU instructions per read, 18 per stage of 18 MFMAs)
            arow[rt] = Ai + (ok ? nat : G * NPIN * PITCH + ((nat - G * NPIN * PITCH) & 15)) + h;
        }
    };
    // Weights: global -> LDS by LDS-DMA (round 4; until then global -> registers -> ds_write, one barrier per stage with the stage's
    // first fragment reads exposed behind it: 1650 cycles per stage for the 2 x 576 MFMA cycles of two co-resident workgroups).
    // Three stage buffers: the one barrier of a stage sits between its two chunks, stage s+2 is fetched behind it into the buffer
    // stage s-1 left, and the next stage's first fragments are read under the second chunk's MFMAs -- tapgemm_f32_kernel's scheme.
    static_assert(KC == 2, "the loop below is written for two chunks per stage");
    constexpr int NDMA = KC * E / 64, NPW = NDMA / 4;   // LDS-DMA wave-instructions per stage, per wave
    static_assert((KC * E) % 256 == 0, "a stage is the same whole number of wave instructions for every wave");
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, 0xffffffffu, 0x00020000);
    unsigned bsrc[NPW];
#pragma unroll
    for (int ii = 0; ii < NPW; ii++) {
        const int e = 64 * (wave + 4 * ii) + lane;
        const int j = e / E, ee = e - j * E;
        const int qq = ee / BN, nn = ee - qq * BN;
        bsrc[ii] = (unsigned)(((j * 4 + qq) * p.Npad + n0 + nn) << 4);
    }
    const unsigned wstage = (unsigned)(KC * 4 * p.Npad) << 4;   // bytes per stage
    auto dma_stage = [&](int stage, int buf) {
#pragma unroll
        for (int ii = 0; ii < NPW; ii++)
            ci_dma16(wrsrc, bsrc[ii], (unsigned)stage * wstage, Bs + buf * (KC * E) + 64 * (wave + 4 * ii));
    };
    auto read_frags = [&](int buf, int j, int cj, f32x4 (&wf)[NT][2], f32x4 (&af)[RT][2]) {   // [..][0] = hi, [..][1] = lo
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            wf[nt][0] = Bs[(buf * KC + j) * E + (0 + h) * BN + wn * (32 * NT) + nt * 32 + l31];
            wf[nt][1] = Bs[(buf * KC + j) * E + (2 + h) * BN + wn * (32 * NT) + nt * 32 + l31];
        }
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            af[rt][0] = arow[rt][cj * 4 + 0];
            af[rt][1] = arow[rt][cj * 4 + 2];
        }
    };
    // one chunk's MFMAs in three groups -- 0: w_hi * a_hi of every tile; 1 / 2: (w_hi * a_lo, w_lo * a_hi) of the first / second half of the
    // tiles -- so that the K loop can place memory instructions between them; per accumulator the order is always hi*hi, hi*lo, lo*hi
    auto mfma_group = [&](const f32x4 (&wf)[NT][2], const f32x4 (&a)[RT][2], int grp) {
        constexpr int T = NT * RT, H = (T + 1) / 2;
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
                const int i = nt * RT + rt;
                if (grp == 1 && i >= H) continue;
                if (grp == 2 && i < H) continue;
                const f16x8 whi = __builtin_bit_cast(f16x8, wf[nt][0]), wlo = __builtin_bit_cast(f16x8, wf[nt][1]);
                const f16x8 ahi = __builtin_bit_cast(f16x8, a[rt][0]), alo = __builtin_bit_cast(f16x8, a[rt][1]);
                if (grp == 0) {
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ahi, acc[rt][nt], 0, 0, 0);
                } else {
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, alo, acc[rt][nt], 0, 0, 0);
                    acc[rt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ahi, acc[rt][nt], 0, 0, 0);
                }
            }
    };
    auto mfma_chunk = [&](const f32x4 (&wf)[NT][2], const f32x4 (&a)[RT][2]) {
        mfma_group(wf, a, 0);
        mfma_group(wf, a, 1);
        mfma_group(wf, a, 2);
    };

    // ---- K loop: (tap, chunk pair) stages; only the weights are fetched per stage ---------------------------
    int t = t0, cc = 0;
    tap_setup(p.tap[t0]);
    int tp_next = p.tap[t0 + 1 < t1 ? t0 + 1 : t0];
    dma_stage(0, 0);
    dma_stage(nstages > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // images + the first two weight stages visible
#ifdef PNN_CI_DIAG
    const unsigned long long dq1 = __builtin_amdgcn_s_memtime();
#endif
    f32x4 wf0[NT][2], wf1[NT][2], af0[RT][2], af1[RT][2];
    read_frags(0, 0, 0, wf0, af0);
    int buf = 0;
    for (int s = 0; s < nstages; s++) {
        const int buf1 = buf == 2 ? 0 : buf + 1, buf2 = buf == 0 ? 2 : buf - 1;    // (s+1) % 3, (s+2) % 3
        mfma_group(wf0, af0, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(buf, 1, cc + 1 < cpt ? cc + 1 : cpt - 1, wf1, af1);           // second chunk's fragments under the first chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(wf0, af0, 1);
        mfma_group(wf0, af0, 2);
        __builtin_amdgcn_sched_barrier(0);
        // every wave's LDS-DMA of stage s+1 has landed (issued a stage ago), and nobody reads buffer (s-1) % 3 any more
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cc += KC;
        if (cc >= cpt && t + 1 < t1) {                // wave-uniform: the next stage starts the next tap
            cc = 0;
            ++t;
            tap_setup(tp_next);
            tp_next = p.tap[t + 1 < t1 ? t + 1 : t];
        }
        // the second chunk's MFMAs with the stage's memory work BETWEEN them: the next stage's first fragments, then the LDS-DMA of
        // stage s+2 (a wave is held while it issues a 1-KiB vector-memory instruction; with one workgroup per CU nobody else feeds the SIMD)
        mfma_group(wf1, af1, 0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(buf1, 0, cc < cpt ? cc : cpt - 1, wf0, af0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(wf1, af1, 1);
        __builtin_amdgcn_sched_barrier(0);
        dma_stage(s + 2 < nstages ? s + 2 : nstages - 1, buf2);                     // (past the end: the last stage again, harmless)
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(wf1, af1, 2);
        buf = buf1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // the epilogue reuses the LDS: every wave done reading, no DMA in flight

#ifdef PNN_CI_DIAG
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long dq2 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    // ---- epilogue (as tapgemm_sp_kernel) -------------------------------------------------------------------------
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
    // Three copies of the unrolled group loop, chosen once (split-f16 only / f32 only / any mix): the loop runs once per
    // launch from a cold instruction cache, so its time follows its code footprint (ring kernel: 7.9k -> 4.1k cycles).
    const bool act = p.act != 0;
    float amax = 0.f;                                // range guard of the split outputs (pnn_device_common.h)
    auto groups = [&](auto kind_tag) {
        constexpr int kKind = decltype(kind_tag)::value;             // 0: split only, 1: f32 only, 2: general
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            if (!mv[rt]) continue;
            const int oy = pi[rt] * p.os + py, ox = pj[rt] * p.os + px;
            const size_t obase = ((((size_t)img0 + pg[rt]) * p.OH + oy) * p.OW + ox) * p.Cout;
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
                            store_split4(p.Yhi, obase, n, v, amax);
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
    // Split-only output (every layer but a net's last): through LDS, like the ring kernel.  A lane of the accumulator layout
    // owns 8-byte fragments of 32 different output rows; stored from there a 48 KB tile took ~10k cycles.  The images are
    // dead after the tap loop, so the tile is laid out in their place as [rows][BN/16 chunks][hi 16 | lo 16] (what a row
    // segment looks like in memory) and leaves in 16-byte pieces, consecutive lanes = consecutive pieces of a pixel.
    constexpr int BM = 32 * RT * WM, OPP = BN / 4 + 1;   // +1 piece of pitch: spreads the rows over the banks
#ifdef PNN_CI_DIRECT_EPILOGUE   // A/B build: the stores from the accumulator layout
    const bool tile_fits = false;
#else
    const bool tile_fits = (size_t)BM * OPP <= (size_t)kCiRing * KC * E + (size_t)G * NPIN * PITCH + ZP;
#endif
    if constexpr (WN == 1 && NT == 2) {
    if (p.k1) {
        // ---- the net's LAST layer on the tile in registers (launch_convimg_sp checked: Cout == 64 == BN, one class, stride-1 grid)
        // The one-output-channel transposed convolution that ends every reference net (TConv1Params) reads exactly what this
        // layer writes, 256 bytes per pixel there and back.  Here a wave holds all 64 channels of its 32 pixels in the
        // accumulator layout -- lane (pixel l31, half h) owns channels 8q + 4h + i -- which IS the operand order of
        // tconv_cout1_mfma_kernel's phase 1 (its K permutation was chosen for the global loads of the same 16-byte pieces), so
        // T[pixel][tap] = sum_c x[pixel][c] w[tap][c] is the same v_mfma_f32_32x32x2_f32 chain on the same values, and phase 2
        // (col2im, bias, HM epilogue) the same sums in the same order: bit-identical to the two launches.  T goes where the
        // images were (dead after the tap loop).
        constexpr int TP = 33;                        // kTcTP of pnn_small.hip
        float* Tl = reinterpret_cast<float*>(lds);
        const int KK = p.k1 * p.k1;
        f32x4 wv[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            wv[q] = l31 < KK ? *reinterpret_cast<const f32x4*>(p.W1 + l31 * 64 + 8 * q + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            f32x16 t = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int nt = q >> 2, g = q & 3;
                f32x4 v = (f32x4){acc[rt][nt][4 * g], acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]} * p.out_scale + bvs[nt][g];
                if (act) {
                    v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) t = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i], wv[q][i], t, 0, 0, 0);
            }
            const int row0 = wm * (32 * RT) + rt * 32;
#pragma unroll
            for (int r = 0; r < 16; r++) Tl[(row0 + 8 * (r >> 2) + 4 * h + (r & 3)) * TP + l31] = t[r];
        }
        __syncthreads();
        const int K1 = p.k1, S1 = p.s1;
        const int OH1 = p.OH * S1, OW1 = p.OW * S1;     // this layer's output map = the last layer's input map
        for (int idx = tid; idx < nimg * OH1 * OW1; idx += 256) {
            const int gi = idx / (OH1 * OW1), o = idx - gi * (OH1 * OW1);
            const int oy = o / OW1, ox = o - oy * OW1;
            float v = 0.f;
            const int ky0 = (oy + p.pad1) % S1, kx0 = (ox + p.pad1) % S1;
            for (int a = 0; a < (K1 + S1 - 1) / S1; a++) {
                const int ky = ky0 + a * S1;
                const int iy = (oy + p.pad1 - ky + S1 * K1) / S1 - K1;
                if (ky >= K1 || iy < 0 || iy > p.OH - 1) continue;
                for (int cc = 0; cc < (K1 + S1 - 1) / S1; cc++) {
                    const int kx = kx0 + cc * S1;
                    const int ix = (ox + p.pad1 - kx + S1 * K1) / S1 - K1;
                    if (kx >= K1 || (unsigned)ix >= (unsigned)p.OW) continue;
                    v += Tl[(gi * SP + iy * p.OW + ix) * TP + ky * K1 + kx];
                }
            }
            v += p.bias1;
            const size_t oo = ((size_t)(img0 + gi) * OH1 + oy) * OW1 + ox;
            if (p.Y1) p.Y1[oo] = v;
            if (p.Yi1) p.Yi1[oo] = hm_round(v, p.mean);
        }
        return;
    }
    }
    if (p.Yhi && !p.Y && !p.Yi && tile_fits) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            const int lrow = wm * (32 * RT) + rt * 32 + l31;
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int nl = wn * (32 * NT) + nt * 32 + 8 * g + 4 * h;
                    f32x4 v = (f32x4){acc[rt][nt][4 * g], acc[rt][nt][4 * g + 1], acc[rt][nt][4 * g + 2], acc[rt][nt][4 * g + 3]} * p.out_scale + bvs[nt][g];
                    if (act) {
                        v[0] = leaky(v[0]); v[1] = leaky(v[1]); v[2] = leaky(v[2]); v[3] = leaky(v[3]);
                    }
                    if (mv[rt] && n0 + nl < p.Cout) amax = amax4(amax, v);      // padding rows / columns hold no activation
                    f16x4 hi, lo;
                    split4(v, hi, lo);                 // same values and rounding as store_split4
                    _Float16* dst = reinterpret_cast<_Float16*>(lds + lrow * OPP) + (nl >> 4) * 32 + (nl & 15);
                    *reinterpret_cast<f16x4*>(dst) = hi;
                    *reinterpret_cast<f16x4*>(dst + 16) = lo;
                }
        }
        __syncthreads();
        f32x4* __restrict__ yo = reinterpret_cast<f32x4*>(p.Yhi);
        const int cq = p.Cout >> 2;
        // A forward convolution's output grid IS its row grid (one class, os = 1): output pixel = first image's first pixel + row, no
        // divisions -- the general mapping below cost ~40 VALU instructions per 16-byte piece (two run-time divisions), a third of
        // this kernel's epilogue instructions on the layers with K = 576 (profiles/r03_conv16_pmc_summary.txt: 9.6 VALU per MFMA)
        const bool linear = p.ncls == 1 && p.os == 1 && p.OH == p.SH && p.OW == p.SW;      // launch-uniform
        if (linear) {
            f32x4* __restrict__ ybase = yo + (size_t)img0 * SP * cq + (n0 >> 2);
            const int rows_here = nimg * SP;
            const bool cols_whole = n0 + BN <= p.Cout;
            for (int i0 = tid; i0 < BM * (BN / 4); i0 += 2 * 256) {
                f32x4 v[2];
                f32x4* dst[2];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const int i = i0 + u * 256;
                    const int ic = i < BM * (BN / 4) ? i : tid;
                    const int row = ic / (BN / 4), q = ic - row * (BN / 4);
                    v[u] = lds[row * OPP + q];
                    dst[u] = (i < BM * (BN / 4) && row < rows_here && (cols_whole || (n0 >> 2) + q < cq)) ? ybase + ((size_t)row * cq + q) : nullptr;
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
                    if (dst[u]) store16_through(dst[u], v[u]);
            }
        } else
        for (int i0 = tid; i0 < BM * (BN / 4); i0 += 2 * 256) {   // two pieces per thread in flight (the store is an asm statement)
            f32x4 v[2];
            f32x4* dst[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * 256;
                const int ic = i < BM * (BN / 4) ? i : tid;
                const int row = ic / (BN / 4), q = ic - row * (BN / 4);
                const int g = row / SP, nq = (n0 >> 2) + q;
                const int r = row - g * SP;
                const int ri = r / p.SW, rj = r - ri * p.SW;
                const size_t opix = (((size_t)img0 + g) * p.OH + ri * p.os + py) * p.OW + rj * p.os + px;
                v[u] = lds[row * OPP + q];
                dst[u] = (i < BM * (BN / 4) && g < nimg && nq < cq) ? yo + (opix * cq + nq) : nullptr;
            }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (dst[u]) store16_through(dst[u], v[u]);
        }
    } else
    if (p.Yhi && !p.Y && !p.Yi) groups(std::integral_constant<int, 0>{});
    else if (p.Y && !p.Yhi && !p.Yi) groups(std::integral_constant<int, 1>{});
    else groups(std::integral_constant<int, 2>{});
    report_range(p.range_flag, amax);
#ifdef PNN_CI_DIAG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.Xlo && tid == 0) {
        unsigned long long* d = (unsigned long long*)p.Xlo + 4 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        const unsigned long long dq3 = __builtin_amdgcn_s_memtime();
        d[0] = dq1 - dq0; d[1] = dq2 - dq1; d[2] = dq3 - dq2; d[3] = nstages;
    }
#endif
}

// X(rt, nt, kc, wm): workgroup rows 32*rt*wm, columns 32*nt*(4/wm)
#define PNN_CI_CFGS(X) \
    X(3, 2, 2, 4) X(1, 2, 2, 4) X(2, 2, 2, 4) X(1, 4, 2, 4) X(2, 4, 2, 4) X(3, 2, 2, 2) X(2, 2, 2, 2) X(2, 1, 2, 2) X(3, 1, 2, 2) \
    X(1, 1, 2, 2) X(1, 2, 2, 2) X(4, 2, 2, 4)

static const TileCfg kCfgsCi[] = {
#define X(rt, nt, kc, wm) {rt, nt, kc, 316, wm},
    PNN_CI_CFGS(X)
#undef X
};

int convimg_sp_num_cfgs() { return (int)(sizeof(kCfgsCi) / sizeof(kCfgsCi[0])); }
TileCfg convimg_sp_cfg(int idx) { return kCfgsCi[idx]; }

// LDS bytes of configuration `t` staging G images of this layer (0 if it cannot run: too big / bad shape).
bool convimg_sp_can_fuse_first(const TapGemmParams& p, const TileCfg& t, int G, int s0, int k0)
{
    const size_t bn = 32 * (size_t)t.nt * (4 / t.wm);
    const size_t scratch_floats = kCiRing * (size_t)t.kc * 4 * bn * 4;         // the weight staging area
    const size_t ph = (size_t)(p.IH - 1) * s0 + k0, pw = (size_t)(p.IW - 1) * s0 + k0;
    return (k0 == 3 || k0 == 5) && (p.Cin == 32 || p.Cin == 64) && (size_t)G * ph * pw <= scratch_floats;
}

// Can this configuration apply the net's last layer (TConv1Params: Cin -> 1 transposed convolution) to its output tile?  Every
// wave must hold all channels of its pixels (one column of waves, two 32-channel tiles: Cout == 64), the layer must be a
// stride-1 single-class one whose rows are whole output maps, the last layer one that tconv_cout1_mfma_kernel takes (the same
// arithmetic), and the T tile ([rows][33] floats) must fit the LDS the kernel has anyway.
bool convimg_sp_can_fuse_last(const TapGemmParams& p, const TileCfg& t, int G, const TConv1Params& last)
{
    if (t.wm != 4 || t.nt != 2 || p.Cout != 64 || p.ncls != 1 || p.os != 1 || p.SH != p.OH || p.SW != p.OW) return false;
    if (last.Cin != 64 || !((last.s == 2 && last.k == 5) || (last.s == 1 && last.k == 3)) || last.pad > last.k - 1) return false;
    if (last.IH != p.OH || last.IW != p.OW) return false;
    return (size_t)32 * t.rt * 4 * 33 * 4 <= convimg_sp_lds_bytes(p, t, G);
}

size_t convimg_sp_lds_bytes(const TapGemmParams& p, const TileCfg& t, int G)
{
    const size_t bn = 32 * (size_t)t.nt * (4 / t.wm);
    const size_t slots = kCiRing * (size_t)t.kc * 4 * bn + (size_t)G * p.IH * p.IW * ((size_t)(p.Cin >> 2) + 1) + (size_t)(p.Cin >> 2) + 16;   // weights | images | zero region
    return slots * 16;
}

template <int RT, int NT, int KC, int WM>
static hipError_t launch_ci(const TapGemmParams& p, int G, hipStream_t s)
{
    constexpr int BN = 32 * NT * (4 / WM);
    const TileCfg t{RT, NT, KC, 316, WM};
    const size_t lds = convimg_sp_lds_bytes(p, t, G);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&convimg_sp_kernel<RT, NT, KC, WM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const long nimg = p.M / (p.SH * p.SW);
    dim3 grid((unsigned)((nimg + G - 1) / G), (p.Cout + BN - 1) / BN, p.ncls);
    pnn_launch(convimg_sp_kernel<RT, NT, KC, WM>, grid, dim3(256), lds, s, p, G);
    return hipGetLastError();
}

hipError_t launch_convimg_sp(const TapGemmParams& p, int idx, int G, hipStream_t s)
{
    if (p.M <= 0) return hipSuccess;
    int i = 0;
#define X(rt, nt, kc, wm) if (idx == i++) return launch_ci<rt, nt, kc, wm>(p, G, s);
    PNN_CI_CFGS(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace pnn
