// Device bodies of two non-GEMM layers -- the channel-wise FC merger and the Cout = 1 last transposed convolution on the fp32 matrix
// cores -- shared by their own kernels (pnn_small.hip) and by the small exact-f32 GEMM kernels that run them as TAILS of the layer
// in front (pnn_gemm_f32_small.hip, round 6): one source, one arithmetic.
#pragma once
#include "pnn_kernels.h"
#include "pnn_device_common.h"

namespace pnn {

// cache-policy bits of a raw buffer load on gfx950: sc0 | sc1 = system scope (what another XCD wrote through)
constexpr int kAuxThrough = 1 | 16;
// A [M][C] tensor that workgroups of ONE launch write through and other workgroups of the same launch read (the tails of
// pnn_gemm_f32_small.hip) is TILE-MAJOR: the 16 x 16 tile (rows 16 R .., channels 16 G ..) is 1 KiB of its own, [G][R][piece g = channels
// 4 g .. 4 g + 3][row][4 floats] -- the producing wave's lanes side by side.  In [row][channel] order two 16-channel tiles share every
// 128-byte line; the L2 of the XCD that wrote one half keeps the line WITH the other half as it was, and a reader on that XCD is served
// from it, system-scope load or not (tools/tails_stress.py: 1 call in 10 000 wrong beside four other contexts).  A line that ONE
// workgroup writes whole is right wherever it is cached.  Byte offset of the piece (row m, channels 16 G + 4 g ..), gx = M / 16:
__device__ __forceinline__ unsigned tile_major_piece(unsigned m, unsigned G, unsigned g, unsigned gx)
{
    return (((G * gx + (m >> 4)) * 64u + g * 16u + (m & 15u)) * 4u) << 2;
}

// One 16-block x 16-channel tile of the merger (see merger_mfma_kernel, pnn_small.hip): rows = blocks first .. first + 15 (clamped to
// B - 1), channels c0 .. c0 + 15; `red` = 48 KiB of LDS.  ONE: every row is block `first` -- the tail of a small pass, where the tile's
// 80 inputs were written a moment ago by workgroups of the SAME launch on other XCDs: they are read past the caches, and only the
// lanes that hold row 0 store.  All 256 threads; no early exit.
template <bool SPLIT, bool ONE>
__device__ __forceinline__ void merger_mfma_tile(const MergerParams& p, f32x4 (*red)[16][64], const int c0, const long first)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    long brow = ONE ? first : first + li;
    if (brow >= p.B) brow = p.B - 1;
    f32x4 xv[5][4], wv[5][4];
    __amdgpu_buffer_rsrc_t arsrc, lrsrc;
    if (ONE) {
        arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
        lrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.L, 0, 0x7fffffff, 0x00020000);
    }
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const int p0 = 4 * (wave + 4 * t);            // first position of this step (wave-uniform): above part or left part
        const float* wr = p.Wp + ((size_t)(p0 + lk) * 16 + li) * p.C + c0;
        if (ONE) {                                    // the branch maps are tile-major here (tile_major_piece)
            const bool ab = p0 < p.na;
            const unsigned np = ab ? p.na : p.nl;
            const unsigned m = (unsigned)brow * np + (unsigned)(ab ? p0 : p0 - p.na) + lk;
#pragma unroll
            for (int q = 0; q < 4; q++)
                xv[t][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ab ? arsrc : lrsrc, tile_major_piece(m, c0 >> 4, q, ((unsigned)p.B * np) >> 4), 0, kAuxThrough));
        } else {
            const float* xr = p0 < p.na ? p.A + ((size_t)brow * p.na + p0 + lk) * p.C + c0
                                        : p.L + ((size_t)brow * p.nl + (p0 - p.na) + lk) * p.C + c0;
#pragma unroll
            for (int q = 0; q < 4; q++) xv[t][q] = *reinterpret_cast<const f32x4*>(xr + 4 * q);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) wv[t][q] = *reinterpret_cast<const f32x4*>(wr + 4 * q);
    }
    __builtin_amdgcn_sched_barrier(0);               // every load is in flight before the first MFMA waits
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 5; t++)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                acc[4 * q + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[t][q][i], wv[t][q][i], acc[4 * q + i], 0, 0, 0);
    if (wave) {
#pragma unroll
        for (int i = 0; i < 16; i++) red[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (acc[i] + red[0][i][lane]) + (red[1][i][lane] + red[2][i][lane]);
    // acc[i][r]: block first + 4 * lk + r (ONE: block `first` in every r), output j = li, channel c0 + i
    f32x4 bv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bv[q] = *reinterpret_cast<const f32x4*>(p.bias + (size_t)li * p.C + c0 + 4 * q);
    float amax = 0.f;                                // range guard of the split output (pnn_device_common.h)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (ONE && (r || lk)) break;
        const long b = ONE ? first : first + 4 * lk + r;
        const size_t pix = (size_t)b * 16 + li;
        if (b < p.B) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = leaky(acc[4 * q + i][r] + bv[q][i]);
                if (SPLIT) store_split4(p.Y, pix * p.C, c0 + 4 * q, v, amax);
                else store4_chain(p.Y + pix * p.C + c0, q, v, p.chain != 0);
            }
        }
    }
    if (SPLIT) report_range(p.range_flag, amax);
}

// One band of output rows of one block of the Cout = 1 last transposed convolution, Cin = 64, on the fp32 matrix cores (see
// tconv_cout1_mfma_kernel, pnn_small.hip): T = LDS, [pixels of the band, padded to 32][kTcTP] floats.  THROUGH: the input map was
// written through by other workgroups of the same launch -- read past the caches.  All 256 threads.
constexpr int kTcTP = 33;                            // LDS pitch of a T row (32 taps + 1)
template <int s, int K, bool THROUGH>
__device__ __forceinline__ void tconv_cout1_mfma_band(const TConv1Params& p, float* T, const long b, const int band)
{
    constexpr int KK = K * K;
    const int OH = p.IH * s, OW = p.IW * s;
    const int TOH = p.ni;                             // output rows per band (set by the launcher)
    const int oy0 = band * TOH;
    const int lo_y = oy0 + p.pad - (K - 1);
    int iy0 = lo_y >= 0 ? lo_y / s : -((-lo_y + s - 1) / s);
    int iy1 = (oy0 + TOH - 1 + p.pad) / s;
    if (iy0 < 0) iy0 = 0;
    if (iy1 > p.IH - 1) iy1 = p.IH - 1;
    const int npx = (iy1 - iy0 + 1) * p.IW;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = lane & 31, h = lane >> 5;
    const int tiles = (npx + 31) >> 5;
    if (wave < tiles) {
        f32x4 wv[8];                                  // B operand: w[tap = col][channels 8q + 4h .. +3]
#pragma unroll
        for (int q = 0; q < 8; q++)
            wv[q] = col < KK ? *reinterpret_cast<const f32x4*>(p.W + col * 64 + 8 * q + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* xb = p.X + ((b * p.IH + iy0) * (long)p.IW) * 64;
        __amdgpu_buffer_rsrc_t xrsrc;
        if (THROUGH) xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0x7fffffff, 0x00020000);   // tile-major (tile_major_piece)
        for (int t = wave; t < tiles; t += 4) {
            int px = t * 32 + col;
            if (px >= npx) px = npx - 1;              // padding rows of the last tile: recomputed, never read
            f32x4 xv[8];
            if (THROUGH) {
                const unsigned m = (unsigned)((b * p.IH + iy0) * (long)p.IW) + (unsigned)px, gx = ((unsigned)p.B * p.IH * p.IW) >> 4;
#pragma unroll
                for (int q = 0; q < 8; q++)       // channels 8 q + 4 h ..: 16-channel group q / 2, piece 2 (q & 1) + h
                    xv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tile_major_piece(m, q >> 1, 2 * (q & 1) + h, gx), 0, kAuxThrough));
            } else {
                const float* xr = xb + (long)px * 64 + 4 * h;
#pragma unroll
                for (int q = 0; q < 8; q++) xv[q] = *reinterpret_cast<const f32x4*>(xr + 8 * q);
            }
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; q++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[q][i], wv[q][i], acc, 0, 0, 0);
            // acc[r]: pixel row 8 * (r / 4) + 4 * h + r % 4 of the tile, tap = col
#pragma unroll
            for (int r = 0; r < 16; r++) T[(t * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * kTcTP + col] = acc[r];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TOH * OW; idx += 256) {
        const int oyl = idx / OW, ox = idx - oyl * OW;
        const int oy = oy0 + oyl;
        if (oy >= OH) break;
        float v = 0.f;
        // taps of this pixel's parity class: ky = ky0, ky0 + s, ... with (oy + pad - ky) % s == 0; the numerators are
        // kept non-negative (+ s * K) so that / and % are shifts
        const int ky0 = (oy + p.pad) % s, kx0 = (ox + p.pad) % s;
#pragma unroll
        for (int a = 0; a < (K + s - 1) / s; a++) {
            const int ky = ky0 + a * s;
            const int iy = (oy + p.pad - ky + s * K) / s - K;
            if (ky >= K || iy < iy0 || iy > iy1) continue;   // band rows are clipped to the image: outside = zero input
#pragma unroll
            for (int c = 0; c < (K + s - 1) / s; c++) {
                const int kx = kx0 + c * s;
                const int ix = (ox + p.pad - kx + s * K) / s - K;
                if (kx >= K || (unsigned)ix >= (unsigned)p.IW) continue;
                v += T[((iy - iy0) * p.IW + ix) * kTcTP + ky * K + kx];
            }
        }
        v += p.bias;
        const size_t o = ((size_t)b * OH + oy) * OW + ox;
        if (p.Y) p.Y[o] = v;
        if (p.Yi) p.Yi[o] = hm_round(v, p.mean);
    }
}

// The general form of the Cout = 1 last transposed convolution (tconv_cout1_kernel, pnn_small.hip): one 16 x 16 output tile `by` of
// images bx * p.ni ..; xt = LDS.  THROUGH as above.  All 256 threads.
template <bool THROUGH>
__device__ __forceinline__ void tconv_cout1_lds_tile(const TConv1Params& p, f32x4* xt, const int bx, const int by)
{
    __amdgpu_buffer_rsrc_t xrsrc;
    if (THROUGH) xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0x7fffffff, 0x00020000);
    const int s = p.s, K = p.k;
    const int OH = p.IH * s, OW = p.IW * s;
    const int TO = OH < 16 ? OH : 16;                 // output tile edge (square images: OH == OW)
    const int tiles_x = (OW + TO - 1) / TO;
    const int ty = by / tiles_x, tx = by - ty * tiles_x;
    const int oy0 = ty * TO, ox0 = tx * TO;
    // input rows/cols that can reach this output tile: iy in [floor((oy0 + pad - (K-1)) / s), (oy0 + TO-1 + pad) / s]
    const int lo_y = oy0 + p.pad - (K - 1), lo_x = ox0 + p.pad - (K - 1);
    const int iy0 = lo_y >= 0 ? lo_y / s : -((-lo_y + s - 1) / s);
    const int ix0 = lo_x >= 0 ? lo_x / s : -((-lo_x + s - 1) / s);
    const int TI = (oy0 + TO - 1 + p.pad) / s - iy0 + 1;
    const int NP = TI * TI, C4 = p.Cin >> 2;
    const int NI = p.ni;                              // images per workgroup (small outputs share a workgroup)
    const long bbase = (long)bx * NI;
    for (int idx = threadIdx.x; idx < NI * NP * C4; idx += 256) {
        const int li = idx / (NP * C4), r0 = idx - li * NP * C4;
        const int pix = r0 / C4, c4 = r0 - pix * C4;
        const int r = pix / TI, c = pix - r * TI;
        const int iy = iy0 + r, ix = ix0 + c;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bbase + li < p.B && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW) {
            if (THROUGH)                              // tile-major, written through by other workgroups of this launch (tile_major_piece)
                v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, tile_major_piece((unsigned)(((bbase + li) * p.IH + iy) * p.IW + ix), (unsigned)c4 >> 2, (unsigned)c4 & 3u,
                                                                                                            ((unsigned)p.B * p.IH * p.IW) >> 4), 0, kAuxThrough));
            else v = *reinterpret_cast<const f32x4*>(p.X + (((bbase + li) * p.IH + iy) * p.IW + ix) * p.Cin + 4 * c4);
        }
        xt[(li * C4 + c4) * NP + pix] = v;
    }
    // the k x k x Cin weights behind the tiles: read per (tap, channel quad) as one broadcast ds_read_b128 in the inner
    // loop (as wave-uniform scalar loads they cost a ~200-cycle round trip per iteration: 26 us for the 16x16 net at
    // batch 1024, most of it waiting)
    f32x4* wl = xt + NI * C4 * NP;
    for (int idx = threadIdx.x; idx < K * K * C4; idx += 256) wl[idx] = reinterpret_cast<const f32x4*>(p.W)[idx];
    __syncthreads();

    // Thread -> (image, output pixel).  Stride 2: wave = parity class (py, px), lanes = NI images x (TO/2)^2
    // pixels of that class, so the tap set is wave-uniform.  Stride 1: plain row-major pixels, every tap valid.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    int li, oy, ox, py, px;
    bool live;
    if (s == 2) {
        const int SG = TO >> 1, SGP = SG * SG;
        py = wave >> 1; px = wave & 1;
        li = lane / SGP;
        const int sub = lane - li * SGP;
        oy = oy0 + 2 * (sub / SG) + py; ox = ox0 + 2 * (sub % SG) + px;
        live = li < NI;
    } else {
        const int PT = TO * TO;
        py = 0; px = 0;
        li = threadIdx.x / PT;
        const int sub = threadIdx.x - li * PT;
        oy = oy0 + sub / TO; ox = ox0 + sub % TO;
        live = li < NI;
    }
    if (!live) li = 0;
    float acc = 0.f;
    for (int ky = 0; ky < K; ky++) {
        if ((py + p.pad - ky) % s) continue;          // wave-uniform: parity class (s = 2) or always taken (s = 1)
        const int ry = (oy + p.pad - ky) / s - iy0;   // exact division for this class; inside the staged tile
        for (int kx = 0; kx < K; kx++) {
            if ((px + p.pad - kx) % s) continue;
            const int rx = (ox + p.pad - kx) / s - ix0;
            const f32x4* xp = xt + (size_t)li * C4 * NP + ry * TI + rx;
            const f32x4* wp = wl + (ky * K + kx) * C4;
            for (int c4 = 0; c4 < C4; c4++) {
                const f32x4 xv = xp[c4 * NP], wv = wp[c4];
                acc += xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3];
            }
        }
    }
    if (live && bbase + li < p.B && oy < OH && ox < OW) {
        const float v = acc + p.bias;
        const size_t o = ((size_t)(bbase + li) * OH + oy) * OW + ox;
        if (p.Y) p.Y[o] = v;
        if (p.Yi) p.Yi[o] = hm_round(v, p.mean);
    }
}

}  // namespace pnn
