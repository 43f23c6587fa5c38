// Rule-based choice of the kernel family and tile configuration of one tap-GEMM launch.  Every configuration of the three
// split-precision families gives bit-identical results (tests/test_gpu_parity.py::test_split_gemm_kernel_families_bit_identical),
// so the rules only decide speed.  They were distilled from on-device sweeps over all legal configurations (the autotuner's
// logs, PNN_DEBUG_TUNE=1; tools/tile_sweep.py reproduces the sweep and profiles/r03_tile_sweep.txt keeps its output).
#include "pnn_ctx.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace pnn {

// tapgemm_f32_kernel (32x32x2 MFMA, few waves with big tiles): the tile that finishes first by a list-scheduling estimate.  Every
// tile runs its loop at 0.95-0.98 of the matrix rate (PNN_F32_DIAG stamps), so a launch costs its padded
// MFMA work spread over 256 CUs, plus a tail of about half a workgroup unless the workgroups fit the chip exactly (FC 8x8 at batch
// 4096 on the 128 x 160 tile: 256 workgroups, one per CU), plus the start-up / epilogue of a workgroup -- which co-resident
// workgroups (R per CU, by LDS and registers) hide behind each other's MFMAs.  Two row tiles per wave measured 20-25 % slower than
// the same area as one (twice the activation loads per MFMA, one resident workgroup).  Big launches are autotuned on top of this
// (run_gemm); all tiles give the same bits.  -1: no legal tile.
int choose_cfg_f32(const pnn_ctx* c, long M, int cout, int ncls, int cin, double k_total, bool fused, double* cost_out)
{
    const int cpt = cin / 16;
    const bool one_tap = (k_total == (double)cin);
    if (c->opt_f32_cfg >= 0 && c->opt_f32_cfg < tapgemm_f32_num_cfgs()) {
        const TileCfg t = tapgemm_f32_cfg((int)c->opt_f32_cfg);
        if ((one_tap || cpt % t.kc == 0) && (!fused || tapgemm_f32_can_fuse((int)c->opt_f32_cfg))) { if (cost_out) *cost_out = 0.0; return (int)c->opt_f32_cfg; }
    }
    int best = -1;
    double best_cost = 1e300;
    const DeviceInfo& di = device_info();
    const double ncu = (double)di.cus, lds_cu = (double)di.lds;
    for (int i = 0; i < tapgemm_f32_num_cfgs(); i++) {
        const TileCfg t = tapgemm_f32_cfg(i);
        if (fused && !tapgemm_f32_can_fuse(i)) continue;
        if (!one_tap && cpt % t.kc) continue;
        const long bm = 128L * t.rt, bn = 32L * t.nt;
        const double wgs = (double)((M + bm - 1) / bm) * (double)((cout + bn - 1) / bn) * ncls;
        const double regs = tapgemm_f32_regs(i);
        const int res = (int)std::max(1.0, std::min(std::min(4.0, std::floor(512.0 / regs)), std::floor(lds_cu / (double)tapgemm_f32_lds_bytes(t, fused))));
        const double chunks = std::ceil(k_total / 16.0 / ncls / t.kc) * t.kc;
        const double mfma = chunks * 8.0 * t.rt * t.nt * 64.0;
        const double fixed = 5000.0 + 2500.0 * t.rt * t.nt;
        const bool exact = std::fmod(wgs, ncu) == 0.0 && wgs / ncu <= res;
        // start-up and epilogue hide behind OTHER workgroups' loops only when a CU's slots turn over (>= 2 rounds); workgroups that are all
        // resident from the start go through them together, and the more of them share a CU the longer those phases last (FC 8x8 first
        // layer, K = 320, 32-column tiles: 4.75 workgroups per CU, 14 k + 10 k cycles of prologue + epilogue around 10 k of MFMAs)
        const double per_cu = wgs / ncu, rounds = per_cu / res;
        const double fixed_eff = rounds >= 2.0 ? 0.4 * fixed : fixed * std::max(1.0, std::min((double)res, per_cu));
        double cost = wgs * mfma / ncu + (exact ? 0.0 : 0.6 * mfma) + fixed_eff;
        if (wgs < ncu) cost = mfma + fixed;                         // under-filled chip: the launch lasts one workgroup
        if (t.rt == 2) cost *= 1.2;
        cost *= 1.0 + 0.01 / (t.rt * t.nt) + (t.kc == 2 ? 0.005 : 0.0);   // ties: the bigger wave tile, the longer stage
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    if (cost_out) *cost_out = best_cost;              // cycles, for the caller's choice between the two forms of a K-segmented layer
    return best;
}

// convimg_sp_kernel: how many images one workgroup of tile `t` stages for this layer (0 = tile cannot run the layer).
int convimg_images(const TapGemmParams& p, const TileCfg& t, bool one_tap)
{
    if (one_tap || (p.Cin / 16) % t.kc) return 0;
    const int rows = 32 * t.rt * t.wm, sp = p.SH * p.SW;
    int g = rows / sp;
    while (g > 0 && convimg_sp_lds_bytes(p, t, g) > (size_t)156 * 1024) --g;
    return g;
}

// Rule-based choice among the convimg tiles: fewest idle rows and columns, then the larger wave tile.  -1 = none fits.
// (A cost model with workgroup counts and residency was tried against the autotuner's per-configuration timings of
// the conv-16/32 layers and picked WORSE tiles overall -- 0.58 vs 0.54 ms per conv-16 pass; the three kernel families
// are within 10-15 % of each other on most layers, so big passes are simply autotuned, see run_gemm_sp.)
// Mid-size passes (tens of blocks: the batching service, small pictures) do not fill the chip with the big tiles: below two
// workgroups per CU the cost grows with the idle share, which takes the choice down to the 64-row tile where the tuner
// ends up too (16x16 net, 100 blocks: 471 -> ~300 us per pass).
int choose_cfg_convimg(const TapGemmParams& p, bool one_tap)
{
    int best = -1;
    double best_cost = 1e300, best_fit = 1e300;
    const long nimg = p.M / (p.SH * p.SW);
    for (int i = 0; i < convimg_sp_num_cfgs(); i++) {
        const TileCfg t = convimg_sp_cfg(i);
        const int g = convimg_images(p, t, one_tap);
        if (g <= 0) continue;
        const long rows = 32L * t.rt * t.wm, bn = 32L * t.nt * (4 / t.wm);
        const long tn = (p.Cout + bn - 1) / bn;
        const double pad = (double)rows * (tn * bn) / ((double)g * p.SH * p.SW * p.Cout);
        // tile height: 128 rows is the sweet spot (tuner logs of the 8x8 / 16x16 nets, K = 576, 64 output channels): taller
        // tiles stage more images per workgroup -- more LDS, fewer co-resident workgroups, a longer serial staging phase
        // (384 rows: 81 us where 128 rows take 49) --, the 64-row tile pays more start-up per MFMA
        double height = rows <= 64 ? 1.15 : rows <= 128 ? 1.0 : rows <= 192 ? 1.05 : rows <= 256 ? 1.25 : rows <= 384 ? 1.6 : 2.0;
        // 32-channel layers (4x4 conv net: K = 288, maps of 16-48 pixels): little work per image, so the tuner settles on EIGHT
        // images per workgroup whatever the map size (384 / 256 / 128 rows for 48 / 32 / 16 pixels) -- the weight stream and the
        // start-up are then shared by enough matrix work
        if (p.Cout <= 32) height = 1.0 + 0.3 * std::fabs(std::log2((double)rows / (8.0 * p.SH * p.SW)));
        const double wgs = (double)((nimg + g - 1) / g) * tn * p.ncls;
        const double fill = wgs >= 512.0 ? 1.0 : 512.0 / wgs;
        const double cost = pad * height * fill;
        // what decides whether the family is used at all: idle rows / columns and an unsuitable height -- for the 32-channel
        // layers the padding alone (their preferred height depends on the batch through `fill`)
        if (cost < best_cost) { best_cost = cost; best_fit = p.Cout <= 32 ? pad : pad * height; best = i; }
    }
    // 32-channel layers (the 4x4 conv net) fill only half of the narrowest tile's 64 columns and still run 30 % faster here than
    // on the register-staged kernel (tuner, batch 4096: 29.6 / 21.7 / 11.0 us for its three layers against a 120 vs 91 us pass)
    return best_fit <= (p.Cout <= 32 ? 3.0 : 1.6) ? best : -1;
}

// Rule-based choice among the ring-kernel tiles for big one-tap (fully-connected) layers, -1 = leave it to the other
// kernels.  Calibrated with tools/ring_prof.hip: a workgroup costs ~(prologue + epilogue) + stages x 1.45 x its MFMA
// cycles (loader and MFMA waves overlap imperfectly), workgroups run one (LDS > 80 KB) or two per CU.
bool pnn_ring_few_images(const TapGemmParams& p, long M, double k_total)
{
    return (p.ncls == 1 || p.ncls == 4) && p.Cout == 64 && k_total / p.ncls >= 512.0 && M / ((long)p.SH * p.SW) < 128 && (M + 63) / 64 * p.ncls >= 128;
}

int choose_cfg_ring(const TapGemmParams& p, long M, bool one_tap, double k_total, bool fused)
{
    if (!fused && !one_tap) {
        // Convolution layers.  With the buffer-descriptor loaders the ring kernel is the fastest of the three families on
        // every layer with >= 128 output channels (tuner logs of the 16x16 / 32x32 nets at 300 ... 1024 blocks: 15-30 % ahead
        // of the register-staged and the LDS-resident-image kernels); the 64-output-channel 3x3 layers stay with the
        // LDS-resident-image kernel.  Tile: 128 columns, the tallest of 192 / 128 / 64 rows that still gives >= 192
        // workgroups (one round of the chip), else 64 rows.  Under-filled long-K layers of any width take the 64-row tile.
        const bool wide = p.Cout % 128 == 0 && k_total / p.ncls >= 1152.0;
        const double col_tiles = (double)((p.Cout + 127) / 128) * p.ncls;
        const bool underfilled = k_total >= 1600.0 && (double)((M + 63) / 64) * col_tiles <= 256.0;
        // stride-2 transposed convolutions to 64 channels (four output-parity classes of 4-9 taps each: short K per class, the
        // classes as blockIdx.z): from 8192 rows on the tuner takes the 128 x 64 ring tile over the LDS-resident-image kernel
        // at every batch looked at (16x16 / 32x32 nets, 512 ... 1024 / 128 ... 256 blocks: 26-28 us against 39 at M = 16384)
        // (round 3: with the 3-deep ring, 28.0 against 32.1 us at M = 16384 -- K per class is 4-9 taps x 8 chunks, the shorter
        // ring starts sooner; tools/tile_sweep.py, profiles/r03_tile_sweep.txt)
        if (p.ncls == 4 && p.Cout == 64 && M >= 8192) {
            for (int d = 3; d <= 4; d++)
                for (int i = 0; i < tapgemm_ring_num_cfgs(); i++) {
                    const TileCfg t = tapgemm_ring_cfg(i);
                    if (t.rt == 1 && t.nt == 2 && t.kc == 2 && t.wm == 4 && t.d == d) return i;
                }
        }
        // 64-channel 3x3 layers of a FEW images (under 128: the image kernel gets one workgroup per image or less and leaves most
        // of the chip idle) but enough rows for >= 128 tiles of 64 rows: the 64 x 128 ring tile, half its columns empty, is what
        // the tuner takes (16x16 net, 64 blocks: 17.9 us against 28)
        const bool few_images = pnn_ring_few_images(p, M, k_total);
        if (!wide && !underfilled && !few_images) return -1;
        int rt = 1, wm = 2, d = 4;                    // 64 x 128
        if (wide) {
            if ((double)((M + 191) / 192) * col_tiles >= 192.0) { rt = 3; d = 4; }        // 192 x 128, four stages = all 160 KB of LDS (round 3: 0.2 % / 0.8 % on the 16x16 / 32x32 passes over the 3-deep ring -- its ~1880-cycle stage for 1152 cycles of MFMA work is LDS bandwidth, 80 KB of fragment reads + 40 KB of DMA writes per stage, not the missing stage of slack)
            else if ((double)((M + 127) / 128) * col_tiles >= 192.0) rt = 2;              // 128 x 128
        }
        for (int i = 0; i < tapgemm_ring_num_cfgs(); i++) {
            const TileCfg t = tapgemm_ring_cfg(i);
            if (t.rt == rt && t.nt == 2 && t.kc == 2 && t.wm == wm && t.d == d) return i;
        }
        return -1;
    }
    if (!fused && ((double)M * p.Cout < 5.0e5 || p.Cin < 64)) return -1;   // FC layers from ~512 rows on (tuner logs: ring 64x128 / 128x64 tiles win there too)
    int best = -1;
    double best_cost = 1e300;
    for (int i = 0; i < tapgemm_ring_num_cfgs(); i++) {
        const TileCfg t = tapgemm_ring_cfg(i);
        if (fused && !tapgemm_ring_can_fuse(i)) continue;
        if (!one_tap && (p.Cin / 16) % t.kc) continue;
        const long bm = 32L * t.rt * t.wm, bn = 32L * t.nt * (4 / t.wm);
        const long nwg = ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn) * p.ncls;
        const int resident = tapgemm_ring_lds_bytes(t) > (size_t)80 * 1024 ? 1 : 2;
        const double stages = std::ceil((k_total / 16.0 / p.ncls) / (double)t.kc);
        const double mfma = 96.0 * t.rt * t.nt * t.kc;
        // stage time / MFMA time and the fixed part, from tools/ring_prof.hip with the buffer-descriptor loaders (FC 1200x1200,
        // M = 4096): D >= 4 rings 1.14-1.19, three-deep rings 1.5-1.9, 16-deep stages 1.3; start-up + epilogue 7-13k cycles
        const double slow = t.kc == 1 ? 1.35 : (resident == 2 ? 1.25 : (t.d >= 4 ? 1.17 : 1.6));
        const double wg = 5000.0 + 1400.0 * t.rt * t.nt + stages * mfma * slow;
        const double rounds = std::ceil(nwg / (256.0 * resident));
        double cost = rounds * wg * (resident == 2 ? 1.6 : 1.0);         // two co-resident workgroups share the CU's MFMA pipes
        cost *= 1.0 + 0.05 * (1.0 - nwg / (256.0 * resident * rounds));   // ties: the tile that leaves fewer CUs idle (M = 1024: 64 x 128 over 128 x 64, as the tuner)
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return best;
}

// Split-precision launch (3 x f16 MFMA): activations as two f16 planes, outputs f32 and/or two f16 planes.
int choose_cfg_sp(const pnn_ctx* c, long M, int cout, int ncls, int cin, double k_total)
{
    const int cpt = cin / 16;
    const bool one_tap = (k_total == (double)cin);
    if (c->opt_sp_cfg >= 0 && c->opt_sp_cfg < tapgemm_sp_num_cfgs()) {
        const TileCfg t = tapgemm_sp_cfg((int)c->opt_sp_cfg);
        if (one_tap || cpt % t.kc == 0) return (int)c->opt_sp_cfg;
    }
    int best = -1;
    double best_cost = 1e300;
    for (int i = 0; i < tapgemm_sp_num_cfgs(); i++) {
        const TileCfg t = tapgemm_sp_cfg(i);
        if (!one_tap && cpt % t.kc) continue;
        const long bm = 32L * t.rt * t.wm, bn = 32L * t.nt * (4 / t.wm);
        const long tm = (M + bm - 1) / bm, tn = (cout + bn - 1) / bn;
        const double wgs = (double)tm * tn * ncls;
        // calibrated on device sweeps (tools/sp_check.py, tools/sp_prof.py): time ~ padded work x (1 + 2/NT) (operand
        // traffic per MFMA), mild tail quantisation, RT = 2 and KC = 4 lose a resident workgroup, KC = 1 adds barriers
        const double per_cu = wgs / 256.0;
        // under-filled chip: idle CUs below one workgroup per CU, no co-resident workgroup below ~1.5
        const double fill = per_cu < 1.0 ? 1.2 * std::pow(1.0 / per_cu, 0.7) : 1.0 + 0.4 * std::max(0.0, 1.5 - per_cu);
        const double pad = (double)(tm * bm) * (tn * bn) / ((double)M * cout);
        const double reuse = 1.0 + 2.0 / (t.nt * (4 / t.wm));
        const double shape = (t.rt * t.wm >= 8 ? 1.15 : 1.0) * (t.kc == 4 ? 1.4 : (t.kc == 1 ? 1.08 : 1.0));
        const double cost = fill * pad * reuse * shape;
        if (cost < best_cost) { best_cost = cost; best = i; }
    }
    return best < 0 ? 0 : best;
}

}  // namespace pnn
