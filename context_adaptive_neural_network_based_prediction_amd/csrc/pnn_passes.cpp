// One pass of a PNN over a chunk of blocks: which kernel family runs each layer, in which order, on which buffers
// (DESIGN.md section 4).  Reference behaviour reproduced (not code): inference_fully_connected / inference_convolutional /
// branch / merger of pnn/components.py:10-261.
#include "pnn_ctx.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace pnn {

// Diagnostic library only (make diag) under PNN_B1_STAMPS: the next kernel's slot of per-workgroup 100 MHz stamps (host_predict prints
// them), or null.  64 bytes per workgroup: [3] loop start, [4] entry, [5] exit (behind the acknowledged stores), [1] loop ticks.
void* diag_stamp_slot(pnn_ctx* c, const char* name, long wgs, double k)
{
    if (!c->diag_stamps || c->diag_launch >= pnn_ctx::kDiagLaunches || wgs > pnn_ctx::kDiagWgs) return nullptr;
    c->diag_names.push_back(name);
    c->diag_wgs.push_back((int)wgs);
    c->diag_k.push_back(k);
    return (char*)c->diag_stamps + (size_t)c->diag_launch++ * pnn_ctx::kDiagWgs * 64;
}

namespace {

// Would run_gemm send this layer (nb blocks, no fused next layer) to the small exact-f32 kernels?  The ONE statement of that rule:
// run_gemm decides by it, and the passes by it which tensors travel in chain order (both ends on those kernels, pnn_gemm_f32_small.hip).
bool f32_small_applies(const pnn_ctx* c, const GemmLayer& L, long nb, bool has_next, bool has_yi)
{
    static const bool big_diag = getenv("PNN_F32_DIAG") != nullptr;
    if (has_next || !c->opt_f32_small || c->opt_f32_cfg >= 0 || big_diag) return false;
    TapGemmParams p = L.proto;
    const long M = nb * p.SH * p.SW;
    if (M > 0x7fffffffL) return false;
    p.M = (int)M;
    if (L.fc_seg_chunks > 0 && L.nseg > 1) {
        p.nseg = L.nseg; p.seg_chunks = (unsigned)L.fc_seg_chunks;
        return fcseg_f32_small_fits(p) && fcseg_f32_small_tiles(p) <= c->opt_f32_small_tiles;
    }
    p.nseg = (!has_yi && L.nseg > 1) ? L.nseg : 1;
    return tapgemm_f32_small_tiles(p) <= c->opt_f32_small_tiles;
}
// ... and may its OUTPUT be written in chain order?  (a K-segmented layer only when its planes are added up inside the launch: the
// seg_reduce kernel writes channel order)
bool f32_small_chain_out_ok(const pnn_ctx* c, const GemmLayer& L, long nb)
{
    if (!c->opt_chain_io || L.proto.Cout % 16) return false;
    if (L.nseg <= 1 || L.fc_seg_chunks > 0) return true;
    TapGemmParams p = L.proto;
    p.M = (int)(nb * p.SH * p.SW); p.nseg = L.nseg;
    return c->opt_seg_fold && c->d_seg_cnt && tapgemm_f32_small_tiles(p) / L.nseg <= pnn_ctx::kSegCntTiles && !c->opt_time_launches;
}

// Exact-f32 launch.  `next` (optional): the net's output layer (<= 64 outputs) applied to this layer's activated tile inside the
// launch (tapgemm_f32_kernel, 128 x 160 tile); `part` then receives the per-column-tile partial sums [tiles][M][64], *tiles_out
// their count, and the caller finishes with launch_fuse_reduce.  Y / Yi must be null in that case.
// chain_io: bit 0 = X is in chain order, bit 1 = write Y in chain order (TapGemmParams::chain_io) -- only for a layer that
// f32_small_applies() sends to the small kernels; anything else is the caller's mistake and refused.
int run_gemm(pnn_ctx* c, const GemmLayer& L, const float* X, float* Y, int32_t* Yi, long nblocks, hipStream_t s,
             const GemmLayer* next = nullptr, float* part = nullptr, int* tiles_out = nullptr, const float* host_rows = nullptr, int chain_io = 0,
             const SmallTail* tail = nullptr, bool* tail_ran = nullptr)
{
    TapGemmParams p = L.proto;
    p.X = X; p.Wp = L.d_w; p.bias = L.d_bias; p.Y = Y; p.Yi = Yi; p.mean = c->mean;
    const long M = nblocks * p.SH * p.SW;
    if (M > 0x7fffffffL) return fail(c, PNN_E_ARG, "batch too large for one pass");
    p.M = (int)M;
    const double xb = 4.0 * (double)nblocks * p.IH * p.IW * p.Cin;
    if (xb >= 2147483648.0) return fail(c, PNN_E_ARG, "activation tensor of %.0f bytes exceeds the 2 GiB descriptor bound", xb);
    p.x_bytes = (unsigned)xb;
    static const bool debug = getenv("PNN_DEBUG") != nullptr;
    static const bool profile = getenv("PNN_PROFILE") != nullptr;   // tuning aid: per-launch timing, synchronous
    const double flops = 2.0 * (double)M * L.k_total * p.Cout + (next ? 2.0 * (double)M * next->k_total * next->proto.Cout : 0.0);
    const bool one_tap = (L.k_total == (double)p.Cin);
    const int cpt = p.Cin / 16;
    constexpr bool f32k = true;                       // (until round 4 an option chose the round-1 kernels here)
    int cfg = -1;
    std::function<hipError_t(int)> launch;
    p.pm_groups = c->opt_ring_pm == 0 ? -1 : c->opt_ring_pm == 2 ? 1 : 0;   // position-major tiles: never / whenever possible / by the planner's model (launch_tapgemm_f32)
    // K segments (GemmLayer::nseg, the canonical order of the deep conv layers): the launch leaves nseg planes of partial sums, a
    // second launch adds them in order, + bias, activation
    // A segmented ONE-TAP layer (FC, GemmLayer::fc_seg_chunks) is folded inside the workgroups at every batch size -- no planes, so it may
    // carry the fused output layer or the HM epilogue: four chains side by side at small M, one after the other into a running total at batch
    const bool fcseg = f32k && L.fc_seg_chunks > 0 && L.nseg > 1;
    const int nseg = (f32k && !fcseg && !next && !Yi && L.nseg > 1) ? L.nseg : 1;
    const size_t out_floats = (size_t)nblocks * (size_t)L.out_per_block;
    if (f32k && !fcseg && L.nseg > 1 && nseg == 1) return fail(c, PNN_E_ARG, "a K-segmented layer cannot carry the HM epilogue or a fused output layer");
    if (fcseg) { p.nseg = L.nseg; p.seg_chunks = (unsigned)L.fc_seg_chunks; p.seg_seq = 1; }
    // Two forms with the same bits: PARALLEL segments (grid z = class x segment, planes of partial sums + seg_reduce_kernel: more,
    // shorter workgroups -- what a launch that does not fill the chip needs) and SEQUENTIAL ones (each workgroup runs its segments
    // one after the other and folds them into a running total: no planes, no second launch -- what a big launch wants).
    // Configuration codes of a segmented layer: [0, ntile) = parallel on tile code, [ntile, 2 ntile) = sequential on tile code - ntile.
    const int ntile = tapgemm_f32_num_cfgs();
    p.persist = (next || one_tap) ? 0 : (int)c->opt_f32_persist;   // persistent workgroups for the convolution layers, see launch_f32 (pnn_gemm_f32.hip)
    TapGemmParams pseq = p;                           // the sequential form: the real Y, bias and activation
    if (nseg > 1) {
        pseq.nseg = nseg; pseq.seg_seq = 1;
        DevBuf& sb = c->seg_part[(c->side_stream && s == c->side_stream) ? 1 : 0];
        int rrc;
        if ((rrc = dev_reserve(c, sb, (size_t)nseg * out_floats * 4))) return rrc;
        if (out_floats >= 0xffffffffull) return fail(c, PNN_E_ARG, "batch too large for one pass");
        p.Y = (float*)sb.p; p.bias = (const float*)c->d_zero; p.act = 0; p.nseg = nseg; p.seg_stride = (unsigned)out_floats; p.seg_seq = 0;
    }
    // Few output tiles (the in-loop single-block calls, the batching service's handfuls): the same fmaf chain per output on the 16x16x4
    // instruction, one wave per 16 x 16 tile over all CUs (pnn_gemm_f32_small.hip) -- bit-identical, 3.2 x shorter dependent chain
    const bool small_f32 = f32k && f32_small_applies(c, L, nblocks, next != nullptr, Yi != nullptr);
    if (chain_io && !small_f32) return fail(c, PNN_E_ARG, "internal: chain-order activations for a layer that does not run on the small kernels");
    if (fcseg && small_f32) {
        TapGemmParams ps = p;
        ps.Wp = L.d_w_ch;                             // the weights in the chain waves' lane order
        ps.chain_io = chain_io;
        if (debug) fprintf(stderr, "[pnn] gemm M=%ld K=%.0f N=%d nseg=%d -> f32 small kernel, %d K segments side by side (%ld tiles of 16 x 16)\n", M, L.k_total, p.Cout, L.nseg, L.nseg, fcseg_f32_small_tiles(p));
        if (void* slot = diag_stamp_slot(c, "fcseg_f32_small", fcseg_f32_small_tiles(p), L.k_total)) ps.Xlo = slot;   // (diagnostic library, PNN_B1_STAMPS)
        if (profile || c->opt_time_launches) {
            pnn_ctx::LaunchRec r;
            HIPCHK(c, hipEventCreate(&r.e0));
            HIPCHK(c, hipEventCreate(&r.e1));
            r.kind = 6; r.flops = flops;
            const LaunchEvents ev{r.e0, r.e1};
            g_launch_events = &ev;
            const hipError_t le = launch_fcseg_f32_small(ps, s);
            g_launch_events = nullptr;
            HIPCHK(c, le);
            if (profile) {
                HIPCHK(c, hipEventSynchronize(r.e1));
                float ms = 0.f;
                HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
                fprintf(stderr, "[pnn-prof] M=%ld K=%.0f N=%d nseg=%d f32-small-fcseg us=%.1f tflops=%.2f\n", M, L.k_total, p.Cout, L.nseg, ms * 1e3, r.flops / (ms * 1e-3) / 1e12);
                (void)hipEventDestroy(r.e0);
                (void)hipEventDestroy(r.e1);
            } else {
                c->launch_recs.push_back(r);
            }
        } else {
            HIPCHK(c, launch_fcseg_f32_small(ps, s));
        }
        c->stat_gemm_launches++; c->stat_launches++;
        c->stat_gemm_flops += flops;
        return PNN_OK;
    }
    if (!fcseg && small_f32) {
        TapGemmParams ps = p;
        ps.Wp = L.d_w_ch;                             // the same weights in the small kernel's lane order
        ps.chain_io = chain_io;
        if ((chain_io & 1) && host_rows) host_rows = nullptr;   // (the argument-block input is the caller's raw context: never in chain order)
        // K segments: added up by the launch itself (the tile's last workgroup to arrive) when the tiles have counters
        const long seg_tiles = nseg > 1 ? tapgemm_f32_small_tiles(p) / nseg : 0;
        const bool fold = nseg > 1 && c->opt_seg_fold && c->d_seg_cnt && seg_tiles <= pnn_ctx::kSegCntTiles && !c->opt_time_launches;
        if (fold) {
            if (c->seg_cnt_dirty) {                   // see pnn_ctx::seg_cnt_dirty
                HIPCHK(c, hipMemsetAsync(c->d_seg_cnt, 0, pnn_ctx::kCntWords * 4, s));
                if (c->side_stream && s != c->side_stream) HIPCHK(c, hipStreamSynchronize(s));   // (the other branch's launches use the second half on the side stream)
                c->seg_cnt_dirty = false;
            }
            ps.seg_cnt = c->d_seg_cnt + ((c->side_stream && s == c->side_stream) ? pnn_ctx::kSegCntTiles : 0);
            ps.seg_Y = Y; ps.bias = L.d_bias; ps.act = L.proto.act;
            DevBuf& sb = c->seg_part[(c->side_stream && s == c->side_stream) ? 1 : 0];   // tile-major planes: 1 KiB per (segment, tile)
            int rrc;
            if ((rrc = dev_reserve(c, sb, (size_t)nseg * seg_tiles * 1024))) return rrc;
            ps.Y = (float*)sb.p;
        }
        if (debug) fprintf(stderr, "[pnn] gemm M=%ld K=%.0f N=%d ncls=%d nseg=%d -> f32 small kernel (%ld tiles of 16 x 16)\n", M, L.k_total, p.Cout, p.ncls, nseg, tapgemm_f32_small_tiles(p));
        if (void* slot = diag_stamp_slot(c, "tapgemm_f32_small", tapgemm_f32_small_tiles(p), L.k_total)) ps.Xlo = slot;   // (diagnostic library, PNN_B1_STAMPS)
        if (profile || c->opt_time_launches) {
            pnn_ctx::LaunchRec r;
            HIPCHK(c, hipEventCreate(&r.e0));
            HIPCHK(c, hipEventCreate(&r.e1));
            r.kind = 6; r.flops = flops;
            const LaunchEvents ev{r.e0, r.e1};
            g_launch_events = &ev;
            const hipError_t le = launch_tapgemm_f32_small(ps, s, host_rows, (int)c->opt_f32_small_deep);
            g_launch_events = nullptr;
            HIPCHK(c, le);
            if (profile) {
                HIPCHK(c, hipEventSynchronize(r.e1));
                float ms = 0.f;
                HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
                fprintf(stderr, "[pnn-prof] M=%ld K=%.0f N=%d ncls=%d f32-small us=%.1f tflops=%.2f\n", M, L.k_total, p.Cout, p.ncls, ms * 1e3, r.flops / (ms * 1e-3) / 1e12);
                (void)hipEventDestroy(r.e0);
                (void)hipEventDestroy(r.e1);
            } else {
                c->launch_recs.push_back(r);
            }
        } else if (tail && tail_ran && nseg == 1 && !(chain_io & 2) && f32_small_cout1_tail_ok(ps, tail->t)) {
            // the net's last layer as this launch's tail (SmallTail kind 2): per block, by the last of the block's tiles to arrive
            if (debug) fprintf(stderr, "[pnn]   ... with the last transposed convolution as its tail\n");
            HIPCHK(c, launch_tapgemm_f32_small_tail(ps, *tail, s, (int)c->opt_f32_small_deep));
            *tail_ran = true;
        } else {
            HIPCHK(c, launch_tapgemm_f32_small(ps, s, host_rows, (int)c->opt_f32_small_deep));
        }
        if (nseg > 1 && !fold) {
            if (chain_io & 2) return fail(c, PNN_E_ARG, "internal: chain-order output of a layer whose K segments are reduced by a second launch");
            HIPCHK(c, launch_seg_reduce(p.Y, nseg, out_floats, p.Cout, L.d_bias, L.proto.act, Y, s));
            c->stat_launches++;
        }
        static const bool sdiag = getenv("PNN_F32S_DIAG") != nullptr;   // diagnostic library only (make diag): the MFMA wave's loop, cycles per chunk and clock
        if (sdiag) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (dev_reserve(c, c->stage_tbs, (size_t)4 << 20)) return PNN_E_NOMEM;
            HIPCHK(c, hipMemset(c->stage_tbs.p, 0, (size_t)4 << 20));
            TapGemmParams q = ps;
            q.Xlo = c->stage_tbs.p;
            HIPCHK(c, launch_tapgemm_f32_small(q, s, host_rows, (int)c->opt_f32_small_deep));
            HIPCHK(c, hipStreamSynchronize(s));
            const size_t nwg = std::min<size_t>((size_t)tapgemm_f32_small_tiles(p), ((size_t)4 << 20) / 64);
            std::vector<unsigned long long> hbuf(8 * nwg);
            HIPCHK(c, hipMemcpy(hbuf.data(), c->stage_tbs.p, hbuf.size() * 8, hipMemcpyDeviceToHost));
            double cyc = 0, ticks = 0, chunks = 0;
            unsigned long long r0 = ~0ull, r1 = 0;
            for (size_t i = 0; i < nwg; i++) { cyc += (double)hbuf[8 * i]; ticks += (double)hbuf[8 * i + 1]; chunks += (double)hbuf[8 * i + 2]; r0 = std::min(r0, hbuf[8 * i + 3]); r1 = std::max(r1, hbuf[8 * i + 3] + hbuf[8 * i + 1]); }
            fprintf(stderr, "[pnn-f32s-diag] M=%ld K=%.0f N=%d ncls=%d nseg=%d: %zu WGs, loop %.0f cycles for %.0f chunks = %.0f cycles per chunk (160 = the chain), %.2f us, clock %.0f MHz; first loop start -> last loop end %.1f us\n",
                    M, L.k_total, p.Cout, p.ncls, nseg, nwg, cyc / nwg, chunks / nwg, cyc / std::max(1.0, chunks), ticks / nwg / 100.0, cyc / std::max(1.0, ticks) * 100.0, (double)(r1 - r0) / 100.0);
        }
        c->stat_gemm_launches++; c->stat_launches++;
        c->stat_gemm_flops += flops;
        return PNN_OK;
    }
    if (f32k) {
        if (next) { p.W2p = next->d_w; p.Npad2 = next->proto.Npad; p.K2chunks = next->proto.chunk_begin[1]; p.part = part; }
        const long seg_mode = c->opt_f32_seg_mode;
        auto legal = [&](int code) {
            if (code >= ntile && (nseg == 1 || seg_mode == 0)) return false;
            if (code < ntile && nseg > 1 && seg_mode == 1) return false;
            const TileCfg t = tapgemm_f32_cfg(code % ntile);
            if (fcseg && L.fc_seg_chunks % (2 * t.kc)) return false;  // an even number of whole stages per segment (tapgemm_f32_kernel, SEQ = 2)
            if (fcseg && t.rt * t.nt >= 8) return false;              // ... and room for the running total beside the accumulators (the 256 x 128 tile spills)
            return (one_tap || cpt % t.kc == 0) && (!next || tapgemm_f32_can_fuse(code % ntile));
        };
        float* const Yreal = Y;
        const float* const bias_real = L.d_bias;
        const int act_real = L.proto.act;
        launch = [&, p, pseq, Yreal, bias_real, act_real, fcseg](int code) {
            if (code >= ntile) return launch_tapgemm_f32(pseq, code - ntile, false, s);
            const hipError_t e = launch_tapgemm_f32(p, code, next != nullptr, s);
            if (e != hipSuccess || p.nseg <= 1 || fcseg) return e;
            return launch_seg_reduce(p.Y, p.nseg, out_floats, p.Cout, bias_real, act_real, Yreal, s);
        };
        double cost_par = 0, cost_seq = 0;
        cfg = choose_cfg_f32(c, M, p.Cout, p.ncls * nseg, p.Cin, L.k_total, next != nullptr, &cost_par);
        if (cfg < 0) return fail(c, PNN_E_ARG, "no tapgemm_f32 tile fits a layer with %d-deep taps", p.Cin);
        if (fcseg && !legal(cfg)) {                   // the rule knows nothing of the segments' stage count: the same tile with the legal stage depth, else any legal one
            const TileCfg want = tapgemm_f32_cfg(cfg);
            int alt = -1;
            for (int i = 0; i < ntile; i++) if (legal(i) && (alt < 0 || (tapgemm_f32_cfg(i).rt == want.rt && tapgemm_f32_cfg(i).nt == want.nt))) alt = i;
            if (alt < 0) return fail(c, PNN_E_ARG, "no tapgemm_f32 tile fits K segments of %d chunks", L.fc_seg_chunks);
            cfg = alt;
        }
        if (nseg > 1) {
            // the planes cost a write and a read of nseg x the output and a launch (in cycles at 2.4 GHz, ~4 TB/s through L2 / MALL)
            cost_par += (double)(nseg + 1) * (double)out_floats * 4.0 / 4.0e12 * 2.4e9 + 9000.0;
            const int cs = choose_cfg_f32(c, M, p.Cout, p.ncls, p.Cin, L.k_total, false, &cost_seq);
            if (seg_mode == 1 || (seg_mode < 0 && cs >= 0 && cost_seq < cost_par)) cfg = ntile + cs;
        }
        bool tune = c->opt_f32_cfg < 0 && (c->opt_autotune == 1 || (c->opt_autotune == 2 && flops >= 4.0e9));
        if (tune) {                                   // never while the caller's stream is being captured into a hipGraph
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) tune = false;
        }
        if (tune) {
            const void* key = (const void*)((const char*)&L + 8 + (next ? 1 : 0));   // offsets 0-7: the split-precision launches of this layer
            float best_us = -1.f;
            const int rule = cfg;
            const int trc = tuned_cfg(c, key, M, nseg > 1 ? 2 * ntile : ntile, rule, legal, launch, s, &cfg, &best_us);
            if (trc) return trc;
            if (best_us >= 0.f && debug) {
                const TileCfg tb = tapgemm_f32_cfg(cfg % ntile), th = tapgemm_f32_cfg(rule % ntile);
                fprintf(stderr, "[pnn] f32 autotune M=%ld K=%.0f N=%d ncls=%d nseg=%d: best {%d,%d,%d}%s %.1f us (rule {%d,%d,%d}%s)\n", M, L.k_total, p.Cout, p.ncls, nseg,
                        tb.rt, tb.nt, tb.kc, cfg >= ntile ? " seq" : "", best_us, th.rt, th.nt, th.kc, rule >= ntile ? " seq" : "");
            }
        }
        if (tiles_out) { const TileCfg t = tapgemm_f32_cfg(cfg % ntile); *tiles_out = (int)((p.Cout + 32L * t.nt - 1) / (32L * t.nt)); }
    }
    const bool seq = f32k && nseg > 1 && cfg >= ntile;
    const TileCfg t = tapgemm_f32_cfg(cfg % ntile);
    if (debug) fprintf(stderr, "[pnn] gemm M=%ld K=%.0f N=%d ncls=%d -> %s cfg %d {rt %d, nt %d, kc %d}%s%s\n", M, L.k_total, p.Cout, p.ncls,
                       f32k ? "f32" : "legacy", cfg, t.rt, t.nt, t.kc, next ? " + fused output layer" : "",
                       nseg > 1 ? (seq ? ", K segments in sequence" : ", K segments in parallel + reduce") : "");
    if (profile || c->opt_time_launches) {
        pnn_ctx::LaunchRec r;
        HIPCHK(c, hipEventCreate(&r.e0));
        HIPCHK(c, hipEventCreate(&r.e1));
        r.kind = 0;
        r.flops = flops;
        const LaunchEvents ev{r.e0, r.e1};            // recorded by the launch itself: the kernel's own begin -> end
        g_launch_events = &ev;
        const hipError_t le = launch(cfg);
        g_launch_events = nullptr;
        HIPCHK(c, le);
        if (profile) {
            HIPCHK(c, hipEventSynchronize(r.e1));
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
            fprintf(stderr, "[pnn-prof] M=%ld K=%.0f N=%d ncls=%d %s cfg=%d rt=%d nt=%d kc=%d mf=%d us=%.1f tflops=%.1f\n", M, L.k_total,
                    p.Cout, p.ncls, f32k ? "f32" : "legacy", cfg, t.rt, t.nt, t.kc, t.mf, ms * 1e3, r.flops / (ms * 1e-3) / 1e12);
            (void)hipEventDestroy(r.e0);
            (void)hipEventDestroy(r.e1);
        } else {
            c->launch_recs.push_back(r);
        }
    } else {
        HIPCHK(c, launch(cfg));
    }
    if (nseg > 1 && !seq) c->stat_launches++;          // the reduction of the parallel form (launched by `launch`)
    static const bool diag = getenv("PNN_F32_DIAG") != nullptr;     // diagnostic library only (make diag): per-workgroup cycle stamps
    if (diag && f32k) {
        HIPCHK(c, hipStreamSynchronize(s));
        if (dev_reserve(c, c->stage_tbs, (size_t)16 << 20)) return PNN_E_NOMEM;
        TapGemmParams q = seq ? pseq : p;
        if (next) { q.W2p = next->d_w; q.Npad2 = next->proto.Npad; q.K2chunks = next->proto.chunk_begin[1]; q.part = part; }
        q.Xlo = c->stage_tbs.p;
        HIPCHK(c, hipMemset(c->stage_tbs.p, 0, (size_t)16 << 20));
        for (int rep = 0; rep < 400; rep++) HIPCHK(c, launch_tapgemm_f32(q, cfg % ntile, next != nullptr, s));   // back to back: the stamps that
                                                                                           // stay are the last launch's, at the steady-state clock
        HIPCHK(c, hipStreamSynchronize(s));
        const size_t nwg = (size_t)((M + 128L * t.rt - 1) / (128L * t.rt)) * ((p.Cout + 32L * t.nt - 1) / (32L * t.nt)) * p.ncls * (seq ? 1 : nseg);
        std::vector<unsigned long long> hbuf(8 * nwg);
        HIPCHK(c, hipMemcpy(hbuf.data(), c->stage_tbs.p, hbuf.size() * 8, hipMemcpyDeviceToHost));
        double sum[4] = {0, 0, 0, 0};
        unsigned long long r0 = ~0ull, r1 = 0;
        for (size_t i = 0; i < nwg; i++) {
            for (int k = 0; k < 4; k++) sum[k] += (double)hbuf[8 * i + k];
            r0 = std::min(r0, hbuf[8 * i + 4]); r1 = std::max(r1, hbuf[8 * i + 4] + hbuf[8 * i + 3]);
        }
        size_t late = 0; unsigned long long life_max = 0;     // workgroups that start > 5 us behind the first; the longest lifetime
        for (size_t i = 0; i < nwg; i++) { late += hbuf[8 * i + 4] > r0 + 500; life_max = std::max(life_max, hbuf[8 * i + 3]); }
        const double chunks = std::ceil(L.k_total / 16.0 / p.ncls / (seq ? 1 : nseg) / t.kc) * t.kc;
        const double cyc = (sum[0] + sum[1] + sum[2]) / nwg, rt_ticks = sum[3] / nwg;
        fprintf(stderr, "[pnn-f32diag] M=%ld K=%.0f N=%d {%d,%d,%d}%s: %zu WGs; wave 0 mean cycles: prologue %.0f  loop %.0f (MFMA work %.0f = %.3f)  epilogue %.0f;"
                " lifetime %.1f us (max %.1f), in-kernel clock %.0f MHz; first start -> last end %.1f us, %zu workgroups start > 5 us late\n", M, L.k_total, p.Cout, t.rt, t.nt, t.kc, next ? "+out" : "", nwg,
                sum[0] / nwg, sum[1] / nwg, chunks * 8 * t.rt * t.nt * 64, chunks * 8 * t.rt * t.nt * 64 / (sum[1] / nwg), sum[2] / nwg, rt_ticks / 100.0, (double)life_max / 100.0,
                cyc / (rt_ticks / 100.0), (double)(r1 - r0) / 100.0, late);
    }
    c->stat_gemm_launches++; c->stat_launches++;
    c->stat_gemm_flops += flops;
    c->stat_gemm_flops_skipped += flops * (1.0 - g_last_issued_frac);
    return PNN_OK;
}

// `next` (optional): a following fully-connected layer with <= 64 outputs that the ring kernel applies to its output tile
// in LDS; `part` then receives the per-column-tile partial products [tiles][M][64] and *tiles_out their count (the caller
// finishes with launch_fuse_reduce).  Y / Yhi / Yi must be null in that case.
int run_gemm_sp(pnn_ctx* c, const GemmLayer& L, const void* Xhi, const void* Xlo, float* Y, void* Yhi, void* Ylo, int32_t* Yi,
                long nblocks, hipStream_t s, const GemmLayer* next = nullptr, float* part = nullptr, int* tiles_out = nullptr,
                const Conv1Params* first = nullptr, bool x_is_f32 = false, int seg_chunks = 0, bool* query_fuse_first = nullptr,
                const TConv1Params* lastp = nullptr)
{
    // `lastp` (optional): the net's last layer (the one-output-channel transposed convolution), which reads this layer's float
    // output Y.  It has NOT been launched: an image-kernel configuration that can applies it to its output tile in registers
    // (no Y round trip, one launch less), any other configuration gets it launched here behind the GEMM.
    // query_fuse_first (optional): launch NOTHING; report whether this call would run the image kernel with `first` fused in
    // -- and is sure of it: no tuning sweep ahead (a sweep launches the other kernel families too, which need the maps in memory)
    if (query_fuse_first) *query_fuse_first = false;
    // x_is_f32: Xhi holds plain f32 rows (an FC net's input as the caller handed it over); only the small-M kernel takes
    // that (it splits in registers), any other choice gets split_kernel launched in front (into ws[2]).
    // seg_chunks > 0: K-segment mode of an FC output layer (small-M kernel only): raw partials to `part`, see fc_pass.
    // `first` (optional): the Cin = 1 convolution that produces this layer's input Xhi.  It has NOT been launched: a
    // convimg configuration computes it inside the kernel (no 50 MB round trip of the maps), any other configuration gets
    // it launched here in front of the GEMM.
    TapGemmParams p = L.proto;
    if (next) {
        p.W2p = next->d_w_sp; p.Npad2 = next->proto.Npad; p.K2chunks = next->proto.chunk_begin[1]; p.part = part;
    }
    p.X = (const float*)Xhi; p.Xlo = Xlo; p.Wp = L.d_w_sp;
    static const bool diag = getenv("PNN_SP_DIAG") != nullptr;   // diagnostic library only: phase stamps of every workgroup
    if (diag) {
        if (dev_reserve(c, c->stage_tbs, (size_t)64 << 20)) return PNN_E_NOMEM;
        p.Xlo = c->stage_tbs.p;
    } p.bias = L.d_bias; p.Y = Y; p.Yhi = Yhi; p.Ylo = Ylo; p.Yi = Yi;
    p.mean = c->mean; p.out_scale = L.sp_inv_scale; p.range_flag = c->h_range;
    p.pm_groups = c->opt_ring_pm == 0 ? -1 : c->opt_ring_pm == 2 ? 1 : 0;   // ring kernel, position-major tiles: never / whenever possible / by its model (launch_tapgemm_ring)
    const long M = nblocks * p.SH * p.SW;
    if (M > 0x7fffffffL) return fail(c, PNN_E_ARG, "batch too large for one pass");
    p.M = (int)M;
    const double xb = 4.0 * (double)nblocks * p.IH * p.IW * p.Cin;
    if (xb >= 2147483648.0) return fail(c, PNN_E_ARG, "activation plane of %.0f bytes exceeds the 2 GiB descriptor bound", xb);
    p.x_bytes = (unsigned)xb;
    const int cpt = p.Cin / 16;
    const bool one_tap = (L.k_total == (double)p.Cin);
    static const bool diag0 = getenv("PNN_SP_DIAG") != nullptr;
    // Few output tiles (the in-loop single-block calls, the batching service's handfuls): one wave per 32 x 32 tile over all
    // CUs instead of one or two big workgroups walking K alone.  Same per-output summation order: bit-identical.
    const bool small = seg_chunks > 0 || (!next && !diag0 && c->opt_small && c->opt_sp_cfg < 0 && tapgemm_small_tiles(p) <= c->opt_small_tiles);
    if (small && query_fuse_first) return PNN_OK;
    if (small) {
        if (seg_chunks > 0) p.part = part;
        if (first) { HIPCHK(c, launch_conv_cin1(*first, s)); c->stat_launches++; }
        static const bool dbg = getenv("PNN_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[pnn] sp-gemm M=%ld K=%.0f N=%d ncls=%d -> small kernel (%ld tiles%s%s)\n", M, L.k_total, p.Cout, p.ncls,
                         tapgemm_small_tiles(p), x_is_f32 ? ", f32 input" : "", seg_chunks ? ", K segments" : "");
        const double flops = 2.0 * (double)M * L.k_total * p.Cout;
        if (c->opt_time_launches) {
            pnn_ctx::LaunchRec r;
            HIPCHK(c, hipEventCreate(&r.e0));
            HIPCHK(c, hipEventCreate(&r.e1));
            r.kind = 5; r.flops = flops;
            const LaunchEvents ev{r.e0, r.e1};
            g_launch_events = &ev;
            const hipError_t le = launch_tapgemm_small(p, x_is_f32, seg_chunks, s, x_is_f32 ? c->host_input : nullptr);
            g_launch_events = nullptr;
            HIPCHK(c, le);
            c->launch_recs.push_back(r);
        } else {
            HIPCHK(c, launch_tapgemm_small(p, x_is_f32, seg_chunks, s, x_is_f32 ? c->host_input : nullptr));
        }
        c->stat_gemm_launches++; c->stat_launches++;
        c->stat_gemm_flops += flops;
        if (tiles_out) *tiles_out = seg_chunks > 0 ? (int)(((long)(L.k_total / 16.0) + seg_chunks - 1) / seg_chunks) : 0;
        if (lastp) { HIPCHK(c, launch_tconv_cout1(*lastp, s)); c->stat_launches++; }
        return PNN_OK;
    }
    if (x_is_f32 && !query_fuse_first) {              // the big-tile kernels read split activations
        const long nin = nblocks * (long)p.IH * p.IW * p.Cin;
        HIPCHK(c, launch_split((const float*)Xhi, nin, c->ws[2].p, nullptr, c->h_range, s));
        c->stat_launches++;
        p.X = (const float*)c->ws[2].p;
    }
    const int nsp = tapgemm_sp_num_cfgs(), nci = convimg_sp_num_cfgs(), nrg = tapgemm_ring_num_cfgs();
    // configuration codes: [0, nsp) = tapgemm_sp_kernel tiles, then the convimg_sp_kernel tiles (images resident in
    // LDS), then the tapgemm_ring_kernel tiles (LDS-DMA ring)
    p.zero = c->d_zero;
    auto cfg_of = [&](int code) { return code < nsp ? tapgemm_sp_cfg(code) : code < nsp + nci ? convimg_sp_cfg(code - nsp) : tapgemm_ring_cfg(code - nsp - nci); };
    auto kind_of = [&](int code) { return code < nsp ? "" : code < nsp + nci ? "img" : "ring"; };
    auto legal = [&](int code) {
        // fused output layer: only the 160-column tile -- its column tiles ARE the K segments of the output layer's canonical
        // summation order (kFuseSegChunks chunks each), which the small-M kernel reproduces for every other batch size
        if (next) return code >= nsp + nci && !diag && c->opt_ring && tapgemm_ring_can_fuse(code - nsp - nci) &&
                         32 * tapgemm_ring_cfg(code - nsp - nci).nt * (4 / tapgemm_ring_cfg(code - nsp - nci).wm) == 16 * kFuseSegChunks;
        if (code < nsp) return one_tap || cpt % tapgemm_sp_cfg(code).kc == 0;
        if (code < nsp + nci) return !diag && c->opt_convimg && convimg_images(p, convimg_sp_cfg(code - nsp), one_tap) > 0;
        return !diag && c->opt_ring && (one_tap || cpt % tapgemm_ring_cfg(code - nsp - nci).kc == 0);
    };
    bool fused_last = false;                          // the last launch(code) had the image kernel apply `lastp`
    auto launch = [&](int code) {
        fused_last = false;
        if (first) {
            if (code >= nsp && code < nsp + nci) {
                const TileCfg t = convimg_sp_cfg(code - nsp);
                const int g = convimg_images(p, t, one_tap);
                if (c->opt_fuse_first && convimg_sp_can_fuse_first(p, t, g, first->s, first->k)) {
                    TapGemmParams q = p;
                    q.X0 = first->X; q.W0 = first->W; q.B0 = first->bias; q.s0 = first->s; q.k0 = first->k; q.pad0 = first->pad;
                    q.W0sp = first->Wsp; q.scale0 = first->out_scale; q.Npad0 = first->npad;
                    if (!first->X && c->lazy.plane) {  // the gather fused in as well (pnn_predict_tbs_device decided so by query)
                        q.plane0 = c->lazy.plane; q.tbs0 = c->lazy.tbs; q.pel0 = c->lazy.pel_bytes; q.unit0 = c->lazy.unit;
                        q.branch0 = first->IH > first->IW ? 1 : 0;       // above portion w x 3w, left portion 2w x w
                        q.w0 = q.branch0 ? first->IW : first->IH;
                    }
                    return launch_convimg_sp(q, code - nsp, g, s);
                }
            }
            if (!first->X) return hipErrorInvalidValue;   // contexts not gathered (fused gather) but this configuration needs the maps in memory: never chosen after the query
            const hipError_t e = launch_conv_cin1(*first, s);
            if (e != hipSuccess) return e;
        }
        hipError_t e;
        if (code < nsp) e = launch_tapgemm_sp(p, code, s);
        else if (code < nsp + nci) {
            const TileCfg t = convimg_sp_cfg(code - nsp);
            const int g = convimg_images(p, t, one_tap);
            // (a last layer that carries the completion signal of a host call is launched: the fused form has no such signal)
            if (lastp && c->opt_fuse_tail && !lastp->done.host_flag && convimg_sp_can_fuse_last(p, t, g, *lastp)) {
                TapGemmParams q = p;
                q.Y = nullptr;
                q.W1 = lastp->W; q.Y1 = lastp->Y; q.Yi1 = lastp->Yi; q.bias1 = lastp->bias; q.k1 = lastp->k; q.s1 = lastp->s; q.pad1 = lastp->pad;
                fused_last = true;
                return launch_convimg_sp(q, code - nsp, g, s);
            }
            e = launch_convimg_sp(p, code - nsp, g, s);
        } else e = launch_tapgemm_ring(p, code - nsp - nci, s);
        if (e != hipSuccess || !lastp) return e;
        return launch_tconv_cout1(*lastp, s);
    };
    int cfg = choose_cfg_sp(c, M, p.Cout, p.ncls, p.Cin, L.k_total);
    if (c->opt_sp_cfg >= nsp && legal((int)c->opt_sp_cfg)) cfg = (int)c->opt_sp_cfg;
    else if (c->opt_sp_cfg < 0) {
        const int ci = c->opt_convimg ? choose_cfg_convimg(p, one_tap) : -1;
        const int ri = c->opt_ring ? choose_cfg_ring(p, M, one_tap, L.k_total, next != nullptr) : -1;
        const bool ring_conv = !one_tap && ri >= 0 && ((p.Cout % 128 == 0 && L.k_total / p.ncls >= 1152.0) || (p.ncls == 4 && p.Cout == 64 && M >= 8192) || pnn_ring_few_images(p, M, L.k_total));   // see choose_cfg_ring
        if (ring_conv && legal(nsp + nci + ri)) cfg = nsp + nci + ri;
        else if (ci >= 0 && legal(nsp + ci) && !one_tap) cfg = nsp + ci;
        else if (ri >= 0 && legal(nsp + nci + ri)) cfg = nsp + nci + ri;
        else if (ci >= 0 && legal(nsp + ci)) cfg = nsp + ci;
    }
    // autotune: 1 = every split GEMM, 2 (default) = only launches of >= 4 GFLOP, where trying all configurations once
    // (~70 x 4 launches) costs a few tens of milliseconds and the choice is worth 10-20 %; 0 = rule-based choice only.
    // All configurations give bit-identical results, so the choice never shows in the predictions.
    bool tune = c->opt_autotune == 1 || (c->opt_autotune == 2 && 2.0 * (double)M * L.k_total * p.Cout >= 4.0e9);
    if (tune && c->tuned.find(std::make_pair((const void*)((const char*)&L + (next ? 1 : 0) + (first ? 2 : 0) + (lastp ? 4 : 0)), M)) == c->tuned.end()) {
        // timing configurations means synchronising on the caller's stream: never while that stream is being captured
        // into a hipGraph (the rule-based choice is used instead, nothing is remembered)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) tune = false;
    }
    if (query_fuse_first) {
        const void* key = (const void*)((const char*)&L + (next ? 1 : 0) + (first ? 2 : 0) + (lastp ? 4 : 0));
        int code = cfg;
        if (tune && c->opt_sp_cfg < 0) {
            auto it = c->tuned.find(std::make_pair(key, M));
            if (it == c->tuned.end()) return PNN_OK;  // a sweep is ahead
            code = it->second;
        }
        if (first && code >= nsp && code < nsp + nci && legal(code)) {
            const TileCfg t = convimg_sp_cfg(code - nsp);
            *query_fuse_first = c->opt_fuse_first && convimg_sp_can_fuse_first(p, t, convimg_images(p, t, one_tap), first->s, first->k);
        }
        return PNN_OK;
    }
    if (tune && c->opt_sp_cfg < 0) {
        const void* key = (const void*)((const char*)&L + (next ? 1 : 0) + (first ? 2 : 0) + (lastp ? 4 : 0));
        const int rule = cfg;
        float best_us = -1.f;
        // The sweep launches every configuration several times; a last layer that carries the completion signal of a host call
        // (`lastp->done`) must not raise it from there -- the flag would already stand at this call's sequence number while the
        // final launch below is still writing its results.  The sweep runs with an unsignalled copy; only the final launch signals.
        TConv1Params quiet;
        const TConv1Params* const signalled = lastp;
        if (lastp && lastp->done.host_flag) { quiet = *lastp; quiet.done = DoneSignal{nullptr, nullptr, 0, 0}; lastp = &quiet; }
        const int trc = tuned_cfg(c, key, M, nsp + nci + nrg, rule, legal, launch, s, &cfg, &best_us);
        lastp = signalled;
        if (trc) return trc;
        if (best_us >= 0.f && getenv("PNN_DEBUG")) {
            const TileCfg tb = cfg_of(cfg), th = cfg_of(rule);
            fprintf(stderr, "[pnn] autotune M=%ld K=%.0f N=%d ncls=%d: best %s{%d,%d,%d,wm%d,d%d} %.1f us (heuristic %s{%d,%d,%d,wm%d,d%d})\n", M,
                    L.k_total, p.Cout, p.ncls, kind_of(cfg), tb.rt, tb.nt, tb.kc, tb.wm, tb.d, best_us, kind_of(rule), th.rt, th.nt,
                    th.kc, th.wm, th.d);
        }
    }
    if (next && !legal(cfg)) {                        // checked BEFORE anything is launched: the caller falls back to separate launches
        cfg = -1;
        for (int i = nsp + nci; i < nsp + nci + nrg && cfg < 0; i++) if (legal(i)) cfg = i;
        if (cfg < 0) return fail(c, PNN_E_ARG, "no ring configuration can fuse the next layer");
    }
    static const bool debug = getenv("PNN_DEBUG") != nullptr;
    static const bool profile = getenv("PNN_PROFILE") != nullptr;
    const TileCfg t = cfg_of(cfg);
    if (debug) fprintf(stderr, "[pnn] sp-gemm M=%ld K=%.0f N=%d ncls=%d -> cfg %d %s{rt %d, nt %d, kc %d, wm %d, d %d}\n", M, L.k_total, p.Cout, p.ncls,
                       cfg, kind_of(cfg), t.rt, t.nt, t.kc, t.wm, t.d);
    if (profile || c->opt_time_launches) {
        pnn_ctx::LaunchRec r;
        HIPCHK(c, hipEventCreate(&r.e0));
        HIPCHK(c, hipEventCreate(&r.e1));
        r.kind = cfg < nsp ? 2 : cfg < nsp + nci ? 3 : 4;
        r.flops = 2.0 * (double)M * L.k_total * p.Cout + (next ? 2.0 * (double)M * next->k_total * next->proto.Cout : 0.0);
        const LaunchEvents ev{r.e0, r.e1};            // recorded by the launch itself: the kernel's own begin -> end
        g_launch_events = &ev;
        const hipError_t le = launch(cfg);
        g_launch_events = nullptr;
        HIPCHK(c, le);
        if (profile) {
            HIPCHK(c, hipEventSynchronize(r.e1));
            float ms = 0.f;
            HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
            fprintf(stderr, "[pnn-prof] M=%ld K=%.0f N=%d ncls=%d cfg=%d rt=%d nt=%d kc=%d mf=%d us=%.1f tflops=%.1f\n", M, L.k_total,
                    p.Cout, p.ncls, cfg, t.rt, t.nt, t.kc, t.mf, ms * 1e3, r.flops / (ms * 1e-3) / 1e12);
            (void)hipEventDestroy(r.e0);
            (void)hipEventDestroy(r.e1);
        } else {
            c->launch_recs.push_back(r);
        }
    } else {
        HIPCHK(c, launch(cfg));
    }
    if (diag) {
        HIPCHK(c, hipStreamSynchronize(s));
        const TileCfg tt = tapgemm_sp_cfg(cfg);   // (diag runs never take the convimg kernel)
        const long bm = 32L * tt.rt * tt.wm, bn = 32L * tt.nt * (4 / tt.wm);
        const size_t nwg = (size_t)((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn) * p.ncls;
        std::vector<unsigned long long> h(4 * nwg);
        HIPCHK(c, hipMemcpy(h.data(), c->stage_tbs.p, h.size() * 8, hipMemcpyDeviceToHost));
        double sum[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < nwg; i++) for (int k = 0; k < 4; k++) sum[k] += (double)h[4 * i + k];
        const double stages = std::ceil(L.k_total / 16.0 / p.ncls / tt.kc);
        fprintf(stderr, "[pnn-diag] M=%ld K=%.0f N=%d cfg {%d,%d,%d,wm%d}: per stage (cycles, wave 0 mean over %zu WGs): issue %.0f  mfma %.0f  store %.0f  barrier %.0f\n",
                M, L.k_total, p.Cout, tt.rt, tt.nt, tt.kc, tt.wm, nwg, sum[0] / nwg / stages, sum[1] / nwg / stages, sum[2] / nwg / stages, sum[3] / nwg / stages);
    }
    c->stat_gemm_launches++; c->stat_launches++;
    if (lastp && !fused_last) c->stat_launches++;    // the net's last layer went out as a launch of its own
    c->stat_gemm_flops += 2.0 * (double)M * L.k_total * p.Cout;
    if (cfg >= nsp + nci) c->stat_gemm_flops_skipped += 2.0 * (double)M * L.k_total * p.Cout * (1.0 - g_last_issued_frac);   // (ring launches may skip padding taps)
    if (next) {
        c->stat_gemm_flops += 2.0 * (double)M * next->k_total * next->proto.Cout;
        if (tiles_out) *tiles_out = (int)((p.Cout + 32L * t.nt * (4 / t.wm) - 1) / (32L * t.nt * (4 / t.wm)));
    }
    return PNN_OK;
}

}  // namespace

long chunk_blocks(const pnn_ctx* c, const Model* m)
{
    const double per_block = 4.0 * (m->is_fc ? 2.0 * kHidden : 2.0 * m->pmax + 80.0 * m->C);
    long n = c->opt_max_chunk > 0 ? c->opt_max_chunk : (long)((double)c->ws_cap_bytes / per_block);
    // every activation tensor of a pass must stay below the 2 GiB bound of a buffer descriptor
    const double biggest = 4.0 * std::max((double)m->pmax, 5.0 * m->width * m->width);
    n = std::min(n, (long)(2147483000.0 / biggest));
    return std::max(1L, std::min(n, 1L << 20));
}

namespace {

// The branches of a conv pass overlap on two streams while one branch leaves most of the chip idle.  The fork/join costs
// ~25 us of event traffic between the two queues (measured: single-block calls of the 16x16 net 88 -> 97 us, 32x32 157 ->
// 147 us, 64x64 261 -> 220 us), so only the nets whose branches are longer than that take it, option "branch_streams" = 2
// forces it.  Not under the per-launch timing modes, which assume one stream.
bool branches_overlap(const pnn_ctx* c, const Model* m, long nb)
{
    static const bool profile = getenv("PNN_PROFILE") != nullptr;
    if (!c->opt_branch_streams || m->is_fc || profile || c->opt_time_launches) return false;
    return nb * m->width * m->width <= 8192 && (m->width >= 32 || c->opt_branch_streams == 2);
}

// Passes at batch on the split-precision kernels overlap their branches too, since position-major tiles (pnn_gemm_ring.hip):
// the later layers of a branch are one or two workgroups per CU, and with unequal workgroups a launch ends with its heaviest
// one while the CUs that drew corner positions idle -- the other branch's launches fill them.  Measured, conv 16x16 at batch
// 1024, same box, rule-based tiles: one stream 0.3775 ms, whole branches on two streams 0.3706 (+1.8 %; before position-major
// tiles the same overlap was 2 % SLOWER), only the layers behind each branch's first GEMM overlapped (the
// image kernels, which fill the chip, one after the other) 0.3824.  Only after a pass of the same shape has run on one stream
// without a tuning sweep (pnn_ctx::overlap_ready), and not under the per-launch timing modes or a forced configuration.
bool branches_overlap_at_batch(const pnn_ctx* c, const Model* m, long nb)
{
    static const bool profile = getenv("PNN_PROFILE") != nullptr;
    if (!c->opt_branch_streams || m->is_fc || profile || c->opt_time_launches || c->opt_sp_cfg >= 0) return false;
    if (m->branch[0].size() < 2 || m->branch[1].size() < 2) return false;
    // (exact-f32 passes too, option f32_kernel: their launches end with a tail of big workgroups -- one or two per CU -- that the other
    // branch's launches fill)
    return nb * m->width * m->width >= 65536 && (pass_uses_split(c, m, nb) || (c->opt_precision == 0 && c->opt_f32_overlap));
}

int ensure_ws(pnn_ctx* c, const Model* m, long nb)
{
    int rc;
    if ((rc = dev_reserve(c, c->ws[0], (size_t)nb * m->pmax * 4))) return rc;
    if ((rc = dev_reserve(c, c->ws[1], (size_t)nb * m->pmax * 4))) return rc;
    if (m->is_fc) {
        if ((rc = dev_reserve(c, c->ws[2], (size_t)nb * 5 * m->width * m->width * 4))) return rc;   // split-precision input planes
    } else {
        if ((rc = dev_reserve(c, c->ws[2], (size_t)nb * 48 * m->C * 4))) return rc;
        if ((rc = dev_reserve(c, c->ws[3], (size_t)nb * 32 * m->C * 4))) return rc;
        if (branches_overlap(c, m, nb) || branches_overlap_at_batch(c, m, nb)) {
            if ((rc = dev_reserve(c, c->ws[4], (size_t)nb * m->pmax * 4))) return rc;
            if ((rc = dev_reserve(c, c->ws[5], (size_t)nb * m->pmax * 4))) return rc;
            if (!c->side_stream) {
                HIPCHK(c, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
                HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                HIPCHK(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
            }
        }
    }
    return PNN_OK;
}

}  // namespace

// Which arithmetic a pass runs on: the context's, whatever the batch size (one summation order at every batch size).
bool pass_uses_split(const pnn_ctx* c, const Model*, long) { return c->opt_precision == 1; }

namespace {

int fc_pass(pnn_ctx* c, Model* m, const float* d_ctx, bool ctx_is_split, long nb, float* d_out, int32_t* d_dst, hipStream_t s)
{
    float* P0 = (float*)c->ws[0].p; float* P1 = (float*)c->ws[1].p;
    int rc;
    if (pass_uses_split(c, m, nb)) {
        // split-precision chain: hidden activations travel in the split f16 layout (same byte count as f32); the input is
        // already split when the gather wrote it, else the first layer's kernel splits it (small-M kernel: in registers)
        const int n_out = m->fc[3].proto.Cout;
        // Output layer of the 4x4 / 8x8 nets (<= 64 outputs): summed in K segments of kFuseSegChunks chunks + fuse_reduce, at
        // EVERY batch size -- by the ring kernel's fused output layer (big batches: the 1200-wide activations of the third
        // hidden layer never leave the workgroups that produce them) or by the small-M kernel's K-segment mode.
        const bool seg_model = n_out <= 64 && n_out % 4 == 0;
        const bool ring_fuse = seg_model && c->opt_fuse_last && c->opt_ring && c->opt_sp_cfg < 0 && nb >= 1024;
        if ((rc = run_gemm_sp(c, m->fc[0], d_ctx, nullptr, nullptr, P0, nullptr, nullptr, nb, s, nullptr, nullptr, nullptr, nullptr, !ctx_is_split))) return rc;
        if ((rc = run_gemm_sp(c, m->fc[1], P0, nullptr, nullptr, P1, nullptr, nullptr, nb, s))) return rc;
        if (seg_model) {
            if ((rc = dev_reserve(c, c->ws[3], (size_t)20 * nb * 64 * 4))) return rc;
            float* part = (float*)c->ws[3].p;
            int tiles = 0;
            if (ring_fuse) {
                if ((rc = run_gemm_sp(c, m->fc[2], P1, nullptr, nullptr, nullptr, nullptr, nullptr, nb, s, &m->fc[3], part, &tiles))) return rc;
            } else {
                if ((rc = run_gemm_sp(c, m->fc[2], P1, nullptr, nullptr, P0, nullptr, nullptr, nb, s))) return rc;
                // small passes: K segments and their reduction in ONE launch (fc_out_small_kernel; the same bits as the small
                // kernel's K-segment mode + fuse_reduce_kernel, two launch floors of ~4 us less per single-block call)
                TapGemmParams q = m->fc[3].proto;
                q.X = P0; q.Wp = m->fc[3].d_w_sp; q.bias = m->fc[3].d_bias; q.out_scale = m->fc[3].sp_inv_scale; q.mean = c->mean;
                q.Y = d_out; q.Yi = d_dst; q.M = (int)nb;
                if (c->opt_fc_out && c->opt_small && c->opt_sp_cfg < 0 && nb <= 2048 && fc_out_small_fits(q, kFuseSegChunks)) {
                    const double flops = 2.0 * (double)nb * m->fc[3].k_total * q.Cout;
                    if (c->opt_time_launches) {
                        pnn_ctx::LaunchRec r;
                        HIPCHK(c, hipEventCreate(&r.e0));
                        HIPCHK(c, hipEventCreate(&r.e1));
                        r.kind = 5; r.flops = flops;
                        const LaunchEvents ev{r.e0, r.e1};
                        g_launch_events = &ev;
                        const hipError_t le = launch_fc_out_small(q, kFuseSegChunks, s, take_done_signal(c));
                        g_launch_events = nullptr;
                        HIPCHK(c, le);
                        c->launch_recs.push_back(r);
                    } else {
                        HIPCHK(c, launch_fc_out_small(q, kFuseSegChunks, s, take_done_signal(c)));
                    }
                    c->stat_gemm_launches++; c->stat_launches++;
                    c->stat_gemm_flops += flops;
                    return PNN_OK;
                }
                if ((rc = run_gemm_sp(c, m->fc[3], P0, nullptr, nullptr, nullptr, nullptr, nullptr, nb, s, nullptr, part, &tiles, nullptr, false, kFuseSegChunks))) return rc;
            }
            if (tiles <= 0 || tiles > 20) return fail(c, PNN_E_ARG, "output layer: %d K segments do not fit the partial buffer", tiles);
            HIPCHK(c, launch_fuse_reduce(part, tiles, (int)nb, n_out, m->fc[3].d_bias, m->fc[3].sp_inv_scale, c->mean, d_out, d_dst, s, take_done_signal(c)));
            c->stat_launches++;
            return PNN_OK;
        }
        if ((rc = run_gemm_sp(c, m->fc[2], P1, nullptr, P0, nullptr, nullptr, nullptr, nb, s))) return rc;
        return run_gemm(c, m->fc[3], P0, d_out, d_dst, nb, s);
    }
    // Small passes: the hidden activations travel in CHAIN ORDER between the three small-kernel launches (pnn_gemm_f32_small.hip: a
    // consumer's loaders then fetch a chunk's activations with one 16-byte instruction instead of four 4-byte ones); the output layer's
    // kernel reads channel order, so the third layer writes that.  All three or none: they have the same number of tiles.
    const bool out64 = m->fc[3].proto.Cout <= 64 && m->fc[3].proto.Cout % 4 == 0;
    const bool chain = c->opt_chain_io && out64 && !(c->opt_fuse_last && nb >= 1024) && f32_small_applies(c, m->fc[0], nb, false, false) &&
                       f32_small_applies(c, m->fc[1], nb, false, false) && f32_small_applies(c, m->fc[2], nb, false, false) &&
                       f32_small_chain_out_ok(c, m->fc[0], nb) && f32_small_chain_out_ok(c, m->fc[1], nb);
    if ((rc = run_gemm(c, m->fc[0], d_ctx, P0, nullptr, nb, s, nullptr, nullptr, nullptr, c->host_input, chain ? 2 : 0))) return rc;
    if ((rc = run_gemm(c, m->fc[1], P0, P1, nullptr, nb, s, nullptr, nullptr, nullptr, nullptr, chain ? 3 : 0))) return rc;
    // Output layer of the 4x4 / 8x8 nets (<= 64 outputs) on the exact-f32 path: summed in K segments of 160 hidden units +
    // fuse_reduce at EVERY batch size -- inside the last hidden layer's launch (big batches: its 1200-wide activations never
    // leave the registers of the waves that produce them; the 29 us / 14 %-of-peak launch of rounds 1-3 is gone) or from the
    // stored activations by fc_out_f32_kernel, which repeats the fused kernel's MFMA chain operand for operand.
    const int n_out = m->fc[3].proto.Cout;
    if (n_out <= 64 && n_out % 4 == 0) {
        const int segs = (m->fc[3].proto.Cin + 159) / 160;
        if ((rc = dev_reserve(c, c->ws[3], (size_t)segs * nb * 64 * 4))) return rc;
        float* part = (float*)c->ws[3].p;
        int tiles = 0;
        if (c->opt_fuse_last && nb >= 1024) {
            if ((rc = run_gemm(c, m->fc[2], P1, nullptr, nullptr, nb, s, &m->fc[3], part, &tiles))) return rc;
        } else {
            if ((rc = run_gemm(c, m->fc[2], P1, P0, nullptr, nb, s, nullptr, nullptr, nullptr, nullptr, chain ? 1 : 0))) return rc;
            TapGemmParams q = m->fc[3].proto;
            q.X = P0; q.Wp = m->fc[3].d_w; q.part = part; q.M = (int)nb; q.x_bytes = (unsigned)(4.0 * (double)nb * q.Cin);
            // small passes: the K segments and their reduction in ONE launch (fc_out_f32_small_kernel: the same bits, one launch less
            // in the chain of a single-block call)
            if (c->opt_fc_out_f32 && nb <= 512 && !c->opt_time_launches) {
                TapGemmParams r = q;
                r.part = nullptr; r.bias = m->fc[3].d_bias; r.mean = c->mean; r.Y = d_out; r.Yi = d_dst;
                if (fc_out_f32_small_fits(r)) {
                    if (void* slot = diag_stamp_slot(c, c->opt_fc_out_f32 == 2 ? "fc_out_f32_small (32x32x2)" : "fc_out_f32_chain", (long)((nb + 15) / 16) * ((n_out + 15) / 16), 1200.0)) r.Xlo = slot;
                    const int nwg = (int)((nb + 15) / 16) * ((n_out + 15) / 16);      // fc_out_f32_chain_kernel's workgroups: a completion flag each
                    HIPCHK(c, launch_fc_out_f32_small(r, s, c->opt_fc_out_f32 == 2 ? take_done_signal(c) : take_done_signal_per_wg(c, nwg), c->opt_fc_out_f32 == 2));
                    c->stat_gemm_launches++; c->stat_launches++;
                    c->stat_gemm_flops += 2.0 * (double)nb * m->fc[3].k_total * n_out;
                    return PNN_OK;
                }
            }
            HIPCHK(c, launch_fc_out_f32(q, s, &tiles));
            c->stat_gemm_launches++; c->stat_launches++;
            c->stat_gemm_flops += 2.0 * (double)nb * m->fc[3].k_total * n_out;
        }
        if (tiles != segs) return fail(c, PNN_E_ARG, "output layer: %d K segments, expected %d", tiles, segs);
        HIPCHK(c, launch_fuse_reduce(part, tiles, (int)nb, n_out, m->fc[3].d_bias, 1.f, c->mean, d_out, d_dst, s, take_done_signal(c)));
        c->stat_launches++;
        return PNN_OK;
    }
    if ((rc = run_gemm(c, m->fc[2], P1, P0, nullptr, nb, s))) return rc;
    return run_gemm(c, m->fc[3], P0, d_out, d_dst, nb, s);
}

int conv_pass(pnn_ctx* c, Model* m, const float* d_above, const float* d_left, long nb, float* d_out, int32_t* d_dst,
              hipStream_t s)
{
    float* P[2] = {(float*)c->ws[0].p, (float*)c->ws[1].p};
    float* F[2] = {(float*)c->ws[2].p, (float*)c->ws[3].p};
    // Split-precision mode: tensors between two tap GEMMs travel in the split f16 layout (same byte count as f32);
    // tensors consumed by the merger / the last transposed convolution stay f32.
    const bool sp = pass_uses_split(c, m, nb);
    int rc;
    const bool side_ok = c->side_stream && c->ws[4].bytes >= (size_t)nb * m->pmax * 4 && c->ws[5].bytes >= (size_t)nb * m->pmax * 4;   // ensure_ws sized them for this pass's chunk
    bool par = branches_overlap(c, m, nb) && side_ok;
    const auto shape = std::make_pair((const void*)m, nb);
    const bool batch_overlap = !par && side_ok && branches_overlap_at_batch(c, m, nb);
    if (batch_overlap) {                              // see branches_overlap_at_batch: once a one-stream pass of this shape needed no tuning sweep
        auto it = c->overlap_ready.find(shape);
        par = it != c->overlap_ready.end() && it->second == c->tune_gen;
    }
    const long gen_before = c->tune_gen;
    hipStream_t const main_stream = s;
    // Small passes (single-block calls, the service's handfuls): the two branches are independent chains of launches that
    // cost ~4 us each whatever they do.  Layer i of both branches goes into ONE launch (conv_cin1_pair_kernel, then
    // tapgemm_small_pair_kernel): 13 -> 9 launches for the 16x16 net, no event traffic between streams.  Same kernels' bodies,
    // same arithmetic: bit-identical to the separate launches.
    static const bool env_profile = getenv("PNN_PROFILE") != nullptr, env_sdiag = getenv("PNN_F32S_DIAG") != nullptr;   // (not per pass: getenv walks the environment)
    bool pair = sp && c->opt_pair && c->opt_small && c->opt_sp_cfg < 0 && !c->opt_time_launches && !env_profile &&
                m->branch[0].size() == m->branch[1].size() && !m->branch[0].empty();
    for (size_t i = 0; pair && i < m->branch[0].size(); i++) {
        long tiles = 0;
        for (int br = 0; br < 2; br++) {
            const TapGemmParams& q = m->branch[br][i].proto;
            tiles += ((nb * q.SH * q.SW + 31) / 32) * ((q.Cout + 31) / 32) * q.ncls;
        }
        pair = tiles <= c->opt_small_tiles;
    }
    if (pair) {
        if ((rc = dev_reserve(c, c->ws[4], (size_t)nb * m->pmax * 4))) return rc;
        if ((rc = dev_reserve(c, c->ws[5], (size_t)nb * m->pmax * 4))) return rc;
        float* Q[2][2] = {{P[0], P[1]}, {(float*)c->ws[4].p, (float*)c->ws[5].p}};
        Conv1Params f[2];
        for (int br = 0; br < 2; br++) {
            f[br] = m->first[br].proto;
            f[br].X = br == 0 ? d_above : d_left; f[br].W = m->first[br].d_w; f[br].bias = m->first[br].d_bias;
            f[br].Wsp = m->first[br].d_w_sp; f[br].out_scale = m->first[br].sp_inv_scale; f[br].npad = m->first[br].npad;
            f[br].B = (int)nb; f[br].range_flag = c->h_range; f[br].Y = Q[br][0]; f[br].split = 1;
        }
        HIPCHK(c, launch_conv_cin1_pair(f[0], f[1], s));
        c->stat_launches++;
        const size_t nl = m->branch[0].size();
        int cur = 0;
        for (size_t i = 0; i < nl; i++) {
            const bool last = i + 1 == nl;
            TapGemmParams q[2];
            for (int br = 0; br < 2; br++) {
                const GemmLayer& L = m->branch[br][i];
                q[br] = L.proto;
                q[br].X = Q[br][cur]; q[br].Wp = L.d_w_sp; q[br].bias = L.d_bias; q[br].mean = c->mean; q[br].out_scale = L.sp_inv_scale;
                q[br].range_flag = c->h_range; q[br].zero = c->d_zero;
                if (last) q[br].Y = F[br]; else q[br].Yhi = Q[br][cur ^ 1];
                q[br].M = (int)(nb * q[br].SH * q[br].SW);
                q[br].x_bytes = (unsigned)(4.0 * (double)nb * q[br].IH * q[br].IW * q[br].Cin);
                c->stat_gemm_flops += 2.0 * (double)q[br].M * L.k_total * q[br].Cout;
            }
            static const bool dbg = getenv("PNN_DEBUG") != nullptr;
            if (dbg) fprintf(stderr, "[pnn] sp-gemm pair: branch layer %zu, M = %d / %d -> one small-kernel launch\n", i + 1, q[0].M, q[1].M);
            HIPCHK(c, launch_tapgemm_small_pair(q[0], q[1], s));
            c->stat_gemm_launches++; c->stat_launches++;
            cur ^= 1;
        }
    }
    // The same on the exact-f32 arithmetic (round 5): conv_cin1_pair_kernel with f32 output (the context gather inside when the pass
    // reads the picture plane itself), then tapgemm_f32_small_pair_kernel per layer; K-segmented layers (32x32 / 64x64 nets) leave
    // their planes and get one seg_reduce launch per branch.
    bool merger_done = false, merger_chain_done = false;   // the merger ran as the tail of the branches' last pair launch (below)
    int merged_in = 0;                                // ... and left the merged map in P[merged_in]
    bool pair32 = !sp && !pair && c->opt_pair && c->opt_f32_small && c->opt_f32_cfg < 0 &&
                  !c->opt_time_launches && !env_profile && !env_sdiag && m->branch[0].size() == m->branch[1].size() && !m->branch[0].empty();
    for (size_t i = 0; pair32 && i < m->branch[0].size(); i++) {
        long tiles = 0;
        for (int br = 0; br < 2; br++) {
            TapGemmParams q = m->branch[br][i].proto;
            q.M = (int)(nb * q.SH * q.SW); q.nseg = m->branch[br][i].nseg;
            tiles += tapgemm_f32_small_tiles(q);
        }
        pair32 = tiles <= c->opt_f32_small_tiles;
    }
    if (pair32) {
        if ((rc = dev_reserve(c, c->ws[4], (size_t)nb * m->pmax * 4))) return rc;
        if ((rc = dev_reserve(c, c->ws[5], (size_t)nb * m->pmax * 4))) return rc;
        float* Q[2][2] = {{P[0], P[1]}, {(float*)c->ws[4].p, (float*)c->ws[5].p}};
        Conv1Params f[2];
        // the first convolutions' maps in chain order too: their consumer is the pair launch of layer 1 (both branches or neither)
        const bool first_chain = c->opt_chain_io && m->first[0].proto.Cout % 16 == 0 && m->first[1].proto.Cout % 16 == 0;
        for (int br = 0; br < 2; br++) {
            f[br] = m->first[br].proto;
            f[br].X = br == 0 ? d_above : d_left; f[br].W = m->first[br].d_w; f[br].bias = m->first[br].d_bias;
            f[br].Wsp = m->first[br].d_w_sp; f[br].out_scale = m->first[br].sp_inv_scale; f[br].npad = m->first[br].npad;
            f[br].B = (int)nb; f[br].range_flag = c->h_range; f[br].Y = Q[br][0]; f[br].split = 0;
            f[br].chain = first_chain ? 1 : 0;
            if (!f[br].X && c->lazy.plane) {
                f[br].plane = c->lazy.plane; f[br].tbs = reinterpret_cast<const TbDev*>(c->lazy.tbs); f[br].pel_bytes = c->lazy.pel_bytes; f[br].unit = c->lazy.unit;
                f[br].w = m->width; f[br].branch = br; f[br].mean = c->mean;
            }
        }
        HIPCHK(c, launch_conv_cin1_pair(f[0], f[1], s));
        c->stat_launches++;
        const size_t nl = m->branch[0].size();
        int cur = 0;
        bool in_chain = first_chain;                  // the tensors layer i reads are in chain order (written so by layer i - 1 of both branches)
        for (size_t i = 0; i < nl; i++) {
            const bool last = i + 1 == nl;
            TapGemmParams q[2];
            float* dst[2];
            size_t out_floats[2];
            bool folded[2] = {false, false};
            // ... and layer i writes that way when layer i + 1 is another launch of this kernel (the last branch layer feeds the merger: channel order)
            const bool out_chain = !last && f32_small_chain_out_ok(c, m->branch[0][i], nb) && f32_small_chain_out_ok(c, m->branch[1][i], nb);
            for (int br = 0; br < 2; br++) {
                const GemmLayer& L = m->branch[br][i];
                q[br] = L.proto;
                q[br].X = Q[br][cur]; q[br].Wp = L.d_w_ch; q[br].bias = L.d_bias; q[br].mean = c->mean;
                dst[br] = last ? F[br] : Q[br][cur ^ 1];
                q[br].Y = dst[br];
                q[br].M = (int)(nb * q[br].SH * q[br].SW);
                q[br].x_bytes = (unsigned)(4.0 * (double)nb * q[br].IH * q[br].IW * q[br].Cin);
                out_floats[br] = (size_t)nb * (size_t)L.out_per_block;
                if (L.nseg > 1) {                     // raw sums into the branch's planes; seg_reduce below applies bias and activation
                    DevBuf& sb = c->seg_part[br];
                    if ((rc = dev_reserve(c, sb, (size_t)L.nseg * out_floats[br] * 4))) return rc;
                    if (out_floats[br] >= 0xffffffffull) return fail(c, PNN_E_ARG, "batch too large for one pass");
                    q[br].Y = (float*)sb.p; q[br].bias = (const float*)c->d_zero; q[br].act = 0; q[br].nseg = L.nseg; q[br].seg_stride = (unsigned)out_floats[br];
                    folded[br] = c->opt_seg_fold && c->d_seg_cnt && tapgemm_f32_small_tiles(q[br]) / L.nseg <= pnn_ctx::kSegCntTiles;
                    if (folded[br]) {                 // ... or the launch adds the planes up itself (tapgemm_f32_small_body; tile-major planes)
                        if ((rc = dev_reserve(c, sb, (size_t)tapgemm_f32_small_tiles(q[br]) * 1024))) return rc;
                        q[br].Y = (float*)sb.p;
                        q[br].seg_cnt = c->d_seg_cnt + br * pnn_ctx::kSegCntTiles;
                        q[br].seg_Y = dst[br]; q[br].bias = L.d_bias; q[br].act = L.proto.act;
                    }
                }
                q[br].chain_io = (in_chain ? 1 : 0) | (out_chain ? 2 : 0);
                c->stat_gemm_flops += 2.0 * (double)q[br].M * L.k_total * q[br].Cout;
                if (void* slot = diag_stamp_slot(c, br ? "f32_small pair, left branch" : "f32_small pair, above branch", tapgemm_f32_small_tiles(q[br]), L.k_total)) q[br].Xlo = slot;
            }
            static const bool dbg = getenv("PNN_DEBUG") != nullptr;
            if (dbg) fprintf(stderr, "[pnn] f32 gemm pair: branch layer %zu, M = %d / %d -> one f32 small-kernel launch\n", i + 1, q[0].M, q[1].M);
            // the merger as the tail of the branches' LAST pair launch (SmallTail kind 1): per (block, channel group), by the last of its
            // five tiles to arrive -- where the launch is at most one workgroup per CU (the tail kernel's registers allow one)
            bool mtail = false;
            SmallTail mt;
            if (last && c->opt_tails && c->d_seg_cnt && nb * (m->C / 16) <= pnn_ctx::kTailCnt &&
                tapgemm_f32_small_tiles(q[0]) + tapgemm_f32_small_tiles(q[1]) <= (long)device_info().cus) {
                mt.kind = 1; mt.cnt = c->d_seg_cnt + 2 * pnn_ctx::kSegCntTiles;
                mt.m = m->merger.proto;
                mt.m.A = F[0]; mt.m.L = F[1]; mt.m.Wp = m->merger.d_w; mt.m.bias = m->merger.d_bias; mt.m.Y = P[cur ^ 1]; mt.m.B = (int)nb;   // (P[cur] is what this launch's above branch still reads)
                mt.m.split = 0; mt.m.range_flag = c->h_range;
                mt.m.chain = (m->tconv.size() > 0 && c->opt_chain_io && mt.m.C % 16 == 0 && f32_small_applies(c, m->tconv[0], nb, false, false)) ? 1 : 0;
                mt.t = TConv1Params{};
                mtail = f32_small_merger_tail_ok(q[0], q[1], mt.m);
            }
            if (mtail) {
                if (c->seg_cnt_dirty) { HIPCHK(c, hipMemsetAsync(c->d_seg_cnt, 0, pnn_ctx::kCntWords * 4, s)); c->seg_cnt_dirty = false; }
                if (dbg) fprintf(stderr, "[pnn]   ... with the merger as its tail\n");
                HIPCHK(c, launch_tapgemm_f32_small_pair_tail(q[0], q[1], mt, s, (int)c->opt_f32_small_deep));
                merger_done = true; merger_chain_done = mt.m.chain != 0; merged_in = cur ^ 1;
            } else {
                HIPCHK(c, launch_tapgemm_f32_small_pair(q[0], q[1], s, (int)c->opt_f32_small_deep));
            }
            c->stat_gemm_launches++; c->stat_launches++;
            for (int br = 0; br < 2; br++) {
                const GemmLayer& L = m->branch[br][i];
                if (L.nseg <= 1 || folded[br]) continue;
                HIPCHK(c, launch_seg_reduce(q[br].Y, L.nseg, out_floats[br], q[br].Cout, L.d_bias, L.proto.act, dst[br], s));
                c->stat_launches++;
            }
            in_chain = out_chain;
            cur ^= 1;
        }
        pair = true;                                  // the branches are done
    }
    float* PB[2][2] = {{P[0], P[1]}, {P[0], P[1]}};
    if (par) { PB[1][0] = (float*)c->ws[4].p; PB[1][1] = (float*)c->ws[5].p; }
    int bcur[2] = {0, 0};
    auto branch_layers = [&](int br, hipStream_t st) -> int {
        const size_t nl = m->branch[br].size();
        Conv1Params f = m->first[br].proto;
        f.X = br == 0 ? d_above : d_left; f.W = m->first[br].d_w; f.bias = m->first[br].d_bias;
        f.Wsp = m->first[br].d_w_sp; f.out_scale = m->first[br].sp_inv_scale; f.npad = m->first[br].npad;
        f.B = (int)nb; f.range_flag = c->h_range;
        f.Y = nl == 0 ? F[br] : PB[br][0];
        f.split = (sp && nl > 0) ? 1 : 0;
        if (!f.X && !sp && c->lazy.plane) {           // exact-f32 pass straight from the picture plane (pnn_predict_tbs_device decided so)
            f.plane = c->lazy.plane; f.tbs = reinterpret_cast<const TbDev*>(c->lazy.tbs); f.pel_bytes = c->lazy.pel_bytes; f.unit = c->lazy.unit;
            f.w = m->width; f.branch = br; f.mean = c->mean;
        }
        const bool delegate = sp && nl > 0;           // run_gemm_sp of the next layer launches or absorbs this convolution
        if (!delegate) {
            HIPCHK(c, launch_conv_cin1(f, st));
            c->stat_launches++;
        }
        for (size_t i = 0; i < nl; i++) {
            const bool last = i + 1 == nl;
            int& cur = bcur[br];
            float* dst = last ? F[br] : PB[br][cur ^ 1];
            int rc2;
            if (sp) rc2 = run_gemm_sp(c, m->branch[br][i], PB[br][cur], nullptr, last ? dst : nullptr, last ? nullptr : dst, nullptr, nullptr, nb, st, nullptr, nullptr,
                                      nullptr, (i == 0 && delegate) ? &f : nullptr);
            else rc2 = run_gemm(c, m->branch[br][i], PB[br][cur], dst, nullptr, nb, st);
            if (rc2) return rc2;
            cur ^= 1;
        }
        return PNN_OK;
    };
    if (!pair) {
        if (par) {
            HIPCHK(c, hipEventRecord(c->ev_fork, main_stream));
            HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        }
        for (int br = 0; br < 2; br++)
            if ((rc = branch_layers(br, par && br == 1 ? c->side_stream : main_stream))) return rc;
        if (par) {
            HIPCHK(c, hipEventRecord(c->ev_join, c->side_stream));
            HIPCHK(c, hipStreamWaitEvent(main_stream, c->ev_join, 0));
        }
        s = main_stream;
    }
    const size_t nt = m->tconv.size();
    MergerParams mp = m->merger.proto;
    mp.A = F[0]; mp.L = F[1]; mp.Wp = m->merger.d_w; mp.bias = m->merger.d_bias; mp.Y = P[0]; mp.B = (int)nb;
    mp.split = (sp && nt > 0) ? 1 : 0;
    mp.range_flag = c->h_range;
    // exact f32, small passes: the merged map in chain order when the first transposed convolution runs on the small kernels
    const bool merger_chain = !sp && nt > 0 && c->opt_chain_io && mp.C % 16 == 0 && f32_small_applies(c, m->tconv[0], nb, false, false);
    mp.chain = merger_chain ? 1 : 0;
    if (merger_done && merger_chain_done != merger_chain) return fail(c, PNN_E_ARG, "internal: the merger's tail and its consumer disagree on the tensor's order");
    if (!merger_done) {
        HIPCHK(c, launch_merger(mp, s));
        c->stat_launches++;
    }
    int cur = merger_done ? merged_in : 0;
    TConv1Params tp = m->last.proto;
    tp.W = m->last.d_w; tp.Y = d_out; tp.Yi = d_dst; tp.B = (int)nb; tp.mean = c->mean;
    // the last layer as the tail of the last GEMM of the transposed stack (SmallTail kind 2; run_gemm takes it where that GEMM runs on
    // the small kernels and the shapes fit): one completion flag per block instead of the counter form
    const bool want_ctail = !sp && nt > 0 && c->opt_tails && c->d_seg_cnt && nb <= pnn_ctx::kTailCnt && nb <= 64 && !c->opt_time_launches && !env_profile && !env_sdiag &&
                            f32_small_applies(c, m->tconv[nt - 1], nb, false, false);
    SmallTail ct;
    if (want_ctail) {
        if (c->seg_cnt_dirty) { HIPCHK(c, hipMemsetAsync(c->d_seg_cnt, 0, pnn_ctx::kCntWords * 4, s)); c->seg_cnt_dirty = false; }
        TapGemmParams probe = m->tconv[nt - 1].proto;
        probe.M = (int)(nb * probe.SH * probe.SW); probe.Y = P[0]; probe.Yi = nullptr; probe.nseg = m->tconv[nt - 1].nseg; probe.chain_io = 0;
        TConv1Params tq = tp;
        tq.X = P[0];
        if (!f32_small_cout1_tail_ok(probe, tq)) { tp.done = take_done_signal(c); ct.kind = 0; }
        else { tp.done = take_done_signal_per_wg(c, (int)nb); ct.kind = 2; }
    } else {
        tp.done = take_done_signal(c);
        ct.kind = 0;
    }
    bool last_done = false;                          // the last layer went out with (or inside) the GEMM in front of it
    bool t_in_chain = merger_chain;                  // exact f32, small passes: the transposed convolutions hand their maps on in chain order (see fc_pass)
    for (size_t i = 0; i < nt; i++) {
        const bool last = i + 1 == nt;
        const bool t_out_chain = !sp && !last && f32_small_applies(c, m->tconv[i], nb, false, false) && f32_small_applies(c, m->tconv[i + 1], nb, false, false) &&
                                 f32_small_chain_out_ok(c, m->tconv[i], nb);
        if (sp && last) {                            // run_gemm_sp launches the net's last layer too, or has the image kernel apply it
            tp.X = P[cur ^ 1];
            rc = run_gemm_sp(c, m->tconv[i], P[cur], nullptr, P[cur ^ 1], nullptr, nullptr, nullptr, nb, s, nullptr, nullptr, nullptr, nullptr, false, 0, nullptr, &tp);
            last_done = true;
        } else if (sp) rc = run_gemm_sp(c, m->tconv[i], P[cur], nullptr, nullptr, P[cur ^ 1], nullptr, nullptr, nb, s);
        else if (last && ct.kind == 2) {
            ct.cnt = c->d_seg_cnt + 2 * pnn_ctx::kSegCntTiles;
            ct.m = MergerParams{};
            ct.t = tp;
            ct.t.X = P[cur ^ 1];
            rc = run_gemm(c, m->tconv[i], P[cur], P[cur ^ 1], nullptr, nb, s, nullptr, nullptr, nullptr, nullptr, (t_in_chain ? 1 : 0), &ct, &last_done);
            if (!rc && !last_done) return fail(c, PNN_E_ARG, "internal: the last layer's tail was planned and not taken");
        }
        else rc = run_gemm(c, m->tconv[i], P[cur], P[cur ^ 1], nullptr, nb, s, nullptr, nullptr, nullptr, nullptr, (t_in_chain ? 1 : 0) | (t_out_chain ? 2 : 0));
        if (rc) return rc;
        t_in_chain = t_out_chain;
        cur ^= 1;
    }
    if (!last_done) {
        tp.X = P[cur];
        HIPCHK(c, launch_tconv_cout1(tp, s));
        c->stat_launches++;
    }
    // a pass of this shape went through on one stream without a tuning sweep: the next one may overlap its branches
    if (batch_overlap && !par && !pair && c->tune_gen == gen_before) c->overlap_ready[shape] = c->tune_gen;
    return PNN_OK;
}

}  // namespace

// Would a pass of nb blocks through convolutional model m fuse BOTH branches' first convolutions into the image kernel (so that
// the context gather can be fused in too)?  Launches nothing.  Exact-f32 passes: conv_cin1_kernel reads the plane itself.
bool conv_pass_fuses_first(pnn_ctx* c, Model* m, long nb)
{
    if (!m->is_fc && c->opt_fuse_gather && !pass_uses_split(c, m, nb)) return true;
    if (m->is_fc || !c->opt_fuse_gather || !pass_uses_split(c, m, nb) || c->opt_time_launches || getenv("PNN_PROFILE")) return false;
    for (int br = 0; br < 2; br++) {
        if (m->branch[br].empty()) return false;
        Conv1Params f = m->first[br].proto;
        f.W = m->first[br].d_w; f.bias = m->first[br].d_bias;
        f.Wsp = m->first[br].d_w_sp; f.out_scale = m->first[br].sp_inv_scale; f.npad = m->first[br].npad;
        f.B = (int)nb; f.split = 1;
        bool fuse = false;
        if (run_gemm_sp(c, m->branch[br][0], nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nb, nullptr, nullptr, nullptr, nullptr, &f, false, 0, &fuse) || !fuse) return false;
    }
    return true;
}

// Runs the net over n blocks in chunks. Inputs per block: FC one [5w^2] row; conv above/left portions.
int run_net(pnn_ctx* c, Model* m, const float* d_a, long pitch_a, const float* d_l, long pitch_l, long n, float* d_out,
            int32_t* d_dst, hipStream_t s, bool ctx_is_split)
{
    const int w = m->width;
    const long chunk = std::min(n, chunk_blocks(c, m));
    int rc = ensure_ws(c, m, chunk);
    if (rc) return rc;
    // host_predict's copy of the caller's f32 rows (small FC inputs ride in the first kernel's argument block) follows the
    // chunks like the device pointers do: with max_chunk below the batch size every chunk must carry ITS rows
    const float* const host_rows = c->host_input;
    for (long b0 = 0; b0 < n; b0 += chunk) {
        const long nb = std::min(chunk, n - b0);
        float* o = d_out ? d_out + b0 * w * w : nullptr;
        int32_t* di = d_dst ? d_dst + b0 * w * w : nullptr;
        c->host_input = host_rows ? host_rows + b0 * pitch_a : nullptr;
        c->done_last_chunk = b0 + nb >= n;            // only the call's last kernel may raise the completion flag
        rc = m->is_fc ? fc_pass(c, m, d_a + b0 * pitch_a, ctx_is_split, nb, o, di, s)
                      : conv_pass(c, m, d_a + b0 * pitch_a, d_l + b0 * pitch_l, nb, o, di, s);
        if (rc) break;
    }
    c->host_input = host_rows;
    c->done_last_chunk = true;
    return rc;
}

}  // namespace pnn
