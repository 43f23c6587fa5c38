// Diagnostic (not part of the product): times tapgemm_ring_kernel on a synthetic FC-shaped problem and prints the
// per-phase cycle sums of wave 0 (PNN_RING_DIAG stamps).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPNN_RING_DIAG -Icontext_adaptive_neural_network_based_prediction_amd/csrc tools/ring_prof.hip -o tools/_bin/ring_prof
//   ./tools/_bin/ring_prof [M] [K] [N]
#include "pnn_gemm_ring.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace pnn;
namespace pnn { thread_local const LaunchEvents* g_launch_events = nullptr; }

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 4096, K = argc > 2 ? atoi(argv[2]) : 1200, N = argc > 3 ? atoi(argv[3]) : 1200;
    const int nchunk = ((K / 16 + kChunkPad - 1) / kChunkPad) * kChunkPad, Npad = ((N + 15) / 16) * 16 + 160;
    const size_t xb = (size_t)M * K * 4, wb = (size_t)nchunk * 4 * Npad * 16;
    std::vector<_Float16> hx(xb / 2), hw(wb / 2);
    srand(1);
    for (auto& v : hx) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    for (auto& v : hw) v = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    void *dx, *dw, *dy, *dz, *dd; float* db;
    hipMalloc(&dx, xb); hipMalloc(&dw, wb); hipMalloc(&dy, (size_t)M * N * 4); hipMalloc(&dz, 4096); hipMalloc(&dd, 1 << 24);
    hipMalloc(&db, Npad * 4);
    hipMemcpy(dx, hx.data(), xb, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), wb, hipMemcpyHostToDevice);
    hipMemset(dz, 0, 4096); hipMemset(db, 0, Npad * 4);
    TapGemmParams p{};
    p.X = (const float*)dx; p.Xlo = dd; p.zero = dz; p.Wp = (const float*)dw; p.bias = db; p.Yhi = dy; p.out_scale = 1.f;
    p.M = M; p.SH = p.SW = 1; p.IH = p.IW = 1; p.Cin = K; p.a = 1; p.OH = p.OW = 1; p.Cout = N; p.os = 1; p.Npad = Npad; p.act = 1; p.x_bytes = (unsigned)xb;
    p.ncls = 1; p.tap_begin[0] = 0; p.tap_begin[1] = 1; p.chunk_begin[0] = 0; p.tap[0] = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < tapgemm_ring_num_cfgs(); i++) {
        const TileCfg t = tapgemm_ring_cfg(i);
        const int bm = 32 * t.rt * t.wm, bn = 32 * t.nt * (4 / t.wm);
        const int nwg = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
        if (launch_tapgemm_ring(p, i, 0) != hipSuccess) { printf("cfg %d: launch failed\n", i); continue; }
        hipEventRecord(e0);
        for (int r = 0; r < 20; r++) launch_tapgemm_ring(p, i, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(4 * (size_t)nwg);
        hipMemcpy(h.data(), dd, h.size() * 8, hipMemcpyDeviceToHost);
        double s[4] = {0, 0, 0, 0};
        for (int w = 0; w < nwg; w++) for (int k = 0; k < 4; k++) s[k] += (double)h[4 * w + k];
        const int nst = (K / 16 + t.kc - 1) / t.kc;
        const double us = ms * 1e3 / 20;
#ifdef PNN_RING_DIAG2
        printf("ring{%d,%d,%d,wm%d,d%d} %3dx%3d  %4d WGs  %7.1f us | wave 0 avg: prologue %6.0f cyc  loop %6.0f cyc (%5.0f/stage, MFMA work %d)  epilogue %6.0f cyc  workgroup lifetime %5.1f us\n",
               t.rt, t.nt, t.kc, t.wm, t.d, bm, bn, nwg, us, s[0] / nwg, s[1] / nwg, s[1] / nwg / nst, t.rt * t.nt * 3 * t.kc * 32, s[2] / nwg, s[3] / nwg / 100.0);
        {
            std::vector<unsigned long long> he(8 * (size_t)nwg);
            hipMemcpy(he.data(), (char*)dd + 8 * (1 << 18), he.size() * 8, hipMemcpyDeviceToHost);
            double e[6] = {0, 0, 0, 0, 0, 0};
            for (int w = 0; w < nwg; w++) for (int k = 0; k < 6; k++) e[k] += (double)he[8 * w + k];
#ifdef PNN_RING_DIAG3
            {
                std::vector<unsigned long long> hl(8 * (size_t)nwg);
                hipMemcpy(hl.data(), (char*)dd + 8 * (1 << 19), hl.size() * 8, hipMemcpyDeviceToHost);
                double l[7] = {0, 0, 0, 0, 0, 0, 0};
                for (int w = 0; w < nwg; w++) { l[0] += (double)((long long)hl[8 * w] - (long long)he[8 * w + 6]); for (int k = 1; k < 7; k++) l[k] += (double)hl[8 * w + k]; }
                printf("      loader wave 4 per stage: waiting for its loads %5.0f | at the barrier %5.0f | issuing the next stage %5.0f\n", l[4] / nwg / nst, l[5] / nwg / nst, l[6] / nwg / nst);
                printf("      loader wave 4: starts %5.0f cycles after MFMA wave 0 | setup %5.0f | issue of D-1 stages %5.0f | wait for stage 0 %5.0f\n", l[0] / nwg, l[1] / nwg,
                       l[2] / nwg, l[3] / nwg);
            }
            {
                unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0, lmin = ~0ull, lmax = 0;
                for (int w = 0; w < nwg; w++) {
                    const unsigned long long st = he[8 * w + 7], life = h[4 * w + 3], en = st + life;
                    s0 = st < s0 ? st : s0; s1 = st > s1 ? st : s1; e0 = en < e0 ? en : e0; e1 = en > e1 ? en : e1;
                    lmin = life < lmin ? life : lmin; lmax = life > lmax ? life : lmax;
                }
                {   // where the slow workgroups are: mean lifetime by XCD (linear workgroup id % 8) and by column tile
                    double bx[8] = {0}, nx[8] = {0};
                    const int gx = (M + bm - 1) / bm, gy = (N + bn - 1) / bn;
                    std::vector<double> by(gy, 0.0);
                    for (int w = 0; w < nwg; w++) { bx[w % 8] += h[4 * w + 3] / 100.0; nx[w % 8] += 1; by[(w / gx) % gy] += h[4 * w + 3] / 100.0 / gx; }
                    printf("      mean lifetime by XCD:");
                    for (int x = 0; x < 8; x++) printf(" %.1f", bx[x] / (nx[x] > 0 ? nx[x] : 1));
                    printf(" us; by column tile:");
                    for (int y = 0; y < gy; y++) printf(" %.1f", by[y]);
                    printf(" us\n");
                }
                printf("      workgroup starts spread over %.1f us, ends over %.1f us, first start -> last end %.1f us, lifetime min %.1f / max %.1f us\n",
                       (s1 - s0) / 100.0, (e1 - e0) / 100.0, (e1 - s0) / 100.0, lmin / 100.0, lmax / 100.0);
            }
            printf("      stage barrier wait (MFMA wave 0): %5.0f cycles per stage\n", e[5] / nwg / nst);
#endif
#ifdef PNN_RING_DIAG4
            {
                std::vector<unsigned long long> h4(2 * (size_t)nwg);
                hipMemcpy(h4.data(), (char*)dd + 8 * (1 << 20), h4.size() * 8, hipMemcpyDeviceToHost);
                double a = 0, b = 0;
                for (int w = 0; w < nwg; w++) { a += (double)h4[2 * w]; b += (double)h4[2 * w + 1]; }
                printf("      group loop: first pass %5.0f cycles, second pass (warm instruction cache) %5.0f\n", a / nwg, b / nwg);
            }
#endif
            printf("      epilogue: barrier A %5.0f | scale/bias/split -> LDS %5.0f | barrier B %5.0f | copy-out issue %5.0f | store drain %5.0f\n", e[0] / nwg, e[1] / nwg,
                   e[2] / nwg, e[3] / nwg, e[4] / nwg);
        }
        continue;
#endif
        printf("ring{%d,%d,%d,wm%d,d%d} %3dx%3d  %4d WGs  %7.1f us  %6.1f TF-eq | per stage, wave 0: mfma-a %5.0f  wait+barrier %5.0f  frag+issue %5.0f  mfma-b %5.0f  (MFMA work %d cyc)\n",
               t.rt, t.nt, t.kc, t.wm, t.d, bm, bn, nwg, us, 2.0 * M * K * N / us / 1e6, s[0] / nwg / nst, s[1] / nwg / nst, s[2] / nwg / nst,
               s[3] / nwg / nst, t.rt * t.nt * 3 * t.kc * 32);
    }
    return 0;
}
