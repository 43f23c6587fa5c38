// Model loading: the flat parameter vector of one PNN (canonical order, weights.py / SURVEY.md Appendix B.7) -> per-layer
// launch prototypes with weights pre-packed in the order the MFMA lane groups consume them.
// Reference behaviour reproduced (not code): the graphs of pnn/components.py:10-261 with the stride tuples of
// pnn/PredictionNeuralNetwork.py:126-132.
#include "pnn_ctx.h"

#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace pnn {
namespace {

int upload(pnn_ctx* c, Model* m, const float* host, size_t n, float** out)
{
    void* d = nullptr;
    std::lock_guard<std::recursive_mutex> guard(unsafe_calls_lock());   // allocation + synchronous copy: never beside another thread's stream capture (pnn_abi.cpp)
    if (hipMalloc(&d, std::max(n, (size_t)4) * sizeof(float)) != hipSuccess)
        return fail(c, PNN_E_NOMEM, "hipMalloc of %zu weight floats failed", n);
    m->allocs.push_back(d);
    HIPCHK(c, hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
    *out = (float*)d;
    return PNN_OK;
}

// Weight packing runs once per model load, but a process of the reference's kind (one HM encoder or decoder) loads five
// models at start-up: 27 M parameters, each written into two strided layouts.  The chunks are independent: a few threads.
template <typename F>
void parallel_chunks(long nchunks, F fn)
{
    const long work = nchunks;
    int nt = (int)std::min<long>(8, std::max<long>(1, work / 64));
    nt = std::min<int>(nt, (int)std::max(1u, std::thread::hardware_concurrency()));
    if (nt <= 1) { fn(0, nchunks); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([=] { fn(nchunks * t / nt, nchunks * (t + 1) / nt); });
    for (auto& x : th) x.join();
}

// [K][N] row-major -> [K/16][4][Npad][4] (k = 16*chunk + 4*q + e), zero-padded columns.
std::vector<float> pack_kn(const std::vector<float>& kn, long K, int N, int npad)
{
    std::vector<float> out((size_t)K * npad, 0.f);
    parallel_chunks(K / 16, [&](long c0, long c1) {
        for (long k = 16 * c0; k < 16 * c1; k++) {
            const long ch = k >> 4; const int q = (k >> 2) & 3, e = k & 3;
            float* dst = out.data() + (((size_t)ch * 4 + q) * npad) * 4 + e;
            const float* src = kn.data() + (size_t)k * N;
            for (int n = 0; n < N; n++) dst[(size_t)n * 4] = src[n];
        }
    });
    return out;
}

// The same weights in the order tapgemm_f32_small_kernel's lane groups consume them (pnn_gemm_f32_small.hip): [K/16][q = 4][Npad][4],
// element i of lane group q = k = 16 chunk + 8 (q & 1) + 2 i + (q >> 1) -- the k that instruction i of the chunk's four
// v_mfma_f32_16x16x4_f32 takes from lane group q when the canonical chain 0, 8, 1, 9, ..., 7, 15 is issued through that instruction.
std::vector<float> pack_kn_chain(const std::vector<float>& kn, long K, int N, int npad)
{
    std::vector<float> out((size_t)K * npad, 0.f);
    parallel_chunks(K / 16, [&](long c0, long c1) {
        for (long ch = c0; ch < c1; ch++)
            for (int q = 0; q < 4; q++)
                for (int i = 0; i < 4; i++) {
                    const long k = 16 * ch + 8 * (q & 1) + 2 * i + (q >> 1);
                    float* dst = out.data() + (((size_t)ch * 4 + q) * npad) * 4 + i;
                    const float* src = kn.data() + (size_t)k * N;
                    for (int n = 0; n < N; n++) dst[(size_t)n * 4] = src[n];
                }
    });
    return out;
}

// Split-precision pack: [K/16][hl = hi/lo][h = k-half][Npad][8 x f16] with w * scale = hi + lo.
std::vector<float> pack_kn_split(const std::vector<float>& kn, long K, int N, int npad, float scale)
{
    std::vector<float> out((size_t)K * npad, 0.f);              // same byte count as the f32 pack
    _Float16* o = reinterpret_cast<_Float16*>(out.data());
    parallel_chunks(K / 16, [&](long c0, long c1) {
        for (long k = 16 * c0; k < 16 * c1; k++) {
            const long ch = k >> 4; const int h = (k >> 3) & 1, j = k & 7;
            const float* src = kn.data() + (size_t)k * N;
            for (int n = 0; n < N; n++) {
                const float w = src[n] * scale;
                const _Float16 hi = (_Float16)w;
                const _Float16 lo = (_Float16)(w - (float)hi);
                o[((((size_t)ch * 2 + 0) * 2 + h) * npad + n) * 8 + j] = hi;
                o[((((size_t)ch * 2 + 1) * 2 + h) * npad + n) * 8 + j] = lo;
            }
        }
    });
    return out;
}

int npad_for(int cout) { return ((cout + 15) / 16) * 16 + 160; }   // slack >= the widest column tile (BN = 160)

// Common tail of the three layer builders. `kn` holds the [K][Cout] rows ordered (class, tap, ci) and
// p.tap_begin / p.Cin / p.ncls are set. Every class is zero-padded to a multiple of kChunkPad 16-deep
// chunks (so that any pipeline stage depth KC <= kChunkPad reads whole stages), packed and uploaded.
int finish_gemm_layer(pnn_ctx* c, Model* m, const std::vector<float>& kn, const float* b, int Cout, GemmLayer* L)
{
    TapGemmParams& p = L->proto;
    const int cpt = p.Cin / 16;
    std::vector<float> padded;
    long chunk = 0;
    double k_real = 0;
    for (int cls = 0; cls < p.ncls; cls++) {
        p.chunk_begin[cls] = (int)chunk;
        const long rows = (long)(p.tap_begin[cls + 1] - p.tap_begin[cls]) * p.Cin;
        const long nch = rows / 16, nch_pad = ((nch + kChunkPad - 1) / kChunkPad) * kChunkPad;
        const float* src = kn.data() + (size_t)p.tap_begin[cls] * p.Cin * Cout;
        padded.insert(padded.end(), src, src + (size_t)rows * Cout);
        padded.resize(padded.size() + (size_t)(nch_pad - nch) * 16 * Cout, 0.f);
        chunk += nch_pad;
        k_real += (double)rows;
    }
    p.chunk_begin[p.ncls] = (int)chunk;
    (void)cpt;
    const int npad = npad_for(Cout);
    std::vector<float> packed = pack_kn(padded, chunk * 16, Cout, npad);
    int rc = upload(c, m, packed.data(), packed.size(), &L->d_w);
    if (rc) return rc;
    packed = pack_kn_chain(padded, chunk * 16, Cout, npad);
    if ((rc = upload(c, m, packed.data(), packed.size(), &L->d_w_ch))) return rc;
    {   // split-precision copy: scale so that max |w| lands in [2^12, 2^13) (hi and lo halves both f16-normal)
        float wmax = 0.f;
        for (float v : padded) wmax = std::max(wmax, std::fabs(v));
        int shift = 0;
        if (wmax > 0.f) { int e; std::frexp(wmax, &e); shift = 13 - e; }
        shift = std::max(-8, std::min(shift, 24));
        const float scale = std::ldexp(1.f, shift);
        L->sp_inv_scale = std::ldexp(1.f, -shift);
        std::vector<float> sp = pack_kn_split(padded, chunk * 16, Cout, npad, scale);
        rc = upload(c, m, sp.data(), sp.size(), &L->d_w_sp);
        if (rc) return rc;
    }
    std::vector<float> bias(((Cout + 3) / 4) * 4 + 4, 0.f);
    std::copy(b, b + Cout, bias.begin());
    rc = upload(c, m, bias.data(), bias.size(), &L->d_bias);
    if (rc) return rc;
    p.Cout = Cout; p.Npad = npad;
    L->k_total = k_real;
    // K segments of the exact-f32 order (GemmLayer::nseg): a class deeper than kSegMinDepth is summed in segments of at most
    // kSegDepth (a 1600-deep chain of v_mfma_f32_32x32x2_f32 is 51 k cycles = 21 us; the 6400-deep third layer of the 64x64 net
    // as ONE chain left a quarter of the chip idle at batch 64 and took 171 us at batch 1), whole taps, never more segments than
    // the shallowest class has taps.
    {
        // The segment layout DEFINES the exact-f32 summation order -- the bits an encoder and its decoder must share -- so in the shipped
        // library it is the pair of constants (reported by pnn_arithmetic_tag); only the diagnostic build (make diag) reads the A/B variables.
#ifdef PNN_F32_DIAG
        static const int seg_depth = getenv("PNN_F32_SEG_DEPTH") ? atoi(getenv("PNN_F32_SEG_DEPTH")) : kSegDepth;
        static const int seg_min = getenv("PNN_F32_SEG_MIN") ? atoi(getenv("PNN_F32_SEG_MIN")) : kSegMinDepth;
#else
        constexpr int seg_depth = kSegDepth, seg_min = kSegMinDepth;
#endif
        int tmax = 0, tmin = 1 << 30;
        for (int cls = 0; cls < p.ncls; cls++) {
            const int t = p.tap_begin[cls + 1] - p.tap_begin[cls];
            tmax = std::max(tmax, t); tmin = std::min(tmin, t);
        }
        L->nseg = 1;
        if (seg_depth > 0 && tmax > 1 && (long)tmax * p.Cin >= seg_min)
            L->nseg = (int)std::min<long>(std::min(tmin, 8), ((long)tmax * p.Cin + seg_depth - 1) / seg_depth);
        // (Round 6, built, measured, not kept: mid-depth layers -- 1152 <= K < 2304, the 16x16 net's 1152- / 1600-deep ones -- in min(taps, 4)
        // segments: a single 16x16 block 86 -> 71 us, but six blocks 96 -> 102 us, the 64x64 net 234 -> 258, and every conv net 2.5-4 % slower
        // at batch: profiles/r06_conv_kseg.txt.)
        if (L->nseg < 1) L->nseg = 1;
        // One-tap layers (FC, round 6): a 1200-deep hidden layer is ONE chain of 75 chunks per output -- 4.7 us of a single-block call's
        // 7.8 us kernel whatever the batch (profiles/r06_b1_stamps.txt).  Deeper than kFcSegChunks chunks it is summed in segments of
        // kFcSegChunks chunks (1200: 320 + 320 + 320 + 240), folded in order INSIDE the workgroup: four chains side by side at small M
        // (fcseg_f32_small_kernel), one after the other into a running total at batch (tapgemm_f32_kernel, seg_seq).
        L->fc_seg_chunks = 0;
        if (tmax == 1 && p.ncls == 1 && p.SH * p.SW == 1 && p.Cin / 16 > kFcSegChunks) {
            L->fc_seg_chunks = kFcSegChunks;
            L->nseg = (p.Cin / 16 + kFcSegChunks - 1) / kFcSegChunks;
        }
    }
    return PNN_OK;
}

// Fully-connected layer as a one-tap GEMM (pnn/components.py:169-176).
int build_fc_layer(pnn_ctx* c, Model* m, const float* W, const float* b, int K, int N, int act, GemmLayer* L)
{
    if (K % 16) return fail(c, PNN_E_MODEL, "FC input size %d is not a multiple of 16", K);
    std::vector<float> kn(W, W + (size_t)K * N);
    TapGemmParams& p = L->proto;
    p.SH = p.SW = p.IH = p.IW = p.OH = p.OW = 1;
    p.a = 1; p.os = 1; p.Cin = K; p.act = act;
    p.ncls = 1; p.tap_begin[0] = 0; p.tap_begin[1] = 1; p.py[0] = p.px[0] = 0; p.tap[0] = pack_tap(0, 0);
    L->out_per_block = N;
    return finish_gemm_layer(c, m, kn, b, N, L);
}

// Forward convolution (pnn/tfutils.py:75-139). W is [k][k][Cin][Cout].
int build_conv_layer(pnn_ctx* c, Model* m, const float* W, const float* b, int IH, int IW, int Cin, int Cout, int s,
                     GemmLayer* L)
{
    const int k = 2 * s + 1, OH = (IH + s - 1) / s, OW = (IW + s - 1) / s;
    const int pad = std::max((OH - 1) * s + k - IH, 0) / 2;
    if (Cin % 16 || Cout % 4) return fail(c, PNN_E_MODEL, "conv layer %d->%d not MFMA-tileable", Cin, Cout);
    const long K = (long)k * k * Cin;
    std::vector<float> kn(W, W + (size_t)K * Cout);
    TapGemmParams& p = L->proto;
    p.SH = OH; p.SW = OW; p.IH = IH; p.IW = IW; p.Cin = Cin; p.a = s;
    p.OH = OH; p.OW = OW; p.os = 1; p.act = 1;
    p.ncls = 1; p.tap_begin[0] = 0; p.tap_begin[1] = k * k; p.py[0] = p.px[0] = 0;
    for (int ky = 0; ky < k; ky++)
        for (int kx = 0; kx < k; kx++) p.tap[ky * k + kx] = pack_tap(ky - pad, kx - pad);
    L->out_per_block = (long)OH * OW * Cout;
    return finish_gemm_layer(c, m, kn, b, Cout, L);
}

// Transposed convolution with Cout >= 4 (Appendix B.3; pnn/tfutils.py:395-462). W is [k][k][Cout][Cin].
// Gather form: y[oy] takes x[iy] through tap ky iff iy*s + ky - pad == oy (pad = 1 for s = 1, 2).
int build_tconv_layer(pnn_ctx* c, Model* m, const float* W, const float* b, int IH, int IW, int Cin, int Cout, int s,
                      int act, GemmLayer* L)
{
    const int k = 2 * s + 1, OH = IH * s, OW = IW * s;
    const int pad = std::max((IH - 1) * s + k - OH, 0) / 2;
    if (Cin % 16 || Cout % 4 || (s != 1 && s != 2)) return fail(c, PNN_E_MODEL, "tconv layer %d->%d not tileable", Cin, Cout);
    TapGemmParams& p = L->proto;
    std::vector<float> kn;                           // rows ordered (class, tap, ci)
    int ntap = 0;
    p.ncls = s * s;
    for (int py = 0; py < s; py++)
        for (int px = 0; px < s; px++) {
            const int cls = py * s + px;
            p.tap_begin[cls] = ntap; p.py[cls] = py; p.px[cls] = px;
            for (int ky = 0; ky < k; ky++) {
                if ((py + pad - ky) % s) continue;   // C++ % keeps the sign; parity test is sign-safe
                for (int kx = 0; kx < k; kx++) {
                    if ((px + pad - kx) % s) continue;
                    p.tap[ntap] = pack_tap((py + pad - ky) / s, (px + pad - kx) / s);
                    const float* wt = W + (size_t)(ky * k + kx) * Cout * Cin;
                    for (int ci = 0; ci < Cin; ci++)
                        for (int co = 0; co < Cout; co++) kn.push_back(wt[(size_t)co * Cin + ci]);
                    ntap++;
                }
            }
        }
    p.tap_begin[p.ncls] = ntap;
    p.SH = IH; p.SW = IW; p.IH = IH; p.IW = IW; p.Cin = Cin; p.a = 1;
    p.OH = OH; p.OW = OW; p.os = s; p.act = act;
    L->out_per_block = (long)OH * OW * Cout;
    return finish_gemm_layer(c, m, kn, b, Cout, L);
}

}  // namespace

void free_model(Model* m)
{
    if (!m) return;
    std::lock_guard<std::recursive_mutex> guard(unsafe_calls_lock());
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
}

int build_model(pnn_ctx* c, int width, int is_fc, const float* params, size_t n, Model** out)
{
    // The range guard of the split-precision kernels tracks max |v| with v_max_f32, which drops NaN operands: a NaN can only
    // enter through the weights (contexts are 8-bit pixels), so a model with a non-finite parameter is refused here.
    for (size_t i = 0; i < n; i++)
        if (!std::isfinite(params[i])) return fail(c, PNN_E_MODEL, "parameter %zu of the width-%d model is not finite", i, width);
    Model* m = new Model();
    m->width = width; m->is_fc = is_fc != 0; m->n_params = (long)n;
    const float* p = params;
    const float* end = params + n;
    int rc = PNN_OK;
    auto need = [&](size_t k) { return (size_t)(end - p) >= k; };
    if (is_fc) {
        if (width != 4 && width != 8 && width != 16) { free_model(m); return fail(c, PNN_E_MODEL, "no FC architecture for width %d", width); }
        const int dims[5] = {5 * width * width, kHidden, kHidden, kHidden, width * width};
        m->fc.resize(4);
        for (int i = 0; i < 4 && rc == PNN_OK; i++) {
            const size_t nw = (size_t)dims[i] * dims[i + 1];
            if (!need(nw + dims[i + 1])) { rc = fail(c, PNN_E_MODEL, "parameter buffer too short"); break; }
            rc = build_fc_layer(c, m, p, p + nw, dims[i], dims[i + 1], i < 3, &m->fc[i]);
            p += nw + dims[i + 1];
        }
        m->n_layers = 4;
        m->pmax = kHidden;
    } else {
        int st[8];
        const int L = strides_for(width, st);
        if (L < 0) { free_model(m); return fail(c, PNN_E_MODEL, "no convolutional architecture for width %d", width); }
        int C = 32;
        for (int br = 0; br < 2 && rc == PNN_OK; br++) {
            int H = br == 0 ? width : 2 * width, Wd = br == 0 ? 3 * width : width, cin = 1, ch = 32;
            for (int i = 0; i < L && rc == PNN_OK; i++) {
                const int s = st[i], k = 2 * s + 1;
                ch *= s;
                const size_t nw = (size_t)k * k * cin * ch;
                if (!need(nw + ch)) { rc = fail(c, PNN_E_MODEL, "parameter buffer too short"); break; }
                const int OH = (H + s - 1) / s, OW = (Wd + s - 1) / s;
                if (i == 0) {
                    Conv1Layer& f = m->first[br];
                    rc = upload(c, m, p, nw, &f.d_w);
                    if (rc == PNN_OK) {
                        // split-precision path: the k*k taps x ch matrix, rows zero-padded to whole 16-deep chunks, in the split
                        // pack of the GEMM layers (max |w| scaled into [2^12, 2^13) so that hi and lo halves are f16-normal)
                        const long krows = ((long)k * k + 15) / 16 * 16;
                        std::vector<float> kn((size_t)krows * ch, 0.f);
                        std::copy(p, p + nw, kn.begin());
                        float wmax = 0.f;
                        for (size_t q = 0; q < nw; q++) wmax = std::max(wmax, std::fabs(p[q]));
                        int shift = 0;
                        if (wmax > 0.f) { int e; std::frexp(wmax, &e); shift = 13 - e; }
                        shift = std::max(-8, std::min(shift, 24));
                        f.npad = npad_for(ch);
                        f.sp_inv_scale = std::ldexp(1.f, -shift);
                        const std::vector<float> sp = pack_kn_split(kn, krows, ch, f.npad, std::ldexp(1.f, shift));
                        rc = upload(c, m, sp.data(), sp.size(), &f.d_w_sp);
                    }
                    if (rc == PNN_OK) {
                        std::vector<float> bias(ch + 4, 0.f);
                        std::copy(p + nw, p + nw + ch, bias.begin());
                        rc = upload(c, m, bias.data(), bias.size(), &f.d_bias);
                    }
                    f.proto.IH = H; f.proto.IW = Wd; f.proto.s = s; f.proto.k = k;
                    f.proto.pad = std::max((OH - 1) * s + k - H, 0) / 2;
                    f.proto.OH = OH; f.proto.OW = OW; f.proto.Cout = ch;
                    f.out_per_block = (long)OH * OW * ch;
                    m->pmax = std::max(m->pmax, f.out_per_block);
                } else {
                    m->branch[br].emplace_back();
                    rc = build_conv_layer(c, m, p, p + nw, H, Wd, cin, ch, s, &m->branch[br].back());
                    if (rc == PNN_OK) m->pmax = std::max(m->pmax, m->branch[br].back().out_per_block);
                }
                p += nw + ch;
                H = OH; Wd = OW; cin = ch;
            }
            if (rc == PNN_OK && ((br == 0 && (H != 4 || Wd != 12)) || (br == 1 && (H != 8 || Wd != 4))))
                rc = fail(c, PNN_E_MODEL, "branch output is %dx%d, expected 4x12 / 8x4", H, Wd);
            C = ch;
        }
        m->C = C;
        if (rc == PNN_OK) {                          // channel-wise FC merger: Wm [C][80][16] -> [80][16][C]
            const size_t nw = (size_t)C * 80 * 16, nb = (size_t)C * 16;
            if (!need(nw + nb)) rc = fail(c, PNN_E_MODEL, "parameter buffer too short");
            else {
                std::vector<float> wp(nw), bp(nb);
                for (int ch = 0; ch < C; ch++)
                    for (int pp = 0; pp < 80; pp++)
                        for (int j = 0; j < 16; j++) wp[((size_t)pp * 16 + j) * C + ch] = p[((size_t)ch * 80 + pp) * 16 + j];
                for (int ch = 0; ch < C; ch++)
                    for (int j = 0; j < 16; j++) bp[(size_t)j * C + ch] = p[nw + (size_t)ch * 16 + j];
                rc = upload(c, m, wp.data(), nw, &m->merger.d_w);
                if (rc == PNN_OK) rc = upload(c, m, bp.data(), nb, &m->merger.d_bias);
                m->merger.proto.C = C; m->merger.proto.na = 48; m->merger.proto.nl = 32; m->merger.proto.nout = 16;
                p += nw + nb;
                m->pmax = std::max(m->pmax, (long)16 * C);
            }
        }
        int H = 4, ci = C;
        for (int i = 0; i < L && rc == PNN_OK; i++) { // merger transposed convolutions, reversed strides
            const int s = st[L - 1 - i], k = 2 * s + 1;
            const bool last = i == L - 1;
            const int co = last ? 1 : ci / s;
            const size_t nw = (size_t)k * k * co * ci;
            if (!need(nw + co)) { rc = fail(c, PNN_E_MODEL, "parameter buffer too short"); break; }
            if (last) {
                rc = upload(c, m, p, nw, &m->last.d_w);   // [k][k][1][Cin] == [k][k][Cin]
                m->last.proto.IH = H; m->last.proto.IW = H; m->last.proto.Cin = ci; m->last.proto.s = s; m->last.proto.k = k;
                m->last.proto.pad = std::max((H - 1) * s + k - H * s, 0) / 2;
                m->last.proto.bias = p[nw];
            } else {
                m->tconv.emplace_back();
                rc = build_tconv_layer(c, m, p, p + nw, H, H, ci, co, s, 1, &m->tconv.back());
                if (rc == PNN_OK) m->pmax = std::max(m->pmax, m->tconv.back().out_per_block);
            }
            p += nw + co;
            H *= s; ci = co;
        }
        if (rc == PNN_OK && H != width) rc = fail(c, PNN_E_MODEL, "merger output width %d != %d", H, width);
        m->n_layers = 3 * L + 1;
    }
    if (rc == PNN_OK && p != end) rc = fail(c, PNN_E_MODEL, "%zu parameters given, architecture needs %zu", n, (size_t)(p - params));
    if (rc != PNN_OK) { free_model(m); return rc; }
    *out = m;
    return PNN_OK;
}

}  // namespace pnn
