// On-device choice of a split-precision tap GEMM's configuration (option "autotune"): the first time a (layer, M) pair is
// seen, every legal configuration runs the real launch -- idempotent: same inputs, same outputs, and all configurations are
// bit-identical -- and the fastest is remembered for the context.
#include "pnn_ctx.h"

#include <cstdio>
#include <cstdlib>

namespace pnn {

int tuned_cfg(pnn_ctx* c, const void* key_ptr, long M, int ncodes, int rule, const std::function<bool(int)>& legal,
              const std::function<hipError_t(int)>& launch, hipStream_t s, int* cfg, float* best_us)
{
    const auto key = std::make_pair(key_ptr, M);
    auto it = c->tuned.find(key);
    if (it != c->tuned.end()) { *cfg = it->second; if (best_us) *best_us = -1.f; return PNN_OK; }
    c->tune_gen++;                                    // a sweep runs: see pnn_ctx::overlap_ready
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    float best_ms = 1e30f, rule_ms = 1e30f;
    int best = rule;
    static const bool debug_tune = getenv("PNN_DEBUG_TUNE") != nullptr;
    for (int i = 0; i < ncodes; i++) {
        if (!legal(i)) continue;
        HIPCHK(c, launch(i));                 // warm
        HIPCHK(c, hipEventRecord(e0, s));
        for (int r = 0; r < 3; r++) HIPCHK(c, launch(i));
        HIPCHK(c, hipEventRecord(e1, s));
        HIPCHK(c, hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (debug_tune) fprintf(stderr, "[pnn]   code %d: %.1f us\n", i, ms * 1e3 / 3);
        if (i == rule) rule_ms = ms;
        if (ms < best_ms) { best_ms = ms; best = i; }
    }
    // Three back-to-back launches of one configuration are a noisy yardstick (no producer in front, caches warm from
    // the same launch): a configuration has to beat the rule-based choice by more than 3 % to replace it.  (Seen on the
    // K = 320 layer of the 8x8 FC net: the tuner took the 3-deep ring for "14.6 vs 14.7 us" where the 4-deep ring of
    // the rule runs the layer in 13.4 us inside the real pass.)
    if (best != rule && rule_ms < 1e29f && rule_ms <= best_ms * 1.03f) { best = rule; best_ms = rule_ms; }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->tuned.emplace(key, best);
    *cfg = best;
    if (best_us) *best_us = best_ms * 1e3f / 3;
    return PNN_OK;
}

}  // namespace pnn
