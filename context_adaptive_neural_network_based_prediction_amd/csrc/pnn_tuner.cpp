// On-device choice of a split-precision tap GEMM's configuration (option "autotune"): the first time a (layer, M) pair is
// seen, every legal configuration runs the real launch -- idempotent: same inputs, same outputs, and all configurations are
// bit-identical -- and the fastest is remembered for the context.
#include "pnn_ctx.h"

#include <algorithm>
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
    static const bool debug_tune = getenv("PNN_DEBUG_TUNE") != nullptr;
    auto timed = [&](int code, int reps, float* ms) -> int {
        HIPCHK(c, hipEventRecord(e0, s));
        for (int r = 0; r < reps; r++) HIPCHK(c, launch(code));
        HIPCHK(c, hipEventRecord(e1, s));
        HIPCHK(c, hipEventSynchronize(e1));
        HIPCHK(c, hipEventElapsedTime(ms, e0, e1));
        *ms /= (float)reps;
        return PNN_OK;
    };
    // Pass 1: every legal configuration, one warm launch + three timed ones.  A noisy yardstick (no producer in front, caches warm
    // from the same launch, and in a fresh process the first configurations run while the clock still ramps -- seen: the FC 8x8
    // f32 pass of one bench run 3.5 % slower than the rule-based tiles run it, with tiles this pass had "measured" faster).
    float ms1[256];
    int best = rule, second = -1;
    float best_ms = 1e30f, second_ms = 1e30f;
    int rc;
    for (int i = 0; i < ncodes && i < 256; i++) {
        ms1[i] = 1e30f;
        if (!legal(i)) continue;
        HIPCHK(c, launch(i));                 // warm
        if ((rc = timed(i, 3, &ms1[i]))) return rc;
        if (debug_tune) fprintf(stderr, "[pnn]   code %d: %.1f us\n", i, ms1[i] * 1e3);
        if (ms1[i] < best_ms) { second = best; second_ms = best_ms; best = i; best_ms = ms1[i]; }
        else if (ms1[i] < second_ms) { second = i; second_ms = ms1[i]; }
    }
    // Pass 2: the finalists -- the two fastest of pass 1 and the rule-based choice -- again, interleaved, three rounds of eight
    // launches each, by their best round; the rule-based choice stays unless beaten by more than 1 %.
    int fin[3] = {best, second, (rule >= 0 && rule < ncodes && rule < 256 && legal(rule)) ? rule : -1};
    float fin_ms[3] = {1e30f, 1e30f, 1e30f};
    for (int round = 0; round < 3; round++)
        for (int k = 0; k < 3; k++) {
            if (fin[k] < 0 || ms1[fin[k]] > 1e29f) continue;
            bool dup = false;
            for (int j = 0; j < k; j++) dup |= fin[j] == fin[k];
            if (dup) continue;
            float ms = 0.f;
            if ((rc = timed(fin[k], 8, &ms))) return rc;
            fin_ms[k] = std::min(fin_ms[k], ms);
        }
    best_ms = 1e30f;
    for (int k = 0; k < 3; k++) if (fin_ms[k] < best_ms) { best_ms = fin_ms[k]; best = fin[k]; }
    if (fin[2] >= 0) {
        float rule_ms = 1e30f;
        for (int k = 0; k < 3; k++) if (fin[k] == rule) rule_ms = std::min(rule_ms, fin_ms[k]);
        if (best != rule && rule_ms <= best_ms * 1.01f) { best = rule; best_ms = rule_ms; }
    }
    if (debug_tune) fprintf(stderr, "[pnn]   finalists %d %d %d: %.1f %.1f %.1f us -> %d\n", fin[0], fin[1], fin[2], fin_ms[0] * 1e3, fin_ms[1] * 1e3, fin_ms[2] * 1e3, best);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->tuned.emplace(key, best);
    *cfg = best;
    if (best_us) *best_us = best_ms * 1e3f;
    return PNN_OK;
}

}  // namespace pnn
