// C-ABI layer of libpnn_hip.so (declared in include/pnn_hip.h): contexts, model files, staging buffers, the prediction
// cache and the entry points.  The launch sequences live in pnn_passes.cpp, model building in pnn_model.cpp, tile rules
// in pnn_tiles.cpp, the autotuner in pnn_tuner.cpp (shared types: pnn_ctx.h).
//
// Reference behaviour reproduced here (not code): TComPrediction::initTempBuff (model selection,
// hm_16_15_substitution/source/Lib/TLibCommon/TComPrediction.cpp:108-178), load_graphs
// (hevc/hm_common/c++/source_common/integration_prediction_neural_network.cpp:29-69) and predict_by_batch_via_pnn
// (pnn/batching.py:7-88).
#include "pnn_ctx.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace pnn;

namespace pnn { std::recursive_mutex& unsafe_calls_lock() { static std::recursive_mutex m; return m; } }
namespace pnn { thread_local const LaunchEvents* g_launch_events = nullptr; thread_local double g_last_issued_frac = 1.0; }
namespace { thread_local std::string g_create_error; }
namespace pnn {

const DeviceInfo& device_info()
{
    static DeviceInfo cache[16];
    static std::once_flag filled[16];                // the service's width workers and HM's loader threads get here at the same time
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) { static const DeviceInfo fallback; return fallback; }
    DeviceInfo& d = cache[dev];
    std::call_once(filled[dev], [&d, dev] {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) == hipSuccess) {
            if (pr.multiProcessorCount > 0) d.cus = pr.multiProcessorCount;
            if (pr.maxSharedMemoryPerMultiProcessor > 0) d.lds = (size_t)pr.maxSharedMemoryPerMultiProcessor;
        }
        d.dev = dev;
    });
    return d;
}

void set_create_error(const std::string& msg) { g_create_error = msg; }


int fail(pnn_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    if (c && code == PNN_E_HIP) c->seg_cnt_dirty = true;   // a launch chain may have stopped half-way: the tiles' arrival counters are zeroed before their next use
    return code;
}

}  // namespace pnn

namespace {

// While ONE thread captures a launch chain (host_predict, option "graphs"), no other thread of the process may allocate, free or copy
// synchronously: in this runtime such a call invalidates the capture whatever the capture mode (thread-local: one HM run in five,
// relaxed: two in three -- the reference's HM loads its five graphs on five threads while the main thread is already predicting).
// Everything of that kind that this library does takes the lock; a capture holds it from begin to end.
#define PNN_UNSAFE_CALLS_GUARD std::lock_guard<std::recursive_mutex> unsafe_guard_(pnn::unsafe_calls_lock())

void cache_clear(pnn_ctx* c)                          // (every option change / model load: cached predictions and captured launch chains go)
{
    for (auto& t : c->cache) { t.clear(); t.shrink_to_fit(); }
    for (auto& kv : c->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    c->graphs.clear();
}

}  // namespace

namespace pnn {

int dev_reserve(pnn_ctx* c, DevBuf& b, size_t bytes)
{
    if (b.bytes >= bytes) return PNN_OK;
    PNN_UNSAFE_CALLS_GUARD;
    // a buffer moves: every captured launch chain of this context carries the old address in its kernels' arguments
    for (auto& kv : c->graphs) if (kv.second.exec) { (void)hipGraphExecDestroy(kv.second.exec); kv.second.exec = nullptr; kv.second.uses = 0; }
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
    const size_t want = std::max(bytes, (size_t)1 << 20);
    if (hipMalloc(&b.p, want) != hipSuccess) return fail(c, PNN_E_NOMEM, "hipMalloc(%zu) failed", want);
    b.bytes = want;
    return PNN_OK;
}

}  // namespace pnn

namespace {

Model* model_for(pnn_ctx* c, int width, int want_fc /* -1 any */, int* rc)
{
    const int idx = width_index(width);
    if (!c) { *rc = PNN_E_ARG; return nullptr; }
    if (idx < 0 || !c->models[idx]) { *rc = fail(c, PNN_E_MODEL, "no model loaded for width %d", width); return nullptr; }
    Model* m = c->models[idx];
    if (want_fc >= 0 && (int)m->is_fc != want_fc) {
        *rc = fail(c, PNN_E_MODEL, "model for width %d is %s", width, m->is_fc ? "fully-connected" : "convolutional");
        return nullptr;
    }
    *rc = PNN_OK;
    return m;
}

void reset_stats(pnn_ctx* c) { c->stat_gemm_launches = 0; c->stat_launches = 0; c->stat_gemm_flops = 0; c->stat_gemm_flops_skipped = 0; }

// Device (asynchronous) entry points cannot wait for their own pass; a pass that left the f16 range is reported by the
// next call on the context (and by pnn_check_range, which waits for the stream).
int pending_range_error(pnn_ctx* c)
{
    if (!c->h_range || !*c->h_range) return PNN_OK;
    *c->h_range = 0;
    return fail(c, PNN_E_RANGE, "an earlier asynchronous pass produced activations outside the f16 range of the split-precision "
                                "kernels (|v| >= 65504): its predictions are invalid; repeat it with pnn_set_option(ctx, \"precision\", 0)");
}

// End of a synchronous host call: wait for the context's stream.  hipStreamSynchronize parks the thread on the completion
// signal (an interrupt and a wake-up: several microseconds on a 45 us call); HM's thread has nothing else to do, so it polls.
int wait_stream(pnn_ctx* c, hipStream_t s)
{
    if (c->opt_spin_wait) {
        hipError_t e;
        while ((e = hipStreamQuery(s)) == hipErrorNotReady) {}
        if (e != hipSuccess) return fail(c, PNN_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
        return PNN_OK;
    }
    HIPCHK(c, hipStreamSynchronize(s));
    return PNN_OK;
}

// The last kernel of the pass took a completion signal (take_done_signal): spin on the flag word it raises in pinned host
// memory behind its results.  Bounded: a launch that failed never raises it, the stream then says why.
int wait_done_flag(pnn_ctx* c, hipStream_t s, long n = 1, int width = 4)
{
    const unsigned* const flag0 = reinterpret_cast<const unsigned*>(c->h_range) + (c->done_nflags > 0 ? pnn_ctx::kDoneFlag0 : 1);
    const int nflags = c->done_nflags > 0 ? c->done_nflags : 1;
    // every flag word of the call stands at its number (one word, or one per workgroup of the last kernel); `first_open` remembers where
    // the scan stopped: words in front of it have been seen raised
    int first_open = 0;
    auto raised = [&]() {
        while (first_open < nflags && __atomic_load_n(flag0 + first_open, __ATOMIC_ACQUIRE) == c->done_seq) ++first_open;
        return first_open == nflags;
    };
    if (c->opt_wait_sleep) {
        // sleep through the predictable part of the wait (see pnn_ctx::opt_wait_sleep), spin for the rest
        constexpr double kMarginUs = 14.0, kMinSleepUs = 12.0;
        int b = 0;
        while ((2L << b) <= n && b < 11) b++;
        const int wi = width_index(width) < 0 ? 0 : width_index(width);
        double& ema = c->wait_ema_us[wi][b];
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        const double nap = ema - kMarginUs;
        bool overslept = false;
        if (nap >= kMinSleepUs) {
            timespec ts{0, (long)(nap * 1e3)};
            nanosleep(&ts, nullptr);
            overslept = raised();
        }
        for (long spins = 0; !raised(); spins++) {
            if (spins < 400000) { __builtin_ia32_pause(); continue; }
            HIPCHK(c, hipStreamSynchronize(s));
            if (!raised()) return fail(c, PNN_E_HIP, "the pass finished without raising its completion flag");
            break;
        }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
        // the flag already stood when the nap ended: the true wait is unknown but shorter -- back off instead of learning the nap's length
        ema = overslept ? ema * 0.85 : (ema == 0.0 ? us : 0.9 * ema + 0.1 * us);
        return PNN_OK;
    }
    for (long spins = 0; !raised(); spins++) {
        if (spins < 200000) { __builtin_ia32_pause(); continue; }   // ~ a few milliseconds
        HIPCHK(c, hipStreamSynchronize(s));
        if (!raised()) return fail(c, PNN_E_HIP, "the pass finished without raising its completion flag");
        break;
    }
    return PNN_OK;
}

// A host pass left the f16 range of the split-precision kernels (*c->h_range raised, stream idle).  Only the blocks that
// overflow ALONE are recomputed on the exact-f32 kernels; every other block keeps its split-precision result -- the value it
// gets in any other batch (canonical_order) -- so one overflowing block behind the batching service does not change the last
// float bits of the other clients' blocks.  `pass(b0, nb)` runs blocks [b0, b0 + nb) of the staged batch into their slots.
// Batches above kPerBlockMax take the whole-batch repeat (a stand-alone caller, not the service, whose batches are <= 256).
template <typename Pass>
int range_fallback(pnn_ctx* c, long n, Pass pass, hipStream_t s)
{
    constexpr long kPerBlockMax = 256;
    *c->h_range = 0;
    c->range_fallbacks++;
    const long keep = c->opt_precision;
    int rc = PNN_OK;
    if (n == 1 || n > kPerBlockMax) {
        c->opt_precision = 0;
        rc = pass(0, n);
        c->opt_precision = keep;
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(s));
        return PNN_OK;
    }
    for (long i = 0; i < n; i++) {
        if ((rc = pass(i, 1))) return rc;             // alone, split precision: the same bits as inside the batch
        HIPCHK(c, hipStreamSynchronize(s));
        if (!*c->h_range) continue;
        *c->h_range = 0;
        c->opt_precision = 0;
        rc = pass(i, 1);
        c->opt_precision = keep;
        if (rc) return rc;
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return PNN_OK;
}

struct PnnwHeader { char magic[4]; uint32_t version, width, is_fc; uint64_t n_params, reserved; };

}  // namespace

extern "C" {

int pnn_create_empty(pnn_ctx** out, float mean, int device)
{
    PNN_UNSAFE_CALLS_GUARD;
    if (!out) return fail(nullptr, PNN_E_ARG, "`out` is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PNN_E_HIP, "no HIP device is visible: libpnn_hip.so has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, PNN_E_ARG, "device %d out of range [0, %d)", device, ndev);
    pnn_ctx* c = new pnn_ctx();
    c->device = device; c->mean = mean;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, PNN_E_HIP, "cannot initialise HIP device %d", device);
    }
    if (const char* e = getenv("PNN_MAX_CHUNK")) c->opt_max_chunk = atol(e);
    if (const char* e = getenv("PNN_PRECISION")) c->opt_precision = atol(e);
    if (const char* e = getenv("PNN_AUTOTUNE")) c->opt_autotune = atol(e);
    if (const char* e = getenv("PNN_F32_CFG")) c->opt_f32_cfg = atol(e);
    if (const char* e = getenv("PNN_F32_OVERLAP")) c->opt_f32_overlap = atol(e);
    if (const char* e = getenv("PNN_F32_SMALL")) c->opt_f32_small = atol(e);
    if (const char* e = getenv("PNN_F32_SMALL_TILES")) c->opt_f32_small_tiles = atol(e);
    if (const char* e = getenv("PNN_CONVIMG")) c->opt_convimg = atol(e);
    if (const char* e = getenv("PNN_RING")) c->opt_ring = atol(e);
    if (const char* e = getenv("PNN_SMALL")) c->opt_small = atol(e);
    if (const char* e = getenv("PNN_FUSE_LAST")) c->opt_fuse_last = atol(e);
    if (const char* e = getenv("PNN_F32_SEG_MODE")) c->opt_f32_seg_mode = atol(e);
    if (const char* e = getenv("PNN_F32_PERSIST")) c->opt_f32_persist = atol(e);
    if (const char* e = getenv("PNN_FUSE_FIRST")) c->opt_fuse_first = atol(e);
    if (const char* e = getenv("PNN_FUSE_GATHER")) c->opt_fuse_gather = atol(e);
    if (const char* e = getenv("PNN_FUSE_TAIL")) c->opt_fuse_tail = atol(e);
    if (const char* e = getenv("PNN_RING_PM")) c->opt_ring_pm = atol(e);
    if (const char* e = getenv("PNN_BRANCH_STREAMS")) c->opt_branch_streams = atol(e);
    if (const char* e = getenv("PNN_CACHE_MB")) c->opt_cache_mb = atol(e);
    if (const char* e = getenv("PNN_FC_OUT")) c->opt_fc_out = atol(e);
    if (const char* e = getenv("PNN_SPIN_WAIT")) c->opt_spin_wait = atol(e);
    if (const char* e = getenv("PNN_FLAG_WAIT")) c->opt_flag_wait = atol(e);
    if (const char* e = getenv("PNN_WAIT_SLEEP")) c->opt_wait_sleep = atol(e);
    if (const char* e = getenv("PNN_GRAPHS")) c->opt_graphs = atol(e);
    if (const char* e = getenv("PNN_F32_SMALL_DEEP")) c->opt_f32_small_deep = atol(e);
    if (const char* e = getenv("PNN_CHAIN_IO")) c->opt_chain_io = atol(e);
    if (const char* e = getenv("PNN_TAILS")) c->opt_tails = atol(e);
    if (hipHostMalloc((void**)&c->h_range, (pnn_ctx::kDoneFlag0 + pnn_ctx::kDoneFlagsMax) * 4, hipHostMallocDefault) != hipSuccess) {
        pnn_destroy(c);
        return fail(nullptr, PNN_E_NOMEM, "hipHostMalloc of the range flag failed");
    }
    *c->h_range = 0;
    for (int i = 1; i < pnn_ctx::kDoneFlag0 + pnn_ctx::kDoneFlagsMax; i++) c->h_range[i] = 0;   // the completion flag(s) of the small host calls (signal_done)
    if (hipMalloc((void**)&c->d_done, 256) != hipSuccess || hipMemset(c->d_done, 0, 256) != hipSuccess) {
        pnn_destroy(c);
        return fail(nullptr, PNN_E_NOMEM, "hipMalloc of the completion counter failed");
    }
    if (hipMalloc((void**)&c->d_seg_cnt, pnn_ctx::kCntWords * 4) != hipSuccess || hipMemset(c->d_seg_cnt, 0, pnn_ctx::kCntWords * 4) != hipSuccess) {
        pnn_destroy(c);
        return fail(nullptr, PNN_E_NOMEM, "hipMalloc of the K-segment counters failed");
    }
    if (hipMalloc(&c->d_zero, 4096) != hipSuccess || hipMemset(c->d_zero, 0, 4096) != hipSuccess) {
        pnn_destroy(c);
        return fail(nullptr, PNN_E_NOMEM, "hipMalloc of the zero page failed");
    }
    *out = c;
    return PNN_OK;
}

int pnn_load_model_params(pnn_ctx* c, int width, int is_fc, const float* params, size_t n)
{
    // (the process-wide lock of the capture-unsafe calls is taken where those calls are -- upload() / free_model() in pnn_model.cpp, the
    // tail below -- not around the file read and the weight packing: HM loads its five models on five threads, and they should overlap)
    if (!c || !params) return fail(c, PNN_E_ARG, "NULL argument");
    const int idx = width_index(width);
    if (idx < 0) return fail(c, PNN_E_ARG, "width %d is not in {4, 8, 16, 32, 64}", width);
    HIPCHK(c, hipSetDevice(c->device));
    Model* m = nullptr;
    const int rc = build_model(c, width, is_fc, params, n, &m);
    if (rc) return rc;
    PNN_UNSAFE_CALLS_GUARD;
    free_model(c->models[idx]);
    c->models[idx] = m;
    c->tuned.clear(); c->tune_gen++;                                 // keys point into the replaced model
    cache_clear(c);
    return PNN_OK;
}

int pnn_load_model_file(pnn_ctx* c, const char* path)
{
    if (!c || !path) return fail(c, PNN_E_ARG, "NULL argument");
    std::vector<char> data;
    if (!read_file(path, &data)) return fail(c, PNN_E_IO, "The model file at \"%s\" cannot be loaded.", path);
    if (data.size() < sizeof(PnnwHeader)) return fail(c, PNN_E_IO, "%s: truncated header", path);
    PnnwHeader h;
    memcpy(&h, data.data(), sizeof h);
    if (memcmp(h.magic, "PNNW", 4) || h.version != 1) return fail(c, PNN_E_IO, "%s is not a PNNW v1 file", path);
    if (data.size() != sizeof h + h.n_params * 4) return fail(c, PNN_E_IO, "%s: size does not match its header", path);
    std::vector<float> params(h.n_params);
    memcpy(params.data(), data.data() + sizeof h, h.n_params * 4);
    return pnn_load_model_params(c, (int)h.width, (int)h.is_fc, params.data(), params.size());
}

int pnn_create(pnn_ctx** out, const char* table_path, int use_pair, float mean, int device)
{
    if (!out) return fail(nullptr, PNN_E_ARG, "`out` is NULL");
    *out = nullptr;
    std::vector<TableEntry> entries;
    std::string err;
    int rc = parse_table(table_path, &entries, &err);
    if (rc) return fail(nullptr, rc, "%s", err.c_str());
    bool have_pair = false;
    for (const TableEntry& e : entries) have_pair |= e.is_pair != 0;
    const int want_pair = (have_pair && use_pair) ? 1 : 0;      // TComPrediction.cpp:156
    pnn_ctx* c = nullptr;
    if ((rc = pnn_create_empty(&c, mean, device))) return rc;
    std::string dir(table_path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    static const int widths[5] = {4, 8, 16, 32, 64};
    for (int wi = 0; wi < 5; wi++) {
        const TableEntry* hit = nullptr;
        for (const TableEntry& e : entries)
            if (e.width == widths[wi] && e.is_pair == want_pair && e.channel == 0) hit = &e;   // later lines overwrite (std::map)
        if (!hit) {
            rc = fail(nullptr, PNN_E_MODEL, "model table has no (%d, %s, luminance) entry", widths[wi], want_pair ? "pair" : "single");
            break;
        }
        std::string p = hit->path;
        if (!p.empty() && p[0] != '/') {
            FILE* f = fopen((dir + "/" + p).c_str(), "rb");
            if (f) { fclose(f); p = dir + "/" + p; }
        }
        rc = pnn_load_model_file(c, p.c_str());
        if (rc) { g_create_error = c->err; break; }
        const Model* m = c->models[wi];
        if (m->width != widths[wi]) { rc = fail(nullptr, PNN_E_MODEL, "%s holds a width-%d model, table says %d", p.c_str(), m->width, widths[wi]); break; }
    }
    if (rc) { pnn_destroy(c); return rc; }
    *out = c;
    return PNN_OK;
}

void pnn_destroy(pnn_ctx* c)
{
    if (!c) return;
    PNN_UNSAFE_CALLS_GUARD;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    cache_clear(c);
    for (Model*& m : c->models) { free_model(m); m = nullptr; }
    for (DevBuf& b : c->ws) if (b.p) (void)hipFree(b.p);
    for (DevBuf& b : c->stage_in) if (b.p) (void)hipFree(b.p);
    for (DevBuf& b : c->stage_out) if (b.p) (void)hipFree(b.p);
    if (c->stage_tbs.p) (void)hipFree(c->stage_tbs.p);
    for (auto& b : c->seg_part) if (b.p) (void)hipFree(b.p);
    if (c->d_zero) (void)hipFree(c->d_zero);
    if (c->d_done) (void)hipFree(c->d_done);
    if (c->d_seg_cnt) (void)hipFree(c->d_seg_cnt);
    if (c->copy_in) {
        (void)hipStreamDestroy(c->copy_in); (void)hipStreamDestroy(c->copy_out);
        for (int i = 0; i < 2; i++) { (void)hipEventDestroy(c->ev_in[i]); (void)hipEventDestroy(c->ev_pass[i]); (void)hipEventDestroy(c->ev_out[i]); }
    }
    for (pnn::DevBuf* b : {&c->stage2_in[0], &c->stage2_in[1], &c->stage2_out[0], &c->stage2_out[1]}) if (b->p) (void)hipFree(b->p);
    if (c->side_stream) { (void)hipStreamSynchronize(c->side_stream); (void)hipStreamDestroy(c->side_stream); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    if (c->h_range) (void)hipHostFree(c->h_range);
    if (c->stream && c->stream_owned) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* pnn_last_error(const pnn_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
float pnn_mean(const pnn_ctx* c) { return c ? c->mean : 0.f; }

int pnn_model_info(const pnn_ctx* c, int width, int* is_fc, int* n_layers, long* n_params)
{
    const int idx = width_index(width);
    if (!c || idx < 0 || !c->models[idx]) return PNN_E_MODEL;
    if (is_fc) *is_fc = c->models[idx]->is_fc;
    if (n_layers) *n_layers = c->models[idx]->n_layers;
    if (n_params) *n_params = c->models[idx]->n_params;
    return PNN_OK;
}

int pnn_arithmetic_tag(const pnn_ctx* c, char* out, size_t bytes)
{
    if (!c || !out || bytes == 0) return PNN_E_ARG;
    // everything that decides the last float bits of a prediction: the arithmetic, its per-output summation order, the library's
    // order revision (bumped whenever a kernel change moves a bit)
    if (c->opt_precision == 0)
        snprintf(out, bytes, "pnn-order-6:f32:fmaf-chain k=0,8,1,9..7,15 per 16:kseg %d/%d:fc-kseg %d:fc-out-seg 160", kSegDepth, kSegMinDepth, 16 * kFcSegChunks);
    else
        snprintf(out, bytes, "pnn-order-5:split-f16x3:hi*hi,hi*lo,lo*hi per 16:fc-out-seg %d", 16 * kFuseSegChunks);
    return PNN_OK;
}

int pnn_num_split_configs(void) { return tapgemm_sp_num_cfgs() + convimg_sp_num_cfgs() + tapgemm_ring_num_cfgs(); }
int pnn_num_f32_configs(void) { return tapgemm_f32_num_cfgs(); }

int pnn_set_option(pnn_ctx* c, const char* name, long value)
{
    if (!c || !name) return PNN_E_ARG;
    PNN_UNSAFE_CALLS_GUARD;
    if (!strcmp(name, "max_chunk")) c->opt_max_chunk = value;
    else if (!strcmp(name, "canonical_order")) {       // kept as a name: one summation order at every batch size is the only mode since round 5
        if (value != 1) return fail(c, PNN_E_ARG, "canonical_order = %ld: the kernels with another summation order were removed; 1 is the only mode", value);
    }
    else if (!strcmp(name, "time_launches")) c->opt_time_launches = value;
    else if (!strcmp(name, "precision")) c->opt_precision = value;
    else if (!strcmp(name, "autotune")) { c->opt_autotune = value; c->tune_gen++; }
    else if (!strcmp(name, "convimg")) { c->opt_convimg = value; c->tuned.clear(); c->tune_gen++; }
    else if (!strcmp(name, "ring")) { c->opt_ring = value; c->tuned.clear(); c->tune_gen++; }
    else if (!strcmp(name, "small")) c->opt_small = value;
    else if (!strcmp(name, "small_max_tiles")) c->opt_small_tiles = value;
    else if (!strcmp(name, "pair")) c->opt_pair = value;
    else if (!strcmp(name, "fc_out")) c->opt_fc_out = value;
    else if (!strcmp(name, "spin_wait")) c->opt_spin_wait = value;
    else if (!strcmp(name, "flag_wait")) c->opt_flag_wait = value;
    else if (!strcmp(name, "wait_sleep")) c->opt_wait_sleep = value;
    else if (!strcmp(name, "graphs")) c->opt_graphs = value;
    else if (!strcmp(name, "stream_priority")) {
        // the context's own stream (the host entry points run on it) at the device's greatest (< 0), default (0) or least (> 0) priority
        HIPCHK(c, hipSetDevice(c->device));
        int least = 0, greatest = 0;
        HIPCHK(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int pr = value < 0 ? greatest : value > 0 ? least : (least + greatest) / 2;
        hipStream_t ns = nullptr;
        HIPCHK(c, hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, pr));
        if (c->stream) { (void)hipStreamSynchronize(c->stream); if (c->stream_owned) (void)hipStreamDestroy(c->stream); }
        c->stream = ns; c->stream_owned = true;
    }
    else if (!strcmp(name, "stream")) {
        // the host entry points run on the caller's stream (a hipStream_t passed as the value; the caller keeps and destroys it) --
        // what pnn_streams_on_distinct_queues is for
        if (!value) return fail(c, PNN_E_ARG, "stream = NULL");
        HIPCHK(c, hipSetDevice(c->device));
        if (c->stream) { (void)hipStreamSynchronize(c->stream); if (c->stream_owned) (void)hipStreamDestroy(c->stream); }
        c->stream = reinterpret_cast<hipStream_t>(value); c->stream_owned = false;
    }
    else if (!strcmp(name, "fuse_last")) c->opt_fuse_last = value;
    else if (!strcmp(name, "f32_seg_mode")) { c->opt_f32_seg_mode = value; c->tuned.clear(); c->tune_gen++; }
    else if (!strcmp(name, "f32_persist")) { c->opt_f32_persist = value; c->tuned.clear(); c->tune_gen++; }
    else if (!strcmp(name, "fuse_first")) { c->opt_fuse_first = value; c->tuned.clear(); c->tune_gen++; }
    else if (!strcmp(name, "fuse_gather")) c->opt_fuse_gather = value;
    else if (!strcmp(name, "fuse_tail")) { c->opt_fuse_tail = value; c->tuned.clear(); c->tune_gen++; }
    else if (!strcmp(name, "ring_pm")) { c->opt_ring_pm = value; c->tuned.clear(); c->tune_gen++; }
    else if (!strcmp(name, "branch_streams")) c->opt_branch_streams = value;
    else if (!strcmp(name, "cache_mb")) { c->opt_cache_mb = value; c->cache_hits = c->cache_misses = 0; }
    else if (!strcmp(name, "sp_cfg")) c->opt_sp_cfg = value;
    else if (!strcmp(name, "f32_cfg")) c->opt_f32_cfg = value;
    else if (!strcmp(name, "f32_overlap")) c->opt_f32_overlap = value;
    else if (!strcmp(name, "f32_small")) c->opt_f32_small = value;
    else if (!strcmp(name, "fc_out_f32")) c->opt_fc_out_f32 = value;
    else if (!strcmp(name, "host_slice")) c->opt_host_slice = value;
    else if (!strcmp(name, "chain_io")) c->opt_chain_io = value;
    else if (!strcmp(name, "tails")) c->opt_tails = value;
    else if (!strcmp(name, "seg_fold")) c->opt_seg_fold = value;
    else if (!strcmp(name, "f32_small_deep")) c->opt_f32_small_deep = value;
    else if (!strcmp(name, "f32_small_max_tiles")) c->opt_f32_small_tiles = value;
    else if (!strcmp(name, "ws_cap_mb")) c->ws_cap_bytes = (size_t)value << 20;
    else return fail(c, PNN_E_ARG, "unknown option %s", name);
    cache_clear(c);                                   // any option may change the arithmetic path: cached predictions are dropped
    return PNN_OK;
}

int pnn_last_call_stats(const pnn_ctx* c, int* n_gemm, double* flops, int* n_launches)
{
    if (!c) return PNN_E_ARG;
    if (n_gemm) *n_gemm = c->stat_gemm_launches;
    if (flops) *flops = c->stat_gemm_flops;
    if (n_launches) *n_launches = c->stat_launches;
    return PNN_OK;
}

int pnn_last_call_issued_flops(const pnn_ctx* c, double* flops)
{
    if (!c || !flops) return PNN_E_ARG;
    *flops = c->stat_gemm_flops - c->stat_gemm_flops_skipped;
    return PNN_OK;
}

int pnn_launch_times(pnn_ctx* c, int kind, int* n_launches, double* total_us, double* total_flops)
{
    if (!c) return PNN_E_ARG;
    int n = 0;
    double us = 0, fl = 0;
    std::vector<pnn_ctx::LaunchRec> keep;
    for (pnn_ctx::LaunchRec& r : c->launch_recs) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        HIPCHK(c, hipEventSynchronize(r.e1));
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, r.e0, r.e1));
        us += ms * 1e3; fl += r.flops; n++;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    c->launch_recs.swap(keep);
    if (n_launches) *n_launches = n;
    if (total_us) *total_us = us;
    if (total_flops) *total_flops = fl;
    return PNN_OK;
}

// ---- device-resident entry points ----------------------------------------------------------------------

int pnn_predict_fc_device(pnn_ctx* c, int width, const float* d_ctx, int n, float* d_out, void* stream)
{
    int rc;
    Model* m = model_for(c, width, 1, &rc);
    if (!m) return rc;
    if (n < 0 || (n > 0 && (!d_ctx || !d_out))) return fail(c, PNN_E_ARG, "bad buffers / batch size");
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = pending_range_error(c))) return rc;
    reset_stats(c);
    hipStream_t s = (hipStream_t)stream;
    return run_net(c, m, d_ctx, 5L * width * width, nullptr, 0, n, d_out, nullptr, s);
}

int pnn_predict_conv_device(pnn_ctx* c, int width, const float* d_above, const float* d_left, int n, float* d_out, void* stream)
{
    int rc;
    Model* m = model_for(c, width, 0, &rc);
    if (!m) return rc;
    if (n < 0 || (n > 0 && (!d_above || !d_left || !d_out))) return fail(c, PNN_E_ARG, "bad buffers / batch size");
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = pending_range_error(c))) return rc;
    reset_stats(c);
    hipStream_t s = (hipStream_t)stream;
    return run_net(c, m, d_above, 3L * width * width, d_left, 2L * width * width, n, d_out, nullptr, s);
}

int pnn_gather_device(pnn_ctx* c, int width, int unit, const void* d_plane, int pel_bytes, const pnn_tb_dev* d_tbs, int n,
                      float* d_above, long pitch_above, float* d_left, long pitch_left, void* stream)
{
    if (!c) return PNN_E_ARG;
    if (width_index(width) < 0 || (unit != 4 && unit != 2) || n < 0 || (n > 0 && (!d_plane || !d_tbs || !d_above || !d_left)))
        return fail(c, PNN_E_ARG, "bad gather arguments");
    if (pel_bytes != 4 && pel_bytes != 1) return fail(c, PNN_E_ARG, "pel_bytes must be 4 (HM Pel) or 1 (uint8)");
    HIPCHK(c, hipSetDevice(c->device));
    GatherParams g;
    g.plane = d_plane; g.pel_bytes = pel_bytes; g.tbs = reinterpret_cast<const TbDev*>(d_tbs); g.N = n; g.w = width;
    g.unit = unit; g.mean = c->mean; g.above = d_above; g.left = d_left; g.pitch_above = pitch_above; g.pitch_left = pitch_left;
    g.split = 0;
    HIPCHK(c, launch_gather(g, (hipStream_t)stream));
    return PNN_OK;
}

int pnn_predict_tbs_device(pnn_ctx* c, int width, const void* d_plane, int pel_bytes, const pnn_tb_dev* d_tbs, int n,
                           int32_t* d_dst, float* d_out_f32, void* stream)
{
    int rc;
    Model* m = model_for(c, width, -1, &rc);
    if (!m) return rc;
    if (n < 0 || (n > 0 && (!d_plane || !d_tbs || (!d_dst && !d_out_f32)))) return fail(c, PNN_E_ARG, "bad buffers / batch size");
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = pending_range_error(c))) return rc;
    reset_stats(c);
    hipStream_t s = (hipStream_t)stream;
    const long w2 = (long)width * width;
    const long chunk = std::min((long)n, chunk_blocks(c, m));
    if ((rc = dev_reserve(c, c->stage_in[0], (size_t)chunk * 5 * w2 * 4))) return rc;
    float* ctxbuf = (float*)c->stage_in[0].p;
    for (long b0 = 0; b0 < n; b0 += chunk) {
        const long nb = std::min(chunk, (long)n - b0);
        float* ab = ctxbuf;
        float* lf = m->is_fc ? ctxbuf + 3 * w2 : ctxbuf + nb * 3 * w2;
        const long pa = m->is_fc ? 5 * w2 : 3 * w2, pl = m->is_fc ? 5 * w2 : 2 * w2;
        const bool split_ctx = m->is_fc && pass_uses_split(c, m, nb);   // the FC chain starts on the split-precision GEMM
        // convolutional nets whose first convolutions run inside the image kernel: the gather goes in there too
        const bool lazy = !m->is_fc && nb <= chunk_blocks(c, m) && conv_pass_fuses_first(c, m, nb);
        if (lazy) {
            c->lazy.plane = d_plane; c->lazy.tbs = reinterpret_cast<const TbDev*>(d_tbs + b0); c->lazy.pel_bytes = pel_bytes; c->lazy.unit = 4;
            ab = lf = nullptr;
        } else {
            GatherParams g;
            g.plane = d_plane; g.pel_bytes = pel_bytes; g.tbs = reinterpret_cast<const TbDev*>(d_tbs + b0); g.N = (int)nb; g.w = width;
            g.unit = 4; g.mean = c->mean; g.above = ab; g.left = lf; g.pitch_above = pa; g.pitch_left = pl; g.split = split_ctx ? 1 : 0;
            HIPCHK(c, launch_gather(g, s));
            c->stat_launches++;
        }
        rc = run_net(c, m, ab, pa, lf, pl, nb, d_out_f32 ? d_out_f32 + b0 * w2 : nullptr, d_dst ? d_dst + b0 * w2 : nullptr, s, split_ctx);
        c->lazy = pnn_ctx::LazyGather();
        if (rc) return rc;
    }
    return PNN_OK;
}

int pnn_block_cost_device(pnn_ctx* c, int width, const void* d_org_plane, int pel_bytes, const pnn_tb_dev* d_tbs, int n,
                          const int32_t* d_pred, int hadamard, uint32_t* d_cost, void* stream)
{
    if (!c) return PNN_E_ARG;
    if (width_index(width) < 0 || n < 0 || (n > 0 && (!d_org_plane || !d_tbs || !d_pred || !d_cost))) return fail(c, PNN_E_ARG, "bad cost arguments");
    if (pel_bytes != 4 && pel_bytes != 1) return fail(c, PNN_E_ARG, "pel_bytes must be 4 (HM Pel) or 1 (uint8)");
    HIPCHK(c, hipSetDevice(c->device));
    BlockCostParams b;
    b.org_plane = d_org_plane; b.pel_bytes = pel_bytes; b.tbs = reinterpret_cast<const TbDev*>(d_tbs); b.N = n; b.w = width;
    b.pred = d_pred; b.hadamard = hadamard; b.cost = d_cost;
    HIPCHK(c, launch_block_cost(b, (hipStream_t)stream));
    return PNN_OK;
}

int pnn_predict_tbs_cost_device(pnn_ctx* c, int width, const void* d_plane, const void* d_org_plane, int pel_bytes,
                                const pnn_tb_dev* d_tbs, int n, int hadamard, uint32_t* d_cost, int32_t* d_dst, void* stream)
{
    if (!c) return PNN_E_ARG;
    if (n > 0 && (!d_org_plane || !d_cost)) return fail(c, PNN_E_ARG, "bad cost arguments");
    int rc;
    if (!d_dst && n > 0) {                            // the predictions themselves are not wanted: keep them in the context's buffer
        if ((rc = dev_reserve(c, c->stage_out[1], (size_t)n * width * width * 4))) return rc;
        d_dst = (int32_t*)c->stage_out[1].p;
    }
    if ((rc = pnn_predict_tbs_device(c, width, d_plane, pel_bytes, d_tbs, n, d_dst, nullptr, stream))) return rc;
    rc = pnn_block_cost_device(c, width, d_org_plane, pel_bytes, d_tbs, n, d_dst, hadamard, d_cost, stream);
    if (rc == PNN_OK) c->stat_launches++;
    return rc;
}

// ---- host-buffer entry points ----------------------------------------------------------------------------

// Hash of the input bytes for the prediction cache: 8 bytes per step (a byte-wise FNV-1a cost 1.3 us per 8x8 lookup and 80 us
// per 64x64 one -- HM's RD search makes ~100 k lookups per picture); the entry is confirmed with memcmp, so only the spread matters.
static uint64_t hash_bytes(const void* data, size_t bytes, uint64_t h = 0x9e3779b97f4a7c15ull)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        h = (h ^ v) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    for (; i < bytes; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h ^ (h >> 29);
}

// No NaN / infinity among n floats: on the bit patterns (exponent all ones), so that the loop vectorises -- std::isfinite with an early
// exit ran at 13 M floats/ms, 0.4 ms of a 0.8 ms FC-8 call of 4096 blocks (PNN_HOST_TRACE).
static bool all_finite(const float* x, size_t n)
{
    uint32_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t u;
        memcpy(&u, x + i, 4);
        bad |= (uint32_t)((u & 0x7f800000u) == 0x7f800000u);
    }
    return bad == 0;
}

static int host_predict(pnn_ctx* c, Model* m, const float* above, const float* left, int n, float* out, int32_t* dst, int dst_stride);

// host_predict for a call of several passes' worth of blocks: slices of `slice` blocks through TWO staging sets on three streams --
// H2D of slice i + 1 and D2H of slice i - 1 beside the pass of slice i.  The copies come from / go to the caller's (pageable) arrays
// through the runtime's own staging, which holds the calling thread for their length: the order of issue below is what overlaps them with
// the device's work (H2D i + 1, then pass i + 1 enqueued behind it, then D2H i).  Same kernels per block, one summation order at every
// batch size: bit-identical to the sequential call (tests/test_gpu_parity.py).
static int host_predict_sliced(pnn_ctx* c, Model* m, const float* above, const float* left, int n, float* out, int32_t* dst, int dst_stride, long slice)
{
    const int w = m->width;
    const long w2 = (long)w * w;
    const size_t pa = (size_t)(m->is_fc ? 5 : 3) * w2, pl = m->is_fc ? 0 : (size_t)2 * w2;
    HIPCHK(c, hipSetDevice(c->device));
    reset_stats(c);
    int rc;
    hipStream_t s = c->stream;
    if (!c->copy_in) {
        PNN_UNSAFE_CALLS_GUARD;
        HIPCHK(c, hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_pass[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming));
        }
    }
    for (int k = 0; k < 2; k++) {
        DevBuf* in = k ? c->stage2_in : c->stage_in;
        DevBuf* ob = k ? c->stage2_out : c->stage_out;
        if ((rc = dev_reserve(c, in[0], (size_t)slice * pa * 4))) return rc;
        if (pl && (rc = dev_reserve(c, in[1], (size_t)slice * pl * 4))) return rc;
        if ((rc = dev_reserve(c, ob[0], (size_t)slice * w2 * 4))) return rc;
        if (dst && (rc = dev_reserve(c, ob[1], (size_t)slice * w2 * 4))) return rc;
    }
    HIPCHK(c, hipStreamSynchronize(s));                // the staging sets may still be read by an earlier asynchronous call of this context
    const long nsl = (n + slice - 1) / slice;
    auto count = [&](long i) { return std::min<long>(slice, (long)n - i * slice); };
    // TWO host threads: a copy from / to the caller's pageable arrays holds its calling thread for its length (the runtime stages it), and
    // scan + copy-in of an FC 8x8 slice (5.2 MB) are as long as its pass -- one thread doing both directions was the bound (11.4 M blocks/s
    // for 17.8 M on the device, profiles/r06_host_rate.txt).  The FEEDER scans and copies slices in, as far ahead as the two staging sets
    // allow; this thread enqueues the passes and copies the results out.  What couples them: `fed` (slices whose copy-in has been issued,
    // their event recorded) and `enqueued` (slices whose pass has been enqueued, its event recorded) under one mutex.
    std::mutex mu;
    std::condition_variable cv;
    long fed = 0, enqueued = 0;
    int feeder_rc = PNN_OK;
    std::string feeder_err;
    bool abort_all = false;
    const int device = c->device;
    std::thread feeder([&] {
        if (hipSetDevice(device) != hipSuccess) { std::lock_guard<std::mutex> lk(mu); feeder_rc = PNN_E_HIP; feeder_err = "hipSetDevice failed in the copy-in thread"; cv.notify_all(); return; }
        for (long i = 0; i < nsl; i++) {
            const int k = (int)(i & 1);
            DevBuf* in = k ? c->stage2_in : c->stage_in;
            hipError_t e = hipSuccess;
            int rcl = PNN_OK;
            // (the non-finite scan of THIS slice, here and not over the whole call up front: 84 MB of a 16-slice FC 8x8 call are 2 ms of one
            // host thread, and the device computes the slices before meanwhile)
            if (!all_finite(above + (size_t)i * slice * pa, (size_t)count(i) * pa) || (pl && !all_finite(left + (size_t)i * slice * pl, (size_t)count(i) * pl))) rcl = PNN_E_ARG;
            if (rcl == PNN_OK && i >= 2) {           // the pass of slice i - 2 reads this set: its event must exist before this stream can wait for it
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return enqueued >= i - 1 || abort_all; });
                if (abort_all) return;
                lk.unlock();
                e = hipStreamWaitEvent(c->copy_in, c->ev_pass[k], 0);
            }
            if (rcl == PNN_OK && e == hipSuccess) e = hipMemcpyAsync(in[0].p, above + (size_t)i * slice * pa, (size_t)count(i) * pa * 4, hipMemcpyHostToDevice, c->copy_in);
            if (rcl == PNN_OK && e == hipSuccess && pl) e = hipMemcpyAsync(in[1].p, left + (size_t)i * slice * pl, (size_t)count(i) * pl * 4, hipMemcpyHostToDevice, c->copy_in);
            if (rcl == PNN_OK && e == hipSuccess) e = hipEventRecord(c->ev_in[k], c->copy_in);
            std::lock_guard<std::mutex> lk(mu);
            if (rcl != PNN_OK || e != hipSuccess) {
                feeder_rc = rcl != PNN_OK ? rcl : PNN_E_HIP;
                feeder_err = rcl != PNN_OK ? "non-finite value in the input contexts" : std::string("copy-in of a slice failed: ") + hipGetErrorString(e);
                cv.notify_all();
                return;
            }
            fed = i + 1;
            cv.notify_all();
        }
    });
    auto pass = [&](long i) -> int {
        const int k = (int)(i & 1);
        DevBuf* in = k ? c->stage2_in : c->stage_in;
        DevBuf* ob = k ? c->stage2_out : c->stage_out;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return fed > i || feeder_rc != PNN_OK; });
            if (fed <= i) return feeder_rc;
        }
        HIPCHK(c, hipStreamWaitEvent(s, c->ev_in[k], 0));
        if (i >= 2) HIPCHK(c, hipStreamWaitEvent(s, c->ev_out[k], 0));                  // the results of slice i - 2 have left this set
        const int r = run_net(c, m, (const float*)in[0].p, (long)pa, (const float*)in[1].p, (long)pl, count(i), (float*)ob[0].p, dst ? (int32_t*)ob[1].p : nullptr, s);
        if (r) return r;
        HIPCHK(c, hipEventRecord(c->ev_pass[k], s));
        { std::lock_guard<std::mutex> lk(mu); enqueued = i + 1; }
        cv.notify_all();
        return PNN_OK;
    };
    auto copy_out = [&](long i) -> int {
        const int k = (int)(i & 1);
        DevBuf* ob = k ? c->stage2_out : c->stage_out;
        HIPCHK(c, hipStreamWaitEvent(c->copy_out, c->ev_pass[k], 0));
        if (out) HIPCHK(c, hipMemcpyAsync(out + (size_t)i * slice * w2, ob[0].p, (size_t)count(i) * w2 * 4, hipMemcpyDeviceToHost, c->copy_out));
        if (dst) {
            if (dst_stride == w) HIPCHK(c, hipMemcpyAsync(dst + (size_t)i * slice * w2, ob[1].p, (size_t)count(i) * w2 * 4, hipMemcpyDeviceToHost, c->copy_out));
            else HIPCHK(c, hipMemcpy2DAsync(dst + (size_t)i * slice * w * dst_stride, (size_t)dst_stride * 4, ob[1].p, (size_t)w * 4, (size_t)w * 4, (size_t)count(i) * w,
                                            hipMemcpyDeviceToHost, c->copy_out));
        }
        HIPCHK(c, hipEventRecord(c->ev_out[k], c->copy_out));
        return PNN_OK;
    };
    // nothing of a failed call stays in flight, and the feeder is never left waiting
    auto finish = [&](int code) {
        { std::lock_guard<std::mutex> lk(mu); abort_all = code != PNN_OK; }
        cv.notify_all();
        feeder.join();
        (void)hipStreamSynchronize(c->copy_in); (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(c->copy_out);
        if (code == PNN_OK && feeder_rc != PNN_OK) code = feeder_rc;
        if (code != PNN_OK && feeder_rc != PNN_OK && !feeder_err.empty()) return fail(c, feeder_rc, "%s", feeder_err.c_str());
        return code;
    };
    if ((rc = pass(0))) return finish(rc);
    for (long i = 0; i < nsl; i++) {
        if (i + 1 < nsl && (rc = pass(i + 1))) return finish(rc);
        if ((rc = copy_out(i))) return finish(rc);
    }
    if ((rc = finish(PNN_OK))) return rc;
    if (c->opt_precision == 1 && *c->h_range) {
        // a slice left the f16 range of the split-precision kernels: the whole call again on the exact-f32 kernels (what the one-pass form
        // does for batches above 256 blocks), sequentially
        *c->h_range = 0;
        c->range_fallbacks++;
        const long keep = c->opt_precision, keep_slice = c->opt_host_slice;
        c->opt_precision = 0; c->opt_host_slice = -1;
        rc = host_predict(c, m, above, left, n, out, dst, dst_stride);
        c->opt_precision = keep; c->opt_host_slice = keep_slice;
        return rc;
    }
    return PNN_OK;
}

static int host_predict(pnn_ctx* c, Model* m, const float* above, const float* left, int n, float* out, int32_t* dst,
                        int dst_stride)
{
    const int w = m->width;
    const long w2 = (long)w * w;
    if (n < 0 || (n > 0 && (!above || (!out && !dst)))) return fail(c, PNN_E_ARG, "bad buffers / batch size");
    if (n == 0) return PNN_OK;
    if (!m->is_fc && !left) return fail(c, PNN_E_ARG, "`left` is NULL for a convolutional model");
    {
        // A call of several passes' worth of blocks (the reference's batched driver slices N into many batches, pnn/batching.py:7-88) runs
        // slice by slice with the neighbouring slices' copies -- and their non-finite scans -- beside each pass (round 6, host_predict_sliced)
        const long slice = c->opt_host_slice > 0 ? c->opt_host_slice : std::max(64L, std::min(4096L, 262144L / w2));   // the bench batches: 4096 / 4096 / 1024 / 256 / 64
        if (c->opt_host_slice >= 0 && !(n == 1 && c->opt_cache_mb > 0) && (long)n >= 2 * slice) return host_predict_sliced(c, m, above, left, n, out, dst, dst_stride, slice);
    }
    {   // the range guard's v_max_f32 drops NaN operands: non-finite inputs are refused here (a few hundred floats per block)
        const size_t ca = (size_t)n * (m->is_fc ? 5 : 3) * w2, cl = m->is_fc ? 0 : (size_t)n * 2 * w2;
        if (!all_finite(above, ca) || !all_finite(left, cl)) return fail(c, PNN_E_ARG, "non-finite value in the input contexts");
    }
    // ---- prediction cache (single-block calls only) ----
    pnn_ctx::CacheEntry* slot = nullptr;
    uint64_t hash = 0;
    const size_t na = (size_t)(m->is_fc ? 5 : 3) * w2, nl = m->is_fc ? 0 : (size_t)2 * w2;
    if (n == 1 && c->opt_cache_mb > 0) {
        const int wi = width_index(w);
        auto& table = c->cache[wi];
        if (table.empty()) {
            const size_t entry = (na + nl + 2 * w2) * 4 + 64;
            table.resize(std::max<size_t>(16, ((size_t)c->opt_cache_mb << 20) / 5 / entry));
        }
        hash = hash_bytes(above, na * 4);
        if (nl) hash = hash_bytes(left, nl * 4, hash);
        slot = &table[hash % table.size()];
        if (slot->valid && slot->hash == hash && !memcmp(slot->in.data(), above, na * 4) && (!nl || !memcmp(slot->in.data() + na, left, nl * 4))) {
            c->cache_hits++;
            if (out) memcpy(out, slot->out.data(), w2 * 4);
            if (dst)
                for (int y = 0; y < w; y++) memcpy(dst + (size_t)y * dst_stride, slot->pel.data() + (size_t)y * w, (size_t)w * 4);
            return PNN_OK;
        }
        c->cache_misses++;
    }
    HIPCHK(c, hipSetDevice(c->device));
    reset_stats(c);
    int rc;
    hipStream_t s = c->stream;
    const size_t in_a = (size_t)n * (m->is_fc ? 5 : 3) * w2 * 4, in_l = m->is_fc ? 0 : (size_t)n * 2 * w2 * 4;
    if ((rc = dev_reserve(c, c->stage_in[0], in_a))) return rc;
    if (in_l && (rc = dev_reserve(c, c->stage_in[1], in_l))) return rc;
    if ((rc = dev_reserve(c, c->stage_out[0], (size_t)n * w2 * 4))) return rc;
    if (dst && (rc = dev_reserve(c, c->stage_out[1], (size_t)n * w2 * 4))) return rc;
    // Single-block calls (what HM issues) and the batching service's handfuls: no copy engine at all -- the first kernel reads
    // the inputs from pinned host memory and the last one writes the results there (two launches and ~12 us less per call
    // than H2D + D2H copies; at these sizes the PCIe reads hide under the weight stream).
    constexpr size_t kPinIn = 64 << 10, kPinOut = 64 << 10;
    if (in_a <= kPinIn && in_l <= kPinIn && (size_t)n * w2 * 4 <= kPinOut) {
        if (!c->h_pin) { PNN_UNSAFE_CALLS_GUARD; HIPCHK(c, hipHostMalloc((void**)&c->h_pin, 2 * kPinIn + 2 * kPinOut, hipHostMallocDefault)); }
        char* hp = c->h_pin;
        memcpy(hp, above, in_a);
        if (in_l) memcpy(hp + kPinIn, left, in_l);
        float* p_out = (float*)(hp + 2 * kPinIn);
        int32_t* p_dst = (int32_t*)(hp + 2 * kPinIn + kPinOut);
        const long pa = m->is_fc ? 5 * w2 : 3 * w2;
        bool inline_input = true;                      // (not in a captured chain: the argument block is part of the graph)
        auto pass = [&](long b0, long nb) {
            c->host_input = (m->is_fc && inline_input) ? above + b0 * pa : nullptr;   // small inputs ride in the first kernel's argument block
            const int r = run_net(c, m, (const float*)hp + b0 * pa, pa, (const float*)(hp + kPinIn) + b0 * 2 * w2, 2 * w2, nb, p_out + b0 * w2,
                                  (dst || slot) ? p_dst + b0 * w2 : nullptr, s);
            c->host_input = nullptr;
            return r;
        };
        static const bool host_trace = getenv("PNN_HOST_TRACE") != nullptr;   // diagnostic: where a single-block call spends its time
        timespec ht0, ht1, ht2;
        if (host_trace) clock_gettime(CLOCK_MONOTONIC, &ht0);
        c->done_want = c->opt_flag_wait != 0;
        c->done_armed = false;
        // The chain as a graph (pnn_ctx::GraphEntry): first call of a shape as ever (it sizes the workspaces), second call captured, then replays.
        pnn_ctx::GraphEntry* ge = nullptr;
        if (c->opt_graphs && n <= 64 && !c->opt_time_launches && !host_trace) ge = &c->graphs[std::make_tuple((const void*)m, n, (dst || slot) ? 1 : 0)];
        bool replay = false;
        if (ge && !ge->failed && !ge->exec && ge->uses >= 1) {
            inline_input = false;
            hipGraph_t g = nullptr;
            // The stream must be IDLE when the capture begins: a host call returns when its last kernel has raised the completion flag,
            // which is before the runtime has retired that kernel (or the previous shape's graph launch) -- and a capture begun in that
            // window came back "invalidated" once in a few hundred HM encodes (tests/test_hm.py under PNN_GRAPHS=1), leaving the stream
            // unusable for plain launches too.
            (void)hipStreamSynchronize(s);
            if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
            // (thread-local mode; the other threads' allocations and copies are kept out by the lock, see unsafe_calls_lock)
            std::lock_guard<std::recursive_mutex> capture_guard(unsafe_calls_lock());
            hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                rc = pass(0, n);                          // nothing runs: the launches are recorded
                e = hipStreamEndCapture(s, &g);
                if (rc == PNN_OK && e == hipSuccess && g) e = hipGraphInstantiate(&ge->exec, g, nullptr, nullptr, 0);
                if (g) (void)hipGraphDestroy(g);
            }
            if (rc != PNN_OK || e != hipSuccess || !ge->exec) {   // this shape stays on plain launches
                static const bool debug = getenv("PNN_DEBUG") != nullptr;
                if (debug) fprintf(stderr, "[pnn] capture of the launch chain failed (width %d, %d blocks): %s / %s -- plain launches for this shape\n", w, n,
                                   rc != PNN_OK ? c->err.c_str() : "pass ok", hipGetErrorString(e));
                // whatever went wrong must not leave a stream in capture mode: the plain pass below runs on them
                for (hipStream_t st : {s, c->side_stream}) {
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    if (st && hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { hipGraph_t junk = nullptr; (void)hipStreamEndCapture(st, &junk); if (junk) (void)hipGraphDestroy(junk); }
                }
                (void)hipGetLastError();
                if (ge->exec) { (void)hipGraphExecDestroy(ge->exec); ge->exec = nullptr; }
                ge->failed = true;
                inline_input = true;
                c->done_armed = false;
            } else {
                ge->armed = c->done_armed; ge->seq = c->done_seq; ge->nflags = c->done_nflags;
                ge->stat_gemm_launches = c->stat_gemm_launches; ge->stat_launches = c->stat_launches;
                ge->stat_gemm_flops = c->stat_gemm_flops; ge->stat_gemm_flops_skipped = c->stat_gemm_flops_skipped;
            }
            rc = PNN_OK;
        }
        if (ge && ge->exec) {
            if (ge->armed) {                          // the number this chain raises may still stand there
                __atomic_store_n(reinterpret_cast<unsigned*>(c->h_range) + 1, 0u, __ATOMIC_RELEASE);
                for (int i = 0; i < ge->nflags; i++) __atomic_store_n(reinterpret_cast<unsigned*>(c->h_range) + pnn_ctx::kDoneFlag0 + i, 0u, __ATOMIC_RELEASE);
            }
            HIPCHK(c, hipGraphLaunch(ge->exec, s));
            c->done_armed = ge->armed; c->done_seq = ge->seq; c->done_nflags = ge->nflags;
            c->stat_gemm_launches = ge->stat_gemm_launches; c->stat_launches = ge->stat_launches;
            c->stat_gemm_flops = ge->stat_gemm_flops; c->stat_gemm_flops_skipped = ge->stat_gemm_flops_skipped;
            replay = true;
        }
#ifdef PNN_F32_DIAG                                 // diagnostic library only: PNN_B1_STAMPS=<k> prints the device-side timeline of this context's k-th small call
        static const long stamp_call = getenv("PNN_B1_STAMPS") ? atol(getenv("PNN_B1_STAMPS")) : 0;
        static thread_local long stamp_calls = 0;
        const bool stamping = stamp_call > 0 && ++stamp_calls == stamp_call && !replay;
        if (stamping) {
            const size_t bytes = (size_t)pnn_ctx::kDiagLaunches * pnn_ctx::kDiagWgs * 64;
            if (!c->diag_stamps) HIPCHK(c, hipMalloc(&c->diag_stamps, bytes));
            HIPCHK(c, hipMemset(c->diag_stamps, 0, bytes));
            HIPCHK(c, hipDeviceSynchronize());
            c->diag_launch = 0; c->diag_names.clear(); c->diag_wgs.clear(); c->diag_k.clear();
        }
        void* const stamps_keep = c->diag_stamps;
        if (!stamping) c->diag_stamps = nullptr;          // diag_stamp_slot hands out slots only during the stamped call
#endif
        if (!replay) {
            reset_stats(c);
            rc = pass(0, n);
            if (ge) ge->uses++;
        }
#ifdef PNN_F32_DIAG
        c->diag_stamps = stamps_keep;
        if (stamping && rc == PNN_OK) {
            timespec tw0, tw1;
            clock_gettime(CLOCK_MONOTONIC, &tw0);
            const int wrc = c->done_armed ? wait_done_flag(c, s, n, w) : wait_stream(c, s);
            clock_gettime(CLOCK_MONOTONIC, &tw1);
            if (wrc) return wrc;
            HIPCHK(c, hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)c->diag_launch * pnn_ctx::kDiagWgs * 8);
            HIPCHK(c, hipMemcpy(h.data(), c->diag_stamps, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t00 = ~0ull;
            for (int l = 0; l < c->diag_launch; l++) for (int i = 0; i < c->diag_wgs[l]; i++) { const unsigned long long e = h[((size_t)l * pnn_ctx::kDiagWgs + i) * 8 + 4]; if (e && e < t00) t00 = e; }
            fprintf(stderr, "[pnn-stamps] width %d, %d block(s): %d stamped launches (us from the first workgroup's entry; per launch: first / median / last over its workgroups); host wait after the last launch call %.1f us\n",
                    w, n, c->diag_launch, (tw1.tv_sec - tw0.tv_sec) * 1e6 + (tw1.tv_nsec - tw0.tv_nsec) * 1e-3);
            for (int l = 0; l < c->diag_launch; l++) {
                std::vector<double> ent, ls, le, ex, stv;
                for (int i = 0; i < c->diag_wgs[l]; i++) {
                    const unsigned long long* d = &h[((size_t)l * pnn_ctx::kDiagWgs + i) * 8];
                    if (!d[4]) continue;
                    ent.push_back((double)(d[4] - t00) / 100.0);
                    if (d[3]) { ls.push_back((double)(d[3] - t00) / 100.0); le.push_back((double)(d[3] + d[1] - t00) / 100.0); }
                    if (d[5]) ex.push_back((double)(d[5] - t00) / 100.0);
                    if (d[6]) stv.push_back((double)(d[6] - t00) / 100.0);
                }
                auto fml = [](std::vector<double>& v, char* buf) { if (v.empty()) { snprintf(buf, 64, "      -      "); return; } std::sort(v.begin(), v.end()); snprintf(buf, 64, "%5.1f /%5.1f /%5.1f", v.front(), v[v.size() / 2], v.back()); };
                char b0[64], b1[64], b2[64], b3[64], b4[64];
                fml(ent, b0); fml(ls, b1); fml(le, b2); fml(ex, b3); fml(stv, b4);
                fprintf(stderr, "[pnn-stamps]  %-28s K %5.0f %4d WGs (%zu stamped) | entry %s | chain start %s | chain end %s | exit %s%s%s\n", c->diag_names[l].c_str(), c->diag_k[l], c->diag_wgs[l], ent.size(), b0, b1, b2, b3,
                        stv.empty() ? "" : " | results stored (before the completion signal) ", stv.empty() ? "" : b4);
            }
        }
#endif
        inline_input = true;                          // (the range fallback's passes are plain launches)
        c->done_want = false;
        if (rc) return rc;
        if (host_trace) clock_gettime(CLOCK_MONOTONIC, &ht1);
        if ((rc = c->done_armed ? wait_done_flag(c, s, n, w) : wait_stream(c, s))) return rc;
        if (host_trace) {
            clock_gettime(CLOCK_MONOTONIC, &ht2);
            static double s_launch = 0, s_wait = 0; static long s_n = 0;
            s_launch += (ht1.tv_sec - ht0.tv_sec) * 1e6 + (ht1.tv_nsec - ht0.tv_nsec) * 1e-3;
            s_wait += (ht2.tv_sec - ht1.tv_sec) * 1e6 + (ht2.tv_nsec - ht1.tv_nsec) * 1e-3;
            if (++s_n % 1000 == 0) { fprintf(stderr, "[pnn-host] width %d: %d launches enqueued in %.1f us, then %.1f us until the stream is idle (mean of 1000 calls)\n", w, c->stat_launches, s_launch / 1000, s_wait / 1000); s_launch = s_wait = 0; }
        }
        if (*c->h_range && (rc = range_fallback(c, n, pass, s))) return rc;
        if (out) memcpy(out, p_out, (size_t)n * w2 * 4);
        if (dst) {
            if (dst_stride == w) memcpy(dst, p_dst, (size_t)n * w2 * 4);
            else for (int y = 0; y < w; y++) memcpy(dst + (size_t)y * dst_stride, p_dst + (size_t)y * w, (size_t)w * 4);   // n == 1 (checked by the caller)
        }
        if (slot) {
            slot->in.resize(na + nl);
            memcpy(slot->in.data(), above, na * 4);
            if (nl) memcpy(slot->in.data() + na, left, nl * 4);
            slot->out.assign(p_out, p_out + w2);
            slot->pel.assign(p_dst, p_dst + w2);
            slot->hash = hash; slot->valid = true;
        }
        return PNN_OK;
    }
    // One copy in, one pass, one copy out -- for ONE pass's worth of blocks.  Streaming a bench batch through in chunks (staging copy,
    // H2D, compute and D2H of successive chunks overlapped on three streams) was built in round 5 and measured SLOWER at every bench batch
    // (FC 8x8 x 4096: 0.58 against 0.48 ms, conv 16x16 x 1024: 1.23 against 1.08): a quarter or half of a bench batch does not fill the
    // chip -- FC 8x8 at 4096 is exactly one workgroup per CU, so half of it takes as long as all of it (profiles/r05_host_rate.txt).
    // (Calls of two slices' worth or more never get here: host_predict_sliced, above.)
    HIPCHK(c, hipMemcpyAsync(c->stage_in[0].p, above, in_a, hipMemcpyHostToDevice, s));
    if (in_l) HIPCHK(c, hipMemcpyAsync(c->stage_in[1].p, left, in_l, hipMemcpyHostToDevice, s));
    if (slot && !dst && (rc = dev_reserve(c, c->stage_out[1], (size_t)n * w2 * 4))) return rc;   // a cached entry serves both result kinds
    float* d_out = (float*)c->stage_out[0].p;
    int32_t* d_dst = (dst || slot) ? (int32_t*)c->stage_out[1].p : nullptr;
    rc = run_net(c, m, (const float*)c->stage_in[0].p, m->is_fc ? 5 * w2 : 3 * w2, (const float*)c->stage_in[1].p, 2 * w2, n,
                 d_out, d_dst, s);
    if (rc) return rc;
    if (c->opt_precision == 1) {                      // the guard costs one synchronisation; the copies below wait for the stream anyway
        HIPCHK(c, hipStreamSynchronize(s));
        if (*c->h_range) {
            const long pa = m->is_fc ? 5 * w2 : 3 * w2;
            auto pass = [&](long b0, long nb) {
                return run_net(c, m, (const float*)c->stage_in[0].p + b0 * pa, pa, (const float*)c->stage_in[1].p + b0 * 2 * w2, 2 * w2, nb,
                               d_out + b0 * w2, d_dst ? d_dst + b0 * w2 : nullptr, s);
            };
            if ((rc = range_fallback(c, n, pass, s))) return rc;
        }
    }
    if (out) HIPCHK(c, hipMemcpyAsync(out, d_out, (size_t)n * w2 * 4, hipMemcpyDeviceToHost, s));
    if (dst) {
        if (dst_stride == w) HIPCHK(c, hipMemcpyAsync(dst, d_dst, (size_t)n * w2 * 4, hipMemcpyDeviceToHost, s));
        else HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)dst_stride * 4, d_dst, (size_t)w * 4, (size_t)w * 4, (size_t)n * w,
                                        hipMemcpyDeviceToHost, s));
    }
    if (slot) {
        slot->valid = false;
        slot->in.resize(na + nl); slot->out.resize(w2); slot->pel.resize(w2);
        memcpy(slot->in.data(), above, na * 4);
        if (nl) memcpy(slot->in.data() + na, left, nl * 4);
        HIPCHK(c, hipMemcpyAsync(slot->out.data(), d_out, w2 * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(slot->pel.data(), d_dst, w2 * 4, hipMemcpyDeviceToHost, s));
    }
    HIPCHK(c, hipStreamSynchronize(s));
    if (slot) { slot->hash = hash; slot->valid = true; }
    return PNN_OK;
}

// `want` streams of the current device that sit on `want` DIFFERENT hardware queues (see probe_queue_shared, pnn_small.hip): streams are
// created until enough pairwise-separate ones exist (at most 16 tries), the others are destroyed again.  Returns how many were found
// (< want when the runtime has fewer queues); the caller hands them to contexts (option "stream") and releases them afterwards.
int pnn_streams_on_distinct_queues(void** out, int want)
{
    if (!out || want < 1 || want > 8) return PNN_E_ARG;
    std::vector<hipStream_t> keep, drop;
    for (int tries = 0; tries < 16 && (int)keep.size() < want; tries++) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
        bool clash = false;
        for (hipStream_t k : keep) {
            bool shared = false;
            if (probe_queue_shared(k, s, &shared) != hipSuccess) { clash = true; break; }
            if (shared) { clash = true; break; }
        }
        (clash ? drop : keep).push_back(s);
    }
    for (hipStream_t s : drop) (void)hipStreamDestroy(s);
    for (size_t i = 0; i < keep.size(); i++) out[i] = keep[i];
    return (int)keep.size();
}

void pnn_streams_release(void** streams, int n)
{
    for (int i = 0; streams && i < n; i++) if (streams[i]) { (void)hipStreamSynchronize((hipStream_t)streams[i]); (void)hipStreamDestroy((hipStream_t)streams[i]); }
}

int pnn_host_alloc(void** out, size_t bytes)
{
    PNN_UNSAFE_CALLS_GUARD;
    if (!out || !bytes) return PNN_E_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, PNN_E_NOMEM, "hipHostMalloc(%zu) failed", bytes); }
    return PNN_OK;
}

void pnn_host_free(void* p)
{
    PNN_UNSAFE_CALLS_GUARD;
    if (p) (void)hipHostFree(p);
}

int pnn_check_range(pnn_ctx* c, void* stream, long* host_fallbacks)
{
    if (!c) return PNN_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize((hipStream_t)stream));
    if (host_fallbacks) *host_fallbacks = c->range_fallbacks;
    return pending_range_error(c);
}

int pnn_cache_stats(pnn_ctx* c, long* hits, long* misses)
{
    if (!c) return PNN_E_ARG;
    if (hits) *hits = c->cache_hits;
    if (misses) *misses = c->cache_misses;
    return PNN_OK;
}

int pnn_predict_fc(pnn_ctx* c, int width, const float* context, int n, float* out)
{
    int rc;
    Model* m = model_for(c, width, 1, &rc);
    if (!m) return rc;
    if (!out) return fail(c, PNN_E_ARG, "`out` is NULL");
    return host_predict(c, m, context, nullptr, n, out, nullptr, 0);
}

int pnn_predict_conv(pnn_ctx* c, int width, const float* above, const float* left, int n, float* out)
{
    int rc;
    Model* m = model_for(c, width, 0, &rc);
    if (!m) return rc;
    if (!out) return fail(c, PNN_E_ARG, "`out` is NULL");
    return host_predict(c, m, above, left, n, out, nullptr, 0);
}

int pnn_predict_pel(pnn_ctx* c, int width, const float* above, const float* left, int n, int32_t* dst, int dst_stride)
{
    int rc;
    Model* m = model_for(c, width, -1, &rc);
    if (!m) return rc;
    if (!dst || dst_stride < width) return fail(c, PNN_E_ARG, "bad destination");
    if (n > 1 && dst_stride != width) return fail(c, PNN_E_ARG, "strided destination needs n == 1");
    return host_predict(c, m, above, left, n, nullptr, dst, dst_stride);
}

int pnn_predict_f32_pel(pnn_ctx* c, int width, const float* above, const float* left, int n, float* out, int32_t* dst)
{
    int rc;
    Model* m = model_for(c, width, -1, &rc);
    if (!m) return rc;
    if (!out && !dst) return fail(c, PNN_E_ARG, "`out` and `dst` are both NULL");
    return host_predict(c, m, above, left, n, out, dst, width);
}

}  // extern "C"
