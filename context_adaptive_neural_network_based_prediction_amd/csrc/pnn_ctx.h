// Internal types of the C-ABI layer, shared by its translation units (not part of the public boundary, include/pnn_hip.h):
//   pnn_model.cpp   weight pre-packing and the per-architecture layer tables (built once per model load)
//   pnn_tiles.cpp   rule-based choice of the kernel family / tile configuration of a tap GEMM
//   pnn_tuner.cpp   on-device choice among the legal configurations, remembered per (layer, M)
//   pnn_passes.cpp  the launch sequences of one pass of the fully-connected / convolutional nets
//   pnn_abi.cpp     the extern "C" entry points, contexts, staging and the prediction cache
#pragma once
#include "../../include/pnn_hip.h"
#include "pnn_kernels.h"
#include "pnn_host.h"

#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <string>
#include <vector>

namespace pnn {

constexpr int kHidden = 1200;                         // pnn/components.py:130-160
// The output layer of an FC net with <= 64 outputs is summed in K segments of 10 chunks (160 hidden units) whose partial
// sums are then added in ascending order (fuse_reduce_kernel): the order the ring kernel's fused output layer produces
// with its 128 x 160 tile, and the order tapgemm_small_kernel's K-segment mode reproduces at any batch size.
constexpr int kFuseSegChunks = 10;
inline int strides_for(int w, int* st)                       // pnn/PredictionNeuralNetwork.py:126-132
{
    switch (w) {
    case 4: st[0] = 1; st[1] = 1; return 2;
    case 8: st[0] = 2; st[1] = 1; return 2;
    case 16: st[0] = 2; st[1] = 1; st[2] = 2; st[3] = 1; return 4;
    case 32: st[0] = 2; st[1] = 2; st[2] = 1; st[3] = 2; st[4] = 1; return 5;
    case 64: st[0] = 2; st[1] = 2; st[2] = 2; st[3] = 2; st[4] = 1; return 5;
    default: return -1;
    }
}

inline int width_index(int w)                                // TComPrediction.cpp:564: log2(w) - 2
{
    switch (w) { case 4: return 0; case 8: return 1; case 16: return 2; case 32: return 3; case 64: return 4; default: return -1; }
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct GemmLayer {                                    // one tap-GEMM launch (all classes)
    TapGemmParams proto{};
    float* d_w = nullptr;
    float* d_w_ch = nullptr;                          // the f32 weights in the lane order of tapgemm_f32_small_kernel (pack_kn_chain, pnn_model.cpp)
    float* d_w_sp = nullptr;                          // split-precision pack: f16 hi/lo, pre-scaled by 2^sp_shift
    float sp_inv_scale = 1.f;
    float* d_bias = nullptr;
    double k_total = 0;                               // sum over classes of taps * Cin
    // Exact-f32 path: the canonical per-output summation order of a DEEP convolution layer is a sum of nseg K segments (each
    // class's taps dealt in order over the segments; within a segment the sequential chain of tapgemm_f32_kernel), added in
    // order, then bias and activation.  A property of the layer (pnn_model.cpp), the same at every batch size and tile.
    int nseg = 1;
    // ... and of a ONE-TAP (FC) layer deeper than kFcSegChunks chunks (round 6): segments of fc_seg_chunks 16-deep chunks of its K, added
    // in order inside the workgroup that computes them (never planes); 0 = the layer is one chain
    int fc_seg_chunks = 0;
    long out_per_block = 0;                           // output floats per block
};
struct Conv1Layer { Conv1Params proto{}; float* d_w = nullptr; float* d_w_sp = nullptr; float sp_inv_scale = 1.f; int npad = 0; float* d_bias = nullptr; long out_per_block = 0; };
struct TConv1Layer { TConv1Params proto{}; float* d_w = nullptr; };
struct MergerLayer { MergerParams proto{}; float* d_w = nullptr; float* d_bias = nullptr; };

struct Model {
    int width = 0;
    bool is_fc = false;
    long n_params = 0;
    int n_layers = 0;
    std::vector<GemmLayer> fc;                        // 4 layers
    Conv1Layer first[2];                              // branch_above / branch_left conv 0
    std::vector<GemmLayer> branch[2];                 // conv 1..L-1
    MergerLayer merger;
    std::vector<GemmLayer> tconv;                     // tconv 0..L-2
    TConv1Layer last;
    int C = 0;                                        // channels at the merger
    long pmax = 0;                                    // largest intermediate activation (floats / block)
    std::vector<void*> allocs;
};

}  // namespace pnn

struct pnn_ctx {
    int device = 0;
    float mean = 0.f;
    hipStream_t stream = nullptr;
    pnn::Model* models[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    pnn::DevBuf ws[6];                                     // P0, P1, F0, F1 (FC uses P0, P1); P2, P3: the left branch's own pair when the branches overlap
    // Small conv passes (the in-loop single-block calls): the two branches are independent chains of 4-5 launches that
    // each fill a fraction of the chip; the left branch runs on a side stream, forked and joined by events.
    bool stream_owned = true;                         // false: adopted through the "stream" option (the caller destroys it)
    long opt_branch_streams = 1;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    pnn::DevBuf stage_in[2], stage_out[2], stage_tbs;
    // Host calls of several passes' worth of blocks (pnn_predict_fc / _conv / _pel with N >= 2 slices; the reference's batched driver,
    // pnn/batching.py:7-88): a second staging set, two copy streams and their events -- slice i + 1 copied in and slice i - 1 copied out
    // while slice i computes (host_predict_sliced, pnn_abi.cpp).  host_slice: blocks per slice, 0 = the bench batch of the width
    // (4096 / 4096 / 1024 / 256 / 64), -1 = never slice (one copy in, passes, one copy out)
    pnn::DevBuf stage2_in[2], stage2_out[2];
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_pass[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    long opt_host_slice = 0;
    pnn::DevBuf seg_part[2];                               // K-segment partial sums of the exact-f32 conv layers: main stream / side stream
    // Prediction cache for the in-loop (n == 1) host calls: HM evaluates the same TB with the same context several
    // times during rate-distortion search.  Direct-mapped per width, exact match on the input bytes.
    struct CacheEntry { uint64_t hash = 0; bool valid = false; std::vector<float> in, out; std::vector<int32_t> pel; };
    std::vector<CacheEntry> cache[5];
    long opt_cache_mb = 0;                            // 0 = off
    long cache_hits = 0, cache_misses = 0;
    char* h_pin = nullptr;                            // pinned, device-visible staging of the single-block host calls (zero-copy)
    // The launch chain of a small host call as a hipGraph (option "graphs", OFF by default: see below): a single-block call is 4 (FC) to 20 (conv
    // 64x64) dependent launches that differ from call to call only in the bytes of the pinned staging -- the third call of a shape
    // (model, blocks, which results) replays the chain the second one captured with ONE hipGraphLaunch: 4-9 us of host time instead of
    // 3.4-4 us per launch (tools/corun_noise.hip: 11 dependent launches 37 -> 9 us inside the launch calls, 51 -> 34 us until complete),
    // and one submission instead of 4-20 for the other streams' launches to contend with.  Same kernels, same arguments: same bits.
    struct GraphEntry {
        hipGraphExec_t exec = nullptr;
        int uses = 0;                                 // calls of this shape so far (0: run and size the buffers, 1: capture, then replay)
        bool failed = false, armed = false;
        unsigned seq = 0;                             // the completion number its last kernel raises
        int nflags = 0;                               // ... in that many per-workgroup flag words (0: the one word)
        int stat_gemm_launches = 0, stat_launches = 0;
        double stat_gemm_flops = 0, stat_gemm_flops_skipped = 0;
    };
    std::map<std::tuple<const void*, int, int>, GraphEntry> graphs;
    // Off by default: the gain is 2-5 us of a 38-235 us call for a thread that calls alone and nothing behind the batching service, and
    // in this runtime a capture is INVALIDATED -- and its stream left unusable -- when any other thread of the process allocates, frees or
    // copies synchronously meanwhile, whatever the capture mode (the reference's HM loads its graphs on threads of their own: one HM run in
    // five failed).  The library's own such calls take a process-wide lock that a capture holds (unsafe_calls_lock, pnn_abi.cpp: 80 HM
    // runs without a failure); a host application's own HIP calls on other threads are not covered, hence opt-in.
    long opt_graphs = 0;
    void* d_zero = nullptr;                           // 4 KiB of zeros: padding source of the LDS-DMA ring GEMM
    long opt_max_chunk = 0;
    // One per-output summation order at every batch size, on either arithmetic: a block's prediction does not depend on the batch it
    // travels in (encoder behind the batching service, decoder alone: no drift).  (Until round 4 an option, canonical_order = 0, let
    // small passes take split-K kernels with another order; those kernels are gone.)
    // 0 (default since round 5): the reference's arithmetic, IEEE float32 products and sums on the f32 matrix instructions
    // (Session::Run in float32, TComPrediction.cpp:572-579,601-608); 1: split f16 (3 x f16 MFMA per product, f32-class accuracy,
    // 2.3-2.7 x the blocks/s at batch) -- an encoder and its decoder must run on the same one
    long opt_precision = 0;
    long opt_sp_cfg = -1;
    // exact-f32 launches of few output tiles (the in-loop single-block calls, the service's handfuls): tapgemm_f32_small_kernel, the same
    // fmaf chain on the 16x16x4 instruction (10 instead of 32 cycles per k of the dependent chain), pnn_gemm_f32_small.hip
    long opt_f32_small = 1;
    long opt_fc_out_f32 = 1;                          // 1: exact-f32 FC passes of <= 512 blocks run the output layer's K segments and their reduction as ONE launch (2: its round-5 form on the 32x32x2 instruction)
    // 1: the K segments of a layer that runs on the small exact-f32 kernel are added up inside its launch (the last workgroup of a tile
    // to arrive, see tapgemm_f32_small_body) instead of by a seg_reduce launch behind it: the same additions in the same order
    long opt_seg_fold = 1;
    // The small exact-f32 kernel's weight ring, 6 or 12 stages ahead of the chain (pnn_gemm_f32_small.hip): 0 = always 6; 1 (default) = 12
    // for the FC layers; 2 = 12 for every launch of at most one workgroup per CU (the batching service's contexts: inside a campaign the
    // weights come from the MALL / HBM, not from L2, and the short ring does not cover that latency)
    long opt_f32_small_deep = 1;
    unsigned* d_seg_cnt = nullptr;                    // the tiles' arrival counters: [2 branches][kSegCntTiles], zero between launches
    // ... which only the LAST workgroup of a tile restores: after any HIP failure on this context (a launch that died part-way, a failed
    // capture) they may not be -- and every later folded launch would wait for an arrival count it never sees.  Zeroed again before the next one.
    bool seg_cnt_dirty = false;
    static constexpr int kSegCntTiles = 2048;
    // ... followed by the arrival counters of the small launches' TAILS (SmallTail, pnn_kernels.h): one per tail instance of a launch
    static constexpr int kTailCnt = 2048, kCntWords = 2 * kSegCntTiles + kTailCnt;
    // 1: in small exact-f32 conv passes the merger runs as the tail of the branches' last pair launch and the last transposed convolution
    // as the tail of the GEMM in front of it (two launches less per call); 0: every layer its own launch
    long opt_tails = 1;
    // 1: tensors between two launches of the small exact-f32 kernels travel in chain order (pnn_gemm_f32_small.hip, XCH: one 16-byte
    // LDS-DMA instruction per chunk of activations instead of four 4-byte ones); 0: channel order everywhere.  Same bits.
    long opt_chain_io = 1;
    long opt_f32_small_tiles = 1024;                  // ... "few" = at most this many 16 x 16 tiles
    long opt_f32_overlap = 1;                         // exact-f32 conv passes at batch: the two branches on two streams (see branches_overlap_at_batch)
    long opt_f32_cfg = -1;                            // tuning aid: force this tapgemm_f32 configuration on every layer it is legal for
    long opt_fuse_first = 1;                          // 1: convimg configurations compute a branch's first (Cin = 1) convolution themselves
    long opt_fuse_last = 1;                           // 1: big FC passes run the output layer inside the last hidden layer's ring kernel
    long opt_f32_seg_mode = 0;                        // K-segmented exact-f32 layers: 0 (default) parallel segments + reduce, 1 in sequence inside the workgroups, -1 by cost model / tuner (same bits)
    long opt_f32_persist = -1;                        // exact-f32 conv launches: -1 (default) two persistent workgroups per CU for 2-4 tiles per CU, 0 never, N > 0 always N per CU
    long opt_ring = 1;                                // 1: split GEMMs may use the LDS-DMA ring kernel (pnn_gemm_ring.hip)
    long opt_small = 1;                               // 1: split GEMMs with few output tiles run on tapgemm_small_kernel (one wave per 32 x 32 tile)
    long opt_small_tiles = 512;                       // ... "few" = at most this many tiles (two one-wave workgroups per CU)
    long opt_pair = 1;                                // 1: small conv passes run the same layer of both branches as ONE launch
    // Two round-3 experiments on the single-block call, both bit-identical, both measured WITHOUT gain and therefore off by default
    // (tools/batch1_latency.py, profiles/r03_batch1_latency.txt): the call's time is the chain of dependent
    // launches on the device (~4 us each whatever they do) plus ~10 us of first-launch latency and completion hand-over.
    long opt_fc_out = 0;                              // 1: small FC passes run the <= 64-output layer (K segments + reduction) as ONE launch (5 -> 4 launches)
    long opt_spin_wait = 0;                           // 1: host calls poll the stream (hipStreamQuery) instead of blocking in hipStreamSynchronize
    long opt_convimg = 1;                             // 1: stride/tap layers whose images fit LDS use convimg_sp_kernel
    long opt_autotune = 2;                            // on-device choice of the split-GEMM configuration: 0 never, 1 always, 2 big launches only
    std::map<std::pair<const void*, long>, int> tuned;
    // conv passes at batch: the branches' later layers overlap on two streams (pnn_passes.cpp, "branch_streams") only once a
    // pass of the same (model, blocks) has run on ONE stream without a tuning sweep -- a sweep times launches on its stream and
    // must not have the other branch beside it.  tune_gen counts sweeps and resets of `tuned`; the map holds its value after such a pass
    long tune_gen = 0;
    std::map<std::pair<const void*, long>, long> overlap_ready;
    long opt_time_launches = 0;                       // 1: bracket every tap-GEMM launch with HIP events (bench roofline)
    struct LaunchRec { hipEvent_t e0, e1; int kind; double flops; };
    std::vector<LaunchRec> launch_recs;
    // Range guard of the split-precision path (pnn_device_common.h): kernels raise *h_range (pinned host memory) when a
    // split-f16 activation leaves the f16 range.  Host entry points then repeat the pass on the exact-f32 kernels; device
    // entry points report PNN_E_RANGE at the next call / pnn_check_range.
    int* h_range = nullptr;
    long range_fallbacks = 0;
    const float* host_input = nullptr;                // host_predict: the caller's f32 input rows (FC nets), valid during the call
    // pnn_predict_tbs_device, convolutional nets: when BOTH branches' second layers take the image kernel with the fused first
    // convolution, the context gather is fused in too -- no gather launch; the kernels read the plane through these (valid
    // during the call; plane == NULL: the contexts were gathered into the staging buffer as usual)
    struct LazyGather { const void* plane = nullptr; const void* tbs = nullptr; int pel_bytes = 0, unit = 0; } lazy;
    long opt_fuse_gather = 1;
    long opt_fuse_tail = 1;                           // conv nets: the image kernel of the last 64-channel layer applies the net's last layer too (pnn_convimg_sp.hip)
    // Completion flag of the small host calls (signal_done, pnn_device_common.h): h_range[1] is the flag word, d_done the
    // workgroup counter; done_want = this pass is the last of a host call that will spin on the flag, done_armed = its last
    // kernel took the signal (kernels that cannot -- the exact-f32 FC output layer -- leave it unset: the call then waits for the stream)
    unsigned* d_done = nullptr;
    unsigned done_seq = 0;                            // the sequence number the running / last pass raises
    unsigned done_seq_alloc = 0;                      // numbers handed out so far (a replayed graph raises the number it was captured with)
    bool done_want = false, done_last_chunk = true, done_armed = false;
    int done_nflags = 0;                              // 0: the one flag word h_range[1]; N > 0: the N per-workgroup words from h_range[kDoneFlag0] on (DoneSignal::per_wg)
    static constexpr int kDoneFlag0 = 16, kDoneFlagsMax = 512;
    long opt_flag_wait = 1;
    // The host thread of a small call spins on the completion flag for as long as the device works: 45-400 us of a CPU per call, and the
    // batching service runs five such threads (19 of its 31 CPU-seconds per Kodak-size campaign, round 4).  wait_sleep = 1: the thread
    // SLEEPS for the part of the wait it can predict -- a running mean of this context's waits per batch-size bucket, minus a margin --
    // and spins only for the rest.  Off by default (a stand-alone codec has nothing else to do with its core); the service turns it on.
    long opt_wait_sleep = 0;
    double wait_ema_us[5][12] = {};                   // by width index (a context with a table holds five nets: 40 us and 235 us calls) and floor(log2(blocks))
    long opt_ring_pm = 1;                             // ring kernel: position-major tiles that skip the taps in the padding (pnn_gemm_ring.hip)
    size_t ws_cap_bytes = (size_t)8 << 30;
    // diagnostic library only (make diag, PNN_B1_STAMPS=<call number>): per-workgroup 100 MHz stamps of the kernels of one small host call
    static constexpr int kDiagLaunches = 32, kDiagWgs = 2048;
    void* diag_stamps = nullptr;
    int diag_launch = 0;
    std::vector<std::string> diag_names;
    std::vector<int> diag_wgs;
    std::vector<double> diag_k;
    std::string err;
    int stat_gemm_launches = 0, stat_launches = 0;
    double stat_gemm_flops = 0;
    double stat_gemm_flops_skipped = 0;               // ... of which position-major tiles skipped (taps that only meet SAME padding)
};

namespace pnn {

int fail(pnn_ctx* c, int code, const char* fmt, ...);
// process-wide: held by a stream capture from begin to end and by every allocation / free / synchronous copy of the library (pnn_abi.cpp)
std::recursive_mutex& unsafe_calls_lock();

#define HIPCHK(c, expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return pnn::fail((c), PNN_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// pnn_abi.cpp
int dev_reserve(pnn_ctx* c, DevBuf& b, size_t bytes);
// pnn_model.cpp
int build_model(pnn_ctx* c, int width, int is_fc, const float* params, size_t n, Model** out);
void free_model(Model* m);
// pnn_tiles.cpp
int convimg_images(const TapGemmParams& p, const TileCfg& t, bool one_tap);
int choose_cfg_convimg(const TapGemmParams& p, bool one_tap);
bool pnn_ring_few_images(const TapGemmParams& p, long M, double k_total);
int choose_cfg_ring(const TapGemmParams& p, long M, bool one_tap, double k_total, bool fused = false);
int choose_cfg_sp(const pnn_ctx* c, long M, int cout, int ncls, int cin, double k_total);
int choose_cfg_f32(const pnn_ctx* c, long M, int cout, int ncls, int cin, double k_total, bool fused, double* cost_out = nullptr);
// pnn_tuner.cpp
// First sighting of (key, M): every legal configuration code in [0, ncodes) runs the real launch (idempotent) on stream
// `s`, the fastest is remembered in c->tuned and returned in *cfg; later sightings return the remembered code.  `rule` =
// the rule-based choice (kept unless beaten by > 3 %).  PNN_OK, or the error of a failed launch.
// the completion signal for the last kernel of the current pass, or an empty one (see pnn_ctx::done_want)
inline DoneSignal take_done_signal(pnn_ctx* c)
{
    if (!c->done_want || !c->done_last_chunk || !c->d_done) return DoneSignal{nullptr, nullptr, 0, 0};
    c->done_armed = true;
    c->done_nflags = 0;
    c->done_seq = ++c->done_seq_alloc;
    return DoneSignal{c->d_done, reinterpret_cast<unsigned*>(c->h_range) + 1, c->done_seq, 0};
}
// ... with a flag word per workgroup of the last kernel (nwg of them): see DoneSignal::per_wg
inline DoneSignal take_done_signal_per_wg(pnn_ctx* c, int nwg)
{
    if (nwg < 1 || nwg > pnn_ctx::kDoneFlagsMax) return take_done_signal(c);
    if (!c->done_want || !c->done_last_chunk || !c->d_done) return DoneSignal{nullptr, nullptr, 0, 0};
    c->done_armed = true;
    c->done_nflags = nwg;
    c->done_seq = ++c->done_seq_alloc;
    return DoneSignal{nullptr, reinterpret_cast<unsigned*>(c->h_range) + pnn_ctx::kDoneFlag0, c->done_seq, 1};
}
int tuned_cfg(pnn_ctx* c, const void* key, long M, int ncodes, int rule, const std::function<bool(int)>& legal,
              const std::function<hipError_t(int)>& launch, hipStream_t s, int* cfg, float* best_us);
// pnn_passes.cpp
void* diag_stamp_slot(pnn_ctx* c, const char* name, long wgs, double k);
long chunk_blocks(const pnn_ctx* c, const Model* m);
bool pass_uses_split(const pnn_ctx* c, const Model* m, long nb);
bool conv_pass_fuses_first(pnn_ctx* c, Model* m, long nb);
int run_net(pnn_ctx* c, Model* m, const float* d_a, long pitch_a, const float* d_l, long pitch_l, long n, float* d_out,
            int32_t* d_dst, hipStream_t s, bool ctx_is_split = false);

}  // namespace pnn
