/*
 * pnn_hip.h -- C ABI of libpnn_hip.so: the MI355X (gfx950) replacement for the TensorFlow-1
 * frozen-graph inference that the reference's modified HM-16.15 calls per transform block.
 *
 * Boundary replaced (reference file:line):
 *   - tensor allocation      hevc/hm_common/c++/source_common/integration_prediction_neural_network.cpp:3-27
 *   - load_graph(s)          integration_prediction_neural_network.cpp:29-69,
 *                            selection logic hm_16_15_substitution/source/Lib/TLibCommon/TComPrediction.cpp:143-178
 *   - Session::Run + epilogue TComPrediction.cpp:554-635 (substitution), :550-661 (switch)
 *   - extract_context_portions hevc/hm_common/c++/source_common/extraction_context.cpp:3-208
 *   - model table parser      hevc/hm_common/c++/source_common/tools.cpp:52-111
 *   - Python batched driver   pnn/batching.py:7-88 (through the *_device entry points)
 *
 * Conventions: plain C types only; every function returns 0 on success and a negative PNN_E_* code on
 * error (pnn_last_error() gives the text); nothing throws across this boundary; the caller owns all
 * buffers; a context is thread-compatible (one context per thread), which is how the reference uses its
 * sessions (one TComPrediction object per encoder/decoder).
 *
 * Tensors keep the frozen graph's layout: float32, NHWC, mean-subtracted in AND out:
 *   FC   (w = 4, 8[, 16]) : node_flattened_context [N][5w^2] = [above w x 3w | left 2w x w]  -> [N][w][w]
 *   conv (w = 4 .. 64)    : node_portion_above [N][w][3w], node_portion_left [N][2w][w]       -> [N][w][w]
 */
#ifndef PNN_HIP_H
#define PNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNN_OK 0
#define PNN_E_ARG (-1)      /* bad argument (the reference's "return -1") */
#define PNN_E_IO (-2)       /* file missing / malformed */
#define PNN_E_MODEL (-3)    /* no model loaded for that width, or wrong kind */
#define PNN_E_HIP (-4)      /* HIP runtime error */
#define PNN_E_NOMEM (-5)
#define PNN_E_RANGE (-6)    /* an asynchronous split-precision pass left the f16 range: its results are invalid */

typedef struct pnn_ctx pnn_ctx;

/* ---- lifetime / models (replaces create_tensors_* + load_graphs + the mean pickle) ------------------ */

/* Creates a context on HIP device `device` with no model. `mean` is mean_training (117.8952234192841 for
 * luminance, sets/results/training_set/means/luminance/mean_training.pkl). */
int pnn_create_empty(pnn_ctx** out, float mean, int device);

/* As TComPrediction::initTempBuff (TComPrediction.cpp:108-178): parses the `width,is_pair,channel,path`
 * table (delimiters ',' and ';', blank lines ignored, tools.cpp:52-111), picks the `pair` models iff the
 * table lists them and use_pair != 0 (the caller passes qp >= 32), and loads the five luminance models
 * (widths 4, 8 fully-connected; 16, 32, 64 convolutional). Paths are `.pnnw` flat weight files (see
 * INTEGRATION.md); relative paths are resolved against the table's directory, then the working directory. */
int pnn_create(pnn_ctx** out, const char* model_table_path, int use_pair, float mean, int device);

/* Loads one model from a `.pnnw` file (width and kind come from its header). */
int pnn_load_model_file(pnn_ctx* ctx, const char* path);
/* Loads one model from host memory: `params` in the canonical flat order (weights.py:tensor_specs). */
int pnn_load_model_params(pnn_ctx* ctx, int width, int is_fc, const float* params, size_t n_params);

int pnn_model_info(const pnn_ctx* ctx, int width, int* is_fc, int* n_layers, long* n_params);
void pnn_destroy(pnn_ctx* ctx);
const char* pnn_last_error(const pnn_ctx* ctx);     /* ctx may be NULL: last error of a failed create */
float pnn_mean(const pnn_ctx* ctx);

/* Options (name, default, meaning).  None of them changes a result bit unless it says so; most exist for A/B measurements.
 *
 * ARITHMETIC -- the one option an encoder and its decoder must agree on:
 *   "precision"      0  the reference's arithmetic: IEEE float32 products and sums (Session::Run in float32, TComPrediction.cpp:572-579,
 *                       601-608; pnn/components.py:169-176) on the f32 matrix instructions.  Default since round 5.
 *                    1  every f32 product from three f16 MFMAs on hi/lo operand halves (22 significand bits per operand, lo x lo dropped:
 *                       f32-class accuracy, within the same +-1 LSB of the oracle) -- 2.3-2.7 x the blocks/s at batch, 5-25 % less per
 *                       single-block call.  Predictions of the two modes differ in the last float bits, i.e. by one LSB on .5 ties.
 *   Within a mode ONE per-output summation order holds at every batch size and on every kernel: a block's prediction is bit-identical
 *   whether it is predicted alone, in a handful or in any batch ("canonical_order" is kept as a name and accepts only 1).
 *
 * EXACT-F32 KERNELS ("precision" 0; also the range fallback of mode 1):
 *   "f32_small"            1   launches of at most "f32_small_max_tiles" (1024) output tiles of 16 x 16 -- single-block calls, the
 *                              batching service's handfuls -- run on tapgemm_f32_small_kernel (the canonical fmaf chain issued through
 *                              v_mfma_f32_16x16x4_f32, one wave per tile); 0: tapgemm_f32_kernel's 128-row tiles at every size
 *   "fc_out_f32"           1   FC passes of <= 512 blocks: output layer's K segments + their reduction in one launch
 *   "chain_io"             1   tensors between two launches of the small kernels travel with every 16-channel group in the order the 16x16x4
 *                              chain consumes it (one 16-byte LDS-DMA instruction per chunk of activations instead of four 4-byte ones); 0: never
 *   "tails"                1   small exact-f32 conv passes: the merger runs inside the branches' last pair launch (per block and channel group, by the
 *                              last of its five tiles to arrive) and the last transposed convolution inside the launch of the GEMM in front of it (per
 *                              block) -- two launches less per single-block call, the same bits; 0: every layer its own launch
 *   "f32_cfg"             -1   >= 0: force tile code [0, pnn_num_f32_configs()) of tapgemm_f32_kernel on every layer it is legal for
 *   "f32_seg_mode"         0   K segments of the deep convolution layers (> 2304 per output: summed in segments of <= 1600, whole taps,
 *                              added in order): 0 separate workgroups + a reduction launch, 1 in sequence inside the workgroups, -1 by model
 *   "f32_persist"         -1   convolution launches of 2-6 tiles per CU on two persistent workgroups per CU; 0 never; N > 0: N per CU
 *   "f32_overlap"          1   the two branches of a conv pass at batch on two streams
 *   "fuse_last"            1   FC passes of >= 1024 blocks run the <= 64-output layer inside the last hidden layer's kernel (both modes)
 * SPLIT-F16 KERNELS ("precision" 1):
 *   "sp_cfg"              -1   >= 0: force configuration code [0, pnn_num_split_configs()) of the three split-GEMM kernel families
 *   "ring" / "convimg"     1   the LDS-DMA ring kernel / the LDS-resident-image convolution kernel may be chosen
 *   "small"                1   GEMMs of at most "small_max_tiles" (512) tiles of 32 x 32 run on tapgemm_small_kernel (one wave per tile)
 *   "fc_out"               0   1: small FC passes run output-layer segments + reduction as one launch (measured no faster)
 *   "fuse_first"           1   the image kernel computes a branch's first (one-input-channel) convolution itself
 *   "fuse_tail"            1   the image kernel of the last 64-channel layer applies the net's last layer to its output tile
 *   "ring_pm"              1   position-major tiles (skip the taps that only meet SAME padding) where the launch model expects a gain;
 *                              2 wherever possible; 0 never.  Applies to tapgemm_f32_kernel's convolution launches too
 *   "autotune"             2   on first sight of a (layer, batch) pair time the legal configurations on the device and keep the fastest:
 *                              2 only launches >= 4 GFLOP, 1 always, 0 rule-based choice only.  Never while the stream is being captured
 * BOTH:
 *   "host_slice"           0   host-array calls (pnn_predict_fc / _conv / _pel) of at least two slices' worth of blocks run slice by slice on two
 *                              staging sets: slice i + 1 is copied in and slice i - 1 copied out while slice i computes.  0: slices of the bench
 *                              batch of the width (4096 / 4096 / 1024 / 256 / 64 blocks); N > 0: N blocks; -1: never (one copy in, passes, one copy out)
 *   "pair"                 1   small conv passes run layer i of BOTH branches as one launch
 *   "branch_streams"       1   small passes of the 32x32 / 64x64 nets, and passes at batch, run the two branches on two streams; 2: every
 *                              small conv pass too; 0: one stream
 *   "fuse_gather"          1   pnn_predict_tbs_device on a conv net: the first convolution reads the picture plane through the descriptors
 *   "cache_mb"             0   > 0: single-block host calls are answered from a direct-mapped cache of that many MiB when the same input
 *                              bytes were predicted before (HM's RD search repeats itself); dropped on any option / model change
 *   "flag_wait"            1   a small host call ends when its LAST kernel raises a sequence number in pinned host memory (3-7 us earlier
 *                              than the runtime's completion signal); "spin_wait" (0): hipStreamQuery polling instead of hipStreamSynchronize
 *   "stream_priority"      0   < 0 / > 0: the context's own stream (host entry points) at the device's greatest / least priority
 *   "stream"               -   a hipStream_t (cast to long): the context's host entry points run on the caller's stream from now on
 *   "wait_sleep"           0   1: the thread of a small host call sleeps through the predictable part of its wait (running mean per batch
 *                              size, minus a margin) and spins only for the rest: the batching service's workers set it (two thirds of
 *                              their CPU time was that spin); a stand-alone codec keeps 0
 *   "seg_fold"             1   exact f32, small calls: the K segments of a deep layer (32x32 / 64x64 nets) are added up inside the
 *                              layer's launch by the last workgroup of each tile to arrive (planes written through, read back past
 *                              the caches, added in plane order) instead of by a reduction launch behind it: 17 -> 11 / 20 -> 11
 *                              launches per single-block call, the same bits
 *   "f32_small_deep"       1   exact f32, small calls: the weight ring of the small kernel 12 instead of 6 stages ahead of the MFMA chain
 *                              (84 instead of 48 KiB of LDS per workgroup) -- 0: never, 1: for the FC layers, 2: for every launch of at
 *                              most one workgroup per CU (the batching service sets 2: inside a campaign the weights come from the
 *                              MALL / HBM, a 4x4 call 55 -> 47 us; alone nothing changes for the FC nets, conv 16x16 82 -> 87 us)
 *   "graphs"               0   1: small host calls (<= 64 blocks): the launch chain of a shape (model, blocks, result kinds) is captured
 *                              on its second call and replayed with one hipGraphLaunch afterwards -- same kernels, same arguments,
 *                              same bits; a single-block call 1-4 us shorter for a thread that calls alone, nothing behind the
 *                              batching service.  Off by default: in this runtime a capture is invalidated when ANOTHER thread of the
 *                              process allocates / frees / copies synchronously meanwhile; the library's own such calls are
 *                              serialised against captures, a host application's own HIP calls are not
 *   "max_chunk" 0 (blocks per pass, 0 = by workspace), "ws_cap_mb" 8192, "time_launches" 0 (HIP events around every tap-GEMM launch)
 */
int pnn_set_option(pnn_ctx* ctx, const char* name, long value);
/* Environment variables read at pnn_create* (same meaning as the options): PNN_PRECISION, PNN_GRAPHS, PNN_AUTOTUNE, PNN_RING, PNN_CONVIMG,
 * PNN_SMALL, PNN_F32_SMALL, PNN_F32_SMALL_TILES, PNN_CACHE_MB, PNN_FC_OUT, PNN_SPIN_WAIT, PNN_FLAG_WAIT, PNN_FUSE_FIRST, PNN_FUSE_GATHER,
 * PNN_FUSE_TAIL, PNN_FUSE_LAST, PNN_RING_PM, PNN_BRANCH_STREAMS, PNN_MAX_CHUNK, PNN_F32_CFG, PNN_F32_OVERLAP, PNN_F32_SEG_MODE,
 * PNN_F32_PERSIST.  Diagnostics: PNN_DEBUG (kernel choice of every GEMM launch on stderr), PNN_DEBUG_TUNE, PNN_PROFILE (synchronous
 * per-launch timing), PNN_HOST_TRACE, PNN_LIB_PATH (Python loader: another build of the library); diagnostic library of `make diag`
 * only: PNN_SP_DIAG, PNN_F32_DIAG, PNN_F32S_DIAG. */
/* Input-range contract of "precision" 1: operands travel as pairs of f16 values, so every intermediate activation must satisfy
 * |v| < 65504.  8-bit contexts through trained models stay two orders of magnitude below that; arbitrary float inputs or
 * models may not.  The kernels detect a violation (never a silent NaN): host entry points then recompute, on the exact-f32 kernels,
 * exactly the blocks that overflow when predicted alone (batches of <= 256; the other blocks keep the bits they get in any batch) and
 * refuse non-finite inputs (PNN_E_ARG); models with a non-finite parameter are refused at load (PNN_E_MODEL).  Device entry points are
 * asynchronous: the NEXT call on the context fails with PNN_E_RANGE, and pnn_check_range -- which waits for `stream` -- tells right
 * away (*host_fallbacks, optional = how many host calls took the exact-f32 repeat so far).  "precision" 0 has no such bound. */
int pnn_check_range(pnn_ctx* ctx, void* stream, long* host_fallbacks);

/* A short string naming everything that decides the last float bits of this context's predictions -- the arithmetic ("precision"), its
 * per-output summation order and the K-segment layout of the deep exact-f32 layers, the library's order revision.  An encoder and its
 * decoder (or an encoder and the batching service it talks to) produce identical predictions iff their tags are equal: compare them
 * once at start-up. */
int pnn_arithmetic_tag(const pnn_ctx* ctx, char* out, size_t bytes);

/* Number of configuration codes "sp_cfg" accepts (tile shapes of tapgemm_sp_kernel, convimg_sp_kernel, tapgemm_ring_kernel). */
int pnn_num_split_configs(void);
/* Number of configuration codes "f32_cfg" accepts (tiles of tapgemm_f32_kernel; all of them give the same bits). */
int pnn_num_f32_configs(void);
/* Hits / misses of the "cache_mb" prediction cache since the option was last set. */
int pnn_cache_stats(pnn_ctx* ctx, long* hits, long* misses);

/* ---- host-buffer entry points (what the HM side binds; synchronous) ---------------------------------- */

/* == Session::Run({{"node_flattened_context", T}}, {"fully_connected/node_output"}) for N stacked inputs. */
int pnn_predict_fc(pnn_ctx* ctx, int width, const float* context, int n, float* out);
/* == Session::Run({{"node_portion_above", A}, {"node_portion_left", L}}, {".../node_output"}). */
int pnn_predict_conv(pnn_ctx* ctx, int width, const float* above, const float* left, int n, float* out);
/* Either of the above followed by the HM epilogue (TComPrediction.cpp:621-635), written with row stride
 * `dst_stride` into `dst` for n == 1, or densely [n][w][w] when dst_stride == width. For FC models
 * `left` is ignored when it equals above + 3w^2 or is NULL (one flattened buffer, TComPattern.cpp:352-353). */
int pnn_predict_pel(pnn_ctx* ctx, int width, const float* above, const float* left, int n, int32_t* dst,
                    int dst_stride);

/* Batched calls of these entry points (more than 64 KiB of input) copy in, compute, copy out.  From PINNED caller arrays the copies
 * run ~10 % of the call faster (no staging inside the runtime): hipHostMalloc / hipHostRegister, or these two (page-locked,
 * device-visible; any thread may free). */
int pnn_host_alloc(void** out, size_t bytes);
void pnn_host_free(void* p);

/* The runtime deals HIP streams onto a few hardware queues (4 by default); streams that share one run their kernels in submission order,
 * so two threads with a context each may find themselves waiting for each other's whole calls.  pnn_streams_on_distinct_queues creates
 * `want` (<= 8) streams that were MEASURED to sit on different hardware queues and returns how many it found; give one to a context with
 * pnn_set_option(ctx, "stream", (long) stream) -- its host entry points then run there -- and release them after the contexts.
 * (The batching service does this for its width workers: pnn_service_run_table.) */
int pnn_streams_on_distinct_queues(void** out_streams, int want);
void pnn_streams_release(void** streams, int n);

/* Both results of one pass: the float prediction (as pnn_predict_fc / pnn_predict_conv) into `out` [n][w][w] and the
 * HM-epilogue Pel values (as pnn_predict_pel, dense) into `dst` [n][w][w]; either may be NULL. */
int pnn_predict_f32_pel(pnn_ctx* ctx, int width, const float* above, const float* left, int n, float* out, int32_t* dst);

/* == extract_context_portions (extraction_context.cpp:3-208), same argument order and error behaviour
 * (returns -1 on NULL pointers, n_avail <= 0, unavailable corner unit). Pure host code. */
int pnn_extract_context(const int32_t* roi_origin, float* above, float* left, const uint8_t* neighbor_flags,
                        int n_avail, int unit_w, int unit_h, int above_units, int left_units, int tu_w,
                        int tu_h, int pic_stride, float mean);

/* Parses the model table; returns the number of entries written (<= max_entries) or a negative code.
 * paths[i] points into an internal buffer valid until the next call on the same thread. */
int pnn_parse_model_table(const char* path, int* widths, int* is_pair, int* channels, const char** paths,
                          int max_entries);

/* ---- device-resident entry points (batched path, asynchronous on `stream`) --------------------------- */

/* All pointers are device pointers on the context's device; `stream` is a hipStream_t (NULL = HIP's
 * default stream, as everywhere in HIP). Calls only enqueue work; the caller synchronises. */
int pnn_predict_fc_device(pnn_ctx* ctx, int width, const float* d_context, int n, float* d_out, void* stream);
int pnn_predict_conv_device(pnn_ctx* ctx, int width, const float* d_above, const float* d_left, int n,
                            float* d_out, void* stream);

/* One transform block of the batched gather. Build it with pnn_make_tb_desc from HM's neighbour flags. */
typedef struct {
    int64_t origin;       /* element index of the TB's top-left pixel from the plane base pointer */
    int32_t stride;       /* plane row stride in elements */
    uint32_t above_mask;  /* bit u: above / above-right unit u (left to right) is available */
    int32_t left_units;   /* number of available left / below-left units, counted from the top */
    int32_t reserved;
} pnn_tb_dev;

/* Translates HM's (bNeighborFlags, iNumIntraNeighbor) of TComPattern.cpp:260-280 into a descriptor with
 * exactly the semantics of extraction_context.cpp:49-205. Returns -1 where the reference returns -1. */
int pnn_make_tb_desc(pnn_tb_dev* out, int64_t origin, int32_t stride, const uint8_t* neighbor_flags,
                     int n_avail, int above_units, int left_units);

/* Batched gather only: Pel plane (pel_bytes 4 = HM `Pel`/int32, 1 = uint8 image) -> mean-subtracted,
 * masked float portions. For FC layouts pass d_left = d_above + 3w^2 and both pitches = 5w^2. */
int pnn_gather_device(pnn_ctx* ctx, int width, int unit, const void* d_plane, int pel_bytes,
                      const pnn_tb_dev* d_tbs, int n, float* d_above, long pitch_above, float* d_left,
                      long pitch_left, void* stream);

/* The whole hot path for n TBs of one width: gather -> net -> (+mean, clamp, round) -> int32 [n][w][w]
 * (and, when d_out_f32 != NULL, the raw float prediction as the frozen graph returns it). */
int pnn_predict_tbs_device(pnn_ctx* ctx, int width, const void* d_plane, int pel_bytes, const pnn_tb_dev* d_tbs,
                           int n, int32_t* d_dst, float* d_out_f32, void* stream);

/* Distortion of n predicted blocks d_pred [n][w][w] against the ORIGINAL picture d_org_plane (same geometry as the
 * reconstructed plane: the descriptors' origin / stride address both), as HM's first intra pass computes it for a
 * candidate mode (TEncSearch.cpp:2376-2389, distParam.DistFunc): hadamard != 0 -> TComRdCost::xGetHADs (8x8 / 4x4
 * Hadamard SATD, TComRdCost.cpp:1753-1824), 0 -> SAD; 8-bit video.  Integer arithmetic, bit-exact. */
int pnn_block_cost_device(pnn_ctx* ctx, int width, const void* d_org_plane, int pel_bytes, const pnn_tb_dev* d_tbs, int n,
                          const int32_t* d_pred, int hadamard, uint32_t* d_cost, void* stream);
/* pnn_predict_tbs_device followed by pnn_block_cost_device on the same stream: only the n cost scalars need to leave
 * the device for candidates that are not selected (d_dst may be NULL). */
int pnn_predict_tbs_cost_device(pnn_ctx* ctx, int width, const void* d_plane, const void* d_org_plane, int pel_bytes,
                                const pnn_tb_dev* d_tbs, int n, int hadamard, uint32_t* d_cost, int32_t* d_dst, void* stream);

/* Per-launch accounting of the last *_device call (for bench.py's roofline object): number of tap-GEMM
 * launches and their algorithmic FLOPs (2 * M * K * N summed, padding excluded). */
int pnn_last_call_stats(const pnn_ctx* ctx, int* n_gemm_launches, double* gemm_flops, int* n_launches);

/* The same FLOPs without the multiply-adds of the taps that position-major tiles skipped (they only meet SAME padding: exact zeros) --
 * what the matrix cores were actually asked to do, for bench.py's executed-MFMA fractions of the convolutional nets. */
int pnn_last_call_issued_flops(const pnn_ctx* ctx, double* flops);

/* With pnn_set_option(ctx, "time_launches", 1) every tap-GEMM launch is bracketed by HIP events on its launch
 * stream. This call waits for them and returns, for kernel family `kind` (0 = tapgemm_f32_kernel, 2 = tapgemm_sp_kernel,
 * 3 = convimg_sp_kernel, 4 = tapgemm_ring_kernel, 5 = tapgemm_small_kernel, 6 = tapgemm_f32_small_kernel), the number of
 * launches since the last call, their summed duration and their summed algorithmic FLOPs. */
int pnn_launch_times(pnn_ctx* ctx, int kind, int* n_launches, double* total_us, double* total_flops);

#ifdef __cplusplus
}
#endif
#endif /* PNN_HIP_H */
