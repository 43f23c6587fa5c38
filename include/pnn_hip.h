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

/* Options: "precision" (1, default: tap GEMMs form every f32 product from three f16 MFMAs on hi/lo operand halves --
 * f32-class accuracy: operands carry 22 significand bits and the lo x lo term is dropped; 2.3-2.7x the blocks/s of 0 on
 * the bench workloads; 0: exact-f32 MFMA, IEEE float32 operands -- the reference's arithmetic), "sp_cfg" / "tile_cfg" (-1 = automatic tile choice; an
 * "sp_cfg" code in [0, pnn_num_split_configs()) forces one configuration of one of the three split-GEMM kernels on
 * every layer it can run -- all of them give bit-identical results), "ring" / "convimg" (1, default: the LDS-DMA ring
 * kernel / the LDS-resident-image convolution kernel may be chosen; 0: never), "fuse_last" (1, default: passes of
 * >= 1024 blocks through a fully-connected PNN with <= 64 outputs run the output layer inside the last hidden layer's
 * kernel; 0: separate launches), "cache_mb" (0, default: off; > 0: single-block host calls -- pnn_predict_pel / _fc /
 * _conv with n == 1, what HM issues -- are answered from a direct-mapped cache of that many MiB when the same input
 * bytes were predicted before: HM's rate-distortion search asks for the same block repeatedly; exact
 * match on the inputs, dropped whenever a model or an option changes),
 * "autotune" (the first call that meets a new (layer, batch size) pair times every legal configuration of the three
 * split-precision GEMM kernels on the device and keeps the fastest -- all of them give bit-identical results, so only
 * the speed depends on it; 2, default: only for launches of >= 4 GFLOP, i.e. big batches, where it costs a few tens
 * of milliseconds once; 1: always; 0: rule-based choice only.  Do the first call outside any timed region; while the
 * stream is being captured into a hipGraph nothing is timed -- the rule-based choice is used),
 * "fuse_first" (1, default: a convolutional net's second layer, when it runs on the LDS-resident-image kernel, computes
 * the branch's first (one-input-channel) convolution itself instead of reading it back from memory; 0: separate launch.
 * Bit-identical either way),
 * "fuse_gather" (1, default: pnn_predict_tbs_device / _cost_device on a convolutional net whose first convolutions run fused
 * inside the image kernel let that kernel read the contexts straight from the picture plane through the TB descriptors -- no
 * gather launch; bit-identical; 0: always gather first),
 * "flag_wait" (1, default: a small host call -- pnn_predict_pel / pnn_predict_f32 / pnn_predict_f32_pel on up to 64 KiB of input,
 * i.e. what HM and the batching service issue -- ends when the call's LAST kernel, behind its results in pinned host memory,
 * raises a sequence number there, which the calling thread spins on, instead of when the runtime reports the stream idle:
 * 3-7 us less per call; 0: hipStreamSynchronize.  Results do not depend on it),
 * "ring_pm" (1, default: convolutions at batch on the LDS-DMA ring kernel take position-major tiles -- a tile = many blocks at
 * ONE position of the feature map, so a tap that only meets the SAME padding there is skipped for the whole tile -- where
 * the launch model of pnn_gemm_ring.hip expects them to finish no later; 2: wherever possible; 0: never.  The skipped products
 * are exact zeros: bit-identical in every mode),
 * "fuse_tail" (1, default: when the last 64-channel layer of a convolutional net runs on the LDS-resident-image kernel, that
 * kernel applies the net's last layer -- the one-output-channel transposed convolution -- to its output tile in registers: no
 * round trip of the 64-channel maps, one launch less; 0: separate launch.  Bit-identical either way),
 * "split_min_px" (-1, default: built-in rule; >= 0: with precision 1, passes through a convolutional net use the
 * split-precision kernels from this many block pixels (blocks x w^2) on and the exact-f32 kernels below -- tuning aid),
 * "branch_streams" (1, default: small passes of the 32x32 / 64x64 convolutional nets -- the in-loop single-block
 * calls -- run the two independent branches on two HIP streams, forked and joined by events, and so do passes at batch
 * (>= 65536 block pixels on the split-precision kernels) from the third pass of a shape on, i.e. once a one-stream pass
 * needed no tuning sweep; 2: also every small conv pass; 0: one stream.  Results do not depend on it),
 * "small" (1, default: split-precision GEMMs with at most "small_max_tiles" (512) output tiles of 32 x 32 -- single-block
 * calls, small batches -- run on tapgemm_small_kernel, one wave per tile spread over the chip, in the SAME per-output
 * summation order as the big-tile kernels; 0: big-tile kernels only),
 * "pair" (1, default: such small passes of a convolutional net run layer i of BOTH branches as one launch -- they do not
 * depend on each other and a launch costs ~4 us whatever it does; bit-identical to separate launches; 0: one launch each),
 * "fc_out" (0, default; 1: small passes through a fully-connected PNN with <= 64 outputs run the output layer's K segments
 * AND their reduction as one launch, fc_out_small_kernel -- bit-identical, one launch less, measured no faster) and
 * "spin_wait" (0, default; 1: synchronous host calls poll the stream with hipStreamQuery instead of blocking in
 * hipStreamSynchronize -- measured no faster): two round-3 experiments on the single-block call kept as switches,
 * "f32_kernel" (1, default: exact-f32 passes -- "precision" 0, the range fallback of host calls -- run their tap GEMMs on
 * tapgemm_f32_kernel: v_mfma_f32_32x32x2_f32, one wave per SIMD, one per-output summation order for every tile and batch size;
 * fully-connected nets with <= 64 outputs sum the output layer in K segments of 160 hidden units at every batch size, inside the
 * last hidden layer's launch from 1024 blocks on ("fuse_last"); convolution layers deeper than 2304 per output are summed in K
 * segments of at most 1600 -- whole taps, added in order: by a second launch over planes of partial sums where the segments run as
 * separate workgroups ("f32_seg_mode" 0, default), inside the workgroups where they run in sequence (1: no planes, no second launch;
 * measured no faster at any batch size; -1: by cost model / tuner); same bits -- at every batch size; "ring_pm", "branch_streams", "fuse_gather" and "autotune" apply to
 * these passes like to the split-precision ones; 0: the round-1 kernels, tapgemm_kernel on 16x16x4 MFMA and the split-K kernel
 * for small M), "f32_cfg" (-1, default; >= 0 forces one tapgemm_f32 tile on every layer it is legal for -- tuning aid, all tiles
 * give the same bits), "f32_overlap" (1, default: the two branches of an exact-f32 conv pass at batch on two streams; 0: one),
 * "f32_persist" (-1, default: a convolution launch of more than two and at most six tiles per CU runs on two PERSISTENT workgroups
 * per CU, each taking its tiles one after the other; 0: never; N > 0: N per CU whenever there are more tiles; same bits),
 * "max_chunk" (blocks per pass, 0 = automatic), "ws_cap_mb", "time_launches",
 * "canonical_order" (1, default: every batch size uses the same per-output summation order, so a block's float
 * prediction is bit-identical whether it is predicted alone or inside any batch -- what an encoder/decoder pair needs
 * (single-block calls then run on tapgemm_small_kernel, the output layer of the 4x4 / 8x8 nets is summed in the same K
 * segments as the big batches' fused output layer); 0: small passes may use the exact-f32 split-K kernels, a few
 * microseconds faster per single-block call, whose float result can differ in the last bits, i.e. by one LSB on an
 * exact .5 tie -- never mix the two modes between an encoder and its decoder). */
int pnn_set_option(pnn_ctx* ctx, const char* name, long value);
/* Environment variables read at pnn_create* (same meaning as the options): PNN_PRECISION, PNN_AUTOTUNE, PNN_RING,
 * PNN_CONVIMG, PNN_SMALL, PNN_CANONICAL_ORDER, PNN_CACHE_MB, PNN_FC_OUT, PNN_SPIN_WAIT, PNN_FLAG_WAIT, PNN_FUSE_FIRST, PNN_FUSE_GATHER, PNN_FUSE_TAIL, PNN_RING_PM, PNN_FUSE_LAST, PNN_BRANCH_STREAMS, PNN_TILE_CFG, PNN_MAX_CHUNK, PNN_F32_KERNEL, PNN_F32_CFG, PNN_F32_OVERLAP, PNN_F32_SEG_MODE, PNN_F32_PERSIST.  Diagnostics:
 * PNN_DEBUG (tile choice of every GEMM launch on stderr), PNN_DEBUG_TUNE (every timed configuration), PNN_PROFILE
 * (synchronous per-launch timing), PNN_LIB_PATH (Python loader: another build of the library).  Experiment switches of
 * individual launchers, not part of the interface: PNN_SK_WAVES, PNN_LDS_PAD, PNN_SP_DIAG, PNN_F32_DIAG (diagnostic library of `make diag` only),
 * PNN_F32_SEG_DEPTH / PNN_F32_SEG_MIN (the K segments of the exact-f32 summation order, read when a model is loaded; 0 = none. They
 * DEFINE that order: an encoder and its decoder must run with the same values -- leave them alone outside A/B measurements). */
/* Input-range contract of the default arithmetic ("precision" = 1): operands travel as pairs of f16 values, so every
 * intermediate activation must satisfy |v| < 65504.  8-bit contexts through trained models stay two orders of magnitude
 * below that; arbitrary float inputs or models may not.  The kernels detect a violation (they never emit
 * a silent NaN; the raw context a convolutional net's first layer splits in registers is checked where it is staged): host entry points (pnn_predict_fc / _conv / _pel / _f32_pel) then recompute, on the exact-f32 kernels and
 * by themselves, exactly the blocks that overflow when predicted alone (batches of <= 256; the other blocks of the batch keep
 * the bits they get in any batch), and they refuse non-finite inputs (PNN_E_ARG) -- the guard's max would drop a NaN; models
 * with a non-finite parameter are refused at load (PNN_E_MODEL).  Device entry points are asynchronous, so the NEXT call on the context fails with PNN_E_RANGE, and
 * pnn_check_range -- which waits for `stream` -- tells right away (returns PNN_OK or PNN_E_RANGE; *host_fallbacks, optional,
 * = how many host calls took the exact-f32 repeat so far). */
int pnn_check_range(pnn_ctx* ctx, void* stream, long* host_fallbacks);

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

/* With pnn_set_option(ctx, "time_launches", 1) every tap-GEMM launch is bracketed by HIP events on its launch
 * stream. This call waits for them and returns, for kernel family `kind` (0 = tapgemm_kernel / tapgemm32_kernel,
 * 1 = tapgemm_splitk_kernel, 2 = tapgemm_sp_kernel, 3 = convimg_sp_kernel, 4 = tapgemm_ring_kernel, 5 = tapgemm_small_kernel), the number of
 * launches since the last call, their summed duration and their summed algorithmic FLOPs. */
int pnn_launch_times(pnn_ctx* ctx, int kind, int* n_launches, double* total_us, double* total_flops);

#ifdef __cplusplus
}
#endif
#endif /* PNN_HIP_H */
