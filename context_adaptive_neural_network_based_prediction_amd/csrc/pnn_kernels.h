// Internal launch interface between the C-ABI layer (pnn_abi.cpp) and the gfx950 kernels
// (pnn_kernels.hip).  Not part of the public boundary -- see include/pnn_hip.h for that.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace pnn {

// see signal_done, pnn_device_common.h.  per_wg = 1 (round 6, fc_out_f32_chain_kernel): workgroup g raises host_flag[g] by itself -- no
// counter, no second fence; the host waits for all of the launch's flags
struct DoneSignal { unsigned* counter; unsigned* host_flag; unsigned seq, per_wg; };

// Per-launch timing (pnn_abi.cpp, option time_launches): when set, the GEMM launchers attach these events to the kernel
// itself (hipExtLaunchKernelGGL), so that their elapsed time is the kernel's own begin -> end -- what rocprofv3
// reports -- instead of the record-to-record time of two hipEventRecord calls around the launch (~4 us more).
struct LaunchEvents { hipEvent_t start, stop; };
extern thread_local const LaunchEvents* g_launch_events;
template <typename K, typename... A>
inline void pnn_launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args)
{
    if (g_launch_events) hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)lds, s, g_launch_events->start, g_launch_events->stop, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
}

// What the launch rules need to know about the CURRENT device (hipGetDevice), read once per device from hipDeviceProp: compute units
// and LDS bytes per CU (MI355X: 256 and 160 KiB; a partitioned GPU -- CPX / DPX -- or another SKU shows other numbers).
struct DeviceInfo { int cus = 256; size_t lds = (size_t)160 << 10; int dev = -1; };
const DeviceInfo& device_info();

hipError_t probe_queue_shared(hipStream_t a, hipStream_t b, bool* shared);   // pnn_small.hip: do two streams sit on one hardware queue?

constexpr int kMaxTaps = 32;
constexpr int kMaxClasses = 4;

// One "tap GEMM": Y[pix(m)][n] = act( sum_{t in class} sum_{ci} X[b, i*a+dy[t], j*a+dx[t], ci] * W[t][ci][n] + bias[n] )
// with m = (b, i, j) over a SH x SW sub-grid per image and the output pixel (i*os+py, j*os+px).
//   forward conv, stride s : one class, a = s, os = 1, dy = ky - pad_before        
//   transposed conv, s = 1 : one class, a = 1, os = 1, dy = pad_before - ky         (Appendix B.3)
//   transposed conv, s = 2 : four output-parity classes, a = 1, os = 2, dy = (py + 1 - ky) / 2
//   fully-connected layer  : one tap, 1x1 "image", Cin = K                            (components.py:169-176)
// Weights are pre-packed per 16-deep K chunk as [chunk][q = 4][Npad][4] floats (k = 16*chunk + 4*q + e),
// the order in which one 16x16x4 f32 MFMA lane group consumes them.
struct TbDev;
struct TapGemmParams {
    const float* X;    // f32 activations, or the hi f16 plane for the split-precision kernel
    const void* Xlo;   // lo f16 plane (split-precision kernel only)
    const void* zero;  // >= 256 bytes of zeros (ring kernel: source of padding / out-of-range pieces)
    // ring kernel, fused next fully-connected layer (<= 64 outputs): its split-packed weights, their Npad, the number of
    // packed 16-deep chunks, and the partial-product buffer [column tiles][M][64] (see pnn_gemm_ring.hip)
    // (the image kernel never reads these; when it applies a net's LAST layer to its output tile -- k1 != 0 below, see
    // pnn_convimg_sp.hip -- that layer's weights [k1][k1][Cout], and the net's outputs (float and / or HM epilogue) live in the same bytes)
    union {
        struct { const float* W2p; int Npad2; int K2chunks; float* part; };
        struct { const float* W1; float* Y1; int32_t* Yi1; };
        // tapgemm_f32_small_kernel, a K-segmented layer (nseg > 1) whose planes are added up INSIDE the launch (seg_cnt != NULL): the
        // workgroup that is the last of a tile's nseg to arrive (a counter per tile in seg_cnt, zero between launches) adds the planes in
        // order, + bias, activation, and stores the layer's real output seg_Y; `bias` and `act` are then the layer's own, Y the planes
        struct { unsigned* seg_cnt; float* seg_Y; };
    };
    // convimg kernel, fused FIRST convolution of a branch (Cin = 1 -> this layer's Cin channels, stride s0, kernel k0 x k0,
    // SAME padding with pad0 before, LeakyReLU): X0 != NULL makes the kernel compute its input maps from the raw f32
    // context X0 [images][IH * s0][IW * s0] instead of reading them from X (see pnn_convimg_sp.hip)
    // (The ring kernel never reads these fields; its position-major launches keep their position order in the same bytes --
    // pos_order below -- so that the argument block stays at 8 lines of 64 bytes: every line is a round trip at kernel entry,
    // and a ninth cost the FC 8x8 pass 0.7 %.)
    union {
        struct {
            const float* X0; const float* W0; const float* B0; int s0, k0, pad0;
            const float* W0sp; float scale0; int Npad0;   // the first convolution's taps x channels matrix in the split pack (FirstConv, pnn_device_common.h)
            // ... and, when plane0 != NULL, the context gather too (extraction_context.cpp:3-208): X0 is not read; image i's raw context
            // comes straight from the picture plane through TB descriptor tbs0[i] -- branch0 = 0: the above portion (rows y - w ..
            // y - 1, columns x - w .. x + 2w - 1, masked per unit0-pixel unit), 1: the left portion (rows y .. y + 2w - 1, columns
            // x - w .. x - 1, the first left_units units) --, Pel (pel0 bytes) -> float, minus `mean`
            const void* plane0; const void* tbs0; int pel0, branch0, unit0, w0;
        };
        // ring kernel, position-major tiles: the positions by decreasing number of in-image taps, one byte each (grids of <= 64
        // positions; larger grids: rank = position)
        unsigned pos_order[16];
    };
    const float* Wp;
    const float* bias;
    float* Y;          // float output (may be null when Yi is set)
    int32_t* Yi;       // optional fused HM epilogue: (int) round(clamp(v + mean, 0, 255))
    void* Yhi;         // optional split-f16 output planes (split-precision kernel only)
    void* Ylo;
    int* range_flag;   // raised (host-visible int) when a split-f16 output leaves the f16 range, see pnn_device_common.h
    float out_scale;   // split-precision kernel: exact power of two undoing the weight pre-scale
    unsigned x_bytes;  // size of X in bytes (< 2^31): bound of the activation buffer descriptor
    int M, SH, SW;
    int IH, IW, Cin, a;
    int OH, OW, Cout, os;
    int Npad;
    int act;
    float mean;
    int ncls;
    int tap_begin[kMaxClasses + 1];
    int chunk_begin[kMaxClasses + 1];   // first packed 16-deep chunk of each class (classes padded to kChunkPad)
    int py[kMaxClasses], px[kMaxClasses];
    int tap[kMaxTaps];   // (dy << 16) | (dx & 0xffff): one scalar load per tap
    // ring kernel, position-major tiles (set by launch_tapgemm_ring, see pnn_gemm_ring.hip): pm_groups > 0 = a workgroup's BM
    // rows are BM BLOCKS at ONE position of the SH x SW grid, so a tap that falls outside the image does so for the whole
    // tile and is skipped; pm_groups = number of block groups, nblk = number of blocks; the position order: pos_order above
    // (tapgemm_f32_small kernels, in nblk's bytes: chain_io bit 0 = the activations X are in CHAIN ORDER -- NHWC with every 16-channel
    // group permuted so that the four values a lane group of the 16x16x4 chain multiplies are contiguous, pnn_gemm_f32_small.hip --,
    // bit 1 = write the output that way; set by the pass for tensors whose producer AND consumer are those kernels)
    int pm_groups; union { int nblk; int chain_io; };
    // image kernel, fused last layer (Cout == 64 -> 1 transposed convolution, kernel k1, stride s1, pad1 before, bias1): k1 != 0
    float bias1; int k1, s1, pad1;
    // tapgemm_f32_kernel, K segments (nseg > 1, see GemmLayer::nseg): grid z = class * nseg + segment; a workgroup walks only its
    // segment's share of the class's taps and stores its sums at Y + segment * seg_stride (floats) -- the caller passes a zero
    // bias and act = 0 and finishes with launch_seg_reduce
    // ONE-TAP layers (FC, round 6; GemmLayer::fc_seg_chunks): segments of seg_chunks 16-deep chunks of the tap, always folded inside the
    // workgroup (seg_seq = 1 in tapgemm_f32_kernel; fcseg_f32_small_kernel runs them as parallel chains of one workgroup) -- seg_stride's bytes
    int nseg; union { unsigned seg_stride; unsigned seg_chunks; };
    // seg_seq = 1: the segments one after the other inside each workgroup instead (grid z = class), folded into a running total in
    // the same order -- same bits, no partial planes, no second launch; the caller passes the real Y, bias and act
    int seg_seq;
    // tapgemm_f32_kernel: the launch's tiles (set by launch_f32; the kernel's grid is 1-D over them, or smaller: persistent workgroups)
    int grid_x, grid_y, grid_z;
    int persist;       // tapgemm_f32 launches: > 0 = persistent workgroups, that many per CU (see launch_f32)
};
static_assert(sizeof(TapGemmParams) <= 512, "the argument block of the tap-GEMM kernels: 8 lines of 64 bytes");
inline int pack_tap(int dy, int dx) { return (int)(((unsigned)dy << 16) | ((unsigned)dx & 0xffffu)); }

constexpr int kChunkPad = 4;   // packed weights: every class is zero-padded to a multiple of 4 chunks
constexpr int kSegDepth = 1600, kSegMinDepth = 2304;   // K segments of the exact-f32 summation order, see finish_gemm_layer (pnn_model.cpp)
constexpr int kFcSegChunks = 20;                       // ... of a one-tap (FC) layer deeper than this many 16-deep chunks: segments of 320 inputs
struct TileCfg { int rt, nt, kc, mf, wm = 4, d = 2; };   // a tile configuration of one of the tap-GEMM kernel families (mf: rows of the MFMA shape)
int tapgemm_sp_num_cfgs();
TileCfg tapgemm_sp_cfg(int idx);
hipError_t launch_tapgemm_sp(const TapGemmParams& p, int idx, hipStream_t s);   // 3 x f16 MFMA, f32-class accuracy
int tapgemm_ring_num_cfgs();
TileCfg tapgemm_ring_cfg(int idx);
size_t tapgemm_ring_lds_bytes(const TileCfg& t);
bool tapgemm_ring_can_fuse(int idx);
// Y[i] = act(bias[i % Cout] + ((part[0][i] + part[1][i]) + ... + part[nseg - 1][i])), i < n (n and Cout multiples of 4): the K segments of
// a tapgemm_f32 layer in their canonical order
hipError_t launch_seg_reduce(const float* part, int nseg, size_t n, int Cout, const float* bias, int act, float* Y, hipStream_t s);
hipError_t launch_fuse_reduce(const float* part, int ntiles, int M, int N2, const float* bias, float scale, float mean, float* Y, int32_t* Yi,
                              hipStream_t s, const DoneSignal& done = DoneSignal{nullptr, nullptr, 0, 0});
hipError_t launch_tapgemm_ring(const TapGemmParams& p, int idx, hipStream_t s);     // LDS-DMA ring pipeline (pnn_gemm_ring.hip)
// Position-major tiles (a workgroup's BM rows = BM blocks at ONE position of the SH x SW grid: the taps that only meet padding are
// skipped): whether a launch of tile BM x BN, KC chunks per stage, `lds` bytes per workgroup takes them, in how many block
// groups, and the position order (pnn_gemm_ring.hip; shared by the ring kernel and tapgemm_f32_kernel).  p.pm_groups on entry:
// -1 never, 1 whenever possible, 0 by the planner's list-scheduling model.
struct PmPlan { bool use = false; int groups = 0; unsigned order[16] = {}; double live_frac = 1.0; };   // live_frac: in-image taps / all taps, over the positions
// Fraction of a tap GEMM's algorithmic multiply-adds that the LAST launch of this thread issued: 1 for block-major tiles, the plan's
// live_frac for position-major ones (the skipped taps only meet SAME padding) -- bench.py's executed-MFMA fractions
extern thread_local double g_last_issued_frac;
const PmPlan& position_major_plan(const TapGemmParams& p, int BM, int BN, int KC, size_t lds, double l2_mb = 4.5);
int convimg_sp_num_cfgs();
TileCfg convimg_sp_cfg(int idx);
size_t convimg_sp_lds_bytes(const TapGemmParams& p, const TileCfg& t, int G);
bool convimg_sp_can_fuse_first(const TapGemmParams& p, const TileCfg& t, int G, int s0, int k0);   // the raw context tiles fit the weight staging area
struct TConv1Params;
bool convimg_sp_can_fuse_last(const TapGemmParams& p, const TileCfg& t, int G, const TConv1Params& last);   // the net's last layer on the output tile (k1 != 0)
hipError_t launch_convimg_sp(const TapGemmParams& p, int idx, int G, hipStream_t s);   // G images per workgroup, resident in LDS
hipError_t launch_split(const float* x, long n, void* hi, void* lo, int* range_flag, hipStream_t s);
// Small-M split-precision tap GEMM (pnn_gemm_small.hip): one wave per 32 x 32 output tile, same per-output summation
// order as the three big-tile kernels.  a_is_f32: p.X holds plain f32 rows (split in registers); seg_chunks > 0: K-segment
// mode of an FC output layer (<= 64 outputs), raw partials to p.part[segment][M][64] for launch_fuse_reduce.
long tapgemm_small_tiles(const TapGemmParams& p);
// host_input (optional, with a_is_f32): the same rows in HOST memory; when they fit they travel inside the argument block
hipError_t launch_tapgemm_small(const TapGemmParams& p, bool a_is_f32, int seg_chunks, hipStream_t s, const float* host_input = nullptr);
// Output layer (<= 64 outputs) of an FC net at small M in one launch: K segments + their reduction (pnn_gemm_small.hip)
bool fc_out_small_fits(const TapGemmParams& p, int seg_chunks);
hipError_t launch_fc_out_small(const TapGemmParams& p, int seg_chunks, hipStream_t s, const DoneSignal& done = DoneSignal{nullptr, nullptr, 0, 0});
hipError_t launch_tapgemm_small_pair(const TapGemmParams& a, const TapGemmParams& b, hipStream_t s);   // two independent layers, one launch
// Exact-f32 tap GEMM on v_mfma_f32_32x32x2_f32, one wave per SIMD (pnn_gemm_f32.hip): the canonical f32 summation order.
// fuse: apply the output layer p.W2p (f32 pack, <= 64 outputs) to the activated tile, partial sums to p.part[column tile][M][64]
int tapgemm_f32_num_cfgs();
TileCfg tapgemm_f32_cfg(int idx);
size_t tapgemm_f32_lds_bytes(const TileCfg& t, bool fuse, bool row_out = false);   // row_out: an FC layer's f32 output leaves through an LDS tile
bool tapgemm_f32_can_fuse(int idx);
int tapgemm_f32_regs(int idx);   // registers per lane of tile idx (residency estimate of choose_cfg_f32)
hipError_t launch_tapgemm_f32(const TapGemmParams& p, int idx, bool fuse, hipStream_t s);
// Small-M form of the same order on v_mfma_f32_16x16x4_f32 (pnn_gemm_f32_small.hip): one wave per 16 x 16 tile, K segments as grid z;
// host_input (optional, FC layers): the same rows in HOST memory; when they fit they travel inside the argument block
long tapgemm_f32_small_tiles(const TapGemmParams& p);
hipError_t launch_tapgemm_f32_small(const TapGemmParams& p, hipStream_t s, const float* host_input = nullptr, int deep_mode = 1);   // deep_mode: pnn_ctx::opt_f32_small_deep
hipError_t launch_tapgemm_f32_small_pair(const TapGemmParams& a, const TapGemmParams& b, hipStream_t s, int deep_mode = 1);   // two independent layers, one launch
// A K-segmented one-tap (FC) layer at small M: one workgroup per 16 x 16 output tile runs the layer's <= 4 segments as 4 chains side by
// side and adds them up in order (p.nseg, p.seg_chunks; the bits of tapgemm_f32_kernel's seg_seq form)
long fcseg_f32_small_tiles(const TapGemmParams& p);
bool fcseg_f32_small_fits(const TapGemmParams& p);
hipError_t launch_fcseg_f32_small(const TapGemmParams& p, hipStream_t s);
// the same output layer from stored activations p.X [M][Cin], in the fused kernel's order: p.part[segment of 160][M][64]
hipError_t launch_fc_out_f32(const TapGemmParams& p, hipStream_t s, int* segments);
// ... and, for small M, the same segments AND their reduction (+ bias, HM epilogue) in one launch: fuse_reduce_kernel's bits
bool fc_out_f32_small_fits(const TapGemmParams& p);
hipError_t launch_fc_out_f32_small(const TapGemmParams& p, hipStream_t s, const DoneSignal& done = DoneSignal{nullptr, nullptr, 0, 0}, bool round5_form = false);

// Cin == 1 forward convolution (first layer of each branch): direct VALU kernel.
struct Conv1Params {
    const float* X; const float* W; const float* bias; float* Y;
    int B, IH, IW, s, k, pad, OH, OW, Cout;
    int split;   // 1: write Y as split activations [pixel][Cout/16][hi 16 x f16 | lo 16 x f16] for the split-precision GEMM
    int band_rows;   // output rows per workgroup (set by the launcher)
    int* range_flag; // split output only: raised when a value leaves the f16 range
    const float* Wsp; float out_scale; int npad;   // split output: the taps x channels matrix in the split pack of the GEMM layers, its inverse pre-scale, its Npad
    // X == NULL: the context gather fused in (extraction_context.cpp:3-208) -- image b's raw context comes straight from the picture
    // plane through TB descriptor tbs[b] (branch 0: the above portion w x 3w, 1: the left portion 2w x w; unit-pixel availability
    // units), Pel (pel_bytes) -> float, minus mean: the values gather_f32x4_kernel would have written
    const void* plane; const TbDev* tbs; int pel_bytes, unit, w, branch; float mean;
    int chain;       // f32 output only: 1 = write it in chain order (its consumer is a small exact-f32 kernel, pnn_gemm_f32_small.hip)
    int mfma;        // f32 output only: 1 = the taps' chain on the f32 matrix instruction instead of the VALU (same bits; set by the launcher)
};
hipError_t launch_conv_cin1(const Conv1Params& p, hipStream_t s);
hipError_t launch_conv_cin1_pair(const Conv1Params& a, const Conv1Params& b, hipStream_t s);   // both branches in one launch (same batch, same kernel size)

// Cout == 1 transposed convolution (last merger layer), optional fused HM epilogue.
struct TConv1Params {
    const float* X; const float* W; /* [k][k][Cin] */ float bias; float* Y; int32_t* Yi;
    int B, IH, IW, Cin, s, k, pad; float mean;
    int ni;   // images per workgroup (set by the launcher)
    DoneSignal done;   // host_flag != NULL: this is the last kernel of a host call, see signal_done (pnn_device_common.h)
};
hipError_t launch_tconv_cout1(const TConv1Params& p, hipStream_t s);

// Channel-wise fully-connected merger + LeakyReLU (Appendix B.4). Wp is [p = 80][j = 16][C].
struct MergerParams {
    const float* A; const float* L; const float* Wp; const float* bias /* [j][C] */; float* Y;
    int B, C, na, nl, nout;
    int split;   // 1: write Y in the split f16 activation layout
    int* range_flag; // split output only: raised when a value leaves the f16 range
    int chain;       // f32 output only: 1 = write it in chain order (its consumer is a small exact-f32 kernel, pnn_gemm_f32_small.hip)
};
hipError_t launch_merger(const MergerParams& p, hipStream_t s);

// A small layer run as the TAIL of the small exact-f32 GEMM launch in front of it (pnn_gemm_f32_small.hip, round 6): kind 1 = the
// merger per (block, channel group) behind the pair launch of the branches' last layers, kind 2 = the last transposed convolution
// (with the HM epilogue and one completion flag per block, DoneSignal::per_wg) per block behind the last GEMM of the transposed
// stack.  cnt: one zeroed counter per instance (kind 1: blocks x C / 16, kind 2: blocks); the launch leaves them zero.
struct SmallTail { int kind; unsigned* cnt; MergerParams m; TConv1Params t; };
bool f32_small_cout1_tail_ok(const TapGemmParams& p, const TConv1Params& t);
bool f32_small_merger_tail_ok(const TapGemmParams& a, const TapGemmParams& b, const MergerParams& m);
hipError_t launch_tapgemm_f32_small_tail(const TapGemmParams& p, const SmallTail& t, hipStream_t s, int deep_mode = 1);
hipError_t launch_tapgemm_f32_small_pair_tail(const TapGemmParams& a, const TapGemmParams& b, const SmallTail& t, hipStream_t s, int deep_mode = 1);

// L-shaped context gather (extraction_context.cpp:3-208) over a descriptor array.
struct TbDev {           // mirrors pnn_tb_dev of include/pnn_hip.h
    int64_t origin;      // element index of the TB's top-left pixel from the plane base
    int32_t stride;      // row stride in elements
    uint32_t above_mask; // bit u = above/above-right unit u (left to right) available
    int32_t left_units;  // number of available left/below-left units, counted from the top
    int32_t reserved;
};
struct GatherParams {
    const void* plane; int pel_bytes; const TbDev* tbs; int N; int w; int unit; float mean;
    float* above; float* left; long pitch_above; long pitch_left;
    int split;   // 1 (FC layout only, one [5w^2] row per TB): write the split f16 activation layout instead of f32
};
hipError_t launch_gather(const GatherParams& p, hipStream_t s);

// (f4) HM distortion (HADs or SAD) of predicted blocks [N][w][w] against the original picture at the descriptors' positions.
struct BlockCostParams {
    const void* org_plane; int pel_bytes; const TbDev* tbs; int N; int w; const int32_t* pred; int hadamard; uint32_t* cost;
};
hipError_t launch_block_cost(const BlockCostParams& p, hipStream_t s);

// Stand-alone HM epilogue for float predictions.
hipError_t launch_epilogue(const float* pred, long n, float mean, int32_t* dst, hipStream_t s);

}  // namespace pnn
