#!/usr/bin/env python3
"""Benchmark of the PNN intra-prediction hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fc8|conv16|...] [--batch B]

One "step" = one pass of the hot path (L-context gather -> PNN -> +mean/clamp/round -> int32 Pel) over one
batch of synthetic transform blocks per GPU, inputs (reconstructed planes + TB descriptors) resident in
HBM.  Default workload = BASELINE.json configs[1]: 8x8 fully-connected PNN, batch 4096, 1 x MI355X.
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank processes its own batch
(independent blocks: no data-path collective, weak scaling) and the time is the max over ranks.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAMP_SECONDS = 0.4          # untimed device ramp-up before the warm-up steps (see main)
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32 matrix peak (v_mfma_f32_16x16x4_f32)
DEFAULT_PRECISION = "1"           # library default (pnn_set_option "precision"); PNN_PRECISION overrides
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 matrix peak

WORKLOADS = {                      # name -> (width, is_fc, default batch per GPU, BASELINE.json config)
    "fc4": (4, True, 4096, "4x4 fully-connected PNN"),
    "fc8": (8, True, 4096, "configs[1]: 8x8 fully-connected PNN, batch 4096"),
    "conv16": (16, False, 1024, "configs[2]: 16x16 convolutional PNN, batch 1024"),
    "conv32": (32, False, 256, "32x32 convolutional PNN"),
    "conv64": (64, False, 64, "64x64 convolutional PNN"),
    "conv8": (8, False, 4096, "8x8 convolutional PNN"),
    "conv4": (4, False, 4096, "4x4 convolutional PNN"),
}


def flops_per_block(width, is_fc):
    """Dense MAC count x 2 (SURVEY.md Appendix A / BASELINE.md section 2)."""
    from context_adaptive_neural_network_based_prediction_amd import weights as wts
    if is_fc:
        h = wts.FC_HIDDEN
        return 2.0 * (5 * width * width * h + 2 * h * h + h * width * width)
    st = wts.STRIDES_BRANCH[width]
    macs = 0
    for (H, W) in ((width, 3 * width), (2 * width, width)):
        cin, c = 1, 32
        for s in st:
            k = 2 * s + 1
            c *= s
            H, W = H // s, W // s
            macs += H * W * k * k * cin * c
            cin = c
    macs += c * 80 * 16
    H, ci = 4, c
    for i, s in enumerate(st[::-1]):
        k = 2 * s + 1
        co = 1 if i == len(st) - 1 else ci // s
        macs += H * H * k * k * ci * co
        H *= s
        ci = co
    return 2.0 * macs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fc8", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="blocks per GPU per step (0 = the workload's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
    from tests import util

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    precision_name = ("f32 (3 x f16 MFMA split products, f32 accumulate)" if os.environ.get("PNN_PRECISION", DEFAULT_PRECISION) == "1"
                      else "f32")
    width, is_fc, default_batch, cfg_name = WORKLOADS[args.workload]
    batch = args.batch or default_batch
    L = _lib.lib()

    # ---- synthetic workload, resident in HBM before the timed region -------------------------------------
    params = util.make_params(width, is_fc, seed=1, out_gain=30.0)     # reference initialiser statistics
    net = PredictionNeuralNetwork(batch, width, is_fc, params=params, device=local_rank)
    net.set_option("autotune", int(os.environ.get("PNN_AUTOTUNE", "1")))   # tile choice measured on the device during warm-up
    plane_h, plane_w = 1088, 1920                                       # one HD luminance plane of int32 Pel
    plane = util.make_plane(plane_h, plane_w, seed=100 + rank, pad=64)
    xs, ys, flags = util.make_tbs(plane_h, plane_w, width, batch, seed=200 + rank, partial_fraction=0.3)
    units = 2 * width // 4
    tbs = (_lib.TbDev * batch)()
    for i in range(batch):
        assert L.pnn_make_tb_desc(ctypes.byref(tbs[i]), int(ys[i]) * plane.shape[1] + int(xs[i]), plane.shape[1],
                                  flags[i].ctypes.data_as(_lib.u8p), int(flags[i].sum()), units, units) == 0
    d_plane = torch.from_numpy(plane).cuda()
    d_tbs = torch.from_numpy(np.frombuffer(tbs, dtype=np.uint8).copy()).cuda()
    d_dst = torch.empty((batch, width, width), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)

    def step():
        rc = L.pnn_predict_tbs_device(net.ctx, width, d_plane.data_ptr(), 4, d_tbs.data_ptr(), batch, d_dst.data_ptr(), None, sp)
        if rc:
            raise RuntimeError(L.pnn_last_error(net.ctx))

    def barrier():
        if dist is not None:
            dist.barrier()

    # Set-up, untimed and outside the W warm-up steps: the first calls autotune the tile configurations, and the device
    # needs a few hundred milliseconds of sustained work before it holds its clocks (measured: 0.126 ms per step right
    # after start, 0.113 ms once warm) -- run steps for RAMP_SECONDS so that W and K see the steady state.
    step()
    torch.cuda.synchronize()
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < RAMP_SECONDS:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = net.last_call_stats()

    gpu_pred = d_dst.cpu().numpy() if rank == 0 else None   # what the timed steps produced; checked in the CPU leg below
    parity, parity_detail = None, None

    # ---- roofline of the dominant kernel (tapgemm_kernel), HIP events on the launch stream ---------------
    # Region = the network alone on pre-gathered contexts: for FC nets exactly 4 tap-GEMM launches per pass.
    above, left = util.make_contexts(width, batch, seed=300 + rank)
    if is_fc:
        d_in = (torch.from_numpy(util.flatten_fc(above, left)).cuda(),)
    else:
        d_in = (torch.from_numpy(above).cuda(), torch.from_numpy(left).cuda())
    d_out = torch.empty((batch, width, width), dtype=torch.float32, device="cuda")

    def net_only():
        if is_fc:
            rc = L.pnn_predict_fc_device(net.ctx, width, d_in[0].data_ptr(), batch, d_out.data_ptr(), sp)
        else:
            rc = L.pnn_predict_conv_device(net.ctx, width, d_in[0].data_ptr(), d_in[1].data_ptr(), batch, d_out.data_ptr(), sp)
        if rc:
            raise RuntimeError(L.pnn_last_error(net.ctx))

    for _ in range(3):
        net_only()
    reps = max(5, min(args.steps, 50))
    torch.cuda.synchronize()
    net.set_option("time_launches", 1)             # HIP events around every tap-GEMM launch, on the launch stream
    for _ in range(reps):
        net_only()
    torch.cuda.synchronize()
    net.set_option("time_launches", 0)
    nstats = net.last_call_stats()
    kinds = {}
    for kind, name in ((0, "tapgemm_kernel (exact f32 MFMA 16x16x4, LDS-staged weights)"),
                       (1, "tapgemm_splitk_kernel (f32 MFMA, small M)"),
                       (2, "tapgemm_sp_kernel (f32-class split products on 3 x f16 MFMA 32x32x16, register-staged operands)"),
                       (3, "convimg_sp_kernel (same split-product MFMAs, feature maps resident in LDS)"),
                       (4, "tapgemm_ring_kernel (same split-product MFMAs; 4 MFMA + 4 loader waves, LDS-DMA ring; incl. the fused output layer)")):
        n_k, us_k, fl_k = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        L.pnn_launch_times(net.ctx, kind, ctypes.byref(n_k), ctypes.byref(us_k), ctypes.byref(fl_k))
        kinds[kind] = {"kernel": name, "launches_timed": n_k.value, "total_us": us_k.value, "flops": fl_k.value}
    dom = max(kinds, key=lambda k: kinds[k]["total_us"])          # dominant kernel of this workload
    gemm_flops_per_launch = kinds[dom]["flops"] / max(kinds[dom]["launches_timed"], 1)
    avg_launch_s = kinds[dom]["total_us"] * 1e-6 / max(kinds[dom]["launches_timed"], 1)
    achieved_tflops = gemm_flops_per_launch / avg_launch_s / 1e12
    # peak: exact-f32 MFMA 157.3 TFLOP/s; the split-precision kernel issues three f16 MFMAs (2.5 PFLOP/s dense) per
    # algorithmic product, so its roof in algorithmic FLOPs is 2500 / 3.
    peak_tflops = PEAK_F32_MFMA_TFLOPS if dom < 2 else PEAK_F16_MFMA_TFLOPS / 3.0
    other_launches = nstats["launches"] - nstats["gemm_launches"]
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # PMC pass results (FETCH_SIZE x2 + WRITE_SIZE, per launch)
    if os.path.exists(pmc):
        rec = json.load(open(pmc)).get(args.workload)
        if rec and batch == default_batch:
            traffic, traffic_src = rec["bytes_per_launch"], rec["source"]

    # ---- CPU baseline: the oracle (a port; TF1 cannot be installed) on this box's host cores --------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pnn_oracle as O
        ncpu = min(batch, 4096 if is_fc else (512 if width <= 16 else 64))
        O.predict_tbs(params, width, is_fc, plane, xs[:8], ys[:8], flags[:8], util.MEAN)      # thread pool warm-up
        c0 = time.perf_counter()
        reps_cpu = 0
        cpu_pred = None
        while reps_cpu < 8 and (time.perf_counter() - c0) < 10.0:
            cpu_pred = O.predict_tbs(params, width, is_fc, plane, xs[:ncpu], ys[:ncpu], flags[:ncpu], util.MEAN)
            reps_cpu += 1
        cdt = time.perf_counter() - c0
        # the CPU leg's output doubles as the parity check of what the GPU steps produced (uint8 LSBs after the HM epilogue)
        pdiff = np.abs(gpu_pred[:ncpu].astype(np.int64) - cpu_pred)
        parity = int(pdiff.max())
        parity_detail = {"pixels_compared": int(pdiff.size), "pixels_differing": int((pdiff != 0).sum())}
        cpu = {"value": ncpu * reps_cpu / cdt, "unit": "blocks/s", "cores": os.cpu_count(), "kind": "port",
               "sample": "%d x %d blocks of the same workload through oracle/pnn_oracle.c (OpenMP, -O3 -mavx2 -mfma), "
                         "batched; stand-in for the reference's TF-1.9 CPU path" % (reps_cpu, ncpu)}

    if rank == 0:
        total_blocks = float(batch) * world * args.steps
        out = {
            "metric": "pnn_intra_pred_blocks_per_s",
            "value": total_blocks / elapsed,
            "unit": "blocks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": precision_name,
            "data": "synthetic",
            "config": {"workload": cfg_name, "width": width, "arch": "fully_connected" if is_fc else "convolutional",
                       "batch_per_gpu": batch, "path": "gather + net + HM epilogue (pnn_predict_tbs_device)",
                       "weights": "seeded random init with the reference initialisers' statistics",
                       "parallelism": "independent blocks sharded over ranks, no data-path collective",
                       "tile_autotune": "on first use, before the warm-up steps (pnn_set_option autotune)",
                       "device_ramp": "%.2f s of untimed steps before the W warm-up steps (clock ramp)" % RAMP_SECONDS},
            "launches_per_step": stats["launches"],
            "max_abs_lsb_vs_oracle": parity,
            "parity_detail": parity_detail,
            "roofline": {"bound": "mfma", "kernel": kinds[dom]["kernel"], "achieved": achieved_tflops,
                         "peak": peak_tflops, "unit": "TFLOP/s", "frac": achieved_tflops / peak_tflops,
                         "traffic": traffic, "traffic_source": traffic_src, "flops_per_launch": gemm_flops_per_launch,
                         "avg_launch_us": avg_launch_s * 1e6, "launches_timed": kinds[dom]["launches_timed"],
                         "peak_note": ("f16 dense MFMA peak 2500 / 3 MFMAs per algorithmic product" if dom >= 2
                                       else "f32 dense MFMA peak at 2.4 GHz"),
                         "frac_of_f32_mfma_peak": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                         "gemm_launches_per_pass": nstats["gemm_launches"], "non_gemm_launches_per_pass": other_launches,
                         "other_gemm_kernels": [{"kernel": v["kernel"], "launches_timed": v["launches_timed"],
                                                 "avg_launch_us": v["total_us"] / max(v["launches_timed"], 1),
                                                 "tflops": v["flops"] / max(v["total_us"], 1e-9) / 1e6}
                                                for k, v in kinds.items() if k != dom and v["launches_timed"]],
                         "algorithmic_flops_per_block": flops_per_block(width, is_fc)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
