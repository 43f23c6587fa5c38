#!/usr/bin/env python3
"""Benchmark of the PNN intra-prediction hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fc8|conv16|...] [--batch B] [--arithmetic f32|split]

One "step" = one pass of the hot path (L-context gather -> PNN -> +mean/clamp/round -> int32 Pel) over one
batch of synthetic transform blocks per GPU, inputs (reconstructed planes + TB descriptors) resident in
HBM.  Default workload = BASELINE.json configs[1]: 8x8 fully-connected PNN, batch 4096, 1 x MI355X.
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank processes its own batch
(independent blocks: no data-path collective, weak scaling) and the time is the max over ranks.

Rank 0 prints ONE JSON line of < 4 KB (build_line; tests/test_host.py checks the size on canned results).  Its top level
-- value, dtype, roofline, cpu_baseline -- is the workload on the REFERENCE's arithmetic, IEEE float32 (`precision` = 0,
pnn/components.py:169-176); `fast_arithmetic` is the same workload on the library's split-f16 mode, and `per_width` the
table BASELINE.json's metric asks for ("blocks/s (per width)": FC 4, FC 8, conv 16 / 32 / 64, both arithmetics, each checked
against the oracle).  Everything else -- every region's time, the CPU legs, the other kernels of the pass -- goes to
`bench_detail.json` (--detail-file).  The HM campaigns (configs[3] / [4]) run only under `--workload hm_kodak | hm_bsds`.
`--gpus N` without a launcher (WORLD_SIZE unset) starts its own ranks: `python -m torch.distributed.run` as a CHILD process.
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

RAMP_SECONDS = 0.4          # untimed device ramp-up before the warm-up steps (see measure)
REPEATS = 5                 # timed regions of K steps each; value = the median region
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense f32 matrix peak (256 CUs x 256 FLOP/clk x 2.4 GHz)
PEAK_F16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 matrix peak
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak (the roof a single-block call's parameter stream is priced against)
SUSTAIN_SECONDS = 2.0       # --sustained: one region of back-to-back steps per (workload, arithmetic), clocks sampled meanwhile
LINE_LIMIT = 4096           # the driver keeps the tail of stdout: the line must fit with room to spare
DTYPE = {0: "f32",
         1: "f32-class split: 2 x f16 per operand, 3 f16-MFMA products, f32 accumulate"}
DTYPE_LONG = {1: "f32 emulated: 2 x f16 per operand (hi + lo, 22-bit significand), 3 of the 4 partial products on f16 MFMA "
                 "(lo*lo dropped), f32 accumulate",
              0: "f32 (IEEE float32 operands on f32 MFMA, f32 accumulate: the reference's arithmetic)"}
ARITH = {"f32": 0, "split": 1}

WORKLOADS = {                      # name -> (width, is_fc, default batch per GPU, BASELINE.json config)
    "fc4": (4, True, 4096, "4x4 fully-connected PNN, batch 4096"),
    "fc8": (8, True, 4096, "configs[1]: 8x8 fully-connected PNN, batch 4096"),
    "conv16": (16, False, 1024, "configs[2]: 16x16 convolutional PNN, batch 1024"),
    "conv32": (32, False, 256, "32x32 convolutional PNN, batch 256"),
    "conv64": (64, False, 64, "64x64 convolutional PNN, batch 64"),
    "conv8": (8, False, 4096, "8x8 convolutional PNN, batch 4096"),
    "conv4": (4, False, 4096, "4x4 convolutional PNN, batch 4096"),
}
PER_WIDTH = ("fc4", "fc8", "conv16", "conv32", "conv64")   # the nets HM uses per width (pnn/PredictionNeuralNetwork.py:119-137, TComPrediction.cpp:130-171)
KERNELS = ((0, "tapgemm_f32_kernel (exact f32 on v_mfma_f32_32x32x2_f32, one wave per SIMD, LDS-DMA weight ring; FC nets: output layer fused in)"),
           (1, "tapgemm_splitk_kernel (f32 MFMA, small M)"),
           (2, "tapgemm_sp_kernel (split products on 3 x f16 MFMA 32x32x16, register-staged operands)"),
           (3, "convimg_sp_kernel (split-product MFMAs, feature maps resident in LDS)"),
           (4, "tapgemm_ring_kernel (split-product MFMAs; 4 MFMA + 4 loader waves, LDS-DMA ring; incl. the fused output layer)"),
           (5, "tapgemm_small_kernel (split-product MFMAs; one 32 x 32 tile per workgroup: small M)"))


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


class Workload:
    """Synthetic batch of one width, resident in HBM: an HD luminance plane of int32 Pel, TB descriptors with HM-style
    availability (30 % partial), seeded weights with the reference initialisers' statistics."""

    def __init__(self, name, batch, rank, local_rank):
        import torch
        from context_adaptive_neural_network_based_prediction_amd import _lib
        from tests import util
        self.name = name
        self.width, self.is_fc, self.default_batch, self.cfg_name = WORKLOADS[name]
        self.batch = batch or self.default_batch
        self.rank, self.local_rank = rank, local_rank
        self.L = L = _lib.lib()
        w, n = self.width, self.batch
        self.params = util.make_params(w, self.is_fc, seed=1, out_gain=30.0)
        plane_h, plane_w = 1088, 1920
        self.plane = util.make_plane(plane_h, plane_w, seed=100 + rank, pad=64)
        self.xs, self.ys, self.flags = util.make_tbs(plane_h, plane_w, w, n, seed=200 + rank, partial_fraction=0.3)
        units = 2 * w // 4
        tbs = (_lib.TbDev * n)()
        for i in range(n):
            assert L.pnn_make_tb_desc(ctypes.byref(tbs[i]), int(self.ys[i]) * self.plane.shape[1] + int(self.xs[i]), self.plane.shape[1],
                                      self.flags[i].ctypes.data_as(_lib.u8p), int(self.flags[i].sum()), units, units) == 0
        self.d_plane = torch.from_numpy(self.plane).cuda()
        self.d_tbs = torch.from_numpy(np.frombuffer(tbs, dtype=np.uint8).copy()).cuda()
        self.d_dst = torch.empty((n, w, w), dtype=torch.int32, device="cuda")
        above, left = util.make_contexts(w, n, seed=300 + rank)
        self.d_in = ((torch.from_numpy(util.flatten_fc(above, left)).cuda(),) if self.is_fc
                     else (torch.from_numpy(above).cuda(), torch.from_numpy(left).cuda()))
        self.d_out = torch.empty((n, w, w), dtype=torch.float32, device="cuda")
        self._oracle_pred = None

    def oracle_sample(self):
        """Oracle predictions of the first blocks of the batch (the checker of every GPU measurement on this workload)."""
        from oracle import pnn_oracle as O
        from tests import util
        if self._oracle_pred is None:
            n = min(self.batch, 4096 if self.is_fc else (512 if self.width <= 16 else 64))
            self._oracle_pred = O.predict_tbs(self.params, self.width, self.is_fc, self.plane, self.xs[:n], self.ys[:n], self.flags[:n], util.MEAN)
        return self._oracle_pred


def stagger(dist, fn):
    """Runs `fn` one rank after the other (first call of a workload = on-device tile autotune: every rank tunes on an
    otherwise idle node, as at N = 1, instead of N tuners timing launches at once)."""
    if dist is None:
        return fn()
    for r in range(dist.get_world_size()):
        if r == dist.get_rank():
            fn()
        dist.barrier()


def measure(wl, precision, steps, warmup, dist, repeats=REPEATS, check=True, sustain_s=0.0, ramp_s=RAMP_SECONDS, global_batch=None):
    """Times the hot path of workload `wl` on arithmetic `precision` (1: split products on f16 MFMA, 0: exact-f32 MFMA)
    and derives the dominant kernel's roofline from HIP events attached to every tap-GEMM launch."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, sharding
    L, w, n = wl.L, wl.width, wl.batch
    net = PredictionNeuralNetwork(n, w, wl.is_fc, params=wl.params, device=wl.local_rank)
    net.set_option("precision", precision)
    net.set_option("autotune", int(os.environ.get("PNN_AUTOTUNE", "1")))   # tile choice measured on the device during set-up
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step():
        rc = L.pnn_predict_tbs_device(net.ctx, w, wl.d_plane.data_ptr(), 4, wl.d_tbs.data_ptr(), n, wl.d_dst.data_ptr(), None, sp)
        if rc:
            raise RuntimeError(L.pnn_last_error(net.ctx))

    # Set-up, untimed and outside the W warm-up steps: the first calls autotune the tile configurations, and the device
    # needs a few hundred milliseconds of sustained work before it holds its clocks (measured: 0.126 ms per step right
    # after start, 0.113 ms once warm) -- run steps for RAMP_SECONDS so that W and K see the steady state.
    def first():
        step()
        torch.cuda.synchronize()
    stagger(dist, first)
    t_ramp = time.perf_counter()
    # ... and the shader clock the chip HOLDS under exactly this load is sampled meanwhile (sysfs, second half of the ramp): the split-f16
    # kernels run into the power cap and hold ~1.9 of the nominal 2.4 GHz, so their roof is quoted at both clocks
    smp = sharding.GpuClockSampler(wl.local_rank, period_s=0.01)
    with smp:
        while time.perf_counter() - t_ramp < ramp_s:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    held = sorted(smp.mhz[len(smp.mhz) // 2:])
    held_mhz = held[len(held) // 2] if held else None
    for _ in range(warmup):
        step()
    regions = [sharding.timed_steps(step, steps, torch.cuda.synchronize, dist, None if os.environ.get("PNN_BENCH_SHARE_GPU") == "1" else "cuda") for _ in range(repeats)]
    if L.pnn_check_range(net.ctx, sp, None) != 0:
        raise RuntimeError(L.pnn_last_error(net.ctx))
    stats = net.last_call_stats()
    elapsed = float(np.median(regions))
    world = dist.get_world_size() if dist is not None else 1
    # blocks the JOB processes per step: N batches of n (weak scaling) or the ONE batch the ranks share (strong: global_batch)
    job_blocks = float(global_batch) if global_batch else float(n) * world
    res = {
        "value": job_blocks * steps / elapsed, "unit": "blocks/s", "ms_per_step": 1e3 * elapsed / steps, "scaling": "strong" if global_batch else "weak",
        "global_batch": int(job_blocks),
        "dtype": DTYPE_LONG[precision], "precision": precision, "steps": steps, "workload": wl.name, "batch_per_gpu": n,
        "repeats": {"n": repeats, "regions_of_steps": steps, "value_is": "median region",
                    "blocks_per_s_min": job_blocks * steps / max(regions), "blocks_per_s_max": job_blocks * steps / min(regions),
                    "ms_per_step_all": [round(1e3 * r / steps, 5) for r in regions]},
        "launches_per_step": stats["launches"], "held_sclk_mhz": held_mhz,
    }
    gpu_pred = wl.d_dst.cpu().numpy()
    if sustain_s > 0:
        # ---- sustained: ONE region of back-to-back steps lasting >= sustain_s, clocks sampled from sysfs meanwhile
        from context_adaptive_neural_network_based_prediction_amd.sharding import GpuClockSampler
        k = max(steps, int(1.1 * sustain_s / (elapsed / steps)))
        with GpuClockSampler(wl.local_rank) as smp:
            t = sharding.timed_steps(step, k, torch.cuda.synchronize, dist, None if os.environ.get("PNN_BENCH_SHARE_GPU") == "1" else "cuda")
        res["sustained"] = {"value": float(n) * world * k / t, "unit": "blocks/s", "ms_per_step": 1e3 * t / k, "steps": k, "seconds": t,
                            "vs_burst": (float(n) * world * k / t) / res["value"], "gpu_clock": smp.summary()}

    # ---- roofline of the dominant GEMM kernel: the network alone on pre-gathered contexts, HIP events on the launch stream
    def net_only():
        if wl.is_fc:
            rc = L.pnn_predict_fc_device(net.ctx, w, wl.d_in[0].data_ptr(), n, wl.d_out.data_ptr(), sp)
        else:
            rc = L.pnn_predict_conv_device(net.ctx, w, wl.d_in[0].data_ptr(), wl.d_in[1].data_ptr(), n, wl.d_out.data_ptr(), sp)
        if rc:
            raise RuntimeError(L.pnn_last_error(net.ctx))

    for _ in range(3):
        net_only()
    reps = max(5, min(steps, 50))
    torch.cuda.synchronize()
    net.set_option("time_launches", 1)
    for _ in range(reps):
        net_only()
    torch.cuda.synchronize()
    net.set_option("time_launches", 0)
    nstats = net.last_call_stats()
    kinds = {}
    for kind, name in KERNELS:
        n_k, us_k, fl_k = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        L.pnn_launch_times(net.ctx, kind, ctypes.byref(n_k), ctypes.byref(us_k), ctypes.byref(fl_k))
        kinds[kind] = {"kernel": name, "launches_timed": n_k.value, "total_us": us_k.value, "flops": fl_k.value}
    dom = max(kinds, key=lambda k: kinds[k]["total_us"])
    fl_launch = kinds[dom]["flops"] / max(kinds[dom]["launches_timed"], 1)
    avg_s = kinds[dom]["total_us"] * 1e-6 / max(kinds[dom]["launches_timed"], 1)
    ach = fl_launch / avg_s / 1e12
    # peak: exact-f32 MFMA 157.3 TFLOP/s; the split-precision kernels issue three f16 MFMAs (2.5 PFLOP/s dense) per
    # algorithmic product, so their roof in algorithmic FLOPs is 2500 / 3.
    peak = PEAK_F32_MFMA_TFLOPS if dom < 2 else PEAK_F16_MFMA_TFLOPS / 3.0
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # PMC pass results (FETCH_SIZE x2 + WRITE_SIZE, per launch)
    if os.path.exists(pmc) and n == wl.default_batch:
        rec = json.load(open(pmc)).get(wl.name if precision == 1 else wl.name + "_f32")   # only the arithmetic it was measured on
        if rec:
            traffic, traffic_src = rec["bytes_per_launch"], rec["source"]
    gemm_us = sum(v["total_us"] for v in kinds.values()) / reps
    res["roofline"] = {
        "bound": "mfma", "kernel": kinds[dom]["kernel"], "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
        "traffic": traffic, "traffic_source": traffic_src, "flops_per_launch": fl_launch, "avg_launch_us": avg_s * 1e6,
        "launches_timed": kinds[dom]["launches_timed"],
        "peak_note": ("f16 dense MFMA peak 2500 / 3 MFMAs per algorithmic product" if dom >= 2 else "f32 dense MFMA peak at 2.4 GHz"),
        # the same fractions against the roof at the clock the chip held under this load (peak x held / 2400 MHz)
        "held_sclk_mhz": held_mhz, "frac_at_held_clock": (ach / (peak * held_mhz / 2400.0)) if held_mhz else None,
        "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
        "gemm_launches_per_pass": nstats["gemm_launches"], "non_gemm_launches_per_pass": nstats["launches"] - nstats["gemm_launches"],
        "gemm_us_per_pass": gemm_us,
        "whole_pass": {"tflops": flops_per_block(w, wl.is_fc) * n / (1e-3 * res["ms_per_step"]) / 1e12,
                       "frac_of_peak": flops_per_block(w, wl.is_fc) * n / (1e-3 * res["ms_per_step"]) / 1e12 / peak,
                       # the same step priced on the multiply-adds its tap GEMMs ISSUED: position-major tiles skip the taps that only
                       # meet SAME padding (SURVEY 8(d) counts them), so this -- not frac_of_peak -- is matrix-pipe utilisation
                       "issued_frac_of_peak": nstats["gemm_flops_issued"] / (1e-3 * res["ms_per_step"]) / 1e12 / peak,
                       "issued_over_algorithmic": nstats["gemm_flops_issued"] / max(nstats["gemm_flops"], 1.0)},
        "other_gemm_kernels": [{"kernel": v["kernel"], "launches_timed": v["launches_timed"],
                                "avg_launch_us": v["total_us"] / max(v["launches_timed"], 1),
                                "tflops": v["flops"] / max(v["total_us"], 1e-9) / 1e6}
                               for k, v in kinds.items() if k != dom and v["launches_timed"]],
        "algorithmic_flops_per_block": flops_per_block(w, wl.is_fc)}

    # ---- parity of what the timed steps produced, against the oracle (rank 0, N = 1: the checker, never the product)
    if check:
        want = wl.oracle_sample()
        m = want.shape[0]
        diff = np.abs(gpu_pred[:m].astype(np.int64) - want)
        res["max_abs_lsb_vs_oracle"] = int(diff.max())
        res["parity_detail"] = {"pixels_compared": int(diff.size), "pixels_differing": int((diff != 0).sum())}
        # pred-PSNR (tools/tools.py:364-401: 10 log10(255^2 / (mse + 1e-6))) of the predictions against the blocks they
        # predict, GPU and oracle; BASELINE.json's "pred-PSNR delta vs ref" with the oracle standing in for the reference
        org = np.stack([wl.plane[int(y):int(y) + w, int(x):int(x) + w] for x, y in zip(wl.xs[:m], wl.ys[:m])]).astype(np.float64)
        psnr = lambda p: float(10.0 * np.log10(255.0 ** 2 / (np.mean((p.astype(np.float64) - org) ** 2) + 1e-6)))
        res["pred_psnr"] = {"gpu_db": psnr(gpu_pred[:m]), "oracle_db": psnr(want), "delta_db": psnr(gpu_pred[:m]) - psnr(want),
                            "note": "random-init weights: only the DELTA means anything.  The trained production weights (FC 4/8, conv 16/32/64) "
                                    "and the Kodak / BSDS pictures are not in the reference checkout, so the paper's pred-PSNR cannot be reproduced here"}
    net.close()
    return res


def single_block_calls(names, device, precision=0, calls=300, warm=80):
    """The reference's own call shape (freezing_graph_pnn.py:100-102, TComPrediction.cpp:572-579,601-608): ONE block per call through the
    host-array entry point HM's Session::Run look-alike binds (pnn_predict_fc / pnn_predict_conv at n = 1: staging, net, wait, copy
    out).  Per workload name: median microseconds of `calls` calls after `warm`, the blocks/s that is, and the parameter stream it
    amounts to (a single-block call is bound by streaming the net's parameters) against the 8 TB/s HBM roof."""
    from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
    from tests import util
    L = _lib.lib()
    out = {}
    for name in names:
        w, fc, _, _ = WORKLOADS[name]
        params = util.make_params(w, fc, 1, out_gain=util.out_gain(w, fc))
        net = PredictionNeuralNetwork(1, w, fc, params=params, device=device)
        net.set_option("precision", precision)
        a, l = util.make_contexts(w, 1, 2)
        x = np.ascontiguousarray(util.flatten_fc(a, l) if fc else a, np.float32)
        l = np.ascontiguousarray(l, np.float32)
        y = np.zeros((w, w), np.float32)
        xp, lp, yp = x.ctypes.data_as(_lib.f32p), l.ctypes.data_as(_lib.f32p), y.ctypes.data_as(_lib.f32p)
        call = (lambda: L.pnn_predict_fc(net.ctx, w, xp, 1, yp)) if fc else (lambda: L.pnn_predict_conv(net.ctx, w, xp, lp, 1, yp))
        for _ in range(warm):
            if call():
                raise RuntimeError(L.pnn_last_error(net.ctx))
        ts = np.empty(calls)
        for i in range(calls):
            t0 = time.perf_counter()
            call()
            ts[i] = time.perf_counter() - t0
        us = float(np.median(ts)) * 1e6
        out[name] = {"us": us, "us_p10": float(np.percentile(ts, 10)) * 1e6, "us_p90": float(np.percentile(ts, 90)) * 1e6, "calls": calls,
                     "blocks_per_s": 1e6 / us, "param_bytes": int(params.nbytes), "param_gbps": params.nbytes / us / 1e3,
                     "frac_of_hbm": params.nbytes / us / 1e3 / PEAK_HBM_GBPS, "launches": net.last_call_stats()["launches"],
                     "entry_point": "pnn_predict_fc" if fc else "pnn_predict_conv"}
        net.close()
    return out


def cpu_leg_worker(kind, workload, batch1, threads, budget_s):
    """One CPU leg in its own process (`bench.py --cpu-leg ...`): a clean thread environment -- two OpenMP runtimes with
    256 spinning threads each in one process (oracle + PyTorch, beside the HIP runtime's own threads) measured each other
    rather than the nets.  Prints one JSON object."""
    from tests import util
    w, fc, default_batch, _ = WORKLOADS[workload]
    params = util.make_params(w, fc, seed=1, out_gain=30.0)
    nb = default_batch                                  # the GPU batch
    above, left = util.make_contexts(w, nb, seed=7)
    ctx = util.flatten_fc(above, left) if fc else None
    if kind == "oracle":
        from oracle import pnn_oracle as M
    else:
        import torch
        from tests import torch_formulation as M
        torch.set_num_threads(threads)
    if batch1:
        k1 = 16
        one = (lambda i: M.fc_forward(params, w, ctx[i:i + 1])) if fc else (lambda i: M.conv_forward(params, w, above[i:i + 1], left[i:i + 1]))
        fn, per_call = (lambda: [one(i) for i in range(k1)]), k1
    else:
        fn, per_call = ((lambda: M.fc_forward(params, w, ctx)) if fc else (lambda: M.conv_forward(params, w, above, left))), nb
    fn()                                                # warm-up (thread pools, page faults)
    ts, t0 = [], time.perf_counter()
    while (len(ts) < 5 and time.perf_counter() - t0 < 3 * budget_s) or (time.perf_counter() - t0 < budget_s and len(ts) < 500):
        a = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - a)
    print(json.dumps({"blocks_per_s": per_call / float(np.median(ts)), "runs": len(ts), "blocks_per_run": per_call, "threads": threads,
                      "blocks_per_s_spread": [per_call / float(np.max(ts)), per_call / float(np.min(ts))]}))   # slowest / fastest run of this leg


def cpu_budget():
    """CPUs this process may actually keep busy (affinity cut to the cgroup quota): sharding.cpu_budget."""
    from context_adaptive_neural_network_based_prediction_amd import sharding
    return sharding.cpu_budget()


def cpu_legs(workload, budget_s=1.5, full=False):
    """the same graph on this box's host cores -- the oracle (a port; TF 1.x cannot be installed) and
    an independent PyTorch-CPU formulation (oneDNN / MKL), each batched (the GPU batch where the oracle finishes it in seconds)
    and at batch 1 sequential (what HM does per TB).  Bounded samples: every leg runs for about `budget_s` in its own process.
    Thread counts: min(64, the CPUs the job may use -- cpu_budget()) batched and 8 at batch 1; `--cpu-legs-full` tries the candidate
    lists {all, 64, 32, 16} / {8, 1} within that budget."""
    import subprocess
    ncores = cpu_budget()

    def run(kind, batch1, threads):
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", kind, "--workload", workload, "--leg-batch1", str(int(batch1)),
                            "--leg-threads", str(threads), "--leg-budget", str(budget_s)], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": r.stderr[-400:]}
        return json.loads(r.stdout.strip().splitlines()[-1])

    batched_threads = sorted({ncores, min(ncores, 64), min(ncores, 32), min(ncores, 16)}, reverse=True) if full else [min(ncores, 64)]
    batch1_threads = sorted({min(ncores, 8), 1}, reverse=True) if full else [min(ncores, 8)]
    legs = {}
    for kind, tag in (("oracle", "oracle"), ("torch", "torch_cpu")):
        cands = [run(kind, False, t) for t in batched_threads]
        legs[tag + "_batched"] = max(cands, key=lambda r: r.get("blocks_per_s", 0.0))
        legs[tag + "_batched"]["threads_tried"] = [c.get("threads") for c in cands]
        cands = [run(kind, True, t) for t in batch1_threads]
        legs[tag + "_batch1"] = max(cands, key=lambda r: r.get("blocks_per_s", 0.0))
    best = max(("oracle_batched", "torch_cpu_batched"), key=lambda k: legs[k].get("blocks_per_s", 0.0))
    best1 = max(("oracle_batch1", "torch_cpu_batch1"), key=lambda k: legs[k].get("blocks_per_s", 0.0))
    return {"value": legs[best].get("blocks_per_s"), "unit": "blocks/s", "cores": legs[best].get("threads", ncores), "host_cores": os.cpu_count(), "cpu_quota": ncores, "kind": "port",
            "value_leg": best, "batch1_value": legs[best1].get("blocks_per_s"), "batch1_leg": best1, "batch1_cores": legs[best1].get("threads"),
            "sample": "batches of %s blocks (batch-1: 16 single-block calls in sequence, what HM issues per TB); median of the runs that fit ~%.1f s "
                      "per leg, each leg its own process; value = the faster of oracle/pnn_oracle.c (OpenMP, -O3 -mavx2 -mfma) and a PyTorch-CPU "
                      "(oneDNN) formulation -- stand-ins for the TF-1.9 CPU path, which cannot be installed" % (legs["oracle_batched"].get("blocks_per_run"), budget_s),
            "legs": legs}


# ---------------------------------------------------------------------------------------------------------------------
# roofline.traffic measured in THIS run: two child `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE: they do not fit one
# pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") over a few steps of the same workload and arithmetic, rule-based tiles.
# Counter-only passes (no tracing); the children are started as child processes, never exec'ed.
# ---------------------------------------------------------------------------------------------------------------------
GEMM_NAMES = {0: ("tapgemm_kernel", "tapgemm32_kernel", "tapgemm_splitk_kernel", "tapgemm_f32"),
              1: ("tapgemm_ring_kernel", "tapgemm_sp_kernel", "convimg_sp_kernel")}


def pmc_child(workload, batch, precision, steps):
    """`bench.py --pmc-child`: a few steps of the hot path and nothing else (the profiler's child)."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork
    wl = Workload(workload, batch, 0, 0)
    net = PredictionNeuralNetwork(wl.batch, wl.width, wl.is_fc, params=wl.params, device=0)
    net.set_option("precision", precision)
    net.set_option("autotune", 0)                     # no tuning launches among the counted dispatches
    for _ in range(steps):
        rc = wl.L.pnn_predict_tbs_device(net.ctx, wl.width, wl.d_plane.data_ptr(), 4, wl.d_tbs.data_ptr(), wl.batch, wl.d_dst.data_ptr(), None, None)
        if rc:
            raise RuntimeError(wl.L.pnn_last_error(net.ctx))
    torch.cuda.synchronize()
    net.close()


def parse_pmc_dir(path, counter, names):
    """Mean per dispatch of `counter` over the dispatches of the kernels in `names`, and the dispatch count."""
    import csv
    import glob
    per = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").replace("pnn::", "")
            if r["Counter_Name"] == counter and k.startswith(names):
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return (sum(per.values()) / len(per), len(per)) if per else (None, 0)


def live_traffic(workload, batch, precision, steps=3, timeout=90):
    """HBM bytes per GEMM launch from this run's own counter passes: 2 x FETCH_SIZE (gfx950 tallies 128-byte reads as 64) +
    WRITE_SIZE, KiB -> bytes (MI355X_MICROARCH.md, HBM / rocprofv3 section).  None when rocprofv3 is absent or a pass fails."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    vals, work = {}, tempfile.mkdtemp(prefix="pnn_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, ctr)
            r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                                "--workload", workload, "--batch", str(batch), "--arithmetic", "f32" if precision == 0 else "split", "--steps", str(steps)],
                               env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", capture_output=True, text=True, timeout=timeout)
            v, nd = parse_pmc_dir(d, ctr, GEMM_NAMES[precision])
            if r.returncode != 0 or v is None:
                return {"error": "rocprofv3 --pmc %s: rc %d, %d GEMM dispatches; %s" % (ctr, r.returncode, nd, r.stderr[-300:])}
            vals[ctr] = (v, nd)
    except Exception as e:                            # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return {"bytes_per_launch": (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0, "fetch_kib_raw_mean": vals["FETCH_SIZE"][0],
            "write_kib_mean": vals["WRITE_SIZE"][0], "launches": vals["FETCH_SIZE"][1],
            "source": "this run: child rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate, counters only) over %d steps, rule-based tiles; "
                      "mean over the pass's GEMM dispatches of 2 x FETCH_SIZE + WRITE_SIZE" % steps}


def preflight_error_line(args, have):
    """The ONE line of a `--gpus N` run on a node that shows fewer than N devices: same shape as a result line, value null, the reason
    spelled out -- so a driver that parses the last JSON line of stdout records WHY there is no number instead of a traceback."""
    return json.dumps({"metric": "pnn_intra_pred_blocks_per_s", "value": None, "unit": "blocks/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                       "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                       "config": {"workload": WORKLOADS[args.workload][3] if args.workload in WORKLOADS else args.workload},
                       "error": "--gpus %d: this node exposes %d HIP device(s) (torch.cuda.device_count()); nothing was launched" % (args.gpus, have),
                       "devices_visible": have}, separators=(",", ":"))


def self_launch(args):
    """`python bench.py --gpus N` run plainly: this process stays off the GPU and starts the N ranks as a CHILD
    (`python -m torch.distributed.run`, one process per GPU, rendezvous on 127.0.0.1), relays rank 0's JSON line and
    exits with the child's code.  Never os.exec*: a process that may have touched HIP must not be replaced."""
    import socket
    import subprocess
    import torch
    if os.environ.get("PNN_BENCH_SHARE_GPU") != "1":
        have = torch.cuda.device_count()             # counting devices does not initialise HIP on this image
        if have < args.gpus:
            print(preflight_error_line(args, have))
            sys.stdout.flush()
            raise SystemExit(1)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line:
        print(line)
    sys.stdout.flush()
    raise SystemExit(r.returncode if r.returncode or line else 1)


def hm_campaigns(which, devices, quick=False, pictures="synthetic", cpu_leg=True, arithmetic="f32"):
    """BASELINE.json configs[3] / configs[4] at their stated picture counts through the reference's own HM binaries
    (tools/hm/campaign.py; built by __graft_entry__.build() where /root/reference exists, they travel with the tree), and -- the
    cpu_baseline leg -- the first pictures of the same campaign with the PNN answered on HOST CORES (the reference's route: inference
    on the CPU; here the CPU oracle behind the same batching service, tools/hm/cpu_pnn_service.py), beside the same pictures on the GPU.
    Returns {name: record}; a missing binary or a failed run is recorded, never raised."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools", "hm"))
    out = {}
    try:
        import campaign
    except Exception as e:                            # noqa: BLE001
        return {"error": "tools/hm/campaign.py not importable: %r" % (e,)}
    if not campaign.binaries_present():
        return {"error": "tools/hm/_build/*/TApp{Encoder,Decoder}Static are missing (built where /root/reference exists: `make -C tools/hm`)"}
    for name in which:
        work = tempfile.mkdtemp(prefix="pnn_bench_hm_")
        try:
            rec = campaign.run_campaign(name, work, devices, pictures=(4 if quick else None), timeout=300, picture_set=pictures, arithmetic=arithmetic)   # per codec process: a wedged service must not hold the line for long
            if cpu_leg:
                k = 2
                small = campaign.run_campaign(name, os.path.join(work, "gpu_small"), devices[:1], pictures=k, timeout=300, picture_set=pictures, yardstick=False, arithmetic=arithmetic)
                cpu = campaign.run_campaign(name, os.path.join(work, "cpu"), devices[:1], pictures=k, timeout=900, picture_set=pictures, yardstick=False, backend="cpu")
                keep = ("pictures", "wall_s_all_encodes_and_decodes", "pictures_per_s", "enc_wall_s", "dec_wall_s", "every_decode_equals_its_encoder", "bits_total", "service")
                rec["cpu_pnn"] = {
                    "what": "the first %d pictures with the PNN answered on host cores (CPU oracle, OpenMP, behind the same batching service) and, beside it, "
                            "the same %d pictures on one MI355X; both with %d encodes in flight" % (k, k, k),
                    "cores": min(8, os.cpu_count() or 8), "host_cores": os.cpu_count(),
                    "cpu": {kk: cpu.get(kk) for kk in keep}, "gpu_same_sample": {kk: small.get(kk) for kk in keep},
                    "pictures_per_s": cpu.get("pictures_per_s"), "gpu_same_sample_pictures_per_s": small.get("pictures_per_s"),
                    "gpu_over_cpu_wall": round(cpu["wall_s_all_encodes_and_decodes"] / small["wall_s_all_encodes_and_decodes"], 2),
                    "same_bits": cpu.get("bits_total") == small.get("bits_total"),
                    "sample": "first %d pictures of the campaign, encode + decode" % k}
            out[name] = rec
        except Exception as e:                        # noqa: BLE001
            out[name] = {"error": repr(e)[:2000]}
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return out


def natural_pred_psnr(device):
    """BASELINE.json's "pred-PSNR delta vs ref" where it means something: the reference's two TRAINED checkpoints (convolutional 4x4 / 8x8,
    tests/golden/conv{4,8}_single.pnnw) on 2500 natural contexts per width (tests/golden/natural_luma.npz), prediction PSNR
    (tools/tools.py:364-401) against the blocks they predict -- the HIP path on both arithmetics beside the oracle."""
    from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, weights as wts
    from oracle import pnn_oracle as O
    from tests import test_natural as tn, util
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "natural_luma.npz")):     # generated by __graft_entry__.build() where the reference checkout exists
        return None
    out = {}
    for w in (4, 8):
        flat, _, _ = wts.load_pnnw(os.path.join(ROOT, "tests", "golden", "conv%d_single.pnnw" % w))
        a8, l8, tgt = tn.natural_contexts(w, tn.N_CONTEXTS)
        above = a8.astype(np.float32) - np.float32(util.MEAN)
        left = l8.astype(np.float32) - np.float32(util.MEAN)
        want = O.epilogue(O.conv_forward(flat, w, above, left), util.MEAN)
        rec = {"contexts": int(tn.N_CONTEXTS), "oracle_db": tn.psnr(want, tgt)}
        for arith, precision in (("f32", 0), ("split", 1)):
            net = PredictionNeuralNetwork(tn.N_CONTEXTS, w, False, params=flat, device=device)
            net.set_option("precision", precision)
            got = net.predict_pel(above, left)
            net.close()
            rec[arith + "_db"] = tn.psnr(got, tgt)
            rec[arith + "_max_abs_lsb_vs_oracle"] = int(np.abs(got.astype(np.int64) - want).max())
        out[str(w)] = rec
    return out


def _r(x, nd=4):
    """Rounded to `nd` significant digits (the line is read by people and by a 4 KB tail buffer)."""
    if x is None or isinstance(x, (str, bool, int)):
        return x
    return float("%.*g" % (nd, x))


def compact(res, with_ms=True):
    """One measurement as the line carries it.  issued_frac only where position-major tiles skipped something (conv nets); the held
    clock and the fraction against the roof at that clock only for the split-f16 mode (the exact-f32 kernels hold the nominal clock)."""
    rf = res["roofline"]
    out = {"value": _r(res["value"], 5), "ms_per_step": _r(res["ms_per_step"], 5), "frac": _r(rf["frac"], 3),
           "pass_frac": _r(rf["whole_pass"]["frac_of_peak"], 3), "launches": res["launches_per_step"], "lsb": res.get("max_abs_lsb_vs_oracle")}
    if not with_ms:                                  # per_width rows: value and batch say it (the line is capped at LINE_LIMIT)
        del out["ms_per_step"]
    if (rf["whole_pass"].get("issued_over_algorithmic") or 1.0) < 0.999:
        out["issued_frac"] = _r(rf["whole_pass"].get("issued_frac_of_peak"), 3)
    if res.get("precision") == 1 and rf.get("frac_at_held_clock"):
        out["held_mhz"], out["frac_held"] = _r(rf.get("held_sclk_mhz"), 4), _r(rf.get("frac_at_held_clock"), 3)
    return out


def build_line(main_res, world, steps, warmup, cfg_name, fast=None, per_width=None, cpu=None, cpu_conv16=None, detail_file=None, extra_config=None,
               rccl_ranks_seen=None, natural=None, single=None):
    """The ONE JSON line of rank 0 (< LINE_LIMIT bytes).  `main_res` / `fast` / `per_width[name][arith]` are measure() results;
    `single` = single_block_calls() results by workload name (exact f32)."""
    rf = main_res["roofline"]
    out = {
        "metric": "pnn_intra_pred_blocks_per_s", "value": _r(main_res["value"], 6), "unit": "blocks/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": _r(main_res["ms_per_step"], 6), "higher_is_better": True, "scaling": main_res.get("scaling", "weak"), "vs_baseline": None,
        "dtype": DTYPE[main_res["precision"]], "data": "synthetic",
        "config": {"workload": cfg_name, "batch_per_gpu": main_res["batch_per_gpu"], "global_batch": main_res.get("global_batch"), "path": "gather + net + HM epilogue (pnn_predict_tbs_device)",
                   "weights": "seeded random init (reference initialisers' statistics)", "parallelism": "independent blocks sharded over ranks, no collective",
                   "timing": "%d regions of K steps, median" % main_res["repeats"]["n"]},
        "max_abs_lsb_vs_oracle": main_res.get("max_abs_lsb_vs_oracle"),
        "pred_psnr_delta_db": _r((main_res.get("pred_psnr") or {}).get("delta_db"), 3),
        "roofline": {"bound": "mfma", "kernel": rf["kernel"].split(" (")[0], "achieved": _r(rf["achieved"]), "peak": _r(rf["peak"]), "unit": "TFLOP/s",
                     "frac": _r(rf["frac"], 3), "traffic": _r(rf["traffic"]), "traffic_source": (rf.get("traffic_source") or "")[:60] or None,
                     "flops_per_launch": _r(rf["flops_per_launch"]), "avg_launch_us": _r(rf["avg_launch_us"]), "launches_timed": rf["launches_timed"],
                     "whole_pass_frac": _r(rf["whole_pass"]["frac_of_peak"], 3)},
        "cpu_baseline": None,
    }
    if extra_config:
        out["config"].update(extra_config)
    if cpu:
        out["cpu_baseline"] = {k: (_r(cpu.get(k)) if k != "sample" else cpu[k][:150]) for k in
                               ("value", "unit", "cores", "host_cores", "cpu_quota", "kind", "value_leg", "batch1_value", "batch1_cores", "sample")}
        if cpu.get("value"):
            out["cpu_baseline"]["gpu_over_cpu"] = _r(main_res["value"] / world / cpu["value"], 3)
    if cpu_conv16 and cpu_conv16.get("value"):
        out["cpu_baseline"]["conv16"] = {"value": _r(cpu_conv16["value"]), "cores": cpu_conv16.get("cores"), "value_leg": cpu_conv16.get("value_leg"),
                                         "batch1_value": _r(cpu_conv16.get("batch1_value"))}
    if fast:
        out["fast_arithmetic"] = dict(compact(fast), dtype=DTYPE[1], peak=_r(fast["roofline"]["peak"]))
    if per_width:
        tab = {}
        for name in PER_WIDTH:
            if name not in per_width:
                continue
            w, fc, _, _ = WORKLOADS[name]
            row = {"arch": "fc" if fc else "conv", "batch": next(iter(per_width[name].values()))["batch_per_gpu"]}
            for arith, r in per_width[name].items():
                row[arith] = compact(r, with_ms=False)
            if single and name in single:
                row["single_block_us"] = _r(single[name]["us"], 3)     # one block per host call, exact f32: the reference's call shape
            tab[str(w)] = row
        out["per_width"] = tab
        out["per_width_note"] = "f32 vs 157.3, split vs 2500/3 TFLOP/s; frac: dominant GEMM, pass_frac: whole step, issued_frac: on issued MACs, frac_held: vs roof at held_mhz; single_block_us: 1 block per host call, f32"
    if single:
        # the reference's call shape -- one block per Session::Run -- beside the CPU's batch-1 leg of the same net
        sb = {}
        for name, cb in (("fc8", cpu), ("conv16", cpu_conv16)):
            if name not in single:
                continue
            r = single[name]
            sb[name] = {"us": _r(r["us"], 3), "blocks_per_s": _r(r["blocks_per_s"], 4), "param_gbps": _r(r["param_gbps"], 3), "frac_of_hbm": _r(r["frac_of_hbm"], 2),
                        "gpu_over_cpu_batch1": _r(r["blocks_per_s"] / cb["batch1_value"], 3) if cb and cb.get("batch1_value") else None}
        out["single_block"] = sb
    if natural:
        out["natural_pred_psnr_db"] = {w: {"gpu_f32": _r(v.get("f32_db"), 5), "gpu_split": _r(v.get("split_db"), 5), "oracle": _r(v.get("oracle_db"), 5)} for w, v in natural.items()}
    if rccl_ranks_seen is not None:
        out["rccl_ranks_seen"] = rccl_ranks_seen
    if detail_file:
        out["detail_file"] = detail_file
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:                       # never let bulk push the head of the line out of the driver's tail buffer
        for k in ("per_width_note", "pred_psnr_delta_db", "detail_file"):
            out.pop(k, None)
        if out.get("cpu_baseline"):
            out["cpu_baseline"].pop("sample", None)
        line = json.dumps(out, separators=(",", ":"))
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fc8", choices=sorted(WORKLOADS) + ["hm_kodak", "hm_bsds"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="--gpus N > 1: weak = every rank its own batch of the workload's size (the contract's default); strong = ONE batch of that "
                         "size split over the ranks (sharding.shard_bounds), at least 200 steps per region")
    ap.add_argument("--force-dist", default=None, choices=["nccl", "gloo"],
                    help="--gpus 1 only: join a ONE-rank process group anyway and run the barrier, the max-over-ranks clock, the device census and a "
                         "gather of the predictions through it -- with nccl the only execution of the RCCL branch a one-GPU box allows")
    ap.add_argument("--arithmetic", default=None, choices=sorted(ARITH), help="top-level arithmetic (default f32 = the reference's; env PNN_PRECISION=1 -> split)")
    ap.add_argument("--hm-quick", action="store_true", help=argparse.SUPPRESS)   # 4 pictures per campaign (plumbing tests)
    ap.add_argument("--hm-pictures", default="synthetic", choices=["synthetic", "natural"], help="hm_* workloads: picture set")
    ap.add_argument("--batch", type=int, default=0, help="blocks per GPU per step (0 = the workload's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-legs-full", action="store_true", help="CPU legs at every candidate thread count (slower)")
    ap.add_argument("--no-extras", action="store_true", help="only the top-level measurement (no fast_arithmetic / per_width / live traffic)")
    ap.add_argument("--no-per-width", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from profiles/pmc_traffic.json instead of this run's own counter passes")
    ap.add_argument("--sustained", action="store_true", help="add a >= %.0f s region per measurement (detail file)" % SUSTAIN_SECONDS)
    ap.add_argument("--no-sustained", action="store_true", help=argparse.SUPPRESS)   # accepted, ignored (the default since round 4)
    ap.add_argument("--no-hm", action="store_true", help=argparse.SUPPRESS)          # accepted, ignored (campaigns run only under --workload hm_*)
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"))
    ap.add_argument("--cpu-leg", default=None, help=argparse.SUPPRESS)       # internal: one CPU-baseline leg, see cpu_leg_worker
    ap.add_argument("--leg-batch1", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--leg-threads", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--leg-budget", type=float, default=3.0, help=argparse.SUPPRESS)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # internal: the profiled child of live_traffic
    args = ap.parse_args()
    if args.cpu_leg:
        return cpu_leg_worker(args.cpu_leg, args.workload, bool(args.leg_batch1), args.leg_threads, args.leg_budget)
    precision = ARITH[args.arithmetic] if args.arithmetic else int(os.environ.get("PNN_PRECISION", "0"))
    if args.pmc_child:
        return pmc_child(args.workload, args.batch, precision, args.steps)

    from context_adaptive_neural_network_based_prediction_amd import sharding
    if args.workload.startswith("hm_"):
        # configs[3] / configs[4]: whole encodes through the reference's HM binaries, one batching service per device; this
        # process never touches the GPU (the services and the codecs are child processes)
        name = args.workload[3:]
        rec = hm_campaigns([name], list(range(args.gpus)), args.hm_quick, args.hm_pictures, cpu_leg=not args.no_cpu_baseline,
                           arithmetic="f32" if precision == 0 else "split")
        r = rec.get(name, rec)
        ok = "error" not in r
        with open(args.detail_file, "w") as f:
            json.dump({"hm": rec}, f, indent=1)
        cpu_pnn = r.get("cpu_pnn") or {}
        print(json.dumps({
            "metric": "pnn_intra_pred_blocks_per_s", "value": _r(r["service"]["pnn_blocks_per_s_over_the_wall"], 6) if ok else None, "unit": "blocks/s",
            "n_gpus": args.gpus, "steps": 1, "warmup": 0, "ms_per_step": _r(1e3 * r["wall_s_all_encodes_and_decodes"], 6) if ok else None,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE[precision], "data": "synthetic",
            "config": {"workload": r.get("config", args.workload), "step": "all encodes + decodes of the campaign", "pictures": r.get("pictures"),
                       "picture_set": args.hm_pictures, "parallelism": "independent encodes dealt over one batching service per device, no collective"},
            "hm": {k: r.get(k) for k in ("variant", "pictures", "picture_set", "arithmetic", "wall_s_all_encodes_and_decodes", "pictures_per_s", "wall_vs_regular",
                                         "every_decode_equals_its_encoder", "arithmetic_tags", "service_start_s", "bits_total", "host_cpu", "error")} if isinstance(r, dict) else None,
            "roofline": None,
            "cpu_baseline": {"value": cpu_pnn.get("pictures_per_s"), "unit": "pictures/s", "cores": cpu_pnn.get("cores"), "kind": "port",
                             "sample": cpu_pnn.get("sample"), "gpu_same_sample": cpu_pnn.get("gpu_same_sample_pictures_per_s"),
                             "gpu_over_cpu_wall": cpu_pnn.get("gpu_over_cpu_wall"), "same_bits": cpu_pnn.get("same_bits"),
                             "hm_16_15_regular_pictures_per_s": (r.get("yardstick_hm_16_15_regular") or {}).get("pictures_per_s")} if ok else None,
            "detail_file": os.path.basename(args.detail_file)}, separators=(",", ":")))
        raise SystemExit(0 if ok else 1)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)                     # before anything initialises HIP in this process
    import torch

    rank, local_rank, world = sharding.rank_env()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    share = os.environ.get("PNN_BENCH_SHARE_GPU") == "1"
    # next to the GPU before the first HIP call: host cores of the device's NUMA node (sysfs only)
    bound = sharding.bind_to_gpu_numa(0 if share else local_rank) if world > 1 else None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # Plumbing check on a ONE-GPU box (tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu): PNN_BENCH_SHARE_GPU=1 puts
    # every rank on device 0 and joins them over gloo (RCCL refuses two ranks on one device).  Never a measurement.
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.force_dist and world != 1:
        raise SystemExit("--force-dist is the ONE-rank form (--gpus 1)")
    backend = args.force_dist or ("gloo" if share else "nccl")
    dist = sharding.init_ranks(backend, None if (share or backend == "gloo") else torch.device("cuda", local_rank), force=bool(args.force_dist))   # "nccl" = RCCL on ROCm; None at N = 1
    ranks_seen = None
    if dist is not None:
        try:
            ndev = sharding.count_distinct_devices(dist, local_rank, share)
        except Exception as e:                        # noqa: BLE001 -- an extra of the line must never cost the measurement
            sys.stderr.write("bench.py: count_distinct_devices failed: %r\n" % (e,))
            ndev = None
        ranks_seen = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "devices": ndev}
        if args.force_dist:
            # ... and the optional all-gather of the predictions through the same group, on the device for RCCL
            probe = torch.arange(3 * 64, dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu").reshape(3, 8, 8)
            back = sharding.gather_predictions(probe, 3, dist)
            ranks_seen["gather_predictions_ok"] = bool(torch.equal(back.cpu(), probe.cpu()))
            ranks_seen["forced_single_rank_group"] = True

    t_start = time.perf_counter()
    strong = args.scaling == "strong" and world > 1
    global_batch = args.batch or WORKLOADS[args.workload][2]
    if strong:
        # ONE batch split over the ranks: rank r takes the contiguous shard shard_bounds(global_batch, r, world); a step of the job is
        # still one pass over the whole batch, so `value` = global_batch * steps / (slowest rank's time).  Shards of a bench batch are
        # short (FC 8x8: 512 blocks at N = 8, ~60 us of device time), so regions are at least 200 steps
        _, mine = sharding.strong_shard(global_batch, rank, world)
        args.steps = max(args.steps, 200)
        wl = Workload(args.workload, mine, rank, local_rank)
    else:
        wl = Workload(args.workload, args.batch, rank, local_rank)
    single = world == 1
    sustain = SUSTAIN_SECONDS if args.sustained else 0.0
    main_res = measure(wl, precision, args.steps, args.warmup, dist, check=(rank == 0 and single), sustain_s=sustain,
                       global_batch=global_batch if strong else None)
    detail = {"cmd": " ".join(sys.argv), "n_gpus": world, "main": main_res,
              "config": {"rank_placement": ("rank 0 bound to cpus %s (its GPU's NUMA node)" % bound) if bound else "no NUMA binding (one node / not exposed)",
                         "tile_autotune": "split kernels: on first use, before the warm-up steps; f32 kernels: rule-based",
                         "device_ramp_s": RAMP_SECONDS, "share_gpu_plumbing_check": share}}
    fast, per_width, cpu, cpu16 = None, None, None, None
    if single and not args.no_extras:
        k_pw = min(args.steps, 20)
        per_width = {args.workload: {("f32" if precision == 0 else "split"): main_res}} if args.workload in PER_WIDTH else {}
        fast = measure(wl, 1 - precision, args.steps, args.warmup, None, sustain_s=sustain)
        if args.workload in PER_WIDTH:
            per_width[args.workload]["f32" if precision == 1 else "split"] = fast
        if precision == 1:                            # `fast_arithmetic` is always the split mode; with --arithmetic split the f32 twin sits in per_width
            fast = None
        if not args.no_per_width:
            for name in PER_WIDTH:
                if name == args.workload:
                    continue
                wn = Workload(name, 0, rank, local_rank)
                per_width[name] = {a: measure(wn, ARITH[a], k_pw, args.warmup, None, repeats=3, ramp_s=0.25, sustain_s=sustain) for a in ("f32", "split")}
                del wn
                torch.cuda.empty_cache()
        detail["per_width"] = per_width
        detail["fast_arithmetic"] = fast
    single_res = None
    if rank == 0 and single and not args.no_extras:
        try:
            single_res = single_block_calls(PER_WIDTH, local_rank)
        except Exception as e:                        # noqa: BLE001 -- an extra of the line must never cost the measurement
            detail["single_block_error"] = repr(e)[:500]
        detail["single_block"] = single_res
    natural = None
    if rank == 0 and single and not args.no_extras:
        try:
            natural = natural_pred_psnr(local_rank)
        except Exception as e:                        # noqa: BLE001 -- fixtures absent / anything else: the kernel line must survive
            natural = None
            detail["natural_pred_psnr_error"] = repr(e)[:500]
        detail["natural_pred_psnr"] = natural
    if rank == 0 and single and not args.no_extras and not args.no_live_traffic:
        lt = live_traffic(args.workload, wl.batch, precision)
        detail["live_traffic"] = lt
        if lt and "bytes_per_launch" in lt:
            main_res["roofline"]["traffic"], main_res["roofline"]["traffic_source"] = lt["bytes_per_launch"], "this run: rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE"
    if rank == 0 and single and not args.no_cpu_baseline:
        cpu = cpu_legs(args.workload, full=args.cpu_legs_full)
        detail["cpu_baseline"] = cpu
        if not args.no_extras and args.workload != "conv16":
            cpu16 = cpu_legs("conv16", full=args.cpu_legs_full)
            detail["cpu_baseline_conv16"] = cpu16
    if rank == 0:
        detail["wall_s"] = time.perf_counter() - t_start
        try:
            with open(args.detail_file, "w") as f:
                json.dump(detail, f, indent=1)
            dfile = os.path.basename(args.detail_file)
        except OSError as e:
            sys.stderr.write("bench.py: cannot write %s: %s\n" % (args.detail_file, e))
            dfile = None
        print(build_line(main_res, world, args.steps, args.warmup, wl.cfg_name, fast, per_width, cpu, cpu16, dfile,
                         {"plumbing_check": "PNN_BENCH_SHARE_GPU=1: all ranks on ONE device"} if share else None, ranks_seen, natural, single_res))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
