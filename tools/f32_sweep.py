"""Every tile of the exact-f32 tap-GEMM kernel (tapgemm_f32_kernel, option `f32_cfg`) forced onto every layer it is legal for, beside
the rule-based choice and the autotuned choice: whole-pass time by wall clock over
synchronised regions, GEMM time and rate from the per-launch events (option `time_launches`).  All tiles of the new kernel must
give the same bits (checked here on the int32 predictions and by tests/test_gpu_parity.py on the floats).

    python tools/f32_sweep.py [workload[:batch] ...]  > profiles/rNN_f32_tile_sweep.txt     (GPU box)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork
name, batch = sys.argv[1], int(sys.argv[2])
wl = bench.Workload(name, batch, 0, 0)
net = PredictionNeuralNetwork(wl.batch, wl.width, wl.is_fc, params=wl.params, device=0)
net.set_option("precision", 0)
def step():
    rc = wl.L.pnn_predict_tbs_device(net.ctx, wl.width, wl.d_plane.data_ptr(), 4, wl.d_tbs.data_ptr(), wl.batch, wl.d_dst.data_ptr(), None, None)
    if rc: raise RuntimeError(wl.L.pnn_last_error(net.ctx))
ref = None
cases = [("rule", {"autotune": 0, "f32_cfg": -1}), ("autotuned", {"autotune": 1})]
cases += [("f32_cfg %%d" %% i, {"autotune": 0, "f32_cfg": i}) for i in range(wl.L.pnn_num_f32_configs())]
for label, opts in cases:
    for k, v in opts.items():
        net.set_option(k, v)
    try:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20)
        got = wl.d_dst.cpu().numpy()
        if label == "rule": ref = got
        net.set_option("time_launches", 1)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        net.set_option("time_launches", 0)
        ks = []
        for kind in (0, 1):
            n_k, us_k, fl_k = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
            wl.L.pnn_launch_times(net.ctx, kind, ctypes.byref(n_k), ctypes.byref(us_k), ctypes.byref(fl_k))
            if n_k.value: ks.append("kind%%d: %%d launches/pass, %%.1f us/pass, %%.1f TFLOP/s" %% (kind, n_k.value // 5, us_k.value / 5, fl_k.value / us_k.value / 1e6))
        diff = "" if ref is None else "  pixels differing from rule: %%d" %% int((got != ref).sum())
        print("%%-16s pass %%.4f ms  %%s%%s" %% (label, 1e3 * sorted(ts)[2], "; ".join(ks), diff))
    except Exception as e:
        print("%%-16s FAILED %%s" %% (label, str(e)[:200]))
    sys.stdout.flush()
''' % ROOT


def main():
    todo = sys.argv[1:] or ["fc8", "conv16"]
    for item in todo:
        name, _, batch = item.partition(":")
        print("==== %s (batch %s), exact-f32 arithmetic" % (name, batch or "default"))
        sys.stdout.flush()
        r = subprocess.run([sys.executable, "-c", CHILD, name, batch or "0"], env=dict(os.environ), stderr=subprocess.PIPE, text=True, cwd=ROOT)
        if r.returncode:
            print("FAILED:", r.stderr[-1500:])


if __name__ == "__main__":
    main()
