"""Every tile configuration of the exact-f32 tap-GEMM kernels (v_mfma_f32_16x16x4_f32 and v_mfma_f32_32x32x2_f32 tiles, option
`tile_cfg`) forced onto every GEMM layer of a workload: whole-pass time by device events, and the per-layer launch times of
one synchronous PNN_PROFILE pass.  The sweep behind the rule-based f32 tile choice (csrc/pnn_tiles.cpp: choose_cfg).

    python tools/f32_sweep.py [workload[:batch] ...]  > profiles/rNN_f32_tile_sweep.txt     (GPU box)
"""
import ctypes
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
name, batch, ncfg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
wl = bench.Workload(name, batch, 0, 0)
net = PredictionNeuralNetwork(wl.batch, wl.width, wl.is_fc, params=wl.params, device=0)
net.set_option("precision", 0)
def step():
    rc = wl.L.pnn_predict_tbs_device(net.ctx, wl.width, wl.d_plane.data_ptr(), 4, wl.d_tbs.data_ptr(), wl.batch, wl.d_dst.data_ptr(), None, None)
    if rc: raise RuntimeError(wl.L.pnn_last_error(net.ctx))
ref = None
for cfg in range(-1, ncfg):
    net.set_option("tile_cfg", cfg)
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
        if ref is None: ref = got
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
        print("cfg %%2d  pass %%.4f ms  maxdiff_vs_rule %%d  %%s" %% (cfg, 1e3 * sorted(ts)[2], int(np.abs(got.astype(np.int64) - ref).max()), "; ".join(ks)))
    except Exception as e:
        print("cfg %%2d  FAILED %%s" %% (cfg, str(e)[:200]))
    sys.stdout.flush()
''' % ROOT


def main():
    todo = sys.argv[1:] or ["fc8", "conv16"]
    ncfg = int(os.environ.get("F32_NCFG", "41"))
    for item in todo:
        name, _, batch = item.partition(":")
        print("==== %s (batch %s), exact-f32 kernels; cfg -1 = the rule-based choice" % (name, batch or "default"))
        sys.stdout.flush()
        r = subprocess.run([sys.executable, "-c", CHILD, name, batch or "0", str(ncfg)], env=dict(os.environ), stderr=subprocess.PIPE, text=True, cwd=ROOT)
        if r.returncode:
            print("FAILED:", r.stderr[-1500:])


if __name__ == "__main__":
    main()
