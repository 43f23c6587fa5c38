"""The sweep behind the rule-based tile choice (csrc/pnn_tiles.cpp): one pass of a workload with the autotuner's log on --
every legal configuration of the three split-precision GEMM families, timed per layer on the device (PNN_DEBUG_TUNE), the
fastest and the rule's own choice (PNN_DEBUG).  All configurations give bit-identical results, so only speed is at stake.

    python tools/tile_sweep.py [workload[:batch] ...]  > profiles/rNN_tile_sweep.txt     (GPU box)

Configuration codes: [0, nsp) tapgemm_sp tiles, then convimg_sp tiles, then tapgemm_ring tiles (pnn_num_split_configs).
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, %r)
import numpy as np, torch
import bench
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork
name, batch = sys.argv[1], int(sys.argv[2])
wl = bench.Workload(name, batch, 0, 0)
net = PredictionNeuralNetwork(wl.batch, wl.width, wl.is_fc, params=wl.params, device=0)
net.set_option("autotune", 1)
rc = wl.L.pnn_predict_tbs_device(net.ctx, wl.width, wl.d_plane.data_ptr(), 4, wl.d_tbs.data_ptr(), wl.batch, wl.d_dst.data_ptr(), None, None)
torch.cuda.synchronize()
assert rc == 0
''' % ROOT


def main():
    todo = sys.argv[1:] or ["fc8", "conv16", "fc4", "conv4", "conv8", "conv32", "conv64", "conv16:64", "conv16:256", "fc8:1024"]
    for item in todo:
        name, _, batch = item.partition(":")
        env = dict(os.environ, PNN_DEBUG="1", PNN_DEBUG_TUNE="1")
        r = subprocess.run([sys.executable, "-c", CHILD, name, batch or "0"], env=env, capture_output=True, text=True, cwd=ROOT)
        print("==== %s (batch %s)" % (name, batch or "default"))
        for line in r.stderr.splitlines():
            if line.startswith("[pnn]"):
                print(line)
        if r.returncode:
            print("FAILED:", r.stderr[-1500:])


if __name__ == "__main__":
    main()
