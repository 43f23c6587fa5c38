"""Single-block calls (what HM issues) for rocprofv3 --kernel-trace: per-kernel time of one width at batch 1.
usage: batch1_trace.py <width> <unused> [calls]      (arithmetic: PNN_PRECISION, default exact f32)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util
w, canon = int(sys.argv[1]), int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 60
fc = w <= 8
L = _lib.lib()
net = PredictionNeuralNetwork(1, w, fc, params=util.make_params(w, fc, 1))
a, l = util.make_contexts(w, 1, 2)
x = util.flatten_fc(a, l) if fc else a
dst = np.zeros((w, w), np.int32)
lp = None if fc else l.ctypes.data_as(_lib.f32p)
for _ in range(calls):
    assert L.pnn_predict_pel(net.ctx, w, x.ctypes.data_as(_lib.f32p), lp, 1, dst.ctypes.data_as(_lib.i32p), w) == 0
