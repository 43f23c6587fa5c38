"""Batch-1 latency of the host-buffer entry point HM binds (pnn_predict_pel: staging + net + epilogue + wait), per width,
with the round-3 switches on and off: `flag_wait` (the call's last kernel raises a flag in pinned host memory behind its
results and the host spins on that word; default on), `fc_out` (FC output layer: K segments + reduction in one launch) and
`spin_wait` (poll the stream instead of blocking in hipStreamSynchronize)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util
L = _lib.lib()
for w, fc in ((4, True), (8, True), (16, False), (32, False), (64, False)):
    net = PredictionNeuralNetwork(1, w, fc, params=util.make_params(w, fc, 1))
    a, l = util.make_contexts(w, 1, 2)
    x = util.flatten_fc(a, l) if fc else a
    dst = np.zeros((w, w), np.int32)
    lp = None if fc else l.ctypes.data_as(_lib.f32p)
    for canonical, fc_out, spin, flag in ((1, 0, 0, 0), (1, 0, 0, 1), (1, 1, 0, 1), (1, 0, 1, 0), (1, 1, 0, 0), (1, 0, 0, 1), (1, 1, 0, 1), (1, 0, 0, 0)):
        if not fc and fc_out == 1:
            continue
        net.set_option("canonical_order", canonical)
        net.set_option("fc_out", fc_out)
        net.set_option("spin_wait", spin)
        net.set_option("flag_wait", flag)
        for _ in range(50):
            L.pnn_predict_pel(net.ctx, w, x.ctypes.data_as(_lib.f32p), lp, 1, dst.ctypes.data_as(_lib.i32p), w)
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                L.pnn_predict_pel(net.ctx, w, x.ctypes.data_as(_lib.f32p), lp, 1, dst.ctypes.data_as(_lib.i32p), w)
            best = min(best, (time.perf_counter() - t0) / n)
        print("width %2d %-4s canonical_order=%d fc_out=%d spin_wait=%d flag_wait=%d: %.1f us per TB call" % (w, "FC" if fc else "conv", canonical, fc_out, spin, flag, best * 1e6))
