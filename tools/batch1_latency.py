"""Batch-1 latency of the host-buffer entry point HM binds (pnn_predict_pel: staging + net + epilogue + wait), per width and
arithmetic, and the weight stream it amounts to (a single-block call is bound by streaming the net's parameters):
    split f16 (option precision = 1; one summation order at every batch size),
    exact f32 in the canonical order (the library default since round 5: tapgemm_f32_small_kernel, the canonical fmaf chain on the 16x16x4 instruction; with
    f32_small = 0 tapgemm_f32_kernel's 128-row tiles at M = 1; fc_out_f32_kernel: the same bits as any batch),
        python tools/batch1_latency.py > profiles/rNN_batch1_latency.txt        (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib, weights as wts
from tests import util
L = _lib.lib()
for w, fc in ((4, True), (8, True), (16, False), (32, False), (64, False)):
    params = util.make_params(w, fc, 1)
    net = PredictionNeuralNetwork(1, w, fc, params=params)
    a, l = util.make_contexts(w, 1, 2)
    x = util.flatten_fc(a, l) if fc else a
    dst = np.zeros((w, w), np.int32)
    lp = None if fc else l.ctypes.data_as(_lib.f32p)
    for label, precision, small in (("exact f32 (the default; 16x16x4 chain)", 0, 1), ("exact f32, f32_small = 0 (128-row tiles)", 0, 0), ("split f16", 1, 1)):
        net.set_option("precision", precision)
        net.set_option("f32_small", small)
        for _ in range(50):
            L.pnn_predict_pel(net.ctx, w, x.ctypes.data_as(_lib.f32p), lp, 1, dst.ctypes.data_as(_lib.i32p), w)
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            n = 200
            for _ in range(n):
                L.pnn_predict_pel(net.ctx, w, x.ctypes.data_as(_lib.f32p), lp, 1, dst.ctypes.data_as(_lib.i32p), w)
            best = min(best, (time.perf_counter() - t0) / n)
        print("width %2d %-4s %-44s %7.1f us per TB call  = %6.1f GB/s of parameters (%.2f MB), %d launches" % (
            w, "FC" if fc else "conv", label, best * 1e6, params.nbytes / best / 1e9, params.nbytes / 1e6, net.last_call_stats()["launches"]))
    net.close()
