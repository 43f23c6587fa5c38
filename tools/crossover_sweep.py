"""Diagnostic: host-call time of a conv pass with the exact-f32 kernels, the split kernels with rule-based tiles and the split kernels autotuned, for a list of batch sizes per width -- the sweep behind pass_uses_split() in pnn_abi.cpp (option split_min_px)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import context_adaptive_neural_network_based_prediction_amd as pnn
from tests import util
for w, sizes in ((8, (150, 250, 350)), (16, (50, 70, 90, 110)), (32, (15, 22, 30, 38)), (64, (5, 8, 11, 15))):
    params = util.make_params(w, False, 3)
    for n in sizes:
        above, left = util.make_contexts(w, n, 4)
        res = []
        for px, at in ((1 << 40, 0), (0, 0), (0, 1)):
            net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
            net.set_option("split_min_px", px); net.set_option("autotune", at)
            for _ in range(5): net.predict(above, left)
            t0 = time.perf_counter(); reps = 30
            for _ in range(reps): net.predict(above, left)
            res.append((time.perf_counter() - t0) / reps * 1e6)
        print("w=%2d n=%4d (pixels %6d): f32 %8.1f us   split rule-based %8.1f us   split autotuned %8.1f us" % (w, n, n * w * w, res[0], res[1], res[2]))
