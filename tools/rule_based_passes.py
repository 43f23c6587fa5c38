"""Diagnostic: rule-based tile choice against big batches of every conv net (autotune off), one line per net -- run it under PNN_LIB_PATH=<other build> for a same-box A/B of the rules."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import context_adaptive_neural_network_based_prediction_amd as pnn
from tests import util
for w, n in ((8, 4096), (16, 1024), (16, 400), (32, 256), (64, 64)):
    params = util.make_params(w, False, 3)
    above, left = util.make_contexts(w, n, 4)
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    net.set_option("autotune", 0)
    for _ in range(5): net.predict(above, left)
    t0 = time.perf_counter(); reps = 20
    for _ in range(reps): net.predict(above, left)
    print("w=%2d n=%4d rule-based tiles: %8.1f us per pass (host call)" % (w, n, (time.perf_counter() - t0) / reps * 1e6))
