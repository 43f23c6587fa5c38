"""Diagnostic: is the FC-8 GEMM chain clock/power-limited? Times the same launches on random and on all-zero data."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util
L = _lib.lib()
w, n = 8, 4096
for name in ("random", "zeros", "random"):
    params = util.make_params(w, True, 1)
    a, l = util.make_contexts(w, n, 2)
    ctx = util.flatten_fc(a, l)
    if name == "zeros":
        params = np.zeros_like(params); ctx = np.zeros_like(ctx)
    net = PredictionNeuralNetwork(n, w, True, params=params)
    d_in = torch.from_numpy(ctx).cuda(); d_out = torch.empty((n, w, w), device="cuda")
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(20):
        L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), sp)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    print("%-7s %.3f ms per pass  %.1f TFLOP/s" % (name, ms, n * 6.6816e6 / ms / 1e9))
