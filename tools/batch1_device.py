import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util
L = _lib.lib()
for w, fc in ((8, True), (16, False), (64, False)):
    net = PredictionNeuralNetwork(1, w, fc, params=util.make_params(w, fc, 1))
    a, l = util.make_contexts(w, 1, 2)
    d_out = torch.empty((1, w, w), device="cuda")
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if fc:
        d_in = torch.from_numpy(util.flatten_fc(a, l)).cuda()
        run = lambda: L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), 1, d_out.data_ptr(), sp)
    else:
        d_a, d_l = torch.from_numpy(a).cuda(), torch.from_numpy(l).cuda()
        run = lambda: L.pnn_predict_conv_device(net.ctx, w, d_a.data_ptr(), d_l.data_ptr(), 1, d_out.data_ptr(), sp)
    for _ in range(20): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 200
    for _ in range(n):
        run(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): run()
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / n
    print("width %2d: device path + sync per call %.1f us; pipelined (no sync between) %.1f us; launches %s" % (w, dt * 1e6, dt2 * 1e6, net.last_call_stats()))
