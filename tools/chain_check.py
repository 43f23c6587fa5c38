"""Diagnostic: the chained FC kernel (option "chain") against the per-layer launches: repeatability, difference, time."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util
L = _lib.lib()
for w, n in [(8, int(x)) for x in sys.argv[1:]] or ((8, 4096), (4, 4096), (8, 2048), (8, 3000)):
    params = util.make_params(w, True, 1, out_gain=util.out_gain(w, True))
    a, l = util.make_contexts(w, n, 2)
    net = PredictionNeuralNetwork(n, w, True, params=params)
    sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    d_out = torch.empty((n, w, w), device="cuda")
    d_in = torch.from_numpy(util.flatten_fc(a, l)).cuda()
    run = lambda: L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), sp)
    net.set_option("autotune", 0)
    net.set_option("chain", 0)
    assert run() == 0, L.pnn_last_error(net.ctx)
    torch.cuda.synchronize()
    want = d_out.cpu().numpy().copy()
    net.set_option("chain", 1)
    bad = 0
    worst = 0.0
    first = None
    for rep in range(20):
        d_out.zero_()
        assert run() == 0, L.pnn_last_error(net.ctx)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        if first is None:
            first = got.copy()
        bad += int((got != first).sum())                 # run-to-run: must be bit-identical (a race would show here)
        worst = max(worst, float(np.abs(got - want).max()))   # vs per-layer launches: the fused output layer may group its sum differently
    st = net.last_call_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(200): run()
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    t1 = e0.elapsed_time(e1) / 200
    net.set_option("chain", 0)
    for _ in range(200): run()
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    t0 = e0.elapsed_time(e1) / 200
    print("w=%d n=%d: chained %s (launches %d)  %.4f ms chained vs %.4f ms per-layer" % (w, n, ("repeatable, max |delta| vs per-layer %.1e" % worst) if bad == 0 else "NOT REPEATABLE %d" % bad, st["launches"], t1, t0), flush=True)
