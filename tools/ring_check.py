"""Diagnostic: every split-GEMM configuration code (tapgemm_sp / convimg / ring kernels) must give BIT-IDENTICAL
predictions; prints the time of a whole pass with each code forced on every split GEMM of the net.

usage: python tools/ring_check.py <width> <batch> <first_code> <last_code> [repeats]"""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util
L = _lib.lib()
w, n = int(sys.argv[1]), int(sys.argv[2])
c0, c1 = int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
fc = w <= 8
params = util.make_params(w, fc, 1, out_gain=util.out_gain(w, fc))
a, l = util.make_contexts(w, n, 2)
net = PredictionNeuralNetwork(n, w, fc, params=params)
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
d_out = torch.empty((n, w, w), device="cuda")
if fc:
    d_in = torch.from_numpy(util.flatten_fc(a, l)).cuda()
    run = lambda: L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), sp)
else:
    d_a, d_l = torch.from_numpy(a).cuda(), torch.from_numpy(l).cuda()
    run = lambda: L.pnn_predict_conv_device(net.ctx, w, d_a.data_ptr(), d_l.data_ptr(), n, d_out.data_ptr(), sp)
net.set_option("canonical_order", 1)
net.set_option("ring", 0); net.set_option("convimg", 0)
assert run() == 0, L.pnn_last_error(net.ctx)
torch.cuda.synchronize()
want = d_out.cpu().numpy().copy()
net.set_option("ring", 1); net.set_option("convimg", 1)
bad = 0
for code in range(c0, c1 + 1):
    net.set_option("sp_cfg", code)
    worst = 0
    for rep in range(reps):                      # races show up as rare wrong tiles: repeat
        d_out.zero_()
        assert run() == 0, L.pnn_last_error(net.ctx)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        worst = max(worst, int((got != want).sum()))
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    print("code %2d: %s  %.3f ms/pass" % (code, "bit-identical" if worst == 0 else "MISMATCH in %d values" % worst, e0.elapsed_time(e1) / 30), flush=True)
    bad += worst != 0
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
