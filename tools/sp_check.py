"""Diagnostic: split-precision (3 x f16 MFMA) FC path vs oracle and vs the f32 path: accuracy and speed."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from oracle import pnn_oracle as O
from tests import util
L = _lib.lib()
w, n = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
params = util.make_params(w, True, 1, out_gain=util.out_gain(w, True))
a, l = util.make_contexts(w, n, 2)
ctx = util.flatten_fc(a, l)
want = O.fc_forward(params, w, ctx[:256])
net = PredictionNeuralNetwork(n, w, True, params=params)
d_in = torch.from_numpy(ctx).cuda(); d_out = torch.empty((n, w, w), device="cuda")
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for prec in (0, 1):
    net.set_option("precision", prec)
    cfgs = [-1] if prec == 0 else [-1] + [0, 2, 7, 8, 10, 11, 12, 13, 14, 15]
    for cfg in cfgs:
        if prec: net.set_option("sp_cfg", cfg)
        rc = L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), sp)
        assert rc == 0, L.pnn_last_error(net.ctx)
        torch.cuda.synchronize()
        got = d_out[:256].cpu().numpy()
        err = np.abs(got - want).max()
        pel = np.abs(O.epilogue(got, util.MEAN).astype(int) - O.epilogue(want, util.MEAN)).max()
        for _ in range(10): L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(100): L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), sp)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print("precision %d cfg %2d: max|err| vs oracle %.2e  max LSB diff %d   %.3f ms/pass  %.1f TFLOP/s-equivalent" % (prec, cfg, err, pel, ms, n * (2*(5*w*w*1200+2*1200*1200+1200*w*w)) / ms / 1e9))
