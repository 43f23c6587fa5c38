"""Single-block host calls (pnn_predict_pel at batch n) under several option sets, alternated on ONE box:
    python tools/b1_opts.py [--widths 4,8] [--n 1] [--rounds 3] [--calls 300] name=value[,name=value...] ...
An option set of "-" is the library's defaults.  Prints the best and the median of the rounds' means per (width, option set), and
checks that every set gives the Pel output of the first one.  With the diagnostic library (PNN_LIB_PATH=.../libpnn_hip_diag.so)
and PNN_B1_STAMPS=<k> the k-th small call of the process prints its device-side timeline (pnn_abi.cpp)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--widths", default="4,8")
ap.add_argument("--n", type=int, default=1)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--calls", type=int, default=300)
ap.add_argument("--conv-small", action="store_true", help="widths 4 / 8 as CONVOLUTIONAL nets (the reference's trained checkpoints are of that kind)")
ap.add_argument("sets", nargs="*", default=["-"])
args = ap.parse_args()
L = _lib.lib()
for w in [int(x) for x in args.widths.split(",")]:
    fc = w <= 8 and not args.conv_small
    n = args.n
    nets = []
    a, l = util.make_contexts(w, n, 2)
    x = util.flatten_fc(a, l) if fc else a
    lp = None if fc else l.ctypes.data_as(_lib.f32p)
    dst = np.zeros((n, w, w), np.int32)
    params = util.make_params(w, fc, 1, out_gain=util.out_gain(w, fc))
    for s in args.sets:
        net = PredictionNeuralNetwork(n, w, fc, params=params)
        if s != "-":
            for kv in s.split(","):
                k, v = kv.split("=")
                net.set_option(k, int(v))
        nets.append(net)
    call = lambda net: L.pnn_predict_pel(net.ctx, w, x.ctypes.data_as(_lib.f32p), lp, n, dst.ctypes.data_as(_lib.i32p), w)
    want = None
    means = [[] for _ in nets]
    for rnd in range(args.rounds):
        for i, net in enumerate(nets):
            for _ in range(60):
                assert call(net) == 0, L.pnn_last_error(net.ctx)
            t0 = time.perf_counter()
            for _ in range(args.calls):
                call(net)
            means[i].append((time.perf_counter() - t0) / args.calls * 1e6)
            if want is None:
                want = dst.copy()
            assert np.array_equal(dst, want), "option set %r changes the Pel output" % args.sets[i]
    for i, s in enumerate(args.sets):
        m = sorted(means[i])
        print("width %2d n=%d %-40s best %6.1f  median %6.1f us per call, %d launches" % (w, n, s, m[0], m[len(m) // 2], nets[i].last_call_stats()["launches"]), flush=True)
    for net in nets:
        net.close()
