"""Stress of the small conv passes beside each other (round 6: the merger / last layer as tails of the GEMM launches, option "tails"):
five host threads, one context each -- the batching service's layout: conv 4x4, conv 8x8, conv 16x16, conv 32x32 and an FC 8x8 net -- issue
host calls of random small batches back to back; every block of every call must equal, bit for bit, what the same block gets in a large batch
through the layer-by-layer launches.    python tools/tails_stress.py [seconds] [tails]      prints mismatching calls per thread"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import context_adaptive_neural_network_based_prediction_amd as pnn
from tests import util

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
tails = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out = {}


def worker(name, w, fc, big, seed, maxn):
    params = util.make_params(w, fc, seed, out_gain=util.out_gain(w, fc))
    above, left = util.make_contexts(w, big, seed + 1)
    net = pnn.PredictionNeuralNetwork(big, w, fc, params=params)
    net.set_option("tails", 0)
    ins = (util.flatten_fc(above, left),) if fc else (above, left)
    want = net.predict_pel(*ins).copy()
    wantf = net.predict(*ins).copy()
    net.set_option("tails", tails)
    net.set_option("f32_small_deep", 2)               # as the service sets it
    rng = np.random.RandomState(seed)
    bar.wait()
    calls = bad = 0
    t0 = time.time()
    first = None
    while time.time() - t0 < seconds:
        n = int(rng.randint(1, maxn + 1))
        o = int(rng.randint(0, big - n + 1))
        sl = tuple(a[o:o + n] for a in ins)
        if calls & 1:
            ok = np.array_equal(net.predict_pel(*sl), want[o:o + n])
        else:
            ok = np.array_equal(net.predict(*sl), wantf[o:o + n])
        calls += 1
        if not ok:
            bad += 1
            if first is None:
                first = (calls, n, o)
    out[name] = (bad, calls, first)
    net.close()


ts = [threading.Thread(target=worker, args=("conv4", 4, False, 256, 11, 24)),
      threading.Thread(target=worker, args=("conv8", 8, False, 256, 13, 12)),
      threading.Thread(target=worker, args=("conv16", 16, False, 128, 15, 6)),
      threading.Thread(target=worker, args=("conv32", 32, False, 32, 17, 3)),
      threading.Thread(target=worker, args=("fc8", 8, True, 512, 19, 16))]
bar = threading.Barrier(len(ts))
for t in ts: t.start()
for t in ts: t.join()
print("tails = %d, %.0f s: mismatching calls / calls (first bad: call number, blocks, offset): %s" % (tails, seconds, out), flush=True)
