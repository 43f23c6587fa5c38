"""Do two builds of the library predict the same BITS?  Runs the float predictions of every net (seeded weights and contexts, a single
block, a handful, a batch) in one subprocess per library and compares the arrays: python tools/lib_ab_bits.py <libA.so> <libB.so>
(used when a kernel is replaced by another form of the same arithmetic: the order revision of pnn_arithmetic_tag stays only if this says equal)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork
from tests import util
out = {}
for w, fc, big in ((4, True, 300), (8, True, 300), (4, False, 200), (8, False, 200), (16, False, 1100), (32, False, 70), (64, False, 9)):
    params = util.make_params(w, fc, 7, out_gain=util.out_gain(w, fc))
    a, l = util.make_contexts(w, big, 8)
    net = PredictionNeuralNetwork(big, w, fc, params=params)
    run = (lambda x, y: net.predict(util.flatten_fc(x, y))) if fc else (lambda x, y: net.predict(x, y))
    for n in (1, 5, big):
        out["%%s%%d_%%d" %% ("fc" if fc else "conv", w, n)] = run(a[:n], l[:n])
    net.close()
np.savez(sys.argv[1], **out)
''' % ROOT
res = []
with tempfile.TemporaryDirectory() as d:
    for i, lib in enumerate(sys.argv[1:3]):
        f = os.path.join(d, "o%d.npz" % i)
        subprocess.check_call([sys.executable, "-c", CHILD, f], env=dict(os.environ, PNN_LIB_PATH=os.path.abspath(lib)))
        res.append(dict(np.load(f)))
bad = 0
for k in sorted(res[0]):
    same = np.array_equal(res[0][k].view(np.uint32), res[1][k].view(np.uint32))
    d = np.abs(res[0][k] - res[1][k]).max()
    print("%-16s %s (max |delta| %.3g)" % (k, "same bits" if same else "DIFFERENT", d))
    bad += not same
print("all equal" if not bad else "%d of %d differ" % (bad, len(res[0])))
sys.exit(1 if bad else 0)
