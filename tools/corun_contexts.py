"""Diagnostic: a conv net (the victim) beside partner contexts in other host threads on the same GPU; every call must reproduce the first result bit for bit.  Env: VW VN VPREC VCFG VCANON PARTNER=fc|conv|none TORCH=1 (see the top of the script).  This is how the packed-fp32 erratum showed in the library."""
# victim conv net beside partners; env: VW (width) VN (batch) VPREC VCFG (sp_cfg) PARTNER=fc|conv|none TORCH=1 (device-resident I/O)
import sys, os, threading, numpy as np
sys.path.insert(0, os.getcwd())
import context_adaptive_neural_network_based_prediction_amd as pnn
from tests import util
E = os.environ.get
def worker(name, w, fc, n, seed, reps, out, opts):
    params = util.make_params(w, fc, seed, out_gain=util.out_gain(w, fc))
    above, left = util.make_contexts(w, n, seed + 1)
    net = pnn.PredictionNeuralNetwork(n, w, fc, params=params)
    for k, v in opts.items(): net.set_option(k, v)
    if E("TORCH") and not fc:
        import torch
        a = torch.from_numpy(above).cuda(); l = torch.from_numpy(left).cuda()
        run = lambda: net.predict(a, l).cpu().numpy()
    else:
        run = (lambda: net.predict(util.flatten_fc(above, left))) if fc else (lambda: net.predict(above, left))
    want = run().copy()
    bar.wait()
    bad = 0
    for r in range(reps):
        bad += not np.array_equal(run(), want)
    out[name] = (bad, reps)
out = {}
vopts = {}
if E("VPREC"): vopts["precision"] = int(E("VPREC"))
if E("VCFG"): vopts["sp_cfg"] = int(E("VCFG"))
if E("VCANON"): vopts["canonical_order"] = int(E("VCANON"))
ts = [threading.Thread(target=worker, args=("victim", int(E("VW", "16")), False, int(E("VN", "64")), 9, 300, out, vopts))]
partner = E("PARTNER", "fc")
if partner == "fc":
    ts += [threading.Thread(target=worker, args=("fc8-a", 8, True, 2048, 5, 300, out, {})), threading.Thread(target=worker, args=("fc8-b", 8, True, 1536, 7, 300, out, {}))]
elif partner == "conv":
    ts += [threading.Thread(target=worker, args=("conv-b", 16, False, 48, 5, 300, out, {}))]
bar = threading.Barrier(len(ts))
for t in ts: t.start()
for t in ts: t.join()
print({k: os.environ[k] for k in ("VW", "VN", "VPREC", "VCFG", "VCANON", "PARTNER", "TORCH") if k in os.environ}, out)
