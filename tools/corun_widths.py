"""What the batching service's five width workers do to each other on ONE GPU: one process and one context per width, each issuing
back-to-back host calls (pnn_predict_pel) of a typical campaign batch -- per width the time per call alone and with the other four
running, on both arithmetics.  (tools/hm/campaign.py's per_width 'us_per_call' is the second number, plus the socket work.)
    python tools/corun_widths.py [seconds per leg]          (GPU box)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ((4, True, 6), (8, True, 3), (16, False, 2), (32, False, 1), (64, False, 1))   # (width, fully connected, blocks per call): configs[3]'s mean batches


def child(w, precision, t_start, t_end):
    import numpy as np
    from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
    from tests import util
    L = _lib.lib()
    fc, n = [(f, b) for ww, f, b in CASES if ww == w][0]
    net = PredictionNeuralNetwork(n, w, fc, params=util.make_params(w, fc, 1))
    net.set_option("precision", precision)
    a, l = util.make_contexts(w, n, 2)
    x = util.flatten_fc(a, l) if fc else a
    dst = np.zeros((n, w, w), np.int32)
    lp = None if fc else l.ctypes.data_as(_lib.f32p)
    xp, dp = x.ctypes.data_as(_lib.f32p), dst.ctypes.data_as(_lib.i32p)
    for _ in range(200):
        L.pnn_predict_pel(net.ctx, w, xp, lp, n, dp, w)
    while time.time() < t_start:
        L.pnn_predict_pel(net.ctx, w, xp, lp, n, dp, w)          # keeps the device busy until everybody is ready
    calls, t0 = 0, time.perf_counter()
    while time.time() < t_end:
        for _ in range(20):
            L.pnn_predict_pel(net.ctx, w, xp, lp, n, dp, w)
        calls += 20
    print("%d %.2f" % (w, (time.perf_counter() - t0) / calls * 1e6), flush=True)


def run(widths, precision, seconds):
    t_start = time.time() + 6.0                       # start-up of a fresh process (imports, context, model load)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(w), str(precision), repr(t_start), repr(t_start + seconds)],
                              stdout=subprocess.PIPE, text=True, cwd=ROOT) for w in widths]
    out = {}
    for p in procs:
        so, _ = p.communicate()
        w, us = so.split()[-2:]
        out[int(w)] = float(us)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), float(sys.argv[5]))
        raise SystemExit(0)
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    for arith, precision in (("f32", 0), ("split", 1)):
        alone = {}
        for w, _, _ in CASES:
            alone.update(run([w], precision, seconds))
        together = run([w for w, _, _ in CASES], precision, seconds)
        small = run([4, 8], precision, seconds)
        for w, fc, n in CASES:
            print("%-5s width %2d %-4s %d blocks per call: alone %6.1f us, with the other four widths running %6.1f us (x %.2f)%s" % (
                arith, w, "FC" if fc else "conv", n, alone[w], together[w], together[w] / alone[w],
                ("; 4 and 8 only: %6.1f us" % small[w]) if w in small else ""), flush=True)
