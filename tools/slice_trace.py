"""Where does a sliced host call spend its time?  (round 6: 16 slices of a bench batch take 10-17 % longer than 16 passes)
   on the box:  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/sl -- python3 tools/slice_trace.py run conv16
                python3 tools/slice_trace.py show /tmp/sl 13        (13 = kernels per pass)
`run` makes 4 calls of 16 slices; `show` prints, for the last call, every slice's pass (first kernel start -> last kernel end), the idle gap in
front of it, its kernels' busy time, and the copies that ran beside it."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(name):
    import ctypes
    import numpy as np
    import bench
    from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
    wl = bench.Workload(name, 0, 0, 0)
    w, n = wl.width, wl.batch * 16
    rng = np.random.RandomState(5)
    if wl.is_fc:
        ins = [rng.uniform(-120, 120, (n, 5 * w * w)).astype(np.float32)]
    else:
        ins = [rng.uniform(-120, 120, (n, w, 3 * w, 1)).astype(np.float32), rng.uniform(-120, 120, (n, 2 * w, w, 1)).astype(np.float32)]
    net = PredictionNeuralNetwork(n, w, wl.is_fc, params=wl.params, device=0)
    L = _lib.lib()
    dst = np.zeros((n, w, w), np.int32)
    a0 = ins[0].ctypes.data_as(_lib.f32p)
    a1 = None if wl.is_fc else ins[1].ctypes.data_as(_lib.f32p)
    import time
    for _ in range(int(os.environ.get("SLICE_TRACE_CALLS", "4"))):
        t0 = time.perf_counter()
        assert L.pnn_predict_pel(net.ctx, w, a0, a1, n, ctypes.cast(dst.ctypes.data, _lib.i32p), w) == 0
        print("call: %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    net.close()


def show(d, per_pass):
    ks, cs = [], []
    for f in glob.glob(d + "/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("pnn::", "").replace("void ", "")[:36]))
    for f in glob.glob(d + "/*/*_memory_copy_trace.csv"):
        for r in csv.DictReader(open(f)):
            cs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?"))))
    ks.sort(); cs.sort()
    ks = [k for k in ks if not k[2].startswith("__amd")]
    last = ks[-16 * per_pass:]
    t0 = last[0][0]
    prev_end = None
    for s in range(16):
        p = last[s * per_pass:(s + 1) * per_pass]
        a, b = p[0][0], p[-1][1]
        busy = sum(k[1] - k[0] for k in p)
        gaps = sum(max(0, p[i][0] - p[i - 1][1]) for i in range(1, len(p)))
        beside = [c for c in cs if c[1] > a and c[0] < b]
        print("slice %2d: starts %8.1f us, gap in front %6.1f us, pass %7.1f us (kernels busy %7.1f, gaps inside %5.1f), copies beside it: %s" % (
            s, (a - t0) / 1e3, 0.0 if prev_end is None else (a - prev_end) / 1e3, (b - a) / 1e3, busy / 1e3, gaps / 1e3,
            ", ".join("%s %s B %.0f us" % (c[2], c[3], (c[1] - c[0]) / 1e3) for c in beside)))
        prev_end = b
    print("first kernel -> last kernel of the call: %.1f us" % ((last[-1][1] - t0) / 1e3))
    first = last[:per_pass]
    for k in first:
        print("   %-36s %7.1f us" % (k[2], (k[1] - k[0]) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        show(sys.argv[2], int(sys.argv[3]))
