"""Host-buffer (PCIe-inclusive) rate of the C ABI's batched entry points, beside bench.py's device-resident figure.

pnn_predict_fc / pnn_predict_conv / pnn_predict_pel take HOST arrays (what the reference's Session::Run takes) and return host
arrays: every call stages its contexts to the device and its predictions back.  bench.py's `value` is measured with the inputs
already in HBM (pnn_predict_tbs_device, the gather reads the picture plane on the device); this prints what the host-array form
of the same batch sustains -- never the bench value, noted in DESIGN.md section 5.   usage: python tools/host_rate.py [--slices S] [fc8 conv16 ...]
--slices S (round 6): ONE call of S bench batches (what the reference's batched driver hands over, pnn/batching.py:7-88): the library runs it slice by
slice with the copies of the neighbouring slices beside each pass ("host_slice"); printed beside the same call with the option off (-1).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import bench
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib


def pinned_like(a):
    """A copy of `a` in pinned host memory (pnn_host_alloc): what a caller that owns its buffers would hand over."""
    L = _lib.lib()
    p = ctypes.c_void_p()
    assert L.pnn_host_alloc(ctypes.byref(p), a.nbytes) == 0
    b = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(a.size,)).reshape(a.shape)
    b[...] = a
    return b, p


def main():
    argv = sys.argv[1:]
    slices = 1
    if "--slices" in argv:
        i = argv.index("--slices")
        slices = int(argv[i + 1])
        del argv[i:i + 2]
    names = argv or ["fc8", "conv16"]
    for name in names:
        wl = bench.Workload(name, 0, 0, 0)
        w, n = wl.width, wl.batch * slices
        rng = np.random.RandomState(5)
        if wl.is_fc:
            ins = [rng.uniform(-120, 120, (n, 5 * w * w)).astype(np.float32)]
        else:
            ins = [rng.uniform(-120, 120, (n, w, 3 * w, 1)).astype(np.float32), rng.uniform(-120, 120, (n, 2 * w, w, 1)).astype(np.float32)]
        for prec, label in ((0, "f32"), (1, "split")):
            net = PredictionNeuralNetwork(n, w, wl.is_fc, params=wl.params, device=0)
            net.set_option("precision", prec)
            in_b = sum(a.nbytes for a in ins) / n
            L = _lib.lib()
            dst = np.zeros((n, w, w), np.int32)
            pins = [pinned_like(a) for a in ins]
            pdst, pdst_h = pinned_like(dst.view(np.float32))
            for how in ("pageable arrays", "pinned arrays (pnn_host_alloc)"):
                arrs = [p[0] for p in pins] if how.startswith("pinned") else ins
                d = pdst if how.startswith("pinned") else dst
                a0 = arrs[0].ctypes.data_as(_lib.f32p)
                a1 = None if wl.is_fc else arrs[1].ctypes.data_as(_lib.f32p)
                dp = ctypes.cast(d.ctypes.data, _lib.i32p)
                call = lambda: L.pnn_predict_pel(net.ctx, w, a0, a1, n, dp, w)
                want = None
                for mode in ((0,) if slices == 1 else (0, -1)):
                    net.set_option("host_slice", mode)
                    for _ in range(5):
                        assert call() == 0
                    if want is None:
                        want = np.array(d, copy=True)
                    assert np.array_equal(np.asarray(d), want), "host_slice changes the result"
                    reps, t0 = 0, time.perf_counter()
                    while time.perf_counter() - t0 < 1.5:
                        call()
                        reps += 1
                    dt = (time.perf_counter() - t0) / reps
                    tag = "" if slices == 1 else (" x %d slices overlapped" % slices if mode == 0 else " x %d, one copy in / out" % slices)
                    print("%-7s %-5s batch %5d%s %-32s %.4f ms per call = %10.0f blocks/s  (%d B in + %d B out per block = %.1f GB/s over the link)"
                          % (name, label, wl.batch, tag, how + ":", dt * 1e3, n / dt, in_b, 4 * w * w, (in_b + 4 * w * w) * n / dt / 1e9), flush=True)
            net.close()
            for _, p in pins + [(None, pdst_h)]:
                L.pnn_host_free(p)


if __name__ == "__main__":
    main()
