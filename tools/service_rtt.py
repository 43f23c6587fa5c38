"""Diagnostic: what one request costs through the batching service (Unix socket, server I/O thread, per-width worker, GPU
call, the way back) against the same single-block call made directly -- the encoders of a campaign are chains of such
requests, so this round trip, not throughput, sets their wall time.

usage: python tools/service_rtt.py [width] [requests] [clients]"""
import ctypes, os, sys, tempfile, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib, service
from tests import util

w = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
nclients = int(sys.argv[3]) if len(sys.argv) > 3 else 1
fc = w <= 8
L = _lib.lib()
params = util.make_params(w, fc, 1, out_gain=util.out_gain(w, fc))
above, left = util.make_contexts(w, n, 2)
rows = util.flatten_fc(above, left) if fc else None
net = PredictionNeuralNetwork(1, w, fc, params=params)
net.set_option("cache_mb", 0)
dst = np.empty((w, w), np.int32)


def direct(i):
    if fc:
        return L.pnn_predict_pel(net.ctx, w, rows[i].ctypes.data_as(_lib.f32p), None, 1, dst.ctypes.data_as(_lib.i32p), w)
    return L.pnn_predict_pel(net.ctx, w, above[i].ctypes.data_as(_lib.f32p), left[i].ctypes.data_as(_lib.f32p), 1, dst.ctypes.data_as(_lib.i32p), w)


for i in range(50):
    assert direct(i) == 0, L.pnn_last_error(net.ctx)
t0 = time.perf_counter()
for i in range(n):
    direct(i)
t_direct = (time.perf_counter() - t0) / n
print("direct single-block call, %dx%d: %.1f us" % (w, w, t_direct * 1e6))

sock = os.path.join(tempfile.mkdtemp(), "pnn.sock")
srv = service.serve_in_thread(sock, ctx=net.ctx, max_batch=256, window_us=0)
per = [0.0] * nclients


def client(k):
    c = ctypes.c_void_p()
    for _ in range(500):
        if L.pnn_client_connect(ctypes.byref(c), sock.encode()) == 0:
            break
        time.sleep(0.01)
    out = np.empty((w, w), np.int32)
    idx = list(range(k, n, nclients))
    call = (lambda i: L.pnn_client_predict_pel(c, w, rows[i].ctypes.data_as(_lib.f32p), None, out.ctypes.data_as(_lib.i32p), w)) if fc else \
           (lambda i: L.pnn_client_predict_pel(c, w, above[i].ctypes.data_as(_lib.f32p), left[i].ctypes.data_as(_lib.f32p), out.ctypes.data_as(_lib.i32p), w))
    for i in idx[:20]:
        assert call(i) == 0
    t = time.perf_counter()
    for i in idx[20:]:
        call(i)
    per[k] = (time.perf_counter() - t) / max(1, len(idx) - 20)
    L.pnn_client_close(c)


ts = [threading.Thread(target=client, args=(k,)) for k in range(nclients)]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
wall = time.perf_counter() - t0
stats = srv.stop()
print("through the service, %d client(s): %.1f us per request round trip (mean over clients), %d requests in %d GPU calls, largest batch %d, %.0f requests/s"
      % (nclients, 1e6 * sum(per) / nclients, stats["requests"], stats["backend_calls"], stats["largest_batch"], stats["requests"] / wall))
