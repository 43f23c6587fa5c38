"""Start-up cost of an HM-like process on libpnn_hip.so (no torch in the process): the first context (HIP runtime
initialisation included), four more contexts, and the five model loads -- what TComPrediction::initTempBuff pays."""
import ctypes, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t_start = time.time()
L = ctypes.CDLL(os.path.join(root, "context_adaptive_neural_network_based_prediction_amd", "libpnn_hip.so"))
L.pnn_create_empty.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_float, ctypes.c_int]
L.pnn_load_model_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
t0 = time.time()
print("dlopen: %.3f s" % (t0 - t_start))
ctxs = []
for i in range(5):
    c = ctypes.c_void_p()
    a = time.time()
    assert L.pnn_create_empty(ctypes.byref(c), ctypes.c_float(117.9), 0) == 0
    print("pnn_create_empty #%d: %.3f s" % (i, time.time() - a))
    ctxs.append(c)
d = sys.argv[1]
for c, w in zip(ctxs, (4, 8, 16, 32, 64)):
    a = time.time()
    assert L.pnn_load_model_file(c, os.path.join(d, "pnn_%d.pnnw" % w).encode()) == 0
    print("load width %d: %.3f s" % (w, time.time() - a))
print("total: %.3f s" % (time.time() - t_start))
