"""How long pnn_load_model_file takes per width (file read + host-side weight packing + upload)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hm"))
import run_hm
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork
table, mean = run_hm.make_models("/tmp/lt_models")
import torch; torch.cuda.init()
PredictionNeuralNetwork(1, 4, True, path_to_model="/tmp/lt_models/pnn_4.pnnw").close()
for w in (4, 8, 16, 32, 64):
    t0 = time.time()
    n = PredictionNeuralNetwork(1, w, w <= 8, path_to_model="/tmp/lt_models/pnn_%d.pnnw" % w)
    t1 = time.time()
    n.close()
    print("width %d: load %.3f s, destroy %.3f s" % (w, t1 - t0, time.time() - t1))
