import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from context_adaptive_neural_network_based_prediction_amd import PredictionNeuralNetwork, _lib
from tests import util
L = _lib.lib()
w, n = 8, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
params = util.make_params(w, True, 1)
rng = np.random.RandomState(0)
ctx = (rng.randint(0, 256, (n, 320)).astype(np.float32) - 117.9)
net = PredictionNeuralNetwork(n, w, True, params=params)
net.set_option("precision", 1); net.set_option("sp_cfg", int(sys.argv[1]))
d_in = torch.from_numpy(ctx).cuda(); d_out = torch.empty((n, w, w), device="cuda")
for _ in range(3): L.pnn_predict_fc_device(net.ctx, w, d_in.data_ptr(), n, d_out.data_ptr(), None)
torch.cuda.synchronize()
