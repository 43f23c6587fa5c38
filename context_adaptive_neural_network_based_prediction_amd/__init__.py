"""MI355X-native PNN intra-prediction forward pass (drop-in for the reference's TF1 frozen-graph path).

Importing the package needs no GPU; creating a predictor or context needs libpnn_hip.so and a HIP
device -- there is no CPU fallback.
"""
from . import weights  # noqa: F401
from .prediction_neural_network import PredictionNeuralNetwork, predict_by_batch_via_pnn  # noqa: F401

__all__ = ["PredictionNeuralNetwork", "predict_by_batch_via_pnn", "weights"]
