"""Python-side evaluation of PNN predictions on images, mirroring the PNN half of the reference's
comparing_pnn_ipfcns_hevc_best_mode.py:162-322 (`predict_mask`) without TensorFlow: contexts by the GPU gather
(context.py), predictions by libpnn_hip.so (predict_by_batch_via_pnn), then the reference's own uint8 cast and
PSNR definitions.  The HEVC-best-mode competitor (Cython, hevc/intraprediction) is out of scope.
"""
import numpy as np

from . import context
from .prediction_neural_network import predict_by_batch_via_pnn


def cast_float_to_uint8(array_float):
    """tools/tools.py:12-49: clip to [0, 255], numpy.round (half to even), cast.  (HM itself rounds half away from zero,
    TComPrediction.cpp:632 -- the two differ only on exact .5 values.)"""
    if not np.issubdtype(array_float.dtype, np.floating):
        raise TypeError('`array_float.dtype` is not smaller than `numpy.float` in type hierarchy.')
    return np.round(array_float.clip(min=0., max=255.)).astype(np.uint8)


def compute_psnr(array_0_uint8, array_1_uint8):
    """tools/tools.py:364-401: 10 log10(255^2 / (mse + 1e-6)) in float64."""
    if array_0_uint8.dtype != np.uint8:
        raise TypeError('`array_0_uint8.dtype` is not equal to `numpy.uint8`.')
    if array_1_uint8.dtype != np.uint8:
        raise TypeError('`array_1_uint8.dtype` is not equal to `numpy.uint8`.')
    mse = np.mean((array_0_uint8.astype(np.float64) - array_1_uint8.astype(np.float64)) ** 2)
    return 10. * np.log10(255. ** 2 / (mse + 1.e-6))


def predict_mask(channels_uint8, width_target, row_1sts, col_1sts, predictor, batch_size, mean_training,
                 tuple_width_height_masks=(0, 0)):
    """Predicts the target patch of every (image, position) pair and scores it.

    Returns {'predictions_pnn_uint8': [N,w,w,1], 'targets_uint8': [N,w,w,1], 'psnrs_pnn': [N], 'mean_psnr_pnn': float}
    with N = images x positions, image-major (the order of sets/common.py:213-263)."""
    batches = context.extract_context_portions_targets_from_channels_plus_preprocessing(
        channels_uint8, width_target, row_1sts, col_1sts, mean_training, tuple_width_height_masks,
        predictor.is_fully_connected, predictor=predictor)
    n = batches[0].shape[0]
    if n % batch_size:
        raise ValueError('`numerator` is not divisible by `denominator`.')
    predictions_float32 = predict_by_batch_via_pnn(batches[0:-1], None, predictor, batch_size)
    targets_off = batches[-1] + np.float32(mean_training)
    if np.any(np.modf(targets_off)[0]):
        raise RuntimeError('The target patches have been altered.')      # comparing_pnn_...py:253-254
    targets_uint8 = cast_float_to_uint8(targets_off)
    predictions_uint8 = cast_float_to_uint8(predictions_float32 + np.float32(mean_training))
    psnrs = np.array([compute_psnr(targets_uint8[i], predictions_uint8[i]) for i in range(n)])
    return {'predictions_pnn_uint8': predictions_uint8, 'targets_uint8': targets_uint8, 'psnrs_pnn': psnrs,
            'mean_psnr_pnn': float(np.mean(psnrs))}
