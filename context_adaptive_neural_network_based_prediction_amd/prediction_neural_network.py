"""Host-side mirror of the reference's Python interface to the PNN forward pass.

Mirrors (same names, argument meaning and error behaviour):
  - pnn/PredictionNeuralNetwork.py:77-137  class PredictionNeuralNetwork(batch_size, width_target, is_fully_connected, ...)
  - pnn/batching.py:7-88                   predict_by_batch_via_pnn(tuple_batches_float32, sess, predictor, batch_size)
The TensorFlow session is replaced by libpnn_hip.so; `sess` is accepted for signature compatibility and
ignored. Inputs may be numpy arrays (staged through the host entry points) or torch CUDA tensors (device
entry points, asynchronous on torch's current stream).
"""
import ctypes

import numpy as np

from . import _lib, weights as wts


class PredictionNeuralNetwork(object):
    """One PNN model (one width) resident on one MI355X.

    Parameters mirror pnn/PredictionNeuralNetwork.py:26-76. `tuple_coeffs` / `dict_reading` configure
    training in the reference and must stay None here (inference only). Weights come from `path_to_model`
    (.pnnw) or `params` (flat float32 in the canonical order of weights.tensor_specs).
    """

    def __init__(self, batch_size, width_target, is_fully_connected, tuple_coeffs=None, dict_reading=None,
                 path_to_model=None, params=None, mean_training=wts.MEAN_TRAINING_LUMINANCE, device=0):
        if tuple_coeffs is not None or dict_reading is not None:
            raise NotImplementedError("training graphs (tuple_coeffs / dict_reading) are out of scope of the MI355X path")
        if is_fully_connected and width_target not in (4, 8, 16):
            raise ValueError("`width_target` does not belong to {4, 8, 16} for a fully-connected PNN.")
        if not is_fully_connected and width_target not in wts.STRIDES_BRANCH:
            raise ValueError("`width_target` does not belong to {4, 8, 16, 32, 64}.")
        self.batch_size = batch_size
        self.width_target = width_target
        self.is_fully_connected = bool(is_fully_connected)
        self.strides_branch = None if is_fully_connected else wts.STRIDES_BRANCH[width_target]
        self.mean_training = float(mean_training)
        self.device = device
        self._L = _lib.lib()
        self._ctx = ctypes.c_void_p()
        _lib.check(self._L.pnn_create_empty(ctypes.byref(self._ctx), ctypes.c_float(mean_training), device))
        if path_to_model is not None:
            _lib.check(self._L.pnn_load_model_file(self._ctx, path_to_model.encode()), self._ctx)
            info_fc = ctypes.c_int()
            rc = self._L.pnn_model_info(self._ctx, width_target, ctypes.byref(info_fc), None, None)
            if rc != 0 or bool(info_fc.value) != self.is_fully_connected:
                raise ValueError("%s does not hold a %s model of width %d" % (
                    path_to_model, "fully-connected" if is_fully_connected else "convolutional", width_target))
        elif params is not None:
            self.load_params(params)

    # -- reference API: PredictionNeuralNetwork.initialization(sess, path_to_restore) (:184-200) --------
    def initialization(self, sess=None, path_to_restore=""):
        """Restores the parameters from a `.pnnw` file or a TF V2 checkpoint prefix."""
        if not path_to_restore:
            raise ValueError("random initialisation is a training feature; give `path_to_restore`.")
        if path_to_restore.endswith(".pnnw"):
            _lib.check(self._L.pnn_load_model_file(self._ctx, path_to_restore.encode()), self._ctx)
        else:
            self.load_params(wts.params_from_tf_bundle(path_to_restore, self.width_target, self.is_fully_connected))

    def load_params(self, flat):
        flat = np.ascontiguousarray(flat, dtype=np.float32)
        _lib.check(self._L.pnn_load_model_params(self._ctx, self.width_target, int(self.is_fully_connected),
                                                 flat.ctypes.data_as(_lib.f32p), flat.size), self._ctx)

    def set_option(self, name, value):
        _lib.check(self._L.pnn_set_option(self._ctx, name.encode(), int(value)), self._ctx)

    @property
    def ctx(self):
        return self._ctx

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._L.pnn_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- prediction ------------------------------------------------------------------------------------
    def _is_torch(self, x):
        return type(x).__module__.startswith("torch")

    def predict(self, *inputs):
        """FC: predict(flattened_contexts [N,5w^2]); conv: predict(portions_above [N,w,3w,1], portions_left [N,2w,w,1]).
        Returns float32 [N,w,w,1] (mean NOT re-added), numpy in -> numpy out, CUDA tensor in -> CUDA tensor out."""
        w = self.width_target
        n_expected = 1 if self.is_fully_connected else 2
        if len(inputs) != n_expected:
            raise ValueError("a %s PNN takes %d input array(s)" % ("fully-connected" if n_expected == 1 else "convolutional", n_expected))
        if self._is_torch(inputs[0]):
            return self._predict_torch(*inputs)
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in inputs]
        n = arrs[0].shape[0]
        self._check_shapes([a.shape for a in arrs])
        out = np.empty((n, w, w, 1), np.float32)
        if self.is_fully_connected:
            rc = self._L.pnn_predict_fc(self._ctx, w, arrs[0].ctypes.data_as(_lib.f32p), n, out.ctypes.data_as(_lib.f32p))
        else:
            rc = self._L.pnn_predict_conv(self._ctx, w, arrs[0].ctypes.data_as(_lib.f32p),
                                          arrs[1].ctypes.data_as(_lib.f32p), n, out.ctypes.data_as(_lib.f32p))
        _lib.check(rc, self._ctx)
        return out

    def _check_shapes(self, shapes):
        w = self.width_target
        n = shapes[0][0]
        if self.is_fully_connected:
            if int(np.prod(shapes[0][1:])) != 5 * w * w:
                raise ValueError("flattened contexts must be [N, %d]" % (5 * w * w))
        else:
            if int(np.prod(shapes[0][1:])) != 3 * w * w or int(np.prod(shapes[1][1:])) != 2 * w * w or shapes[1][0] != n:
                raise ValueError("portions must be [N,%d,%d,1] and [N,%d,%d,1]" % (w, 3 * w, 2 * w, w))

    def _predict_torch(self, *inputs):
        import torch
        w = self.width_target
        ts = [t.contiguous().float() for t in inputs]
        if not all(t.is_cuda for t in ts):
            raise ValueError("torch inputs must be CUDA tensors (use numpy arrays for host data)")
        self._check_shapes([tuple(t.shape) for t in ts])
        n = ts[0].shape[0]
        out = torch.empty((n, w, w, 1), dtype=torch.float32, device=ts[0].device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(ts[0].device).cuda_stream)
        if self.is_fully_connected:
            rc = self._L.pnn_predict_fc_device(self._ctx, w, ts[0].data_ptr(), n, out.data_ptr(), stream)
        else:
            rc = self._L.pnn_predict_conv_device(self._ctx, w, ts[0].data_ptr(), ts[1].data_ptr(), n, out.data_ptr(), stream)
        _lib.check(rc, self._ctx)
        return out

    def predict_pel(self, *inputs):
        """predict() followed by the HM epilogue (TComPrediction.cpp:621-635): int32 [N,w,w] in 0..255."""
        w = self.width_target
        arrs = [np.ascontiguousarray(a, dtype=np.float32) for a in inputs]
        self._check_shapes([a.shape for a in arrs])
        n = arrs[0].shape[0]
        dst = np.empty((n, w, w), np.int32)
        left = arrs[1].ctypes.data_as(_lib.f32p) if len(arrs) > 1 else None
        _lib.check(self._L.pnn_predict_pel(self._ctx, w, arrs[0].ctypes.data_as(_lib.f32p), left, n,
                                           dst.ctypes.data_as(_lib.i32p), w), self._ctx)
        return dst

    def arithmetic_tag(self):
        """What decides the last float bits of this network's predictions (pnn_arithmetic_tag): equal tags <=> identical predictions --
        what an encoder and its decoder compare once at start-up."""
        buf = ctypes.create_string_buffer(160)
        if self._L.pnn_arithmetic_tag(self._ctx, buf, len(buf)) != 0:
            raise _lib.PnnError("pnn_arithmetic_tag failed")
        return buf.value.decode()

    def last_call_stats(self):
        ng, fl, nl = ctypes.c_int(), ctypes.c_double(), ctypes.c_int()
        self._L.pnn_last_call_stats(self._ctx, ctypes.byref(ng), ctypes.byref(fl), ctypes.byref(nl))
        issued = ctypes.c_double()
        self._L.pnn_last_call_issued_flops(self._ctx, ctypes.byref(issued))
        return {"gemm_launches": ng.value, "gemm_flops": fl.value, "gemm_flops_issued": issued.value, "launches": nl.value}

    def cache_stats(self):
        """(hits, misses) of the single-block prediction cache (option "cache_mb")."""
        h, m = ctypes.c_long(), ctypes.c_long()
        self._L.pnn_cache_stats(self._ctx, ctypes.byref(h), ctypes.byref(m))
        return h.value, m.value


def divide_ints_check_divisible(numerator, denominator):
    """tools/tools.py:403-434."""
    if not isinstance(numerator, int):
        raise TypeError('`numerator` is not an instance of `int`.')
    if not isinstance(denominator, int):
        raise TypeError('`denominator` is not an instance of `int`.')
    if numerator % denominator != 0:
        raise ValueError('`numerator` is not divisible by `denominator`.')
    return numerator // denominator


def predict_by_batch_via_pnn(tuple_batches_float32, sess, predictor, batch_size):
    """pnn/batching.py:7-88: N inputs, `batch_size` per run, float32 [N,w,w,1] (mean not re-added).

    `sess` is ignored (no TensorFlow). Same checks: N must be divisible by batch_size (ValueError); for a
    fully-connected PNN the width is recovered as sqrt(cols / 5) and must be whole (ValueError).
    """
    nb_predictions = int(tuple_batches_float32[0].shape[0])
    nb_batches = divide_ints_check_divisible(nb_predictions, batch_size)
    if predictor.is_fully_connected:
        width_float = float(np.sqrt(float(tuple_batches_float32[0].shape[1]) / 5.))
        if not width_float.is_integer():
            raise ValueError('`numpy.sqrt(float(tuple_batches_float32[0].shape[1])/5.)` is not a whole number.')
        width_target = int(width_float)
    else:
        width_target = tuple_batches_float32[0].shape[1]
    if width_target != predictor.width_target:
        raise ValueError("inputs are for width %d, the predictor is for width %d" % (width_target, predictor.width_target))
    # The reference runs its nb_batches batches strictly one after the other (batching.py:64-86).  Here they are ONE call of the C ABI
    # whose slices are those batches ("host_slice" = batch_size): batch i + 1 is copied in and batch i - 1 copied out while batch i
    # computes (host_predict_sliced, csrc/pnn_abi.cpp) -- the same predictions bit for bit, a block's result does not depend on its batch.
    sliced = nb_batches >= 2 and hasattr(predictor, "set_option") and hasattr(predictor, "_is_torch")   # (any object with predict() may stand in for the predictor)
    if sliced and not predictor._is_torch(tuple_batches_float32[0]):
        predictor.set_option("host_slice", batch_size)
        try:
            return predictor.predict(*tuple_batches_float32[:1 if predictor.is_fully_connected else 2])
        finally:
            predictor.set_option("host_slice", 0)
    predictions_float32 = np.zeros((nb_predictions, width_target, width_target, 1), dtype=np.float32)
    for i in range(nb_batches):
        sl = slice(i * batch_size, (i + 1) * batch_size)
        if predictor.is_fully_connected:
            predictions_float32[sl] = predictor.predict(tuple_batches_float32[0][sl])
        else:
            predictions_float32[sl] = predictor.predict(tuple_batches_float32[0][sl], tuple_batches_float32[1][sl])
    return predictions_float32
