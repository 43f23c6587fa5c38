"""Parity of the HIP path (through the C ABI) against the CPU oracle -- the GPU tests proper.

Tolerances: the nets compute in float32 on both sides but in different summation orders (MFMA k-chunks
vs the oracle's sequential loops), so raw float predictions are compared with an absolute tolerance of
2e-3 on values of magnitude up to ~300, and HM-epilogue outputs (uint8 range) within +-1 LSB -- the
tolerance BASELINE.json states -- with at most 0.1 % of the pixels allowed to differ at all (exact .5
ties).  Gather outputs are integer-valued minus a constant: bit-exact.
"""
import ctypes

import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu

FLOAT_ATOL = 2e-3


@pytest.fixture(scope="module")
def pnn():
    import context_adaptive_neural_network_based_prediction_amd as P
    return P


def _check_pel(got, want):
    diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert diff.max() <= 1, "max |delta| = %d LSB" % diff.max()
    assert (diff != 0).mean() <= 1e-3, "%.4f %% of pixels differ" % (100 * (diff != 0).mean())


@pytest.mark.parametrize("w,n", [(4, 1), (4, 257), (8, 1), (8, 64), (8, 1000), (16, 33)])
def test_fc_matches_oracle(pnn, oracle, w, n):
    params = util.make_params(w, True, seed=10 + w, out_gain=60.0)
    above, left = util.make_contexts(w, n, seed=w * 1000 + n)
    ctx = util.flatten_fc(above, left)
    net = pnn.PredictionNeuralNetwork(n, w, True, params=params)
    got = net.predict(ctx)
    want = oracle.fc_forward(params, w, ctx)
    assert got.shape == (n, w, w, 1)
    np.testing.assert_allclose(got[..., 0], want, rtol=0, atol=FLOAT_ATOL)
    _check_pel(net.predict_pel(ctx), oracle.epilogue(want, util.MEAN))
    if n >= 64:
        assert want.min() + util.MEAN < 0 and want.max() + util.MEAN > 255, "test must exercise both clamps"


@pytest.mark.parametrize("w,n", [(4, 1), (4, 130), (8, 1), (8, 77), (16, 1), (16, 40), (32, 5), (64, 2)])
def test_conv_matches_oracle(pnn, oracle, w, n):
    params = util.make_params(w, False, seed=20 + w, out_gain=40.0)
    above, left = util.make_contexts(w, n, seed=w * 1000 + n + 1)
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    got = net.predict(above[..., None], left[..., None])
    want = oracle.conv_forward(params, w, above, left)
    np.testing.assert_allclose(got[..., 0], want, rtol=0, atol=FLOAT_ATOL)
    _check_pel(net.predict_pel(above, left), oracle.epilogue(want, util.MEAN))


def _device_tbs(pnn, xs, ys, flags, stride, w):
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    units = 2 * w // 4
    arr = (_lib.TbDev * len(xs))()
    for i in range(len(xs)):
        rc = L.pnn_make_tb_desc(ctypes.byref(arr[i]), int(ys[i]) * stride + int(xs[i]), stride,
                                flags[i].ctypes.data_as(_lib.u8p), int(flags[i].sum()), units, units)
        assert rc == 0
    return np.frombuffer(arr, dtype=np.uint8).copy()


@pytest.mark.parametrize("w", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("holes", [False, True])
def test_gather_bit_exact(pnn, oracle, w, holes):
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    n = 200
    plane = util.make_plane(320, 448, seed=w, pad=16)
    xs, ys, flags = util.make_tbs(320, 448, w, n, seed=w + 7, partial_fraction=0.6, holes=holes)
    net = pnn.PredictionNeuralNetwork(n, w, w <= 8, params=util.make_params(w, w <= 8, 1))
    d_plane = torch.from_numpy(plane).cuda()
    d_tbs = torch.from_numpy(_device_tbs(pnn, xs, ys, flags, plane.shape[1], w)).cuda()
    d_above = torch.full((n, w, 3 * w), float("nan"), device="cuda")
    d_left = torch.full((n, 2 * w, w), float("nan"), device="cuda")
    rc = L.pnn_gather_device(net.ctx, w, 4, d_plane.data_ptr(), 4, d_tbs.data_ptr(), n, d_above.data_ptr(), 3 * w * w,
                             d_left.data_ptr(), 2 * w * w, None)
    assert rc == 0
    torch.cuda.synchronize()
    ga, gl = d_above.cpu().numpy(), d_left.cpu().numpy()
    for i in range(n):
        rc, a, l = oracle.extract_context(plane, int(xs[i]), int(ys[i]), w, flags[i], util.MEAN)
        assert rc == 0
        assert np.array_equal(ga[i], a) and np.array_equal(gl[i], l), "TB %d differs" % i


@pytest.mark.parametrize("w,is_fc,n", [(4, True, 500), (8, True, 700), (16, False, 96), (8, False, 100), (32, False, 6)])
def test_fused_tbs_matches_oracle(pnn, oracle, w, is_fc, n):
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, is_fc, seed=30 + w, out_gain=30.0)
    plane = util.make_plane(256, 384, seed=100 + w, pad=8)
    xs, ys, flags = util.make_tbs(256, 384, w, n, seed=w + 11)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    d_plane = torch.from_numpy(plane).cuda()
    d_tbs = torch.from_numpy(_device_tbs(pnn, xs, ys, flags, plane.shape[1], w)).cuda()
    d_dst = torch.full((n, w, w), -1, dtype=torch.int32, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.pnn_predict_tbs_device(net.ctx, w, d_plane.data_ptr(), 4, d_tbs.data_ptr(), n, d_dst.data_ptr(), None, stream)
    assert rc == 0, L.pnn_last_error(net.ctx)
    torch.cuda.synchronize()
    want = oracle.predict_tbs(params, w, is_fc, plane, xs, ys, flags, util.MEAN)
    _check_pel(d_dst.cpu().numpy(), want)
