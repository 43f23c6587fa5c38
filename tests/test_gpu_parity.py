"""Parity of the HIP path (through the C ABI) against the CPU oracle -- the GPU tests proper.

Tolerances: the nets compute in float32 on both sides but in different summation orders (MFMA k-chunks
vs the oracle's sequential loops), so raw float predictions are compared with an absolute tolerance of
2e-3 on values of magnitude up to ~300, and HM-epilogue outputs (uint8 range) within +-1 LSB -- the
tolerance BASELINE.json states -- with at most 0.02 % of the pixels allowed to differ at all (exact .5
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


@pytest.fixture(autouse=True, params=["f32", "split_f16"])
def precision(request, monkeypatch):
    """Every test runs on both arithmetic paths of the tap GEMMs: exact-f32 MFMA and the 3 x f16 split-product MFMA
    (f32-class accuracy); the tolerances are the same for both."""
    monkeypatch.setenv("PNN_PRECISION", "0" if request.param == "f32" else "1")
    return request.param


def _check_pel(got, want):
    diff = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert diff.max() <= 1, "max |delta| = %d LSB" % diff.max()
    # exact .5 ties only: the bench's own count is 13 of 262 144 pixels (0.005 %); 0.02 % so that a regression shows
    # (+ one pixel, for cases of a few hundred pixels)
    assert (diff != 0).sum() <= 2e-4 * diff.size + 1, "%.4f %% of pixels differ" % (100 * (diff != 0).mean())


@pytest.mark.parametrize("w,n", [(4, 1), (4, 257), (8, 1), (8, 64), (8, 1000), (16, 33)])
def test_fc_matches_oracle(pnn, oracle, w, n):
    params = util.make_params(w, True, seed=10 + w, out_gain=util.out_gain(w, True))
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


# (8, 400), (16, 130), (32, 48), (64, 16): mid-size passes just above the f32 / split-precision crossover, where the
# rule-based tile choice (small LDS-resident-image tile, 64x128 ring tile) differs from the big-batch one
@pytest.mark.parametrize("w,n", [(4, 1), (4, 130), (8, 1), (8, 77), (16, 1), (16, 40), (32, 5), (64, 2),
                                 (8, 400), (16, 130), (32, 48), (64, 16)])
def test_conv_matches_oracle(pnn, oracle, w, n):
    params = util.make_params(w, False, seed=20 + w, out_gain=util.out_gain(w, False))
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
    params = util.make_params(w, is_fc, seed=30 + w, out_gain=util.out_gain(w, is_fc))
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


# ---- committed golden fixtures through the HIP path ----------------------------------------------------------
import os  # noqa: E402

from context_adaptive_neural_network_based_prediction_amd import weights as wts  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("is_fc,w", [(True, 4), (True, 8), (True, 16), (False, 4), (False, 8), (False, 16), (False, 32),
                                     (False, 64)])
def test_golden_nets(pnn, is_fc, w):
    g = np.load(os.path.join(GOLD, "nets.npz"))
    tag = "%s%d" % ("fc" if is_fc else "conv", w)
    seed, n = int(g[tag + "_seed"]), int(g[tag + "_n"])
    params = util.make_params(w, is_fc, seed, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, seed + 1)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    got = net.predict(util.flatten_fc(above, left)) if is_fc else net.predict(above, left)
    np.testing.assert_allclose(got[..., 0], g[tag + "_out"], rtol=0, atol=FLOAT_ATOL)


@pytest.mark.parametrize("w", [4, 8])
def test_real_checkpoints_on_gpu(pnn, oracle, w):
    """The reference's two trained models (converted from its TF checkpoints) through the HIP path."""
    g = np.load(os.path.join(GOLD, "nets.npz"))
    net = pnn.PredictionNeuralNetwork(8, w, False, path_to_model=os.path.join(GOLD, "conv%d_single.pnnw" % w))
    got = net.predict(g["real%d_above" % w], g["real%d_left" % w])
    np.testing.assert_allclose(got[..., 0], g["real%d_out" % w], rtol=0, atol=FLOAT_ATOL)
    _check_pel(net.predict_pel(g["real%d_above" % w], g["real%d_left" % w]), oracle.epilogue(g["real%d_out" % w], util.MEAN))
    img = np.tile(np.linspace(60, 180, 3 * w)[None, :], (3 * w, 1))
    img[:, w + 1] = 230                                             # SURVEY.md Appendix A anchor
    pred = net.predict_pel((img[None, 0:w, :] - util.MEAN).astype(np.float32), (img[None, w:3 * w, 0:w] - util.MEAN).astype(np.float32))[0]
    assert (pred.argmax(axis=1) == 1).all()
    if w == 4:
        assert np.abs(pred - np.array([[102, 225, 126, 131], [105, 228, 132, 136], [100, 225, 131, 137], [112, 213, 133, 140]])).max() <= 1


def test_python_gather_twin(pnn):
    """context.extract_context_portions_targets_from_channels_plus_preprocessing == sets/common.py outputs."""
    from context_adaptive_neural_network_based_prediction_amd import context
    g = np.load(os.path.join(GOLD, "gather_python.npz"))
    img = g["images"]
    net = pnn.PredictionNeuralNetwork(1, 4, True, params=util.make_params(4, True, 1))
    for k in range(int(g["n_cases"])):
        w, mw, mh, is_fc = [int(v) for v in g["k%d_meta" % k]]
        res = context.extract_context_portions_targets_from_channels_plus_preprocessing(
            img, w, g["k%d_rows" % k], g["k%d_cols" % k], util.MEAN, (mw, mh), bool(is_fc), predictor=net)
        assert len(res) == (2 if is_fc else 3)
        for i, r in enumerate(res):
            assert r.shape == g["k%d_out%d" % (k, i)].shape and np.array_equal(r, g["k%d_out%d" % (k, i)]), (k, i)
    with pytest.raises(ValueError):
        context.extract_context_portions_targets_from_channels_plus_preprocessing(
            img, 8, np.array([0]), np.array([0]), util.MEAN, (6, 0), True, predictor=net)
    with pytest.raises(TypeError):
        context.extract_context_portions_targets_from_channels_plus_preprocessing(
            img.astype(np.float32), 8, np.array([0]), np.array([0]), util.MEAN, (0, 0), True, predictor=net)


def test_predict_by_batch_via_pnn(pnn, oracle):
    """pnn/batching.py semantics: N = 6, batch 2 (test_pnn.py:451-506): identical inputs give identical outputs and
    the result does not depend on the batch size."""
    w = 8
    params = util.make_params(w, True, 7, out_gain=util.out_gain(w, True))
    above, left = util.make_contexts(w, 2, 8)
    ctx = util.flatten_fc(above, left)
    ctx = np.concatenate([ctx[:1], np.repeat(ctx[1:2], 5, axis=0)], axis=0)
    net = pnn.PredictionNeuralNetwork(2, w, True, params=params)
    out2 = pnn.predict_by_batch_via_pnn((ctx,), None, net, 2)
    out6 = pnn.predict_by_batch_via_pnn((ctx,), None, net, 6)
    assert out2.shape == (6, w, w, 1) and out2.dtype == np.float32
    assert np.array_equal(out2, out6)
    assert all(np.array_equal(out2[i], out2[1]) for i in range(2, 6)) and not np.array_equal(out2[0], out2[1])
    np.testing.assert_allclose(out2[..., 0], oracle.fc_forward(params, w, ctx), rtol=0, atol=FLOAT_ATOL)
    wc = 16
    pc = util.make_params(wc, False, 9, out_gain=util.out_gain(wc, False))
    a, l = util.make_contexts(wc, 4, 10)
    netc = pnn.PredictionNeuralNetwork(2, wc, False, params=pc)
    outc = pnn.predict_by_batch_via_pnn((a[..., None], l[..., None]), None, netc, 2)
    np.testing.assert_allclose(outc[..., 0], oracle.conv_forward(pc, wc, a, l), rtol=0, atol=FLOAT_ATOL)


def test_model_table_create_and_hm_call_sequence(pnn, oracle, tmp_path):
    """pnn_create from a `width,is_pair,channel,path` table (TComPrediction.cpp:143-178), then exactly what HM does per
    TB: host gather into the per-width tensor, one Run, epilogue into a strided Pel block (TComPrediction.cpp:554-635)."""
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    entries = []
    params = {}
    for pair in (0, 1):
        for w in (4, 8, 16, 32, 64):
            is_fc = w <= 8
            flat = wts.init_params(w, is_fc, seed=100 * pair + w, bias_std=0.05)
            if w == 8 or w == 16:
                flat = util.make_params(w, is_fc, seed=100 * pair + w, out_gain=util.out_gain(w, is_fc))
            params[(w, pair)] = flat
            name = "w%d_%s.pnnw" % (w, "pair" if pair else "single")
            wts.save_pnnw(str(tmp_path / name), flat, w, is_fc)
            entries.append((w, pair, 0, name))                       # relative to the table's directory
    table = wts.write_model_table(str(tmp_path / "table.txt"), entries)
    for use_pair in (0, 1):
        ctx = ctypes.c_void_p()
        rc = L.pnn_create(ctypes.byref(ctx), table.encode(), use_pair, ctypes.c_float(util.MEAN), 0)
        assert rc == 0, L.pnn_last_error(None)
        fc, nl, npar = ctypes.c_int(), ctypes.c_int(), ctypes.c_long()
        assert L.pnn_model_info(ctx, 8, ctypes.byref(fc), ctypes.byref(nl), ctypes.byref(npar)) == 0
        assert (fc.value, nl.value, npar.value) == (1, 4, 3344464)
        assert L.pnn_model_info(ctx, 16, ctypes.byref(fc), ctypes.byref(nl), ctypes.byref(npar)) == 0
        assert (fc.value, nl.value, npar.value) == (0, 13, 1339073)
        for w in (8, 16):
            plane = util.make_plane(96, 128, seed=w, pad=8)
            xs, ys, flags = util.make_tbs(96, 128, w, 1, seed=w + use_pair, partial_fraction=1.0)
            units = 2 * w // 4
            buf = np.zeros(5 * w * w, np.float32)                    # one tensor for w <= 8, left at +3w^2 (TComPattern.cpp:352-353)
            above, left = buf[:3 * w * w], buf[3 * w * w:]
            origin = ctypes.cast(plane.ctypes.data + 4 * (int(ys[0]) * plane.shape[1] + int(xs[0])), _lib.i32p)
            assert L.pnn_extract_context(origin, above.ctypes.data_as(_lib.f32p), left.ctypes.data_as(_lib.f32p),
                                         flags[0].ctypes.data_as(_lib.u8p), int(flags[0].sum()), 4, 4, units, units, w, w,
                                         plane.shape[1], ctypes.c_float(util.MEAN)) == 0
            stride = 64
            dst = np.full((w, stride), -7, np.int32)
            rc = L.pnn_predict_pel(ctx, w, above.ctypes.data_as(_lib.f32p), left.ctypes.data_as(_lib.f32p), 1,
                                   dst.ctypes.data_as(_lib.i32p), stride)
            assert rc == 0, L.pnn_last_error(ctx)
            want = oracle.predict_tbs(params[(w, use_pair)], w, w <= 8, plane, xs, ys, flags, util.MEAN)[0]
            _check_pel(dst[:, :w], want)
            assert (dst[:, w:] == -7).all()                          # nothing written outside the block
        # wrong kind / missing width are errors, not fallbacks
        out = np.zeros((1, 8, 8), np.float32)
        assert L.pnn_predict_conv(ctx, 8, out.ctypes.data_as(_lib.f32p), out.ctypes.data_as(_lib.f32p), 1,
                                  out.ctypes.data_as(_lib.f32p)) == -3
        L.pnn_destroy(ctx)
    bad = tmp_path / "bad.txt"
    bad.write_text("4,0,0,w4_single.pnnw\n")
    ctx = ctypes.c_void_p()
    assert L.pnn_create(ctypes.byref(ctx), str(bad).encode(), 0, ctypes.c_float(util.MEAN), 0) == -3   # 8..64 missing


def test_torch_device_path_and_empty_batch(pnn, oracle):
    import torch
    w = 16
    params = util.make_params(w, False, 11, out_gain=util.out_gain(w, False))
    a, l = util.make_contexts(w, 9, 12)
    net = pnn.PredictionNeuralNetwork(9, w, False, params=params)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                                       # a non-default stream: the library must follow it
        out = net.predict(torch.from_numpy(a).cuda()[..., None], torch.from_numpy(l).cuda()[..., None])
    s.synchronize()
    assert out.is_cuda and out.shape == (9, w, w, 1)
    np.testing.assert_allclose(out.cpu().numpy()[..., 0], oracle.conv_forward(params, w, a, l), rtol=0, atol=FLOAT_ATOL)
    assert net.predict(a[:0, ..., None], l[:0, ..., None]).shape == (0, w, w, 1)
    with pytest.raises(ValueError):
        net.predict(a)                                               # conv nets take two inputs


@pytest.mark.parametrize("w,is_fc,n", [(8, True, 300), (4, False, 130), (16, False, 21), (32, False, 3)])
def test_split_gemm_kernel_families_bit_identical(pnn, precision, w, is_fc, n):
    """The three split-precision GEMM kernels (register-staged tapgemm_sp, LDS-resident-image convimg_sp, LDS-DMA ring
    with loader waves) and all their tile shapes keep one per-output summation order: forcing any configuration code on
    every layer it can run must not change a single bit of the float predictions (ragged M, idle rows/columns, taps
    that leave the image, stride-2 transposed-convolution classes are all in these nets)."""
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, is_fc, 31, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 32)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    run = (lambda: net.predict(util.flatten_fc(above, left))) if is_fc else (lambda: net.predict(above, left))
    net.set_option("canonical_order", 1)
    net.set_option("ring", 0)
    net.set_option("convimg", 0)
    want = run()
    net.set_option("ring", 1)
    net.set_option("convimg", 1)
    ncodes = L.pnn_num_split_configs()
    assert ncodes >= 40
    for rep in range(2):                                             # a pipeline race would show as a rare wrong tile
        for code in range(ncodes):
            net.set_option("sp_cfg", code)
            assert np.array_equal(run(), want), "split-GEMM configuration code %d changes the result" % code


@pytest.mark.parametrize("w,n", [(16, 384), (16, 300), (16, 1200), (8, 1024), (32, 130)])
def test_ring_position_major_tiles_bit_identical(pnn, oracle, precision, w, n):
    """Convolutions at batch on the ring kernel take position-major tiles (a tile = many blocks at ONE position of the map)
    and skip the taps that only meet the SAME padding there.  The skipped products are exact zeros: with the option off
    (ring_pm = 0: block-major tiles, every tap) not one bit of the float predictions may differ -- whole block groups (384 = 3 x 128),
    a ragged last group (300, 130), a partial chunk of groups behind a whole one (1200 = 9.4 x 128), stride-2 convolutions and the four classes of the stride-2 transposed ones."""
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, False, 77, out_gain=util.out_gain(w, False))
    above, left = util.make_contexts(w, n, 78)
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    net.set_option("canonical_order", 1)
    net.set_option("autotune", 0)
    ncodes = L.pnn_num_split_configs()
    want = None
    for code in range(-1, ncodes):                                   # -1: the rule-based choice; the ring kernel's codes are among the rest
        net.set_option("sp_cfg", code)
        net.set_option("ring_pm", 0)
        plain = net.predict(above, left)
        for mode in (2, 1):                                          # 2: wherever possible; 1 (default): where the launch model expects a gain
            net.set_option("ring_pm", mode)
            assert np.array_equal(net.predict(above, left), plain), "position-major tiles change the result (configuration code %d)" % code
        if want is None:
            want = plain
        assert np.array_equal(plain, want)
    m = min(n, 48)
    np.testing.assert_allclose(want[:m, ..., 0], oracle.conv_forward(params, w, above[:m], left[:m]), rtol=0, atol=FLOAT_ATOL)
    net.close()


@pytest.mark.parametrize("w,is_fc,n", [(16, False, 384), (16, False, 300), (8, False, 1024), (32, False, 130), (64, False, 9), (8, True, 1500), (4, True, 2048), (16, False, 3)])
def test_f32_tiles_and_position_major_bit_identical(pnn, oracle, precision, w, is_fc, n):
    """The exact-f32 path (tapgemm_f32_kernel): every tile configuration gives the same float bits (one per-output summation order),
    with block-major tiles and with position-major ones (which skip the taps that only meet SAME padding: exact zeros) -- whole
    block groups, ragged ones, stride-2 layers, the four classes of the stride-2 transposed convolutions; FC nets with the
    output layer fused into the last hidden layer's launch (n >= 1024) or from stored activations."""
    if precision != "f32":
        pytest.skip("exact-f32 kernels only")
    params = util.make_params(w, is_fc, 177, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 178)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    net.set_option("autotune", 0)
    run = (lambda: net.predict(util.flatten_fc(above, left))) if is_fc else (lambda: net.predict(above, left))
    net.set_option("ring_pm", 0)
    want = run()
    from context_adaptive_neural_network_based_prediction_amd import _lib
    for cfg in range(-1, _lib.lib().pnn_num_f32_configs()):
        net.set_option("f32_cfg", cfg)
        for mode in (0, 2, 1):
            net.set_option("ring_pm", mode)
            assert np.array_equal(run(), want), "tapgemm_f32 configuration %d, position-major mode %d changes the result" % (cfg, mode)
    net.set_option("f32_cfg", -1)
    net.set_option("fuse_last", 0)                                   # FC: the output layer from stored activations
    assert np.array_equal(run(), want)
    if not is_fc:
        # the deep layers' K segments (32x32 / 64x64 nets): as separate workgroups + seg_reduce_kernel, and in sequence inside the
        # workgroups with a running total -- the same order of additions, the same bits, on every tile
        for seg_mode in (0, 1):
            net.set_option("f32_seg_mode", seg_mode)
            for cfg in range(-1, _lib.lib().pnn_num_f32_configs()):
                net.set_option("f32_cfg", cfg)
                for mode in (0, 2):
                    net.set_option("ring_pm", mode)
                    assert np.array_equal(run(), want), "K segments in form %d on tile %d, position-major mode %d change the result" % (seg_mode, cfg, mode)
        net.set_option("f32_seg_mode", 0); net.set_option("ring_pm", 1)
        # persistent workgroups (each takes its tiles one after the other): never, one, two, three per CU
        for persist in (0, 1, 2, 3):
            net.set_option("f32_persist", persist)
            for cfg in (-1, 5, 6, 7, 8):
                net.set_option("f32_cfg", cfg)
                assert np.array_equal(run(), want), "%d persistent workgroups per CU on tile %d change the result" % (persist, cfg)
        net.set_option("f32_persist", -1); net.set_option("f32_cfg", -1)
    m = min(n, 48)
    ref = oracle.fc_forward(params, w, util.flatten_fc(above[:m], left[:m])) if is_fc else oracle.conv_forward(params, w, above[:m], left[:m])
    np.testing.assert_allclose(want[:m, ..., 0], ref, rtol=0, atol=FLOAT_ATOL)
    net.close()


@pytest.mark.parametrize("w,is_fc,n", [(4, True, 37), (8, True, 50), (4, False, 45), (8, False, 21), (16, False, 11), (32, False, 5), (64, False, 2)])
def test_f32_small_kernel_bit_identical(pnn, oracle, precision, w, is_fc, n):
    """Round 5: tapgemm_f32_small_kernel -- the canonical f32 order (a k-ordered fmaf chain, per 16-deep chunk k = 0, 8, 1, 9, ...) issued
    through v_mfma_f32_16x16x4_f32 with its lane groups fed the k the chain visits next, one wave per 16 x 16 tile -- gives the float bits
    of tapgemm_f32_kernel's 32x32x2 chain: single blocks, handfuls, ragged row tiles, stride-2 layers, the four classes of the stride-2
    transposed convolutions, the K segments of the deep layers (32x32 / 64x64 nets), FC inputs riding in the argument block."""
    if precision != "f32":
        pytest.skip("exact-f32 kernels only")
    params = util.make_params(w, is_fc, 277, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 278)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    net.set_option("autotune", 0)
    run = (lambda a, l: net.predict(util.flatten_fc(a, l))) if is_fc else (lambda a, l: net.predict(a, l))
    net.set_option("f32_small", 0)
    want = run(above, left)
    alone = [run(above[i:i + 1], left[i:i + 1]) for i in range(min(n, 3))]
    for i, a in enumerate(alone):
        assert np.array_equal(a[0], want[i])
    net.set_option("f32_small", 1)
    for limit in (1 << 30, 1024, 40):                               # every layer on the small kernel / the default rule / only the smallest
        net.set_option("f32_small_max_tiles", limit)
        assert np.array_equal(run(above, left), want), "f32_small_max_tiles = %d changes the result" % limit
        for i in range(min(n, 3)):
            assert np.array_equal(run(above[i:i + 1], left[i:i + 1])[0], want[i]), "block %d alone, f32_small_max_tiles = %d" % (i, limit)
        for m in (2, 3, 7, 17):
            if m <= n:
                assert np.array_equal(run(above[n - m:], left[n - m:]), want[n - m:]), "%d blocks, f32_small_max_tiles = %d" % (m, limit)
    pel = net.predict_pel(*((util.flatten_fc(above, left),) if is_fc else (above, left)))
    net.set_option("f32_small_max_tiles", 1024)
    assert np.array_equal(run(above[:1], left[:1])[0], want[0])
    launches = net.last_call_stats()["launches"]
    net.set_option("seg_fold", 0)                                    # K-segmented layers (32x32 / 64x64 nets): planes + seg_reduce launch instead of the in-launch sum
    assert np.array_equal(run(above, left), want)
    assert np.array_equal(run(above[:1], left[:1])[0], want[0])
    assert net.last_call_stats()["launches"] == launches + ({32: 6, 64: 9}.get(w, 0) if not is_fc else 0)   # what the in-launch sum saves per single-block call
    net.set_option("seg_fold", 1)
    for deep in (0, 2, 1):                                           # the weight ring 6 or 12 stages ahead of the chain: the same sums
        net.set_option("f32_small_deep", deep)
        assert np.array_equal(run(above, left), want), "f32_small_deep = %d" % deep
        assert np.array_equal(run(above[:1], left[:1])[0], want[0]), "f32_small_deep = %d, one block" % deep
    for _ in range(3):                                               # the tiles' counters go back to zero: launch after launch
        assert np.array_equal(run(above[:1], left[:1])[0], want[0])
    for chain in (0, 1):                                             # hidden tensors in channel order / in the chain's order (round 6): the same sums
        net.set_option("chain_io", chain)
        assert np.array_equal(run(above, left), want), "chain_io = %d" % chain
        for m in (1, 2, 17):
            if m <= n:
                assert np.array_equal(run(above[:m], left[:m]), want[:m]), "chain_io = %d, %d blocks" % (chain, m)
    net.set_option("fc_out_f32", 0)                                  # FC: the output layer's K segments and their reduction as two launches
    assert np.array_equal(run(above, left), want)
    assert np.array_equal(net.predict_pel(*((util.flatten_fc(above, left),) if is_fc else (above, left))), pel)
    net.set_option("f32_small", 0)
    assert np.array_equal(net.predict_pel(*((util.flatten_fc(above, left),) if is_fc else (above, left))), pel)
    m = min(n, 24)
    ref = oracle.fc_forward(params, w, util.flatten_fc(above[:m], left[:m])) if is_fc else oracle.conv_forward(params, w, above[:m], left[:m])
    np.testing.assert_allclose(want[:m, ..., 0], ref, rtol=0, atol=FLOAT_ATOL)
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w", [4, 8, 16, 32])
def test_small_conv_passes_run_merger_and_last_layer_as_tails(pnn, precision, w):
    """Small exact-f32 conv passes (round 6): the merger runs as the TAIL of the branches' last pair launch -- per (block, channel group),
    by the last of its five tiles to arrive -- and the last transposed convolution as the tail of the GEMM in front of it, per block
    (option "tails", pnn_gemm_f32_small.hip).  Same bodies, same bits: float and Pel predictions of 1 ... 9 blocks equal those of the
    layer-by-layer launches and of a large batch; two launches less per call where the shapes allow it; call after call (the
    counters go back to zero); from the picture plane too (pnn_predict_tbs_device's small passes)."""
    if precision != "f32":
        pytest.skip("exact-f32 kernels only")
    params = util.make_params(w, False, 71, out_gain=util.out_gain(w, False))
    big = {4: 400, 8: 300, 16: 150, 32: 40}[w]
    above, left = util.make_contexts(w, big, 72)
    net = pnn.PredictionNeuralNetwork(big, w, False, params=params)
    want_f, want_p = net.predict(above, left), net.predict_pel(above, left)
    saved = {}
    for n in (1, 2, 3, 6, 9):
        res = {}
        for tails in (0, 1):
            net.set_option("tails", tails)
            for rep in range(3 if tails else 1):
                f = net.predict(above[:n], left[:n])
                launches = net.last_call_stats()["launches"]
                q = net.predict_pel(above[:n], left[:n])
                assert np.array_equal(f, want_f[:n]), "tails = %d, %d blocks, call %d: float predictions differ from the large batch's" % (tails, n, rep)
                assert np.array_equal(q, want_p[:n]), "tails = %d, %d blocks, call %d: Pel blocks differ" % (tails, n, rep)
            res[tails] = launches
        saved[n] = res[0] - res[1]
        assert saved[n] in (0, 1, 2), saved
    if w in (4, 8, 16):
        assert saved[1] == 2 and saved[3] == 2, saved                # both tails (32x32: K segments in front of the merger, four classes in front of the last layer)
    # ... and with other options that move the small kernels' data around
    for opts in ({"chain_io": 0}, {"f32_small_deep": 2}, {"pair": 0}, {"flag_wait": 0}):
        for k, v in opts.items():
            net.set_option(k, v)
        assert np.array_equal(net.predict(above[:2], left[:2]), want_f[:2]), opts
        assert np.array_equal(net.predict_pel(above[:1], left[:1]), want_p[:1]), opts
        for k in opts:
            net.set_option(k, {"chain_io": 1, "f32_small_deep": 1, "pair": 1, "flag_wait": 1}[k])
    net.close()


@pytest.mark.gpu
def test_small_passes_of_five_contexts_side_by_side_keep_their_bits(precision):
    """tools/tails_stress.py: five host threads, one context each (the batching service's layout), random small batches back to back
    for 8 s -- every call equals the large batch's bits.  The first form of the tails failed this 1 call in 10 000 (tiles of two
    workgroups sharing 128-byte lines between XCDs) and passed every single-context test."""
    if precision != "f32":
        pytest.skip("exact-f32 kernels only")
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "tails_stress.py"), "8", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("tails =")][-1]
    import re
    counts = re.findall(r"'(\w+)': \((\d+), (\d+)", line)
    assert len(counts) == 5 and all(int(bad) == 0 and int(calls) > 1000 for _, bad, calls in counts), line


@pytest.mark.gpu
@pytest.mark.parametrize("w,is_fc,slice_blocks,n", [(8, True, 512, 2300), (4, True, 1024, 5000), (16, False, 128, 700), (32, False, 32, 100)])
def test_host_calls_of_several_slices_overlap_and_keep_the_bits(pnn, precision, w, is_fc, slice_blocks, n):
    """Host-array calls of several passes' worth of blocks (the reference's batched driver, pnn/batching.py:7-88) run slice
    by slice on two staging sets -- slice i + 1 copied in and slice i - 1 copied out beside the pass of slice i ("host_slice"): the float
    predictions, the Pel blocks (also through a strided destination) and predict_by_batch_via_pnn give the bits of the one-copy-in /
    one-copy-out call, ragged last slice included."""
    from context_adaptive_neural_network_based_prediction_amd import _lib, predict_by_batch_via_pnn
    params = util.make_params(w, is_fc, 511, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 512)
    ins = (util.flatten_fc(above, left),) if is_fc else (above[..., None], left[..., None])
    net = pnn.PredictionNeuralNetwork(slice_blocks, w, is_fc, params=params)
    net.set_option("host_slice", -1)
    want_f, want_p = net.predict(*ins), net.predict_pel(*ins)
    net.set_option("host_slice", slice_blocks)
    assert np.array_equal(net.predict(*ins), want_f)
    assert np.array_equal(net.predict_pel(*ins), want_p)
    assert np.array_equal(net.predict(*[a[:2 * slice_blocks] for a in ins]), want_f[:2 * slice_blocks])       # exactly two slices
    assert np.array_equal(net.predict(*[a[:2 * slice_blocks - 1] for a in ins]), want_f[:2 * slice_blocks - 1])   # one short of two: the one-pass form
    net.set_option("host_slice", 0)                                    # the default slice (a bench batch): this call is below two of them or not -- same bits
    assert np.array_equal(net.predict(*ins), want_f)
    nb = (n // slice_blocks) * slice_blocks
    got = predict_by_batch_via_pnn(tuple(a[:nb] for a in ins), None, net, slice_blocks)
    assert got.shape == (nb, w, w, 1) and np.array_equal(got, want_f[:nb])
    net.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,is_fc,slice_blocks,n", [(8, True, 256, 1100), (16, False, 64, 300)])
def test_sliced_host_call_edge_cases(pnn, precision, w, is_fc, slice_blocks, n):
    """The sliced host call's own corners: a non-finite value in a LATE slice (found by the feeder thread while earlier slices already
    run) fails the call like the one-pass form does and leaves the context usable; in the split mode a block that leaves the f16 range
    in a middle slice sends the call through the exact-f32 kernels -- the result of the unsliced call, bit for bit."""
    from context_adaptive_neural_network_based_prediction_amd import _lib
    params = util.make_params(w, is_fc, 611, out_gain=util.out_gain(w, is_fc)).copy()
    above, left = util.make_contexts(w, n, 612, masked_fraction=0.0)
    pack = lambda a, l: (util.flatten_fc(a, l),) if is_fc else (a, l)
    net = pnn.PredictionNeuralNetwork(slice_blocks, w, is_fc, params=params)
    net.set_option("host_slice", slice_blocks)
    want = net.predict(*pack(above, left))
    bad = above.copy()
    bad[n - 5].reshape(-1)[3] = np.nan                                  # last slice
    with pytest.raises(_lib.PnnError, match="non-finite"):
        net.predict(*pack(bad, left))
    assert np.array_equal(net.predict(*pack(above, left)), want)        # nothing of the failed call is left behind
    net.close()
    if precision != "split_f16":
        return
    specs = wts.tensor_specs(w, is_fc)
    offs = np.concatenate([[0], np.cumsum([int(np.prod(sh)) for _, sh, _ in specs])])
    gain = 300.0 if is_fc else 1000.0                                   # as in test_range_fallback_touches_only_the_overflowing_block
    params[offs[0]:offs[2]] *= gain
    params[offs[-3]:offs[-2]] /= gain
    hot = 2 * slice_blocks + 7                                          # a block of the third slice
    above[hot] *= 40.0
    left[hot] *= 40.0
    net = pnn.PredictionNeuralNetwork(slice_blocks, w, is_fc, params=params)
    net.set_option("host_slice", -1)
    one_pass = net.predict(*pack(above, left))
    net.set_option("host_slice", slice_blocks)
    got = net.predict(*pack(above, left))
    assert np.isfinite(got).all()
    # both forms repeat a call of more than 256 blocks WHOLE on the exact-f32 kernels when one block overflowed (pnn_abi.cpp): same bits
    assert np.array_equal(got, one_pass)
    fb = ctypes.c_long()
    assert _lib.lib().pnn_check_range(net.ctx, None, ctypes.byref(fb)) == 0 and fb.value >= 1
    net.close()


@pytest.mark.gpu
def test_bench_one_rank_through_rccl(precision):
    """`python bench.py --gpus 1 --force-dist nccl` (VERDICT r5 #8a): the RCCL branch of the N > 1 path executed on the hardware that is
    there -- a ONE-rank group on the real device; the opening barrier and the max-over-ranks clock of every timed region, the device
    census and a gather of predictions all go through RCCL.  The line says so (rccl_ranks_seen) and is otherwise the N = 1 line."""
    import json
    import subprocess
    import sys
    if precision != "f32":
        pytest.skip("once is enough")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PNN_AUTOTUNE="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    for k in ("PNN_PRECISION", "WORLD_SIZE", "RANK", "LOCAL_RANK", "PNN_BENCH_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "nccl", "--steps", "10", "--warmup", "2",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["dtype"] == "f32" and d["value"] > 1e6
    assert d["rccl_ranks_seen"] == {"backend": "nccl", "world_size": 1, "devices": 1, "gather_predictions_ok": True, "forced_single_rank_group": True}
    assert d["max_abs_lsb_vs_oracle"] is not None and d["max_abs_lsb_vs_oracle"] <= 1


@pytest.mark.parametrize("w,is_fc", [(4, True), (8, True), (16, False), (32, False), (64, False)])
def test_small_calls_replayed_as_graphs(pnn, precision, w, is_fc):
    """Option "graphs" (opt-in): the launch chain of a small host call is captured on the second call of a shape (model, blocks,
    result kinds) and replayed afterwards with one hipGraphLaunch.  Every call -- first (plain launches), second (captured), later
    (replayed), with new inputs each time and shapes interleaved -- gives the bits of the same call with the option off; an option
    change drops the captured chains."""
    params = util.make_params(w, is_fc, 811, out_gain=util.out_gain(w, is_fc))
    net = pnn.PredictionNeuralNetwork(8, w, is_fc, params=params)
    net.set_option("graphs", 0)
    calls = []
    for i, n in enumerate((1, 1, 3, 1, 3, 1, 3, 2, 1, 3, 7, 1)):
        above, left = util.make_contexts(w, n, 820 + i)
        ins = (util.flatten_fc(above, left),) if is_fc else (above, left)
        calls.append((ins, net.predict(*ins).copy(), net.predict_pel(*ins).copy()))
    launches_plain = net.last_call_stats()["launches"]
    net.set_option("graphs", 1)
    for rnd in range(2):
        for i, (ins, want_f, want_p) in enumerate(calls):
            assert np.array_equal(net.predict(*ins), want_f), "float result, call %d of round %d" % (i, rnd)
            assert np.array_equal(net.predict_pel(*ins), want_p), "Pel result, call %d of round %d" % (i, rnd)
    assert net.last_call_stats()["launches"] == launches_plain        # a replayed call reports the chain it stands for
    net.set_option("cache_mb", 0)                                      # any option change: the chains are captured anew
    for ins, want_f, want_p in calls[:6]:
        assert np.array_equal(net.predict_pel(*ins), want_p)
    net.close()


def test_streams_on_distinct_hardware_queues(pnn):
    """pnn_streams_on_distinct_queues: four streams measured to sit on four different hardware queues (the runtime deals streams onto
    four); a context that adopts one (option "stream") computes the same predictions there, before and after, and the streams outlive it."""
    import ctypes
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    w = 8
    params = util.make_params(w, True, 71, out_gain=util.out_gain(w, True))
    above, left = util.make_contexts(w, 5, 72)
    ctx = util.flatten_fc(above, left)
    net = pnn.PredictionNeuralNetwork(5, w, True, params=params)
    want = net.predict_pel(ctx).copy()
    streams = (ctypes.c_void_p * 4)()
    n = L.pnn_streams_on_distinct_queues(streams, 4)
    assert n == 4 and len({int(s) for s in streams}) == 4
    for s in streams:
        net.set_option("stream", int(s))
        for m in (1, 5):
            assert np.array_equal(net.predict_pel(ctx[:m]), want[:m])
    more = (ctypes.c_void_p * 8)()
    assert 4 <= L.pnn_streams_on_distinct_queues(more, 8) <= 8          # (as many as the runtime has queues; never fewer than the four above)
    L.pnn_streams_release(more, 8)
    assert L.pnn_streams_on_distinct_queues(more, 0) == -1 and L.pnn_set_option(net.ctx, b"stream", 0) == -1
    net.close()                                                         # the adopted stream is not the context's to destroy
    L.pnn_streams_release(streams, 4)


def test_arithmetic_tag_names_the_summation_order(pnn):
    """pnn_arithmetic_tag: one string per arithmetic, the same for every context and width of a library build; it changes with the
    "precision" option and with nothing else -- what an encoder and its decoder compare once at start-up."""
    nets = [pnn.PredictionNeuralNetwork(1, w, fc, params=util.make_params(w, fc, 5)) for w, fc in ((4, True), (16, False))]
    tags = {}
    for prec in (0, 1):
        for net in nets:
            net.set_option("precision", prec)
            net.set_option("autotune", 0); net.set_option("pair", 0)        # options that do not touch the bits do not touch the tag
            tags.setdefault(prec, set()).add(net.arithmetic_tag())
    assert len(tags[0]) == 1 and len(tags[1]) == 1 and tags[0] != tags[1]
    assert "f32" in next(iter(tags[0])) and "kseg 1600/2304" in next(iter(tags[0])) and "split" in next(iter(tags[1]))
    for net in nets:
        net.close()


@pytest.mark.parametrize("w,is_fc,n", [(8, True, 3001), (16, False, 333)])
def test_pinned_caller_arrays(pnn, oracle, precision, w, is_fc, n):
    """Batched host-array calls (the shape of Session::Run / pnn/batching.py:7-88) from PINNED caller arrays (pnn_host_alloc: the copy
    engines reach them directly) give the bits of the same call from pageable arrays; a non-finite input is refused from either."""
    import ctypes
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, is_fc, 311, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 312)
    ins = (util.flatten_fc(above, left),) if is_fc else (above, left)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    want_f, want_p = net.predict(*ins), net.predict_pel(*ins)
    handles, pinned = [], []
    for a in ins:
        p = ctypes.c_void_p()
        assert L.pnn_host_alloc(ctypes.byref(p), a.nbytes) == 0
        b = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_float)), shape=(a.size,)).reshape(a.shape)
        b[...] = a
        handles.append(p); pinned.append(b)
    po, pd = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.pnn_host_alloc(ctypes.byref(po), n * w * w * 4) == 0 and L.pnn_host_alloc(ctypes.byref(pd), n * w * w * 4) == 0
    out = np.ctypeslib.as_array(ctypes.cast(po, ctypes.POINTER(ctypes.c_float)), shape=(n * w * w,))
    dst = np.ctypeslib.as_array(ctypes.cast(pd, ctypes.POINTER(ctypes.c_int32)), shape=(n * w * w,))
    out[...] = -1; dst[...] = -1
    call = lambda: L.pnn_predict_f32_pel(net.ctx, w, pinned[0].ctypes.data_as(_lib.f32p), None if is_fc else pinned[1].ctypes.data_as(_lib.f32p), n,
                                         ctypes.cast(po, _lib.f32p), ctypes.cast(pd, _lib.i32p))
    assert call() == 0, L.pnn_last_error(net.ctx)
    assert np.array_equal(out.reshape(want_f.shape), want_f) and np.array_equal(dst.reshape(want_p.shape), want_p)
    pinned[0].reshape(-1)[pinned[0].size // 2 + 3] = np.inf              # the vectorised finite check sees one bad value anywhere
    assert call() == -1 and b"non-finite" in L.pnn_last_error(net.ctx)
    net.close()
    for h in handles + [po, pd]:
        L.pnn_host_free(h)


@pytest.mark.parametrize("w,is_fc", [(8, True), (16, False)])
def test_prediction_cache_for_single_block_calls(pnn, w, is_fc):
    """`cache_mb`: a repeated single-block call (HM's RDO re-evaluates the same TB) is answered from the cache with the
    very same values, for both result kinds; a different context misses; changing an option drops the entries."""
    params = util.make_params(w, is_fc, 41, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, 3, 42)
    net = pnn.PredictionNeuralNetwork(1, w, is_fc, params=params)
    one = (lambda i: (util.flatten_fc(above[i:i + 1], left[i:i + 1]),)) if is_fc else (lambda i: (above[i:i + 1], left[i:i + 1]))
    ref_f = [net.predict(*one(i)).copy() for i in range(3)]
    ref_p = [net.predict_pel(*one(i)).copy() for i in range(3)]
    net.set_option("cache_mb", 8)
    assert net.cache_stats() == (0, 0)
    assert np.array_equal(net.predict_pel(*one(0)), ref_p[0])      # miss, fills the entry
    assert net.cache_stats() == (0, 1)
    assert np.array_equal(net.predict_pel(*one(0)), ref_p[0])      # hit
    assert np.array_equal(net.predict(*one(0)), ref_f[0])          # hit: the entry holds the float prediction too
    assert net.cache_stats() == (2, 1)
    assert np.array_equal(net.predict(*one(1)), ref_f[1])          # other context: miss
    assert np.array_equal(net.predict_pel(*one(2)), ref_p[2])
    assert net.cache_stats() == (2, 3)
    net.set_option("canonical_order", 1)                            # any option change drops the cache
    net.predict_pel(*one(0))
    assert net.cache_stats()[1] == 4


@pytest.mark.parametrize("w,is_fc,n", [(4, True, 300), (8, True, 257), (16, False, 40), (64, False, 3)])
@pytest.mark.parametrize("pel_bytes", [4, 1])
def test_block_cost_bit_exact_and_fused_entry(pnn, oracle, w, is_fc, n, pel_bytes):
    """(f4) HM's first-pass distortion of the PNN candidate on the device: HADs and SAD equal the oracle (itself pinned on
    the reference's TComRdCost) bit for bit, from int32 and uint8 pictures, and the fused predict+cost entry returns the
    costs of its own predictions."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    plane = util.make_plane(320, 448, seed=9, pad=16)
    org = np.clip(plane + np.random.RandomState(3).randint(-6, 7, plane.shape), 0, 255).astype(np.int32)
    xs, ys, flags = util.make_tbs(320, 448, w, n, seed=10)
    params = util.make_params(w, is_fc, 51, out_gain=util.out_gain(w, is_fc))
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    tbs = _device_tbs(pnn, xs, ys, flags, plane.shape[1], w)
    dt = np.int32 if pel_bytes == 4 else np.uint8
    d_plane = torch.from_numpy(plane.astype(dt)).cuda()
    d_org = torch.from_numpy(org.astype(dt)).cuda()
    d_tbs = torch.from_numpy(tbs).cuda()
    d_dst = torch.empty((n, w, w), dtype=torch.int32, device="cuda")
    d_cost = torch.empty(n, dtype=torch.int32, device="cuda")
    for had in (1, 0):
        assert L.pnn_predict_tbs_cost_device(net.ctx, w, d_plane.data_ptr(), d_org.data_ptr(), pel_bytes, d_tbs.data_ptr(), n, had,
                                             d_cost.data_ptr(), d_dst.data_ptr(), None) == 0, L.pnn_last_error(net.ctx)
        torch.cuda.synchronize()
        pred = d_dst.cpu().numpy()
        want = oracle.block_costs(org, xs, ys, w, pred, had)
        assert np.array_equal(d_cost.cpu().numpy().view(np.uint32), want)
        d_cost2 = torch.full((n,), -1, dtype=torch.int32, device="cuda")    # without the predictions, and the stand-alone entry
        assert L.pnn_predict_tbs_cost_device(net.ctx, w, d_plane.data_ptr(), d_org.data_ptr(), pel_bytes, d_tbs.data_ptr(), n, had,
                                             d_cost2.data_ptr(), None, None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_cost2.cpu().numpy().view(np.uint32), want)
        rnd = torch.from_numpy(np.random.RandomState(4).randint(0, 256, (n, w, w)).astype(np.int32)).cuda()
        assert L.pnn_block_cost_device(net.ctx, w, d_org.data_ptr(), pel_bytes, d_tbs.data_ptr(), n, rnd.data_ptr(), had, d_cost.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d_cost.cpu().numpy().view(np.uint32), oracle.block_costs(org, xs, ys, w, rnd.cpu().numpy(), had))


def test_batching_service_on_gpu(pnn, tmp_path):
    """include/pnn_service.h end to end: client threads send single-block requests over the socket, the server answers
    them from batched pnn_predict_pel calls on one context; under canonical_order every answer equals the direct
    single-block call bit for bit."""
    import threading
    from context_adaptive_neural_network_based_prediction_amd import service
    w, n = 8, 48
    params = util.make_params(w, True, 61, out_gain=util.out_gain(w, True))
    above, left = util.make_contexts(w, n, 62)
    ctxs = util.flatten_fc(above, left)
    net = pnn.PredictionNeuralNetwork(1, w, True, params=params)
    net.set_option("canonical_order", 1)
    want = [net.predict_pel(ctxs[i:i + 1])[0].copy() for i in range(n)]
    sock = str(tmp_path / "pnn.sock")
    srv = service.serve_in_thread(sock, ctx=net.ctx, max_batch=16, window_us=2000)
    bad = []

    def client(k):
        c = service.Client(sock)
        for i in range(k, n, 4):
            if not np.array_equal(c.predict_pel(w, ctxs[i]), want[i]):
                bad.append(i)
        c.close()

    threads = [threading.Thread(target=client, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stats = srv.stop()
    assert not bad, bad
    assert stats["requests"] == n and stats["largest_batch"] >= 2


def _natural_contexts(w, n, seed):
    """Contexts cut from a smooth synthetic picture with HM-style availability (what the trained nets were made for)."""
    from oracle import pnn_oracle as O
    plane = util.make_plane(192, 256, seed=seed)
    xs, ys, flags = util.make_tbs(192, 256, w, n, seed=seed + 1, partial_fraction=0.4)
    ab = np.zeros((n, w, 3 * w), np.float32)
    lf = np.zeros((n, 2 * w, w), np.float32)
    for i in range(n):
        _, ab[i], lf[i] = O.extract_context(plane, int(xs[i]), int(ys[i]), w, flags[i], util.MEAN)
    return ab, lf


@pytest.mark.parametrize("w", [4, 8])
def test_trained_checkpoints_through_the_split_kernels(pnn, oracle, precision, w):
    """The reference's two trained models on the 3 x f16 split-product kernels (ring / register-staged / LDS-resident-image),
    1024 natural-like contexts in one pass and the committed 8 contexts (small-M kernels: the same summation order).  Float
    predictions within FLOAT_ATOL of the oracle, Pel within one LSB."""
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    path = os.path.join(GOLD, "conv%d_single.pnnw" % w)
    flat = wts.load_pnnw(path)[0]
    n = 1024
    ab, lf = _natural_contexts(w, n, 300 + w)
    net = pnn.PredictionNeuralNetwork(n, w, False, path_to_model=path)
    want = oracle.conv_forward(flat, w, ab, lf)
    got = net.predict(ab, lf)
    assert net.last_call_stats()["gemm_launches"] >= 3
    np.testing.assert_allclose(got[..., 0], want, rtol=0, atol=FLOAT_ATOL)
    _check_pel(net.predict_pel(ab, lf), oracle.epilogue(want, util.MEAN))
    assert want.max() - want.min() > 60                              # real pictures, not a flat answer
    g = np.load(os.path.join(GOLD, "nets.npz"))
    net2 = pnn.PredictionNeuralNetwork(8, w, False, path_to_model=path)
    got8 = net2.predict(g["real%d_above" % w], g["real%d_left" % w])
    np.testing.assert_allclose(got8[..., 0], g["real%d_out" % w], rtol=0, atol=FLOAT_ATOL)
    _check_pel(net2.predict_pel(g["real%d_above" % w], g["real%d_left" % w]), oracle.epilogue(g["real%d_out" % w], util.MEAN))
    net2.close()


@pytest.mark.parametrize("w,n", [(16, 200), (16, 2), (8, 300), (32, 40)])
def test_f16_range_guard_on_the_raw_context(pnn, oracle, precision, w, n):
    """The first convolution of a branch splits the RAW context into f16 pairs in registers (FirstConv): a finite input element
    of 1e5 would become hi = inf, lo = -inf and a NaN that the output-side guard's fmaxf drops.  The staging loops check the
    inputs (leaves_f16): host calls repeat the pass on the exact-f32 kernels and still match the oracle, device calls report
    PNN_E_RANGE -- at batch (image kernel with the fused first convolution) and for a handful (conv_cin1 kernels)."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, False, 91, out_gain=util.out_gain(w, False)).copy()
    specs = wts.tensor_specs(w, False)
    sizes = [int(np.prod(sh)) for _, sh, _ in specs]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    above, left = util.make_contexts(w, n, 92)
    above[n // 2, 1, 2] = 1.0e5                                      # one finite, huge element in ONE block's above portion
    left[n - 1, 3, 1] = -2.0e5                                       # ... and one in another block's left portion
    params[offs[-3]:offs[-2]] /= 200.0                               # keep that block's prediction finite-sized
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    net.set_option("canonical_order", 1)
    want = oracle.conv_forward(params, w, above, left)
    got = net.predict(above, left)
    fallbacks = ctypes.c_long()
    assert L.pnn_check_range(net.ctx, None, ctypes.byref(fallbacks)) == 0
    assert fallbacks.value == (1 if precision == "split_f16" else 0)
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[..., 0], want, rtol=2e-6, atol=FLOAT_ATOL * 4)
    ts = [torch.from_numpy(np.ascontiguousarray(a)).cuda()[..., None] for a in (above, left)]
    out = net.predict(*ts)
    rc = L.pnn_check_range(net.ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None)
    if precision == "split_f16":
        assert rc == -6 and b"f16 range" in L.pnn_last_error(net.ctx)
    else:
        assert rc == 0
        np.testing.assert_allclose(out.cpu().numpy()[..., 0], want, rtol=2e-6, atol=FLOAT_ATOL * 4)
    net.close()


@pytest.mark.parametrize("w,is_fc,n,layer", [(8, True, 600, 0), (8, True, 1, 0), (16, False, 90, 0), (16, False, 90, 1), (16, False, 1, 1)])
def test_f16_range_guard(pnn, oracle, precision, w, is_fc, n, layer):
    """Split precision carries activations as f16 pairs: |v| >= 65504 must never turn into a silent NaN -> 255.  First-layer
    weights are scaled until hidden activations pass 1e5 (the last layer is scaled back, so the prediction stays in range):
    host calls must still match the oracle (they repeat the pass on the exact-f32 kernels and count it), device calls
    must report PNN_E_RANGE."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, is_fc, 81, out_gain=util.out_gain(w, is_fc)).copy()
    specs = wts.tensor_specs(w, is_fc)
    sizes = [int(np.prod(sh)) for _, sh, _ in specs]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    gain = 3000.0 if is_fc else 30000.0
    # first layer (or, layer = 1, the second: its outputs leave through the GEMM kernels' own epilogues): weights and biases
    # (LeakyReLU is positively homogeneous)
    params[offs[2 * layer]:offs[2 * layer + 2]] *= gain
    params[offs[-3]:offs[-2]] /= gain                                # last layer's weights undo it
    above, left = util.make_contexts(w, n, 82)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    net.set_option("canonical_order", 1)                             # split kernels at every batch size
    if is_fc:
        x = (util.flatten_fc(above, left),)
        want = oracle.fc_forward(params, w, x[0])
    else:
        x = (above, left)
        want = oracle.conv_forward(params, w, above, left)
    got = net.predict(*x)
    fallbacks = ctypes.c_long()
    assert L.pnn_check_range(net.ctx, None, ctypes.byref(fallbacks)) == 0
    if precision == "split_f16":
        assert fallbacks.value == 1, "hidden activations were meant to leave the f16 range"
    else:
        assert fallbacks.value == 0
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got[..., 0], want, rtol=0, atol=FLOAT_ATOL * 4)   # 1e5-sized intermediates: a little more float noise
    _check_pel(net.predict_pel(*x), oracle.epilogue(want, util.MEAN))
    # device entry point: asynchronous, so the violation is reported, not repaired
    ts = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in x]
    out = net.predict(*[t if is_fc else t[..., None] for t in ts])
    rc = L.pnn_check_range(net.ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None)
    if precision == "split_f16":
        assert rc == -6 and b"f16 range" in L.pnn_last_error(net.ctx)
        assert L.pnn_check_range(net.ctx, None, None) == 0           # reported once
        out = net.predict(*[t if is_fc else t[..., None] for t in ts])
        torch.cuda.synchronize()
        with pytest.raises(_lib.PnnError):                           # ... or by the next call on the context
            net.predict(*[t if is_fc else t[..., None] for t in ts])
        net.set_option("precision", 0)
        out = net.predict(*[t if is_fc else t[..., None] for t in ts])
        assert L.pnn_check_range(net.ctx, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), None) == 0
    else:
        assert rc == 0
    np.testing.assert_allclose(out.cpu().numpy()[..., 0], want, rtol=0, atol=FLOAT_ATOL * 4)


def test_reference_gather_fixtures_through_the_hip_gather(pnn):
    """tests/golden/gather_ref.npz -- outputs of the REFERENCE's own extract_context_portions (the tests.cpp ramp scenarios
    and 40 seeded cases incl. flag patterns with holes) -- straight through pnn_make_tb_desc + pnn_gather_device, from
    int32 and from uint8 planes: bit-exact."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    g = np.load(os.path.join(GOLD, "gather_ref.npz"))
    nets = {}
    for k in range(int(g["n_cases"])):
        w, mean = int(g["c%d_w" % k]), float(g["c%d_mean" % k])
        plane, (x, y), flags = g["c%d_plane" % k], g["c%d_xy" % k], g["c%d_flags" % k]
        key = (w <= 8, mean)
        if key not in nets:
            nets[key] = pnn.PredictionNeuralNetwork(1, 4, True, params=util.make_params(4, True, 1), mean_training=mean)
        net = nets[key]
        units = 2 * w // 4
        tb = (_lib.TbDev * 1)()
        assert L.pnn_make_tb_desc(ctypes.byref(tb[0]), int(y) * plane.shape[1] + int(x), plane.shape[1],
                                  flags.ctypes.data_as(_lib.u8p), int(flags.sum()), units, units) == 0
        d_tbs = torch.from_numpy(np.frombuffer(tb, dtype=np.uint8).copy()).cuda()
        variants = [(plane.astype(np.int32), 4)]
        if plane.max() < 256:
            variants.append((plane.astype(np.uint8), 1))
        for pl, pel_bytes in variants:
            d_plane = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
            d_above = torch.full((w, 3 * w), float("nan"), device="cuda")
            d_left = torch.full((2 * w, w), float("nan"), device="cuda")
            assert L.pnn_gather_device(net.ctx, w, 4, d_plane.data_ptr(), pel_bytes, d_tbs.data_ptr(), 1, d_above.data_ptr(), 3 * w * w,
                                       d_left.data_ptr(), 2 * w * w, None) == 0
            torch.cuda.synchronize()
            assert np.array_equal(d_above.cpu().numpy(), g["c%d_above" % k]), "case %d (w %d, pel_bytes %d): above differs" % (k, w, pel_bytes)
            assert np.array_equal(d_left.cpu().numpy(), g["c%d_left" % k]), "case %d (w %d, pel_bytes %d): left differs" % (k, w, pel_bytes)


@pytest.mark.parametrize("w", [4, 8, 16, 32])
@pytest.mark.parametrize("holes", [False, True])
def test_gather_chroma_units(pnn, oracle, w, holes):
    """Chroma planes of 4:2:0 video reach the PNN with unitWidth = unitHeight = 2 (TEncSearch.cpp:1197-1200):
    availability flags then cover 2-pixel units, 2w/2 of them per side.  HIP gather == oracle (== reference, test_oracle)."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    n, unit = 150, 2
    units = 2 * w // unit
    plane = util.make_plane(200, 280, seed=40 + w, pad=4)
    rng = np.random.RandomState(50 + w)
    xs = 4 * rng.randint((w + 3) // 4, (280 - 2 * w) // 4 + 1, n)
    ys = 4 * rng.randint((w + 3) // 4, (200 - 2 * w) // 4 + 1, n)
    flags = np.ones((n, 2 * units + 1), np.uint8)
    for i in range(n):
        if rng.rand() < 0.7:
            if holes:
                flags[i] = rng.randint(0, 2, 2 * units + 1)
                flags[i, units] = 1
            else:
                kl, ka = rng.randint(0, units // 2 + 1, 2)
                if kl:
                    flags[i, :kl] = 0
                if ka:
                    flags[i, 2 * units + 1 - ka:] = 0
    net = pnn.PredictionNeuralNetwork(n, 4, True, params=util.make_params(4, True, 1))
    arr = (_lib.TbDev * n)()
    for i in range(n):
        assert L.pnn_make_tb_desc(ctypes.byref(arr[i]), int(ys[i]) * plane.shape[1] + int(xs[i]), plane.shape[1],
                                  flags[i].ctypes.data_as(_lib.u8p), int(flags[i].sum()), units, units) == 0
    d_plane = torch.from_numpy(plane).cuda()
    d_tbs = torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy()).cuda()
    d_above = torch.full((n, w, 3 * w), float("nan"), device="cuda")
    d_left = torch.full((n, 2 * w, w), float("nan"), device="cuda")
    assert L.pnn_gather_device(net.ctx, w, unit, d_plane.data_ptr(), 4, d_tbs.data_ptr(), n, d_above.data_ptr(), 3 * w * w,
                               d_left.data_ptr(), 2 * w * w, None) == 0
    torch.cuda.synchronize()
    ga, gl = d_above.cpu().numpy(), d_left.cpu().numpy()
    for i in range(n):
        rc, a, l = oracle.extract_context(plane, int(xs[i]), int(ys[i]), w, flags[i], util.MEAN, unit=unit)
        assert rc == 0
        assert np.array_equal(ga[i], a) and np.array_equal(gl[i], l), "TB %d differs" % i


def _random_size_cases():
    rng = np.random.RandomState(7)
    cases = []
    for w, fc, hi in ((4, True, 1500), (8, True, 1500), (4, False, 900), (8, False, 700), (16, False, 260), (32, False, 70), (64, False, 22)):
        cases += [(w, fc, int(rng.randint(1, hi + 1))) for _ in range(2)]
    # one block either side of the rule that switches between the exact-f32 and the split-precision kernels
    cases += [(8, False, 156), (8, False, 157), (16, False, 70), (16, False, 71), (32, False, 34), (32, False, 35),
              (64, False, 17), (64, False, 18), (8, True, 511), (8, True, 513)]
    return cases


@pytest.mark.parametrize("w,is_fc,n", _random_size_cases())
def test_random_batch_sizes_match_oracle(pnn, oracle, w, is_fc, n):
    """Random (ragged) batch sizes for every net, and the sizes around the kernel-family switch: whatever tile rule a
    size lands on, float predictions stay within FLOAT_ATOL of the oracle and the HM epilogue within one LSB."""
    params = util.make_params(w, is_fc, 100 + w + n, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 200 + n)
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    if is_fc:
        x = util.flatten_fc(above, left)
        got, want, pel = net.predict(x), oracle.fc_forward(params, w, x), net.predict_pel(x)
    else:
        got, want, pel = net.predict(above, left), oracle.conv_forward(params, w, above, left), net.predict_pel(above, left)
    np.testing.assert_allclose(got[..., 0], want, rtol=0, atol=FLOAT_ATOL)
    _check_pel(pel, oracle.epilogue(want, util.MEAN))


@pytest.mark.parametrize("w,n", [(16, 1024), (8, 2048), (32, 96)])
def test_fused_first_convolution_bit_identical(pnn, precision, w, n):
    """Option "fuse_first" (default on): an LDS-resident-image kernel computes its branch's first (Cin = 1) convolution itself
    instead of reading conv_cin1_kernel's output back.  Same arithmetic in the same order, so predictions must not change
    by a bit -- at a batch with several workgroups per CU in flight (the prototype of this fusion was not repeatable
    there until the packed-fp32 erratum was understood), on repeated and on permuted batches."""
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    params = util.make_params(w, False, 91, out_gain=util.out_gain(w, False))
    above, left = util.make_contexts(w, n, 92)
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    net.set_option("autotune", 0)
    net.set_option("fuse_first", 0)
    ref = net.predict(above, left).copy()
    net.set_option("fuse_first", 1)
    for _ in range(5):
        assert np.array_equal(net.predict(above, left), ref)
    perm = np.random.RandomState(0).permutation(n)
    assert np.array_equal(net.predict(above[perm], left[perm]), ref[perm])
    net.set_option("autotune", 1)                                     # whichever configuration the tuner picks
    for _ in range(3):
        assert np.array_equal(net.predict(above, left), ref)


def test_contexts_sharing_the_gpu_stay_repeatable(pnn):
    """Four contexts in four host threads on one GPU (two big FC passes = matrix-core kernels, a conv net in small passes =
    VALU-heavy first layers, a conv net at batch = two streams of its own): every call must reproduce the context's first result bit for bit.  Before the library was built
    without packed-fp32 instructions the conv net showed rare one-pixel errors here (gfx950: an in-place v_pk_fma_f32 can
    read an already-overwritten half while another wave on its SIMD issues MFMAs -- tools/pkfma_probe.hip)."""
    import threading
    out, errs = {}, []
    bar = threading.Barrier(4)

    def worker(name, w, fc, n, seed, reps):
        try:
            params = util.make_params(w, fc, seed, out_gain=util.out_gain(w, fc))
            above, left = util.make_contexts(w, n, seed + 1)
            net = pnn.PredictionNeuralNetwork(n, w, fc, params=params)
            run = (lambda: net.predict(util.flatten_fc(above, left))) if fc else (lambda: net.predict(above, left))
            want = run().copy()
            bar.wait()
            out[name] = sum(not np.array_equal(run(), want) for _ in range(reps))
        except Exception as e:                                        # pragma: no cover
            errs.append((name, repr(e)))
            bar.abort()

    ts = [threading.Thread(target=worker, args=("fc8-a", 8, True, 2048, 5, 150)),
          threading.Thread(target=worker, args=("fc8-b", 8, True, 1536, 7, 150)),
          threading.Thread(target=worker, args=("conv16", 16, False, 64, 9, 150)),
          # round 3: a conv net at batch -- position-major ring tiles, the two branches side by side on two streams (from its
          # third pass on), the last layer inside the image kernel -- beside the others
          threading.Thread(target=worker, args=("conv16-batch", 16, False, 384, 11, 120))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert out == {"fc8-a": 0, "fc8-b": 0, "conv16": 0, "conv16-batch": 0}, out


@pytest.mark.parametrize("w,n", [(16, 384), (32, 96)])
def test_conv_branches_overlap_at_batch(pnn, oracle, precision, w, n):
    """Passes at batch overlap the two branches on two streams as well -- from the third pass of a shape on (the first may
    tune, the second proves that nothing is left to tune).  Same predictions as on one stream, pass after pass, also when a
    smaller batch and single-block calls (their own overlap rules, the shared side buffers) come in between."""
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    params = util.make_params(w, False, 83, out_gain=util.out_gain(w, False))
    above, left = util.make_contexts(w, n, 84)
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    net.set_option("branch_streams", 0)
    want = net.predict(above, left).copy()
    small = net.predict(above[:5], left[:5]).copy()
    net.set_option("branch_streams", 1)
    for rep in range(6):
        assert np.array_equal(net.predict(above, left), want), "pass %d differs" % rep
        if rep == 3:
            assert np.array_equal(net.predict(above[:5], left[:5]), small)
            assert np.array_equal(net.predict(above[:1], left[:1]), small[:1])
    m = min(n, 32)
    np.testing.assert_allclose(want[:m, ..., 0], oracle.conv_forward(params, w, above[:m], left[:m]), rtol=0, atol=FLOAT_ATOL)
    net.close()


@pytest.mark.parametrize("w", [16, 32, 64])
def test_conv_branches_on_two_streams(pnn, oracle, w):
    """Option "branch_streams": small conv passes run the two branches concurrently on two HIP streams (fork / join by
    events, the left branch on its own buffer pair).  The result must not depend on it -- a missing dependency would show
    as a difference or as run-to-run noise -- also when single-block and batched calls alternate on one context (the side
    buffers are sized per pass)."""
    params = util.make_params(w, False, 81, out_gain=util.out_gain(w, False))
    above, left = util.make_contexts(w, 6, 82)
    net = pnn.PredictionNeuralNetwork(6, w, False, params=params)
    net.set_option("branch_streams", 0)
    ref1 = [net.predict(above[i:i + 1], left[i:i + 1]).copy() for i in range(6)]
    ref3 = net.predict(above[:3], left[:3]).copy()
    for mode in (2, 1):
        net.set_option("branch_streams", mode)
        for rep in range(3):
            for i in range(6):
                assert np.array_equal(net.predict(above[i:i + 1], left[i:i + 1]), ref1[i]), (mode, rep, i)
            assert np.array_equal(net.predict(above[:3], left[:3]), ref3), (mode, rep)
    np.testing.assert_allclose(ref1[0][..., 0], oracle.conv_forward(params, w, above[:1], left[:1]), rtol=0, atol=FLOAT_ATOL)


# ---- BASELINE.json sizes: size-independent properties ---------------------------------------------------------
@pytest.mark.parametrize("w,is_fc,n", [(8, True, 4096), (16, False, 1024)])
def test_full_size_properties(pnn, oracle, w, is_fc, n):
    """configs[1] / configs[2] at full size: (1) a prediction does not depend on its batch position or on the batch
    it travels in (chunked vs whole, permuted), (2) the WHOLE batch equals the oracle (FC; a 256-block sample for the conv net,
    whose oracle takes 50 ms per block), (3) duplicated
    inputs give identical outputs, (4) the fused gather+net+epilogue entry equals gather -> net -> epilogue."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, is_fc, 21, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 22)
    above[n // 2] = above[3]
    left[n // 2] = left[3]
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    run = (lambda a, l: net.predict(util.flatten_fc(a, l))) if is_fc else (lambda a, l: net.predict(a, l))
    full = run(above, left)
    perm = np.random.RandomState(0).permutation(n)
    assert np.array_equal(run(above[perm], left[perm]), full[perm])
    net.set_option("max_chunk", 300)                                 # ragged chunks: 300, 300, ..., remainder
    assert np.array_equal(run(above, left), full)                    # one summation order (the default): chunking changes nothing, bit for bit
    assert np.array_equal(run(above[5:6], left[5:6])[0], full[5])    # ... down to a batch of one
    net.set_option("max_chunk", 0)
    assert np.array_equal(full[n // 2], full[3])
    idx = np.arange(n) if is_fc else np.random.RandomState(1).choice(n, 256, replace=False)
    want = oracle.fc_forward(params, w, util.flatten_fc(above[idx], left[idx])) if is_fc else oracle.conv_forward(params, w, above[idx], left[idx])
    np.testing.assert_allclose(full[idx, ..., 0], want, rtol=0, atol=FLOAT_ATOL)
    plane = util.make_plane(544, 960, seed=5, pad=32)
    xs, ys, flags = util.make_tbs(544, 960, w, n, seed=6)
    d_plane = torch.from_numpy(plane).cuda()
    d_tbs = torch.from_numpy(_device_tbs(pnn, xs, ys, flags, plane.shape[1], w)).cuda()
    d_dst = torch.empty((n, w, w), dtype=torch.int32, device="cuda")
    d_f32 = torch.empty((n, w, w), dtype=torch.float32, device="cuda")
    assert L.pnn_predict_tbs_device(net.ctx, w, d_plane.data_ptr(), 4, d_tbs.data_ptr(), n, d_dst.data_ptr(), d_f32.data_ptr(), None) == 0
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    assert got.min() >= 0 and got.max() <= 255 and got.min() == 0 and got.max() == 255
    assert np.array_equal(got, oracle.epilogue(d_f32.cpu().numpy(), util.MEAN))   # epilogue fused == epilogue applied after
    sample = idx if is_fc else idx[:128]
    _check_pel(got[sample], oracle.predict_tbs(params, w, is_fc, plane, xs[sample], ys[sample], flags[sample], util.MEAN))


@pytest.mark.parametrize("w,n", [(16, 1024), (8, 2048), (4, 1500), (32, 256), (64, 64)])
def test_gather_fused_into_the_image_kernel(pnn, oracle, precision, w, n):
    """Big passes of pnn_predict_tbs_device through a conv net: when both branches' first convolutions run inside the image kernel,
    the context gather does too (the kernel reads the picture plane through the TB descriptors): one launch less, and exactly the
    predictions of gather -> net, for int32 and uint8 planes and for descriptors with missing units."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    L = _lib.lib()
    params = util.make_params(w, False, 141, out_gain=util.out_gain(w, False))
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    net.set_option("autotune", 0)
    plane = util.make_plane(544, 960, seed=15, pad=32)
    xs, ys, flags = util.make_tbs(544, 960, w, n, seed=16, partial_fraction=0.5)
    d_tbs = torch.from_numpy(_device_tbs(pnn, xs, ys, flags, plane.shape[1], w)).cuda()
    outs = {}
    for pel_bytes, pl in ((4, plane), (1, plane.astype(np.uint8))):
        d_plane = torch.from_numpy(np.ascontiguousarray(pl)).cuda()
        for fuse in (0, 1):
            net.set_option("fuse_gather", fuse)
            d_dst = torch.full((n, w, w), -1, dtype=torch.int32, device="cuda")
            d_f32 = torch.zeros((n, w, w), dtype=torch.float32, device="cuda")
            assert L.pnn_predict_tbs_device(net.ctx, w, d_plane.data_ptr(), pel_bytes, d_tbs.data_ptr(), n, d_dst.data_ptr(), d_f32.data_ptr(), None) == 0, L.pnn_last_error(net.ctx)
            torch.cuda.synchronize()
            outs[(pel_bytes, fuse)] = (d_dst.cpu().numpy(), d_f32.cpu().numpy(), net.last_call_stats()["launches"])
        assert np.array_equal(outs[(pel_bytes, 1)][0], outs[(pel_bytes, 0)][0]) and np.array_equal(outs[(pel_bytes, 1)][1], outs[(pel_bytes, 0)][1])
    assert np.array_equal(outs[(1, 1)][0], outs[(4, 1)][0])
    net.set_option("max_chunk", max(8, n // 3 + 1))                  # several passes per call: every chunk reads ITS descriptors
    d_dst = torch.full((n, w, w), -1, dtype=torch.int32, device="cuda")
    assert L.pnn_predict_tbs_device(net.ctx, w, d_plane.data_ptr(), 1, d_tbs.data_ptr(), n, d_dst.data_ptr(), None, None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_dst.cpu().numpy(), outs[(4, 1)][0])
    net.set_option("max_chunk", 0)
    if w in (16, 8):
        assert outs[(4, 1)][2] == outs[(4, 0)][2] - 1, "the gather launch is gone"
    m = min(n, 64)
    _check_pel(outs[(4, 1)][0][:m], oracle.predict_tbs(params, w, False, plane, xs[:m], ys[:m], flags[:m], util.MEAN))
    net.close()


@pytest.mark.parametrize("w,is_fc,n", [(4, True, 1500), (8, True, 600), (16, False, 300), (32, False, 60)])
def test_completion_flag_of_small_host_calls(pnn, precision, w, is_fc, n):
    """Option "flag_wait": a small host call returns when its last kernel has raised the completion flag behind its results in
    pinned host memory.  A call that returned early would hand back the PREVIOUS call's block: n different blocks one after
    the other, then in handfuls (several workgroups count themselves in) and in chunks (only the last chunk's last kernel
    may signal), must equal what the same context returns when it waits for the stream."""
    params = util.make_params(w, is_fc, 91, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, 92)
    rows = util.flatten_fc(above, left) if is_fc else None
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    net.set_option("canonical_order", 1)
    call = (lambda a, b: net.predict_pel(rows[a:b])) if is_fc else (lambda a, b: net.predict_pel(above[a:b], left[a:b]))
    net.set_option("flag_wait", 0)
    want = np.concatenate([call(i, min(i + 50, n)) for i in range(0, n, 50)])
    net.set_option("flag_wait", 1)
    for i in range(n):
        assert np.array_equal(call(i, i + 1)[0], want[i]), "single-block call %d" % i
    for i in range(0, n - 7, 7):
        assert np.array_equal(call(i, i + 7), want[i:i + 7]), "handful at %d" % i
    net.set_option("max_chunk", 2)
    for i in range(0, min(n, 60) - 5, 5):
        assert np.array_equal(call(i, i + 5), want[i:i + 5]), "chunked call at %d" % i
    net.close()


@pytest.mark.parametrize("w,n", [(16, 1024), (16, 333), (8, 2048), (32, 128)])
def test_last_layer_fused_into_the_image_kernel(pnn, oracle, precision, w, n):
    """Option "fuse_tail": the image kernel of the last 64-channel layer applies the net's last layer (64 -> 1 transposed
    convolution) to its tile in registers.  Same MFMA chain on the same values, same col2im order: float predictions and HM
    blocks equal the two-launch path bit for bit, one launch less where it applies."""
    import torch
    from context_adaptive_neural_network_based_prediction_amd import _lib
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    L = _lib.lib()
    params = util.make_params(w, False, 95, out_gain=util.out_gain(w, False))
    above, left = util.make_contexts(w, n, 96)
    net = pnn.PredictionNeuralNetwork(n, w, False, params=params)
    net.set_option("autotune", 0)
    net.set_option("branch_streams", 0)
    d_a, d_l = torch.from_numpy(above).cuda(), torch.from_numpy(left).cuda()
    res = {}
    for fuse in (0, 1):
        net.set_option("fuse_tail", fuse)
        d_out = torch.zeros((n, w, w), dtype=torch.float32, device="cuda")
        assert L.pnn_predict_conv_device(net.ctx, w, d_a.data_ptr(), d_l.data_ptr(), n, d_out.data_ptr(), None) == 0, L.pnn_last_error(net.ctx)
        torch.cuda.synchronize()
        res[fuse] = (d_out.cpu().numpy(), net.predict_pel(above, left), net.last_call_stats()["launches"])
    assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1])
    if w == 16 and n == 1024:
        assert res[1][2] == res[0][2] - 1, "the last layer's launch is gone"
    m = min(n, 32)
    np.testing.assert_allclose(res[1][0][:m], oracle.conv_forward(params, w, above[:m], left[:m]), rtol=0, atol=FLOAT_ATOL)
    net.close()


def test_chunked_host_calls_carry_their_own_rows(pnn, oracle, precision):
    """with max_chunk below the batch size, every chunk of a host call through an FC net must be predicted
    from ITS rows -- the inline copy of small inputs (first kernel's argument block) used to stay on the first chunk's."""
    for w, chunk, n in ((4, 1, 3), (4, 2, 7), (8, 1, 3)):
        params = util.make_params(w, True, 31, out_gain=util.out_gain(w, True))
        above, left = util.make_contexts(w, n, 32 + w)
        ctx = util.flatten_fc(above, left)
        net = pnn.PredictionNeuralNetwork(n, w, True, params=params)
        whole = net.predict(ctx)
        net.set_option("max_chunk", chunk)
        got = net.predict(ctx)
        assert np.array_equal(got, whole)
        np.testing.assert_allclose(got[..., 0], oracle.fc_forward(params, w, ctx), rtol=0, atol=FLOAT_ATOL)
        assert not np.array_equal(got[0], got[1])
        net.close()


@pytest.mark.parametrize("w,is_fc,n", [(8, True, 6), (16, False, 5), (64, False, 3)])
def test_range_fallback_touches_only_the_overflowing_block(pnn, oracle, precision, w, is_fc, n):
    """one block of a host batch leaves the f16 range -- only THAT block is recomputed on the exact-f32
    kernels; the others keep, bit for bit, what they get alone or in any other batch (the batching service's promise)."""
    if precision != "split_f16":
        pytest.skip("split-precision kernels only")
    from context_adaptive_neural_network_based_prediction_amd import _lib
    L = _lib.lib()
    params = util.make_params(w, is_fc, 91, out_gain=util.out_gain(w, is_fc)).copy()
    specs = wts.tensor_specs(w, is_fc)
    offs = np.concatenate([[0], np.cumsum([int(np.prod(sh)) for _, sh, _ in specs])])
    gain = 300.0 if is_fc else 1000.0                              # ordinary contexts stay below 65504 (max |hidden| ~17 k / 22 k) ...
    params[offs[0]:offs[2]] *= gain
    params[offs[-3]:offs[-2]] /= gain
    above, left = util.make_contexts(w, n, 92, masked_fraction=0.0)
    bad = n // 2
    above[bad] *= 40.0                                              # ... this one does not
    left[bad] *= 40.0
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    run = (lambda a, l: net.predict(util.flatten_fc(a, l))) if is_fc else (lambda a, l: net.predict(a, l))
    fb = ctypes.c_long()
    alone = []
    for i in range(n):
        alone.append(run(above[i:i + 1], left[i:i + 1])[0])
        assert L.pnn_check_range(net.ctx, None, ctypes.byref(fb)) == 0
        assert fb.value == (1 if i >= bad else 0), "only block %d was meant to leave the f16 range" % bad
    batch = run(above, left)
    assert L.pnn_check_range(net.ctx, None, ctypes.byref(fb)) == 0 and fb.value == 2
    assert np.isfinite(batch).all()
    for i in range(n):
        assert np.array_equal(batch[i], alone[i]), "block %d changed because block %d overflowed" % (i, bad)
    want = oracle.fc_forward(params, w, util.flatten_fc(above, left)) if is_fc else oracle.conv_forward(params, w, above, left)
    np.testing.assert_allclose(batch[..., 0], want, rtol=0, atol=FLOAT_ATOL * 40)
    # non-finite inputs are refused, non-finite parameters too (the guard's max would drop a NaN)
    a2 = above.copy()
    a2[0, 0, 0] = np.nan
    with pytest.raises(_lib.PnnError):
        run(a2, left)
    p2 = params.copy()
    p2[5] = np.inf
    with pytest.raises(_lib.PnnError):
        pnn.PredictionNeuralNetwork(1, w, is_fc, params=p2)
    net.close()


@pytest.mark.parametrize("w,is_fc,big", [(4, True, 1500), (8, True, 2048), (4, False, 700), (8, False, 600), (16, False, 200), (32, False, 40),
                                         (64, False, 5)])
def test_one_summation_order_at_every_batch_size(pnn, precision, w, is_fc, big):
    """canonical_order: a block's float prediction is the same bit pattern whether it is predicted alone (tapgemm_small_kernel,
    one wave per 32 x 32 tile, K-segment output layer), in a handful, in a mid-size pass or in a big batch (ring / sp /
    convimg kernels, fused output layer) -- what an encoder behind the batching service and a stand-alone decoder need.
    On the exact-f32 arithmetic too: every tile of tapgemm_f32_kernel sums in one order, the FC output layer is summed in the
    fused kernel's K segments at every batch size (fc_out_f32_kernel below 1024 blocks)."""
    params = util.make_params(w, is_fc, 111, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, big, 112)
    net = pnn.PredictionNeuralNetwork(big, w, is_fc, params=params)
    net.set_option("canonical_order", 1)
    run = (lambda a, l: net.predict(util.flatten_fc(a, l))) if is_fc else (lambda a, l: net.predict(a, l))
    full = run(above, left)
    net.set_option("small", 0)                                       # the big-tile kernels alone
    assert np.array_equal(run(above, left), full)
    few = run(above[:3], left[:3])
    net.set_option("small", 1)
    assert np.array_equal(few, full[:3])
    net.set_option("pair", 0)                                        # conv nets: the two branches as separate launches
    assert np.array_equal(run(above[:3], left[:3]), full[:3])
    assert np.array_equal(run(above[4:5], left[4:5])[0], full[4])
    net.set_option("pair", 1)
    for n in (1, 2, 3, 17, 33, min(160, big)):
        lo = big - n
        got = run(above[lo:], left[lo:])
        assert np.array_equal(got, full[lo:]), "a batch of %d differs from the same blocks inside a batch of %d" % (n, big)
    assert np.array_equal(net.predict_pel(*((util.flatten_fc(above[2:3], left[2:3]),) if is_fc else (above[2:3], left[2:3])))[0],
                          net.predict_pel(*((util.flatten_fc(above, left),) if is_fc else (above, left)))[2])


@pytest.mark.parametrize("w,is_fc,big", [(4, True, 2600), (8, True, 2300), (4, False, 1500), (8, False, 1300), (16, False, 520), (32, False, 150), (64, False, 36)])
def test_random_batch_sizes_keep_the_bits(pnn, oracle, precision, w, is_fc, big):
    """A seeded walk over batch sizes and offsets -- ragged row tiles, ragged position-major block groups, the FC output layer fused
    (>= 1024 blocks) and from stored activations, chunked passes (max_chunk) -- on both arithmetics: every sub-batch must reproduce,
    bit for bit, the rows it has inside the big batch, and the big batch must match the oracle."""
    params = util.make_params(w, is_fc, 211, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, big, 212)
    net = pnn.PredictionNeuralNetwork(big, w, is_fc, params=params)
    net.set_option("canonical_order", 1)
    run = (lambda a, l: net.predict(util.flatten_fc(a, l))) if is_fc else (lambda a, l: net.predict(a, l))
    full = run(above, left)
    m = min(big, 64 if w <= 16 else 12)
    ref = oracle.fc_forward(params, w, util.flatten_fc(above[:m], left[:m])) if is_fc else oracle.conv_forward(params, w, above[:m], left[:m])
    np.testing.assert_allclose(full[:m, ..., 0], ref, rtol=0, atol=FLOAT_ATOL)
    rng = np.random.RandomState(213 + w)
    sizes = sorted(set([1, 2, 31, 32, 33, 127, 128, 129, big - 1] + [int(x) for x in rng.randint(1, big, 8)]))
    for n in [x for x in sizes if 0 < x <= big]:
        lo = int(rng.randint(0, big - n + 1))
        got = run(above[lo:lo + n], left[lo:lo + n])
        assert np.array_equal(got, full[lo:lo + n]), "%d blocks at offset %d differ from the same rows of the batch of %d" % (n, lo, big)
    net.set_option("max_chunk", max(1, big // 3 + 1))                # three passes per call
    assert np.array_equal(run(above, left), full)
    net.close()


def test_tf_compat_session_run(pnn, oracle, tmp_path):
    """The TensorFlow-look-alike session API of include/pnn_tf_compat.h (what HM's C++ calls) against the oracle."""
    import subprocess
    from context_adaptive_neural_network_based_prediction_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pf = util.make_params(8, True, 41, out_gain=util.out_gain(8, True))
    pc = util.make_params(16, False, 42, out_gain=util.out_gain(16, False))
    wts.save_pnnw(str(tmp_path / "fc8.pnnw"), pf, 8, True)
    wts.save_pnnw(str(tmp_path / "conv16.pnnw"), pc, 16, False)
    exe = str(tmp_path / "hm_sample")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++11", "-I" + os.path.join(root, "include", "tf_compat"), "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "hm_callsite_sample.cpp"), "-o", exe, "-L" + libdir, "-lpnn_hip",
                           "-Wl,-rpath," + libdir])
    out = subprocess.check_output([exe, str(tmp_path / "fc8.pnnw"), "8", str(tmp_path / "conv16.pnnw"), "16"]).decode()
    a, b = out.split("----\n")
    got_fc = np.array([float(x) for x in a.split()], np.float32).reshape(8, 8)
    got_cv = np.array([float(x) for x in b.split()], np.float32).reshape(16, 16)
    mean = np.float32(117.8952234192841)
    ctx = (np.arange(320) * 37 % 256).astype(np.float32) - mean
    np.testing.assert_allclose(got_fc, oracle.fc_forward(pf, 8, ctx[None])[0], rtol=0, atol=FLOAT_ATOL)
    ab = ((np.arange(768) * 37 % 256).astype(np.float32) - mean).reshape(1, 16, 48)
    lf = (((np.arange(512) * 53 + 11) % 256).astype(np.float32) - mean).reshape(1, 32, 16)
    np.testing.assert_allclose(got_cv, oracle.conv_forward(pc, 16, ab, lf)[0], rtol=0, atol=FLOAT_ATOL)
    # Session::Create loads the weights in a background thread (PNN_ASYNC_LOAD, default) that the first Run joins: the same
    # output with the load inside Create, and a model file that is cut off behind its header passes Create (the header is
    # read there) but fails at the first Run with a message -- no crash, no hang
    env = dict(os.environ, PNN_ASYNC_LOAD="0")
    assert subprocess.check_output([exe, str(tmp_path / "fc8.pnnw"), "8", str(tmp_path / "conv16.pnnw"), "16"], env=env).decode() == out
    blob = open(str(tmp_path / "fc8.pnnw"), "rb").read()
    open(str(tmp_path / "cut.pnnw"), "wb").write(blob[:len(blob) // 3])
    for mode in ("1", "0"):
        r = subprocess.run([exe, str(tmp_path / "cut.pnnw"), "8", str(tmp_path / "conv16.pnnw"), "16"], env=dict(os.environ, PNN_ASYNC_LOAD=mode),
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "cut.pnnw" in r.stderr, (mode, r.returncode, r.stderr)


def test_python_evaluator_real_weights(pnn, oracle):
    """evaluation.predict_mask (the PNN half of comparing_pnn_ipfcns_hevc_best_mode.py:162-322) with the reference's
    trained conv-8 model on a smooth synthetic image: predictions equal the oracle's, PSNRs follow tools.compute_psnr,
    and a trained model predicts a smooth image far better than chance."""
    from context_adaptive_neural_network_based_prediction_amd import evaluation
    w = 8
    net = pnn.PredictionNeuralNetwork(4, w, False, path_to_model=os.path.join(GOLD, "conv%d_single.pnnw" % w))
    yy, xx = np.mgrid[0:64, 0:96]
    img = np.clip(90 + 0.9 * xx + 0.5 * yy + 12 * np.sin(xx / 7.0), 0, 255).astype(np.uint8)[None, :, :, None]
    rows = np.array([0, 8, 24, 40], dtype=np.int32)
    cols = np.array([4, 32, 60, 72], dtype=np.int32)
    res = evaluation.predict_mask(img, w, rows, cols, net, 2, util.MEAN, (0, 0))
    assert res['predictions_pnn_uint8'].shape == (4, w, w, 1) and res['psnrs_pnn'].shape == (4,)
    flat, _, _ = wts.load_pnnw(os.path.join(GOLD, "conv%d_single.pnnw" % w))
    for i, (r, c) in enumerate(zip(rows, cols)):
        rc, a, l = oracle.extract_context_u8_rect(img[0, :, :, 0], w, int(r), int(c), util.MEAN, 0, 0)
        want = oracle.conv_forward(flat, w, a[None], l[None])[0] + np.float32(util.MEAN)
        got = res['predictions_pnn_uint8'][i, :, :, 0].astype(int)
        assert np.abs(got - np.round(np.clip(want, 0, 255))).max() <= 1
        tgt = img[0, r + w:r + 2 * w, c + w:c + 2 * w, 0]
        assert np.array_equal(res['targets_uint8'][i, :, :, 0], tgt)
        assert abs(res['psnrs_pnn'][i] - evaluation.compute_psnr(tgt, res['predictions_pnn_uint8'][i, :, :, 0])) < 1e-9
    assert res['mean_psnr_pnn'] > 25.0
    with pytest.raises(TypeError):
        evaluation.compute_psnr(img.astype(np.float32), img)


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_on_one_gpu(precision, tmp_path, launcher):
    """bench.py's N > 1 path end to end on the hardware that is there, with PNN_BENCH_SHARE_GPU=1 (both ranks on device 0,
    joined over gloo -- RCCL refuses two ranks on one device): rank / world plumbing, per-rank workloads, barriers, staggered
    autotune, the max-over-ranks clock, ONE JSON line from rank 0 whose value counts both ranks' blocks.
    `self`: plain `python bench.py --gpus 2` -- the parent starts its own ranks as a child process (what the driver's SCALE
    step may run); `torchrun`: the launcher form of the contract.  A plumbing check, not a measurement."""
    import json
    import subprocess
    import sys
    if precision != "split_f16":
        pytest.skip("once is enough")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PNN_BENCH_SHARE_GPU="1", PNN_AUTOTUNE="0")
    for k in ("PNN_PRECISION", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--batch", "1024", "--no-sustained"]
    head = [sys.executable] if launcher == "self" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                         "--master-addr", "127.0.0.1", "--master-port", "29611"]
    r = subprocess.run(head + tail, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 10
    assert d["config"]["batch_per_gpu"] == 1024
    assert abs(d["value"] - 2 * 1024 * 10 / (d["ms_per_step"] * 1e-3 * 10)) < 2e-5 * d["value"]   # whole-job blocks over the slowest rank's time (6 significant digits each)
    assert d["cpu_baseline"] is None and "fast_arithmetic" not in d and "per_width" not in d   # the extras are N = 1 only
    assert d["dtype"] == "f32" and len(lines[0]) < 4096
    assert d["rccl_ranks_seen"] == {"backend": "gloo", "world_size": 2, "devices": 1}          # share mode: both ranks on device 0, joined over gloo


@pytest.mark.parametrize("w,is_fc", [(4, True), (8, True), (8, False), (16, False), (32, False)])
def test_known_answer_behaviours_on_gpu(pnn, oracle, w, is_fc):
    """SURVEY 8(c) behaviours on the HIP path: an all-zero (= flat training-mean) context through a net with zero biases gives
    exactly zero -- the "grey square" of test_pnn.py:456-459, Pel = round(mean) = 118 after the HM epilogue -- alone and in
    any batch position; scaling the context by a positive factor scales the prediction of a bias-free net (LeakyReLU is
    positively homogeneous) -- a cheap check that no stage adds a stray constant."""
    params = wts.init_params(w, is_fc, seed=77, bias_std=0.0)        # the reference's initialisation: zero biases
    n = 5
    above, left = util.make_contexts(w, n, 78, masked_fraction=0.0)
    above[2] = 0.0
    left[2] = 0.0
    net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
    run = (lambda a, l: net.predict(util.flatten_fc(a, l))) if is_fc else (lambda a, l: net.predict(a, l))
    out = run(above, left)
    assert np.all(out[2] == 0.0)
    assert np.all(run(above[2:3], left[2:3]) == 0.0)
    pel = net.predict_pel(*((util.flatten_fc(above, left),) if is_fc else (above, left)))
    assert np.all(pel[2] == 118)
    half = run(0.5 * above, 0.5 * left)
    np.testing.assert_allclose(half, 0.5 * out, rtol=0, atol=FLOAT_ATOL)


_TF_OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_outputs.npz")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(_TF_OUT), reason="tests/golden/tf_outputs.npz absent: made with TensorFlow 1.x by tools/tf_goldens.py")
def test_hip_matches_tensorflow_outputs(pnn, precision):
    """north_star's bar against the reference ITSELF: the HIP path within 1 LSB per pixel (uint8, after HM's epilogue) of what the
    reference's TensorFlow graphs predict for the seeded weights and contexts of nets.npz -- both arithmetics, every architecture."""
    from oracle import pnn_oracle as O                                # the epilogue only (TComPrediction.cpp:623-635); TF supplies the predictions
    tf_out = np.load(_TF_OUT)
    g = np.load(os.path.join(os.path.dirname(_TF_OUT), "nets.npz"))
    for is_fc, w in [(True, 4), (True, 8), (False, 4), (False, 8), (False, 16), (False, 32), (False, 64)]:
        tag = "%s%d" % ("fc" if is_fc else "conv", w)
        seed, n = int(g[tag + "_seed"]), int(g[tag + "_n"])
        params = util.make_params(w, is_fc, seed, out_gain=util.out_gain(w, is_fc))
        above, left = util.make_contexts(w, n, seed + 1)
        net = pnn.PredictionNeuralNetwork(n, w, is_fc, params=params)
        got = net.predict_pel(*((util.flatten_fc(above, left),) if is_fc else (above, left)))
        want = O.epilogue(tf_out[tag + "_out"], util.MEAN)
        assert np.abs(got.astype(np.int64) - want).max() <= 1, tag
        net.close()
