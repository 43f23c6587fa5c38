"""Evidence, on natural content, that the oracle (and the HIP path) computes what the authors trained.

TensorFlow cannot be run here and the reference checkout holds no output vector of its networks, so the op
semantics of the restated graphs -- TF's SAME padding rule, the crop of conv2d_transpose, the order of the channel-wise merger,
the LeakyReLU slope -- are pinned on what IS there: the two trained checkpoints (convolutional 4x4 / 8x8, pnn/results/...,
converted to tests/golden/conv{4,8}_single.pnnw) and the natural pictures of the checkout (tests/golden/natural_luma.npz,
tests/golden/make_natural.py).  A network trained under semantics S predicts natural blocks well when it is EVALUATED under S
and worse under anything else: with >= 2000 natural contexts per width the restated semantics must beat a DC predictor by a stated
margin, and every single perturbation of one restated rule must lose prediction PSNR (tools/tools.py:364-401) -- most of them
several dB.  The knobbed forward pass below is an independent PyTorch formulation; with all knobs at their restated values it
must equal the oracle, and the HIP path must equal the oracle on the same contexts (GPU test).
"""
import os

import numpy as np
import pytest

from context_adaptive_neural_network_based_prediction_amd import weights as wts
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
N_CONTEXTS = 2500
# measured with `python -m tests.test_natural` (2500 contexts per width; table in DESIGN.md section 2): restated semantics 24.95 dB
# (4x4) / 23.21 dB (8x8) against 21.69 / 20.34 for the DC predictor; the mildest perturbation (LeakyReLU -> ReLU on the 8x8 net)
# costs 0.87 dB, a one-pixel padding or crop error 3.5-12 dB
MARGIN_OVER_DC_DB = {4: 2.5, 8: 2.2}
MIN_DROP_DB = 0.5                                      # every perturbation must cost at least this much


def natural_contexts(w, n, seed=5):
    """n (above [w, 3w], left [2w, w], target [w, w]) triples of uint8 natural luminance, all context available, blocks on the
    w-grid of the five fixture pictures (the layout of extraction_context.cpp:3-208 / sets/common.py:466-473)."""
    if not os.path.exists(os.path.join(GOLD, "natural_luma.npz")):
        pytest.skip("tests/golden/natural_luma.npz is generated from the reference checkout by __graft_entry__.build() (tests/golden/make_natural.py)")
    pics = np.load(os.path.join(GOLD, "natural_luma.npz"))
    rng = np.random.RandomState(seed)
    names = sorted(pics.files)
    above, left, target = [], [], []
    for k in range(n):
        img = pics[names[k % len(names)]]
        H, W = img.shape
        y = w * rng.randint(1, (H - 2 * w) // w + 1)
        x = w * rng.randint(1, (W - 2 * w) // w + 1)
        above.append(img[y - w:y, x - w:x + 2 * w])
        left.append(img[y:y + 2 * w, x - w:x])
        target.append(img[y:y + w, x:x + w])
    return np.stack(above), np.stack(left), np.stack(target)


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10.0 * np.log10(255.0 ** 2 / (mse + 1e-6))      # tools/tools.py:364-401


def epilogue(pred):
    return np.floor(np.clip(pred.astype(np.float32) + np.float32(util.MEAN), 0.0, 255.0) + 0.5)   # TComPrediction.cpp:623-635


def knobbed_conv_forward(flat, w, above, left, pad_before_s2=1, tconv_crop=1, merger="restated", slope=0.1, flip=False):
    """The convolutional PNN (pnn/components.py:182-261) in PyTorch, with ONE rule per knob:
    pad_before_s2   rows / columns of zeros BEFORE the map of a stride-2, 5x5 SAME convolution (TF: pad_total // 2 = 1, 2 after)
    tconv_crop      first kept row / column of the full transposed convolution (TF conv2d_transpose SAME: 1)
    merger          "restated": per channel [above 4x12 row-major | left 8x4 row-major] -> 16 -> 4x4 row-major (tfutils.py:42-73);
                    "left_first", "above_colmajor", "out_colmajor": one ordering changed
    slope           LeakyReLU slope (tfutils.py:192: 0.1)
    flip            True: true convolution (kernels flipped) instead of TF's cross-correlation"""
    import torch
    import torch.nn.functional as F
    t = wts.split_params(np.asarray(flat, np.float32), w, False)
    st = wts.STRIDES_BRANCH[w]
    leaky = lambda x: torch.maximum(slope * x, x)

    def conv(x, W, b, s):
        k = W.shape[0]
        tot = s + 1                                    # (out - 1) s + k - in for in = out * s, k = 2 s + 1
        pb = pad_before_s2 if s == 2 else tot // 2
        x = F.pad(x, (pb, tot - pb, pb, tot - pb))
        Wt = torch.from_numpy(np.ascontiguousarray(W.transpose(3, 2, 0, 1)))
        if flip:
            Wt = torch.flip(Wt, (2, 3))
        return F.conv2d(x, Wt, torch.from_numpy(b), stride=s)

    def tconv(x, W, b, s):
        H, Wd = x.shape[2], x.shape[3]
        Wt = torch.from_numpy(np.ascontiguousarray(W.transpose(3, 2, 0, 1)))
        if flip:
            Wt = torch.flip(Wt, (2, 3))
        y = F.conv_transpose2d(x, Wt, None, stride=s)
        y = y[:, :, tconv_crop:tconv_crop + H * s, tconv_crop:tconv_crop + Wd * s]
        return y + torch.from_numpy(b).view(1, -1, 1, 1)

    feats = []
    for name, inp, shape in (("branch_above", above, (w, 3 * w)), ("branch_left", left, (2 * w, w))):
        x = torch.from_numpy(np.asarray(inp, np.float32).reshape(-1, 1, *shape))
        for i, s in enumerate(st):
            p = "convolutional/%s/convolution_%d/" % (name, i)
            x = leaky(conv(x, t[p + "weights"], t[p + "biases"], s))
        feats.append(x)
    a, l = feats
    n, c = a.shape[0], a.shape[1]
    av = a.transpose(2, 3).reshape(n, c, 48) if merger == "above_colmajor" else a.reshape(n, c, 48)
    lv = l.reshape(n, c, 32)
    v = torch.cat([lv, av], dim=2) if merger == "left_first" else torch.cat([av, lv], dim=2)
    m = "convolutional/merger/"
    Wm = torch.from_numpy(t[m + "channelwise_fully_connected_merger/weights"])
    bm = torch.from_numpy(t[m + "channelwise_fully_connected_merger/biases"])
    o = leaky(torch.einsum("ncp,cpj->ncj", v, Wm) + bm.unsqueeze(0)).reshape(n, c, 4, 4)
    x = o.transpose(2, 3) if merger == "out_colmajor" else o
    rev = st[::-1]
    for i, s in enumerate(rev):
        p = m + "transpose_convolution_%d/" % i
        x = tconv(x, t[p + "weights"], t[p + "biases"], s)
        if i != len(rev) - 1:
            x = leaky(x)
    return x.reshape(n, w, w).numpy()


PERTURBATIONS = [("pad_before_s2", 2), ("pad_before_s2", 0), ("tconv_crop", 0), ("tconv_crop", 2), ("merger", "left_first"), ("merger", "above_colmajor"),
                 ("merger", "out_colmajor"), ("slope", 0.2), ("slope", 0.0), ("slope", 0.3), ("flip", True)]


def evaluate(w, n=N_CONTEXTS):
    flat, _, _ = wts.load_pnnw(os.path.join(GOLD, "conv%d_single.pnnw" % w))
    a8, l8, tgt = natural_contexts(w, n)
    above = a8.astype(np.float32) - np.float32(util.MEAN)
    left = l8.astype(np.float32) - np.float32(util.MEAN)
    res = {"restated": psnr(epilogue(knobbed_conv_forward(flat, w, above, left)), tgt)}
    dc = np.floor((a8.reshape(n, -1).sum(1) + l8.reshape(n, -1).sum(1)) / (5.0 * w * w) + 0.5)
    res["dc"] = psnr(np.broadcast_to(dc[:, None, None], tgt.shape), tgt)
    for knob, value in PERTURBATIONS:
        if knob == "pad_before_s2" and 2 not in wts.STRIDES_BRANCH[w]:
            continue                                   # the 4x4 net has no stride-2 layer
        res["%s=%s" % (knob, value)] = psnr(epilogue(knobbed_conv_forward(flat, w, above, left, **{knob: value})), tgt)
    return res, (flat, above, left, tgt)


@pytest.mark.parametrize("w", [4, 8])
def test_restated_semantics_are_the_trained_ones(oracle, w):
    res, (flat, above, left, tgt) = evaluate(w)
    # all knobs at their restated values: the knobbed formulation IS the oracle's graph
    m = 200
    np.testing.assert_allclose(knobbed_conv_forward(flat, w, above[:m], left[:m]), oracle.conv_forward(flat, w, above[:m], left[:m]), rtol=0, atol=2e-3)
    assert psnr(epilogue(oracle.conv_forward(flat, w, above, left)), tgt) == pytest.approx(res["restated"], abs=1e-3)
    assert res["restated"] >= res["dc"] + MARGIN_OVER_DC_DB[w], res
    for k, v in res.items():
        if k not in ("restated", "dc"):
            assert v <= res["restated"] - MIN_DROP_DB, "perturbation %s does not lose PSNR: %r" % (k, res)


@pytest.mark.gpu
@pytest.mark.parametrize("w", [4, 8])
def test_hip_path_on_natural_contexts(oracle, w):
    """The HIP path on the same natural contexts, trained weights, both arithmetics: Pel predictions within 1 LSB of the oracle's
    (ties at .5 only), pred-PSNR delta 0 to three decimals."""
    import context_adaptive_neural_network_based_prediction_amd as P
    flat, _, _ = wts.load_pnnw(os.path.join(GOLD, "conv%d_single.pnnw" % w))
    a8, l8, tgt = natural_contexts(w, N_CONTEXTS)
    above = a8.astype(np.float32) - np.float32(util.MEAN)
    left = l8.astype(np.float32) - np.float32(util.MEAN)
    want = oracle.epilogue(oracle.conv_forward(flat, w, above, left), util.MEAN)
    for precision in (0, 1):
        net = P.PredictionNeuralNetwork(N_CONTEXTS, w, False, params=flat)
        net.set_option("precision", precision)
        got = net.predict_pel(above, left)
        d = np.abs(got.astype(np.int64) - want)
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, (precision, int(d.max()), float((d != 0).mean()))
        assert abs(psnr(got, tgt) - psnr(want, tgt)) < 1e-3
        net.close()


if __name__ == "__main__":
    for w in (4, 8):
        r, _ = evaluate(w)
        print("width %d (%d natural contexts):" % (w, N_CONTEXTS))
        for k, v in r.items():
            print("    %-24s %7.3f dB  (%+.3f)" % (k, v, v - r["restated"]))
