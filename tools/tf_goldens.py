#!/usr/bin/env python
"""Closes the one pin this repository cannot close by itself: the reference's TensorFlow arithmetic at the network boundary.

The nets' arithmetic lives in TensorFlow 1.x (pnn/tfutils.py:134-139,455-462, pnn/components.py:169-176), which is not installable in
the build container and of which the reference holds no numeric output: the CPU oracle is pinned on the graphs'
TOPOLOGY and on natural-content statistics, not on TensorFlow's own float32 outputs ("parity unpinned").
Whoever has TensorFlow 1.x (1.4 ... 1.15, CPU is enough) and the reference checkout runs THIS script once:

    python tools/tf_goldens.py --reference /path/to/context_adaptive_neural_network_based_prediction [--out tests/golden/tf_outputs.npz]

It builds the reference's OWN graphs (pnn.PredictionNeuralNetwork.PredictionNeuralNetwork, inference only), assigns the seeded weights
of tests/golden/nets.npz to the variables by their names (SURVEY.md Appendix B.7 = weights.tensor_specs), feeds the seeded contexts of
the same file, and writes the fetched predictions -- plus, for the two trained checkpoints of the checkout (conv 4x4 / 8x8), the
predictions of the RESTORED models on the committed natural-like contexts.  Commit the result; from then on

    tests/test_oracle.py::test_oracle_matches_tensorflow_outputs          (CPU)   oracle == TensorFlow within 1e-3, HM epilogue within 1 LSB
    tests/test_gpu_parity.py::test_hip_matches_tensorflow_outputs         (GPU)   the HIP path, both arithmetics, within 1 LSB per pixel

run instead of skipping, and DESIGN.md's parity row may say "pinned".  Nothing here is run by the build, the tests or the bench:
this container has no TensorFlow.  The script imports nothing from oracle/ -- it produces reference outputs, it does not check them.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [(True, 4), (True, 8), (True, 16), (False, 4), (False, 8), (False, 16), (False, 32), (False, 64)]   # tests/golden/make_golden.py: gen_nets


def run_seeded(tf, pnn_cls, wts, util, is_fc, w, seed, n):
    """The reference's graph for (kind, width) with the seeded parameters assigned, run on the seeded contexts."""
    params = util.make_params(w, is_fc, seed, out_gain=util.out_gain(w, is_fc))
    above, left = util.make_contexts(w, n, seed + 1)
    tensors = wts.split_params(params, w, is_fc)
    graph = tf.Graph()
    with graph.as_default():
        net = pnn_cls(n, w, is_fc)                                # batch_size, width_target, is_fully_connected: inference graph only
        by_name = {v.op.name: v for v in tf.global_variables()}
        missing = sorted(set(tensors) - set(by_name))
        if missing:
            raise SystemExit("the graph has no variable named %s (has: %s ...)" % (missing[0], sorted(by_name)[:4]))
        with tf.Session(graph=graph) as sess:
            sess.run(tf.global_variables_initializer())
            for name, value in tensors.items():
                by_name[name].load(value, sess)
            if is_fc:
                feed = {net.node_flattened_contexts_float32: util.flatten_fc(above, left)}
            else:
                feed = {net.node_portions_above_float32: above[..., None], net.node_portions_left_float32: left[..., None]}
            out = sess.run(net.node_predictions_float32, feed_dict=feed)
    return np.asarray(out, np.float32).reshape(n, w, w)


def run_trained(tf, pnn_cls, reference, w, above, left):
    """The reference's trained conv checkpoint of width w (the only complete trained weights the checkout ships), restored by TF's own Saver."""
    prefix = os.path.join(reference, "pnn", "results", "width_target_%d" % w, "convolutional", "single", "luminance", "1_0", "masks_tr_random",
                          "model_800000.ckpt")
    n = above.shape[0]
    graph = tf.Graph()
    with graph.as_default():
        net = pnn_cls(n, w, False)
        saver = tf.train.Saver(tf.global_variables())
        with tf.Session(graph=graph) as sess:
            saver.restore(sess, prefix)
            out = sess.run(net.node_predictions_float32, feed_dict={net.node_portions_above_float32: above[..., None],
                                                                    net.node_portions_left_float32: left[..., None]})
    return np.asarray(out, np.float32).reshape(n, w, w)


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--reference", required=True, help="checkout of thierrydumas/context_adaptive_neural_network_based_prediction")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "tf_outputs.npz"))
    args = ap.parse_args()
    try:
        import tensorflow as tf
    except ImportError:
        raise SystemExit("tools/tf_goldens.py needs TensorFlow 1.x (the reference's: 1.4 - 1.5 for Python, README.md); there is none in this environment")
    if int(tf.__version__.split(".")[0]) != 1:
        tf = tf.compat.v1                                         # TF 2.x: the 1.x graph API, eager off
        tf.disable_v2_behavior()
    sys.path.insert(0, args.reference)
    from pnn.PredictionNeuralNetwork import PredictionNeuralNetwork as pnn_cls
    from context_adaptive_neural_network_based_prediction_amd import weights as wts
    from tests import util
    gold = np.load(os.path.join(ROOT, "tests", "golden", "nets.npz"))
    rec = {"tf_version": np.array(str(getattr(tf, "__version__", "compat.v1")))}
    for is_fc, w in CASES:
        tag = "%s%d" % ("fc" if is_fc else "conv", w)
        seed, n = int(gold[tag + "_seed"]), int(gold[tag + "_n"])
        rec[tag + "_out"] = run_seeded(tf, pnn_cls, wts, util, is_fc, w, seed, n)
        d = np.abs(rec[tag + "_out"] - gold[tag + "_out"]).max()
        print("%-7s TensorFlow vs the committed oracle outputs: max |delta| = %.3e" % (tag, d))
    for w in (4, 8):
        rec["real%d_out" % w] = run_trained(tf, pnn_cls, args.reference, w, gold["real%d_above" % w], gold["real%d_left" % w])
        d = np.abs(rec["real%d_out" % w] - gold["real%d_out" % w]).max()
        print("real%d   TensorFlow (restored checkpoint) vs the committed oracle outputs: max |delta| = %.3e" % (w, d))
    np.savez_compressed(args.out, **rec)
    print("wrote %s -- commit it: tests/test_oracle.py and tests/test_gpu_parity.py then compare against TensorFlow itself" % args.out)


if __name__ == "__main__":
    main()
