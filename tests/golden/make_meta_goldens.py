"""Golden vectors from the reference's OWN serialized graphs.  Run in the build container (needs /root/reference):

    python tests/golden/make_meta_goldens.py

The reference ships no TensorFlow outputs, but it ships 13 TF-written MetaGraphDefs (`model_*.ckpt.meta`) -- the graphs
its TensorFlow built from pnn/components.py / pnn/tfutils.py -- two of them beside complete checkpoints.
`oracle/tf_graph_interp.py` executes the inference subgraph of each file with every structural parameter (strides,
padding, Transpose perms, concat axis, Conv2DBackpropInput output shapes, operand orders) taken from the file.

meta_graphs.npz, per graph <tag>:
  <tag>_path        path below /root/reference/pnn
  <tag>_info        [width, is_fc, batch size baked into the graph, seed (-1 = the real checkpoint), blocks kept]
  <tag>_out         float32 [kept][w][w]: the graph's node_output for the first `kept` blocks of
                    util.make_contexts(width, batch, seed + 1) (seeded graphs: variables = util.make_params(width, is_fc,
                    seed, out_gain); real checkpoints: the variables of the checkpoint itself, read by weights.read_tf_bundle)
  <tag>_structure   JSON: oracle.tf_graph_interp.structure(graph) -- the op list with attrs, as the file states it
ref_meta/*.meta.gz  the two .meta files the reference's own test_pnn.py:465-494 uses as test data (FC 4x4, conv 16x16),
                    gzipped: TF-written bytes for the protobuf walker and the interpreter where /root/reference is absent.
"""
import glob
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from context_adaptive_neural_network_based_prediction_amd import weights as wts  # noqa: E402
from oracle import tf_graph_interp as tfi  # noqa: E402
from tests import util  # noqa: E402

REF_PNN = "/root/reference/pnn"
KEEP = {4: 100, 8: 100, 16: 32, 32: 8}


def tag_of(rel):
    parts = rel.split("/")
    if parts[0] == "pseudo_data":
        return "pseudo_w%s" % parts[2].split("_")[-1]
    return "%s_%s_w%s" % ("fc" if parts[2] == "fully_connected" else "conv", parts[3], parts[1].split("_")[-1])


def graph_facts(nodes):
    """(width, is_fc, batch) as the FILE states them: from the last Reshape / Conv2DBackpropInput output shape."""
    out, ins, _ = tfi.network_io(nodes)
    is_fc = out.startswith("fully_connected/")
    rows = tfi.structure(nodes)
    if is_fc:
        shape = [r for r in rows if r[1] == "Reshape"][-1][5]
    else:
        shape = [r for r in rows if r[1] == "Conv2DBackpropInput"][-1][5]
    assert shape[1] == shape[2] and shape[3] == 1, shape
    return int(shape[1]), is_fc, int(shape[0])


def run_graph(nodes, variables, width, is_fc, batch, seed):
    out, ins, _ = tfi.network_io(nodes)
    above, left = util.make_contexts(width, batch, seed)
    if is_fc:
        assert len(ins) == 1, ins
        feeds = {ins[0]: util.flatten_fc(above, left)}
    else:
        assert len(ins) == 2, ins
        feeds = {ins[0]: above[..., None], ins[1]: left[..., None]}
    y = tfi.Interpreter(nodes, variables).run(out, feeds)
    assert y.shape == (batch, width, width, 1), y.shape
    return y[..., 0]


def main():
    rec = {}
    tags = []
    for path in sorted(glob.glob(REF_PNN + "/**/*.meta", recursive=True)):
        rel = os.path.relpath(path, REF_PNN)
        tag = tag_of(rel)
        nodes = wts.read_meta_graph(path)
        width, is_fc, batch = graph_facts(nodes)
        keep = min(KEEP[width], batch)
        cases = [(tag, 4000 + width + (100 if is_fc else 0))]
        prefix = path[:-len(".meta")]
        if os.path.exists(prefix + ".data-00000-of-00001") and os.path.getsize(prefix + ".data-00000-of-00001") > 4096:
            cases.append((tag + "_real", -1))
        for t, seed in cases:
            if seed < 0:
                variables = wts.read_tf_bundle(prefix)
                in_seed = 4500 + width
            else:
                flat = util.make_params(width, is_fc, seed, out_gain=util.out_gain(width, is_fc))
                variables = wts.split_params(flat, width, is_fc)
                in_seed = seed + 1
            y = run_graph(nodes, variables, width, is_fc, batch, in_seed)
            rec[t + "_path"] = rel
            rec[t + "_info"] = np.array([width, int(is_fc), batch, seed, keep, in_seed])
            rec[t + "_out"] = y[:keep].astype(np.float32)
            rec[t + "_structure"] = json.dumps(tfi.structure(nodes), default=lambda b: b.decode())
            tags.append(t)
            print("%-22s w=%-2d %-4s batch %-3d seed %-5d out range [%.1f, %.1f]" % (
                t, width, "FC" if is_fc else "conv", batch, seed, y.min(), y.max()))
    rec["tags"] = np.array(tags)
    np.savez_compressed(os.path.join(HERE, "meta_graphs.npz"), **rec)
    os.makedirs(os.path.join(HERE, "ref_meta"), exist_ok=True)
    for w in (4, 16):
        src = "%s/pseudo_data/predict_by_batch_via_pnn/width_target_%d/model.ckpt.meta" % (REF_PNN, w)
        with open(src, "rb") as f, gzip.GzipFile(os.path.join(HERE, "ref_meta", "pseudo_w%d.meta.gz" % w), "wb", mtime=0) as g:
            g.write(f.read())
    print("meta_graphs.npz: %d graphs" % len(tags))


if __name__ == "__main__":
    main()
