"""A numpy interpreter for the inference subgraph of the reference's own TF-written graph files.

TEST INFRASTRUCTURE (build container only): used by tests/golden/make_meta_goldens.py and tests/test_oracle.py to pin
the CPU oracle's TOPOLOGY -- op list, strides, padding, Transpose perms, concat axis, Conv2DBackpropInput output
shapes, BatchMatMul operand order -- on what the authors' TensorFlow serialized, instead of on a reading of the Python
that built it (pnn/components.py:10-261, pnn/tfutils.py:8-139,395-462).  Every structural parameter below comes FROM
THE FILE (node attrs and Const nodes), nothing from this repository's architecture tables.

Input files: the `model_*.ckpt.meta` MetaGraphDefs under /root/reference/pnn (training graphs: the network sits between
the input queue's dequeue node and `.../node_output`, with VariableV2/Variable nodes where a frozen graph has Consts), or
a frozen GraphDef written by freezing_graph_pnn.py:131-143 (placeholders `node_portion_above` / `node_portion_left` /
`node_flattened_context`, freezing_graph_pnn.py:100-143).

Per-op semantics are TensorFlow 1.x's documented ones (tensorflow/core/ops/{nn_ops,array_ops,math_ops}.cc):
  * Conv2D, NHWC, filter [kh, kw, in, out], cross-correlation; padding SAME: out = ceil(in / s),
    pad_total = max((out - 1) * s + k - in, 0), pad_before = pad_total // 2 (the extra pixel goes AFTER).
  * Conv2DBackpropInput(input_sizes, filter, out_backprop) = gradient of that Conv2D w.r.t. its input of shape
    input_sizes: every out_backprop pixel scatters filter taps into the input positions the forward conv read.
  * BatchMatMul over the leading dimension with adj_x / adj_y; MatMul with transpose_a / transpose_b.
  * ConcatV2(values..., axis) and the pre-1.0 Concat(concat_dim, values...).
Arithmetic runs in float64 so that the result is the graph's value up to ~1e-12: the fixtures made from it carry no
summation-order noise of their own, and every float32 implementation (TF's Eigen kernels, the oracle, the HIP kernels)
sits within its own rounding of it.
"""
import numpy as np


def _same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def _check_nhwc(node):
    fmt = node.attr("data_format", b"NHWC")
    if fmt != b"NHWC":
        raise NotImplementedError("%s: data_format %r" % (node.name, fmt))


def _strides(node):
    st = node.attr("strides")
    if len(st) != 4 or st[0] != 1 or st[3] != 1:
        raise NotImplementedError("%s: strides %r" % (node.name, st))
    return int(st[1]), int(st[2])


def _geometry(node, H, W, kh, kw):
    """-> (sy, sx, out_h, out_w, pad_top, pad_left) of the FORWARD convolution over an H x W input."""
    sy, sx = _strides(node)
    padding = node.attr("padding")
    if padding == b"SAME":
        oh, pt, _ = _same_pad(H, kh, sy)
        ow, pl, _ = _same_pad(W, kw, sx)
    elif padding == b"VALID":
        oh, pt = (H - kh) // sy + 1, 0
        ow, pl = (W - kw) // sx + 1, 0
    else:
        raise NotImplementedError("%s: padding %r" % (node.name, padding))
    return sy, sx, oh, ow, pt, pl


def conv2d(node, x, f):
    _check_nhwc(node)
    N, H, W, C = x.shape
    kh, kw, ci, co = f.shape
    assert ci == C, (node.name, x.shape, f.shape)
    sy, sx, oh, ow, pt, pl = _geometry(node, H, W, kh, kw)
    xp = np.zeros((N, (oh - 1) * sy + kh, (ow - 1) * sx + kw, C), x.dtype)
    hh, ww = min(H, xp.shape[1] - pt), min(W, xp.shape[2] - pl)
    xp[:, pt:pt + hh, pl:pl + ww] = x[:, :hh, :ww]
    y = np.zeros((N, oh, ow, co), x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + (oh - 1) * sy + 1:sy, kx:kx + (ow - 1) * sx + 1:sx]
            y += patch.reshape(-1, C).dot(f[ky, kx]).reshape(N, oh, ow, co)
    return y


def conv2d_backprop_input(node, input_sizes, f, g):
    _check_nhwc(node)
    N, H, W, C = [int(v) for v in input_sizes]
    kh, kw, ci, co = f.shape                                   # the FORWARD conv's filter: [kh, kw, in, out]
    assert ci == C and g.shape[3] == co, (node.name, input_sizes, f.shape, g.shape)
    sy, sx, oh, ow, pt, pl = _geometry(node, H, W, kh, kw)
    assert g.shape == (N, oh, ow, co), (node.name, g.shape, (N, oh, ow, co))
    dxp = np.zeros((N, (oh - 1) * sy + kh, (ow - 1) * sx + kw, C), g.dtype)
    for ky in range(kh):
        for kx in range(kw):
            contrib = g.reshape(-1, co).dot(f[ky, kx].T).reshape(N, oh, ow, C)
            dxp[:, ky:ky + (oh - 1) * sy + 1:sy, kx:kx + (ow - 1) * sx + 1:sx] += contrib
    out = np.zeros((N, H, W, C), g.dtype)
    hh, ww = min(H, dxp.shape[1] - pt), min(W, dxp.shape[2] - pl)
    out[:, :hh, :ww] = dxp[:, pt:pt + hh, pl:pl + ww]
    return out


def _matmul(a, b, ta, tb):
    if ta:
        a = np.swapaxes(a, -1, -2)
    if tb:
        b = np.swapaxes(b, -1, -2)
    return np.matmul(a, b)


class Interpreter(object):
    """interp = Interpreter(nodes, variables); interp.run(fetch, feeds) with feeds keyed by TENSOR name
    ("node" = "node:0", "node:1" ...).  `variables`: {variable node name: ndarray}; a Variable's `shape` attr
    must equal the array's.  `trace` lists the (op, name) pairs evaluated, in evaluation order."""

    def __init__(self, nodes, variables, dtype=np.float64):
        self.nodes, self.variables, self.dtype = nodes, variables, dtype
        self.trace = []

    def run(self, fetch, feeds):
        self.values = {}
        for k, v in feeds.items():
            self.values[k if ":" in k else k + ":0"] = np.asarray(v, dtype=self.dtype)
        self.trace = []
        return self._tensor(fetch)

    def _tensor(self, name):
        if name.startswith("^"):
            raise ValueError("control input %s has no value" % name)
        key = name if ":" in name else name + ":0"
        if key not in self.values:
            node_name, idx = key.rsplit(":", 1)
            if int(idx) != 0:
                raise KeyError("tensor %s is not fed and its node has no second output here" % key)
            self.values[key] = self._eval(self.nodes[node_name])
        return self.values[key]

    def _eval(self, node):
        op = node.op
        ins = [i for i in node.inputs if not i.startswith("^")]
        self.trace.append((op, node.name))
        if op == "Const":
            v = node.attr("value")
            return v.astype(self.dtype) if v.dtype.kind == "f" else v
        if op in ("Variable", "VariableV2"):
            v = np.asarray(self.variables[node.name])
            shape = node.attr("shape")
            if list(v.shape) != list(shape):
                raise ValueError("%s: the graph declares shape %s, the checkpoint holds %s" % (node.name, shape, v.shape))
            return v.astype(self.dtype)
        if op == "Placeholder":
            raise KeyError("placeholder %s was not fed" % node.name)
        a = [self._tensor(i) for i in ins]
        if op == "Identity":
            return a[0]
        if op == "Conv2D":
            return conv2d(node, a[0], a[1])
        if op == "Conv2DBackpropInput":
            return conv2d_backprop_input(node, a[0], a[1], a[2])
        if op == "BiasAdd":
            _check_nhwc(node)
            assert a[1].ndim == 1 and a[1].shape[0] == a[0].shape[-1], (node.name, a[0].shape, a[1].shape)
            return a[0] + a[1]
        if op == "Mul":
            return a[0] * a[1]
        if op == "Add":
            return a[0] + a[1]
        if op == "Maximum":
            return np.maximum(a[0], a[1])
        if op == "Reshape":
            return a[0].reshape([int(v) for v in a[1]])
        if op == "Transpose":
            return np.transpose(a[0], [int(v) for v in a[1]])
        if op == "ConcatV2":
            assert node.attr("N") == len(a) - 1
            return np.concatenate(a[:-1], axis=int(a[-1]))
        if op == "Concat":
            assert node.attr("N") == len(a) - 1
            return np.concatenate(a[1:], axis=int(a[0]))
        if op == "BatchMatMul":
            return _matmul(a[0], a[1], node.attr("adj_x", False), node.attr("adj_y", False))
        if op == "MatMul":
            return _matmul(a[0], a[1], node.attr("transpose_a", False), node.attr("transpose_b", False))
        if op == "ExpandDims":
            return np.expand_dims(a[0], int(a[1]))
        if op == "Tile":
            return np.tile(a[0], [int(v) for v in a[1]])
        raise NotImplementedError("op %s (%s)" % (op, node.name))


def network_io(nodes):
    """-> (output tensor name, [input tensor names]) of the PNN inside a graph: the output is the one node named
    `.../node_output` (components.py:177-180,251-256); inputs are where a backward walk leaves the
    `convolutional/` / `fully_connected/` scopes (the dequeue node's outputs in a training graph, the placeholders in a
    frozen one), in the order [above, left] resp. [flattened context]."""
    outs = [n for n in nodes if n.endswith("/node_output")]
    if len(outs) != 1:
        raise ValueError("expected one node_output, found %r" % outs)
    seen, boundary = set(), set()

    def walk(t):
        n = t.split(":")[0].lstrip("^")
        if not (n.startswith("convolutional/") or n.startswith("fully_connected/")):
            boundary.add(t if ":" in t else t + ":0")
            return
        if n in seen:
            return
        seen.add(n)
        for i in nodes[n].inputs:
            walk(i)
    walk(outs[0])
    order = {"node_portion_above:0": 0, "node_portion_left:0": 1}
    return outs[0], sorted(boundary, key=lambda t: (order.get(t, 0), t)), seen


def structure(nodes):
    """The structural facts of a PNN graph, read from the file: a list of
    (scope-relative layer name, op, strides, padding, filter/variable shape, extra) in evaluation order of the compute ops."""
    out, _, seen = network_io(nodes)
    rows = []
    for name in nodes:                                          # file order == construction order
        if name not in seen:
            continue
        nd = nodes[name]
        if nd.op in ("Conv2D", "Conv2DBackpropInput"):
            w = nodes[nodes[nd.inputs[1]].inputs[0]] if nodes[nd.inputs[1]].op == "Identity" else nodes[nd.inputs[1]]
            shape = w.attr("shape") if w.op.startswith("Variable") else list(w.attr("value").shape)
            extra = None
            if nd.op == "Conv2DBackpropInput":
                extra = [int(v) for v in nodes[nd.inputs[0]].attr("value")]
            rows.append((name, nd.op, nd.attr("strides"), nd.attr("padding"), shape, extra))
        elif nd.op in ("MatMul", "BatchMatMul"):
            w = nodes[nodes[nd.inputs[1]].inputs[0]] if nodes[nd.inputs[1]].op == "Identity" else nodes[nd.inputs[1]]
            shape = w.attr("shape") if w.op.startswith("Variable") else list(w.attr("value").shape)
            rows.append((name, nd.op, None, None, shape,
                         (nd.attr("transpose_a", False), nd.attr("transpose_b", False), nd.attr("adj_x", False), nd.attr("adj_y", False))))
        elif nd.op in ("Transpose", "Reshape", "Tile", "ExpandDims"):
            rows.append((name, nd.op, None, None, None, [int(v) for v in np.atleast_1d(nodes[nd.inputs[1]].attr("value"))]))
        elif nd.op in ("ConcatV2", "Concat"):
            ax = nd.inputs[-1] if nd.op == "ConcatV2" else nd.inputs[0]
            rows.append((name, nd.op, None, None, None, int(nodes[ax].attr("value"))))
        elif nd.op in ("Mul",):
            c = nodes[nd.inputs[0]]
            rows.append((name, nd.op, None, None, None, float(c.attr("value")) if c.op == "Const" else None))
        elif nd.op in ("Maximum", "BiasAdd", "Add"):
            rows.append((name, nd.op, None, None, None, None))
    return rows
