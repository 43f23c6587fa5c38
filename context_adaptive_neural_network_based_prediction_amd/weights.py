"""Weight tooling for the PNN models: architecture tables, the canonical flat parameter order, the
`.pnnw` flat weight file consumed by libpnn_hip.so, seeded initialisation with the reference's
initialiser statistics, and a dependency-free reader of TensorFlow "V2 bundle" checkpoints.

Reference: pnn/PredictionNeuralNetwork.py:119-137 (stride tuples), pnn/components.py:103-180 (FC),
pnn/components.py:10-101,182-261 + pnn/tfutils.py:8-139,395-462 (conv), SURVEY.md Appendix B.7 (names).
"""
import os
import struct

import numpy as np

FC_HIDDEN = 1200                     # pnn/components.py:130-160
MEAN_TRAINING_LUMINANCE = 117.8952234192841   # sets/results/training_set/means/luminance/mean_training.pkl
STRIDES_BRANCH = {4: (1, 1), 8: (2, 1), 16: (2, 1, 2, 1), 32: (2, 2, 1, 2, 1), 64: (2, 2, 2, 2, 1)}
PNNW_MAGIC = b"PNNW"
PNNW_VERSION = 1
PNNW_HEADER = struct.Struct("<4sIIIQQ")   # magic, version, width, is_fc, n_params, reserved


def tensor_specs(width, is_fc):
    """[(tf_variable_name, shape, init_std)] in the canonical flat order (== oracle_param_count order)."""
    specs = []
    if is_fc:
        h = FC_HIDDEN
        dims = [(5 * width * width, h, 0.01), (h, h, 0.029), (h, h, 0.029), (h, width * width, 0.01)]
        for i, (k, n, std) in enumerate(dims):                       # components.py:130-166
            specs.append(("fully_connected/weights_%d" % i, (k, n), std))
            specs.append(("fully_connected/biases_%d" % i, (n,), 0.0))
        return specs
    strides = STRIDES_BRANCH[width]
    c = 32
    for branch in ("branch_above", "branch_left"):
        cin, c = 1, 32
        for i, s in enumerate(strides):
            k = 2 * s + 1
            c *= s                                                    # components.py:37
            std = 0.01 if i == 0 else 1.0 / np.sqrt(cin * k * k)      # tfutils.py:112-122, components.py:39-40
            specs.append(("convolutional/%s/convolution_%d/weights" % (branch, i), (k, k, cin, c), std))
            specs.append(("convolutional/%s/convolution_%d/biases" % (branch, i), (c,), 0.0))
            cin = c
    m = "convolutional/merger/"
    specs.append((m + "channelwise_fully_connected_merger/weights", (c, 80, 16), 1.0 / np.sqrt(80.0)))  # tfutils.py:48-55
    specs.append((m + "channelwise_fully_connected_merger/biases", (c, 16), 0.0))
    ci = c
    rev = strides[::-1]                                               # components.py:80
    for i, s in enumerate(rev):
        k = 2 * s + 1
        last = i == len(rev) - 1
        co = 1 if last else ci // s                                   # components.py:243-249
        std = 0.01 if last else 1.0 / np.sqrt(ci * k * k)             # tfutils.py:433-437
        specs.append((m + "transpose_convolution_%d/weights" % i, (k, k, co, ci), std))
        specs.append((m + "transpose_convolution_%d/biases" % i, (co,), 0.0))
        ci = co
    return specs


def param_count(width, is_fc):
    return int(sum(int(np.prod(s)) for _, s, _ in tensor_specs(width, is_fc)))


def output_node_name(width, is_fc):
    """Frozen-graph fetch name (freezing_graph_pnn.py:100-143; TComPrediction.cpp:571,593-599)."""
    if is_fc:
        return "fully_connected/node_output"
    return "convolutional/merger/transpose_convolution_%d/node_output" % (len(STRIDES_BRANCH[width]) - 1)


def init_params(width, is_fc, seed, bias_std=0.0):
    """Seeded N(0, std) draw with the reference initialisers' std.  The reference zero-initialises
    biases; `bias_std` > 0 draws them too so that parity tests exercise the bias path."""
    rng = np.random.RandomState(seed)
    chunks = []
    for name, shape, std in tensor_specs(width, is_fc):
        if std > 0:
            chunks.append((rng.standard_normal(int(np.prod(shape))) * std).astype(np.float32))
        elif bias_std > 0:
            chunks.append((rng.standard_normal(int(np.prod(shape))) * bias_std).astype(np.float32))
        else:
            chunks.append(np.zeros(int(np.prod(shape)), np.float32))
    return np.concatenate(chunks)


def split_params(flat, width, is_fc):
    """flat float32 -> {name: ndarray(shape)} (views)."""
    out, off = {}, 0
    for name, shape, _ in tensor_specs(width, is_fc):
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].reshape(shape)
        off += n
    assert off == flat.size, (off, flat.size)
    return out


def save_pnnw(path, flat, width, is_fc):
    flat = np.ascontiguousarray(flat, dtype="<f4")
    assert flat.size == param_count(width, is_fc)
    with open(path, "wb") as f:
        f.write(PNNW_HEADER.pack(PNNW_MAGIC, PNNW_VERSION, width, int(is_fc), flat.size, 0))
        f.write(flat.tobytes())


def load_pnnw(path):
    with open(path, "rb") as f:
        magic, ver, width, is_fc, n, _ = PNNW_HEADER.unpack(f.read(PNNW_HEADER.size))
        if magic != PNNW_MAGIC or ver != PNNW_VERSION:
            raise ValueError("%s is not a PNNW v%d file" % (path, PNNW_VERSION))
        flat = np.frombuffer(f.read(4 * n), dtype="<f4")
    if flat.size != n or n != param_count(width, bool(is_fc)):
        raise ValueError("%s: parameter count mismatch" % path)
    return flat.copy(), width, bool(is_fc)


# ---------------------------------------------------------------------------------------------
# TensorFlow V2 bundle checkpoint reader -- no TensorFlow needed.
# ---------------------------------------------------------------------------------------------
def _varint(buf, pos):
    res = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        res |= (b & 0x7F) << shift
        if not b & 0x80:
            return res, pos
        shift += 7


def _block_entries(buf, off, size):
    blk = buf[off:off + size]
    n_restarts = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(blk, pos)
        non_shared, pos = _varint(blk, pos)
        vlen, pos = _varint(blk, pos)
        key = key[:shared] + blk[pos:pos + non_shared]
        pos += non_shared
        yield key, blk[pos:pos + vlen]
        pos += vlen


def _proto_fields(buf):
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]; pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]; pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, val


def read_tf_bundle(prefix):
    """{tensor_name: float32/int32 ndarray} from `<prefix>.index` + `<prefix>.data-00000-of-00001`."""
    with open(prefix + ".index", "rb") as f:
        idx = f.read()
    with open(prefix + ".data-00000-of-00001", "rb") as f:
        data = f.read()
    footer = idx[-48:]
    if footer[-8:] != struct.pack("<Q", 0xdb4775248b80fb57):
        raise ValueError("bad table magic in %s.index" % prefix)
    pos = 0
    _, pos = _varint(footer, pos); _, pos = _varint(footer, pos)      # metaindex handle
    ioff, pos = _varint(footer, pos); isz, pos = _varint(footer, pos)  # index handle
    tensors = {}
    for _, handle in _block_entries(idx, ioff, isz):
        boff, p = _varint(handle, 0)
        bsz, p = _varint(handle, p)
        for key, val in _block_entries(idx, boff, bsz):
            if not key:
                continue                                               # BundleHeaderProto
            dtype, shape, offset, size = 0, [], 0, 0
            for field, wt, v in _proto_fields(val):
                if field == 1: dtype = v
                elif field == 2:
                    for f2, _, dim in _proto_fields(v):
                        if f2 == 2:
                            sz = 0
                            for f3, _, v3 in _proto_fields(dim):
                                if f3 == 1: sz = v3
                            shape.append(sz)
                elif field == 4: offset = v
                elif field == 5: size = v
            np_dt = {1: "<f4", 3: "<i4", 9: "<i8"}.get(dtype)
            if np_dt is None:
                continue
            tensors[key.decode()] = np.frombuffer(data[offset:offset + size], dtype=np_dt).reshape(shape).copy()
    return tensors


def params_from_tf_bundle(prefix, width, is_fc):
    """Flat canonical parameters from a reference checkpoint (Adam slots and counters dropped)."""
    t = read_tf_bundle(prefix)
    chunks = []
    for name, shape, _ in tensor_specs(width, is_fc):
        if name not in t:
            raise KeyError("%s missing from %s" % (name, prefix))
        if tuple(t[name].shape) != tuple(shape):
            raise ValueError("%s: shape %s, expected %s" % (name, t[name].shape, shape))
        chunks.append(t[name].astype(np.float32).ravel())
    return np.concatenate(chunks)


# ---------------------------------------------------------------------------------------------
# GraphDef / MetaGraphDef walker, no TensorFlow needed.  Two kinds of TF-written
# files carry the PNN graphs: the frozen `graph_output.pbtxt` files of freezing_graph_pnn.py:131-143
# (BINARY GraphDefs whose weights are Const nodes named like the variables of Appendix B.7) and the
# `model_*.ckpt.meta` files beside every checkpoint (MetaGraphDef: field 2 = the training GraphDef, whose
# inference subgraph holds the same ops with VariableV2 nodes where freezing puts Consts).
# Field numbers: tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape}.proto (TF 1.x).
# ---------------------------------------------------------------------------------------------
_TF_DTYPES = {1: "<f4", 2: "<f8", 3: "<i4", 9: "<i8", 10: "?"}   # DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64, DT_BOOL


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(wt, v):
    if wt == 0:
        return [_signed64(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed64(x))
    return out


def parse_tensor_shape(buf):
    """TensorShapeProto -> list of dims (dim = 2 {size = 1}); unknown_rank (3) -> None."""
    shape = []
    for f, wt, v in _proto_fields(buf):
        if f == 2:
            sz = 0
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1: sz = _signed64(v2)
            shape.append(sz)
        elif f == 3 and v:
            return None
    return shape


def parse_tensor_proto(buf):
    """TensorProto -> ndarray (dtype = 1, tensor_shape = 2, tensor_content = 4, float_val = 5, double_val = 6,
    int_val = 7, int64_val = 10, bool_val = 11); string tensors -> None.  A tensor whose repeated *_val field
    holds ONE value is that value broadcast over the shape (how TF stores constant-filled tensors)."""
    dtype, shape, content, vals = 0, [], None, []
    for f, wt, v in _proto_fields(buf):
        if f == 1: dtype = v
        elif f == 2: shape = parse_tensor_shape(v)
        elif f == 4: content = v
        elif f == 5: vals += list(np.frombuffer(v, dtype="<f4"))              # packed or one fixed32
        elif f == 6: vals += list(np.frombuffer(v, dtype="<f8"))
        elif f in (7, 10, 11): vals += _packed_varints(wt, v)
    np_dt = _TF_DTYPES.get(dtype)
    if np_dt is None or shape is None:
        return None
    n = int(np.prod(shape)) if shape else 1
    if content is not None and len(content) == n * np.dtype(np_dt).itemsize:
        arr = np.frombuffer(content, dtype=np_dt).copy()
    elif len(vals) == n:
        arr = np.array(vals, dtype=np_dt)
    elif len(vals) == 1:
        arr = np.full(n, vals[0], dtype=np_dt)
    elif n == 0:
        arr = np.zeros(0, dtype=np_dt)
    else:
        return None
    return arr.reshape(shape)


def parse_attr_value(buf):
    """AttrValue -> Python value: s (2) bytes, i (3) int, f (4) float, b (5) bool, type (6) int, shape (7) list,
    tensor (8) ndarray, list (1) -> list of the same."""
    for f, wt, v in _proto_fields(buf):
        if f == 2: return bytes(v)
        if f == 3: return _signed64(v)
        if f == 4: return float(np.frombuffer(v, dtype="<f4")[0])
        if f == 5: return bool(v)
        if f == 6: return int(v)
        if f == 7: return parse_tensor_shape(v)
        if f == 8: return parse_tensor_proto(v)
        if f == 1:
            out = []
            for f2, w2, v2 in _proto_fields(v):
                if f2 == 2: out.append(bytes(v2))
                elif f2 in (3, 6): out += _packed_varints(w2, v2)
                elif f2 == 4: out += [float(x) for x in np.frombuffer(v2, dtype="<f4")]
                elif f2 == 5: out += [bool(x) for x in _packed_varints(w2, v2)]
                elif f2 == 7: out.append(parse_tensor_shape(v2))
                elif f2 == 8: out.append(parse_tensor_proto(v2))
            return out
    return None


class GraphNode(object):
    """One NodeDef: name (1), op (2), inputs (3, repeated), attr (5, map<string, AttrValue>) decoded on demand."""
    __slots__ = ("name", "op", "inputs", "_attrs")

    def __init__(self, name, op, inputs, attrs):
        self.name, self.op, self.inputs, self._attrs = name, op, inputs, attrs

    def attr(self, key, default=None):
        raw = self._attrs.get(key)
        return default if raw is None else parse_attr_value(raw)

    def attr_names(self):
        return sorted(self._attrs)


def parse_graph_def(buf):
    """Binary GraphDef -> {node name: GraphNode}, in file order."""
    nodes = {}
    for field, wt, node in _proto_fields(buf):
        if field != 1 or wt != 2:                       # GraphDef.node
            continue
        name, op, inputs, attrs = None, None, [], {}
        for f2, w2, v2 in _proto_fields(node):
            if f2 == 1: name = v2.decode()
            elif f2 == 2: op = v2.decode()
            elif f2 == 3: inputs.append(v2.decode())
            elif f2 == 5:                               # map entry {key = 1, value = 2}
                key, val = None, b""
                for f3, _, v3 in _proto_fields(v2):
                    if f3 == 1: key = v3.decode()
                    elif f3 == 2: val = v3
                attrs[key] = val
        if name is not None:
            nodes[name] = GraphNode(name, op, inputs, attrs)
    return nodes


def read_graph_def(path):
    """{node name: GraphNode} of a binary GraphDef file (a frozen `graph_output.pbtxt`)."""
    with open(path, "rb") as f:
        return parse_graph_def(f.read())


def meta_graph_def_bytes(path):
    """The serialized GraphDef (field 2) inside a MetaGraphDef file (`model_*.ckpt.meta`)."""
    with open(path, "rb") as f:
        buf = f.read()
    for field, wt, v in _proto_fields(buf):
        if field == 2 and wt == 2:
            return bytes(v)
    raise ValueError("%s holds no graph_def" % path)


def read_meta_graph(path):
    """{node name: GraphNode} of the graph_def of a MetaGraphDef file."""
    return parse_graph_def(meta_graph_def_bytes(path))


def read_frozen_graph_consts(path):
    """{node_name: float32 ndarray} for every float Const node of a binary GraphDef."""
    consts = {}
    for name, node in read_graph_def(path).items():
        if node.op != "Const":
            continue
        arr = node.attr("value")
        if isinstance(arr, np.ndarray) and arr.dtype == np.float32:
            consts[name] = arr
    return consts


def params_from_frozen_graph(path, width, is_fc):
    """Flat canonical parameters from a frozen graph written by the reference's freezing_graph_pnn.py."""
    t = read_frozen_graph_consts(path)
    chunks = []
    for name, shape, _ in tensor_specs(width, is_fc):
        if name not in t:
            raise KeyError("%s: no Const node named %s" % (path, name))
        if tuple(t[name].shape) != tuple(shape):
            raise ValueError("%s: shape %s, expected %s" % (name, t[name].shape, shape))
        chunks.append(t[name].astype(np.float32).ravel())
    return np.concatenate(chunks)


def convert_model(src, dst, width, is_fc):
    """Any supported source (frozen GraphDef `.pbtxt`/`.pb`, TF V2 checkpoint prefix, `.pnnw`) -> `.pnnw`."""
    if src.endswith(".pnnw"):
        flat, w, fc = load_pnnw(src)
        if (w, fc) != (width, bool(is_fc)):
            raise ValueError("%s holds a width-%d %s model" % (src, w, "FC" if fc else "conv"))
    elif os.path.exists(src + ".index"):
        flat = params_from_tf_bundle(src, width, is_fc)
    else:
        flat = params_from_frozen_graph(src, width, is_fc)
    save_pnnw(dst, flat, width, is_fc)
    return flat


def write_model_table(path, entries):
    """entries: [(width, is_pair, channel, path)] -> the `width,is_pair,channel,path` text table of
    hevc/hm_common/paths_to_graphs_output/{single,pair}.txt."""
    with open(path, "w") as f:
        for w, pair, ch, p in entries:
            f.write("%d,%d,%d,%s\n" % (w, int(pair), ch, p))
    return os.path.abspath(path)
