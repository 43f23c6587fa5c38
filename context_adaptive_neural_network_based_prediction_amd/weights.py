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
# Frozen GraphDef reader: the `graph_output.pbtxt` files written by the
# reference's freezing_graph_pnn.py:131-143 are BINARY GraphDefs whose weights are Const nodes
# named like the variables of Appendix B.7.  No TensorFlow needed.
# ---------------------------------------------------------------------------------------------
def read_frozen_graph_consts(path):
    """{node_name: float32 ndarray} for every float Const node of a binary GraphDef."""
    with open(path, "rb") as f:
        buf = f.read()
    consts = {}
    for field, wt, node in _proto_fields(buf):
        if field != 1 or wt != 2:                       # GraphDef.node
            continue
        name, op, tensor = None, None, None
        for f2, w2, v2 in _proto_fields(node):
            if f2 == 1: name = v2.decode()
            elif f2 == 2: op = v2.decode()
            elif f2 == 5:                               # NodeDef.attr map entry {key = 1, value = 2 (AttrValue)}
                key, val = None, None
                for f3, _, v3 in _proto_fields(v2):
                    if f3 == 1: key = v3.decode()
                    elif f3 == 2: val = v3
                if key == "value" and val is not None:
                    for f4, w4, v4 in _proto_fields(val):
                        if f4 == 8 and w4 == 2: tensor = v4   # AttrValue.tensor (TensorProto)
        if op != "Const" or tensor is None:
            continue
        dtype, shape, content, float_vals = 0, [], None, []
        for f5, w5, v5 in _proto_fields(tensor):
            if f5 == 1: dtype = v5
            elif f5 == 2:
                for f6, _, dim in _proto_fields(v5):
                    if f6 == 2:
                        sz = 0
                        for f7, _, v7 in _proto_fields(dim):
                            if f7 == 1: sz = v7
                        shape.append(sz)
            elif f5 == 4: content = v5
            elif f5 == 5:                               # float_val: packed (wire type 2) or single fixed32 (wire type 5)
                float_vals += list(np.frombuffer(v5, dtype="<f4")) if w5 in (2, 5) else []
        if dtype != 1:
            continue
        n = int(np.prod(shape)) if shape else 1
        if content is not None and len(content) == 4 * n:
            arr = np.frombuffer(content, dtype="<f4").copy()
        elif len(float_vals) == n:
            arr = np.array(float_vals, np.float32)
        elif len(float_vals) == 1:                      # TF stores a constant-filled tensor as one value
            arr = np.full(n, float_vals[0], np.float32)
        else:
            continue
        consts[name] = arr.reshape(shape)
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
