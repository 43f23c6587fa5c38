"""ctypes front-end of the CPU oracle (oracle/pnn_oracle.c) and of the reference gather built into
oracle/_ref/.  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
            os.path.join(_HERE, "pnn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(
            os.path.join(_HERE, "_ref", "libref_extract.so")) or not os.path.exists(
            os.path.join(_HERE, "_ref", "libref_rdcost.so"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.oracle_extract_context.restype = ctypes.c_int
        L.oracle_extract_context.argtypes = [_i32p, _f32p, _f32p, _u8p] + [ctypes.c_int] * 8 + [ctypes.c_float]
        L.oracle_extract_context_u8_rect.restype = ctypes.c_int
        L.oracle_extract_context_u8_rect.argtypes = [_u8p] + [ctypes.c_int] * 5 + [ctypes.c_float] + \
            [ctypes.c_int] * 2 + [_f32p, _f32p]
        L.oracle_param_count.restype = ctypes.c_long
        L.oracle_param_count.argtypes = [ctypes.c_int, ctypes.c_int]
        L.oracle_fc_forward.restype = ctypes.c_int
        L.oracle_fc_forward.argtypes = [_f32p, ctypes.c_int, _f32p, ctypes.c_int, _f32p]
        L.oracle_conv_forward.restype = ctypes.c_int
        L.oracle_conv_forward.argtypes = [_f32p, ctypes.c_int, _f32p, _f32p, ctypes.c_int, _f32p]
        L.oracle_epilogue.restype = None
        L.oracle_epilogue.argtypes = [_f32p, ctypes.c_long, ctypes.c_float, _i32p]
        L.oracle_conv2d_same.restype = None
        L.oracle_conv2d_same.argtypes = [_f32p] + [ctypes.c_int] * 4 + [_f32p, _f32p] + [ctypes.c_int] * 3 + [_f32p]
        L.oracle_tconv2d_same.restype = None
        L.oracle_tconv2d_same.argtypes = L.oracle_conv2d_same.argtypes
        L.oracle_merger_cfc.restype = None
        L.oracle_merger_cfc.argtypes = [_f32p, _f32p] + [ctypes.c_int] * 5 + [_f32p, _f32p, _f32p]
        L.oracle_dense.restype = None
        L.oracle_dense.argtypes = [_f32p, _f32p, _f32p, _f32p] + [ctypes.c_int] * 4
        L.oracle_predict_tbs.restype = ctypes.c_int
        L.oracle_predict_tbs.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, _i32p, ctypes.c_int, _i32p, _i32p,
                                         _u8p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_float, _i32p]
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own extract_context_portions (oracle/_ref); None when it was never built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_extract.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference"):
                build()
            else:
                return None
        R = ctypes.CDLL(path)
        R.ref_extract_context_portions.restype = ctypes.c_int
        R.ref_extract_context_portions.argtypes = [_i32p, _f32p, _f32p, _u8p] + [ctypes.c_int] * 9 + [ctypes.c_float]
        _REF = R
    return _REF


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def extract_context(plane, x, y, w, flags, mean, unit=4, use_ref=False):
    """One TB: plane int32 [H][stride], TB top-left (x, y); flags as HM orders them (index 0 = bottom-most
    below-left unit, 2w/unit = corner, then above -> above-right).  Returns (rc, above[w][3w], left[2w][w])."""
    plane = _c(plane, np.int32)
    flags = _c(flags, np.uint8)
    units = 2 * w // unit
    above = np.full((w, 3 * w), np.nan, np.float32)
    left = np.full((2 * w, w), np.nan, np.float32)
    stride = plane.shape[1]
    origin = ctypes.cast(ctypes.addressof(_p(plane, _i32p).contents) + 4 * (y * stride + x), _i32p)
    n_avail = int(flags[:2 * units + 1].sum())
    if use_ref:
        rc = ref_lib().ref_extract_context_portions(origin, _p(above, _f32p), _p(left, _f32p), _p(flags, _u8p),
                                                    flags.size, n_avail, unit, unit, units, units, w, w, stride,
                                                    np.float32(mean))
    else:
        rc = lib().oracle_extract_context(origin, _p(above, _f32p), _p(left, _f32p), _p(flags, _u8p), n_avail,
                                          unit, unit, units, units, w, w, stride, np.float32(mean))
    return rc, above, left


def extract_context_u8_rect(img, w, row, col, mean, mask_w, mask_h):
    img = _c(img, np.uint8)
    H, W = img.shape
    above = np.zeros((w, 3 * w), np.float32)
    left = np.zeros((2 * w, w), np.float32)
    rc = lib().oracle_extract_context_u8_rect(_p(img, _u8p), H, W, w, row, col, np.float32(mean), mask_w, mask_h,
                                              _p(above, _f32p), _p(left, _f32p))
    return rc, above, left


def param_count(w, is_fc):
    return int(lib().oracle_param_count(w, int(is_fc)))


def fc_forward(params, w, ctx):
    params = _c(params, np.float32)
    ctx = _c(ctx, np.float32).reshape(-1, 5 * w * w)
    assert params.size == param_count(w, True)
    out = np.empty((ctx.shape[0], w, w), np.float32)
    rc = lib().oracle_fc_forward(_p(params, _f32p), w, _p(ctx, _f32p), ctx.shape[0], _p(out, _f32p))
    assert rc == 0
    return out


def conv_forward(params, w, above, left):
    params = _c(params, np.float32)
    above = _c(above, np.float32).reshape(-1, w, 3 * w)
    left = _c(left, np.float32).reshape(-1, 2 * w, w)
    assert params.size == param_count(w, False) and above.shape[0] == left.shape[0]
    out = np.empty((above.shape[0], w, w), np.float32)
    rc = lib().oracle_conv_forward(_p(params, _f32p), w, _p(above, _f32p), _p(left, _f32p), above.shape[0],
                                   _p(out, _f32p))
    assert rc == 0
    return out


def epilogue(pred, mean):
    pred = _c(pred, np.float32)
    dst = np.empty(pred.shape, np.int32)
    lib().oracle_epilogue(_p(pred, _f32p), pred.size, np.float32(mean), _p(dst, _i32p))
    return dst


def conv2d_same(x, W, b, s, act):
    x = _c(x, np.float32); W = _c(W, np.float32); b = _c(b, np.float32)
    B, H, Wd, Cin = x.shape
    Cout = W.shape[3]
    y = np.empty((B, (H + s - 1) // s, (Wd + s - 1) // s, Cout), np.float32)
    lib().oracle_conv2d_same(_p(x, _f32p), B, H, Wd, Cin, _p(W, _f32p), _p(b, _f32p), s, Cout, int(act), _p(y, _f32p))
    return y


def tconv2d_same(x, W, b, s, act):
    x = _c(x, np.float32); W = _c(W, np.float32); b = _c(b, np.float32)
    B, H, Wd, Cin = x.shape
    Cout = W.shape[2]
    y = np.empty((B, H * s, Wd * s, Cout), np.float32)
    lib().oracle_tconv2d_same(_p(x, _f32p), B, H, Wd, Cin, _p(W, _f32p), _p(b, _f32p), s, Cout, int(act), _p(y, _f32p))
    return y


def merger_cfc(a, l, Wm, bm):
    a = _c(a, np.float32); l = _c(l, np.float32); Wm = _c(Wm, np.float32); bm = _c(bm, np.float32)
    B, C = a.shape[0], a.shape[-1]
    na = a.shape[1] * a.shape[2]
    nl = l.shape[1] * l.shape[2]
    nout = Wm.shape[2]
    out = np.empty((B, nout, C), np.float32)
    lib().oracle_merger_cfc(_p(a, _f32p), _p(l, _f32p), B, C, na, nl, nout, _p(Wm, _f32p), _p(bm, _f32p), _p(out, _f32p))
    return out


def dense(x, W, b, act):
    x = _c(x, np.float32); W = _c(W, np.float32); b = _c(b, np.float32)
    y = np.empty((x.shape[0], W.shape[1]), np.float32)
    lib().oracle_dense(_p(x, _f32p), _p(W, _f32p), _p(b, _f32p), _p(y, _f32p), x.shape[0], W.shape[0], W.shape[1], int(act))
    return y


def block_costs(org_plane, xs, ys, w, pred, hadamard=True, use_ref=False):
    """Distortion (HM's HADs / SAD, first intra pass) of N predicted blocks [N][w][w] against the original picture.
    use_ref=True: the reference's own TComRdCost (oracle/_ref/libref_rdcost.so), block by block."""
    org_plane = _c(org_plane, np.int32); pred = _c(pred, np.int32)
    xs = _c(xs, np.int32); ys = _c(ys, np.int32)
    N = xs.size
    cost = np.empty(N, np.uint32)
    if use_ref:
        R = ref_rdcost_lib()
        for i in range(N):
            blk = np.ascontiguousarray(org_plane[ys[i]:ys[i] + w, xs[i]:xs[i] + w])
            cost[i] = R.ref_block_cost(_p(blk, _i32p), w, _p(np.ascontiguousarray(pred[i]), _i32p), w, w, w, int(hadamard))
        return cost
    L = lib()
    L.oracle_block_costs.restype = None
    L.oracle_block_costs(_p(org_plane, _i32p), org_plane.shape[1], _p(xs, _i32p), _p(ys, _i32p), N, w, _p(pred, _i32p),
                         int(hadamard), cost.ctypes.data_as(ctypes.c_void_p))
    return cost


_REF_RD = None


def ref_rdcost_lib():
    """The reference's TComRdCost behind a C shim (oracle/_ref); None when it was never built."""
    global _REF_RD
    if _REF_RD is None:
        path = os.path.join(_HERE, "_ref", "libref_rdcost.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference"):
                build()
            if not os.path.exists(path):
                return None
        R = ctypes.CDLL(path)
        R.ref_block_cost.restype = ctypes.c_uint
        R.ref_block_cost.argtypes = [_i32p, ctypes.c_int, _i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _REF_RD = R
    return _REF_RD


def predict_tbs(params, w, is_fc, plane, xs, ys, flags, mean):
    """gather -> net -> (+mean, clamp, round): int32 [N][w][w].  flags uint8 [N][pitch]."""
    params = _c(params, np.float32)
    plane = _c(plane, np.int32)
    xs = _c(xs, np.int32); ys = _c(ys, np.int32)
    flags = _c(flags, np.uint8)
    N = xs.size
    units = 2 * w // 4
    n_avail = _c(flags[:, :2 * units + 1].sum(axis=1), np.int32)
    dst = np.empty((N, w, w), np.int32)
    rc = lib().oracle_predict_tbs(_p(params, _f32p), w, int(is_fc), _p(plane, _i32p), plane.shape[1], _p(xs, _i32p),
                                  _p(ys, _i32p), _p(flags, _u8p), flags.shape[1], _p(n_avail, _i32p), N,
                                  np.float32(mean), _p(dst, _i32p))
    assert rc == 0, rc
    return dst
