"""ctypes binding of libpnn_hip.so (the C ABI of include/pnn_hip.h).

There is deliberately no fallback: if the shared library is missing this module raises, and if no HIP
device is visible `pnn_create_empty` fails -- the product path never computes on the CPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PNN_LIB_PATH") or os.path.join(_HERE, "libpnn_hip.so")   # override: diagnostic builds

f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int32)
u8p = ctypes.POINTER(ctypes.c_uint8)
vp = ctypes.c_void_p
ci = ctypes.c_int


class TbDev(ctypes.Structure):
    """pnn_tb_dev of include/pnn_hip.h (24 bytes)."""
    _fields_ = [("origin", ctypes.c_int64), ("stride", ctypes.c_int32), ("above_mask", ctypes.c_uint32),
                ("left_units", ctypes.c_int32), ("reserved", ctypes.c_int32)]


BACKEND = ctypes.CFUNCTYPE(ci, vp, ci, f32p, f32p, ci, i32p, f32p)   # pnn_service_backend of include/pnn_service.h

# include/pnn_service.h (cross-process batching service); same conventions
SERVICE_SIGNATURES = {
    "pnn_service_run_backend": (ci, [ctypes.c_char_p, BACKEND, vp, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_long)]),
    "pnn_service_run": (ci, [ctypes.c_char_p, vp, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_long)]),
    "pnn_service_run_table": (ci, [ctypes.c_char_p, ctypes.c_char_p, ci, ctypes.c_float, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_long)]),
    "pnn_client_connect": (ci, [ctypes.POINTER(vp), ctypes.c_char_p]),
    "pnn_client_predict_pel": (ci, [vp, ci, f32p, f32p, i32p, ci]),
    "pnn_client_predict_f32": (ci, [vp, ci, f32p, f32p, f32p]),
    "pnn_client_cache_stats": (ci, [vp, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]),
    "pnn_client_arithmetic_tag": (ci, [vp, ci, ctypes.c_char_p, ctypes.c_size_t]),
    "pnn_client_close": (None, [vp]),
}

# name -> (restype, argtypes); also the list the symbol-export test walks.
SIGNATURES = {
    "pnn_create_empty": (ci, [ctypes.POINTER(vp), ctypes.c_float, ci]),
    "pnn_create": (ci, [ctypes.POINTER(vp), ctypes.c_char_p, ci, ctypes.c_float, ci]),
    "pnn_load_model_file": (ci, [vp, ctypes.c_char_p]),
    "pnn_load_model_params": (ci, [vp, ci, ci, f32p, ctypes.c_size_t]),
    "pnn_model_info": (ci, [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_long)]),
    "pnn_destroy": (None, [vp]),
    "pnn_last_error": (ctypes.c_char_p, [vp]),
    "pnn_mean": (ctypes.c_float, [vp]),
    "pnn_set_option": (ci, [vp, ctypes.c_char_p, ctypes.c_long]),
    "pnn_num_split_configs": (ci, []),
    "pnn_num_f32_configs": (ci, []),
    "pnn_check_range": (ci, [vp, vp, ctypes.POINTER(ctypes.c_long)]),
    "pnn_last_call_issued_flops": (ci, [vp, ctypes.POINTER(ctypes.c_double)]),
    "pnn_arithmetic_tag": (ci, [vp, ctypes.c_char_p, ctypes.c_size_t]),
    "pnn_host_alloc": (ci, [ctypes.POINTER(vp), ctypes.c_size_t]),
    "pnn_host_free": (None, [vp]),
    "pnn_streams_on_distinct_queues": (ci, [ctypes.POINTER(vp), ci]),
    "pnn_streams_release": (None, [ctypes.POINTER(vp), ci]),
    "pnn_cache_stats": (ci, [vp, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]),
    "pnn_predict_fc": (ci, [vp, ci, f32p, ci, f32p]),
    "pnn_predict_conv": (ci, [vp, ci, f32p, f32p, ci, f32p]),
    "pnn_predict_pel": (ci, [vp, ci, f32p, f32p, ci, i32p, ci]),
    "pnn_predict_f32_pel": (ci, [vp, ci, f32p, f32p, ci, f32p, i32p]),
    "pnn_extract_context": (ci, [i32p, f32p, f32p, u8p] + [ci] * 8 + [ctypes.c_float]),
    "pnn_parse_model_table": (ci, [ctypes.c_char_p, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci),
                                   ctypes.POINTER(ctypes.c_char_p), ci]),
    "pnn_predict_fc_device": (ci, [vp, ci, vp, ci, vp, vp]),
    "pnn_predict_conv_device": (ci, [vp, ci, vp, vp, ci, vp, vp]),
    "pnn_make_tb_desc": (ci, [ctypes.POINTER(TbDev), ctypes.c_int64, ctypes.c_int32, u8p, ci, ci, ci]),
    "pnn_gather_device": (ci, [vp, ci, ci, vp, ci, vp, ci, vp, ctypes.c_long, vp, ctypes.c_long, vp]),
    "pnn_predict_tbs_device": (ci, [vp, ci, vp, ci, vp, ci, vp, vp, vp]),
    "pnn_block_cost_device": (ci, [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp]),
    "pnn_predict_tbs_cost_device": (ci, [vp, ci, vp, vp, ci, vp, ci, ci, vp, vp, vp]),
    "pnn_last_call_stats": (ci, [vp, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ci)]),
    "pnn_launch_times": (ci, [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}

_lib = None
SKIP_TORCH = False      # set by processes that never touch torch (the batching service): skips its import, seconds on a cold box


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or `make -C %s/csrc`). There is no CPU fallback." % (LIB_PATH, _HERE))
        # PyTorch-ROCm bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both own
        # the GPU. Importing torch first makes the loader resolve our DT_NEEDED libamdhip64.so.7 to the
        # copy already mapped, so the library and torch share one runtime (streams, device pointers).
        if not SKIP_TORCH:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(SERVICE_SIGNATURES.items()):
            if os.environ.get("PNN_LIB_PATH") and not hasattr(L, name):
                continue                               # an A/B build of an older revision (tools/ab.sh): it lacks the newer entry points
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class PnnError(RuntimeError):
    pass


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().pnn_last_error(ctx)
        raise PnnError("libpnn_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
