"""Cross-process batching service (include/pnn_service.h).

One server process owns the GPU context; encoder processes (HM links the C client stub, Python tools use `Client`)
send their single-block requests over a Unix-domain socket and the server coalesces the requests of one width that are
pending at the same time into one batched `pnn_predict_pel` call.

    python -m context_adaptive_neural_network_based_prediction_amd.service --socket /tmp/pnn.sock --table single.txt

`serve_in_thread` runs the same loop in a background thread of this process (tests, notebooks).
"""
import argparse
import ctypes
import threading

import numpy as np

from . import _lib


class Server:
    """The C server loop (`pnn_service_run` / `pnn_service_run_backend`) in a background thread."""

    def __init__(self, socket_path, ctx=None, backend=None, table=None, max_batch=256, window_us=0, pair=0,
                 mean=117.8952234192841, device=0):
        if sum(x is not None for x in (ctx, backend, table)) != 1:
            raise ValueError("give exactly one of `ctx` (a pnn context handle), `backend` (a Python callable) and `table` (a model table: "
                             "the server then owns five contexts, one per width, each on its own worker thread)")
        self._L = _lib.lib()
        self._stop = ctypes.c_int(0)
        self._stats = (ctypes.c_long * 4)()
        self.rc = None
        path = socket_path.encode()
        if backend is not None:
            def trampoline(user, width, above, left, n, dst, out_f32):
                w2 = width * width
                na = (3 if left else 5) * w2
                a = np.ctypeslib.as_array(above, shape=(n, na))
                l = np.ctypeslib.as_array(left, shape=(n, 2 * w2)) if left else None
                try:
                    # a stand-in backend returns the Pel blocks, or (Pel blocks, float predictions)
                    res = backend(width, a, l)
                    pel, f32 = res if isinstance(res, tuple) else (res, np.asarray(res, np.float32))
                    if dst:
                        np.ctypeslib.as_array(dst, shape=(n, width, width))[...] = pel
                    if out_f32:
                        np.ctypeslib.as_array(out_f32, shape=(n, width, width))[...] = f32
                    return 0
                except Exception:                      # nothing may propagate into the C loop
                    return -1
            self._cb = _lib.BACKEND(trampoline)       # keep the callback object alive
            target = lambda: self._L.pnn_service_run_backend(path, self._cb, None, max_batch, window_us, ctypes.byref(self._stop), self._stats)
        elif table is not None:
            target = lambda: self._L.pnn_service_run_table(path, table.encode(), pair, ctypes.c_float(mean), device, max_batch, window_us,
                                                           ctypes.byref(self._stop), self._stats)
        else:
            target = lambda: self._L.pnn_service_run(path, ctx, max_batch, window_us, ctypes.byref(self._stop), self._stats)

        def run():
            self.rc = target()
        self._thread = threading.Thread(target=run, daemon=True)
        self._thread.start()

    def stop(self):
        """Stops the loop and returns {"requests", "backend_calls", "largest_batch", "clients"}."""
        self._stop.value = 1
        self._thread.join()
        s = self._stats
        return {"requests": s[0], "backend_calls": s[1], "largest_batch": s[2], "clients": s[3]}


def serve_in_thread(socket_path, **kw):
    return Server(socket_path, **kw)


class Client:
    """`pnn_client_*`: what an encoder process uses instead of a GPU context."""

    def __init__(self, socket_path, retries=200):
        import time
        self._L = _lib.lib()
        self._c = ctypes.c_void_p()
        for _ in range(retries):                       # the server may still be binding its socket
            if self._L.pnn_client_connect(ctypes.byref(self._c), socket_path.encode()) == 0:
                return
            time.sleep(0.01)
        raise ConnectionError("no PNN service at %s" % socket_path)

    def predict_pel(self, width, above, left=None):
        a = np.ascontiguousarray(above, np.float32)
        l = None if left is None else np.ascontiguousarray(left, np.float32)
        dst = np.empty((width, width), np.int32)
        rc = self._L.pnn_client_predict_pel(self._c, width, a.ctypes.data_as(_lib.f32p), None if l is None else l.ctypes.data_as(_lib.f32p),
                                            dst.ctypes.data_as(_lib.i32p), width)
        if rc != 0:
            raise _lib.PnnError("service returned %d" % rc)
        return dst

    def predict_f32(self, width, above, left=None):
        """The float prediction, as `Session::Run` returns it (mean not re-added)."""
        a = np.ascontiguousarray(above, np.float32)
        l = None if left is None else np.ascontiguousarray(left, np.float32)
        out = np.empty((width, width), np.float32)
        rc = self._L.pnn_client_predict_f32(self._c, width, a.ctypes.data_as(_lib.f32p), None if l is None else l.ctypes.data_as(_lib.f32p),
                                            out.ctypes.data_as(_lib.f32p))
        if rc != 0:
            raise _lib.PnnError("service returned %d" % rc)
        return out

    def arithmetic_tag(self, width):
        """`pnn_arithmetic_tag` of the server context that serves `width` (compare with the decoder's: INTEGRATION.md)."""
        buf = ctypes.create_string_buffer(256)
        rc = self._L.pnn_client_arithmetic_tag(self._c, width, buf, 256)
        if rc != 0:
            raise _lib.PnnError("service returned %d" % rc)
        return buf.value.decode()

    def close(self):
        if self._c:
            self._L.pnn_client_close(self._c)
            self._c = ctypes.c_void_p()


def die_with_parent():
    """SIGTERM for this process when the one that started it dies (PR_SET_PDEATHSIG), so a campaign killed from outside -- a `timeout`
    around bench.py sends SIGTERM, which skips `finally` -- leaves no service alive on the GPU.  Called first thing in main(), from
    the service itself rather than from a preexec_fn of its parent; PNN_SERVICE_PARENT = the starter's pid closes the window in which
    the parent died before the call."""
    import os
    import signal
    parent = os.environ.get("PNN_SERVICE_PARENT")
    if not parent:
        return
    try:
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM, 0, 0, 0)      # PR_SET_PDEATHSIG
    except OSError:
        return
    if os.getppid() != int(parent):
        raise SystemExit("pnn service: the process that started it (%s) is gone" % parent)


def main():
    die_with_parent()
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--socket", required=True)
    ap.add_argument("--table", required=True, help="model table (width,is_pair,channel,path per line)")
    ap.add_argument("--pair", type=int, default=0)
    ap.add_argument("--mean", type=float, default=117.8952234192841)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--max-batch", type=int, default=256)
    ap.add_argument("--window-us", type=int, default=0,
                    help="extra wait for stragglers before an idle worker dispatches; 0 (default) measured fastest: requests that arrive "
                         "while a batch is on the GPU form the next one by themselves")
    args = ap.parse_args()
    import signal
    import sys
    import time
    t_main = time.time()
    _lib.SKIP_TORCH = True                            # this process owns a GPU context through libpnn_hip.so alone
    # five contexts (one per width, each with its own worker thread and stream) inside the C server; a block gets the same
    # prediction whatever batch it travels in (one summation order at every batch size is the library's default)
    import os
    try:                                              # a stale file of a killed earlier run must not read as "bound"
        os.unlink(args.socket)
    except OSError:
        pass
    srv = Server(args.socket, table=args.table, pair=args.pair, mean=args.mean, device=args.device, max_batch=args.max_batch,
                 window_us=args.window_us)
    done = threading.Event()
    for sig in (signal.SIGTERM, signal.SIGINT):       # handlers run in this (main) thread; the C loop runs in the server's thread
        signal.signal(sig, lambda *_: done.set())

    # the stale path is gone, so the file's existence now means THIS server bound it (bind and listen are back to back in the C
    # loop, after the models are loaded); no probing connection: it would count as a client in the server's statistics
    t0 = time.time()
    while srv.rc is None and not os.path.exists(args.socket) and time.time() - t0 < 300:
        time.sleep(0.02)
    if srv.rc is not None or not os.path.exists(args.socket):
        raise SystemExit("pnn service: cannot start (%s): %s" % (srv.rc, (_lib.lib().pnn_last_error(None) or b"").decode()))
    print("pnn service: listening on %s" % args.socket, flush=True)
    if os.environ.get("PNN_SERVICE_DEBUG"):
        sys.stderr.write("[pnn-service] start-up: %.2f s from main() to listening (five contexts created, five models loaded)\n" % (time.time() - t_main))
    while not done.is_set() and srv.rc is None:
        time.sleep(0.05)
    st = srv.stop()
    print("pnn service: %(requests)d requests in %(backend_calls)d batched calls (largest batch %(largest_batch)d), %(clients)d clients" % st,
          flush=True)


if __name__ == "__main__":
    main()
