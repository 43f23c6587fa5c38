"""BASELINE.json configs[3] / configs[4] at their stated picture counts, on the hardware that is there.

    configs[3]  "Kodak 24-image first-frame encode via hm_16_15_substitution, CTU-row batched PNN on 1 GPU"
                (comparing_rate_distortion.py:477-561, hevc/running.py:94-127): 24 pictures of 768 x 512, luminance only
    configs[4]  "BSDS100 luminance encode via hm_16_15_switch, all widths 4-64, CTU rows sharded over 8 x MI355X"
                (hevc/unifiedloading.py:71-76 crops 481 x 321 to 480 x 320): 100 pictures of 480 x 320

The datasets are not in the reference checkout (download scripts, no network) and neither are the trained production models:
seeded synthetic pictures (run_hm.make_frame) and seeded random-init models in the reference's five architectures stand in.
Inside one encode the PNN calls are serially dependent, so the data-parallel axis is ACROSS encodes: one
batching-service process per visible device, encodes dealt round-robin, all of them in flight up to the host-core count --
replicas, no collective.  `hm_16_15_regular` (the reference's stock HM-16.15, CPU only) encodes the same
pictures beside it as the yardstick.

    python tools/hm/campaign.py --config kodak [--devices 0,1,...] [--pictures 24] [--in-flight N]

Prints one JSON object: wall clock of all encodes + decodes, HM's own `Total Time` (CPU seconds, summed), PNN calls per width,
cache rates, the services' request / call / batch statistics, PNN blocks per second through the services, and whether every
decoded picture equals its encoder's reconstruction.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import run_hm  # noqa: E402

CONFIGS = {
    "kodak": {"variant": "substitution", "pictures": 24, "width": 768, "height": 512,
              "baseline": "configs[3]: Kodak 24-image first-frame encode via hm_16_15_substitution"},
    "bsds": {"variant": "switch", "pictures": 100, "width": 480, "height": 320,
             "baseline": "configs[4]: BSDS100 luminance encode via hm_16_15_switch, all widths 4-64"},
}


def binaries_present(variants=("substitution", "switch", "regular")):
    try:
        for v in variants:
            run_hm.exe(v, "Encoder"), run_hm.exe(v, "Decoder")
        return True
    except FileNotFoundError:
        return False


def np_cat(above, left):
    import numpy as np
    return np.concatenate([above.reshape(-1), left.reshape(-1)])       # sets/common.py:466-473: [above | left], both row-major


def _service_stats(stdout_text, stderr_text):
    out = {}
    m = re.search(r"(\d+) requests in (\d+) batched calls \(largest batch (\d+)\), (\d+) clients", stdout_text)
    if m:
        req, calls, largest, clients = map(int, m.groups())
        out = {"requests": req, "backend_calls": calls, "largest_batch": largest, "mean_batch": round(req / max(calls, 1), 3), "clients": clients}
    workers = {}
    for m in re.finditer(r"worker (\d+): ([0-9.]+) s inside the backend, (\d+) calls \(([0-9.]+) us each\), (\d+) requests"
                         r"(?: \([0-9.]+ per call\); per request ([0-9.]+) us queued before its batch is taken, ([0-9.]+) us from last byte in to reply out)?", stderr_text):
        k = int(m.group(1))
        workers[(4, 8, 16, 32, 64)[k]] = {"backend_busy_s": float(m.group(2)), "calls": int(m.group(3)), "us_per_call": float(m.group(4)),
                                          "requests": int(m.group(5))}
        if m.group(6):
            workers[(4, 8, 16, 32, 64)[k]].update(queued_us_per_request=float(m.group(6)), in_server_us_per_request=float(m.group(7)))
    if workers:
        out["per_width"] = workers
    return out


def spot_check_contexts(width, k=6, seed=31):
    """k seeded uint8-valued contexts of one width, mean-subtracted (what HM hands to Session::Run)."""
    import numpy as np
    rng = np.random.RandomState(seed + width)
    above = rng.randint(0, 256, (k, width, 3 * width)).astype(np.float32) - np.float32(run_hm.MEAN)
    left = rng.randint(0, 256, (k, 2 * width, width)).astype(np.float32) - np.float32(run_hm.MEAN)
    above[1, :, 2 * width:] = 0.0                      # an unavailable above-right unit row / below-left rows, as the gather leaves them
    left[2, width:, :] = 0.0
    return above, left


def _cgroup_cpu():
    """usage_usec / nr_throttled of this job's cgroup (v2), or None: what the campaign costs the HOST -- the GPU boxes grant 16 CPUs."""
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat").read().strip().splitlines())}
    except (OSError, ValueError):
        return None


def _cpu_quota():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(float(q) / float(per), 2)
    except (OSError, ValueError):
        return None


def _thread_cpu_s(pid):
    """{thread name: CPU seconds} of a live process, threads of one name summed (the service names its I/O threads and width workers)."""
    out = {}
    try:
        for tid in os.listdir("/proc/%d/task" % pid):
            try:
                f = open("/proc/%d/task/%s/stat" % (pid, tid)).read()
                name = f[f.index("(") + 1:f.rindex(")")]
                g = f.rsplit(")", 1)[1].split()
                out[name] = round(out.get(name, 0.0) + (int(g[11]) + int(g[12])) / os.sysconf("SC_CLK_TCK"), 2)
            except (OSError, ValueError, IndexError):
                pass
    except OSError:
        pass
    return out


def _proc_cpu_s(pid):
    """user + system CPU seconds of a live process, all its threads."""
    try:
        f = open("/proc/%d/stat" % pid).read().rsplit(")", 1)[1].split()
        return (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
    except (OSError, ValueError, IndexError):
        return 0.0


ARITHMETIC = {"f32": "0", "split": "1"}               # value of PNN_PRECISION in the service processes (csrc/pnn_ctx.h: opt_precision)


def run_campaign(config, work, devices=(0,), pictures=None, in_flight=None, qp=32, seed=1, yardstick=True, timeout=1800, picture_set="synthetic",
                 backend="gpu", spot_check=False, cpu_threads=None, arithmetic=None):
    """backend "gpu": one batching service per device (the product).  backend "cpu": ONE service whose backend answers from the CPU
    oracle (tools/hm/cpu_pnn_service.py) -- the reference's route, PNN inference on host cores, as bench.py's cpu_baseline leg.
    picture_set "natural": windows of the natural fixtures, widths 4 / 8 on the reference's trained (convolutional) checkpoints.
    arithmetic "f32": the services compute on the reference's IEEE-float32 arithmetic (Session::Run in float32, TComPrediction.cpp:572-579,
    601-608), "split": on the split-f16 mode, None: the library's default (f32 since round 5).
    spot_check: before the services stop, a client asks each of them for the predictions of seeded contexts of every width; they come
    back under "_spot_check" (numpy arrays, popped by the callers that serialise the record) for the tests to hold against the oracle."""
    cfg = CONFIGS[config]
    n = int(pictures or cfg["pictures"])
    h, w, variant = cfg["height"], cfg["width"], cfg["variant"]
    if backend == "cpu":
        devices = [0]
    if not in_flight:
        # by the CPUs the job may really use (cgroup quota), not by the cores the box shows: sharding.encodes_in_flight
        from context_adaptive_neural_network_based_prediction_amd import sharding
        in_flight = sharding.encodes_in_flight(n, len(devices))
    in_flight = int(in_flight)
    os.makedirs(work, exist_ok=True)
    natural = picture_set == "natural"
    table, mean_path = run_hm.make_models(os.path.join(work, "models"), trained_small=natural)
    frames = run_hm.natural_frames(h, w, n, seed) if natural else [run_hm.make_frame(h, w, seed + j) for j in range(n)]
    servers, socks, logs = [], [], []
    t_start = time.time()
    for k, dev in enumerate(devices):
        sock = os.path.join(work, "pnn%d.sock" % k)
        env = dict(os.environ, PNN_SERVICE_DEBUG="1")
        if arithmetic is not None:
            env["PNN_PRECISION"] = ARITHMETIC[arithmetic]
        log = open(os.path.join(work, "service%d.err" % k), "w+")      # a file, not a pipe: a chatty service must never block on a full pipe
        logs.append(log)
        if backend == "cpu":
            cmd = [sys.executable, os.path.join(HERE, "cpu_pnn_service.py"), "--socket", sock, "--table", table, "--threads", str(cpu_threads or 0)]
        else:
            cmd = [sys.executable, "-m", "context_adaptive_neural_network_based_prediction_amd.service", "--socket", sock,
                   "--table", table, "--device", str(dev), "--max-batch", "256", "--window-us", "0"]
        # (the services ask for SIGTERM on their parent's death themselves, service.die_with_parent: no preexec_fn -- Python code between
        # fork and exec of a process whose runtime threads are alive can deadlock)
        env["PNN_SERVICE_PARENT"] = str(os.getpid())
        servers.append(subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=log, text=True, start_new_session=True))
        socks.append(sock)
    results, stats, spot, tags = [], [], {}, []
    try:
        import select
        for srv, log in zip(servers, logs):            # the start-up line, with a deadline
            ready, _, _ = select.select([srv.stdout], [], [], 300.0)
            line = srv.stdout.readline() if ready else ""
            if "listening" not in line:
                log.seek(0)
                raise RuntimeError("batching service did not start: %r %s" % (line, log.read()[-2000:]))
        t_up = time.time() - t_start
        # The arithmetic contract: every service is asked for the tag of each width's context before the first encode;
        # the encoders run behind it, and every DECODER gets that tag as $PNN_EXPECT_TAG -- one that would be answered on another
        # arithmetic or summation order refuses to start (run_hm.ArithmeticMismatch) instead of decoding a picture that drifts.
        from context_adaptive_neural_network_based_prediction_amd import service as svc_mod
        for k, sock in enumerate(socks):
            cl = svc_mod.Client(sock)
            per_width = {wd: cl.arithmetic_tag(wd) for wd in (4, 8, 16, 32, 64)}
            cl.close()
            if len(set(per_width.values())) != 1:
                raise RuntimeError("service %d serves its widths on different arithmetics: %r" % (k, per_width))
            tags.append(per_width[4])

        def job(j):
            return run_hm.encode_decode(variant, frames[j], qp, table, mean_path, os.path.join(work, "enc"), tag=str(j),
                                        env={"PNN_SERVICE_SOCKET": socks[j % len(socks)]}, timeout=timeout, expect_tag=tags[j % len(socks)])
        cg0 = _cgroup_cpu()
        t0 = time.time()
        with ThreadPoolExecutor(in_flight) as ex:
            results = list(ex.map(job, range(n)))
        wall = time.time() - t0
        cg1 = _cgroup_cpu()
        host_cpu = {"service_cpu_s": round(sum(_proc_cpu_s(srv.pid) for srv in servers), 2),     # since the services started (their start-up included)
                    "service_threads_cpu_s": _thread_cpu_s(servers[0].pid)}
        if cg0 and cg1:
            host_cpu.update({"all_processes_cpu_s": round(cg1["usage_usec"] / 1e6 - cg0["usage_usec"] / 1e6, 2), "cpu_quota": _cpu_quota(),
                             "times_throttled": cg1["nr_throttled"] - cg0["nr_throttled"]})
        if spot_check:
            from context_adaptive_neural_network_based_prediction_amd import service as svc
            for k, sock in enumerate(socks):
                cl = svc.Client(sock)
                for wd in (4, 8, 16, 32, 64):
                    above, left = spot_check_contexts(wd)
                    fc = run_hm.model_params(wd, trained_small=natural)[1]
                    pel = [cl.predict_pel(wd, np_cat(above[i], left[i]) if fc else above[i], None if fc else left[i]) for i in range(above.shape[0])]
                    spot[(k, wd)] = {"above": above, "left": left, "pel": pel, "is_fc": fc}
                cl.close()
    finally:
        for srv in servers:
            try:
                os.killpg(srv.pid, 15)                 # the service and anything it started (its own session: see Popen above)
            except OSError:
                pass
        for srv, log in zip(servers, logs):
            try:
                so, _ = srv.communicate(timeout=30)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(srv.pid, 9)
                except OSError:
                    pass
                so, _ = srv.communicate()
            log.seek(0)
            se = log.read()
            log.close()
            for line in se.splitlines():               # whatever the service complained about goes to this process's stderr, not only into the statistics
                if line.startswith("[pnn-service]") and "worker" not in line and "start-up" not in line:
                    sys.stderr.write(line + "\n")
            stats.append(_service_stats(so or "", se or ""))
    calls = {}
    for side in ("enc_pnn", "dec_pnn"):
        for wd in (4, 8, 16, 32, 64):
            runs = sum(r[side].get(wd, {}).get("runs", 0) for r in results)
            hits = sum(r[side].get(wd, {}).get("cache_hits", 0) for r in results)
            calls.setdefault(side, {})[wd] = {"session_run_calls": runs, "cache_hits": hits, "cache_rate": round(hits / max(runs, 1), 4)}
    requests = sum(s.get("requests", 0) for s in stats)
    backend_calls = sum(s.get("backend_calls", 0) for s in stats)
    busy = {}
    for s in stats:
        for wd, v in s.get("per_width", {}).items():
            b = busy.setdefault(wd, {"backend_busy_s": 0.0, "calls": 0, "requests": 0})
            b["backend_busy_s"] += v["backend_busy_s"]; b["calls"] += v["calls"]; b["requests"] += v["requests"]
            for key in ("queued_us_per_request", "in_server_us_per_request"):
                if key in v:
                    b[key] = b.get(key, 0.0) + v[key] * v["requests"]
    for wd, b in busy.items():
        b["blocks_per_s_inside_backend"] = round(b["requests"] / b["backend_busy_s"], 1) if b["backend_busy_s"] > 0 else None
        b["mean_batch"] = round(b["requests"] / max(b["calls"], 1), 3)
        b["backend_busy_s"] = round(b["backend_busy_s"], 3)
        for key in ("queued_us_per_request", "in_server_us_per_request"):
            if key in b:
                b[key] = round(b[key] / max(b["requests"], 1), 1)
    out = {
        "config": cfg["baseline"], "variant": "hm_16_15_" + variant, "pictures": n, "picture_size": "%dx%d 4:0:0" % (w, h), "qp": qp,
        "picture_set": picture_set, "pnn_backend": backend,
        "arithmetic": "f32 (CPU oracle)" if backend == "cpu" else (arithmetic or {"0": "f32", "1": "split"}.get(os.environ.get("PNN_PRECISION", "0"), "f32")),
        "data": ("windows of the natural fixtures (tests/golden/natural_luma.npz); widths 4 / 8 on the reference's trained convolutional checkpoints, 16 / 32 / 64 "
                 "seeded random init" if natural else "seeded synthetic pictures + seeded random-init models") + " (Kodak / BSDS and the trained production models "
                "are not in the reference checkout)",
        "devices": list(devices), "services": len(devices), "encodes_in_flight": in_flight, "host_cores": os.cpu_count(), "cpu_budget": _cpu_quota() or os.cpu_count(),
        "wall_s_all_encodes_and_decodes": round(wall, 3), "service_start_s": round(t_up, 3), "host_cpu": host_cpu,
        "pictures_per_s": round(n / wall, 3),
        "hm_total_time_s_sum": {"encoders": round(sum(r["enc_total_time_s"] or 0 for r in results), 2),
                                "decoders": round(sum(r["dec_total_time_s"] or 0 for r in results), 2),
                                "note": "HM's clock(): CPU seconds of the process; time blocked on the service socket is not in it"},
        "enc_wall_s": {"mean": round(sum(r["enc_wall_s"] for r in results) / n, 3), "max": max(r["enc_wall_s"] for r in results)},
        "dec_wall_s": {"mean": round(sum(r["dec_wall_s"] for r in results) / n, 3), "max": max(r["dec_wall_s"] for r in results)},
        "bits_total": int(sum(r["bits"] for r in results)), "psnr_rec_db_mean": round(sum(r["psnr_rec_db"] for r in results) / n, 3),
        "pnn_calls": calls,
        "service": {"requests": requests, "backend_calls": backend_calls, "mean_batch": round(requests / max(backend_calls, 1), 3),
                    "largest_batch": max([s.get("largest_batch", 0) for s in stats] or [0]), "per_service": stats, "per_width": busy,
                    "pnn_blocks_per_s_over_the_wall": round(requests / wall, 1)},
        "every_decode_equals_its_encoder": bool(all(r["decoder_equals_encoder"] and not r["decoder_hash_error"] for r in results)),
        # pnn_arithmetic_tag of what answered the encoders (asked through each service's socket before the first encode) and of what
        # answered the decoders: here the same services, and every decoder ran with $PNN_EXPECT_TAG = its encoder's tag (it refuses to
        # start on a mismatch, include/pnn_tf_compat.h)
        "arithmetic_tags": {"encoder_side": tags, "decoder_side": tags, "decoders_checked_expect_tag": True},
    }
    if yardstick and binaries_present(("regular",)):
        def ref_job(j):
            return run_hm.encode_decode("regular", frames[j], qp, None, None, os.path.join(work, "reg"), tag=str(j), timeout=timeout)
        t0 = time.time()
        with ThreadPoolExecutor(in_flight) as ex:
            reg = list(ex.map(ref_job, range(n)))
        rw = time.time() - t0
        out["yardstick_hm_16_15_regular"] = {
            "what": "the reference's stock HM-16.15 (no PNN), CPU only, the same pictures, the same number in flight",
            "wall_s_all_encodes_and_decodes": round(rw, 3), "pictures_per_s": round(n / rw, 3),
            "hm_total_time_s_sum_encoders": round(sum(r["enc_total_time_s"] or 0 for r in reg), 2),
            "enc_wall_s_mean": round(sum(r["enc_wall_s"] for r in reg) / n, 3), "bits_total": int(sum(r["bits"] for r in reg)),
            "every_decode_equals_its_encoder": bool(all(r["decoder_equals_encoder"] for r in reg))}
        out["wall_vs_regular"] = round(wall / rw, 3)
    if spot_check:
        out["_spot_check"] = spot
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", default="kodak", choices=sorted(CONFIGS))
    ap.add_argument("--devices", default="0")
    ap.add_argument("--pictures", type=int, default=0)
    ap.add_argument("--in-flight", type=int, default=0)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--picture-set", default="synthetic", choices=["synthetic", "natural"])
    ap.add_argument("--backend", default="gpu", choices=["gpu", "cpu"])
    ap.add_argument("--arithmetic", default=None, choices=sorted(ARITHMETIC), help="arithmetic of the GPU services (default: the library's = f32)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hm_campaign"))
    args = ap.parse_args()
    import shutil
    import tempfile
    work = tempfile.mkdtemp(prefix="hm_campaign_")
    try:
        res = run_campaign(args.config, work, [int(d) for d in args.devices.split(",")], args.pictures or None, args.in_flight or None, args.qp,
                           picture_set=args.picture_set, backend=args.backend, arithmetic=args.arithmetic)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
