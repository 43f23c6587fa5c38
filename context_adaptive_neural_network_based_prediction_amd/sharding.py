"""Multi-GPU decomposition of the batched PNN path: independent transform blocks are split contiguously over
ranks (one process per GPU), weights are replicated, and there is no collective on the data path.  The only
exchanges are the bench barrier / max-over-ranks clock and the optional result gather for a single consumer.
(the reference itself has no multi-device code.)"""


import os
import time


def rank_env(environ=None):
    """(rank, local_rank, world_size) as torch.distributed.run exports them; (0, 0, 1) for a plain process."""
    e = os.environ if environ is None else environ
    return int(e.get("RANK", "0")), int(e.get("LOCAL_RANK", "0")), int(e.get("WORLD_SIZE", "1"))


# True: the exchanges below run through the process group even when it has ONE rank (bench.py --force-dist: the only execution of the
# RCCL branch a one-GPU box allows -- communicator set-up, barrier, all-reduce, all-gather on the real device)
FORCE_COLLECTIVES = False


def _single(dist):
    return dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_COLLECTIVES)


def init_ranks(backend, device=None, force=False):
    """Joins the job's process group when WORLD_SIZE > 1 (backend "nccl" = RCCL on the GPUs, "gloo" on the CPU) and
    returns the torch.distributed module, or None for a single process.  Rendezvous defaults to 127.0.0.1.
    force: a group even for ONE rank, and every exchange of this module goes through it (FORCE_COLLECTIVES)."""
    global FORCE_COLLECTIVES
    rank, _, world = rank_env()
    if world <= 1 and not force:
        return None
    if force:
        FORCE_COLLECTIVES = True
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    kw = {"device_id": device} if device is not None else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def timed_steps(step, n_steps, sync, dist=None, device=None):
    """The bench contract's timed region: barrier + device synchronisation, EXACTLY n_steps calls of `step`, device
    synchronisation, and the job's time = the MAX over ranks (seconds, the same value on every rank).  The clock of a rank
    stops when ITS device is idle, before any exchange: the closing rendezvous is the max-reduction itself (an all-reduce
    every rank must reach), so its 30-100 us on RCCL never enter a region that may last only a few milliseconds.
    `sync` waits for this rank's device work (torch.cuda.synchronize on a GPU rank, a no-op on the CPU)."""
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed, dist, device)      # the closing barrier: every rank waits here for the slowest one


def count_distinct_devices(dist, local_rank, share=False):
    """How many different (host, device) pairs the ranks of the initialised group sit on -- gathered over the group itself, so
    a bench line that says N ranks were on N devices has been through N-way collectives (RCCL when the backend is "nccl")."""
    import socket
    objs = [None] * dist.get_world_size()
    dist.all_gather_object(objs, (socket.gethostname(), 0 if share else int(local_rank)))
    return len(set(objs))


def shard_bounds(n_items, rank, world_size):
    """[begin, end) of `rank`'s contiguous shard; the first n % world ranks take one extra item."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    base, extra = divmod(int(n_items), world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def max_over_ranks(seconds, dist=None, device=None):
    """The job's step time is the slowest rank's (bench contract)."""
    if _single(dist):
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_predictions(local, n_total, dist=None):
    """All ranks' [n_local, w, w] predictions concatenated in rank order on every rank (equal or ragged shards)."""
    if _single(dist):
        return local
    import torch
    world = dist.get_world_size()
    sizes = [e - b for b, e in (shard_bounds(n_total, r, world) for r in range(world))]
    biggest = max(sizes)                                  # all_gather wants equal shapes: pad ragged shards
    padded = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    outs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(outs, padded)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)


def cpu_budget(cgroup_root="/sys/fs/cgroup"):
    """CPUs this process may actually keep busy: its affinity mask, cut to the cgroup's CPU quota.  The GPU boxes show 256 cores and
    grant the job `cpu.max` = 16 CPUs: 64 threads there run in bursts between throttled periods -- CPU legs, encoder pools and
    services size themselves by THIS, never by os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open(os.path.join(cgroup_root, "cpu.max")).read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = int(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_quota_us")).read())
            per = int(open(os.path.join(cgroup_root, "cpu", "cpu.cfs_period_us")).read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def encodes_in_flight(n_encodes, n_services, budget=None):
    """How many codec processes a campaign keeps in flight.  An encoder behind the batching service is LATENCY-bound, not CPU-bound: it
    spends four fifths of its wall clock blocked on the socket (configs[3]: 24 encoders use 24 CPU-seconds over a 5.3 s wall), and the
    service batches better the more of them wait at once -- 14 in flight instead of 24 doubled the campaign's wall (10.4 against 5.3 s,
    round 5).  So the pool oversubscribes the CPUs the job may really use (the cgroup quota, not the cores the box shows) eight times,
    after one CPU per service for its I/O and launch threads; at least one encode per service, never more than there are."""
    b = cpu_budget() if budget is None else int(budget)
    return int(max(n_services, min(int(n_encodes), 8 * max(1, b - int(n_services)))))


def strong_shard(global_batch, rank, world_size):
    """Blocks of `rank` when ONE batch of `global_batch` blocks is split over the ranks (bench.py --scaling strong): shard_bounds,
    returned as (begin, count); every rank must get at least one block."""
    b, e = shard_bounds(global_batch, rank, world_size)
    if e <= b:
        raise ValueError("a batch of %d blocks cannot be split over %d ranks" % (global_batch, world_size))
    return b, e - b


# ---------------------------------------------------------------------------------------------------------------------
# Host-side placement of a rank next to its GPU, read from sysfs only (nothing here initialises HIP: it runs BEFORE the
# first HIP call of a rank, and in the self-launching parent of bench.py, which must never touch the GPU).
# ---------------------------------------------------------------------------------------------------------------------
KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of sysfs `local_cpulist`)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_render_minors(kfd_nodes=KFD_NODES):
    """DRM render minors of the GPUs in KFD topology order -- the order ROCr / HIP enumerate devices in (a node with
    simd_count > 0 is a GPU; CPU nodes have none)."""
    minors = []
    try:
        names = sorted(os.listdir(kfd_nodes), key=lambda n: int(n))
    except (OSError, ValueError):
        return minors
    for n in names:
        props = {}
        try:
            with open(os.path.join(kfd_nodes, n, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    props[k] = v
        except OSError:
            continue
        if int(props.get("simd_count", "0") or 0) > 0 and int(props.get("drm_render_minor", "0") or 0) > 0:
            minors.append(int(props["drm_render_minor"]))
    return minors


def visible_device_index(local_rank, environ=None):
    """Index into the physical enumeration of HIP device `local_rank` under HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES
    (plain integer lists only; anything else -> the identity)."""
    e = os.environ if environ is None else environ
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = e.get(var, "").strip()
        if v:
            try:
                ids = [int(x) for x in v.split(",") if x.strip() != ""]
                return ids[local_rank]
            except (ValueError, IndexError):
                return local_rank
    return local_rank


def gpu_sysfs_dir(local_rank, kfd_nodes=KFD_NODES, drm="/sys/class/drm"):
    """/sys/class/drm/renderD<minor>/device of HIP device `local_rank`, or None."""
    minors = gpu_render_minors(kfd_nodes)
    idx = visible_device_index(local_rank)
    if not 0 <= idx < len(minors):
        return None
    d = os.path.join(drm, "renderD%d" % minors[idx], "device")
    return d if os.path.isdir(d) else None


def bind_to_gpu_numa(local_rank, kfd_nodes=KFD_NODES, drm="/sys/class/drm"):
    """Pins this process to the host cores of the NUMA node its GPU hangs off (sysfs `local_cpulist`), so that the launch
    thread and the pinned staging buffers of a rank sit next to its device on a multi-socket node.  Best effort:
    returns the CPU list it bound to, or None when sysfs does not say (single-node boxes, containers)."""
    d = gpu_sysfs_dir(local_rank, kfd_nodes, drm)
    if d is None:
        return None
    try:
        with open(os.path.join(d, "local_cpulist")) as f:
            cpus = parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus if c in allowed]
        if not cpus or len(cpus) == len(allowed):
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError):
        return None


class GpuClockSampler(object):
    """Samples the GPU's shader clock and busy percentage from sysfs in a thread while a measurement runs (what
    `rocm-smi --showclocks --showuse` prints): hwmon `freq1_input` (Hz) when present, else the starred level of
    `pp_dpm_sclk`; `gpu_busy_percent`.  `summary()` -> {"sclk_mhz_mean", "sclk_mhz_min", "sclk_mhz_max", "busy_pct_mean",
    "power_w_mean", "samples", "source"} or None when nothing is readable."""

    def __init__(self, local_rank=0, period_s=0.02):
        import glob
        self.period = period_s
        self.dir = gpu_sysfs_dir(local_rank)
        self.freq_file, self.source = None, None
        if self.dir:
            hw = sorted(glob.glob(os.path.join(self.dir, "hwmon", "hwmon*", "freq1_input")))
            if hw:
                self.freq_file, self.source = hw[0], "hwmon freq1_input"
            elif os.path.exists(os.path.join(self.dir, "pp_dpm_sclk")):
                self.freq_file, self.source = os.path.join(self.dir, "pp_dpm_sclk"), "pp_dpm_sclk (current level)"
        self.busy_file = os.path.join(self.dir, "gpu_busy_percent") if self.dir else None
        pw = sorted(glob.glob(os.path.join(self.dir, "hwmon", "hwmon*", "power1_input"))) if self.dir else []
        self.power_file = pw[0] if pw else None       # microwatts
        self.power_cap_w = None
        if pw:
            try:
                with open(os.path.join(os.path.dirname(pw[0]), "power1_cap")) as f:
                    self.power_cap_w = float(f.read().strip()) / 1e6
            except (OSError, ValueError):
                pass
        self.mhz, self.busy, self.watts = [], [], []
        self._stop = None
        self._thread = None

    def _read_mhz(self):
        try:
            with open(self.freq_file) as f:
                text = f.read()
            if self.source.startswith("hwmon"):
                return float(text.strip()) / 1e6
            for line in text.splitlines():
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        return None

    def _loop(self):
        while not self._stop.is_set():
            if self.freq_file:
                v = self._read_mhz()
                if v is not None:
                    self.mhz.append(v)
            if self.busy_file:
                try:
                    with open(self.busy_file) as f:
                        self.busy.append(float(f.read().strip()))
                except (OSError, ValueError):
                    pass
            if self.power_file:
                try:
                    with open(self.power_file) as f:
                        self.watts.append(float(f.read().strip()) / 1e6)
                except (OSError, ValueError):
                    pass
            self._stop.wait(self.period)

    def __enter__(self):
        import threading
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join()
        return False

    def summary(self):
        if not self.mhz and not self.busy:
            return None
        out = {"samples": max(len(self.mhz), len(self.busy)), "source": self.source}
        if self.mhz:
            out.update(sclk_mhz_mean=sum(self.mhz) / len(self.mhz), sclk_mhz_min=min(self.mhz), sclk_mhz_max=max(self.mhz))
        if self.busy:
            out["busy_pct_mean"] = sum(self.busy) / len(self.busy)
        if self.watts:
            out["power_w_mean"] = sum(self.watts) / len(self.watts)
            out["power_cap_w"] = self.power_cap_w
        return out
