"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The reference is single-process ``nn.DataParallel`` (/root/reference/tools/train_3d.py:140):
scatter batch -> broadcast all parameters -> threads -> gather -> reduce-add grads, all through
GPU 0.  Here frames are sharded by rank (the path has no cross-frame dependency, SURVEY.md
§8(e)): inference needs NO collective; training all-reduces gradients only (DDP buckets over
RCCL/xGMI, overlapped with backward).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def env_world():
    """(rank, world, local_rank) from torchrun's environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


_DATA_GROUP = None            # set by init_split: the RCCL group gradients travel on (None: the default group)


def init(backend: str | None = None, device: torch.device | None = None):
    rank, world, _ = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


class _StdoutToStderr:
    """File-descriptor-level redirect of stdout to stderr while a gloo group connects: the library announces "[Gloo] Rank r is
    connected to n peer ranks" on the process's STDOUT, and bench.py's contract is that rank 0 prints exactly one line there."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def init_split(data_backend: str = "nccl", single_rank: bool = False):
    """bench.py's process groups: the CONTROL plane (barriers around timed regions, max / sum of a few host numbers) is the
    default group on gloo - host memory, TCP on 127.0.0.1, nothing on the GPU's queues - and the DATA plane (DDP's gradient
    buckets) is a second group on ``data_backend`` ("nccl" = RCCL over xGMI), created lazily by its first collective.  The
    forward-only path has no data-path collective, so its number depends on RCCL in no way: a
    communicator that cannot be built fails the train leg (an error entry in the line), not the job.  ``single_rank``
    builds the two groups at world size 1 too (tests/test_gpu_rccl_single_rank.py: the layout on the real library).
    Returns (rank, world)."""
    global _DATA_GROUP
    rank, world, _ = env_world()
    if (world > 1 or single_rank) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        with _StdoutToStderr():
            dist.init_process_group("gloo")
            _DATA_GROUP = dist.new_group(backend=data_backend)
            dist.barrier()                          # the gloo mesh connects (and says so) here, not inside a timed region
            if data_backend == "gloo":
                dist.barrier(group=_DATA_GROUP)
    return rank, world


def shutdown():
    """destroy the process groups of init / init_split (no-op when none exists); never raises - a peer may be gone already"""
    global _DATA_GROUP
    _DATA_GROUP = None
    try:
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass


def data_group():
    """the group gradients are reduced on (None = the default group)"""
    return _DATA_GROUP


def data_ranks_seen(device=None) -> int:
    """all-reduce(SUM) of a one on the DATA group: how many ranks the communicator gradients travel on really spans.
    On "nccl" this is the first collective of the RCCL communicator (it is built here); proof for a bench line that RCCL
    saw N ranks, and the place a broken fabric shows up before the train leg depends on it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 1
    group = _DATA_GROUP
    on_host = (dist.get_backend(group) if group is not None else dist.get_backend()) == "gloo" and \
        (device is None or device.type != "cuda")
    t = torch.ones(1, dtype=torch.float32, device="cpu" if on_host or device is None else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    if t.is_cuda:
        torch.cuda.synchronize(t.device)
    return int(round(float(t.item())))


def allreduce_alone_ms(nbytes: int, device, bucket_bytes: int = 32 << 20, reps: int = 5, warmup: int = 2) -> float:
    """milliseconds to all-reduce `nbytes` of fp32 gradients in DDP-sized buckets on the data group with NOTHING else
    running (no backward to hide behind): the yardstick for how much of the collective a train step overlaps"""
    import time
    n = max(1, int(nbytes) // 4)
    per = max(1, bucket_bytes // 4)
    buckets = [torch.zeros(min(per, n - i), dtype=torch.float32, device=device) for i in range(0, n, per)]

    def once():
        works = [dist.all_reduce(b, group=_DATA_GROUP, async_op=True) for b in buckets]
        for w in works:
            w.wait()
    for _ in range(warmup):
        once()
    barrier_sync(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    barrier_sync(device)
    return 1e3 * max_over_ranks(time.perf_counter() - t0, device) / reps


def _host_group() -> bool:
    """is the default group a host-memory one (gloo)?  Then the few numbers the timing rule exchanges stay on the host."""
    try:
        return dist.get_backend() == "gloo"
    except Exception:
        return False


def shard_frames(num_frames: int, rank: int, world: int) -> List[int]:
    """DistributedSampler semantics without padding: rank r owns frames r, r+world, ..."""
    return list(range(rank, num_frames, world))


def shard_batch(tensors: Sequence[torch.Tensor], rank: int, world: int):
    """slice dim 0 of every tensor the way ``shard_frames`` assigns frames"""
    return [t[rank::world] for t in tensors]


def max_over_ranks(value: float, device=None) -> float:
    """bench timing rule: the job takes as long as its slowest rank"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None and not _host_group() else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_predictions(pred: torch.Tensor, num_frames: int | None = None) -> torch.Tensor:
    """all-gather per-rank predictions (B_r, ...) back into frame order (evaluation only).  Shards may be uneven
    (``shard_frames`` does not pad: with 7 frames on 2 ranks rank 0 holds 4, rank 1 holds 3): counts are exchanged
    first, every rank pads to the largest shard, and the padding is dropped while de-interleaving."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return pred
    world = dist.get_world_size()
    n = torch.tensor([pred.shape[0]], dtype=torch.int64, device=pred.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    padded = pred.contiguous()
    if padded.shape[0] < m:
        padded = torch.cat([padded, padded.new_zeros((m - padded.shape[0],) + tuple(pred.shape[1:]))], 0)
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    total = sum(counts) if num_frames is None else int(num_frames)
    out = pred.new_empty((total,) + tuple(pred.shape[1:]))
    for r in range(world):                          # rank r holds frames r, r+world, ...
        out[r::world][:counts[r]] = parts[r][:counts[r]]
    return out


def sum_over_ranks(*values: float, device=None):
    """all-reduce(SUM) of a few python numbers (evaluation counters: every rank must score the WHOLE set)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tuple(float(v) for v in values)
    t = torch.tensor(values, dtype=torch.float64, device=device if device is not None and not _host_group() else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(float(v) for v in t.tolist())


def barrier_sync(device=None):
    """barrier + device synchronise: both sides of a timed region (bench.py contract)"""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step, steps: int, warmup: int, device=None):
    """bench.py's timing rule as a function: `warmup` untimed steps, then EXACTLY `steps` steps between two
    barrier+synchronise points; returns (seconds of the slowest rank, last step's output)."""
    import time
    out = None
    for _ in range(warmup):
        out = step()
    barrier_sync(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier_sync(device)
    return max_over_ranks(time.perf_counter() - t0, device), out


def timed_steps_host(step, steps: int, device=None):
    """timed_steps without warm-up that also returns what the HOST spent: (seconds, CPU seconds of the calling Python
    thread, CPU seconds of the whole process) over the `steps` steps - how close a rank is to being launch-bound"""
    import time
    barrier_sync(device)
    t0, c0, p0 = time.perf_counter(), time.thread_time(), time.process_time()
    for _ in range(steps):
        step()
    c1 = time.thread_time()
    barrier_sync(device)
    return max_over_ranks(time.perf_counter() - t0, device), c1 - c0, time.process_time() - p0


def _die_with_parent():
    """preexec hook of a child process: SIGKILL it when the parent dies (Linux PR_SET_PDEATHSIG = 1)"""
    try:
        import ctypes
        import signal
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)
    except Exception:
        pass


class HostShare:
    """Put THIS process where one of 8 ranks of a node would live: all its threads pinned to 1/8 of the host's cores
    (the first share), one compute thread (OMP_NUM_THREADS=1) - and, with ``burners``, 7 neighbour "ranks" beside it: on each
    of the other 7 shares ``busy_per_rank`` spinning processes (a rank of this framework keeps ~2 host threads busy: the
    Python thread and the HIP runtime's; measured process CPU time / step time = 1.6-2.0).  No 8-GPU box needed to see
    whether the launch path of a step survives its share of the host (the reference's DataParallel runs ONE process for 8
    GPUs, /root/reference/tools/train_3d.py:105-140; here it is one process per GPU).
    (Round 5, first form: a spinner on EVERY other logical CPU, 224 of them - the step took 11x longer although none shared
    a core with this process: the box's container has a CPU quota, which the spinners exhaust for everybody.  Recorded in
    docs/history_r1-r5.md; eight real ranks need ~16 cores, not 256.)"""

    def __init__(self, burners: bool = False, shares: int = 8, busy_per_rank: int = 2, max_seconds: int = 600):
        self.burners, self.shares, self.busy_per_rank, self.max_seconds = burners, shares, busy_per_rank, int(max_seconds)
        self.procs_started = 0
        self.procs, self.prev, self.prev_threads = [], {}, None

    @staticmethod
    def _tids():
        import os
        return [int(t) for t in os.listdir("/proc/self/task")]

    def __enter__(self):
        import os
        import subprocess
        cores = sorted(os.sched_getaffinity(0))
        per = max(1, len(cores) // self.shares)
        mine = cores[:per]
        for tid in self._tids():
            try:
                self.prev[tid] = os.sched_getaffinity(tid)
                os.sched_setaffinity(tid, mine)
            except OSError:
                pass
        self.prev_threads = torch.get_num_threads()
        torch.set_num_threads(1)
        self.cores_used, self.cores_total = len(mine), len(cores)
        if self.burners:
            for r in range(1, self.shares):                    # neighbour rank r: its own share of the cores
                share = cores[r * per:(r + 1) * per] or cores[-1:]
                for k in range(self.busy_per_rank):
                    c = share[k % len(share)]
                    # a spinner must not outlive this process however it ends (SIGKILL, a timeout of the caller): the kernel
                    # sends it SIGKILL when its parent dies (PR_SET_PDEATHSIG survives taskset's and bash's exec; no fork in
                    # between, so the Popen pid IS the spinner), and the loop ends by itself after max_seconds regardless
                    self.procs.append(subprocess.Popen(["taskset", "-c", str(c), "bash", "-c",
                                                        f"while [ $SECONDS -lt {self.max_seconds} ]; do :; done"],
                                                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                                       preexec_fn=_die_with_parent))
            import atexit
            atexit.register(self._kill)
        return self

    def _kill(self):
        for p in self.procs:                                   # exactly the processes started above
            if p.poll() is None:
                p.kill()
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                pass
        self.procs = [p for p in self.procs if p.poll() is None]

    def __exit__(self, *exc):
        import os
        n = len(self.procs)
        self._kill()
        self.procs_started = n
        for tid, mask in self.prev.items():
            try:
                os.sched_setaffinity(tid, mask)
            except OSError:
                pass
        torch.set_num_threads(self.prev_threads)
        return False


def _cpu_quota():
    """the container's CPU quota in cores (cgroup v2 cpu.max / v1 cfs quota), or None"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(p), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / p, 2)
    except Exception:
        return None


def host_contention(step, steps: int, device=None, base_ms=None):
    """{condition: step time + host CPU time per step} for: the host as it is, this rank's 1/8 share of the cores, the
    same share with 7 neighbour ranks' worth of busy host threads on the other shares.  `step` must already be warm."""
    out = {}
    for name, ctx in (("unconstrained", None), ("one_eighth_of_the_cores_1_thread", HostShare(False)),
                      ("one_eighth_of_the_cores_7_busy_neighbour_ranks", HostShare(True))):
        if ctx is None:
            el, cpu_t, cpu_p = timed_steps_host(step, steps, device)
            rec = {}
        else:
            with ctx:
                timed_steps_host(step, max(2, steps // 4), device)          # settle on the new cores
                el, cpu_t, cpu_p = timed_steps_host(step, steps, device)
                n_burn = len(ctx.procs)
                rec = {"cores": ctx.cores_used, "of": ctx.cores_total, "burner_processes": n_burn}
        rec.update({"ms_per_step": round(1e3 * el / steps, 4), "host_ms_per_step": round(1e3 * cpu_t / steps, 4),
                    "process_cpu_ms_per_step": round(1e3 * cpu_p / steps, 4), "steps": steps})
        out[name] = rec
    ref = out["unconstrained"]["ms_per_step"]
    for name, rec in out.items():
        rec["vs_unconstrained"] = round(rec["ms_per_step"] / ref, 4)
    out["container_cpu_quota_cores"] = _cpu_quota()
    return out


class Deadline:
    """Bound on a region that may contain a collective which never returns (a communicator whose peer died, a fabric
    link that is down): if the region is not left within ``seconds``, ``on_expire()`` runs on a helper thread and the
    process ends with ``exit_code`` - every rank arms the same deadline, so the job ends together and the launcher sees a
    clean exit instead of a hang.  bench.py arms it around the legs that follow the headline measurement at world > 1:
    rank 0's ``on_expire`` prints the line it already has.  ``seconds <= 0`` disables it."""

    def __init__(self, seconds: float, on_expire=None, exit_code: int = 0, _exit=None):
        self.seconds, self.on_expire, self.exit_code = float(seconds), on_expire, exit_code
        self._exit = _exit if _exit is not None else os._exit
        self._timer, self.expired = None, False

    def _fire(self):
        self.expired = True
        import sys
        try:
            if self.on_expire is not None:
                self.on_expire()
        except Exception as e:                      # the process ends whatever the handler did
            print(f"[deadline] handler failed: {type(e).__name__}: {e}", file=sys.stderr)
        sys.stdout.flush()
        sys.stderr.flush()
        self._exit(self.exit_code)

    def __enter__(self):
        if self.seconds > 0:
            import threading
            self._timer = threading.Timer(self.seconds, self._fire)
            self._timer.daemon = True
            self._timer.start()
        return self

    def __exit__(self, *exc):
        if self._timer is not None:
            self._timer.cancel()
        return False


def job_throughput(units_per_rank_per_step: int, steps: int, seconds: float, world: int) -> float:
    """whole-job units/s under weak scaling: every rank processes its own `units_per_rank_per_step`"""
    return world * units_per_rank_per_step * steps / seconds


def rank_seed(base_seed: int, rank: int) -> int:
    """distinct, reproducible stream per rank (augmentation, synthetic-root sampling: identical draws on every rank
    would make N GPUs train on N copies of one sample)"""
    return (int(base_seed) * 1_000_003 + 7919 * (int(rank) + 1)) % (2 ** 31 - 1)


def rank_generator(base_seed: int, rank: int | None = None) -> torch.Generator:
    r = env_world()[0] if rank is None else rank
    return torch.Generator().manual_seed(rank_seed(base_seed, r))


def needs_find_unused(cfg) -> bool:
    """Does DDP have to search for parameters without gradients every step (find_unused_parameters)?  No: every
    parameter that requires grad gets one on every rank in every iteration, by construction -
    * frozen sub-nets have requires_grad=False (tools/train_3d.py select_trainable, following the reference's
      tools/train_3d.py:48-75; this includes the root net when proposals come from ground truth, USE_GT);
    * a sub-net that trains but reaches no loss term in an iteration - the pose net when a rank's frames hold no valid
      proposal, the attention net on the TRAIN_ONLY_2D / TRAIN_ONLY_ROOTNET / INIT_TRAIN_EPOCHS_ROOTNET / SINGLE_AUG return
      paths, the root net without a 3D target, the backbone when heat-maps are handed in - is tied to the loss with zero
      weight on EVERY return path of the models' training forwards (engine.anchor_unreached: the models keep the set of
      sub-nets their loss terms went through), as the reference does with zero-weighted dummy forwards
      (lib/models/multi_person_posenet_ssv.py:290,429,496,499); tests/test_distributed_gloo.py runs those paths on 2 ranks.
    So the static-graph fast path of DDP applies, and no rank can skip backward() (which would hang the others)."""
    return False


def wrap_ddp(model: torch.nn.Module, device: torch.device | None = None, find_unused: bool = True):
    """DistributedDataParallel with gradient-only all-reduce; no-op for a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    if device is not None and device.type == "cuda":
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], output_device=device.index,
                                                         find_unused_parameters=find_unused, bucket_cap_mb=32,
                                                         gradient_as_bucket_view=True, process_group=_DATA_GROUP)
    return torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=find_unused, process_group=_DATA_GROUP)
