"""Multi-GPU plumbing of the batch engine: one process per GPU, no collective on the data path.

The path shards embarrassingly: operation i of a batch belongs to exactly
one rank and the devices never exchange data.  Two layouts:
  * weak:   every rank owns its own batch of the same size            (bench.py default)
  * strong: one global batch, contiguous slices [g*n/G, (g+1)*n/G)    (shard_range; BASELINE
            config 5: 2^24 verifications over 8 GPUs = 2^21 per GPU)
The only collectives are control-plane: the barrier around the timed region, MAX of the elapsed
time and an all-gather of per-rank figures.

bench.py uses everything in this file:
  launch_ranks()   `python bench.py --gpus N` without a launcher: start N fresh children, one per
                   rank, BEFORE anything in this process has touched a GPU (this module imports
                   neither torch nor HIP at module level; a process that has initialised the GPU
                   must never re-exec, and this one never initialises it at all)
  init_group()     the process group of a rank: gloo always, plus RCCL ("nccl") on top when every rank has
                   a GPU of its own AND the ranks agree (over gloo) that it came up on all of them
  timed_region()   W untimed steps, barrier + sync, K timed steps, barrier + sync
  max_over_ranks / sum_over_ranks / gather_over_ranks
"""
import os
import socket
import subprocess
import sys
import time


def shard_range(n, rank, world):
    """Contiguous slice of [0, n) owned by `rank`; slices are disjoint, ordered and cover [0, n)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d not in [0, %d)" % (rank, world))
    return (n * rank) // world, (n * (rank + 1)) // world


def device_for_rank(local_rank, visible_devices):
    """Ranks map onto the visible devices modulo their count (so a 1-GPU box can run a 2-rank job)."""
    if visible_devices < 1:
        raise ValueError("no visible device")
    return local_rank % visible_devices


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    return env


LAUNCH_TIMEOUT_S = 3600.0   # a whole bench run of one rank, generously; launch_ranks(timeout=None) waits forever


def launch_ranks(argv, world, timeout=LAUNCH_TIMEOUT_S, poll_s=0.05):
    """Run `python argv...` once per rank (fresh processes, rendezvous on 127.0.0.1) and return the
    largest exit code.  Rank 0's stdout is this process's stdout (the one JSON line); the other
    ranks' stdout goes to stderr.  Nothing here touches a GPU.
    All children are polled together: the first one that ends with a non-zero code ends the launch --
    the others (probably waiting for it in a barrier) are killed at once instead of sitting out the
    process group's timeout -- and so does the deadline (exit code 124)."""
    port = free_port()
    procs = []
    for rank in range(world):
        out = None if rank == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=rank_env(rank, world, port), stdout=out))
    deadline = None if timeout is None else time.time() + timeout
    code = 0
    try:
        running = list(procs)
        while running:
            for p in list(running):
                rc = p.poll()
                if rc is None:
                    continue
                running.remove(p)
                code = max(code, abs(rc))
            if code:                      # a rank failed: do not wait for the ones it left hanging
                break
            if deadline is not None and time.time() > deadline:
                code = 124
                break
            if running:
                time.sleep(poll_s)
    finally:
        for p in procs:           # exactly the children started here, by pid
            if p.poll() is None:
                p.kill()
                p.wait()
    return code


RCCL_RANKS_SEEN = [None]   # set by init_group when an RCCL bring-up was attempted: the number of ranks it succeeded on


class Group(object):
    """The control-plane process group of a rank: torch.distributed bound to ONE group, so that callers
    (bench.py, the helpers below) write dist.barrier() / dist.all_reduce(t) whichever backend carries it."""

    def __init__(self, dist, group):
        self._dist, self.group = dist, group
        self.ReduceOp = dist.ReduceOp

    def is_initialized(self):
        return self._dist.is_initialized()

    def get_world_size(self):
        return self._dist.get_world_size(group=self.group)

    def barrier(self):
        self._dist.barrier(group=self.group)

    def all_reduce(self, t, op=None):
        self._dist.all_reduce(t, op=op if op is not None else self._dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, out, t):
        self._dist.all_gather(out, t, group=self.group)

    def destroy_process_group(self):
        self._dist.destroy_process_group()


def init_group(world, rank, visible_devices, use_gpu=True, rccl_timeout_s=120):
    """The process group of this rank, or (None, None) when world == 1.  Returns (Group, backend).

    The DEFAULT group is always gloo (it comes up wherever TCP on 127.0.0.1 does).  When every rank has a
    GPU of its own an RCCL ("nccl") group is brought up on top of it, and the ranks AGREE over gloo whether
    that worked for all of them: RCCL carries the barrier / MAX / all-gather only if every rank's bring-up
    succeeded, otherwise every rank uses gloo -- no rank decides on its own, so a partial RCCL failure can
    neither hang the others in a barrier nor split the job over two backends.  GOLDILOCKS_BENCH_BACKEND=gloo
    skips RCCL.  The group only carries the control plane; nothing of the data path."""
    if world <= 1 and not os.environ.get("GOLDILOCKS_BENCH_FORCE_DIST"):   # the knob lets a 1-GPU box exercise RCCL
        return None, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    world = max(world, 1)
    import datetime
    import torch
    import torch.distributed as dist
    own_device = use_gpu and visible_devices >= world
    want = os.environ.get("GOLDILOCKS_BENCH_BACKEND") or ("nccl" if own_device else "gloo")
    os.environ.setdefault("NCCL_DEBUG", "WARN")          # keep RCCL's banner off stdout: one JSON line only

    # gloo and RCCL print banners on the C-level stdout; rank 0's stdout carries ONE JSON line only
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
        dist.barrier()
        group, backend = None, "gloo"
        if want == "nccl":
            mine, rccl = 1, None
            try:
                rccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=rccl_timeout_s))
                dist.barrier(group=rccl)                 # connections are made (and announced) lazily
                torch.cuda.synchronize()
            except Exception as e:   # noqa: BLE001 -- whatever RCCL raises on this node
                mine = 0
                print("shard.init_group: rank %d: RCCL bring-up failed (%s)" % (rank, e), file=sys.stderr)
            agreed = torch.tensor([mine], dtype=torch.int32)
            dist.all_reduce(agreed, op=dist.ReduceOp.SUM)    # over gloo: every rank learns the same answer
            RCCL_RANKS_SEEN[0] = int(agreed.item())          # how many ranks' RCCL bring-up succeeded (the line reports it)
            if RCCL_RANKS_SEEN[0] == world:
                group, backend = rccl, "nccl"
            elif rank == 0:
                print("shard.init_group: RCCL did not come up on every rank; the control plane stays on gloo",
                      file=sys.stderr)
    finally:
        try:   # what the C libraries wrote sits in stdio's buffer: push it out while fd 1 still is stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:   # noqa: BLE001
            pass
        os.dup2(saved, 1)
        os.close(saved)
    return Group(dist, group), backend


def _reduce(value, op_name, dist, backend, dtype_name):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    dev = "cuda" if backend == "nccl" else "cpu"
    t = torch.tensor([value], dtype=getattr(torch, dtype_name), device=dev)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op_name))
    return t.item()


def max_over_ranks(value, dist=None, backend="gloo"):
    """MAX-reduce a python float over the process group (identity without one)."""
    return float(_reduce(float(value), "MAX", dist, backend, "float64"))


def sum_over_ranks(value, dist=None, backend="gloo"):
    """SUM-reduce a python int over the process group (identity without one)."""
    return int(_reduce(int(value), "SUM", dist, backend, "int64"))


def gather_over_ranks(values, dist=None, backend="gloo"):
    """All-gather a short list of python floats; returns one list per rank, in rank order."""
    vals = [float(v) for v in values]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [vals]
    import torch
    dev = "cuda" if backend == "nccl" else "cpu"
    mine = torch.tensor(vals, dtype=torch.float64, device=dev)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [[float(x) for x in t.cpu().tolist()] for t in out]


def timed_region(step, steps, warmup, sync, dist=None, backend="gloo", after_step=None):
    """The driver's timing contract: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by
    barrier + device sync on both sides.  Returns this rank's seconds and the MAX over ranks.
    after_step(i) runs inside the loop right after step i was enqueued (event records)."""
    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    if after_step:
        after_step(-1)
    for i in range(steps):
        step()
        if after_step:
            after_step(i)
    sync()
    mine = time.perf_counter() - t0
    barrier()
    return mine, max_over_ranks(mine, dist, backend)
