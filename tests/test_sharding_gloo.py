"""CPU test of the N > 1 path: world_size-2 gloo process group, the sharding helpers bench.py uses
(contiguous disjoint shards, MAX-over-ranks timing, SUM of counters).  The per-rank "kernel" here
is the oracle (test infrastructure) so the test runs without a GPU."""
import hashlib
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here); sys.path.insert(0, os.path.dirname(here))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import _gen
    from _libs import oracle
    from libgoldilocks_amd.shard import max_over_ranks, shard_range, sum_over_ranks
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O = oracle()
    lo, hi = shard_range(n, rank, world)
    s = _gen.stream_scalars(n, b"shard-test")           # every rank derives the same global batch
    enc = _gen.oracle_encode(_gen.oracle_fixed(O, s[lo:hi]))
    dist.barrier()
    slow = max_over_ranks(1.0 + rank, dist)             # rank-dependent "elapsed time"
    total = sum_over_ranks(hi - lo, dist)
    q.put((rank, lo, hi, hashlib.sha256(enc.tobytes()).hexdigest(), slow, total))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_properties():
    from libgoldilocks_amd.shard import shard_range
    for n in (0, 1, 7, 1000, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 4, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def test_two_rank_gloo_sharding(O):
    import torch.multiprocessing as mp
    import _gen
    n, world = 300, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = _gen.oracle_encode(_gen.oracle_fixed(O, _gen.stream_scalars(n, b"shard-test")))
    for rank, lo, hi, digest, slow, total in res:
        assert digest == hashlib.sha256(whole[lo:hi].tobytes()).hexdigest()
        assert slow == 2.0 and total == n
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n
