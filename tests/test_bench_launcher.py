"""bench.py's N > 1 path, without a GPU: `python bench.py --gpus 2` must start its own ranks (the
driver runs it exactly like that), print ONE JSON line with the aggregate and the per-GPU figures,
and never import torch in the launching process (a process that touched the GPU may not re-exec;
this one never touches it).  The step is a stub (--stub-step-ms): what is under test is
libgoldilocks_amd/shard.py -- launcher, process group, barrier-bracketed timing, MAX over ranks,
all-gather of per-rank rows -- the same code the GPU ranks run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _check_line(out, world, step_ms, steps):
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["steps"] == steps and line["scaling"] == "weak"
    assert [g["rank"] for g in line["per_gpu"]] == list(range(world))
    # rank r sleeps (1 + r) * step_ms per step: the job's time is the slowest rank's (MAX over ranks)
    assert line["ms_per_step"] >= world * step_ms * 0.95
    slowest = min(g["value"] for g in line["per_gpu"])
    assert line["value"] == pytest.approx(world * slowest, rel=0.05)
    return line


def test_plain_invocation_launches_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-step-ms", "20",
                        "--log2-batch", "10"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    _check_line(r.stdout, 2, 20, 3)
    launcher = [json.loads(l) for l in r.stderr.splitlines() if l.startswith('{"launcher"')]
    assert launcher and launcher[0]["launcher"]["ranks"] == 2
    assert launcher[0]["launcher"]["torch_imported_by_launcher"] is False


def test_under_torch_distributed_run():
    from libgoldilocks_amd.shard import free_port
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH, "--gpus", "2",
                        "--steps", "2", "--warmup", "1", "--stub-step-ms", "20"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    _check_line(r.stdout, 2, 20, 2)


def test_failing_rank_fails_the_launch():
    from libgoldilocks_amd import shard
    code = shard.launch_ranks(["-c", "import os, sys; sys.exit(3 if os.environ['RANK'] == '1' else 0)"], 2, timeout=60)
    assert code == 3


def test_shard_module_is_torch_free_at_import():
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import libgoldilocks_amd.shard; "
                        "print(any(m.split('.')[0] == 'torch' for m in sys.modules))" % ROOT],
                       capture_output=True, text=True, timeout=120)
    assert r.stdout.strip() == "False", r.stdout + r.stderr


def test_rank_to_device_mapping_and_strong_slices():
    from libgoldilocks_amd.shard import device_for_rank, shard_range
    assert [device_for_rank(r, 1) for r in range(4)] == [0, 0, 0, 0]
    assert [device_for_rank(r, 8) for r in range(8)] == list(range(8))
    with pytest.raises(ValueError):
        device_for_rank(0, 0)
    # BASELINE config 5: 2^24 verifications over 8 GPUs = 2^21 each
    assert [shard_range(1 << 24, r, 8)[1] - shard_range(1 << 24, r, 8)[0] for r in range(8)] == [1 << 21] * 8


def test_a_dead_rank_ends_the_launch_promptly():
    """Rank 1 exits with code 3 while rank 0 would sit in a barrier (here: sleeps a minute): the launcher polls all
    children, returns the failure within seconds and leaves no child behind (a 600-second
    process-group timeout is not how a dead rank should show up on the first 8-GPU run)."""
    import time
    from libgoldilocks_amd import shard
    t0 = time.time()
    code = shard.launch_ranks(["-c", "import os, sys, time\n"
                               "if os.environ['RANK'] == '1': sys.exit(3)\n"
                               "time.sleep(60)"], 2)
    assert code == 3 and time.time() - t0 < 5.0


def test_the_launch_deadline_is_a_default():
    import inspect
    import time
    from libgoldilocks_amd import shard
    assert inspect.signature(shard.launch_ranks).parameters["timeout"].default == shard.LAUNCH_TIMEOUT_S
    t0 = time.time()
    assert shard.launch_ranks(["-c", "import time; time.sleep(60)"], 2, timeout=1.0) == 124
    assert time.time() - t0 < 5.0


def test_eight_ranks_as_the_scaling_run_launches_them():
    """The first real 8-GPU run starts eight ranks: rendezvous on a free port, eight gloo processes, barriers, the MAX
    over ranks and the all-gather of eight rows -- here with the stub step, in both of the driver's forms: independent
    batches (weak) and BASELINE config 5's contiguous slices of one global batch (strong)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--stub-step-ms", "5",
                        "--log2-batch", "12"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _check_line(r.stdout, 8, 5, 2)
    assert [g["batch"] for g in line["per_gpu"]] == [1 << 12] * 8 and line["config"]["control_plane"] == "gloo"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--stub-step-ms", "5",
                        "--workload", "verify", "--global-log2-batch", "15"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong"
    slices = [g["slice"] for g in line["per_gpu"]]
    assert slices == [[i << 12, (i + 1) << 12] for i in range(8)]           # disjoint, contiguous, the whole batch
    assert sum(g["batch"] for g in line["per_gpu"]) == 1 << 15


def test_host_feed_probe_runs_without_a_device():
    """tools/hostfeed: the host side of goldilocks_ed448_verify_batch_ex (csrc/host_pack.hpp, the library's own packing
    code) for 1 / 2 / 4 / 8 shards with the device calls stubbed -- CPU only; here at a small size, as a smoke test."""
    src, exe = os.path.join(ROOT, "tools", "hostfeed.cpp"), os.path.join(ROOT, "tools", "hostfeed")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "libgoldilocks_amd", "csrc"), "-o", exe, src],
                   check=True, timeout=300)
    r = subprocess.run([exe, "--log2n", "17", "--reps", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = [l.split() for l in r.stdout.splitlines() if l.split() and l.split()[0].isdigit()]
    assert [int(x[0]) for x in rows] == [1, 2, 4, 8] and all(float(x[2]) > 1e5 for x in rows), r.stdout
    assert "cores usable by this process" in r.stdout


def test_reference_tool_rows_of_the_cpu_baseline():
    """bench.py's cpu_baseline.reference_tool: the reference's own test/bench_goldilocks.cxx, built by oracle/Makefile against
    the reference library where /root/reference is, runs on the host and its Ed448 rows parse (seconds per operation)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = bench.reference_tool_rows("x86_64")
    if rows is None:
        pytest.skip("oracle/_ref/bench_goldilocks_x86_64 not built here (no /root/reference)")
    assert "error" not in rows, rows
    t = rows["seconds_per_op"]
    assert 1e-5 < t["Point scalarmul"] < 1e-2 and 1e-5 < t["EdDSA verify"] < 1e-2
    assert abs(rows["point_scalarmul_per_s"] * t["Point scalarmul"] - 1) < 1e-9
