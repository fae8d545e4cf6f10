"""bench.py as the driver runs it, on the GPU box: `python bench.py --gpus 2 ...` without a launcher must
start its own two ranks (mapped onto the visible devices modulo their count: both on device 0 of a 1-GPU
box), run the real kernels and print ONE JSON line with n_gpus = 2, the aggregate and the per-GPU
figures; and the BASELINE config-5 invocation (verify, one global batch cut into per-rank slices)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), r.stderr


def test_two_ranks_self_launched_variable_base():
    line, err = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "16", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["parity_spot_check"] == "ok"
    assert [g["rank"] for g in line["per_gpu"]] == [0, 1] and all(g["value"] > 0 for g in line["per_gpu"])
    assert line["value"] > 0 and line["roofline"]["kernel"] == "k_point_scalarmul_ct"   # the library's default mode
    assert line["config"]["table_access"] == "index-independent" and line["config"]["table_access_is_library_default"]
    assert "configs" not in line and "cpu_baseline" not in line and "end_to_end" not in line   # N = 1 only
    launcher = [json.loads(l) for l in err.splitlines() if l.startswith('{"launcher"')]
    assert launcher and launcher[0]["launcher"]["torch_imported_by_launcher"] is False


def test_config5_invocation_strong_slices_verify():
    """BASELINE config 5 in miniature: one global batch of 2^15 verifications over 2 ranks = 2^14 each."""
    line, _ = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "verify", "--global-log2-batch", "15"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["unit"] == "verifies/s"
    assert [g["batch"] for g in line["per_gpu"]] == [1 << 14, 1 << 14]
    assert line["config"]["parity_spot_check"] == "ok"


def test_default_line_carries_configs_end_to_end_and_cpu_baseline():
    """The driver's N = 1 invocation at a reduced batch: the headline in the library's default
    (index-independent) mode + configs 3, 3', 4 and the opt-in fast tables, every config's first lanes
    re-computed by the oracle + the host-array (PCIe-inclusive) rates + the CPU baseline of both reference builds."""
    line, _ = _run(["--steps", "2", "--warmup", "1", "--log2-batch", "15"])
    assert line["n_gpus"] == 1 and line["config"]["parity_spot_check"] == "ok"
    assert line["roofline"]["kernel"] == "k_point_scalarmul_ct" and line["config"]["table_access"] == "index-independent"
    assert "oracle's goldilocks_448_point_scalarmul" in line["config"]["check"]
    assert set(line["configs"]) == {"fixed", "base", "verify", "verify_distinct_keys", "base_fast", "verify_table24"}
    for c in line["configs"].values():
        assert c["parity_spot_check"] == "ok" and c["value"] > 0 and c["kernel_ms_avg"] > 0
        assert "equal the oracle's" in c["check"]                        # not a self-comparison
        assert c["device_memory_bytes"] > 0                              # every config states what the library holds
    # config 4 at the library's default table (20 bits, bounded footprint) and at the opt-in 24 bits
    assert line["configs"]["verify"]["base_table_bits"] == 20 and line["configs"]["verify"]["base_table_bits_is_library_default"]
    assert line["configs"]["verify_table24"]["base_table_bits"] == 24
    assert line["configs"]["verify"]["device_memory_bytes"] < 10 << 30 < line["configs"]["verify_table24"]["device_memory_bytes"]
    peaks = line["roofline"]["mac"]["peaks"]
    assert line["roofline"]["mac"]["peak"] == peaks[peaks["used"]] == 38.4
    assert "rejects among them" in line["configs"]["verify"]["check"]
    assert line["configs"]["base"]["kernel"] == "k_base_scalarmul_ct" and line["configs"]["base_fast"]["kernel"] == "k_base_scalarmul"
    assert set(line["end_to_end"]) == {"varbase", "fixed", "verify", "link_gbs"}
    assert line["end_to_end"]["link_gbs"]["h2d"] > 1 and line["end_to_end"]["link_gbs"]["d2h"] > 1    # the link, measured in the same run
    for k, e in line["end_to_end"].items():
        if k != "link_gbs":
            assert e["value"] > 0 and e["host_memory"] == "pageable" and 0 < e["pcie_frac"] < 1.5
    assert line["end_to_end"]["varbase"]["value"] < line["value"]        # PCIe-inclusive: never the headline
    cb = line["cpu_baseline"]
    assert set(cb["single_thread_by_build"]) <= {"x86_64_generic", "x86_64_v3", "oracle_port"} and cb["build"] in cb["single_thread_by_build"]
    assert cb["cores"] >= 1 and cb["single_thread"]["us_per_op"] > 0
    # the reference's own tool (test/bench_goldilocks.cxx, built by oracle/Makefile where /root/reference is) ran beside it
    tool = cb["reference_tool"]
    assert tool is None or 0 < tool["seconds_per_op"]["Point scalarmul"] < 1e-3
    # the stated core count is consistent with the speed-up over one thread (within 2x)
    ratio = cb["value"] / cb["single_thread"]["value"]
    assert cb["cores"] / 2.0 <= max(ratio, 1.0) * 2.0 and ratio <= cb["cores"] * 2.0


def test_control_plane_over_rccl_on_one_gpu():
    """The barrier / MAX / all-gather path over RCCL ("nccl"), as the ranks of a real multi-GPU run use it:
    a one-rank group on this box's GPU."""
    line, _ = _run(["--steps", "2", "--warmup", "1", "--log2-batch", "14", "--no-cpu-baseline", "--no-configs",
                    "--no-end-to-end"], GOLDILOCKS_BENCH_FORCE_DIST="1")
    assert line["config"]["control_plane"] == "nccl" and line["n_gpus"] == 1
    assert line["config"]["parity_spot_check"] == "ok"


def test_eight_ranks_on_one_device_both_forms():
    """What the first 8-GPU run does, on this box's one device (ranks map to the visible devices modulo their count):
    eight self-launched ranks, eight library contexts building their tables at once, one agreed control-plane backend,
    eight per-GPU rows -- independent batches, then BASELINE config 5's form (verify, one global batch cut into eight
    disjoint slices)."""
    line, err = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--log2-batch", "12", "--no-cpu-baseline"])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["parity_spot_check"] == "ok"
    assert [g["rank"] for g in line["per_gpu"]] == list(range(8)) and all(g["value"] > 0 and g["batch"] == 1 << 12 for g in line["per_gpu"])
    assert line["config"]["control_plane"] in ("gloo", "nccl")            # ONE backend for all ranks (agreed over gloo)
    line, _ = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--workload", "verify", "--global-log2-batch", "15"])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["unit"] == "verifies/s"
    assert [g["slice"] for g in line["per_gpu"]] == [[i << 12, (i + 1) << 12] for i in range(8)]
    assert line["config"]["parity_spot_check"] == "ok"
