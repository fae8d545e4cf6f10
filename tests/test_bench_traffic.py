"""bench.py reports roofline.traffic -- HBM bytes per launch from the PMC passes of tools/profile_round.sh, a measurement
of ANOTHER run -- only when it was taken on the kernel sources of this tree (profiles/pmc_traffic.json carries their digest);
otherwise the figure goes to traffic_stale and traffic is null.  CPU test of that rule on the file as committed."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_traffic_is_reported_only_for_the_sources_it_was_measured_on():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    stamp = d["_measured_on"]
    assert len(stamp["kernel_source_sha16"]) == 16 and stamp["round"]
    current = stamp["kernel_source_sha16"] == bench.kernel_source_sha16()
    r = bench.roofline("varbase", "k_point_scalarmul_ct", 1 << 20, 34.0, "index-independent")
    assert r["traffic_measured_on"]["current"] == current
    if current:
        assert r["traffic"] == d["k_point_scalarmul_ct"] and "traffic_stale" not in r
        assert 1.0 < r["traffic"] / (568 * (1 << 20)) < 5.0            # a few times the algorithmic bytes, not hundreds
    else:
        assert r["traffic"] is None and r["traffic_stale"] == d["k_point_scalarmul_ct"]
    # a verification step is three kernels: their traffic is summed
    names = ("k_ed448_verify_keycomb_xwide", "k_verify_key_combs", "k_verify_base_part")
    total, info = bench.pmc_traffic(names)
    assert total == sum(d[k] for k in names) and info["kernels"] == list(names)


def test_multiply_accumulates_and_gathers_follow_the_width_of_the_base_table():
    """The verification and base-point figures are priced for 28 digits of 16 bits; the device's table may have wider
    digits (fewer additions, each 7 multiplications of 192 multiply-accumulates), and its gathers -- one 192-byte entry,
    two 128-byte lines, per digit -- are named beside the measured traffic."""
    n, names = 1 << 20, ("k_ed448_verify_keycomb_xwide", "k_verify_key_combs", "k_verify_base_part")
    at = lambda bits: bench.roofline("verify", names[0], n, 7.6, "index-independent", names, bits)
    assert at(16)["mac"]["macs_per_op"] == bench.WORKLOADS["verify"]["macs"] == at(0)["mac"]["macs_per_op"]
    assert at(24)["mac"]["macs_per_op"] == bench.WORKLOADS["verify"]["macs"] - 9 * 7 * 192         # 19 digits instead of 28
    assert at(20)["mac"]["macs_per_op"] == bench.WORKLOADS["verify"]["macs"] - 5 * 7 * 192
    assert at(24)["traffic_of_base_table_gathers"] == 19 * 256 * n and "traffic_note" in at(24)
    assert [bench.base_table_windows(b) for b in (8, 16, 18, 20, 22, 24)] == [56, 28, 25, 23, 21, 19]
    # the base point's own multiplication takes the table only with digit-addressed tables
    fast = bench.roofline("base", "k_base_scalarmul", n, 1.7, "fast", None, 24)
    assert fast["mac"]["macs_per_op"] == bench.WORKLOADS["base"]["macs"] - 9 * 7 * 192 and "traffic_of_base_table_gathers" in fast
    default = bench.roofline("base", "k_base_scalarmul_ct", n, 5.3, "index-independent", None, 24)
    assert default["mac"]["macs_per_op"] == bench.WORKLOADS["base"]["macs_index_independent"] and "traffic_of_base_table_gathers" not in default
