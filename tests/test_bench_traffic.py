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
    names = ("k_ed448_verify_keycomb_wide", "k_ed448_verify_keycomb_finish", "k_verify_base_part")
    total, info = bench.pmc_traffic(names)
    assert total == sum(d[k] for k in names) and info["kernels"] == list(names)
