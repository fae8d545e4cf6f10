"""Prints the figures of one bench.py line (a JSON file) one per row: what a gpurun call's tail should show."""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("headline %.3f M/s  %.3f ms/step  mac.frac %.3f  hbm.frac %.5f  device memory %d MiB" % (
    d["value"] / 1e6, d["ms_per_step"], r.get("mac", {}).get("frac", 0), r["frac"], d["config"].get("device_memory_bytes", 0) >> 20))
for k, v in d.get("configs", {}).items():
    print("%-22s %8.2f M/s  kernel %7.3f ms  mac %.3f  table bits %2d  device memory %6d MiB  %s" % (
        k, v["value"] / 1e6, v["kernel_ms_avg"], v["mac_frac"] or 0, v["base_table_bits"], v.get("device_memory_bytes", 0) >> 20,
        v["parity_spot_check"]))
for k, v in d.get("end_to_end", {}).items():
    if isinstance(v, dict) and "value" in v:
        print("end_to_end %-12s %8.2f M/s  %s" % (k, v["value"] / 1e6, {a: v[a] for a in ("ms", "ms_per_call", "pcie_frac") if a in v}))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("cpu_baseline %.1f k/s on %d cores (%s)" % (c["value"] / 1e3, c["cores"], c["kind"]))
