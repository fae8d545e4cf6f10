#!/bin/bash
# round-4 GPU call 21: the driver's own commands on the final tree, for the record: python bench.py (defaults)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call21
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python bench.py > "$OUT/bench_driver_line.json" 2> "$OUT/bench_driver_err.txt"; echo "rc $?"
python - "$OUT/bench_driver_line.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline %.2f M/s  %.3f ms  mac %.3f  traffic %s" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["mac"]["frac"], d["roofline"]["traffic"]))
for k, c in d["configs"].items():
    print("%-22s %7.1f M/s %7.3f ms  mac_frac %s  traffic %s  bits %s %s" % (k, c["value"] / 1e6, c["kernel_ms_avg"], c.get("mac_frac"), c["roofline"].get("traffic"), c.get("base_table_bits"), c["parity_spot_check"]))
e = d["end_to_end"]
for k in ("varbase", "fixed", "verify"):
    print("e2e %-8s %7.1f M/s %7.3f ms  pcie_frac %.2f" % (k, e[k]["value"] / 1e6, e[k]["ms_per_call"], e[k]["pcie_frac"]))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"])
PY
