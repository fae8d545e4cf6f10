#!/bin/bash
# round-4 GPU call 3: S*B ahead of the key-comb verification (auxiliary stream), the reworked host-array pipeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call3
mkdir -p "$OUT"
cd "$ROOT"
timeout 200 python bench.py --workload verify --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/bench_verify.json" 2>"$OUT/bench_verify.err"; cut -c1-300 "$OUT/bench_verify.json"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_verify" -- python3 "$ROOT/bench.py" --workload verify --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/trace_verify.log" 2>&1 )
python tools/trace_timeline.py "$OUT/trace_verify" k_verify_dedupe | tee "$OUT/timeline_verify.txt"
rm -rf "$OUT/trace_verify"
timeout 1800 python -m pytest tests -m gpu -x -q > "$OUT/gputest.txt" 2>&1; tail -15 "$OUT/gputest.txt"
timeout 600 python bench.py > "$OUT/bench_default.json" 2>"$OUT/bench_default.err"; python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["kernel_ms_avg"])
for k,v in d["configs"].items(): print(k, round(v["value"]/1e6,2), "M/s", round(v["kernel_ms_avg"],3), "ms mac_frac", v["mac_frac"])
for k,v in d["end_to_end"].items(): print(k, v if k=="link_gbs" else (round(v["value"]/1e6,2), v["ms_every_call"], round(v["pcie_frac"],3)))
PY
