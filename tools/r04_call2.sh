#!/bin/bash
# round-4 GPU call 2: the phased verification (first pass / finish kernels, chunked host-array pipeline), 8-rank launches,
# key-comb reach, host feed rate on the GPU box's host, traffic of the key-comb kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call2
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/gputest.txt" 2>&1; tail -15 "$OUT/gputest.txt"
timeout 200 python bench.py --workload verify --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/bench_verify.json" 2>"$OUT/bench_verify.err"; cut -c1-300 "$OUT/bench_verify.json"
timeout 600 python bench.py > "$OUT/bench_default.json" 2>"$OUT/bench_default.err"; python - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", d["value"], d["roofline"]["kernel_ms_avg"], d["roofline"].get("traffic"), d["roofline"].get("traffic_measured_on"))
for k,v in d["configs"].items(): print(k, round(v["value"]/1e6,2), "M/s", round(v["kernel_ms_avg"],3), "ms mac_frac", v["mac_frac"])
print(json.dumps(d["end_to_end"])[:1500])
PY
timeout 300 python tests/key_pool_probe.py --quick > "$OUT/key_pool_probe.txt" 2>&1; tail -25 "$OUT/key_pool_probe.txt"
for o in sequential scattered; do for s in none memcpy; do tools/hostfeed --log2n 22 --stage $s --order $o; done; done > "$OUT/hostfeed.txt" 2>&1; cat "$OUT/hostfeed.txt"
bash tools/pmc_once.sh verify FETCH_SIZE > "$OUT/pmc_verify_fetch.txt" 2>&1; grep keycomb "$OUT/pmc_verify_fetch.txt"
bash tools/pmc_once.sh verify WRITE_SIZE > "$OUT/pmc_verify_write.txt" 2>&1; grep keycomb "$OUT/pmc_verify_write.txt"
