#!/bin/bash
# One extra rocprofv3 --pmc pass over a bench workload:  bash tools/pmc_once.sh <workload> <COUNTER> [COUNTER...]
# (counters only -- never combined with a trace domain); prints the per-kernel averages.
# BENCH_ARGS="--table-access fast" adds bench.py arguments.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=$1; shift
OUT=$ROOT/gpurun_out/pmc_once
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_X_$WL" -- \
    python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-end-to-end ${BENCH_ARGS:-} > "$OUT/log.txt" 2>&1
tail -3 "$OUT/log.txt" | cut -c1-300
python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$OUT/summary" > /dev/null
python3 - "$OUT/summary/rocprofv3_pmc_summary.json" <<'PY'
import json, sys
for r in json.load(open(sys.argv[1])):
    if r["dispatches"] >= 2:
        print("%-26s %-24s x%d  %.5g" % (r["kernel"], r["counter"], r["dispatches"], r["avg_per_dispatch"]))
PY
