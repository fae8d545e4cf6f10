#!/bin/bash
# round-4 GPU call 12: the next entry of the base point's window table requested before the current addition in the
# standalone ladder (ladder_bwt): k_base_scalarmul, signing with fast tables, S*B ahead of a verification
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call12
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do
  timeout 300 python tests/variant_sweep.py "--workload base --table-access fast --steps 20 --warmup 5" >> "$OUT/sweep.txt" 2>&1
  timeout 300 python tests/variant_sweep.py "--workload sign --table-access fast --steps 20 --warmup 5" >> "$OUT/sweep.txt" 2>&1
  timeout 300 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep.txt" 2>&1
done
cat "$OUT/sweep.txt"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_verify" -- python3 "$ROOT/bench.py" --workload verify --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/trace_verify.log" 2>&1 )
python tools/trace_timeline.py "$OUT/trace_verify" k_verify_dedupe | tee "$OUT/timeline_verify.txt"
rm -rf "$OUT/trace_verify"
