#!/bin/bash
# round-4 GPU call 1: parity of the reworked ladders, the swap A/B, two-ladder kernels, key-comb phase diagnostics
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call1
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests -m gpu -x -q > "$OUT/gputest.txt" 2>&1; tail -3 "$OUT/gputest.txt"
timeout 300 python tests/variant_sweep.py "--workload varbase --steps 10 --warmup 3" > "$OUT/sweep_varbase.txt" 2>&1
timeout 300 python tests/variant_sweep.py "--workload varbase --steps 10 --warmup 3" >> "$OUT/sweep_varbase.txt" 2>&1
timeout 200 python tests/variant_sweep.py "--workload x448 --steps 10 --warmup 3" > "$OUT/sweep_x448.txt" 2>&1
cat "$OUT/sweep_varbase.txt" "$OUT/sweep_x448.txt"
timeout 200 python tests/ct_varbase_probe.py > "$OUT/ct_varbase_probe.txt" 2>&1; cat "$OUT/ct_varbase_probe.txt"
timeout 100 python bench.py --workload verify --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/bench_verify.json" 2>&1; cut -c1-400 "$OUT/bench_verify.json"
for v in t7m0x0 t8m0x0 t8m0x1 t8m1x0 t7m2x0 t7m2x1; do
  timeout 100 tools/keycombphases_$v > "$OUT/keycombphases_$v.txt" 2>&1; cat "$OUT/keycombphases_$v.txt" | tail -9
done
