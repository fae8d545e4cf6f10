#!/bin/bash
# round-4 GPU call 24: with the widest combs the entries kernel takes 0.74 ms (0.43 before): is there room for a third and
# fourth round of S*B ahead now?  2 (product) / 3 / 4 rounds, config 4 and config 5's share
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call24
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2 3; do
  timeout 900 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify_ahead_xwide.txt" 2>&1
done
timeout 900 python tests/variant_sweep.py "--workload verify --log2-batch 21 --steps 10 --warmup 5" >> "$OUT/sweep_verify_ahead_xwide.txt" 2>&1
cat "$OUT/sweep_verify_ahead_xwide.txt"
