#!/bin/bash
# round-4 GPU call 4: how many rounds of S*B ahead of the key-comb verification (0 / 2 / 3 = product / 4), device-resident
# and through the host-array pipeline, alternating on one box
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call4
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do
  timeout 400 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify.txt" 2>&1
done
cat "$OUT/sweep_verify.txt"
for rep in 1 2; do
  for v in product ahead0 ahead2 ahead4; do
    lib=$ROOT/variants/libgoldilocks_amd_$v.so; [ $v = product ] && lib=$ROOT/libgoldilocks_amd/libgoldilocks_amd.so
    echo "== $v" >> "$OUT/e2e_verify.txt"
    GOLDILOCKS_AMD_LIB=$lib timeout 200 python tests/e2e_trace_probe.py 2>&1 | grep call >> "$OUT/e2e_verify.txt"
  done
done
cat "$OUT/e2e_verify.txt"
