#!/bin/bash
# round-4 GPU call 14: the base point's window table with wider digits (18 / 20 / 22 / 24 bits: 0.6 / 2.2 / 7.9 / 28.5 GiB
# in HBM instead of 168 MiB in the Infinity Cache; 25 / 23 / 21 / 19 additions instead of 28): verification on 2^10 keys,
# on distinct keys, the base point's operations with digit-addressed tables, time to build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call14
mkdir -p "$OUT"
cd "$ROOT"
for v in product bwt18 bwt20 bwt22 bwt24; do
  lib=$ROOT/variants/libgoldilocks_amd_$v.so; [ $v = product ] && lib=$ROOT/libgoldilocks_amd/libgoldilocks_amd.so
  echo "== $v" >> "$OUT/init_time.txt"
  GOLDILOCKS_AMD_LIB=$lib timeout 300 python tests/init_time.py >> "$OUT/init_time.txt" 2>&1
done
cat "$OUT/init_time.txt"
for rep in 1 2; do
  timeout 900 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify.txt" 2>&1
done
cat "$OUT/sweep_verify.txt"
timeout 900 python tests/variant_sweep.py "--workload base --table-access fast --steps 20 --warmup 5" > "$OUT/sweep_base_fast.txt" 2>&1; cat "$OUT/sweep_base_fast.txt"
timeout 900 python tests/variant_sweep.py "--workload sign --table-access fast --steps 20 --warmup 5" > "$OUT/sweep_sign_fast.txt" 2>&1; cat "$OUT/sweep_sign_fast.txt"
timeout 900 python tests/variant_sweep.py "--workload verify_distinct --steps 10 --warmup 3" > "$OUT/sweep_verify_distinct.txt" 2>&1; cat "$OUT/sweep_verify_distinct.txt"
