#!/bin/bash
# round-4 GPU call 9: the entries kernel's segment: adaptive (8 .. 64 by the number of keys: product) against fixed 16
# (round 3), fixed 8, and 4 .. 64; on 2^10 keys (latency) and 2^13 .. 2^16 keys (throughput)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call9
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do
  timeout 500 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify.txt" 2>&1
done
cat "$OUT/sweep_verify.txt"
for v in product seg16fixed seg8fixed seg4; do
  lib=$ROOT/variants/libgoldilocks_amd_$v.so; [ $v = product ] && lib=$ROOT/libgoldilocks_amd/libgoldilocks_amd.so
  echo "== $v" >> "$OUT/key_counts.txt"
  GOLDILOCKS_AMD_LIB=$lib timeout 300 python tests/comb_seg_probe.py >> "$OUT/key_counts.txt" 2>&1
done
cat "$OUT/key_counts.txt"
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -k "verif or pipeline or config5 or ten_thousand" 2>&1 | tail -3
