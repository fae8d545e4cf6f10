#!/bin/bash
# round-4 GPU call 11: the device entry point's first pass finishing in its own launch (fused) against the finish kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call11
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2 3; do
  timeout 400 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify.txt" 2>&1
done
cat "$OUT/sweep_verify.txt"
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_soak.py -x -q -k "verif or pipeline or config5 or ten_thousand or sign" 2>&1 | tail -3
