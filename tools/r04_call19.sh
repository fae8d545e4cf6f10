#!/bin/bash
# round-4 GPU call 19: the keys' kernels (teeth, entries) at wave priority 3, so that S*B running beside them does not
# take their issue slots -- with 2 / 3 / 4 rounds of S*B ahead; and the step's timeline with it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call19
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2 3; do
  timeout 900 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify_prio.txt" 2>&1
done
cat "$OUT/sweep_verify_prio.txt"
for v in product prio3a4; do
  lib=$ROOT/variants/libgoldilocks_amd_$v.so; [ $v = product ] && lib=$ROOT/libgoldilocks_amd/libgoldilocks_amd.so
  ( cd /tmp && export TMPDIR=/tmp && GOLDILOCKS_AMD_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$v" -- python3 "$ROOT/bench.py" --workload verify --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/trace_$v.log" 2>&1 )
  echo "== $v"; python tools/trace_timeline.py "$OUT/trace_$v" k_verify_dedupe | tee "$OUT/timeline_$v.txt"
  rm -rf "$OUT/trace_$v"
done
