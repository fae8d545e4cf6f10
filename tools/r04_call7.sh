#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call7
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do for l in 18 19 20; do
  echo "== later chunks 2^$l" >> "$OUT/e2e_chunks.txt"
  GOLDILOCKS_AMD_VERIFY_LATER_CHUNK_LOG2=$l timeout 200 python tests/e2e_trace_probe.py 2>&1 | grep "^call" >> "$OUT/e2e_chunks.txt"
done; done
cat "$OUT/e2e_chunks.txt"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "pipeline or config5 or repeated_keys or ten_thousand" 2>&1 | tail -3
