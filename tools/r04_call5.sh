#!/bin/bash
# round-4 GPU call 5: where the host-array verification's time goes (kernels and copies of one call)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call5
mkdir -p "$OUT"
cd "$ROOT"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace_e2e" -- python3 "$ROOT/tests/e2e_trace_probe.py" > "$OUT/trace_e2e.log" 2>&1 )
tail -5 "$OUT/trace_e2e.log"
python tools/trace_timeline.py "$OUT/trace_e2e" k_verify_dedupe --copies | tee "$OUT/timeline_e2e.txt"
ls "$OUT/trace_e2e"/*/ | head
rm -rf "$OUT/trace_e2e"
