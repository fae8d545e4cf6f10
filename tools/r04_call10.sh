#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call10
mkdir -p "$OUT"
cd "$ROOT"
timeout 400 python tests/key_pool_probe.py --quick > "$OUT/key_pool_probe.txt" 2>&1; cat "$OUT/key_pool_probe.txt"
timeout 400 python tests/key_pool_probe.py --small > "$OUT/key_pool_probe_small.txt" 2>&1; cat "$OUT/key_pool_probe_small.txt"
