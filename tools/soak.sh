#!/bin/bash
# Extended soak on the GPU box: the soak and property modules (every kernel in three modes against the oracle, a sample
# against the real reference build) over N further input streams.   gpurun --timeout 3000 -- 'bash tools/soak.sh 60 r04a'
set -u
N=${1:-30}
TAG=${2:-soak}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/soak
mkdir -p "$OUT"
cd "$ROOT"
fail=0
t0=$(date +%s)
for i in $(seq 1 "$N"); do
    if ! GOLDILOCKS_SOAK_SEED="$TAG-$i" timeout 600 python -m pytest tests/test_gpu_soak.py tests/test_gpu_properties.py -x -q -k "not soak_logs_belong" > "$OUT/last.txt" 2>&1; then
        fail=$((fail + 1))
        cp "$OUT/last.txt" "$OUT/fail_$TAG-$i.txt"
    fi
done
python - <<'PY' | tee "$OUT/soak_$TAG.stamp.txt"
import json, sys
sys.path.insert(0, ".")
import libgoldilocks_amd as ga
b = ga.build_info()
print("toolchain: " + b["toolchain"])
print("library_sha256: " + b["library_sha256"])
PY
echo "soak + property modules ($TAG): $N further input streams, $fail failures, $(( $(date +%s) - t0 )) s; last run: $(tail -1 "$OUT/last.txt")" | tee "$OUT/soak_$TAG.txt"
cat "$OUT/soak_$TAG.stamp.txt" >> "$OUT/soak_$TAG.txt"
