#!/bin/bash
# Config 4's kernel timeline for the product and for every variants/*.so, on one box:
#     gpurun -- 'bash tools/timeline_ab.sh gpurun_out/tl'
OUT=${1:-gpurun_out/tl}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in product $(ls "$ROOT"/variants/*.so 2>/dev/null); do
    name=$(basename "$lib" .so); name=${name#libgoldilocks_amd_}
    if [ "$lib" = product ]; then unset GOLDILOCKS_AMD_LIB; else export GOLDILOCKS_AMD_LIB="$lib"; fi
    rm -rf "/tmp/tl_$name"
    rocprofv3 --kernel-trace --output-format csv -d "/tmp/tl_$name" -- python3 "$ROOT/bench.py" --workload verify --steps 4 --warmup 2 > "/tmp/tl_$name.log" 2>&1
    python3 "$ROOT/tools/trace_timeline.py" "/tmp/tl_$name" k_verify_dedupe > "$ROOT/$OUT/timeline_$name.txt" 2>&1
done
cd "$ROOT"
for f in "$OUT"/timeline_*.txt; do echo "== $f"; grep -E "k_verify_dedupe|k_verify_key_teeth |k_verify_base_part|k_verify_key_combs|xwide|finish|step:" "$f"; done
