#!/bin/bash
# round-4 GPU call 16: S*B ahead of the verification with the 24-bit base table (19 additions instead of 28: the rounds
# computed ahead cost less -- 1 / 2 (product) / 3 / 4 rounds), the release of the library's device memory, the host-array
# verification with the wider table
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call16
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_base_table.py -x -q 2>&1 | tail -15 | tee "$OUT/gputest_base_table.txt"
for rep in 1 2 3; do
  timeout 900 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify_ahead.txt" 2>&1
done
cat "$OUT/sweep_verify_ahead.txt"
for bits in 16 0; do
  GOLDILOCKS_AMD_BASE_TABLE_BITS=$bits timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null \
    | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=l['end_to_end']; print('bits', $bits, json.dumps({k: e[k] for k in e if k.startswith('verify') or k == 'link_gbs'}))" | tee -a "$OUT/e2e_verify_widths.txt"
done
