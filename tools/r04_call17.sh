#!/bin/bash
# round-4 GPU call 17: BASELINE config 4 with every signature a signature of its own (bench.py drew 4 096 distinct
# signatures 2^20 times until now: repeated S and challenges, i.e. table entries already in the caches): 16-bit and
# 24-bit base tables, resident and from host arrays; tests/key_pool_probe.py (always distinct signatures) beside it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call17
mkdir -p "$OUT"
cd "$ROOT"
for rep in 1 2; do
  for bits in 16 20 22 24; do
    GOLDILOCKS_AMD_BASE_TABLE_BITS=$bits timeout 300 python bench.py --workload verify --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-end-to-end 2>/dev/null \
      | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bits', $bits, '->', l['config']['base_table_bits'], '%.1f M/s' % (l['value']/1e6), 'kernel %.3f ms' % l['roofline']['kernel_ms_avg'], 'mac_frac %.3f' % l['roofline']['mac']['frac'], l['config']['parity_spot_check'])" | tee -a "$OUT/bench_verify_widths.txt"
  done
done
for bits in 16 24; do
  echo "== base table: $bits bits" | tee -a "$OUT/key_pool_probe_quick.txt"
  GOLDILOCKS_AMD_BASE_TABLE_BITS=$bits timeout 600 python tests/key_pool_probe.py --quick 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/key_pool_probe_quick.txt"
done
for bits in 16 0; do
  GOLDILOCKS_AMD_BASE_TABLE_BITS=$bits timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs 2>/dev/null \
    | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=l['end_to_end']; print('bits', $bits, json.dumps(e['verify']))" | tee -a "$OUT/e2e_verify_widths.txt"
done
