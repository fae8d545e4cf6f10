#!/bin/bash
# round-4 GPU call 15: the base point's window table with a width chosen at run time (goldilocks_amd_set_base_table_bits,
# GOLDILOCKS_AMD_BASE_TABLE_BITS; built at first use by k_build_bwt, 64 entries per lane and inversion): parity at every
# width, time to build, BASELINE config 4 through bench.py at 16 / 20 / 22 / 24 bits and the default
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call15
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_base_table.py -x -q 2>&1 | tail -15 | tee "$OUT/gputest_base_table.txt"
timeout 600 python tests/base_table_probe.py 2>&1 | tee "$OUT/base_table_probe.txt"
for rep in 1 2; do
  for bits in 16 20 22 24 0; do
    GOLDILOCKS_AMD_BASE_TABLE_BITS=$bits timeout 300 python bench.py --workload verify --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-end-to-end 2>/dev/null \
      | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bits', $bits, '->', l['config']['base_table_bits'], '%.1f M/s' % (l['value']/1e6), 'kernel %.3f ms' % l['roofline']['kernel_ms_avg'], 'mac_frac %.3f' % l['roofline']['mac']['frac'], l['config']['parity_spot_check'])" | tee -a "$OUT/bench_verify_widths.txt"
  done
done
