#!/bin/bash
# round-4 GPU call 20: the whole GPU suite with the base point's window table forced to other widths (8 and 22 bits:
# spans 448 and 462 of the recoding; the suite's default run has 24, the soak ran 20)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call20
mkdir -p "$OUT"
cd "$ROOT"
for bits in 8 22; do
  GOLDILOCKS_AMD_BASE_TABLE_BITS=$bits timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_base_table.py > "$OUT/gputest_bits$bits.txt" 2>&1
  echo "base table $bits bits: $(tail -1 "$OUT/gputest_bits$bits.txt")" | tee -a "$OUT/summary.txt"
done
