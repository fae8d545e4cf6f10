#!/bin/bash
# round-4 GPU call 8: two accumulator chains in the wave kernels' row multiplication (teeth kernel, single calls), the
# entries kernel's segment size (8 / 16 / 32 entries per lane), x1 as the second operand in the ladder step
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call8
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | tee "$OUT/gputest_wave.txt"
for rep in 1 2; do
  timeout 500 python tests/variant_sweep.py "--workload verify --steps 20 --warmup 5" >> "$OUT/sweep_verify.txt" 2>&1
done
cat "$OUT/sweep_verify.txt"
timeout 300 python tests/variant_sweep.py "--workload varbase --steps 10 --warmup 3" product onechain > "$OUT/sweep_varbase.txt" 2>&1; cat "$OUT/sweep_varbase.txt"
for v in product onechain; do
  lib=$ROOT/variants/libgoldilocks_amd_$v.so; [ $v = product ] && lib=$ROOT/libgoldilocks_amd/libgoldilocks_amd.so
  echo "== $v" >> "$OUT/wave_probe.txt"
  GOLDILOCKS_AMD_LIB=$lib timeout 300 python tests/wave_probe.py >> "$OUT/wave_probe.txt" 2>&1
done
cat "$OUT/wave_probe.txt"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_verify" -- python3 "$ROOT/bench.py" --workload verify --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/trace_verify.log" 2>&1 )
python tools/trace_timeline.py "$OUT/trace_verify" k_verify_dedupe | tee "$OUT/timeline_verify.txt"
rm -rf "$OUT/trace_verify"
