#!/bin/bash
# round-4 GPU call 25: the keys' combs built comb by comb -- the entries of comb j on a stream of their own while the
# teeth of comb j + 1 are still being doubled: parity, the step's timeline, config 4 and config 5's share
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call25
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_fullsize.py tests/test_gpu_base_table.py -x -q -k "verif or config5 or pipeline or keys or ten_thousand or release" 2>&1 | tail -5 | tee "$OUT/gputest.txt"
for lb in 20 21; do
  timeout 600 python bench.py --workload verify --log2-batch $lb --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-end-to-end 2>/dev/null \
    | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lb', '%.1f M/s' % (l['value']/1e6), 'kernel %.3f ms' % l['roofline']['kernel_ms_avg'], l['roofline']['kernel'], l['config']['parity_spot_check'])" | tee -a "$OUT/bench_verify.txt"
done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --workload verify --steps 4 --warmup 2 --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/trace.log" 2>&1 )
python tools/trace_timeline.py "$OUT/trace" k_verify_dedupe | tee "$OUT/timeline.txt"
rm -rf "$OUT/trace"
timeout 600 python tests/key_pool_probe.py --quick 2>&1 | grep -v amdgpu.ids | tee "$OUT/key_pool_probe_quick.txt"
