#!/bin/bash
# round-4 GPU call 18: BASELINE config 5's per-GPU share (2^21 signatures of 2^10 keys) and the eight-rank invocation on
# one device with the final library and batch; a verification step behind an idle device (tests/idle_gap_probe.py); the
# host-array call's laps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call18
mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python bench.py --workload verify --log2-batch 21 --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-end-to-end 2>/dev/null | tail -1 > "$OUT/bench_verify_2p21.json"
python -c "import json; l=json.load(open('$OUT/bench_verify_2p21.json')); print('2^21:', '%.1f M/s' % (l['value']/1e6), '%.3f ms' % l['roofline']['kernel_ms_avg'], l['config']['parity_spot_check'], l['config']['base_table_bits'])"
timeout 900 python bench.py --gpus 8 --workload verify --global-log2-batch 21 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_verify_8ranks_one_device.json"
python -c "import json; l=json.load(open('$OUT/bench_verify_8ranks_one_device.json')); print('8 ranks:', '%.1f M/s' % (l['value']/1e6), [round(g['value']/1e6,1) for g in l['per_gpu']], l['config']['parity_spot_check'])"
timeout 600 python tests/idle_gap_probe.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/idle_gap_probe.txt"
GOLDILOCKS_AMD_TRACE=1 timeout 600 python tests/e2e_trace_probe.py 2>&1 | grep -v amdgpu.ids | tail -40 > "$OUT/e2e_laps.txt"; tail -20 "$OUT/e2e_laps.txt"
