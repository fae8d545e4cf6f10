#!/bin/bash
# round-4 GPU call 13: BASELINE config 5's per-GPU share (2^21 verifications of 2^10 keys in one call) through bench.py,
# and the same as 8 self-launched ranks of 2^18 each on this box's one device (the invocation the scaling run uses)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call13
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python bench.py --workload verify --log2-batch 21 --steps 10 --warmup 3 --no-cpu-baseline --no-configs --no-end-to-end > "$OUT/bench_verify_2p21.json" 2> "$OUT/err1.txt"; cut -c1-330 "$OUT/bench_verify_2p21.json"
timeout 900 python bench.py --gpus 8 --workload verify --global-log2-batch 21 --steps 5 --warmup 2 > "$OUT/bench_verify_8ranks.json" 2> "$OUT/err2.txt"; python - "$OUT/bench_verify_8ranks.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["n_gpus"], d["scaling"], round(d["value"]/1e6,2), "M/s", d["config"]["control_plane"], [ (g["rank"], g["device"], g["slice"], round(g["value"]/1e6,1)) for g in d["per_gpu"]])
PY
