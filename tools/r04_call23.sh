#!/bin/bash
# round-4 GPU call 23: the widest combs by the refined rule (a thousand signatures per key; two thousand beyond 1 024
# keys): library defaults against 8 teeth, bench.py on config 4 and config 5's share
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_call23
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench.py -x -q -k "verif or keys or default_line" 2>&1 | tail -4 | tee "$OUT/gputest.txt"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee "$OUT/xwide_defaults_probe.txt"
import os, sys
sys.path.insert(0, "tests")
import numpy as np, torch, libgoldilocks_amd as ga, _gen
from key_pool_probe_lib import make, timeit
for n, nk in ((1 << 20, 256), (1 << 20, 512), (1 << 20, 1024), (1 << 20, 2048), (1 << 21, 1024), (1 << 21, 2048), (1 << 21, 4096), (1 << 19, 512)):
    sig, pk, msg = make(n, nk)
    st = torch.empty(n, dtype=torch.int32, device="cuda")
    f = lambda: ga.dev("ed448_verify", st.data_ptr(), sig.data_ptr(), pk.data_ptr(), msg.data_ptr(), None, 32, 0, None, 0, n, None)
    ga.set_verify_key_combs_xwide(0); a = timeit(f); ta = ga.last_verify_key_counts(teeth=True)[3]
    ga.set_verify_key_combs_xwide(); b = timeit(f); tb = ga.last_verify_key_counts(teeth=True)[3]
    assert int((st == -1).sum()) == n
    print("n=2^%d keys=%-5d (%6.1f per key)  without the widest: %d teeth %7.3f ms   library default: %d teeth %7.3f ms  %+.1f %%"
          % (n.bit_length() - 1, nk, n / nk, ta, a, tb, b, 100 * (b - a) / a), flush=True)
PY
for lb in 20 21; do
  timeout 600 python bench.py --workload verify --log2-batch $lb --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-end-to-end 2>/dev/null \
    | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2^$lb', '%.1f M/s' % (l['value']/1e6), 'kernel %.3f ms' % l['roofline']['kernel_ms_avg'], l['roofline']['kernel'], 'macs', l['roofline']['mac']['macs_per_op'], 'mac_frac %.3f' % l['roofline']['mac']['frac'], l['config']['parity_spot_check'])" | tee -a "$OUT/bench_verify.txt"
done
