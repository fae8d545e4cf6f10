#!/bin/bash
# What clock and power the chip holds while the headline kernel runs:  gpurun -- 'bash tools/clock_probe.sh > gpurun_out/clock.txt'
# (bench.py's roofline.valu_issue is quoted against the nominal 2.4 GHz; this says what the chip actually holds.)
# rocm-smi is sampled twice a second for the whole life of a 600-step run; every distinct reading of every device is printed once.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
python bench.py --workload varbase --steps 600 --warmup 3 --no-cpu-baseline --no-configs --no-end-to-end > /tmp/clock_probe_bench.json 2>/dev/null &
PID=$!
: > /tmp/clock_samples.txt
while kill -0 $PID 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr -s '\t ' ' ' >> /tmp/clock_samples.txt
    sleep 0.5
done
sort /tmp/clock_samples.txt | uniq -c | sort -k2,2 -k1,1nr | head -60
python tools/gpu_summarise_bench.py /tmp/clock_probe_bench.json | head -1
