#!/bin/bash
# Same-box A/B of library variants (variants/*.so against the product) over a list of bench workloads, run twice in
# alternation so that a drift of the box shows up as a difference between the two passes:
#     gpurun -- 'bash tools/ab_sweep.sh <out file> "varbase" "verify" "fixed" ...'
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for pass in 1 2; do
    for wl in "$@"; do
        python tools/probes/variant_sweep.py "--workload $wl" >> "$OUT" 2>&1
    done
done
cat "$OUT"
