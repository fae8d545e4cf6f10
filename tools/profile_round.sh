#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#     gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01'
# For each bench workload: one `rocprofv3 --kernel-trace --stats` run, then SEPARATE --pmc passes
# (FETCH_SIZE, WRITE_SIZE, SQ instruction counters; never combined with a trace domain).  Raw output
# goes to gpurun_out/prof/<tag>/; the summaries that get committed are written to
# gpurun_out/profiles_<round>/ by tools/summarize_prof.py (copy them to profiles/<round>/).
set -u
ROUND=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT" "$ROOT/gpurun_out/profiles_$ROUND"
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles_$ROUND"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py"
for wl in varbase fixed base verify sign x448 direct; do
    python3 "$BENCH" --workload $wl > "$ROOT/gpurun_out/profiles_$ROUND/bench_$wl.json" 2> "$OUT/bench_$wl.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -- \
        python3 "$BENCH" --workload $wl --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/stats_$wl.log" 2>&1
done
for wl in sign base; do   # the index-independent (LDS comb + shuffle gather) variants, bench line only
    python3 "$BENCH" --workload $wl --table-access index-independent --no-cpu-baseline \
        > "$ROOT/gpurun_out/profiles_$ROUND/bench_${wl}_index_independent.json" 2> "$OUT/bench_${wl}_ii.err"
done
for wl in varbase fixed verify; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $ctr --output-format csv -d "$OUT/pmc_${ctr}_$wl" -- \
            python3 "$BENCH" --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_${ctr}_$wl.log" 2>&1
    done
done
for wl in varbase verify x448; do
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d "$OUT/pmc_SQ1_$wl" -- \
        python3 "$BENCH" --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_SQ1_$wl.log" 2>&1
done
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_SQ2_varbase" -- \
    python3 "$BENCH" --workload varbase --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_SQ2_varbase.log" 2>&1
"$ROOT/tools/fieldbench" > "$ROOT/gpurun_out/profiles_$ROUND/fieldbench.txt" 2>&1
python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$ROOT/gpurun_out/profiles_$ROUND"
ls -la "$ROOT/gpurun_out/profiles_$ROUND"
