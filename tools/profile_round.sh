#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#     gpurun --timeout 1500 -- 'bash tools/profile_round.sh r02'
# One `rocprofv3 --kernel-trace --stats` run of the SAME command the driver times (bench.py defaults)
# and of each workload, then SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters; never
# combined with a trace domain).  Raw output goes to gpurun_out/prof/; the summaries that get
# committed are written to gpurun_out/profiles_<round>/ by tools/summarize_prof.py (copy them to
# profiles/<round>/).  The program after `--` is always python3 itself (no env/bash hop).
set -u
ROUND=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
DST=$ROOT/gpurun_out/profiles_$ROUND
rm -rf "$OUT" "$DST"
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py"
II="--table-access index-independent"

# the driver's command: headline + configs + cpu_baseline on one line
python3 "$BENCH" > "$DST/bench_default.json" 2> "$OUT/bench_default.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -- \
    python3 "$BENCH" --steps 5 --warmup 1 > "$OUT/stats_default.log" 2>&1

for wl in varbase fixed base verify sign x448 direct; do
    python3 "$BENCH" --workload $wl --no-cpu-baseline --no-configs > "$DST/bench_$wl.json" 2> "$OUT/bench_$wl.err"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -- \
        python3 "$BENCH" --workload $wl --steps 5 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/stats_$wl.log" 2>&1
done
for wl in varbase base sign direct; do   # the library's default: index-independent table access
    python3 "$BENCH" --workload $wl $II --no-cpu-baseline --no-configs > "$DST/bench_${wl}_index_independent.json" 2> "$OUT/bench_${wl}_ii.err"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_varbaseii" -- \
    python3 "$BENCH" --workload varbase $II --steps 5 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/stats_varbaseii.log" 2>&1

pmc() {   # pmc <tag> <workload> <extra bench args...> -- <counters...>
    local tag=$1 wl=$2; shift 2
    local extra=()
    while [ "$1" != "--" ]; do extra+=("$1"); shift; done
    shift
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_${tag}_$wl" -- \
        python3 "$BENCH" --workload $wl "${extra[@]}" --steps 2 --warmup 1 --no-cpu-baseline --no-configs > "$OUT/pmc_${tag}_$wl.log" 2>&1
}
for wl in varbase fixed base verify; do
    pmc FETCH $wl -- FETCH_SIZE
    pmc WRITE $wl -- WRITE_SIZE
done
pmc FETCHII varbase $II -- FETCH_SIZE
pmc WRITEII varbase $II -- WRITE_SIZE
for wl in varbase verify; do
    pmc SQ1 $wl -- SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES
    pmc SQ2 $wl -- SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
done
pmc SQ1II varbase $II -- SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES
pmc SQ2II varbase $II -- SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
pmc GRBM varbase -- GRBM_GUI_ACTIVE

for probe in gpu_probe wave_probe ladder_probe base_double_probe ct_base_probe direct_probe single_call_probe encode_probe crossover_probe; do
    python3 "$ROOT/tests/$probe.py" > "$DST/$probe.txt" 2>&1
done
"$ROOT/tools/fieldbench" > "$DST/fieldbench.txt" 2>&1
python3 "$ROOT/tests/batch_sweep.py" > "$DST/batch_sweep.txt" 2>&1
python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$DST"
ls -la "$DST"
