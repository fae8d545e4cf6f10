#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#     gpurun --timeout 2400 -- 'bash tools/profile_round.sh r04'
# One `rocprofv3 --kernel-trace --stats` run of the SAME command the driver times (bench.py defaults)
# and of each workload, then SEPARATE --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters; never
# combined with a trace domain).  Raw output goes to gpurun_out/prof/; the summaries that get
# committed are written to gpurun_out/profiles_<round>/ by tools/summarize_prof.py (copy them to
# profiles/<round>/).  The program after `--` is always python3 itself (no env/bash hop).
set -u
ROUND=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
DST=$ROOT/gpurun_out/profiles_$ROUND
rm -rf "$OUT" "$DST"
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py"
FAST="--table-access fast"            # the opt-in for public scalars; bench.py's default is the library's (index-independent)
Q="--no-cpu-baseline --no-configs --no-end-to-end"

# the driver's command: headline + configs + cpu_baseline on one line (written at the end, with this library's counters)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_default" -- \
    python3 "$BENCH" --steps 5 --warmup 1 > "$OUT/stats_default.log" 2>&1
# ... and without its end_to_end leg (which runs the same kernels in one-residency chunks): per-kernel averages of
# full-size launches only, comparable with the line's kernel_ms_avg figures
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_defaultresident" -- \
    python3 "$BENCH" --steps 5 --warmup 1 --no-end-to-end > "$OUT/stats_defaultresident.log" 2>&1

for wl in varbase fixed base verify verify_distinct sign x448 direct; do
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$wl" -- \
        python3 "$BENCH" --workload $wl --steps 5 --warmup 1 $Q > "$OUT/stats_$wl.log" 2>&1
done
pmc() {   # pmc <tag> <workload> <extra bench args...> -- <counters...>
    local tag=$1 wl=$2; shift 2
    local extra=()
    while [ "$1" != "--" ]; do extra+=("$1"); shift; done
    shift
    rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_${tag}_$wl" -- \
        python3 "$BENCH" --workload $wl "${extra[@]}" --steps 2 --warmup 1 $Q > "$OUT/pmc_${tag}_$wl.log" 2>&1
}
for wl in varbase fixed base verify verify_distinct; do
    pmc FETCH $wl -- FETCH_SIZE
    pmc WRITE $wl -- WRITE_SIZE
done
pmc FETCHFAST base $FAST -- FETCH_SIZE
pmc WRITEFAST base $FAST -- WRITE_SIZE
for wl in varbase verify verify_distinct; do
    pmc SQ1 $wl -- SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES
    pmc SQ2 $wl -- SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
done
# config 4 at the opt-in 24-bit base table (28.5 GiB): its own passes, kept as run "verify24" (the table's width comes from the
# environment, which the profiled python inherits: no env hop behind rocprofv3)
export GOLDILOCKS_AMD_BASE_TABLE_BITS=24
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_FETCH_verify24" -- python3 "$BENCH" --workload verify --steps 2 --warmup 1 $Q > "$OUT/pmc_FETCH_verify24.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_WRITE_verify24" -- python3 "$BENCH" --workload verify --steps 2 --warmup 1 $Q > "$OUT/pmc_WRITE_verify24.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d "$OUT/pmc_SQ1_verify24" -- python3 "$BENCH" --workload verify --steps 2 --warmup 1 $Q > "$OUT/pmc_SQ1_verify24.log" 2>&1
unset GOLDILOCKS_AMD_BASE_TABLE_BITS
for wl in fixed base sign x448 direct; do       # the ceiling that binds, for every workload of the line
    pmc SQ1 $wl -- SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES
done
pmc SQ1FAST base $FAST -- SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES
pmc GRBM varbase -- GRBM_GUI_ACTIVE

for probe in gpu_probe h2d_probe wave_probe key_pool_probe wide_comb_probe small_batch_probe ct_varbase_probe base_double_probe ct_base_probe direct_probe single_call_probe encode_probe crossover_probe; do
    python3 "$ROOT/tools/probes/$probe.py" > "$DST/$probe.txt" 2>&1
done
# round 4: which kernels of a verification step overlap (S*B beside the combs' build); the host-array verification's laps;
# the host feed rate of the 8-shard host-array call without a device
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_verify" -- \
    python3 "$BENCH" --workload verify --steps 4 --warmup 2 $Q > "$OUT/trace_verify.log" 2>&1
python3 "$ROOT/tools/trace_timeline.py" "$OUT/trace_verify" k_verify_dedupe > "$DST/timeline_verify.txt" 2>&1
GOLDILOCKS_AMD_TRACE=1 python3 "$ROOT/tools/probes/e2e_trace_probe.py" > "$DST/e2e_laps.txt" 2>&1
for o in sequential scattered; do for s in none memcpy; do "$ROOT/tools/hostfeed" --log2n 22 --stage $s --order $o; done; done > "$DST/hostfeed.txt" 2>&1
"$ROOT/tools/fieldbench" > "$DST/fieldbench.txt" 2>&1
"$ROOT/tools/stepbench" > "$DST/stepbench.txt" 2>&1
"$ROOT/tools/fp64gate" > "$DST/fp64gate.txt" 2>&1
[ -x "$ROOT/tools/combsphases" ] && "$ROOT/tools/combsphases" > "$DST/combsphases.txt" 2>&1
[ -x "$ROOT/tools/verifyphases" ] && "$ROOT/tools/verifyphases" > "$DST/verifyphases.txt" 2>&1
"$ROOT/tools/keycombphases" > "$DST/keycombphases.txt" 2>&1   # 9 teeth, 20-bit base table, XCD-aware positions: the product's geometry for config 4
python3 "$ROOT/tools/probes/batch_sweep.py" > "$DST/batch_sweep.txt" 2>&1
python3 "$ROOT/tools/summarize_prof.py" "$OUT" "$DST"
# the bench lines once more, now that the counters of THIS library are in place (bench.py reports `traffic` and `valu_issue`
# only from counters stamped with the current kernel sources: the lines above were written before there were any)
cp "$DST/pmc_traffic.json" "$ROOT/profiles/pmc_traffic.json"
python3 "$BENCH" > "$DST/bench_default.json" 2> "$OUT/bench_default.err"
for wl in varbase fixed base verify verify_distinct sign x448 direct; do
    python3 "$BENCH" --workload $wl $Q > "$DST/bench_$wl.json" 2> "$OUT/bench_$wl.err"
done
for wl in base sign; do
    python3 "$BENCH" --workload $wl $FAST $Q > "$DST/bench_${wl}_fast.json" 2> "$OUT/bench_${wl}_fast.err"
done
ls -la "$DST"
