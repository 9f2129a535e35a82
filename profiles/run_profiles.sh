#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/run_profiles.sh <tag> <cfg> [<cfg> ...]
# For every workload: one rocprofv3 kernel-trace/stats pass and three separate --pmc passes (FETCH_SIZE, WRITE_SIZE,
# SQ counters; counters never share a run with a trace domain other than the kernel trace) over a short bench.py run of
# that workload alone.  Summaries land under gpurun_out/prof_<tag>/<cfg>/ ; tools/refresh_profiles.py copies them to
# profiles/<round>/ and recomputes profiles/traffic.json.
set -o pipefail
TAG=${1:-r02}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-variants --no-per-config"
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for CFG in "$@"; do
  OUT=$REPO/gpurun_out/prof_$TAG/$CFG
  mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$REPO/bench.py" --config $CFG $ARGS > "$OUT/bench.json" 2> "$OUT/trace.err" || { echo "trace pass failed for $CFG"; tail -5 "$OUT/trace.err"; exit 1; }
  cp "$(find "$OUT/trace" -name 't_kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
  for PASS in FETCH_SIZE WRITE_SIZE SQ; do
    if [ $PASS = SQ ]; then CTRS="$SQ"; else CTRS=$PASS; fi
    rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pmc_$PASS" -o p -- python3 "$REPO/bench.py" --config $CFG $ARGS > /dev/null 2> "$OUT/pmc_$PASS.err" || { echo "pmc pass $PASS failed for $CFG"; tail -5 "$OUT/pmc_$PASS.err"; exit 1; }
    python3 "$REPO/tools/pmc_summary.py" "$(find "$OUT/pmc_$PASS" -name 'p_counter_collection.csv' | head -1)" apm_ > "$OUT/pmc_$PASS.txt"
    rm -rf "$OUT/pmc_$PASS"
  done
  rm -rf "$OUT/trace"
  echo "== $CFG"; python3 "$REPO/tools/trace_summary.py" "$OUT/kernel_stats.csv"; grep -h -A1 "FETCH_SIZE\|WRITE_SIZE" "$OUT/pmc_FETCH_SIZE.txt" "$OUT/pmc_WRITE_SIZE.txt" | head -20
done
