#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash profiles/collect.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats of a short bench run under gpurun_out/prof_<tag>/.
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o trace -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_stdout.log" 2> "$OUT/bench_stderr.log"
rc=$?
echo "rocprofv3 rc=$rc"
find "$OUT" -name "*stats*.csv" | head
exit $rc
