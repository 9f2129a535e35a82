#!/bin/bash
# Run ON THE GPU BOX (via gpurun):  bash profiles/pmc.sh <tag> "<counters>" [bench args...]
# One rocprofv3 --pmc pass (counters in their own run, kernel-trace only) over a short bench run.
set -o pipefail
TAG=${1:-pmc}; shift
CTRS=${1}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT" -o pmc -- python3 "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-variants "$@" > "$OUT/stdout.log" 2> "$OUT/stderr.log"
echo "rocprofv3 rc=$?"
ls "$OUT"
