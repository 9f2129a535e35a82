#!/bin/bash
# Run ON THE GPU BOX: bash profiles/pmc_tool.sh <tag> "<counters>" <cfg> [kernel]   (env ABLATIONS, APM_* pass through)
set -o pipefail
TAG=$1; CTRS=$2; CFG=$3; KERN=$4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmct_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT" -o p -- python3 "$REPO/tools/ablate.py" $CFG $KERN > "$OUT/stdout.log" 2> "$OUT/stderr.log"
echo "rc=$?"
python3 "$REPO/tools/pmc_summary.py" "$OUT/p_counter_collection.csv" | grep -A10 "apm_filter\|apm_bitpar\|apm_wavefront" | head -40
