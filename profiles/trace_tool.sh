#!/bin/bash
# Run ON THE GPU BOX: bash profiles/trace_tool.sh <tag> <cfg> [kernel]  -> kernel-trace stats of tools/ablate.py
set -o pipefail
TAG=$1; CFG=$2; KERN=$3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ABLATIONS=${TRACE_ABL:-0} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$REPO/tools/ablate.py" $CFG $KERN > "$OUT/stdout.log" 2> "$OUT/stderr.log"
echo "rc=$?"; tail -1 "$OUT/stdout.log"; head -8 "$OUT/t_kernel_stats.csv" | cut -c1-150
