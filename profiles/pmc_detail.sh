#!/bin/bash
# Run ON THE GPU BOX: bash profiles/pmc_detail.sh <tag> <cfg>  -- texture-addresser / L1 / L2 / TLB counters of the scan
# kernels of one workload, one --pmc pass per group (kernel trace only beside them; every pass under its own timeout:
# the TA_BUFFER_* group aborted rocprofv3 on this image and left it hanging, so it is not collected).  Output: gpurun_out/prof_<tag>/<cfg>/pmc_detail.txt
set -o pipefail
TAG=${1:-r02}; CFG=${2:-cfg3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-variants --no-per-config"
OUT=$REPO/gpurun_out/prof_$TAG/$CFG
mkdir -p "$OUT"; : > "$OUT/pmc_detail.txt"
i=0
for CTRS in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  echo "pass $i: $CTRS"
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pd_$i" -o p -- python3 "$REPO/bench.py" --config $CFG $ARGS > /dev/null 2> "$OUT/pd_$i.err" || { echo "pass $i ($CTRS) failed"; tail -3 "$OUT/pd_$i.err"; continue; }
  python3 "$REPO/tools/pmc_summary.py" "$(find "$OUT/pd_$i" -name 'p_counter_collection.csv' | head -1)" apm_ | grep -v synth >> "$OUT/pmc_detail.txt"
  rm -rf "$OUT/pd_$i"
done
cat "$OUT/pmc_detail.txt"
