set -o pipefail
mkdir -p gpurun_out/$TAG
P=inf560-approximate-pattern-matching_amd
for v in "" b6 c6 a7p2; do
  for c in cfg3 cfg5 cfg4; do
    if [ -n "$v" ]; then export APM_LIB_PATH=$PWD/$P/libapm_var_$v.so; else unset APM_LIB_PATH; fi
    APM_FUSED=1 timeout -k 10 120 python bench.py --config $c --no-cpu-baseline --no-variants --no-per-config > gpurun_out/$TAG/bench_${c}_f${v}.json 2> gpurun_out/$TAG/bench_${c}_f${v}.err || echo "FAILED $c $v"
  done
done
unset APM_LIB_PATH
for c in cfg3 cfg5 cfg4; do timeout -k 10 120 python bench.py --config $c --no-cpu-baseline --no-variants --no-per-config > gpurun_out/$TAG/bench_${c}_base.json 2>/dev/null; done
python3 tools/show_bench.py gpurun_out/$TAG/bench_*.json
