# Run ON THE GPU BOX: bash tools/ab_libs.sh <tag> "<cfg ...>" "<variant ...>"   ("" = the product library; a variant
# name picks inf560-approximate-pattern-matching_amd/libapm_var_<name>.so through APM_LIB_PATH).  One box, one call: boxes differ by 5-10 %.
TAG=$1; CFGS=$2; VARS=$3
P=$PWD/inf560-approximate-pattern-matching_amd
mkdir -p gpurun_out/$TAG
for round in 1 2; do
for v in base $VARS; do
  for c in $CFGS; do
    if [ "$v" != base ]; then export APM_LIB_PATH=$P/libapm_var_$v.so; else unset APM_LIB_PATH; fi
    timeout -k 10 120 python bench.py --config $c --no-cpu-baseline --no-variants --no-per-config > gpurun_out/$TAG/bench_${c}_${v}_$round.json 2> gpurun_out/$TAG/err.txt || echo "FAILED $c $v"
    echo "$c $v $round: $(python3 tools/show_bench.py gpurun_out/$TAG/bench_${c}_${v}_$round.json | head -2 | tr '\n' ' ')"
  done
done
done
unset APM_LIB_PATH
