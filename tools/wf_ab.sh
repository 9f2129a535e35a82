P=$PWD/inf560-approximate-pattern-matching_amd
mkdir -p gpurun_out/r02w2
for v in new old new old; do
  if [ $v = old ]; then export APM_LIB_PATH=$P/libapm_var_oldwf.so; else unset APM_LIB_PATH; fi
  for c in cfg2 cfg3; do
    timeout -k 10 200 python bench.py --config $c --kernel wavefront --bytes-per-gpu 67108864 --no-cpu-baseline --no-variants --no-per-config > gpurun_out/r02w2/b_${c}_$v.json 2>gpurun_out/r02w2/err.txt
    python3 -c "
import json,sys; b=json.load(open('gpurun_out/r02w2/b_${c}_$v.json')); print('$c $v', b['roofline']['kernel_ms_avg'], b['value'])"
  done
done
unset APM_LIB_PATH
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export APM_LIB_PATH=$P/libapm_var_oldwf.so; else unset APM_LIB_PATH; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02w2/pmc_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config cfg2 --kernel wavefront --bytes-per-gpu 67108864 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-per-config > /dev/null 2>&1
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$(find $GRAFT_REPO_ROOT/gpurun_out/r02w2/pmc_$v -name 'p_counter_collection.csv' | head -1)" apm_wavefront
done
