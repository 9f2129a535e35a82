#!/bin/bash
# Build container: bash tools/build_variant.sh <name> <-D flags...>  ->  inf560-approximate-pattern-matching_amd/libapm_var_<name>.so
# An A/B build of the library that differs from the product in apm_sieve.hip's compile-time knobs only (the other
# objects are the product's); picked up on the GPU box through APM_LIB_PATH (tools/ab_libs.sh).  Never shipped.
set -e
NAME=$1; shift
P=$(dirname "$0")/../inf560-approximate-pattern-matching_amd
make -s -C "$P" lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -Wno-unused-value "$@" -c "$P/csrc/apm_sieve.hip" -o "$P/csrc/apm_sieve.var_$NAME.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$P/libapm_var_$NAME.so" "$P/csrc/apm_kernels.o" "$P/csrc/apm_bitpar_wide.o" "$P/csrc/apm_bitlong.o" "$P/csrc/apm_nfa.o" "$P/csrc/apm_sieve.var_$NAME.o" "$P/csrc/apm_runtime.o" "$P/csrc/apm_refshim.o" -ldl -lpthread
rm -f "$P/csrc/apm_sieve.var_$NAME.o"
echo "built libapm_var_$NAME.so"
