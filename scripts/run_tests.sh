#!/bin/bash
# Batch runner for the GPU CLI, in the spirit of the reference's scripts/basic_test.batch:9-18 and
# scripts/run_tests:11-68: runs the same invocations (minus salloc/mpirun) through host/apm_parallel
# and diffs ONLY the `Number of matches` lines (the parity surface; banner and timing lines differ by
# design) against tests/golden/expected/*.txt, which oracle/gen_golden.py wrote from the reference
# binary.  Needs a GPU.  Usage: scripts/run_tests.sh [iterations]   (exit code = number of failures)
root=$(cd "$(dirname "$0")/.." && pwd)
apm_executable=${APM_EXECUTABLE:-$root/inf560-approximate-pattern-matching_amd/host/apm_parallel}
data_dir=${APM_DATA_DIR:-$root/tests/golden/dna}
expected_dir=$root/tests/golden/expected
iterations=${1:-3}
failures=0

[ -x "$apm_executable" ] || make -C "$root/inf560-approximate-pattern-matching_amd" all || exit 99

validate() { # <name> <expected file> <output file>
    if diff <(grep '^Number of matches' "$3") "$2" >/dev/null; then
        echo -e "\033[0;32m$1: result OK\033[0m"
    else
        echo -e "\033[0;31m$1: fail\033[0m"
        diff <(grep '^Number of matches' "$3") "$2"
        failures=$((failures + 1))
    fi
}

out=$(mktemp)
trap 'rm -f "$out"' EXIT
P_NONE=$(cat "$data_dir/line_non_existent.fa")
P_20783=$(cat "$data_dir/line_20783.fa")
P_10=$(cat "$data_dir/line_10.fa")
P_20=$(cat "$data_dir/line_20.fa")

echo "Running basic test (three spellings of the reference's batch file)"
for flag in "" PATTERNS_OVER_RANKS DB_OVER_RANKS; do
    "$apm_executable" 0 "$data_dir/small_chrY_x100.fa" $P_NONE $P_20783 $P_20783 $P_20783 $P_20783 $P_20783 $flag > "$out"
    validate "basic_test ${flag:-SEQUENTIAL}" "$expected_dir/basic_test.txt" "$out"
done

for ((i = 1; i <= iterations; i++)); do
    echo "Test iteration $i"
    "$apm_executable" 0 "$data_dir/easy.fa" 123 456 78934 PATTERNS_OVER_RANKS > "$out"
    validate "easy input" "$expected_dir/easy.txt" "$out"
    "$apm_executable" 0 "$data_dir/small_chrY_x100.fa" $P_10 $P_20 $P_NONE $P_10 $P_20 $P_NONE PATTERNS_OVER_RANKS > "$out"
    validate "complex input" "$expected_dir/complex.txt" "$out"
done
exit $failures
