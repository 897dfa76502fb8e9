#!/bin/bash
# Which lines and branches of the CPU oracle (= the restated reference path) does the test suite execute?  gcov over oracle/zl_oracle.c under
# both test tiers.  Run on a GPU box (scripts/oracle_coverage.sh) or here with TIERS='not gpu'.  Output: gpurun_out/oracle_cov/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
W=/tmp/zl_cov; rm -rf $W; mkdir -p $W $R/gpurun_out/oracle_cov
cp $R/oracle/_build/libzl_oracle.so $W/orig.so
( cd $W && gcc -std=c11 -O0 --coverage -ffp-contract=off -fPIC -c $R/oracle/zl_oracle.c -o $W/zl_oracle.o && gcc -shared --coverage -o $R/oracle/_build/libzl_oracle.so $W/zl_oracle.o -lm -lpthread ) || exit 1
if [ -n "$TIERS" ]; then tiers=("$TIERS"); else tiers=("gpu" "not gpu"); fi      # TIERS='not gpu': one tier only
for t in "${tiers[@]}"; do
  ( cd $R && timeout -k 10 900 python -m pytest tests -q -m "$t" 2>&1 | tail -2 )
done
( cd $W && gcov -b -o $W $R/oracle/zl_oracle.c | head -6 | tee $R/gpurun_out/oracle_cov/summary.txt; cp zl_oracle.c.gcov $R/gpurun_out/oracle_cov/ )
cp $W/orig.so $R/oracle/_build/libzl_oracle.so
