#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
export PA_FUZZ_SEEDS=${1:-120}
rc=0
for t in tests/test_gpu_agg_fuzz.py tests/test_gpu_join_fuzz.py tests/test_gpu_sort_fuzz.py tests/test_gpu_dynamic_filter.py tests/test_gpu_expr_fuzz.py; do
  n=$(basename $t .py)
  timeout -k 10 280 python -m pytest -q -m gpu $t -x > $O/r3_fuzz_$n.log 2>&1 || { rc=1; tail -30 $O/r3_fuzz_$n.log; }
  tail -1 $O/r3_fuzz_$n.log
done
exit $rc
