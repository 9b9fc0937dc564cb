#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_order_by.py --durations=3 > $O/r3_sort_tests.log 2>&1 || { tail -60 $O/r3_sort_tests.log; exit 1; }
tail -6 $O/r3_sort_tests.log
python3 scripts/bench_operators.py next > $O/r3_sort_bench.txt 2>&1 || { tail -20 $O/r3_sort_bench.txt; exit 1; }
cat $O/r3_sort_bench.txt

