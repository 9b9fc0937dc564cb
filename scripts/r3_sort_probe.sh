#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
PA_FUZZ_SEEDS=${1:-24} timeout -k 10 800 python -m pytest -x -q -m gpu tests/test_gpu_order_by.py tests/test_gpu_sort_fuzz.py --durations=3 > $O/r3_sort_tests.log 2>&1 || { tail -60 $O/r3_sort_tests.log; exit 1; }
tail -6 $O/r3_sort_tests.log
