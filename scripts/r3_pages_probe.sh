#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
PA_FUZZ_SEEDS=${1:-40} timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_small_pages.py --durations=4 > $O/r3_pages_tests.log 2>&1 || { tail -60 $O/r3_pages_tests.log; exit 1; }
tail -7 $O/r3_pages_tests.log
