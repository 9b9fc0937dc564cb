#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
PA_FUZZ_SEEDS=${1:-12} timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_varchar_keys.py "tests/test_gpu_agg_fuzz.py::test_random_min_max_over_strings" --durations=5 > $O/r3_minmax_tests.log 2>&1 || { tail -80 $O/r3_minmax_tests.log; exit 1; }
tail -9 $O/r3_minmax_tests.log
