#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
PRESTO_AMD_POOL_SCRUB=0xA5 PA_FUZZ_SEEDS=30 timeout -k 10 900 python -m pytest -x -q -m gpu tests/test_gpu_varchar_keys.py "tests/test_gpu_agg_fuzz.py::test_random_min_max_over_strings" tests/test_gpu_join.py tests/test_gpu_join_fuzz.py > $O/r3_scrub_tests.log 2>&1 || { tail -80 $O/r3_scrub_tests.log; exit 1; }
tail -4 $O/r3_scrub_tests.log
