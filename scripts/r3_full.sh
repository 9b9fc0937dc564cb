#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 1000 python -m pytest -q -m gpu tests --durations=12 > $O/r3_full_tests.log 2>&1 || { tail -60 $O/r3_full_tests.log; exit 1; }
tail -40 $O/r3_full_tests.log
