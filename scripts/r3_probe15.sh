#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 500 python -m pytest -x -q -m gpu tests/test_gpu_fused.py -k "tier_hand_overs" --durations=5 > $O/r3_p15_tests.log 2>&1 || { tail -60 $O/r3_p15_tests.log; exit 1; }
tail -8 $O/r3_p15_tests.log
