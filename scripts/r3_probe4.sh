#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest -x -q -m gpu tests/test_gpu_fused.py tests/test_gpu_agg_fuzz.py "tests/test_gpu_fullsize.py::test_grouped_aggregation_over_a_full_size_page" > $O/r3_p4_tests.log 2>&1 || { tail -40 $O/r3_p4_tests.log; exit 1; }
tail -3 $O/r3_p4_tests.log
timeout -k 10 120 python3 scripts/micro/op_create.py > $O/r3_op_create.txt 2>&1; cat $O/r3_op_create.txt
AGG_GROUPS=3000000,1000000 timeout -k 10 240 python3 scripts/bench_operators.py agg > $O/r3_agg_d.txt 2>&1; cat $O/r3_agg_d.txt
