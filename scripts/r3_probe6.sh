#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_join.py tests/test_gpu_join_fuzz.py tests/test_gpu_q3_pipeline.py tests/test_gpu_fused_join.py > $O/r3_p6_tests.log 2>&1 || { tail -40 $O/r3_p6_tests.log; exit 1; }
tail -3 $O/r3_p6_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_join_b
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_join_b -- python3 $R/scripts/bench_operators.py join > $O/r3_join_b.txt 2> $O/r3_join_b.err
cat $O/r3_join_b.txt
