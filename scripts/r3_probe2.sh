#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_exchange.py tests/test_gpu_fused.py tests/test_gpu_agg_fuzz.py tests/test_gpu_join.py "tests/test_gpu_fullsize.py::test_grouped_aggregation_over_a_full_size_page" > $O/r3_p2_tests.log 2>&1 || { tail -30 $O/r3_p2_tests.log; exit 1; }
tail -3 $O/r3_p2_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_agg3m_b
AGG_GROUPS=3000000,1000000,300000 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_agg3m_b -- python3 $R/scripts/bench_operators.py agg > $O/r3_agg3m_b.txt 2> $O/r3_agg3m_b.err
cat $O/r3_agg3m_b.txt
