#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_fused_join.py tests/test_gpu_q3_pipeline.py tests/test_gpu_exchange.py tests/test_gpu_fused.py "tests/test_gpu_fullsize.py::test_grouped_aggregation_over_a_full_size_page" > $O/r3_p3_tests.log 2>&1 || { tail -40 $O/r3_p3_tests.log; exit 1; }
tail -3 $O/r3_p3_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_agg3m_c $O/r3_q3_c
AGG_GROUPS=3000000 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_agg3m_c -- python3 $R/scripts/bench_operators.py agg > $O/r3_agg3m_c.txt 2> $O/r3_agg3m_c.err
cat $O/r3_agg3m_c.txt
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_q3_c -- python3 $R/scripts/bench_q3.py --steps 5 --warmup 1 > $O/r3_q3_c.json 2> $O/r3_q3_c.err
cat $O/r3_q3_c.json
cd $R
python3 scripts/kernel_timeline.py $O/r3_q3_c pa_fp_count 2 > $O/r3_q3_c_timeline.txt
