#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 400 python -m pytest -x -q -m gpu tests/test_gpu_fused.py tests/test_gpu_agg_fuzz.py tests/test_gpu_q3_pipeline.py tests/test_gpu_fused_join.py tests/test_gpu_real.py tests/test_gpu_varchar_keys.py tests/test_gpu_partial_final.py tests/test_gpu_states.py tests/test_gpu_nullability.py "tests/test_gpu_fullsize.py::test_grouped_aggregation_over_a_full_size_page" > $O/r3_agg_tests.log 2>&1 || { tail -40 $O/r3_agg_tests.log; exit 1; }
tail -3 $O/r3_agg_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_agg_e
AGG_GROUPS=3000000,1000000,100000,20000 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_agg_e -- python3 $R/scripts/bench_operators.py agg > $O/r3_agg_e.txt 2> $O/r3_agg_e.err
cat $O/r3_agg_e.txt
f=$(ls $O/r3_agg_e/*/*_kernel_stats.csv | head -1)
head -14 $f | cut -c1-150
