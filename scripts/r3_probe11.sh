#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 500 python -m pytest -x -q -m gpu tests/test_gpu_topn.py tests/test_gpu_sort_fuzz.py tests/test_gpu_real.py tests/test_gpu_q3_pipeline.py --durations=5 > $O/r3_p11_tests.log 2>&1 || { tail -60 $O/r3_p11_tests.log; exit 1; }
tail -3 $O/r3_p11_tests.log
timeout -k 10 300 python3 scripts/bench_operators.py next > $O/r3_next_a.txt 2>&1 || { tail -20 $O/r3_next_a.txt; exit 1; }
cat $O/r3_next_a.txt
timeout -k 10 240 python3 scripts/bench_q3.py --steps 5 --warmup 2 > $O/r3_q3_f.json 2> $O/r3_q3_f.err
tail -c 500 $O/r3_q3_f.json
