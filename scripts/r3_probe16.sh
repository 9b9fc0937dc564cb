#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 800 python -m pytest -x -q -m gpu tests/test_gpu_topn.py tests/test_gpu_sort_fuzz.py tests/test_gpu_q3_pipeline.py tests/test_gpu_real.py --durations=6 > $O/r3_p16_tests.log 2>&1 || { tail -60 $O/r3_p16_tests.log; exit 1; }
tail -10 $O/r3_p16_tests.log
timeout -k 10 240 python3 scripts/bench_q3.py --steps 6 --warmup 2 > $O/r3_q3_g.json 2> $O/r3_q3_g.err
tail -c 600 $O/r3_q3_g.json
