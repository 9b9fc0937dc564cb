#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest -x -q -m gpu tests/test_gpu_small_pages.py tests/test_gpu_fused.py > $O/r3_p7_tests.log 2>&1 || { tail -60 $O/r3_p7_tests.log; exit 1; }
tail -3 $O/r3_p7_tests.log
for layout in shuffled; do
  timeout -k 10 200 scripts/page_sweep --sf 100 --steps 3 --layout $layout --rows 1048576,65536,8192 > $O/r3_sweep_$layout.txt 2>&1 || { tail -20 $O/r3_sweep_$layout.txt; exit 1; }
  cat $O/r3_sweep_$layout.txt
done
