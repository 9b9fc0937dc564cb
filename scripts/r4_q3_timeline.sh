#!/bin/bash
# kernel timeline of one Q3 step (scripts/q3_native, C++ Driver loop) under rocprofv3 --kernel-trace
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_q3_tl -- $R/scripts/q3_native --sf 100 --steps 4 --warmup 1 > $O/r4_q3_tl.json 2> $O/r4_q3_tl.err
python3 $R/scripts/kernel_timeline.py $O/r_q3_tl pa_fp_count 2 > $O/r4_q3_timeline.txt
cp $(ls -t $O/r_q3_tl/*/*_kernel_stats.csv | head -1) $O/r4_q3_kernel_stats.csv
rm -rf $O/r_q3_tl
cut -c1-150 $O/r4_q3_timeline.txt
