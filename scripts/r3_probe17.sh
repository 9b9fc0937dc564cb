#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_q3_h
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_q3_h -- python3 $R/scripts/bench_q3.py --steps 3 --warmup 1 > $O/r3_q3_h.json 2> $O/r3_q3_h.err
cd $R
python3 scripts/kernel_timeline.py $O/r3_q3_h pa_fp_count 2 > $O/r3_q3_h_timeline.txt
grep -n "pa_fused" $O/r3_q3_h_timeline.txt | tail -1
