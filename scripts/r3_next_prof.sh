#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_next
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_next -- python3 $R/scripts/bench_operators.py next > $O/r3_next.txt 2> $O/r3_next.err
f=$(ls $O/r3_next/*/*_kernel_stats.csv | head -1)
head -24 $f | cut -c1-200
rm -rf $O/r3_next
