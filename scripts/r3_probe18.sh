#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_join_d
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_join_d -- python3 $R/scripts/bench_operators.py join > $O/r3_join_d.txt 2> $O/r3_join_d.err
cat $O/r3_join_d.txt
