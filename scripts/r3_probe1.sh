#!/bin/bash
# round 3, first measurements: readback round trips, the 3 M-group aggregation's kernels, the new bench line
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 60 scripts/micro/readback 1048576 > $O/r3_readback.txt 2>&1
timeout -k 10 60 scripts/micro/readback 1024 >> $O/r3_readback.txt 2>&1
echo "readback done"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_agg3m
AGG_GROUPS=3000000 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_agg3m -- python3 $R/scripts/bench_operators.py agg > $O/r3_agg3m.txt 2> $O/r3_agg3m.err
echo "agg trace done"
cd $R
timeout -k 10 900 python3 bench.py > $O/r3_bench1.json 2> $O/r3_bench1.err
echo "bench done"
