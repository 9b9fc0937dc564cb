#!/bin/bash
# kernel statistics of the hash builds (bench_operators.py join)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_build
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_build -- python3 $R/scripts/bench_operators.py join > $O/r3_build.txt 2> $O/r3_build.err
cat $O/r3_build.txt
f=$(ls $O/r3_build/*/*_kernel_stats.csv | head -1)
head -30 $f | cut -c1-170
cp $f $O/r3_build_kernel_stats.csv
rm -rf $O/r3_build
