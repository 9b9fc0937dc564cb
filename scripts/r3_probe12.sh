#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_next_b
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_next_b -- python3 $R/scripts/bench_operators.py next > $O/r3_next_b.txt 2> $O/r3_next_b.err
cat $O/r3_next_b.txt | head -3
cd $R
rm -f $O/r3_host_trace2.txt
PRESTO_AMD_HOST_TRACE=$O/r3_host_trace2.txt timeout -k 10 200 python3 scripts/bench_operators.py next > /dev/null 2>&1
grep -c . $O/r3_host_trace2.txt
