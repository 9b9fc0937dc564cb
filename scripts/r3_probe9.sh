#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
rm -f $O/r3_host_trace.txt
PRESTO_AMD_HOST_TRACE=$O/r3_host_trace.txt timeout -k 10 240 python3 scripts/bench_q3.py --steps 4 --warmup 2 > $O/r3_q3_e.json 2> $O/r3_q3_e.err
tail -c 700 $O/r3_q3_e.json
wc -l $O/r3_host_trace.txt
