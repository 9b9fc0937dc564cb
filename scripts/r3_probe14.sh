#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_jd
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_jd -- python3 $R/scripts/micro/join_dups.py > $O/r3_jd.txt 2> $O/r3_jd.err || { tail $O/r3_jd.err; exit 1; }
cat $O/r3_jd.txt
