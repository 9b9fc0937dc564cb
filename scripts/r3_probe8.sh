#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r3_q3_d $O/r3_q3_d_fetch $O/r3_q3_d_write
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3_q3_d -- python3 $R/scripts/bench_q3.py --steps 3 --warmup 1 > $O/r3_q3_d.json 2> $O/r3_q3_d.err
python3 $R/scripts/kernel_timeline.py $O/r3_q3_d > $O/r3_q3_d_timeline.txt 2>&1 || true
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r3_q3_d_fetch -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r3_q3_d_fetch.json 2> $O/r3_q3_d_fetch.err
python3 $R/scripts/pmc_by_kernel.py $O/r3_q3_d_fetch > $O/r3_q3_d_fetch.txt
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r3_q3_d_write -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r3_q3_d_write.json 2> $O/r3_q3_d_write.err
python3 $R/scripts/pmc_by_kernel.py $O/r3_q3_d_write > $O/r3_q3_d_write.txt
grep -h "k_gt_emit\|k_topn" $O/r3_q3_d_fetch.txt $O/r3_q3_d_write.txt
tail -c 600 $O/r3_q3_d.json
