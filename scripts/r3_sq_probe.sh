#!/bin/bash
# issue / wait split of the Q3 kernels (one SQ counter pass; never combined with API traces)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/r_q3_sq -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r_q3_sq.json 2> $O/r_q3_sq.err
python3 $R/scripts/pmc_by_kernel.py $O/r_q3_sq pa_f > $O/r_q3_sq.txt
cat $O/r_q3_sq.txt
rm -rf $O/r_q3_sq
