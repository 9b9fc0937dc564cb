#!/bin/bash
# issue / wait split of the Q3 probe kernel, plain loop against the software pipeline (one SQ counter pass each; never combined with API traces)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in 3 0; do
    export PRESTO_AMD_BROW_PIPE=$L
    timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/r_q3_sq$L -- $R/scripts/q3_native --sf 100 --steps 2 --warmup 1 > $O/r4_q3_sq$L.json 2> $O/r4_q3_sq$L.err
    python3 $R/scripts/pmc_by_kernel.py $O/r_q3_sq$L pa_fused_probe > $O/r4_q3_sq_pipe$L.txt
    cat $O/r4_q3_sq_pipe$L.txt
    rm -rf $O/r_q3_sq$L
done
