#!/bin/bash
# the Q3 probe kernel's own duration (rocprofv3 --kernel-trace --stats of scripts/q3_native) under measurement switches:
#   CONFIGS="PIPE=3 SLOTWISE=1" scripts/r4_probe_kernel_ab.sh   (each word: PRESTO_AMD_BROW_<word>)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for run in a b; do
for cfg in ${CONFIGS:-"PIPE=3" "PIPE=0"}; do
    export PRESTO_AMD_BROW_${cfg}
    rm -rf $O/r_ab
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_ab -- $R/scripts/q3_native --sf 100 --steps 6 --warmup 1 > $O/r_ab.json 2> $O/r_ab.err
    unset PRESTO_AMD_BROW_${cfg%%=*}
    echo "$cfg $run: $(grep pa_fused_probe $(ls -t $O/r_ab/*/*_kernel_stats.csv | head -1) | cut -d, -f1-4)"
done
done
rm -rf $O/r_ab
