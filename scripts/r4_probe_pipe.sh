#!/bin/bash
# Q3's lineitem probe kernel under measurement switches, interleaved twice: per-pipeline ms of scripts/q3_native at SF100.
#   PRESTO_AMD_BROW_PIPE   0 plain loop / 1 next quad's columns loaded ahead / 3 four-stage software pipeline (default)
#   PRESTO_AMD_BROW_WAVES  amdgpu_waves_per_eu of the kernel
# RUN_TESTS=1: the tests of the fused probe on the default setting afterwards.
set -e
O=gpurun_out/pipe
mkdir -p $O
for run in a b; do
    for cfg in ${CONFIGS:-"PIPE=3" "PIPE=0"}; do
        env PRESTO_AMD_BROW_${cfg//,/ PRESTO_AMD_BROW_} ./scripts/q3_native --sf 100 --steps 20 --warmup 3 > $O/q3_${cfg}_$run.json 2> $O/q3_${cfg}_$run.err
        echo "$cfg run $run: $(cut -c100-330 $O/q3_${cfg}_$run.json)"
    done
done
if [ "$RUN_TESTS" = 1 ]; then
    python -m pytest tests/test_gpu_fused_join.py tests/test_gpu_q3_pipeline.py tests/test_gpu_join.py -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
    tail -2 $O/tests.log
fi
