#!/bin/bash
# Collects, on the GPU box, everything profiles/ is made of (run through gpurun from the repo root):
#   1. kernel trace + stats of `python3 bench.py`          -> gpurun_out/r_trace   (+ the bench line of that run)
#   2. FETCH_SIZE and WRITE_SIZE counter passes (separate) -> gpurun_out/r_fetch, gpurun_out/r_write
#   3. the default `python3 bench.py` (with q3, h2d, cpu_baseline) -> gpurun_out/r_bench_default.json
#   4. kernel trace + stats of scripts/bench_q3.py         -> gpurun_out/r_q3 (+ its line)
# scripts/summarize_profile.py <tag> r_trace r_fetch r_write r_trace_bench.json r_q3_fetch.txt then writes profiles/<tag>_*.  Counter passes never combine --pmc with API traces.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r_trace $O/r_fetch $O/r_write $O/r_q3
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_trace -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-rows 0 --q3 0 --h2d-rows 0 --sf300 0 > $O/r_trace_bench.json 2> $O/r_trace.err
echo "trace done"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 --q3 0 --h2d-rows 0 --sf300 0 > $O/r_fetch.json 2> $O/r_fetch.err
echo "fetch done"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-rows 0 --q3 0 --h2d-rows 0 --sf300 0 > $O/r_write.json 2> $O/r_write.err
echo "write done"
cd $R
timeout -k 10 540 python3 bench.py > $O/r_bench_default.json 2> $O/r_bench_default.err
echo "default bench done"
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_q3 -- python3 $R/scripts/bench_q3.py --steps 5 --warmup 1 > $O/r_q3_bench.json 2> $O/r_q3.err
echo "q3 done"
cd $R
python3 scripts/kernel_timeline.py $O/r_q3 pa_fp_count 2 > $O/r_q3_timeline.txt
cd /tmp
# HBM bytes of Q3's kernels (counter pass of its own)
rm -rf $O/r_q3_fetch
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r_q3_fetch -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r_q3_fetch.json 2> $O/r_q3_fetch.err
cd $R
python3 scripts/pmc_by_kernel.py $O/r_q3_fetch > $O/r_q3_fetch.txt
echo "q3 counters done"
