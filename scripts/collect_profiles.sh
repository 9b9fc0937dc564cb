#!/bin/bash
# Collects, on the GPU box, what the headline's roofline is reproduced from (run through gpurun from the repo root, in a call of its own):
#   1. kernel trace + stats of Q1 + Q6 only   -> gpurun_out/r_trace  (+ the line and the detail file of that run)
#   2. FETCH_SIZE and WRITE_SIZE counter passes of the same command (separate runs) -> gpurun_out/r_fetch, gpurun_out/r_write
#   3. scripts/bench_q3.py: kernel trace + stats -> gpurun_out/r_q3, FETCH_SIZE pass -> gpurun_out/r_q3_fetch
# then, here:  python3 scripts/summarize_profile.py <tag> gpurun_out/r_trace gpurun_out/r_fetch gpurun_out/r_write gpurun_out/r_trace_detail.json gpurun_out/r_q3_fetch
# Counter passes never combine --pmc with API traces; the traced command runs NOTHING but the headline's two kernels' plans
# (--operators 0 --sf300 0 --q3 0 --h2d-rows 0 --cpu-rows 0), so every row of the statistics file is a kernel of the headline.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
HEADLINE="--operators 0 --sf300 0 --q3 0 --h2d-rows 0 --cpu-rows 0"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r_trace $O/r_fetch $O/r_write $O/r_q3 $O/r_q3_fetch
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_trace -- python3 $R/bench.py --steps 10 --warmup 2 $HEADLINE --detail $O/r_trace_detail.json > $O/r_trace_bench.json 2> $O/r_trace.err
echo "trace done"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r_fetch -- python3 $R/bench.py --steps 3 --warmup 1 $HEADLINE --detail $O/r_fetch_detail.json > $O/r_fetch.json 2> $O/r_fetch.err
echo "fetch done"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r_write -- python3 $R/bench.py --steps 3 --warmup 1 $HEADLINE --detail $O/r_write_detail.json > $O/r_write.json 2> $O/r_write.err
echo "write done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_q3 -- python3 $R/scripts/bench_q3.py --steps 5 --warmup 1 > $O/r_q3_bench.json 2> $O/r_q3.err
echo "q3 done"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r_q3_fetch -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r_q3_fetch.json 2> $O/r_q3_fetch.err
echo "q3 counters done"
cd $R
python3 scripts/kernel_timeline.py $O/r_q3 pa_fp_count 2 > $O/r_q3_timeline.txt || true
