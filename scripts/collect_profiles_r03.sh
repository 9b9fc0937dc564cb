#!/bin/bash
# Round-3 additions to scripts/collect_profiles.sh (run through gpurun from the repo root, in a call of its own):
#   1. Q3: TCC_EA0_ATOMIC (memory-side atomic requests) and WRITE_SIZE passes of scripts/bench_q3.py
#   2. grouped aggregation (scripts/bench_operators.py agg, 64 M rows): kernel trace + stats, FETCH_SIZE and WRITE_SIZE passes
#   3. joins (scripts/bench_operators.py join) and TopN / OrderBy / DynamicFilterSource (next): kernel trace + stats
#   4. the page-size sweep (scripts/page_sweep, C++ Driver loop) over the table / shuffled / separate layouts
# scripts/summarize_r03.py then writes profiles/r03_*.  Counter passes never combine --pmc with API traces.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r_q3_atomic $O/r_q3_write $O/r_agg $O/r_agg_fetch $O/r_agg_write $O/r_join $O/r_next
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $O/r_q3_atomic -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r_q3_atomic.json 2> $O/r_q3_atomic.err || \
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_ATOMIC --kernel-trace --output-format csv -d $O/r_q3_atomic -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r_q3_atomic.json 2> $O/r_q3_atomic.err
python3 $R/scripts/pmc_by_kernel.py $O/r_q3_atomic > $O/r_q3_atomic.txt
echo "q3 atomics done"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r_q3_write -- python3 $R/scripts/bench_q3.py --steps 2 --warmup 1 > $O/r_q3_write.json 2> $O/r_q3_write.err
python3 $R/scripts/pmc_by_kernel.py $O/r_q3_write > $O/r_q3_write.txt
echo "q3 writes done"
export AGG_GROUPS=4,1000,100000,1000000,3000000
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_agg -- python3 $R/scripts/bench_operators.py agg > $O/r_agg.txt 2> $O/r_agg.err
export AGG_GROUPS=3000000
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r_agg_fetch -- python3 $R/scripts/bench_operators.py agg > $O/r_agg_fetch.out 2> $O/r_agg_fetch.err
python3 $R/scripts/pmc_by_kernel.py $O/r_agg_fetch > $O/r_agg_fetch.txt
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r_agg_write -- python3 $R/scripts/bench_operators.py agg > $O/r_agg_write.out 2> $O/r_agg_write.err
python3 $R/scripts/pmc_by_kernel.py $O/r_agg_write > $O/r_agg_write.txt
echo "agg done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_join -- python3 $R/scripts/bench_operators.py join > $O/r_join.txt 2> $O/r_join.err
echo "join done"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r_next -- python3 $R/scripts/bench_operators.py next > $O/r_next.txt 2> $O/r_next.err
echo "next done"
cd $R
rm -f $O/r_sweep_*.jsonl
for layout in table shuffled separate; do
  timeout -k 10 150 scripts/page_sweep --sf 100 --steps 3 --layout $layout --rows 4194304,1048576,65536,8192 > $O/r_sweep_$layout.jsonl 2> $O/r_sweep_$layout.err || echo "sweep $layout failed"
done
timeout -k 10 150 scripts/page_sweep --sf 100 --steps 3 --layout separate --shared-stream --rows 4194304,1048576,65536,8192 > $O/r_sweep_separate_shared.jsonl 2> $O/r_sweep_separate_shared.err || echo "sweep separate (caller's stream) failed"
echo "sweep done"
