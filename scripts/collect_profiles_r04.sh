#!/bin/bash
# Round-4 additions to scripts/collect_profiles.sh (run through gpurun from the repo root, in a call of its own):
#   1. operator benchmarks (scripts/run_bench_ops.py: bench_ops.run without the CPU twins), one traced run per section -> timelines
#   2. the page-size sweep (scripts/page_sweep, C++ Driver loop): table / shuffled / separate / retained (+ separate on the caller's stream)
#   3. the default bench (`python3 bench.py`, as the driver runs it) -> line + detail
# scripts/summarize_r04.py then writes profiles/r04_operators.md, r04_page_sweep.{json,md}, r04_bench_default.json, r04_bench_detail.json.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for section in hash_agg hash_join order_by; do
  rm -rf $O/r4_ops_$section
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r4_ops_$section -- python3 $R/scripts/run_bench_ops.py $section > $O/r4_ops_$section.json 2> $O/r4_ops_$section.err
  echo "ops $section done"
done
cd $R
rm -f $O/r4_sweep_*.jsonl
for layout in table shuffled separate retained; do
  timeout -k 10 150 scripts/page_sweep --sf 100 --steps 3 --layout $layout --rows 4194304,1048576,65536,8192 > $O/r4_sweep_$layout.jsonl 2> $O/r4_sweep_$layout.err || echo "sweep $layout failed"
done
timeout -k 10 150 scripts/page_sweep --sf 100 --steps 3 --layout separate --shared-stream --rows 4194304,1048576,65536,8192 > $O/r4_sweep_separate_shared.jsonl 2> $O/r4_sweep_separate_shared.err || echo "sweep separate (caller's stream) failed"
echo "sweep done"
timeout -k 10 540 python3 bench.py --detail $O/r4_bench_default_detail.json > $O/r4_bench_default.json 2> $O/r4_bench_default.err
echo "default bench done"
