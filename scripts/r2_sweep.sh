#!/bin/bash
# Round-2 page-size sweep (run through gpurun from the repo root): bench.py at several --page-rows (Python Driver loop) and
# scripts/page_sweep (C++ Driver loop: no Python per page) over the page layouts.  Results -> <dir>/.
O=${1:-gpurun_out/r2_sweep}
mkdir -p $O
for pr in 268435456 4194304 1048576; do
  timeout -k 10 300 python bench.py --page-rows $pr --steps 10 --warmup 1 --cpu-rows 0 --q3 0 --h2d-rows 0 > $O/bench_pr$pr.json 2> $O/bench_pr$pr.err || echo "bench pr $pr failed"
done
timeout -k 10 300 python bench.py --page-rows 4194304 --page-order shuffled --steps 10 --warmup 1 --cpu-rows 0 --q3 0 --h2d-rows 0 > $O/bench_pr4194304_shuffled.json 2> $O/bench_pr4194304_shuffled.err || echo "bench shuffled failed"
for layout in table shuffled; do
  timeout -k 10 600 scripts/page_sweep --layout $layout --steps 5 > $O/sweep_$layout.jsonl 2> $O/sweep_$layout.err || echo "sweep $layout failed"
done
for layout in host hostcopy separate; do
  timeout -k 10 600 scripts/page_sweep --layout $layout --steps 3 --rows 4194304,1048576,131072,65536,8192 > $O/sweep_$layout.jsonl 2> $O/sweep_$layout.err || echo "sweep $layout failed"
done
echo sweep done
