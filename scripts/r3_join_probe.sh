#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 700 python -m pytest -x -q -m gpu tests/test_gpu_join.py tests/test_gpu_join_fuzz.py tests/test_gpu_fused_join.py tests/test_gpu_q3_pipeline.py --durations=5 > $O/r3_join_tests.log 2>&1 || { tail -60 $O/r3_join_tests.log; exit 1; }
tail -8 $O/r3_join_tests.log
python3 - <<'PY' > $O/r3_join_c.txt 2>&1
import bench_ops
out = bench_ops.run(cpu=False)
for e in out["hash_join"]:
    print(e["case"], "| build %.3g rows/s (%.2f ms)" % (e["build"]["value"], e["build"]["ms_median"]), "| probe %.3g rows/s (%.2f ms)" % (e["probe"]["value"], e["probe"]["ms_median"]))
for e in out["hash_agg"]:
    print("agg", e["rows"], e["groups"], "%.3g rows/s (%.2f ms)" % (e["value"], e["ms_median"]))
print("topn %.3g" % out["topn"]["value"], "order_by %.3g" % out["order_by"]["value"])
PY
cat $O/r3_join_c.txt
timeout -k 10 240 python scripts/bench_q3.py --steps 8 --warmup 2 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('q3 ms_per_step', d['ms_per_step'], {k: round(v, 3) for k, v in d['rank0'].items() if k.endswith('_ms') and isinstance(v, float)})"
