#!/bin/bash
# the grouped-aggregation operator benchmarks alone: rows, groups, ms, rows/s
python3 scripts/run_bench_ops.py hash_agg > gpurun_out/r4_agg_ops.json 2> gpurun_out/r4_agg_ops.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_agg_ops.json"))
for e in d["hash_agg"]:
    print("%10d rows %8d groups  %.3f ms  %.3g rows/s" % (e["rows"], e["groups"], e["ms_median"], e["value"]))
PY
