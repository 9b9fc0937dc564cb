#!/bin/bash
# the hash-join operator benchmarks alone: case, build ms / rows/s, probe ms / rows/s
python3 scripts/run_bench_ops.py hash_join > gpurun_out/r4_join_ops.json 2> gpurun_out/r4_join_ops.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4_join_ops.json"))
for e in d["hash_join"]:
    b, p = e["build"], e["probe"]
    print("%-70s build %.3f ms %.3g rows/s   probe %.3f ms %.3g rows/s" % (e["case"][:70], b["ms_median"], b["value"], p["ms_median"], p["value"]))
PY
