#!/bin/bash
# Q3 per step: the C++ Driver loop over the C ABI (scripts/q3_native) next to the Python one (scripts/bench_q3.py), same box
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
timeout -k 10 240 scripts/q3_native --steps 10 --warmup 3 > $O/r3_native.json 2> $O/r3_native.err
timeout -k 10 240 python scripts/bench_q3.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/r3_native_py.json
timeout -k 10 240 scripts/q3_native --steps 10 --warmup 3 >> $O/r3_native.json 2>> $O/r3_native.err
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out")
for l in open(os.path.join(O, "r3_native.json")):
    d = json.loads(l)
    print("native", {k: v for k, v in d.items() if "ms" in k})
d = json.loads(open(os.path.join(O, "r3_native_py.json")).read())
print("python", d["ms_per_step"], {k: round(v, 3) for k, v in d["rank0"].items() if k.endswith("_ms") and isinstance(v, float)})
PY
