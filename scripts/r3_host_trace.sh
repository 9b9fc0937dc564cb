#!/bin/bash
# host-side listing of one Q3 step: every C-ABI call with its start and duration (the last step of the run)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
rm -f $O/r3_host_trace.txt
PRESTO_AMD_HOST_TRACE=$O/r3_host_trace.txt timeout -k 10 240 python scripts/bench_q3.py --steps 3 --warmup 2 > $O/r3_host_trace.json
python - <<'PY'
import os
p = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r3_host_trace.txt")
recs = [(float(a), float(b), c.strip()) for a, b, c in (l.split(None, 2) for l in open(p))]
recs.sort()
# the last step: from the last pa_filter_project / factory call after a long pause ... simply the last 400 records
tail = recs[-400:]
out = []
prev_end = None
for s, d, n in tail:
    gap = s - prev_end if prev_end is not None else 0.0
    out.append("%12.1f %8.1f  gap %7.1f  %s" % (s, d, gap, n))
    prev_end = max(prev_end or 0.0, s + d)
open(p.replace(".txt", "_tail.txt"), "w").write("\n".join(out) + "\n")
PY
tail -3 $O/r3_host_trace_tail.txt
