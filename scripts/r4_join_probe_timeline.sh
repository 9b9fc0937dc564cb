#!/bin/bash
# kernel timeline of the last probe of 2^26 random keys (unique build, then five rows per key) of the operator benchmarks
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/r_jp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/r_jp -- python3 $R/scripts/run_bench_ops.py hash_join > $O/r_jp.json 2> $O/r_jp.err
python3 - <<PY
import csv, glob, os
f = sorted(glob.glob("$O/r_jp/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
cand = [i for i, r in enumerate(rows) if "k_join_probe_count_keyed4" in r["Kernel_Name"]]
gmax = max(int(rows[i]["Grid_Size_X"]) for i in cand) if cand else 0
big = [i for i in cand if int(rows[i]["Grid_Size_X"]) == gmax]
print("launches of the counting kernel:", len(cand), "largest grid", gmax, "x", len(big))
# a probe of 2^26 rows = 4 pages of 2^24: print the last life of each template instantiation
for tag in ("<false>", "<true>"):
    idx = [i for i in big if tag in rows[i]["Kernel_Name"]]
    if not idx:
        continue
    start = idx[-4] if len(idx) >= 4 else idx[0]
    t0 = int(rows[start]["Start_Timestamp"]); prev = t0
    print("--- count", tag)
    for r in rows[start:start + 60]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (s - t0) / 1e6 > 4.0: break
        print("%9.3f %8.1f gap %7.1f  %s grid=%s" % ((s - t0) / 1e6, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:70], r["Grid_Size_X"]))
        prev = e
PY
rm -rf $O/r_jp
