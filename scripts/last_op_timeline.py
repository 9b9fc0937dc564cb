"""Kernel timeline of the last run of an operator in a rocprofv3 --kernel-trace directory: python scripts/last_op_timeline.py <dir> <anchor>
prints start (us from the last launch of the anchor kernel), duration, gap to the previous kernel, name and grid from there on."""
import csv, glob, os, sys
d, anchor = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
f = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
start = max(idx[-1] - back, 0)
t0 = prev = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f %8.1f gap %6.1f  %s grid=%s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:64], r["Grid_Size_X"]))
    prev = e
