"""Timeline of the last step of a rocprofv3 --kernel-trace run: python scripts/kernel_timeline.py <trace dir> <anchor kernel> <anchors per step>
Prints start (ms from the step's first anchor), duration (us), gap to the previous kernel (us), name and grid of every kernel from
the last step's first launch of the anchor kernel on.  Used for DESIGN.md's per-pipeline breakdown of Q3."""
import csv
import glob
import os
import sys


def main():
    directory, anchor, per_step = sys.argv[1], sys.argv[2], int(sys.argv[3])
    files = sorted(glob.glob(os.path.join(directory, "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    start = idx[-per_step]
    t0 = int(rows[start]["Start_Timestamp"])
    prev_end = t0
    for r in rows[start:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.3f %8.1f gap %7.1f  %s grid=%s wg=%s" % ((s - t0) / 1e6, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:64], r["Grid_Size_X"], r["Workgroup_Size_X"]))
        prev_end = e


if __name__ == "__main__":
    main()
