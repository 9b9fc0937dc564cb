"""Average of every collected counter per kernel name (and grid) of a rocprofv3 --pmc run:
python scripts/pmc_by_kernel.py <dir> [name filter].  Values are raw (gfx950: FETCH_SIZE counts 32-B-sector units as KiB -- apply
the corrections of MI355X_MICROARCH.md before quoting bytes)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    directory = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = sorted(glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(files[-1])))
    acc = defaultdict(lambda: [0.0, 0])
    for r in rows:
        name = r["Kernel_Name"][:48]
        if flt and flt not in name:
            continue
        k = (name, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""), r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
    for k in sorted(acc):
        total, n = acc[k]
        print("%-48s grid=%-10s %-14s avg %.1f over %d launches" % (k[0], k[1], k[2], total / n, n))


if __name__ == "__main__":
    main()
