#!/usr/bin/env python3
"""profiles/r03_* from what scripts/collect_profiles_r03.sh left under gpurun_out/ (run here, after the gpurun call):
Q3 counter passes, the grouped-aggregation / join / TopN kernel stats and counters, the page-size sweep."""
import csv
import glob
import json
import os
import shutil

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(root, "gpurun_out")
P = os.path.join(root, "profiles")


def newest(directory, pattern):
    hits = sorted(glob.glob(os.path.join(O, directory, "**", pattern), recursive=True), key=os.path.getmtime, reverse=True)
    return hits[0] if hits else None


def stats_table(directory, top=16):
    src = newest(directory, "*_kernel_stats.csv")
    if not src:
        return "(no kernel stats)\n"
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for i, r in enumerate(csv.DictReader(open(src))):
        if i >= top:
            break
        lines.append("| `%s` | %s | %.3f | %.1f | %.2f |" % (r["Name"][:72], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return "\n".join(lines) + "\n"


def text(name, nonzero=False):
    path = os.path.join(O, name)
    if not os.path.exists(path):
        return "(missing)"
    lines = open(path).read().strip().splitlines()
    if nonzero:  # counter listings: the kernels that count at all
        lines = [l for l in lines if " avg " in l and float(l.split(" avg ")[1].split()[0]) >= 1.0]
    return "\n".join(lines)


for d, tag in (("r_agg", "r03_agg_kernel_stats.csv"), ("r_join", "r03_join_kernel_stats.csv"), ("r_next", "r03_next_kernel_stats.csv")):
    src = newest(d, "*_kernel_stats.csv")
    if src:
        shutil.copy(src, os.path.join(P, tag))

with open(os.path.join(P, "r03_q3_counters.md"), "w") as f:
    f.write("# Q3 (scripts/bench_q3.py, SF100, 1 x MI355X): counter passes of round 3\n\n")
    f.write("Each pass is a run of its own: `rocprofv3 --pmc <counter> --kernel-trace -- python3 scripts/bench_q3.py --steps 2 --warmup 1`; values are raw per-launch averages "
            "(scripts/pmc_by_kernel.py).  FETCH_SIZE / WRITE_SIZE are KiB (FETCH_SIZE x 2 for wide streaming reads on gfx950, MI355X_MICROARCH.md).\n\n")
    f.write("## memory-side atomic requests (TCC_EA0_ATOMIC)\n\nRound 2 measured ~3.7 M requests per 2^28-row lineitem page for the fused probe kernel (`pa_fused`, DESIGN.md; that pass was left "
            "under gpurun_out/ and not committed).  Round 3 (per-wave window of build positions in LDS, flushed 64 positions per instruction):\n\n```\n%s\n```\n\n" % text("r_q3_atomic.txt", True))
    f.write("## WRITE_SIZE\n\n```\n%s\n```\n\n## FETCH_SIZE\n\n```\n%s\n```\n" % (text("r_q3_write.txt", True), text("r_q3_fetch.txt", True)))

with open(os.path.join(P, "r03_operators.md"), "w") as f:
    f.write("# Operator benchmarks under rocprofv3, round 3 (scripts/bench_operators.py, 1 x MI355X)\n\n")
    f.write("## grouped aggregation, 64 M rows of (BIGINT key, DOUBLE value) = 16 B/row (BenchmarkGroupByHash.java:68-71 shape)\n\n```\n%s\n```\n\n" % text("r_agg.txt"))
    f.write(stats_table("r_agg"))
    f.write("\n### HBM traffic at 3 M groups (counter passes of their own; KiB per launch, raw)\n\nFETCH_SIZE:\n```\n%s\n```\nWRITE_SIZE:\n```\n%s\n```\n\n" % (text("r_agg_fetch.txt", True), text("r_agg_write.txt", True)))
    f.write("## joins (BenchmarkHashBuildAndJoinOperators.java:103-110, 192-199 shapes at device sizes)\n\n```\n%s\n```\n\n" % text("r_join.txt"))
    f.write(stats_table("r_join"))
    f.write("\n## TopN / OrderBy / DynamicFilterSource / PagesSerde\n\n```\n%s\n```\n\n" % text("r_next.txt"))
    f.write(stats_table("r_next"))

rows = []
for layout in ("table", "shuffled", "separate", "separate_shared"):
    path = os.path.join(O, "r_sweep_%s.jsonl" % layout)
    if os.path.exists(path):
        for line in open(path):
            line = line.strip()
            if line.startswith("{"):
                rows.append(json.loads(line))
json.dump(rows, open(os.path.join(P, "r03_page_sweep.json"), "w"), indent=1)
with open(os.path.join(P, "r03_page_sweep.md"), "w") as f:
    f.write("# Page-size sweep, round 3 (1 x MI355X, SF100 lineitem, Q1 + Q6 fused operators; scripts/page_sweep: C++ Driver loop)\n\n")
    f.write("Round 2 (`profiles/r02_page_sweep.md`): shuffled 65 536-row pages Q6 1346 GB/s, Q1 79.7 GB/s (VARCHAR channels were not gathered); separate 65 536-row pages Q1 56.7 GB/s.\n"
            "Round 3: stable device pages that do not continue each other are taken in place as a table of row ranges (no copy); device pages with buffers of their own are gathered "
            "into the arena with a byte cursor in HBM for their VARCHAR channels.\n\n| layout | page rows | rows/s (Q1+Q6) | Q6 GB/s | Q1 GB/s |\n|---|---|---|---|---|\n")
    for r in rows:
        f.write("| %s | %d | %.3g | %.1f | %.1f |\n" % (r["layout"], r["page_rows"], r["rows_per_s"], r["q6_GBps"], r["q1_GBps"]))
print("profiles/r03_* written")
