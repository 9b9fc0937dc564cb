#!/usr/bin/env python3
"""profiles/r04_* from what scripts/collect_profiles_r04.sh left under gpurun_out/ (run here, after the gpurun call): timelines of the
operator benchmarks, the page-size sweep, the default bench line and its detail file."""
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


def trace_rows(directory):
    src = newest(directory, "*_kernel_trace.csv")
    if not src:
        return []
    rows = list(csv.DictReader(open(src)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def timeline(rows, first, last):
    """start (ms), duration (us), gap to the previous kernel (us), name, grid"""
    out = []
    t0 = int(rows[first]["Start_Timestamp"])
    prev = t0
    for r in rows[first:last]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.append("%9.3f %8.1f gap %7.1f  %s grid=%s" % ((s - t0) / 1e6, (e - s) / 1e3, (s - prev) / 1e3, r["Kernel_Name"][:76], r["Grid_Size_X"]))
        prev = e
    return "\n".join(out)


def one_life(rows, anchor, occurrence, before, after):
    """the kernels around the occurrence-th launch of `anchor` (a kernel every life of the operator launches once)"""
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(anchor)]
    if len(idx) <= occurrence:
        return "(no %s in the trace)" % anchor
    a = max(idx[occurrence] - before, 0)
    return timeline(rows, a, min(idx[occurrence] + after, len(rows)))


def entries(name):
    path = os.path.join(O, name)
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return {}


with open(os.path.join(P, "r04_operators.md"), "w") as f:
    f.write("# Operator benchmarks under rocprofv3, round 4 (scripts/run_bench_ops.py = bench_ops.run without the CPU twins, 1 x MI355X)\n\n"
            "One traced run per section (`rocprofv3 --kernel-trace --stats`); the timelines show ONE whole operator life each: start (ms), duration (us), "
            "idle gap before the kernel (us).  The gaps are host round trips (a count or key range read back, operator creation).\n\n")
    agg = entries("r4_ops_hash_agg.json").get("hash_agg", [])
    f.write("## grouped aggregation (BIGINT key, DOUBLE value; sum + count(*))\n\n| rows | groups | rows/s | ms | frac |\n|---|---|---|---|---|\n")
    for e in agg:
        f.write("| %d | %d | %.3g | %.3f | %.3f |\n" % (e["rows"], e["groups"], e["value"], e["ms_median"], e["frac"]))
    rows = trace_rows("r4_ops_hash_agg")
    if rows:
        f.write("\n### one life at 10 M rows / 3 M groups (BenchmarkGroupByHash.java:68-71)\n\n```\n%s\n```\n" % one_life(rows, "pa_fused_hash", 3, 9, 13))
        src = newest("r4_ops_hash_agg", "*_kernel_stats.csv")
        if src:
            shutil.copy(src, os.path.join(P, "r04_agg_kernel_stats.csv"))
    join = entries("r4_ops_hash_join.json").get("hash_join", [])
    f.write("\n## hash join (BIGINT key + BIGINT payload)\n\n| case | build rows/s | build ms | frac | probe rows/s | probe ms | frac |\n|---|---|---|---|---|---|---|\n")
    for e in join:
        f.write("| %s | %.3g | %.3f | %.3f | %.3g | %.3f | %.3f |\n" % (e["case"], e["build"]["value"], e["build"]["ms_median"], e["build"]["frac"], e["probe"]["value"],
                                                                    e["probe"]["ms_median"], e["probe"]["frac"]))
    rows = trace_rows("r4_ops_hash_join")
    if rows:
        f.write("\n### one build of 8 M unique keys (key rank index)\n\n```\n%s\n```\n" % one_life(rows, "pa::k_join_rank_tile_zip", 2, 12, 4))
        f.write("\n### one build of 8 M rows, five per key (partitioned build)\n\n```\n%s\n```\n" % one_life(rows, "pa::k_join_part_ids", 3, 6, 16))
        f.write("\n### probe pages of 1.4 M rows against the unique build (one round trip per page since round 4)\n\n```\n%s\n```\n"
                % one_life(rows, "pa::k_join_probe_count_keyed4", 8, 2, 16))
        src = newest("r4_ops_hash_join", "*_kernel_stats.csv")
        if src:
            shutil.copy(src, os.path.join(P, "r04_join_kernel_stats.csv"))
    ob = entries("r4_ops_order_by.json").get("order_by")
    if ob:
        f.write("\n## OrderBy, 2^24 (DOUBLE, BIGINT) rows by the BIGINT key: %.3g rows/s, %.3f ms, frac %.3f\n" % (ob["value"], ob["ms_median"], ob["frac"]))
    rows = trace_rows("r4_ops_order_by")
    if rows:
        f.write("\n### one life (rocPRIM's histogram + one-sweep passes are the `trampoline_kernel` rows)\n\n```\n%s\n```\n" % one_life(rows, "pa::(anonymous namespace)::k_iota_i32", 3, 2, 28))

sweep = []
for layout in ("table", "shuffled", "separate", "separate_shared", "retained"):
    path = os.path.join(O, "r4_sweep_%s.jsonl" % layout)
    if os.path.exists(path):
        for line in open(path):
            line = line.strip()
            if line.startswith("{"):
                sweep.append(json.loads(line))
if sweep:
    json.dump(sweep, open(os.path.join(P, "r04_page_sweep.json"), "w"), indent=1)
    with open(os.path.join(P, "r04_page_sweep.md"), "w") as f:
        f.write("# Page-size sweep, round 4 (1 x MI355X, SF100 lineitem, Q1 + Q6 fused operators; scripts/page_sweep: C++ Driver loop)\n\n"
                "Layouts: `table` = consecutive row ranges of resident columns (PA_PAGE_STABLE) in table order; `shuffled` = the same pages in a seeded random "
                "order; `separate` = every page's blocks in buffers of their own, plain pages (read before add_input returns: a copy launch and, on the "
                "operator's own stream, a drained stream per page); `retained` = the same buffers handed over as PA_PAGE_RETAINED with a release callback "
                "(round 4: the owner keeps the page until the operator releases it -- every page is released exactly once, checked by the sweep).\n"
                "Round 3 (`profiles/r03_page_sweep.md`): separate 65 536-row pages Q6 83.5 / Q1 59.1 GB/s, 8 192-row pages 11.6 / 10.4 GB/s; shuffled "
                "4 Mi-row pages 3216 / 2078 GB/s against 5026 / 4494 at 1 Mi rows (each 4 Mi-row range had a launch and its merges of its own: "
                "ranges now wait for each other in a range table up to 2^24 rows).\n\n| layout | page rows | rows/s (Q1+Q6) | Q6 GB/s | Q1 GB/s |\n|---|---|---|---|---|\n")
        for r in sweep:
            f.write("| %s | %d | %.3g | %.1f | %.1f |\n" % (r["layout"], r["page_rows"], r["rows_per_s"], r["q6_GBps"], r["q1_GBps"]))

for src, dst in (("r4_bench_default.json", "r04_bench_default.json"), ("r4_bench_default_detail.json", "r04_bench_detail.json")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
for src, dst in (("r_q3_timeline.txt", "r04_q3_timeline.txt"),):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
src = newest("r_q3", "*_kernel_stats.csv")
if src:
    shutil.copy(src, os.path.join(P, "r04_q3_kernel_stats.csv"))
print("profiles/r04_* written")
