#!/usr/bin/env python3
"""Turns rocprofv3 output directories (kernel trace + stats, FETCH_SIZE pass, WRITE_SIZE pass) of a bench.py run into
the committed summaries under profiles/:  <tag>_kernel_stats.csv, <tag>_summary.md and pmc_traffic.json
(per-launch HBM bytes of the fused kernels, corrected as MI355X_MICROARCH.md 'HBM' prescribes: on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, WRITE_SIZE is exact; both in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, trace_dir, fetch_dir, write_dir = sys.argv[1:5]
bench_json = sys.argv[5] if len(sys.argv) > 5 else None
q3_fetch_txt = sys.argv[6] if len(sys.argv) > 6 else None  # scripts/pmc_by_kernel.py output of the Q3 FETCH_SIZE pass
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def one(directory, *patterns):
    """First file under `directory` (any depth) matching one of the patterns: rocprofv3 writes CSVs directly, or a
    rocpd database that `rocpd2csv` / `rocpd2summary --format csv` turn into the same tables."""
    for pattern in patterns:
        # (gpurun merges every run's files into the same local directory: take the newest)
        hits = sorted(glob.glob(os.path.join(directory, "**", pattern), recursive=True), key=os.path.getmtime, reverse=True)
        if hits:
            return hits[0]
    raise SystemExit("no %s under %s" % (" / ".join(patterns), directory))


stats_src = one(trace_dir, "*_kernel_stats.csv", "kernels_summary.csv")
shutil.copy(stats_src, os.path.join(out, tag + "_kernel_stats.csv"))
stats = []
for r in csv.DictReader(open(stats_src)):  # column names differ between the two producers
    stats.append({"Name": r["Name"], "Calls": r["Calls"], "TotalDurationNs": r.get("TotalDurationNs") or r["Duration (Nsec)"],
                  "AverageNs": r.get("AverageNs") or r["Average (Nsec)"],
                  "Percentage": "%.2f" % float(r.get("Percentage") or r["Percent (Inc)"])})
trace = list(csv.DictReader(open(one(trace_dir, "*kernel_trace.csv"))))


def kernel_class(r):
    if r["Kernel_Name"] != "pa_fused":
        return None
    wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size")
    return "q1_lds" if int(wg) == 64 else "q6_global"


dur = collections.defaultdict(list)
for r in trace:
    k = kernel_class(r)
    if k:
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)

pmc = {}
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    rows = list(csv.DictReader(open(one(d, "*counter_collection*.csv"))))
    acc = collections.defaultdict(list)
    for r in rows:
        k = kernel_class(r)
        if k and r["Counter_Name"] == name:
            acc[k].append(float(r["Counter_Value"]))
    pmc[name] = {k: sum(v) / len(v) for k, v in acc.items()}

traffic = {}
for k in dur:
    fetch_kib = pmc["FETCH_SIZE"].get(k, 0.0)
    write_kib = pmc["WRITE_SIZE"].get(k, 0.0)
    traffic[k] = {
        "avg_launch_ms": sum(dur[k]) / len(dur[k]), "launches": len(dur[k]),
        "fetch_size_kib_raw": fetch_kib, "write_size_kib": write_kib,
        "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
        "correction": "FETCH_SIZE x2 (gfx950 wide streaming reads), WRITE_SIZE x1; KiB -> bytes",
    }
if q3_fetch_txt and os.path.exists(q3_fetch_txt):
    # the fused probe kernel of Q3's lineitem pipeline (launches of `pa_fused` in scripts/bench_q3.py): reads only -- its writes are
    # memory-side atomics into the build-row table, not counted by WRITE_SIZE
    for line in open(q3_fetch_txt):
        parts = line.split()
        if parts and parts[0] == "pa_fused" and "FETCH_SIZE" in parts:
            kib = float(parts[parts.index("avg") + 1])
            traffic["q3_probe"] = {"fetch_size_kib_raw": kib, "launches": int(parts[parts.index("over") + 1]),
                                   "hbm_bytes_per_launch": 2.0 * kib * 1024.0,
                                   "correction": "FETCH_SIZE x2 (gfx950 wide streaming reads); KiB -> bytes; command: python3 scripts/bench_q3.py (SF100)"}
json.dump({"tag": tag, "command": "python3 bench.py --steps N --warmup W --cpu-rows 0 --q3 0 --h2d-rows 0 (SF100, 2^28-row pages)", "kernels": traffic},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)

with open(os.path.join(out, tag + "_summary.md"), "w") as f:
    f.write("# %s -- rocprofv3 summary of `python3 bench.py` (SF100 Q1+Q6, 1 x MI355X)\n\n" % tag)
    f.write("Collected with `rocprofv3 --kernel-trace --stats`, then `--pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` in separate runs.\n\n")
    f.write("## kernel stats (all kernels of the run)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in stats:
        f.write("| `%s` | %s | %.3f | %.1f | %s |\n" % (r["Name"][:70], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    f.write("\n## fused kernels (per launch = one 2^28-row page (the last page of the table is shorter))\n\n| kernel | launches | avg ms | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM bytes / launch (corrected) |\n|---|---|---|---|---|---|\n")
    for k, t in traffic.items():
        f.write("| pa_fused %s | %d | %.4f | %.0f | %.1f | %.4g |\n" % (k, t["launches"], t.get("avg_launch_ms", float("nan")), t["fetch_size_kib_raw"], t.get("write_size_kib", 0.0),
                                                                        t["hbm_bytes_per_launch"]))
    if bench_json and os.path.exists(bench_json):
        f.write("\n## bench.py line of the traced run\n\n```\n%s\n```\n" % open(bench_json).read().strip())
print(json.dumps(traffic, indent=1))
