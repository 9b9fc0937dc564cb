#!/usr/bin/env python3
"""Turns rocprofv3 output directories of ONE command -- kernel trace + stats, a FETCH_SIZE pass and a WRITE_SIZE pass, all of
    python3 bench.py --operators 0 --sf300 0 --q3 0 --h2d-rows 0 --cpu-rows 0     (Q1 + Q6 over the SF100 table, nothing else)
-- into the committed summaries under profiles/:
    <tag>_kernel_stats.csv   rocprofv3's own per-kernel statistics of the traced run, unedited.  Every generated kernel has a name of its
                             own (pa_fused_<tier>_<key8>: tier and code-object key), so the Q1 row and the Q6 row are what they say.
    <tag>_summary.md         the same as a table, the per-kernel HBM traffic, and how to recompute bench.py's roofline from the CSV
    pmc_traffic.json         per kernel NAME: launches, average duration, FETCH_SIZE / WRITE_SIZE per launch and the HBM bytes per launch,
                             corrected as MI355X_MICROARCH.md 'HBM' prescribes (gfx950: FETCH_SIZE reports exactly half of the bytes of a
                             wide coalesced streaming read -> x2; WRITE_SIZE is exact; both in KiB), plus the algorithmic bytes per launch
                             of the traced run's own bench line -- bench.py quotes `traffic` only for a kernel of the same name over the
                             same launch population.
usage: summarize_profile.py <tag> <trace_dir> <fetch_dir> <write_dir> <detail json of the traced run> [<q3 fetch dir>]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, trace_dir, fetch_dir, write_dir = sys.argv[1:5]
detail_json = sys.argv[5] if len(sys.argv) > 5 else None
q3_fetch_dir = sys.argv[6] if len(sys.argv) > 6 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
COMMAND = "python3 bench.py --steps N --warmup W --operators 0 --sf300 0 --q3 0 --h2d-rows 0 --cpu-rows 0 (SF100, 2^28-row pages)"


def one(directory, *patterns):
    """Newest file under `directory` (any depth) matching one of the patterns (gpurun merges every run's files into the same
    local directory)."""
    for pattern in patterns:
        hits = sorted(glob.glob(os.path.join(directory, "**", pattern), recursive=True), key=os.path.getmtime, reverse=True)
        if hits:
            return hits[0]
    raise SystemExit("no %s under %s" % (" / ".join(patterns), directory))


stats_src = one(trace_dir, "*_kernel_stats.csv", "kernels_summary.csv")
shutil.copy(stats_src, os.path.join(out, tag + "_kernel_stats.csv"))
stats = []
for r in csv.DictReader(open(stats_src)):  # column names differ between the two producers
    stats.append({"Name": r["Name"], "Calls": int(r["Calls"]), "TotalDurationNs": int(float(r.get("TotalDurationNs") or r["Duration (Nsec)"])),
                  "AverageNs": float(r.get("AverageNs") or r["Average (Nsec)"]),
                  "Percentage": float(r.get("Percentage") or r["Percent (Inc)"])})


def generated(name):
    return name.startswith("pa_fused_") or name.startswith("pa_fp_") or name.startswith("pa_brow_keys")


def counters(directory, counter, per_pass=None):
    """{kernel name: (average counter value per launch, launches)}.  per_pass: {name: launches of the kernel per pass over the table} --
    the first (launches mod that) launches of the kernel are left out: the first operator of a plan cuts a 2^20-row probe launch off its
    first page (op_fused.cpp, the few-groups tier's verdict), a launch the timed passes of the bench line do not have."""
    rows = list(csv.DictReader(open(one(directory, "*counter_collection*.csv"))))
    acc = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == counter and generated(r["Kernel_Name"]):
            acc[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), float(r["Counter_Value"])))
    out = {}
    for k, v in acc.items():
        v.sort()
        extra = len(v) % per_pass[k] if per_pass and per_pass.get(k) else 0
        vals = [x for _, x in v[extra:]]
        out[k] = (sum(vals) / len(vals), len(vals))
    return out


def trace_durations(directory, per_pass):
    """{kernel name: (average duration ns, launches)} from the kernel trace, over the same launch population as `counters`"""
    rows = list(csv.DictReader(open(one(directory, "*_kernel_trace.csv"))))
    acc = collections.defaultdict(list)
    for r in rows:
        if generated(r["Kernel_Name"]):
            acc[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for k, v in acc.items():
        v.sort()
        extra = len(v) % per_pass[k] if per_pass.get(k) else 0
        vals = [x for _, x in v[extra:]]
        out[k] = (sum(vals) / len(vals), len(vals))
    return out


detail = json.load(open(detail_json)) if detail_json and os.path.exists(detail_json) else {}
roofs = {r["kernel"]: r for r in (detail.get("roofline"), detail.get("roofline_q6")) if r}
steps = (detail.get("line") or detail).get("steps") or 0
per_pass = {name: int(round(r["launches"] / steps)) for name, r in roofs.items() if steps}
fetch, write = counters(fetch_dir, "FETCH_SIZE", per_pass), counters(write_dir, "WRITE_SIZE", per_pass)
traced = trace_durations(trace_dir, per_pass)

kernels = {}
for r in stats:
    name = r["Name"]
    if not generated(name) or name.endswith("_tail_" + name.rsplit("_", 1)[-1]):
        continue
    f, fl = fetch.get(name, (0.0, 0))
    w, wl = write.get(name, (0.0, 0))
    avg_ns, calls = traced.get(name, (r["AverageNs"], r["Calls"]))
    k = {"launches_traced": calls, "launches_in_stats_file": r["Calls"], "avg_launch_ms": avg_ns / 1e6, "fetch_size_kib_raw": f, "fetch_launches": fl,
         "write_size_kib": w, "write_launches": wl, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
         "correction": "FETCH_SIZE x2 (gfx950 wide streaming reads), WRITE_SIZE x1; KiB -> bytes"}
    if name in roofs:
        # the three runs launch the kernel over the same pages: launches per step and rows per launch are the bench line's
        k["algorithmic_bytes_per_launch"] = roofs[name]["algorithmic_bytes_per_launch"]
        k["rows_per_launch"] = roofs[name].get("rows_per_launch")
        k["traffic_over_algorithmic"] = k["hbm_bytes_per_launch"] / k["algorithmic_bytes_per_launch"] if fl else None
        k["achieved_GBps_from_trace"] = k["algorithmic_bytes_per_launch"] / (k["avg_launch_ms"] / 1e3) / 1e9
        k["frac_from_trace"] = k["achieved_GBps_from_trace"] / 8000.0
        k["bench_line_avg_launch_ms"] = roofs[name]["avg_launch_ms"]
        k["bench_line_frac"] = roofs[name]["frac"]
    kernels[name] = k

if q3_fetch_dir and os.path.isdir(q3_fetch_dir):
    # Q3's kernels (scripts/bench_q3.py, SF100): reads only -- the probe kernel's table updates are memory-side atomics
    for name, (kib, n) in counters(q3_fetch_dir, "FETCH_SIZE").items():
        if name not in kernels:
            kernels[name] = {"fetch_size_kib_raw": kib, "fetch_launches": n, "hbm_bytes_per_launch": 2.0 * kib * 1024.0,
                             "correction": "FETCH_SIZE x2; KiB -> bytes; command: python3 scripts/bench_q3.py (SF100); reads only"}

json.dump({"tag": tag, "command": COMMAND, "kernels": kernels}, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)

with open(os.path.join(out, tag + "_summary.md"), "w") as f:
    f.write("# %s -- rocprofv3 summary of `%s` (1 x MI355X)\n\n" % (tag, COMMAND))
    f.write("Three runs of the same command: `rocprofv3 --kernel-trace --stats`, then `--pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE "
            "--kernel-trace` (counter passes never combined with API traces).  `%s_kernel_stats.csv` is rocprofv3's own statistics file of the "
            "first run.\n\n" % tag)
    f.write("## kernel stats (all kernels of the traced run)\n\n| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in stats:
        f.write("| `%s` | %d | %.3f | %.1f | %.2f |\n" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"] / 1e6, r["AverageNs"] / 1e3, r["Percentage"]))
    f.write("\n## generated kernels: roofline from the trace alone\n\n"
            "`achieved` = algorithmic bytes per launch / average launch duration of the kernel in the kernel trace behind `%s_kernel_stats.csv`; algorithmic "
            "bytes per launch = rows per launch x bytes per row (SURVEY 8d: Q1 46 B/row, Q6 28 B/row), rows per launch = 600 037 902 rows of SF100 lineitem / "
            "launches per pass (warm-up passes launch the same pages, so the average over the calls is the average of the timed ones).  A kernel whose call "
            "count is not a multiple of its launches per pass had probe launches in its first pass (the few-groups tier cuts 2^20 rows off the first page of "
            "the first operator of a plan): the first (calls mod launches per pass) launches are left out of the averages, durations and counters alike, and "
            "`calls` below is what remains (the statistics file's own average includes them).  `frac` = achieved / 8000 GB/s.\n\n"
            "| kernel | calls | avg ms (trace) | rows / launch | algorithmic bytes / launch | achieved GB/s | frac | bench line: avg ms | frac | FETCH_SIZE KiB raw | "
            "WRITE_SIZE KiB | HBM bytes / launch (corrected) | traffic / algorithmic |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n" % tag)
    for name, k in kernels.items():
        if "algorithmic_bytes_per_launch" not in k:
            continue
        f.write("| `%s` | %d | %.4f | %.0f | %.4g | %.0f | %.3f | %.4f | %.3f | %.0f | %.1f | %.4g | %s |\n" % (
            name, k["launches_traced"], k["avg_launch_ms"], k.get("rows_per_launch") or 0, k["algorithmic_bytes_per_launch"], k["achieved_GBps_from_trace"],
            k["frac_from_trace"], k["bench_line_avg_launch_ms"], k["bench_line_frac"], k["fetch_size_kib_raw"], k["write_size_kib"], k["hbm_bytes_per_launch"],
            "%.3f" % k["traffic_over_algorithmic"] if k.get("traffic_over_algorithmic") else "-"))
    others = [(n, k) for n, k in kernels.items() if "algorithmic_bytes_per_launch" not in k]
    if others:
        f.write("\n## other generated kernels (counter passes only)\n\n| kernel | launches | FETCH_SIZE KiB raw | HBM bytes / launch (corrected) |\n|---|---|---|---|\n")
        for name, k in others:
            f.write("| `%s` | %d | %.0f | %.4g |\n" % (name, k.get("fetch_launches", 0), k["fetch_size_kib_raw"], k["hbm_bytes_per_launch"]))
    if detail:
        f.write("\n## bench.py line of the traced run\n\n```\n%s\n```\n" % json.dumps(detail.get("line", detail)))
print(json.dumps(kernels, indent=1))
