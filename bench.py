#!/usr/bin/env python3
"""bench.py -- rows/s through the operator pipeline, TPC-H Q1 + Q6 over device-resident lineitem pages.

One step = one Q6 pass (scan-filter-project -> global SUM) plus one Q1 pass (scan-filter-project ->
hash aggregation on (returnflag, linestatus), 8 aggregates) over the rank's lineitem shard, each through a
fresh fused operator driven with the Operator protocol (addInput per page, finish, getOutput).
Weak scaling: every rank holds `--sf` worth of lineitem rows (its slice of the SF x N table); the path
shards by row range and needs no data-path collective, the final 4-group / 1-row partials
are not merged across ranks inside the timed region.

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  roofline     for the dominant kernel (the Q1 fused kernel; Q6's is reported next to it), measured with
               HIP events on the operator's stream around every launch in the timed region;
  cpu_baseline the oracle's hand-written-twin pipelines timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured streaming)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sf", type=float, default=100.0, help="TPC-H scale factor of the lineitem shard per GPU")
    ap.add_argument("--page-rows", type=int, default=1 << 28, help="rows per device-resident page")
    ap.add_argument("--cpu-rows", type=int, default=16_000_000, help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--queries", default="q1,q6")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; 'gloo' only to rehearse the multi-rank "
                    "control flow with several ranks on one GPU (RCCL refuses two ranks on one device)")
    return ap.parse_args()


def cpu_baseline(sf, rows):
    """Oracle twins of the two pipelines (HandTpchQuery6 / HandTpchQuery1 shape) on a sample of the same
    synthetic workload: one Driver thread each first, then T threads over disjoint row ranges."""
    import numpy as np
    from oracle import oracle as O
    from presto_amd import abi, tpch
    O.build()
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    cols = {}
    union = sorted(set(tpch.Q1_COLUMNS + tpch.Q6_COLUMNS))

    def gen(lo, hi, out):
        for c in union:
            out[c] = O.tpch_column(c, sf, lo, hi - lo)

    # generate the sample in parallel slices (generation is not timed)
    bounds = [rows * i // threads for i in range(threads + 1)]
    parts = [dict() for _ in range(threads)]
    ts = [threading.Thread(target=gen, args=(bounds[i], bounds[i + 1], parts[i])) for i in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]

    def q6_args(p):
        return [p[c][0] for c in tpch.Q6_COLUMNS]

    def q1_args(p):
        rf, ls = p[abi.L_RETURNFLAG], p[abi.L_LINESTATUS]
        return [rf[0], rf[1], ls[0], ls[1]] + [p[c][0] for c in tpch.Q1_COLUMNS[2:]]

    def run_all(fn, nthreads):
        res = [None] * nthreads
        # nthreads == 1: one Driver walks every slice; else slice i on thread i
        if nthreads == 1:
            t0 = time.perf_counter()
            for i in range(threads):
                fn(parts[i])
            return time.perf_counter() - t0
        ts = [threading.Thread(target=lambda i=i: res.__setitem__(i, fn(parts[i]))) for i in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        return time.perf_counter() - t0

    t6_1 = run_all(lambda p: O.q6(*q6_args(p)), 1)
    t1_1 = run_all(lambda p: O.q1(q1_args(p)), 1)
    t6_t = run_all(lambda p: O.q6(*q6_args(p)), threads)
    t1_t = run_all(lambda p: O.q1(q1_args(p)), threads)
    one = 2 * rows / (t6_1 + t1_1)
    many = 2 * rows / (t6_t + t1_t)
    return {
        "value": many, "unit": "rows/s", "cores": threads, "kind": "port",
        "sample": "%d lineitem rows (SF%g generator), Q6+Q1 hand-written-twin pipelines of the oracle; "
                  "1 thread: %.3g rows/s (q6 %.3g, q1 %.3g); %d threads over disjoint row ranges: q6 %.3g, q1 %.3g rows/s"
                  % (rows, sf, one, rows / t6_1, rows / t1_1, threads, rows / t6_t, rows / t1_t),
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    device = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend)

    from presto_amd import _lib, abi, tpch
    from presto_amd.operators import FusedAggregationOperatorFactory
    _lib.init(device)

    rows = tpch.lineitem_rows(args.sf)
    queries = args.queries.split(",")
    columns = sorted(set((tpch.Q1_COLUMNS if "q1" in queries else []) + (tpch.Q6_COLUMNS if "q6" in queries else [])))
    keep = []

    def allocator(nbytes):
        t = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        keep.append(t)
        return t

    # this rank's slice of the SF x world lineitem table, generated on device
    table = tpch.DeviceColumns(columns, args.sf * world, rows, allocator=allocator, first_row=rank * rows)
    torch.cuda.synchronize()

    def pages_of(cols):
        sub = tpch.DeviceColumns.__new__(tpch.DeviceColumns)
        sub.columns = cols
        sub.rows = table.rows
        sub._bufs = table._bufs
        return list(sub.pages(args.page_rows))

    q6_pages = pages_of(tpch.Q6_COLUMNS) if "q6" in queries else []
    q1_pages = pages_of(tpch.Q1_COLUMNS) if "q1" in queries else []
    ktime = {"q6": [0.0, 0], "q1": [0.0, 0]}
    results = {}

    # the planner's part, once per query plan: OperatorFactory objects holding the serialised descriptors
    # (LocalExecutionPlanner builds the factories; every Driver then calls createOperator)
    factories = {
        "q6": FusedAggregationOperatorFactory(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES),
        "q1": FusedAggregationOperatorFactory(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY,
                                              tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS),
    }

    def run_query(name, timed):
        op = factories[name].createOperator()  # a fresh operator per pass: operators are single-use
        pages = q6_pages if name == "q6" else q1_pages
        for p in pages:
            op.addInput(p)
        op.finish()
        out = op.getOutput()
        results[name] = out.to_rows()
        ms, n = op.kernelTime()  # also during warm-up: the first event query of a process pays a one-off cost
        if timed:
            ktime[name][0] += ms
            ktime[name][1] += n
        op.close()

    def step(timed):
        for q in queries:
            run_query(q, timed)

    for _ in range(args.warmup):
        step(False)
    # a full collection over torch's import graph takes tens of ms: keep the cyclic GC out of the timed region
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(args.steps):
        ts0 = time.perf_counter()
        step(True)
        step_times.append(time.perf_counter() - ts0)
    te = time.perf_counter()
    torch.cuda.synchronize()
    if os.environ.get("BENCH_DEBUG"):
        print("step ms:", ["%.2f" % (x * 1e3) for x in step_times], "final sync ms: %.2f" % ((time.perf_counter() - te) * 1e3), file=sys.stderr)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rows_per_step = rows * len(queries)
    value = rows_per_step * args.steps * world / elapsed

    # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/pmc_traffic.json,
    # written by scripts/summarize_profile.py); only valid for the default workload shape they were collected on
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and args.sf == 100.0 and args.page_rows == 1 << 28:
        pmc = json.load(open(pmc_path)).get("kernels", {})

    def roof(name, bytes_per_row, pages):
        ms, n = ktime[name]
        if n == 0:
            return None
        # algorithmic bytes of the timed region / summed duration of the operator's kernel launches; n counts the
        # launches of the dominant kernel `pa_fused` only (a page whose row count is not a multiple of 256 adds one
        # <256-row `pa_fused_tail` launch: its ~10 us are in `ms`, it is not a launch of the dominant kernel)
        total_bytes = sum(p.position_count for p in pages) * bytes_per_row * args.steps
        achieved = total_bytes / (ms / 1e3) / 1e9
        traffic = pmc.get({"q1": "q1_lds", "q6": "q6_global"}[name], {}).get("hbm_bytes_per_launch")
        return {"bound": "hbm", "kernel": "pa_fused (%s)" % name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 on gfx950), average per launch" if pmc else None,
                "avg_launch_ms": ms / n, "launches": n, "algorithmic_bytes_per_launch": total_bytes / n}

    if rank == 0:
        line = {
            "metric": "rows/s through operator pipeline, TPC-H Q1+Q6 SF100, 1/2/4/8 GPUs vs CPU ref",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "TPC-H SF%g %s fused scan-filter-project-aggregate over device-resident lineitem pages, "
                                   "%d rows per GPU, %d-row pages" % (args.sf, "+".join(q.upper() for q in queries), rows,
                                                                        args.page_rows),
                       "scale_factor_per_gpu": args.sf, "rows_per_gpu": rows, "page_rows": args.page_rows,
                       "queries": queries, "parallelism": "row-range shards, %d rank(s), no data-path collective" % world},
        }
        r1 = roof("q1", tpch.Q1_BYTES_PER_ROW, q1_pages) if "q1" in queries else None
        r6 = roof("q6", tpch.Q6_BYTES_PER_ROW, q6_pages) if "q6" in queries else None
        line["roofline"] = r1 or r6
        if r1 and r6:
            line["roofline_q6"] = r6
        line["results"] = {k: [[x.decode() if isinstance(x, bytes) else x for x in r] for r in v] for k, v in results.items()}
        if world == 1 and args.cpu_rows > 0:
            line["cpu_baseline"] = cpu_baseline(args.sf, args.cpu_rows)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
