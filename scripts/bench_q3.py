#!/usr/bin/env python3
"""TPC-H Q3 operator pipelines (presto_amd/q3.py) over device-resident synthetic tables: rows/s = rows entering the
first operator of the three pipelines (customer + orders + lineitem) / wall time, the reference's
input_rows_per_second (testing/trino-benchmark/.../AbstractOperatorBenchmark.java:305-334).

One rank per GPU (launch with torch.distributed.run for N > 1): rank r holds rows [r n, (r+1) n) of the SF x N
tables and the stages are connected by hash-partitioned RCCL all-to-all exchanges.  Not part of bench.py's headline
metric (BASELINE.json: Q1+Q6); prints one JSON line on rank 0."""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sf", type=float, default=100.0, help="scale factor per GPU")
    ap.add_argument("--page-rows", type=int, default=1 << 28)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--force-exchange", action="store_true", help="run the exchange steps even with one rank")
    ap.add_argument("--with-count", action="store_true", help="also compute count(*) per group (the parity tests' extra column; TPC-H Q3 itself has sum(revenue) only)")
    ap.add_argument("--unfused", action="store_true", help="FilterAndProject, LookupJoin and HashAggregation as operators of their own "
                    "(what a plan without the fused descriptors runs)")
    ap.add_argument("--top-n", type=int, default=10, help="the query's ORDER BY revenue DESC, orderdate LIMIT n (0 = stop at the grouped result)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        kw = {"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)
    from presto_amd import _lib, abi, q3, tpch
    _lib.init(local)
    total_sf = args.sf * world
    nc, no, nl = tpch.customer_rows(args.sf), tpch.orders_rows(args.sf), tpch.lineitem_rows(args.sf)
    customer = tpch.DeviceColumns(tpch.CUSTOMER_COLUMNS, total_sf, nc, first_row=rank * nc)
    orders = tpch.DeviceColumns(tpch.ORDERS_COLUMNS, total_sf, no, first_row=rank * no)
    lineitem = tpch.DeviceColumns(tpch.Q3_LINEITEM_COLUMNS, total_sf, nl, first_row=rank * nl)
    stream = _lib.DeviceStream()
    distributed = world > 1 or args.force_exchange
    comm = None
    if distributed:
        from presto_amd.exchange import Comm
        comm = Comm.single() if world == 1 else (Comm.rccl() if args.backend == "nccl" else Comm.host())

    def step():
        out, counters = q3.run(customer.pages(args.page_rows - args.page_rows % 20), orders.pages(args.page_rows),
                               lineitem.pages(args.page_rows), stream.handle, comm=comm, distributed=distributed,
                               result_mem=abi.MEM_HOST if args.top_n else abi.MEM_DEVICE, top_n=args.top_n,
                               with_count=args.with_count, fused_probe=not args.unfused)
        groups = sum(p.position_count for p in out)
        return groups, counters

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        groups, counters = step()
    gc.collect()
    gc.freeze()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        groups, counters = step()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rows = nc + no + nl
    if rank == 0:
        print(json.dumps({
            "metric": "rows/s through the TPC-H Q3 operator pipelines (customer+orders+lineitem input rows)",
            "value": rows * world * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "scaling": "weak", "dtype": "f64", "data": "synthetic",
            "config": {"workload": "TPC-H SF%g Q3: 2 hash joins + grouped sum%s, %d-row pages, exchange steps %s" %
                                   (args.sf, " + top %d" % args.top_n if args.top_n else "", args.page_rows, "on" if distributed else "off (one rank)"),
                       "rows_per_gpu": {"customer": nc, "orders": no, "lineitem": nl}},
            "rank0": {"result_rows": groups, **counters}}))
    stream.destroy()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
