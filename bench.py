#!/usr/bin/env python3
"""bench.py -- rows/s through the operator pipeline, TPC-H Q1 + Q6 over device-resident lineitem pages (the headline),
with TPC-H Q3 (two hash joins + grouped sum, exchange steps between the ranks when N > 1) timed beside it.

Headline step = one Q6 pass (scan-filter-project -> global SUM) plus one Q1 pass (scan-filter-project -> hash
aggregation on (returnflag, linestatus), 8 aggregates) over the rank's lineitem shard, each through a fresh fused
operator driven with the Operator protocol (addInput per page, finish, getOutput).  The headline is the metric's own
scaling mode -- STRONG: the ONE SF100 table split by row range over the N ranks ("SF100 at 1/2/4/8 GPUs"); the weak
figure (every rank holds `--sf` worth of rows) is timed beside it at N > 1.  The path shards by row range and needs no
data-path collective for the scans; the ranks' Step.PARTIAL states (4 groups / 1 row) are merged by a
Step.FINAL operator on rank 0 inside the step.

Processes.  One rank per GPU.  A rank holds ONE ROCm stack: the library's (/opt/rocm: HIP, RCCL, hiprtc).  A rank of the device
workload never imports torch -- `import torch` puts the wheel's own libamdhip64 / librccl / libhiprtc into the global symbol scope
and a library loaded afterwards binds to THOSE (presto_amd/control.py) -- `torch_imported` in the detail file is asserted false.
The control plane (rendezvous, barriers around the timed region, the max-over-ranks clock, shipping the RCCL unique id) is a local
socket between the ranks of the node (presto_amd/control.py, named after the launcher's MASTER_ADDR / MASTER_PORT); tables come from
pa_device_malloc, every byte of the data plane moves through pa_comm (RCCL over xGMI), and a fresh communicator is pre-flighted
(checked 1 MB all-to-all + all-reduce + all-gather under a watchdog) before the first step.  (Checker workloads of the CPU tests
bring torch.distributed themselves: for them the control plane is the gloo group.)

Output.  ONE JSON line of at most 4 KB on rank 0's stdout: the contract's keys, `roofline`, `cpu_baseline` and one-number
summaries of the side legs.  Everything else -- per-stage times, results, operator benchmark entries, samples' descriptions --
goes to the detail file (`--detail`, default bench_detail.json next to this file) and to stderr.

The `q3` object of the line is its own timed region (W warm-up + K steps between barriers, max over ranks): the three
Driver pipelines of presto_amd/q3.py over the rank's slices of customer / orders / lineitem.  With N > 1 every build
and probe side is hash-partitioned on its join key and shuffled between the ranks (RCCL over xGMI) -- BASELINE config #4.

Prints ONE JSON line on rank 0 (see the contract in the task description), including
  roofline     for the dominant kernel (the Q1 fused kernel; Q6's is reported next to it), measured with
               HIP events on the operator's stream around every launch in the timed region;
  cpu_baseline the oracle's hand-written-twin pipelines timed on the host cores on a bounded sample;
  q3           ms/step, input rows/s, per-pipeline ms, achieved HBM GB/s against the algorithmic bytes of SURVEY 8d,
               exchange bytes and GB/s over xGMI when N > 1;
  h2d          the Q6 pipeline fed with PA_MEM_HOST pages (what a JNI shim hands over): PCIe-inclusive rows/s, N = 1 only;
  sf300        BASELINE config #5's tables on the one GPU (1.80 G lineitem rows, 82.8 GB of Q1 / Q6 columns, plus the Q3
               tables): Q1+Q6 rows/s and Q3 ms/step over a few steps, N = 1 only (the SF100 tables are released first).
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
PCIE_PEAK_GBS = 63.0   # MI355X_MICROARCH.md: PCIe Gen5 x16 host link


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--sf", type=float, default=100.0, help="TPC-H scale factor of the shard per GPU")
    ap.add_argument("--page-rows", type=int, default=1 << 28, help="rows per device-resident page")
    ap.add_argument("--page-order", default="table", choices=["table", "shuffled"],
                    help="table: pages arrive in table order (consecutive pages continue each other in memory); shuffled: in a "
                         "seeded random order, so that no page continues its predecessor (page-size sweeps)")
    ap.add_argument("--cpu-rows", type=int, default=96_000_000, help="rows of the CPU-baseline sample (0 = skip): about 11 s of one-thread work")
    ap.add_argument("--queries", default="q1,q6", help="headline queries")
    ap.add_argument("--q3", type=int, default=1, help="1 = also time the Q3 pipelines (the `q3` object), 0 = skip")
    ap.add_argument("--q3-sf", type=float, default=0.0, help="scale factor per GPU of the Q3 tables (0 = --sf)")
    ap.add_argument("--q3-timeout", type=int, default=300, help="N > 1: seconds after which the line is printed without the Q3 leg (0 = wait for ever)")
    ap.add_argument("--h2d-rows", type=int, default=1 << 25, help="rows of the host-page (PCIe-inclusive) Q6 leg, N = 1 only (0 = skip)")
    ap.add_argument("--sf300", type=int, default=1, help="1 = also run BASELINE config #5's tables (SF300: 1.80 G lineitem rows, 82.8 GB of "
                    "Q1 / Q6 columns + the Q3 tables) on this one GPU for a few steps (the `sf300` object), N = 1 only; 0 = skip")
    ap.add_argument("--backend", default="nccl", help="the DATA plane between the ranks: 'nccl' (= 'rccl') the library's RCCL communicator, one "
                    "rank per GPU; 'gloo' (= 'host') the library's host transport over the control plane, to rehearse the multi-rank control "
                    "flow with several ranks on one GPU (RCCL refuses two ranks on one device)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="what `value` measures at N > 1.  strong (the metric's 'SF100 at 1/2/4/8 GPUs'): the ranks split the ONE --sf table "
                         "by row range; weak: every rank holds --sf worth of rows (its slice of the SF x N table).  The other "
                         "mode is timed beside it and reported as an object of that name (--other-scaling 0 skips it)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where rank 0 writes the full report (the stdout line is its summary)")
    ap.add_argument("--preflight-timeout", type=int, default=120, help="N > 1: seconds the communicator's creation + pre-flight may take before the rank "
                    "leaves with exit code 4 instead of waiting inside RCCL for ever")
    ap.add_argument("--other-scaling", type=int, default=1)
    ap.add_argument("--operators", type=int, default=1, help="1 = also run the operator benchmarks (the `operators` object: hash aggregation at "
                    "several cardinalities, hash join build / probe, OrderBy, TopN -- the reference's micro-benchmark shapes -- each beside its oracle twin "
                    "on the host), N = 1 only; 0 = skip")
    ap.add_argument("--workload", default="device", help="'device' = the product path; 'module:Class' = a checker workload with the same "
                    "surface (tests/rehearsal_workload.py: the oracle's operators on CPU ranks, to rehearse the multi-rank control flow)")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: this process becomes the launcher -- it starts N ranks as
    fresh child processes (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) BEFORE anything here has touched the GPU or
    imported torch, waits for them and leaves with their exit code.  The ranks inherit stdout: rank 0's JSON line is the output."""
    import subprocess
    # --standalone: torchrun picks and owns the rendezvous port itself (two benches started together cannot collide on one)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo among the CPUs this process may run on (hyper-thread siblings
    count once); falls back to the affinity count."""
    allowed = os.sched_getaffinity(0)
    cores, cpu, phys = set(), None, 0
    try:
        for line in open("/proc/cpuinfo"):
            key, _, value = line.partition(":")
            key = key.strip()
            if key == "processor":
                cpu = int(value)
            elif key == "physical id":
                phys = int(value)
            elif key == "core id" and cpu in allowed:
                cores.add((phys, int(value)))
    except (OSError, ValueError):
        pass
    return len(cores) or len(allowed)


def cpu_baseline(sf, rows, one_thread_rows=8_000_000, warmups=3, runs=10):
    """Oracle twins of the two pipelines (HandTpchQuery6 / HandTpchQuery1 shape) on a sample of the same synthetic workload, as
    BASELINE.md section 2 asks: one Driver thread, and T threads over disjoint row ranges with T = all physical cores; `warmups`
    warm-up passes + `runs` measured passes each, median and min (AbstractOperatorBenchmark.java:305-334: rows / wall seconds).
    The T-thread passes walk the whole sample, the one-thread passes its first `one_thread_rows` rows."""
    from oracle import oracle as O
    from presto_amd import abi, tpch
    O.build()
    threads = max(1, physical_cores())
    union = sorted(set(tpch.Q1_COLUMNS + tpch.Q6_COLUMNS))

    def gen(lo, hi, out):
        for c in union:
            out[c] = O.tpch_column(c, sf, lo, hi - lo)

    # the sample in `threads` slices, generated in parallel (generation is not timed); the one-thread sample in front of them
    one_rows = min(one_thread_rows, rows)
    bounds = [rows * i // threads for i in range(threads + 1)]
    parts = [dict() for _ in range(threads)]
    head = {}
    ts = [threading.Thread(target=gen, args=(bounds[i], bounds[i + 1], parts[i])) for i in range(threads)]
    ts.append(threading.Thread(target=gen, args=(0, one_rows, head)))
    [t.start() for t in ts]
    [t.join() for t in ts]

    def q6(p):
        return O.q6(*[p[c][0] for c in tpch.Q6_COLUMNS])

    def q1(p):
        rf, ls = p[abi.L_RETURNFLAG], p[abi.L_LINESTATUS]
        return O.q1([rf[0], rf[1], ls[0], ls[1]] + [p[c][0] for c in tpch.Q1_COLUMNS[2:]])

    def one(fn):
        t0 = time.perf_counter()
        fn(head)
        return time.perf_counter() - t0

    def many(fn):
        ts = [threading.Thread(target=fn, args=(parts[i],)) for i in range(threads)]
        t0 = time.perf_counter()
        [t.start() for t in ts]
        [t.join() for t in ts]
        return time.perf_counter() - t0

    def measure(run, fn):
        for _ in range(warmups):
            run(fn)
        t = sorted(run(fn) for _ in range(runs))
        return t[len(t) // 2], t[0]

    (m6_1, b6_1), (m1_1, b1_1) = measure(one, q6), measure(one, q1)
    (m6_t, b6_t), (m1_t, b1_t) = measure(many, q6), measure(many, q1)
    one_v, many_v = 2 * one_rows / (m6_1 + m1_1), 2 * rows / (m6_t + m1_t)
    return {
        "value": many_v, "unit": "rows/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
        "host_threads_available": len(os.sched_getaffinity(0)), "physical_cores": threads, "warmups": warmups, "runs": runs,
        "one_thread": {"value": one_v, "best": 2 * one_rows / (b6_1 + b1_1), "q6": one_rows / m6_1, "q1": one_rows / m1_1, "rows": one_rows, "unit": "rows/s"},
        "all_threads": {"value": many_v, "best": 2 * rows / (b6_t + b1_t), "q6": rows / m6_t, "q1": rows / m1_t, "threads": threads, "rows": rows, "unit": "rows/s"},
        "sample_short": "%d lineitem rows of the SF%g generator through the oracle's Q6+Q1 twins (C -O2, scalar): %d threads over disjoint row ranges, "
                        "median of %d passes; one_thread = one Driver over the first %d rows" % (rows, sf, threads, runs, one_rows),
        "sample": "%d lineitem rows (SF%g generator), Q6+Q1 hand-written-twin pipelines of the oracle (C, -O2, scalar); 1 thread = one reference "
                  "Driver over the first %d rows; %d threads (one per physical core) over disjoint row ranges = task_concurrency Drivers; %d warm-up + "
                  "%d measured passes each, `value` = rows / median wall time of the %d-thread passes (`best`: / min)"
                  % (rows, sf, one_rows, threads, warmups, runs, threads),
    }


def q3_cpu_baseline(sf=1.0, warmups=1, runs=3):
    """The oracle's composition of the Q3 operators (oracle.q3: FilterAndProject, HashBuilder / LookupJoin twice, HashAggregation, TopN)
    on one thread over SF`sf` tables of the same generator: input rows / median wall time."""
    from oracle import oracle as O
    from presto_amd import abi, tpch
    from presto_amd.page import Block, Page
    O.build()

    def table(columns, n):
        blocks = []
        for c in columns:
            v, o = O.tpch_column(c, sf, 0, n)
            t = abi.TPCH_COLUMN_TYPE[c]
            blocks.append(Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v))
        return Page(blocks, n)
    nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
    customer, orders, lineitem = table(tpch.CUSTOMER_COLUMNS, nc), table(tpch.ORDERS_COLUMNS, no), table(tpch.Q3_LINEITEM_COLUMNS, nl)
    ts = []
    for i in range(warmups + runs):
        t0 = time.perf_counter()
        O.q3(customer, orders, lineitem, top_n=10)
        if i >= warmups:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    rows = nc + no + nl
    # ... and on T threads (T = the physical cores): T instances of the same plan side by side over the same read-only tables -- what T
    # tasks of the reference would do with T copies of the work (no exchange between them: an upper bound for a partitioned join)
    threads = max(1, physical_cores())

    def many():
        workers = [threading.Thread(target=O.q3, args=(customer, orders, lineitem), kwargs={"top_n": 10}) for _ in range(threads)]
        t0 = time.perf_counter()
        [w.start() for w in workers]
        [w.join() for w in workers]
        return time.perf_counter() - t0
    many()
    tm = sorted(many() for _ in range(runs))
    return {"value": rows / ts[len(ts) // 2], "best": rows / ts[0], "unit": "rows/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
            "warmups": warmups, "runs": runs,
            "all_threads": {"value": threads * rows / tm[len(tm) // 2], "best": threads * rows / tm[0], "threads": threads, "unit": "rows/s",
                            "sample": "%d instances of the same composition side by side, one per physical core, over the same tables" % threads},
            "sample": "TPC-H SF%g tables of the same generator (%d + %d + %d rows), the oracle's Q3 operator composition incl. TopN on one thread "
                      "(one reference Driver per pipeline)" % (sf, nc, no, nl)}


class DeviceWorkload:
    """The product path: device-resident synthetic tables, operators through the C ABI (presto_amd.operators)."""

    def __init__(self, args, rank, world, device, scaling=None, with_q3=None, comm=None):
        from presto_amd import _lib, abi, tpch
        from presto_amd.operators import FusedAggregationOperatorFactory
        self.args, self.rank, self.world = args, rank, world
        self._lib, self.abi, self.tpch = _lib, abi, tpch
        self.scaling = scaling or args.scaling
        _lib.init(device)
        self.queries = [q for q in args.queries.split(",") if q]
        q3_sf = args.q3_sf or args.sf
        self.q3_on = bool(args.q3) if with_q3 is None else with_q3
        self.q3_sf = q3_sf
        # the communicator of the data plane (Q3's exchange steps, the PARTIAL states on their way to the FINAL operator): the
        # library's RCCL, one rank per GPU (rehearsals with ranks sharing a GPU: its host transport over gloo).  Made once per
        # process by main() and pre-flighted there; a second workload of the process (the other scaling mode) is handed the same one.
        self.comm, self.owns_comm = comm, False
        if world > 1 and comm is None:
            self.comm, self.owns_comm = make_comm(args), True
        if getattr(args, "rccl_failed", None):
            self.q3_on = False   # (its exchange steps are what RCCL is for)

        # Row-range shards.  weak: rank r holds rows [r n, (r + 1) n) of the SF x world table, n = rows of one SF table;
        # strong: the ranks split the ONE SF table, rank r holds rows [N r / W, N (r + 1) / W) (boundaries on multiples of `multiple`).
        def shard(rows_of, sf, multiple=4):
            if self.scaling == "weak" or world == 1:
                n = rows_of(sf)
                return sf * world, rank * n, n
            n = rows_of(sf)
            lo = n * rank // world // multiple * multiple
            hi = n if rank == world - 1 else n * (rank + 1) // world // multiple * multiple
            return sf, lo, hi - lo

        total_sf, first, self.rows = shard(tpch.lineitem_rows, args.sf)
        self.job_rows = tpch.lineitem_rows(args.sf) * (world if self.scaling == "weak" else 1)  # lineitem rows of the whole job
        columns = set((tpch.Q1_COLUMNS if "q1" in self.queries else []) + (tpch.Q6_COLUMNS if "q6" in self.queries else []))
        share_lineitem = self.q3_on and q3_sf == args.sf
        if share_lineitem:
            columns |= set(tpch.Q3_LINEITEM_COLUMNS)
        self.hbm_bytes = 0

        def allocator(nbytes):   # HBM through the C ABI (pa_device_malloc): the rank's one ROCm stack
            self.hbm_bytes += int(nbytes)
            return _lib.DeviceAllocation(nbytes)

        # this rank's slice of the table, generated on device
        self.table = tpch.DeviceColumns(sorted(columns), total_sf, self.rows, allocator=allocator, first_row=first)
        self.q6_pages = self._pages_of(self.table, tpch.Q6_COLUMNS, args.page_rows) if "q6" in self.queries else []
        self.q1_pages = self._pages_of(self.table, tpch.Q1_COLUMNS, args.page_rows) if "q1" in self.queries else []
        self.ktime = {"q6": [0.0, 0], "q1": [0.0, 0]}
        self.kname = {}
        self.results = {}
        # the planner's part, once per query plan: OperatorFactory objects holding the serialised descriptors
        # (LocalExecutionPlanner builds the factories; every Driver then calls createOperator).  With more than one rank the
        # aggregations run as Step.PARTIAL on every rank and Step.FINAL on rank 0 (HashAggregationOperator.java:390), the
        # ranks' intermediate pages combined in rank order -- inside the step.
        step = abi.STEP_PARTIAL if world > 1 else abi.STEP_SINGLE
        # the two pipelines of a step are two Drivers of one task on ONE stream of the task's (what desc.stream is for): their
        # kernels run one after the other -- so that a kernel's HIP-event time is its own -- while the host work of one pipeline
        # (descriptor, launch, result read-back) overlaps the other's kernels
        self.stream = _lib.DeviceStream()
        self.factories = {
            "q6": FusedAggregationOperatorFactory(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES, step=step,
                                                  stream=self.stream.handle),
            "q1": FusedAggregationOperatorFactory(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY,
                                                  tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS, step=step, stream=self.stream.handle),
        }
        self.merger = None
        if world > 1:
            from presto_amd.exchange import PartialStateMerger, partial_layout
            from presto_amd.operators import AggregationOperator, HashAggregationOperator
            self.merger = PartialStateMerger(comm=self.comm)
            t6, f6 = partial_layout([], tpch.Q6_AGGREGATES)
            t1, f1 = partial_layout([abi.VARCHAR, abi.VARCHAR], tpch.Q1_AGGREGATES)
            self.final_operators = {
                "q6": lambda: AggregationOperator(t6, f6, step=abi.STEP_FINAL),
                "q1": lambda: HashAggregationOperator(t1, [0, 1], f1, step=abi.STEP_FINAL, type_params=[1, 1] + [0] * (len(t1) - 2)),
            }
        if self.q3_on:
            self.q3_rows_job = tuple(f(q3_sf) * (world if self.scaling == "weak" else 1)
                                     for f in (tpch.customer_rows, tpch.orders_rows, tpch.lineitem_rows))
            ct, cf, nc = shard(tpch.customer_rows, q3_sf, 20)
            ot, of, no = shard(tpch.orders_rows, q3_sf)
            lt, lf, nl = shard(tpch.lineitem_rows, q3_sf)
            self.q3_rows = (nc, no, nl)
            self.customer = tpch.DeviceColumns(tpch.CUSTOMER_COLUMNS, ct, nc, allocator=allocator, first_row=cf)
            self.orders = tpch.DeviceColumns(tpch.ORDERS_COLUMNS, ot, no, allocator=allocator, first_row=of)
            self.q3_lineitem = self.table if share_lineitem else tpch.DeviceColumns(
                tpch.Q3_LINEITEM_COLUMNS, lt, nl, allocator=allocator, first_row=lf)
            pr = args.page_rows
            self.q3_pages = (self._pages_of(self.customer, tpch.CUSTOMER_COLUMNS, max(20, pr - pr % 20)),
                             self._pages_of(self.orders, tpch.ORDERS_COLUMNS, pr),
                             self._pages_of(self.q3_lineitem, tpch.Q3_LINEITEM_COLUMNS, pr))
            self.q3_stream = _lib.DeviceStream()
            self.q3_counters = {}
        _lib.device_synchronize()

    def _pages_of(self, table, cols, page_rows):
        sub = self.tpch.DeviceColumns.__new__(self.tpch.DeviceColumns)
        sub.columns = cols
        sub.rows = table.rows
        sub._bufs = table._bufs
        pages = list(sub.pages(page_rows))
        if self.args.page_order == "shuffled":
            import random
            random.Random(0x5EED).shuffle(pages)
        for p in pages:
            p.to_c()  # the C view of a page is what the JNI shim is handed: built once, outside the timed region
        return pages

    def synchronize(self):
        self._lib.device_synchronize()

    # ---- headline ----
    def step(self, timed):
        """One pass of every query: each pipeline through a fresh operator (operators are single-use) driven with the Operator protocol
        -- addInput per page, finish, getOutput.  The Drivers of the step's pipelines are interleaved the way the task executor
        interleaves them: every pipeline's pages are handed over first, then the results are collected."""
        ops = []
        for name in self.queries:
            op = self.factories[name].createOperator()
            for p in (self.q6_pages if name == "q6" else self.q1_pages):
                op.addInput(p)
            op.finish()
            ops.append((name, op))
        self.partials = {}
        for name, op in ops:
            out = op.getOutput()
            if self.world > 1:
                self.partials[name] = out   # this rank's intermediate states (a host page of a few rows)
            else:
                self.results[name] = out.to_rows()
            ms, n = op.kernelTime()  # also during warm-up: the first event query of a process pays a one-off cost
            if timed:
                self.ktime[name][0] += ms
                self.ktime[name][1] += n
                if name not in self.kname:
                    self.kname[name] = op.kernelName()
            op.close()
        if self.world > 1:
            # PARTIAL -> FINAL across the ranks, inside the step: one small all-gather, the FINAL operators on rank 0
            final = self.merger.merge(self.partials, self.final_operators)
            if final is not None:
                self.results.update({q: ([] if p is None else p.to_rows()) for q, p in final.items()})

    def rows_per_step(self):
        """lineitem rows entering the first operators of one step, on THIS rank"""
        return self.rows * len(self.queries)

    def job_rows_per_step(self):
        """... on all ranks together"""
        return self.job_rows * len(self.queries)

    def roofline(self, name, steps, pmc):
        tpch = self.tpch
        bytes_per_row = {"q1": tpch.Q1_BYTES_PER_ROW, "q6": tpch.Q6_BYTES_PER_ROW}[name]
        pages = self.q1_pages if name == "q1" else self.q6_pages
        ms, n = self.ktime[name]
        if n == 0:
            return None
        # algorithmic bytes of the timed region / summed duration of the operator's kernel launches; n counts the launches of
        # the dominant kernel only (a page whose row count is not a multiple of 256 adds one <256-row `..._tail` launch: its
        # ~10 us are in `ms`, it is not a launch of the dominant kernel)
        total_bytes = sum(p.position_count for p in pages) * bytes_per_row * steps
        achieved = total_bytes / (ms / 1e3) / 1e9
        kernel = self.kname.get(name, "")
        # HBM bytes per launch from the committed counter passes -- only when they are of THIS kernel (same generated code: the
        # name carries the code object's key) launched over the same pages (same launches per pass and rows per launch)
        t = pmc.get(kernel)
        traffic = None
        if t and abs(t.get("algorithmic_bytes_per_launch", 0) - total_bytes / n) <= 1e-6 * total_bytes / n:
            traffic = t.get("hbm_bytes_per_launch")
        return {"bound": "hbm", "kernel": kernel, "query": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --operators 0 --sf300 0 --q3 0 "
                                  "--h2d-rows 0 --cpu-rows 0`, launches of this kernel name only, FETCH_SIZE x2 on gfx950, average per launch" if traffic else None,
                "avg_launch_ms": ms / n, "launches": n, "algorithmic_bytes_per_launch": total_bytes / n,
                "rows_per_launch": sum(p.position_count for p in pages) * steps / n}

    def workload_name(self):
        a = self.args
        if self.world > 1 and self.scaling == "strong":
            return ("TPC-H SF%g %s fused scan-filter-project-aggregate over device-resident lineitem pages, the %d rows of the table split by row "
                    "range over %d GPUs, PARTIAL -> FINAL merge inside the step, %d-row pages"
                    % (a.sf, "+".join(q.upper() for q in self.queries), self.job_rows, self.world, a.page_rows))
        return ("TPC-H SF%g %s fused scan-filter-project-aggregate over device-resident lineitem pages, %d rows per GPU, %d-row pages"
                % (a.sf, "+".join(q.upper() for q in self.queries), self.rows, a.page_rows))

    # ---- Q3 ----
    def q3_step(self):
        from presto_amd import q3
        out, counters = q3.run(self.q3_pages[0], self.q3_pages[1], self.q3_pages[2], self.q3_stream.handle, comm=self.comm,
                               distributed=self.world > 1, result_mem=self.abi.MEM_HOST, top_n=10, with_count=False)
        self.q3_counters = counters
        self.results["q3"] = [r for p in out for r in p.to_rows()]

    def q3_input_rows(self):
        return sum(self.q3_rows)

    def q3_job_input_rows(self):
        return sum(self.q3_rows_job)

    def q3_algorithmic_bytes(self):
        # customer 21 B/row, orders 24 B/row, lineitem 28 B/row (columns read once by the scans)
        nc, no, nl = self.q3_rows
        return nc * 21 + no * 24 + nl * 28

    # ---- host pages ----
    def h2d(self, rows):
        """Q6 over PA_MEM_HOST pages (pageable numpy buffers, 4 Mi-row pages): the rate an unmodified Driver sees when its
        pages start on the host.  Bounded by the host link (28 B/row over PCIe), not by the kernel."""
        from presto_amd.operators import FusedAggregationOperator, download_page
        tpch = self.tpch
        rows = min(rows, self.rows)
        sub = self._pages_of(self.table, tpch.Q6_COLUMNS, rows)[0]
        host = download_page(sub)
        page_rows = 1 << 22
        pages = [host.get_region(i, min(page_rows, rows - i)) for i in range(0, rows, page_rows)]
        best = None
        for _ in range(3):
            op = FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
            t0 = time.perf_counter()
            for p in pages:
                op.addInput(p)
            op.finish()
            op.getOutput()
            dt = time.perf_counter() - t0
            op.close()
            best = dt if best is None else min(best, dt)
        gbs = rows * tpch.Q6_BYTES_PER_ROW / best / 1e9
        out = {"value": rows / best, "unit": "rows/s", "GBps": gbs, "peak_GBps": PCIE_PEAK_GBS, "frac": gbs / PCIE_PEAK_GBS,
               "workload": "Q6 fused pipeline over %d rows handed over as PA_MEM_HOST pages of %d rows (pageable host buffers)" % (rows, page_rows)}
        # the same rows as the pages a Driver delivers: 1 MB (PageProcessor.java:56-58: MAX_PAGE_SIZE_IN_BYTES; 32 768 rows of Q6's 28 B)
        try:
            small_rows = 32768
            small = [host.get_region(i, min(small_rows, rows - i)) for i in range(0, rows, small_rows)]
            sbest = None
            for _ in range(3):
                op = FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
                t0 = time.perf_counter()
                for p in small:
                    op.addInput(p)
                op.finish()
                op.getOutput()
                dt = time.perf_counter() - t0
                op.close()
                sbest = dt if sbest is None else min(sbest, dt)
            sg = rows * tpch.Q6_BYTES_PER_ROW / sbest / 1e9
            out["small_pages"] = {"value": rows / sbest, "unit": "rows/s", "GBps": sg, "frac": sg / PCIE_PEAK_GBS, "page_rows": small_rows, "pages": len(small),
                                  "workload": "the same rows as %d pageable host pages of %d rows (1 MB: what PageProcessor emits), Python Driver loop -- "
                                              "%.1f us per page all told" % (len(small), small_rows, sbest / len(small) * 1e6)}
        except Exception as e:
            out["small_pages"] = {"error": "%s: %s" % (type(e).__name__, e)}
        return out

    def close(self):
        if getattr(self, "stream", None) is not None:
            self.stream.destroy()
            self.stream = None
        if self.q3_on:
            self.q3_stream.destroy()
        if self.comm is not None and self.owns_comm:
            self.comm.destroy()
        self.comm = None
        # the tables go back to the library's pool (DeviceAllocation.free): the next leg's tables reuse the HBM
        for t in ("table", "customer", "orders", "q3_lineitem"):
            cols = getattr(self, t, None)
            if cols is not None:
                for values, offsets in cols._bufs.values():
                    for b in (values, offsets):
                        if b is not None and hasattr(b, "free"):
                            b.free()
                setattr(self, t, None)
        self.q6_pages = self.q1_pages = []
        self.q3_pages = ()


def rccl_transport(args):
    return args.backend in ("nccl", "rccl")


CONTROL = None   # the process's control plane (main() makes it)


def make_comm(args):
    """The process's pa_comm, made and pre-flighted under a watchdog: a peer that never arrives leaves ncclCommInitRank (or the first
    collective) waiting for ever -- after --preflight-timeout seconds this rank leaves with PREFLIGHT_FAILED_EXIT_CODE instead."""
    from presto_amd.exchange import Comm

    def bail():
        print("bench.py: the communicator was not up and checked within %d s -- leaving" % args.preflight_timeout, file=sys.stderr, flush=True)
        os._exit(PREFLIGHT_FAILED_EXIT_CODE)
    watchdog = None
    if args.preflight_timeout > 0:
        watchdog = threading.Timer(args.preflight_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    comm, failure = None, None
    try:
        comm = Comm.rccl(CONTROL) if rccl_transport(args) else Comm.host(CONTROL)
        comm.preflight(1 << 20)
    except Exception as e:
        failure = "%s: %s" % (type(e).__name__, e)
        print("bench.py: communicator pre-flight failed: %s" % failure, file=sys.stderr, flush=True)
        if not rccl_transport(args):
            os._exit(PREFLIGHT_FAILED_EXIT_CODE)
    finally:
        if watchdog is not None:
            watchdog.cancel()
    if rccl_transport(args):
        # The ranks decide TOGETHER, over the control plane (which does not depend on RCCL): when RCCL answered with an error on some rank
        # (and every rank came back -- one that hangs inside RCCL is the watchdog's, and then the control plane's time-out ends the others),
        # all of them go on over the library's host transport.  The headline needs no more than that -- the scans run without a data-path
        # collective and the PARTIAL states are a few KB per step -- so the job still yields its number; the Q3 leg, whose exchange moves
        # GBs between the GPUs, is left out, and the line says so (config.data_plane, q3.error).
        failures = [f for f in CONTROL.all_gather(failure) if f]
        if failures:
            args.rccl_failed = failures[0]
            print("bench.py: RCCL is not available to this job (%s) -- the ranks go on over the host transport, without the Q3 leg" % failures[0],
                  file=sys.stderr, flush=True)
            try:
                comm = Comm.host(CONTROL)
                comm.preflight(1 << 16)
            except Exception as e:
                print("bench.py: host transport pre-flight failed: %s: %s" % (type(e).__name__, e), file=sys.stderr, flush=True)
                os._exit(PREFLIGHT_FAILED_EXIT_CODE)
    return comm


def timed_region(workload, dist, world, fn, steps, warmup):
    """W untimed + K timed calls of fn between barrier + device synchronisation on both sides; MAX over the ranks.  The barrier and
    the MAX run over the control plane (`dist`); the device synchronisation is the library's (pa_device_synchronize)."""
    import gc
    for _ in range(warmup):
        fn(False)
    # a full collection over a large import graph takes tens of ms: keep the cyclic GC out of the timed region
    gc.collect()
    gc.freeze()
    gc.disable()
    workload.synchronize()
    if world > 1:
        dist.barrier()
    workload.synchronize()
    t0 = time.perf_counter()
    step_times = []
    for _ in range(steps):
        ts0 = time.perf_counter()
        fn(True)
        step_times.append(time.perf_counter() - ts0)
    workload.synchronize()
    if world > 1:
        dist.barrier()
    workload.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    gc.unfreeze()
    if os.environ.get("BENCH_DEBUG"):
        print("step ms:", ["%.2f" % (x * 1e3) for x in step_times], file=sys.stderr)
    if world > 1:
        elapsed = float(dist.all_reduce_max(elapsed))
    return elapsed


def printable(rows):
    return [[x.decode() if isinstance(x, bytes) else x for x in r] for r in rows]


def sf300_leg(args, dist, device, steps=3):
    """BASELINE config #5's tables on one GPU: Q1+Q6 over 1.80 G lineitem rows (82.8 GB of columns) and Q3 over the SF300
    customer / orders / lineitem tables, `steps` timed steps each after one warm-up step.  The SF100 tables of the headline
    are released first (the caller closed that workload: its HBM went back to the library's pool)."""
    import copy
    import gc
    gc.collect()
    a = copy.copy(args)
    a.sf, a.q3_sf, a.h2d_rows = 300.0, 0.0, 0
    t0 = time.perf_counter()
    w = DeviceWorkload(a, 0, 1, device)
    gen_s = time.perf_counter() - t0
    try:
        elapsed = timed_region(w, dist, 1, w.step, steps, 1)
        out = {"workload": w.workload_name(), "value": w.rows_per_step() * steps / elapsed, "unit": "rows/s", "steps": steps,
               "ms_per_step": elapsed / steps * 1e3, "table_generation_s": gen_s, "hbm_bytes_resident": w.hbm_bytes,
               "results": {k: printable(v) for k, v in w.results.items()}}
        r1, r6 = w.roofline("q1", steps, {}), w.roofline("q6", steps, {})
        out["roofline_frac"] = {"q1": r1 and r1["frac"], "q6": r6 and r6["frac"]}
        if w.q3_on:
            q3_elapsed = timed_region(w, dist, 1, lambda timed: w.q3_step(), steps, 1)
            out["q3"] = {"ms_per_step": q3_elapsed / steps * 1e3, "value": w.q3_input_rows() * steps / q3_elapsed, "unit": "rows/s",
                         "input_rows": w.q3_input_rows(), "result": printable(w.results["q3"]),
                         "stage_ms": {k: v for k, v in w.q3_counters.items() if k.endswith("_pipeline_ms")}}
        return out
    finally:
        w.close()


def load_workload_class(spec):
    if spec == "device":
        return DeviceWorkload
    import importlib
    mod, _, cls = spec.partition(":")
    return getattr(importlib.import_module(mod), cls)


def q3_line_object(args, workload, world, q3_elapsed, pmc):
    c = dict(workload.q3_counters)
    rows_in = workload.q3_input_rows()
    job_rows = workload.q3_job_input_rows() if hasattr(workload, "q3_job_input_rows") else rows_in * world
    ms = q3_elapsed / args.steps * 1e3
    alg = workload.q3_algorithmic_bytes()
    q3 = {"metric": "input rows/s through the TPC-H Q3 operator pipelines (customer + orders + lineitem rows entering the three scans)",
          "value": job_rows * args.steps / q3_elapsed, "unit": "rows/s", "ms_per_step": ms, "steps": args.steps,
          "scaling": getattr(workload, "scaling", "weak"),
          "scale_factor_per_gpu": workload.q3_sf if getattr(workload, "scaling", "weak") == "weak" else workload.q3_sf / world,
          "input_rows_per_gpu": rows_in,
          "stage_ms_rank0": {k: v for k, v in c.items() if k.endswith("_pipeline_ms")},
          "rank0": {k: v for k, v in c.items() if not k.endswith("_pipeline_ms")},
          "roofline": {"bound": "hbm", "scope": "whole step (all kernels of the three pipelines)", "achieved": alg / (ms / 1e3) / 1e9,
                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                       "algorithmic_bytes_per_step": alg},
          "exchange": "none (one rank)"}
    if "lineitem_fused_kernel_ms" in c and c["lineitem_fused_kernel_ms"] > 0:
        # the dominant kernel of Q3: lineitem's filter -> probe -> aggregate as one generated kernel; algorithmic bytes =
        # the 28 B/row of SURVEY 8d (orderkey 8, extendedprice 8, discount 8, shipdate 4) -- the kernel itself streams 12 B/row
        # and reads price and discount for the matching rows only, so its HBM traffic is below the algorithmic figure
        k_ms, k_n = c["lineitem_fused_kernel_ms"], c["lineitem_fused_launches"]
        k_alg = workload.q3_rows[2] * 28
        kernel = c.get("lineitem_fused_kernel", "pa_fused_probe_brow")
        t = pmc.get(kernel) or {}
        q3["roofline_dominant"] = {"bound": "hbm", "kernel": kernel, "what": "probe stage (lineitem: filter -> key rank index -> accumulate by build row)",
                                   "achieved": k_alg / (k_ms / 1e3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": k_alg / (k_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": k_alg,
                                   "kernel_ms_per_step": k_ms, "launches_per_step": k_n,
                                   "traffic": t.get("hbm_bytes_per_launch"),
                                   "note": "the kernel physically reads less than the algorithmic 28 B/row: price and discount only for the matching rows"}
    if world > 1:
        sent, t_ms = c.get("exchange_bytes_remote", 0), c.get("exchange_transfer_ms", 0.0)
        q3["exchange"] = {"what": "4 hash-partitioned exchanges per step (customer keys, orders, orders JOIN customer, lineitem), each one "
                                  "count all-gather + one grouped ncclSend/ncclRecv all-to-all; dynamic-filter bitmaps combined by all-reduce; "
                                  "all partitions of a build side arrive before any probe (PartitionedLookupSourceFactory.java:179-206)",
                          "transport": "RCCL over xGMI" if rccl_transport(args) else "host transport over gloo",
                          "rank0_bytes_to_other_ranks_per_step": sent, "rank0_all_to_all_ms_per_step": t_ms,
                          "xgmi_GBps": (sent / (t_ms / 1e3) / 1e9) if t_ms > 0 else None,
                          "note": "xgmi_GBps = rank 0's payload bytes sent to the other ranks / device time of its all-to-alls (HIP events)"}
    return q3


def sig(x, digits=6):
    """numbers of the stdout line: `digits` significant digits (the detail file keeps everything)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


LINE_LIMIT = 4096   # the driver keeps ~8 KB of stdout: the line stays far below


def summary_line(detail):
    """The ONE stdout line: the contract's keys + `roofline` + `cpu_baseline` verbatim (trimmed to their contract fields), one-number
    summaries of every side leg.  The full objects are in the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: sig(detail[k]) for k in keep}
    c = detail["config"]
    line["config"] = {k: c[k] for k in ("workload", "scale_factor_job", "rows_per_gpu", "page_rows", "queries", "parallelism") if k in c}

    def roof(r, extra=()):
        if not r:
            return None
        keys = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch") + tuple(extra)
        return {k: sig(r.get(k)) for k in keys}
    line["roofline"] = roof(detail.get("roofline"))
    if detail.get("roofline_q6"):
        line["roofline_q6"] = roof(detail["roofline_q6"])
    cb = detail.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb.get("sample_short", cb["sample"][:240]), "one_thread": sig(cb.get("one_thread", {}).get("value"))}
    for name in ("strong", "weak"):
        o = detail.get(name)
        if o:
            line[name] = {"error": o["error"][:120]} if "error" in o else {"value": sig(o["value"]), "ms_per_step": sig(o["ms_per_step"])}
    q3 = detail.get("q3")
    if q3:
        if "error" in q3:
            line["q3"] = {"error": q3["error"][:160]}
        else:
            line["q3"] = {"value": sig(q3["value"]), "ms_per_step": sig(q3["ms_per_step"]), "frac": sig(q3["roofline"]["frac"]),
                          "cpu_rows_s": sig((q3.get("cpu_baseline") or {}).get("value")),
                          "cpu_rows_s_all_cores": sig(((q3.get("cpu_baseline") or {}).get("all_threads") or {}).get("value"))}
            if isinstance(q3.get("exchange"), dict):
                line["q3"]["xgmi_GBps"] = sig(q3["exchange"].get("xgmi_GBps"))
    h = detail.get("h2d")
    if h:
        line["h2d"] = {"error": h["error"][:120]} if "error" in h else {k: sig(h[k]) for k in ("value", "GBps", "frac") if k in h}
        if "small_pages" in h and "error" not in h["small_pages"]:
            line["h2d"]["small_pages_GBps"] = sig(h["small_pages"]["GBps"])
    s3 = detail.get("sf300")
    if s3:
        line["sf300"] = {"error": s3["error"][:120]} if "error" in s3 else {"value": sig(s3["value"]), "ms_per_step": sig(s3["ms_per_step"]),
                                                                           "q3_ms_per_step": sig((s3.get("q3") or {}).get("ms_per_step"))}
    ops = detail.get("operators")
    if ops:
        if "error" in ops:
            line["operators"] = {"error": ops["error"][:160]}
        else:
            o = {}
            for e in ops.get("hash_agg", []):
                o["agg_%dr_%dg" % (e["rows"], e["groups"])] = [sig(e["value"], 4), sig(e["frac"], 3)]
            for i, e in enumerate(ops.get("hash_join", [])):
                o["join%d_build" % i] = [sig(e["build"]["value"], 4), sig(e["build"]["frac"], 3)]
                o["join%d_probe" % i] = [sig(e["probe"]["value"], 4), sig(e["probe"]["frac"], 3)]
            for k in ("order_by", "topn"):
                if k in ops:
                    o[k] = [sig(ops[k]["value"], 4), sig(ops[k]["frac"], 3)]
            line["operators"] = o
    line["detail"] = os.path.basename(detail.get("detail_path", "bench_detail.json"))
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:   # never: but a line the driver cannot read is worth nothing -- drop side legs, biggest first
        for k in ("operators", "sf300", "h2d", "roofline_q6", "strong", "weak", "q3"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) < LINE_LIMIT:
                break
    return text


Q3_FAILED_EXIT_CODE = 3          # the line was printed, but a rank failed inside the Q3 leg (or the leg timed out)
PREFLIGHT_FAILED_EXIT_CODE = 4   # the communicator did not come up (or its pre-flight found damaged bytes): nothing was measured


def main(argv=None, workload_factory=None, out=None):
    """workload_factory(args, rank, world, device) -> workload: tests rehearse the multi-rank control flow (barriers, the
    max-over-ranks clock, the exchange rounds of Q3, the JSON line) on CPU ranks with a checker workload; the default is
    the device workload (--workload names another one for subprocess runs).  Returns the process exit code."""
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args, argv)   # nothing here has touched the GPU (torch is not even imported yet)
    world = int(env_world or "1")
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (it launches its own ranks) or under "
              "torch.distributed.run with --nproc-per-node equal to --gpus" % (args.gpus, world), file=sys.stderr)
        return 2
    # ONE JSON line on stdout: whatever libraries print there while the bench runs (RCCL greets with its version) goes to stderr
    line_fd = None
    if out is None:
        sys.stdout.flush()
        line_fd = os.dup(1)
        os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    global CONTROL
    dist = None
    device = None
    make_workload = workload_factory or load_workload_class(args.workload)
    on_device = make_workload is DeviceWorkload
    torch_group = False
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_device:
            # the control plane of device ranks: a local socket, no torch in the process (see the module docstring)
            from presto_amd.control import ControlPlane
            dist = ControlPlane(rank, world, timeout=max(args.preflight_timeout, 60))
        else:
            # checker workloads run the oracle's operators on CPU ranks and exchange over torch.distributed themselves
            import torch.distributed as torch_dist
            from presto_amd.control import TorchControlPlane
            torch_dist.init_process_group("gloo")
            torch_group = True
            dist = TorchControlPlane()
        CONTROL = dist
    if on_device:
        from presto_amd._lib import lib
        n_dev = lib().pa_device_count()
        if n_dev <= 0:
            print("bench.py: no gfx950 device (pa_device_count = %d)" % n_dev, file=sys.stderr)
            return 2
        # RCCL: one rank per GPU.  Host transport (rehearsal): the ranks share what is there.  (More ranks than GPUs under RCCL: the
        # ranks share devices too -- RCCL refuses that with an error, and make_comm's fall-back to the host transport takes over.)
        device = local_rank % n_dev

    workload = make_workload(args, rank, world, device)
    scaling = getattr(workload, "scaling", "weak")
    comm = getattr(workload, "comm", None)

    def job_rows(w):
        return w.job_rows_per_step() if hasattr(w, "job_rows_per_step") else w.rows_per_step() * world

    elapsed = timed_region(workload, dist, world, workload.step, args.steps, args.warmup)
    value = job_rows(workload) * args.steps / elapsed

    # HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, written by
    # scripts/summarize_profile.py), keyed by kernel name: a kernel whose generated code changed has another name and gets null
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and world == 1:
        try:
            pmc = json.load(open(pmc_path)).get("kernels", {})
        except ValueError:
            pmc = {}

    emitted = False
    exit_code = 0

    def emit(q3, other, side_legs=True):
        nonlocal workload, emitted
        if rank != 0 or emitted:
            return
        emitted = True
        queries = workload.queries
        detail = {
            "metric": "rows/s through operator pipeline, TPC-H Q1+Q6 SF100, 1/2/4/8 GPUs vs CPU ref",
            "value": value, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload.workload_name(),
                       "scale_factor_per_gpu": args.sf if scaling == "weak" else args.sf / world, "scale_factor_job": args.sf * (world if scaling == "weak" else 1),
                       "rows_per_gpu": workload.rows, "page_rows": args.page_rows,
                       "page_order": args.page_order,
                       "queries": queries, "parallelism": "row-range shards x%d%s" % (world, "" if world == 1 else ", PARTIAL per rank -> FINAL on rank 0 inside the step"),
                       "parallelism_note": "no data-path collective in the Q1/Q6 scans; with more than one rank the Step.PARTIAL states travel in one small "
                                           "all-gather over the library's communicator to the Step.FINAL operators on rank 0, inside the step; Q3 (the `q3` "
                                           "object) shuffles its join sides between the ranks",
                       "data_plane": None if world == 1 else (("host transport over the control plane -- RCCL failed: %s" % args.rccl_failed)[:300] if getattr(args, "rccl_failed", None)
                                                               else ("RCCL (library's /opt/rocm stack)" if rccl_transport(args) else "host transport over gloo (rehearsal)")),
                       "control_plane": None if world == 1 else ("local socket between the ranks (presto_amd/control.py), no torch in the process" if on_device
                                                                   else "torch.distributed gloo (checker workload)")},
            "detail_path": args.detail,
        }
        if on_device:
            detail["torch_imported"] = "torch" in sys.modules   # (bench_ops and the legs of this file bring none)
        r1 = workload.roofline("q1", args.steps, pmc) if "q1" in queries else None
        r6 = workload.roofline("q6", args.steps, pmc) if "q6" in queries else None
        detail["roofline"] = r1 or r6
        if r1 and r6:
            detail["roofline_q6"] = r6
        if other is not None:
            detail[other["scaling"]] = other
        if q3 is None and getattr(args, "rccl_failed", None) and args.q3:
            q3 = {"error": "left out: the exchange steps of Q3 need the RCCL communicator, which failed its creation or pre-flight"}
        if q3 is not None:
            detail["q3"] = q3
        detail["results"] = {k: printable(v) for k, v in workload.results.items()}
        if side_legs and world == 1 and args.h2d_rows > 0 and hasattr(workload, "h2d"):
            try:
                detail["h2d"] = workload.h2d(args.h2d_rows)
            except Exception as e:
                detail["h2d"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if side_legs and world == 1 and args.sf300 and args.sf == 100.0 and on_device:
            try:
                workload.close()
                workload = None
                detail["sf300"] = sf300_leg(args, dist, device)
            except Exception as e:
                detail["sf300"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if side_legs and world == 1 and args.operators and on_device:
            try:
                if workload is not None:
                    workload.close()
                    workload = None
                import gc
                gc.collect()
                import bench_ops
                detail["operators"] = bench_ops.run(cpu=args.cpu_rows > 0)
            except Exception as e:
                detail["operators"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if side_legs and world == 1 and args.cpu_rows > 0:
            detail["cpu_baseline"] = cpu_baseline(args.sf, args.cpu_rows)
            if isinstance(detail.get("q3"), dict) and "error" not in detail["q3"]:
                try:
                    detail["q3"]["cpu_baseline"] = q3_cpu_baseline()
                except Exception as e:
                    detail["q3"]["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        text = summary_line(detail)
        detail["line"] = json.loads(text)
        try:
            with open(args.detail, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:
            print("bench.py: could not write %s: %s" % (args.detail, e), file=sys.stderr)
        print(json.dumps(detail), file=sys.stderr, flush=True)
        if out is not None:
            print(text, file=out, flush=True)
        else:
            sys.stdout.flush()
            os.write(line_fd, (text + "\n").encode())

    # the other scaling mode, timed beside the headline (Q1 + Q6 only; its tables are generated now and released afterwards)
    other = None
    if world > 1 and args.other_scaling:
        other_mode = "strong" if scaling == "weak" else "weak"
        try:
            import inspect
            params = inspect.signature(make_workload).parameters
            kw = {"scaling": other_mode, "with_q3": False} if "scaling" in params else None
            if kw is not None:
                if "comm" in params:
                    kw["comm"] = comm
                w2 = make_workload(args, rank, world, device, **kw)
                try:
                    e2 = timed_region(w2, dist, world, w2.step, args.steps, args.warmup)
                    other = {"scaling": other_mode, "value": job_rows(w2) * args.steps / e2, "unit": "rows/s", "ms_per_step": e2 / args.steps * 1e3,
                             "steps": args.steps, "rows_per_gpu_rank0": w2.rows, "job_rows": job_rows(w2) // max(len(w2.queries), 1),
                             "workload": w2.workload_name(), "results": {k: printable(v) for k, v in w2.results.items()}}
                    ro = w2.roofline("q1", args.steps, {}) if "q1" in w2.queries else None
                    if ro:
                        other["roofline_frac_q1_rank0"] = ro["frac"]
                finally:
                    w2.close()
                    del w2
        except Exception as e:
            other = {"scaling": other_mode, "error": "%s: %s" % (type(e).__name__, e)}

    # N > 1: the Q3 leg runs collectives; a rank that fails inside one would leave the others waiting for ever.  The headline was
    # measured above: if the leg does not come back within --q3-timeout seconds, rank 0 prints the line without it and every
    # rank leaves -- with a non-zero exit code: the line is there, the run is not clean.
    watchdog = None
    if world > 1 and getattr(workload, "q3_on", False) and args.q3_timeout > 0:
        def bail():
            try:
                emit({"error": "the Q3 leg did not finish within %d s on %d ranks" % (args.q3_timeout, world)}, other, side_legs=False)
            finally:
                os._exit(Q3_FAILED_EXIT_CODE)
        watchdog = threading.Timer(args.q3_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    q3 = None
    if getattr(workload, "q3_on", False):
        try:
            q3_elapsed = timed_region(workload, dist, world, lambda timed: workload.q3_step(), args.steps, max(1, min(args.warmup, 2)))
            q3 = q3_line_object(args, workload, world, q3_elapsed, pmc)
        except Exception as e:  # the headline must survive a failing side leg
            q3 = {"error": "%s: %s" % (type(e).__name__, e)}
            exit_code = Q3_FAILED_EXIT_CODE
    emit(q3, other)
    if workload is not None:
        workload.close()
    if world > 1:
        dist.barrier()  # (still under the watchdog: a rank whose Q3 leg failed arrives here while the others wait inside a collective)
        if watchdog is not None:
            watchdog.cancel()
        if on_device and "torch" in sys.modules:   # a second ROCm stack in this rank: the run is not what it claims to be
            print("bench.py: torch was imported in rank %d" % rank, file=sys.stderr)
            exit_code = exit_code or 5
        dist.close()
        if torch_group:
            import torch.distributed as torch_dist
            torch_dist.destroy_process_group()
    if line_fd is not None:
        sys.stdout.flush()
        os.dup2(line_fd, 1)
        os.close(line_fd)
    return exit_code


if __name__ == "__main__":
    sys.exit(main())
