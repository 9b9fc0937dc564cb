"""world_size > 1 gloo tests of the sharded paths on CPU ranks.

No GPU here, so the per-row work is the oracle's and the exchange is the stand-in of tests/gloo_standin.py (same contract as
the native exchange: routing by the reference's partition rules, (source rank, source position) order, one collective per
exchange whatever the page counts).  Under test: the sharded paths themselves -- partitioned join = single-process join as
a multiset with every key on exactly one rank (PartitionedLookupSource semantics,
…/operator/join/PartitionedLookupSource.java:143-152), PARTIAL -> FINAL aggregation across ranks, and bench.py's
multi-rank control flow (world size 8)."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tables(seed):
    from presto_amd import abi
    rng = np.random.default_rng(seed)
    nb, npr = 3000, 8000
    build = [rng.integers(0, 2000, nb).astype(np.int64), rng.random(nb), np.arange(nb, dtype=np.int32)]
    probe = [rng.integers(-50, 2200, npr).astype(np.int64), np.arange(npr, dtype=np.int32)]
    return build, [abi.BIGINT, abi.DOUBLE, abi.INTEGER], probe, [abi.BIGINT, abi.INTEGER]


def oracle_join_rows(O, build, btypes, probe, ptypes):
    from presto_amd.page import Block, Page
    j = O.HashJoin(btypes, [0], [1, 2])
    j.add_build_page(Page([Block.flat(t, c) for c, t in zip(build, btypes)], len(build[0])))
    j.build()
    out, pi, bi = j.probe(Page([Block.flat(t, c) for c, t in zip(probe, ptypes)], len(probe[0])), ptypes, [0], [0, 1])
    return out.to_rows()


def init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def shuffle(O, pages, types, channels):
    """pages through one exchange; returns (received page or None, the operator)"""
    from tests.gloo_standin import StandinExchangeOperator
    ex = StandinExchangeOperator(O, types, channels)
    for p in pages:
        ex.addInput(p)
    ex.finish()
    return ex.getOutput(), ex


def worker(rank, world, port, result_queue):
    init(rank, world, port)
    from oracle import oracle as O
    from presto_amd.page import Block, Page
    build, btypes, probe, ptypes = make_tables(1234)

    # every rank starts with a contiguous row range of both tables (the scan shards by row range)
    def shard(cols, types):
        n = len(cols[0])
        lo, hi = n * rank // world, n * (rank + 1) // world
        return Page([Block.flat(t, np.ascontiguousarray(c[lo:hi])) for c, t in zip(cols, types)], hi - lo)
    b_recv, bx = shuffle(O, [shard(build, btypes)], btypes, [0])
    p_recv, px = shuffle(O, [shard(probe, ptypes)], ptypes, [0])
    rows = oracle_join_rows(O, [b.values for b in b_recv.blocks], btypes, [b.values for b in p_recv.blocks], ptypes)
    keys = sorted(set(b_recv.blocks[0].values.tolist()))
    result_queue.put((rank, rows, keys, bx.rows_received, px.rows_received))
    dist.barrier()
    dist.destroy_process_group()


class OracleFinalOperator:
    """Operator-protocol adapter over the oracle's Step.FINAL aggregation (checker side of merge_partial_aggregations)."""

    def __init__(self, O, types, group_by, aggregates):
        from presto_amd import abi
        self.agg = O.HashAggregation(types, group_by, aggregates, step=abi.STEP_FINAL)

    def addInput(self, page):
        self.agg.add_page(page)

    def finish(self):
        pass

    def getOutput(self):
        return self.agg.build_result()


def agg_worker(rank, world, port, result_queue):
    init(rank, world, port)
    from oracle import oracle as O
    from presto_amd import abi
    from presto_amd.exchange import merge_partial_aggregations, partial_layout
    from presto_amd.page import Block, Page
    keys, vals = agg_table()
    n = len(keys)
    lo, hi = n * rank // world, n * (rank + 1) // world
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    part = O.HashAggregation([abi.BIGINT, abi.DOUBLE], [0], aggs, step=abi.STEP_PARTIAL)
    part.add_page(Page([Block.bigint(keys[lo:hi]), Block.double(vals[lo:hi])], hi - lo))
    ptypes, faggs = partial_layout([abi.BIGINT], aggs)
    out = merge_partial_aggregations(part.build_result(), lambda: OracleFinalOperator(O, ptypes, [0], faggs))
    result_queue.put((rank, None if out is None else out.to_rows()))
    dist.barrier()
    dist.destroy_process_group()


def agg_table():
    rng = np.random.default_rng(77)
    return rng.integers(0, 11, 5000).astype(np.int64), rng.random(5000)


def test_partial_final_aggregation_over_two_gloo_ranks(oracle):
    from presto_amd import abi
    from presto_amd.page import Block, Page
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=agg_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert results[1] is None
    keys, vals = agg_table()
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    single = oracle.HashAggregation([abi.BIGINT, abi.DOUBLE], [0], aggs)
    single.add_page(Page([Block.bigint(keys), Block.double(vals)], len(keys)))
    expected = sorted(single.build_result().to_rows())
    got = sorted(results[0])
    assert len(got) == len(expected) == 11
    for g, e in zip(got, expected):
        assert g[0] == e[0] and g[3] == e[3]
        assert abs(g[1] - e[1]) <= 1e-12 * abs(e[1]) and abs(g[2] - e[2]) <= 1e-12 * abs(e[2])


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(target, world, timeout=180):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=timeout) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return sorted(results, key=lambda r: r[0])


def test_partitioned_join_over_two_gloo_ranks(oracle):
    world = 2
    results = run_ranks(worker, world)
    build, btypes, probe, ptypes = make_tables(1234)
    expected = oracle_join_rows(oracle, build, btypes, probe, ptypes)
    got = [r for _, rows, _, _, _ in results for r in rows]
    assert sorted(got) == sorted(expected) and len(got) > 0
    # every build key lives on exactly one rank, and that rank is the one the reference's local rule names
    k0, k1 = set(results[0][2]), set(results[1][2])
    assert not (k0 & k1)
    from presto_amd.page import Block, Page
    allkeys = np.array(sorted(k0 | k1), dtype=np.int64)
    h = oracle.hash_page(Page([Block.bigint(allkeys)], len(allkeys)), [0])
    part = oracle.partition_ids(h, world, local=True)
    assert set(allkeys[part == 0].tolist()) == k0 and set(allkeys[part == 1].tolist()) == k1
    # every row arrives exactly once
    assert results[0][3] + results[1][3] == len(build[0])
    assert results[0][4] + results[1][4] == len(probe[0])


def exchange_operator_table():
    rng = np.random.default_rng(5)
    return rng.integers(0, 10 ** 6, 4000).astype(np.int64), rng.random(4000)


def exchange_operator_worker(rank, world, port, result_queue):
    init(rank, world, port)
    from oracle import oracle as O
    from presto_amd import abi
    from presto_amd.page import Block, Page
    keys, vals = exchange_operator_table()
    # uneven page counts: rank 0 feeds three pages, rank 1 one page, rank 2 none at all
    n = len(keys)
    ranges = {0: [(0, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4)], 1: [(3 * n // 4, n)], 2: []}[rank]
    types = [abi.BIGINT, abi.DOUBLE]
    pages = [Page([Block.bigint(keys[lo:hi]), Block.double(vals[lo:hi])], hi - lo) for lo, hi in ranges]
    out, ex = shuffle(O, pages, types, [0])
    got = [] if out is None else out.to_rows()
    result_queue.put((rank, got, ex.rows_sent, ex.rows_received))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_with_uneven_page_counts(oracle):
    """One collective per exchange whatever the page counts (3 / 1 / 0 pages on the three ranks; the remote partition rule,
    three is no power of two): every row arrives exactly once, on the rank its key hashes to, ordered by (source rank,
    source position)."""
    from presto_amd.page import Block, Page
    world = 3
    results = run_ranks(exchange_operator_worker, world)
    keys, vals = exchange_operator_table()
    h = oracle.hash_page(Page([Block.bigint(keys)], len(keys)), [0])
    part = oracle.partition_ids(h, world, local=False)
    for rank, got, sent, received in results:
        mine = part == rank
        assert got == list(zip(keys[mine].tolist(), vals[mine].tolist()))   # global row order = (source rank, source position) here
        assert received == int(mine.sum())
    assert [r[2] for r in results] == [3000, 1000, 0]


def varchar_table():
    rng = np.random.default_rng(21)
    words = [b"", b"a", b"BUILDING", b"AUTOMOBILE", b"0123456789abcdefghijklmnopqrstuvwxyz", b"\xc3\xa9", b"zz"]
    n = 3001
    names = [words[i] + str(int(j)).encode() for i, j in zip(rng.integers(0, len(words), n), rng.integers(0, 50, n))]
    names = [None if k % 17 == 0 else s for k, s in enumerate(names)]   # NULLs travel too
    return rng.integers(0, 10 ** 6, n).astype(np.int64), names


def varchar_worker(rank, world, port, result_queue):
    init(rank, world, port)
    from oracle import oracle as O
    from presto_amd import abi
    from presto_amd.page import Block, Page
    keys, names = varchar_table()
    n = len(keys)
    lo, hi = (0, n // 3) if rank == 0 else (n // 3, n)   # uneven shards
    page = Page([Block.bigint(keys[lo:hi]), Block.varchar(names[lo:hi])], hi - lo)
    out, ex = shuffle(O, [page], [abi.BIGINT, abi.VARCHAR], [1])   # partitioned BY THE STRING
    result_queue.put((rank, out.to_rows(), ex.rows_received))
    dist.barrier()
    dist.destroy_process_group()


def test_varchar_columns_and_nulls_travel_through_the_exchange(oracle):
    from presto_amd.page import Block, Page
    world = 2
    results = run_ranks(varchar_worker, world)
    keys, names = varchar_table()
    h = oracle.hash_page(Page([Block.varchar(names)], len(names)), [0])
    part = oracle.partition_ids(h, world, local=True)
    for rank, rows, received in results:
        mine = [(int(k), nm) for k, nm, p in zip(keys, names, part) if p == rank]
        assert rows == mine and received == len(mine)
    assert any(nm is None for _, rows, _ in results for _, nm in rows)


# ---- bench.py on 8 CPU ranks ------------------------------------------------------------------------------------------------
def bench_worker(rank, world, port, result_queue):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world)})
    import io
    import bench
    from tests.rehearsal_workload import RehearsalWorkload
    seen = {}

    def factory(args, r, w, device):
        seen["w"] = RehearsalWorkload(args, r, w, device)
        return seen["w"]

    out = io.StringIO()
    detail = os.path.join(os.environ.get("BENCH_TEST_DIR", "/tmp"), "bench_detail_8.json")
    bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--sf", "0.002", "--cpu-rows", "0", "--backend", "gloo", "--scaling", "weak",
                "--detail", detail], workload_factory=factory, out=out)
    result_queue.put((rank, out.getvalue(), seen["w"].results.get("q3")))


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline")


def check_line_shape(text):
    """What the driver needs of the stdout line: ONE line, far below the ~8 KB of stdout it keeps, parseable from its tail, with
    the contract's keys."""
    import bench
    assert text.count("\n") == 1 and text.endswith("\n")
    assert len(text) < bench.LINE_LIMIT, len(text)
    line = json.loads(text[-8000:])   # the driver's view: the last 8000 bytes of stdout
    for k in CONTRACT_KEYS:
        assert k in line, k
    assert isinstance(line["config"]["workload"], str) and "model" not in line["config"]
    return line


def test_bench_control_flow_on_eight_gloo_ranks(oracle, tmp_path, monkeypatch):
    """bench.py --gpus 8 --backend gloo with the checker workload: one JSON line from rank 0 only, whole-job value over the
    max-over-ranks clock, a `q3` object whose exchange steps ran between 8 ranks -- and the union of the ranks' Q3 top-10
    rows holds the single-process top 10."""
    from presto_amd import abi, tpch
    from tests.rehearsal_workload import host_table
    from tests.test_gpu_q3_pipeline import oracle_q3
    world, sf = 8, 0.002
    monkeypatch.setenv("BENCH_TEST_DIR", str(tmp_path))
    results = run_ranks(bench_worker, world, timeout=300)
    lines = [r[1] for r in results]
    assert all(l == "" for l in lines[1:])
    line = check_line_shape(lines[0])
    detail = json.load(open(tmp_path / "bench_detail_8.json"))
    assert detail["line"] == line
    rows = tpch.lineitem_rows(sf)
    assert line["n_gpus"] == world and line["scaling"] == "weak" and line["steps"] == 2 and line["unit"] == "rows/s"
    assert abs(detail["value"] - 2 * rows * 2 * world / (detail["ms_per_step"] * 2 / 1e3)) <= 1e-6 * detail["value"]
    assert abs(line["value"] - detail["value"]) <= 1e-5 * detail["value"]
    assert "error" not in line["q3"] and line["q3"]["value"] > 0 and detail["q3"]["input_rows_per_gpu"] == sum(
        (tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)))
    # Q3 parity of the sharded run: global top 10 = top 10 of the union of the ranks' (disjoint) groups
    total = sf * world
    grouped, _, _ = oracle_q3(oracle, host_table(tpch.CUSTOMER_COLUMNS, total, 0, tpch.customer_rows(sf) * world),
                              host_table(tpch.ORDERS_COLUMNS, total, 0, tpch.orders_rows(sf) * world),
                              host_table(tpch.Q3_LINEITEM_COLUMNS, total, 0, tpch.lineitem_rows(sf) * world))
    expected = sorted(grouped, key=lambda r: (-r[3], r[1]))[:10]
    union = [tuple(r) for _, _, q3 in results for r in (q3 or [])]
    keys = [r[0] for r in union]
    assert len(keys) == len(set(keys))   # an orderkey lives on one rank
    got = sorted(union, key=lambda r: (-r[3], r[1]))[:10]
    assert len(expected) == 10 and [g[:3] for g in got] == [e[:3] for e in expected]
    assert all(abs(g[3] - e[3]) <= 1e-9 * abs(e[3]) for g, e in zip(got, expected))


def bench_worker_one_rank_fails(rank, world, port, path):
    sys.path.insert(0, ROOT)
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world)})
    import bench
    from tests.rehearsal_workload import RehearsalWorkload

    class FailsOnRankOne(RehearsalWorkload):
        def q3_step(self):
            if self.rank == 1:
                raise RuntimeError("rank 1 gives up before its first exchange")
            return super().q3_step()

    with open("%s.%d" % (path, rank), "w") as out:
        sys.exit(bench.main(["--gpus", str(world), "--steps", "1", "--warmup", "1", "--sf", "0.002", "--cpu-rows", "0", "--backend", "gloo",
                             "--q3-timeout", "8", "--other-scaling", "0", "--detail", path + ".detail"], workload_factory=FailsOnRankOne, out=out))


def test_bench_line_survives_a_rank_failing_inside_q3(oracle, tmp_path):
    """One rank raising inside the Q3 leg leaves the others inside a collective: after --q3-timeout the headline line is
    printed by rank 0 without the Q3 numbers and every rank leaves -- with a non-zero exit code: the line is there, the run was
    not clean."""
    world = 2
    ctx = mp.get_context("spawn")
    port = free_port()
    path = str(tmp_path / "line")
    procs = [ctx.Process(target=bench_worker_one_rank_fails, args=(r, world, port, path)) for r in range(world)]
    [p.start() for p in procs]
    [p.join(timeout=120) for p in procs]
    import bench
    assert all(p.exitcode == bench.Q3_FAILED_EXIT_CODE for p in procs)
    text = open(path + ".0").read()
    assert open(path + ".1").read() == ""
    line = check_line_shape(text)
    assert line["n_gpus"] == world and line["value"] > 0 and "error" in line["q3"]


# ---- bench.py as the driver starts it: a plain subprocess, no torchrun, no environment -----------------------------------
def run_bench(argv, env_extra=None, timeout=600):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout, cwd=ROOT)


def oracle_q1_q6(oracle, sf, rows):
    from presto_amd import tpch
    q6_cols = [oracle.tpch_column(c, sf, 0, rows)[0] for c in tpch.Q6_COLUMNS]
    cols = [oracle.tpch_column(c, sf, 0, rows) for c in tpch.Q1_COLUMNS]
    q1 = oracle.q1([cols[0][0], cols[0][1], cols[1][0], cols[1][1]] + [c[0] for c in cols[2:]])
    return oracle.q6(*q6_cols)[0], sorted(q1)


def check_q1_q6(results, q6_sum, q1_rows):
    assert abs(results["q6"][0][0] - q6_sum) <= 1e-9 * abs(q6_sum)
    got = sorted((r[0].encode(), r[1].encode(), *r[2:]) for r in results["q1"])
    assert len(got) == len(q1_rows)
    for g, e in zip(got, q1_rows):
        assert g[:2] == e[:2] and g[-1] == e[-1]
        assert all(abs(a - b) <= 1e-9 * abs(b) for a, b in zip(g[2:-1], e[2:-1]))


def test_bench_launches_its_own_ranks(oracle, tmp_path):
    """`python bench.py --gpus 8` with no torchrun environment starts 8 ranks itself (before anything touches a GPU), prints ONE
    line with n_gpus 8, the STRONG headline (the metric's mode: the ONE SF table split by row range) and the `weak` object, both with
    the PARTIAL -> FINAL merge inside the step: the merged results are those of one process over the whole tables."""
    from presto_amd import tpch
    world, sf = 8, 0.002
    path = str(tmp_path / "detail.json")
    r = run_bench(["--gpus", str(world), "--backend", "gloo", "--workload", "tests.rehearsal_workload:RehearsalWorkload", "--sf", str(sf),
                   "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--detail", path])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = check_line_shape(r.stdout.decode())
    detail = json.load(open(path))
    assert line["n_gpus"] == world and line["scaling"] == "strong" and detail["weak"]["scaling"] == "weak" and line["weak"]["value"] > 0
    rows = tpch.lineitem_rows(sf)
    assert abs(detail["value"] - 2 * rows * 2 / (detail["ms_per_step"] * 2 / 1e3)) <= 1e-6 * detail["value"]
    assert detail["weak"]["job_rows"] == rows * world
    assert abs(detail["weak"]["value"] - 2 * rows * world * 2 / (detail["weak"]["ms_per_step"] * 2 / 1e3)) <= 1e-6 * detail["weak"]["value"]
    assert "error" not in detail["q3"] and detail["q3"]["exchange"]["transport"].startswith("host transport")
    check_q1_q6(detail["results"], *oracle_q1_q6(oracle, sf, rows))                         # strong: the SF table
    check_q1_q6(detail["weak"]["results"], *oracle_q1_q6(oracle, sf * world, rows * world))   # weak: the SF x 8 table


def test_bench_weak_headline_on_two_ranks(oracle, tmp_path):
    from presto_amd import tpch
    sf = 0.002
    path = str(tmp_path / "detail.json")
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--workload", "tests.rehearsal_workload:RehearsalWorkload", "--sf", str(sf), "--steps", "1",
                   "--warmup", "1", "--cpu-rows", "0", "--scaling", "weak", "--q3", "0", "--detail", path])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = check_line_shape(r.stdout.decode())
    detail = json.load(open(path))
    assert line["scaling"] == "weak" and detail["strong"]["scaling"] == "strong" and line["n_gpus"] == 2
    check_q1_q6(detail["results"], *oracle_q1_q6(oracle, sf * 2, tpch.lineitem_rows(sf) * 2))
    check_q1_q6(detail["strong"]["results"], *oracle_q1_q6(oracle, sf, tpch.lineitem_rows(sf)))


def test_summary_line_of_a_full_detail_object_stays_small():
    """The line bench.py prints is a summary of the detail object: with every side leg present (the shape of a default N = 1 run,
    operator benchmarks included) it stays below 4 KB and its last 8000 bytes parse."""
    import bench
    roof = {"bound": "hbm", "kernel": "pa_fused_lds_0123abcd", "query": "q1", "achieved": 6063.123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.757890123,
            "traffic": 7.05e9, "traffic_source": "x" * 300, "avg_launch_ms": 1.138123456, "launches": 80, "algorithmic_bytes_per_launch": 6.9e9,
            "rows_per_launch": 1.5e8}
    cpu = {"value": 2.4e8, "unit": "rows/s", "cores": 128, "kind": "port", "sample": "s" * 900, "sample_short": "t" * 200, "one_thread": {"value": 1.87e7}}
    entry = {"rows": 67108864, "groups": 3000000, "value": 4.1e10, "frac": 0.0831234, "cpu": cpu, "reference_shape": "y" * 80}
    join = {"case": "c" * 90, "build": dict(entry), "probe": dict(entry)}
    detail = {"metric": "rows/s through operator pipeline, TPC-H Q1+Q6 SF100, 1/2/4/8 GPUs vs CPU ref", "value": 1.5712345678e11, "unit": "rows/s", "n_gpus": 1,
              "steps": 20, "warmup": 5, "ms_per_step": 7.64123456, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
              "data": "synthetic", "config": {"workload": "w" * 200, "scale_factor_job": 100.0, "rows_per_gpu": 600037902, "page_rows": 1 << 28,
                                              "queries": ["q1", "q6"], "parallelism": "row-range shards x1", "parallelism_note": "n" * 500},
              "roofline": roof, "roofline_q6": roof, "cpu_baseline": cpu, "results": {"q1": [["A", "F"] + [1.0] * 8] * 4},
              "q3": {"value": 1.6e11, "ms_per_step": 4.66, "roofline": {"frac": 0.556}, "cpu_baseline": {"value": 1.1e7}, "exchange": "none (one rank)",
                     "stage_ms_rank0": {"a": 1.0}},
              "h2d": {"value": 2e9, "GBps": 54.8, "frac": 0.87, "workload": "h" * 200, "small_pages": {"GBps": 31.0}},
              "sf300": {"value": 1.5e11, "ms_per_step": 23.0, "q3": {"ms_per_step": 14.0}, "results": {"q1": [[1.0] * 10] * 4}},
              "operators": {"hash_agg": [dict(entry, groups=g) for g in (3000000, 4, 1000, 100000, 3000001)], "hash_join": [join] * 8,
                            "order_by": entry, "topn": entry},
              "detail_path": "/somewhere/bench_detail.json"}
    text = bench.summary_line(detail) + "\n"
    line = check_line_shape(text)
    assert line["roofline"]["kernel"] == "pa_fused_lds_0123abcd" and line["roofline"]["frac"] == 0.75789 and line["cpu_baseline"]["cores"] == 128
    assert len(line["operators"]) == 5 + 16 + 2 and line["q3"]["ms_per_step"] == 4.66 and line["detail"] == "bench_detail.json"
    # a detail object larger than anything the bench produces still yields a parseable line: side legs are dropped, never the contract
    detail["operators"]["hash_join"] = [join] * 400
    line = check_line_shape(bench.summary_line(detail) + "\n")
    assert "operators" not in line and line["roofline"]["frac"] == 0.75789


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = run_bench(["--gpus", "4", "--backend", "gloo", "--workload", "tests.rehearsal_workload:RehearsalWorkload", "--sf", "0.002"],
                  env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}, timeout=120)
    assert r.returncode == 2 and r.stdout == b"" and b"WORLD_SIZE" in r.stderr
