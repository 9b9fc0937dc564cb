"""world_size-2 gloo test of the exchange logic (presto_amd/exchange.py) on CPU ranks.

The per-row kernels are supplied by the oracle here (checker implementation of the `ops` interface); what is under
test is the sharded path itself: partition -> count exchange -> all-to-all per column -> per-rank build/probe,
which must reproduce the single-process join as a multiset and keep every key on exactly one rank
(PartitionedLookupSource semantics, …/operator/join/PartitionedLookupSource.java:143-152)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """CPU checker implementation of the exchange `ops` interface."""

    def __init__(self):
        from oracle import oracle as O
        self.O = O

    def _page(self, columns, types):
        from presto_amd import abi
        from presto_amd.exchange import rows_of
        from presto_amd.page import Block, Page
        blocks = [Block.varwidth(c[0].numpy(), c[1].numpy()) if t == abi.VARCHAR else Block.flat(t, c.numpy()) for c, t in zip(columns, types)]
        return Page(blocks, rows_of(columns[0]))

    def hash_rows(self, columns, types, channels):
        return torch.from_numpy(self.O.hash_page(self._page(columns, types), channels))

    def partition_ids(self, raw_hash, partition_count, local):
        return torch.from_numpy(self.O.partition_ids(raw_hash.numpy(), partition_count, local))

    def partition_positions(self, partition, partition_count):
        pos, counts = self.O.partition_positions(partition.numpy(), partition_count)
        return torch.from_numpy(pos), [int(c) for c in counts]

    def gather(self, column, positions):
        return column[positions.long()]

    def gather_varwidth(self, values, offsets, positions):
        v, o, p = values.numpy(), offsets.numpy().astype(np.int64), positions.numpy().astype(np.int64)
        lengths = (o[p + 1] - o[p]).astype(np.int32)
        out_off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int32)
        out = np.concatenate([v[o[i]:o[i + 1]] for i in p]) if len(p) else np.zeros(0, np.uint8)
        return torch.from_numpy(np.ascontiguousarray(out.astype(np.uint8))), torch.from_numpy(out_off), torch.from_numpy(lengths)

    def offsets_from_lengths(self, lengths):
        out = np.concatenate([[0], np.cumsum(lengths.numpy().astype(np.int64))]).astype(np.int32)
        return torch.from_numpy(out), int(out[-1])


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


def worker(rank, world, port, result_queue):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from presto_amd.exchange import exchange_columns
    ops = OracleOps()
    build, btypes, probe, ptypes = make_tables(1234)
    # every rank starts with a contiguous row range of both tables (the scan shards by row range)
    def shard(cols):
        n = len(cols[0])
        lo, hi = n * rank // world, n * (rank + 1) // world
        return [torch.from_numpy(np.ascontiguousarray(c[lo:hi])) for c in cols]
    b_recv, b_counts = exchange_columns(ops, shard(build), btypes, [0])
    p_recv, p_counts = exchange_columns(ops, shard(probe), ptypes, [0])
    rows = oracle_join_rows(O, [c.numpy() for c in b_recv], btypes, [c.numpy() for c in p_recv], ptypes)
    keys = sorted(set(b_recv[0].tolist()))
    result_queue.put((rank, rows, keys, b_counts, p_counts))
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
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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


def test_partitioned_join_over_two_gloo_ranks(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    results.sort()
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
    # counts are consistent: what rank r received from s is what s sent to r
    assert sum(results[0][3]) + sum(results[1][3]) == len(build[0])
    assert sum(results[0][4]) + sum(results[1][4]) == len(probe[0])


class CollectOperator:
    """Sink with the Operator protocol: keeps the pages it is given."""

    def __init__(self):
        self.pages, self._finished = [], False

    def needsInput(self):
        return not self._finished

    def addInput(self, page):
        self.pages.append(page)

    def getOutput(self):
        return None

    def finish(self):
        self._finished = True

    def isFinished(self):
        return self._finished


def exchange_operator_worker(rank, world, port, result_queue):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from presto_amd import abi
    from presto_amd.operators import Driver
    from presto_amd.q3 import ExchangeOperator, page_of, tensor_of
    keys, vals = exchange_operator_table()
    # uneven page counts: rank 0 feeds three pages, rank 1 one page -- rank 1 must keep answering rank 0's rounds
    n = len(keys)
    bounds = [0, n // 4, n // 2, 3 * n // 4] if rank == 0 else [3 * n // 4, n]
    if rank == 0:
        ranges = list(zip(bounds[:-1], bounds[1:]))
    else:
        ranges = [(bounds[0], bounds[1])]
    types = [abi.BIGINT, abi.DOUBLE]
    pages = [page_of([torch.from_numpy(np.ascontiguousarray(keys[lo:hi])), torch.from_numpy(np.ascontiguousarray(vals[lo:hi]))], types)
             for lo, hi in ranges]
    ex = ExchangeOperator(types, [0], None, ops=OracleOps(), device=torch.device("cpu"))
    sink = CollectOperator()
    Driver(pages, [ex, sink]).run()
    cpu = torch.device("cpu")
    got_keys = np.concatenate([tensor_of(p.blocks[0], cpu).numpy() for p in sink.pages]) if sink.pages else np.zeros(0, np.int64)
    got_vals = np.concatenate([tensor_of(p.blocks[1], cpu).numpy() for p in sink.pages]) if sink.pages else np.zeros(0)
    result_queue.put((rank, got_keys.tolist(), got_vals.tolist(), ex.rows_sent, ex.rows_received, len(sink.pages)))
    dist.barrier()
    dist.destroy_process_group()


def exchange_operator_table():
    rng = np.random.default_rng(5)
    return rng.integers(0, 10 ** 6, 4000).astype(np.int64), rng.random(4000)


def test_exchange_operator_with_uneven_page_counts(oracle):
    """ExchangeOperator (the exchange step of the multi-GPU Q3 pipelines): collective rounds stay matched when the
    ranks feed different numbers of pages; every row arrives exactly once, on the rank its key hashes to."""
    from presto_amd.page import Block, Page
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=exchange_operator_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    keys, vals = exchange_operator_table()
    h = oracle.hash_page(Page([Block.bigint(keys)], len(keys)), [0])
    part = oracle.partition_ids(h, world, local=True)
    for rank, got_keys, got_vals, sent, received, pages in results:
        mine = part == rank
        assert sorted(zip(got_keys, got_vals)) == sorted(zip(keys[mine].tolist(), vals[mine].tolist()))
        assert received == int(mine.sum())
        assert pages == 3  # one output page per collective round (rank 0 fed three pages)
    assert results[0][3] == 3000 and results[1][3] == 1000


def varchar_worker(rank, world, port, result_queue):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from presto_amd import abi
    from presto_amd.exchange import exchange_columns
    from presto_amd.page import Block
    keys, names = varchar_table()
    n = len(keys)
    lo, hi = (0, n // 3) if rank == 0 else (n // 3, n)   # uneven shards
    b = Block.varchar(names[lo:hi])
    cols = [torch.from_numpy(np.ascontiguousarray(keys[lo:hi])), (torch.from_numpy(b.values.copy()), torch.from_numpy(b.offsets.copy()))]
    recv, counts = exchange_columns(OracleOps(), cols, [abi.BIGINT, abi.VARCHAR], [1])   # partitioned BY THE STRING
    rk = recv[0].numpy().tolist()
    rb, ro = recv[1][0].numpy().tobytes(), recv[1][1].numpy().tolist()
    rows = [(rk[i], rb[ro[i]:ro[i + 1]]) for i in range(len(rk))]
    result_queue.put((rank, rows, counts))
    dist.barrier()
    dist.destroy_process_group()


def varchar_table():
    rng = np.random.default_rng(21)
    words = [b"", b"a", b"BUILDING", b"AUTOMOBILE", b"0123456789abcdefghijklmnopqrstuvwxyz", b"\xc3\xa9", b"zz"]
    n = 3001
    return rng.integers(0, 10 ** 6, n).astype(np.int64), [words[i] + str(int(j)).encode() for i, j in zip(rng.integers(0, len(words), n), rng.integers(0, 50, n))]


def test_varchar_columns_travel_through_the_exchange(oracle):
    """VARCHAR columns (and a VARCHAR partitioning key): per-row lengths split by rows, bytes split by the byte totals of
    the destinations, offsets rebuilt by the receiver; every row arrives once, on the rank its string hashes to."""
    from presto_amd.page import Block, Page
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=varchar_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    keys, names = varchar_table()
    h = oracle.hash_page(Page([Block.varchar(names)], len(names)), [0])
    part = oracle.partition_ids(h, world, local=True)
    for rank, rows, counts in results:
        mine = [(int(k), nm) for k, nm, p in zip(keys, names, part) if p == rank]
        assert sorted(rows) == sorted(mine) and len(rows) == sum(counts)
        # source-rank order, then source position: rank 0's rows first, each run in ascending source position
        split = counts[0]
        first = [(int(k), nm) for k, nm, p in zip(keys[:len(keys) // 3], names[:len(keys) // 3], part[:len(keys) // 3]) if p == rank]
        assert rows[:split] == first
