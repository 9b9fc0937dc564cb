"""Many Drivers at once (the reference's TaskExecutor runs 2 x cores Driver threads, SURVEY 8b "Threading"): different operator
handles driven from different host threads concurrently -- every thread its own pipeline, its own stream -- must give
the single-threaded results.  Exercises the process-wide pools, the JIT / generated-kernel caches and the per-thread
error slots of the library."""
import threading

import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd.expr import field
from presto_amd.operators import (FilterAndProjectOperator, FusedAggregationOperator, HashAggregationOperator, HashBuilderOperator,
                                  LookupJoinOperator, LookupSourceFactory, to_pages)
from presto_amd.page import Block, Page

pytestmark = pytest.mark.gpu


def host_lineitem(oracle, columns, sf, n):
    blocks = []
    for c in columns:
        v, o = oracle.tpch_column(c, sf, 0, n)
        t = abi.TPCH_COLUMN_TYPE[c]
        blocks.append(Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v))
    return Page(blocks, n)


def test_concurrent_drivers(gpu, oracle):
    n = 200000
    q6_page = host_lineitem(oracle, tpch.Q6_COLUMNS, 1.0, n)
    q1_page = host_lineitem(oracle, tpch.Q1_COLUMNS, 1.0, n)
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 5000, n).astype(np.int64)
    vals = rng.random(n)
    agg_page = Page([Block.bigint(keys), Block.double(vals)], n)
    build = Page([Block.bigint(np.arange(3000) * 2), Block.integer(np.arange(3000, dtype=np.int32))], 3000)
    probe = Page([Block.bigint(rng.integers(0, 7000, n)), Block.integer(np.arange(n, dtype=np.int32))], n)

    def q6():
        op = FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
        return [p.to_rows() for p in to_pages(op, [q6_page, q6_page])]

    def q1():
        op = FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                      type_params=tpch.Q1_TYPE_PARAMS)
        return sorted(r for p in to_pages(op, [q1_page]) for r in p.to_rows())

    def agg():
        op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MAX, 1, abi.DOUBLE)])
        return sorted(r for p in to_pages(op, [agg_page, agg_page]) for r in p.to_rows())

    def join():
        bridge = LookupSourceFactory()
        to_pages(HashBuilderOperator(bridge, [abi.BIGINT, abi.INTEGER], [0], [1]), [build])
        j = LookupJoinOperator(bridge, [abi.BIGINT, abi.INTEGER], [0], [1])
        return [r for p in to_pages(j, [probe]) for r in p.to_rows()]

    def fp():
        op = FilterAndProjectOperator([abi.BIGINT, abi.DOUBLE], field(0, abi.BIGINT) < 100, [field(0, abi.BIGINT), field(1, abi.DOUBLE) * 2.0])
        return [r for p in to_pages(op, [agg_page]) for r in p.to_rows()]

    jobs = [q6, q1, agg, join, fp]
    expected = [j() for j in jobs]          # single-threaded first (also warms the code caches for half of the shapes)
    results, errors = {}, []

    def worker(i):
        try:
            for rep in range(3):
                k = (i + rep) % len(jobs)
                results[(i, rep)] = (k, jobs[k]())
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(10)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    assert len(results) == 30
    for (i, rep), (k, got) in results.items():
        if k in (0, 1):  # DOUBLE sums: identical launch shapes give identical bits
            assert got == expected[k], (i, rep, k)
        else:
            assert got == expected[k], (i, rep, k)
