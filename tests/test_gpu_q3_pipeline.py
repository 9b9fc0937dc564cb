"""Q3-shaped operator pipeline (BASELINE config #4 on one GPU): three Driver pipelines chained through device-resident
pages on one stream -- customer -> filter -> HashBuilder; orders -> filter -> LookupJoin -> HashBuilder;
lineitem -> filter/project -> LookupJoin -> HashAggregation(orderkey, orderdate, shippriority; sum(revenue)) --
against the same composition of the oracle's operators.  Join multiset, BIGINT/DATE keys and group counts are
bit-exact; revenue sums within 1e-9 (stated tolerance)."""
import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd._lib import DeviceStream
from presto_amd.expr import field
from presto_amd.page import Block, Page
from tests.util import rows_equal_ignore_order

pytestmark = pytest.mark.gpu


def host_table(oracle, columns, sf, n):
    blocks = []
    for c in columns:
        v, o = oracle.tpch_column(c, sf, 0, n)
        t = abi.TPCH_COLUMN_TYPE[c]
        blocks.append(Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v))
    return Page(blocks, n)


def host_table_at(oracle, columns, sf, first, n):
    """rows [first, first + n) of a table of the generator"""
    blocks = []
    for c in columns:
        v, o = oracle.tpch_column(c, sf, first, n)
        t = abi.TPCH_COLUMN_TYPE[c]
        blocks.append(Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v))
    return Page(blocks, n)


def oracle_q3(oracle, customer, orders, lineitem):
    return oracle.q3(customer, orders, lineitem)


def _device_tables(sf):
    nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
    return (tpch.DeviceColumns(tpch.CUSTOMER_COLUMNS, sf, nc), tpch.DeviceColumns(tpch.ORDERS_COLUMNS, sf, no),
            tpch.DeviceColumns(tpch.Q3_LINEITEM_COLUMNS, sf, nl))


def _expected(oracle, sf):
    nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
    return oracle_q3(oracle, host_table(oracle, tpch.CUSTOMER_COLUMNS, sf, nc), host_table(oracle, tpch.ORDERS_COLUMNS, sf, no),
                     host_table(oracle, tpch.Q3_LINEITEM_COLUMNS, sf, nl))


@pytest.mark.parametrize("fused_probe", [True, False])
@pytest.mark.parametrize("sf,page_rows", [(0.02, 1 << 14), (0.1, 1 << 17)])
def test_q3_pipeline_matches_oracle(gpu, oracle, sf, page_rows, fused_probe):
    """fused_probe: the lineitem pipeline as one generated kernel (filter, bitmap test, probe, accumulate by build row) / as the
    three operators FilterAndProject -> LookupJoin -> HashAggregation."""
    from presto_amd import q3
    expected, exp_orders, exp_joined = _expected(oracle, sf)
    stream = DeviceStream()
    customer, orders, lineitem = _device_tables(sf)
    out, counters = q3.run(customer.pages(page_rows - page_rows % 20), orders.pages(page_rows), lineitem.pages(page_rows), stream.handle,
                           distributed=False, fused_probe=fused_probe)
    rows = [r for p in out for r in p.to_rows()]
    assert len(expected) > 100
    rows_equal_ignore_order(rows, expected, rel=1e-9)
    assert counters["build2_rows"] == exp_orders  # orders JOIN customer rows that reached the second build side
    assert sum(r[4] for r in rows) == exp_joined
    stream.destroy()


def test_q3_with_exchange_steps_on_one_rank_over_rccl(gpu, oracle):
    """The multi-GPU form of the pipelines (an exchange before every build and probe, shared dynamic-filter bitmaps) with a
    one-rank RCCL communicator: every all-to-all is a self copy, so the result must equal the oracle's, and every row must
    come back."""
    from presto_amd import q3
    from presto_amd.exchange import Comm
    sf, page_rows = 0.05, 1 << 16
    expected, exp_orders, exp_joined = _expected(oracle, sf)
    comm = Comm.single()
    try:
        stream = DeviceStream()
        customer, orders, lineitem = _device_tables(sf)
        out, counters = q3.run(customer.pages(page_rows - page_rows % 20), orders.pages(page_rows), lineitem.pages(page_rows),
                               stream.handle, comm=comm, distributed=True)
        rows = [r for p in out for r in p.to_rows()]
        rows_equal_ignore_order(rows, expected, rel=1e-9)
        assert counters["build2_rows"] == exp_orders
        assert sum(r[4] for r in rows) == exp_joined
        assert counters["orders_dynamic_filter"] is True and counters["lineitem_dynamic_filter"] is True
        assert counters["exchange_rows_sent"] == counters["exchange_rows_received"] > 0 and counters["exchange_bytes_remote"] == 0
        stream.destroy()
    finally:
        comm.destroy()


def q3_shared_gpu_worker(rank, world, port, sf, page_rows, q):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from presto_amd import _lib, q3
    from presto_amd.exchange import Comm
    _lib.init(0)
    comm = Comm.host()
    try:
        nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
        total = sf * world
        # rank r holds rows [r n, (r + 1) n) of the SF x world tables, as bench.py lays them out
        customer = tpch.DeviceColumns(tpch.CUSTOMER_COLUMNS, total, nc, first_row=rank * nc)
        orders = tpch.DeviceColumns(tpch.ORDERS_COLUMNS, total, no, first_row=rank * no)
        lineitem = tpch.DeviceColumns(tpch.Q3_LINEITEM_COLUMNS, total, nl, first_row=rank * nl)
        stream = DeviceStream()
        pr = page_rows * (rank + 1)   # different page counts per rank
        out, counters = q3.run(customer.pages(pr - pr % 20), orders.pages(pr), lineitem.pages(pr), stream.handle, comm=comm)
        q.put((rank, [r for p in out for r in p.to_rows()], counters))
        dist.barrier()
        stream.destroy()
    finally:
        comm.destroy()
        dist.destroy_process_group()


def test_q3_between_ranks_sharing_the_gpu(gpu, oracle):
    """BASELINE config #4's shape with 2 ranks: each holds a row range of the three tables, every join side is hash-partitioned
    and exchanged natively (host transport over gloo: the ranks share the box's one GPU), the dynamic filters are the union
    of both ranks' build keys.  The union of the ranks' grouped results must be the single-process oracle result, every group
    on exactly one rank."""
    import socket
    import torch.multiprocessing as mp
    world, sf, page_rows = 2, 0.03, 1 << 15
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=q3_shared_gpu_worker, args=(r, world, port, sf, page_rows, q)) for r in range(world)]
    [p.start() for p in procs]
    results = dict((r[0], r[1:]) for r in (q.get(timeout=300) for _ in range(world)))
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    nc, no, nl = tpch.customer_rows(sf) * world, tpch.orders_rows(sf) * world, tpch.lineitem_rows(sf) * world
    total = sf * world
    expected, exp_orders, exp_joined = oracle_q3(oracle, host_table(oracle, tpch.CUSTOMER_COLUMNS, total, nc),
                                                 host_table(oracle, tpch.ORDERS_COLUMNS, total, no),
                                                 host_table(oracle, tpch.Q3_LINEITEM_COLUMNS, total, nl))
    rows = results[0][0] + results[1][0]
    assert len(results[0][0]) > 0 and len(results[1][0]) > 0
    rows_equal_ignore_order(rows, expected, rel=1e-9)
    assert not ({r[0] for r in results[0][0]} & {r[0] for r in results[1][0]})   # an orderkey (group) lives on one rank
    assert results[0][1]["build2_rows"] + results[1][1]["build2_rows"] == exp_orders
    for r in range(world):
        c = results[r][1]
        assert c["orders_dynamic_filter"] is True and c["lineitem_dynamic_filter"] is True and c["exchange_bytes_remote"] > 0
    assert sum(results[r][1]["exchange_rows_sent"] for r in range(world)) == sum(results[r][1]["exchange_rows_received"] for r in range(world))


def test_q3_with_topn(gpu, oracle):
    """The whole query: ... GROUP BY ... ORDER BY revenue DESC, orderdate LIMIT 10 (TopNOperator behind the aggregation)."""
    from presto_amd import q3
    sf, page_rows = 0.1, 1 << 17
    expected, _, _ = _expected(oracle, sf)
    # oracle TopN over the oracle's grouped result: rows are (orderkey, orderdate, shippriority, revenue, count)
    from presto_amd.page import Block, Page
    cols = list(zip(*expected))
    page = Page([Block.bigint(cols[0]), Block.date(cols[1]), Block.integer(cols[2]), Block.double(cols[3]), Block.bigint(cols[4])], len(expected))
    top = oracle.topn([page], 10, [3, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST])
    stream = DeviceStream()
    customer, orders, lineitem = _device_tables(sf)
    out, _ = q3.run(customer.pages(page_rows - page_rows % 20), orders.pages(page_rows), lineitem.pages(page_rows), stream.handle,
                    distributed=False, top_n=10)
    rows = [r for p in out for r in p.to_rows()]
    assert len(rows) == 10
    for g, e in zip(rows, top):
        assert g[:3] == e[:3] and g[4] == e[4] and abs(g[3] - e[3]) <= 1e-9 * abs(e[3])
    stream.destroy()


@pytest.mark.parametrize("sf", [100.0, 300.0])
def test_q3_full_size_independent_paths_agree(gpu, sf):
    """BASELINE config #4 at full size (SF100: 765 M input rows) and config #5's Q3 leg (SF300: 2.3 G input rows on the one GPU,
    lineitem row offsets beyond 2^30 and byte offsets beyond 2^32), tied to the small-scale oracle parity above through a
    size-independent property: the TopN result must not depend on the path -- with / without the joins' dynamic filters (rows
    dropped before the probe vs inside it), with / without the extra count(*) (implicit vs explicit count word), 2^28- vs
    2^26-row pages, fused probe kernel with build-row accumulators vs FilterAndProject -> LookupJoin -> HashAggregation with a
    hashed table."""
    from presto_amd import q3
    customer, orders, lineitem = _device_tables(sf)
    assert lineitem.rows == int(6001215 * sf)
    stream = DeviceStream()

    def run(page_rows, **kw):
        out, counters = q3.run(customer.pages(page_rows - page_rows % 20), orders.pages(page_rows), lineitem.pages(page_rows), stream.handle,
                               distributed=False, top_n=10, **kw)
        return [r for p in out for r in p.to_rows()], counters

    base, counters = run(1 << 28)
    assert len(base) == 10 and counters["lineitem_dynamic_filter"] == "fused" and counters["build2_rows"] > 100_000 * sf
    revenue = [r[3] for r in base]
    assert revenue == sorted(revenue, reverse=True)
    plain, counters2 = run(1 << 28, dynamic_filters=False)
    assert "lineitem_dynamic_filter" not in counters2

    def same(a, b, cols):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x[:3] == y[:3] and all(abs(x[c] - y[c]) <= 1e-9 * abs(y[c]) for c in cols), (x, y)

    same(base, plain, [3])
    assert [r[4] for r in base] == [r[4] for r in plain]  # count(*) per group, exactly
    # the aggregation told that a TopN is its only consumer (thousands of groups leave the table) vs every group emitted
    assert counters["topn_hint"] is True
    every, counters4 = run(1 << 28, topn_hint=False)
    assert "topn_hint" not in counters4
    same(every, base, [3])
    assert [r[4] for r in base] == [r[4] for r in every]
    same(run(1 << 28, with_count=False)[0], base, [3])
    same(run(1 << 26)[0], base, [3])
    # the lineitem pipeline as one generated kernel (above) vs as three operators with a hashed group table
    unfused, counters3 = run(1 << 28, fused_probe=False)
    assert counters3["lineitem_dynamic_filter"] is True
    same(unfused, base, [3])
    assert [r[4] for r in base] == [r[4] for r in unfused]
    stream.destroy()


def test_q3_sf100_far_offset_sample_matches_oracle(gpu, oracle):
    """BASELINE config #4 at full size against the ORACLE, on a sample far into the tables: the generator is regular -- 7 orders own 28
    consecutive lineitem rows, order row o has orderkey sparse(o) -- so the orders [o0, o0 + m) and their lineitem rows
    [4 o0, 4 (o0 + m)) form a closed slice of the join graph once every customer is known.  The oracle's Q3 composition over (the whole
    SF100 customer table, that orders slice, that lineitem slice) must give exactly the groups the device's full-size run (765 M input
    rows, every group emitted) holds for those orderkeys: keys, dates, priorities and count(*) bit-exact, sum(revenue) to 1e-9."""
    import numpy as np
    from presto_amd import q3
    sf = 100.0
    customer, orders, lineitem = _device_tables(sf)
    stream = DeviceStream()
    out, counters = q3.run(customer.pages(1 << 28), orders.pages(1 << 28), lineitem.pages(1 << 28), stream.handle, distributed=False, top_n=0)
    cols = [np.concatenate([np.asarray(p.blocks[c].values)[:p.position_count] for p in out]) for c in range(5)]
    assert len(cols[0]) == len(np.unique(cols[0])) > 1_000_000 * 10   # one group per joined orderkey
    checked = 0
    for o0, m in ((7 * 15_000_000, 7 * 20_000), (7 * 21_428_000, 7 * 5_000)):   # order rows 105 M .. and the table's last orders
        m = min(m, tpch.orders_rows(sf) - o0)
        # (the table's last order owns every lineitem row behind 4 x orders: the slice then runs to the end of lineitem)
        l_rows = (tpch.lineitem_rows(sf) if o0 + m == tpch.orders_rows(sf) else 4 * (o0 + m)) - 4 * o0
        ref_rows, _, _ = oracle.q3(host_table(oracle, tpch.CUSTOMER_COLUMNS, sf, tpch.customer_rows(sf)),
                                   host_table_at(oracle, tpch.ORDERS_COLUMNS, sf, o0, m), host_table_at(oracle, tpch.Q3_LINEITEM_COLUMNS, sf, 4 * o0, l_rows))
        expected = {r[0]: r for r in ref_rows}
        lo = min(expected) if expected else 0
        hi = max(expected) if expected else 0
        sel = np.nonzero((cols[0] >= lo) & (cols[0] <= hi))[0]
        got = {int(cols[0][i]): tuple(c[i].item() for c in cols) for i in sel.tolist()}
        assert len(expected) > m // 20 and set(got) == set(expected)
        for k, e in expected.items():
            g = got[k]
            assert g[:3] == e[:3] and g[4] == e[4] and abs(g[3] - e[3]) <= 1e-9 * abs(e[3]), (g, e)
        checked += len(expected)
    assert checked > 5_000
    stream.destroy()
