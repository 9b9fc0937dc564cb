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


def oracle_q3(oracle, customer, orders, lineitem):
    c = oracle.filter_project(customer, tpch.q3_customer_filter(), [field(0, abi.BIGINT)])
    j1 = oracle.HashJoin([abi.BIGINT], [0], [])
    j1.add_build_page(c)
    j1.build()
    o = oracle.filter_project(orders, tpch.q3_orders_filter(), [field(i, t) for i, t in enumerate(tpch.ORDERS_TYPES)])
    oc, _, _ = j1.probe(o, tpch.ORDERS_TYPES, [1], [0, 2, 3])
    j2 = oracle.HashJoin([abi.BIGINT, abi.DATE, abi.INTEGER], [0], [1, 2])
    j2.add_build_page(oc)
    j2.build()
    l = oracle.filter_project(lineitem, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections())
    joined, _, _ = j2.probe(l, [abi.BIGINT, abi.DOUBLE], [0], [0, 1])
    agg = oracle.HashAggregation([abi.BIGINT, abi.DOUBLE, abi.DATE, abi.INTEGER], [0, 2, 3], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)],
                                 expected_groups=100000)
    agg.add_page(joined)
    return agg.build_result().to_rows(), oc.position_count, joined.position_count


def _device_tables(sf):
    nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
    return (tpch.DeviceColumns(tpch.CUSTOMER_COLUMNS, sf, nc), tpch.DeviceColumns(tpch.ORDERS_COLUMNS, sf, no),
            tpch.DeviceColumns(tpch.Q3_LINEITEM_COLUMNS, sf, nl))


def _expected(oracle, sf):
    nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
    return oracle_q3(oracle, host_table(oracle, tpch.CUSTOMER_COLUMNS, sf, nc), host_table(oracle, tpch.ORDERS_COLUMNS, sf, no),
                     host_table(oracle, tpch.Q3_LINEITEM_COLUMNS, sf, nl))


@pytest.mark.parametrize("sf,page_rows", [(0.02, 1 << 14), (0.1, 1 << 17)])
def test_q3_pipeline_matches_oracle(gpu, oracle, sf, page_rows):
    from presto_amd import q3
    expected, exp_orders, exp_joined = _expected(oracle, sf)
    stream = DeviceStream()
    customer, orders, lineitem = _device_tables(sf)
    out, counters = q3.run(customer.pages(page_rows - page_rows % 20), orders.pages(page_rows), lineitem.pages(page_rows), stream.handle,
                           distributed=False)
    rows = [r for p in out for r in p.to_rows()]
    assert len(expected) > 100
    rows_equal_ignore_order(rows, expected, rel=1e-9)
    assert counters["build2_rows"] == exp_orders  # orders JOIN customer rows that reached the second build side
    assert sum(r[4] for r in rows) == exp_joined
    stream.destroy()


def test_q3_with_exchange_steps_on_one_rank_over_rccl(gpu, oracle):
    """The multi-GPU form of the pipelines (an ExchangeOperator before every build and probe) with a one-rank RCCL
    group: every all-to-all is a self copy, so the result must equal the oracle's, and every row must come back."""
    import os
    import torch
    import torch.distributed as dist
    from presto_amd import q3
    sf, page_rows = 0.05, 1 << 16
    expected, exp_orders, exp_joined = _expected(oracle, sf)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.zeros(1, device="cuda")  # initialise torch's view of the device before the process group asks for it
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        stream = DeviceStream()
        customer, orders, lineitem = _device_tables(sf)
        out, counters = q3.run(customer.pages(page_rows - page_rows % 20), orders.pages(page_rows), lineitem.pages(page_rows),
                               stream.handle, distributed=True)
        torch.cuda.synchronize()
        rows = [r for p in out for r in p.to_rows()]
        rows_equal_ignore_order(rows, expected, rel=1e-9)
        assert counters["build2_rows"] == exp_orders
        assert sum(r[4] for r in rows) == exp_joined
        stream.destroy()
    finally:
        dist.destroy_process_group()


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


def test_q3_sf100_independent_paths_agree(gpu):
    """BASELINE config #4 at full size (765 M input rows), tied to the small-scale oracle parity above through a size-independent
    property: the TopN result must not depend on the path -- with / without the joins' dynamic filters (rows dropped before
    the probe vs inside it), with / without the extra count(*) (implicit vs explicit count word), 2^28- vs 2^26-row pages."""
    from presto_amd import q3
    sf = 100.0
    customer, orders, lineitem = _device_tables(sf)
    stream = DeviceStream()

    def run(page_rows, **kw):
        out, counters = q3.run(customer.pages(page_rows - page_rows % 20), orders.pages(page_rows), lineitem.pages(page_rows), stream.handle,
                               distributed=False, top_n=10, **kw)
        return [r for p in out for r in p.to_rows()], counters

    base, counters = run(1 << 28)
    assert len(base) == 10 and counters["lineitem_dynamic_filter"] is True and counters["build2_rows"] > 10_000_000
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
    same(run(1 << 28, with_count=False)[0], base, [3])
    same(run(1 << 26)[0], base, [3])
    stream.destroy()
