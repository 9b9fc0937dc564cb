"""Q3-shaped operator pipeline (BASELINE config #4 on one GPU): three Driver pipelines chained through device-resident
pages on one stream -- customer -> filter -> HashBuilder; orders -> filter -> LookupJoin -> HashBuilder;
lineitem -> filter/project -> LookupJoin -> HashAggregation(orderkey, orderdate, shippriority; sum(revenue)) --
against the same composition of the oracle's operators.  Join multiset, BIGINT/DATE keys and group counts are
bit-exact; revenue sums within 1e-9 (stated tolerance)."""
import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd._lib import DeviceStream
from presto_amd.operators import (Driver, FilterAndProjectOperator, HashAggregationOperator, HashBuilderOperator, LookupJoinOperator,
                                  LookupSourceFactory)
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


@pytest.mark.parametrize("sf,page_rows", [(0.02, 1 << 14), (0.1, 1 << 17)])
def test_q3_pipeline_matches_oracle(gpu, oracle, sf, page_rows):
    nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
    expected, exp_orders, exp_joined = oracle_q3(oracle, host_table(oracle, tpch.CUSTOMER_COLUMNS, sf, nc),
                                                 host_table(oracle, tpch.ORDERS_COLUMNS, sf, no),
                                                 host_table(oracle, tpch.Q3_LINEITEM_COLUMNS, sf, nl))
    stream = DeviceStream()
    s = stream.handle
    dev = abi.MEM_DEVICE
    customer = tpch.DeviceColumns(tpch.CUSTOMER_COLUMNS, sf, nc)
    orders = tpch.DeviceColumns(tpch.ORDERS_COLUMNS, sf, no)
    lineitem = tpch.DeviceColumns(tpch.Q3_LINEITEM_COLUMNS, sf, nl)

    # pipeline 1: customer -> FilterAndProject -> HashBuilder (JoinBridge b1)
    b1 = LookupSourceFactory()
    Driver(customer.pages(page_rows - page_rows % 20), [
        FilterAndProjectOperator(tpch.CUSTOMER_TYPES, tpch.q3_customer_filter(), [field(0, abi.BIGINT)], output_mem=dev, stream=s),
        HashBuilderOperator(b1, [abi.BIGINT], [0], [], stream=s)]).run()
    # pipeline 2: orders -> FilterAndProject -> LookupJoin(b1) -> HashBuilder (b2)
    b2 = LookupSourceFactory()
    join1 = LookupJoinOperator(b1, tpch.ORDERS_TYPES, [1], [0, 2, 3], output_mem=dev, stream=s)
    Driver(orders.pages(page_rows), [
        FilterAndProjectOperator(tpch.ORDERS_TYPES, tpch.q3_orders_filter(), [field(i, t) for i, t in enumerate(tpch.ORDERS_TYPES)], output_mem=dev, stream=s),
        join1,
        HashBuilderOperator(b2, [abi.BIGINT, abi.DATE, abi.INTEGER], [0], [1, 2], stream=s)]).run()
    # pipeline 3: lineitem -> FilterAndProject -> LookupJoin(b2) -> HashAggregation
    agg = HashAggregationOperator([abi.BIGINT, abi.DOUBLE, abi.DATE, abi.INTEGER], [0, 2, 3], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)],
                                  expected_groups=100000, stream=s)
    out = Driver(lineitem.pages(page_rows), [
        FilterAndProjectOperator(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), output_mem=dev, stream=s),
        LookupJoinOperator(b2, [abi.BIGINT, abi.DOUBLE], [0], [0, 1], output_mem=dev, stream=s),
        agg]).run()
    rows = [r for p in out for r in p.to_rows()]
    assert len(expected) > 100
    rows_equal_ignore_order(rows, expected, rel=1e-9)
    key, links = b2.tables()
    assert len(links) == exp_orders  # orders JOIN customer rows that reached the second build side
    assert sum(r[4] for r in rows) == exp_joined
    stream.destroy()
