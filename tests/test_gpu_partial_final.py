"""Step.PARTIAL -> Step.FINAL on device equals Step.SINGLE (counts, keys, BIGINT sums exactly; DOUBLE sums to 1e-12),
and the intermediate pages equal the oracle's."""
import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd.exchange import partial_layout
from presto_amd.operators import FusedAggregationOperator, HashAggregationOperator, AggregationOperator, to_pages
from presto_amd.page import Block, Page
from tests.util import rows_equal_ignore_order

pytestmark = pytest.mark.gpu


def test_partial_then_final_equals_single(gpu, oracle):
    rng = np.random.default_rng(8)
    n = 120007
    page = Page([Block.bigint(rng.integers(0, 6, n)), Block.double(rng.random(n), rng.random(n) < 0.1), Block.bigint(rng.integers(-1000, 1000, n))], n)
    types = [abi.BIGINT, abi.DOUBLE, abi.BIGINT]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_COUNT, 1, abi.DOUBLE),
            (abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_AVG, 2, abi.BIGINT)]
    single = [r for p in to_pages(HashAggregationOperator(types, [0], aggs), [page]) for r in p.to_rows()]
    halves = [page.get_region(0, 50001), page.get_region(50001, n - 50001)]
    ptypes, faggs = partial_layout([abi.BIGINT], aggs)
    partial_pages = []
    for h in halves:
        out = to_pages(HashAggregationOperator(types, [0], aggs, step=abi.STEP_PARTIAL), [h])
        assert [b.type for b in out[0].blocks] == ptypes
        ref = oracle.HashAggregation(types, [0], aggs, step=abi.STEP_PARTIAL)
        ref.add_page(h)
        rows_equal_ignore_order(out[0].to_rows(), ref.build_result().to_rows(), rel=1e-12)
        partial_pages += out
    final = [r for p in to_pages(HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL), partial_pages) for r in p.to_rows()]
    rows_equal_ignore_order(final, single, rel=1e-12)
    oref = oracle.HashAggregation(ptypes, [0], faggs, step=abi.STEP_FINAL)
    for p in partial_pages:
        oref.add_page(p)
    rows_equal_ignore_order(final, oref.build_result().to_rows(), rel=1e-12)


def test_q1_partial_final_and_global_q6(gpu, oracle):
    n, sf = 200000, 0.1
    dev = tpch.DeviceColumns(tpch.Q1_COLUMNS, sf, n)
    mk = lambda step: FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                               type_params=tpch.Q1_TYPE_PARAMS, step=step)
    single = [r for p in to_pages(mk(abi.STEP_SINGLE), list(dev.pages(1 << 16))) for r in p.to_rows()]
    pages = list(dev.pages(1 << 16))
    parts = to_pages(mk(abi.STEP_PARTIAL), pages[:2]) + to_pages(mk(abi.STEP_PARTIAL), pages[2:])
    ptypes, faggs = partial_layout([abi.VARCHAR, abi.VARCHAR], tpch.Q1_AGGREGATES)
    final = [r for p in to_pages(HashAggregationOperator(ptypes, [0, 1], faggs, step=abi.STEP_FINAL, type_params=[1, 1] + [0] * (len(ptypes) - 2)), parts)
             for r in p.to_rows()]
    rows_equal_ignore_order(final, single, rel=1e-12)
    # global aggregation (Q6 shape)
    dev6 = tpch.DeviceColumns(tpch.Q6_COLUMNS, sf, n)
    aggs6 = tpch.Q6_AGGREGATES + [(abi.AGG_COUNT_STAR, -1, None)]
    mk6 = lambda step: FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], aggs6, step=step)
    (s, c), = to_pages(mk6(abi.STEP_SINGLE), list(dev6.pages(1 << 16)))[0].to_rows()
    pages6 = list(dev6.pages(1 << 16))
    parts6 = to_pages(mk6(abi.STEP_PARTIAL), pages6[:1]) + to_pages(mk6(abi.STEP_PARTIAL), pages6[1:])
    ptypes6, faggs6 = partial_layout([], aggs6)
    (fs, fc), = to_pages(AggregationOperator(ptypes6, faggs6, step=abi.STEP_FINAL), parts6)[0].to_rows()
    assert fc == c and abs(fs - s) <= 1e-12 * abs(s)
