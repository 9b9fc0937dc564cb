"""DECIMAL on the device against the oracle (which the reference's own known answers pin: tests/test_oracle_decimal.py): every
result is an exact integer -- no tolerance anywhere.  Expressions (DecimalOperators' add / subtract / multiply with rescaling,
short and long results, casts, comparisons), sum / avg / min / max / count through every tier of the fused aggregation,
PARTIAL -> FINAL, decimal group keys, and TPC-H Q6 / Q1 over DECIMAL(12, 2) columns end to end."""
import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd.exchange import partial_layout
from presto_amd.expr import and_, constant, field
from presto_amd._lib import PrestoAmdError
from presto_amd.operators import AggregationOperator, FilterAndProjectOperator, FusedAggregationOperator, HashAggregationOperator, to_pages, upload_page
from presto_amd.page import Block, Page
from tests.util import rows_equal_ignore_order

pytestmark = pytest.mark.gpu
D = abi.decimal(12, 2)


def decimal_page(rng, n, null_rate=0.1):
    v = rng.integers(-10 ** 11, 10 ** 11, n)
    w = rng.integers(0, 11, n)
    k = rng.integers(0, 7, n)
    return Page([Block.decimal(v, rng.random(n) < null_rate), Block.decimal(w), Block.bigint(k)], n)


TYPES = [D, D, abi.BIGINT]


@pytest.mark.parametrize("device_pages", [False, True])
def test_decimal_expressions_bit_exact(gpu, oracle, device_pages):
    rng = np.random.default_rng(1)
    page = decimal_page(rng, 20011)
    a, b = field(0, D), field(1, D)
    one = constant(1, abi.decimal(10, 0))
    projections = [a + b, a - b, a * b, (one - b), a * (one - b), a * (one - b) * (one + b), -a, (a * b) + (a * b),
                   field(2, abi.BIGINT).cast(abi.decimal(18, 3)), a.cast(abi.decimal(14, 4)), a.cast(abi.decimal(12, 0)), a.cast(abi.decimal(30, 10))]
    flt = and_(b >= constant(2, D), b.between(constant(1, D), constant(9, D)), (b * b) > constant(3, abi.decimal(24, 4)))
    expected = oracle.filter_project(page, flt, projections)
    op = FilterAndProjectOperator(TYPES, flt, projections)
    out = to_pages(op, [upload_page(page) if device_pages else page])
    assert [blk.type for blk in out[0].blocks] == [blk.type for blk in expected.blocks]
    assert out[0].to_rows() == expected.to_rows() and out[0].position_count > 5000
    assert abi.LONG_DECIMAL in [blk.type for blk in out[0].blocks] and any(r[0] is None for r in out[0].to_rows())


def test_decimal_overflow_is_numeric_value_out_of_range(gpu):
    big = abi.decimal(38, 0)
    page = Page([Block.long_decimal([10 ** 38 - 1, 5]), Block.long_decimal([1, 1])], 2)
    op = FilterAndProjectOperator([big, big], None, [field(0, big) + field(1, big)])
    with pytest.raises(PrestoAmdError) as err:
        to_pages(op, [page])
    assert err.value.status == abi.ERR_NUMERIC_VALUE_OUT_OF_RANGE
    op = FilterAndProjectOperator([abi.BIGINT], None, [field(0, abi.BIGINT).cast(abi.decimal(5, 2))])
    with pytest.raises(PrestoAmdError) as err:
        to_pages(op, [Page([Block.bigint([999, 1000])], 2)])
    assert err.value.status == abi.ERR_NUMERIC_VALUE_OUT_OF_RANGE


AGGS = [(abi.AGG_SUM, 0, D), (abi.AGG_AVG, 0, D), (abi.AGG_MIN, 0, D), (abi.AGG_MAX, 0, D), (abi.AGG_COUNT, 0, D), (abi.AGG_COUNT_STAR, -1, None),
        (abi.AGG_SUM, 1, D)]


@pytest.mark.parametrize("groups", [0, 5, 300, 20000])
def test_decimal_aggregates_through_every_tier(gpu, oracle, groups):
    """global / few groups / LDS table / HBM table: sums are DECIMAL(38, 2) values, averages round half up in DECIMAL(12, 2)."""
    rng = np.random.default_rng(groups)
    pages = []
    for _ in range(3):
        n = 40000
        k = rng.integers(0, max(groups, 1), n)
        pages.append(Page([Block.decimal(rng.integers(-10 ** 11, 10 ** 11, n), rng.random(n) < 0.1), Block.decimal(rng.integers(0, 11, n)), Block.bigint(k)], n))
    gb = [2] if groups else []
    ref = oracle.HashAggregation(TYPES, gb, AGGS)
    for p in pages:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    op = HashAggregationOperator(TYPES, gb, AGGS, expected_groups=max(groups, 1)) if groups else AggregationOperator(TYPES, AGGS)
    rows = [r for p in to_pages(op, pages) for r in p.to_rows()]
    rows_equal_ignore_order(rows, expected)
    # PARTIAL on two halves -> FINAL
    ptypes, faggs = partial_layout([abi.BIGINT] if groups else [], AGGS)
    mk = (lambda **kw: HashAggregationOperator(TYPES, gb, AGGS, **kw)) if groups else (lambda **kw: AggregationOperator(TYPES, AGGS, **kw))
    parts = to_pages(mk(step=abi.STEP_PARTIAL), pages[:1]) + to_pages(mk(step=abi.STEP_PARTIAL), pages[1:])
    assert [b.type for b in parts[0].blocks] == [int(t) for t in ptypes]
    fin = (HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL) if groups else AggregationOperator(ptypes, faggs, step=abi.STEP_FINAL))
    final = [r for p in to_pages(fin, parts) for r in p.to_rows()]
    rows_equal_ignore_order(final, expected)


def test_decimal_group_keys_and_hash_channel(gpu, oracle):
    rng = np.random.default_rng(9)
    n = 30000
    page = Page([Block.decimal(rng.integers(-50, 50, n), rng.random(n) < 0.05), Block.decimal(rng.integers(0, 1000, n)), Block.bigint(rng.integers(0, 3, n))], n)
    aggs = [(abi.AGG_SUM, 1, D), (abi.AGG_COUNT_STAR, -1, None)]
    ref = oracle.HashAggregation(TYPES, [0, 2], aggs)
    ref.add_page(page)
    rows = [r for p in to_pages(HashAggregationOperator(TYPES, [0, 2], aggs), [page]) for r in p.to_rows()]
    rows_equal_ignore_order(rows, ref.build_result().to_rows())


def decimal_lineitem(oracle, columns, sf, n):
    """the synthetic lineitem columns with money / quantity as DECIMAL(12, 2): the generator's two-decimal doubles times 100"""
    blocks = []
    for c in columns:
        v, o = oracle.tpch_column(c, sf, 0, n)
        t = abi.TPCH_COLUMN_TYPE[c]
        if t == abi.VARCHAR:
            blocks.append(Block.varwidth(v, o))
        elif t == abi.DOUBLE:
            blocks.append(Block.decimal(np.rint(v * 100).astype(np.int64)))
        else:
            blocks.append(Block.flat(t, v))
    return Page(blocks, n)


def test_tpch_q6_and_q1_with_decimal_columns_bit_exact(gpu, oracle):
    """Q6 and Q1 over DECIMAL(12, 2) columns, fused scan-filter-project-aggregate: equal to the oracle's operators digit for digit
    (a parity upgrade over the DOUBLE form, whose sums depend on the order of addition)."""
    n, sf = 200003, 0.1
    p6 = decimal_lineitem(oracle, tpch.Q6_COLUMNS, sf, n)
    proj6, aggs6 = tpch.q6_decimal_projections(), tpch.q6_decimal_aggregates() + [(abi.AGG_COUNT_STAR, -1, None)]
    ref = oracle.HashAggregation([proj6[0].type], [], aggs6)
    ref.add_page(oracle.filter_project(p6, tpch.q6_decimal_filter(), proj6))
    expected6 = ref.build_result().to_rows()
    op = FusedAggregationOperator(tpch.Q6_DECIMAL_TYPES, tpch.q6_decimal_filter(), proj6, [], aggs6)
    pages = [p6.get_region(i, min(65536, n - i)) for i in range(0, n, 65536)]
    assert to_pages(op, [upload_page(p) for p in pages])[0].to_rows() == expected6 and expected6[0][1] > 3000
    # the DOUBLE form agrees to the tolerance it states
    d6 = [oracle.tpch_column(c, sf, 0, n)[0] for c in tpch.Q6_COLUMNS]
    assert abs(oracle.q6(*d6)[0] - expected6[0][0] / 10 ** 4) <= 1e-9 * expected6[0][0] / 10 ** 4

    p1 = decimal_lineitem(oracle, tpch.Q1_COLUMNS, sf, n)
    proj1, aggs1 = tpch.q1_decimal_projections(), tpch.q1_decimal_aggregates()
    tp = [1, 1] + [abi.type_param(tpch.DEC)] * 4 + [0]
    ref = oracle.HashAggregation([p.type for p in proj1], tpch.Q1_GROUP_BY, aggs1)
    ref.add_page(oracle.filter_project(p1, tpch.q1_filter(), proj1))
    expected1 = ref.build_result().to_rows()
    assert len(expected1) == 4
    for page_rows in (n, 50000):
        op = FusedAggregationOperator(tpch.Q1_DECIMAL_TYPES, tpch.q1_filter(), proj1, tpch.Q1_GROUP_BY, aggs1, type_params=tp)
        pages = [upload_page(p1.get_region(i, min(page_rows, n - i))) for i in range(0, n, page_rows)]
        rows = [r for p in to_pages(op, pages) for r in p.to_rows()]
        rows_equal_ignore_order(rows, expected1)
