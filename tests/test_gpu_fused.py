"""GPU parity of the fused scan-filter-project-aggregate operator (Q6 / Q1 shapes) against the oracle.

DOUBLE tolerance: the reference adds f64 values strictly left to right (DoubleSumAggregation.java:33-38);
the device adds per lane / wave / workgroup partials, so sums agree to rounding only.  Stated tolerance
(BASELINE.md section 3): |gpu - cpu| <= 1e-9 * max(|gpu|, |cpu|) against the sequential CPU sum -- whose own
rounding error grows with the row count -- and, tighter, |gpu - exact| <= 1e-12 * |exact| against the correctly
rounded sum (math.fsum) of the same projected values.  The reference's own tests compare 5 significant digits.
Counts, group keys and row selection are bit-exact.
"""
import math

import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd.expr import and_, constant, field
from presto_amd.operators import (AggregationOperator, FusedAggregationOperator, HashAggregationOperator, download_page,
                                  to_pages, upload_page)
from presto_amd.page import Block, Page, sequence_page
from tests.util import rows_equal_ignore_order

pytestmark = pytest.mark.gpu
REL = 1e-9
REL_EXACT = 1e-12


def host_lineitem(oracle, columns, sf, first, n):
    blocks = []
    for col in columns:
        v, o = oracle.tpch_column(col, sf, first, n)
        t = abi.TPCH_COLUMN_TYPE[col]
        blocks.append(Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v))
    return Page(blocks, n)


@pytest.mark.parametrize("column", list(range(14)))
def test_tpch_generator_matches_oracle(gpu, oracle, column):
    n, first = 10007 if column != abi.C_MKTSEGMENT else 10005, 12345
    dev = tpch.DeviceColumns([column], 0.05, first + n)
    page = download_page(dev.page(first, n))
    v, o = oracle.tpch_column(column, 0.05, first, n)
    b = page.blocks[0]
    if o is None:
        assert np.array_equal(b.values.view(np.uint8), v.view(np.uint8))
    else:
        # device offsets are relative to row 0 of the generated table, oracle's to `first`
        base = int(b.offsets[0])
        assert np.array_equal(b.offsets - base, o)
        assert bytes(b.values[base:base + int(o[-1])]) == bytes(v[:int(o[-1])])


def q6_operator(**kw):
    aggs = tpch.Q6_AGGREGATES + [(abi.AGG_COUNT_STAR, -1, None)]
    return FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], aggs, **kw)


@pytest.mark.parametrize("n", [1, 3, 4, 5, 1000, 250003])
def test_q6_fused_matches_oracle(gpu, oracle, n):
    sf = 0.1
    host = host_lineitem(oracle, tpch.Q6_COLUMNS, sf, 0, n)
    ref_sum, ref_count = oracle.q6(*(b.values for b in host.blocks))
    dev = tpch.DeviceColumns(tpch.Q6_COLUMNS, sf, n)
    op = q6_operator()
    out = to_pages(op, list(dev.pages(100000)))
    assert len(out) == 1
    (revenue, count), = out[0].to_rows()
    assert count == ref_count
    if ref_count == 0:
        assert revenue is None  # DoubleSumAggregation.output: NULL when no input row
    else:
        assert abs(revenue - ref_sum) <= REL * abs(ref_sum)
    # and through the generic oracle operators (PageProcessor -> AggregationOperator)
    projected = oracle.filter_project(host, tpch.q6_filter(), tpch.q6_projections())
    agg = oracle.HashAggregation([abi.DOUBLE], [], [(abi.AGG_SUM, 0, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)])
    if projected is not None:
        agg.add_page(projected)
    (osum, ocount), = agg.build_result().to_rows()
    assert ocount == ref_count and (osum is None) == (ref_count == 0)
    if ref_count:
        assert osum == ref_sum  # the two oracle paths add in the same order: bit-exact
        exact = math.fsum(projected.blocks[0].values.tolist())
        assert abs(revenue - exact) <= REL_EXACT * abs(exact)
    op.close()


def test_q6_host_pages_with_nulls(gpu, oracle):
    rng = np.random.default_rng(7)
    n = 40001
    host = host_lineitem(oracle, tpch.Q6_COLUMNS, 0.1, 0, n)
    blocks = []
    for b in host.blocks:
        nulls = (rng.random(n) < 0.1).astype(np.uint8)
        blocks.append(Block(b.type, abi.FLAT, n, values=b.values, nulls=nulls))
    host = Page(blocks, n)
    op = q6_operator()
    pages = [host.get_region(0, 12345), host.get_region(12345, n - 12345)]  # second region is unaligned
    out = to_pages(op, pages)
    (revenue, count), = out[0].to_rows()
    projected = oracle.filter_project(host, tpch.q6_filter(), tpch.q6_projections())
    agg = oracle.HashAggregation([abi.DOUBLE], [], [(abi.AGG_SUM, 0, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)])
    agg.add_page(projected)
    (osum, ocount), = agg.build_result().to_rows()
    assert count == ocount
    assert abs(revenue - osum) <= REL * abs(osum)


def q1_reference(oracle, host):
    projected = oracle.filter_project(host, tpch.q1_filter(), tpch.q1_projections())
    agg = oracle.HashAggregation([p.type for p in tpch.q1_projections()], tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES)
    agg.add_page(projected)
    return agg.build_result().to_rows()


@pytest.mark.parametrize("n", [5, 1000, 300001])
def test_q1_fused_matches_oracle(gpu, oracle, n):
    sf = 0.1
    host = host_lineitem(oracle, tpch.Q1_COLUMNS, sf, 0, n)
    expected = q1_reference(oracle, host)
    dev = tpch.DeviceColumns(tpch.Q1_COLUMNS, sf, n)
    op = FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                  type_params=tpch.Q1_TYPE_PARAMS)
    out = to_pages(op, list(dev.pages(65536)))
    rows = [r for p in out for r in p.to_rows()]
    rows_equal_ignore_order(rows, expected, rel=REL)
    # against correctly rounded per-group sums of the projected values
    projected = oracle.filter_project(host, tpch.q1_filter(), tpch.q1_projections())
    cols = [b.to_pylist() if b.type == abi.VARCHAR else b.values for b in projected.blocks]
    for row in rows:
        sel = np.array([a == row[0] and b == row[1] for a, b in zip(cols[0], cols[1])])
        assert int(sel.sum()) == row[9]
        for out_col, src in ((2, 2), (3, 3), (4, 4), (5, 5)):
            exact = math.fsum(cols[src][sel].tolist())
            assert abs(row[out_col] - exact) <= REL_EXACT * abs(exact)
        for out_col, src in ((6, 2), (7, 3), (8, 6)):
            exact = math.fsum(cols[src][sel].tolist()) / row[9]
            assert abs(row[out_col] - exact) <= REL_EXACT * abs(exact)
    op.close()


def test_q1_without_type_params_uses_two_word_keys(gpu, oracle):
    n = 50000
    host = host_lineitem(oracle, tpch.Q1_COLUMNS, 0.1, 0, n)
    expected = q1_reference(oracle, host)
    op = FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES)
    rows = [r for p in to_pages(op, [host]) for r in p.to_rows()]
    rows_equal_ignore_order(rows, expected, rel=REL)


def test_hash_aggregation_many_groups_escalates_to_hbm_table(gpu, oracle):
    """TestHashAggregationOperator.testHashAggregation shape (…/TestHashAggregationOperator.java:160-219):
    3 pages x 40 000 rows, BIGINT key -> count 3, sum 3i, avg i per key."""
    pages = [sequence_page(40000, [(abi.BIGINT, 0), (abi.BIGINT, 0)]) for _ in range(3)]
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_AVG, 1, abi.BIGINT)]
    op = HashAggregationOperator([abi.BIGINT, abi.BIGINT], [0], aggs, expected_groups=100000)
    rows = [r for p in to_pages(op, pages) for r in p.to_rows()]
    assert len(rows) == 40000
    assert sorted(rows) == [(i, 3, 3 * i, float(i)) for i in range(40000)]
    ref = oracle.HashAggregation([abi.BIGINT, abi.BIGINT], [0], aggs, expected_groups=100000)
    for p in pages:
        ref.add_page(p)
    rows_equal_ignore_order(rows, ref.build_result().to_rows(), rel=0.0)


def test_aggregation_operator_kat(gpu, oracle):
    """TestAggregationOperator.testAggregation subset (…/TestAggregationOperator.java:119-156): 100-row
    sequence page -> count 100, sum(bigint) 4950, avg 49.5, sum(double @500) 54950.0."""
    page = sequence_page(100, [(abi.BIGINT, 0), (abi.DOUBLE, 500)])
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 0, abi.BIGINT), (abi.AGG_AVG, 0, abi.BIGINT), (abi.AGG_SUM, 1, abi.DOUBLE)]
    op = AggregationOperator([abi.BIGINT, abi.DOUBLE], aggs)
    out = to_pages(op, [page])
    assert out[0].to_rows() == [(100, 4950, 49.5, 54950.0)]


def test_aggregation_function_sequence_kats_on_device(gpu):
    """The reference's per-function tests (TestCountColumnAggregation, TestLong/DoubleSum/Average/Min/MaxAggregation, TestDateMaxAggregation,
    TestShortDecimalMaxAggregation, TestBooleanMax / MinAggregation, TestRealSumAggregation) over AbstractTestAggregationFunction's cases (…/aggregation/AbstractTestAggregationFunction.java:84-140),
    as tests/test_oracle_operators.py::test_aggregation_function_sequence_kats restates them for the oracle -- here through the device's
    AggregationOperator (ungrouped) and HashAggregationOperator (one group)."""
    from tests.test_oracle_operators import SEQUENCE_CASES
    D = abi.decimal(10, 5)
    for name, start, length, shape in SEQUENCE_CASES:
        seq = list(range(start, start + length))
        if shape == "all_null":
            values, nulls = [0] * 10, [True] * 10
        elif shape == "alternating":
            values, nulls = [v for x in seq for v in (0, x)], [n for _ in seq for n in (True, False)]
        else:
            values, nulls = seq, [False] * len(seq)
        n = len(values)
        nl = np.array(nulls, dtype=bool)
        blocks = [Block.bigint(np.array(values, dtype=np.int64), nl), Block.double(np.array(values, dtype=np.float64), nl),
                  Block.date(np.array(values, dtype=np.int32), nl), Block.decimal(np.array(values, dtype=np.int64), nl),
                  Block.boolean(np.array([v % 2 != 0 for v in values], dtype=bool), nl), Block.boolean(np.array([v % 2 == 0 for v in values], dtype=bool), nl),
                  Block.real(np.array(values, dtype=np.float32), nl)]
        types = [abi.BIGINT, abi.DOUBLE, abi.DATE, D, abi.BOOLEAN, abi.BOOLEAN, abi.REAL]
        aggs = [(abi.AGG_COUNT, 0, abi.BIGINT), (abi.AGG_SUM, 0, abi.BIGINT), (abi.AGG_AVG, 0, abi.BIGINT), (abi.AGG_SUM, 1, abi.DOUBLE),
                (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_MIN, 0, abi.BIGINT), (abi.AGG_MAX, 0, abi.BIGINT), (abi.AGG_MIN, 1, abi.DOUBLE),
                (abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_MAX, 2, abi.DATE), (abi.AGG_MAX, 3, D), (abi.AGG_MAX, 4, abi.BOOLEAN), (abi.AGG_MIN, 5, abi.BOOLEAN),
                (abi.AGG_SUM, 6, abi.REAL)]
        total, lo, hi = sum(seq), start, start + length - 1
        bool_max = length > 1 or start % 2 == 1
        expected = (0,) + (None,) * 13 if length == 0 else (length, total, float(total) / length, float(total), float(total) / length, lo, hi,
                                                            float(lo), float(hi), hi, hi, bool_max, not bool_max, float(np.float32(total)))
        pages = [Page(blocks, n)] if n else []
        assert [r for p in to_pages(AggregationOperator(types, aggs), pages) for r in p.to_rows()] == [expected], name
        if n:   # the same values as one group of a grouped aggregation
            grouped = [Page([Block.bigint(np.full(n, 7, dtype=np.int64))] + blocks, n)]
            gaggs = [(a[0], a[1] + 1, a[2]) for a in aggs]
            got = [r for p in to_pages(HashAggregationOperator([abi.BIGINT] + types, [0], gaggs), grouped) for r in p.to_rows()]
            assert got == [(7,) + expected], name


def test_bigint_sum_overflow_raises(gpu):
    from presto_amd._lib import PrestoAmdError
    page = Page([Block.bigint([2 ** 62, 2 ** 62, 5])])
    op = AggregationOperator([abi.BIGINT], [(abi.AGG_SUM, 0, abi.BIGINT)])
    with pytest.raises(PrestoAmdError) as e:
        to_pages(op, [page])
    assert e.value.status == abi.ERR_NUMERIC_VALUE_OUT_OF_RANGE


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 63, 64, 65, 257, 1027])
def test_new_groups_in_partial_waves(gpu, oracle, n):
    """Regression: first occurrences of a group in a wave iteration where only some lanes have a row (tiny pages, page
    tails) must update the wave's key table consistently."""
    keys = (np.arange(n) * 7) % 8
    vals = np.arange(n, dtype=np.float64) + 0.5
    pages = [Page([Block.bigint(keys), Block.double(vals)], n), Page([Block.bigint(keys[::-1].copy()), Block.double(vals)], n)]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], aggs)
    rows = [r for p in to_pages(op, pages) for r in p.to_rows()]
    ref = oracle.HashAggregation([abi.BIGINT, abi.DOUBLE], [0], aggs)
    for p in pages:
        ref.add_page(p)
    rows_equal_ignore_order(rows, ref.build_result().to_rows(), rel=1e-12)


@pytest.mark.parametrize("groups,pages,rows", [(9, 2, 5000), (64, 3, 70001), (700, 2, 200003), (3000, 3, 150001), (50000, 2, 300007)])
def test_hash_aggregation_across_cardinalities(gpu, oracle, groups, pages, rows):
    """Every tier of the grouped aggregation against the oracle: the wave-register table (<= 8 groups), the workgroup's
    LDS table (tens to ~1000 groups; rows beyond its capacity go to the HBM table), the HBM table.  Nullable key and
    value, a second key column, a mask channel; BIGINT sums and counts bit-exact, DOUBLE within 1e-9."""
    rng = np.random.default_rng(groups)
    plist = []
    for p in range(pages):
        k1 = rng.integers(0, groups, rows).astype(np.int64) * 1000003 - 17
        k1_null = rng.random(rows) < 0.01
        k2 = (k1 % 3).astype(np.int32)
        v = rng.random(rows) * 100 - 50
        v_null = rng.random(rows) < 0.05
        b = rng.integers(-1000, 1000, rows).astype(np.int64)
        m = rng.random(rows) < 0.7
        plist.append(Page([Block.bigint(k1, k1_null), Block.integer(k2), Block.double(v, v_null), Block.bigint(b), Block.boolean(m)], rows))
    types = [abi.BIGINT, abi.INTEGER, abi.DOUBLE, abi.BIGINT, abi.BOOLEAN]
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_COUNT, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.BIGINT),
            (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 3, abi.BIGINT, 4)]
    op = HashAggregationOperator(types, [0, 1], aggs)
    got = [r for p in to_pages(op, plist) for r in p.to_rows()]
    ref = oracle.HashAggregation(types, [0, 1], aggs)
    for p in plist:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    assert len(expected) >= min(groups, 9)
    rows_equal_ignore_order(got, expected, rel=1e-9)


@pytest.mark.parametrize("expected_groups", [10000, 3_000_000])
@pytest.mark.parametrize("shape", ["few_then_many", "many_from_the_start", "many_then_few", "device_pages"])
def test_tier_hand_overs_at_high_cardinality(gpu, oracle, shape, expected_groups):
    """Page sequences that cross the tiers at hundreds of thousands of groups: the few-groups tier's first launch giving up
    (the page restarts on the next tier), the probe launch whose groups are given up when the page goes to the partition-owned
    tables (its rows are redone there), pages that arrive before / after the tier is known, tables that already hold groups
    when the cardinality jumps (nothing may be dropped then).  Every row counted exactly once."""
    from presto_amd.operators import upload_page
    rng = np.random.default_rng(len(shape) + expected_groups % 7)

    def page(rows, groups, base=0):
        k = rng.integers(0, groups, rows).astype(np.int64) * 7919 + base
        return Page([Block.bigint(k), Block.bigint(rng.integers(-100, 100, rows)), Block.double(rng.random(rows))], rows)

    if shape == "few_then_many":
        pages = [page(150_000, 5), page(1_200_000, 400_000), page(500_000, 400_000, 13)]
    elif shape == "many_then_few":
        pages = [page(1_300_000, 450_000), page(200_000, 6), page(400_000, 450_000)]
    else:
        pages = [page(1_400_000, 420_000), page(500_000, 420_000), page(300_001, 250_000, 5)]
    types = [abi.BIGINT, abi.BIGINT, abi.DOUBLE]
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_MAX, 1, abi.BIGINT)]
    ref = oracle.HashAggregation(types, [0], aggs)
    for p in pages:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    fed = [upload_page(p) for p in pages] if shape == "device_pages" else pages
    op = HashAggregationOperator(types, [0], aggs, expected_groups=expected_groups)
    got = [r for p in to_pages(op, fed) for r in p.to_rows()]
    assert len(expected) > 300_000
    rows_equal_ignore_order(got, expected, rel=1e-9)


@pytest.mark.parametrize("groups,rows", [(0, 30001), (5, 20000), (200, 100003), (5000, 200001), (60000, 200003)])
def test_min_max_across_variants_and_types(gpu, oracle, groups, rows):
    """min / max (AbstractMinMaxAggregationFunction: the type's COMPARISON operator; Double.compare for DOUBLE, so NaN is the
    maximum and -0.0 < 0.0) over BIGINT / INTEGER / DATE / DOUBLE / BOOLEAN in every variant of the aggregation -- global,
    wave-register table, workgroup LDS table, HBM table -- with nullable inputs and a mask; bit-exact against the oracle."""
    rng = np.random.default_rng(groups + 1)
    plist = []
    for p in range(2):
        d = rng.standard_normal(rows) * 100
        d[rng.random(rows) < 0.002] = np.nan
        d[rng.random(rows) < 0.01] = -0.0
        d[rng.random(rows) < 0.01] = 0.0
        plist.append(Page([Block.bigint(rng.integers(0, max(groups, 1), rows)), Block.double(d, rng.random(rows) < 0.2),
                           Block.bigint(rng.integers(-2 ** 62, 2 ** 62, rows), rng.random(rows) < 0.5), Block.integer(rng.integers(-1000, 1000, rows)),
                           Block.date(rng.integers(8000, 12000, rows)), Block.boolean(rng.random(rows) < 0.5, rng.random(rows) < 0.3),
                           Block.boolean(rng.random(rows) < 0.6)], rows))
    types = [abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.INTEGER, abi.DATE, abi.BOOLEAN, abi.BOOLEAN]
    aggs = [(abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_MIN, 2, abi.BIGINT), (abi.AGG_MAX, 2, abi.BIGINT),
            (abi.AGG_MIN, 3, abi.INTEGER), (abi.AGG_MAX, 4, abi.DATE), (abi.AGG_MIN, 5, abi.BOOLEAN), (abi.AGG_MAX, 5, abi.BOOLEAN),
            (abi.AGG_MAX, 1, abi.DOUBLE, 6), (abi.AGG_SUM, 3, abi.INTEGER), (abi.AGG_COUNT_STAR, -1, None)]
    keys = [0] if groups else []
    make = (lambda **kw: HashAggregationOperator(types, keys, aggs, **kw)) if groups else (lambda **kw: AggregationOperator(types, aggs, **kw))
    got = [r for p in to_pages(make(), plist) for r in p.to_rows()]
    ref = oracle.HashAggregation(types, keys, aggs)
    for p in plist:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    assert bits_of(sorted(got, key=lambda r: r[0] if groups else 0)) == bits_of(sorted(expected, key=lambda r: r[0] if groups else 0))
    # PARTIAL per page, FINAL over the partial pages == SINGLE
    from presto_amd.exchange import partial_layout
    ptypes, faggs = partial_layout([abi.BIGINT] if groups else [], aggs)
    partial_pages = []
    for p in plist:
        partial_pages += to_pages(make(step=abi.STEP_PARTIAL), [p])
    fkeys = [0] if groups else []
    final = HashAggregationOperator(ptypes, fkeys, faggs, step=abi.STEP_FINAL) if groups else AggregationOperator(ptypes, faggs, step=abi.STEP_FINAL)
    got2 = [r for p in to_pages(final, partial_pages) for r in p.to_rows()]
    assert bits_of(sorted(got2, key=lambda r: r[0] if groups else 0)) == bits_of(sorted(expected, key=lambda r: r[0] if groups else 0))


def bits_of(rows):
    import struct
    out = []
    for r in rows:
        out.append(tuple(("nan" if v != v else struct.pack("<d", v)) if isinstance(v, float) else v for v in r))
    return out


@pytest.mark.parametrize("groups", [0, 6, 300, 5000, 200000])
@pytest.mark.parametrize("step", ["single", "partial"])
def test_nullability_changes_between_pages(gpu, oracle, groups, step):
    """valueIsNull arrays come and go from page to page (LongArrayBlock.java:37-41: null when the block has no NULLs).  A key or
    an aggregate input that turns nullable needs NULL flags / its own count word: the operator keeps the states collected so
    far as they are, continues in a new generation and combines the generations' states at the end like partial
    aggregations.  Page order here: no NULLs anywhere -> NULLs in the second key -> none -> NULLs in an aggregate input and
    the mask -> NULLs in the first key -> none."""
    rng = np.random.default_rng(groups + 17)
    rows = 60000

    def page(null_k1, null_k2, null_x, null_m):
        def nulls(on):
            return (rng.random(rows) < 0.07) if on else None
        g = max(groups, 1)
        return Page([Block.boolean(rng.random(rows) < 0.5, nulls(null_k1)), Block.integer(rng.integers(0, g, rows), nulls(null_k2)),
                     Block.double(rng.random(rows) * 10, nulls(null_x)), Block.bigint(rng.integers(-50, 50, rows), nulls(null_x)),
                     Block.boolean(rng.random(rows) < 0.8, nulls(null_m))], rows)

    pages = [page(0, 0, 0, 0), page(0, 1, 0, 0), page(0, 0, 0, 0), page(0, 0, 1, 1), page(1, 0, 0, 0), page(0, 0, 0, 0)]
    types = [abi.BOOLEAN, abi.INTEGER, abi.DOUBLE, abi.BIGINT, abi.BOOLEAN]
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_COUNT, 2, abi.DOUBLE), (abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_AVG, 3, abi.BIGINT),
            (abi.AGG_SUM, 3, abi.BIGINT), (abi.AGG_MIN, 2, abi.DOUBLE), (abi.AGG_MAX, 3, abi.BIGINT), (abi.AGG_SUM, 3, abi.BIGINT, 4),
            (abi.AGG_COUNT_STAR, -1, None, 4)]
    keys = [0, 1] if groups else []
    ref = oracle.HashAggregation(types, keys, aggs)
    for p in pages:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    make = (lambda **kw: HashAggregationOperator(types, keys, aggs, expected_groups=max(groups, 1), **kw)) if groups else \
        (lambda **kw: AggregationOperator(types, aggs, **kw))
    if step == "single":
        got = [r for p in to_pages(make(), pages) for r in p.to_rows()]
    else:
        from presto_amd.exchange import partial_layout
        ptypes, faggs = partial_layout([abi.BOOLEAN, abi.INTEGER] if groups else [], aggs)
        partial_pages = to_pages(make(step=abi.STEP_PARTIAL), pages[:4]) + to_pages(make(step=abi.STEP_PARTIAL), pages[4:])
        final = HashAggregationOperator(ptypes, [0, 1], faggs, step=abi.STEP_FINAL) if groups else AggregationOperator(ptypes, faggs, step=abi.STEP_FINAL)
        got = [r for p in to_pages(final, partial_pages) for r in p.to_rows()]
    rows_equal_ignore_order(got, expected, rel=1e-9)


def test_group_tables_on_recycled_memory(gpu, oracle):
    """The same aggregation forty times in a row: every operator's group table lives in blocks the previous one gave back to the pool,
    still holding ITS keys.  Keys whose packed halves XOR to the same value -- (day, priority) = (9172, 1) and (9173, 0) -- used to
    collide in all 32 hash bits, and a cached look at an unclaimed slot's leftover key words then sent rows to the colliding key's
    group, about once in sixty runs: the key arrays are cleared when a table is made, and the hash mixes before it folds."""
    rng = np.random.default_rng(606)
    n = 75000
    page = Page([Block.date(rng.integers(9100, 9400, n)), Block.integer(rng.integers(0, 3, n)), Block.double(rng.random(n) * 100),
                 Block.integer(rng.integers(-50, 50, n))], n)
    types = [abi.DATE, abi.INTEGER, abi.DOUBLE, abi.INTEGER]
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 3, abi.INTEGER), (abi.AGG_SUM, 2, abi.DOUBLE)]
    ref = oracle.HashAggregation(types, [0, 1], aggs)
    ref.add_page(page)
    expected = sorted(ref.build_result().to_rows())
    dev = upload_page(page)
    for _ in range(40):
        got = sorted(r for p in to_pages(HashAggregationOperator(types, [0, 1], aggs), [dev]) for r in p.to_rows())
        assert len(got) == len(expected)
        for g, e in zip(got, expected):
            assert g[:4] == e[:4] and abs(g[4] - e[4]) <= 1e-9 * abs(e[4])
