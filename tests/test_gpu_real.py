"""REAL (RealType: IEEE single values in an IntArrayBlock) on the device path: expressions, aggregates and payload channels, against
the oracle's restatement of RealOperators / RealSumAggregation / RealAverageAggregation.  Values and arithmetic are bit-exact
(Java float arithmetic = IEEE single arithmetic); sums run in DOUBLE in both (tolerance 1e-9 on the DOUBLE sum, i.e. at most the
last bit of the REAL result)."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.exchange import partial_layout
from presto_amd.expr import and_, constant, field
from presto_amd.operators import (FilterAndProjectOperator, FusedAggregationOperator, HashAggregationOperator, HashBuilderOperator, LookupJoinOperator,
                                  LookupSourceFactory, TopNOperator, download_page, to_pages)
from presto_amd.page import Block, Page, deserialize_page, serialize_page

pytestmark = pytest.mark.gpu

F32 = np.float32


def real_page(rng, n, groups=5):
    v = (rng.standard_normal(n) * 100).astype(F32)
    v[rng.random(n) < 0.01] = F32(0.0)
    v[rng.random(n) < 0.01] = F32(-0.0)
    w = (rng.random(n) * 10 + 0.5).astype(F32)
    return Page([Block.real(v, rng.random(n) < 0.05), Block.real(w), Block.bigint(rng.integers(0, groups, n)), Block.double(rng.standard_normal(n)),
                 Block.integer(rng.integers(-1000, 1000, n))], n)


TYPES = [abi.REAL, abi.REAL, abi.BIGINT, abi.DOUBLE, abi.INTEGER]


def same_rows(got, expected, rel=0.0):
    assert len(got) == len(expected)
    for g, e in zip(got, expected):
        assert len(g) == len(e)
        for x, y in zip(g, e):
            if isinstance(y, float) and isinstance(x, float):
                assert (np.isnan(x) and np.isnan(y)) or x == y or abs(x - y) <= rel * max(abs(x), abs(y)), (g, e)
            else:
                assert x == y, (g, e)


def test_real_expressions_bit_exact(gpu, oracle):
    """RealOperators: + - * / % and negation in float arithmetic, comparisons, BETWEEN / IN, casts REAL <-> DOUBLE and from the
    integer types, REAL constants, three-valued logic over NULL REALs."""
    rng = np.random.default_rng(1)
    a, b, d, i = field(0, abi.REAL), field(1, abi.REAL), field(3, abi.DOUBLE), field(4, abi.INTEGER)
    c = lambda x: constant(x, abi.REAL)
    flt = and_(a > c(-150.0), (a * b).between(c(-900.5), c(900.25)), b.ne(c(3.0)))
    proj = [a, a + b, a - b, a * b, a / b, a % b, -a, a.cast(abi.DOUBLE) * d, d.cast(abi.REAL), i.cast(abi.REAL) + c(0.1), field(2, abi.BIGINT).cast(abi.REAL),
            a.isin(c(0.0), c(1.5)), (a + b) * c(0.3333333) - b / c(7.0)]
    pages = [real_page(rng, n) for n in (1, 1000, 70001)]
    op = FilterAndProjectOperator(TYPES, flt, proj)
    for p in pages:
        expected = oracle.filter_project(p, flt, proj)
        got = to_pages(op, [p]) if False else None
    op = FilterAndProjectOperator(TYPES, flt, proj)
    out = to_pages(op, pages)
    got = [r for p in out for r in p.to_rows()]
    expected = [r for p in pages for e in [oracle.filter_project(p, flt, proj)] if e is not None for r in e.to_rows()]
    assert [b.type for b in out[-1].blocks][:7] == [abi.REAL] * 7 and len(got) > 30000
    same_rows(got, expected)          # bit-exact, -0.0 vs 0.0 aside (== treats them alike; the bits are checked below)
    gb = np.concatenate([p.blocks[3].values[:p.position_count].view(np.uint32) for p in out])
    eb = np.concatenate([e.blocks[3].values[:e.position_count].view(np.uint32) for p in pages for e in [oracle.filter_project(p, flt, proj)] if e is not None])
    assert np.array_equal(gb, eb)     # a * b: the raw float bits


@pytest.mark.parametrize("groups", [0, 5, 3000, 200000])
def test_real_aggregates(gpu, oracle, groups):
    """sum / avg / min / max / count over REAL: DOUBLE accumulation of the widened values, REAL results (RealSumAggregation.java:36-67,
    RealAverageAggregation.java:132-158); global, few groups, and the table tiers."""
    rng = np.random.default_rng(groups + 2)
    pages = [real_page(rng, 120000, max(groups, 1)) for _ in range(2)]
    aggs = [(abi.AGG_SUM, 0, abi.REAL), (abi.AGG_AVG, 0, abi.REAL), (abi.AGG_MIN, 0, abi.REAL), (abi.AGG_MAX, 1, abi.REAL), (abi.AGG_COUNT, 0, abi.REAL),
            (abi.AGG_SUM, 3, abi.DOUBLE)]
    keys = [2] if groups else []
    op = HashAggregationOperator(TYPES, keys, aggs, expected_groups=max(groups, 1))
    got = sorted(r for p in to_pages(op, pages) for r in p.to_rows())
    ref = oracle.HashAggregation(TYPES, keys, aggs)
    for p in pages:
        ref.add_page(p)
    expected = sorted(ref.build_result().to_rows())
    same_rows(got, expected, rel=2e-7)   # one ulp of a float: the DOUBLE sums differ in their last bits by summation order
    out_types = [b.type for b in to_pages(HashAggregationOperator(TYPES, keys, aggs), pages[:1])[0].blocks]
    assert out_types[len(keys):] == [abi.REAL, abi.REAL, abi.REAL, abi.REAL, abi.BIGINT, abi.DOUBLE]


def test_real_partial_final(gpu, oracle):
    rng = np.random.default_rng(4)
    pages = [real_page(rng, 50000, 40) for _ in range(3)]
    aggs = [(abi.AGG_SUM, 0, abi.REAL), (abi.AGG_AVG, 1, abi.REAL), (abi.AGG_MAX, 0, abi.REAL)]
    ref = oracle.HashAggregation(TYPES, [2], aggs)
    for p in pages:
        ref.add_page(p)
    expected = sorted(ref.build_result().to_rows())
    ptypes, faggs = partial_layout([abi.BIGINT], aggs)
    assert ptypes == [abi.BIGINT, abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.REAL]
    partial = [o for p in pages for o in to_pages(HashAggregationOperator(TYPES, [2], aggs, step=abi.STEP_PARTIAL), [p])]
    final = sorted(r for p in to_pages(HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL), partial) for r in p.to_rows())
    same_rows(final, expected, rel=2e-7)


def test_real_fused_filter_and_payload_channels(gpu, oracle):
    """REAL through the fused filter -> aggregate kernel, as a join / TopN payload and through the wire format."""
    rng = np.random.default_rng(6)
    pages = [real_page(rng, 30000, 7) for _ in range(2)]
    flt = field(1, abi.REAL) > constant(3.0, abi.REAL)
    proj = [field(2, abi.BIGINT), field(0, abi.REAL) * field(1, abi.REAL)]
    aggs = [(abi.AGG_SUM, 1, abi.REAL), (abi.AGG_COUNT_STAR, -1, None)]
    got = sorted(r for p in to_pages(FusedAggregationOperator(TYPES, flt, proj, [0], aggs), pages) for r in p.to_rows())
    ref = oracle.HashAggregation([abi.BIGINT, abi.REAL], [0], aggs)
    for p in pages:
        fp = oracle.filter_project(p, flt, proj)
        ref.add_page(fp)
    same_rows(got, sorted(ref.build_result().to_rows()), rel=2e-7)
    # join payload
    build = Page([Block.bigint(np.arange(100)), Block.real(np.arange(100, dtype=F32) / F32(3))], 100)
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, [abi.BIGINT, abi.REAL], [0], [1]), [build])
    rows = [r for p in to_pages(LookupJoinOperator(bridge, TYPES, [2], [0, 2]), pages[:1]) for r in p.to_rows()]
    j = oracle.HashJoin([abi.BIGINT, abi.REAL], [0], [1])
    j.add_build_page(build)
    j.build()
    assert rows == j.probe(pages[0], TYPES, [2], [0, 2])[0].to_rows() and len(rows) == 30000
    # TopN payload (sort key BIGINT / DOUBLE)
    top = [r for p in to_pages(TopNOperator(TYPES, 10, [3], [abi.DESC_NULLS_LAST]), pages) for r in p.to_rows()]
    assert top == oracle.topn(pages, 10, [3], [abi.DESC_NULLS_LAST])
    # wire format: INT_ARRAY, typed on the way back
    frame = serialize_page(pages[0])
    assert frame == oracle.serialize_page(pages[0])
    back = download_page(deserialize_page(frame, types=TYPES))
    assert [b.type for b in back.blocks] == TYPES
    same_rows(back.to_rows(), pages[0].to_rows())


def test_real_keys_are_refused(gpu):
    from presto_amd._lib import PrestoAmdError
    from presto_amd.operators import OrderByOperator
    cases = [lambda: HashAggregationOperator([abi.REAL], [0], [(abi.AGG_COUNT_STAR, -1, None)]),
             lambda: HashBuilderOperator(LookupSourceFactory(), [abi.REAL], [0], []),
             lambda: TopNOperator([abi.REAL], 5, [0], [abi.ASC_NULLS_LAST]),
             lambda: OrderByOperator([abi.REAL], [0], [0], [abi.ASC_NULLS_LAST])]
    for make in cases:
        with pytest.raises(PrestoAmdError) as e:
            op = make()
            to_pages(op, [Page([Block.real([1.0, 2.0])], 2)])
        assert e.value.status == abi.ERR_NOT_SUPPORTED
