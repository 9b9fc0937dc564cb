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


def key_pool(rng, distinct):
    return np.concatenate([(rng.standard_normal(distinct) * 50).astype(F32), np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1e-42, -1e-42], dtype=F32)])


def key_page(rng, n, distinct, nulls=True, pool=None):
    """a REAL key channel with both zeros, NaN, +-Inf, denormals and NULLs among `distinct` ordinary values, and two payload channels"""
    if pool is None:
        pool = key_pool(rng, distinct)
    k = pool[rng.integers(0, len(pool), n)]
    return Page([Block.real(k, rng.random(n) < 0.04 if nulls else None), Block.bigint(rng.integers(-50, 50, n)), Block.double(rng.random(n))], n)


KEY_TYPES = [abi.REAL, abi.BIGINT, abi.DOUBLE]


def norm(rows):  # NaN != NaN, and the key of the zero group may come out with either sign
    return sorted((tuple("nan" if isinstance(v, float) and v != v else (0.0 if isinstance(v, float) and v == 0 else v) for v in r) for r in rows), key=repr)


@pytest.mark.parametrize("distinct,n", [(3, 2000), (500, 60000), (40000, 300000)])
def test_group_by_a_real_key(gpu, oracle, distinct, n):
    """REAL group keys: -0 and +0 are one group, NaN is one group (IS NOT DISTINCT FROM, RealType.java:127-140), NULL is a group;
    every tier (few groups in registers / LDS, many in the HBM table).  Sums of integers and counts exact, the DOUBLE sum to 1e-9."""
    rng = np.random.default_rng(distinct)
    pool = key_pool(rng, distinct)
    pages = [key_page(rng, n, distinct, pool=pool), key_page(rng, n // 3 + 1, distinct, pool=pool)]
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_MIN, 1, abi.BIGINT)]
    ref = oracle.HashAggregation(KEY_TYPES, [0], aggs)
    for p in pages:
        ref.add_page(p)
    expected = norm(ref.build_result().to_rows())
    got = norm([r for p in to_pages(HashAggregationOperator(KEY_TYPES, [0], aggs, expected_groups=distinct + 8), pages) for r in p.to_rows()])
    assert len(got) == len(expected)
    for g, e in zip(got, expected):
        assert g[:3] == e[:3] and g[4] == e[4] and abs(g[3] - e[3]) <= 1e-9 * max(abs(e[3]), 1.0)
    # the two-column key (REAL, BIGINT) through the fused operator
    fused = FusedAggregationOperator(KEY_TYPES, None, [field(0, abi.REAL), field(1, abi.BIGINT), field(2, abi.DOUBLE)], [0, 1], [(abi.AGG_COUNT_STAR, -1, None)])
    ref2 = oracle.HashAggregation(KEY_TYPES, [0, 1], [(abi.AGG_COUNT_STAR, -1, None)])
    for p in pages:
        ref2.add_page(p)
    assert norm([r for p in to_pages(fused, pages) for r in p.to_rows()]) == norm(ref2.build_result().to_rows())


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_PROBE_OUTER, abi.JOIN_FULL_OUTER])
def test_join_on_a_real_key(gpu, oracle, join_type):
    """REAL join keys: RealType.equalOperator is ==, so NaN matches nothing and -0 matches +0 (their hash is the same);
    rows, order and chains as the oracle's PagesHash."""
    rng = np.random.default_rng(41 + join_type)
    pool = key_pool(rng, 400)   # (both sides draw from the same values: chains of ~7 build rows per key)
    build, probe = [key_page(rng, 3000, 400, pool=pool)], [key_page(rng, 5000, 400, pool=pool), key_page(rng, 1, 400, pool=pool)]
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, KEY_TYPES, [0], [0, 1]), build)
    join = LookupJoinOperator(bridge, KEY_TYPES, [0], [0, 2], join_type=join_type)
    got = [r for p in to_pages(join, probe) for r in p.to_rows()]
    ref = oracle.HashJoin(KEY_TYPES, [0], [0, 1])
    ref.add_build_page(build[0])
    ref.build()
    expected = [r for p in probe for r in ref.probe(p, KEY_TYPES, [0], [0, 2], join_type=join_type)[0].to_rows()]

    def nn(rows):
        return [tuple("nan" if isinstance(v, float) and v != v else v for v in r) for r in rows]
    assert nn(got) == nn(expected) and len(got) > 20000


def test_sort_by_a_real_key(gpu, oracle):
    """TopN and OrderBy by a REAL channel: Float.compare order (-Inf < .. < -0.0 < 0.0 < .. < Inf < NaN), NULL placement, both
    directions, a second key breaking ties."""
    from presto_amd.operators import OrderByOperator
    rng = np.random.default_rng(77)
    pages = [key_page(rng, 20000, 300), key_page(rng, 777, 300)]
    for order in (abi.ASC_NULLS_LAST, abi.DESC_NULLS_FIRST, abi.ASC_NULLS_FIRST, abi.DESC_NULLS_LAST):
        expected = oracle.topn(pages, 200, [0, 1], [order, abi.ASC_NULLS_LAST])
        got = [r for p in to_pages(TopNOperator(KEY_TYPES, 200, [0, 1], [order, abi.ASC_NULLS_LAST]), pages) for r in p.to_rows()]

        def keys(rows):   # rows tied on both keys may come in either order
            return [tuple("nan" if isinstance(v, float) and v != v else (repr(v) if isinstance(v, float) else v) for v in r[:2]) for r in rows]
        assert keys(got) == keys(expected)
        ordered = [r for p in to_pages(OrderByOperator(KEY_TYPES, [0, 1], [0, 1], [order, abi.ASC_NULLS_LAST]), pages) for r in p.to_rows()]
        assert keys(ordered) == keys(oracle.order_by(pages, [0, 1], [0, 1], [order, abi.ASC_NULLS_LAST]))


def test_hash_of_a_real_channel(gpu, oracle):
    """$hashvalue over a REAL channel: AbstractLongType.hash(floatToIntBits(v)) with +0 for both zeros and one NaN -- what routes a row
    through a partitioned exchange."""
    import ctypes as C

    from presto_amd._lib import DeviceAllocation, check, lib
    from presto_amd.operators import DeviceBuffer, download, upload_page

    def hash_page(page, channels):
        cpage, keep = upload_page(page).to_c()
        buf = DeviceAllocation(8 * page.position_count)
        check(lib().pa_hash_page(C.byref(cpage), len(channels), abi.int32_array(channels), buf.ptr, None))
        check(lib().pa_stream_synchronize(None))
        return download(DeviceBuffer(buf.ptr, 8 * page.position_count), np.int64, page.position_count)

    rng = np.random.default_rng(5)
    page = key_page(rng, 5000, 100)
    assert np.array_equal(hash_page(page, [0]), oracle.hash_page(page, [0]))
    assert np.array_equal(hash_page(page, [1, 0]), oracle.hash_page(page, [1, 0]))
    zeros = Page([Block.real(np.array([0.0, -0.0], dtype=F32))], 2)
    h = hash_page(zeros, [0])
    assert h[0] == h[1]


def test_real_dynamic_filter_channels(gpu, oracle):
    """REAL build-side channels of a DynamicFilterSource (TestDynamicFilterSourceOperator.java:296-312, 327-338): values without
    NaN, one element for the two zeros, no min / max collection (DynamicFilterSourceOperator.java:187-188)."""
    from presto_amd.operators import DynamicFilterSourceOperator
    nan = float("nan")
    pages = [Page([Block.real([42.0, nan, -0.0, 0.0, 1.5, -3.25, 1.5])], 7), Page([Block.real([7.0, nan], [False, True])], 2)]
    ref = oracle.DynamicFilterSource([abi.REAL], [0], 100, 1 << 20, 100)
    for p in pages:
        ref.add_page(p)
    ref.finish()
    op = DynamicFilterSourceOperator([abi.REAL], [0], 100, 1 << 20, 100)
    for p in pages:
        op.addInput(p)
        op.getOutput()
    op.finish()
    got = op.predicate()
    op.close()
    assert got == ref.predicate == [("values", [-3.25, 0.0, 1.5, 7.0, 42.0])]
    # more distinct values than the limit and no orderable channel: TupleDomain.all()
    op = DynamicFilterSourceOperator([abi.REAL], [0], 100, 1 << 20, 1000000)
    op.addInput(Page([Block.real(np.arange(101, dtype=np.float32))], 101))
    op.getOutput()
    op.finish()
    assert op.predicate() == "all"
    op.close()
