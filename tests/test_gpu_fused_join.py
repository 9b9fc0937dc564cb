"""FilterAndProject -> LookupJoin -> (Hash)Aggregation behind one handle (pa_fused_join_aggregation_create) against the oracle's
composition of the three operators, and against the library's own three separate operators (independent paths: generated
probe kernel vs probe-count / emit / gather kernels + hashed group table).

Keys, counts, integer sums and min / max are bit-exact; DOUBLE sums within 1e-9 (atomic summation order)."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.expr import constant, field
from presto_amd.operators import (Driver, FilterAndProjectOperator, FusedJoinAggregationOperator, HashAggregationOperator, HashBuilderOperator,
                                  LookupJoinOperator, LookupSourceFactory, to_pages)
from presto_amd.page import Block, Page
from tests.util import rows_equal_ignore_order

pytestmark = pytest.mark.gpu

PROBE_TYPES = [abi.BIGINT, abi.DOUBLE, abi.INTEGER, abi.DATE]          # key, amount, quantity, day
BUILD_TYPES = [abi.BIGINT, abi.DATE, abi.INTEGER, abi.DOUBLE, abi.BOOLEAN]  # key, date, priority, weight, flag


def probe_pages(rng, pages, n, key_range, null_keys=True, clustered=False):
    out = []
    for _ in range(pages):
        key = np.sort(rng.integers(0, key_range, n)) if clustered else rng.integers(0, key_range, n)
        out.append(Page([Block.bigint(key, rng.random(n) < 0.05 if null_keys else None), Block.double(rng.random(n) * 100, rng.random(n) < 0.1),
                         Block.integer(rng.integers(-50, 50, n)), Block.date(rng.integers(9000, 9400, n))], n))
    return out


def build_page(rng, keys, nullable=True, null_keys=None):
    n = len(keys)
    nulls = (lambda p: rng.random(n) < p) if nullable else (lambda p: None)
    key_nulls = nulls(0.02) if null_keys is None or null_keys else None
    return Page([Block.bigint(keys, key_nulls), Block.date(rng.integers(8000, 8100, n), nulls(0.2)), Block.integer(rng.integers(0, 3, n)),
                 Block.double(rng.random(n), nulls(0.1)), Block.boolean(rng.random(n) < 0.5)], n)


FILTER = field(3, abi.DATE) > constant(9100, abi.DATE)
PROJECTIONS = [field(0, abi.BIGINT), field(1, abi.DOUBLE) * constant(2.0, abi.DOUBLE), field(2, abi.INTEGER), field(3, abi.DATE)]
PROJECTED = [abi.BIGINT, abi.DOUBLE, abi.INTEGER, abi.DATE]


def oracle_rows(oracle, probe, build, build_out, probe_out, group_by, aggregates, flt=FILTER):
    j = oracle.HashJoin(BUILD_TYPES, [0], build_out)
    for p in build:
        j.add_build_page(p)
    j.build()
    joined_types = [PROJECTED[c] for c in probe_out] + [BUILD_TYPES[c] for c in build_out]
    agg = oracle.HashAggregation(joined_types, group_by, aggregates, expected_groups=1000)
    for p in probe:
        fp = oracle.filter_project(p, flt, PROJECTIONS)
        if fp is None or fp.position_count == 0:
            continue
        joined, _, _ = j.probe(fp, PROJECTED, [0], probe_out)
        if joined.position_count:
            agg.add_page(joined)
    return agg.build_result().to_rows(), joined_types


def fused_rows(probe, build, build_out, probe_out, joined_types, group_by, aggregates, flt=FILTER, expected_groups=1000):
    bridge = LookupSourceFactory()
    builder = HashBuilderOperator(bridge, BUILD_TYPES, [0], build_out)
    op = FusedJoinAggregationOperator(bridge, PROBE_TYPES, flt, PROJECTIONS, [0], probe_out, joined_types, group_by, aggregates,
                                      expected_groups=expected_groups)
    assert not op.needsInput() and op.isBlocked()          # the lookup source future
    Driver(build, [builder]).run()
    assert op.needsInput() and not op.isBlocked()
    return [r for p in to_pages(op, probe) for r in p.to_rows()]


def chain_rows(probe, build, build_out, probe_out, joined_types, group_by, aggregates, flt=FILTER):
    bridge = LookupSourceFactory()
    Driver(build, [HashBuilderOperator(bridge, BUILD_TYPES, [0], build_out)]).run()
    dev = abi.MEM_DEVICE
    out = Driver(probe, [FilterAndProjectOperator(PROBE_TYPES, flt, PROJECTIONS, output_mem=dev),
                         LookupJoinOperator(bridge, PROJECTED, [0], probe_out, output_mem=dev),
                         HashAggregationOperator(joined_types, group_by, aggregates)]).run()
    return [r for p in out for r in p.to_rows()]


AGGS = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 2, abi.INTEGER), (abi.AGG_MIN, 3, abi.DATE),
        (abi.AGG_COUNT, 1, abi.DOUBLE)]


@pytest.mark.parametrize("case", ["build_rows", "hashed", "few_groups", "global"])
@pytest.mark.parametrize("n", [1, 777, 70000])
@pytest.mark.parametrize("lookup", ["table", "rank", "rank, rows in key order"])
def test_fused_probe_matches_oracle(gpu, oracle, case, n, lookup):
    """Unique build keys: the one-kernel execution.  build_rows: group by (probe key, build date, build priority) -- the group is
    the build row; hashed: group by (probe day, build priority); few_groups: group by the build priority alone; global: no keys.
    NULL probe keys and NULL build keys match nothing; NULL build values are group key values.  lookup: NULL build keys leave the
    lookup to the slot table, without them the key rank index answers (build rows shuffled, or arriving in key order)."""
    rng = np.random.default_rng(n + len(case))
    key_range = max(4 * n // 3, 8)
    keys = rng.permutation(key_range)[: max(key_range // 2, 1)]
    if lookup == "rank, rows in key order":
        keys = np.sort(keys)
    build = [build_page(rng, keys, null_keys=(lookup == "table"))]
    probe = probe_pages(rng, 3, n, key_range, clustered=(n == 70000))
    probe_out, build_out = [0, 1, 2, 3], [1, 2, 3]          # joined page: key, amount, quantity, day, date, priority, weight
    group_by = {"build_rows": [0, 4, 5], "hashed": [3, 5], "few_groups": [5], "global": []}[case]
    aggs = AGGS + [(abi.AGG_SUM, 6, abi.DOUBLE)]           # a build column as aggregate input
    expected, joined_types = oracle_rows(oracle, probe, build, build_out, probe_out, group_by, aggs)
    rows = fused_rows(probe, build, build_out, probe_out, joined_types, group_by, aggs)
    rows_equal_ignore_order(rows, expected, rel=1e-9)
    rows_equal_ignore_order(chain_rows(probe, build, build_out, probe_out, joined_types, group_by, aggs), expected, rel=1e-9)


def test_duplicate_build_keys_take_the_operator_chain(gpu, oracle):
    """A key on several build rows: every match is a join output row (DefaultPageJoiner.joinCurrentPosition walks the chain);
    the handle then runs the three operators behind each other."""
    rng = np.random.default_rng(5)
    build = [build_page(rng, rng.integers(0, 300, 500)), build_page(rng, rng.integers(0, 300, 100))]
    probe = probe_pages(rng, 4, 5000, 400)
    expected, joined_types = oracle_rows(oracle, probe, build, [1, 2], [0, 1, 2, 3], [0, 4, 5], AGGS)
    assert sum(r[4] for r in expected) > 5000               # more joined rows than probe rows survive: real duplicates
    rows_equal_ignore_order(fused_rows(probe, build, [1, 2], [0, 1, 2, 3], joined_types, [0, 4, 5], AGGS), expected, rel=1e-9)


def test_no_match_no_rows_and_empty_inputs(gpu, oracle):
    rng = np.random.default_rng(6)
    build = [build_page(rng, np.arange(1000, 1100))]
    probe = probe_pages(rng, 2, 3000, 500)                    # keys 0..499: nothing matches
    jt = PROJECTED + [abi.DATE]
    assert fused_rows(probe, build, [1], [0, 1, 2, 3], jt, [0, 4], AGGS) == []
    rows = fused_rows(probe, build, [1], [0, 1, 2, 3], jt, [], [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 1, abi.DOUBLE)])
    assert rows == [(0, None)]                               # a global aggregation always has its one row
    # no probe page at all, and an empty build side
    assert fused_rows([], build, [1], [0, 1, 2, 3], jt, [0, 4], AGGS) == []
    empty = Page([Block.bigint([]), Block.date([]), Block.integer([]), Block.double([]), Block.boolean([])], 0)
    assert fused_rows(probe, [empty], [1], [0, 1, 2, 3], jt, [0, 4], AGGS) == []


def test_group_keys_from_the_build_side_only_with_the_build_key(gpu, oracle):
    """Group by (build date, build key column as an output channel): the join key reaches the aggregation through the build
    side; and a plan whose group keys leave the join key out is grouped through the hashed table (two build rows may share
    the group)."""
    rng = np.random.default_rng(8)
    build = [build_page(rng, rng.permutation(4000)[:2500], nullable=False)]
    probe = probe_pages(rng, 2, 20000, 4000, null_keys=False)
    probe_out, build_out = [1, 2], [0, 1, 2]                  # joined: amount, quantity, build key, date, priority
    aggs = [(abi.AGG_SUM, 0, abi.DOUBLE), (abi.AGG_MAX, 1, abi.INTEGER), (abi.AGG_COUNT_STAR, -1, None)]
    for group_by in ([3, 2], [3, 4]):
        expected, jt = oracle_rows(oracle, probe, build, build_out, probe_out, group_by, aggs)
        rows_equal_ignore_order(fused_rows(probe, build, build_out, probe_out, jt, group_by, aggs), expected, rel=1e-9)


def test_many_groups_over_many_pages(gpu, oracle):
    """2 x 10^5 build rows, 40 pages: the build-row table and the hashed table (PRESTO_AMD_NO_BROW) agree with the oracle."""
    import os
    rng = np.random.default_rng(9)
    nb = 200000
    build = [build_page(rng, rng.permutation(2 * nb)[:nb])]
    probe = probe_pages(rng, 40, 25000, 2 * nb, clustered=True)
    group_by, aggs = [0, 4, 5], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    expected, jt = oracle_rows(oracle, probe, build, [1, 2], [0, 1, 2, 3], group_by, aggs)
    assert len(expected) > 100000
    rows_equal_ignore_order(fused_rows(probe, build, [1, 2], [0, 1, 2, 3], jt, group_by, aggs, expected_groups=nb), expected, rel=1e-9)
    os.environ["PRESTO_AMD_BROW_TAGS"] = "1"   # the build-row table with stored tags (what plans without an always-updated word get)
    try:
        rows_equal_ignore_order(fused_rows(probe, build, [1, 2], [0, 1, 2, 3], jt, group_by, aggs, expected_groups=nb), expected, rel=1e-9)
    finally:
        del os.environ["PRESTO_AMD_BROW_TAGS"]
    os.environ["PRESTO_AMD_NO_BROW"] = "1"
    try:
        rows_equal_ignore_order(fused_rows(probe, build, [1, 2], [0, 1, 2, 3], jt, group_by, aggs, expected_groups=2000000), expected, rel=1e-9)
        rows_equal_ignore_order(fused_rows(probe, build, [1, 2], [0, 1, 2, 3], jt, group_by, aggs, expected_groups=100), expected, rel=1e-9)
    finally:
        del os.environ["PRESTO_AMD_NO_BROW"]


def test_group_existence_without_an_always_updated_word(gpu, oracle):
    """sum over a nullable probe column, min over a nullable build column, no count: no accumulator word is touched by every row of
    a group, so the build-row table stores tags; groups whose every input is NULL still come out (sum NULL)."""
    rng = np.random.default_rng(12)
    build = [build_page(rng, rng.permutation(3000)[:2000])]
    probe = probe_pages(rng, 3, 9000, 3000)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_MIN, 4, abi.DATE)]
    expected, jt = oracle_rows(oracle, probe, build, [1, 2], [0, 1, 2, 3], [0, 5], aggs)
    assert any(r[2] is None for r in expected)
    rows_equal_ignore_order(fused_rows(probe, build, [1, 2], [0, 1, 2, 3], jt, [0, 5], aggs), expected, rel=1e-9)


def test_sum_of_negative_zeros_is_positive_zero(gpu, oracle):
    """The build-row table reads "has a group" off the DOUBLE sum word (it starts at -0.0 and takes x + 0.0): a group whose inputs are
    all -0.0 must still exist, with the reference's sum +0.0 (DoubleSumAggregation starts from 0.0)."""
    n = 64
    key = np.arange(n, dtype=np.int64) // 2
    amount = np.where(key % 3 == 0, -0.0, 1.5)
    probe = [Page([Block.bigint(key), Block.double(amount), Block.integer(np.zeros(n, dtype=np.int32)), Block.date(np.full(n, 9200, dtype=np.int32))], n)]
    rng = np.random.default_rng(1)
    build = [build_page(rng, np.arange(40), nullable=False)]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE)]
    expected, jt = oracle_rows(oracle, probe, build, [1], [0, 1, 2, 3], [0, 4], aggs)
    rows = fused_rows(probe, build, [1], [0, 1, 2, 3], jt, [0, 4], aggs)
    rows_equal_ignore_order(rows, expected, rel=0.0)
    zeros = [r for r in rows if r[0] % 3 == 0]
    assert len(zeros) == 11 and all(r[2] == 0.0 and np.signbit(r[2]) == False for r in zeros)


def test_fused_probe_of_long_min_value(gpu, oracle):
    """Long.MIN_VALUE as a join key (10 % of the probe rows): no key value may double as an "empty slot" mark."""
    rng = np.random.default_rng(14)
    low = -2 ** 63
    bkeys = np.concatenate([np.arange(100, 1100), [low]]).astype(np.int64)
    build = [build_page(rng, bkeys[rng.permutation(len(bkeys))], nullable=False)]
    n = 4000
    key = np.where(rng.random(n) < 0.1, low, rng.integers(0, 1500, n)).astype(np.int64)
    probe = [Page([Block.bigint(key), Block.double(rng.random(n)), Block.integer(rng.integers(0, 9, n)), Block.date(np.full(n, 9300, dtype=np.int32))], n)]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    expected, jt = oracle_rows(oracle, probe, build, [1, 2], [0, 1, 2, 3], [0, 4, 5], aggs)
    assert any(r[0] == low and r[4] > 300 for r in expected)
    rows_equal_ignore_order(fused_rows(probe, build, [1, 2], [0, 1, 2, 3], jt, [0, 4, 5], aggs), expected, rel=1e-9)


def test_refused_shapes(gpu):
    from presto_amd._lib import PrestoAmdError
    bridge = LookupSourceFactory()
    builder = HashBuilderOperator(bridge, BUILD_TYPES, [0], [1])
    with pytest.raises(PrestoAmdError):   # the aggregation's input page must be the join's output page
        FusedJoinAggregationOperator(bridge, PROBE_TYPES, FILTER, PROJECTIONS, [0], [0, 1], [abi.BIGINT, abi.DOUBLE], [0], [(abi.AGG_COUNT_STAR, -1, None)])
    with pytest.raises(PrestoAmdError):   # probe / build key types differ
        FusedJoinAggregationOperator(bridge, PROBE_TYPES, FILTER, PROJECTIONS, [2], [0, 1], [abi.BIGINT, abi.DOUBLE, abi.DATE], [0], [(abi.AGG_COUNT_STAR, -1, None)])
    builder.close()


# ---- FilterAndProject -> LookupJoin behind one handle (pa_fused_join_create) ------------------------------------------------
def joined_rows_oracle(oracle, probe, build, build_out, probe_out, flt=FILTER):
    j = oracle.HashJoin(BUILD_TYPES, [0], build_out)
    for p in build:
        j.add_build_page(p)
    j.build()
    rows = []
    for p in probe:
        fp = oracle.filter_project(p, flt, PROJECTIONS)
        if fp is None or fp.position_count == 0:
            continue
        rows += j.probe(fp, PROJECTED, [0], probe_out)[0].to_rows()
    return rows


def joined_rows_device(probe, build, build_out, probe_out, flt=FILTER, output_mem=abi.MEM_HOST):
    from presto_amd.operators import FusedJoinOperator, download_page
    bridge = LookupSourceFactory()
    builder = HashBuilderOperator(bridge, BUILD_TYPES, [0], build_out)
    op = FusedJoinOperator(bridge, PROBE_TYPES, flt, PROJECTIONS, [0], probe_out, output_mem=output_mem)
    assert not op.needsInput() and op.isBlocked()
    Driver(build, [builder]).run()
    assert op.needsInput() and not op.isBlocked()
    if output_mem != abi.MEM_DEVICE:
        return [r for p in to_pages(op, probe) for r in p.to_rows()]
    rows = []   # device output buffers belong to the operator until its next call: read every page at once
    for p in probe:
        op.addInput(p)
        out = op.getOutput()
        if out is not None:
            rows += download_page(out).to_rows()
    op.finish()
    assert op.getOutput() is None and op.isFinished()
    return rows


@pytest.mark.parametrize("n", [1, 1000, 70000])
@pytest.mark.parametrize("duplicates,null_keys", [(False, True), (True, True), (False, False)])
def test_fused_join_rows_and_order(gpu, oracle, n, duplicates, null_keys):
    """FilterAndProject -> LookupJoin: the same rows in the same (probe) order as the oracle's two operators -- through the
    one-pass form (unique build keys) and through the operator chain (duplicate keys: matches of a row in chain order)."""
    rng = np.random.default_rng(n + duplicates)
    key_range = max(4 * n // 3, 8)
    keys = rng.integers(0, key_range, max(key_range // 2, 1)) if duplicates else rng.permutation(key_range)[: max(key_range // 2, 1)]
    build = [build_page(rng, keys, null_keys=null_keys)]   # (no NULL and no duplicate key: the key rank index instead of the table)
    probe = probe_pages(rng, 3, n, key_range, clustered=(n == 70000))
    for probe_out, build_out in (([0, 1, 2, 3], [1, 2, 3, 4]), ([3, 1], [2]), ([0], [])):
        expected = joined_rows_oracle(oracle, probe, build, build_out, probe_out)
        assert joined_rows_device(probe, build, build_out, probe_out) == expected
    assert joined_rows_device(probe, build, [1, 3], [0, 2], output_mem=abi.MEM_DEVICE) == joined_rows_oracle(oracle, probe, build, [1, 3], [0, 2])


def test_fused_join_every_row_and_no_row(gpu, oracle):
    """Every probe row selected (identity projections of a fully selected page are views of the input; the build columns are still
    computed), no filter at all, and a probe side without a single match."""
    rng = np.random.default_rng(9)
    n = 5000
    build = [build_page(rng, np.arange(n), nullable=False)]
    probe = [Page([Block.bigint(rng.permutation(n)), Block.double(rng.random(n)), Block.integer(rng.integers(0, 9, n)), Block.date(np.full(n, 9300, dtype=np.int32))], n)]
    idp = [field(0, abi.BIGINT), field(1, abi.DOUBLE), field(2, abi.INTEGER), field(3, abi.DATE)]
    from presto_amd.operators import FusedJoinOperator
    for flt in (None, FILTER):
        bridge = LookupSourceFactory()
        Driver(build, [HashBuilderOperator(bridge, BUILD_TYPES, [0], [1, 2])]).run()
        rows = [r for p in to_pages(FusedJoinOperator(bridge, PROBE_TYPES, flt, idp, [0], [0, 1, 2, 3]), probe) for r in p.to_rows()]
        j = oracle.HashJoin(BUILD_TYPES, [0], [1, 2])
        j.add_build_page(build[0])
        j.build()
        assert rows == j.probe(probe[0], PROBE_TYPES, [0], [0, 1, 2, 3])[0].to_rows() and len(rows) == n
    none = [Page([Block.bigint(np.arange(n) + 10 * n), Block.double(rng.random(n)), Block.integer(np.zeros(n, dtype=np.int32)), Block.date(np.full(n, 9300, dtype=np.int32))], n)]
    assert joined_rows_device(none, build, [1], [0, 1]) == []


def test_fused_join_with_varchar_probe_channels(gpu, oracle):
    from presto_amd.operators import FusedJoinOperator
    rng = np.random.default_rng(10)
    n = 20000
    words = [b"", b"a", b"BUILDING", b"0123456789abcdefghij", None]
    types = [abi.BIGINT, abi.VARCHAR, abi.DOUBLE]
    probe = [Page([Block.bigint(rng.integers(0, 3000, n), rng.random(n) < 0.03), Block.varchar([words[i] for i in rng.integers(0, len(words), n)]), Block.double(rng.random(n))], n)
             for _ in range(2)]
    build = [build_page(rng, rng.permutation(3000)[:1500])]
    proj = [field(0, abi.BIGINT), field(1, abi.VARCHAR), field(2, abi.DOUBLE) * constant(3.0, abi.DOUBLE)]
    flt = field(2, abi.DOUBLE) < constant(0.8, abi.DOUBLE)
    bridge = LookupSourceFactory()
    Driver(build, [HashBuilderOperator(bridge, BUILD_TYPES, [0], [1, 4])]).run()
    rows = [r for p in to_pages(FusedJoinOperator(bridge, types, flt, proj, [0], [1, 2, 0]), probe) for r in p.to_rows()]
    j = oracle.HashJoin(BUILD_TYPES, [0], [1, 4])
    j.add_build_page(build[0])
    j.build()
    expected = []
    for p in probe:
        fp = oracle.filter_project(p, flt, proj)
        expected += j.probe(fp, types, [0], [1, 2, 0])[0].to_rows()
    assert rows == expected and len(rows) > 5000


def test_operator_chain_on_an_owned_stream_with_a_large_duplicate_build_side(gpu, oracle):
    """No stream in any descriptor (the JNI default) and a build side with duplicate keys: the handle runs FilterAndProject ->
    LookupJoin -> HashAggregation behind each other, device pages handed over inside the handle -- on ONE stream the handle owns
    (members with pooled streams of their own would race: nothing orders two private streams).  Large pages, so that a page's
    gather kernels are still running when the next operator takes it."""
    rng = np.random.default_rng(77)
    nb = 200_000
    build = [build_page(rng, rng.integers(0, 150_000, nb), nullable=False)]
    probe = probe_pages(rng, 6, 300_000, 160_000, null_keys=False)
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 2, abi.INTEGER)]
    expected, joined_types = oracle_rows(oracle, probe, build, [1, 2], [0, 1, 2, 3], [0], aggs)
    assert sum(r[2] for r in expected) > 1_000_000
    for _ in range(3):
        rows = fused_rows(probe, build, [1, 2], [0, 1, 2, 3], joined_types, [0], aggs, expected_groups=200_000)
        rows_equal_ignore_order(rows, expected, rel=1e-9)


@pytest.mark.parametrize("n", [3, 255, 1021, 262147])
@pytest.mark.parametrize("clustered", [True, False])
def test_pipelined_probe_loop_equals_the_plain_loop(gpu, oracle, n, clustered, monkeypatch):
    """The build-row tier's page loop as a software pipeline (four quads in flight per lane, the default over a key rank index) against
    the plain loop (PRESTO_AMD_BROW_PIPE=0) and the loop that only loads the next quad's columns ahead (=1), on pages whose row counts
    leave partial quads, partial waves and ranges shorter than the pipeline is deep -- and against the oracle."""
    rng = np.random.default_rng(n)
    key_range = max(2 * n, 64)
    keys = np.sort(rng.permutation(key_range)[: key_range // 2])
    build = [build_page(rng, keys, null_keys=False)]
    probe = probe_pages(rng, 3, n, key_range, clustered=clustered)
    group_by, aggs = [0, 4, 5], AGGS
    expected, jt = oracle_rows(oracle, probe, build, [1, 2], [0, 1, 2, 3], group_by, aggs)
    for level in (None, "0", "1"):
        if level is not None:
            monkeypatch.setenv("PRESTO_AMD_BROW_PIPE", level)
        rows = fused_rows(probe, build, [1, 2], [0, 1, 2, 3], jt, group_by, aggs, expected_groups=len(keys))
        rows_equal_ignore_order(rows, expected, rel=1e-9)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("PA_FUZZ_SEEDS", "16")))))
def test_random_fused_probes_over_a_key_rank_index(gpu, oracle, seed):
    """Seeded shapes of the one-kernel execution over a lookup source with a key rank index (the pipelined page loop of the build-row
    tier, and the other tiers' quad loops): page sizes that leave partial quads and waves, clustered or not, several pages, group keys
    that make the group the build row or not, aggregates with lazy (probe-side) and build-side inputs."""
    rng = np.random.default_rng(9100 + seed)
    n = int(rng.choice([1, 2, 5, 63, 64, 257, 1023, 4099, 30011, 131075]))
    pages = int(rng.integers(1, 4))
    clustered = bool(rng.integers(0, 2))
    key_range = max(int(n * rng.choice([0.5, 1.5, 4.0])), 8)
    keys = rng.permutation(key_range)[: max(key_range // 2, 1)]
    if rng.integers(0, 2):
        keys = np.sort(keys)
    build = [build_page(rng, keys, null_keys=False)]
    probe = probe_pages(rng, pages, n, key_range, null_keys=bool(rng.integers(0, 2)), clustered=clustered)
    group_by = [[0, 4, 5], [3, 5], [5], []][int(rng.integers(0, 4))]
    aggs = AGGS + ([(abi.AGG_SUM, 6, abi.DOUBLE)] if rng.integers(0, 2) else [])
    probe_out, build_out = [0, 1, 2, 3], [1, 2, 3]
    expected, jt = oracle_rows(oracle, probe, build, build_out, probe_out, group_by, aggs)
    rows = fused_rows(probe, build, build_out, probe_out, jt, group_by, aggs, expected_groups=max(len(keys), 16))
    rows_equal_ignore_order(rows, expected, rel=1e-9)
