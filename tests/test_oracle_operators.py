"""The reference's own behavioural known-answer tests restated against the oracle.  These are
what pins the CPU restatement: inputs are SequencePageBuilder pages, expected values are copied from the cited
reference tests."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.expr import and_, constant, field
from presto_amd.page import Block, Page, sequence_page


def test_fp1_filter_and_project(oracle):
    """TestFilterAndProjectOperator.test (…/operator/TestFilterAndProjectOperator.java:78-124), the reference's literal
    expressions: filter LESS_THAN_OR_EQUAL(field 1, 9), projections field 0 and ADD(field 1, 5)"""
    page = sequence_page(100, [(abi.VARCHAR, 0), (abi.BIGINT, 0)])
    out = oracle.filter_project(page, field(1, abi.BIGINT) <= 9, [field(0, abi.VARCHAR), field(1, abi.BIGINT) + 5])
    assert out.to_rows() == [(str(i).encode(), i + 5) for i in range(10)]


def test_fp2_merge_output(oracle):
    """TestFilterAndProjectOperator.testMergeOutput (…/TestFilterAndProjectOperator.java:126-161): 4 x the same page,
    filter c1 = 10, project c1, minOutputPageSize 64 kB / minOutputPageRowCount 2 -> ONE page [10,10,10,10]"""
    page = sequence_page(100, [(abi.VARCHAR, 0), (abi.BIGINT, 0)])
    merge = oracle.MergePages(64 * 1024, 2)
    out = []
    for _ in range(4):
        out += merge.process(oracle.filter_project(page, field(1, abi.BIGINT).eq(10), [field(1, abi.BIGINT)]))
    out += merge.finish()
    assert [p.to_rows() for p in out] == [[(10,)] * 4]


def test_page_processor_selection_shapes(oracle):
    """TestPageProcessor partial / all / none / projection-less (…/operator/project/TestPageProcessor.java:90-200)"""
    page = sequence_page(100, [(abi.BIGINT, 0)])
    c0 = field(0, abi.BIGINT)
    rng = and_(c0 >= 25, c0 < 75)
    assert oracle.filter_positions(page, rng)[0] is True
    assert oracle.filter_positions(page, rng)[1].tolist() == list(range(25, 75))
    assert [r[0] for r in oracle.filter_project(page, rng, [c0]).to_rows()] == list(range(25, 75))
    assert oracle.filter_positions(page, c0 >= 0) == (False, 100)      # positionsRange(0, 100)
    assert oracle.filter_positions(page, c0 < 0) == (False, 0)          # positionsRange(0, 0)
    assert oracle.filter_project(page, c0 < 0, [c0]) is None             # no page
    out = oracle.filter_project(page, rng, [])
    assert out.channel_count == 0 and out.position_count == 50
    empty = Page([Block.bigint([])], 0)
    assert oracle.filter_project(empty, None, [c0]) is None


def test_agg1_global_aggregates(oracle):
    """TestAggregationOperator.testAggregation (…/operator/TestAggregationOperator.java:119-156), the aggregates on
    the device path: count 100, sum(bigint@0) 4950, avg 49.5, count(varchar) 100, sum(bigint@500) 54950, sum(double@500) 54950.0"""
    page = sequence_page(100, [(abi.VARCHAR, 0), (abi.BIGINT, 0), (abi.VARCHAR, 300), (abi.BIGINT, 500), (abi.DOUBLE, 500), (abi.VARCHAR, 500)])
    types = [abi.VARCHAR, abi.BIGINT, abi.VARCHAR, abi.BIGINT, abi.DOUBLE, abi.VARCHAR]
    agg = oracle.HashAggregation(types, [], [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_AVG, 1, abi.BIGINT),
                                             (abi.AGG_COUNT, 0, abi.VARCHAR), (abi.AGG_SUM, 3, abi.BIGINT), (abi.AGG_SUM, 4, abi.DOUBLE)])
    agg.add_page(page)
    assert agg.build_result().to_rows() == [(100, 4950, 49.5, 100, 54950, 54950.0)]


def test_agg_mask_with_dirty_nulls(oracle):
    """TestAggregationOperator.testMaskWithDirtyNulls (…/TestAggregationOperator.java:91-117): values {1,2,3,4},
    mask bytes {0,27,0,75} with nulls {T,T,F,F} -> count = 1"""
    mask = Block(abi.BOOLEAN, abi.FLAT, 4, values=np.array([0, 27, 0, 75], dtype=np.uint8), nulls=np.array([1, 1, 0, 0], dtype=np.uint8))
    page = Page([Block.bigint([1, 2, 3, 4]), mask], 4)
    agg = oracle.HashAggregation([abi.BIGINT, abi.BOOLEAN], [], [(abi.AGG_COUNT_STAR, -1, None, 1)])
    agg.add_page(page)
    assert agg.build_result().to_rows() == [(1,)]


def test_agg_empty_input_default_row(oracle):
    """AggregationOperator emits one row even without input: count 0, sum NULL (DoubleSumAggregation.java:54-63)"""
    agg = oracle.HashAggregation([abi.DOUBLE], [], [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 0, abi.DOUBLE), (abi.AGG_AVG, 0, abi.DOUBLE)])
    assert agg.build_result().to_rows() == [(0, None, None)]


def hagg1_input(oracle, hashed, rows=40000):
    """rowPagesBuilder(hashEnabled, hashChannels = [1], VARCHAR, VARCHAR, VARCHAR, BIGINT, BOOLEAN) with three
    addSequencePage(rows, 100, 0, {1,2,3}00_000, 0, 500) pages (TestHashAggregationOperator.java:172-177); hashEnabled appends the
    $hashvalue of the hash channels as a last channel (RowPagesBuilder.java)."""
    types = [abi.VARCHAR, abi.VARCHAR, abi.VARCHAR, abi.BIGINT, abi.BOOLEAN]
    pages = [sequence_page(rows, [(abi.VARCHAR, 100), (abi.VARCHAR, 0), (abi.VARCHAR, start), (abi.BIGINT, 0), (abi.BOOLEAN, 500)])
             for start in (100_000, 200_000, 300_000)]
    hc = -1
    if hashed:
        pages = [Page(p.blocks + [Block.bigint(oracle.hash_page(p, [1]))], rows) for p in pages]
        types, hc = types + [abi.BIGINT], 5
    return types, pages, hc


# COUNT, LONG_SUM(3), LONG_AVERAGE(3), max(varchar @2), count(varchar @0), count(boolean @4): TestHashAggregationOperator.java:186-191
HAGG1_AGGREGATES = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 3, abi.BIGINT), (abi.AGG_AVG, 3, abi.BIGINT), (abi.AGG_MAX, 2, abi.VARCHAR),
                    (abi.AGG_COUNT, 0, abi.VARCHAR), (abi.AGG_COUNT, 4, abi.BOOLEAN)]


def check_hagg1_rows(oracle, rows, hashed, n=40000):
    """expectedBuilder.row(Integer.toString(i), 3L, 3L * i, (double) i, Integer.toString(300_000 + i), 3L, 3L), compared ignoring order
    (assertPagesEqualIgnoreOrder, the $hashvalue column dropped: TestHashAggregationOperator.java:205-216)"""
    assert len(rows) == n
    seen = set()
    for r in rows:
        key, rest = r[0], r[1:]
        i = int(key)
        if hashed:
            assert rest[0] == oracle.combine_hash(0, oracle._s64(oracle.xxh64(key)))
            rest = rest[1:]
        assert key == str(i).encode() and rest == (3, 3 * i, float(i), str(300_000 + i).encode(), 3, 3), r
        seen.add(i)
    assert seen == set(range(n))


def drive(op, pages):
    """toPages (OperatorAssertion.java): the Driver's loop -- needsInput / addInput per page, finish, getOutput until finished."""
    out = []
    for p in pages:
        assert op.needsInput()
        op.addInput(p)
        o = op.getOutput()
        if o is not None:
            out.append(o)
    op.finish()
    for _ in range(1000):
        if op.isFinished():
            break
        o = op.getOutput()
        if o is not None and o.position_count:
            out.append(o)
    assert op.isFinished()
    return out


@pytest.mark.parametrize("hashed", [False, True])
def test_hagg1_hash_aggregation(oracle, hashed):
    """TestHashAggregationOperator.testHashAggregation (…/TestHashAggregationOperator.java:160-219): 3 pages x 40 000 rows keyed by a
    VARCHAR sequence, all six aggregates; more than one output page (:213)."""
    types, pages, hc = hagg1_input(oracle, hashed)
    op = oracle.HashAggregationOperator(types, [1], HAGG1_AGGREGATES, hash_channel=hc, expected_groups=100_000)
    out = drive(op, pages)
    assert len(out) > 1   # "Expected more than one output page"
    check_hagg1_rows(oracle, [r for p in out for r in p.to_rows()], hashed)


# testHashAggregationWithGlobals (…/TestHashAggregationOperator.java:221-272).  The reference builds its factory with group-by TYPES
# (VARCHAR, BIGINT) over channels (1, 2) of a (VARCHAR, VARCHAR, VARCHAR, BIGINT, BIGINT, BOOLEAN) page layout -- no page ever flows, so
# the mismatch of channel 2 never shows.  Here the key types come from the channels: channel 2 is declared BIGINT and max(varchar) reads a
# VARCHAR channel of its own (6); the operator sees what the reference's sees: keys (VARCHAR, BIGINT), group id in key column 1, ids 42, 49.
GLOBALS_TYPES = [abi.VARCHAR, abi.VARCHAR, abi.BIGINT, abi.BIGINT, abi.BIGINT, abi.BOOLEAN, abi.VARCHAR]
GLOBALS_AGGREGATES = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MIN, 4, abi.BIGINT), (abi.AGG_AVG, 4, abi.BIGINT), (abi.AGG_MAX, 6, abi.VARCHAR),
                      (abi.AGG_COUNT, 0, abi.VARCHAR), (abi.AGG_COUNT, 5, abi.BOOLEAN)]
GLOBALS_EXPECTED = [(None, 42, 0, None, None, None, 0, 0), (None, 49, 0, None, None, None, 0, 0)]


def globals_hash(oracle, group_id):
    """calculateDefaultOutputHash (HashAggregationOperator.java:589-600) over (NULL, id)"""
    return oracle.combine_hash(oracle.combine_hash(0, 0), oracle.hash_bigint(group_id))


@pytest.mark.parametrize("hashed", [False, True])
def test_hash_aggregation_with_globals(oracle, hashed):
    types, hc = (GLOBALS_TYPES + [abi.BIGINT], 7) if hashed else (GLOBALS_TYPES, -1)
    op = oracle.HashAggregationOperator(types, [1, 2], GLOBALS_AGGREGATES, hash_channel=hc, expected_groups=100_000,
                                        global_aggregation_group_ids=[42, 49], group_id_channel=1, produce_default_output=True)
    out = drive(op, [])
    rows = [r for p in out for r in p.to_rows()]
    if hashed:
        assert [r[2] for r in rows] == [globals_hash(oracle, 42), globals_hash(oracle, 49)]
        rows = [r[:2] + r[3:] for r in rows]
    assert rows == GLOBALS_EXPECTED
    # with input the default rows are not produced (inputProcessed, :386, 488); without produceDefaultOutput nothing is emitted at all
    op = oracle.HashAggregationOperator(GLOBALS_TYPES, [1, 2], GLOBALS_AGGREGATES, global_aggregation_group_ids=[42, 49], group_id_channel=1,
                                        produce_default_output=True)
    page = Page([Block.varchar(["a"]), Block.varchar(["k"]), Block.bigint([7]), Block.bigint([1]), Block.bigint([5]), Block.boolean([True]),
                 Block.varchar(["z"])], 1)
    assert [r for p in drive(op, [page]) for r in p.to_rows()] == [(b"k", 7, 1, 5, 5.0, b"z", 1, 1)]
    op = oracle.HashAggregationOperator(GLOBALS_TYPES, [1, 2], GLOBALS_AGGREGATES, global_aggregation_group_ids=[42, 49], group_id_channel=1)
    assert drive(op, []) == []
    # a PARTIAL step emits the empty intermediate states (evaluateIntermediate, :576-578): [count 0] / [count 0, value NULL or 0]
    op = oracle.HashAggregationOperator(GLOBALS_TYPES, [1, 2], GLOBALS_AGGREGATES, step=abi.STEP_PARTIAL, global_aggregation_group_ids=[42],
                                        group_id_channel=1, produce_default_output=True)
    (row,), = [p.to_rows() for p in drive(op, [])]
    assert row[:2] == (None, 42) and row[2] == 0 and row[3:5] == (0, None) and row[5] == 0 and row[7:9] == (0, None) and row[9:] == (0, 0)


def hash_builder_resize_pages(oracle, hashed):
    """testHashBuilderResize's input (…/TestHashAggregationOperator.java:360-399): addSequencePage(10, 100), one row holding a
    200 000-byte string of zero bytes (larger than MAX_BLOCK_SIZE_IN_BYTES), addSequencePage(10, 100)"""
    big = Page([Block.varchar([b"\0" * 200_000])], 1)
    pages = [sequence_page(10, [(abi.VARCHAR, 100)]), big, sequence_page(10, [(abi.VARCHAR, 100)])]
    types, hc = [abi.VARCHAR], -1
    if hashed:
        pages = [Page(p.blocks + [Block.bigint(oracle.hash_page(p, [0]))], p.position_count) for p in pages]
        types, hc = types + [abi.BIGINT], 1
    return types, pages, hc


def check_hash_builder_resize_rows(rows, hashed):
    got = sorted((r[0], r[-1]) for r in rows)
    assert got == sorted([(str(100 + i).encode(), 2) for i in range(10)] + [(b"\0" * 200_000, 1)])


@pytest.mark.parametrize("hashed", [False, True])
def test_hash_builder_resize(oracle, hashed):
    """The reference asserts only that the operator gets through (toPages); the rows are checked here as well."""
    types, pages, hc = hash_builder_resize_pages(oracle, hashed)
    op = oracle.HashAggregationOperator(types, [0], [(abi.AGG_COUNT_STAR, -1, None)], hash_channel=hc, expected_groups=100_000)
    check_hash_builder_resize_rows([r for p in drive(op, pages) for r in p.to_rows()], hashed)


MULTI_SLICE_POSITIONS = int(1.5 * 1024 * 1024 / 32)   # testMultiSliceAggregationOutput: 1.5 x DEFAULT_MAX_PAGE_SIZE_IN_BYTES / fixedWidthSize


def multi_slice_input(oracle, hashed):
    page = sequence_page(MULTI_SLICE_POSITIONS, [(abi.BIGINT, 0), (abi.BIGINT, 0)])
    types, hc = [abi.BIGINT, abi.BIGINT], -1
    if hashed:
        page = Page(page.blocks + [Block.bigint(oracle.hash_page(page, [1]))], page.position_count)
        types, hc = types + [abi.BIGINT], 2
    return types, [page], hc


@pytest.mark.parametrize("hashed", [False, True])
def test_multi_slice_aggregation_output(oracle, hashed):
    """TestHashAggregationOperator.testMultiSliceAggregationOutput (…/TestHashAggregationOperator.java:477-510): 49 152 BIGINT groups,
    count + avg -> the result leaves as TWO pages (the PageBuilder fills at 1 MB)."""
    types, pages, hc = multi_slice_input(oracle, hashed)
    op = oracle.HashAggregationOperator(types, [1], [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_AVG, 1, abi.BIGINT)], hash_channel=hc, expected_groups=100_000)
    out = drive(op, pages)
    assert len(out) == 2
    rows = [r for p in out for r in p.to_rows()]
    assert [(r[0], r[-2], r[-1]) for r in rows] == [(i, 1, float(i)) for i in range(MULTI_SLICE_POSITIONS)]


def test_gbh1_group_ids_are_first_seen_ordinals(oracle):
    """TestGroupByHash.testAddPage / testGetGroupIds (…/operator/TestGroupByHash.java:69-140): values 0..499 one page at a
    time, twice, each page added 10 times -> group id == value, count stable on re-add"""
    gbh = oracle.HashAggregation([abi.BIGINT], [0], [], expected_groups=100)
    for tries in range(2):
        for value in range(500):
            page = Page([Block.bigint([value])], 1)
            for _ in range(3):
                ids = gbh.add_page(page, want_group_ids=True)
                assert ids.tolist() == [value]
                assert gbh.group_count() == (value + 1 if tries == 0 else 500)


def test_gbh_null_group_and_rehash(oracle):
    """TestGroupByHash.testNullGroup (…/TestGroupByHash.java:98-118): NULL, then 1..132747 forces rehashes; contains(0) false"""
    gbh = oracle.HashAggregation([abi.BIGINT], [0], [], expected_groups=100)
    gbh.add_page(Page([Block.bigint([0], [1])], 1))
    gbh.add_page(Page([Block.bigint(np.arange(1, 132748))], 132747))
    assert gbh.group_count() == 132748
    assert not gbh.contains(Page([Block.bigint([0])], 1), 0)
    assert gbh.contains(Page([Block.bigint([5])], 1), 0) and gbh.contains(Page([Block.bigint([0], [1])], 1), 0)
    assert gbh.capacity() == 262144  # 256 doubled while groups >= 0.75 * capacity


def test_gbh_force_rehash_multichannel(oracle):
    """TestGroupByHash.testForceRehash (…/TestGroupByHash.java:238-253): 100 VARCHAR keys into a table sized for 4"""
    page = sequence_page(100, [(abi.VARCHAR, 0)])
    gbh = oracle.HashAggregation([abi.VARCHAR], [0], [], expected_groups=4)
    ids = gbh.add_page(page, want_group_ids=True)
    assert ids.tolist() == list(range(100))
    assert all(gbh.contains(page, i) for i in range(100))
    assert gbh.capacity() == 256


def test_gbh_contains_multiple_columns(oracle):
    """TestGroupByHash.testContainsMultipleColumns (…/TestGroupByHash.java:218-235)"""
    page = Page([Block.double(np.arange(10, dtype=np.float64)), Block.varchar([str(i) for i in range(10)])], 10)
    gbh = oracle.HashAggregation([abi.DOUBLE, abi.VARCHAR], [0, 1], [], expected_groups=100)
    gbh.add_page(page)
    assert gbh.contains(Page([Block.double([3.0]), Block.varchar(["3"])], 1), 0)
    assert not gbh.contains(Page([Block.double([3.0]), Block.varchar(["4"])], 1), 0)


def test_links1_position_links_chain_order(oracle):
    """TestPositionLinks.testArrayPositionLinks (…/join/TestPositionLinks.java:39-64): link(new, head) chains descend"""
    keys = [7, 7, 7, 7, 1, 2, 3, 4, 5, 6, 9, 9, 9]  # positions 0..3 share a key, 10..12 share a key
    j = oracle.HashJoin([abi.BIGINT], [0], [0])
    j.add_build_page(Page([Block.bigint(keys)], len(keys)))
    j.build()
    key, links = j.tables()
    assert links.tolist() == [-1, 0, 1, 2, -1, -1, -1, -1, -1, -1, -1, 10, 11]
    assert sorted(k for k in key.tolist() if k >= 0) == [3, 4, 5, 6, 7, 8, 9, 12]  # heads = last inserted of each key


@pytest.mark.parametrize("probe_hash,build_hash", [(False, False), (True, False), (False, True), (True, True)])
def test_join1_inner_join(oracle, probe_hash, build_hash):
    """TestHashJoinOperator.testInnerJoin (…/join/TestHashJoinOperator.java:192-229)"""
    btypes = ptypes = [abi.VARCHAR, abi.BIGINT, abi.BIGINT]
    build = sequence_page(10, [(abi.VARCHAR, 20), (abi.BIGINT, 30), (abi.BIGINT, 40)])
    probe = sequence_page(1000, [(abi.VARCHAR, 0), (abi.BIGINT, 1000), (abi.BIGINT, 2000)])
    bh = ph = -1
    if build_hash:
        build, btypes, bh = Page(build.blocks + [Block.bigint(oracle.hash_page(build, [0]))], 10), btypes + [abi.BIGINT], 3
    if probe_hash:
        probe, ptypes, ph = Page(probe.blocks + [Block.bigint(oracle.hash_page(probe, [0]))], 1000), ptypes + [abi.BIGINT], 3
    j = oracle.HashJoin(btypes, [0], [0, 1, 2], hash_channel=bh)
    j.add_build_page(build)
    j.build()
    out, pi, bi = j.probe(probe, ptypes, [0], [0, 1, 2], ph)
    assert out.to_rows() == [(str(20 + i).encode(), 1020 + i, 2020 + i, str(20 + i).encode(), 30 + i, 40 + i) for i in range(10)]
    assert pi.tolist() == list(range(20, 30)) and bi.tolist() == list(range(10))


def test_join_nulls_never_match(oracle):
    """TestHashJoinOperator NULL cases (…/join/TestHashJoinOperator.java:694-850): NULL keys on either side never match"""
    build = Page([Block.varchar(["a", None, None, "a", "b"])], 5)
    probe = Page([Block.varchar(["a", None, "b", "c"])], 4)
    j = oracle.HashJoin([abi.VARCHAR], [0], [0])
    j.add_build_page(build)
    j.build()
    out, pi, bi = j.probe(probe, [abi.VARCHAR], [0], [0])
    assert out.to_rows() == [(b"a", b"a"), (b"a", b"a"), (b"b", b"b")]
    assert pi.tolist() == [0, 0, 2] and bi.tolist() == [3, 0, 4]  # chain: last inserted first


def test_join_rle_probe(oracle):
    """…/TestHashJoinOperator.java:232-267: an RLE probe block replicates its matches"""
    build = Page([Block.bigint([5, 6, 5])], 3)
    probe = Page([Block.rle(Block.bigint([5]), 3)], 3)
    j = oracle.HashJoin([abi.BIGINT], [0], [0])
    j.add_build_page(build)
    j.build()
    out, pi, bi = j.probe(probe, [abi.BIGINT], [0], [0])
    assert pi.tolist() == [0, 0, 1, 1, 2, 2] and bi.tolist() == [2, 0, 2, 0, 2, 0]


def test_bigint_sum_overflow_is_an_error(oracle):
    agg = oracle.HashAggregation([abi.BIGINT], [], [(abi.AGG_SUM, 0, abi.BIGINT)])
    with pytest.raises(oracle.OracleError) as e:
        agg.add_page(Page([Block.bigint([2 ** 62, 2 ** 62])], 2))
    assert e.value.status == abi.ERR_NUMERIC_VALUE_OUT_OF_RANGE


def test_q1_group_structure_against_sf1_answers(oracle):
    """Distribution sanity of the synthetic generator against the reference's SF1 answers (tests/golden/tpch_sf1_answers.json):
    the same 4 (returnflag, linestatus) groups with matching proportions."""
    import json, os
    from presto_amd import tpch
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tpch_sf1_answers.json")))
    n = 400000
    cols = [oracle.tpch_column(c, 1.0, 0, n) for c in tpch.Q1_COLUMNS]
    args = [cols[0][0], cols[0][1], cols[1][0], cols[1][1]] + [c[0] for c in cols[2:]]
    rows = oracle.q1(args)
    got = {(r[0].decode(), r[1].decode()): r[9] for r in rows}
    ref = {(x["returnflag"], x["linestatus"]): x["count_order"] for x in g["q01_groups"]}
    assert set(got) == set(ref)
    tg, tr = sum(got.values()), sum(ref.values())
    for k in ref:
        assert abs(got[k] / tg - ref[k] / tr) < 0.03, (k, got[k] / tg, ref[k] / tr)


# ---- MergePages (core/trino-main/src/test/java/io/trino/operator/project/TestMergePages.java) ------------------------
MERGE_TYPES = [(abi.VARCHAR, 0), (abi.BIGINT, 0), (abi.DOUBLE, 0)]


def merge_all(m, pages):
    out = []
    for p in pages:
        out += m.process(p)
    return out + m.finish()


def test_merge_pages_kats(oracle):
    from presto_amd.page import sequence_page
    page = sequence_page(10, MERGE_TYPES)
    size = oracle.page_size_in_bytes(page)
    assert size == (10 + 5 * 10) + 9 * 10 + 9 * 10  # "0".."9" + (4 + 1) per position; (8 + 1) per position twice
    # testMinPageSizeThreshold (:44-59): a page of exactly minPageSizeInBytes passes through
    out = merge_all(oracle.MergePages(size, 2 ** 31 - 1, 2 ** 31 - 1), [page])
    assert len(out) == 1 and out[0] is page
    # testMinRowCountThreshold (:61-76)
    out = merge_all(oracle.MergePages(1024 * 1024, 10, 2 ** 31 - 1), [page])
    assert len(out) == 1 and out[0] is page
    # testBufferSmallPages (:106-123): two halves come out as the whole
    whole = sequence_page(20, MERGE_TYPES)
    halves = [whole.get_region(0, 10), whole.get_region(10, 10)]
    out = merge_all(oracle.MergePages(oracle.page_size_in_bytes(whole) + 1, 21, 2 ** 31 - 1), halves)
    assert [p.to_rows() for p in out] == [whole.to_rows()]
    # testFlushOnBigPage (:125-142): the buffered small page first, then the big one
    small, big = sequence_page(10, MERGE_TYPES), sequence_page(100, MERGE_TYPES)
    out = merge_all(oracle.MergePages(oracle.page_size_in_bytes(big), 100, 2 ** 31 - 1), [small, big])
    assert [p.to_rows() for p in out] == [small.to_rows(), big.to_rows()]
    # testFlushOnFullPage (:144-163): BIGINT halves, max page size = the whole page: two whole pages
    whole = sequence_page(20, [(abi.BIGINT, 0)])
    halves = [whole.get_region(0, 10), whole.get_region(10, 10)]
    size = oracle.page_size_in_bytes(whole)
    out = merge_all(oracle.MergePages(size // 2 + 1, 11, size), halves + halves)
    assert [p.to_rows() for p in out] == [whole.to_rows(), whole.to_rows()]


# ---- outer joins (core/trino-main/src/test/java/io/trino/operator/join/TestHashJoinOperator.java:850-1390) ------------
def _varchar_join(oracle, build, probe, join_type):
    j = oracle.HashJoin([abi.VARCHAR], [0], [0])
    if build:
        j.add_build_page(Page([Block.varchar(build)], len(build)))
    j.build()
    out, pi, bi = j.probe(Page([Block.varchar(probe)], len(probe)), [abi.VARCHAR], [0], [0], join_type=join_type)
    return j, out.to_rows()


def test_probe_outer_join_kats(oracle):
    # testProbeOuterJoin (:850-894): build seq 10 rows @20,30,40; probe seq 15 rows @20,1020,2020
    types = [abi.VARCHAR, abi.BIGINT, abi.BIGINT]
    j = oracle.HashJoin(types, [0], [0, 1, 2])
    j.add_build_page(sequence_page(10, [(abi.VARCHAR, 20), (abi.BIGINT, 30), (abi.BIGINT, 40)]))
    j.build()
    out, _, _ = j.probe(sequence_page(15, [(abi.VARCHAR, 20), (abi.BIGINT, 1020), (abi.BIGINT, 2020)]), types, [0], [0, 1, 2],
                        join_type=abi.JOIN_PROBE_OUTER)
    expected = [(str(20 + i).encode(), 1020 + i, 2020 + i, str(20 + i).encode(), 30 + i, 40 + i) for i in range(10)]
    expected += [(str(30 + i).encode(), 1030 + i, 2030 + i, None, None, None) for i in range(5)]
    assert out.to_rows() == expected
    # testOuterJoinWithNullProbe (:945-985)
    assert _varchar_join(oracle, ["a", "b", "c"], ["a", None, None, "a", "b"], abi.JOIN_PROBE_OUTER)[1] == \
        [(b"a", b"a"), (None, None), (None, None), (b"a", b"a"), (b"b", b"b")]
    # testOuterJoinWithNullBuild (:1032-1071)
    assert _varchar_join(oracle, ["a", None, None, "a", "b"], ["a", "b", "c"], abi.JOIN_PROBE_OUTER)[1] == \
        [(b"a", b"a"), (b"a", b"a"), (b"b", b"b"), (b"c", None)]
    # testOuterJoinWithNullOnBothSides (:1118-1158)
    assert _varchar_join(oracle, ["a", None, None, "a", "b"], ["a", "b", None, "c"], abi.JOIN_PROBE_OUTER)[1] == \
        [(b"a", b"a"), (b"a", b"a"), (b"b", b"b"), (None, None), (b"c", None)]
    # testProbeOuterJoinWithEmptyLookupSource (:1297-1343) / testFullOuterJoinWithEmptyLookupSource (:1345-1390)
    for jt in (abi.JOIN_PROBE_OUTER, abi.JOIN_FULL_OUTER):
        assert _varchar_join(oracle, [], ["a", "b", None, "c"], jt)[1] == [(b"a", None), (b"b", None), (None, None), (b"c", None)]


def test_lookup_outer_and_full_outer(oracle):
    # the build rows no probe row joined with come out afterwards, ascending, probe side NULL
    # (OuterLookupSource.java:120-160; testLookupOuterJoinWithEmptyLookupSource :1258-1295 for the empty case)
    j, rows = _varchar_join(oracle, ["a", None, "x", "a", "b", "y"], ["a", "c", None], abi.JOIN_FULL_OUTER)
    assert rows == [(b"a", b"a"), (b"a", b"a"), (b"c", None), (None, None)]
    assert j.outer([abi.VARCHAR], [0]).to_rows() == [(None, None), (None, b"x"), (None, b"b"), (None, b"y")]
    j, rows = _varchar_join(oracle, ["a", "b"], ["b", "z"], abi.JOIN_LOOKUP_OUTER)
    assert rows == [(b"b", b"b")]
    assert j.outer([abi.VARCHAR], [0]).to_rows() == [(None, b"a")]
    j, rows = _varchar_join(oracle, [], ["test"], abi.JOIN_LOOKUP_OUTER)
    assert rows == [] and j.outer([abi.VARCHAR], [0]).position_count == 0


def test_page_wire_format_known_frame(oracle):
    """The reference's tests only round-trip PagesSerde (core/trino-main/src/test/java/io/trino/execution/buffer/TestPagesSerde.java),
    they hold no golden bytes; this frame is computed by hand from the cited encodings (PagesSerdeUtil.java:45-74,
    LongArrayBlockEncoding.java:38-61, VariableWidthBlockEncoding.java:37-58, EncoderUtil.java:35-72): (BIGINT 1, NULL, 3),
    (VARCHAR 'ab', '', NULL)."""
    import struct
    page = Page([Block.bigint([1, 0, 3], [0, 1, 0]), Block.varchar([b"ab", b"", None])], 3)
    payload = struct.pack("<i", 2)
    payload += struct.pack("<i", 10) + b"LONG_ARRAY" + struct.pack("<i", 3) + bytes([1, 0b01000000]) + struct.pack("<iqq", 2, 1, 3)
    payload += struct.pack("<i", 14) + b"VARIABLE_WIDTH" + struct.pack("<i", 3) + struct.pack("<iii", 2, 2, 2) + bytes([1, 0b00100000]) + struct.pack("<i", 2) + b"ab"
    frame = struct.pack("<ibii", 3, 0, len(payload), len(payload)) + payload
    assert oracle.serialize_page(page) == frame
    assert oracle.deserialize_page(frame).to_rows() == [(1, b"ab"), (None, b""), (3, None)]
    rng = np.random.default_rng(3)
    n = 1001
    big = Page([Block.double(rng.random(n), rng.random(n) < 0.2), Block.integer(rng.integers(-5, 5, n)), Block.boolean(rng.random(n) < 0.5, rng.random(n) < 0.5)], n)
    back = oracle.deserialize_page(oracle.serialize_page(big))
    assert back.blocks[1].to_pylist() == big.blocks[1].to_pylist()
    assert [None if v is None else int(v) for v in back.blocks[2].to_pylist()] == [None if v is None else int(v) for v in big.blocks[2].to_pylist()]
    assert [None if v is None else np.int64(v).view(np.float64) for v in back.blocks[0].to_pylist()] == big.blocks[0].to_pylist()


# ---- DynamicFilterSourceOperator KATs (TestDynamicFilterSourceOperator.java; defaults :121-124: 100 distinct values, 10 kB,
# 1 000 000 rows for min / max) ------------------------------------------------------------------------------------------
def dynamic_filter_kats():
    """(name, types, filter channels, pages as column lists, (max distinct, max bytes, min/max row limit), expected)"""
    seq = lambda a, b: list(range(a, b))
    nan = float("nan")
    B, I, D, BO, V, R = abi.BIGINT, abi.INTEGER, abi.DOUBLE, abi.BOOLEAN, abi.VARCHAR, abi.REAL
    dflt = (100, 10240, 1000000)
    text = b"A" * 10241
    ta, tb = b"A" * 5121, b"B" * 5121
    return [
        ("multiple operators 1", [B], [0], [[[1, 2]], [[3, 5]]], dflt, [("values", [1, 2, 3, 5])]),  # :181-207
        ("multiple operators 2", [B], [0], [[[2, 3]], [[1, 4]]], dflt, [("values", [1, 2, 3, 4])]),
        ("multiple columns", [BO, D], [0, 1], [[[True, True], [1.5, 3.0]], [[False], [4.5]]], dflt,
         [("values", [False, True]), ("values", [1.5, 3.0, 4.5])]),  # :209-223
        ("only first column", [BO, D], [0], [[[True, True], [1.5, 3.0]], [[False], [4.5]]], dflt, [("values", [False, True])]),
        ("only last column", [BO, D], [1], [[[True, True], [1.5, 3.0]], [[False], [4.5]]], dflt, [("values", [1.5, 3.0, 4.5])]),
        ("nulls", [I], [0], [[[1, 2, 3]], [[3, None, 4]], [[4, 5]]], dflt, [("values", [1, 2, 3, 4, 5])]),  # :255-276
        ("double NaN", [D], [0], [[[42.0, nan]]], dflt, [("values", [42.0])]),  # :278-294
        ("too many rows double", [D], [0], [[[float(i) for i in seq(0, 101)]], [[nan] * 101]], dflt, "all"),  # :314-325
        ("real NaN", [R], [0], [[[42.0, nan]]], dflt, [("values", [42.0])]),  # :296-312
        ("too many rows real", [R], [0], [[[float(i) for i in seq(0, 101)]], [[nan] * 101]], dflt, "all"),  # :327-338
        ("no filters", [B], [], [[[1, 2, 3]]], dflt, "all"),  # :370-379
        ("empty build side", [B], [0], [], dflt, [("none",)]),  # :381-389
        ("min max when too many positions", [B], [0], [[seq(0, 101)]], dflt, [("range", 0, 100)]),  # :391-406
        ("below the distinct limit", [B, B, B], [0, 1, 2], [[seq(0, 101), seq(100, 201), seq(200, 301)]], (101, 10240, 1000000),
         [("values", seq(0, 101)), ("values", seq(100, 201)), ("values", seq(200, 301))]),  # :408-429
        ("min max when too many distinct", [B, B], [0, 1], [[seq(0, 101), [200] * 101]], dflt, [("range", 0, 100), ("range", 200, 200)]),  # :431-447 (COLOR channel left out)
        ("min max with nulls", [B, B], [0, 1], [[[None] * 100, seq(200, 300)]], (99, 10240, 1000000), [("none",), ("range", 200, 299)]),  # :449-462 shape
        ("too many bytes", [V], [0], [[[text]]], (100, 10240, 100), [("range", text, text)]),  # :464-483
        ("too many bytes, two columns", [V, V], [0, 1], [[[ta], [tb]]], (100, 10240, 100), [("range", ta, ta), ("range", tb, tb)]),  # :485-507
        ("multiple large pages", [B], [0], [[seq(50, 151)], [seq(0, 101)], [seq(100, 201)]], dflt, [("range", 0, 200)]),  # :509-526
        ("deduplication", [B], [0], [[[7] * 1000], [[None] * 1000]], dflt, [("values", [7])]),  # :528-542
        ("min max limit, single page", [B], [0], [[seq(0, 201)]], (100, 10240, 200), "all"),  # :544-555
        ("min max limit, multiple pages", [B], [0], [[seq(0, 101)], [seq(0, 101)]], (100, 10240, 201), "all"),  # :557-570
    ]


def dynamic_filter_pages(types, pages):
    make = {abi.BIGINT: Block.bigint, abi.INTEGER: Block.integer, abi.DOUBLE: Block.double, abi.BOOLEAN: Block.boolean, abi.REAL: Block.real}
    out = []
    for cols in pages:
        blocks = []
        for t, col in zip(types, cols):
            if t == abi.VARCHAR:
                blocks.append(Block.varchar(col))
            else:
                nulls = [v is None for v in col]
                blocks.append(make[t]([0 if v is None else v for v in col], nulls if any(nulls) else None))
        out.append(Page(blocks, len(cols[0])))
    return out


@pytest.mark.parametrize("kat", dynamic_filter_kats(), ids=lambda k: k[0])
def test_dynamic_filter_source_kats(oracle, kat):
    name, types, channels, pages, (max_distinct, max_bytes, row_limit), expected = kat
    op = oracle.DynamicFilterSource(types, channels, max_distinct, max_bytes, row_limit)
    for p in dynamic_filter_pages(types, pages):
        assert op.add_page(p) is p  # verifyPassthrough
    op.finish()
    got = "all" if op.predicate == [] else op.predicate
    assert got == expected


def test_join_output_single_match_kat(oracle):
    """TestHashJoinOperator.testInnerJoinWithOutputSingleMatch (…/operator/join/TestHashJoinOperator.java:733-767):
    build a, a, b; probe a, b, c -> (a, a), (b, b)"""
    j = oracle.HashJoin([abi.VARCHAR], [0], [0])
    j.add_build_page(Page([Block.varchar([b"a", b"a", b"b"])], 3))
    j.build()
    out, pi, bi = j.probe(Page([Block.varchar([b"a", b"b", b"c"])], 3), [abi.VARCHAR], [0], [0], output_single_match=True)
    assert out.to_rows() == [(b"a", b"a"), (b"b", b"b")] and pi.tolist() == [0, 1]


def test_real_type_hand_computed(oracle):
    """RealType / RealOperators / RealSumAggregation / RealAverageAggregation restated: values are IEEE singles, arithmetic is Java
    float arithmetic (0.1f + 0.2f = 0.3f rounded in single precision, 1e8f + 1f = 1e8f), sums accumulate the widened values in
    double and narrow on output, avg = (float)(double sum / count).  Expected values computed with numpy float32 / float64."""
    import numpy as np
    from presto_amd.expr import constant, field
    from presto_amd.page import Block, Page
    F = np.float32
    page = Page([Block.real([F(0.1), F(1e8), F(3.0), F(-2.5)], [0, 0, 1, 0]), Block.real([F(0.2), F(1.0), F(1.0), F(4.0)])], 4)
    a, b = field(0, abi.REAL), field(1, abi.REAL)
    out = oracle.filter_project(page, b < constant(3.5, abi.REAL), [a + b, a * b, a / b, a % b, -a, a.cast(abi.DOUBLE), constant(16777217, abi.BIGINT).cast(abi.REAL)])
    assert out.to_rows() == [
        (float(F(0.1) + F(0.2)), float(F(0.1) * F(0.2)), float(F(0.1) / F(0.2)), float(np.fmod(F(0.1), F(0.2))), float(-F(0.1)), float(F(0.1)), 16777216.0),
        (100000000.0, 100000000.0, 100000000.0, 0.0, -100000000.0, 100000000.0, 16777216.0),
        (None, None, None, None, None, None, 16777216.0)]
    assert [blk.type for blk in out.blocks] == [abi.REAL] * 5 + [abi.DOUBLE, abi.REAL]
    agg = oracle.HashAggregation([abi.REAL, abi.REAL], [], [(abi.AGG_SUM, 0, abi.REAL), (abi.AGG_AVG, 1, abi.REAL), (abi.AGG_MIN, 0, abi.REAL), (abi.AGG_MAX, 0, abi.REAL),
                                                            (abi.AGG_COUNT, 0, abi.REAL)])
    agg.add_page(page)
    dsum = float(F(0.1)) + 1e8 + float(F(-2.5))                       # the double state of RealSumAggregation
    davg = (float(F(0.2)) + 1.0 + 1.0 + 4.0) / 4
    assert agg.build_result().to_rows() == [(float(F(dsum)), float(F(davg)), -2.5, 100000000.0, 3)]
    # the wire format carries the raw bits as INT_ARRAY
    frame = oracle.serialize_page(page)
    assert b"INT_ARRAY" in frame and oracle.deserialize_page(frame).position_count == 4


def test_real_keys_hand_computed(oracle):
    """RealType as a key (core/trino-spi/.../type/RealType.java:101-140): hash = AbstractLongType.hash(floatToIntBits(v == 0 ? 0 : v)) --
    computed here from the constants of AbstractLongType.java:126-130, independently of the oracle's C --, -0 and +0 one group, NaN
    one group, NaN joins nothing, Float.compare order."""
    import struct
    M = (1 << 64) - 1

    def long_hash(v):  # rotateLeft(v * 0xC2B2AE3D27D4EB4F, 31) * 0x9E3779B185EBCA87
        x = (v * 0xC2B2AE3D27D4EB4F) & M
        x = ((x << 31) | (x >> 33)) & M
        x = (x * 0x9E3779B185EBCA87) & M
        return x - (1 << 64) if x >> 63 else x

    def real_hash(f):
        bits = struct.unpack("<i", struct.pack("<f", 0.0 if f == 0 else f))[0] if f == f else 0x7fc00000
        return long_hash(bits & M if bits >= 0 else (bits + (1 << 64)))

    values = np.array([1.0, -2.5, 0.0, -0.0, np.nan, np.inf, 1e-42], dtype=np.float32)
    page = Page([Block.real(values), Block.bigint(np.arange(len(values)))], len(values))
    # row hash of one channel = 31 * 0 + hash(value)
    assert list(oracle.hash_page(page, [0])) == [real_hash(float(v)) for v in values]
    assert real_hash(0.0) == real_hash(-0.0)
    agg = oracle.HashAggregation([abi.REAL, abi.BIGINT], [0], [(abi.AGG_COUNT_STAR, -1, None)])
    agg.add_page(Page([Block.real(np.array([0.0, -0.0, np.nan, np.nan, 1.0, 1.0, 1.0], dtype=np.float32)), Block.bigint(np.zeros(7, dtype=np.int64))], 7))
    counts = sorted(r[1] for r in agg.build_result().to_rows())
    assert counts == [2, 2, 3]
    join = oracle.HashJoin([abi.REAL, abi.BIGINT], [0], [1])
    join.add_build_page(Page([Block.real(np.array([0.0, np.nan, 2.0], dtype=np.float32)), Block.bigint(np.array([10, 11, 12]))], 3))
    join.build()
    out, _, _ = join.probe(Page([Block.real(np.array([-0.0, np.nan, 2.0, 3.0], dtype=np.float32)), Block.bigint(np.arange(4))], 4), [abi.REAL, abi.BIGINT], [0], [1])
    assert out.to_rows() == [(0, 10), (2, 12)]          # -0 = +0; NaN = nothing
    rows = oracle.topn([Page([Block.real(np.array([np.nan, 0.0, -0.0, -np.inf, 5.0], dtype=np.float32))], 5)], 5, [0], [abi.ASC_NULLS_LAST])
    assert [repr(r[0]) for r in rows] == ["-inf", "-0.0", "0.0", "5.0", "nan"]


def test_tiny_q6_config_1_on_the_cpu_operators(oracle):
    """BASELINE config #1 (TPC-H tiny Q6, lineitem scan -> filter -> SUM on the CPU operators; plumbing, no GPU): over the
    60 175 lineitem rows of SF0.01 the oracle's hand-written twin of the pipeline (HandTpchQuery6.java:95-141) and the oracle's
    operator composition FilterAndProject -> Aggregation (what TpchQueryRunner's Driver runs) give the same count and -- both
    adding left to right -- bit-identical sums, through 8192-row pages as the Driver delivers them."""
    import numpy as np
    from presto_amd import abi, tpch
    from presto_amd.page import Block, Page
    n, sf = 60_175, 0.01
    cols = [oracle.tpch_column(c, sf, 0, n)[0] for c in tpch.Q6_COLUMNS]
    twin_sum, twin_count = oracle.q6(*cols)
    host = Page([Block.flat(t, c) for t, c in zip(tpch.Q6_TYPES, cols)], n)
    agg = oracle.HashAggregation([abi.DOUBLE], [], [(abi.AGG_SUM, 0, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)])
    selected = 0
    for i in range(0, n, 8192):
        out = oracle.filter_project(host.get_region(i, min(8192, n - i)), tpch.q6_filter(), tpch.q6_projections())
        if out is not None and out.position_count:
            selected += out.position_count
            agg.add_page(out)
    (s, c), = agg.build_result().to_rows()
    assert c == twin_count == selected and 500 < c < 2000
    assert np.float64(s).view(np.int64) == np.float64(twin_sum).view(np.int64)
    # and against numpy over the whole table (independent of both)
    m = (cols[0] >= 8766) & (cols[0] < 9131) & (cols[1] >= 0.05) & (cols[1] <= 0.07) & (cols[2] < 24.0)
    assert int(m.sum()) == c and abs(float((cols[3][m] * cols[1][m]).sum()) - s) <= 1e-12 * s


def test_c_sort_and_topn_twins_agree_with_the_python_restatements(oracle):
    """The timing-grade C twins of OrderBy / TopN (bench.py's CPU legs) against the Python restatements that are the parity
    checkers, and against numpy: PagesIndexOrdering.quickSort by one BIGINT key sorts and keeps every row once; TopNProcessor's
    heap under (DOUBLE DESC, BIGINT ASC) keeps the same rows in the same order, ties on both channels included."""
    import numpy as np
    from presto_amd import abi
    from presto_amd.page import Block, Page
    rng = np.random.default_rng(12)
    for n, hi in ((0, 5), (1, 5), (6, 3), (7, 3), (41, 10), (100_000, 1000), (100_000, 1 << 40)):
        k = rng.integers(-hi, hi, n)
        p = oracle.sort_positions_bigint(k)
        assert np.array_equal(k[p], np.sort(k)) and np.array_equal(np.sort(p), np.arange(n))
    k = np.arange(50_000)[::-1].copy()           # descending input, and all-equal keys
    assert np.array_equal(oracle.sort_positions_bigint(k), np.arange(50_000)[::-1])
    assert sorted(oracle.sort_positions_bigint(np.zeros(1000, np.int64)).tolist()) == list(range(1000))
    v = rng.random(5000)
    v[::7] = 0.5
    v[3], v[4], v[5] = -0.0, 0.0, np.nan
    kk = rng.integers(0, 50, 5000)
    for limit in (1, 10, 100, 5000, 6000):
        got = oracle.topn_positions_double_desc_bigint_asc(v, kk, limit)
        ref = oracle.topn([Page([Block.double(v), Block.bigint(kk)], 5000)], limit, [0, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST])
        assert len(got) == min(limit, 5000)
        for i, r in zip(got, ref):
            assert (np.float64(v[i]).view(np.int64) == np.float64(r[0]).view(np.int64) or (np.isnan(v[i]) and np.isnan(r[0]))) and kk[i] == r[1]


def varbinary_sequence(start, length):
    """TestVarBinaryMaxAggregation.getSequenceBlocks (…/operator/aggregation/TestVarBinaryMaxAggregation.java:33-41): the big-endian four
    bytes of start .. start + length - 1, aggregated as VARCHAR."""
    import struct
    return [struct.pack(">i", i) for i in range(start, start + length)]


# AbstractTestAggregationFunction's cases (…/operator/aggregation/AbstractTestAggregationFunction.java:84-140) with the answers
# TestVarBinaryMaxAggregation / TestVarBinaryMinAggregation.getExpectedValue compute: the natural order of Slices -- unsigned bytes, so
# the negative integers' 0xff.. images are the largest strings
VARBINARY_MINMAX_CASES = [
    ("testNoPositions", [], None, None),
    ("testSinglePosition", varbinary_sequence(0, 1), b"\x00\x00\x00\x00", b"\x00\x00\x00\x00"),
    ("testMultiplePositions", varbinary_sequence(0, 5), b"\x00\x00\x00\x00", b"\x00\x00\x00\x04"),
    ("testAllPositionsNull", [None] * 10, None, None),
    ("testMixedNullAndNonNullPositions", [v for x in varbinary_sequence(0, 10) for v in (None, x)], b"\x00\x00\x00\x00", b"\x00\x00\x00\x09"),
    ("testNegativeOnlyValues", varbinary_sequence(-10, 5), b"\xff\xff\xff\xf6", b"\xff\xff\xff\xfa"),
    ("testPositiveOnlyValues", varbinary_sequence(2, 4), b"\x00\x00\x00\x02", b"\x00\x00\x00\x05"),
]


@pytest.mark.parametrize("case", VARBINARY_MINMAX_CASES, ids=[c[0] for c in VARBINARY_MINMAX_CASES])
def test_min_max_over_strings_kats(oracle, case):
    _, values, lo, hi = case
    agg = oracle.HashAggregation([abi.VARCHAR], [], [(abi.AGG_MIN, 0, abi.VARCHAR), (abi.AGG_MAX, 0, abi.VARCHAR)])
    if values:
        agg.add_page(Page([Block.varchar(values)], len(values)))
    assert agg.build_result().to_rows() == [(lo, hi)]
    # a value that mixes both signs (not one of the reference's cases): the unsigned order puts -1 on top
    mixed = oracle.HashAggregation([abi.VARCHAR], [], [(abi.AGG_MIN, 0, abi.VARCHAR), (abi.AGG_MAX, 0, abi.VARCHAR)])
    mixed.add_page(Page([Block.varchar(varbinary_sequence(-5, 10))], 10))
    assert mixed.build_result().to_rows() == [(b"\x00\x00\x00\x00", b"\xff\xff\xff\xff")]


# The reference's per-function tests over AbstractTestAggregationFunction's cases (…/operator/aggregation/AbstractTestAggregationFunction
# .java:84-140: getSequenceBlocks(start, length) = the integers start .. start + length - 1 in the function's type):
# TestCountColumnAggregation (length), TestLongSumAggregation / TestDoubleSumAggregation (the sum, NULL without input),
# TestLongAverageAggregation / TestDoubleAverageAggregation (sum / length), TestLongMinAggregation / TestLongMaxAggregation /
# TestDoubleMinAggregation / TestDoubleMaxAggregation / TestDateMaxAggregation / TestShortDecimalMaxAggregation (start, start + length - 1),
# TestBooleanMaxAggregation / TestBooleanMinAggregation, TestRealSumAggregation
SEQUENCE_CASES = [("testNoPositions", 0, 0, "plain"), ("testSinglePosition", 0, 1, "plain"), ("testMultiplePositions", 0, 5, "plain"),
                  ("testAllPositionsNull", 0, 0, "all_null"), ("testMixedNullAndNonNullPositions", 0, 10, "alternating"),
                  ("testNegativeOnlyValues", -10, 5, "plain"), ("testPositiveOnlyValues", 2, 4, "plain")]


@pytest.mark.parametrize("case", SEQUENCE_CASES, ids=[c[0] for c in SEQUENCE_CASES])
def test_aggregation_function_sequence_kats(oracle, case):
    _, start, length, shape = case
    seq = list(range(start, start + length))
    if shape == "all_null":
        values, nulls = [0] * 10, [True] * 10
    elif shape == "alternating":   # createAlternatingNullsBlock: a NULL in front of every value
        values, nulls = [v for x in seq for v in (0, x)], [n for _ in seq for n in (True, False)]
    else:
        values, nulls = seq, [False] * len(seq)
    n = len(values)
    nl = np.array(nulls, dtype=bool)
    D = abi.decimal(10, 5)
    blocks = [Block.bigint(np.array(values, dtype=np.int64), nl), Block.double(np.array(values, dtype=np.float64), nl),
              Block.date(np.array(values, dtype=np.int32), nl), Block.decimal(np.array(values, dtype=np.int64), nl),
              Block.boolean(np.array([v % 2 != 0 for v in values], dtype=bool), nl), Block.boolean(np.array([v % 2 == 0 for v in values], dtype=bool), nl),
              Block.real(np.array(values, dtype=np.float32), nl)]
    types = [abi.BIGINT, abi.DOUBLE, abi.DATE, D, abi.BOOLEAN, abi.BOOLEAN, abi.REAL]
    aggs = [(abi.AGG_COUNT, 0, abi.BIGINT), (abi.AGG_SUM, 0, abi.BIGINT), (abi.AGG_AVG, 0, abi.BIGINT), (abi.AGG_SUM, 1, abi.DOUBLE),
            (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_MIN, 0, abi.BIGINT), (abi.AGG_MAX, 0, abi.BIGINT), (abi.AGG_MIN, 1, abi.DOUBLE),
            (abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_MAX, 2, abi.DATE), (abi.AGG_MAX, 3, D), (abi.AGG_MAX, 4, abi.BOOLEAN), (abi.AGG_MIN, 5, abi.BOOLEAN),
            (abi.AGG_SUM, 6, abi.REAL)]
    agg = oracle.HashAggregation(types, [], aggs)
    if n:
        agg.add_page(Page(blocks, n))
    (row,) = agg.build_result().to_rows()
    if length == 0:
        assert row == (0,) + (None,) * 13
        return
    total, lo, hi = sum(seq), start, start + length - 1
    # TestBooleanMaxAggregation (= TestBooleanOrAggregation over false, true, false, ...), TestBooleanMinAggregation (= ...And over true,
    # false, ...), TestRealSumAggregation (the float sum)
    bool_max, bool_min = length > 1 or start % 2 == 1, not (length > 1 or start % 2 == 1)
    assert row == (length, total, float(total) / length, float(total), float(total) / length, lo, hi, float(lo), float(hi), hi, hi,
                   bool_max, bool_min, float(np.float32(total)))


def test_lookup_join_page_builder_positions(oracle):
    """TestLookupJoinPageBuilder.testDifferentPositions (…/join/TestLookupJoinPageBuilder.java:85-150) at the operator's level: a probe page
    of 0 .. 99 against the build page 0 .. 99 -- no probe row joined gives an empty page; every second probe row joined gives probe and
    build columns 0, 2, 4, ...; every row joined gives both columns 0 .. 99, in probe order."""
    build = Page([Block.bigint(np.arange(100, dtype=np.int64))], 100)
    j = oracle.HashJoin([abi.BIGINT], [0], [0])
    j.add_build_page(build)
    j.build()
    # "empty": no probe key exists on the build side
    out, pi, bi = j.probe(Page([Block.bigint(np.arange(100, dtype=np.int64) + 1000)], 100), [abi.BIGINT], [0], [0], -1)
    assert out is None or out.position_count == 0
    # "the probe covers non-sequential positions": odd positions carry keys the build side does not hold
    keys = np.arange(100, dtype=np.int64)
    keys[1::2] += 1000
    out, pi, bi = j.probe(Page([Block.bigint(keys)], 100), [abi.BIGINT], [0], [0], -1)
    assert out.to_rows() == [(2 * i, 2 * i) for i in range(50)] and pi.tolist() == list(range(0, 100, 2)) and bi.tolist() == list(range(0, 100, 2))
    # "the probe covers everything"
    out, pi, bi = j.probe(build, [abi.BIGINT], [0], [0], -1)
    assert out.to_rows() == [(i, i) for i in range(100)] and pi.tolist() == bi.tolist() == list(range(100))
