"""GPU parity of the FilterAndProject operator against the oracle's PageProcessor restatement, including the
reference's own known-answer cases.  Row selection and every projected value are
bit-exact (projections are compiled with -ffp-contract=off, as Java evaluates them unfused)."""
import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd.expr import and_, coalesce, constant, field, if_, not_, or_
from presto_amd.operators import FilterAndProjectOperator, to_pages, upload_page
from presto_amd.page import Block, Page, sequence_page

pytestmark = pytest.mark.gpu


def run(op, pages):
    out = to_pages(op, pages)
    rows = [r for p in out for r in p.to_rows()]
    return out, rows


def oracle_rows(oracle, pages, f, projections):
    rows = []
    for p in pages:
        o = oracle.filter_project(p, f, projections)
        if o is not None:
            rows += o.to_rows()
    return rows


def bits(rows):
    """float -> raw bits so that comparisons are bit-exact (and NaN-safe)"""
    return [tuple(np.float64(v).view(np.int64).item() if isinstance(v, float) else v for v in r) for r in rows]


def test_fp1_filter_and_project_kat(gpu, oracle):
    """TestFilterAndProjectOperator.test (…/TestFilterAndProjectOperator.java:78-124), the reference's literal expressions:
    (VARCHAR, BIGINT) sequence page (100, 0, 0), filter LESS_THAN_OR_EQUAL(c1, 9), projections c0 and ADD(c1, 5)."""
    page = sequence_page(100, [(abi.VARCHAR, 0), (abi.BIGINT, 0)])
    f = field(1, abi.BIGINT) <= 9
    proj = [field(0, abi.VARCHAR), field(1, abi.BIGINT) + 5]
    op = FilterAndProjectOperator([abi.VARCHAR, abi.BIGINT], f, proj)
    out, rows = run(op, [page])
    assert rows == [(str(i).encode(), i + 5) for i in range(10)]
    assert rows == oracle_rows(oracle, [page], f, proj)
    assert op.selectedPositions()[1].tolist() == list(range(10))


def test_pp_partial_all_none(gpu, oracle):
    """TestPageProcessor partial / all / none filters over seq 0..99 (…/project/TestPageProcessor.java:90-200)."""
    page = sequence_page(100, [(abi.BIGINT, 0)])
    c0 = field(0, abi.BIGINT)
    # range(25, 50): positions 25..74
    op = FilterAndProjectOperator([abi.BIGINT], and_(c0 >= 25, c0 < 75), [c0])
    out, rows = run(op, [page])
    assert [r[0] for r in rows] == list(range(25, 75))
    is_list, pos = op.selectedPositions()
    assert is_list and pos.tolist() == list(range(25, 75))
    # all rows: positionsRange(0, 100), identity projection is the input block itself
    op = FilterAndProjectOperator([abi.BIGINT], c0 >= 0, [c0])
    out, rows = run(op, [page])
    assert [r[0] for r in rows] == list(range(100))
    assert op.selectedPositions() == (False, 100)
    # no rows: no output page at all
    op = FilterAndProjectOperator([abi.BIGINT], c0 < 0, [c0])
    out, rows = run(op, [page])
    assert out == [] and op.selectedPositions() == (False, 0)
    # no projections: a channel-less page carrying only the selected count
    op = FilterAndProjectOperator([abi.BIGINT], and_(c0 >= 25, c0 < 75), [])
    out = to_pages(op, [page])
    assert len(out) == 1 and out[0].position_count == 50 and out[0].channel_count == 0


@pytest.mark.parametrize("n", [1, 3, 1023, 1024, 1025, 5000, 100003])
@pytest.mark.parametrize("device_input", [False, True])
def test_q6_shape_positions_and_projection_bit_exact(gpu, oracle, n, device_input):
    cols = [oracle.tpch_column(c, 0.1, 7, n)[0] for c in tpch.Q6_COLUMNS]
    host = Page([Block.flat(t, v) for t, v in zip(tpch.Q6_TYPES, cols)], n)
    page = upload_page(host) if device_input else host
    op = FilterAndProjectOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections() + [field(0, abi.DATE)])
    out, rows = run(op, [page])
    expected = oracle_rows(oracle, [host], tpch.q6_filter(), tpch.q6_projections() + [field(0, abi.DATE)])
    assert bits(rows) == bits(expected)
    o_is_list, o_pos = oracle.filter_positions(host, tpch.q6_filter())
    is_list, pos = op.selectedPositions()
    assert is_list == o_is_list
    if is_list:
        assert np.array_equal(pos, o_pos)
    else:
        assert pos == o_pos


def test_q1_shape_projections_bit_exact_with_varchar(gpu, oracle):
    n = 70001
    blocks = []
    for c in tpch.Q1_COLUMNS:
        v, o = oracle.tpch_column(c, 0.1, 0, n)
        t = abi.TPCH_COLUMN_TYPE[c]
        blocks.append(Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v))
    host = Page(blocks, n)
    # make the filter selective enough to exercise the VARCHAR gather
    f = and_(tpch.q1_filter(), field(2, abi.DOUBLE) < constant(30.0, abi.DOUBLE))
    op = FilterAndProjectOperator(tpch.Q1_TYPES, f, tpch.q1_projections())
    out, rows = run(op, [host, host.get_region(100, 5001)])
    expected = oracle_rows(oracle, [host, host.get_region(100, 5001)], f, tpch.q1_projections())
    assert bits(rows) == bits(expected)


def test_null_logic_and_or_not_if_coalesce_in_between(gpu, oracle):
    rng = np.random.default_rng(11)
    n = 20011
    a = Block.bigint(rng.integers(-50, 50, n), rng.random(n) < 0.2)
    b = Block.double(rng.integers(-5, 5, n).astype(np.float64) / 4, rng.random(n) < 0.2)
    c = Block.boolean(rng.random(n) < 0.5, rng.random(n) < 0.3)
    d = Block.integer(rng.integers(-1000, 1000, n), rng.random(n) < 0.1)
    page = Page([a, b, c, d], n)
    types = [abi.BIGINT, abi.DOUBLE, abi.BOOLEAN, abi.INTEGER]
    fa, fb, fc, fd = field(0, abi.BIGINT), field(1, abi.DOUBLE), field(2, abi.BOOLEAN), field(3, abi.INTEGER)
    filters = [
        and_(fa > 0, or_(fb < constant(0.5, abi.DOUBLE), fc), not_(fd.eq(7))),
        or_(fa.is_null_(), fb.between(constant(-0.5, abi.DOUBLE), constant(0.75, abi.DOUBLE))),
        fa.isin(1, 2, 3, constant(None, abi.BIGINT)),
        not_(fc),
        if_(fc, fa > 10, fb > constant(0.0, abi.DOUBLE)),
    ]
    proj = [fa * 3 - 7, fb * fb + constant(1.5, abi.DOUBLE), coalesce(fa, constant(-1, abi.BIGINT)), fc, fd + 1,
            if_(fa > 0, fa, -fa), fa.cast(abi.DOUBLE) * fb, fd.cast(abi.BIGINT) + fa, fa % 7, fb / constant(3.0, abi.DOUBLE)]
    for f in filters:
        op = FilterAndProjectOperator(types, f, proj)
        out, rows = run(op, [page])
        expected = oracle_rows(oracle, [page], f, proj)
        assert bits(rows) == bits(expected)
        op.close()


def test_varchar_comparisons(gpu, oracle):
    segs = [b"AUTOMOBILE", b"BUILDING", b"FURNITURE", b"HOUSEHOLD", b"MACHINERY", b"", None]
    rng = np.random.default_rng(5)
    n = 5003
    vals = [segs[i] for i in rng.integers(0, len(segs), n)]
    page = Page([Block.varchar(vals), Block.bigint(np.arange(n))], n)
    s = field(0, abi.VARCHAR)
    for f in [s.eq(constant("BUILDING", abi.VARCHAR)), s < constant("FURNITURE", abi.VARCHAR), s.ne(constant("", abi.VARCHAR)),
              s.isin("MACHINERY", "AUTOMOBILE"), s >= constant("HOUSEHOLD", abi.VARCHAR)]:
        op = FilterAndProjectOperator([abi.VARCHAR, abi.BIGINT], f, [field(1, abi.BIGINT), s])
        out, rows = run(op, [page])
        assert rows == oracle_rows(oracle, [page], f, [field(1, abi.BIGINT), s])
        op.close()


def test_errors_surface_like_the_reference(gpu):
    from presto_amd._lib import PrestoAmdError
    page = Page([Block.bigint([1, 2, 0, 4]), Block.bigint([2 ** 62, 5, 6, 7])])
    a, b = field(0, abi.BIGINT), field(1, abi.BIGINT)
    op = FilterAndProjectOperator([abi.BIGINT, abi.BIGINT], None, [b / a])
    with pytest.raises(PrestoAmdError) as e:
        to_pages(op, [page])
    assert e.value.status == abi.ERR_DIVISION_BY_ZERO
    op = FilterAndProjectOperator([abi.BIGINT, abi.BIGINT], None, [b * 4])
    with pytest.raises(PrestoAmdError) as e:
        to_pages(op, [page])
    assert e.value.status == abi.ERR_NUMERIC_VALUE_OUT_OF_RANGE
    # AND short-circuits left to right: a row excluded by an earlier conjunct must not raise (AndCodeGenerator.java:63-71)
    op = FilterAndProjectOperator([abi.BIGINT, abi.BIGINT], and_(a.ne(0), (b / a) > 0), [a])
    rows = [r for p in to_pages(op, [page]) for r in p.to_rows()]
    assert rows == [(1,), (2,), (4,)]


def test_dictionary_and_rle_inputs(gpu, oracle):
    d = Block.bigint([10, 20, 30, 40], [0, 0, 1, 0])
    ids = np.array([3, 2, 1, 0, 0, 1, 2, 3, 3, 3], dtype=np.int32)
    page = Page([Block.dictionary_block(d, ids), Block.rle(Block.double([2.5]), 10)], 10)
    f = field(0, abi.BIGINT) >= 20
    proj = [field(0, abi.BIGINT), field(1, abi.DOUBLE) * constant(2.0, abi.DOUBLE)]
    op = FilterAndProjectOperator([abi.BIGINT, abi.DOUBLE], f, proj)
    rows = [r for p in to_pages(op, [page]) for r in p.to_rows()]
    assert rows == oracle_rows(oracle, [page], f, proj)


def test_varchar_dictionary_and_rle_inputs(gpu, oracle):
    """DictionaryBlock / RunLengthEncodedBlock over a VariableWidthBlock (what ORC / Parquet readers produce for strings):
    filter on the string, project it, group by it."""
    from presto_amd.operators import HashAggregationOperator
    rng = np.random.default_rng(8)
    n = 20000
    d = Block.varchar([b"BUILDING", b"", None, b"AUTOMOBILE", b"x", b"MACHINERY"])
    ids = rng.integers(0, 6, n).astype(np.int32)
    page = Page([Block.dictionary_block(d, ids), Block.rle(Block.varchar([b"const"]), n), Block.bigint(np.arange(n))], n)
    types = [abi.VARCHAR, abi.VARCHAR, abi.BIGINT]
    f = or_(field(0, abi.VARCHAR).eq(constant(b"BUILDING", abi.VARCHAR)), field(0, abi.VARCHAR) > constant(b"M", abi.VARCHAR))
    proj = [field(0, abi.VARCHAR), field(1, abi.VARCHAR), field(2, abi.BIGINT)]
    rows = [r for p in to_pages(FilterAndProjectOperator(types, f, proj), [page]) for r in p.to_rows()]
    assert rows == oracle_rows(oracle, [page], f, proj) and len(rows) > 1000
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 2, abi.BIGINT)]
    got = sorted((r for p in to_pages(HashAggregationOperator(types, [0, 1], aggs), [page]) for r in p.to_rows()), key=repr)
    ref = oracle.HashAggregation(types, [0, 1], aggs)
    ref.add_page(page)
    assert got == sorted(ref.build_result().to_rows(), key=repr) and len(got) == 6


def test_dictionary_aware_filter(gpu, oracle):
    """DictionaryAwarePageFilter (…/operator/project/DictionaryAwarePageFilter.java:57-142): a filter over one channel whose
    block is a DictionaryBlock / RLE is evaluated on the dictionary and looked up through the ids.  Same rows as the plain
    filter: filter channel projected or not, NULL entries, RLE (all rows / no row), a dictionary larger than the page
    (ordinary path), an entry no row uses that would fail the filter expression (:105-111), one that a row does use."""
    from presto_amd._lib import PrestoAmdError
    rng = np.random.default_rng(12)
    n = 50000
    d = Block.bigint(np.arange(100) * 3 - 50, (np.arange(100) % 17 == 0))
    ids = rng.integers(0, 100, n).astype(np.int32)
    other = Block.double(rng.random(n))
    x, y = field(0, abi.BIGINT), field(1, abi.DOUBLE)
    types = [abi.BIGINT, abi.DOUBLE]
    page = Page([Block.dictionary_block(d, ids), other], n)
    for f in (x > 40, x.between(constant(-20, abi.BIGINT), constant(100, abi.BIGINT)), x > 10 ** 9, x > -10 ** 9):
        for proj in ([y * constant(3.0, abi.DOUBLE)], [x + 1, y], [x]):
            op = FilterAndProjectOperator(types, f, proj)
            rows = [r for p in to_pages(op, [page]) for r in p.to_rows()]
            assert rows == oracle_rows(oracle, [page], f, proj)
    # selectedPositions: a range when no or every row passes (PageFilter.java:37-39), a list otherwise
    op = FilterAndProjectOperator(types, x > 40, [y])
    op.addInput(page)
    out = op.getOutput()
    is_list, pos = op.selectedPositions()
    assert is_list and pos.tolist() == [i for i, v in enumerate(page.blocks[0].to_pylist()) if v is not None and v > 40]
    assert out.position_count == len(pos)
    # RLE filter channel
    for value, expect in ((Block.bigint([7]), n), (Block.bigint([-7]), 0), (Block.bigint([0], [1]), 0)):
        p = Page([Block.rle(value, n), other], n)
        rows = [r for q in to_pages(FilterAndProjectOperator(types, x > 0, [y, x]), [p]) for r in q.to_rows()]
        assert len(rows) == expect and rows == oracle_rows(oracle, [p], x > 0, [y, x])
    # dictionary larger than the page: processed row by row
    small = Page([Block.dictionary_block(d, ids[:30]), Block.double(rng.random(30))], 30)
    rows = [r for q in to_pages(FilterAndProjectOperator(types, x > 40, [x, y]), [small]) for r in q.to_rows()]
    assert rows == oracle_rows(oracle, [small], x > 40, [x, y])
    # an unused entry that fails the expression: the dictionary pass is dropped, the rows decide
    dz = Block.bigint([0, 1, 2, 5])
    f = constant(10, abi.BIGINT) / x > 3
    ok = Page([Block.dictionary_block(dz, rng.integers(1, 4, 1000).astype(np.int32)), Block.double(rng.random(1000))], 1000)
    rows = [r for q in to_pages(FilterAndProjectOperator(types, f, [x]), [ok]) for r in q.to_rows()]
    assert rows == oracle_rows(oracle, [ok], f, [x]) and len(rows) > 0
    bad = Page([Block.dictionary_block(dz, rng.integers(0, 4, 1000).astype(np.int32)), Block.double(rng.random(1000))], 1000)
    with pytest.raises(PrestoAmdError) as e:
        to_pages(FilterAndProjectOperator(types, f, [x]), [bad])
    assert e.value.status == abi.ERR_DIVISION_BY_ZERO


def test_dictionary_aware_filter_on_device_pages(gpu, oracle):
    """The same with the page resident in HBM (ids, dictionary and the other channel are device pointers), VARCHAR dictionary."""
    from presto_amd.operators import download_page, upload_page
    rng = np.random.default_rng(13)
    n = 200000
    words = [b"AIR", b"MAIL", b"SHIP", None, b"TRUCK", b"REG AIR", b"FOB", b"RAIL"]
    page = Page([Block.dictionary_block(Block.varchar(words), rng.integers(0, len(words), n).astype(np.int32)), Block.bigint(rng.integers(0, 1000, n))], n)
    types = [abi.VARCHAR, abi.BIGINT]
    m = field(0, abi.VARCHAR)
    f = or_(m.eq(constant(b"MAIL", abi.VARCHAR)), m.eq(constant(b"SHIP", abi.VARCHAR)))
    for proj in ([field(1, abi.BIGINT) * 2], [m, field(1, abi.BIGINT)]):
        op = FilterAndProjectOperator(types, f, proj, output_mem=abi.MEM_DEVICE)
        rows = [r for p in to_pages(op, [upload_page(page)]) for r in download_page(p).to_rows()]
        assert rows == oracle_rows(oracle, [page], f, proj)


# ---- MergePages behind the PageProcessor (a7) ------------------------------------------------------------------------
def merged_reference(oracle, pages, f, projections, min_bytes, min_rows, max_bytes=0):
    m = oracle.MergePages(min_bytes, min_rows, max_bytes)
    out = []
    for p in pages:
        q = oracle.filter_project(p, f, projections)
        if q is not None and q.position_count > 0:
            out += m.process(q)
    return [p.to_rows() for p in out + m.finish()]


@pytest.mark.parametrize("device_output", [False, True])
def test_merge_pages_kats_on_device(gpu, oracle, device_output):
    """TestMergePages.java:44-163 through FilterAndProjectOperator(minOutputPageSize, minOutputPageRowCount): identity
    projections, no filter, so the PageProcessor hands the input pages to MergePages unchanged."""
    from presto_amd.operators import download_page
    types = [abi.VARCHAR, abi.BIGINT, abi.DOUBLE]
    cols = [(abi.VARCHAR, 0), (abi.BIGINT, 0), (abi.DOUBLE, 0)]
    proj = [field(i, t) for i, t in enumerate(types)]
    mem = abi.MEM_DEVICE if device_output else abi.MEM_HOST

    def go(pages, min_bytes, min_rows, max_bytes=0, ptypes=types, pproj=proj):
        op = FilterAndProjectOperator(ptypes, None, pproj, output_mem=mem, min_output_page_size=min_bytes, min_output_page_row_count=min_rows,
                                      max_output_page_size=max_bytes)
        got = []
        for p in pages:  # Driver order: addInput, then getOutput until none
            assert op.needsInput()
            op.addInput(p)
            while True:
                o = op.getOutput()
                if o is None:
                    break
                got.append((download_page(o) if device_output else o).to_rows())
        op.finish()
        while not op.isFinished():
            o = op.getOutput()
            if o is not None:
                got.append((download_page(o) if device_output else o).to_rows())
        assert got == merged_reference(oracle, pages, None, pproj, min_bytes, min_rows, max_bytes)
        return got

    page = sequence_page(10, cols)
    assert go([page], oracle.page_size_in_bytes(page), 2 ** 31 - 1) == [page.to_rows()]
    assert go([page], 1024 * 1024, 10) == [page.to_rows()]
    whole = sequence_page(20, cols)
    assert go([whole.get_region(0, 10), whole.get_region(10, 10)], oracle.page_size_in_bytes(whole) + 1, 21) == [whole.to_rows()]
    small, big = sequence_page(10, cols), sequence_page(100, cols)
    assert go([small, big], oracle.page_size_in_bytes(big), 100) == [small.to_rows(), big.to_rows()]
    w1 = sequence_page(20, [(abi.BIGINT, 0)])
    halves = [w1.get_region(0, 10), w1.get_region(10, 10)]
    size = oracle.page_size_in_bytes(w1)
    assert go(halves + halves, size // 2 + 1, 11, size, [abi.BIGINT], [field(0, abi.BIGINT)]) == [w1.to_rows(), w1.to_rows()]


def test_merge_pages_selective_filter_with_nulls_and_varchar(gpu, oracle):
    """The shape MergePages exists for: a selective filter over many small pages (8192-row connector pages, ~2 % pass);
    defaults of the session properties (500 kB / 256 rows, SystemSessionProperties.java:106-107) and a small maximum so
    that full-buffer flushes happen; nullable and VARCHAR columns; expressions and identity projections mixed."""
    rng = np.random.default_rng(11)
    words = [b"", b"a", b"BUILDING", b"MACHINERY", None]
    pages = []
    for i in range(40):
        n = int(rng.integers(1, 4000))
        pages.append(Page([Block.bigint(rng.integers(0, 1000, n), rng.random(n) < 0.1), Block.double(rng.random(n)),
                           Block.varchar([words[j] for j in rng.integers(0, len(words), n)]), Block.integer(rng.integers(-5, 5, n))], n))
    types = [abi.BIGINT, abi.DOUBLE, abi.VARCHAR, abi.INTEGER]
    f = field(1, abi.DOUBLE) < constant(0.03, abi.DOUBLE)
    proj = [field(0, abi.BIGINT), field(2, abi.VARCHAR), field(1, abi.DOUBLE) * constant(2.0, abi.DOUBLE), field(3, abi.INTEGER)]
    for min_bytes, min_rows, max_bytes in [(500 * 1000, 256, 0), (3000, 120, 6000), (1, 1, 0)]:
        op = FilterAndProjectOperator(types, f, proj, min_output_page_size=min_bytes, min_output_page_row_count=min_rows,
                                      max_output_page_size=max_bytes)
        got = [p.to_rows() for p in to_pages(op, pages)]
        expected = merged_reference(oracle, pages, f, proj, min_bytes, min_rows, max_bytes)
        assert len(got) == len(expected)
        assert bits([r for p in got for r in p]) == bits([r for p in expected for r in p])
        assert [len(p) for p in got] == [len(p) for p in expected]


def test_fp2_merge_output_kat(gpu):
    """TestFilterAndProjectOperator.testMergeOutput (…/TestFilterAndProjectOperator.java:126-161)"""
    page = sequence_page(100, [(abi.VARCHAR, 0), (abi.BIGINT, 0)])
    op = FilterAndProjectOperator([abi.VARCHAR, abi.BIGINT], field(1, abi.BIGINT).eq(10), [field(1, abi.BIGINT)],
                                  min_output_page_size=64 * 1024, min_output_page_row_count=2)
    out = to_pages(op, [page] * 4)
    assert [p.to_rows() for p in out] == [[(10,)] * 4]
