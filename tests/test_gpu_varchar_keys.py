"""VARCHAR group keys of any length (MultiChannelGroupByHash compares the bytes, …/MultiChannelGroupByHash.java:441-452).
Keys whose declared bound fits 15 bytes travel inside the packed key words; every other VARCHAR key is interned on the
device (string -> dense id, byte-exact) and grouped by id.  These tests drive the interned path through every tier of
the aggregation, with NULLs, empty strings, strings that differ only in their last byte or only in length, PARTIAL /
FINAL, the $hashvalue channel and device-resident output -- all against the oracle."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.exchange import partial_layout
from presto_amd.operators import FusedAggregationOperator, HashAggregationOperator, download_page, to_pages
from presto_amd.page import Block, Page
from presto_amd.expr import constant, field

pytestmark = pytest.mark.gpu

TRICKY = [b"", b"a", b"a\0", b"a\0\0", b"12345678", b"12345678\0", b"123456789", b"1234567812345678", b"1234567812345679",
          b"1234567812345678x", b"Customer#000000001 special requests", b"Customer#000000001 special requestt",
          b"Customer#000000001 special request", "zażółć gęślą jaźń".encode(), b"\xff" * 40, b"\xff" * 41, None]


def long_keys(rng, n, card, null_share=0.01):
    ids = rng.integers(0, card, n)
    base = [("supplier#%09d--%s" % (i, "x" * (i % 23))).encode() for i in range(card)]
    keys = [base[i] for i in ids]
    for i in np.nonzero(rng.random(n) < null_share)[0]:
        keys[i] = None
    return keys


def run_both(oracle, types, keys, aggs, pages, **kw):
    op = HashAggregationOperator(types, keys, aggs, **kw)
    got = [r for p in to_pages(op, pages) for r in p.to_rows()]
    ref = oracle.HashAggregation(types, keys, aggs, hash_channel=kw.get("hash_channel", -1))
    for p in pages:
        ref.add_page(p)
    return got, ref.build_result().to_rows()


def assert_same(got, expected, nkeys=1):
    assert len(got) == len(expected)
    g = {r[:nkeys]: r for r in got}
    e = {r[:nkeys]: r for r in expected}
    assert len(g) == len(got) and set(g) == set(e)
    for k, er in e.items():
        for gv, ev in zip(g[k][nkeys:], er[nkeys:]):
            if isinstance(ev, float):
                assert gv == ev or abs(gv - ev) <= 1e-9 * max(abs(gv), abs(ev)), (k, g[k], er)  # DOUBLE sums: DESIGN tolerance
            else:
                assert gv == ev, (k, g[k], er)


def test_tricky_strings_are_distinct_groups(gpu, oracle):
    rng = np.random.default_rng(1)
    n = 50000
    pick = rng.integers(0, len(TRICKY), n)
    page = Page([Block.varchar([TRICKY[i] for i in pick]), Block.bigint(rng.integers(0, 100, n))], n)
    types = [abi.VARCHAR, abi.BIGINT]
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT, 0, abi.VARCHAR)]
    got, expected = run_both(oracle, types, [0], aggs, [page, page])
    assert len(expected) == len(TRICKY)
    assert_same(got, expected)


@pytest.mark.parametrize("card,rows,pages", [(6, 30000, 2), (300, 100000, 2), (20000, 150000, 3), (400000, 500000, 2)])
def test_long_keys_through_every_tier(gpu, oracle, card, rows, pages):
    rng = np.random.default_rng(card)
    plist = []
    for _ in range(pages):
        plist.append(Page([Block.varchar(long_keys(rng, rows, card)), Block.double(rng.random(rows), rng.random(rows) < 0.1),
                           Block.bigint(rng.integers(-1000, 1000, rows))], rows))
    types = [abi.VARCHAR, abi.DOUBLE, abi.BIGINT]
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 2, abi.BIGINT), (abi.AGG_MIN, 2, abi.BIGINT),
            (abi.AGG_COUNT, 0, abi.VARCHAR)]
    got, expected = run_both(oracle, types, [0], aggs, plist, expected_groups=card)
    assert_same(got, expected)


@pytest.mark.parametrize("card", [50, 100000])
def test_hash_channel_and_second_key(gpu, oracle, card):
    """$hashvalue of an interned key is the hash of the string (VarcharType.hash = XxHash64 of the bytes), combined with
    the other key's hash as InterpretedHashGenerator does."""
    rng = np.random.default_rng(card + 5)
    rows = 200000
    plist = []
    for _ in range(2):
        p = Page([Block.varchar(long_keys(rng, rows, card)), Block.integer(rng.integers(0, 3, rows)), Block.bigint(rng.integers(0, 10, rows))], rows)
        plist.append(Page(p.blocks + [Block.bigint(oracle.hash_page(p, [0, 1]))], rows))
    types = [abi.VARCHAR, abi.INTEGER, abi.BIGINT, abi.BIGINT]
    aggs = [(abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None)]
    got, expected = run_both(oracle, types, [0, 1], aggs, plist, hash_channel=3, expected_groups=card * 3)
    assert_same(got, expected, nkeys=2)
    for r in got[:2000]:
        h = 0 if r[0] is None else oracle._s64(oracle.xxh64(r[0]))
        assert r[2] == oracle.combine_hash(oracle.combine_hash(0, h), oracle.hash_integer(r[1]))


@pytest.mark.parametrize("card", [40, 30000])
def test_partial_final_with_long_keys(gpu, oracle, card):
    rng = np.random.default_rng(card + 9)
    rows = 120000
    plist = [Page([Block.varchar(long_keys(rng, rows, card)), Block.double(rng.random(rows))], rows) for _ in range(3)]
    types = [abi.VARCHAR, abi.DOUBLE]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MAX, 1, abi.DOUBLE)]
    ref = oracle.HashAggregation(types, [0], aggs)
    for p in plist:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    ptypes, faggs = partial_layout([abi.VARCHAR], aggs)
    partial_pages = []
    for p in plist:
        partial_pages += to_pages(HashAggregationOperator(types, [0], aggs, step=abi.STEP_PARTIAL, expected_groups=card), [p])
    assert all(p.blocks[0].type == abi.VARCHAR for p in partial_pages)
    final = HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL, expected_groups=card)
    got = [r for p in to_pages(final, partial_pages) for r in p.to_rows()]
    assert_same(got, expected)


def test_declared_long_bound_and_fused_filter(gpu, oracle):
    """VARCHAR(40) keys behind a filter and projections on other channels (the fused ScanFilterAndProject -> HashAggregation);
    a long key that the filter itself reads stays with the Java operators."""
    rng = np.random.default_rng(77)
    rows, card = 200000, 1500
    page = Page([Block.varchar(long_keys(rng, rows, card, 0.0)), Block.double(rng.random(rows) * 10), Block.bigint(rng.integers(0, 50, rows))], rows)
    types = [abi.VARCHAR, abi.DOUBLE, abi.BIGINT]
    x, q = field(1, abi.DOUBLE), field(2, abi.BIGINT)
    filt = q < constant(25, abi.BIGINT)
    proj = [field(0, abi.VARCHAR), x * constant(2.0, abi.DOUBLE), q + 1]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None)]
    op = FusedAggregationOperator(types, filt, proj, [0], aggs, type_params=[40, 0, 0])
    got = [r for p in to_pages(op, [page]) for r in p.to_rows()]
    fp = oracle.filter_project(page, filt, proj)
    ref = oracle.HashAggregation([abi.VARCHAR, abi.DOUBLE, abi.BIGINT], [0], aggs)
    ref.add_page(fp)
    assert_same(got, ref.build_result().to_rows())
    from presto_amd._lib import PrestoAmdError
    with pytest.raises(PrestoAmdError) as e:
        FusedAggregationOperator(types, field(0, abi.VARCHAR).eq(constant("supplier#000000001--x", abi.VARCHAR)), proj, [0], aggs, type_params=[40, 0, 0])
    assert e.value.status == abi.ERR_NOT_SUPPORTED


def test_device_resident_output(gpu, oracle):
    rng = np.random.default_rng(3)
    rows, card = 300000, 50000
    page = Page([Block.varchar(long_keys(rng, rows, card)), Block.bigint(rng.integers(0, 9, rows))], rows)
    types = [abi.VARCHAR, abi.BIGINT]
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT)]
    op = HashAggregationOperator(types, [0], aggs, output_mem=abi.MEM_DEVICE, expected_groups=card)
    got = [r for p in to_pages(op, [page]) for r in download_page(p).to_rows()]
    ref = oracle.HashAggregation(types, [0], aggs)
    ref.add_page(page)
    assert_same(got, ref.build_result().to_rows())


@pytest.mark.parametrize("device", [False, True])
def test_dictionary_blocks_as_long_keys(gpu, oracle, device):
    """MultiChannelGroupByHash's dictionary path (…/MultiChannelGroupByHash.java:465-512): key pages that arrive as
    DictionaryBlock / RLE over strings -- a different dictionary per page, NULL entries, unused entries, then a plain
    VariableWidthBlock page with overlapping strings -- group exactly like their decoded form."""
    from presto_amd.operators import upload_page
    rng = np.random.default_rng(21)
    base = [("part-%05d-%s" % (i, "q" * (i % 29))).encode() for i in range(400)]
    pages = []
    for k in range(3):
        n = 60000 + 1000 * k
        entries = [base[i] for i in rng.choice(400, 150, replace=False)] + [None]
        d = Block.varchar(entries)
        ids = rng.integers(0, len(entries) - (k == 1), n).astype(np.int32)  # page 1 never uses its NULL entry
        pages.append(Page([Block.dictionary_block(d, ids), Block.bigint(rng.integers(0, 100, n))], n))
    pages.append(Page([Block.rle(Block.varchar([base[7]]), 5000), Block.bigint(rng.integers(0, 100, 5000))], 5000))
    pages.append(Page([Block.rle(Block.varchar([None]), 300), Block.bigint(rng.integers(0, 100, 300))], 300))
    plain = [base[i] for i in rng.integers(0, 400, 20000)]
    pages.append(Page([Block.varchar(plain), Block.bigint(rng.integers(0, 100, 20000))], 20000))
    types = [abi.VARCHAR, abi.BIGINT]
    aggs = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT, 0, abi.VARCHAR)]
    op = HashAggregationOperator(types, [0], aggs)
    got = [r for p in to_pages(op, [upload_page(p) for p in pages] if device else pages) for r in p.to_rows()]
    ref = oracle.HashAggregation(types, [0], aggs)
    for p in pages:
        ref.add_page(p)
    assert_same(got, ref.build_result().to_rows())


# ---- min / max over VARCHAR ------------------------------------------------------------------------------------------------------
SHORT = [b"", b"a", b"a\0", b"ab", b"b", b"REG AIR", b"RAIL", b"\xff", b"\xff\x00", b"\x80abc", b"zzzzzzz", b"zzzzzz", None]


@pytest.mark.parametrize("groups", [0, 3, 300, 40000])
def test_min_max_over_short_varchar(gpu, oracle, groups):
    """min / max over a channel declared VARCHAR(n), n <= 7: Slice.compareTo order (unsigned bytes, then length; a proper prefix
    first), NULL inputs skipped, NULL for groups without a value -- global, a few groups, and through the table tiers."""
    rng = np.random.default_rng(groups + 5)
    n = 150000
    pages = []
    for _ in range(3):
        strings = [SHORT[i] for i in rng.integers(0, len(SHORT), n)]
        key = rng.integers(0, max(groups, 1), n)
        if groups == 40000:   # most groups see one or two values only
            key = rng.integers(0, groups, n)
        pages.append(Page([Block.bigint(key), Block.varchar(strings), Block.double(rng.random(n))], n))
    types = [abi.BIGINT, abi.VARCHAR, abi.DOUBLE]
    aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR), (abi.AGG_COUNT, 1, abi.VARCHAR), (abi.AGG_SUM, 2, abi.DOUBLE)]
    keys = [0] if groups else []
    got, expected = run_both(oracle, types, keys, aggs, pages, type_params=[0, 7, 0], expected_groups=max(groups, 1))
    assert_same(got, expected, nkeys=len(keys))
    if groups == 3:
        assert all(r[1] == b"" and r[2] == b"\xff\x00" for r in got)   # unsigned bytes; a proper prefix (b"\xff") sorts first


def test_min_max_varchar_partial_final(gpu, oracle):
    rng = np.random.default_rng(3)
    n = 60000
    pages = [Page([Block.bigint(rng.integers(0, 50, n)), Block.varchar([SHORT[i] for i in rng.integers(0, len(SHORT), n)])], n) for _ in range(2)]
    types, aggs = [abi.BIGINT, abi.VARCHAR], [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR)]
    single, expected = run_both(oracle, types, [0], aggs, pages, type_params=[0, 7])
    assert_same(single, expected)
    ptypes, faggs = partial_layout([abi.BIGINT], aggs)
    assert ptypes == [abi.BIGINT, abi.BIGINT, abi.VARCHAR, abi.BIGINT, abi.VARCHAR]
    partial_pages = []
    for p in pages:
        out = to_pages(HashAggregationOperator(types, [0], aggs, step=abi.STEP_PARTIAL, type_params=[0, 7]), [p])
        ref = oracle.HashAggregation(types, [0], aggs, step=abi.STEP_PARTIAL)
        ref.add_page(p)
        assert sorted(out[0].to_rows(), key=lambda r: r[0]) == sorted(ref.build_result().to_rows(), key=lambda r: r[0])
        partial_pages += out
    final = [r for p in to_pages(HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL, type_params=[0, 0, 7, 0, 7]), partial_pages) for r in p.to_rows()]
    assert_same(final, expected)


def long_strings(rng, count, pool):
    """`count` strings out of `pool` distinct ones: lengths 0..40, shared prefixes (a proper prefix sorts first), bytes >= 0x80 (unsigned
    order), some NULL."""
    stems = [b"", b"a", b"customer#", b"customer#0000", b"\xff\xfe", b"\x80", b"special requests ", b"zz"]
    words = [None]
    for i in range(pool):
        stem = stems[i % len(stems)]
        words.append(stem + bytes(rng.integers(0, 256, int(rng.integers(0, 41 - len(stem)))).astype(np.uint8)))
    return [words[i] for i in rng.integers(0, len(words), count)]


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("groups", [0, 3, 300, 40000])
def test_min_max_over_long_varchar(gpu, oracle, groups, device):
    """min / max over VARCHAR channels without a bound of <= 7 bytes (undeclared, and VARCHAR(40)): the strings travel as ranks in the
    channel's dictionary.  Every page brings strings the pages before did not hold -- what was accumulated is re-ranked each time --,
    the last page only repeats known ones.  Slice.compareTo order, NULL inputs skipped, NULL for groups without a value; ungrouped, a
    few groups, many groups."""
    from presto_amd.operators import upload_page
    rng = np.random.default_rng(groups + 77)
    n = 60000
    pages = []
    for k in range(4):
        a = long_strings(rng, n, 50 if groups == 3 else 5000)
        b = long_strings(rng, n, 700)
        key = rng.integers(0, max(groups, 1), n)
        pages.append(Page([Block.bigint(key), Block.varchar(a), Block.double(rng.random(n)), Block.varchar(b)], n))
    pages.append(pages[1].get_region(100, 20000))
    types = [abi.BIGINT, abi.VARCHAR, abi.DOUBLE, abi.VARCHAR]
    aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR), (abi.AGG_COUNT, 1, abi.VARCHAR), (abi.AGG_SUM, 2, abi.DOUBLE),
            (abi.AGG_MAX, 3, abi.VARCHAR), (abi.AGG_COUNT_STAR, -1, None)]
    keys = [0] if groups else []
    inputs = [upload_page(p) for p in pages] if device else pages
    op = HashAggregationOperator(types, keys, aggs, type_params=[0, 0, 0, 40], expected_groups=max(groups, 1))
    got = [r for p in to_pages(op, inputs) for r in p.to_rows()]
    ref = oracle.HashAggregation(types, keys, aggs)
    for p in pages:
        ref.add_page(p)
    assert_same(got, ref.build_result().to_rows(), nkeys=len(keys))


@pytest.mark.parametrize("param", [0, 7])
def test_min_max_over_strings_reference_kats(gpu, param):
    """TestVarBinaryMaxAggregation / TestVarBinaryMinAggregation (max / min over VARCHAR values holding the big-endian bytes of an integer
    sequence) through AbstractTestAggregationFunction's cases, on the device: by rank (undeclared length) and by image (VARCHAR(7))."""
    from tests.test_oracle_operators import VARBINARY_MINMAX_CASES
    for name, values, lo, hi in VARBINARY_MINMAX_CASES:
        op = HashAggregationOperator([abi.VARCHAR], [], [(abi.AGG_MIN, 0, abi.VARCHAR), (abi.AGG_MAX, 0, abi.VARCHAR)], type_params=[param])
        pages = [Page([Block.varchar(values)], len(values))] if values else []
        assert [r for p in to_pages(op, pages) for r in p.to_rows()] == [(lo, hi)], name


def test_min_max_long_varchar_empty_and_all_null(gpu, oracle):
    types, aggs = [abi.BIGINT, abi.VARCHAR], [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR), (abi.AGG_COUNT_STAR, -1, None)]
    # no input at all: one row of NULLs for the ungrouped aggregation, no row for the grouped one
    assert [r for p in to_pages(HashAggregationOperator(types, [], aggs), []) for r in p.to_rows()] == [(None, None, 0)]
    assert [r for p in to_pages(HashAggregationOperator(types, [0], aggs), []) for r in p.to_rows()] == []
    # a group whose strings are all NULL, next to one holding the empty string
    page = Page([Block.bigint([1, 1, 2, 2]), Block.varchar([None, None, b"", b"a much longer string than seven bytes"])], 4)
    got, expected = run_both(oracle, types, [0], aggs, [page])
    assert_same(got, expected)
    assert sorted(got) == [(1, None, None, 2), (2, b"", b"a much longer string than seven bytes", 2)]


def test_min_max_long_varchar_partial_final(gpu, oracle):
    """PARTIAL emits [count, the string itself]; FINAL takes those state pages -- from several PARTIAL operators with dictionaries of
    their own -- and ranks the state strings in its own dictionary."""
    rng = np.random.default_rng(8)
    n = 40000
    pages = [Page([Block.bigint(rng.integers(0, 500, n)), Block.varchar(long_strings(rng, n, 3000))], n) for _ in range(3)]
    types, aggs = [abi.BIGINT, abi.VARCHAR], [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR)]
    single, expected = run_both(oracle, types, [0], aggs, pages)
    assert_same(single, expected)
    ptypes, faggs = partial_layout([abi.BIGINT], aggs)
    assert ptypes == [abi.BIGINT, abi.BIGINT, abi.VARCHAR, abi.BIGINT, abi.VARCHAR]
    partial_pages = []
    for p in pages:
        out = to_pages(HashAggregationOperator(types, [0], aggs, step=abi.STEP_PARTIAL), [p])
        ref = oracle.HashAggregation(types, [0], aggs, step=abi.STEP_PARTIAL)
        ref.add_page(p)
        assert sorted(out[0].to_rows(), key=lambda r: r[0]) == sorted(ref.build_result().to_rows(), key=lambda r: r[0])
        partial_pages += out
    final = [r for p in to_pages(HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL), partial_pages) for r in p.to_rows()]
    assert_same(final, expected)


@pytest.mark.parametrize("groups", [0, 200])
def test_min_max_long_varchar_page_shapes(gpu, oracle, groups):
    """The forms a page takes -- VariableWidthBlock without and with NULLs (the second changes the channel's layout: a new generation of
    the operator's state, combined at the end through the strings themselves), DictionaryBlock and RLE over strings, small device pages,
    device-resident output -- behind a filter and projections on other channels (the fused ScanFilterAndProject -> aggregation)."""
    from presto_amd.operators import upload_page
    rng = np.random.default_rng(groups + 31)

    def strings(n, with_null):
        v = long_strings(rng, n, 900)
        return v if with_null else [x if x is not None else b"never null here" for x in v]
    pages = []
    for k, n in enumerate((30000, 30000, 2000, 800, 25000)):
        key, q = rng.integers(0, max(groups, 1), n), rng.integers(0, 50, n)
        if k == 2:
            entries = strings(300, True)
            col = Block.dictionary_block(Block.varchar(entries), rng.integers(0, len(entries), n).astype(np.int32))
        elif k == 3:
            col = Block.rle(Block.varchar([b"the one string of this run-length page"]), n)
        else:
            col = Block.varchar(strings(n, k >= 1))
        pages.append(Page([Block.bigint(key), col, Block.bigint(q)], n))
    types = [abi.BIGINT, abi.VARCHAR, abi.BIGINT]
    q = field(2, abi.BIGINT)
    filt = q < constant(40, abi.BIGINT)
    proj = [field(0, abi.BIGINT), field(1, abi.VARCHAR), q + 1]
    aggs = [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR), (abi.AGG_COUNT, 1, abi.VARCHAR), (abi.AGG_SUM, 2, abi.BIGINT)]
    keys = [0] if groups else []
    ref = oracle.HashAggregation(types, keys, aggs)
    for p in pages:
        ref.add_page(oracle.filter_project(p, filt, proj))
    expected = ref.build_result().to_rows()
    inputs = [pages[0], pages[1], pages[2], pages[3]] + [upload_page(pages[4].get_region(i, min(5000, 25000 - i))) for i in range(0, 25000, 5000)]
    for mem in (abi.MEM_HOST, abi.MEM_DEVICE):
        op = FusedAggregationOperator(types, filt, proj, keys, aggs, output_mem=mem)
        got = [r for p in to_pages(op, inputs) for r in (download_page(p) if mem == abi.MEM_DEVICE else p).to_rows()]
        assert_same(got, expected, nkeys=len(keys))


def test_min_max_long_varchar_partial_flushes(gpu, oracle):
    """Step.PARTIAL under maxPartialMemory: the operator flushes its groups whenever they pass the budget -- each flush hands the strings
    out and starts over; the FINAL step over all flushed pages gives the SINGLE result."""
    rng = np.random.default_rng(12)
    n = 30000
    pages = [Page([Block.bigint(rng.integers(0, 3000, n)), Block.varchar(long_strings(rng, n, 2000))], n) for _ in range(5)]
    types, aggs = [abi.BIGINT, abi.VARCHAR], [(abi.AGG_MIN, 1, abi.VARCHAR), (abi.AGG_MAX, 1, abi.VARCHAR)]
    ref = oracle.HashAggregation(types, [0], aggs)
    for p in pages:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    ptypes, faggs = partial_layout([abi.BIGINT], aggs)
    partial = HashAggregationOperator(types, [0], aggs, step=abi.STEP_PARTIAL, max_partial_memory=64 << 10)
    flushed = to_pages(partial, pages)
    assert len(flushed) >= 3
    final = [r for p in to_pages(HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL), flushed) for r in p.to_rows()]
    assert_same(final, expected)


def test_min_max_varchar_outside_the_device_subset(gpu):
    from presto_amd._lib import PrestoAmdError
    types = [abi.BIGINT, abi.VARCHAR]
    # the long string is also the group key, or an expression reads it: the strings themselves would be needed next to their ranks
    with pytest.raises(PrestoAmdError) as e:
        HashAggregationOperator([abi.VARCHAR, abi.BIGINT], [0], [(abi.AGG_MAX, 0, abi.VARCHAR)], type_params=[20, 0])
    assert e.value.status == abi.ERR_NOT_SUPPORTED
    # a string longer than the declared length of a short channel fails the query (as a VARCHAR(n) cast would have upstream)
    op = HashAggregationOperator(types, [0], [(abi.AGG_MAX, 1, abi.VARCHAR)], type_params=[0, 7])
    page = Page([Block.bigint([1, 1]), Block.varchar([b"short", b"12345678"])], 2)
    with pytest.raises(PrestoAmdError):
        to_pages(op, [page])
