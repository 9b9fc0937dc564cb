"""OrderByOperator on device (stable sorts of (key image, row id) pairs, channel by channel) against the reference's known answers
(core/trino-main/src/test/java/io/trino/operator/TestOrderByOperator.java:130-232) and the oracle on random pages: every type as sort
key, all four SortOrders, multi-channel keys, long VARCHAR keys, ties in arrival order."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import OrderByOperator, download_page, to_pages, upload_page
from presto_amd.page import Block, Page

pytestmark = pytest.mark.gpu


def rows_of(pages):
    return [r for p in pages for r in p.to_rows()]


def test_order_by_kats(gpu, oracle):
    pages = [Page([Block.bigint([1, 2]), Block.double([0.1, 0.2])]), Page([Block.bigint([-1, 4]), Block.double([-0.1, 0.4])])]
    types = [abi.BIGINT, abi.DOUBLE]
    # testSingleFieldKey (:130-163)
    assert rows_of(to_pages(OrderByOperator(types, [1], [0], [abi.ASC_NULLS_LAST]), pages)) == [(-0.1,), (0.1,), (0.2,), (0.4,)]
    # testReverseOrder (:200-233)
    assert rows_of(to_pages(OrderByOperator(types, [0], [0], [abi.DESC_NULLS_LAST]), pages)) == [(4,), (2,), (1,), (-1,)]
    # testMultiFieldKey (:165-198)
    vp = [Page([Block.varchar(["a", "b"]), Block.bigint([1, 2])]), Page([Block.varchar(["b", "a"]), Block.bigint([3, 4])])]
    got = rows_of(to_pages(OrderByOperator([abi.VARCHAR, abi.BIGINT], [0, 1], [0, 1], [abi.ASC_NULLS_LAST, abi.DESC_NULLS_LAST]), vp))
    assert got == [(b"a", 4), (b"a", 1), (b"b", 3), (b"b", 2)]
    assert oracle.order_by(vp, [0, 1], [0, 1], [abi.ASC_NULLS_LAST, abi.DESC_NULLS_LAST]) == got


@pytest.mark.parametrize("first", ["bigint", "double", "varchar", "date", "boolean"])
@pytest.mark.parametrize("order", [abi.ASC_NULLS_FIRST, abi.ASC_NULLS_LAST, abi.DESC_NULLS_FIRST, abi.DESC_NULLS_LAST])
def test_order_by_matches_oracle(gpu, oracle, first, order):
    rng = np.random.default_rng(hash((first, order, 7)) % 2 ** 32)
    words = [b"", b"a", b"ab", b"ab\x00", b"abcdefgh", b"abcdefghX", b"abcdefghXYZ0123456789", b"abcdefghXYZ0123456788", b"zz", b"\xc3\xa9", None]
    pages = []
    for _ in range(3):
        n = int(rng.integers(1, 20000))
        nulls = rng.random(n) < 0.05
        if first == "bigint":
            k = Block.bigint(rng.integers(-50, 50, n), nulls)
        elif first == "double":
            v = np.round(rng.standard_normal(n), 1)
            v[rng.random(n) < 0.02] = np.nan
            v[rng.random(n) < 0.02] = -0.0
            k = Block.double(v, nulls)
        elif first == "varchar":
            k = Block.varchar([words[j] for j in rng.integers(0, len(words), n)])
        elif first == "date":
            k = Block.date(rng.integers(8000, 8020, n), nulls)
        else:
            k = Block.boolean(rng.random(n) < 0.5, nulls)
        # second key has few values too: many fully tied rows, whose arrival order must survive; payload = arrival index
        pages.append(Page([k, Block.integer(rng.integers(0, 3, n), rng.random(n) < 0.1), Block.bigint(np.arange(n) + 100000 * len(pages))], n))
    types = [pages[0].blocks[0].type, abi.INTEGER, abi.BIGINT]
    orders = [order, abi.DESC_NULLS_FIRST]
    expected = oracle.order_by(pages, [2, 0, 1], [0, 1], orders)
    got = rows_of(to_pages(OrderByOperator(types, [2, 0, 1], [0, 1], orders), pages))
    assert len(got) == len(expected)
    for g, e in zip(got, expected):
        assert g[0] == e[0], (g, e)   # the arrival index pins the whole permutation, ties included


def test_order_by_device_pages_and_output(gpu, oracle):
    rng = np.random.default_rng(4)
    pages = [Page([Block.double(rng.random(100000)), Block.bigint(np.arange(100000) + 100000 * i)], 100000) for i in range(3)]
    op = OrderByOperator([abi.DOUBLE, abi.BIGINT], [1], [0], [abi.DESC_NULLS_LAST], output_mem=abi.MEM_DEVICE)
    for p in [upload_page(p) for p in pages]:
        op.addInput(p)
    op.finish()
    out = download_page(op.getOutput())
    assert out.to_rows() == oracle.order_by(pages, [1], [0], [abi.DESC_NULLS_LAST])
    assert op.isFinished()


@pytest.mark.parametrize("n", [300, 5000, 150_000])
@pytest.mark.parametrize("kind", ["bigint_wide", "bigint_narrow", "integer", "date"])
def test_first_sort_channel_as_output_and_every_sort_size(gpu, oracle, n, kind):
    """The pair sort's regimes (one workgroup, merge sort; the one-sweep radix passes in the test below) over key ranges whose varying bits
    end at bit 64 (negative and positive keys), lie in the middle of the word, or start at bit 0; the first sort channel -- an integer channel
    without NULL rows -- is also an output channel, which the operator then writes from the sorted keys instead of gathering it.  A second
    sort channel breaks ties; the arrival index pins the permutation."""
    rng = np.random.default_rng(n % 9973 + len(kind))
    page = Page([sort_keys(rng, kind, n), Block.integer(rng.integers(0, 4, n), rng.random(n) < 0.1), Block.bigint(np.arange(n))], n)
    types = [page.blocks[0].type, abi.INTEGER, abi.BIGINT]
    for orders in ([abi.ASC_NULLS_LAST, abi.DESC_NULLS_FIRST], [abi.DESC_NULLS_FIRST, abi.ASC_NULLS_LAST]):
        expected = oracle.order_by([page], [0, 2, 1], [0, 1], orders)
        halves = [page.get_region(0, n // 3), page.get_region(n // 3, n - n // 3)]
        got = rows_of(to_pages(OrderByOperator(types, [0, 2, 1], [0, 1], orders), halves))
        assert got == expected


def sort_keys(rng, kind, n):
    if kind == "bigint_wide":
        return Block.bigint(rng.integers(-2 ** 62, 2 ** 62, n))
    if kind == "bigint_narrow":
        return Block.bigint((rng.integers(0, 1 << 20, n).astype(np.int64) << 12) + (7 << 40))   # bits 12..31 vary
    if kind == "integer":
        return Block.integer(rng.integers(-1000, 1000, n))
    return Block.date(rng.integers(8000, 8300, n))


@pytest.mark.parametrize("kind", ["bigint_wide", "bigint_narrow", "integer"])
def test_sorts_beyond_a_million_rows(gpu, kind):
    """Above 2^20 rows the pair sort runs its one-sweep radix passes: checked against numpy's stable sort (the oracle's row-at-a-time
    quicksort would take a minute here) -- keys in order, ties in arrival order, both directions, the key column written from the keys."""
    n = 1_300_000
    rng = np.random.default_rng(len(kind))
    key = sort_keys(rng, kind, n)
    page = Page([key, Block.bigint(np.arange(n))], n)
    types = [key.type, abi.BIGINT]
    values = np.asarray(key.values).astype(np.int64)
    for order in (abi.ASC_NULLS_LAST, abi.DESC_NULLS_FIRST):
        out = to_pages(OrderByOperator(types, [0, 1], [0], [order]), [page.get_region(0, 400_000), page.get_region(400_000, n - 400_000)])
        got_keys = np.concatenate([np.asarray(p.blocks[0].values).astype(np.int64) for p in out])
        got_rows = np.concatenate([np.asarray(p.blocks[1].values) for p in out])
        perm = np.argsort(values if order == abi.ASC_NULLS_LAST else ~values, kind="stable")
        assert np.array_equal(got_rows, perm)
        assert np.array_equal(got_keys, values[perm])


@pytest.mark.parametrize("mix", ["one stable", "one retained", "several retained", "retained then host then stable", "retained with a VARCHAR channel"])
def test_device_pages_that_stay_are_kept_not_copied(gpu, oracle, mix):
    """PagesIndex.addPage keeps the Page (PagesIndex.java:150-170).  Device pages flagged PA_PAGE_STABLE / PA_PAGE_RETAINED with flat
    fixed-width blocks are listed by the operator instead of copied: one alone is sorted where it lies, several are laid behind each other
    when the input ends, a plain page in between forces the listed ones to be copied first (arrival order).  Retained pages are released
    exactly once, never before their last read (the release overwrites the buffers), at close at the latest."""
    from tests.test_gpu_small_pages import retained_pages
    rng = np.random.default_rng(len(mix))
    n = 50_000
    nulls = rng.random(n) < 0.05
    blocks = [Block.bigint(rng.integers(-1000, 1000, n)), Block.flat(abi.DOUBLE, rng.standard_normal(n), nulls), Block.integer(np.arange(n, dtype=np.int32))]
    types = [abi.BIGINT, abi.DOUBLE, abi.INTEGER]
    if mix == "retained with a VARCHAR channel":
        blocks.append(Block.varchar([b"k%d" % (i % 7) for i in range(n)]))
        types.append(abi.VARCHAR)
    host = Page(blocks, n)
    released = []
    if mix == "one stable":
        dev = upload_page(host)
        pages = [Page(dev.blocks, n, abi.MEM_DEVICE, stable=True)]
        keep = dev
    elif mix in ("one retained", "retained with a VARCHAR channel"):
        pages = retained_pages(host, [0, n], released)
    elif mix == "several retained":
        pages = retained_pages(host, [0, 7, 20_000, 20_001, n], released)
    else:
        first = retained_pages(host.get_region(0, 10_000), [0, 10_000], released)
        dev = upload_page(host.get_region(30_000, n - 30_000))
        pages = [first[0], host.get_region(10_000, 20_000), Page(dev.blocks, n - 30_000, abi.MEM_DEVICE, stable=True)]
        keep = dev
    outs = list(range(len(types)))
    op = OrderByOperator(types, outs, [0, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_FIRST])
    got = [row for p in to_pages(op, pages) for row in p.to_rows()]
    want = oracle.order_by([host], outs, [0, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_FIRST])
    op.close()

    def norm(rows):
        return [tuple(repr(v) if isinstance(v, float) else v for v in r) for r in rows]
    assert norm(got) == norm(want)
    handed = sum(1 for p in pages if getattr(p, "on_release", None) is not None)
    assert sorted(released) == sorted(set(released)) and len(released) == handed, (released, handed)
