"""OrderByOperator on device (stable LSD radix sort of a row permutation) against the reference's known answers
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
