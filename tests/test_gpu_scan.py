"""ScanFilterAndProjectOperator on the device: the reference's known-answer cases
(core/trino-main/src/test/java/io/trino/operator/TestScanFilterAndProjectOperator.java:98-255) through pa_scan_filter_project_create,
plus the lazy-load rule and the accounting on a selective filter, against the oracle's PageProcessor / MergePages restatement."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.expr import constant, field
from presto_amd.operators import LazyBlock, ScanFilterAndProjectOperator, source_to_pages
from presto_amd.page import Block, Page, sequence_page

pytestmark = pytest.mark.gpu


class FixedPageSource:
    """FixedPageSource (core/trino-spi/.../connector/FixedPageSource.java): hands out its pages in order."""

    def __init__(self, pages):
        self.pages, self.closed = list(pages), 0

    def getNextPage(self):
        return self.pages.pop(0) if self.pages else None

    def close(self):
        self.closed += 1


def test_page_source(gpu):
    """testPageSource (:98-128): VARCHAR sequence page of 10 000 rows, identity projection, no filter -> the input."""
    page = sequence_page(10_000, [(abi.VARCHAR, 0)])
    source = FixedPageSource([page])
    op = ScanFilterAndProjectOperator(source, [abi.VARCHAR], None, [field(0, abi.VARCHAR)])
    assert not op.needsInput()
    out = source_to_pages(op)
    assert [r for p in out for r in p.to_rows()] == page.to_rows()
    assert source.closed == 1 and op.stats()[0] == 10_000
    op.close()
    assert source.closed == 1


def test_page_source_merge_output(gpu):
    """testPageSourceMergeOutput (:130-177): 4 x sequence page (100, 0); filter c0 = 10; project c0; min output page 64 KB / 2 rows
    -> ONE page holding [10, 10, 10, 10]."""
    pages = [sequence_page(100, [(abi.BIGINT, 0)]) for _ in range(4)]
    op = ScanFilterAndProjectOperator(FixedPageSource(pages), [abi.BIGINT], field(0, abi.BIGINT).eq(constant(10, abi.BIGINT)), [field(0, abi.BIGINT)],
                                      min_output_page_size=64 * 1024, min_output_page_row_count=2)
    out = source_to_pages(op)
    assert len(out) == 1 and out[0].to_rows() == [(10,), (10,), (10,), (10,)]
    op.close()


def test_page_source_lazy_load(gpu):
    """testPageSourceLazyLoad (:179-215): channel 1 is a LazyBlock whose loader fails the test; the processor (select-all filter,
    projection of channel 0) must never load it."""
    def must_not_load():
        raise AssertionError("Lazy block should not be loaded")

    lazy = LazyBlock(100, must_not_load)
    page = Page.__new__(Page)
    page.blocks, page.position_count, page.mem, page.stable, page.pinned = [Block.bigint(np.arange(100)), lazy], 100, abi.MEM_HOST, False, False
    op = ScanFilterAndProjectOperator(FixedPageSource([page]), [abi.BIGINT, abi.BIGINT], field(0, abi.BIGINT) >= constant(0, abi.BIGINT), [field(0, abi.BIGINT)])
    out = source_to_pages(op)
    assert [r[0] for p in out for r in p.to_rows()] == list(range(100))
    assert lazy.loaded is None
    op.close()


def test_record_cursor_source(gpu):
    """testRecordCursorSource (:217-255): the same rows through a RecordPageSource -- a source that builds its pages row by
    row from a cursor (here: a generator cutting the sequence into pages of 1 000 rows)."""
    class RecordPageSource:
        def __init__(self, rows, page_rows):
            self.rows, self.page_rows, self.at = rows, page_rows, 0

        def getNextPage(self):
            if self.at >= len(self.rows):
                return None
            chunk = self.rows[self.at:self.at + self.page_rows]
            self.at += len(chunk)
            return Page([Block.varchar(chunk)], len(chunk))

    rows = [str(i).encode() for i in range(10_000)]
    op = ScanFilterAndProjectOperator(RecordPageSource(rows, 1000), [abi.VARCHAR], None, [field(0, abi.VARCHAR)])
    assert [r[0] for p in source_to_pages(op) for r in p.to_rows()] == rows
    op.close()


def make_lazy_page(blocks, loads):
    page = Page.__new__(Page)
    n = blocks[0].position_count

    def loader(i, b):
        def load():
            loads.append(i)
            return b
        return load

    page.blocks = [LazyBlock(n, loader(i, b)) for i, b in enumerate(blocks)]
    page.position_count, page.mem, page.stable, page.pinned = n, abi.MEM_HOST, False, False
    return page


def test_lazy_blocks_are_loaded_by_need_and_accounted(gpu, oracle):
    """PageProcessor.java:307-347 on the device: per page the filter's channel is loaded first; the projections' other channels
    only when a position survives; a channel no expression reads never.  Pages 0, 2, 4 select nothing.  Output rows equal the
    oracle's; materialised bytes follow Block.getSizeInBytes of the loaded blocks (recordMaterializedBytes, :391)."""
    rng = np.random.default_rng(12)
    n = 5000
    loads_per_page, pages, host_pages = [], [], []
    for k in range(6):
        key = np.full(n, 1000, dtype=np.int64) if k % 2 == 0 else rng.integers(0, 100, n)
        blocks = [Block.bigint(key), Block.double(rng.random(n)), Block.varchar([b"s%d" % i for i in range(n)]), Block.integer(rng.integers(0, 9, n))]
        loads = []
        loads_per_page.append(loads)
        pages.append(make_lazy_page(blocks, loads))
        host_pages.append(Page(blocks, n))
    types = [abi.BIGINT, abi.DOUBLE, abi.VARCHAR, abi.INTEGER]
    flt = field(0, abi.BIGINT) < constant(50, abi.BIGINT)
    projections = [field(1, abi.DOUBLE) * constant(2.0, abi.DOUBLE), field(2, abi.VARCHAR), field(0, abi.BIGINT)]
    op = ScanFilterAndProjectOperator(FixedPageSource(pages), types, flt, projections)
    rows = [r for p in source_to_pages(op) for r in p.to_rows()]
    expected = []
    for p in host_pages:
        out = oracle.filter_project(p, flt, projections)   # None: no position selected (PageProcessor.java:127-129)
        expected += [] if out is None else out.to_rows()
    assert rows == expected and len(rows) > 1000
    for k, loads in enumerate(loads_per_page):
        assert loads == ([0] if k % 2 == 0 else [0, 1, 2]), (k, loads)     # channel 3 is never loaded
    positions, bytes_loaded, loaded, skipped = op.stats()
    assert positions == 6 * n and loaded == 6 + 3 * 2 and skipped == 3 * 2
    varchar_bytes = sum(len(b"s%d" % i) for i in range(n)) + 5 * n
    assert bytes_loaded == 6 * n * 9 + 3 * (n * 9 + varchar_bytes)
    op.close()


def test_page_source_error_surfaces(gpu):
    class Failing:
        def getNextPage(self):
            raise ValueError("connector failed")

    op = ScanFilterAndProjectOperator(Failing(), [abi.BIGINT], None, [field(0, abi.BIGINT)])
    with pytest.raises(ValueError, match="connector failed"):
        op.getOutput()
    op.close()
