"""Small pages through the fused aggregation operator (what an unmodified Driver delivers: <= 1 MB / 8192 rows,
PageProcessor.java:56-58): consecutive PA_PAGE_STABLE device pages that continue each other in memory are processed as one
range without a copy, other small pages are gathered in the operator's arenas, and the launches of the few-groups tier are
confirmed late (needsInput / isBlocked instead of a wait inside addInput).  None of it may change a result: every case is
compared with the oracle and with the same rows handed over as one page."""
import numpy as np
import pytest

from presto_amd import abi, tpch
from presto_amd.operators import FusedAggregationOperator, HashAggregationOperator, to_pages, upload_page
from presto_amd.page import Block, DeviceBuffer, Page

pytestmark = pytest.mark.gpu


def stable_regions(dev_page, bounds):
    """Page.getRegion views of a device page, flagged PA_PAGE_STABLE (the test keeps dev_page alive past the operator)."""
    out = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        n = hi - lo
        blocks = []
        for b in dev_page.blocks:
            nulls = DeviceBuffer(b.nulls.ptr + lo, n, b.nulls) if b.nulls is not None else None
            if b.encoding == abi.VARWIDTH:
                blocks.append(Block(b.type, abi.VARWIDTH, n, values=b.values, offsets=DeviceBuffer(b.offsets.ptr + 4 * lo, 4 * (n + 1), b.offsets), nulls=nulls))
            else:
                w = abi.TYPE_WIDTH[b.type]
                blocks.append(Block(b.type, abi.FLAT, n, values=DeviceBuffer(b.values.ptr + w * lo, w * n, b.values), nulls=nulls))
        out.append(Page(blocks, n, abi.MEM_DEVICE, stable=True))
    return out


def bounds_of(n, page_rows):
    return list(range(0, n, page_rows)) + [n]


def q6_host(oracle, sf, n):
    cols = [oracle.tpch_column(c, sf, 0, n)[0] for c in tpch.Q6_COLUMNS]
    page = Page([Block.flat(t, c) for t, c in zip(tpch.Q6_TYPES, cols)], n)
    return page, oracle.q6(*cols)


def run_q6(pages, extra=()):
    op = FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [],
                                  tpch.Q6_AGGREGATES + [(abi.AGG_COUNT_STAR, -1, None)])
    (revenue, count), = to_pages(op, pages)[0].to_rows()
    op.close()
    return revenue, count


@pytest.mark.parametrize("page_rows", [8192, 4096 + 4, 65536])
def test_q6_over_small_pages_every_route(gpu, oracle, monkeypatch, page_rows):
    n, sf = 600_011, 0.1
    host, (ref_sum, ref_count) = q6_host(oracle, sf, n)
    dev = upload_page(host)
    bounds = bounds_of(n, page_rows)
    routes = {
        "stable ranges, one launch at finish": stable_regions(dev, bounds),
        "device pages of their own (arena, segment copies)": [upload_page(host.get_region(lo, hi - lo)) for lo, hi in zip(bounds[:-1], bounds[1:])][:40]
        + stable_regions(dev, [bounds[40], n] if len(bounds) > 41 else [n, n])[:1],
        "host pages (arena, H2D copies)": [host.get_region(lo, hi - lo) for lo, hi in zip(bounds[:-1], bounds[1:])],
    }
    for name, pages in routes.items():
        pages = [p for p in pages if p.position_count > 0]
        assert sum(p.position_count for p in pages) == n, name
        revenue, count = run_q6(pages)
        assert count == ref_count, name
        assert abs(revenue - ref_sum) <= 1e-9 * abs(ref_sum), name
    # a launch threshold below the page size: every stable page is launched on its own (range of one page)
    monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", "1000")
    revenue, count = run_q6(stable_regions(dev, bounds))
    assert count == ref_count and abs(revenue - ref_sum) <= 1e-9 * abs(ref_sum)
    # ranges broken by a gap: pages 0, 2, 4 ... then 1, 3, 5 ...
    monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", str(1 << 26))
    regs = stable_regions(dev, bounds)
    revenue, count = run_q6(regs[0::2] + regs[1::2])
    assert count == ref_count and abs(revenue - ref_sum) <= 1e-9 * abs(ref_sum)


@pytest.mark.parametrize("page_rows", [8192, 100_000])
def test_q1_over_small_stable_pages(gpu, oracle, page_rows):
    """VARCHAR(1) keys + the few-groups tier: stable ranges are merged (the shared byte array + consecutive offsets continue
    each other), the launches confirmed late."""
    n, sf = 1_300_003, 0.25
    cols = [oracle.tpch_column(c, sf, 0, n) for c in tpch.Q1_COLUMNS]
    args = [cols[0][0], cols[0][1], cols[1][0], cols[1][1]] + [c[0] for c in cols[2:]]
    expected = sorted(oracle.q1(args))
    dev = tpch.DeviceColumns(tpch.Q1_COLUMNS, sf, n)
    for pages in (list(dev.pages(page_rows)), [dev.page(0, n)]):
        op = FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                      type_params=tpch.Q1_TYPE_PARAMS)
        rows = sorted(r for p in to_pages(op, pages) for r in p.to_rows())
        op.close()
        assert len(rows) == len(expected) == 4
        for a, e in zip(rows, expected):
            assert a[:2] == e[:2] and a[-1] == e[-1]
            assert np.allclose(a[2:-1], e[2:-1], rtol=1e-9, atol=0)


def test_more_groups_appear_after_the_first_launches(gpu, oracle, monkeypatch):
    """The few-groups tier is confirmed one launch late for retained pages: rows that overflow its register table AFTER the
    first (probe) launch was confirmed must be redone on the next tier exactly once -- no row lost, none counted twice."""
    monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", str(1 << 18))
    rng = np.random.default_rng(3)
    n = (1 << 21) + 12345
    keys = rng.integers(0, 4, n).astype(np.int64)
    keys[(1 << 20) + (1 << 19):] = rng.integers(0, 40, n - (1 << 20) - (1 << 19))   # 40 groups from row 1.5 M on
    vals = rng.integers(-1000, 1000, n).astype(np.int64)
    host = Page([Block.bigint(keys), Block.bigint(vals)], n)
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MAX, 1, abi.BIGINT)]
    ref = oracle.HashAggregation([abi.BIGINT, abi.BIGINT], [0], aggs)
    ref.add_page(host)
    expected = sorted(ref.build_result().to_rows())
    dev = upload_page(host)
    for page_rows in (8192, 1 << 19, n):
        op = HashAggregationOperator([abi.BIGINT, abi.BIGINT], [0], aggs, expected_groups=8)
        blocked = 0
        pages = stable_regions(dev, bounds_of(n, page_rows))
        for p in pages:
            while not op.needsInput():
                assert op.isBlocked() or op.needsInput()   # only ever refuses input because launches are unconfirmed
                blocked += 1
            op.addInput(p)
        op.finish()
        rows = []
        while not op.isFinished():
            out = op.getOutput()
            if out is not None:
                rows += out.to_rows()
        op.close()
        assert sorted(rows) == expected, page_rows


def test_nullability_changes_between_small_pages(gpu, oracle):
    """Arena pages fix the nullability of their channels with the first page: a page that differs starts the next arena (and,
    where the state layout changes, the next generation)."""
    rng = np.random.default_rng(8)
    pages, keys_all = [], []
    for k in range(30):
        n = 3000 + k
        nulls = (rng.random(n) < 0.2) if k % 3 == 1 else None
        pages.append(Page([Block.bigint(rng.integers(0, 5, n)), Block.double(rng.random(n), nulls)], n))
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    ref = oracle.HashAggregation([abi.BIGINT, abi.DOUBLE], [0], aggs)
    for p in pages:
        ref.add_page(p)
    expected = sorted(ref.build_result().to_rows())
    for device in (False, True):
        op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], aggs)
        rows = sorted(r for p in to_pages(op, [upload_page(p) if device else p for p in pages]) for r in p.to_rows())
        op.close()
        assert len(rows) == len(expected)
        for a, e in zip(rows, expected):
            assert a[0] == e[0] and a[2:] == e[2:] and abs(a[1] - e[1]) <= 1e-12 * abs(e[1])


class Pinned:
    """Pinned host memory through the C ABI (pa_host_malloc_pinned) with numpy views: what the JNI shim's PinnedPagePool holds."""

    def __init__(self):
        self.allocs = []

    def copy(self, arr):
        import ctypes as C
        from presto_amd._lib import check, lib
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        check(lib().pa_host_malloc_pinned(C.byref(p), max(arr.nbytes, 16) + 16))
        self.allocs.append(p)
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(arr.nbytes, 1),))[:arr.nbytes].view(arr.dtype)
        out[...] = arr
        return out

    def page(self, page):
        blocks = []
        for b in page.blocks:
            nulls = None if b.nulls is None else self.copy(b.nulls)
            if b.encoding == abi.VARWIDTH:
                blocks.append(Block(b.type, abi.VARWIDTH, b.position_count, values=self.copy(b.values), offsets=self.copy(b.offsets), nulls=nulls))
            else:
                blocks.append(Block(b.type, abi.FLAT, b.position_count, values=self.copy(b.values), nulls=nulls))
        return Page(blocks, page.position_count, abi.MEM_HOST, stable=True, pinned=True)

    def free(self):
        from presto_amd._lib import lib
        for p in self.allocs:
            lib().pa_host_free_pinned(p)
        self.allocs = []


@pytest.mark.parametrize("page_rows", [8192, 50_000])
def test_q1_q6_over_small_host_pages_pageable_and_pinned(gpu, oracle, page_rows):
    """Host pages as a Driver delivers them (<= 1 MB): pageable buffers are copied block array by block array into the arena
    (VARCHAR offsets rebased behind the copy), pinned + stable ones (PA_PAGE_PINNED | PA_PAGE_STABLE, the JNI shim's staging
    slabs) are read by the device itself, all pages of an arena in one copy launch."""
    n, sf = 300_007, 0.05
    q1cols = [oracle.tpch_column(c, sf, 0, n) for c in tpch.Q1_COLUMNS]
    args = [q1cols[0][0], q1cols[0][1], q1cols[1][0], q1cols[1][1]] + [c[0] for c in q1cols[2:]]
    expected = sorted(oracle.q1(args))
    q1host = Page([Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v) for (v, o), t in zip(q1cols, tpch.Q1_TYPES)], n)
    q6host, (ref_sum, ref_count) = q6_host(oracle, sf, n)
    bounds = bounds_of(n, page_rows)

    def regions(page):
        return [page.get_region(lo, hi - lo) for lo, hi in zip(bounds[:-1], bounds[1:])]

    pinned = Pinned()
    try:
        for name, wrap in (("pageable", lambda p: p), ("pinned", pinned.page)):
            op = FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                          type_params=tpch.Q1_TYPE_PARAMS)
            rows = sorted(r for p in to_pages(op, [wrap(p) for p in regions(q1host)]) for r in p.to_rows())
            op.close()
            assert len(rows) == len(expected) == 4, name
            for a, e in zip(rows, expected):
                assert a[:2] == e[:2] and a[-1] == e[-1], name
                assert np.allclose(a[2:-1], e[2:-1], rtol=1e-9, atol=0), name
            revenue, count = run_q6([wrap(p) for p in regions(q6host)])
            assert count == ref_count and abs(revenue - ref_sum) <= 1e-9 * abs(ref_sum), name
    finally:
        pinned.free()


def q1_expected(oracle, sf, n):
    cols = [oracle.tpch_column(c, sf, 0, n) for c in tpch.Q1_COLUMNS]
    args = [cols[0][0], cols[0][1], cols[1][0], cols[1][1]] + [c[0] for c in cols[2:]]
    host = Page([Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v) for (v, o), t in zip(cols, tpch.Q1_TYPES)], n)
    return host, sorted(oracle.q1(args))


@pytest.mark.parametrize("in_place", [True, False])
@pytest.mark.parametrize("page_rows", [8192, 65536])
def test_q1_over_small_device_pages_with_varchar_channels(gpu, oracle, monkeypatch, page_rows, in_place):
    """Device pages whose VARCHAR blocks start at an offset only the device knows: stable pages that do not continue each
    other (a shuffled scan) are gathered by one planning + one copy launch per arena, pages with buffers of their own (what a
    device operator upstream hands over) by one launch each -- a byte cursor in HBM places the bytes and rebases the offsets --
    instead of a fused launch and its merges per page (PageProcessor.java:56-58 page sizes).  in_place: the few-groups tier takes
    the stable pages as a TABLE of row ranges, without any copy (the default); otherwise they are gathered like the others."""
    if not in_place:
        monkeypatch.setenv("PRESTO_AMD_NO_RANGES", "1")
    n, sf = 1_000_003, 0.2
    host, expected = q1_expected(oracle, sf, n)
    dev = upload_page(host)
    bounds = bounds_of(n, page_rows)
    stable = stable_regions(dev, bounds)
    order = np.random.default_rng(5).permutation(len(stable))
    own = [upload_page(host.get_region(lo, hi - lo)) for lo, hi in zip(bounds[:-1], bounds[1:])]
    routes = {
        "stable, shuffled": [stable[i] for i in order],
        "buffers of their own": own,
        "both kinds and host pages in turn": [(stable[i], own[i], host.get_region(bounds[i], bounds[i + 1] - bounds[i]))[i % 3] for i in order],
    }
    for name, pages in routes.items():
        op = FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                      type_params=tpch.Q1_TYPE_PARAMS)
        rows = sorted(r for p in to_pages(op, pages) for r in p.to_rows())
        _, launches = op.kernelTime()
        op.close()
        assert len(rows) == len(expected) == 4, name
        for a, e in zip(rows, expected):
            assert a[:2] == e[:2] and a[-1] == e[-1], name
            assert np.allclose(a[2:-1], e[2:-1], rtol=1e-9, atol=0), name
        if name != "both kinds and host pages in turn":
            assert launches <= max(len(pages) // 8, 3), (name, launches, len(pages))


def test_ragged_varchar_keys_over_small_device_pages(gpu, oracle):
    """Strings of 0..12 bytes with NULLs as a group key: blocks whose bytes start anywhere and are not aligned with where they
    land, pages of 1 to a few thousand rows, some empty."""
    rng = np.random.default_rng(17)
    words = [b"", b"a", b"bc", b"\xc3\xa9t\xc3\xa9", b"lineitem", b"twelve bytes", None]
    sizes = [1, 0, 4097, 3, 2500, 64, 8191, 2, 1000, 777] * 6
    types = [abi.VARCHAR, abi.BIGINT]
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None)]
    host_pages = []
    for k, n in enumerate(sizes):
        keys = [words[i] for i in rng.integers(0, len(words), n)]
        host_pages.append(Page([Block.varchar(keys), Block.bigint(rng.integers(-50, 50, n))], n))
    ref = oracle.HashAggregation(types, [0], aggs)
    for p in host_pages:
        ref.add_page(p)
    expected = sorted(ref.build_result().to_rows(), key=repr)
    own = [upload_page(p) for p in host_pages]
    # the same pages as regions of one device page (stable), visited out of order
    total = sum(sizes)
    merged = Page([Block.varchar([s for p in host_pages for s in p.blocks[0].to_pylist()]),
                   Block.bigint(np.concatenate([p.blocks[1].values for p in host_pages]))], total)
    dev = upload_page(merged)
    stable = stable_regions(dev, [0] + list(np.cumsum(sizes)))
    order = rng.permutation(len(stable))
    # ... and as pageable host pages: laid into pinned memory, one copy launch per page (PinnedPageCopy)
    for name, pages in (("own", own), ("stable", [stable[i] for i in order]), ("mixed", [(own[i], stable[i])[i & 1] for i in order]),
                        ("host", host_pages), ("host and device in turn", [(host_pages[i], own[i], stable[i])[i % 3] for i in range(len(own))])):
        op = HashAggregationOperator(types, [0], aggs, type_params=[12, 0])
        rows = sorted((r for p in to_pages(op, pages) for r in p.to_rows()), key=repr)
        op.close()
        assert rows == expected, name


def test_pageable_host_pages_either_side_of_the_one_copy_limit(gpu, oracle):
    """Pageable host pages up to 256 KB of block arrays go through pinned memory as one copy launch, larger ones array by array
    (device_page.hpp PinnedPageCopy, PageStager::kPackedLimit): pages of both kinds in turn, nullable channels, exact sums."""
    rng = np.random.default_rng(23)
    types = [abi.BIGINT, abi.BIGINT, abi.INTEGER]
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT, 1, abi.BIGINT), (abi.AGG_MAX, 2, abi.INTEGER), (abi.AGG_COUNT_STAR, -1, None)]
    sizes = [12_000, 13_200, 1, 12_900, 40_000, 12_901, 5, 12_483, 12_484, 3000] * 3   # 256 KB = 12 483 rows of 8 + 8 + 4 + 1 bytes
    pages = []
    for n in sizes:
        nulls = rng.random(n) < 0.1
        pages.append(Page([Block.bigint(rng.integers(0, 7, n)), Block.flat(abi.BIGINT, rng.integers(-10**12, 10**12, n), nulls),
                           Block.flat(abi.INTEGER, rng.integers(-2**31, 2**31 - 1, n).astype(np.int32))], n))
    ref = oracle.HashAggregation(types, [0], aggs)
    for p in pages:
        ref.add_page(p)
    expected = sorted(ref.build_result().to_rows())
    op = HashAggregationOperator(types, [0], aggs)
    rows = sorted(r for p in to_pages(op, pages) for r in p.to_rows())
    op.close()
    assert rows == expected


def test_varchar_bytes_beyond_the_declared_bound_are_refused(gpu):
    """The arena of device pages is sized by the declared VARCHAR(n): blocks that hold more than 4 n bytes per row of an arena
    do not fit -- nothing is written past the buffer, the operator fails with INVALID_ARGUMENT."""
    from presto_amd import expr as E
    from presto_amd._lib import PrestoAmdError
    n = 200_000
    strings = np.full(n * 120, ord("x"), dtype=np.uint8)
    offsets = np.arange(n + 1, dtype=np.int32) * 120
    pages = [upload_page(Page([Block.varwidth(strings, offsets), Block.bigint(np.ones(n, dtype=np.int64))], n)) for _ in range(2)]
    op = FusedAggregationOperator([abi.VARCHAR, abi.BIGINT], E.field(0, abi.VARCHAR).eq(E.constant("y", abi.VARCHAR)), [E.field(1, abi.BIGINT)], [],
                                  [(abi.AGG_SUM, 0, abi.BIGINT)], type_params=[1, 0])
    with pytest.raises(PrestoAmdError) as err:
        to_pages(op, pages)
    op.close()
    assert err.value.status == abi.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("page_rows", [8192, 4096 + 4, 65536])
def test_row_range_tables_ungrouped_and_few_groups(gpu, oracle, monkeypatch, page_rows):
    """Stable device pages that do not continue each other are taken in place as a table of row ranges by ONE launch of the
    ungrouped (Q6) / few-groups (Q1-like) kernels: ragged range sizes, unaligned ranges (scalar rows), NULL flags that appear in
    some ranges only, a launch threshold that cuts the table several times."""
    monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", str(1 << 18))
    n, sf = 700_001, 0.1
    host, (ref_sum, ref_count) = q6_host(oracle, sf, n)
    dev = upload_page(host)
    bounds = bounds_of(n, page_rows)
    regs = stable_regions(dev, bounds)
    order = np.random.default_rng(9).permutation(len(regs))
    revenue, count = run_q6([regs[i] for i in order])
    assert count == ref_count and abs(revenue - ref_sum) <= 1e-9 * abs(ref_sum)
    # few groups, BIGINT key, a nullable DOUBLE channel whose NULL flags exist in every third range only
    rng = np.random.default_rng(10)
    keys = rng.integers(0, 6, n).astype(np.int64)
    vals = rng.random(n)
    nulls = rng.random(n) < 0.1
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MIN, 1, abi.DOUBLE)]
    with_nulls = upload_page(Page([Block.bigint(keys), Block.double(vals, nulls)], n))
    pages, ref = [], oracle.HashAggregation([abi.BIGINT, abi.DOUBLE], [0], aggs)
    for i, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        reg = stable_regions(with_nulls, [lo, hi])[0]
        has = i % 3 == 0
        if not has:
            reg = Page([reg.blocks[0], Block(abi.DOUBLE, abi.FLAT, hi - lo, values=reg.blocks[1].values)], hi - lo, abi.MEM_DEVICE, stable=True)
        pages.append(reg)
        ref.add_page(Page([Block.bigint(keys[lo:hi]), Block.double(vals[lo:hi], nulls[lo:hi] if has else None)], hi - lo))
    expected = sorted(ref.build_result().to_rows())
    op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], aggs, expected_groups=8)
    rows = sorted(r for p in to_pages(op, [pages[i] for i in order]) for r in p.to_rows())
    _, launches = op.kernelTime()
    op.close()
    assert launches <= 2 + n // (1 << 18) + 2, launches
    assert len(rows) == len(expected)
    for a, e in zip(rows, expected):
        assert a[0] == e[0] and a[2:4] == e[2:4] and a[4] == e[4] and abs(a[1] - e[1]) <= 1e-9 * abs(e[1])


def test_row_range_table_when_the_few_groups_tier_gives_up(gpu, oracle, monkeypatch):
    """A table of ranges whose rows hold more groups than the wave-level tables: the launch is redone range by range on the next
    tier, and later ranges are gathered the old way -- no row lost, none counted twice."""
    monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", str(1 << 17))
    rng = np.random.default_rng(12)
    n = 500_000
    keys = rng.integers(0, 4, n).astype(np.int64)
    keys[300_000:] = rng.integers(0, 5000, n - 300_000)
    vals = rng.integers(-1000, 1000, n).astype(np.int64)
    aggs = [(abi.AGG_SUM, 1, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MAX, 1, abi.BIGINT)]
    host = Page([Block.bigint(keys), Block.bigint(vals)], n)
    ref = oracle.HashAggregation([abi.BIGINT, abi.BIGINT], [0], aggs)
    ref.add_page(host)
    expected = sorted(ref.build_result().to_rows())
    dev = upload_page(host)
    regs = stable_regions(dev, bounds_of(n, 8192))
    # every other page first, then the rest: no page continues its predecessor, the many-groups rows arrive mid-way
    op = HashAggregationOperator([abi.BIGINT, abi.BIGINT], [0], aggs, expected_groups=8)
    rows = sorted(r for p in to_pages(op, regs[0::2] + regs[1::2]) for r in p.to_rows())
    op.close()
    assert rows == expected


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("PA_FUZZ_SEEDS", "8")))))
def test_random_mix_of_page_routes(gpu, oracle, monkeypatch, seed):
    """Pages of random sizes handed over in random order through a random mix of routes -- stable views of one device page (merged
    when they continue each other, taken as a table of row ranges when they do not), device pages with buffers of their own, plain
    host pages -- with a launch threshold that cuts the stream several times: a VARCHAR(3) + BIGINT grouped aggregation with NULL keys
    and NULL values, and an ungrouped one, must give the oracle's result whatever the mix."""
    rng = np.random.default_rng(4000 + seed)
    monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", str(int(rng.choice([1 << 15, 1 << 17, 1 << 26]))))
    n = int(rng.integers(150_000, 400_000))
    words = [b"", b"a", b"ab", b"abc", b"zz", None]
    skeys = [words[i] for i in rng.integers(0, len(words), n)]
    ikeys = rng.integers(0, int(rng.choice([3, 40])), n).astype(np.int64)
    vals = rng.random(n) * 10 - 5
    vnull = rng.random(n) < 0.1
    host = Page([Block.varchar(skeys), Block.bigint(ikeys), Block.double(vals, vnull)], n)
    dev = upload_page(host)
    cuts = sorted(set(int(x) for x in rng.integers(1, n, int(rng.integers(5, 60)))))
    bounds = [0] + cuts + [n]
    views = stable_regions(dev, bounds)
    order = rng.permutation(len(views)) if rng.random() < 0.7 else np.arange(len(views))
    pages = []
    for i in order:
        lo, hi = bounds[i], bounds[i + 1]
        route = rng.integers(0, 3)
        pages.append(views[i] if route == 0 else (upload_page(host.get_region(lo, hi - lo)) if route == 1 else host.get_region(lo, hi - lo)))
    types = [abi.VARCHAR, abi.BIGINT, abi.DOUBLE]
    aggs = [(abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_COUNT, 2, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MAX, 1, abi.BIGINT)]
    for keys in ([0, 1], []):
        ref = oracle.HashAggregation(types, keys, aggs)
        ref.add_page(host)
        expected = sorted(ref.build_result().to_rows(), key=repr)
        op = HashAggregationOperator(types, keys, aggs, type_params=[3, 0, 0])
        got = sorted((r for p in to_pages(op, pages) for r in p.to_rows()), key=repr)
        op.close()
        assert len(got) == len(expected)
        for g, e in zip(got, expected):
            assert g[:len(keys)] == e[:len(keys)] and g[len(keys) + 1:] == e[len(keys) + 1:], (g, e)
            assert (g[len(keys)] is None and e[len(keys)] is None) or abs(g[len(keys)] - e[len(keys)]) <= 1e-9 * max(abs(e[len(keys)]), 1.0), (g, e)


# ---- PA_PAGE_RETAINED: pages kept by their owner until the operator releases them -----------------------------------------------
def retained_pages(host, bounds, released, scribble=True):
    """Every row range of `host` uploaded to buffers of its own and handed over as a retained page.  The release callback does what a
    staging pool's would -- it takes the buffers back: here it overwrites them with garbage, so an operator that releases a page
    before its last read of it computes a wrong result."""
    from presto_amd._lib import check, lib
    pages = []
    for i, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
        dev = upload_page(host.get_region(lo, hi - lo))

        def on_release(i=i, dev=dev):
            released.append(i)
            if scribble:
                for b in dev.blocks:
                    for buf in (b.values, b.offsets):
                        if buf is not None and buf.nbytes:
                            junk = np.full(buf.nbytes, 0x7F, dtype=np.uint8)
                            check(lib().pa_memcpy_h2d(buf.ptr, junk.ctypes.data, buf.nbytes, None))
        pages.append(Page(dev.blocks, dev.position_count, abi.MEM_DEVICE, on_release=on_release))
    return pages


@pytest.mark.parametrize("page_rows", [8192, 65536, 1 << 22])
def test_retained_pages_are_released_once_and_only_after_their_last_read(gpu, oracle, monkeypatch, page_rows):
    n, sf = 1_300_003, 0.25
    host6, (ref_sum, ref_count) = q6_host(oracle, sf, n)
    bounds = bounds_of(n, page_rows)
    for gather in (None, "20000"):   # one table of ranges at finish / a launch (and a round of releases) every few pages
        if gather:
            monkeypatch.setenv("PRESTO_AMD_GATHER_ROWS", gather)
        released = []
        pages = retained_pages(host6, bounds, released)
        revenue, count = run_q6(pages)
        assert count == ref_count and abs(revenue - ref_sum) <= 1e-9 * abs(ref_sum)
        assert sorted(released) == list(range(len(pages)))     # every page, exactly once (the operator is closed)
    monkeypatch.delenv("PRESTO_AMD_GATHER_ROWS", raising=False)
    # Q1: VARCHAR channels, the few-groups tier with its late confirmation
    cols = [oracle.tpch_column(c, sf, 0, n) for c in tpch.Q1_COLUMNS]
    host1 = Page([Block.varwidth(*c) if t == abi.VARCHAR else Block.flat(t, c[0]) for t, c in zip(tpch.Q1_TYPES, cols)], n)
    expected = sorted(oracle.q1([cols[0][0], cols[0][1], cols[1][0], cols[1][1]] + [c[0] for c in cols[2:]]))
    released = []
    pages = retained_pages(host1, bounds, released)
    op = FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS)
    rows = sorted(r for p in to_pages(op, pages) for r in p.to_rows())
    op.close()
    assert sorted(released) == list(range(len(pages)))
    assert len(rows) == len(expected)
    for a, e in zip(rows, expected):
        assert a[:2] == e[:2] and a[-1] == e[-1] and np.allclose(a[2:-1], e[2:-1], rtol=1e-9, atol=0)


def test_retained_pages_through_operators_that_do_not_hold_pages(gpu, oracle):
    """An operator without retention (here: HashAggregation over a high-cardinality key handed over as dictionary blocks, and a
    FilterAndProject) gets the page as a plain one; the library releases it before add_input returns."""
    from presto_amd.expr import field
    from presto_amd.operators import FilterAndProjectOperator
    released = []
    host = Page([Block.bigint(np.arange(50_000) % 977), Block.double(np.arange(50_000, dtype=np.float64))], 50_000)
    pages = retained_pages(host, bounds_of(50_000, 7_000), released, scribble=False)
    op = FilterAndProjectOperator([abi.BIGINT, abi.DOUBLE], field(0, abi.BIGINT) < 10, [field(0, abi.BIGINT), field(1, abi.DOUBLE)])
    seen = 0
    for i, p in enumerate(pages):
        assert op.needsInput()
        op.addInput(p)
        assert released == list(range(i + 1))   # handed back when the call returned
        out = op.getOutput()
        seen += out.position_count if out is not None else 0
    op.finish()
    op.close()
    assert seen == int((np.arange(50_000) % 977 < 10).sum())
