"""Randomised OrderBy / TopN against the oracle: 1-3 sort channels of every type (DOUBLE with NaN / -0.0, VARCHAR with shared
prefixes and empty strings, narrow and wide integer ranges so that some radix passes are skipped and others are not), every
SortOrder, NULLs, several pages, ties (which must keep arrival order)."""
import os

import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import OrderByOperator, TopNOperator, to_pages
from presto_amd.page import Block, Page

pytestmark = pytest.mark.gpu

WORDS = [b"", b"a", b"ab", b"abc", b"abcdefgh", b"abcdefghi", b"abcdefgh\x00", b"zz", b"Zz", b"\xff\xfe", b"abcdefghabcdefghX", b"abcdefghabcdefghY"]


def column(rng, t, n, card):
    nulls = (rng.random(n) < 0.08) if rng.random() < 0.6 else None
    if t == abi.BIGINT:
        span = int(rng.choice([3, 1 << 9, 1 << 33, 1 << 62]))
        return Block.bigint(rng.integers(-span, span, n), nulls)
    if t == abi.INTEGER:
        return Block.integer(rng.integers(-card, card + 1, n), nulls)
    if t == abi.DATE:
        return Block.date(rng.integers(0, card + 1, n), nulls)
    if t == abi.BOOLEAN:
        return Block.boolean(rng.random(n) < 0.5, nulls)
    if t == abi.DOUBLE:
        pool = np.concatenate([rng.standard_normal(max(card, 2)) * 1e3, [0.0, -0.0, np.nan, np.inf, -np.inf]])
        return Block.double(pool[rng.integers(0, len(pool), n)], nulls)
    return Block.varchar([None if (nulls is not None and nulls[i]) else WORDS[j] for i, j in enumerate(rng.integers(0, len(WORDS), n))])


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("PA_FUZZ_SEEDS", "24")))))
def test_random_sorts(gpu, oracle, seed):
    rng = np.random.default_rng(7300 + seed)
    kinds = [abi.BIGINT, abi.INTEGER, abi.DATE, abi.BOOLEAN, abi.DOUBLE, abi.VARCHAR]
    ncols = int(rng.integers(2, 5))
    types = [kinds[int(i)] for i in rng.integers(0, len(kinds), ncols)]
    card = int(rng.choice([2, 40, 5000]))
    pages = []
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.choice([1, 33, 2000, 30000]))
        pages.append(Page([column(rng, t, n, card) for t in types], n))
    nsort = int(rng.integers(1, min(3, ncols) + 1))
    sort_channels = [int(c) for c in rng.choice(ncols, nsort, replace=False)]
    orders = [int(o) for o in rng.integers(0, 4, nsort)]
    outs = list(range(ncols))

    def norm(rows):
        return [tuple("nan" if isinstance(v, float) and v != v else (repr(v) if isinstance(v, float) else v) for v in r) for r in rows]

    got = [r for p in to_pages(OrderByOperator(types, outs, sort_channels, orders), pages) for r in p.to_rows()]
    assert norm(got) == norm(oracle.order_by(pages, outs, sort_channels, orders))
    total = sum(p.position_count for p in pages)
    n = int(rng.choice([1, 7, 100, total + 5]))
    top = [r for p in to_pages(TopNOperator(types, n, sort_channels, orders), pages) for r in p.to_rows()]
    assert norm(top) == norm(oracle.topn(pages, n, sort_channels, orders))


def _sorted_rows(keys, descending=False):
    """(key, arrival index) rows in PagesIndexOrdering's order for one BIGINT channel: by key, ties in arrival order."""
    order = np.argsort(-keys if descending else keys, kind="stable")
    return keys[order], order


def _expected_sort(keys):
    """Which sort launch_sort_pairs takes for these BIGINT keys (sort_kernels.hip's plan, restated): images = keys with the sign bit
    flipped, sorted over the bits in which they differ; ~1024 pairs per bucket of the top bits, at most 14; a bucket beyond 2048 pairs
    with bits left to sort by -> the library."""
    n = len(keys)
    img = keys.astype(np.uint64) ^ np.uint64(1 << 63)
    varying = int(np.bitwise_or.reduce(img)) ^ int(np.bitwise_and.reduce(img))
    begin, end = (varying & -varying).bit_length() - 1, varying.bit_length()
    if end == 64:
        begin = 0
    width = end - begin
    if n <= 2048:
        return "pa_sort_buckets"
    top = min(width, 14, max(1, int(np.ceil(np.log2((n + 1023) // 1024)))))
    if width == top:
        return "pa_sort_buckets"
    buckets = (img >> np.uint64(end - top)) & np.uint64((1 << top) - 1)
    return "rocprim_radix_sort_pairs" if np.bincount(buckets.astype(np.int64)).max() > 2048 else "pa_sort_buckets"


SORT_SIZES = [1, 2, 63, 2048, 2049, 4095, 4097, 10_000, 262_147, (1 << 20) + 5, 3_000_001]


@pytest.mark.parametrize("n", SORT_SIZES)
@pytest.mark.parametrize("spread", ["40 bits", "63 bits", "5 bits", "14 bits", "wide with duplicates", "hot prefix", "one outlier"])
def test_pair_sort_paths(gpu, n, spread):
    """The pair sort under OrderBy (sort_kernels.hip) at the sizes where its plan changes -- one LDS bucket (<= 2048 pairs), one or two
    partition passes, tiles that end inside a bucket -- and over key spreads that take each of its routes: uniform keys (partition passes
    + LDS bucket sort), ranges no wider than the partition passes (no LDS sort), keys crowded under one bit prefix (the library sort
    takes over after the bucket sizes are known).  Expected: numpy's stable sort of (key, arrival index)."""
    rng = np.random.default_rng(n * 31 + len(spread))
    if spread == "40 bits":
        keys = rng.integers(0, 1 << 40, n, dtype=np.int64)
    elif spread == "63 bits":
        keys = rng.integers(-(1 << 62), 1 << 62, n, dtype=np.int64)
    elif spread == "5 bits":
        keys = rng.integers(0, 32, n, dtype=np.int64)
    elif spread == "14 bits":
        keys = rng.integers(0, 1 << 14, n, dtype=np.int64) * 8 + 3
    elif spread == "wide with duplicates":
        keys = rng.integers(0, 1 << 40, max(n // 16, 1), dtype=np.int64)[rng.integers(0, max(n // 16, 1), n)]
    elif spread == "hot prefix":
        keys = rng.integers(0, 1 << 20, n, dtype=np.int64)
        keys[rng.random(n) < 0.02] += 1 << 50     # the varying range is 51 bits wide, nearly every key sits under one prefix of it
    else:
        keys = np.full(n, 7, dtype=np.int64)
        keys[n // 2] = 1 << 45
    descending = n % 2 == 1
    # (the INTEGER and DOUBLE channels ride along with the pairs -- 4- and 8-byte payload columns of the sort -- when the hand-written
    # sort runs, and are gathered by the sorted row ids when the library's does)
    page = Page([Block.bigint(keys), Block.integer(np.arange(n, dtype=np.int32)), Block.double(np.arange(n) * 0.5)], n)
    op = OrderByOperator([abi.BIGINT, abi.INTEGER, abi.DOUBLE], [0, 1, 2], [0], [abi.DESC_NULLS_LAST if descending else abi.ASC_NULLS_LAST])
    out = to_pages(op, [page])
    name = op.kernelName()
    op.close()
    got_keys = np.concatenate([p.blocks[0].values for p in out])
    got_rows = np.concatenate([p.blocks[1].values for p in out])
    want_keys, want_rows = _sorted_rows(keys, descending)
    assert np.array_equal(got_keys, want_keys)
    assert np.array_equal(got_rows, want_rows.astype(np.int32))
    assert np.array_equal(np.concatenate([p.blocks[2].values for p in out]), want_rows * 0.5)
    if len(np.unique(keys)) > 1:
        assert name == _expected_sort(keys), (name, n, spread)


@pytest.mark.parametrize("n", [3_001, 262_144, 1_000_003, 5_000_000, 10_000_000])
@pytest.mark.parametrize("spread", ["uniform [0, 1)", "normal", "exponential", "signed, huge range", "specials", "few values", "REAL uniform"])
def test_pair_sort_by_sampled_bounds(gpu, n, spread):
    """DOUBLE / REAL sort keys: their images crowd under a few bit prefixes (the exponent), so the pair sort takes its bucket bounds from a
    sorted sample of the keys (sort_kernels.hip, PA_SORT_HINT_CROWDED; 16 sampled keys per bucket of ~512 rows, up to 9.8 M rows); values
    that fill more than a bucket (few distinct values) send it to the library sort.  Expected: numpy's stable sort by the key in Double.compare order, ties in arrival order;
    an INTEGER and a BIGINT channel ride along."""
    rng = np.random.default_rng(n % 1000 + len(spread))
    real = spread.startswith("REAL")
    if spread in ("uniform [0, 1)", "REAL uniform"):
        keys = rng.random(n)
    elif spread == "normal":
        keys = rng.standard_normal(n) * 1e6
    elif spread == "exponential":
        keys = rng.exponential(1e-3, n)
    elif spread == "signed, huge range":
        keys = rng.standard_normal(n) * 10.0 ** rng.integers(-200, 200, n)
    elif spread == "specials":
        keys = rng.standard_normal(n)
        keys[rng.integers(0, n, n // 50)] = rng.choice([0.0, -0.0, np.inf, -np.inf, np.nan], n // 50)
    else:
        keys = rng.integers(0, 7, n).astype(np.float64) * 1.5
    if real:
        keys = keys.astype(np.float32)
    descending = n % 2 == 0
    t = abi.REAL if real else abi.DOUBLE
    page = Page([Block.flat(t, keys), Block.integer(np.arange(n, dtype=np.int32)), Block.bigint(np.arange(n, dtype=np.int64) * 3)], n)
    op = OrderByOperator([t, abi.INTEGER, abi.BIGINT], [0, 1, 2], [0], [abi.DESC_NULLS_LAST if descending else abi.ASC_NULLS_LAST])
    out = to_pages(op, [page])
    name = op.kernelName()
    op.close()
    # Double.compare order as an integer image: -0.0 < 0.0, one NaN above everything
    wide = keys.astype(np.float64)
    bits = np.where(np.isnan(wide), np.float64(np.nan), wide).view(np.int64).copy()
    bits[np.isnan(wide)] = 0x7ff8000000000000
    image = np.where(bits < 0, ~bits, bits | np.int64(-2**63)).view(np.uint64)
    order = np.argsort(~image if descending else image, kind="stable")
    got_rows = np.concatenate([p.blocks[1].values for p in out])
    assert np.array_equal(got_rows, order.astype(np.int32))
    assert np.array_equal(np.concatenate([p.blocks[2].values for p in out]), order.astype(np.int64) * 3)
    got_keys = np.concatenate([p.blocks[0].values for p in out])
    assert np.array_equal(got_keys.view(np.uint32 if real else np.uint64), keys[order].view(np.uint32 if real else np.uint64))
    crowded = (spread == "few values" and n // 7 > 2048) or (spread == "specials" and n // 50 // 5 > 2048)   # (a value that fills more than the LDS copy)
    assert name == ("rocprim_radix_sort_pairs" if crowded or n > 9_830_400 else "pa_sort_buckets"), (name, n, spread)
