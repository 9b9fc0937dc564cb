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
