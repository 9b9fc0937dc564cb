"""valueIsNull arrays that come and go between the pages of one operator (a block without NULLs has none,
LongArrayBlock.java:37-41): every operator that keeps rows or state across pages -- OrderBy, TopN, the join build and probe,
MergePages behind FilterAndProject -- against the oracle.  (The aggregation's case is in test_gpu_fused.py.)"""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.expr import field
from presto_amd.operators import (FilterAndProjectOperator, HashBuilderOperator, LookupJoinOperator, LookupSourceFactory, OrderByOperator,
                                  TopNOperator, to_pages)
from presto_amd.page import Block, Page

pytestmark = pytest.mark.gpu


def make_pages(rng, patterns, rows=4000, card=500):
    pages = []
    for null_a, null_b, null_s in patterns:
        def nulls(on):
            return (rng.random(rows) < 0.1) if on else None
        strs = [None if (null_s and rng.random() < 0.1) else b"s%03d" % v for v in rng.integers(0, card, rows)]
        pages.append(Page([Block.bigint(rng.integers(0, card, rows), nulls(null_a)), Block.double(rng.random(rows), nulls(null_b)), Block.varchar(strs)], rows))
    return pages


PATTERNS = [(0, 0, 0), (1, 0, 0), (0, 0, 0), (0, 1, 1), (1, 1, 0), (0, 0, 0)]
TYPES = [abi.BIGINT, abi.DOUBLE, abi.VARCHAR]


def test_order_by_and_topn(gpu, oracle):
    rng = np.random.default_rng(1)
    pages = make_pages(rng, PATTERNS)
    orders = [abi.ASC_NULLS_LAST, abi.DESC_NULLS_FIRST, abi.ASC_NULLS_FIRST]
    got = [r for p in to_pages(OrderByOperator(TYPES, [0, 1, 2], [0, 2, 1], orders), pages) for r in p.to_rows()]
    assert got == oracle.order_by(pages, [0, 1, 2], [0, 2, 1], orders)
    got = [r for p in to_pages(TopNOperator(TYPES, 300, [1, 0, 2], orders), pages) for r in p.to_rows()]
    assert got == oracle.topn(pages, 300, [1, 0, 2], orders)


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_PROBE_OUTER])
def test_join_build_and_probe(gpu, oracle, join_type):
    rng = np.random.default_rng(2)
    build = make_pages(rng, PATTERNS, rows=3000, card=2000)
    probe = make_pages(rng, PATTERNS[::-1], rows=5000, card=2000)
    bridge = LookupSourceFactory()
    builder = HashBuilderOperator(bridge, TYPES, [0], [0, 1, 2])
    to_pages(builder, build)
    join = LookupJoinOperator(bridge, TYPES, [0], [0, 1, 2], join_type=join_type)
    got = [r for p in to_pages(join, probe) for r in p.to_rows()]
    ref = oracle.HashJoin(TYPES, [0], [0, 1, 2])
    for p in build:
        ref.add_build_page(p)
    ref.build()
    expected = [r for p in probe for r in ref.probe(p, TYPES, [0], [0, 1, 2], join_type=join_type)[0].to_rows()]
    assert got == expected and len(got) > 1000


def test_merge_pages(gpu, oracle):
    rng = np.random.default_rng(3)
    pages = make_pages(rng, PATTERNS, rows=700)
    proj = [field(i, t) for i, t in enumerate(TYPES)]
    op = FilterAndProjectOperator(TYPES, field(0, abi.BIGINT) < 400, proj, min_output_page_size=1 << 14, min_output_page_row_count=2500, max_output_page_size=1 << 15)
    got = [p.to_rows() for p in to_pages(op, pages)]
    m = oracle.MergePages(1 << 14, 2500, 1 << 15)
    expected = []
    for p in pages:
        q = oracle.filter_project(p, field(0, abi.BIGINT) < 400, proj)
        if q is not None and q.position_count:
            expected += m.process(q)
    expected = [p.to_rows() for p in expected + m.finish()]
    assert got == expected and len(got) >= 2
