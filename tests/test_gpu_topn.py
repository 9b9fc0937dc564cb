"""TopNOperator on device against the reference's known answers (core/trino-main/src/test/java/io/trino/operator/
TestTopNOperator.java:77-182) and the oracle's restatement on random pages (every type as first sort key, NULL
placement, ties on the first key resolved by the second)."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import TopNOperator, download_page, to_pages
from presto_amd.page import Block, Page

pytestmark = pytest.mark.gpu


def rows_of(pages):
    return [r for p in pages for r in p.to_rows()]


def test_topn_kats(gpu, oracle):
    pages = [Page([Block.bigint([1, 2]), Block.double([0.1, 0.2])]), Page([Block.bigint([-1, 4]), Block.double([-0.1, 0.4])]),
             Page([Block.bigint([5, 4, 6]), Block.double([0.5, 0.41, 0.6])])]
    types = [abi.BIGINT, abi.DOUBLE]
    # testSingleFieldKey (:77-104)
    assert rows_of(to_pages(TopNOperator(types, 2, [0], [abi.DESC_NULLS_LAST]), pages)) == [(6, 0.6), (5, 0.5)]
    # testReverseOrder (:136-164)
    assert rows_of(to_pages(TopNOperator(types, 2, [0], [abi.ASC_NULLS_LAST]), pages)) == [(-1, -0.1), (1, 0.1)]
    # testMultiFieldKey (:106-134)
    vp = [Page([Block.varchar(["a", "b"]), Block.bigint([1, 2])]), Page([Block.varchar(["f", "a"]), Block.bigint([3, 4])]),
          Page([Block.varchar(["d", "d", "e"]), Block.bigint([5, 7, 6])])]
    got = rows_of(to_pages(TopNOperator([abi.VARCHAR, abi.BIGINT], 3, [0, 1], [abi.DESC_NULLS_LAST, abi.DESC_NULLS_LAST]), vp))
    assert got == [(b"f", 3), (b"e", 6), (b"d", 7)]
    # testLimitZero (:166-182)
    op = TopNOperator([abi.BIGINT], 0, [0], [abi.DESC_NULLS_LAST])
    assert op.getOutput() is None and op.isFinished() and not op.needsInput() and op.getOutput() is None
    # the oracle agrees on the same inputs
    assert oracle.topn(pages, 2, [0], [abi.DESC_NULLS_LAST]) == [(6, 0.6), (5, 0.5)]
    assert oracle.topn(vp, 3, [0, 1], [abi.DESC_NULLS_LAST] * 2) == got


@pytest.mark.parametrize("first", ["bigint", "double", "varchar", "date", "boolean"])
@pytest.mark.parametrize("order", [abi.ASC_NULLS_FIRST, abi.ASC_NULLS_LAST, abi.DESC_NULLS_FIRST, abi.DESC_NULLS_LAST])
def test_topn_matches_oracle(gpu, oracle, first, order):
    rng = np.random.default_rng(hash((first, order)) % 2 ** 32)
    words = [b"", b"a", b"ab", b"abcdefgh", b"abcdefghX", b"abcdefghY", b"zz", b"\xc3\xa9", None]
    pages = []
    for _ in range(5):
        n = int(rng.integers(1, 30000))
        nulls = rng.random(n) < 0.02
        if first == "bigint":
            k = Block.bigint(rng.integers(-2 ** 62, 2 ** 62, n), nulls)
        elif first == "double":
            v = rng.standard_normal(n) * 1e3
            v[rng.random(n) < 0.01] = np.nan
            v[rng.random(n) < 0.01] = -0.0
            v[rng.random(n) < 0.01] = 0.0
            k = Block.double(v, nulls)
        elif first == "varchar":
            k = Block.varchar([words[j] for j in rng.integers(0, len(words), n)])
        elif first == "date":
            k = Block.date(rng.integers(8000, 8100, n), nulls)     # few distinct values: many ties on the first key
        else:
            k = Block.boolean(rng.random(n) < 0.5, nulls)
        pages.append(Page([k, Block.integer(rng.permutation(n).astype(np.int32) + 1000000 * len(pages)), Block.double(rng.random(n))], n))
    types = [pages[0].blocks[0].type, abi.INTEGER, abi.DOUBLE]
    for limit in (1, 10, 2500):
        orders = [order, abi.DESC_NULLS_LAST]   # the second channel is unique: no fully tied rows
        expected = oracle.topn(pages, limit, [0, 1], orders)
        got = rows_of(to_pages(TopNOperator(types, limit, [0, 1], orders), pages))
        assert len(got) == len(expected)
        for g, e in zip(got, expected):
            assert g[1] == e[1] and g[2] == e[2], (g, e)
            assert g[0] == e[0] or (isinstance(e[0], float) and np.isnan(e[0]) and np.isnan(g[0]))


def test_topn_device_pages_and_device_output(gpu, oracle):
    from presto_amd.operators import upload_page
    rng = np.random.default_rng(3)
    pages = [Page([Block.double(rng.random(200000)), Block.bigint(np.arange(200000) + 200000 * i)], 200000) for i in range(3)]
    op = TopNOperator([abi.DOUBLE, abi.BIGINT], 10, [0, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST], output_mem=abi.MEM_DEVICE)
    dev_pages = [upload_page(p) for p in pages]
    for p in dev_pages:
        op.addInput(p)
    op.finish()
    out = download_page(op.getOutput())
    assert out.to_rows() == oracle.topn(pages, 10, [0, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST])
    assert op.isFinished()


@pytest.mark.parametrize("shape", ["all_equal", "two_values", "three_channels", "fully_tied", "varchar_second"])
def test_topn_with_masses_of_ties_on_the_first_key(gpu, oracle, shape):
    """ORDER BY <low-cardinality column>, ... LIMIT n: hundreds of thousands of rows tie with the bound on the first key.  The page's
    ties are cut down on the device, channel by channel (a page contributes at most its own n best rows), instead of crossing to
    the host whole; fully tied rows come out in arrival order.  Results equal the oracle's, and the call stays fast."""
    import time
    rng = np.random.default_rng(len(shape))
    n = 120000
    pages = []
    for p in range(3):
        first = {"all_equal": np.zeros(n), "two_values": (rng.random(n) < 0.5).astype(np.float64), "three_channels": np.zeros(n),
                 "fully_tied": np.zeros(n), "varchar_second": np.zeros(n)}[shape]
        second = rng.integers(0, 5, n) if shape in ("three_channels", "varchar_second") else rng.permutation(n) + p * n
        if shape == "fully_tied":
            second = np.zeros(n, dtype=np.int64)
        blocks = [Block.double(first, rng.random(n) < 0.001), Block.bigint(second), Block.integer(rng.permutation(n).astype(np.int32) + p * n)]
        if shape == "varchar_second":
            blocks[1] = Block.varchar([[b"abcdefghA", b"abcdefghB", b"x"][j] for j in rng.integers(0, 3, n)])
        pages.append(Page(blocks, n))
    types = [abi.DOUBLE, abi.VARCHAR if shape == "varchar_second" else abi.BIGINT, abi.INTEGER]
    for limit, orders in ((10, [abi.ASC_NULLS_LAST, abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST]), (2000, [abi.DESC_NULLS_FIRST, abi.ASC_NULLS_LAST, abi.DESC_NULLS_LAST])):
        expected = oracle.topn(pages, limit, [0, 1, 2], orders)
        t0 = time.perf_counter()
        got = rows_of(to_pages(TopNOperator(types, limit, [0, 1, 2], orders), pages))
        elapsed = time.perf_counter() - t0
        if shape == "fully_tied":   # the third channel decides (unique); rows tied on all three do not exist
            assert got == expected
        else:
            assert got == expected, (got[:3], expected[:3])
        assert elapsed < (20.0 if shape == "varchar_second" else 3.0), elapsed


@pytest.mark.parametrize("shape", ["random", "ascending", "descending", "fifty_values", "three_values", "nulls_first"])
def test_topn_large_pages_take_the_sampled_single_pass(gpu, oracle, shape):
    """Pages of >= 2^18 rows: the page's bound is drawn from a sample and the candidates are collected in one pass over the first sort
    channel (op_topn.cpp add_filtered); inputs that get better page after page, ties with the bound (within what the sample
    promised, and beyond it: the exact way takes over) and NULLs sorting first must all give the oracle's rows in the oracle's
    order.  (The oracle is handed the rows that can matter -- everything not worse than the 2 N-th best first key, in arrival
    order -- so that it finishes in seconds.)"""
    rng = np.random.default_rng(len(shape) + 40)
    n = 270_001 if shape == "three_values" else 700_001
    pages = []
    for p in range(1 if shape == "three_values" else 3):
        if shape == "random":
            first = rng.standard_normal(n) * 1e6
        elif shape == "ascending":      # DESC order below: every page holds better rows than all before it
            first = np.arange(n, dtype=np.float64) + p * n
        elif shape == "descending":
            first = -(np.arange(n, dtype=np.float64) + p * n)
        elif shape == "fifty_values":
            first = rng.integers(0, 50, n).astype(np.float64)
        elif shape == "three_values":
            first = rng.integers(0, 3, n).astype(np.float64)
        else:
            first = rng.random(n)
        nulls = (rng.random(n) < (0.01 if shape == "nulls_first" else 0.001))
        pages.append(Page([Block.double(first, nulls), Block.bigint(rng.permutation(n) + p * n)], n))
    types = [abi.DOUBLE, abi.BIGINT]
    nulls_first = shape == "nulls_first"
    orders = [abi.DESC_NULLS_FIRST if nulls_first else abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST]
    first_all = np.concatenate([p.blocks[0].values for p in pages])
    nulls_all = np.concatenate([p.blocks[0].nulls for p in pages])
    second_all = np.concatenate([p.blocks[1].values for p in pages])
    score = np.where(nulls_all != 0, np.inf if nulls_first else -np.inf, first_all)
    for limit in (100, 20_000):
        cut = np.partition(score, len(score) - 2 * limit)[len(score) - 2 * limit]
        keep = score >= cut
        relevant = Page([Block.double(first_all[keep], nulls_all[keep] != 0), Block.bigint(second_all[keep])], int(keep.sum()))
        expected = oracle.topn([relevant], limit, [0, 1], orders)
        got = rows_of(to_pages(TopNOperator(types, limit, [0, 1], orders), pages))
        assert got == expected, (shape, limit, got[:3], expected[:3])


@pytest.mark.parametrize("shape", ["sum_desc", "key_asc_ties", "count_desc_few_values", "nullable_min_desc_nulls_first"])
def test_aggregation_told_that_a_topn_is_its_only_consumer(gpu, oracle, shape):
    """pa_aggregation_set_output_topn_hint: a HashAggregation over hundreds of thousands of groups whose output feeds a TopN emits
    only the groups that can be among the n best (a bound on the first sort channel drawn from a sample of the table); the TopN's
    result must equal the one over every group -- sort channel an aggregate or a key, ties on the first channel (second channel
    decides), massive ties (fewer distinct values than n: the sample cannot vouch, everything is emitted), NULL results first."""
    from presto_amd.operators import HashAggregationOperator
    rng = np.random.default_rng(len(shape) + 3)
    rows, groups = 1_500_000, 400_000
    keys = rng.integers(0, groups, rows).astype(np.int64) * 13 - 7
    vals = rng.standard_normal(rows) * 100
    vnull = rng.random(rows) < (0.5 if shape.startswith("nullable") else 0.0)
    small = rng.integers(0, 3, rows).astype(np.int64)
    page = Page([Block.bigint(keys), Block.double(vals, vnull if vnull.any() else None), Block.bigint(small)], rows)
    types = [abi.BIGINT, abi.DOUBLE, abi.BIGINT]
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MIN, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.BIGINT)]
    out_types = [abi.BIGINT, abi.DOUBLE, abi.BIGINT, abi.DOUBLE, abi.BIGINT]
    sort = {"sum_desc": ([1, 0], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST]), "key_asc_ties": ([4, 0], [abi.ASC_NULLS_LAST, abi.DESC_NULLS_LAST]),
            "count_desc_few_values": ([2, 0], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST]),
            "nullable_min_desc_nulls_first": ([3, 0], [abi.DESC_NULLS_FIRST, abi.ASC_NULLS_LAST])}[shape]
    for limit in (10, 3000):
        results = []
        for hint in (False, True):
            agg = HashAggregationOperator(types, [0], aggs, expected_groups=groups, output_mem=abi.MEM_DEVICE)
            if hint:
                assert agg.setOutputTopNHint(limit, sort[0], sort[1]) is True
            grouped = to_pages(agg, [page])
            emitted = sum(p.position_count for p in grouped)
            top = TopNOperator(out_types, limit, sort[0], sort[1])
            results.append((rows_of(to_pages(top, grouped)), emitted))
            agg.close()
        (every, n_all), (hinted, n_hint) = results
        assert n_all > 300_000 and len(every) == limit
        assert len(hinted) == limit
        for a, b in zip(hinted, every):
            assert a[0] == b[0] and a[2] == b[2] and a[4] == b[4] and (a[3] == b[3] or (a[3] is None and b[3] is None))
            assert a[1] is None and b[1] is None or abs(a[1] - b[1]) <= 1e-9 * max(abs(b[1]), 1e-300)
        if shape in ("sum_desc", "nullable_min_desc_nulls_first") or (shape == "key_asc_ties" and False):
            assert n_hint < n_all // 4, (n_hint, n_all)   # the bound did cut the output down
