"""Randomised group-by shapes against the oracle: 1-3 key columns drawn from every key type (nullable BIGINT, INTEGER, DATE,
nullable BOOLEAN, DOUBLE with -0.0 / NaN, short VARCHAR with a declared length bound, longer VARCHAR, interned text), random aggregates
(count(*), count, sum, avg, min, max; masks; nullable inputs), cardinalities from a handful to tens of thousands -- so the
key packing (bit fields, two-word strings, NULL flags) and every tier of the aggregation see shapes nobody wrote by hand."""
import os

import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import HashAggregationOperator, to_pages
from presto_amd.page import Block, Page

pytestmark = pytest.mark.gpu

WORDS_SHORT = [b"", b"A", b"F", b"N", b"ab", b"abc", b"xyzw", b"1234567", None]
WORDS_LONG = [b"", b"BUILDING", b"AUTOMOBILE", b"MACHINERY", b"0123456789abcde", b"x", None, b"HOUSEHOLD"]


def key_column(rng, kind, n, card):
    if kind == "bigint":
        return abi.BIGINT, Block.bigint(rng.integers(0, card, n) * 1000003 - 5, rng.random(n) < 0.02), 0
    if kind == "integer":
        return abi.INTEGER, Block.integer(rng.integers(-card // 2, card // 2 + 1, n)), 0
    if kind == "date":
        return abi.DATE, Block.date(rng.integers(8000, 8000 + max(card, 1), n)), 0
    if kind == "boolean":
        return abi.BOOLEAN, Block.boolean(rng.random(n) < 0.5, rng.random(n) < 0.1), 0
    if kind == "double":
        pool = np.concatenate([rng.standard_normal(max(card, 4)), [0.0, -0.0, np.nan, np.inf]])
        return abi.DOUBLE, Block.double(pool[rng.integers(0, len(pool), n)], rng.random(n) < 0.02), 0
    if kind == "short":
        return abi.VARCHAR, Block.varchar([WORDS_SHORT[i] for i in rng.integers(0, len(WORDS_SHORT), n)]), 7
    if kind == "text":  # longer than the packed key holds: interned on the device
        ids = rng.integers(0, max(card, 1), n)
        nulls = rng.random(n) < 0.02
        return abi.VARCHAR, Block.varchar([None if z else b"Customer#%09d %s" % (i, b"y" * (i % 19)) for i, z in zip(ids, nulls)]), int(rng.choice([0, 64]))
    return abi.VARCHAR, Block.varchar([WORDS_LONG[i] for i in rng.integers(0, len(WORDS_LONG), n)]), 0


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("PA_FUZZ_SEEDS", "24")))))  # PA_FUZZ_SEEDS=N: a longer one-off run
def test_random_group_by_shapes(gpu, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    kinds = ["bigint", "integer", "date", "boolean", "double", "short", "long", "text"]
    nkeys = int(rng.integers(1, 4))
    chosen = [kinds[i] for i in rng.choice(len(kinds), nkeys, replace=False)]
    card = int([3, 40, 700, 20000][seed % 4])
    pages = []
    types, params = None, None
    for _ in range(int(rng.integers(1, 4))):
        n = int(rng.integers(1, 60000))
        blocks, types, params = [], [], []
        state = rng.bit_generator.state
        for kind in chosen:
            t, b, p = key_column(rng, kind, n, card)
            blocks.append(b)
            types.append(t)
            params.append(p)
        # value columns: nullable DOUBLE, BIGINT, INTEGER, and a nullable mask
        blocks += [Block.double(rng.random(n) * 200 - 100, rng.random(n) < 0.1), Block.bigint(rng.integers(-10 ** 6, 10 ** 6, n), rng.random(n) < 0.1),
                   Block.integer(rng.integers(-500, 500, n)), Block.boolean(rng.random(n) < 0.7, rng.random(n) < 0.05)]
        types += [abi.DOUBLE, abi.BIGINT, abi.INTEGER, abi.BOOLEAN]
        params += [0, 0, 0, 0]
        pages.append(Page(blocks, n))
    v0 = nkeys
    pool = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_COUNT, v0, abi.DOUBLE), (abi.AGG_SUM, v0, abi.DOUBLE), (abi.AGG_AVG, v0, abi.DOUBLE),
            (abi.AGG_SUM, v0 + 1, abi.BIGINT), (abi.AGG_AVG, v0 + 1, abi.BIGINT), (abi.AGG_MIN, v0, abi.DOUBLE), (abi.AGG_MAX, v0 + 1, abi.BIGINT),
            (abi.AGG_MIN, v0 + 2, abi.INTEGER), (abi.AGG_SUM, v0 + 2, abi.INTEGER), (abi.AGG_SUM, v0, abi.DOUBLE, v0 + 3), (abi.AGG_COUNT_STAR, -1, None, v0 + 3),
            (abi.AGG_MAX, v0, abi.DOUBLE, v0 + 3)]
    aggs = [pool[i] for i in sorted(rng.choice(len(pool), int(rng.integers(1, 7)), replace=False))]
    keys = list(range(nkeys))
    op = HashAggregationOperator(types, keys, aggs, type_params=params)
    got = [r for p in to_pages(op, pages) for r in p.to_rows()]
    ref = oracle.HashAggregation(types, keys, aggs)
    for p in pages:
        ref.add_page(p)
    expected = ref.build_result().to_rows()

    def key_of(r):  # group keys are exact; a DOUBLE key of -0.0 comes out as +0.0 on the device (same group; DESIGN)
        out = []
        for v in r[:nkeys]:
            if isinstance(v, float):
                out.append("nan" if v != v else repr(0.0 if v == 0.0 else v))
            else:
                out.append(repr(v))
        return tuple(out)

    assert len(got) == len(expected)
    g, e = {key_of(r): r for r in got}, {key_of(r): r for r in expected}
    assert len(g) == len(got) and set(g) == set(e)
    for k, er in e.items():
        for gv, ev in zip(g[k][nkeys:], er[nkeys:]):
            if isinstance(ev, float) and ev == ev:
                assert gv == ev or abs(gv - ev) <= 1e-9 * max(abs(gv), abs(ev)), (k, g[k], er)
            elif isinstance(ev, float):
                assert gv != gv, (k, g[k], er)
            else:
                assert gv == ev, (k, g[k], er)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("PA_FUZZ_SEEDS", "16")))))
def test_random_page_sequences(gpu, oracle, seed):
    """Sequences of pages whose shape changes from page to page: valueIsNull arrays that come and go per column (state layout
    generations, implicit count words), keys in runs / sorted (run combining on the table tier), the VARCHAR key as a plain,
    dictionary or RLE block (dictionary route of the interned keys), tiny and large pages -- against the oracle."""
    rng = np.random.default_rng(9100 + seed)
    card = int([4, 300, 6000, 90000][seed % 4])
    clustered = bool(rng.integers(0, 2))
    text = [b"item-%06d-%s" % (i, b"w" * (i % 21)) for i in range(min(card, 5000))]
    pages = []
    for _ in range(int(rng.integers(2, 6))):
        n = int(rng.choice([3, 70, 5000, 120000]))
        k = rng.integers(0, card, n)
        if clustered:
            k = np.sort(k)
        def nulls(p=0.06):
            return (rng.random(n) < p) if rng.random() < 0.4 else None
        tk = k % len(text)
        form = int(rng.integers(0, 3))
        if form == 0:
            tblock = Block.varchar([None if (rng.random() < 0.02) else text[i] for i in tk])
        elif form == 1:
            tblock = Block.dictionary_block(Block.varchar(text + [None]), np.where(rng.random(n) < 0.02, len(text), tk).astype(np.int32))
        else:
            tblock = Block.rle(Block.varchar([text[int(tk[0])]]), n)
        pages.append(Page([Block.bigint(k * 13 - 7, nulls()), tblock, Block.double(rng.random(n) * 8, nulls()), Block.bigint(rng.integers(-99, 99, n), nulls()),
                           Block.boolean(rng.random(n) < 0.75, nulls())], n))
    types = [abi.BIGINT, abi.VARCHAR, abi.DOUBLE, abi.BIGINT, abi.BOOLEAN]
    keys = [[0], [1], [0, 1]][seed % 3]
    pool = [(abi.AGG_COUNT_STAR, -1, None), (abi.AGG_SUM, 2, abi.DOUBLE), (abi.AGG_SUM, 3, abi.BIGINT), (abi.AGG_MIN, 2, abi.DOUBLE), (abi.AGG_MAX, 3, abi.BIGINT),
            (abi.AGG_AVG, 2, abi.DOUBLE), (abi.AGG_COUNT, 3, abi.BIGINT), (abi.AGG_SUM, 2, abi.DOUBLE, 4)]
    aggs = [pool[i] for i in sorted(rng.choice(len(pool), int(rng.integers(1, 5)), replace=False))]
    op = HashAggregationOperator(types, keys, aggs, expected_groups=card)
    got = [r for p in to_pages(op, pages) for r in p.to_rows()]
    ref = oracle.HashAggregation(types, keys, aggs)
    for p in pages:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    nk = len(keys)
    assert len(got) == len(expected)
    g, e = {r[:nk]: r for r in got}, {r[:nk]: r for r in expected}
    assert len(g) == len(got) and set(g) == set(e)
    for kk, er in e.items():
        for gv, ev in zip(g[kk][nk:], er[nk:]):
            if isinstance(ev, float):
                assert gv == ev or abs(gv - ev) <= 1e-9 * max(abs(gv), abs(ev)), (kk, g[kk], er)
            else:
                assert gv == ev, (kk, g[kk], er)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("PA_FUZZ_SEEDS", "12")))))
def test_random_min_max_over_strings(gpu, oracle, seed):
    """min / max (masked and not) and count over VARCHAR channels of any length -- by rank in the channel's dictionary --
    next to sums, under random group keys (none, integers, short and interned strings), page counts and sizes, NULL shares and pools of
    distinct strings; SINGLE, or PARTIAL per page -> FINAL."""
    from presto_amd.exchange import partial_layout
    rng = np.random.default_rng(9100 + seed)
    kinds = ["bigint", "integer", "short", "text"]
    nkeys = int(rng.integers(0, 3))
    chosen = [kinds[i] for i in rng.choice(len(kinds), nkeys, replace=False)]
    card = int([3, 40, 700, 20000][seed % 4])
    pool_size = int(rng.choice([5, 300, 20000]))
    stems = [b"", b"a", b"ab", b"order comment: ", b"\xf0\x9f\x98\x80", b"\x80\x81", b"zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz"]
    words = [stems[i % len(stems)] + bytes(rng.integers(0, 256, int(rng.integers(0, 30))).astype(np.uint8)) for i in range(pool_size)]
    null_share = float(rng.choice([0.0, 0.05, 0.6]))
    pages = []
    types, params = None, None
    for k in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, 50000))
        blocks, types, params = [], [], []
        for kind in chosen:
            t, b, p = key_column(rng, kind, n, card)
            blocks.append(b)
            types.append(t)
            params.append(p)
        # page k draws from a growing prefix of the pool: later pages bring strings the earlier ones did not hold
        upto = max(1, pool_size * (k + 1) // 4)
        strings = [None if z else words[i] for i, z in zip(rng.integers(0, min(upto, pool_size), n), rng.random(n) < (null_share if k != 1 else 0.0))]
        blocks += [Block.varchar(strings), Block.double(rng.random(n) * 10, rng.random(n) < 0.1), Block.boolean(rng.random(n) < 0.6, rng.random(n) < 0.05)]
        types += [abi.VARCHAR, abi.DOUBLE, abi.BOOLEAN]
        params += [int(rng.choice([0, 64])), 0, 0]
        pages.append(Page(blocks, n))
    v = nkeys
    pool = [(abi.AGG_MIN, v, abi.VARCHAR), (abi.AGG_MAX, v, abi.VARCHAR), (abi.AGG_COUNT, v, abi.VARCHAR), (abi.AGG_MAX, v, abi.VARCHAR, v + 2),
            (abi.AGG_MIN, v, abi.VARCHAR, v + 2), (abi.AGG_SUM, v + 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    picks = sorted(set([int(rng.integers(0, 2))] + [int(i) for i in rng.choice(len(pool), int(rng.integers(1, 5)), replace=False)]))
    aggs = [pool[i] for i in picks]
    keys = list(range(nkeys))
    params[v] = params[v] if nkeys else 0
    ref = oracle.HashAggregation(types, keys, aggs)
    for p in pages:
        ref.add_page(p)
    expected = ref.build_result().to_rows()
    masked = any(len(a) > 3 for a in aggs)
    if seed % 3 == 2 and not masked:
        ptypes, faggs = partial_layout([types[i] for i in keys], aggs)
        partials = []
        for p in pages:
            partials += to_pages(HashAggregationOperator(types, keys, aggs, step=abi.STEP_PARTIAL, type_params=params), [p])
        pparams = [params[i] for i in keys] + [0] * (len(ptypes) - nkeys)
        got = [r for p in to_pages(HashAggregationOperator(ptypes, keys, faggs, step=abi.STEP_FINAL, type_params=pparams), partials) for r in p.to_rows()]
    else:
        got = [r for p in to_pages(HashAggregationOperator(types, keys, aggs, type_params=params), pages) for r in p.to_rows()]
    assert len(got) == len(expected)
    g, e = {tuple(map(repr, r[:nkeys])): r for r in got}, {tuple(map(repr, r[:nkeys])): r for r in expected}
    assert len(g) == len(got) and set(g) == set(e)
    for k, er in e.items():
        for gv, ev in zip(g[k][nkeys:], er[nkeys:]):
            if isinstance(ev, float):
                assert gv == ev or abs(gv - ev) <= 1e-9 * max(abs(gv), abs(ev)), (k, g[k], er)
            else:
                assert gv == ev, (k, g[k], er)
