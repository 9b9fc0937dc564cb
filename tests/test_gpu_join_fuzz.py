"""Randomised joins against the oracle: one or two key channels of every key type (one integer key = the keyed probe-side
table with its existence bitmap; anything else = the tagged table), duplicates on both sides, NULL keys, sparse and dense key
ranges (bitmap or not), every join type, outputSingleMatch, several build and probe pages, a $hashvalue channel or not -- output
rows AND their order, and the lookup-outer rows."""
import os

import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import HashBuilderOperator, LookupJoinOperator, LookupOuterOperator, LookupSourceFactory, to_pages
from presto_amd.page import Block, Page

pytestmark = pytest.mark.gpu


def key_block(rng, t, n, card, sparse, null_share, values=None):
    nulls = (rng.random(n) < null_share) if null_share else None
    v = rng.integers(0, card, n) if values is None else values
    if t == abi.BIGINT:
        return Block.bigint(v * (1 << 33 if sparse else 3) - 11, nulls)
    if t == abi.INTEGER:
        return Block.integer(v * (40000 if sparse else 1) - 5, nulls)
    if t == abi.DATE:
        return Block.date(v + 7000, nulls)
    if t == abi.DOUBLE:
        pool = np.concatenate([rng.standard_normal(max(card, 2)), [0.0, -0.0, np.nan]])
        return Block.double(pool[rng.integers(0, len(pool), n)], nulls)
    return Block.varchar([None if (nulls is not None and nulls[i]) else b"k%05d%s" % (x, b"_" * (x % 9)) for i, x in enumerate(v)])


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("PA_FUZZ_SEEDS", "64")))))
def test_random_joins(gpu, oracle, seed):
    rng = np.random.default_rng(5200 + seed)
    kinds = [abi.BIGINT, abi.INTEGER, abi.DATE, abi.DOUBLE, abi.VARCHAR]
    nkeys = 1 if seed % 3 else 2
    key_types = [kinds[i] for i in rng.choice(len(kinds), nkeys, replace=False)] if nkeys == 2 else [kinds[int(rng.integers(0, 3)) if seed % 2 else int(rng.integers(0, 5))]]
    card = int(rng.choice([3, 50, 4000, 60000]))
    sparse = bool(rng.integers(0, 2))
    join_type = int(rng.choice([abi.JOIN_INNER, abi.JOIN_PROBE_OUTER, abi.JOIN_LOOKUP_OUTER, abi.JOIN_FULL_OUTER]))
    single = bool(rng.integers(0, 4) == 0)
    hashed = bool(rng.integers(0, 2))
    types = key_types + [abi.BIGINT]
    join_ch = list(range(nkeys))

    # one integer key, now and then without duplicate or NULL build keys (in key order or not): dense enough, such a build side is
    # looked up through the key rank index instead of the table
    unique = nkeys == 1 and key_types[0] in (abi.BIGINT, abi.INTEGER, abi.DATE) and card >= 4000 and bool(rng.integers(0, 3) == 0)
    pool = rng.permutation(card)
    if unique and rng.integers(0, 2):
        pool = np.sort(pool[: card // 2])
    taken = 0

    def pages(count, rows_choice, null_share, build_side=False):
        nonlocal taken
        out = []
        for _ in range(count):
            n = int(rng.choice(rows_choice))
            values = None
            if unique and build_side:
                n = min(n, len(pool) - taken)
                if n == 0:
                    continue
                values = pool[taken:taken + n]
                taken += n
            blocks = [key_block(rng, t, n, card, sparse, 0 if values is not None else (null_share if rng.random() < 0.6 else 0), values) for t in key_types]
            p = Page(blocks + [Block.bigint(rng.integers(0, 1 << 40, n))], n)
            if hashed:
                p = Page(p.blocks + [Block.bigint(oracle.hash_page(p, join_ch))], n)
            out.append(p)
        return out

    # (few distinct keys = long chains: keep the pages small there, the output has build x probe / card rows)
    build = pages(int(rng.integers(1, 4)), [1, 40, 300] if card < 100 else [1, 40, 3000, 20000], 0.05, build_side=True)
    probe = pages(int(rng.integers(1, 4)), [1, 60, 2000] if card < 100 else [1, 60, 5000, 50000], 0.05)
    ptypes = types + ([abi.BIGINT] if hashed else [])
    hc = len(types) if hashed else -1
    out_ch = list(range(len(types)))
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, ptypes, join_ch, out_ch, hash_channel=hc), build)
    join = LookupJoinOperator(bridge, ptypes, join_ch, out_ch, probe_hash_channel=hc, join_type=join_type, output_single_match=single)
    got = [r for p in to_pages(join, probe) for r in p.to_rows()]
    ref = oracle.HashJoin(ptypes, join_ch, out_ch, hash_channel=hc)
    for p in build:
        ref.add_build_page(p)
    ref.build()
    expected = [r for p in probe for r in ref.probe(p, ptypes, join_ch, out_ch, hc, join_type=join_type, output_single_match=single)[0].to_rows()]

    def norm(rows):  # NaN != NaN in tuple comparison
        return [tuple("nan" if isinstance(v, float) and v != v else v for v in r) for r in rows]

    assert norm(got) == norm(expected)
    if join_type in (abi.JOIN_LOOKUP_OUTER, abi.JOIN_FULL_OUTER):
        outer = [r for p in to_pages(LookupOuterOperator(bridge, ptypes, out_ch, join_type=join_type), []) for r in p.to_rows()]
        assert sorted(norm(outer), key=repr) == sorted(norm(ref.outer(ptypes, out_ch).to_rows()), key=repr)
