"""GPU parity of HashBuilderOperator + LookupJoinOperator against the oracle's PagesHash / DefaultPageJoiner
restatement: identical output rows in identical order (ascending probe position, chains in descending build
position), identical positionLinks, and the reference's own known-answer cases (SURVEY 9.5 join-1, join-null)."""
import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import HashBuilderOperator, LookupJoinOperator, LookupSourceFactory, to_pages
from presto_amd.page import Block, Page, sequence_page

pytestmark = pytest.mark.gpu


def gpu_join(build_pages, build_types, join_ch, out_ch, probe_pages, probe_types, probe_join_ch, probe_out_ch, build_hash=-1, probe_hash=-1):
    bridge = LookupSourceFactory()
    builder = HashBuilderOperator(bridge, build_types, join_ch, out_ch, hash_channel=build_hash)
    to_pages(builder, build_pages)
    join = LookupJoinOperator(bridge, probe_types, probe_join_ch, probe_out_ch, probe_hash_channel=probe_hash)
    rows, pairs = [], []
    for p in probe_pages:
        assert join.needsInput()
        join.addInput(p)
        out = join.getOutput()
        if out is not None:
            rows += out.to_rows()
        pairs.append(join.matchPairs())
    join.finish()
    assert join.isFinished()
    return rows, pairs, bridge


def oracle_join(oracle, build_pages, build_types, join_ch, out_ch, probe_pages, probe_types, probe_join_ch, probe_out_ch, build_hash=-1, probe_hash=-1):
    j = oracle.HashJoin(build_types, join_ch, out_ch, hash_channel=build_hash)
    for p in build_pages:
        j.add_build_page(p)
    j.build()
    rows, pairs = [], []
    for p in probe_pages:
        out, pi, bi = j.probe(p, probe_types, probe_join_ch, probe_out_ch, probe_hash)
        rows += out.to_rows()
        pairs.append((pi, bi))
    return rows, pairs, j


def test_join1_kat(gpu, oracle):
    """TestHashJoinOperator.testInnerJoin (…/join/TestHashJoinOperator.java:192-229): build (VARCHAR,BIGINT,BIGINT)
    10 rows @20,30,40; probe 1000 rows @0,1000,2000; key = channel 0 -> 10 rows for keys '20'..'29'."""
    types = [abi.VARCHAR, abi.BIGINT, abi.BIGINT]
    build = [sequence_page(10, [(abi.VARCHAR, 20), (abi.BIGINT, 30), (abi.BIGINT, 40)])]
    probe = [sequence_page(1000, [(abi.VARCHAR, 0), (abi.BIGINT, 1000), (abi.BIGINT, 2000)])]
    rows, pairs, _ = gpu_join(build, types, [0], [0, 1, 2], probe, types, [0], [0, 1, 2])
    assert rows == [(str(20 + i).encode(), 1020 + i, 2020 + i, str(20 + i).encode(), 30 + i, 40 + i) for i in range(10)]
    orows, opairs, _ = oracle_join(oracle, build, types, [0], [0, 1, 2], probe, types, [0], [0, 1, 2])
    assert rows == orows


@pytest.mark.parametrize("hashed", [False, True])
def test_duplicates_nulls_and_chain_order(gpu, oracle, hashed):
    rng = np.random.default_rng(3)
    nb, npr = 30011, 50021
    bkeys = rng.integers(0, 4000, nb)
    bnull = rng.random(nb) < 0.05
    pkeys = rng.integers(-100, 4500, npr)
    pnull = rng.random(npr) < 0.05
    build = Page([Block.bigint(bkeys, bnull), Block.double(rng.random(nb)), Block.integer(np.arange(nb))], nb)
    probe = Page([Block.integer(np.arange(npr)), Block.bigint(pkeys, pnull)], npr)
    btypes, ptypes = [abi.BIGINT, abi.DOUBLE, abi.INTEGER], [abi.INTEGER, abi.BIGINT]
    bh = ph = -1
    if hashed:
        build = Page(build.blocks + [Block.bigint(oracle.hash_page(build, [0]))], nb)
        probe = Page(probe.blocks + [Block.bigint(oracle.hash_page(probe, [1]))], npr)
        btypes, ptypes, bh, ph = btypes + [abi.BIGINT], ptypes + [abi.BIGINT], 3, 2
    # build arrives in several pages; probe in two
    bpages = [build.get_region(0, 10000), build.get_region(10000, 1), build.get_region(10001, nb - 10001)]
    ppages = [probe.get_region(0, 25000), probe.get_region(25000, npr - 25000)]
    rows, pairs, bridge = gpu_join(bpages, btypes, [0], [1, 2], ppages, ptypes, [1], [0, 1], bh, ph)
    orows, opairs, oj = oracle_join(oracle, bpages, btypes, [0], [1, 2], ppages, ptypes, [1], [0, 1], bh, ph)
    assert len(rows) == len(orows) and len(rows) > npr  # many-to-many
    assert rows == orows
    for (gp, gb), (op_, ob) in zip(pairs, opairs):
        assert np.array_equal(gp, op_) and np.array_equal(gb, ob)
    # positionLinks are the reference's: every chain descends from the highest position of its key
    key, links = bridge.tables()
    okey, olinks = oj.tables()
    assert np.array_equal(links, olinks)
    assert len(key) == len(okey) and sorted(key[key >= 0].tolist()) == sorted(okey[okey >= 0].tolist())


def test_multi_channel_keys_with_varchar_and_double(gpu, oracle):
    rng = np.random.default_rng(9)
    nb, npr = 4001, 9001
    words = [b"a", b"bb", b"ccc", b"", b"dddd"]
    def mk(n):
        return Page([Block.varchar([words[i] for i in rng.integers(0, 5, n)]),
                     Block.double(rng.integers(0, 4, n).astype(np.float64) * 0.5),
                     Block.bigint(np.arange(n))], n)
    build, probe = mk(nb), mk(npr)
    types = [abi.VARCHAR, abi.DOUBLE, abi.BIGINT]
    rows, pairs, _ = gpu_join([build], types, [0, 1], [2, 0], [probe], types, [0, 1], [2, 0, 1])
    orows, opairs, _ = oracle_join(oracle, [build], types, [0, 1], [2, 0], [probe], types, [0, 1], [2, 0, 1])
    assert rows == orows


def test_empty_build_and_no_match(gpu, oracle):
    types = [abi.BIGINT]
    rows, pairs, _ = gpu_join([], types, [0], [0], [sequence_page(100, [(abi.BIGINT, 0)])], types, [0], [0])
    assert rows == []
    rows, pairs, _ = gpu_join([sequence_page(10, [(abi.BIGINT, 1000)])], types, [0], [0], [sequence_page(100, [(abi.BIGINT, 0)])], types, [0], [0])
    assert rows == []
