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


@pytest.mark.parametrize("duplicates", [False, True])
def test_extreme_key_values(gpu, oracle, duplicates):
    """Extreme key values (Long.MIN_VALUE, -1, 0, Long.MAX_VALUE) on both sides of a keyed join, with and without chains: no key
    value may double as an "empty slot" mark."""
    rng = np.random.default_rng(21)
    low = -2 ** 63
    bkeys = np.concatenate([rng.permutation(5000)[:3000], [low] * (3 if duplicates else 1), [0, -1, 2 ** 63 - 1]]).astype(np.int64)
    if duplicates:
        bkeys = np.concatenate([bkeys, [7, 7, 2 ** 63 - 1]]).astype(np.int64)
    order = rng.permutation(len(bkeys))
    bkeys = bkeys[order]
    nb = len(bkeys)
    build = [Page([Block.bigint(bkeys), Block.integer(np.arange(nb))], nb)]
    pkeys = np.concatenate([rng.integers(-10, 5010, 4000), [low, low, 0, -1, 2 ** 63 - 1, low + 1]]).astype(np.int64)
    pkeys = pkeys[rng.permutation(len(pkeys))]
    probe = [Page([Block.bigint(pkeys), Block.integer(np.arange(len(pkeys)))], len(pkeys))]
    types = [abi.BIGINT, abi.INTEGER]
    rows, pairs, _ = gpu_join(build, types, [0], [1], probe, types, [0], [0, 1])
    orows, opairs, _ = oracle_join(oracle, build, types, [0], [1], probe, types, [0], [0, 1])
    assert rows == orows and sum(1 for r in rows if r[0] == low) == (6 if duplicates else 2)
    for (gp, gb), (op_, ob) in zip(pairs, opairs):
        assert np.array_equal(gp, op_) and np.array_equal(gb, ob)


@pytest.mark.parametrize("key_type,duplicates", [(abi.BIGINT, False), (abi.INTEGER, False), (abi.BIGINT, True)])
def test_large_build_side_is_built_in_partitions(gpu, oracle, key_type, duplicates, monkeypatch):
    """>= 2^20 build rows with one integer key: the probe-side table is assembled partition by partition in LDS (rows regrouped by
    the partition of their home slot; probe sequences wrap inside a partition).  Same rows, same (probe, build) pairs as the oracle
    -- and as the table built slot by slot with atomics in HBM, which duplicate keys still get."""
    rng = np.random.default_rng(31)
    nb, npr = (1 << 20) + 12345, 300000
    if key_type == abi.BIGINT:
        keys = rng.permutation(3 * nb)[:nb].astype(np.int64) * 977 - 10 ** 9
    else:
        keys = (rng.permutation(3 * nb)[:nb] - nb).astype(np.int32)
    if duplicates:
        keys[rng.integers(0, nb, 1000)] = keys[rng.integers(0, nb, 1000)]
    kb = Block.bigint if key_type == abi.BIGINT else Block.integer
    build = [Page([kb(keys, rng.random(nb) < 0.01), Block.integer(np.arange(nb))], nb)]
    pk = np.where(rng.random(npr) < 0.6, keys[rng.integers(0, nb, npr)], rng.integers(-2 ** 31, 2 ** 31 - 1, npr)).astype(keys.dtype)
    probe = [Page([kb(pk, rng.random(npr) < 0.02), Block.integer(np.arange(npr))], npr)]
    types = [key_type, abi.INTEGER]
    orows, opairs, _ = oracle_join(oracle, build, types, [0], [1], probe, types, [0], [0, 1])
    assert len(orows) > 100000
    for switch in (None, "1"):
        if switch:
            monkeypatch.setenv("PRESTO_AMD_NO_PARTITIONED_BUILD", switch)
        rows, pairs, _ = gpu_join(build, types, [0], [1], probe, types, [0], [0, 1])
        assert rows == orows
        for (gp, gb), (op_, ob) in zip(pairs, opairs):
            assert np.array_equal(gp, op_) and np.array_equal(gb, ob)


@pytest.mark.parametrize("shape", ["five_per_key", "one_long_chain", "a_chain_beyond_the_rounds"])
def test_partitioned_build_links_the_rows_of_a_key(gpu, oracle, shape):
    """>= 2^20 build rows where keys have several rows: the partitioned build puts the chains together in LDS (round by round: the
    highest position not yet on the chain becomes the next link) -- positionLinks and emission order equal the oracle's
    (ArrayPositionLinks.java:45-50: from the highest position down).  A key with more rows than the rounds take sends the build
    the other way."""
    rng = np.random.default_rng(len(shape) + 7)
    nb, npr = (1 << 20) + 4321, 120_000
    distinct = nb // 5 if shape == "five_per_key" else nb
    keys = (rng.integers(0, distinct, nb).astype(np.int64) * 7919) - 12345
    if shape != "five_per_key":
        heavy = 150 if shape == "one_long_chain" else 400
        keys[rng.permutation(nb)[:heavy]] = 77_777_777_777
    build = [Page([Block.bigint(keys, rng.random(nb) < 0.001), Block.integer(np.arange(nb))], nb)]
    pk = np.where(rng.random(npr) < 0.5, keys[rng.integers(0, nb, npr)], rng.integers(-2 ** 40, 2 ** 40, npr)).astype(np.int64)
    pk[:3] = 77_777_777_777
    probe = [Page([Block.bigint(pk), Block.integer(np.arange(npr))], npr)]
    types = [abi.BIGINT, abi.INTEGER]
    orows, opairs, _ = oracle_join(oracle, build, types, [0], [1], probe, types, [0], [0, 1])
    rows, pairs, _ = gpu_join(build, types, [0], [1], probe, types, [0], [0, 1])
    assert len(orows) > 100_000 and rows == orows
    for (gp, gb), (op_, ob) in zip(pairs, opairs):
        assert np.array_equal(gp, op_) and np.array_equal(gb, ob)


def test_partitioned_build_with_an_overfull_partition_falls_back(gpu, oracle, monkeypatch):
    """Keys chosen so that more of them have their home slot in ONE 8192-slot partition than the partition may hold: the build
    notices (no slot is dropped, nothing overflows the LDS table) and assembles the table slot by slot instead."""
    P1, P2 = np.uint64(0x9E3779B185EBCA87), np.uint64(0xC2B2AE3D27D4EB4F)

    def home(keys, mask):   # AbstractLongType.hash + fastutil murmurHash3 (pa_hash_bigint, pa_murmur3_fmix)
        with np.errstate(over="ignore"):
            h = keys.astype(np.uint64) * P2
            h = ((h << np.uint64(31)) | (h >> np.uint64(33))) * P1
            h ^= h >> np.uint64(33)
            h *= np.uint64(0xff51afd7ed558ccd)
            h ^= h >> np.uint64(33)
            h *= np.uint64(0xc4ceb9fe1a85ec53)
            h ^= h >> np.uint64(33)
        return h & np.uint64(mask)

    monkeypatch.setenv("PRESTO_AMD_NO_RANK_INDEX", "1")  # (dense unique keys: the rank index would stand in for the table)
    nb = (1 << 20) + 5000
    slots = 1 << 22                                   # the probe-side table of nb rows: >= 2 nb slots, a power of two
    cand = np.arange(1, 6_000_000, dtype=np.int64)
    in_first = cand[home(cand, slots - 1) < np.uint64(8192)]
    assert len(in_first) > 7800                        # > 7/8 of a partition's 8192 slots
    rest = cand[home(cand, slots - 1) >= np.uint64(8192)][: nb - len(in_first)]
    keys = np.concatenate([in_first, rest])
    rng = np.random.default_rng(2)
    keys = keys[rng.permutation(len(keys))]
    build = [Page([Block.bigint(keys), Block.integer(np.arange(len(keys)))], len(keys))]
    pk = np.concatenate([in_first[:5000], rng.integers(1, 7_000_000, 100000)]).astype(np.int64)
    probe = [Page([Block.bigint(pk), Block.integer(np.arange(len(pk)))], len(pk))]
    types = [abi.BIGINT, abi.INTEGER]
    rows, pairs, _ = gpu_join(build, types, [0], [1], probe, types, [0], [0, 1])
    orows, opairs, _ = oracle_join(oracle, build, types, [0], [1], probe, types, [0], [0, 1])
    assert rows == orows and len(rows) > 5000


@pytest.mark.parametrize("key_type", [abi.BIGINT, abi.INTEGER, abi.DATE])
@pytest.mark.parametrize("order", ["key order", "shuffled", "descending"])
@pytest.mark.parametrize("nb", [1, 777, 300_001])
def test_unique_dense_keys_are_looked_up_by_rank(gpu, oracle, key_type, order, nb, monkeypatch):
    """A build side with one integer key, no NULL and no duplicate key over a dense enough range is looked up through the key rank
    index (bitmap words with running counts; the rank of a key names its build row, directly when the rows came in key order) and no
    table is built.  Same rows and (probe, build) pairs as the oracle and as the table (PRESTO_AMD_NO_RANK_INDEX=1)."""
    rng = np.random.default_rng(nb + len(order))
    dtype = np.int64 if key_type == abi.BIGINT else np.int32
    base = -5_000_000_000 if key_type == abi.BIGINT else -1000
    keys = (np.sort(rng.permutation(4 * nb + 100)[:nb]) * 3 + base).astype(dtype)   # unique, sparse (x3), 64 words and more
    if order == "shuffled":
        keys = keys[rng.permutation(nb)]
    elif order == "descending":
        keys = keys[::-1].copy()
    kb = {abi.BIGINT: Block.bigint, abi.INTEGER: Block.integer, abi.DATE: Block.date}[key_type]
    build = Page([kb(keys), Block.double(rng.random(nb)), Block.integer(np.arange(nb))], nb)
    bpages = [build] if nb < 1000 else [build.get_region(0, 1000), build.get_region(1000, nb - 1000)]
    npr = 50_000
    pk = np.where(rng.random(npr) < 0.5, keys[rng.integers(0, nb, npr)], (rng.integers(-200, 12 * nb + 500, npr) + base)).astype(dtype)
    pk[:4] = [keys.min(), keys.max(), keys.min() - 1, keys.max() + 1]
    probe = [Page([kb(pk, rng.random(npr) < 0.03), Block.integer(np.arange(npr))], npr)]
    btypes, ptypes = [key_type, abi.DOUBLE, abi.INTEGER], [key_type, abi.INTEGER]
    orows, opairs, _ = oracle_join(oracle, bpages, btypes, [0], [1, 2], probe, ptypes, [0], [0, 1])
    assert len(orows) > npr // 3
    for switch in (None, "1"):
        if switch:
            monkeypatch.setenv("PRESTO_AMD_NO_RANK_INDEX", switch)
        rows, pairs, _ = gpu_join(bpages, btypes, [0], [1, 2], probe, ptypes, [0], [0, 1])
        assert rows == orows
        for (gp, gb), (op_, ob) in zip(pairs, opairs):
            assert np.array_equal(gp, op_) and np.array_equal(gb, ob)


@pytest.mark.parametrize("spoiler", ["duplicate key", "null key"])
def test_rank_index_is_not_used_with_duplicate_or_null_keys(gpu, oracle, spoiler):
    """One duplicate key or one NULL key among dense keys: the build falls back to the table; chains and NULL rules as ever."""
    rng = np.random.default_rng(8)
    nb, npr = 20_000, 30_000
    keys = (rng.permutation(3 * nb)[:nb]).astype(np.int64)
    nulls = np.zeros(nb, dtype=bool)
    if spoiler == "duplicate key":
        keys[nb - 1] = keys[0]
    else:
        nulls[nb // 2] = True
    build = [Page([Block.bigint(keys, nulls if nulls.any() else None), Block.integer(np.arange(nb))], nb)]
    pk = rng.integers(-5, 3 * nb + 5, npr).astype(np.int64)
    pk[:2] = [keys[0], keys[nb // 2]]
    probe = [Page([Block.bigint(pk), Block.integer(np.arange(npr))], npr)]
    types = [abi.BIGINT, abi.INTEGER]
    rows, pairs, _ = gpu_join(build, types, [0], [1], probe, types, [0], [0, 1])
    orows, opairs, _ = oracle_join(oracle, build, types, [0], [1], probe, types, [0], [0, 1])
    assert rows == orows and len(rows) > 5000
    for (gp, gb), (op_, ob) in zip(pairs, opairs):
        assert np.array_equal(gp, op_) and np.array_equal(gb, ob)


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


# ---- outer joins: OperatorFactories.probeOuterJoin / lookupOuterJoin / fullOuterJoin -------------------------------------
def gpu_outer_join(join_type, build_pages, build_types, join_ch, out_ch, probe_pages, probe_types, probe_join_ch, probe_out_ch, device_output=False):
    from presto_amd.operators import LookupOuterOperator, download_page
    mem = abi.MEM_DEVICE if device_output else abi.MEM_HOST
    take = (lambda p: download_page(p)) if device_output else (lambda p: p)
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, build_types, join_ch, out_ch), build_pages)
    join = LookupJoinOperator(bridge, probe_types, probe_join_ch, probe_out_ch, join_type=join_type, output_mem=mem)
    rows, pairs = [], []
    for p in probe_pages:
        join.addInput(p)
        out = join.getOutput()
        if out is not None:
            rows += take(out).to_rows()
        pairs.append(join.matchPairs())
    join.finish()
    outer_rows = None
    if join_type in (abi.JOIN_LOOKUP_OUTER, abi.JOIN_FULL_OUTER):
        outer = LookupOuterOperator(bridge, probe_types, probe_out_ch, join_type=join_type, output_mem=mem)
        assert not outer.needsInput()
        out = outer.getOutput()
        outer_rows = take(out).to_rows() if out is not None else []
        assert outer.isFinished() and outer.getOutput() is None
    return rows, pairs, outer_rows


def oracle_outer_join(oracle, join_type, build_pages, build_types, join_ch, out_ch, probe_pages, probe_types, probe_join_ch, probe_out_ch):
    j = oracle.HashJoin(build_types, join_ch, out_ch)
    for p in build_pages:
        j.add_build_page(p)
    j.build()
    rows, pairs = [], []
    for p in probe_pages:
        out, pi, bi = j.probe(p, probe_types, probe_join_ch, probe_out_ch, join_type=join_type)
        rows += out.to_rows()
        pairs.append((pi, bi))
    outer_rows = j.outer(probe_types, probe_out_ch).to_rows() if join_type in (abi.JOIN_LOOKUP_OUTER, abi.JOIN_FULL_OUTER) else None
    return rows, pairs, outer_rows


def test_probe_outer_join_kats(gpu, oracle):
    """TestHashJoinOperator.testProbeOuterJoin / testOuterJoinWithNull{Probe,Build,OnBothSides} /
    test{Probe,Full}OuterJoinWithEmptyLookupSource (…/join/TestHashJoinOperator.java:850-894, 945-1158, 1297-1390)"""
    types = [abi.VARCHAR, abi.BIGINT, abi.BIGINT]
    build = [sequence_page(10, [(abi.VARCHAR, 20), (abi.BIGINT, 30), (abi.BIGINT, 40)])]
    probe = [sequence_page(15, [(abi.VARCHAR, 20), (abi.BIGINT, 1020), (abi.BIGINT, 2020)])]
    rows, _, _ = gpu_outer_join(abi.JOIN_PROBE_OUTER, build, types, [0], [0, 1, 2], probe, types, [0], [0, 1, 2])
    expected = [(str(20 + i).encode(), 1020 + i, 2020 + i, str(20 + i).encode(), 30 + i, 40 + i) for i in range(10)]
    expected += [(str(30 + i).encode(), 1030 + i, 2030 + i, None, None, None) for i in range(5)]
    assert rows == expected

    def vj(build, probe, jt):
        b = [Page([Block.varchar(build)], len(build))] if build else []
        return gpu_outer_join(jt, b, [abi.VARCHAR], [0], [0], [Page([Block.varchar(probe)], len(probe))], [abi.VARCHAR], [0], [0])

    assert vj(["a", "b", "c"], ["a", None, None, "a", "b"], abi.JOIN_PROBE_OUTER)[0] == [(b"a", b"a"), (None, None), (None, None), (b"a", b"a"), (b"b", b"b")]
    assert vj(["a", None, None, "a", "b"], ["a", "b", "c"], abi.JOIN_PROBE_OUTER)[0] == [(b"a", b"a"), (b"a", b"a"), (b"b", b"b"), (b"c", None)]
    assert vj(["a", None, None, "a", "b"], ["a", "b", None, "c"], abi.JOIN_PROBE_OUTER)[0] == \
        [(b"a", b"a"), (b"a", b"a"), (b"b", b"b"), (None, None), (b"c", None)]
    for jt in (abi.JOIN_PROBE_OUTER, abi.JOIN_FULL_OUTER):
        assert vj([], ["a", "b", None, "c"], jt)[0] == [(b"a", None), (b"b", None), (None, None), (b"c", None)]
    rows, _, outer = vj([], ["test"], abi.JOIN_LOOKUP_OUTER)  # testLookupOuterJoinWithEmptyLookupSource (:1258-1295)
    assert rows == [] and outer == []
    rows, _, outer = vj(["a", None, "x", "a", "b", "y"], ["a", "c", None], abi.JOIN_FULL_OUTER)
    assert rows == [(b"a", b"a"), (b"a", b"a"), (b"c", None), (None, None)]
    assert outer == [(None, None), (None, b"x"), (None, b"b"), (None, b"y")]


@pytest.mark.parametrize("join_type", [abi.JOIN_PROBE_OUTER, abi.JOIN_LOOKUP_OUTER, abi.JOIN_FULL_OUTER])
@pytest.mark.parametrize("device_output", [False, True])
@pytest.mark.parametrize("unique_keys", [False, True])
def test_outer_joins_match_oracle(gpu, oracle, join_type, device_output, unique_keys):
    """Duplicates, NULL keys on both sides, several probe pages (the visited marks accumulate), nullable / VARCHAR build
    payload: output rows and their order, (probe, build) pairs and the LookupOuterOperator's rows equal the oracle's.
    unique_keys: no duplicate and no NULL build key -- the lookup goes through the key rank index instead of the table."""
    rng = np.random.default_rng(100 + join_type)
    nb = 5000
    words = [b"", b"x", b"payload", None, b"0123456789abcdef"]
    bk = rng.permutation(6000)[:nb].astype(np.int64) if unique_keys else rng.integers(0, 3000, nb).astype(np.int64)
    build = [Page([Block.bigint(bk, None if unique_keys else rng.random(nb) < 0.03), Block.varchar([words[i] for i in rng.integers(0, len(words), nb)]),
                   Block.double(rng.random(nb), rng.random(nb) < 0.1), Block.integer(np.arange(nb, dtype=np.int32))], nb)]
    btypes = [abi.BIGINT, abi.VARCHAR, abi.DOUBLE, abi.INTEGER]
    probes = []
    for i in range(3):
        n = 4000 + 17 * i
        probes.append(Page([Block.integer(np.arange(n, dtype=np.int32) + 10000 * i), Block.bigint(rng.integers(-100, 3500, n), rng.random(n) < 0.05)], n))
    ptypes = [abi.INTEGER, abi.BIGINT]
    got = gpu_outer_join(join_type, build, btypes, [0], [1, 2, 3, 0], probes, ptypes, [1], [0, 1], device_output)
    exp = oracle_outer_join(oracle, join_type, build, btypes, [0], [1, 2, 3, 0], probes, ptypes, [1], [0, 1])
    assert got[0] == exp[0]
    for (gp, gb), (ep, eb) in zip(got[1], exp[1]):
        assert np.array_equal(gp, ep) and np.array_equal(gb, eb)
    assert got[2] == exp[2]
    if exp[2] is not None:
        assert len(exp[2]) > 100


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_PROBE_OUTER, abi.JOIN_FULL_OUTER])
def test_output_single_match(gpu, oracle, join_type):
    """LookupJoinOperatorFactory.outputSingleMatch (DefaultPageJoiner.java:276-278; the semi-join-like shape the planner
    asks for when only the existence of a match matters): one output row per probe row -- the first position of its chain
    (the chain head is the highest build position, ArrayPositionLinks), NULL-extended rows of the outer joins as usual,
    and only the emitted build row counts as visited for the lookup-outer side."""
    from presto_amd.operators import LookupOuterOperator
    rng = np.random.default_rng(31)
    types = [abi.BIGINT, abi.BIGINT]
    build = [Page([Block.bigint(rng.integers(0, 300, 2000), rng.random(2000) < 0.05), Block.bigint(np.arange(2000) + 2000 * k)], 2000) for k in range(3)]
    probe = [Page([Block.bigint(rng.integers(0, 600, 5000), rng.random(5000) < 0.05), Block.bigint(np.arange(5000))], 5000) for _ in range(2)]
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, types, [0], [0, 1]), build)
    join = LookupJoinOperator(bridge, types, [0], [0, 1], join_type=join_type, output_single_match=True)
    got = [r for p in to_pages(join, probe) for r in p.to_rows()]
    ref = oracle.HashJoin(types, [0], [0, 1])
    for p in build:
        ref.add_build_page(p)
    ref.build()
    expected = [r for p in probe for r in ref.probe(p, types, [0], [0, 1], join_type=join_type, output_single_match=True)[0].to_rows()]
    assert got == expected
    plain = [r for p in probe for r in ref.probe(p, types, [0], [0, 1], join_type=abi.JOIN_INNER)[0].to_rows()]
    assert len(plain) > 2 * len([r for r in got if r[2] is not None])  # the build side holds ~20 rows per key
    if join_type == abi.JOIN_FULL_OUTER:
        outer = [r for p in to_pages(LookupOuterOperator(bridge, types, [0, 1], join_type=join_type), []) for r in p.to_rows()]
        assert sorted(outer, key=repr) == sorted(ref.outer(types, [0, 1]).to_rows(), key=repr) and len(outer) > 3000


def test_inner_join_with_output_single_match_kat(gpu, oracle):
    """TestHashJoinOperator.testInnerJoinWithOutputSingleMatch (…/operator/join/TestHashJoinOperator.java:733-767):
    build a, a, b; probe a, b, c -> (a, a), (b, b)"""
    types = [abi.VARCHAR]
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, types, [0], [0]), [Page([Block.varchar([b"a", b"a", b"b"])], 3)])
    join = LookupJoinOperator(bridge, types, [0], [0], output_single_match=True)
    got = [r for p in to_pages(join, [Page([Block.varchar([b"a", b"b", b"c"])], 3)]) for r in p.to_rows()]
    assert got == [(b"a", b"a"), (b"b", b"b")]


@pytest.mark.parametrize("key_type", [abi.BIGINT, abi.INTEGER, abi.DATE])
def test_dynamic_filter_upstream_of_the_probe(gpu, oracle, key_type):
    """pa_filter_project_set_dynamic_filter: the FilterAndProject feeding a probe also drops the rows whose key matches no build key
    (Trino applies a join's dynamic filter in the probe-side scan).  Its output = the plain filter's output restricted to keys
    that exist on the build side (NULL keys never match) -- and the join over it equals the join over the unfiltered rows."""
    from presto_amd._lib import PrestoAmdError
    from presto_amd.expr import field
    from presto_amd.operators import FilterAndProjectOperator
    rng = np.random.default_rng(41)
    mk = {abi.BIGINT: Block.bigint, abi.INTEGER: Block.integer, abi.DATE: Block.date}[key_type]
    build = [Page([mk(rng.integers(1000, 30000, 4000) * 3, rng.random(4000) < 0.02), Block.bigint(np.arange(4000))], 4000)]
    n = 120000
    probe = [Page([mk(rng.integers(0, 100000, n), rng.random(n) < 0.03), Block.double(rng.random(n))], n) for _ in range(2)]
    types = [key_type, abi.DOUBLE]
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, [key_type, abi.BIGINT], [0], [0, 1]), build)
    f = field(1, abi.DOUBLE) < 0.6
    proj = [field(0, key_type), field(1, abi.DOUBLE)]
    fp = FilterAndProjectOperator(types, f, proj)
    assert fp.setDynamicFilter(0, bridge) is True
    got = [r for p in to_pages(fp, probe) for r in p.to_rows()]
    keys = {v for v in build[0].blocks[0].to_pylist() if v is not None}
    plain = [r for p in probe for r in oracle.filter_project(p, f, proj).to_rows()]
    assert got == [r for r in plain if r[0] in keys] and 0 < len(got) < len(plain) // 5
    # same join result with and without the dynamic filter
    def join(pages):
        return [r for p in to_pages(LookupJoinOperator(bridge, types, [0], [0, 1]), pages) for r in p.to_rows()]
    filtered = to_pages(FilterAndProjectOperator(types, f, proj), probe)
    fp2 = FilterAndProjectOperator(types, f, proj)
    fp2.setDynamicFilter(0, bridge)
    assert join(to_pages(fp2, probe)) == join(filtered)
    # no filter expression at all: the dynamic filter alone selects
    fp3 = FilterAndProjectOperator(types, None, proj)
    fp3.setDynamicFilter(0, bridge)
    assert [r for p in to_pages(fp3, probe) for r in p.to_rows()] == [r for p in probe for r in p.to_rows() if r[0] in keys]
    # too late after the first page
    fp4 = FilterAndProjectOperator(types, f, proj)
    fp4.addInput(probe[0])
    fp4.getOutput()
    with pytest.raises(PrestoAmdError) as e:
        fp4.setDynamicFilter(0, bridge)
    assert e.value.status == abi.ERR_ILLEGAL_STATE


def test_dynamic_filter_not_offered_for_varchar_keys(gpu):
    from presto_amd.expr import field
    from presto_amd.operators import FilterAndProjectOperator
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, [abi.VARCHAR], [0], [0]), [Page([Block.varchar([b"a", b"b"])], 2)])
    fp = FilterAndProjectOperator([abi.VARCHAR, abi.BIGINT], None, [field(1, abi.BIGINT)])
    assert fp.setDynamicFilter(1, bridge) is False
    page = Page([Block.varchar([b"a", b"x", b"b"]), Block.bigint([1, 2, 3])], 3)
    assert [r for p in to_pages(fp, [page]) for r in p.to_rows()] == [(1,), (2,), (3,)]


def test_dynamic_filter_combined_over_partitions(gpu, oracle):
    """The cross-rank form (presto_amd/q3.py with exchange steps): every rank holds one partition of the build keys; the ranks
    agree on the union key range, each fills its bits (pa_lookup_source_key_bitmap), the bitmaps are OR-ed and installed with
    pa_filter_project_set_dynamic_filter_bitmap.  Two lookup sources on one GPU stand in for two ranks."""
    import torch
    from presto_amd.expr import field
    from presto_amd.operators import FilterAndProjectOperator
    rng = np.random.default_rng(43)
    keys = rng.choice(np.arange(5000, 90000), 6000, replace=False)
    parts = [keys[keys % 2 == 0], keys[keys % 2 == 1]]
    bridges = []
    for part in parts:
        b = LookupSourceFactory()
        to_pages(HashBuilderOperator(b, [abi.BIGINT], [0], [0]), [Page([Block.bigint(part)], len(part))])
        bridges.append(b)
    ranges = [b.keyRange() for b in bridges]
    assert ranges[0] == (int(parts[0].min()), int(parts[0].max()))
    lo, hi = min(r[0] for r in ranges), max(r[1] for r in ranges)
    words = ((hi - lo) >> 6) + 1
    maps = [torch.full((words,), -1, dtype=torch.int64, device="cuda") for _ in bridges]  # the call clears them
    torch.cuda.synchronize()
    for b, m in zip(bridges, maps):
        b.fillKeyBitmap(lo, hi - lo, m.data_ptr())
    gpu.check(gpu.lib().pa_stream_synchronize(None))
    both = maps[0] | maps[1]
    torch.cuda.synchronize()
    n = 200000
    probe = Page([Block.bigint(rng.integers(0, 100000, n), rng.random(n) < 0.02), Block.double(rng.random(n))], n)
    fp = FilterAndProjectOperator([abi.BIGINT, abi.DOUBLE], None, [field(0, abi.BIGINT), field(1, abi.DOUBLE)])
    fp.setDynamicFilterBitmap(0, both.data_ptr(), lo, hi - lo, keep=both)
    got = [r for p in to_pages(fp, [probe]) for r in p.to_rows()]
    want = set(int(k) for k in keys)
    assert got == [r for r in probe.to_rows() if r[0] in want] and len(got) > 5000


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_PROBE_OUTER])
def test_probe_page_with_more_matches_than_one_output_page_holds(gpu, oracle, monkeypatch, join_type):
    """LookupJoinPageBuilder bounds its output pages (LookupJoinPageBuilder.java:51-56); the device operator joins a whole probe
    page at once, so a probe page whose matches exceed the bound of an output page (2^30 rows; the match total is kept in 64
    bits) comes out as several pages, each the join of a row range of the probe page.  With the bound lowered to 700 rows: same
    rows, same order as the oracle, more than one page, VARCHAR and NULL-extended columns included."""
    monkeypatch.setenv("PRESTO_AMD_JOIN_MAX_OUTPUT_ROWS", "700")
    rng = np.random.default_rng(17)
    nb, npr = 4000, 3001
    bkeys = rng.integers(0, 300, nb)     # ~13 build rows per key
    build = Page([Block.bigint(bkeys), Block.varchar([b"b%d" % i for i in range(nb)])], nb)
    pkeys = rng.integers(-50, 350, npr)
    pkeys[100:1500] = 1000               # a long stretch without any match
    probe = Page([Block.bigint(pkeys, rng.random(npr) < 0.03), Block.varchar([b"p%d" % i if i % 7 else None for i in range(npr)])], npr)
    btypes = ptypes = [abi.BIGINT, abi.VARCHAR]
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, btypes, [0], [1, 0]), [build])
    join = LookupJoinOperator(bridge, ptypes, [0], [1, 0], join_type=join_type)
    pages = to_pages(join, [probe])
    j = oracle.HashJoin(btypes, [0], [1, 0])
    j.add_build_page(build)
    j.build()
    expected, _, _ = j.probe(probe, ptypes, [0], [1, 0], join_type=join_type)
    assert len(pages) > 5 and all(0 < p.position_count <= 700 for p in pages)
    assert [r for p in pages for r in p.to_rows()] == expected.to_rows()


def test_split_probe_page_behind_a_device_operator(gpu, oracle, monkeypatch):
    """The probe pages come from a FilterAndProject on the device, whose output buffers are its own again as soon as it is given the
    next page -- which a Driver does while the join still emits the ranges of the page before.  The join keeps the probe channels
    of a page it has to emit range by range: same rows as over host pages."""
    from presto_amd.expr import field
    from presto_amd.operators import Driver, FilterAndProjectOperator
    monkeypatch.setenv("PRESTO_AMD_JOIN_MAX_OUTPUT_ROWS", "500")
    rng = np.random.default_rng(23)
    nb = 3000
    build = Page([Block.bigint(rng.integers(0, 200, nb)), Block.integer(np.arange(nb))], nb)
    probes = []
    for i in range(4):
        n = 2000 + i
        probes.append(Page([Block.bigint(rng.integers(-20, 230, n), rng.random(n) < 0.03), Block.varchar([b"p%d_%d" % (i, r) if r % 5 else None for r in range(n)]),
                            Block.integer(np.arange(n) + 10000 * i)], n))
    btypes, ptypes = [abi.BIGINT, abi.INTEGER], [abi.BIGINT, abi.VARCHAR, abi.INTEGER]
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, btypes, [0], [1]), [build])
    fp = FilterAndProjectOperator(ptypes, field(2, abi.INTEGER) >= 0, [field(0, abi.BIGINT), field(1, abi.VARCHAR), field(2, abi.INTEGER)], output_mem=abi.MEM_DEVICE)
    join = LookupJoinOperator(bridge, ptypes, [0], [1, 2, 0])
    pages = Driver(probes, [fp, join]).run()
    j = oracle.HashJoin(btypes, [0], [1])
    j.add_build_page(build)
    j.build()
    expected = [r for p in probes for r in j.probe(p, ptypes, [0], [1, 2, 0])[0].to_rows()]
    assert len(pages) > 20 and all(0 < p.position_count <= 500 for p in pages)
    assert [r for p in pages for r in p.to_rows()] == expected


# ---- JoinFilterFunction ------------------------------------------------------------------------------------------------------
def _filter_case(rng, nb=3000, npr=5000, keys=400):
    build = Page([Block.bigint(rng.integers(0, keys, nb), rng.random(nb) < 0.03), Block.integer(rng.integers(0, 100, nb), rng.random(nb) < 0.1),
                  Block.varchar([[b"a", b"bb", b"ccc", None][i] for i in rng.integers(0, 4, nb)]), Block.double(rng.random(nb))], nb)
    probe = Page([Block.integer(rng.integers(0, 100, npr), rng.random(npr) < 0.1), Block.bigint(rng.integers(-20, keys + 50, npr), rng.random(npr) < 0.03),
                  Block.varchar([[b"a", b"bb", b"zz"][i] for i in rng.integers(0, 3, npr)])], npr)
    return build, [abi.BIGINT, abi.INTEGER, abi.VARCHAR, abi.DOUBLE], probe, [abi.INTEGER, abi.BIGINT, abi.VARCHAR]


@pytest.mark.parametrize("join_type", [abi.JOIN_INNER, abi.JOIN_PROBE_OUTER])
@pytest.mark.parametrize("single_match", [False, True])
@pytest.mark.parametrize("which", ["less", "strings", "never", "always"])
def test_join_filter_function(gpu, oracle, join_type, single_match, which):
    """build.key = probe.key AND <filter over the pair>: eligible positions only, in chain order; the first eligible one under
    outputSingleMatch; NULL-extended probe rows when none is eligible (probe-outer); a NULL filter result is not a match.
    Filter channels: 0..3 = build (key, quantity, tag, weight), 4..6 = probe (quantity, key, tag)."""
    from presto_amd.expr import and_, constant, field
    rng = np.random.default_rng(40 + len(which))
    build, btypes, probe, ptypes = _filter_case(rng)
    flt = {"less": field(1, abi.INTEGER) < field(4, abi.INTEGER),                                   # build.quantity < probe.quantity (NULLs on both sides)
           "strings": and_(field(2, abi.VARCHAR).eq(field(6, abi.VARCHAR)), field(3, abi.DOUBLE) > constant(0.25, abi.DOUBLE)),
           "never": field(0, abi.BIGINT) > field(5, abi.BIGINT),                                  # contradicts the equi-condition
           "always": field(0, abi.BIGINT).eq(field(5, abi.BIGINT))}[which]
    expected, epairs, _ = oracle.join_with_filter([build], btypes, [0], [1, 2], probe, ptypes, [1], [0, 1, 2], flt, join_type, single_match)
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, btypes, [0], [1, 2]), [build.get_region(0, 1000), build.get_region(1000, 2000)])
    join = LookupJoinOperator(bridge, ptypes, [1], [0, 1, 2], join_type=join_type, output_single_match=single_match, filter=flt)
    rows, pairs = [], []
    for page, base in ((probe.get_region(0, 2000), 0), (probe.get_region(2000, 3000), 2000)):
        join.addInput(page)
        out = join.getOutput()
        if out is not None:
            rows += out.to_rows()
            gp, gb = join.matchPairs()
            pairs += [(int(p) + base, int(b)) for p, b in zip(gp, gb)]
    assert rows == expected and pairs == epairs
    if which == "never":
        assert len(rows) == (probe.position_count if join_type == abi.JOIN_PROBE_OUTER else 0)
    else:
        assert len(rows) > 500


def test_join_filter_decides_which_build_rows_a_full_outer_join_has_visited(gpu, oracle):
    from presto_amd.expr import field
    from presto_amd.operators import LookupOuterOperator
    rng = np.random.default_rng(77)
    build, btypes, probe, ptypes = _filter_case(rng, nb=800, npr=1500, keys=300)
    flt = field(1, abi.INTEGER) < field(4, abi.INTEGER)
    expected, _, visited = oracle.join_with_filter([build], btypes, [0], [0, 1], probe, ptypes, [1], [0, 1], flt, abi.JOIN_FULL_OUTER)
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, btypes, [0], [0, 1]), [build])
    join = LookupJoinOperator(bridge, ptypes, [1], [0, 1], join_type=abi.JOIN_FULL_OUTER, filter=flt)
    outer = LookupOuterOperator(bridge, ptypes, [0, 1], join_type=abi.JOIN_FULL_OUTER)
    rows = [r for p in to_pages(join, [probe]) for r in p.to_rows()]
    assert rows == expected
    unvisited = [r for p in to_pages(outer, []) for r in p.to_rows()]
    brows = build.to_rows()
    assert 0 < len(unvisited) < 800
    assert unvisited == [(None, None, brows[b][0], brows[b][1]) for b in range(800) if b not in visited]


def test_large_build_side_with_a_hash_channel(gpu, oracle):
    """>= 2^20 build rows AND a $hashvalue channel on both sides: build and probe must place / look for a key by the same hash --
    the channel's -- so the partitioned build (which hashes the key itself) is not taken.  The channel carries the reference's
    hash here; a second run carries another function of the key, which only stays consistent if both sides use the channel."""
    rng = np.random.default_rng(41)
    nb, npr = (1 << 20) + 777, 200_000
    keys = rng.permutation(2 * nb)[:nb].astype(np.int64) * 31 - 5
    pk = np.where(rng.random(npr) < 0.7, keys[rng.integers(0, nb, npr)], rng.integers(-10 ** 9, 10 ** 9, npr)).astype(np.int64)
    types = [abi.BIGINT, abi.INTEGER, abi.BIGINT]

    def pages(hash_of):
        b = [Page([Block.bigint(keys), Block.integer(np.arange(nb)), Block.bigint(hash_of(keys))], nb)]
        p = [Page([Block.bigint(pk), Block.integer(np.arange(npr)), Block.bigint(hash_of(pk))], npr)]
        return b, p

    ref_hash = lambda k: oracle.hash_page(Page([Block.bigint(k)], len(k)), [0])
    other_hash = lambda k: (k * np.int64(0x9E3779B97F4A7C15 - (1 << 64))) ^ (k >> np.int64(7))
    for hash_of in (ref_hash, other_hash):
        build, probe = pages(hash_of)
        rows, pairs, _ = gpu_join(build, types, [0], [1], probe, types, [0], [0, 1], build_hash=2, probe_hash=2)
        orows, opairs, _ = oracle_join(oracle, build, types, [0], [1], probe, types, [0], [0, 1], build_hash=2, probe_hash=2)
        assert len(orows) > 100_000 and rows == orows
        for (gp, gb), (op_, ob) in zip(pairs, opairs):
            assert np.array_equal(gp, op_) and np.array_equal(gb, ob)


def test_lookup_join_page_builder_positions_on_device(gpu, oracle):
    """TestLookupJoinPageBuilder.testDifferentPositions (…/join/TestLookupJoinPageBuilder.java:85-150) at the operator's level, on the
    device: a probe page of 0 .. 99 against the build page 0 .. 99 -- (1) no probe row joins: no page; (2) only the last position joins;
    (3) every position joins once: both columns 0 .. 99 in probe order; (4) every second position joins: 0, 2, 4, ...; (5) every position
    joins twice (two build rows per key): each probe row twice, build rows in descending build position.  Rows, order and the
    (probe position, build position) pairs equal the oracle's LookupJoinPageBuilder restatement."""
    types = [abi.BIGINT]
    build = [Page([Block.bigint(np.arange(100, dtype=np.int64))], 100)]
    keys_even = np.arange(100, dtype=np.int64)
    keys_even[1::2] += 1000
    keys_last = np.arange(100, dtype=np.int64) + 1000
    keys_last[99] = 99
    cases = {
        "empty": np.arange(100, dtype=np.int64) + 1000,
        "last position only": keys_last,
        "every position": np.arange(100, dtype=np.int64),
        "even positions": keys_even,
    }
    for name, keys in cases.items():
        probe = [Page([Block.bigint(keys)], 100)]
        rows, pairs, _ = gpu_join(build, types, [0], [0], probe, types, [0], [0])
        orows, opairs, _ = oracle_join(oracle, build, types, [0], [0], probe, types, [0], [0])
        assert rows == orows, name
        assert pairs[0][0].tolist() == opairs[0][0].tolist() and pairs[0][1].tolist() == opairs[0][1].tolist(), name
    assert cases and oracle_join(oracle, build, types, [0], [0], [Page([Block.bigint(cases["empty"])], 100)], types, [0], [0])[0] == []
    # (5) each probe position joined twice
    twice = [Page([Block.bigint(np.concatenate([np.arange(100), np.arange(100)]).astype(np.int64)), Block.bigint(np.arange(200, dtype=np.int64))], 200)]
    probe = [Page([Block.bigint(np.arange(100, dtype=np.int64))], 100)]
    rows, pairs, _ = gpu_join(twice, [abi.BIGINT, abi.BIGINT], [0], [1], probe, types, [0], [0])
    assert rows == [(i, b) for i in range(100) for b in (100 + i, i)]          # descending build position within a probe row
    orows, opairs, _ = oracle_join(oracle, twice, [abi.BIGINT, abi.BIGINT], [0], [1], probe, types, [0], [0])
    assert rows == orows and pairs[0][0].tolist() == opairs[0][0].tolist() and pairs[0][1].tolist() == opairs[0][1].tolist()


def test_group_by_hash_contains_on_device(gpu, oracle):
    """TestGroupByHash.testContains / testContainsMultipleColumns (…/operator/TestGroupByHash.java:202-235): membership of a row in the
    set of keys seen.  GroupByHash.contains is the primitive under the reference's ChannelSet (SetBuilderOperator -> semi join); on the device
    the set is a lookup source and the question a probe with outputSingleMatch: DOUBLE keys 0 .. 9 contain 3.0 and not 11.0;
    (DOUBLE, VARCHAR) keys (i, str(i)) contain (3.0, "3") and not (3.0, "4").  -0.0 finds 0.0 and NaN finds NaN (IS NOT DISTINCT keys)."""
    from presto_amd.operators import LookupJoinOperator as Join

    def contains(build_page, build_types, probe_page):
        bridge = LookupSourceFactory()
        to_pages(HashBuilderOperator(bridge, build_types, list(range(len(build_types))), []), [build_page])
        j = Join(bridge, build_types, list(range(len(build_types))), [0], output_single_match=True)
        j.addInput(probe_page)
        out = j.getOutput()
        j.finish()
        return 0 if out is None else out.position_count

    keys = Page([Block.double(np.arange(10, dtype=np.float64))], 10)
    assert contains(keys, [abi.DOUBLE], Page([Block.double([3.0])], 1)) == 1
    assert contains(keys, [abi.DOUBLE], Page([Block.double([11.0])], 1)) == 0
    assert contains(keys, [abi.DOUBLE], Page([Block.double([-0.0])], 1)) == 1
    two = Page([Block.double(np.arange(10, dtype=np.float64)), Block.varchar([str(i) for i in range(10)])], 10)
    assert contains(two, [abi.DOUBLE, abi.VARCHAR], Page([Block.double([3.0]), Block.varchar(["3"])], 1)) == 1
    assert contains(two, [abi.DOUBLE, abi.VARCHAR], Page([Block.double([3.0]), Block.varchar(["4"])], 1)) == 0
    # the oracle's GroupByHash agrees
    gbh = oracle.HashAggregation([abi.DOUBLE, abi.VARCHAR], [0, 1], [], expected_groups=100)
    gbh.add_page(two)
    assert gbh.contains(Page([Block.double([3.0]), Block.varchar(["3"])], 1), 0) and not gbh.contains(Page([Block.double([3.0]), Block.varchar(["4"])], 1), 0)


def test_retained_build_page_is_read_in_place_until_the_lookup_source_goes(gpu, oracle):
    """A build side that arrives as one PA_PAGE_RETAINED device page is not copied: the build columns are the page's block arrays
    (PagesIndex holds on to the Page's blocks the same way) and the page is released when the lookup source is destroyed -- not before:
    the test overwrites the buffers in the release callback, a probe after an early release would join garbage.  A second build page
    moves the borrowed arrays into the builder's own; a retained page that had to be copied is handed back at once."""
    from presto_amd._lib import check, lib
    from presto_amd.operators import upload_page
    rng = np.random.default_rng(11)
    nb, npr = 20_000, 50_000
    bhost = Page([Block.bigint(rng.permutation(nb).astype(np.int64) * 3), Block.double(rng.random(nb))], nb)
    phost = Page([Block.bigint(rng.integers(0, 3 * nb, npr).astype(np.int64)), Block.integer(np.arange(npr))], npr)
    btypes, ptypes = [abi.BIGINT, abi.DOUBLE], [abi.BIGINT, abi.INTEGER]
    orows, _, _ = oracle_join(oracle, [bhost], btypes, [0], [1], [phost], ptypes, [0], [0, 1])

    def retained(host, released, tag):
        dev = upload_page(host)

        def on_release():
            released.append(tag)
            for b in dev.blocks:
                junk = np.full(b.values.nbytes, 0x5A, dtype=np.uint8)
                check(lib().pa_memcpy_h2d(b.values.ptr, junk.ctypes.data, b.values.nbytes, None))
        return Page(dev.blocks, dev.position_count, abi.MEM_DEVICE, on_release=on_release)

    released = []
    bridge = LookupSourceFactory()
    builder = HashBuilderOperator(bridge, btypes, [0], [1])
    to_pages(builder, [retained(bhost, released, "build")])
    builder.close()
    assert released == []                                  # the lookup source reads the page where it is
    for _ in range(2):                                     # two probe operators of the same bridge
        join = LookupJoinOperator(bridge, ptypes, [0], [0, 1])
        join.addInput(phost)
        assert join.getOutput().to_rows() == orows
        join.finish()
        join.close()
    assert released == []
    bridge.destroy()
    assert released == ["build"]
    # two pages: the first is borrowed, the second makes the builder take copies; a retained second page is released right away
    released = []
    half = nb // 2
    bridge = LookupSourceFactory()
    builder = HashBuilderOperator(bridge, btypes, [0], [1])
    to_pages(builder, [retained(bhost.get_region(0, half), released, "first"), retained(bhost.get_region(half, nb - half), released, "second")])
    assert released == ["second"]
    join = LookupJoinOperator(bridge, ptypes, [0], [0, 1])
    join.addInput(phost)
    assert join.getOutput().to_rows() == orows
    join.finish()
    join.close()
    builder.close()
    bridge.destroy()
    assert sorted(released) == ["first", "second"]


@pytest.mark.parametrize("duplicates", [False, True])
def test_keyed_table_with_a_hash_channel_and_no_rank_index(gpu, oracle, duplicates):
    """One integer key, a $hashvalue channel on both sides and a NULL build key (no key rank index): the probe reads the slot table row by
    row and takes a match's output rows from the slot's chain length -- 1 for a key on one build row, set when the slot is claimed (a build
    without duplicate keys never runs the pass that counts the chains: round 4's first version emitted nothing here)."""
    rng = np.random.default_rng(77)
    nb, npr = 3000, 20_000
    keys = (rng.permutation(3 * nb)[:nb] * 7 - 100).astype(np.int64)
    if duplicates:
        keys[nb // 2:] = keys[: nb - nb // 2]
    nulls = np.zeros(nb, dtype=bool)
    nulls[5] = True
    pk = np.where(rng.random(npr) < 0.6, keys[rng.integers(0, nb, npr)], rng.integers(-500, 25 * nb, npr)).astype(np.int64)
    types = [abi.BIGINT, abi.INTEGER, abi.BIGINT]
    h = lambda k: oracle.hash_page(Page([Block.bigint(k)], len(k)), [0])
    build = [Page([Block.bigint(keys, nulls), Block.integer(np.arange(nb)), Block.bigint(h(keys))], nb)]
    probe = [Page([Block.bigint(pk), Block.integer(np.arange(npr)), Block.bigint(h(pk))], npr)]
    rows, pairs, _ = gpu_join(build, types, [0], [1], probe, types, [0], [0, 1], build_hash=2, probe_hash=2)
    orows, opairs, _ = oracle_join(oracle, build, types, [0], [1], probe, types, [0], [0, 1], build_hash=2, probe_hash=2)
    assert len(orows) > npr // 3 and rows == orows
    for (gp, gb), (op_, ob) in zip(pairs, opairs):
        assert np.array_equal(gp, op_) and np.array_equal(gb, ob)
