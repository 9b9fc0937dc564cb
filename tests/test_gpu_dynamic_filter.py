"""DynamicFilterSourceOperator on the device against the reference's own cases (TestDynamicFilterSourceOperator.java, restated
in tests/test_oracle_operators.py::dynamic_filter_kats) and against the oracle on random build sides: distinct value sets,
the fall-back to min / max, giving up, NULL / NaN / -0.0 handling, VARCHAR channels, pass-through of the pages."""
import math

import os

import numpy as np
import pytest

from presto_amd import abi
from presto_amd.operators import DynamicFilterSourceOperator
from presto_amd.page import Block, Page
from tests.test_oracle_operators import dynamic_filter_kats, dynamic_filter_pages

pytestmark = pytest.mark.gpu


def drive(op, pages):
    """verifyPassthrough: addInput / getOutput alternate, the output is the input page."""
    for p in pages:
        assert op.needsInput()
        op.addInput(p)
        assert not op.needsInput()
        out = op.getOutput()
        assert out is p and op.getOutput() is None
    op.finish()
    op.finish()  # Driver may call finish() twice
    assert op.isFinished()
    return op.predicate()


def same(got, expected):
    if isinstance(expected, str) or isinstance(got, str):
        return got == expected
    if len(got) != len(expected):
        return False
    for g, e in zip(got, expected):
        if g[0] != e[0]:
            return False
        gv = list(g[1]) if g[0] == "values" else list(g[1:])
        ev = list(e[1]) if e[0] == "values" else list(e[1:])
        if len(gv) != len(ev):
            return False
        for a, b in zip(gv, ev):
            if isinstance(b, float):
                if a != b or math.copysign(1, a) != math.copysign(1, b):
                    return False
            elif a != b:
                return False
    return True


@pytest.mark.parametrize("kat", dynamic_filter_kats(), ids=lambda k: k[0])
def test_reference_cases(gpu, kat):
    name, types, channels, pages, (max_distinct, max_bytes, row_limit), expected = kat
    op = DynamicFilterSourceOperator(types, channels, max_distinct, max_bytes, row_limit)
    got = drive(op, dynamic_filter_pages(types, pages))
    assert same(got, expected), (got, expected)


def test_predicate_is_not_ready_before_finish_unless_given_up(gpu):
    op = DynamicFilterSourceOperator([abi.DOUBLE], [0], 10, 10240, 1000)
    page = Page([Block.double(np.arange(5, dtype=np.float64))], 5)
    op.addInput(page)
    op.getOutput()
    assert op.predicate() is None
    big = Page([Block.double(np.arange(50, dtype=np.float64))], 50)
    op.addInput(big)  # more than 10 distinct DOUBLEs and no orderable channel: TupleDomain.all() right away
    assert op.predicate() == "all"
    op.getOutput()
    op.finish()
    assert op.predicate() == "all"


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("PA_FUZZ_SEEDS", "16")))))  # PA_FUZZ_SEEDS=N: a longer one-off run
def test_random_build_sides_against_oracle(gpu, oracle, seed):
    rng = np.random.default_rng(900 + seed)
    kinds = [abi.BIGINT, abi.INTEGER, abi.DATE, abi.DOUBLE, abi.BOOLEAN, abi.VARCHAR, abi.REAL]
    nch = int(rng.integers(1, 4))
    types = [kinds[i] for i in rng.integers(0, len(kinds), nch)] + [abi.BIGINT]
    channels = list(range(nch))
    card = int([5, 60, 3000, 200000][seed % 4])
    max_distinct = int([20, 100, 5000, 100000][(seed // 4) % 4])
    max_bytes = int(rng.choice([2000, 10 ** 9]))
    row_limit = int(rng.choice([0, 50000, 10 ** 9]))
    pages = []
    for _ in range(int(rng.integers(1, 5))):
        n = int(rng.integers(1, 70000))
        blocks = []
        for t in types:
            nulls = rng.random(n) < float(rng.choice([0.0, 0.05]))
            nulls = nulls if nulls.any() else None
            if t == abi.BIGINT:
                blocks.append(Block.bigint(rng.integers(-card, card, n) * 7919, nulls))
            elif t == abi.INTEGER:
                blocks.append(Block.integer(rng.integers(-card, card, n), nulls))
            elif t == abi.DATE:
                blocks.append(Block.date(rng.integers(0, card, n), nulls))
            elif t == abi.BOOLEAN:
                blocks.append(Block.boolean(rng.random(n) < 0.5, nulls))
            elif t == abi.DOUBLE:
                pool = np.concatenate([rng.standard_normal(card), [0.0, -0.0, np.nan, np.inf, -np.inf]])
                blocks.append(Block.double(pool[rng.integers(0, len(pool), n)], nulls))
            elif t == abi.REAL:
                pool = np.concatenate([rng.standard_normal(card), [0.0, -0.0, np.nan, np.inf, -np.inf]]).astype(np.float32)
                blocks.append(Block.real(pool[rng.integers(0, len(pool), n)], nulls))
            else:
                ids = rng.integers(0, card, n)
                blocks.append(Block.varchar([None if (nulls is not None and nulls[i]) else b"key-%d-%s" % (v, b"z" * (v % 11)) for i, v in enumerate(ids)]))
        pages.append(Page(blocks, n))
    ref = oracle.DynamicFilterSource(types, channels, max_distinct, max_bytes, row_limit)
    for p in pages:
        ref.add_page(p)
    ref.finish()
    op = DynamicFilterSourceOperator(types, channels, max_distinct, max_bytes, row_limit)
    got = drive(op, pages)
    assert same(got, ref.predicate), (got if isinstance(got, str) else [g[:1] for g in got], ref.predicate if isinstance(ref.predicate, str) else [e[:1] for e in ref.predicate])


def test_device_pages_pass_through(gpu, oracle):
    import torch
    from presto_amd.page import DeviceBuffer
    keys = torch.randint(0, 1 << 40, (1 << 22,), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    page = Page([Block(abi.BIGINT, abi.FLAT, keys.numel(), values=DeviceBuffer(keys.data_ptr(), keys.numel() * 8, keys))], keys.numel(), abi.MEM_DEVICE)
    op = DynamicFilterSourceOperator([abi.BIGINT], [0], 1000, 1 << 20, 1 << 30)
    got = drive(op, [page])
    assert got == [("range", int(keys.min()), int(keys.max()))]
