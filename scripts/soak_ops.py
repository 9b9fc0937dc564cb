"""Dev tool: the operators beyond the headline path, created / driven / closed over and over in one process -- device memory
must stay flat and results identical (pools, per-operator dictionaries, generations, probe-side tables, multisplit buffers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from presto_amd import _lib, abi
from presto_amd.expr import constant, field
from presto_amd.operators import (DynamicFilterSourceOperator, FilterAndProjectOperator, HashAggregationOperator, HashBuilderOperator,
                                  LookupJoinOperator, LookupSourceFactory, OrderByOperator, TopNOperator, to_pages)
from presto_amd.page import Block, Page
torch.cuda.set_device(0)
_lib.init(0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(5)
n = 400000


def agg_pages(card, nullable_late):
    out = []
    for k in range(3):
        nulls = (rng.random(n) < 0.05) if (nullable_late and k == 2) else None
        keys = [b"customer-%07d-%s" % (v, b"x" * (v % 13)) for v in rng.integers(0, card, n // 8)]
        out.append(Page([Block.bigint(rng.integers(0, card, n), nulls), Block.double(rng.random(n)),
                         Block.varchar(keys * 8)], n))
    return out


cases = [(c, agg_pages(c, c == 3000)) for c in (5, 700, 3000, 60000, 250000)]
types = [abi.BIGINT, abi.DOUBLE, abi.VARCHAR]
aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_MIN, 1, abi.DOUBLE)]
build = [Page([Block.bigint(rng.integers(0, 50000, 100000)), Block.double(rng.random(100000))], 100000)]
probe = [Page([Block.bigint(rng.integers(0, 100000, n)), Block.double(rng.random(n))], n) for _ in range(2)]
d = Block.varchar([b"AIR", b"MAIL", None, b"SHIP", b"TRUCK"])
dict_page = Page([Block.dictionary_block(d, rng.integers(0, 5, n).astype(np.int32)), Block.bigint(np.arange(n))], n)
first = None
for step in range(steps):
    t0 = time.perf_counter()
    sig = []
    for card, pages in cases:
        for keys in ([0], [2], [0, 2]):
            rows = [r for p in to_pages(HashAggregationOperator(types, keys, aggs, expected_groups=card), pages) for r in p.to_rows()]
            sig.append((len(rows), sum(r[len(keys) + 1] for r in rows)))
    bridge = LookupSourceFactory()
    to_pages(HashBuilderOperator(bridge, [abi.BIGINT, abi.DOUBLE], [0], [0, 1]), build)
    for jt in (abi.JOIN_INNER, abi.JOIN_PROBE_OUTER):
        sig.append(sum(p.position_count for p in to_pages(LookupJoinOperator(bridge, [abi.BIGINT, abi.DOUBLE], [0], [0, 1], join_type=jt), probe)))
    df = DynamicFilterSourceOperator([abi.BIGINT, abi.DOUBLE], [0], 1000, 1 << 20, 1 << 30)
    to_pages(df, build)
    sig.append(repr(df.predicate())[:60])
    fp = FilterAndProjectOperator([abi.VARCHAR, abi.BIGINT], field(0, abi.VARCHAR).eq(constant(b"MAIL", abi.VARCHAR)), [field(1, abi.BIGINT)])
    sig.append(sum(p.position_count for p in to_pages(fp, [dict_page])))
    sig.append(len([r for p in to_pages(TopNOperator([abi.BIGINT, abi.DOUBLE], 100, [1], [abi.DESC_NULLS_LAST]), probe) for r in p.to_rows()]))
    sig.append(sum(p.position_count for p in to_pages(OrderByOperator([abi.BIGINT, abi.DOUBLE], [0, 1], [0], [abi.ASC_NULLS_LAST]), probe[:1])))
    import gc
    gc.collect()
    dt = time.perf_counter() - t0
    if first is None:
        first = sig
    assert sig == first, (sig, first)
    if step % 10 == 0 or step == steps - 1:
        free, total = torch.cuda.mem_get_info()
        print("step %4d  %.1f ms  device memory in use %.3f GB" % (step, dt * 1e3, (total - free) / 1e9), flush=True)
