import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from presto_amd import _lib, abi
from presto_amd.exchange import partial_layout
from presto_amd.operators import HashAggregationOperator, to_pages
from presto_amd.page import Block, Page
_lib.init(0)
rng = np.random.default_rng(8)
n = int(sys.argv[1]); K = int(sys.argv[2])
page = Page([Block.bigint(rng.integers(0, K, n)), Block.double(rng.random(n), rng.random(n) < 0.1), Block.bigint(rng.integers(-1000, 1000, n))], n)
types = [abi.BIGINT, abi.DOUBLE, abi.BIGINT]
for aggs in ([(abi.AGG_COUNT_STAR, -1, None)],
             [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_AVG, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None), (abi.AGG_COUNT, 1, abi.DOUBLE), (abi.AGG_SUM, 2, abi.BIGINT), (abi.AGG_AVG, 2, abi.BIGINT)]):
    single = sorted(r for p in to_pages(HashAggregationOperator(types, [0], aggs), [page]) for r in p.to_rows())
    halves = [page.get_region(0, n // 2 - 7), page.get_region(n // 2 - 7, n - (n // 2 - 7))]
    ptypes, faggs = partial_layout([abi.BIGINT], aggs)
    parts = []
    for h in halves:
        parts += to_pages(HashAggregationOperator(types, [0], aggs, step=abi.STEP_PARTIAL), [h])
    print("partials", [sorted(p.to_rows()) for p in parts])
    final = sorted(r for p in to_pages(HashAggregationOperator(ptypes, [0], faggs, step=abi.STEP_FINAL), parts) for r in p.to_rows())
    print("final ", final)
    print("single", single)
