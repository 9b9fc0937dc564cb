"""BenchmarkGroupByHash's shape (10 M rows, 3 M groups, BenchmarkGroupByHash.java:68-71) through HashAggregationOperator, a few times:
what a kernel trace / PRESTO_AMD_HOST_TRACE listing of the fixed costs of one such run is taken of."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from presto_amd import _lib, abi
from presto_amd.operators import HashAggregationOperator
from presto_amd.page import Block, DeviceBuffer, Page
import bench_ops
_lib.init(0)
rng = np.random.default_rng(3)
rows, groups = 10_000_000, 3_000_000
keys = bench_ops.DeviceArray(rng.integers(0, groups, rows, dtype=np.int64))
vals = bench_ops.DeviceArray(rng.random(rows))
aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
def dev_block(type_, t):
    return Block(type_, abi.FLAT, t.numel(), values=DeviceBuffer(t.alloc.ptr, t.host.itemsize * t.numel(), t))
page = Page([dev_block(abi.BIGINT, keys), dev_block(abi.DOUBLE, vals)], rows, abi.MEM_DEVICE, stable=True)
def agg():
    op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], aggs, expected_groups=groups, output_mem=abi.MEM_DEVICE)
    op.addInput(page)
    op.finish()
    n = op.getOutput().position_count
    op.close()
    return n
for _ in range(3):
    agg()
_lib.device_synchronize()
time.sleep(0.05)
for _ in range(3):
    t0 = time.perf_counter()
    n = agg()
    _lib.device_synchronize()
    print("%d groups, %.3f ms" % (n, (time.perf_counter() - t0) * 1e3))
