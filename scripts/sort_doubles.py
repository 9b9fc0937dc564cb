"""OrderBy of (DOUBLE key, BIGINT payload) rows by the DOUBLE channel over several key distributions: the pair sort with bucket bounds from
a sample (PA_SORT_HINT_CROWDED) against the library's radix sort (PRESTO_AMD_SORT_LIBRARY=1) -- python3 scripts/sort_doubles.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from presto_amd import _lib, abi
from presto_amd.operators import OrderByOperator
from presto_amd.page import Block, DeviceBuffer, Page
import bench_ops
_lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 23
rng = np.random.default_rng(5)
pay = bench_ops.DeviceArray(np.arange(n, dtype=np.int64))
def dev_block(type_, t):
    return Block(type_, abi.FLAT, t.numel(), values=DeviceBuffer(t.alloc.ptr, t.host.itemsize * t.numel(), t))
cases = (("uniform [0, 1)", rng.random(n)), ("normal", rng.standard_normal(n)), ("exponential", rng.exponential(1.0, n)),
         ("prices (2 decimals, 0 .. 10^5)", np.round(rng.random(n) * 1e5, 2)))
for name, host in cases:
    keys = bench_ops.DeviceArray(host)
    page = Page([dev_block(abi.DOUBLE, keys), dev_block(abi.BIGINT, pay)], n, abi.MEM_DEVICE, stable=True)
    ts = []
    for _ in range(4):
        _lib.device_synchronize()
        t0 = time.perf_counter()
        op = OrderByOperator([abi.DOUBLE, abi.BIGINT], [0, 1], [0], [abi.ASC_NULLS_LAST], output_mem=abi.MEM_DEVICE)
        op.addInput(page)
        op.finish()
        op.getOutput()
        ms, _ = op.kernelTime()
        kernel = op.kernelName()
        op.close()
        _lib.device_synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%-32s operator %.3f ms (%.1f G rows/s), sort passes %.3f ms (%s)" % (name, min(ts), n / min(ts) / 1e6, ms, kernel), flush=True)
    keys.free()
