"""OrderBy of 2^24 (BIGINT key, BIGINT payload) rows over keys of W uniform bits, W = 14 .. 62: the pair sort's cost per LDS pass
(a kernel trace of this shows k_sort_buckets with 0, 1, 2, ... passes) -- python3 scripts/sort_widths.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from presto_amd import _lib, abi
from presto_amd.operators import OrderByOperator
from presto_amd.page import Block, DeviceBuffer, Page
import bench_ops
_lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
rng = np.random.default_rng(5)
pay = bench_ops.DeviceArray(np.arange(n, dtype=np.int64))
def dev_block(type_, t):
    return Block(type_, abi.FLAT, t.numel(), values=DeviceBuffer(t.alloc.ptr, t.host.itemsize * t.numel(), t))
for width in (14, 21, 28, 35, 40, 48, 62):
    keys = bench_ops.DeviceArray(rng.integers(0, 1 << width, n, dtype=np.int64))
    page = Page([dev_block(abi.BIGINT, keys), dev_block(abi.BIGINT, pay)], n, abi.MEM_DEVICE)
    ts = []
    for _ in range(4):
        _lib.device_synchronize()
        t0 = time.perf_counter()
        op = OrderByOperator([abi.BIGINT, abi.BIGINT], [0, 1], [0], [abi.ASC_NULLS_LAST], output_mem=abi.MEM_DEVICE)
        op.addInput(page)
        op.finish()
        op.getOutput()
        ms, _ = op.kernelTime()
        name = op.kernelName()
        op.close()
        _lib.device_synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print("W=%2d: operator %.3f ms, sort passes %.3f ms (%s)" % (width, min(ts), ms, name), flush=True)
    keys.free()
