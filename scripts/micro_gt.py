"""Dev tool: GT (HBM group table) aggregation at high cardinality -- clustered keys (equal keys adjacent, the shape of
a join output ordered by orderkey) vs the same rows shuffled vs unique keys.  Kernel time from pa_op_kernel_time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from presto_amd import _lib, abi
from presto_amd.operators import HashAggregationOperator
from presto_amd.page import Block, DeviceBuffer, Page
torch.cuda.set_device(0)
_lib.init(0)
rows = 1 << 24
groups = rows * 5 // 12
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
clustered = (torch.arange(rows, device="cuda", dtype=torch.int64) * 5 // 12) * 7 + 3
perm = torch.randperm(rows, device="cuda", generator=g)
cases = {"clustered": clustered, "shuffled": clustered[perm].contiguous(), "unique": torch.arange(rows, device="cuda", dtype=torch.int64) * 7 + 3}


def blk(t, ty):
    return Block(ty, abi.FLAT, t.numel(), values=DeviceBuffer(t.data_ptr(), t.numel() * t.element_size(), t))


for name, keys in cases.items():
    page = Page([blk(keys, abi.BIGINT), blk(vals, abi.DOUBLE)], rows, abi.MEM_DEVICE)
    for rep in range(3):
        torch.cuda.synchronize()
        op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)],
                                     expected_groups=rows, output_mem=abi.MEM_DEVICE)
        t0 = time.perf_counter()
        op.addInput(page)
        op.finish()
        out = op.getOutput()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = op.kernelTime()
        ng = out.position_count
        op.close()
    print("%-10s %d rows -> %d groups: wall %.2f ms, fused kernels %.2f ms (%d launches) = %.2f G rows/s" % (name, rows, ng, dt * 1e3, ms, n, rows / ms / 1e6))
