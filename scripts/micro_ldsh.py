"""Dev tool: one HashAggregation over 64 M rows at a given cardinality (for rocprofv3 counter passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from presto_amd import _lib, abi
from presto_amd.operators import HashAggregationOperator
from presto_amd.page import Block, DeviceBuffer, Page
torch.cuda.set_device(0)
_lib.init(0)
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows = 1 << 26
g = torch.Generator(device="cuda").manual_seed(1)
vals = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
keys = torch.randint(0, groups, (rows,), dtype=torch.int64, device="cuda", generator=g)
torch.cuda.synchronize()
page = Page([Block(abi.BIGINT, abi.FLAT, rows, values=DeviceBuffer(keys.data_ptr(), rows * 8, keys)),
             Block(abi.DOUBLE, abi.FLAT, rows, values=DeviceBuffer(vals.data_ptr(), rows * 8, vals))], rows, abi.MEM_DEVICE)
for rep in range(2):
    op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)], output_mem=abi.MEM_DEVICE)
    op.addInput(page)
    op.finish()
    out = op.getOutput()
    print(out.position_count, op.kernelTime())
    op.close()
