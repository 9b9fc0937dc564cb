"""Dev tool: many bench steps in one process; device memory and step time must stay flat (pools, caches, handles)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from presto_amd import _lib, tpch
from presto_amd.operators import FusedAggregationOperatorFactory
torch.cuda.set_device(0)
_lib.init(0)
sf, steps = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0, int(sys.argv[2]) if len(sys.argv) > 2 else 300
rows = tpch.lineitem_rows(sf)
cols = sorted(set(tpch.Q1_COLUMNS + tpch.Q6_COLUMNS))
table = tpch.DeviceColumns(cols, sf, rows)
def pages_of(c):
    sub = tpch.DeviceColumns.__new__(tpch.DeviceColumns); sub.columns = c; sub.rows = table.rows; sub._bufs = table._bufs
    return list(sub.pages(1 << 24))
q6p, q1p = pages_of(tpch.Q6_COLUMNS), pages_of(tpch.Q1_COLUMNS)
f6 = FusedAggregationOperatorFactory(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
f1 = FusedAggregationOperatorFactory(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS)
first = None
for step in range(steps):
    t0 = time.perf_counter()
    for f, pages in ((f6, q6p), (f1, q1p)):
        op = f.createOperator()
        for p in pages:
            op.addInput(p)
        op.finish()
        out = sorted(op.getOutput().to_rows())
        op.close()
    dt = time.perf_counter() - t0
    if first is None:
        first = out
    assert out == first
    if step % 50 == 0 or step == steps - 1:
        free, total = torch.cuda.mem_get_info()
        print("step %4d  %.2f ms  device memory in use %.3f GB" % (step, dt * 1e3, (total - free) / 1e9), flush=True)
