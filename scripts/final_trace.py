"""One FINAL HashAggregation over a 4-row PARTIAL page of Q1, for an API-level trace (rocprofv3 --hip-trace): which calls and round trips
the plain operator path spends on a tiny page."""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from presto_amd import _lib, abi, tpch
from presto_amd.exchange import partial_layout
from presto_amd.operators import HashAggregationOperator, FusedAggregationOperatorFactory
_lib.init(0)
n = 500_000
dev1 = tpch.DeviceColumns(tpch.Q1_COLUMNS, 1.0, n)
f1 = FusedAggregationOperatorFactory(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS, step=abi.STEP_PARTIAL)
op = f1.createOperator()
for p in dev1.pages(1 << 28):
    op.addInput(p)
op.finish()
page = op.getOutput()
op.close()
t1, a1 = partial_layout([abi.VARCHAR, abi.VARCHAR], tpch.Q1_AGGREGATES)
def run():
    op = HashAggregationOperator(t1, [0, 1], a1, step=abi.STEP_FINAL, type_params=[1, 1] + [0] * (len(t1) - 2))
    op.addInput(page)
    op.finish()
    o = op.getOutput()
    op.close()
    return o
for _ in range(5):
    run()
_lib.device_synchronize()
time.sleep(0.05)      # a gap in the timeline: the traced run is what follows it
t0 = time.perf_counter()
run()
print("one FINAL: %.1f us" % ((time.perf_counter() - t0) * 1e6))
