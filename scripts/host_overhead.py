"""Where does a query's wall time go on the host side?  (dev tool)"""
import sys, time, os
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); _x = torch.empty(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from presto_amd import _lib, abi, tpch
from presto_amd.operators import FusedAggregationOperator
_lib.init(0)
sf = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
rows = tpch.lineitem_rows(sf)
cols = sorted(set(tpch.Q1_COLUMNS + tpch.Q6_COLUMNS))
t = tpch.DeviceColumns(cols, sf, rows)
def pages_of(c):
    sub = tpch.DeviceColumns.__new__(tpch.DeviceColumns); sub.columns = c; sub.rows = rows; sub._bufs = t._bufs
    return list(sub.pages(1 << 26))
P = {"q6": pages_of(tpch.Q6_COLUMNS), "q1": pages_of(tpch.Q1_COLUMNS)}
def mk(q):
    if q == "q6":
        return FusedAggregationOperator(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
    return FusedAggregationOperator(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS)
for it in range(4):
    for q in ("q6", "q1"):
        t0 = time.perf_counter(); op = mk(q); t1 = time.perf_counter()
        ts = []
        for p in P[q]:
            a = time.perf_counter(); op.addInput(p); ts.append(time.perf_counter() - a)
        t2 = time.perf_counter(); op.finish(); out = op.getOutput(); t3 = time.perf_counter()
        kt = op.kernelTime(); t4 = time.perf_counter(); op.close(); t5 = time.perf_counter()
        print("%s create %.3f ms | addInput %s ms | finish+getOutput %.3f | kernelTime %.3f (%s) | close %.3f | total %.3f" % (
            q, (t1 - t0) * 1e3, " ".join("%.3f" % (x * 1e3) for x in ts), (t3 - t2) * 1e3, (t4 - t3) * 1e3, kt, (t5 - t4) * 1e3, (t5 - t0) * 1e3))
