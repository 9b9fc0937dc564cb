import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from presto_amd import _lib, abi, tpch
from presto_amd.exchange import Comm, PartialStateMerger, partial_layout
from presto_amd.operators import AggregationOperator, HashAggregationOperator, FusedAggregationOperatorFactory
_lib.init(0)
comm = Comm.single()
comm.preflight(1 << 20)
merger = PartialStateMerger(comm=comm)
n = 2_000_000
dev6 = tpch.DeviceColumns(tpch.Q6_COLUMNS, 1.0, n)
dev1 = tpch.DeviceColumns(tpch.Q1_COLUMNS, 1.0, n)
f6 = FusedAggregationOperatorFactory(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES, step=abi.STEP_PARTIAL)
f1 = FusedAggregationOperatorFactory(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS, step=abi.STEP_PARTIAL)
parts = {}
for name, f, dev in (("q6", f6, dev6), ("q1", f1, dev1)):
    op = f.createOperator()
    for p in dev.pages(1 << 28):
        op.addInput(p)
    op.finish()
    parts[name] = op.getOutput()
    op.close()
t6, a6 = partial_layout([], tpch.Q6_AGGREGATES)
t1, a1 = partial_layout([abi.VARCHAR, abi.VARCHAR], tpch.Q1_AGGREGATES)
finals = {"q6": lambda: AggregationOperator(t6, a6, step=abi.STEP_FINAL),
          "q1": lambda: HashAggregationOperator(t1, [0, 1], a1, step=abi.STEP_FINAL, type_params=[1, 1] + [0] * (len(t1) - 2))}
for _ in range(5):
    merger.merge(parts, finals)
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    out = merger.merge(parts, finals)
    ts.append(time.perf_counter() - t0)
ts.sort()
print("merge (world 1: all-gather over RCCL with itself + 2 FINAL operators): median %.1f us, min %.1f us" % (ts[10] * 1e6, ts[0] * 1e6))
import pickle, struct
t0 = time.perf_counter()
for _ in range(100):
    chunks = merger._all_gather(b"x" * 2000)
print("all-gather alone: %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
print(out["q1"].to_rows()[:1])

# ---- where the time goes ----
from presto_amd.exchange import _state_payload, _concat_payloads, _state_page
def clock(label, fn, reps=50):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    print("  %-34s %7.1f us" % (label, (time.perf_counter() - t0) / reps * 1e6))
    return r
for name in ("q6", "q1"):
    print(name)
    payload = clock("payload (host page -> python)", lambda: _state_payload(parts[name]))
    blob = clock("pickle", lambda: pickle.dumps(payload))
    cols = clock("concat", lambda: _concat_payloads([payload]))
    page = clock("state page (python -> host page)", lambda: _state_page(cols))
    clock("operator create + close", lambda: finals[name]().close())
    def run():
        op = finals[name]()
        op.addInput(page)
        op.finish()
        o = op.getOutput()
        op.close()
        return o
    clock("create/addInput/finish/getOutput", run)
    op = finals[name]()
    c0 = time.perf_counter(); op.addInput(page); c1 = time.perf_counter(); op.finish(); c2 = time.perf_counter(); o = op.getOutput(); c3 = time.perf_counter()
    print("  one run: addInput %.1f finish %.1f getOutput %.1f us" % ((c1 - c0) * 1e6, (c2 - c1) * 1e6, (c3 - c2) * 1e6))
    op.close()
