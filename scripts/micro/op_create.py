"""Host cost of the operator life cycle without device work: createOperator + close for the operator factories of Q3 / Q1 / Q6
(what a Driver pays per pipeline before the first page), microseconds per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from presto_amd import _lib, abi, tpch, q3
from presto_amd.expr import field
from presto_amd.operators import (FilterAndProjectOperatorFactory, FusedAggregationOperatorFactory, FusedJoinAggregationOperatorFactory, FusedJoinOperatorFactory,
                                  HashBuilderOperatorFactory, LookupSourceFactory, TopNOperatorFactory)
torch.cuda.set_device(0)
_lib.init(0)
s = _lib.DeviceStream().handle
dev = abi.MEM_DEVICE


def t(name, make, n=300):
    for _ in range(20):
        make()
    t0 = time.perf_counter()
    for _ in range(n):
        make()
    print("%-44s %7.1f us" % (name, (time.perf_counter() - t0) / n * 1e6))


f = FilterAndProjectOperatorFactory(tpch.CUSTOMER_TYPES, tpch.q3_customer_filter(), [field(0, abi.BIGINT)], output_mem=dev, stream=s)
t("FilterAndProject (customer)", lambda: f.createOperator().close())
hb = HashBuilderOperatorFactory([abi.BIGINT], [0], [], stream=s)
def build():
    b = LookupSourceFactory(); op = hb.createOperator(b); op.close(); b.destroy()
t("LookupSourceFactory + HashBuilder", build)
b1 = LookupSourceFactory(); keep = hb.createOperator(b1)
fj = FusedJoinOperatorFactory(tpch.ORDERS_TYPES, tpch.q3_orders_filter(), [field(i, ty) for i, ty in enumerate(tpch.ORDERS_TYPES)], [1], [0, 2, 3], output_mem=dev, stream=s)
t("FusedJoin (orders)", lambda: fj.createOperator(b1).close())
hb2 = HashBuilderOperatorFactory(q3.ORDERS_JOINED_TYPES, [0], [1, 2], stream=s)
b2 = LookupSourceFactory(); keep2 = hb2.createOperator(b2)
fa = FusedJoinAggregationOperatorFactory(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1], q3.AGG_TYPES, q3.AGG_GROUP_BY,
                                         q3.AGG_AGGREGATES[:1], expected_groups=14_000_000, output_mem=dev, stream=s)
t("FusedJoinAggregation (lineitem)", lambda: fa.createOperator(b2).close())
tn = TopNOperatorFactory(q3.RESULT_TYPES[:4], 10, [3, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST], output_mem=abi.MEM_HOST, stream=s)
t("TopN", lambda: tn.createOperator().close())
q6 = FusedAggregationOperatorFactory(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
t("FusedAggregation (Q6)", lambda: q6.createOperator().close())
q1 = FusedAggregationOperatorFactory(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES, type_params=tpch.Q1_TYPE_PARAMS)
t("FusedAggregation (Q1)", lambda: q1.createOperator().close())
