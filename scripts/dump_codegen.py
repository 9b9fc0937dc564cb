"""Writes the generated source of every cell of tests/test_codegen_tiers.py's full matrix under a directory, one file per cell
(<shape>.<tier>.<mask>.hip), plus the probe-stage kernels: `diff -r` of two dumps shows what a change of the generator changed."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_codegen_tiers as T
from presto_amd import abi, tpch, q3
from presto_amd._lib import lib
from presto_amd.operators import fused_join_aggregation_desc, fused_join_desc, hash_builder_desc
from presto_amd.expr import field

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
n = 0
for name, variant, mask in T.cells(True):
    src = T.source(name, variant, mask)
    with open(os.path.join(out, "%s.%s.%x.hip" % (name, T.TIER_NAMES[variant], mask)), "w") as f:
        f.write(src if src is not None else "REFUSED: " + lib().pa_last_error().decode() + "\n")
    n += 1
L = lib()
build, kb = hash_builder_desc(q3.ORDERS_JOINED_TYPES, [0], [1, 2])
for ai, aggs in enumerate((q3.AGG_AGGREGATES, [(abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_SUM, 0, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None)])):
    for gi, (group_by, variants) in enumerate(((q3.AGG_GROUP_BY, (6, 2, 3)), ([2, 3], (2, 3, 1)), ([], (0,)))):
        d, keep = fused_join_aggregation_desc(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1], q3.AGG_TYPES, group_by, aggs)
        for v in variants:
            need = L.pa_codegen_fused_join(C.byref(d), C.byref(build), v, None, 0)
            buf = C.create_string_buffer(max(need, 1))
            if need > 0:
                L.pa_codegen_fused_join(C.byref(d), C.byref(build), v, buf, need)
            open(os.path.join(out, "probe.%d.%d.%s.hip" % (ai, gi, T.TIER_NAMES[v])), "wb").write(buf.value if need > 0 else b"REFUSED\n")
            n += 1
print(n, "sources under", out)
