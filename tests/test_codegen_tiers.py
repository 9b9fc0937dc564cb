"""Every tier of the fused aggregation x nullability signature x key width, generated and compiled for gfx950 without a GPU
(hiprtc in process): a wave-uniformity slip, a hash fold or a missing declaration in one tier's generated source shows here, before
a GPU lease -- round 2 compiled only the Q1 / Q6 / Q3 shapes off the GPU.  The default run takes a spread of the matrix (a couple of
minutes on 8 cores); PA_CODEGEN_FULL=1 takes all of it.  scripts/dump_codegen.py writes the same sources out, one file per cell,
which is how a refactoring of the generator is shown to change nothing."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import pytest

from presto_amd import abi, tpch
from presto_amd._lib import lib
from presto_amd.expr import and_, constant, field
from presto_amd.operators import fused_aggregation_desc, fused_join_aggregation_desc, hash_builder_desc

V_GLOBAL, V_LDS, V_GT, V_LDSH, V_HASH, V_LDSP, V_BROW, V_GLOBAL_R, V_LDS_R = range(9)
TIER_NAMES = ["global", "lds", "gt", "ldsh", "hash", "ldsp", "brow", "global_ranges", "lds_ranges"]
D = abi.decimal(12, 2)


def shapes():
    """name -> (input types, type params, filter, projections, group-by channels, aggregates): one shape per key layout"""
    bigint, dbl, date, integer, boolean, varchar, real = abi.BIGINT, abi.DOUBLE, abi.DATE, abi.INTEGER, abi.BOOLEAN, abi.VARCHAR, abi.REAL
    common = [(abi.AGG_SUM, 1, dbl), (abi.AGG_COUNT_STAR, -1, None)]
    rich = [(abi.AGG_SUM, 1, dbl), (abi.AGG_AVG, 1, dbl), (abi.AGG_MIN, 1, dbl), (abi.AGG_MAX, 2, bigint), (abi.AGG_COUNT, 1, dbl), (abi.AGG_SUM, 2, bigint),
            (abi.AGG_COUNT_STAR, -1, None, 3)]
    return {
        "one_bigint_key": ([bigint, dbl], None, None, [field(0, bigint), field(1, dbl)], [0], common),
        "bigint_date_integer_keys": ([bigint, dbl, date, integer], None, field(2, date) > constant(9000, date),
                                     [field(0, bigint), field(1, dbl) * constant(2.0, dbl), field(2, date), field(3, integer)], [0, 2, 3], common),
        "two_varchar1_keys_q1": (tpch.Q1_TYPES, tpch.Q1_TYPE_PARAMS, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES),
        "varchar12_key": ([varchar, dbl], [12, 0], None, [field(0, varchar), field(1, dbl)], [0], common),
        "double_boolean_real_keys_masked": ([dbl, dbl, bigint, boolean, real], None, None,
                                            [field(0, dbl), field(1, dbl), field(2, bigint), field(3, boolean), field(4, real)], [0, 3, 4], rich),
        "decimal_key_and_sums": ([D, D, bigint], None, field(1, D) >= constant(5, D), [field(0, D), field(1, D) * field(0, D), field(2, bigint)], [0],
                                 [(abi.AGG_SUM, 1, abi.decimal(24, 4)), (abi.AGG_AVG, 0, D), (abi.AGG_MIN, 0, D), (abi.AGG_COUNT_STAR, -1, None)]),
    }


def global_shapes():
    dbl, bigint = abi.DOUBLE, abi.BIGINT
    return {
        "q6": (tpch.Q6_TYPES, None, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES),
        "q6_decimal": (tpch.Q6_DECIMAL_TYPES, None, tpch.q6_decimal_filter(), tpch.q6_decimal_projections(), [], tpch.q6_decimal_aggregates()),
        "every_aggregate": ([dbl, bigint, abi.BOOLEAN], None, None, [field(0, dbl), field(1, bigint), field(2, abi.BOOLEAN)], [],
                            [(abi.AGG_SUM, 0, dbl), (abi.AGG_AVG, 1, bigint), (abi.AGG_MIN, 0, dbl), (abi.AGG_MAX, 1, bigint), (abi.AGG_COUNT, 0, dbl),
                             (abi.AGG_SUM, 1, bigint, 2)]),
    }


def cells(full):
    out = []
    for name, shape in global_shapes().items():
        n = len(shape[0])
        for mask in ([0, (1 << n) - 1] if full else [0, 0b10 if n > 1 else 1]):
            out.append((name, V_GLOBAL, mask))
            out.append((name, V_GLOBAL_R, mask))   # the same kernel over a table of row ranges
    for name, shape in shapes().items():
        n = len(shape[0])
        masks = [0, (1 << n) - 1, 0b01, 0b10] if full else [0, (1 << n) - 1]
        for variant in (V_LDS, V_GT, V_LDSH, V_HASH, V_LDSP):
            for mask in masks:
                if not full and variant in (V_HASH, V_LDSP) and mask:
                    continue
                out.append((name, variant, mask))
        for mask in masks[:2]:
            out.append((name, V_LDS_R, mask))
    return out


def descriptor(name):
    types, params, flt, proj, gb, aggs = {**shapes(), **global_shapes()}[name]
    return fused_aggregation_desc(types, flt, proj, gb, aggs, type_params=params)


def generate(name, variant, mask, compile_it):
    """(size | negative status, message)"""
    d, keep = descriptor(name)
    L = lib()
    L.pa_codegen_fused_layout.restype = C.c_int64
    L.pa_codegen_fused_layout.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_char_p, C.c_int64]
    rc = L.pa_codegen_fused_layout(C.byref(d), variant, mask, 1 if compile_it else 0, None, 0)
    return rc, (L.pa_last_error().decode() if rc < 0 else "")


def source(name, variant, mask):
    d, keep = descriptor(name)
    L = lib()
    L.pa_codegen_fused_layout.restype = C.c_int64
    L.pa_codegen_fused_layout.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_char_p, C.c_int64]
    need = L.pa_codegen_fused_layout(C.byref(d), variant, mask, 0, None, 0)
    if need < 0:
        return None
    buf = C.create_string_buffer(need)
    L.pa_codegen_fused_layout(C.byref(d), variant, mask, 0, buf, need)
    return buf.value.decode()


# shapes a tier refuses by design (PA_ERR_NOT_SUPPORTED: the operator then takes the next tier)
def refused_by_design(name, variant):
    return (name == "two_varchar1_keys_q1" and variant in (V_HASH, V_LDSP)) or False


def test_every_tier_generates_and_compiles_for_gfx950():
    full = os.environ.get("PA_CODEGEN_FULL") == "1"
    todo = cells(full)
    assert len(todo) >= 30

    def one(cell):
        name, variant, mask = cell
        rc, msg = generate(name, variant, mask, True)
        return cell, rc, msg
    with ThreadPoolExecutor(max_workers=8) as pool:   # (hiprtc releases the GIL: the compiles run side by side)
        results = list(pool.map(one, todo))
    failures = []
    refused = 0
    for (name, variant, mask), rc, msg in results:
        if rc == abi.ERR_NOT_SUPPORTED:
            refused += 1
            # a refusal must be about the tier's capacity, never about the code
            assert "too many" in msg or "too wide" in msg or "VARCHAR" in msg or "wider" in msg, (name, TIER_NAMES[variant], mask, msg)
            continue
        if rc < 1000:
            failures.append((name, TIER_NAMES[variant], bin(mask), rc, msg[-800:]))
    assert not failures, failures
    assert refused < len(todo) // 4


def test_probe_stage_tiers_generate_and_compile():
    """the probe stage (FilterAndProject -> LookupJoin -> aggregation as one kernel) behind every tier that takes it"""
    from presto_amd import q3
    L = lib()
    build, kb = hash_builder_desc(q3.ORDERS_JOINED_TYPES, [0], [1, 2])
    for aggs in (q3.AGG_AGGREGATES, [(abi.AGG_MAX, 1, abi.DOUBLE), (abi.AGG_SUM, 0, abi.BIGINT), (abi.AGG_COUNT_STAR, -1, None)]):
        for group_by, variants in ((q3.AGG_GROUP_BY, (V_BROW, V_GT, V_LDSH)), ([2, 3], (V_GT, V_LDSH, V_LDS)), ([], (V_GLOBAL,))):
            d, keep = fused_join_aggregation_desc(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1], q3.AGG_TYPES,
                                                  group_by, aggs)
            for v in variants:
                rc = L.pa_codegen_compile_fused_join(C.byref(d), C.byref(build), v)
                assert rc > 1000, (group_by, TIER_NAMES[v], L.pa_last_error())
