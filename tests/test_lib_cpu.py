"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/presto_amd.h declares, refuses to
run without a gfx950 device (no CPU fallback), and its query-time code generator produces HIP that compiles for
gfx950 (hiprtc needs no GPU).  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

from presto_amd import abi, tpch
from presto_amd._lib import PrestoAmdError, check, lib
from presto_amd.expr import and_, coalesce, constant, field, if_, not_, or_
from presto_amd.operators import _filter_project_desc, fused_aggregation_desc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "presto_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    symbols = sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", header)))
    assert len(symbols) >= 35
    L = lib()
    for s in symbols:
        assert getattr(L, s) is not None, s
    assert L.pa_abi_version() == abi.ABI_VERSION


def has_gpu():
    return lib().pa_device_count() > 0


@pytest.mark.skipif(has_gpu(), reason="container without a GPU only")
def test_no_device_fails_loudly():
    with pytest.raises(PrestoAmdError) as e:
        check(lib().pa_init(0))
    assert e.value.status == abi.ERR_NO_DEVICE
    d, keep = fused_aggregation_desc(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
    h = C.c_void_p()
    assert lib().pa_fused_aggregation_create(C.byref(d), C.byref(h)) == abi.ERR_NO_DEVICE
    assert b"no CPU fallback" in lib().pa_last_error()
    fp, keep2 = _filter_project_desc(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), abi.MEM_HOST, None)
    assert lib().pa_filter_project_create(C.byref(fp), C.byref(h)) == abi.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "presto_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in text and "presto_oracle" not in text and "import oracle" not in text and "from oracle" not in text, f


def fused_source(desc, variant=-1):
    L = lib()
    need = L.pa_codegen_fused(C.byref(desc), variant, None, 0, None)
    assert need > 0, L.pa_last_error()
    buf = C.create_string_buffer(need)
    key = C.create_string_buffer(32)
    L.pa_codegen_fused(C.byref(desc), variant, buf, need, key)
    return buf.value.decode(), key.value.decode()


def test_q6_q1_kernels_generate_and_compile_for_gfx950():
    d6, k6 = fused_aggregation_desc(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
    src, key = fused_source(d6)
    assert "pa_fused" in src and len(key) == 16
    # Java evaluates double arithmetic unfused: constants are exact hex literals, no fast-math anywhere
    assert "0x1.999999999999ap-5" in src
    assert lib().pa_codegen_compile_fused(C.byref(d6), -1) > 1000
    d1, k1 = fused_aggregation_desc(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                    type_params=tpch.Q1_TYPE_PARAMS)
    for variant in (1, 2, 3, 4):  # register/LDS table, HBM table (with run combining), workgroup LDS table (150 KB), partition-id pass
        src, key = fused_source(d1, variant)
        assert "#define PA_KW 1\n" in src  # both VARCHAR(1) keys share one packed word
        assert "#define PA_NW 6\n" in src  # 5 sums + 1 shared count for the 8 aggregates
        assert lib().pa_codegen_compile_fused(C.byref(d1), variant) > 1000, lib().pa_last_error()
    assert "pa_flush(a, acc, true)" in fused_source(d1, 2)[0] and "pa_lt_upsert" in fused_source(d1, 3)[0]


def fused_join_source(desc, build, variant=-1):
    L = lib()
    need = L.pa_codegen_fused_join(C.byref(desc), C.byref(build), variant, None, 0)
    assert need > 0, L.pa_last_error()
    buf = C.create_string_buffer(need)
    L.pa_codegen_fused_join(C.byref(desc), C.byref(build), variant, buf, need)
    return buf.value.decode()


def test_q3_probe_stage_kernels_generate_and_compile_for_gfx950():
    """FilterAndProject -> LookupJoin -> HashAggregation of Q3's lineitem pipeline as one kernel: the build side is
    (orderkey, orderdate, shippriority) keyed by orderkey; the group keys are the join key and two build columns, so the
    accumulators can be indexed by build position (variant 6); the hashed table (2) and the LDS tables (1, 3) take the same rows."""
    from presto_amd import q3
    from presto_amd.operators import fused_join_aggregation_desc, hash_builder_desc
    build, kb = hash_builder_desc(q3.ORDERS_JOINED_TYPES, [0], [1, 2])
    d, keep = fused_join_aggregation_desc(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1],
                                          q3.AGG_TYPES, q3.AGG_GROUP_BY, q3.AGG_AGGREGATES)
    src = fused_join_source(d, build)  # default: the build-row table
    assert "pa_brow_keys" in src and "pa_join_probe_keyed(a, " in src and "#define PA_KW 1\n" in src and "#define PA_TW 2\n" in src
    # the vector loop probes the four rows of a quad together, between pa_pre (filter + key) and pa_post (accumulate)
    assert "pa_join_probe4(a, js, jk, jb);" in src and src.count("pa_pre(a, true, (i32)(4 * q + ") == 4
    # extendedprice and discount are read for the matches only: filter and key take shipdate and orderkey
    pre_fn = src[src.index("void pa_pre("):]
    assert pre_fn[:pre_fn.index(")")].endswith("const i32 row, i64 c0, i64 c3, bool& sel0, u64& jk")
    assert "if (jb[2] >= 0) { c1_2 = ((const double*)a.v[1])[4 * q + 2]; c2_2 = ((const double*)a.v[2])[4 * q + 2]; }" in src
    for variant in (6, 2, 1, 3):
        assert lib().pa_codegen_compile_fused_join(C.byref(d), C.byref(build), variant) > 1000, lib().pa_last_error()
    assert "pa_brow_keys" not in fused_join_source(d, build, 2)
    # a global aggregate behind the probe
    d0, k0 = fused_join_aggregation_desc(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1],
                                         q3.AGG_TYPES, [], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_MAX, 2, abi.DATE)])
    assert lib().pa_codegen_compile_fused_join(C.byref(d0), C.byref(build), -1) > 1000, lib().pa_last_error()
    # group keys that do not name the build row (revenue is a probe column): only the hashed variants exist
    d2, k2 = fused_join_aggregation_desc(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1],
                                         q3.AGG_TYPES, [2, 3], [(abi.AGG_SUM, 1, abi.DOUBLE)])
    assert lib().pa_codegen_compile_fused_join(C.byref(d2), C.byref(build), 6) == abi.ERR_NOT_SUPPORTED
    assert lib().pa_codegen_compile_fused_join(C.byref(d2), C.byref(build), 2) > 1000, lib().pa_last_error()
    # a VARCHAR key is no keyed lookup source
    bad, kb2 = hash_builder_desc([abi.VARCHAR, abi.DATE], [0], [1])
    dv, kv = fused_join_aggregation_desc([abi.VARCHAR], None, [field(0, abi.VARCHAR)], [0], [0], [abi.VARCHAR, abi.DATE], [1],
                                         [(abi.AGG_COUNT_STAR, -1, None)])
    assert lib().pa_codegen_compile_fused_join(C.byref(dv), C.byref(bad), -1) == abi.ERR_NOT_SUPPORTED


def test_q3_probe_kernel_page_loops_generate_and_compile_for_gfx950(monkeypatch):
    """The build-row tier's page loop in its three forms: the four-stage software pipeline over a key rank index (the default, beside the
    plain loop for lookup sources without one), the plain loop alone (PRESTO_AMD_BROW_PIPE=0) and the loop that loads only the next quad's
    columns ahead (=1).  In the pipeline every load is unconditional: no `if (jb` in front of a lazy channel's load."""
    from presto_amd import q3
    from presto_amd.operators import fused_join_aggregation_desc, hash_builder_desc
    build, kb = hash_builder_desc(q3.ORDERS_JOINED_TYPES, [0], [1, 2])
    d, keep = fused_join_aggregation_desc(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1],
                                          q3.AGG_TYPES, q3.AGG_GROUP_BY, q3.AGG_AGGREGATES)
    src = fused_join_source(d, build, 6)
    assert "if (a.jrank) {" in src and "pa_join_rank4_issue(a, djs, djk, dw);" in src and "pa_join_rank4_read(a, djs, djk, dw, jb);" in src
    kernel = src[src.index("void PA_K(pa_fused_probe_brow)("):]
    piped = kernel[kernel.index("if (a.jrank) {"):kernel.index("    } else {\n")]
    assert "(pjb0 >= 0 ? 4 * pq + 0 : rs)" in piped and "if (jb" not in piped and "it < n_it + 2" in piped
    sources = {src}
    for level in ("0", "1"):
        monkeypatch.setenv("PRESTO_AMD_BROW_PIPE", level)
        s = fused_join_source(d, build, 6)
        assert "pa_join_rank4_issue" not in s[s.index("void PA_K(pa_fused_probe_brow)("):]
        assert lib().pa_codegen_compile_fused_join(C.byref(d), C.byref(build), 6) > 1000, lib().pa_last_error()
        sources.add(s)
    assert len(sources) == 3


def test_expression_forms_compile():
    types = [abi.BIGINT, abi.DOUBLE, abi.BOOLEAN, abi.INTEGER, abi.VARCHAR, abi.DATE]
    a, b, c, d, s, t = (field(i, ty) for i, ty in enumerate(types))
    f = and_(or_(a > 0, not_(c)), b.between(constant(0.5, abi.DOUBLE), constant(2.5, abi.DOUBLE)), s.eq(constant("BUILDING", abi.VARCHAR)),
             d.isin(1, 2, 3), t < constant(9204, abi.DATE), coalesce(a, constant(0, abi.BIGINT)).ne(7), if_(c, a / 3 > 1, a % 5 > 1))
    proj = [a * 2 + 1, b / constant(3.0, abi.DOUBLE), a.cast(abi.DOUBLE) * b, d + 1, -a, s, t, c]
    fp, keep = _filter_project_desc(types, f, proj, abi.MEM_HOST, None)
    size = lib().pa_codegen_compile_filter_project(C.byref(fp))
    assert size > 1000, lib().pa_last_error()


def test_unsupported_shapes_report_not_supported():
    # VARCHAR-valued expression projections are left to the Java operators
    types = [abi.VARCHAR, abi.BOOLEAN]
    fp, keep = _filter_project_desc(types, None, [if_(field(1, abi.BOOLEAN), field(0, abi.VARCHAR), field(0, abi.VARCHAR))], abi.MEM_HOST, None)
    assert lib().pa_codegen_compile_filter_project(C.byref(fp)) == abi.ERR_NOT_SUPPORTED
    # DOUBLE compared with BIGINT needs the planner's explicit CAST
    fp, keep = _filter_project_desc([abi.DOUBLE, abi.BIGINT], field(0, abi.DOUBLE) > field(1, abi.BIGINT), [], abi.MEM_HOST, None)
    assert lib().pa_codegen_compile_filter_project(C.byref(fp)) == abi.ERR_NOT_SUPPORTED
    # min / max over VARCHAR: channels declared VARCHAR(n), n <= 7, through an order-preserving 64-bit image, any other channel through
    # its strings' ranks -- unless something else (here: the grouping) needs the strings themselves
    d, keep = fused_aggregation_desc([abi.VARCHAR], None, [field(0, abi.VARCHAR)], [0], [(abi.AGG_MAX, 0, abi.VARCHAR)], type_params=[20])
    assert lib().pa_codegen_compile_fused(C.byref(d), -1) == abi.ERR_NOT_SUPPORTED
    d, keep = fused_aggregation_desc([abi.VARCHAR], None, [field(0, abi.VARCHAR)], [], [(abi.AGG_MAX, 0, abi.VARCHAR)])
    assert lib().pa_codegen_compile_fused(C.byref(d), -1) > 0
    d, keep = fused_aggregation_desc([abi.VARCHAR], None, [field(0, abi.VARCHAR)], [], [(abi.AGG_MAX, 0, abi.VARCHAR)], type_params=[7])
    assert lib().pa_codegen_compile_fused(C.byref(d), -1) > 0
    d, keep = fused_aggregation_desc([abi.DOUBLE], None, [field(0, abi.DOUBLE)], [], [(abi.AGG_MAX, 0, abi.DOUBLE)])
    assert lib().pa_codegen_compile_fused(C.byref(d), -1) > 0


def test_header_is_plain_c():
    """The boundary is a C ABI: include/presto_amd.h must compile as C99 (what a JNI shim written in C includes), and a C
    translation unit that references every declared function must link against libpresto_amd.so."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = os.path.join(root, "include", "presto_amd.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", header], check=True)
    names = sorted(set(re.findall(r"\b(pa_[a-z0-9_]+)\s*\(", open(header).read())) - {"pa_column"})
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "link_all.c")
        with open(src, "w") as f:
            f.write('#include "presto_amd.h"\n#include <stdio.h>\nint main(void) {\n    const void* fns[] = {\n')
            f.write("".join("        (const void*)%s,\n" % n for n in names))
            f.write('    };\n    printf("%d\\n", (int)(sizeof fns / sizeof fns[0]));\n    return 0;\n}\n')
        exe = os.path.join(d, "link_all")
        lib_dir = os.path.join(root, "presto_amd")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), src, "-L", lib_dir, "-lpresto_amd",
                        "-Wl,-rpath," + lib_dir, "-Wl,--allow-shlib-undefined", "-o", exe], check=True)
        assert len(names) > 40


def test_java_side_expects_this_abi_version():
    """GpuNative.java checks pa_abi_version against the version it was written for: keep the two in step."""
    import re
    text = open(os.path.join(ROOT, "java", "io", "trino", "gpu", "GpuNative.java")).read()
    assert [int(v) for v in re.findall(r"abiVersion\(\) != (\d+)", text)] == [abi.ABI_VERSION]


def test_jni_shim_compiles_against_the_header_and_matches_the_java_natives():
    """The JVM side (jni/presto_amd_jni.c, java/io/trino/gpu/*.java) cannot be built in this image (no JDK); what can drift
    silently is the shim against include/presto_amd.h and the Java `native` declarations against the shim's exports.  The C
    file is compiled (-fsyntax-only) against a stand-in jni.h holding the JNI types and the functions it uses, and the two
    lists of native methods are compared."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = os.path.join(root, "jni", "presto_amd_jni.c")
    subprocess.check_call(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(root, "jni", "stub"), "-I" + os.path.join(root, "include"), shim])
    exported = set(re.findall(r"Java_io_trino_gpu_GpuNative_(\w+)\(", open(shim).read()))
    declared = set(re.findall(r"static native [\w\[\]]+ (\w+)\(", open(os.path.join(root, "java", "io", "trino", "gpu", "GpuNative.java")).read()))
    assert exported == declared and len(declared) > 25
    # every C-ABI entry the shim calls is declared in the header (the compile above proves it) and exported by the library
    from presto_amd._lib import lib
    for sym in set(re.findall(r"\b(pa_[a-z0-9_]+)\(", open(shim).read())):
        getattr(lib(), sym)


def test_q3_orders_probe_inside_filter_and_project_generates_and_compiles_for_gfx950():
    """FilterAndProject -> LookupJoin of Q3's orders pipeline (pa_fused_join_desc): the probe runs inside the FilterAndProject
    kernels, the four rows of a quad side by side in both passes."""
    from presto_amd import q3
    from presto_amd.expr import field
    from presto_amd.operators import fused_join_desc, hash_builder_desc
    build, kb = hash_builder_desc([abi.BIGINT], [0], [])
    projections = [field(i, t) for i, t in enumerate(tpch.ORDERS_TYPES)]
    d, keep = fused_join_desc(tpch.ORDERS_TYPES, tpch.q3_orders_filter(), projections, [1], [0, 2, 3], output_mem=abi.MEM_DEVICE)
    L = lib()
    need = L.pa_codegen_fused_join_probe(C.byref(d), C.byref(build), None, 0)
    assert need > 0, L.pa_last_error()
    buf = C.create_string_buffer(need)
    L.pa_codegen_fused_join_probe(C.byref(d), C.byref(build), buf, need)
    src = buf.value.decode()
    count = src[src.index("void PA_K(pa_fp_count)("):src.index("void PA_K(pa_fp_scatter)(")]
    scatter = src[src.index("void PA_K(pa_fp_scatter)("):]
    # the counting pass asks whether a key exists (the key bitmap is exact: the slot table is only read where there is none)
    assert count.count("pa_keep(a, pa_k[") == 4 and "pa_join_exists4(a, pa_s, pa_k, pa_hit);" in count
    # no build column in the output: the second pass does not probe at all
    assert "pa_join_probe4(" not in scatter and "pa_key_of(a" not in scatter
    assert L.pa_codegen_compile_fused_join_probe(C.byref(d), C.byref(build)) > 1000, L.pa_last_error()
    # a build side that carries columns: they become output channels read at the build position
    build2, kb2 = hash_builder_desc(q3.ORDERS_JOINED_TYPES, [0], [1, 2])
    lp = [field(i, t) for i, t in enumerate(tpch.Q3_LINEITEM_TYPES)]
    d2, keep2 = fused_join_desc(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), lp, [0], [0, 1, 2], output_mem=abi.MEM_DEVICE)
    need = L.pa_codegen_fused_join_probe(C.byref(d2), C.byref(build2), None, 0)
    buf = C.create_string_buffer(need)
    L.pa_codegen_fused_join_probe(C.byref(d2), C.byref(build2), buf, need)
    scatter2 = buf.value.decode()
    scatter2 = scatter2[scatter2.index("void PA_K(pa_fp_scatter)("):]
    assert scatter2.count("pa_key_of(a") >= 4 and "pa_join_probe4(a, pa_s, pa_k, pa_jb);" in scatter2 and "pa_jb[3]" in scatter2
    assert L.pa_codegen_compile_fused_join_probe(C.byref(d2), C.byref(build2)) > 1000, L.pa_last_error()


def test_committed_counter_passes_are_of_the_kernels_the_bench_runs():
    """profiles/pmc_traffic.json is keyed by kernel NAME (tier + the first 8 hex digits of the code object's key), and bench.py quotes
    `roofline.traffic` only for a kernel of that name.  A change of the generated source (pa_device.h, the generator, the expression
    printer) gives the kernels new names and the line `traffic: null`: this test says so -- as a warning, the line stays valid -- so
    that the counter passes (scripts/collect_profiles.sh + scripts/summarize_profile.py) are collected again."""
    import json
    import os
    import warnings
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        pytest.skip("no committed counter passes")
    names = set(json.load(open(path)).get("kernels", {}))
    d6, k6 = fused_aggregation_desc(tpch.Q6_TYPES, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES)
    d1, k1 = fused_aggregation_desc(tpch.Q1_TYPES, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES,
                                    type_params=tpch.Q1_TYPE_PARAMS)
    expected = {"pa_fused_global_" + fused_source(d6, 0)[1][:8], "pa_fused_lds_" + fused_source(d1, 1)[1][:8]}
    missing = expected - names
    if missing:
        warnings.warn("profiles/pmc_traffic.json holds no counter pass of %s: bench.py will print roofline.traffic = null until "
                      "scripts/collect_profiles.sh has been run again" % sorted(missing))
    assert all(len(n.rsplit("_", 1)[1]) == 8 for n in expected)
