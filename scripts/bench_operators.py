"""Operator micro-benchmarks on device-resident pages (dev tool; numbers quoted in DESIGN.md):
FilterAndProject (Q6 shape), HashAggregation at several cardinalities (BenchmarkGroupByHash shape), hash join build /
probe (BenchmarkHashBuildAndJoinOperators shape).  rows/s = input rows / wall time including the final sync."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from presto_amd import _lib, abi, tpch
from presto_amd.expr import field
from presto_amd.operators import (FilterAndProjectOperator, HashAggregationOperator, HashBuilderOperator, LookupJoinOperator,
                                  LookupSourceFactory)
from presto_amd.page import Block, DeviceBuffer, Page
torch.cuda.set_device(0)
_lib.init(0)
which = sys.argv[1:] or ["fp", "agg", "join"]


def dev_block(type_, t):
    return Block(type_, abi.FLAT, t.numel(), values=DeviceBuffer(t.data_ptr(), t.numel() * t.element_size(), t))


def timeit(fn, reps=5):
    torch.cuda.synchronize()  # inputs were produced on torch's stream; operators run on their own
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


if "fp" in which:
    rows = 1 << 27
    dev = tpch.DeviceColumns(tpch.Q6_COLUMNS, 100.0, rows)
    pages = list(dev.pages(1 << 26))
    for name, f in (("q6 filter (1.9% pass)", tpch.q6_filter()), ("q1 filter (96% pass)", field(0, abi.DATE) <= 10471)):
        def run():
            op = FilterAndProjectOperator(tpch.Q6_TYPES, f, tpch.q6_projections() + [field(0, abi.DATE)], output_mem=abi.MEM_DEVICE)
            for p in pages:
                op.addInput(p)
                op.getOutput()
            op.finish()
            ms, n = op.kernelTime()
            op.close()
            return ms
        dt = timeit(run)
        print("FilterAndProject %-24s %.3g rows/s  (%.1f GB/s of the 28 B/row inputs)" % (name, rows / dt, rows * 28 / dt / 1e9))

if "agg" in which:
    rows = 1 << 26
    g = torch.Generator(device="cuda").manual_seed(1)
    vals = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
    for groups in [int(x) for x in os.environ.get("AGG_GROUPS", "4,64,1000,100000,3000000").split(",")]:
        keys = torch.randint(0, groups, (rows,), dtype=torch.int64, device="cuda", generator=g)
        page = Page([dev_block(abi.BIGINT, keys), dev_block(abi.DOUBLE, vals)], rows, abi.MEM_DEVICE)
        sub = [Page([Block(abi.BIGINT, abi.FLAT, 1 << 24, values=DeviceBuffer(keys.data_ptr() + 8 * i, 8 << 24, keys)),
                     Block(abi.DOUBLE, abi.FLAT, 1 << 24, values=DeviceBuffer(vals.data_ptr() + 8 * i, 8 << 24, vals))], 1 << 24, abi.MEM_DEVICE,
                    stable=os.environ.get("AGG_STABLE", "1") == "1")   # row ranges of resident tensors: PA_PAGE_STABLE (AGG_STABLE=0: pages of their own)
               for i in range(0, rows, 1 << 24)]
        def run():
            op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)],
                                         expected_groups=groups, output_mem=abi.MEM_DEVICE)
            for p in sub:
                op.addInput(p)
            op.finish()
            out = op.getOutput()
            n = out.position_count
            run.kernel_ms, run.launches = op.kernelTime()
            op.close()
            return n
        dt = timeit(run, reps=3)
        print("HashAggregation %8d groups: %.3g rows/s (%.1f GB/s of the 16 B/row inputs), %d groups out; wall %.2f ms, fused kernels %.2f ms in %d launches"
              % (groups, rows / dt, rows * 16 / dt / 1e9, run(), dt * 1e3, run.kernel_ms, run.launches))

if "dict" in which:
    # l_shipmode as a DictionaryBlock over 7 strings (what an ORC / Parquet reader hands over) under
    # `shipmode = 'MAIL' OR shipmode = 'SHIP'`, projecting another column: the dictionary-aware filter against the same
    # rows as a plain VariableWidthBlock
    from presto_amd.expr import constant, or_
    rows = 1 << 26
    g = torch.Generator(device="cuda").manual_seed(4)
    words = [b"REG AIR", b"AIR", b"RAIL", b"SHIP", b"TRUCK", b"MAIL", b"FOB"]
    ids = torch.randint(0, 7, (rows,), dtype=torch.int32, device="cuda", generator=g)
    qty = torch.randint(0, 50, (rows,), dtype=torch.int64, device="cuda", generator=g)
    dbytes = torch.tensor(list(b"".join(words)), dtype=torch.uint8, device="cuda")
    doffs = torch.tensor(np.cumsum([0] + [len(w) for w in words]), dtype=torch.int32, device="cuda")
    dictionary = Block(abi.VARCHAR, abi.VARWIDTH, 7, values=DeviceBuffer(dbytes.data_ptr(), dbytes.numel(), dbytes), offsets=DeviceBuffer(doffs.data_ptr(), 32, doffs))
    dpage = Page([Block(abi.VARCHAR, abi.DICTIONARY, rows, ids=DeviceBuffer(ids.data_ptr(), rows * 4, ids), dictionary=dictionary), dev_block(abi.BIGINT, qty)], rows, abi.MEM_DEVICE)
    lens = torch.tensor([len(w) for w in words], dtype=torch.int32, device="cuda")[ids.long()]
    offs = torch.zeros(rows + 1, dtype=torch.int32, device="cuda")
    offs[1:] = torch.cumsum(lens, 0).to(torch.int32)
    total = int(offs[-1])
    pos = torch.arange(total, device="cuda", dtype=torch.int64)
    row_of = torch.repeat_interleave(torch.arange(rows, device="cuda"), lens.long())
    flat = dbytes[(doffs.long()[ids.long()][row_of] + (pos - offs.long()[row_of]))].contiguous()
    fpage = Page([Block(abi.VARCHAR, abi.VARWIDTH, rows, values=DeviceBuffer(flat.data_ptr(), total, flat), offsets=DeviceBuffer(offs.data_ptr(), (rows + 1) * 4, offs)),
                  dev_block(abi.BIGINT, qty)], rows, abi.MEM_DEVICE)
    m = field(0, abi.VARCHAR)
    f = or_(m.eq(constant(b"MAIL", abi.VARCHAR)), m.eq(constant(b"SHIP", abi.VARCHAR)))
    for name, page in (("DictionaryBlock (dictionary-aware filter)", dpage), ("VariableWidthBlock (row by row)", fpage)):
        def run():
            op = FilterAndProjectOperator([abi.VARCHAR, abi.BIGINT], f, [field(1, abi.BIGINT)], output_mem=abi.MEM_DEVICE)
            op.addInput(page)
            out = op.getOutput()
            op.finish()
            n = out.position_count
            op.close()
            return n
        dt = timeit(run)
        print("FilterAndProject shipmode IN (MAIL, SHIP), %-42s %.3g rows/s, %d rows out" % (name, rows / dt, run()))

if "next" in which:
    # the SURVEY 8(f) operators: TopN, OrderBy, DynamicFilterSource over device pages; the page wire format over PCIe
    import ctypes as C
    from presto_amd.operators import DynamicFilterSourceOperator, OrderByOperator, TopNOperator
    from presto_amd.page import deserialize_page, serialize_page
    g = torch.Generator(device="cuda").manual_seed(5)
    rows = 1 << 26
    v = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
    k = torch.randint(0, 1 << 40, (rows,), dtype=torch.int64, device="cuda", generator=g)
    page = Page([dev_block(abi.DOUBLE, v), dev_block(abi.BIGINT, k)], rows, abi.MEM_DEVICE)
    def topn():
        op = TopNOperator([abi.DOUBLE, abi.BIGINT], 100, [0, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST])
        op.addInput(page); op.finish(); out = op.getOutput(); op.close(); return out.position_count
    dt = timeit(topn, reps=3)
    print("TopN 100 of %d rows (DOUBLE desc, BIGINT): %.3g rows/s (%.2f ms)" % (rows, rows / dt, dt * 1e3))
    srows = 1 << 24
    spage = Page([Block(abi.DOUBLE, abi.FLAT, srows, values=DeviceBuffer(v.data_ptr(), srows * 8, v)),
                  Block(abi.BIGINT, abi.FLAT, srows, values=DeviceBuffer(k.data_ptr(), srows * 8, k))], srows, abi.MEM_DEVICE)
    def orderby():
        op = OrderByOperator([abi.DOUBLE, abi.BIGINT], [0, 1], [1], [abi.ASC_NULLS_LAST], output_mem=abi.MEM_DEVICE)
        op.addInput(spage); op.finish(); out = op.getOutput(); n = out.position_count; op.close(); return n
    dt = timeit(orderby, reps=3)
    print("OrderBy %d rows by one BIGINT key, 16 B rows: %.3g rows/s (%.2f ms; stable sort of (image, row id) pairs over the images' varying bits)" % (srows, srows / dt, dt * 1e3))
    small = torch.randint(0, 5000, (rows,), dtype=torch.int64, device="cuda", generator=g)
    dpage = Page([dev_block(abi.BIGINT, small), dev_block(abi.DOUBLE, v)], rows, abi.MEM_DEVICE)
    def dynf():
        op = DynamicFilterSourceOperator([abi.BIGINT, abi.DOUBLE], [0], 10000, 1 << 20, 1 << 30)
        op.addInput(dpage); op.getOutput(); op.finish(); p = op.predicate(); op.close(); return len(p[0][1])
    dt = timeit(dynf, reps=3)
    print("DynamicFilterSource %d rows, one BIGINT channel with 5000 distinct values (limit 10000): %.3g rows/s (%.2f ms), %d values collected"
          % (rows, rows / dt, dt * 1e3, dynf()))
    wrows = 1 << 22
    wpage = Page([Block(abi.DOUBLE, abi.FLAT, wrows, values=DeviceBuffer(v.data_ptr(), wrows * 8, v)),
                  Block(abi.BIGINT, abi.FLAT, wrows, values=DeviceBuffer(k.data_ptr(), wrows * 8, k))], wrows, abi.MEM_DEVICE)
    from presto_amd._lib import check, lib
    cpage, keep = wpage.to_c()
    buf = np.zeros(wrows * 16 + 4096, dtype=np.uint8)
    size = lib().pa_page_serialize(C.byref(cpage), buf.ctypes.data, buf.nbytes, None)  # warm
    t0 = time.perf_counter()
    for _ in range(5):
        size = lib().pa_page_serialize(C.byref(cpage), buf.ctypes.data, buf.nbytes, None)
    t1 = time.perf_counter()
    handles = []
    for _ in range(5):
        h = C.c_void_p()
        check(lib().pa_page_deserialize(buf.ctypes.data, size, None, C.byref(h)))
        handles.append(h)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for h in handles:
        lib().pa_page_buffer_free(h)
    print("PagesSerde %d rows x 16 B through the C ABI (pageable host buffer): device page -> wire bytes %.1f GB/s, wire bytes -> device page %.1f GB/s"
          % (wrows, 5 * size / (t1 - t0) / 1e9, 5 * size / (t2 - t1) / 1e9))

if "vagg" in which:
    # VARCHAR(25) keys ("Customer#%09d" + padding, c_name / s_name shape): interned on the device, then grouped by id
    rows, width = 1 << 25, 24
    g = torch.Generator(device="cuda").manual_seed(3)
    vals = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g)
    for groups in [int(x) for x in os.environ.get("AGG_GROUPS", "4,1000,100000,3000000").split(",")]:
        ids = torch.randint(0, groups, (rows,), dtype=torch.int64, device="cuda", generator=g)
        digits = torch.stack([(ids // 10 ** k) % 10 + 48 for k in range(8, -1, -1)], dim=1).to(torch.uint8)
        text = torch.cat([torch.tensor(list(b"Customer#"), dtype=torch.uint8, device="cuda").expand(rows, 9), digits,
                          torch.full((rows, width - 18), 120, dtype=torch.uint8, device="cuda")], dim=1).contiguous()
        offs = (torch.arange(rows + 1, dtype=torch.int64, device="cuda") * width).to(torch.int32)
        chunk = 1 << 23
        sub = []
        for i in range(0, rows, chunk):
            o = offs[i:i + chunk + 1].contiguous()
            kb = Block(abi.VARCHAR, abi.VARWIDTH, chunk, values=DeviceBuffer(text.data_ptr(), text.numel(), text), offsets=DeviceBuffer(o.data_ptr(), o.numel() * 4, o))
            sub.append(Page([kb, Block(abi.DOUBLE, abi.FLAT, chunk, values=DeviceBuffer(vals.data_ptr() + 8 * i, 8 * chunk, vals))], chunk, abi.MEM_DEVICE))
        def run():
            op = HashAggregationOperator([abi.VARCHAR, abi.DOUBLE], [0], [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)],
                                         expected_groups=groups, output_mem=abi.MEM_DEVICE, type_params=[25, 0])
            for p in sub:
                op.addInput(p)
            op.finish()
            out = op.getOutput()
            n = out.position_count
            run.kernel_ms, run.launches = op.kernelTime()
            op.close()
            return n
        dt = timeit(run, reps=3)
        print("HashAggregation VARCHAR(25) key %8d groups: %.3g rows/s (%.1f GB/s of the 36 B/row inputs), %d groups out; wall %.2f ms, fused kernels %.2f ms"
              % (groups, rows / dt, rows * 36 / dt / 1e9, run(), dt * 1e3, run.kernel_ms))

if "join" in which:
    g = torch.Generator(device="cuda").manual_seed(2)
    for nb, npr in ((1 << 20, 1 << 26), (15_000_000, 1 << 26)):
        bkeys = torch.randperm(nb, device="cuda", generator=g).to(torch.int64) * 4
        bpay = torch.arange(nb, dtype=torch.int32, device="cuda")
        pkeys = torch.randint(0, nb * 8, (npr,), dtype=torch.int64, device="cuda", generator=g)  # ~50 % match
        pval = torch.rand(npr, dtype=torch.float64, device="cuda", generator=g)
        build = Page([dev_block(abi.BIGINT, bkeys), dev_block(abi.INTEGER, bpay)], nb, abi.MEM_DEVICE)
        probes = [Page([Block(abi.BIGINT, abi.FLAT, 1 << 24, values=DeviceBuffer(pkeys.data_ptr() + 8 * i, 8 << 24, pkeys)),
                        Block(abi.DOUBLE, abi.FLAT, 1 << 24, values=DeviceBuffer(pval.data_ptr() + 8 * i, 8 << 24, pval))], 1 << 24, abi.MEM_DEVICE)
                  for i in range(0, npr, 1 << 24)]
        bridge = LookupSourceFactory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = HashBuilderOperator(bridge, [abi.BIGINT, abi.INTEGER], [0], [1], expected_positions=nb)
        b.addInput(build)
        b.finish()
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        def run():
            j = LookupJoinOperator(bridge, [abi.BIGINT, abi.DOUBLE], [0], [0, 1], output_mem=abi.MEM_DEVICE)
            total = 0
            for p in probes:
                j.addInput(p)
                out = j.getOutput()
                total += out.position_count if out is not None else 0
            j.finish()
            j.close()
            return total
        dt = timeit(run, reps=3)
        print("HashJoin build %9d rows: %.3g rows/s (incl. allocation); probe %d rows: %.3g rows/s, %d matches" % (nb, nb / tb, npr, npr / dt, run()))
