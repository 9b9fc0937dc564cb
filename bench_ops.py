"""Operator benchmarks of bench.py's `operators` object: the reference's own micro-benchmark shapes (and larger ones of the same
kind) through the device operators, each with the oracle's twin of the same operator timed on the host beside it.

  hash_agg   HashAggregationOperator, (BIGINT key, DOUBLE value), sum + count(*):
             10 M rows / 3 M groups   core/trino-main/src/test/java/io/trino/operator/BenchmarkGroupByHash.java:68-71 (POSITIONS, GROUP_COUNT)
             64 M rows at 4 / 1 K / 100 K / 3 M groups
  hash_join  HashBuilderOperator + LookupJoinOperator on one BIGINT key:
             8 M build rows (every key once / five times), 1.4 M probe rows at match rate 0.1 / 1 / 2
                                      core/trino-main/src/test/java/io/trino/operator/join/BenchmarkHashBuildAndJoinOperators.java:103-110,192-199,260-303
             15 M unique random build keys, 2^26 random probe keys (one in eight matches); the same with about five build rows per key
  order_by   OrderByOperator, 2^24 rows x 16 B by one BIGINT key        topn   TopNOperator, 100 of 2^26 rows (DOUBLE desc, BIGINT)

Device figures: W warm-ups + K measured runs, median and min of the wall time of a whole operator life (create, addInput of
device-resident pages, finish, getOutput into HBM, close; device drained), rows/s from the median.  `frac` = algorithmic bytes
(the operator's input columns read once: SURVEY 8d's convention) / median time / 8 TB/s.  CPU figures: the oracle's operator
(test infrastructure, C, one thread = one reference Driver) on the same pages or on a stated sample of them, one run each --
they are reported baselines, not targets.  Only bench.py calls this module.

No torch here: the inputs are numpy arrays uploaded through the C ABI (pa_device_malloc + pa_memcpy_h2d) -- the process holds ONE
ROCm stack, the library's."""
import time

HBM_PEAK_GBS = 8000.0


def _timed(fn, sync, warmup=2, runs=5):
    for _ in range(warmup):
        fn()
    sync()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def _entry(rows, alg_bytes, median, best, **more):
    gbs = alg_bytes / median / 1e9
    e = {"rows": rows, "value": rows / median, "unit": "rows/s", "ms_median": median * 1e3, "ms_min": best * 1e3,
         "algorithmic_bytes": alg_bytes, "achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS}
    e.update(more)
    return e


def _cpu(rows, seconds, what):
    return {"value": rows / seconds, "unit": "rows/s", "rows": rows, "seconds": seconds, "cores": 1, "kind": "port", "sample": what}


class DeviceArray:
    """A numpy array and its copy in HBM (the host copy feeds the CPU twin)."""

    def __init__(self, host):
        import numpy as np
        from presto_amd._lib import DeviceAllocation, check, lib
        self.host = np.ascontiguousarray(host)
        self.alloc = DeviceAllocation(max(self.host.nbytes, 16))
        check(lib().pa_memcpy_h2d(self.alloc.ptr, self.host.ctypes.data, self.host.nbytes, None))

    def numel(self):
        return self.host.size

    def free(self):
        self.alloc.free()


def run(cpu=True, only=None):
    """only: None = everything, else a set of section names out of {'hash_agg', 'hash_join', 'order_by', 'topn'} (profiling runs)"""
    want = lambda name: only is None or name in only
    import numpy as np
    from presto_amd import _lib, abi
    from presto_amd.operators import (HashAggregationOperator, HashBuilderOperator, LookupJoinOperator, LookupSourceFactory, OrderByOperator,
                                      TopNOperator)
    from presto_amd.page import Block, DeviceBuffer, Page
    if cpu:
        from oracle import oracle as O
        O.build()

    def sync():
        _lib.device_synchronize()

    def dev_block(type_, t, first=0, n=None):
        n = t.numel() - first if n is None else n
        w = t.host.itemsize
        return Block(type_, abi.FLAT, n, values=DeviceBuffer(t.alloc.ptr + w * first, w * n, t))

    def host_page(types, arrays, n):
        return Page([Block.flat(t, x.host[:n]) for t, x in zip(types, arrays)], n)

    out = {}
    rng = np.random.default_rng(1)

    # ---- HashAggregation ----
    aggs = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
    vals = DeviceArray(rng.random(1 << 26))
    out["hash_agg"] = []
    agg_cases = ((10_000_000, 3_000_000, "BenchmarkGroupByHash.java:68-71"), (1 << 26, 4, None), (1 << 26, 1000, None),
                 (1 << 26, 100_000, None), (1 << 26, 3_000_000, None))
    for rows, groups, shape in (agg_cases if want("hash_agg") else ()):
        keys = DeviceArray(rng.integers(0, groups, rows, dtype=np.int64))
        sync()
        chunk = 1 << 24
        pages = [Page([dev_block(abi.BIGINT, keys, i, min(chunk, rows - i)), dev_block(abi.DOUBLE, vals, i, min(chunk, rows - i))],
                      min(chunk, rows - i), abi.MEM_DEVICE, stable=True) for i in range(0, rows, chunk)]
        seen = {}

        def agg():
            op = HashAggregationOperator([abi.BIGINT, abi.DOUBLE], [0], aggs, expected_groups=groups, output_mem=abi.MEM_DEVICE)
            for p in pages:
                op.addInput(p)
            op.finish()
            seen["groups"] = op.getOutput().position_count
            op.close()
        med, best = _timed(agg, sync, warmup=2, runs=5)
        e = _entry(rows, rows * 16, med, best, groups=groups, groups_out=seen["groups"], reference_shape=shape)
        if cpu:
            n = min(rows, 8_000_000)
            page = host_page([abi.BIGINT, abi.DOUBLE], [keys, vals], n)
            t0 = time.perf_counter()
            ref = O.HashAggregation([abi.BIGINT, abi.DOUBLE], [0], aggs, expected_groups=groups)
            ref.add_page(page)
            ref.build_result()
            e["cpu"] = _cpu(n, time.perf_counter() - t0, "the first %d rows of the same page through the oracle's HashAggregationOperator "
                                                         "(BigintGroupByHash, expectedGroups as on the device)" % n)
            ref.close()
        out["hash_agg"].append(e)
        del pages
        keys.free()

    # ---- HashBuilder + LookupJoin ----
    types = [abi.BIGINT, abi.BIGINT]
    out["hash_join"] = []

    def join_case(name, bkeys, pkeys, shape, cpu_probe_rows):
        nb, npr = bkeys.numel(), pkeys.numel()
        bpay = DeviceArray(np.arange(nb, dtype=np.int64))
        ppay = DeviceArray(np.arange(npr, dtype=np.int64))
        sync()
        # the build side arrives as ONE page whose owner keeps it alive until the lookup source lets go of it (PA_PAGE_RETAINED: what a
        # reference does for a Java Page in PagesIndex) -- the build columns are the page's block arrays, nothing is copied
        released = []
        build = Page([dev_block(abi.BIGINT, bkeys), dev_block(abi.BIGINT, bpay)], nb, abi.MEM_DEVICE, on_release=lambda: released.append(1))
        chunk = 1 << 24
        probes = [Page([dev_block(abi.BIGINT, pkeys, i, min(chunk, npr - i)), dev_block(abi.BIGINT, ppay, i, min(chunk, npr - i))],
                       min(chunk, npr - i), abi.MEM_DEVICE, stable=True) for i in range(0, npr, chunk)]
        state = {}

        def do_build():
            old = state.pop("bridge", None)
            if old is not None:
                state.pop("builder").close()
                old.destroy()
            bridge = LookupSourceFactory()
            b = HashBuilderOperator(bridge, types, [0], [1], expected_positions=nb)
            b.addInput(build)
            b.finish()
            state["bridge"], state["builder"] = bridge, b
        bmed, bbest = _timed(do_build, sync, warmup=1, runs=5)

        def do_probe():
            j = LookupJoinOperator(state["bridge"], types, [0], [1], output_mem=abi.MEM_DEVICE)
            total = 0
            for p in probes:
                j.addInput(p)
                o = j.getOutput()
                total += o.position_count if o is not None else 0
            j.finish()
            j.close()
            state["matches"] = total
        pmed, pbest = _timed(do_probe, sync, warmup=1, runs=5)
        # algorithmic bytes: the build reads key + payload once (16 B/row); the probe reads its key column and the output
        # channel of the probe rows (16 B/row) and writes 8 B per output row (the build payload is gathered per match: 8 B)
        e = {"case": name, "reference_shape": shape, "build": _entry(nb, nb * 16, bmed, bbest),
             "probe": _entry(npr, npr * 16 + state["matches"] * 16, pmed, pbest, matches=state["matches"])}
        if cpu:
            n = min(npr, cpu_probe_rows)
            bp = host_page(types, [bkeys, bpay], nb)
            pp = host_page(types, [pkeys, ppay], n)
            t0 = time.perf_counter()
            j = O.HashJoin(types, [0], [1])
            j.add_build_page(bp)
            j.build()
            t1 = time.perf_counter()
            j.probe(pp, types, [0], [1])
            t2 = time.perf_counter()
            j.close()
            e["build"]["cpu"] = _cpu(nb, t1 - t0, "the same build page through the oracle's PagesIndex + PagesHash")
            e["probe"]["cpu"] = _cpu(n, t2 - t1, "the first %d probe rows through the oracle's JoinProbe / DefaultPageJoiner" % n)
        state.pop("builder").close()
        state.pop("bridge").destroy()
        bpay.free()
        ppay.free()
        out["hash_join"].append(e)

    ref = "BenchmarkHashBuildAndJoinOperators.java:103-110,192-199,260-303"
    for repetition in ((1, 5) if want("hash_join") else ()):
        nb = 8_000_000
        max_value = nb // repetition + 40
        # addSequencePage(newRows, ..., (rows + 30) % maxValue, ...) page by page (1024 rows): BIGINT channel 1 = (rows + 30) % maxValue + i
        i = np.arange(nb, dtype=np.int64)
        bkeys = DeviceArray((i // 1024 * 1024 + 30) % max_value + i % 1024)
        for match_rate in (0.1, 1, 2):
            prng = np.random.default_rng(42)
            remaining, keys = 1_400_000, []
            rolls = prng.random(1_400_000)
            if match_rate < 1:
                k = 30 + np.arange(remaining, 0, -1, dtype=np.int64)
                keys = np.where(rolls > match_rate, -k, k)
            elif match_rate > 1:
                counts = np.floor(rolls * 2 * match_rate + 1).astype(np.int64)
                ends = np.cumsum(counts)
                take = int(np.searchsorted(ends, remaining)) + 1
                rem = remaining - np.concatenate([[0], ends[:take - 1]])
                keys = np.repeat(30 + rem, counts[:take])[:remaining]
            else:
                keys = 30 + np.arange(remaining, 0, -1, dtype=np.int64)
            pkeys = DeviceArray(np.ascontiguousarray(keys, dtype=np.int64))
            join_case("8 M build rows x%d, 1.4 M probe rows, match rate %g" % (repetition, match_rate), bkeys, pkeys, ref, 1_400_000)
            pkeys.free()
        bkeys.free()
    if want("hash_join"):
        nb = 15_000_000
        unique = DeviceArray(rng.permutation(nb).astype(np.int64) * 4)
        pkeys = DeviceArray(rng.integers(0, nb * 8, 1 << 26, dtype=np.int64))
        join_case("15 M unique random build keys, 2^26 random probe keys (one in eight matches)", unique, pkeys, None, 4_000_000)
        unique.free()
        pkeys.free()
        dup = DeviceArray(rng.integers(0, nb // 5, nb, dtype=np.int64) * 4)
        pkeys5 = DeviceArray(rng.integers(0, nb // 5 * 8, 1 << 26, dtype=np.int64))
        join_case("15 M random build keys, about 5 rows per key, 2^26 random probe keys (one in eight matches, about 5 rows each)", dup, pkeys5, None, 2_000_000)
        dup.free()
        pkeys5.free()

    # ---- OrderBy / TopN ----
    rows = 1 << 26
    v = vals   # 2^26 uniform doubles
    k = DeviceArray(rng.integers(0, 1 << 40, rows, dtype=np.int64))
    sync()
    srows = 1 << 24
    # (a stable page, like the aggregation's: PagesIndex.addPage keeps the reference's Page, it does not copy it)
    spage = Page([dev_block(abi.DOUBLE, v, 0, srows), dev_block(abi.BIGINT, k, 0, srows)], srows, abi.MEM_DEVICE, stable=True)

    def order_by():
        op = OrderByOperator([abi.DOUBLE, abi.BIGINT], [0, 1], [1], [abi.ASC_NULLS_LAST], output_mem=abi.MEM_DEVICE)
        op.addInput(spage)
        op.finish()
        op.getOutput()
        op.close()
    if want("order_by"):
        med, best = _timed(order_by, sync, warmup=1, runs=5)
        out["order_by"] = _entry(srows, srows * 16 * 2, med, best, shape="2^24 rows of (DOUBLE, BIGINT) in one stable device page, by the BIGINT channel ascending; bytes = rows read + written once")
    if cpu and want("order_by"):
        n = 1 << 22
        kk = np.ascontiguousarray(k.host[:n])
        t0 = time.perf_counter()
        O.sort_positions_bigint(kk)
        out["order_by"]["cpu"] = _cpu(n, time.perf_counter() - t0, "the first %d rows: PagesIndexOrdering.quickSort over row positions (oracle, C); "
                                                                   "the sort only, without gathering the output page" % n)
    page = Page([dev_block(abi.DOUBLE, v), dev_block(abi.BIGINT, k)], rows, abi.MEM_DEVICE)

    def topn():
        op = TopNOperator([abi.DOUBLE, abi.BIGINT], 100, [0, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST])
        op.addInput(page)
        op.finish()
        op.getOutput()
        op.close()
    if not want("topn"):
        return out
    med, best = _timed(topn, sync, warmup=1, runs=5)
    out["topn"] = _entry(rows, rows * 16, med, best, shape="100 of 2^26 rows of (DOUBLE, BIGINT), DOUBLE descending then BIGINT ascending")
    if cpu:
        vv, kk = v.host, k.host
        t0 = time.perf_counter()
        O.topn_positions_double_desc_bigint_asc(vv, kk, 100)
        out["topn"]["cpu"] = _cpu(rows, time.perf_counter() - t0, "the same page: TopNProcessor's bounded heap (oracle, C)")
    return out
