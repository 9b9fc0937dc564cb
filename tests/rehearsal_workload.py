"""Checker workload for rehearsing bench.py's multi-rank control flow on CPU ranks (gloo): the same step / q3_step /
roofline surface as bench.DeviceWorkload, computed by the oracle on tiny row-range shards, with the stand-in exchange
(tests/gloo_standin.py) where the device workload runs the native one.  Test infrastructure only."""
from oracle import oracle as O
from presto_amd import abi, tpch
from presto_amd.expr import field
from presto_amd.page import Block, Page
from tests.gloo_standin import StandinExchangeOperator


def host_table(columns, sf, first_row, n):
    blocks = []
    for c in columns:
        v, o = O.tpch_column(c, sf, first_row, n)
        t = abi.TPCH_COLUMN_TYPE[c]
        blocks.append(Block.varwidth(v, o) if t == abi.VARCHAR else Block.flat(t, v))
    return Page(blocks, n)


def exchange(page, types, channels):
    ex = StandinExchangeOperator(O, types, channels)
    if page is not None and page.position_count:
        ex.addInput(page)
    ex.finish()
    return ex.getOutput()


def empty_page(types):
    return Page([Block.varchar([]) if t == abi.VARCHAR else Block.flat(t, []) for t in types], 0)


class OracleFinalOperator:
    """Operator-protocol adapter over the oracle's Step.FINAL aggregation."""

    def __init__(self, types, group_by, aggregates):
        self.agg = O.HashAggregation(types, group_by, aggregates, step=abi.STEP_FINAL)

    def addInput(self, page):
        self.agg.add_page(page)

    def finish(self):
        pass

    def getOutput(self):
        return self.agg.build_result()


class RehearsalWorkload:
    def __init__(self, args, rank, world, device, scaling=None, with_q3=None):
        O.build()
        self.args, self.rank, self.world = args, rank, world
        self.scaling = scaling or getattr(args, "scaling", "weak")
        self.queries = ["q1", "q6"]
        self.results = {}
        self.q3_on = bool(args.q3) if with_q3 is None else with_q3
        self.q3_sf = args.q3_sf or args.sf
        self.q3_counters = {}

        def shard(rows_of, sf, multiple=4):   # bench.DeviceWorkload's row-range shards
            n = rows_of(sf)
            if self.scaling == "weak" or world == 1:
                return sf * world, rank * n, n
            lo = n * rank // world // multiple * multiple
            hi = n if rank == world - 1 else n * (rank + 1) // world // multiple * multiple
            return sf, lo, hi - lo

        self.total_sf, first, self.rows = shard(tpch.lineitem_rows, args.sf)
        self.job_rows = tpch.lineitem_rows(args.sf) * (world if self.scaling == "weak" else 1)
        self.q6_page = host_table(tpch.Q6_COLUMNS, self.total_sf, first, self.rows)
        self.q1_page = host_table(tpch.Q1_COLUMNS, self.total_sf, first, self.rows)
        self.merger = None
        if world > 1:
            from presto_amd.exchange import PartialStateMerger, partial_layout
            self.merger = PartialStateMerger()
            t6, f6 = partial_layout([], tpch.Q6_AGGREGATES)
            t1, f1 = partial_layout([abi.VARCHAR, abi.VARCHAR], tpch.Q1_AGGREGATES)
            self.final_operators = {"q6": lambda: OracleFinalOperator(t6, [], f6), "q1": lambda: OracleFinalOperator(t1, [0, 1], f1)}
        if self.q3_on:
            sf = self.q3_sf
            self.q3_rows_job = tuple(f(sf) * (world if self.scaling == "weak" else 1) for f in (tpch.customer_rows, tpch.orders_rows, tpch.lineitem_rows))
            ct, cf, nc = shard(tpch.customer_rows, sf, 20)
            ot, of, no = shard(tpch.orders_rows, sf)
            lt, lf, nl = shard(tpch.lineitem_rows, sf)
            self.q3_rows = (nc, no, nl)
            self.customer = host_table(tpch.CUSTOMER_COLUMNS, ct, cf, nc)
            self.orders = host_table(tpch.ORDERS_COLUMNS, ot, of, no)
            self.lineitem = host_table(tpch.Q3_LINEITEM_COLUMNS, lt, lf, nl)

    def synchronize(self):
        pass

    def _aggregate(self, page, filter_expr, projections, group_by, aggregates, step):
        agg = O.HashAggregation([p.type for p in projections], group_by, aggregates, step=step)
        if page.position_count:
            selected = O.filter_project(page, filter_expr, projections)
            if selected is not None and selected.position_count:
                agg.add_page(selected)
        return agg.build_result()

    def step(self, timed):
        step = abi.STEP_PARTIAL if self.world > 1 else abi.STEP_SINGLE
        out = {"q6": self._aggregate(self.q6_page, tpch.q6_filter(), tpch.q6_projections(), [], tpch.Q6_AGGREGATES, step),
               "q1": self._aggregate(self.q1_page, tpch.q1_filter(), tpch.q1_projections(), tpch.Q1_GROUP_BY, tpch.Q1_AGGREGATES, step)}
        if self.world > 1:
            out = self.merger.merge(out, self.final_operators)
        if out is not None:
            self.results.update({q: ([] if p is None else p.to_rows()) for q, p in out.items()})

    def rows_per_step(self):
        return 2 * self.rows

    def job_rows_per_step(self):
        return 2 * self.job_rows

    def roofline(self, name, steps, pmc):
        return None

    def workload_name(self):
        return "rehearsal: oracle operators on CPU ranks, SF%g, %s scaling" % (self.args.sf, self.scaling)

    def q3_step(self):
        """presto_amd/q3.py's three pipelines with the oracle's operators and the stand-in exchange."""
        c = O.filter_project(self.customer, tpch.q3_customer_filter(), [field(0, abi.BIGINT)])
        c = exchange(c, [abi.BIGINT], [0])
        j1 = O.HashJoin([abi.BIGINT], [0], [])
        if c is not None:
            j1.add_build_page(c)
        j1.build()
        o = O.filter_project(self.orders, tpch.q3_orders_filter(), [field(i, t) for i, t in enumerate(tpch.ORDERS_TYPES)])
        o = exchange(o, tpch.ORDERS_TYPES, [1]) or empty_page(tpch.ORDERS_TYPES)
        oc, _, _ = j1.probe(o, tpch.ORDERS_TYPES, [1], [0, 2, 3])
        joined_types = [abi.BIGINT, abi.DATE, abi.INTEGER]
        oc = exchange(oc, joined_types, [0])
        j2 = O.HashJoin(joined_types, [0], [1, 2])
        if oc is not None:
            j2.add_build_page(oc)
        j2.build()
        l = O.filter_project(self.lineitem, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections())
        l = exchange(l, [abi.BIGINT, abi.DOUBLE], [0]) or empty_page([abi.BIGINT, abi.DOUBLE])
        joined, _, _ = j2.probe(l, [abi.BIGINT, abi.DOUBLE], [0], [0, 1])
        agg = O.HashAggregation([abi.BIGINT, abi.DOUBLE, abi.DATE, abi.INTEGER], [0, 2, 3], [(abi.AGG_SUM, 1, abi.DOUBLE)], expected_groups=1000)
        if joined.position_count:
            agg.add_page(joined)
        grouped = agg.build_result()
        rows = O.topn([grouped], 10, [3, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST]) if grouped is not None and grouped.position_count else []
        self.results["q3"] = [list(r) for r in rows]
        self.q3_counters = {"build2_rows": 0 if oc is None else oc.position_count, "lineitem_pipeline_ms": 0.0}

    def q3_input_rows(self):
        return sum(self.q3_rows)

    def q3_job_input_rows(self):
        return sum(self.q3_rows_job)

    def q3_algorithmic_bytes(self):
        nc, no, nl = self.q3_rows
        return nc * 21 + no * 24 + nl * 28

    def close(self):
        pass
