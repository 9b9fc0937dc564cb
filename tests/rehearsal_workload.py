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


class RehearsalWorkload:
    def __init__(self, args, rank, world, device):
        O.build()
        self.args, self.rank, self.world = args, rank, world
        self.queries = ["q1", "q6"]
        self.rows = tpch.lineitem_rows(args.sf)
        self.total_sf = args.sf * world
        self.results = {}
        self.q3_on = bool(args.q3)
        self.q3_sf = args.q3_sf or args.sf
        self.q3_counters = {}
        first = rank * self.rows
        self.q6_cols = [O.tpch_column(c, self.total_sf, first, self.rows)[0] for c in tpch.Q6_COLUMNS]
        cols = [O.tpch_column(c, self.total_sf, first, self.rows) for c in tpch.Q1_COLUMNS]
        self.q1_args = [cols[0][0], cols[0][1], cols[1][0], cols[1][1]] + [c[0] for c in cols[2:]]
        if self.q3_on:
            sf = self.q3_sf
            nc, no, nl = tpch.customer_rows(sf), tpch.orders_rows(sf), tpch.lineitem_rows(sf)
            self.q3_rows = (nc, no, nl)
            t = sf * world
            self.customer = host_table(tpch.CUSTOMER_COLUMNS, t, rank * nc, nc)
            self.orders = host_table(tpch.ORDERS_COLUMNS, t, rank * no, no)
            self.lineitem = host_table(tpch.Q3_LINEITEM_COLUMNS, t, rank * nl, nl)

    def synchronize(self):
        pass

    def step(self, timed):
        self.results["q6"] = [O.q6(*self.q6_cols)]
        self.results["q1"] = O.q1(self.q1_args)

    def rows_per_step(self):
        return 2 * self.rows

    def roofline(self, name, steps, pmc):
        return None

    def workload_name(self):
        return "rehearsal: oracle operators on CPU ranks, SF%g per rank" % self.args.sf

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

    def q3_algorithmic_bytes(self):
        nc, no, nl = self.q3_rows
        return nc * 21 + no * 24 + nl * 28

    def close(self):
        pass
