"""TPC-H Q3 as three Driver pipelines of device operators (BASELINE config #4), one rank per GPU:

    customer -> FilterAndProject(mktsegment = 'BUILDING'; custkey)            [-> exchange by custkey]  -> HashBuilder(b1)
    orders   -> FilterAndProject(orderdate < DATE)  [-> exchange by custkey]  -> LookupJoin(b1)
                                                    [-> exchange by orderkey] -> HashBuilder(b2)
    lineitem -> FilterAndProject(shipdate > DATE; orderkey, revenue) [-> exchange by orderkey] -> LookupJoin(b2)
             -> HashAggregation(orderkey, orderdate, shippriority; sum(revenue), count(*))
             [-> TopN(10; revenue DESC, orderdate ASC)]      the query's ORDER BY ... LIMIT 10, when asked for

The bracketed steps exist when the process group has more than one rank: they are the reference's hash-partitioned
exchange between the stages of a distributed join (PartitionedOutputOperator.java:411-431 routing rule,
LocalPartitionGenerator.java:45-65 for a power-of-two fan-out): the native exchange of include/presto_amd.h
(`presto_amd.exchange.ExchangeOperator` = PartitionedOutput sink + exchange source; one count all-gather and one variable
all-to-all over RCCL per exchange).  After the last exchange every orderkey lives on exactly one rank, so the grouped result
of a rank is final (disjoint groups; no PARTIAL/FINAL merge is needed), as in the reference's plan where the final
aggregation is partitioned on the group keys.

Every rank runs the same pipelines in the same order, so every rank reaches the collectives of an exchange (and of a
shared dynamic filter) in the same order; page counts may differ per rank -- an exchange transfers once, when the
producing side of the rank has finished."""
from . import abi, tpch
from .exchange import ExchangeOperator
from .expr import field
from .operators import (Driver, FilterAndProjectOperator, FilterAndProjectOperatorFactory, FusedJoinAggregationOperator,
                        FusedJoinAggregationOperatorFactory, FusedJoinOperatorFactory, HashAggregationOperator,
                        HashBuilderOperatorFactory, LookupJoinOperator, LookupSourceFactory, TopNOperatorFactory)


AGG_TYPES = [abi.BIGINT, abi.DOUBLE, abi.DATE, abi.INTEGER]       # lineitem JOIN orders: orderkey, revenue, orderdate, shippriority
AGG_GROUP_BY = [0, 2, 3]
AGG_AGGREGATES = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
ORDERS_JOINED_TYPES = [abi.BIGINT, abi.DATE, abi.INTEGER]           # orders JOIN customer: orderkey, orderdate, shippriority
RESULT_TYPES = [abi.BIGINT, abi.DATE, abi.INTEGER, abi.DOUBLE, abi.BIGINT]  # orderkey, orderdate, shippriority, revenue, count


# The planner's part of the one-rank plan with fused probes (LocalExecutionPlanner: OperatorFactory objects, made once per plan and
# stream; a Driver then only calls createOperator): the serialised descriptors of the operators the three pipelines create.
_FACTORIES = {}


def _factory(key, make):
    f = _FACTORIES.get(key)
    if f is None:
        if len(_FACTORIES) > 64:
            _FACTORIES.clear()
        f = _FACTORIES[key] = make()
    return f


def run(customer_pages, orders_pages, lineitem_pages, stream, comm=None, expected_groups=100000, distributed=None,
        result_mem=abi.MEM_HOST, top_n=0, with_count=True, dynamic_filters=True, fused_probe=True, topn_hint=True):
    """Runs the three pipelines on this rank's pages; returns (result pages, counters).  `stream` is the HIP stream
    handle every operator (and the exchange) runs on; `comm` the presto_amd.exchange.Comm of the ranks (None: one rank, no
    exchange steps; distributed=True with a one-rank comm still runs them); result_mem = where the grouped result is left
    (PA_MEM_DEVICE when a device operator consumes it).  top_n > 0 appends the query's TopN (revenue DESC, orderdate ASC):
    every rank then returns its own top_n rows -- the groups of different ranks are disjoint, so the query result is the top_n
    of their union.  with_count=False runs the query as TPC-H states it (sum(revenue) only); the count(*) column is there for
    the parity tests.  dynamic_filters: the filters upstream of the two probes also drop the rows whose join key matches no
    build key (the joins' dynamic filters, applied where Trino applies them); with exchange steps the filter is the union of
    every rank's build-key bitmap (pa_lookup_source_shared_key_bitmap), so that rows are dropped before they are exchanged.
    fused_probe: the orders pipeline's FilterAndProject -> LookupJoin run behind one handle (pa_fused_join_create), and
    the lineitem pipeline's FilterAndProject -> LookupJoin -> HashAggregation run behind one handle
    (pa_fused_join_aggregation_create: one generated kernel over the lineitem pages -- orderkey is unique on the build side);
    with exchange steps the filter stays in front of the exchange and the fused operator takes the exchanged pages.
    False: the three operators on their own (the independent path of the parity tests).  topn_hint (with top_n): the final
    aggregation is told that a TopN is its only consumer (pa_aggregation_set_output_topn_hint)."""
    if distributed is None:
        distributed = comm is not None and comm.world > 1
    if distributed and comm is None:
        raise ValueError("exchange steps need a Comm")
    dev = abi.MEM_DEVICE
    s = stream
    exchanges = []

    def exchange(types, channels):
        if not distributed:
            return []
        ex = ExchangeOperator(comm, types, channels, stream=s)
        exchanges.append(ex)
        return [ex]

    import time
    from ._lib import check, lib
    counters = {}

    def dynamic_filter(fp, channel, bridge, name):
        """Installs the join's dynamic filter in the FilterAndProject upstream of its probe.  With exchange steps the filter
        runs BEFORE the exchange, so it must know the build keys of every rank (collective)."""
        if not dynamic_filters:
            return
        if not distributed:
            counters[name] = fp.setDynamicFilter(channel, bridge)
            return
        shared = bridge.sharedKeyBitmap(comm, True, s)  # the builds are partitioned on their join key
        if shared is None:
            counters[name] = False  # no keys anywhere, or too sparse for a bitmap to pay
            return
        bits, lo, key_range = shared
        fp.setDynamicFilterBitmap(channel, bits, lo, key_range, keep=bridge)
        counters[name] = True

    t0 = time.perf_counter()

    def lap(name):  # wall time of a pipeline incl. its device work (the next pipeline needs its lookup source anyway)
        nonlocal t0
        check(lib().pa_stream_synchronize(s))
        t1 = time.perf_counter()
        counters[name + "_ms"] = (t1 - t0) * 1e3
        t0 = t1

    # pipeline 1
    b1 = LookupSourceFactory()
    Driver(customer_pages, [
        # (the pages that feed a HashBuilder are handed over with their buffers: the build side is read where the filter wrote it)
        _factory(("customer", s), lambda: FilterAndProjectOperatorFactory(tpch.CUSTOMER_TYPES, tpch.q3_customer_filter(), [field(0, abi.BIGINT)],
                                                                          output_mem=dev, stream=s, output_handover=True)).createOperator(),
        *exchange([abi.BIGINT], [0]),
        _factory(("build1", s), lambda: HashBuilderOperatorFactory([abi.BIGINT], [0], [], stream=s)).createOperator(b1)]).run()
    lap("customer_pipeline")
    # pipeline 2
    b2 = LookupSourceFactory()
    if fused_probe and not distributed:
        # FilterAndProject -> LookupJoin as one operator: custkey is unique on the build side (the join's dynamic filter is the
        # fused kernels' own bitmap test)
        if dynamic_filters:
            counters["orders_dynamic_filter"] = "fused"
        orders_head = [_factory(("orders", s), lambda: FusedJoinOperatorFactory(
            tpch.ORDERS_TYPES, tpch.q3_orders_filter(), [field(i, t) for i, t in enumerate(tpch.ORDERS_TYPES)], [1], [0, 2, 3], output_mem=dev,
            stream=s, output_handover=True)).createOperator(b1)]
    else:
        orders_projections = [field(i, t) for i, t in enumerate(tpch.ORDERS_TYPES)]
        orders_fp = FilterAndProjectOperator(tpch.ORDERS_TYPES, tpch.q3_orders_filter(), orders_projections, output_mem=dev, stream=s)
        dynamic_filter(orders_fp, 1, b1, "orders_dynamic_filter")
        orders_head = [orders_fp, *exchange(tpch.ORDERS_TYPES, [1]), LookupJoinOperator(b1, tpch.ORDERS_TYPES, [1], [0, 2, 3], output_mem=dev, stream=s)]
    Driver(orders_pages, [
        *orders_head,
        *exchange(ORDERS_JOINED_TYPES, [0]),
        _factory(("build2", s), lambda: HashBuilderOperatorFactory(ORDERS_JOINED_TYPES, [0], [1, 2], stream=s)).createOperator(b2)]).run()
    lap("orders_pipeline")
    # pipeline 3
    # orderkey is unique on the build side, so its row count bounds the groups (what the planner's stats estimate)
    expected_groups = max(expected_groups, min(b2.positionCount(), 1 << 28))
    aggregates = AGG_AGGREGATES if with_count else AGG_AGGREGATES[:1]
    result_types = RESULT_TYPES if with_count else RESULT_TYPES[:4]
    agg_mem = dev if top_n else result_mem
    top = [_factory(("top", s, top_n, with_count, result_mem), lambda: TopNOperatorFactory(
        result_types, top_n, [3, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST], output_mem=result_mem, stream=s)).createOperator()] if top_n else []
    if fused_probe and not distributed:
        # (the join's dynamic filter is the fused kernel's own bitmap test)
        if dynamic_filters:
            counters["lineitem_dynamic_filter"] = "fused"
        head = [_factory(("lineitem", s, with_count, expected_groups, agg_mem), lambda: FusedJoinAggregationOperatorFactory(
            tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), [0], [0, 1], AGG_TYPES, AGG_GROUP_BY, aggregates,
            expected_groups=expected_groups, output_mem=agg_mem, stream=s)).createOperator(b2)]
    else:
        lineitem_fp = FilterAndProjectOperator(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), output_mem=dev, stream=s)
        dynamic_filter(lineitem_fp, 0, b2, "lineitem_dynamic_filter")
        exchanged = [abi.BIGINT, abi.DOUBLE]
        if fused_probe:
            tail = [FusedJoinAggregationOperator(b2, exchanged, None, [field(0, abi.BIGINT), field(1, abi.DOUBLE)], [0], [0, 1], AGG_TYPES, AGG_GROUP_BY,
                                                 aggregates, expected_groups=expected_groups, output_mem=agg_mem, stream=s)]
        else:
            tail = [LookupJoinOperator(b2, exchanged, [0], [0, 1], output_mem=dev, stream=s),
                    HashAggregationOperator(AGG_TYPES, AGG_GROUP_BY, aggregates, expected_groups=expected_groups, output_mem=agg_mem, stream=s)]
        head = [lineitem_fp, *exchange(exchanged, [0]), *tail]
    if top_n and topn_hint:
        # the planner's note (LocalExecutionPlanner.visitTopN over visitAggregation): the aggregation's only consumer is this TopN, so
        # groups that cannot be among its rows may stay in the table -- thousands of rows leave it instead of millions
        agg_op = head[0] if (fused_probe and not distributed) else head[-1]
        counters["topn_hint"] = agg_op.setOutputTopNHint(top_n, [3, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST])
    out = Driver(lineitem_pages, head + top).run()
    lap("lineitem_pipeline")
    if fused_probe and not distributed:
        # HIP-event time of the fused filter -> probe -> aggregate launches (pa_op_kernel_time of the handle)
        counters["lineitem_fused_kernel_ms"], counters["lineitem_fused_launches"] = head[0].kernelTime()
        counters["lineitem_fused_kernel"] = head[0].kernelName()
    counters["build1_rows"] = b1.positionCount()
    counters["build2_rows"] = b2.positionCount()
    if exchanges:
        st = [ex.stats() for ex in exchanges]
        counters["exchange_rows_sent"] = sum(x[0] for x in st)
        counters["exchange_rows_received"] = sum(x[1] for x in st)
        counters["exchange_bytes_remote"] = sum(x[2] for x in st)
        counters["exchange_transfer_ms"] = sum(x[3] for x in st)
        for ex in exchanges:
            ex.close()
    return out, counters
