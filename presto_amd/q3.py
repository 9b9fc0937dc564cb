"""TPC-H Q3 as three Driver pipelines of device operators (BASELINE config #4), one rank per GPU:

    customer -> FilterAndProject(mktsegment = 'BUILDING'; custkey)            [-> exchange by custkey]  -> HashBuilder(b1)
    orders   -> FilterAndProject(orderdate < DATE)  [-> exchange by custkey]  -> LookupJoin(b1)
                                                    [-> exchange by orderkey] -> HashBuilder(b2)
    lineitem -> FilterAndProject(shipdate > DATE; orderkey, revenue) [-> exchange by orderkey] -> LookupJoin(b2)
             -> HashAggregation(orderkey, orderdate, shippriority; sum(revenue), count(*))
             [-> TopN(10; revenue DESC, orderdate ASC)]      the query's ORDER BY ... LIMIT 10, when asked for

The bracketed steps exist when the process group has more than one rank: they are the reference's hash-partitioned
exchange between the stages of a distributed join (PartitionedOutputOperator.java:411-431 routing rule,
LocalPartitionGenerator.java:45-65 for a power-of-two fan-out) done as one RCCL all-to-all per column
(`presto_amd.exchange`).  After the last exchange every orderkey lives on exactly one rank, so the grouped result of a
rank is final (disjoint groups; no PARTIAL/FINAL merge is needed), as in the reference's plan where the final
aggregation is partitioned on the group keys.

Every rank must run the same number of pipelines; page counts may differ per rank (ExchangeOperator keeps taking part
in the collectives until every rank has finished)."""
import torch
import torch.distributed as dist

from . import abi, tpch
from .exchange import DeviceOps, exchange_columns
from .expr import field
from .operators import (Driver, FilterAndProjectOperator, HashAggregationOperator, HashBuilderOperator, LookupJoinOperator,
                        LookupSourceFactory, TopNOperator)
from .page import Block, DeviceBuffer, Page

_TORCH_DTYPE = {abi.BIGINT: torch.int64, abi.INTEGER: torch.int32, abi.DATE: torch.int32, abi.DOUBLE: torch.float64,
                abi.BOOLEAN: torch.uint8}
_TYPESTR = {abi.BIGINT: "<i8", abi.INTEGER: "<i4", abi.DATE: "<i4", abi.DOUBLE: "<f8", abi.BOOLEAN: "|u1"}


class _DeviceArray:
    """__cuda_array_interface__ view of a flat device column, so torch can wrap it without a copy."""

    def __init__(self, ptr, n, typestr, owner):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}
        self._owner = owner


def tensor_of(block, device):
    """torch view of a flat, fixed-width, non-null PA_MEM_DEVICE block."""
    if block.encoding != abi.FLAT or block.type == abi.VARCHAR or block.nulls is not None:
        raise NotImplementedError("only flat fixed-width non-null columns are exchanged on device")
    n = block.position_count
    if n == 0:
        return torch.empty(0, dtype=_TORCH_DTYPE[block.type], device=device)
    owner = getattr(block.values, "owner", None)
    if isinstance(owner, torch.Tensor) and owner.numel() == n and owner.data_ptr() == block.values.ptr:
        return owner  # the column already is a torch tensor (pages made by page_of)
    return torch.as_tensor(_DeviceArray(block.values.ptr, n, _TYPESTR[block.type], block), device=device)


def page_of(tensors, types):
    n = int(tensors[0].shape[0]) if tensors else 0
    blocks = [Block(t, abi.FLAT, n, values=DeviceBuffer(c.data_ptr(), c.numel() * c.element_size(), c)) for c, t in zip(tensors, types)]
    return Page(blocks, n, abi.MEM_DEVICE)


class ExchangeOperator:
    """Operator-protocol wrapper of one hash-partitioned exchange step: every page added is partitioned on device by
    the hash of `hash_channels` and shuffled with one all-to-all per column; the rows this rank receives come out
    as one device page.  addInput and finish are collective: finish keeps answering the other ranks' rounds with
    empty pages until all ranks have finished."""

    def __init__(self, types, hash_channels, stream, group=None, ops=None, device=None):
        self.types = list(types)
        self.hash_channels = list(hash_channels)
        self.group = group
        self.ops = ops or DeviceOps()
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._stream = torch.cuda.ExternalStream(stream) if (stream and self.device.type == "cuda") else None
        self._queue = []
        self._alive = []          # received tensors of the pages handed out last (consumers may still read them)
        self._finishing = False
        self._finished = False
        self.rows_sent = 0
        self.rows_received = 0

    def _round(self, page):
        """One collective round; returns False when no rank had a page (all ranks are finishing)."""
        flag = torch.tensor([1 if page is not None else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, group=self.group)
        if int(flag.item()) == 0:
            return False
        if page is not None:
            columns = [tensor_of(b, self.device) for b in page.blocks]
        else:
            columns = [torch.empty(0, dtype=_TORCH_DTYPE[t], device=self.device) for t in self.types]
        received, counts = exchange_columns(self.ops, columns, self.types, self.hash_channels, group=self.group)
        self.rows_sent += int(columns[0].shape[0])
        self.rows_received += sum(counts)
        if sum(counts) > 0:
            self._queue.append(received)
        return True

    def _in_stream(self, fn, *args):
        if self._stream is None:
            return fn(*args)
        with torch.cuda.stream(self._stream):
            return fn(*args)

    def needsInput(self):
        return not self._finishing and not self._queue

    def addInput(self, page):
        if self._finishing:
            raise RuntimeError("addInput after finish")
        self._in_stream(self._round, page)

    def getOutput(self):
        if not self._queue:
            return None
        tensors = self._queue.pop(0)
        self._alive = (self._alive + [tensors])[-2:]
        return page_of(tensors, self.types)

    def finish(self):
        if self._finishing:
            return
        self._finishing = True
        while self._in_stream(self._round, None):
            pass
        self._finished = True

    def isFinished(self):
        return self._finished and not self._queue

    def close(self):
        self._queue, self._alive = [], []


AGG_TYPES = [abi.BIGINT, abi.DOUBLE, abi.DATE, abi.INTEGER]       # lineitem JOIN orders: orderkey, revenue, orderdate, shippriority
AGG_GROUP_BY = [0, 2, 3]
AGG_AGGREGATES = [(abi.AGG_SUM, 1, abi.DOUBLE), (abi.AGG_COUNT_STAR, -1, None)]
ORDERS_JOINED_TYPES = [abi.BIGINT, abi.DATE, abi.INTEGER]           # orders JOIN customer: orderkey, orderdate, shippriority
RESULT_TYPES = [abi.BIGINT, abi.DATE, abi.INTEGER, abi.DOUBLE, abi.BIGINT]  # orderkey, orderdate, shippriority, revenue, count


def run(customer_pages, orders_pages, lineitem_pages, stream, group=None, ops=None, device=None, expected_groups=100000,
        distributed=None, result_mem=abi.MEM_HOST, top_n=0, with_count=True, dynamic_filters=True):
    """Runs the three pipelines on this rank's pages; returns (result pages, counters).  `stream` is the HIP stream
    handle every operator (and the exchange) runs on; result_mem = where the grouped result is left (PA_MEM_DEVICE
    when a device operator consumes it).  top_n > 0 appends the query's TopN (revenue DESC, orderdate ASC): every rank then
    returns its own top_n rows -- the groups of different ranks are disjoint, so the query result is the top_n of their union.
    with_count=False runs the query as TPC-H states it (sum(revenue) only); the count(*) column is there for the parity tests.
    dynamic_filters: the filters upstream of the two probes also drop the rows whose join key matches no build key (the joins'
    dynamic filters, applied where Trino applies them); with exchange steps the filter is the OR of every rank's build-key
    bitmap, so that rows are dropped before they are exchanged."""
    if distributed is None:
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    dev = abi.MEM_DEVICE
    s = stream

    def exchange(types, channels):
        return [ExchangeOperator(types, channels, s, group, ops, device)] if distributed else []

    import time
    from ._lib import check, lib
    counters = {}

    def dynamic_filter(fp, channel, bridge, name):
        """Installs the join's dynamic filter in the FilterAndProject upstream of its probe.  With exchange steps the filter
        runs BEFORE the exchange, so it must know the build keys of every rank: the ranks agree on the key range, each sets the
        bits of its partition's keys, the bitmaps are all-gathered and OR-ed (RCCL has no bitwise reduction)."""
        if not dynamic_filters:
            return
        if not distributed:
            counters[name] = fp.setDynamicFilter(channel, bridge)
            return
        import torch
        dev_t = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        local = bridge.keyRange()
        big = (1 << 62)
        ends = torch.tensor([local[0] if local else big, -local[1] if local else big], dtype=torch.int64, device=dev_t)
        dist.all_reduce(ends, op=dist.ReduceOp.MIN, group=group)
        lo, hi = int(ends[0]), -int(ends[1])
        rows = torch.tensor([bridge.positionCount()], dtype=torch.int64, device=dev_t)
        dist.all_reduce(rows, group=group)
        key_range = hi - lo
        if lo == big or key_range < 0 or key_range >= 64 * max(int(rows[0]), 1) or key_range >= (1 << 36):
            counters[name] = False  # no keys anywhere, or too sparse for a bitmap to pay
            return
        words = (key_range >> 6) + 1
        world = dist.get_world_size(group)
        mine = torch.zeros(words, dtype=torch.int64, device=dev_t)
        torch.cuda.synchronize()
        if local:
            bridge.fillKeyBitmap(lo, key_range, mine.data_ptr(), s)
        check(lib().pa_stream_synchronize(s))
        gathered = torch.empty(world * words, dtype=torch.int64, device=dev_t)
        dist.all_gather_into_tensor(gathered, mine, group=group)
        bits = gathered.view(world, words)[0].clone()
        for r in range(1, world):
            bits |= gathered.view(world, words)[r]
        torch.cuda.synchronize()
        fp.setDynamicFilterBitmap(channel, bits.data_ptr(), lo, key_range, keep=bits)
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
        FilterAndProjectOperator(tpch.CUSTOMER_TYPES, tpch.q3_customer_filter(), [field(0, abi.BIGINT)], output_mem=dev, stream=s),
        *exchange([abi.BIGINT], [0]),
        HashBuilderOperator(b1, [abi.BIGINT], [0], [], stream=s)]).run()
    lap("customer_pipeline")
    # pipeline 2
    b2 = LookupSourceFactory()
    orders_fp = FilterAndProjectOperator(tpch.ORDERS_TYPES, tpch.q3_orders_filter(), [field(i, t) for i, t in enumerate(tpch.ORDERS_TYPES)],
                                         output_mem=dev, stream=s)
    dynamic_filter(orders_fp, 1, b1, "orders_dynamic_filter")
    Driver(orders_pages, [
        orders_fp,
        *exchange(tpch.ORDERS_TYPES, [1]),
        LookupJoinOperator(b1, tpch.ORDERS_TYPES, [1], [0, 2, 3], output_mem=dev, stream=s),
        *exchange(ORDERS_JOINED_TYPES, [0]),
        HashBuilderOperator(b2, ORDERS_JOINED_TYPES, [0], [1, 2], stream=s)]).run()
    lap("orders_pipeline")
    # pipeline 3
    # orderkey is unique on the build side, so its row count bounds the groups (what the planner's stats estimate)
    expected_groups = max(expected_groups, min(b2.positionCount(), 1 << 28))
    aggregates = AGG_AGGREGATES if with_count else AGG_AGGREGATES[:1]
    result_types = RESULT_TYPES if with_count else RESULT_TYPES[:4]
    agg = HashAggregationOperator(AGG_TYPES, AGG_GROUP_BY, aggregates, expected_groups=expected_groups,
                                  output_mem=dev if top_n else result_mem, stream=s)
    lineitem_fp = FilterAndProjectOperator(tpch.Q3_LINEITEM_TYPES, tpch.q3_lineitem_filter(), tpch.q3_lineitem_projections(), output_mem=dev, stream=s)
    dynamic_filter(lineitem_fp, 0, b2, "lineitem_dynamic_filter")
    out = Driver(lineitem_pages, [
        lineitem_fp,
        *exchange([abi.BIGINT, abi.DOUBLE], [0]),
        LookupJoinOperator(b2, [abi.BIGINT, abi.DOUBLE], [0], [0, 1], output_mem=dev, stream=s),
        agg,
        *([TopNOperator(result_types, top_n, [3, 1], [abi.DESC_NULLS_LAST, abi.ASC_NULLS_LAST], output_mem=result_mem, stream=s)] if top_n else [])]).run()
    lap("lineitem_pipeline")
    counters["build1_rows"] = b1.positionCount()
    counters["build2_rows"] = b2.positionCount()
    return out, counters
