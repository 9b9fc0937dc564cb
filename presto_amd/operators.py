"""Host-side mirror of the reference's Operator / OperatorFactory interfaces over the C ABI.

Mirrors io.trino.operator.Operator (core/trino-main/src/main/java/io/trino/operator/Operator.java:21-103)
with the same method names and call protocol (needsInput / addInput / getOutput / finish / isFinished /
close), so that the parity tests read like the reference's operator tests
(core/trino-main/src/test/java/io/trino/operator/OperatorAssertion.java:62-138).  Each class is what the
Java `GpuOperator implements Operator` of INTEGRATION.md would be: a handle plus one native call per method.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import DeviceAllocation, check, lib
from .expr import serialize, serialize_many
from .page import Block, DeviceBuffer, Page, page_from_c


class Operator:
    """Operator.java:21-103"""

    def __init__(self, handle, keep):
        self._h = handle
        self._keep = keep

    def needsInput(self):
        return bool(check(lib().pa_op_needs_input(self._h)))

    def addInput(self, page):
        cpage, keep = page.to_c()
        L = lib()
        # Driver.java:355-457 only hands a page over when needsInput(); while the operator reports isBlocked (device work in
        # flight, its queue full) the Driver would yield and poll -- so does this
        while not L.pa_op_needs_input(self._h) and L.pa_op_is_blocked(self._h) > 0:
            pass
        page.retain_until_released()   # (PA_PAGE_RETAINED pages: kept until the native side calls their release)
        check(L.pa_op_add_input(self._h, C.byref(cpage)))
        self._last_input = keep  # Pages may be retained by the operator until its work is done
        if page.stable:  # PA_PAGE_STABLE: the operator may still read the page until it is closed
            if not hasattr(self, "_stable_inputs"):
                self._stable_inputs = []
            self._stable_inputs.append(page)

    def getOutput(self):
        """Returns a host Page (copy) for PA_MEM_HOST operators, a device Page view otherwise, or None."""
        out = abi.pa_page()
        if not check(lib().pa_op_get_output(self._h, C.byref(out))):
            return None
        if out.mem == abi.MEM_HOST:
            return page_from_c(out)
        return device_page_from_c(out, owner=self)

    def finish(self):
        check(lib().pa_op_finish(self._h))

    def isFinished(self):
        return bool(check(lib().pa_op_is_finished(self._h)))

    def isBlocked(self):
        return bool(check(lib().pa_op_is_blocked(self._h)))

    def memoryBytes(self):
        return lib().pa_op_memory_bytes(self._h)

    def kernelTime(self):
        """(total ms, launches) of the operator's dominant kernel, from HIP events on its stream."""
        ms = C.c_double()
        n = C.c_int64()
        check(lib().pa_op_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernelName(self):
        """Name of that kernel as a kernel trace shows it (pa_fused_<tier>_<key8>); '' before its first launch."""
        buf = C.create_string_buffer(128)
        check(lib().pa_op_kernel_name(self._h, buf, 128))
        return buf.value.decode()

    def selectedPositions(self):
        """FilterAndProject only: SelectedPositions of the last page -> (is_list, ndarray | count)."""
        ptr, count, is_list = C.c_void_p(), C.c_int32(), C.c_int32()
        check(lib().pa_filter_project_selected_positions(self._h, C.byref(ptr), C.byref(count), C.byref(is_list)))
        if not is_list.value:
            return False, count.value
        return True, download(DeviceBuffer(ptr.value, 4 * count.value), np.int32, count.value)

    def matchPairs(self):
        """LookupJoin only: (probe positions, build positions) of the last probe page in emission order."""
        p, b, count = C.c_void_p(), C.c_void_p(), C.c_int32()
        check(lib().pa_lookup_join_match_pairs(self._h, C.byref(p), C.byref(b), C.byref(count)))
        n = count.value
        if n == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int32)
        return (download(DeviceBuffer(p.value, 4 * n), np.int32, n), download(DeviceBuffer(b.value, 4 * n), np.int32, n))

    def setDynamicFilter(self, channel, lookup_source_factory):
        """FilterAndProject only, before the first page: drop the rows whose `channel` value matches no build key of the (built)
        join bridge -- the join's dynamic filter applied upstream of the probe.  True when the filter is active."""
        return bool(check(lib().pa_filter_project_set_dynamic_filter(self._h, channel, lookup_source_factory._h)))

    def setOutputTopNHint(self, n, sort_channels, sort_orders):
        """(Hash)Aggregation operators whose only consumer is TopNOperator(n, sort_channels, sort_orders) over their output: groups
        that cannot be among its n best rows may be left out (pa_aggregation_set_output_topn_hint).  True when the hint is taken."""
        ch, od = abi.int32_array(sort_channels), abi.int32_array(sort_orders)
        return bool(check(lib().pa_aggregation_set_output_topn_hint(self._h, n, len(sort_channels), C.cast(ch, C.POINTER(C.c_int32)),
                                                                    C.cast(od, C.POINTER(C.c_int32)))))

    def setDynamicFilterBitmap(self, channel, bits_ptr, min_key, key_range, keep=None):
        """The same with a bitmap combined over the ranks of a partitioned join (`keep` = whatever owns the bitmap)."""
        check(lib().pa_filter_project_set_dynamic_filter_bitmap(self._h, channel, bits_ptr, min_key, key_range))
        self._dyn_keep = keep

    def close(self):
        if self._h:
            lib().pa_op_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_page_from_c(cpage, owner=None):
    blocks = []
    n = cpage.position_count
    for i in range(cpage.channel_count):
        col = cpage.columns[i]
        w = abi.TYPE_WIDTH[col.type]
        nulls = DeviceBuffer(col.nulls, n, owner) if col.nulls else None
        if col.encoding == abi.ROW_FIELDS:
            sub = abi.pa_page()
            sub.position_count, sub.channel_count, sub.columns, sub.mem = n, col.dictionary_size, col.dictionary, abi.MEM_DEVICE
            blocks.append(Block(abi.ROW, abi.ROW_FIELDS, n, nulls=nulls, fields=device_page_from_c(sub, owner).blocks))
            continue
        if col.encoding == abi.VARWIDTH:
            blocks.append(Block(col.type, abi.VARWIDTH, n, values=DeviceBuffer(col.values, 0, owner),
                                offsets=DeviceBuffer(col.offsets, 4 * (n + 1), owner), nulls=nulls))
        else:
            blocks.append(Block(col.type, abi.FLAT, n, values=DeviceBuffer(col.values, w * n, owner), nulls=nulls))
    page = Page(blocks, n, abi.MEM_DEVICE)
    if cpage.flags & abi.PAGE_RETAINED and cpage.release:
        # a page that was handed over (pa_filter_project_desc.output_handover): its release travels with it to the operator that takes it
        page.native_release = (cpage.release, cpage.release_ctx)
    return page


# ---- descriptors -----------------------------------------------------------------------------------
def _params_of(input_types, type_params):
    """DECIMAL channels carry PA_DECIMAL_PARAM(precision, scale) as their type parameter (abi.DecimalType knows it)."""
    if type_params is None and any(isinstance(t, abi.DecimalType) for t in input_types):
        return [abi.type_param(t) for t in input_types]
    return type_params


def _filter_project_desc(input_types, filter_expr, projections, output_mem, stream, type_params=None, output_handover=False):
    keep = []
    type_params = _params_of(input_types, type_params)
    d = abi.pa_filter_project_desc()
    types = abi.int32_array(input_types)
    d.input_channel_count = len(input_types)
    d.input_types = C.cast(types, C.POINTER(C.c_int32))
    keep.append(types)
    if type_params is not None:
        tp = abi.int32_array(type_params)
        d.input_type_params = C.cast(tp, C.POINTER(C.c_int32))
        keep.append(tp)
    if filter_expr is not None:
        f, kf = serialize(filter_expr)
        keep += [f, kf]
        d.filter = C.pointer(f)
    arr, kp = serialize_many(projections)
    keep += [arr, kp]
    d.projection_count = len(projections)
    d.projections = C.cast(arr, C.POINTER(abi.pa_expr))
    d.output_mem = output_mem
    d.stream = stream
    d.output_handover = 1 if output_handover else 0
    return d, keep


def _aggregates(aggregates):
    arr = (abi.pa_aggregate * max(len(aggregates), 1))()
    for i, a in enumerate(aggregates):
        arr[i].fn = a[0]
        arr[i].input_channel = a[1]
        arr[i].input_type = a[2] if a[2] is not None else 0
        arr[i].mask_channel = a[3] if len(a) > 3 else -1
    return arr


def _hash_agg_desc(input_types, group_by_channels, aggregates, hash_channel, expected_groups, output_mem, stream,
                   type_params=None, step=abi.STEP_SINGLE, max_partial_memory=0, state_format=abi.STATES_FLAT,
                   global_aggregation_group_ids=None, group_id_channel=None, produce_default_output=False):
    keep = []
    type_params = _params_of(input_types, type_params)
    d = abi.pa_hash_aggregation_desc()
    types = abi.int32_array(input_types)
    gb = abi.int32_array(group_by_channels)
    aggs = _aggregates(aggregates)
    d.input_channel_count = len(input_types)
    d.input_types = C.cast(types, C.POINTER(C.c_int32))
    if type_params is not None:
        tp = abi.int32_array(type_params)
        d.input_type_params = C.cast(tp, C.POINTER(C.c_int32))
        keep.append(tp)
    d.group_by_count = len(group_by_channels)
    d.group_by_channels = C.cast(gb, C.POINTER(C.c_int32))
    d.hash_channel = hash_channel
    d.step = step
    d.aggregate_count = len(aggregates)
    d.aggregates = C.cast(aggs, C.POINTER(abi.pa_aggregate))
    d.expected_groups = expected_groups
    d.output_mem = output_mem
    d.stream = stream
    d.max_partial_memory = int(max_partial_memory)
    d.state_format = state_format
    d.produce_default_output = 1 if produce_default_output else 0
    d.group_id_channel = -1 if group_id_channel is None else group_id_channel
    if global_aggregation_group_ids:
        ids = abi.int32_array(global_aggregation_group_ids)
        d.global_aggregation_group_id_count = len(global_aggregation_group_ids)
        d.global_aggregation_group_ids = C.cast(ids, C.POINTER(C.c_int32))
        keep.append(ids)
    keep += [types, gb, aggs]
    return d, keep


def fused_aggregation_desc(input_types, filter_expr, projections, group_by_channels, aggregates, hash_channel=-1,
                           expected_groups=10000, output_mem=abi.MEM_HOST, stream=None, type_params=None, step=abi.STEP_SINGLE,
                           max_partial_memory=0, state_format=abi.STATES_FLAT):
    """aggregates: list of (fn, projection index, input type[, mask projection index])."""
    fp, k1 = _filter_project_desc(input_types, filter_expr, projections, output_mem, stream, type_params)
    ag, k2 = _hash_agg_desc([p.type for p in projections], group_by_channels, aggregates, hash_channel, expected_groups,
                            output_mem, stream, None, step, max_partial_memory, state_format)
    d = abi.pa_fused_aggregation_desc()
    d.filter_project = fp
    d.aggregation = ag
    return d, [k1, k2, fp, ag]


# ---- factories (OperatorFactory.createOperator) ----------------------------------------------------------
def FilterAndProjectOperator(input_types, filter_expr, projections, output_mem=abi.MEM_HOST, stream=None,
                             type_params=None, min_output_page_size=0, min_output_page_row_count=0, max_output_page_size=0):
    """FilterAndProjectOperator.createOperatorFactory (…/operator/FilterAndProjectOperator.java:73-178) with
    PageProcessor(Optional<PageFilter>, List<PageProjection>); minOutputPageSize / minOutputPageRowCount configure
    the MergePages step behind it (0, 0 = every page passes through, as in the reference's operator tests)."""
    return FilterAndProjectOperatorFactory(input_types, filter_expr, projections, output_mem, stream, type_params, min_output_page_size,
                                           min_output_page_row_count, max_output_page_size).createOperator()


def FilterAndProjectOperatorFactory(input_types, filter_expr, projections, output_mem=abi.MEM_HOST, stream=None,
                                    type_params=None, min_output_page_size=0, min_output_page_row_count=0, max_output_page_size=0,
                                    output_handover=False):
    """output_handover: device output pages leave with their buffers (PA_PAGE_RETAINED, released by the operator that takes them) --
    what feeds a HashBuilder, which then reads the build side in place."""
    d, keep = _filter_project_desc(input_types, filter_expr, projections, output_mem, stream, type_params, output_handover)
    d.min_output_page_bytes = int(min_output_page_size)
    d.min_output_page_rows = int(min_output_page_row_count)
    d.max_output_page_bytes = int(max_output_page_size)
    return OperatorFactory(lib().pa_filter_project_create, d, keep)


class LazyBlock:
    """LazyBlock (core/trino-spi/src/main/java/io/trino/spi/block/LazyBlock.java): a block whose loader runs on first use."""

    def __init__(self, position_count, loader):
        self.position_count = position_count
        self.loader = loader
        self.loaded = None

    def getLoadedBlock(self):
        if self.loaded is None:
            self.loaded = self.loader()
            assert self.loaded.position_count == self.position_count
        return self.loaded


class ScanFilterAndProjectOperator(Operator):
    """ScanFilterAndProjectOperator (…/operator/ScanFilterAndProjectOperator.java:67-114): a source operator over a
    ConnectorPageSource -- here any object with getNextPage() -> Page | None (None = finished; blocks may be LazyBlocks) and
    an optional close().  The native operator pulls pages through pa_page_source and loads lazy blocks by need."""

    def __init__(self, page_source, input_types, filter_expr, projections, output_mem=abi.MEM_HOST, stream=None, type_params=None,
                 min_output_page_size=0, min_output_page_row_count=0):
        d, keep = _filter_project_desc(input_types, filter_expr, projections, output_mem, stream, type_params)
        d.min_output_page_bytes = int(min_output_page_size)
        d.min_output_page_rows = int(min_output_page_row_count)
        self._source = page_source
        self._types = list(input_types)
        self._current = None     # (page, c columns array, keepalive)
        self.errors = []

        def next_page(ctx, out):
            try:
                page = page_source.getNextPage()
                if page is None:
                    return 0
                cols = (abi.pa_column * max(len(page.blocks), 1))()
                keepalive = []
                for i, b in enumerate(page.blocks):
                    if isinstance(b, LazyBlock) and b.loaded is None:
                        cols[i].type = self._types[i]
                        cols[i].encoding = abi.VARWIDTH if self._types[i] == abi.VARCHAR else abi.FLAT
                    else:
                        (b.loaded if isinstance(b, LazyBlock) else b).fill_c(cols[i], keepalive)
                out[0].position_count = page.position_count
                out[0].channel_count = len(page.blocks)
                out[0].columns = C.cast(cols, C.POINTER(abi.pa_column))
                out[0].mem = page.mem
                out[0].flags = 0
                self._current = (page, cols, keepalive)
                return 1
            except Exception as e:  # never let an exception cross the C boundary
                self.errors.append(e)
                return abi.ERR_DEVICE

        def load_block(ctx, channel, out):
            try:
                page, cols, keepalive = self._current
                block = page.blocks[channel].getLoadedBlock()
                block.fill_c(out[0], keepalive)
                return 0
            except Exception as e:
                self.errors.append(e)
                return abi.ERR_DEVICE

        def close(ctx):
            if hasattr(page_source, "close"):
                page_source.close()

        src = abi.pa_page_source()
        src.ctx = None
        src.next_page = abi.NEXT_PAGE(next_page)
        src.load_block = abi.LOAD_BLOCK(load_block)
        src.close = abi.CLOSE_SOURCE(close)
        h = C.c_void_p()
        check(lib().pa_scan_filter_project_create(C.byref(d), C.byref(src), C.byref(h)))
        super().__init__(h, [keep, src])

    def getOutput(self):
        try:
            return super().getOutput()
        except Exception:
            if self.errors:
                raise self.errors[-1]   # the page source's own exception, as the reference would surface it
            raise

    def stats(self):
        """(positions pulled, bytes of the blocks loaded, blocks loaded, projection blocks left unloaded)"""
        v = [C.c_int64() for _ in range(4)]
        check(lib().pa_scan_stats(self._h, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)


def AggregationOperator(input_types, aggregates, output_mem=abi.MEM_HOST, stream=None, step=abi.STEP_SINGLE, state_format=abi.STATES_FLAT):
    """AggregationOperator.AggregationOperatorFactory (…/operator/AggregationOperator.java:40-95)."""
    d = abi.pa_aggregation_desc()
    types = abi.int32_array(input_types)
    aggs = _aggregates(aggregates)
    d.input_channel_count = len(input_types)
    d.input_types = C.cast(types, C.POINTER(C.c_int32))
    d.aggregate_count = len(aggregates)
    d.aggregates = C.cast(aggs, C.POINTER(abi.pa_aggregate))
    d.step = step
    d.output_mem = output_mem
    d.stream = stream
    d.state_format = state_format
    h = C.c_void_p()
    check(lib().pa_aggregation_create(C.byref(d), C.byref(h)))
    return Operator(h, [types, aggs])


def HashAggregationOperator(input_types, group_by_channels, aggregates, hash_channel=-1, expected_groups=10000,
                            output_mem=abi.MEM_HOST, stream=None, type_params=None, step=abi.STEP_SINGLE, max_partial_memory=0,
                            state_format=abi.STATES_FLAT, global_aggregation_group_ids=None, group_id_channel=None, produce_default_output=False):
    """HashAggregationOperatorFactory (…/operator/HashAggregationOperator.java:120-202); max_partial_memory = maxPartialMemory
    in bytes (Step.PARTIAL: flush when the aggregation is "full"), state_format = abi.STATES_FLAT / STATES_REFERENCE;
    global_aggregation_group_ids / group_id_channel (an index among the group-by columns) / produce_default_output: the default rows
    of the global grouping sets when no page arrives (:486-492, 545-587)."""
    d, keep = _hash_agg_desc(input_types, group_by_channels, aggregates, hash_channel, expected_groups, output_mem, stream,
                             type_params, step, max_partial_memory, state_format, global_aggregation_group_ids, group_id_channel,
                             produce_default_output)
    h = C.c_void_p()
    check(lib().pa_hash_aggregation_create(C.byref(d), C.byref(h)))
    return Operator(h, keep)


class OperatorFactory:
    """OperatorFactory (core/trino-main/src/main/java/io/trino/operator/OperatorFactory.java:18-50): built once per plan
    node by the planner -- the descriptor (types, serialised RowExpressions, aggregates) is made here, once --
    createOperator() is then called per Driver and only hands the descriptor to the native factory."""

    def __init__(self, create, desc, keep):
        self._create, self._desc, self._keep = create, desc, keep

    def createOperator(self):
        h = C.c_void_p()
        check(self._create(C.byref(self._desc), C.byref(h)))
        return Operator(h, self._keep)


class JoinOperatorFactory(OperatorFactory):
    """The factories of a join's operators (HashBuilderOperatorFactory, LookupJoinOperatorFactory and the fused forms) share a
    JoinBridge; here the LookupSourceFactory of the Driver's query comes with createOperator."""

    def createOperator(self, bridge):
        h = C.c_void_p()
        check(self._create(C.byref(self._desc), bridge._h, C.byref(h)))
        return Operator(h, [self._keep, bridge])


def FusedAggregationOperatorFactory(input_types, filter_expr, projections, group_by_channels, aggregates, **kw):
    d, keep = fused_aggregation_desc(input_types, filter_expr, projections, group_by_channels, aggregates, **kw)
    return OperatorFactory(lib().pa_fused_aggregation_create, d, keep)


def FusedAggregationOperator(input_types, filter_expr, projections, group_by_channels, aggregates, **kw):
    """[Scan]FilterAndProject -> (Hash)Aggregation collapsed into one device pass."""
    return FusedAggregationOperatorFactory(input_types, filter_expr, projections, group_by_channels, aggregates, **kw).createOperator()


def TopNOperator(input_types, n, sort_channels, sort_orders, output_mem=abi.MEM_HOST, stream=None):
    """TopNOperator.createOperatorFactory (…/operator/TopNOperator.java:43-90)."""
    return TopNOperatorFactory(input_types, n, sort_channels, sort_orders, output_mem, stream).createOperator()


def TopNOperatorFactory(input_types, n, sort_channels, sort_orders, output_mem=abi.MEM_HOST, stream=None):
    d = abi.pa_topn_desc()
    types = abi.int32_array(input_types)
    sc = abi.int32_array(sort_channels)
    so = abi.int32_array(sort_orders)
    d.input_channel_count = len(input_types)
    d.input_types = C.cast(types, C.POINTER(C.c_int32))
    d.n = n
    d.sort_channel_count = len(sort_channels)
    d.sort_channels = C.cast(sc, C.POINTER(C.c_int32))
    d.sort_orders = C.cast(so, C.POINTER(C.c_int32))
    d.output_mem = output_mem
    d.stream = stream
    return OperatorFactory(lib().pa_topn_create, d, [types, sc, so])


def OrderByOperator(input_types, output_channels, sort_channels, sort_orders, output_mem=abi.MEM_HOST, stream=None):
    """OrderByOperator.OrderByOperatorFactory (…/operator/OrderByOperator.java:45-120)."""
    d = abi.pa_order_by_desc()
    types = abi.int32_array(input_types)
    oc = abi.int32_array(output_channels)
    sc = abi.int32_array(sort_channels)
    so = abi.int32_array(sort_orders)
    d.input_channel_count = len(input_types)
    d.input_types = C.cast(types, C.POINTER(C.c_int32))
    d.output_channel_count = len(output_channels)
    d.output_channels = C.cast(oc, C.POINTER(C.c_int32))
    d.sort_channel_count = len(sort_channels)
    d.sort_channels = C.cast(sc, C.POINTER(C.c_int32))
    d.sort_orders = C.cast(so, C.POINTER(C.c_int32))
    d.output_mem = output_mem
    d.stream = stream
    h = C.c_void_p()
    check(lib().pa_order_by_create(C.byref(d), C.byref(h)))
    return Operator(h, [types, oc, sc, so])


class DynamicFilterSourceOperator(Operator):
    """DynamicFilterSourceOperator.DynamicFilterSourceOperatorFactory (…/operator/DynamicFilterSourceOperator.java:74-139):
    pass-through on the build side of a join that collects the build values of `filter_channels`."""

    def __init__(self, input_types, filter_channels, max_distinct_values, max_filter_size_bytes, min_max_collection_limit, stream=None):
        d = abi.pa_dynamic_filter_source_desc()
        types = abi.int32_array(input_types)
        fc = abi.int32_array(filter_channels)
        d.input_channel_count = len(input_types)
        d.input_types = C.cast(types, C.POINTER(C.c_int32))
        d.filter_channel_count = len(filter_channels)
        d.filter_channels = C.cast(fc, C.POINTER(C.c_int32))
        d.max_distinct_values = max_distinct_values
        d.min_max_collection_limit = min_max_collection_limit
        d.max_filter_size_bytes = max_filter_size_bytes
        d.stream = stream
        h = C.c_void_p()
        check(lib().pa_dynamic_filter_source_create(C.byref(d), C.byref(h)))
        super().__init__(h, [types, fc])
        self._filters = len(filter_channels)

    def getOutput(self):
        """The page handed to addInput (the reference returns the same Page object)."""
        out = abi.pa_page()
        if not check(lib().pa_op_get_output(self._h, C.byref(out))):
            return None
        return self._last_page

    def addInput(self, page):
        super().addInput(page)
        self._last_page = page

    def predicate(self):
        """What dynamicPredicateConsumer received: None while it has not been called, "all" for TupleDomain.all(), else one
        entry per filter channel: ("all",) | ("none",) | ("values", [..ascending..]) | ("range", low, high)."""
        is_all = C.c_int32()
        doms = (abi.pa_domain * self._filters)()
        if not check(lib().pa_dynamic_filter_poll(self._h, C.byref(is_all), doms, self._filters)):
            return None
        if is_all.value:
            return "all"
        out = []
        for d in doms:
            if d.kind == abi.DOMAIN_ALL:
                out.append(("all",))
                continue
            if d.kind == abi.DOMAIN_NONE:
                out.append(("none",))
                continue
            page = abi.pa_page()
            page.position_count = d.value_count
            page.channel_count = 1
            page.columns = C.pointer(d.values)
            page.mem = abi.MEM_HOST
            vals = [r[0] for r in page_from_c(page).to_rows()]
            out.append(("values", vals) if d.kind == abi.DOMAIN_VALUES else ("range", vals[0], vals[1]))
        return out


class LookupSourceFactory:
    """JoinBridge between a HashBuilderOperator and its LookupJoinOperators
    (…/operator/join/PartitionedLookupSourceFactory.java, JoinBridgeManager.java)."""

    def __init__(self):
        h = C.c_void_p()
        check(lib().pa_lookup_source_create(C.byref(h)))
        self._h = h

    def tables(self):
        """(key[], positionLinks[]) of the built PagesHash."""
        key, links, hs, n = C.c_void_p(), C.c_void_p(), C.c_int32(), C.c_int32()
        check(lib().pa_lookup_source_tables(self._h, C.byref(key), C.byref(hs), C.byref(links), C.byref(n)))
        return (download(DeviceBuffer(key.value, 4 * hs.value), np.int32, hs.value),
                download(DeviceBuffer(links.value, 4 * n.value), np.int32, n.value))

    def keyRange(self):
        """(min, max) of this rank's build keys, or None (no single integer join key / no non-NULL key)."""
        lo, hi = C.c_int64(), C.c_int64()
        if not check(lib().pa_lookup_source_key_range(self._h, C.byref(lo), C.byref(hi))):
            return None
        return lo.value, hi.value

    def fillKeyBitmap(self, min_key, key_range, bits_ptr, stream=None):
        """Sets bit (key - min_key) of the device bitmap at bits_ptr ((key_range >> 6) + 1 words, cleared first) for every build key."""
        check(lib().pa_lookup_source_key_bitmap(self._h, min_key, key_range, bits_ptr, stream))

    def sharedKeyBitmap(self, comm, partitioned_by_key=True, stream=None):
        """Collective over the ranks of `comm`: the existence bitmap of every rank's build keys over their union key range ->
        (device pointer, min key, range), or None when no filter is possible (pa_lookup_source_shared_key_bitmap)."""
        bits, lo, rng = C.c_void_p(), C.c_int64(), C.c_uint64()
        if not check(lib().pa_lookup_source_shared_key_bitmap(self._h, comm._h, 1 if partitioned_by_key else 0, stream, C.byref(bits),
                                                              C.byref(lo), C.byref(rng))):
            return None
        return bits.value, lo.value, rng.value

    def positionCount(self):
        """Build positions of the published lookup source (LookupSource.getJoinPositionCount)."""
        return check(lib().pa_lookup_source_position_count(self._h))

    def destroy(self):
        if self._h:
            lib().pa_lookup_source_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def hash_builder_desc(input_types, join_channels, output_channels, hash_channel=-1, expected_positions=0, stream=None):
    d = abi.pa_hash_builder_desc()
    types = abi.int32_array(input_types)
    jc = abi.int32_array(join_channels)
    oc = abi.int32_array(output_channels)
    d.input_channel_count = len(input_types)
    d.input_types = C.cast(types, C.POINTER(C.c_int32))
    d.join_channel_count = len(join_channels)
    d.join_channels = C.cast(jc, C.POINTER(C.c_int32))
    d.hash_channel = hash_channel
    d.output_channel_count = len(output_channels)
    d.output_channels = C.cast(oc, C.POINTER(C.c_int32))
    d.expected_positions = expected_positions
    d.stream = stream
    return d, [types, jc, oc]


def HashBuilderOperator(bridge, input_types, join_channels, output_channels, hash_channel=-1, expected_positions=0,
                        stream=None):
    """HashBuilderOperator.HashBuilderOperatorFactory (…/operator/join/HashBuilderOperator.java:56-180)."""
    return HashBuilderOperatorFactory(input_types, join_channels, output_channels, hash_channel, expected_positions, stream).createOperator(bridge)


def HashBuilderOperatorFactory(input_types, join_channels, output_channels, hash_channel=-1, expected_positions=0, stream=None):
    d, keep = hash_builder_desc(input_types, join_channels, output_channels, hash_channel, expected_positions, stream)
    return JoinOperatorFactory(lib().pa_hash_builder_create, d, keep)


def FusedJoinOperator(bridge, input_types, filter_expr, projections, probe_join_channels, probe_output_channels, output_mem=abi.MEM_HOST,
                      stream=None, type_params=None):
    """[Scan]FilterAndProject -> LookupJoin (INNER) behind one handle (pa_fused_join_create): filter, probe and the join's output page
    [probe output channels, build output channels] in two passes over the page when the lookup source has a single integer key
    without duplicates; the two device operators otherwise."""
    return FusedJoinOperatorFactory(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, output_mem, stream,
                                    type_params).createOperator(bridge)


def FusedJoinOperatorFactory(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, output_mem=abi.MEM_HOST,
                             stream=None, type_params=None, output_handover=False):
    d, keep = fused_join_desc(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, output_mem, stream, type_params,
                              output_handover)
    return JoinOperatorFactory(lib().pa_fused_join_create, d, keep)


def fused_join_desc(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, output_mem=abi.MEM_HOST, stream=None,
                    type_params=None, output_handover=False):
    d = abi.pa_fused_join_desc()
    fp, k1 = _filter_project_desc(input_types, filter_expr, projections, abi.MEM_DEVICE, stream, type_params, output_handover)
    jd, k2 = _lookup_join_desc([p.type for p in projections], probe_join_channels, probe_output_channels, -1, output_mem, stream, abi.JOIN_INNER)
    d.filter_project, d.join = fp, jd
    return d, [k1, k2]


def fused_join_aggregation_desc(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, joined_types,
                                group_by_channels, aggregates, hash_channel=-1, expected_groups=10000, output_mem=abi.MEM_HOST, stream=None,
                                type_params=None, step=abi.STEP_SINGLE):
    """pa_fused_join_aggregation_desc: FilterAndProject(input_types; filter, projections) -> LookupJoin(probe page = the
    projections; probe_join_channels / probe_output_channels index them) -> (Hash)Aggregation over the join's output page
    (joined_types = [probe outputs, build outputs]; group_by_channels / aggregates index it)."""
    d = abi.pa_fused_join_aggregation_desc()
    fp, k1 = _filter_project_desc(input_types, filter_expr, projections, abi.MEM_DEVICE, stream, type_params)
    jd, k2 = _lookup_join_desc([p.type for p in projections], probe_join_channels, probe_output_channels, -1, abi.MEM_DEVICE, stream, abi.JOIN_INNER)
    ag, k3 = _hash_agg_desc(joined_types, group_by_channels, aggregates, hash_channel, expected_groups, output_mem, stream, None, step)
    d.filter_project, d.join, d.aggregation = fp, jd, ag
    return d, [k1, k2, k3]


def FusedJoinAggregationOperator(bridge, input_types, filter_expr, projections, probe_join_channels, probe_output_channels, joined_types,
                                 group_by_channels, aggregates, **kw):
    """[Scan]FilterAndProject -> LookupJoin (INNER) -> (Hash)Aggregation behind one handle (pa_fused_join_aggregation_create): one
    generated kernel when the lookup source has a single integer key without duplicates, the three device operators otherwise."""
    return FusedJoinAggregationOperatorFactory(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, joined_types,
                                               group_by_channels, aggregates, **kw).createOperator(bridge)


def FusedJoinAggregationOperatorFactory(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, joined_types,
                                        group_by_channels, aggregates, **kw):
    d, keep = fused_join_aggregation_desc(input_types, filter_expr, projections, probe_join_channels, probe_output_channels, joined_types,
                                          group_by_channels, aggregates, **kw)
    return JoinOperatorFactory(lib().pa_fused_join_aggregation_create, d, keep)


def _lookup_join_desc(probe_types, probe_join_channels, probe_output_channels, probe_hash_channel, output_mem, stream, join_type):
    d = abi.pa_lookup_join_desc()
    types = abi.int32_array(probe_types)
    jc = abi.int32_array(probe_join_channels)
    oc = abi.int32_array(probe_output_channels)
    d.probe_channel_count = len(probe_types)
    d.probe_types = C.cast(types, C.POINTER(C.c_int32))
    d.join_channel_count = len(probe_join_channels)
    d.probe_join_channels = C.cast(jc, C.POINTER(C.c_int32))
    d.probe_hash_channel = probe_hash_channel
    d.probe_output_channel_count = len(probe_output_channels)
    d.probe_output_channels = C.cast(oc, C.POINTER(C.c_int32))
    d.output_mem = output_mem
    d.stream = stream
    d.join_type = join_type
    return d, [types, jc, oc]


def LookupOuterOperator(bridge, probe_types, probe_output_channels, join_type=abi.JOIN_LOOKUP_OUTER, output_mem=abi.MEM_HOST, stream=None):
    """LookupOuterOperator (…/operator/join/LookupOuterOperator.java:40-215): created by OperatorFactories.lookupOuterJoin /
    fullOuterJoin next to the probe operators; pull from it once they are finished."""
    d, keep = _lookup_join_desc(probe_types, [], probe_output_channels, -1, output_mem, stream, join_type)
    h = C.c_void_p()
    check(lib().pa_lookup_outer_create(C.byref(d), bridge._h, C.byref(h)))
    return Operator(h, keep + [bridge])


def LookupJoinOperator(bridge, probe_types, probe_join_channels, probe_output_channels, probe_hash_channel=-1,
                       output_mem=abi.MEM_HOST, stream=None, join_type=abi.JOIN_INNER, output_single_match=False, filter=None):
    """OperatorFactories.innerJoin / probeOuterJoin / lookupOuterJoin / fullOuterJoin (…/operator/OperatorFactories.java:27-84)
    -> LookupJoinOperator; join_type = abi.JOIN_*.  filter = the JoinFilterFunction as an expression over [build page channels,
    probe page channels] (JoinFilterFunctionCompiler's numbering), or None."""
    d = abi.pa_lookup_join_desc()
    types = abi.int32_array(probe_types)
    jc = abi.int32_array(probe_join_channels)
    oc = abi.int32_array(probe_output_channels)
    d.probe_channel_count = len(probe_types)
    d.probe_types = C.cast(types, C.POINTER(C.c_int32))
    d.join_channel_count = len(probe_join_channels)
    d.probe_join_channels = C.cast(jc, C.POINTER(C.c_int32))
    d.probe_hash_channel = probe_hash_channel
    d.probe_output_channel_count = len(probe_output_channels)
    d.probe_output_channels = C.cast(oc, C.POINTER(C.c_int32))
    d.output_mem = output_mem
    d.stream = stream
    d.join_type = join_type
    d.output_single_match = 1 if output_single_match else 0
    keep = [types, jc, oc, bridge]
    if filter is not None:
        f, kf = serialize(filter)
        d.filter = C.pointer(f)
        keep += [f, kf]
    h = C.c_void_p()
    check(lib().pa_lookup_join_create(C.byref(d), bridge._h, C.byref(h)))
    return Operator(h, keep)


# ---- driver loop ---------------------------------------------------------------------------------------
def to_pages(operator, input_pages):
    """OperatorAssertion.toPages (core/trino-main/src/test/java/io/trino/operator/OperatorAssertion.java:62-138):
    feed while needsInput, drain getOutput, then finish and drain until isFinished."""
    out = []
    it = iter(input_pages)
    pending = next(it, None)
    for _ in range(1 << 20):
        progressed = False
        if pending is not None and operator.needsInput():
            if pending.position_count > 0:  # Driver.java:391 never hands over empty pages
                operator.addInput(pending)
            pending = next(it, None)
            progressed = True
        page = operator.getOutput()
        if page is not None:
            if page.position_count > 0:
                out.append(page)
            progressed = True
        if pending is None and not progressed:
            break
    operator.finish()
    for _ in range(1 << 20):
        if operator.isFinished():
            break
        page = operator.getOutput()
        if page is not None and page.position_count > 0:
            out.append(page)
    assert operator.isFinished(), "operator did not finish"
    return out


def source_to_pages(operator, limit=1 << 22):
    """OperatorAssertion.toPages for a SourceOperator (no input pages): pull getOutput until isFinished."""
    out = []
    for _ in range(limit):
        if operator.isFinished():
            return out
        page = operator.getOutput()
        if page is not None and page.position_count > 0:
            out.append(page)
    raise RuntimeError("source operator did not finish")


class Driver:
    """Single-threaded operator pipeline, the shape of Driver.processInternal
    (core/trino-main/src/main/java/io/trino/operator/Driver.java:355-457): a page source feeding operators[0], every
    pass moves at most one page between each adjacent pair, finish() propagates downstream once an operator is
    finished; the pages of the last operator are collected."""

    def __init__(self, source_pages, operators):
        self.source = iter(source_pages)
        self.operators = list(operators)
        self.output = []

    def run(self):
        ops = self.operators
        pending = next(self.source, None)
        source_done = pending is None
        if source_done:
            ops[0].finish()
        for _ in range(1 << 22):
            moved = False
            if pending is not None and ops[0].needsInput():
                if pending.position_count > 0:
                    ops[0].addInput(pending)
                pending = next(self.source, None)
                moved = True
                if pending is None and not source_done:
                    source_done = True
                    ops[0].finish()
            for i in range(len(ops) - 1):
                cur, nxt = ops[i], ops[i + 1]
                if not cur.isFinished() and nxt.needsInput():
                    page = cur.getOutput()
                    if page is not None and page.position_count > 0:
                        nxt.addInput(page)
                        moved = True
                if cur.isFinished():
                    nxt.finish()  # idempotent (Driver.java:365-372)
            last = ops[-1]
            page = last.getOutput()
            if page is not None:
                if page.position_count > 0:
                    self.output.append(page)
                moved = True
            if last.isFinished():
                return self.output
            if not moved and pending is None:
                # nothing moved and nothing is left to feed: operators must be draining (finish already sent)
                if all(op.isFinished() for op in ops):
                    return self.output
        raise RuntimeError("pipeline did not finish")


# ---- device pages without torch (JNI-style hosts, tests) -----------------------------------------------
def upload_page(page):
    """Copies a host Page into HBM through the C ABI; returns a PA_MEM_DEVICE Page."""
    blocks = []

    def up(arr):
        if arr is None:
            return None
        arr = np.ascontiguousarray(arr)
        alloc = DeviceAllocation(max(arr.nbytes, 16))
        check(lib().pa_memcpy_h2d(alloc.ptr, arr.ctypes.data, arr.nbytes, None))
        return DeviceBuffer(alloc.ptr, arr.nbytes, alloc)

    def up_block(b):
        if b.encoding == abi.ROW_FIELDS:
            return Block(abi.ROW, abi.ROW_FIELDS, b.position_count, nulls=up(b.nulls), fields=[up_block(f) for f in b.fields])
        dictionary = up_block(b.dictionary) if b.dictionary is not None else None  # DictionaryBlock / RunLengthEncodedBlock
        return Block(b.type, b.encoding, b.position_count, values=up(b.values), offsets=up(b.offsets), nulls=up(b.nulls),
                     ids=up(b.ids), dictionary=dictionary)

    for b in page.blocks:
        blocks.append(up_block(b))
    return Page(blocks, page.position_count, abi.MEM_DEVICE)


def download(buf, dtype, count):
    """DeviceBuffer -> numpy array."""
    out = np.zeros(count, dtype=dtype)
    if count:
        check(lib().pa_memcpy_d2h(out.ctypes.data, buf.ptr, out.nbytes, None))
    return out


def download_page(page):
    """PA_MEM_DEVICE Page (flat / varwidth blocks) -> host Page."""
    n = page.position_count
    blocks = []
    for b in page.blocks:
        nulls = download(b.nulls, np.uint8, n) if b.nulls is not None else None
        if b.encoding == abi.ROW_FIELDS:
            blocks.append(Block.row(download_page(Page(b.fields, n, abi.MEM_DEVICE)).blocks, nulls))
            continue
        if b.encoding == abi.VARWIDTH:
            off = download(b.offsets, np.int32, n + 1)
            total = int(off[n]) if n else 0
            vals = download(b.values, np.uint8, total) if total else np.zeros(1, np.uint8)
            blocks.append(Block(b.type, abi.VARWIDTH, n, values=vals, offsets=off, nulls=nulls))
        elif b.type == abi.LONG_DECIMAL:
            blocks.append(Block(b.type, abi.FLAT, n, values=download(b.values, np.uint64, 2 * n).reshape(n, 2), nulls=nulls))
        else:
            dt = {abi.BIGINT: np.int64, abi.INTEGER: np.int32, abi.DATE: np.int32, abi.DOUBLE: np.float64,
                  abi.BOOLEAN: np.uint8, abi.REAL: np.float32, abi.DECIMAL: np.int64}[b.type]
            blocks.append(Block(b.type, abi.FLAT, n, values=download(b.values, dt, n), nulls=nulls))
    return Page(blocks, n, abi.MEM_HOST)
