"""Hash-partitioned exchange between the GPUs of one node -- ctypes caller of the native exchange (op_exchange.cpp, comm.cpp).

Reference: rows are routed by
    partition = (int) XxHash64.hash(Long.reverse(rawHash)) & (P - 1)     local exchange / PartitionedLookupSource
                (core/trino-main/src/main/java/io/trino/operator/exchange/LocalPartitionGenerator.java:45-65,
                 core/trino-main/src/main/java/io/trino/operator/join/PartitionedLookupSource.java:143-152)
    partition = (rawHash & MAX_LONG) % P                                  remote exchange
                (core/trino-main/src/main/java/io/trino/operator/HashGenerator.java:24-35,
                 core/trino-main/src/main/java/io/trino/operator/PartitionedOutputOperator.java:411-431)
with rawHash = InterpretedHashGenerator over the partition channels, appended per partition in ascending position
order (core/trino-main/src/main/java/io/trino/operator/exchange/PartitioningExchanger.java:59-82), then serialised
and pulled over HTTP.  Here: one rank per GPU, P = world size.  Everything on the data path is behind the C ABI
(include/presto_amd.h, "hash-partitioned exchange"): the PartitionedOutput sink regroups every page by destination on
the device, the exchange source runs ONE count all-gather and ONE variable all-to-all (grouped ncclSend / ncclRecv, RCCL
over xGMI) per exchange and hands the received rows out as one page.  This module only creates the handles:

  Comm          pa_comm: the RCCL communicator (the 128-byte unique id travels through the control plane -- presto_amd/control.py, or a
                torch.distributed group in the CPU tests; in Trino the coordinator would ship it), or the host transport, whose two
                collectives this module does over the control plane -- several ranks sharing one GPU in the tests
  Exchange      pa_exchange: the OutputBuffer + ExchangeClient pair of one exchange on this rank
  ExchangeOperator   sink + source behind one Operator, for a Driver pipeline that contains the exchange step
"""
import ctypes as C

from . import abi
from ._lib import check, lib
from .page import Block, Page


class Comm:
    """pa_comm.  Comm.rccl(group) / Comm.host(group) are collective over the ranks of the torch.distributed group."""

    def __init__(self, handle, rank, world, keep=None):
        self._h, self.rank, self.world, self._keep = handle, rank, world, keep

    @staticmethod
    def single():
        """World of one rank (RCCL with itself): the exchange path on one GPU."""
        ident = (C.c_uint8 * abi.COMM_ID_BYTES)()
        check(lib().pa_comm_unique_id(ident))
        h = C.c_void_p()
        check(lib().pa_comm_create(ident, 0, 1, C.byref(h)))
        return Comm(h, 0, 1)

    @staticmethod
    def rccl(control=None, group=None):
        """ncclCommInitRank over the ranks of `control` (presto_amd.control.ControlPlane: the torch-free control plane of bench.py's
        ranks) or, without one, of the torch.distributed group.  Collective."""
        if control is None:
            from .control import TorchControlPlane
            control = TorchControlPlane(group)
        ident = (C.c_uint8 * abi.COMM_ID_BYTES)()
        if control.rank == 0:
            check(lib().pa_comm_unique_id(ident))
        ident = (C.c_uint8 * abi.COMM_ID_BYTES).from_buffer_copy(control.broadcast(bytes(ident), src=0))
        h = C.c_void_p()
        check(lib().pa_comm_create(ident, control.rank, control.world, C.byref(h)))
        return Comm(h, control.rank, control.world)

    @staticmethod
    def host(control=None, group=None):
        """The two collectives done by the host, over the control plane (pa_host_transport): several ranks sharing one GPU in the
        tests -- RCCL refuses that -- not a data path."""
        import numpy as np
        if control is None:
            from .control import TorchControlPlane
            control = TorchControlPlane(group)
        rank, world = control.rank, control.world

        def all_gather(ctx, send, recv, count):
            try:
                mine = np.ctypeslib.as_array(send, shape=(count,)).tobytes()
                got = control.all_gather(mine)
                np.ctypeslib.as_array(recv, shape=(world * count,))[:] = np.frombuffer(b"".join(got), dtype=np.int64)
                return 0
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return abi.ERR_DEVICE

        def all_to_all(ctx, send, soff, sbytes, recv, roff, rbytes):
            try:
                so = [int(soff[i]) for i in range(world)]
                sb = [int(sbytes[i]) for i in range(world)]
                ro = [int(roff[i]) for i in range(world)]
                rb = [int(rbytes[i]) for i in range(world)]
                stotal = max([o + b for o, b in zip(so, sb)] + [0])
                rtotal = max([o + b for o, b in zip(ro, rb)] + [0])
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(max(stotal, 1),))
                dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(max(rtotal, 1),))
                got = control.all_to_all([src[o:o + b].tobytes() for o, b in zip(so, sb)])
                for off, b, blob in zip(ro, rb, got):
                    if len(blob) != b:
                        raise ValueError("all_to_all_v: %d bytes arrived where %d were announced" % (len(blob), b))
                    dst[off:off + b] = np.frombuffer(blob, dtype=np.uint8)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return abi.ERR_DEVICE

        t = abi.pa_host_transport()
        t.ctx = None
        t.all_gather_i64 = abi.ALL_GATHER_I64(all_gather)
        t.all_to_all_v = abi.ALL_TO_ALL_V(all_to_all)
        h = C.c_void_p()
        check(lib().pa_comm_create_host(C.byref(t), rank, world, C.byref(h)))
        return Comm(h, rank, world, keep=t)

    def preflight(self, bytes_per_peer=1 << 20, stream=None):
        """pa_comm_preflight: a checked all-to-all + all-reduce + all-gather over the fresh communicator (collective)."""
        check(lib().pa_comm_preflight(self._h, bytes_per_peer, stream))

    def allReduce(self, values, op=abi.COMM_SUM, stream=None):
        arr = (C.c_int64 * len(values))(*[int(v) for v in values])
        check(lib().pa_comm_all_reduce_i64(self._h, arr, len(values), op, stream))
        return [int(v) for v in arr]

    def destroy(self):
        if self._h:
            lib().pa_comm_destroy(self._h)
            self._h = None


class Exchange:
    """pa_exchange: one hash-partitioned exchange on this rank (what sits between a PartitionedOutputOperator and the
    ExchangeOperator of the consuming pipeline)."""

    def __init__(self, comm, types, partition_channels, hash_channel=-1, partition_rule=-1, sink_count=1):
        d = abi.pa_exchange_desc()
        t = abi.int32_array(types)
        pc = abi.int32_array(partition_channels)
        d.channel_count = len(types)
        d.types = C.cast(t, C.POINTER(C.c_int32))
        d.partition_channel_count = len(partition_channels)
        d.partition_channels = C.cast(pc, C.POINTER(C.c_int32))
        d.hash_channel = hash_channel
        d.partition_rule = partition_rule
        d.sink_count = sink_count
        h = C.c_void_p()
        check(lib().pa_exchange_create(C.byref(d), comm._h, C.byref(h)))
        self._h, self.comm, self.types = h, comm, list(types)

    def stats(self):
        """(rows sent, rows received, payload bytes sent to other ranks, device ms of the all-to-all)"""
        a, b, c, ms = C.c_int64(), C.c_int64(), C.c_int64(), C.c_double()
        check(lib().pa_exchange_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(ms)))
        return a.value, b.value, c.value, ms.value

    def destroy(self):
        if self._h:
            lib().pa_exchange_destroy(self._h)
            self._h = None


def PartitionedOutputOperator(exchange, stream=None):
    """The sink of the producing pipeline (PartitionedOutputOperator.java:411-431)."""
    from .operators import Operator
    h = C.c_void_p()
    check(lib().pa_partitioned_output_create(exchange._h, stream, C.byref(h)))
    return Operator(h, [exchange])


def ExchangeSourceOperator(exchange, output_mem=abi.MEM_DEVICE, stream=None):
    """The source of the consuming pipeline (ExchangeOperator.java)."""
    from .operators import Operator
    h = C.c_void_p()
    check(lib().pa_exchange_source_create(exchange._h, output_mem, stream, C.byref(h)))
    return Operator(h, [exchange])


class ExchangeOperator:
    """One exchange step inside a Driver pipeline: pages added go to the PartitionedOutput sink; finish() finishes the
    sink, after which getOutput pulls the exchange source -- the collective: every rank's Driver gets here once per
    exchange, in program order -- and returns the rows this rank received as one device page."""

    def __init__(self, comm, types, partition_channels, stream=None, output_mem=abi.MEM_DEVICE, hash_channel=-1):
        self.exchange = Exchange(comm, types, partition_channels, hash_channel=hash_channel)
        self.sink = PartitionedOutputOperator(self.exchange, stream)
        self.source = ExchangeSourceOperator(self.exchange, output_mem, stream)
        self._finishing = False

    def needsInput(self):
        return not self._finishing

    def addInput(self, page):
        self.sink.addInput(page)

    def finish(self):
        if not self._finishing:
            self._finishing = True
            self.sink.finish()

    def getOutput(self):
        return self.source.getOutput() if self._finishing else None

    def isFinished(self):
        return self._finishing and self.source.isFinished()

    def isBlocked(self):
        return False

    def stats(self):
        return self.exchange.stats()

    def close(self):
        self.sink.close()
        self.source.close()
        self.exchange.destroy()


# ---- partial -> final aggregation across ranks --------------------------------------------------------------------
def partial_layout(key_types, aggregates):
    """Channel types of a Step.PARTIAL output page and the aggregate list of the matching Step.FINAL operator.

    aggregates: list of (fn, input_channel, input_type[, mask]) of the SINGLE-step aggregation.  The intermediate
    channels follow include/presto_amd.h: [count] for count / count(*), [count, sum] for sum / avg, [count, value] for min / max."""
    types = list(key_types)
    final = []
    for a in aggregates:
        fn, in_type = a[0], a[2]
        first = len(types)
        types.append(abi.BIGINT)
        if fn in (abi.AGG_SUM, abi.AGG_AVG) and in_type in (abi.DECIMAL, abi.LONG_DECIMAL):
            # sum / avg over DECIMAL(p, s): [count, sum DECIMAL(38, s)]; the FINAL aggregate's input_type names its RESULT type
            scale = in_type.scale if isinstance(in_type, abi.DecimalType) else 0
            types.append(abi.decimal(38, scale))
            final.append((fn, first, abi.decimal(38, scale) if fn == abi.AGG_SUM else in_type))
        elif fn in (abi.AGG_SUM, abi.AGG_AVG):
            value_type = abi.DOUBLE if (fn == abi.AGG_AVG or in_type in (abi.DOUBLE, abi.REAL)) else abi.BIGINT
            types.append(value_type)
            # (sum / avg over REAL: a DOUBLE state; the FINAL step narrows its result, so it keeps the input type)
            final.append((fn, first, abi.REAL if in_type == abi.REAL else value_type))
        elif fn in (abi.AGG_MIN, abi.AGG_MAX):
            types.append(in_type)  # [count, value of the input type]
            final.append((fn, first, in_type))
        else:
            final.append((fn, first, abi.BIGINT))
    return types, final


def _state_payload(page):
    """A host Page of intermediate states as (type, type parameter, python values) per channel -- what travels between the ranks."""
    if page is None or page.position_count == 0:
        return None
    params = getattr(page, "type_params", None) or [0] * len(page.blocks)
    return [(int(b.type), int(params[k]) if k < len(params) else 0, b.to_pylist()) for k, b in enumerate(page.blocks)]


def _concat_payloads(payloads):
    """The ranks' payloads of one aggregation behind each other, in rank order: ONE page for the FINAL operator (its combine order
    is the rows' order: rank 0's states first)."""
    payloads = [p for p in payloads if p is not None]
    if not payloads:
        return None
    return [(t, param, [v for p in payloads for v in p[k][2]]) for k, (t, param, _values) in enumerate(payloads[0])]


def _state_page(cols):
    """The host Page of a received payload; every channel type of a PARTIAL state (include/presto_amd.h): BIGINT counts, DOUBLE /
    BIGINT / DECIMAL(38, s) sums (LONG_DECIMAL: two words per value), min / max values of any type, VARCHAR keys."""
    blocks = []
    for t, _param, values in cols:
        if t == abi.VARCHAR:
            blocks.append(Block.varchar(values))
        elif t == abi.LONG_DECIMAL:
            blocks.append(Block.long_decimal(values))      # None = NULL
        else:
            nulls = [v is None for v in values]
            blocks.append(Block.flat(t, [0 if v is None else v for v in values], nulls if any(nulls) else None))
    return Page(blocks, len(cols[0][2]) if cols else 0)


def merge_partial_aggregations(partial_page, make_final_operator, group=None, dst=0):
    """The FINAL step of a row-range-sharded aggregation (HashAggregationOperator Step.PARTIAL on every rank ->
    Step.FINAL on one; the reference ships the partial pages through its exchange).  `partial_page` is this rank's
    host Page of intermediate states (or None when the rank produced no group); the pages are tiny (Q1: 4 rows), so
    they travel as objects.  Returns the final host Page on rank `dst`, None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gathered = [None] * world
    dist.all_gather_object(gathered, _state_payload(partial_page), group=group)
    if rank != dst:
        return None
    op = make_final_operator()
    for cols in gathered:  # rank order: a fixed combine order
        if cols is None:
            continue
        op.addInput(_state_page(cols))
    op.finish()
    return op.getOutput()


class PartialStateMerger:
    """The FINAL step of row-range-sharded aggregations, for a step loop: ONE fixed-size all-gather per call carries the Step.PARTIAL
    pages of several aggregations (Q1: 4 rows, Q6: 1 row per rank), the FINAL operators run on rank `dst` over the ranks' pages
    in rank order -- a fixed combine order (HashAggregationOperator.java:390 Step.PARTIAL -> Step.FINAL;
    DoubleSumAggregation.java:47-52 combine).  Same result as merge_partial_aggregations; the buffers are made once.

    comm: a Comm -- the all-gather runs over the library's communicator (pa_comm_all_gather_i64: RCCL over xGMI, or the host
    transport); None: over torch.distributed host tensors of `group` (CPU ranks of the tests)."""

    CAPACITY = 1 << 13   # bytes per rank and call

    def __init__(self, group=None, dst=0, comm=None):
        self.group, self.dst, self.comm = group, dst, comm
        if comm is not None:
            self.world, self.rank = comm.world, comm.rank
            words = self.CAPACITY // 8
            self.send = (C.c_int64 * words)()
            self.recv = (C.c_int64 * (words * self.world))()
        else:
            import torch
            import torch.distributed as dist
            self.dist = dist
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
            self.send = torch.zeros(self.CAPACITY, dtype=torch.uint8)
            self.recv = torch.zeros(self.CAPACITY * self.world, dtype=torch.uint8)

    def _all_gather(self, blob):
        """blob (bytes, <= CAPACITY) of every rank -> list of CAPACITY-byte chunks in rank order"""
        if self.comm is not None:
            C.memmove(self.send, blob, len(blob))
            check(lib().pa_comm_all_gather_i64(self.comm._h, self.send, self.recv, self.CAPACITY // 8, None))
            got = bytes(self.recv)
        else:
            import numpy as np
            import torch
            self.send[:len(blob)] = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy())
            self.dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
            got = self.recv.numpy().tobytes()
        return [got[r * self.CAPACITY:(r + 1) * self.CAPACITY] for r in range(self.world)]

    def merge(self, partial_pages, make_final_operator):
        """partial_pages: {name: host Page | None} on every rank (same names everywhere); make_final_operator: {name: () -> Operator};
        returns {name: final host Page | None} on rank dst, None elsewhere.  Collective."""
        import pickle
        import struct
        payload = {name: _state_payload(page) for name, page in partial_pages.items()}
        blob = pickle.dumps(payload, protocol=pickle.HIGHEST_PROTOCOL)
        if len(blob) + 8 > self.CAPACITY:
            raise ValueError("partial aggregation states of %d bytes: not a few-groups result, use an exchange" % len(blob))
        chunks = self._all_gather(struct.pack("<q", len(blob)) + blob)
        if self.rank != self.dst:
            return None
        per_rank = []
        for chunk in chunks:
            n = struct.unpack("<q", chunk[:8])[0]
            per_rank.append(pickle.loads(chunk[8:8 + n]))
        # every FINAL operator is given its page before the first result is asked for: the operators work on streams of their own,
        # so the launch chains of Q6's and Q1's FINAL steps overlap instead of following each other
        ops = {}
        for name in partial_pages:
            op = make_final_operator[name]()
            cols = _concat_payloads([payload[name] for payload in per_rank])   # rank order
            if cols is not None:
                op.addInput(_state_page(cols))
            op.finish()
            ops[name] = op
        out = {}
        for name, op in ops.items():
            out[name] = op.getOutput()
            if hasattr(op, "close"):
                op.close()
        return out
