"""Hash-partitioned exchange between the GPUs of one node: the RCCL counterpart of the reference's page shuffle.

Reference: rows are routed by
    partition = (int) XxHash64.hash(Long.reverse(rawHash)) & (P - 1)     local exchange / PartitionedLookupSource
                (core/trino-main/src/main/java/io/trino/operator/exchange/LocalPartitionGenerator.java:45-65,
                 core/trino-main/src/main/java/io/trino/operator/join/PartitionedLookupSource.java:143-152)
    partition = (rawHash & MAX_LONG) % P                                  remote exchange
                (core/trino-main/src/main/java/io/trino/operator/HashGenerator.java:24-35,
                 core/trino-main/src/main/java/io/trino/operator/PartitionedOutputOperator.java:411-431)
with rawHash = InterpretedHashGenerator over the partition channels, appended per partition in ascending position
order (core/trino-main/src/main/java/io/trino/operator/exchange/PartitioningExchanger.java:59-82), then serialised
and pulled over HTTP.  Here: one rank per GPU, P = world size; the per-row work (hash, partition id, stable
partition, gather into per-destination send buffers) runs in HIP kernels behind the C ABI, and the transfer is one
all-to-all per column over xGMI (torch.distributed `nccl` == RCCL): raw column bytes, no serialisation, no
compression.  Row order inside what a rank receives = (source rank, source position), i.e. what a consumer that
drains the producers in rank order would see.

The `ops` object supplies the per-row kernels so that the exchange logic can be exercised on CPU ranks (gloo) in
the tests with a checker implementation; the product implementation is DeviceOps.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import abi
from .page import Block, DeviceBuffer, Page

_TORCH_DTYPE = {abi.BIGINT: torch.int64, abi.INTEGER: torch.int32, abi.DATE: torch.int32, abi.DOUBLE: torch.float64,
                abi.BOOLEAN: torch.uint8}


class DeviceOps:
    """Per-row exchange kernels on this rank's GPU through libpresto_amd.so; columns are torch CUDA tensors."""

    def __init__(self):
        from ._lib import check, lib
        self._check, self._lib = check, lib()

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream or None

    @staticmethod
    def _page(columns, types):
        n = rows_of(columns[0]) if columns else 0
        blocks = []
        for c, t in zip(columns, types):
            if t == abi.VARCHAR:  # (bytes uint8, offsets int32[n + 1])
                v, o = c
                blocks.append(Block(t, abi.VARWIDTH, n, values=DeviceBuffer(v.data_ptr(), v.numel(), v), offsets=DeviceBuffer(o.data_ptr(), o.numel() * 4, o)))
            else:
                blocks.append(Block(t, abi.FLAT, n, values=DeviceBuffer(c.data_ptr(), c.numel() * c.element_size(), c)))
        return Page(blocks, n, abi.MEM_DEVICE)

    def hash_rows(self, columns, types, channels):
        page = self._page(columns, types)
        cpage, keep = page.to_c()
        out = torch.empty(page.position_count, dtype=torch.int64, device=columns[0].device)
        ch = abi.int32_array(channels)
        self._check(self._lib.pa_hash_page(C.byref(cpage), len(channels), ch, out.data_ptr(), self._stream()))
        return out

    def partition_ids(self, raw_hash, partition_count, local):
        out = torch.empty(raw_hash.numel(), dtype=torch.int32, device=raw_hash.device)
        self._check(self._lib.pa_partition_ids(raw_hash.data_ptr(), raw_hash.numel(), partition_count, 1 if local else 0,
                                               out.data_ptr(), self._stream()))
        return out

    def partition_positions(self, partition, partition_count):
        pos = torch.empty(partition.numel(), dtype=torch.int32, device=partition.device)
        counts = (C.c_int64 * partition_count)()
        self._check(self._lib.pa_partition_positions(partition.data_ptr(), partition.numel(), partition_count, pos.data_ptr(),
                                                     counts, self._stream()))
        return pos, [int(c) for c in counts]

    def partition_columns(self, partition, partition_count, columns, want_positions):
        """The flat columns regrouped by partition in one stable multisplit pass (row order kept inside a partition) instead of
        partition_positions + one gather per column -> (regrouped columns, positions or None, rows per partition).  None when
        the partition count is beyond the stable kernel's 256."""
        if partition_count > 256:
            return None
        n = partition.numel()
        cols = list(columns)
        if want_positions:
            cols.append(torch.arange(n, dtype=torch.int32, device=partition.device))
        outs = [torch.empty_like(c) for c in cols]
        vp = C.c_void_p
        ins = (vp * max(len(cols), 1))(*[c.data_ptr() for c in cols])
        ous = (vp * max(len(cols), 1))(*[c.data_ptr() for c in outs])
        widths = (C.c_int32 * max(len(cols), 1))(*[c.element_size() for c in cols])
        counts = (C.c_int64 * partition_count)()
        self._check(self._lib.pa_partition_columns_stable(partition.data_ptr(), n, partition_count, ins, ous, widths, len(cols), counts, self._stream()))
        positions = outs.pop() if want_positions else None
        return outs, positions, [int(c) for c in counts]

    def gather_varwidth(self, values, offsets, positions):
        """Block.copyPositions for a VARCHAR column -> (bytes, offsets, per-row lengths)."""
        n = positions.numel()
        lengths = torch.empty(n, dtype=torch.int32, device=values.device)
        out_offsets = torch.empty(n + 1, dtype=torch.int32, device=values.device)
        total = C.c_int64()
        self._check(self._lib.pa_varwidth_gather_offsets(offsets.data_ptr(), positions.data_ptr(), n, lengths.data_ptr(), out_offsets.data_ptr(),
                                                         C.byref(total), self._stream()))
        out = torch.empty(max(total.value, 1), dtype=torch.uint8, device=values.device)
        self._check(self._lib.pa_varwidth_gather_bytes(values.data_ptr(), offsets.data_ptr(), positions.data_ptr(), n, out_offsets.data_ptr(),
                                                       out.data_ptr(), self._stream()))
        return out[:total.value], out_offsets, lengths

    def offsets_from_lengths(self, lengths):
        n = lengths.numel()
        out = torch.empty(n + 1, dtype=torch.int32, device=lengths.device)
        total = C.c_int64()
        self._check(self._lib.pa_offsets_from_lengths(lengths.data_ptr() if n else None, n, out.data_ptr(), C.byref(total), self._stream()))
        return out, total.value

    def gather(self, column, positions):
        out = torch.empty(positions.numel(), dtype=column.dtype, device=column.device)
        self._check(self._lib.pa_gather_flat(column.data_ptr(), column.element_size(), positions.data_ptr(), positions.numel(),
                                             out.data_ptr(), self._stream()))
        return out


def rows_of(column):
    """rows of a column: a 1-D tensor, or (bytes, offsets[n + 1]) for VARCHAR"""
    return int(column[1].shape[0]) - 1 if isinstance(column, (tuple, list)) else int(column.shape[0])


def partition_rows(ops, columns, types, hash_channels, partition_count, local=True, raw_hash=None):
    """Returns (positions grouped by partition, rows per partition)."""
    if raw_hash is None:
        raw_hash = ops.hash_rows(columns, types, hash_channels)
    part = ops.partition_ids(raw_hash, partition_count, local)
    return ops.partition_positions(part, partition_count)


def exchange_columns(ops, columns, types, hash_channels, group=None, local=None, raw_hash=None):
    """All-to-all of the rows of `columns` by the hash of `hash_channels`.  A column is a 1-D tensor, or for VARCHAR the pair
    (bytes uint8, offsets int32[n + 1]): its rows travel as per-row lengths (split by rows) plus the bytes (split by the byte
    totals of the destinations), and the receiver rebuilds the offsets with a scan.

    Returns (received columns, rows received from every source rank)."""
    world = dist.get_world_size(group)
    if local is None:
        local = (world & (world - 1)) == 0  # LocalPartitionGenerator needs a power of two
    device = (columns[0][0] if isinstance(columns[0], (tuple, list)) else columns[0]).device
    rows = rows_of(columns[0])
    regrouped = {}
    if rows > 0:
        fast = None
        if hasattr(ops, "partition_columns"):
            # device path: all flat columns regrouped by destination in one stable multisplit pass (same row order as the
            # position list gives); a position list is only made for VARCHAR columns
            if raw_hash is None:
                raw_hash = ops.hash_rows(columns, types, hash_channels)
            part = ops.partition_ids(raw_hash, world, local)
            flat = [i for i, t in enumerate(types) if t != abi.VARCHAR]
            fast = ops.partition_columns(part, world, [columns[i] for i in flat], any(t == abi.VARCHAR for t in types))
        if fast is not None:
            outs, positions, send_counts = fast
            regrouped = dict(zip(flat, outs))
        else:
            positions, send_counts = partition_rows(ops, columns, types, hash_channels, world, local, raw_hash)
    else:  # a rank with nothing to send still takes part in the collectives
        positions, send_counts = None, [0] * world
    # 8 x 8 count matrix: every rank learns how much it receives from each source
    sc = torch.tensor(send_counts, dtype=torch.int64, device=device)
    rc = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(rc, sc, group=group)
    recv_counts = [int(x) for x in rc.tolist()]
    received = []
    for ci, (col, t) in enumerate(zip(columns, types)):
        if t == abi.VARCHAR:
            values, offsets = col
            if rows > 0:
                send_bytes, send_offsets, send_lengths = ops.gather_varwidth(values, offsets, positions)
                bounds = torch.tensor([0] + list(torch.tensor(send_counts).cumsum(0).tolist()), dtype=torch.int64, device=device)
                ends = send_offsets.index_select(0, bounds).tolist()  # byte offset at every destination boundary
                send_byte_counts = [int(ends[i + 1] - ends[i]) for i in range(world)]
            else:
                send_bytes = torch.empty(0, dtype=torch.uint8, device=device)
                send_lengths = torch.empty(0, dtype=torch.int32, device=device)
                send_byte_counts = [0] * world
            sb = torch.tensor(send_byte_counts, dtype=torch.int64, device=device)
            rb = torch.empty(world, dtype=torch.int64, device=device)
            dist.all_to_all_single(rb, sb, group=group)
            recv_byte_counts = [int(x) for x in rb.tolist()]
            recv_lengths = torch.empty(sum(recv_counts), dtype=torch.int32, device=device)
            dist.all_to_all_single(recv_lengths, send_lengths, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
            recv_bytes = torch.empty(sum(recv_byte_counts), dtype=torch.uint8, device=device)
            dist.all_to_all_single(recv_bytes, send_bytes.contiguous(), output_split_sizes=recv_byte_counts, input_split_sizes=send_byte_counts, group=group)
            recv_offsets, total = ops.offsets_from_lengths(recv_lengths)
            assert total == sum(recv_byte_counts)
            received.append((recv_bytes, recv_offsets))
            continue
        send = (regrouped[ci] if ci in regrouped else ops.gather(col, positions)) if rows > 0 else col
        recv = torch.empty(sum(recv_counts), dtype=col.dtype, device=device)
        dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
        received.append(recv)
    return received, recv_counts


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
        if fn in (abi.AGG_SUM, abi.AGG_AVG):
            value_type = abi.DOUBLE if (fn == abi.AGG_AVG or in_type == abi.DOUBLE) else abi.BIGINT
            types.append(value_type)
            final.append((fn, first, value_type))
        elif fn in (abi.AGG_MIN, abi.AGG_MAX):
            types.append(in_type)  # [count, value of the input type]
            final.append((fn, first, in_type))
        else:
            final.append((fn, first, abi.BIGINT))
    return types, final


def merge_partial_aggregations(partial_page, make_final_operator, group=None, dst=0):
    """The FINAL step of a row-range-sharded aggregation (HashAggregationOperator Step.PARTIAL on every rank ->
    Step.FINAL on one; the reference ships the partial pages through its exchange).  `partial_page` is this rank's
    host Page of intermediate states (or None when the rank produced no group); the pages are tiny (Q1: 4 rows), so
    they travel as objects.  Returns the final host Page on rank `dst`, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    payload = None
    if partial_page is not None and partial_page.position_count > 0:
        payload = [(b.type, b.to_pylist()) for b in partial_page.blocks]
    gathered = [None] * world
    dist.all_gather_object(gathered, payload, group=group)
    if rank != dst:
        return None
    op = make_final_operator()
    for cols in gathered:  # rank order: a fixed combine order
        if cols is None:
            continue
        blocks = []
        for t, values in cols:
            nulls = [v is None for v in values]
            if t == abi.VARCHAR:
                blocks.append(Block.varchar(values))
            else:
                blocks.append(Block.flat(t, [0 if v is None else v for v in values], nulls))
        op.addInput(Page(blocks, len(cols[0][1]) if cols else 0))
    op.finish()
    return op.getOutput()
