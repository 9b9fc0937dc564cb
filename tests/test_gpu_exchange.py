"""GPU parity of the exchange kernels (row hash, partition id, stable partition, multisplit) against the oracle, and the native
exchange (pa_exchange_*): on one rank over RCCL, and between several ranks sharing the GPU through the host transport."""
import os

import numpy as np
import pytest
import torch

from presto_amd import abi
from presto_amd.operators import download, upload_page
from presto_amd.page import Block, DeviceBuffer, Page

pytestmark = pytest.mark.gpu


def test_hash_page_all_types_bit_exact(gpu, oracle):
    import ctypes as C
    rng = np.random.default_rng(2)
    n = 30011
    words = [b"", b"A", b"BUILDING", b"0123456789abcdefghijklmnopqrstuvwxyz0123456789", None, b"xyz"]
    page = Page([
        Block.bigint(rng.integers(-2 ** 62, 2 ** 62, n), rng.random(n) < 0.1),
        Block.integer(rng.integers(-2 ** 31, 2 ** 31 - 1, n)),
        Block.date(rng.integers(8000, 11000, n)),
        Block.double(np.where(rng.random(n) < 0.1, -0.0, rng.standard_normal(n)), rng.random(n) < 0.05),
        Block.boolean(rng.random(n) < 0.5),
        Block.varchar([words[i] for i in rng.integers(0, len(words), n)]),
    ], n)
    dev = upload_page(page)
    cpage, keep = dev.to_c()
    for channels in ([0], [5], [3, 4], [0, 1, 2, 3, 4, 5], [5, 0]):
        out = C.c_void_p()
        from presto_amd._lib import DeviceAllocation, check, lib
        buf = DeviceAllocation(8 * n)
        check(lib().pa_hash_page(C.byref(cpage), len(channels), abi.int32_array(channels), buf.ptr, None))
        check(lib().pa_stream_synchronize(None))
        got = download(DeviceBuffer(buf.ptr, 8 * n), np.int64, n)
        assert np.array_equal(got, oracle.hash_page(page, channels)), channels


@pytest.mark.parametrize("partitions,local", [(2, True), (8, True), (64, True), (8, False), (7, False), (3, False)])
def test_partition_ids_and_stable_positions(gpu, oracle, partitions, local):
    from presto_amd._lib import check, lib
    rng = np.random.default_rng(4)
    n = 123457
    raw = rng.integers(-2 ** 63, 2 ** 63 - 1, n, dtype=np.int64)
    d_raw = torch.from_numpy(raw).cuda()
    part = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    check(lib().pa_partition_ids(d_raw.data_ptr(), n, partitions, 1 if local else 0, part.data_ptr(), None))
    expected = oracle.partition_ids(raw, partitions, local)
    check(lib().pa_stream_synchronize(None))
    assert np.array_equal(part.cpu().numpy(), expected)
    pos = torch.empty(n, dtype=torch.int32, device="cuda")
    counts = np.zeros(partitions, dtype=np.int64)
    check(lib().pa_partition_positions(part.data_ptr(), n, partitions, pos.data_ptr(), counts.ctypes.data, None))
    epos, ecounts = oracle.partition_positions(expected, partitions)
    assert counts.tolist() == ecounts.tolist()
    assert np.array_equal(pos.cpu().numpy(), epos)  # ascending positions inside every partition


# ---- the native exchange (pa_exchange_*): PartitionedOutput sink -> one all-to-all -> exchange source ----
def exchange_pages(seed, n_pages, rows, nulls_from_page=1):
    """Pages of (BIGINT key, DOUBLE, VARCHAR, INTEGER): NULLs appear from page `nulls_from_page` on (a channel that turns
    nullable after rows were already buffered), the VARCHAR holds empty and long strings."""
    rng = np.random.default_rng(seed)
    words = [b"", b"a", b"BUILDING", b"0123456789abcdefghijklmnopqrstuvwxyz", b"\xc3\xa9", b"zz"]
    pages = []
    for k in range(n_pages):
        n = rows + 17 * k
        with_nulls = k >= nulls_from_page
        names = [words[i] + str(int(j)).encode() for i, j in zip(rng.integers(0, len(words), n), rng.integers(0, 1000, n))]
        if with_nulls:
            names = [None if rng.random() < 0.1 else s for s in names]
        pages.append(Page([
            Block.bigint(rng.integers(0, 10 ** 9, n), (rng.random(n) < 0.05) if with_nulls else None),
            Block.double(rng.random(n)),
            Block.varchar(names),
            Block.integer(rng.integers(-1000, 1000, n), (rng.random(n) < 0.2) if (with_nulls and k % 2 == 1) else None),
        ], n))
    return pages


EX_TYPES = [abi.BIGINT, abi.DOUBLE, abi.VARCHAR, abi.INTEGER]


def run_exchange(comm, pages, channels, device_pages, output_mem=abi.MEM_HOST):
    from presto_amd.exchange import ExchangeOperator
    from presto_amd.operators import download_page
    ex = ExchangeOperator(comm, EX_TYPES, channels, output_mem=output_mem)
    keep = []
    for p in pages:
        if device_pages:
            p = upload_page(p)
            keep.append(p)
        assert ex.needsInput() and ex.getOutput() is None
        ex.addInput(p)
    ex.finish()
    out = ex.getOutput()
    assert ex.isFinished() and ex.getOutput() is None
    stats = ex.stats()
    rows = [] if out is None else (out if output_mem == abi.MEM_HOST else download_page(out)).to_rows()
    ex.close()
    return rows, stats


def expected_for_rank(oracle, pages_by_rank, channels, world, rank, local):
    """What `rank` must receive: the rows its partition id names, by (source rank, page, position)."""
    rows = []
    for src in range(world):
        for p in pages_by_rank[src]:
            part = oracle.partition_ids(oracle.hash_page(p, channels), world, local)
            rows += [r for r, d in zip(p.to_rows(), part.tolist()) if d == rank]
    return rows


@pytest.mark.parametrize("device_pages,output_mem", [(False, abi.MEM_HOST), (True, abi.MEM_DEVICE)])
@pytest.mark.parametrize("channels", [[0], [2], [0, 2, 3]])
def test_native_exchange_on_one_rank_over_rccl(gpu, oracle, channels, device_pages, output_mem):
    """World of one rank: the count all-gather and the grouped ncclSend / ncclRecv run over RCCL against the rank itself, so
    every row must come back, in page order (one partition: the stable regrouping is the identity)."""
    from presto_amd.exchange import Comm
    comm = Comm.single()
    try:
        pages = exchange_pages(11, 4, 20011)
        rows, (sent, received, remote, ms) = run_exchange(comm, pages, channels, device_pages, output_mem)
        assert rows == [r for p in pages for r in p.to_rows()]
        assert sent == received == len(rows) and remote == 0 and ms >= 0
        assert any(r[0] is None for r in rows) and any(r[2] is None for r in rows) and any(r[3] is None for r in rows)
        # nothing to send at all: no output page, finished all the same
        rows, stats = run_exchange(comm, [], channels, device_pages, output_mem)
        assert rows == [] and stats[0] == 0
    finally:
        comm.destroy()


def test_comm_all_reduce_on_one_rank(gpu):
    from presto_amd.exchange import Comm
    comm = Comm.single()
    try:
        assert comm.allReduce([5, -7], abi.COMM_MIN) == [5, -7] and comm.allReduce([3], abi.COMM_SUM) == [3]
    finally:
        comm.destroy()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def shared_gpu_worker(rank, world, port, q):
    """One of `world` processes sharing the one GPU: native exchange with the host transport over gloo."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from presto_amd import _lib
    from presto_amd.exchange import Comm
    _lib.init(0)
    comm = Comm.host()
    try:
        pages = exchange_pages(100 + rank, [3, 1, 0][rank % 3], 5003, nulls_from_page=1 if rank == 0 else 99)
        out = {}
        for name, channels in (("key", [0]), ("string", [2])):
            out[name] = run_exchange(comm, pages, channels, device_pages=(rank % 2 == 0))
        out["reduce"] = (comm.allReduce([rank + 1, 10 - rank], abi.COMM_MIN), comm.allReduce([rank + 1], abi.COMM_SUM), comm.allReduce([rank], abi.COMM_MAX))
        q.put((rank, out))
        dist.barrier()
    finally:
        comm.destroy()
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_native_exchange_between_ranks_sharing_the_gpu(gpu, oracle, world):
    """The multi-rank logic of the native exchange (count matrix, blob layouts, per-source unpacking, NULL flags that only some
    ranks carry, ranks with 3 / 1 / 0 pages, power-of-two and other world sizes) with `world` processes on the one GPU of the
    box: RCCL refuses several ranks on one device, so the two collectives go through the library's host transport over gloo.
    Every rank must receive exactly the rows its partition id names, ordered by (source rank, source position)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=shared_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = dict(q.get(timeout=300) for _ in range(world))
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    pages_by_rank = [exchange_pages(100 + r, [3, 1, 0][r % 3], 5003, nulls_from_page=1 if r == 0 else 99) for r in range(world)]
    local = (world & (world - 1)) == 0
    total = sum(p.position_count for ps in pages_by_rank for p in ps)
    for name, channels in (("key", [0]), ("string", [2])):
        got_total = 0
        for r in range(world):
            rows, (sent, received, remote, ms) = results[r][name]
            assert rows == expected_for_rank(oracle, pages_by_rank, channels, world, r, local), (name, r)
            assert received == len(rows) and sent == sum(p.position_count for p in pages_by_rank[r])
            got_total += len(rows)
        assert got_total == total
    for r in range(world):
        assert results[r]["reduce"] == ([1, 10 - (world - 1)], [world * (world + 1) // 2], [world - 1])


@pytest.mark.parametrize("n,parts", [(1, 1), (1000, 3), (8192, 256), (8193, 257), (300001, 513), (2_000_003, 1024), (1 << 22, 64)])
def test_partition_columns_multisplit(gpu, n, parts):
    """pa_partition_columns: every column regrouped by partition with ONE permutation -- a partition's rows are contiguous,
    complete and the same rows in every column (their order inside the partition is free)."""
    import ctypes as C
    import torch
    from presto_amd._lib import check, lib
    g = torch.Generator(device="cuda").manual_seed(n)
    part = torch.randint(0, parts, (n,), dtype=torch.int32, device="cuda", generator=g)
    if parts > 4:
        part[part == 2] = 3  # an empty partition
    rowid = torch.arange(n, dtype=torch.int64, device="cuda")
    c4 = (rowid * 7 + 1).to(torch.int32)
    c1 = (rowid % 251).to(torch.uint8)
    cols = [rowid, c4, c1]
    outs = [torch.empty_like(c) for c in cols]
    torch.cuda.synchronize()
    vp = C.c_void_p
    ins = (vp * 3)(*[c.data_ptr() for c in cols])
    ous = (vp * 3)(*[c.data_ptr() for c in outs])
    widths = (C.c_int32 * 3)(8, 4, 1)
    counts = np.zeros(parts, dtype=np.int64)
    check(lib().pa_partition_columns(part.data_ptr(), n, parts, ins, ous, widths, 3, counts.ctypes.data, None))
    expected = torch.bincount(part.long(), minlength=parts).cpu().numpy()
    assert counts.tolist() == expected.tolist()
    ids = outs[0]
    assert torch.equal(outs[1], (ids * 7 + 1).to(torch.int32)) and torch.equal(outs[2], (ids % 251).to(torch.uint8))  # one permutation
    assert torch.equal(torch.sort(ids).values, rowid)  # a permutation of the rows
    bounds = np.concatenate([[0], np.cumsum(counts)])
    got_part = part[ids.long()] if n else part
    want = torch.repeat_interleave(torch.arange(parts, device="cuda", dtype=torch.int32), torch.tensor(counts, device="cuda"))
    assert torch.equal(got_part, want) and bounds[-1] == n


@pytest.mark.parametrize("n,parts", [(1, 1), (5000, 2), (8193, 8), (300001, 256), (1 << 22, 8)])
def test_partition_columns_stable(gpu, oracle, n, parts):
    """pa_partition_columns_stable: the columns regrouped by partition with ascending row order inside every partition -- exactly
    the gather through pa_partition_positions' position list (the oracle's stable partition)."""
    import ctypes as C
    import torch
    from presto_amd._lib import check, lib
    g = torch.Generator(device="cuda").manual_seed(n + 1)
    part = torch.randint(0, parts, (n,), dtype=torch.int32, device="cuda", generator=g)
    rowid = torch.arange(n, dtype=torch.int32, device="cuda")
    c8 = torch.randint(-(1 << 60), 1 << 60, (n,), dtype=torch.int64, device="cuda", generator=g)
    c1 = (rowid % 7).to(torch.uint8)
    cols = [rowid, c8, c1]
    outs = [torch.empty_like(c) for c in cols]
    torch.cuda.synchronize()
    vp = C.c_void_p
    ins = (vp * 3)(*[c.data_ptr() for c in cols])
    ous = (vp * 3)(*[c.data_ptr() for c in outs])
    widths = (C.c_int32 * 3)(4, 8, 1)
    counts = np.zeros(parts, dtype=np.int64)
    check(lib().pa_partition_columns_stable(part.data_ptr(), n, parts, ins, ous, widths, 3, counts.ctypes.data, None))
    pos, ocounts = oracle.partition_positions(part.cpu().numpy(), parts)
    assert counts.tolist() == ocounts.tolist()
    assert outs[0].cpu().numpy().tolist() == pos.tolist()
    p = torch.from_numpy(pos).cuda().long()
    assert torch.equal(outs[1], c8[p]) and torch.equal(outs[2], c1[p])


def test_two_sinks_on_two_driver_threads(gpu, oracle):
    """pa_exchange_desc.sink_count = 2: two PartitionedOutput operators, each on a Driver thread and a stream of its own, append to
    the same destination buffers (growing them: reserve_keep copies and recycles a buffer the other sink's copy kernel may still
    be writing to unless the appends are ordered).  Every row arrives once; per sink, in page order."""
    import threading
    from presto_amd.exchange import Comm, Exchange, ExchangeSourceOperator, PartitionedOutputOperator
    comm = Comm.single()
    try:
        for attempt in range(3):
            ex = Exchange(comm, EX_TYPES, [0], sink_count=2)
            sinks = [PartitionedOutputOperator(ex), PartitionedOutputOperator(ex)]
            source = ExchangeSourceOperator(ex, abi.MEM_HOST)
            pages = [exchange_pages(100 + attempt, 12, 30011, nulls_from_page=3), exchange_pages(200 + attempt, 12, 25013, nulls_from_page=5)]
            errors = []

            def drive(sink, mine):
                try:
                    for p in mine:
                        sink.addInput(p)
                    sink.finish()
                except Exception as e:  # surfaces in the main thread
                    errors.append(e)
            assert source.isBlocked()
            threads = [threading.Thread(target=drive, args=(s_, p_)) for s_, p_ in zip(sinks, pages)]
            [t.start() for t in threads]
            [t.join() for t in threads]
            assert not errors, errors
            out = source.getOutput()
            rows = out.to_rows()
            expected = [[r for p in mine for r in p.to_rows()] for mine in pages]
            assert len(rows) == len(expected[0]) + len(expected[1])
            key = lambda r: tuple((x is None, x) for x in r)
            assert sorted(rows, key=key) == sorted(expected[0] + expected[1], key=key)
            # per sink the rows keep their order: the rows of sink k, as a subsequence of the output
            for mine in expected:
                want = set(map(key, mine))
                got = [r for r in rows if key(r) in want]
                if len(want) == len(mine) and not (want & set(map(key, expected[1] if mine is expected[0] else expected[0]))):
                    assert got == mine
            for s_ in sinks:
                s_.close()
            source.close()
            ex.destroy()
    finally:
        comm.destroy()
