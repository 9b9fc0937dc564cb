"""GPU parity of the exchange kernels (row hash, partition id, stable partition, gather) against the oracle, and the
full exchange path on one rank over RCCL (a 1-rank all-to-all is a self copy: every row must come back, grouped)."""
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
    from presto_amd.exchange import DeviceOps
    rng = np.random.default_rng(4)
    n = 123457
    raw = rng.integers(-2 ** 63, 2 ** 63 - 1, n, dtype=np.int64)
    ops = DeviceOps()
    part = ops.partition_ids(torch.from_numpy(raw).cuda(), partitions, local)
    expected = oracle.partition_ids(raw, partitions, local)
    assert np.array_equal(part.cpu().numpy(), expected)
    pos, counts = ops.partition_positions(part, partitions)
    epos, ecounts = oracle.partition_positions(expected, partitions)
    assert counts == ecounts.tolist()
    assert np.array_equal(pos.cpu().numpy(), epos)  # ascending positions inside every partition


def test_exchange_on_one_rank_over_rccl(gpu, oracle):
    import torch.distributed as dist
    from presto_amd.exchange import DeviceOps, exchange_columns
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(6)
        n = 50001
        keys = rng.integers(0, 10 ** 9, n).astype(np.int64)
        vals = rng.random(n)
        cols = [torch.from_numpy(keys).cuda(), torch.from_numpy(vals).cuda()]
        recv, counts = exchange_columns(DeviceOps(), cols, [abi.BIGINT, abi.DOUBLE], [0])
        torch.cuda.synchronize()
        assert counts == [n]
        assert np.array_equal(recv[0].cpu().numpy(), keys) and np.array_equal(recv[1].cpu().numpy(), vals)
    finally:
        dist.destroy_process_group()


def test_varchar_exchange_on_one_rank_over_rccl(gpu, oracle):
    """VARCHAR column + VARCHAR partitioning key through the device ops and a one-rank RCCL group (self copy): rows come back
    grouped by partition, strings intact."""
    import torch.distributed as dist
    from presto_amd.exchange import DeviceOps, exchange_columns
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    torch.zeros(1, device="cuda")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(9)
        n = 40001
        words = [b"", b"a", b"BUILDING", b"0123456789abcdefghijklmnopqrstuvwxyz", b"zz"]
        names = [words[i] + str(int(j)).encode() for i, j in zip(rng.integers(0, len(words), n), rng.integers(0, 1000, n))]
        b = Block.varchar(names)
        keys = rng.integers(0, 10 ** 9, n).astype(np.int64)
        cols = [torch.from_numpy(keys).cuda(), (torch.from_numpy(b.values.copy()).cuda(), torch.from_numpy(b.offsets.copy()).cuda())]
        recv, counts = exchange_columns(DeviceOps(), cols, [abi.BIGINT, abi.VARCHAR], [1])
        torch.cuda.synchronize()
        assert counts == [n]
        rb, ro = recv[1][0].cpu().numpy().tobytes(), recv[1][1].cpu().numpy().tolist()
        got = [(int(k), rb[ro[i]:ro[i + 1]]) for i, k in enumerate(recv[0].cpu().numpy().tolist())]
        assert got == list(zip(keys.tolist(), names))   # one partition: the stable order is the input order
        # the row hash of the string column equals the oracle's (what routes the rows)
        h = DeviceOps().hash_rows(cols, [abi.BIGINT, abi.VARCHAR], [1]).cpu().numpy()
        assert np.array_equal(h, oracle.hash_page(Page([Block.bigint(keys), Block.varchar(names)], n), [1]))
    finally:
        dist.destroy_process_group()


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
