"""presto_amd/control.py: the torch-free control plane of bench.py's ranks (a star of local sockets) -- rendezvous in any arrival
order, all_gather / barrier / max / broadcast / all_to_all between four processes, and a rank that never arrives fails the others
instead of hanging them."""
import multiprocessing as mp
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, name, queue, delay):
    sys.path.insert(0, ROOT)
    from presto_amd.control import ControlPlane
    time.sleep(delay)
    c = ControlPlane(rank, world, name=name, timeout=20)
    out = {"gather": c.all_gather({"rank": rank, "blob": bytes([rank]) * (1 << 16)})}
    c.barrier()
    out["max"] = c.all_reduce_max(1.5 * rank)
    out["bcast"] = c.broadcast(b"id-%d" % rank, src=2)
    out["a2a"] = c.all_to_all([b"%d->%d" % (rank, p) for p in range(world)])
    c.barrier()
    c.close()
    assert "torch" not in sys.modules
    queue.put((rank, out))


def test_four_ranks_every_operation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, name = 4, "presto_amd.test.%d" % os.getpid()
    # rank 0 (the listener) arrives LAST: the others retry until it is there
    procs = [ctx.Process(target=worker, args=(r, world, name, q, 0.5 if r == 0 else 0.0)) for r in range(world)]
    [p.start() for p in procs]
    results = dict(q.get(timeout=60) for _ in range(world))
    [p.join(timeout=30) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, out in results.items():
        assert [g["rank"] for g in out["gather"]] == list(range(world)) and all(len(g["blob"]) == 1 << 16 for g in out["gather"])
        assert out["max"] == 1.5 * (world - 1) and out["bcast"] == b"id-2"
        assert out["a2a"] == [b"%d->%d" % (src, rank) for src in range(world)]


def lonely(rank, world, name, queue):
    sys.path.insert(0, ROOT)
    from presto_amd.control import ControlError, ControlPlane
    try:
        ControlPlane(rank, world, name=name, timeout=1.5)
        queue.put("connected")
    except ControlError as e:
        queue.put("refused: %s" % e)


@pytest.mark.parametrize("rank", [0, 1])
def test_a_missing_rank_fails_the_rendezvous_instead_of_hanging(rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=lonely, args=(rank, 2, "presto_amd.test.lonely.%d.%d" % (os.getpid(), rank), q))
    p.start()
    assert q.get(timeout=30).startswith("refused")
    p.join(timeout=10)


def test_single_rank_needs_no_socket():
    from presto_amd.control import ControlPlane
    c = ControlPlane(0, 1)
    assert c.all_gather(7) == [7] and c.all_reduce_max(2.5) == 2.5 and c.broadcast(b"x") == b"x" and c.all_to_all([b"s"]) == [b"s"]
    c.barrier()
    c.close()
