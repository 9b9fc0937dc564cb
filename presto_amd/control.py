"""Control plane of the ranks of ONE node, without torch: rendezvous, barrier, small reductions and broadcasts over a local socket.

Why not torch.distributed: `import torch` puts the torch wheel's own ROCm stack (libamdhip64.so, libhsa-runtime64.so, librccl.so,
libhiprtc.so under torch/lib) into the process's global symbol scope, and a library loaded afterwards binds its hip* / nccl* calls to
THOSE copies -- a rank would run the device path on another runtime than the one it was built and profiled with (/opt/rocm).  A rank
process of bench.py therefore never imports torch: the data plane is the library's pa_comm (RCCL over xGMI), and what is left --
a barrier around the timed region, the max over the ranks' clocks, shipping the 128-byte RCCL unique id -- needs a few hundred bytes
per call between processes of one node.

Topology: a star.  Rank 0 listens on an abstract unix socket named after MASTER_ADDR / MASTER_PORT (what torchrun exports; the name
lives in no file system), the other ranks connect to it.  Every operation is collective: each rank sends one message to rank 0, rank 0
answers every rank when all have arrived.  Messages are length-prefixed pickles.  A rank that does not arrive within `timeout` seconds
fails the operation on every rank that waits for it (socket timeout) instead of hanging the job."""
import os
import pickle
import socket
import struct
import time


class ControlError(RuntimeError):
    pass


def _send(sock, obj):
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<q", len(blob)) + blob)


def _recv(sock):
    head = _recv_exact(sock, 8)
    (n,) = struct.unpack("<q", head)
    return pickle.loads(_recv_exact(sock, n))


def _recv_exact(sock, n):
    parts = []
    while n > 0:
        chunk = sock.recv(min(n, 1 << 20))
        if not chunk:
            raise ControlError("control plane: a peer closed its connection")
        parts.append(chunk)
        n -= len(chunk)
    return b"".join(parts)


class ControlPlane:
    """rank / world from the launcher's environment (RANK, WORLD_SIZE); collective calls only."""

    def __init__(self, rank=None, world=None, name=None, timeout=300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.timeout = timeout
        self.peers = []     # rank 0: socket of rank r at peers[r - 1]
        self.sock = None    # other ranks: the connection to rank 0
        if self.world == 1:
            return
        if name is None:
            name = "presto_amd.%s.%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"))
        address = "\0" + name   # abstract namespace: gone with the last socket, nothing to clean up
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            server = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            server.bind(address)
            server.listen(self.world)
            server.settimeout(timeout)
            by_rank = {}
            try:
                while len(by_rank) < self.world - 1:
                    conn, _ = server.accept()
                    conn.settimeout(timeout)
                    by_rank[_recv(conn)] = conn
            except socket.timeout:
                raise ControlError("control plane: %d of %d ranks arrived within %g s" % (len(by_rank) + 1, self.world, timeout))
            finally:
                server.close()
            self.peers = [by_rank[r] for r in range(1, self.world)]
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(address)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise ControlError("control plane: rank 0 did not open %r within %g s" % (name, timeout))
                    time.sleep(0.01)
            s.settimeout(timeout)
            _send(s, self.rank)
            self.sock = s

    # ---- the one primitive: every rank contributes a value, every rank gets the list in rank order ----
    def all_gather(self, value):
        if self.world == 1:
            return [value]
        try:
            if self.rank == 0:
                values = [value] + [_recv(p) for p in self.peers]
                for p in self.peers:
                    _send(p, values)
                return values
            _send(self.sock, value)
            return _recv(self.sock)
        except socket.timeout:
            raise ControlError("control plane: a rank did not arrive within %g s" % self.timeout)

    def barrier(self):
        self.all_gather(None)

    def all_reduce_max(self, x):
        return max(self.all_gather(x))

    def broadcast(self, value, src=0):
        return self.all_gather(value if self.rank == src else None)[src]

    def all_to_all(self, blobs):
        """blobs[p] = bytes for rank p; returns the list of what every rank sent to this one, in rank order (relayed by rank 0: the
        rehearsal transport of ranks that share one GPU, not a data path)."""
        everything = self.all_gather(list(blobs))
        return [everything[src][self.rank] for src in range(self.world)]

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None


class TorchControlPlane:
    """The same surface over an initialised torch.distributed group (CPU ranks of the tests, whose checker workloads use
    torch.distributed themselves)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_gather(self, value):
        out = [None] * self.world
        self.dist.all_gather_object(out, value, group=self.group)
        return out

    def barrier(self):
        self.dist.barrier(group=self.group)

    def all_reduce_max(self, x):
        return max(self.all_gather(x))

    def broadcast(self, value, src=0):
        return self.all_gather(value if self.rank == src else None)[src]

    def all_to_all(self, blobs):
        everything = self.all_gather(list(blobs))
        return [everything[src][self.rank] for src in range(self.world)]

    def close(self):
        pass
