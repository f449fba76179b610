"""Launcher plumbing for one-process-per-GPU runs WITHOUT torch: a tiny TCP rendezvous that ships the 128-byte RCCL unique id
from rank 0 to the other ranks and provides the host barrier / max / gather a benchmark needs around its timed region.

It replaces what the reference gets from `mpirun` + `MpiComm::new` (src/parallel/mpi_comm.rs:49-55): the data path never goes
through here (inner products and halo planes travel over RCCL inside libkryst_hip.so).

Environment (the variables `python -m torch.distributed.run` and most launchers export): RANK, WORLD_SIZE, LOCAL_RANK,
MASTER_ADDR, MASTER_PORT.  The rendezvous listens on MASTER_PORT + KRYST_RDZV_PORT_OFFSET (default 1) and the next few ports if
that one is taken; every message starts with a token derived from MASTER_PORT / TORCHELASTIC_RUN_ID so that a foreign
listener on one of those ports is recognised and skipped.
"""
import json
import os
import socket
import struct
import time


def _send(sock, obj):
    data = json.dumps(obj).encode()
    sock.sendall(struct.pack("!Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf.extend(chunk)
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("!Q", _recv_exact(sock, 8))
    return json.loads(_recv_exact(sock, n).decode())


class Rendezvous:
    """All ranks construct it with the same (world, addr, port, token); every method is collective (call it on every rank)."""

    TRIES = 8                      # consecutive ports tried when the first one is taken

    def __init__(self, rank, world, addr="127.0.0.1", port=29511, token="kryst", timeout=600.0):
        self.rank, self.world, self.token = rank, world, token
        self.peers = []            # rank 0: sockets of ranks 1..world-1 (index r-1); others: [socket to rank 0]
        if world == 1:
            return
        deadline = time.time() + timeout
        if rank == 0:
            srv = None
            for k in range(self.TRIES):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((addr, port + k))
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                raise RuntimeError(f"rendezvous: no free port in [{port}, {port + self.TRIES})")
            srv.listen(world)
            srv.settimeout(1.0)
            slots = [None] * (world - 1)
            while any(p is None for p in slots):
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: not every rank connected")
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                c.settimeout(2.0)                           # a stale or foreign connection that says nothing must not block the accept
                try:                                        # loop: the long timeout is set only once the hello has been validated
                    hello = _recv(c)
                except Exception:
                    c.close()
                    continue
                if hello.get("token") != token or not (1 <= hello.get("rank", 0) < world) or slots[hello["rank"] - 1] is not None:
                    c.close()
                    continue
                c.settimeout(timeout)
                _send(c, {"token": token, "ok": True})
                slots[hello["rank"] - 1] = c
            srv.close()
            self.peers = slots
        else:
            sock = None
            while sock is None:
                if time.time() > deadline:
                    raise TimeoutError("rendezvous: rank 0 not reachable")
                for k in range(self.TRIES):
                    try:
                        s = socket.create_connection((addr, port + k), timeout=2.0)
                        s.settimeout(2.0)                   # a foreign listener on this port does not answer the hello
                        _send(s, {"token": token, "rank": rank})
                        if _recv(s).get("token") == token:
                            s.settimeout(timeout)
                            sock = s
                            break
                        s.close()
                    except Exception:
                        continue
                if sock is None:
                    time.sleep(0.05)
            self.peers = [sock]

    @classmethod
    def from_env(cls, timeout=600.0):
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", "29500"))
        port = base + int(os.environ.get("KRYST_RDZV_PORT_OFFSET", "1"))
        token = f"kryst-{base}-{os.environ.get('TORCHELASTIC_RUN_ID', '')}"
        return cls(rank, world, addr, port, token, timeout)

    # every collective is "gather to rank 0, combine, send the result back"
    def _collective(self, value, combine):
        if self.world == 1:
            return combine([value])
        if self.rank == 0:
            vals = [value] + [_recv(p) for p in self.peers]
            out = combine(vals)
            for p in self.peers:
                _send(p, out)
            return out
        _send(self.peers[0], value)
        return _recv(self.peers[0])

    def broadcast_bytes(self, payload):
        """payload: bytes on rank 0 (ignored elsewhere) -> the same bytes on every rank."""
        return bytes.fromhex(self._collective(payload.hex() if self.rank == 0 else None, lambda v: v[0]))

    def barrier(self):
        self._collective(0, lambda v: 0)

    def allreduce_max(self, x):
        return self._collective(float(x), lambda v: max(v))

    def gather(self, obj):
        """JSON-serialisable obj from every rank -> the list [rank 0's, rank 1's, ...] on every rank."""
        return self._collective(obj, lambda v: v)

    def close(self):
        for p in self.peers:
            try:
                p.close()
            except Exception:
                pass
        self.peers = []
