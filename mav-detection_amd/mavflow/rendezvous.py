"""Torch-free rendezvous for the frame-parallel multi-GPU run (the reference has no counterpart -- it is one process).

What N ranks on one node have to agree on is tiny: the 128-byte ncclUniqueId of the record all-gather, a barrier on both sides of the
timed region, the maximum of N elapsed times, and whether everybody got that far.  This module does it over a localhost TCP socket
with the Python standard library, so that a rank loads exactly one HIP runtime and one RCCL -- the ones under /opt/rocm that
libmavflow.so was built against -- instead of the copies bundled with torch (same SONAMEs, other versions; whichever is mapped first
wins).  torch.distributed stays available as the fallback path of bench.py.

    Store           a key-value server thread (rank 0, or the parent that spawns the ranks): SET / GET (blocking) / ADD (counter).
    Client          one per rank: set / get / barrier / allreduce_max / allgather of small byte strings.
    from_env()      how a rank finds the store:
                      MAVFLOW_RDZV=host:port   (bench.py's own launcher: the parent hosts the store before any child starts)
                      or, under `python -m torch.distributed.run`, rank 0 hosts it on an ephemeral port and publishes the port in
                      a file named after MASTER_PORT and the launcher's pid (every worker's parent), which the other ranks poll.

Protocol: one request per line, `SET key hex`, `GET key timeout_s`, `ADD key n`; one reply per line (`OK`, the hex value, the new
count, or `TIMEOUT`).  Values are hex strings: no framing, no pickling, nothing executable crosses the socket.  The first line of every
connection is `AUTH token`: the store draws a random token when it starts and hands it to the ranks with its address (the launcher's
environment, or the 0600 port file), so that another user's process on the node can neither publish a forged ncclUniqueId or elapsed
time nor flip the "communicator came up" flags; a connection that does not present it is closed.
"""
from __future__ import annotations

import os
import secrets
import socket
import socketserver
import tempfile
import threading
import time
from typing import Dict, List, Optional


class Store:
    """The key-value server.  start() -> (host, port); stop() closes it.  Thread per connection (N <= a few dozen ranks)."""

    def __init__(self, host: str = "127.0.0.1", port: int = 0):
        data: Dict[str, str] = {}
        cond = threading.Condition()
        token = self.token = secrets.token_hex(16)

        class Handler(socketserver.StreamRequestHandler):
            def handle(self):
                first = self.rfile.readline().decode("ascii", "replace").split()
                if len(first) != 2 or first[0] != "AUTH" or not secrets.compare_digest(first[1], token):
                    return                                # not one of ours: no reply, connection closed
                self.wfile.write(b"OK\n")
                self.wfile.flush()
                for raw in self.rfile:
                    parts = raw.decode("ascii", "replace").split()
                    if not parts:
                        continue
                    cmd = parts[0]
                    if cmd == "SET" and len(parts) == 3:
                        with cond:
                            data[parts[1]] = parts[2]
                            cond.notify_all()
                        reply = "OK"
                    elif cmd == "GET" and len(parts) == 3:
                        deadline = time.monotonic() + float(parts[2])
                        with cond:
                            while parts[1] not in data:
                                left = deadline - time.monotonic()
                                if left <= 0:
                                    break
                                cond.wait(left)
                            reply = data.get(parts[1], "TIMEOUT")
                    elif cmd == "ADD" and len(parts) == 3:
                        with cond:
                            v = int(data.get(parts[1], "0")) + int(parts[2])
                            data[parts[1]] = str(v)
                            cond.notify_all()
                        reply = str(v)
                    else:
                        reply = "ERR"
                    self.wfile.write((reply + "\n").encode("ascii"))
                    self.wfile.flush()

        class Server(socketserver.ThreadingTCPServer):
            allow_reuse_address = True
            daemon_threads = True

        self._server = Server((host, port), Handler)
        self.host, self.port = self._server.server_address[:2]
        self._thread = threading.Thread(target=self._server.serve_forever, kwargs={"poll_interval": 0.05}, daemon=True)

    def start(self):
        self._thread.start()
        return self.host, self.port

    @property
    def address(self) -> str:
        """host:port:token -- what a rank needs to join (MAVFLOW_RDZV, the port file)"""
        return f"{self.host}:{self.port}:{self.token}"

    def stop(self):
        self._server.shutdown()
        self._server.server_close()


class RendezvousError(RuntimeError):
    pass


class Client:
    def __init__(self, host: str, port: int, rank: int, world: int, timeout: float = 120.0, store: Optional[Store] = None, token: str = ""):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self._broken = False
        self._store = store                               # kept alive (and stopped) by the rank that hosts it
        deadline = time.monotonic() + self.timeout
        while True:
            try:
                self._sock = socket.create_connection((host, port), timeout=self.timeout)
                break
            except OSError:
                if time.monotonic() > deadline:
                    raise RendezvousError(f"rank {rank}: cannot reach the rendezvous store at {host}:{port}")
                time.sleep(0.05)
        self._sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self._file = self._sock.makefile("rwb")
        self._epoch = 0
        if self._ask(f"AUTH {token or (store.token if store is not None else '-')}") != "OK":
            raise RendezvousError(f"rank {rank}: the rendezvous store refused the token")

    def _ask(self, line: str, wait: Optional[float] = None) -> str:
        """One request, one reply.  `wait`: how long the STORE may hold the reply back (a blocking GET); the socket gives it that long
        plus a margin, whatever the connection's own timeout is.  A reply that does not come leaves request and reply out of step:
        the client is unusable from then on and says so."""
        if self._broken:
            raise RendezvousError(f"rank {self.rank}: the rendezvous connection is out of step after a timeout")
        try:
            self._sock.settimeout((self.timeout if wait is None else max(wait, 0.0)) + 10.0)
            self._file.write((line + "\n").encode("ascii"))
            self._file.flush()
            reply = self._file.readline()
        except (socket.timeout, TimeoutError):
            self._broken = True
            raise RendezvousError(f"rank {self.rank}: no reply from the rendezvous store to '{line.split()[0]} {line.split()[1] if ' ' in line else ''}'")
        if not reply:
            raise RendezvousError(f"rank {self.rank}: the rendezvous store closed the connection")
        return reply.decode("ascii").strip()

    def set(self, key: str, value: bytes) -> None:
        if self._ask(f"SET {key} {value.hex() or '-'}") != "OK":
            raise RendezvousError(f"rank {self.rank}: SET {key} refused")

    def get(self, key: str, timeout: Optional[float] = None) -> bytes:
        t = self.timeout if timeout is None else float(timeout)
        r = self._ask(f"GET {key} {t}", wait=t)
        if r == "TIMEOUT":
            raise RendezvousError(f"rank {self.rank}: timed out waiting for '{key}'")
        return b"" if r == "-" else bytes.fromhex(r)

    def allgather(self, name: str, value: bytes, timeout: Optional[float] = None) -> List[bytes]:
        """Every rank's value, in rank order.  `name` must be used once per collective (an epoch counter makes repeated calls distinct)."""
        self._epoch += 1
        base = f"{name}.{self._epoch}"
        self.set(f"{base}.{self.rank}", value)
        return [self.get(f"{base}.{r}", timeout) for r in range(self.world)]

    def barrier(self, name: str = "barrier", timeout: Optional[float] = None) -> None:
        self.allgather(name, b"\x01", timeout)

    def allreduce_max(self, name: str, x: float) -> float:
        return max(float(v.decode("ascii")) for v in self.allgather(name, repr(float(x)).encode("ascii")))

    def broadcast(self, name: str, value: Optional[bytes], src: int = 0) -> bytes:
        self._epoch += 1
        key = f"{name}.{self._epoch}"
        if self.rank == src:
            self.set(key, value)
            return value
        return self.get(key)

    def close(self) -> None:
        """Leave.  The rank that hosts the store goes last: it waits (bounded) until every other rank has said goodbye, so that nobody's
        final replies are cut off."""
        try:
            if self._store is None:
                self.set(f"bye.{self.rank}", b"\x01")
            else:
                for r in range(self.world):
                    if r != self.rank:
                        try:
                            self.get(f"bye.{r}", 30.0)
                        except RendezvousError:
                            pass
            self._file.close()
            self._sock.close()
        except OSError:
            pass
        finally:
            if self._store is not None:
                self._store.stop()
                self._store = None


def _port_file(master_port: str) -> str:
    return os.path.join(tempfile.gettempdir(), f"mavflow_rdzv_{os.getuid()}_{master_port}_{os.getppid()}")


def from_env(timeout: float = 120.0) -> Client:
    """The rank's client, from RANK / WORLD_SIZE and either MAVFLOW_RDZV=host:port or (under torch.distributed.run) MASTER_PORT + the
    launcher's pid.  In the second case rank 0 hosts the store; the file that publishes its port is removed once every rank has
    connected."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    where = os.environ.get("MAVFLOW_RDZV")
    if where:
        host, port, token = (where.split(":") + [""])[:3]
        return Client(host, int(port), rank, world, timeout, token=token)
    path = _port_file(os.environ.get("MASTER_PORT", "0"))
    if rank == 0:
        store = Store()
        host, port = store.start()
        tmp = f"{path}.{os.getpid()}"
        try:
            os.unlink(tmp)
        except OSError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)   # ours, fresh, no symlink followed
        with os.fdopen(fd, "w") as f:
            f.write(store.address)
        os.replace(tmp, path)                             # atomic: a reader never sees a half-written file
        c = Client(host, port, rank, world, timeout, store=store)
        try:
            c.barrier("connected", timeout)
        finally:
            try:
                os.unlink(path)
            except OSError:
                pass
        return c
    deadline = time.monotonic() + timeout
    while True:
        try:
            if os.lstat(path).st_uid != os.getuid():      # the temporary directory is shared: only a file of our own is believed
                raise OSError("rendezvous port file is not ours")
            with open(path) as f:
                host, port, token = f.read().strip().split(":")
            if host != "127.0.0.1":
                raise ValueError("rendezvous store must be on localhost")
            break
        except (OSError, ValueError):
            if time.monotonic() > deadline:
                raise RendezvousError(f"rank {rank}: rank 0 never published the rendezvous port ({path})")
            time.sleep(0.02)
    c = Client(host, int(port), rank, world, timeout, token=token)
    c.barrier("connected", timeout)
    return c


def spawn_ranks(argv: List[str], n: int, env: Optional[dict] = None) -> int:
    """bench.py's own launcher: host the store HERE (this process never touches the GPU), start n ranks of `argv` with RANK / LOCAL_RANK /
    WORLD_SIZE / MAVFLOW_RDZV set, relay their output, return the largest exit code."""
    import subprocess
    store = Store()
    host, port = store.start()
    with socket.socket() as sk:                           # a free port for torch.distributed, should the ranks fall back to it
        sk.bind(("127.0.0.1", 0))
        master_port = sk.getsockname()[1]
    procs = []
    try:
        for r in range(n):
            e = dict(os.environ if env is None else env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MAVFLOW_RDZV=store.address,
                     MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port),
                     HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen(argv, env=e))
        codes = [p.wait() for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        store.stop()
    return max(abs(c) for c in codes)
