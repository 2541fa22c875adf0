"""mavflow.rendezvous -- the torch-free way the ranks of a multi-GPU run meet -- at world size 2 and 8 on the CPU.
Ids only: a fake 128-byte unique id, the 32-byte-per-pair record blocks, barrier, max-over-ranks.  Both ways a rank finds the store:
MAVFLOW_RDZV from bench.py's own launcher (rendezvous.spawn_ranks) and, as under torch.distributed.run, rank 0 hosting the store and
publishing its port in a file named after MASTER_PORT and the common parent's pid."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, os.path.join(%r, "mav-detection_amd"))
    from mavflow import rendezvous
    from mavflow.dist import shard
    c = rendezvous.from_env(timeout=60)
    rank, world = c.rank, c.world
    uid = c.broadcast("uid", bytes(range(128)) if rank == 0 else None)
    assert uid == bytes(range(128)), uid
    dt = np.dtype([("box", np.int32, (4,)), ("foe", np.float64, (2,))])
    B = 64
    for step in range(3):
        rec = np.zeros(B, dt)
        rec["box"][:, 0] = rank * B + np.arange(B); rec["box"][:, 1] = step
        rec["foe"][:, 0] = 0.5 * (rank * B + np.arange(B))
        blocks = c.allgather("rec", rec.tobytes())
        allrec = np.frombuffer(b"".join(blocks), dt)
        assert allrec.shape == (world * B,)
        assert (allrec["box"][:, 0] == np.arange(world * B)).all() and (allrec["box"][:, 1] == step).all()
        assert (allrec["foe"][:, 0] == 0.5 * np.arange(world * B)).all()
    lo, hi = shard(512, rank, world)
    assert hi - lo == 512 // world
    c.barrier("align")                                    # everybody is here (process start-up skew is behind us) ...
    if rank == world - 1:
        time.sleep(0.4)                                   # ... then one rank straggles: the barrier must hold the others
    t0 = time.monotonic()
    c.barrier("b")
    waited = time.monotonic() - t0
    assert c.allreduce_max("t", 1.0 + rank) == float(world)
    flags = c.allgather("up", b"1" if rank != 1 else b"no: made-up failure on rank 1")
    assert [f == b"1" for f in flags] == [r != 1 for r in range(world)]
    assert c.allgather("empty", b"") == [b""] * world
    try:
        c.get("never-set", 0.2)
        raise SystemExit("expected a timeout")
    except rendezvous.RendezvousError:
        pass
    c.close()
    print("rank", rank, "ok", "waited" if waited > 0.1 else "straggler" if rank == world - 1 else "fast")
""")


def _script(tmp_path):
    p = tmp_path / "worker.py"
    p.write_text(WORKER % ROOT)
    return str(p)


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_started_by_a_launcher_that_publishes_no_address(tmp_path, world):
    """As under `python -m torch.distributed.run`: only RANK / WORLD_SIZE / MASTER_PORT and a common parent."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = _script(tmp_path)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("MAVFLOW_RDZV", None)
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode(errors="replace"))
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert f"rank {rank} ok" in out
    assert sum("waited" in o for o in outs) >= world - 1        # everybody but the straggler sat in the barrier
    from mavflow import rendezvous
    assert not os.path.exists(rendezvous._port_file(str(port)).replace(str(os.getppid()), str(os.getpid())))   # rank 0 removed the port file


@pytest.mark.parametrize("world", [2, 8])
def test_bench_s_own_launcher_hosts_the_store(tmp_path, world):
    from mavflow import rendezvous
    script = _script(tmp_path)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MAVFLOW_RDZV"):
        env.pop(k, None)
    assert rendezvous.spawn_ranks([sys.executable, script], world, env=env) == 0
    bad = tmp_path / "bad.py"
    bad.write_text("import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    assert rendezvous.spawn_ranks([sys.executable, str(bad)], world, env=env) == 3     # the largest exit code comes back


def test_store_values_are_plain_hex_and_a_missing_store_times_out():
    from mavflow import rendezvous
    store = rendezvous.Store()
    host, port = store.start()
    c = rendezvous.Client(host, port, 0, 1, timeout=5, token=store.token)
    c.set("k", b"\x00\xff\x10")
    assert c.get("k") == b"\x00\xff\x10"
    assert c._ask("ADD n 2") == "2" and c._ask("ADD n 3") == "5" and c._ask("BOGUS") == "ERR"
    c.close()
    store.stop()
    with pytest.raises(rendezvous.RendezvousError):
        rendezvous.Client("127.0.0.1", port, 0, 1, timeout=0.3)


def test_store_refuses_connections_without_its_token():
    """the store accepted SET / ADD from any local process.  Every connection now opens with the store's random token
    (handed to the ranks with the address: the launcher's environment or the 0600 port file); anything else is closed unanswered."""
    import socket
    from mavflow import rendezvous
    store = rendezvous.Store()
    host, port = store.start()
    assert len(store.token) == 32 and store.address == f"{host}:{port}:{store.token}"
    with pytest.raises(rendezvous.RendezvousError):
        rendezvous.Client(host, port, 0, 1, timeout=2, token="0" * 32)
    with socket.create_connection((host, port), timeout=2) as sk:             # a raw client that skips the handshake gets nothing done
        sk.sendall(b"SET uid.1 deadbeef\n")
        sk.settimeout(2)
        assert sk.recv(16) == b""                                             # closed, no reply
    ok = rendezvous.Client(host, port, 0, 1, timeout=2, token=store.token)
    with pytest.raises(rendezvous.RendezvousError):
        ok.get("uid.1", 0.2)                                                  # the forged value never landed
    ok.close()
    store.stop()


def test_a_get_may_wait_longer_than_the_connection_timeout_and_a_lost_reply_breaks_the_client():
    """the connection's 120 s socket timeout used to cut a 300 s GET short with a TimeoutError, leaving request and reply
    out of step.  A request now gives the socket its own wait + a margin; a reply that still does not come raises RendezvousError and
    the client refuses further use."""
    import threading
    import time
    from mavflow import rendezvous
    store = rendezvous.Store()
    host, port = store.start()
    a = rendezvous.Client(host, port, 0, 2, timeout=0.5, token=store.token)   # connection timeout 0.5 s ...
    b = rendezvous.Client(host, port, 1, 2, timeout=0.5, token=store.token)
    threading.Timer(1.5, lambda: b.set("late", b"\x07")).start()
    t0 = time.monotonic()
    assert a.get("late", timeout=5.0) == b"\x07"                              # ... and a GET that is answered after 1.5 s
    assert 1.0 < time.monotonic() - t0 < 4.0
    with pytest.raises(rendezvous.RendezvousError, match="timed out"):
        a.get("never", timeout=0.2)                                           # the store's own TIMEOUT reply: client still in step
    assert a.get("late") == b"\x07"
    a.close(); b.close(); store.stop()
    # a store that accepts the token and then falls silent: the reply never comes
    import socket
    srv = socket.socket(); srv.bind(("127.0.0.1", 0)); srv.listen(1)
    conns = []

    def mute():
        cn, _ = srv.accept()
        conns.append(cn)
        cn.recv(200)
        cn.sendall(b"OK\n")                                                    # the AUTH reply, nothing afterwards
    th = threading.Thread(target=mute, daemon=True); th.start()
    c = rendezvous.Client("127.0.0.1", srv.getsockname()[1], 0, 1, timeout=0.2, token="x")
    c.timeout = -9.7                                                          # a request's wait + the 10 s margin = 0.3 s
    t0 = time.monotonic()
    with pytest.raises(rendezvous.RendezvousError, match="no reply"):
        c._ask("SET k 00")
    assert time.monotonic() - t0 < 2.0
    with pytest.raises(rendezvous.RendezvousError, match="out of step"):
        c.get("k", 0.1)
    for cn in conns:
        cn.close()
    srv.close()
