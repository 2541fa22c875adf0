"""The N > 1 orchestration on CPU: gloo processes (world size 2 and 8) shard a batch of pairs and all-gather their 32-byte
result records as bench.py does over RCCL -- including BASELINE configs 4 and 5's shard tables (512 -> 64 per rank, 128 -> 16 per
rank) and ragged totals.  No GPU, no libmavflow compute."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges():
    from mavflow.dist import shard
    assert [shard(512, r, 8) for r in range(8)] == [(64 * r, 64 * (r + 1)) for r in range(8)]
    parts = [shard(10, r, 4) for r in range(4)]
    assert parts == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [shard(2, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]      # ragged: empty shards allowed
    assert shard(0, 0, 1) == (0, 0)
    with pytest.raises(ValueError):
        shard(4, 4, 4)


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(%r, "mav-detection_amd"))
    from mavflow import dist as mdist
    dist, rank, world, _ = mdist.init_process_group("gloo")
    dt = np.dtype([("box", np.int32, (4,)), ("foe", np.float64, (2,))])
    assert dt.itemsize == mdist.RECORD_BYTES
    total, per = 6, 3
    lo, hi = mdist.shard(total, rank, world)
    rec = np.zeros(per, dt)
    for k, pair in enumerate(range(lo, hi)):
        rec[k]["box"] = (pair, pair + 1, pair + 2, pair + 3)
        rec[k]["foe"] = (pair * 0.5, -pair * 0.25)
    allrec = mdist.allgather_numpy(dist, rec)
    assert allrec.shape == (world * per,)
    for pair in range(total):
        assert tuple(allrec[pair]["box"]) == (pair, pair + 1, pair + 2, pair + 3), (rank, pair, allrec[pair])
        assert tuple(allrec[pair]["foe"]) == (pair * 0.5, -pair * 0.25)
    # max-over-ranks timing reduction used by bench.py
    import torch
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_world_size_2_gloo_allgather(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode(errors="replace"))
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\\n{out}"
        assert f"rank {rank} ok" in out


WORKER8 = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, os.path.join(%r, "mav-detection_amd"))
    from mavflow import dist as mdist
    dist, rank, world, _ = mdist.init_process_group("gloo")
    assert world == 8
    dt = np.dtype([("box", np.int32, (4,)), ("foe", np.float64, (2,))])
    def record(pair):
        r = np.zeros((), dt)
        r["box"] = (pair, 2 * pair, pair + 7, 3 * pair + 1)
        r["foe"] = (pair * 0.5 + 0.25, -pair * 0.125)
        return r
    # C4: 1920x1080, 512 pairs -> 64 per rank;  C5: 3840x2160, 128 pairs -> 16 per rank;  ragged totals incl. empty shards
    for total, per_rank in ((512, 64), (128, 16), (100, None), (5, None), (0, None)):
        lo, hi = mdist.shard(total, rank, world)
        if per_rank is not None:
            assert (lo, hi) == (rank * per_rank, (rank + 1) * per_rank)
        mine = np.array([record(p) for p in range(lo, hi)], dt).reshape(hi - lo)
        allrec = mdist.allgather_pairs(dist, mine, total)
        assert allrec.shape == (total,), (total, allrec.shape)
        want = np.array([record(p) for p in range(total)], dt).reshape(total)
        assert allrec.tobytes() == want.tobytes(), (rank, total)
    if rank == 0:                                       # a shard of the wrong length is refused, not silently padded
        try:
            mdist.allgather_pairs(dist, np.zeros(3, dt), 512)
            raise SystemExit("expected ValueError")
        except ValueError:
            pass
    import torch
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 8.0
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_world_size_8_shard_tables_and_ragged_gather(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % ROOT)
    procs = []
    for rank in range(8):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="8", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode(errors="replace"))
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert f"rank {rank} ok" in out
