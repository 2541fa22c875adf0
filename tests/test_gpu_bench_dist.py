"""bench.py's distributed code path on one GPU.  Two ways the ranks can meet (bench.py --rendezvous):

  socket (default)  mavflow.rendezvous over a localhost socket -> ncclUniqueId -> mav_comm_init with RCCL and the HIP runtime from
                    /opt/rocm, no torch in the process -> the record all-gather inside every step on the context's stream
                    (mav_allgather_results) -> the gathered block checked against the local records.
  torch             torch.distributed process group (nccl = RCCL) -> the same, with torch loaded first, so that libmavflow.so (built by
                    this tree's hipcc) binds to the HIP runtime and RCCL torch ships -- a version seam nothing else exercises.  Also
                    where the socket path lands when it does not come up on every rank (agreed through the store).

World size 1 with the real communicator; two ranks sharing the one GPU as a functional rehearsal (records exchanged on the host: RCCL
refuses two ranks on one device).  bench.py runs as a CHILD process (it must own its process group and its GPU context)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


SMALL = ["--steps", "2", "--warmup", "1", "--no-profile", "--cpu-pairs", "0", "--no-configs", "--no-api-loop"]


def _env(**kw):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MAVFLOW_BENCH_DIST", "MAVFLOW_RDZV", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(kw)
    return env


def _line(p):
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                             # rank 0 alone prints
    return json.loads(lines[0])


def test_bench_socket_rendezvous_world_size_1():
    """The default path of an N > 1 run, at world size 1: the rank hosts the store itself (as rank 0 does under torch.distributed.run),
    the library opens its communicator with the RCCL under /opt/rocm, and torch never enters the process."""
    env = _env(MAVFLOW_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    d = _line(p)
    cfg = d["config"]
    assert cfg["record_exchange"].startswith("mav_allgather_results") and "mavflow.rendezvous" in cfg["record_exchange"], cfg["record_exchange"]
    assert cfg["comm_ranks"] == 1 and cfg["torch_in_process"] is False
    assert d["verified_pairs"] == [0, 63], d.get("verification")
    rt = cfg["runtime"]
    print("\nruntime (socket path):", rt)
    assert rt["hip_runtime_major"] == int(rt["built_with_hip"].split(".")[0]) and rt["rccl"] > 0


def test_socket_path_falls_back_to_torch_when_a_rank_reports_failure():
    env = _env(MAVFLOW_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--simulate-socket-failure"] + SMALL, env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    d = _line(p)
    assert "falling back to --rendezvous torch" in p.stderr
    assert d["config"]["torch_in_process"] is True and "torch.distributed" in d["config"]["record_exchange"]
    assert d["config"]["comm_ranks"] == 1 and d["verified_pairs"] == [0, 63]


def test_bench_distributed_path_world_size_1():
    env = dict(os.environ, MAVFLOW_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rendezvous", "torch"] + SMALL
    # (the child imports torch: on a freshly started box that alone can take one to two minutes while the image pages in -- the only
    # part of this suite with that kind of variance; bounded here, and pytest.ini makes every run print its slowest tests)
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2
    assert d["config"]["record_exchange"].startswith("mav_allgather_results"), d["config"]["record_exchange"]
    assert d["config"]["comm_ranks"] == 1 and d["config"]["torch_in_process"] is True
    assert d["verified_pairs"] == [0, 63], d.get("verification")
    rt = d["config"]["runtime"]
    print("\nruntime seam:", rt)
    built_major = int(rt["built_with_hip"].split(".")[0])
    assert rt["hip_runtime_major"] == built_major            # mav_comm_init refuses anything else (MAV_ERR_STATE)
    assert rt["rccl"] > 0                                     # the library's communicator really went through RCCL


def test_comm_init_reports_versions(mav):
    from mavflow import _lib
    rt = _lib.runtime_info()
    assert rt["hip_runtime"] > 0 and rt["hip_runtime_major"] == int(rt["built_with_hip"].split(".")[0])


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """The N > 1 path of bench.py with real GPU work in every rank: two processes started by torch.distributed.run exactly as the driver
    starts them, both on device 0 (--rehearse-on-one-gpu: process group over gloo, records through torch's all-gather -- RCCL refuses two
    ranks on one device).  Exercises what a world-size-1 run cannot: per-rank content and sample seeds, the barrier and the
    max-over-ranks time, the gathered blocks of BOTH ranks (own block == local records, the blocks differ), rank 0 printing the one
    line with the whole-job value.  Not a scaling measurement (the ranks share the GPU)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MAVFLOW_BENCH_DIST"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--rehearse-on-one-gpu",
           "--rendezvous", "torch", "--no-profile", "--cpu-pairs", "0", "--no-configs", "--no-api-loop"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                             # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["gathered_rank_blocks_distinct"] == 2
    assert d["verified_pairs"] == [0, 7] and d["verification"]["all_pairs_equal_plain_schedule"]
    assert "rehearsal" in d and d["config"]["record_exchange"].startswith("torch.distributed.all_gather_into_tensor")
    assert abs(d["value"] - 16 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"] + 0.5   # whole-job pairs / max-over-ranks time


@pytest.mark.parametrize("launcher", ["own", "torch.distributed.run"])
def test_bench_two_ranks_rehearsed_through_the_socket_rendezvous(launcher):
    """The same rehearsal on the default (torch-free) path, started both ways: by bench.py's own launcher (the parent hosts the store and
    spawns the ranks) and by torch.distributed.run exactly as the driver starts it (rank 0 hosts the store, the port travels through a
    file named after MASTER_PORT and the launcher's pid).  Ids only: the ncclUniqueId is created and broadcast, the records move
    through the store."""
    import socket
    tail = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--rehearse-on-one-gpu", "--no-profile", "--cpu-pairs", "0",
            "--no-configs", "--no-api-loop"]
    if launcher == "own":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(port), os.path.join(ROOT, "bench.py")] + tail
    p = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=420)
    d = _line(p)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and d["scaling"] == "weak"
    assert d["gathered_rank_blocks_distinct"] == 2
    assert d["verified_pairs"] == [0, 7] and d["verification"]["all_pairs_equal_plain_schedule"]
    assert d["config"]["record_exchange"].startswith("rendezvous store all-gather") and d["config"]["torch_in_process"] is False
    assert abs(d["value"] - 16 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"] + 0.5


@pytest.mark.parametrize("launcher", ["own", "torch.distributed.run"])
def test_two_ranks_agree_to_fall_back_to_the_torch_path(launcher):
    """One of two ranks reports that its communicator did not come up (--simulate-socket-failure): BOTH ranks must leave the socket path,
    re-run in child processes on the torch.distributed path (under the launcher's process group, or, with bench.py's own launcher, the
    MASTER_PORT it reserved for exactly this) and rank 0's child prints the one line."""
    import socket
    tail = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--rehearse-on-one-gpu", "--simulate-socket-failure", "--no-profile",
            "--cpu-pairs", "0", "--no-configs", "--no-api-loop"]
    if launcher == "own":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + tail
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
               str(port), os.path.join(ROOT, "bench.py")] + tail
    p = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=600)
    d = _line(p)
    assert "falling back to --rendezvous torch" in p.stderr
    assert d["n_gpus"] == 2 and d["gathered_rank_blocks_distinct"] == 2 and d["config"]["torch_in_process"] is True
    assert d["config"]["record_exchange"].startswith("torch.distributed.all_gather_into_tensor") and d["verified_pairs"] == [0, 7]
