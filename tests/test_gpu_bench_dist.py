"""bench.py's distributed code path on one GPU: torch.distributed process group (nccl = RCCL) at world size 1 -> ncclUniqueId
broadcast -> mav_comm_init -> the record all-gather inside every step on the context's stream (mav_allgather_results) -> the
gathered block checked against the local records.  This is the path the driver's 1/2/4/8-GPU scaling run takes; with torch loaded
first libmavflow.so (built by this tree's hipcc) binds to the HIP runtime and RCCL torch ships, a version seam nothing else
exercises.  Runs bench.py as a CHILD process (it must own its process group and its GPU context)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_distributed_path_world_size_1():
    env = dict(os.environ, MAVFLOW_BENCH_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-profile", "--cpu-pairs", "0", "--no-configs"]
    # (the child imports torch: on a freshly started box that alone can take one to two minutes while the image pages in -- the only
    # part of this suite with that kind of variance; bounded here, and pytest.ini makes every run print its slowest tests)
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2
    assert d["config"]["record_exchange"].startswith("mav_allgather_results"), d["config"]["record_exchange"]
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
           "--no-profile", "--cpu-pairs", "0", "--no-configs"]
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
