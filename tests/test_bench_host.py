"""Host side of bench.py that needs no GPU: the ranks of an N-GPU run pin themselves to disjoint core sets, the BLAS pool is capped
before numpy loads, and the bench line stays a numbers-only record."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = """
import os, sys, json
sys.path.insert(0, {root!r})
import bench
before = sorted(os.sched_getaffinity(0))
usable = bench.usable_cores()
n = bench.pin_rank_to_its_cores(int(os.environ["LOCAL_RANK"]), int(os.environ["LOCAL_WORLD_SIZE"]))
print(json.dumps({{"n": n, "before": before, "after": sorted(os.sched_getaffinity(0)), "usable": usable,
                  "blas": os.environ.get("OPENBLAS_NUM_THREADS")}}))
"""


@pytest.mark.parametrize("world", [1, 2, 8])
def test_ranks_pin_themselves_to_disjoint_core_sets(world, tmp_path):
    """eight ranks on one node share its host cores (on this pool: one cgroup quota); each rank's threads are confined to
    usable // ranks cores of its own before anything touches the GPU.  World size 8 on whatever cores this machine has."""
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    script = tmp_path / "pin.py"
    script.write_text(SCRIPT.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    outs = []
    for r in range(world):
        e = dict(env, LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(world))
        outs.append(json.loads(subprocess.check_output([sys.executable, str(script)], env=e, text=True).strip().splitlines()[-1]))
    avail = outs[0]["before"]
    if world == 1:
        assert outs[0]["after"] == avail and outs[0]["n"] == outs[0]["usable"] >= 1           # a single rank keeps every core
        return
    per = max(1, min(len(avail), outs[0]["usable"]) // world)
    seen = set()
    for r, o in enumerate(outs):
        assert o["n"] == len(o["after"]) == per and set(o["after"]) <= set(avail), (r, o)
        if len(avail) >= world:
            assert not (seen & set(o["after"])), (r, "core sets overlap")
        seen |= set(o["after"])
    assert 1 <= int(outs[0]["blas"]) <= max(1, (os.cpu_count() or 2) // 2)                     # capped before numpy is imported


def test_compact_line_carries_numbers_not_prose():
    """the line on stdout is a numbers-only record (the driver keeps its tail); api_loop and the roofline come before
    the per-configuration legs; the prose lives in the --detail file."""
    sys.path.insert(0, ROOT)
    import bench
    roof = {"bound": "hbm", "served_by": "infinity_cache", "achieved": 5700.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.71, "achieved_from_trace": 5400.0,
            "frac_from_trace": 0.675, "traced_step_ms": 26.6, "untraced_step_ms": 23.6, "traffic": 83900000, "alg_bytes_per_launch_avg": 79965388,
            "avg_launch_ms": 0.0265, "launches_per_step": 1440, "kernel_busy_ms": 20.2, "launches_in_flight": 1.6, "frac_of_measured_ceiling": 0.78,
            "kernel_share_of_step": 0.84, "device_busy_ms": 25.4, "achieved_is": "x" * 700, "note": "y" * 500, "traffic_source": "z" * 300,
            "measured_ceiling": {"infinity_cache_GBs": 7200.0, "hbm_GBs": 4600.0, "kernel": "k" * 60}, "all_kernels_ms": {"a": 1.0}}
    leg = {"workload": "w" * 200, "ms_per_call_hip_events": 0.3, "ms_per_pair": 0.3, "pairs_per_s": 3300.0, "frac": 0.42, "p99_ms": 0.33, "max_ms": 0.35,
           "median_ms": 0.31, "latency_calls": 1200, "frac_is": "f" * 100, "schedule": {"x": list(range(200))}, "roofline": roof, "verified_pairs": [0],
           "failed_pairs": [], "flow_epe_px": {"mean": 1e-6, "p99.9": 1e-4, "max": 1e-3, "against": "oracle", "pairs": 1},
           "lanes": {"is": "i" * 300, "1_context": {"ms_per_pair": 0.3, "frac": 0.42}, "records_identical_across_contexts": True}}
    full = {"metric": "frame-pairs/sec at 1920x1080", "value": 2700.0, "unit": "frame-pairs/s", "n_gpus": 1, "steps": 20, "warmup": 3, "ms_per_step": 23.6,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1920x1080, batch=64 frame pairs per GPU, ...", "global_batch": 64, "parallelism": "frame-parallel x1", "record_exchange": "none (1 GPU)",
                       "comm_ranks": None, "torch_in_process": False, "host_cores_per_rank": 16, "schedule": {"x": list(range(300))}, "runtime": {"rccl": 0}},
            "device_busy_ms": 25.4, "roofline": roof, "source_hash": "abc",
            "cpu_baseline": {"value": 1.5, "unit": "frame-pairs/s", "cores": 1, "kind": "port", "sample": "16 of the pairs, " + "s" * 300,
                             "all_cores": {"value": 19.0, "cores": 16, "host_cores": 256, "pairs": 128, "seconds": 6.7, "kind": "port"}},
            "api_loop": {"workload": "w" * 300, "run_detection": {"flow_seam": "s" * 50, "frames": 96, "ms_per_frame": 0.44, "lanes": 3}, "batched_and_unbatched_results_identical": True},
            "verified_pairs": [0, 63], "verification": {"failed_pairs": [], "all_pairs_equal_plain_schedule": True, "checked": "c" * 300,
                                                         "flow_epe_px": {"mean": 1e-6, "p99.9": 1e-4, "max": 1e-3, "against": "oracle", "pairs": 2}},
            "configs": {"C2": leg, "C5_share": dict(leg)}}
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 6000
    keys = list(line)
    assert keys.index("roofline") < keys.index("cpu_baseline") < keys.index("api_loop") < keys.index("configs")
    assert line["roofline"]["achieved_from_trace"] == 5400.0 and line["roofline"]["traced_step_ms"] == 26.6 and line["device_busy_ms"] == 25.4
    assert line["configs"]["C2"]["p99_ms"] == 0.33 and line["configs"]["C2"]["max_ms"] == 0.35 and "schedule" not in line["configs"]["C2"]
    assert not any(len(v) > 160 for v in _strings(line)), "prose belongs in the detail file"
    assert line["cpu_baseline"]["all_cores"]["cores"] == 16 and line["config"]["host_cores_per_rank"] == 16


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, (list, tuple)):
        for v in o:
            yield from _strings(v)
