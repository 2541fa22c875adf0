"""Soak of the reference-shaped loops: many frames through Processor.run_detection and run_detection_batched while watching the host's
resident set, the GPU's free memory and the idle page-locked pool -- a leak in the handle / slot / marker plumbing shows as growth.
    python tools/api_loop_soak.py [frames] [W H] [lane stream priority] [lanes]"""
import sys, logging, time, resource
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1280, 720)
if len(sys.argv) > 4:
    from mavflow import pipeline
    pipeline.LANE_STREAM_PRIORITY = int(sys.argv[4])
LANES = int(sys.argv[5]) if len(sys.argv) > 5 else None


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


probe = _lib.Context(64, 64, 1)
ok = True
for loop, batch in (("run_detection", 1), ("run_detection_batched", 16)):
    ds = SyntheticDataset(W, H, 2, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001), lanes=LANES)
    for i in range(8):
        ds._pair(i); ds.get_gt_of(i)
    ds._bgr = {0: np.zeros((H, W, 3), np.uint8)}
    ds.get_frame = lambda: ds._bgr[0]                      # (the BGR frame is not part of the path; one array for all)
    p = Processor(RunConfig(logging.getLogger("soak"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
    marks = []
    chunk = max(batch * 4, F // 8)
    done = 0
    while done < F:
        ds.N = chunk + 1
        p.frame_index = 0; p.detection_results = {}; p.config.results = {}; p.detection_boxes = {}
        t0 = time.perf_counter()
        (p.run_detection_batched(batch=batch) if batch > 1 else p.run_detection())
        dt = time.perf_counter() - t0
        done += chunk
        held = np.asarray(p.estimate_fixed).sum()            # read a handle now and then: the lazy path must keep working
        marks.append((done, rss_mb(), probe.mem_info()["dev_free"] / 2**20, _lib._pinned.idle_bytes / 2**20, 1e3 * dt / chunk))
    for m in marks:
        print(f"{loop:22s} after {m[0]:6d} frames: max RSS {m[1]:8.1f} MB   GPU free {m[2]:9.1f} MB   idle pinned {m[3]:7.1f} MB   {m[4]:.4f} ms per frame", flush=True)
    grow_rss = marks[-1][1] - marks[1][1]
    grow_gpu = marks[1][2] - marks[-1][2]
    # a LEAK takes memory chunk after chunk.  What these loops show instead, once per process and at a random time (3 .. 40 s in;
    # 400 000 frames: one step, then flat for 200 000 more -- profiles/r06/api_loop_soak_720p.txt): the HIP runtime grows one of its
    # own pools by 188 MB of host memory and 2 MB of device memory.  So: at most ONE chunk after the first may raise the resident set
    # by more than 1 MB, and never by more than 256 MB in all.
    rises = sum(1 for a, b in zip(marks[1:], marks[2:]) if b[1] - a[1] > 1.0)
    print(f"{loop}: RSS growth after the first chunk {grow_rss:.1f} MB in {rises} step(s), GPU memory taken after the first chunk {grow_gpu:.1f} MB", flush=True)
    ok = ok and rises <= 1 and grow_rss < 256 and grow_gpu < 64
    p.release()
probe.close()
assert ok, "a loop keeps taking memory"
print("soak ok")
