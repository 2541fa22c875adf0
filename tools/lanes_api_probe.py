"""The one-frame loop (Processor.run_detection, Farneback seam) with 1 / 2 / 3 lanes, same process, interleaved repetitions.
    python tools/lanes_api_probe.py [W H frames reps]"""
import sys, logging, time
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig
a = sys.argv[1:]
W, H, F, REPS = (int(a[0]), int(a[1]), int(a[2]), int(a[3])) if len(a) >= 4 else (1920, 1080, 96, 4)
N = F + 1
LANES = tuple(int(v) for v in a[4].split(",")) if len(a) >= 5 else (1, 2, 3)
procs = {}
for lanes in LANES:
    ds = SyntheticDataset(W, H, N, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001), lanes=lanes)
    for i in range(8):
        ds._pair(i); ds.get_gt_of(i)
    for _ in range(N):
        ds.get_frame()
    ds.get_segmentation(0); ds.get_sky_segmentation(0); ds.get_depth(0)
    p = Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
    p.run_detection()                                      # warm: contexts, workspaces, slots
    procs[lanes] = (p, ds)
for rep in range(REPS):
    row = []
    for lanes in LANES:
        p, ds = procs[lanes]
        p.frame_index = 0; p.detection_results = {}; p.config.results = {}
        np.random.seed(7)
        c0 = time.thread_time(); t0 = time.perf_counter(); p.run_detection(); dt = time.perf_counter() - t0; cpu = time.thread_time() - c0
        row.append(f"lanes {lanes}: {1e3 * dt / F:.4f} (loop thread busy {1e3 * cpu / F:.4f})")
    print(f"{W}x{H}, {F} frames, ms per frame   " + "   ".join(row), flush=True)
for p, ds in procs.values():
    p.release()
