"""Do the lanes of the one-frame loop get hardware queues of their own?  (profiles/r06/lane_queue_priority.txt; run on the GPU box)

    python tools/lane_queue_probe.py MODE [lane stream priority: -1 | 0] [lanes] [W H]

Processor.run_detection (Farneback seam, frames on the host) over 2 000 frames, three repetitions in one process (the first warms up).
MODE plain: nothing else in the process; probe_first: ONE idle 64 x 64 context is created before anything else (what tools/api_loop_soak.py
does to watch the GPU's free memory) -- in the default priority class that context's stream shifts the queue assignment of the lanes."""
import sys, logging, time
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib, pipeline
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig
a = sys.argv[1:]
mode = a[0] if a else "plain"
if len(a) > 1:
    pipeline.LANE_STREAM_PRIORITY = int(a[1])
lanes = int(a[2]) if len(a) > 2 else None
W, H = (int(a[3]), int(a[4])) if len(a) > 4 else (1280, 720)
probe = _lib.Context(64, 64, 1) if mode == "probe_first" else None
ds = SyntheticDataset(W, H, 2, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001), lanes=lanes)
for i in range(8):
    ds._pair(i); ds.get_gt_of(i)
ds._bgr = {0: np.zeros((H, W, 3), np.uint8)}
ds.get_frame = lambda: ds._bgr[0]                          # (the BGR frame is not part of the path; one array for all)
p = Processor(RunConfig(logging.getLogger("probe"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
for rep in range(3):
    ds.N = 2001
    p.frame_index = 0; p.detection_results = {}; p.config.results = {}; p.detection_boxes = {}
    t0 = time.perf_counter(); p.run_detection(); dt = time.perf_counter() - t0
    print(f"{W}x{H} {mode} lanes {lanes or pipeline.auto_lanes(W, H, 1)} priority {pipeline.LANE_STREAM_PRIORITY} rep {rep}: {1e3 * dt / 2000:.4f} ms per frame", flush=True)
p.release()
