import sys, logging, time
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig
mode = sys.argv[1]
if len(sys.argv) > 2:
    from mavflow import pipeline
    pipeline.LANE_STREAM_PRIORITY = int(sys.argv[2])
W, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1280, 720)
LANES = int(sys.argv[3]) if len(sys.argv) > 3 else None
probe = _lib.Context(64, 64, 1) if mode in ("probe_first", "probe_first_getframe") else None
ds = SyntheticDataset(W, H, 2, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001), lanes=LANES)
for i in range(8):
    ds._pair(i); ds.get_gt_of(i)
if mode in ("getframe", "probe_first_getframe"):
    ds._bgr = {0: np.zeros((H, W, 3), np.uint8)}
    ds.get_frame = lambda: ds._bgr[0]
else:
    for _ in range(8):
        ds.get_frame()
    ds._frame_cursor = 0
p = Processor(RunConfig(logging.getLogger("soak"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
for rep in range(3):
    ds.N = 2001
    p.frame_index = 0; p.detection_results = {}; p.config.results = {}; p.detection_boxes = {}
    t0 = time.perf_counter(); p.run_detection(); dt = time.perf_counter() - t0
    print(mode, rep, f"{1e3 * dt / 2000:.4f} ms per frame", flush=True)
p.release()
