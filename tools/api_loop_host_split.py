"""Where the one-frame loop's wall time goes on its own thread: waiting for results (wait_step) vs everything else (sample draws, step
assembly, post, FrameResult tail).  python tools/api_loop_host_split.py [W H frames]"""
import sys, logging, time
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import _lib
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig
a = sys.argv[1:]
W, H, F = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1280, 720, 384)
acc = {"wait": 0.0, "post": 0.0}
_w, _p = _lib.Context.wait_step, _lib.Context.post_step


def wait_step(self, ticket, marker=None):
    t = time.perf_counter(); _w(self, ticket, marker); acc["wait"] += time.perf_counter() - t


def post_step(self, step):
    t = time.perf_counter(); r = _p(self, step); acc["post"] += time.perf_counter() - t
    return r


_lib.Context.wait_step, _lib.Context.post_step = wait_step, post_step
for lanes in (1, 2, 3):
    ds = SyntheticDataset(W, H, F + 1, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001), lanes=lanes)
    for i in range(8):
        ds._pair(i); ds.get_gt_of(i)
    for _ in range(F + 1):
        ds.get_frame()
    p = Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
    p.run_detection()
    for rep in range(2):
        p.frame_index = 0; p.detection_results = {}; p.config.results = {}
        acc["wait"] = acc["post"] = 0.0
        np.random.seed(7)
        t0 = time.perf_counter(); p.run_detection(); dt = time.perf_counter() - t0
        print(f"{W}x{H} lanes {lanes}: {1e3 * dt / F:.4f} ms per frame = waiting for results {1e3 * acc['wait'] / F:.4f} + posting {1e3 * acc['post'] / F:.4f} "
              f"+ the loop's own Python {1e3 * (dt - acc['wait'] - acc['post']) / F:.4f}", flush=True)
    p.release()
