"""cProfile of one Processor loop at 1080p (where do the milliseconds of an iteration go on the host?).
    python tools/profile_loop.py [run_detection|run_detection_staged|run_detection_batched] [farneback|host]"""
import cProfile, logging, pstats, sys
sys.path.insert(0, "mav-detection_amd")
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig

loop = sys.argv[1] if len(sys.argv) > 1 else "run_detection"
use_fb = (sys.argv[2] if len(sys.argv) > 2 else "host") == "farneback"
W, H, N = 1920, 1080, 8
ds = SyntheticDataset(W, H, N, use_farneback=use_fb)
for i in range(N):
    ds._pair(i)
mk = lambda: Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
getattr(mk(), loop)()                  # warm-up
p = mk()
pr = cProfile.Profile()
pr.enable()
getattr(p, loop)()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
