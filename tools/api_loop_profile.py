"""cProfile of Processor.run_detection (one frame per iteration) at 1080p: where the host's share of a frame goes."""
import sys, logging, cProfile, pstats
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig
W, H, N = (int(sys.argv[1]), int(sys.argv[2]), 201) if len(sys.argv) > 2 else (1920, 1080, 201)
ds = SyntheticDataset(W, H, N, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001))
for i in range(8):
    ds._pair(i); ds.get_gt_of(i)
for _ in range(N):
    ds.get_frame()
p = Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
ds.N = 20
p.run_detection()
p.frame_index = 0; p.detection_results = {}
ds.N = N
pr = cProfile.Profile()
pr.enable()
p.run_detection()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
