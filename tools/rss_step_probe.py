"""What is the one-time 188 MB step in the resident set of a long one-frame loop (tools/api_loop_soak.py)?  Runs Processor.run_detection in
chunks of 4 000 frames and, whenever VmRSS has grown by more than 32 MB over a chunk, prints the memory mappings that are new or grew
(/proc/self/maps + smaps Rss): anonymous heap, or device-visible host memory mapped through /dev/kfd / /dev/dri?
    python tools/rss_step_probe.py [frames]"""
import sys, logging, re
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig

F = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
W, H = 1280, 720


def smaps():
    out, cur = {}, None
    for line in open("/proc/self/smaps"):
        m = re.match(r"([0-9a-f]+)-([0-9a-f]+) (\S+) \S+ \S+ \S+\s*(.*)", line)
        if m:
            cur = (int(m.group(1), 16), m.group(4).strip() or "[anon]", m.group(3))
            out[cur] = [int(m.group(2), 16) - int(m.group(1), 16), 0]
        elif line.startswith("Rss:") and cur is not None:
            out[cur][1] = int(line.split()[1]) * 1024
    return out


def rss():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"):
            return int(line.split()[1]) / 1024.0


ds = SyntheticDataset(W, H, 2, use_farneback=True, distinct=8, dangle=(0.004, -0.002, 0.001))
for i in range(8):
    ds._pair(i); ds.get_gt_of(i)
ds._bgr = {0: np.zeros((H, W, 3), np.uint8)}
ds.get_frame = lambda: ds._bgr[0]
p = Processor(RunConfig(logging.getLogger("probe"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
chunk, done = 4000, 0
prev_maps, prev_rss = None, None
while done < F:
    ds.N = chunk + 1
    p.frame_index = 0; p.detection_results = {}; p.config.results = {}; p.detection_boxes = {}
    p.run_detection()
    done += chunk
    now_maps, now_rss = smaps(), rss()
    if prev_rss is not None and now_rss - prev_rss > 32:
        print(f"after {done} frames: VmRSS {prev_rss:.1f} -> {now_rss:.1f} MB; mappings whose resident part grew by more than 4 MB:", flush=True)
        for key, (size, r) in sorted(now_maps.items(), key=lambda kv: -kv[1][1]):
            old = prev_maps.get(key, [0, 0])[1]
            if r - old > (4 << 20):
                print(f"   {key[0]:016x}  size {size / 2**20:8.1f} MB  resident {old / 2**20:8.1f} -> {r / 2**20:8.1f} MB  {key[2]}  {key[1]}", flush=True)
    prev_maps, prev_rss = now_maps, now_rss
print(f"{done} frames, VmRSS {prev_rss:.1f} MB")
p.release()
