"""Where a batch of Processor.run_detection_batched goes on the host: wall time inside submit() (since round 6: assembling ONE
mav_frame_step and posting it to the context's worker thread, which gathers the frames and enqueues the batch), inside collect()
(waiting for the batch's step and marker) and in the FrameResult tail, per batch.
usage: python tools/api_loop_breakdown.py [batch] [batches] [upload_threads] [distinct pairs] [video]"""
import sys, time, logging
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import numpy as np
from mavflow import pipeline
from mavflow.processor import Processor, SyntheticDataset
from mavflow.run_config import RunConfig

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 8
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
distinct = int(sys.argv[4]) if len(sys.argv) > 4 else 8          # distinct pairs the dataset cycles through (8 x 4 MB of frames stay in the host's L3)
video = len(sys.argv) > 5 and sys.argv[5] == "video"             # pair i = (frame i, frame i + 1) of ONE sequence: frames shared between pairs
W, H = 1920, 1080
N = batch * batches + 1
ds = SyntheticDataset(W, H, N, use_farneback=True, distinct=distinct, dangle=(0.004, -0.002, 0.001), video=video)
for i in range(distinct):
    ds._pair(i); ds.get_gt_of(i)
p = Processor(RunConfig(logging.getLogger("t"), ds, "", False, False, False, True, False, False, "FLOW_FOE_CLUSTERING"))
if threads:
    [c.set_option("upload_threads", threads) for c in p._own_ctxs(batch, 1)]
acc = {"submit": [], "collect": [], "total": []}
orig_submit, orig_collect = pipeline.DetectPipeline.submit, pipeline.DetectPipeline.collect


def submit(self, *a, **k):
    t0 = time.perf_counter(); r = orig_submit(self, *a, **k); acc["submit"].append(time.perf_counter() - t0); return r


def collect(self, *a, **k):
    t0 = time.perf_counter(); r = orig_collect(self, *a, **k); acc["collect"].append(time.perf_counter() - t0); return r


pipeline.DetectPipeline.submit, pipeline.DetectPipeline.collect = submit, collect
# ... and the library calls inside submit()
from mavflow import _lib
lib = _lib.load()
for name in ("mav_frame_step_post", "mav_frame_step_wait", "mav_worker_drain", "mav_upload_gather", "mav_marker_wait"):
    acc[name] = []

    def wrap(fn, name=name):
        def w(*a):
            t0 = time.perf_counter(); r = fn(*a); acc[name].append(time.perf_counter() - t0); return r
        return w
    setattr(lib, name, wrap(getattr(lib, name)))
ds.N = 3 * batch + 1                                            # (three slots: all of them get their buffers in the warm run)
p.run_detection_batched(batch=batch)
p.frame_index = 0; p.detection_results = {}
p.flow_uv = p.estimate_fixed = p.total_mask = None          # a fresh run holds no handles of an earlier one
for k in acc:
    acc[k].clear()
ds.N = N
t0 = time.perf_counter()
p.run_detection_batched(batch=batch)
dt = time.perf_counter() - t0
ms = lambda v: " ".join(f"{1e3 * x:6.2f}" for x in v)
print(f"{W}x{H} batch {batch} x {batches}, upload_threads {threads or 'default'}, {distinct} distinct pairs{' of one video' if video else ''}: {(N - 1) / dt:.1f} pairs/s, {1e3 * dt / batches:.2f} ms per batch")
print(f"  submit  per batch (ms): {ms(acc['submit'])}")
print(f"  collect per batch (ms): {ms(acc['collect'])}")
for name in ("mav_frame_step_post", "mav_frame_step_wait", "mav_worker_drain", "mav_upload_gather", "mav_marker_wait"):
    v = acc[name]
    print(f"  {name:28s} {len(v):4d} calls, {1e3 * sum(v) / batches:7.2f} ms per batch; per call (ms): {ms(v[:12])}")
other = 1e3 * dt - 1e3 * sum(acc['submit']) - 1e3 * sum(acc['collect'])
print(f"  everything else (frame_pair, samples, FrameResult tail): {other / batches:.2f} ms per batch")
p.release()
