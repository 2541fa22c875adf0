"""The reference-shaped loops at 1080p: pairs/s of Processor.run_detection_batched(batch) and ms per frame of run_detection /
run_detection_staged on a pre-generated SyntheticDataset (host frames in, filled FrameResults out; frame synthesis excluded).
bench.py prints the same figures as `api_loop`.  Run on the GPU box:  python tools/api_loop_probe.py [batch] [batches]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "mav-detection_amd")
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
out = bench.api_loop_leg(batch=batch, n_batches=batches, staged=True)
for k, v in out.items():
    print(f"{k:40s} {v}")
