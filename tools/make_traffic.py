#!/usr/bin/env python3
"""profiles/<round>/traffic*.json from a tools/pmc_passes.sh summary: HBM-side bytes of the sweep kernels in ONE bench step.

    python tools/make_traffic.py <pmc_dir> <out.json> --width W --height H --batch B --levels L

FETCH_SIZE / WRITE_SIZE are reported in KB and come from separate rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 0`
(one step = one process_batch_dev call).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-byte
read requests of a wide coalesced stream at 64 bytes, so it is doubled; WRITE_SIZE is exact.  The record carries the hash of the
kernel sources it was measured on; bench.py reports `traffic` only when hash, shape, batch and launch count all match.
"""
import argparse, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("pmc_dir"); ap.add_argument("out")
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--levels", type=int, default=1)
a = ap.parse_args()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mav-detection_amd"))
from mavflow import _lib
with _lib.Context(a.width, a.height, a.batch, _lib.fb_defaults(levels=a.levels)) as _c:        # (runs on the GPU box, like the passes)
    schedule = _c.schedule_info(a.batch)
kern, cur = {}, None
for line in open(os.path.join(a.pmc_dir, "summary.txt")):
    m = re.match(r"== (\S.*?)\s+\(dispatches per pass: (\d+)\)", line)
    if m:
        cur = kern.setdefault(m.group(1), {"dispatches": int(m.group(2))})
        continue
    m = re.match(r"\s+(\w+)\s+([0-9.e+]+)", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
sweeps = {k: v for k, v in kern.items() if k.startswith("k_blur_iter_fast")}
fetch = sum(v.get("FETCH_SIZE", 0.0) for v in sweeps.values())
write = sum(v.get("WRITE_SIZE", 0.0) for v in sweeps.values())
rdreq = sum(v.get("TCC_EA0_RDREQ_sum", 0.0) for v in sweeps.values())
launches = sum(v["dispatches"] for v in sweeps.values())
rec = {"kernel": sorted(sweeps), "source": f"{a.pmc_dir}/summary.txt (rocprofv3 --pmc FETCH_SIZE ; --pmc WRITE_SIZE GRBM_GUI_ACTIVE ; separate passes of bench.py --steps 1 --warmup 0)",
       "source_hash": bench.source_hash(schedule), "schedule": schedule, "width": a.width, "height": a.height, "batch": a.batch, "levels": a.levels, "launches": launches,
       "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "TCC_EA0_RDREQ_sum": rdreq, "fetch_correction": 2.0,
       "hbm_bytes_sweeps_per_step": (2.0 * fetch + write) * 1024.0,
       "note": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while the requests of a wide stream are 128 B: doubled (MI355X_MICROARCH.md). "
               "Memory-side request bytes: Infinity-Cache hits are included, so this is fabric traffic, an upper bound on DRAM traffic."}
json.dump(rec, open(a.out, "w"), indent=1)
print(json.dumps(rec, indent=1))
