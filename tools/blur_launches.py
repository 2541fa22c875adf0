#!/usr/bin/env python3
"""Per-launch durations of the layer-image kernels (blur3 / fused / multi / two-pass h and v) and the expansions, grouped by grid
shape, from a rocprofv3 kernel trace:   python tools/blur_launches.py <..._kernel_trace.csv>"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(k in n for k in ("blur_resize", "blur_multi", "blur3", "polyexp")):
        d[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    print(f"{k[0]:24s} grid {k[1]:>9s} x {k[2]:>4s} x {k[3]:>3s}  launches {len(v):4d}  avg {sum(v) / len(v):8.1f} us  min {min(v):8.1f} us")
