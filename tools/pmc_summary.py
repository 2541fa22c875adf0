#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, sum of every counter over all dispatches (+ dispatch count)."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(set)
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:36]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add((f, r["Dispatch_Id"]))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    if k.startswith("__amd") or agg[k].get("SQ_WAVE_CYCLES", 0) < 1e6:
        continue
    print(f"== {k}  (dispatches per pass: {len(ndisp[k]) // max(1, len(glob.glob(sys.argv[1] + '/p*/')))})")
    for a, b in sorted(agg[k].items()):
        print(f"   {a:40s} {b:.5g}")
