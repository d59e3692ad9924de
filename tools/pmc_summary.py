#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, per counter, the mean value per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
for f in glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "vrt::" not in k:
            continue
        k = k.replace("void vrt::", "").replace("(vrt::FrameParams)", "")
        acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
for k in sorted(acc):
    print(f"== {k}")
    for c in sorted(acc[k]):
        v = list(acc[k][c].values())
        print(f"  {c:36s} dispatches={len(v):3d} mean/dispatch={sum(v)/len(v):.6g}")
